// Packing of embedding rows into the tile-contiguous bank layout (include/imagescry_hip.h: isc_bank_pack,
// isc_bank_unpack, isc_bank_packed_bytes).
#include "bank_layout.h"
#include "isc_common.h"

namespace {

// One wave per row: optional L2 normalisation (float32, the F.normalize formula), cast, scatter the row's
// 16-byte chunks to their K-step blocks; columns past D are zero.
template <typename TIN, typename TOUT>
__global__ __launch_bounds__(256) void k_bank_pack(const TIN* __restrict__ x, int64_t n_rows, int d, int64_t ldx,
                                                   int64_t first_row, int normalize, float eps,
                                                   unsigned char* __restrict__ packed, int ks) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const TIN* p = x + r * ldx;
    float inv = 1.f;
    float denom = 1.f;
    if (normalize) {
        float acc = 0.f;
        for (int i = lane; i < d; i += 64) {
            const float v = (float)p[i];
            acc += v * v;
        }
        denom = fmaxf(sqrtf(isc_wave_sum(acc)), eps);
    }
    (void)inv;
    constexpr int PER_CHUNK = 16 / (int)sizeof(TOUT);
    const int chunks = ks * 8;
    const int64_t row = first_row + r;
    for (int c = lane; c < chunks; c += 64) {
        TOUT v[PER_CHUNK];
#pragma unroll
        for (int j = 0; j < PER_CHUNK; ++j) {
            const int e = c * PER_CHUNK + j;
            float f = e < d ? (float)p[e] : 0.f;
            if (normalize && e < d) f = __fdiv_rn(f, denom);
            v[j] = (TOUT)f;
        }
        unsigned char* dst = packed + isc_packed_offset(row, c >> 3, ks) + (c & 7) * 16;
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_bank_unpack(const unsigned char* __restrict__ packed, int d, int ks,
                                                     int64_t first_row, int64_t n_rows, T* __restrict__ y,
                                                     int64_t ldy) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int lane = threadIdx.x & 63;
    for (int e = lane; e < d; e += 64) y[r * ldy + e] = isc_packed_load<T>(packed, first_row + r, e, ks);
}

int check_dtype(int dtype) { return dtype == ISC_F16 || dtype == ISC_F32; }

}  // namespace

extern "C" int isc_bank_packed_bytes(int dtype, int64_t N, int D, size_t* bytes) {
    ISC_REQUIRE(bytes && check_dtype(dtype) && N > 0 && D > 0);
    const int esz = dtype == ISC_F16 ? 2 : 4;
    const int64_t tiles = isc_ceil_div<int64_t>(N, ISC_TILE_ROWS);
    *bytes = (size_t)tiles * isc_ksteps(D, esz) * ISC_TILE_KSTEP_BYTES;
    return ISC_OK;
}

extern "C" int isc_bank_pack(const void* rows, int in_dtype, int64_t n_rows, int D, int64_t ldx, int64_t first_row,
                             int normalize, float eps, void* packed, int dtype, void* stream) {
    ISC_REQUIRE(rows && packed && check_dtype(in_dtype) && check_dtype(dtype));
    ISC_REQUIRE(n_rows > 0 && D > 0 && ldx >= D && first_row >= 0);
    if (!isc_aligned(packed, 16)) return ISC_ERR_ALIGNMENT;
    const int64_t blocks = isc_ceil_div<int64_t>(n_rows, 4);
    if (blocks > 0x7fffffff) return ISC_ERR_UNSUPPORTED;
    const int ks = isc_ksteps(D, dtype == ISC_F16 ? 2 : 4);
    unsigned char* out = static_cast<unsigned char*>(packed);
    hipStream_t s = isc_stream(stream);
    const dim3 grid((unsigned)blocks), block(256);
#define ISC_PACK(TIN, TOUT)                                                                                            \
    hipLaunchKernelGGL((k_bank_pack<TIN, TOUT>), grid, block, 0, s, static_cast<const TIN*>(rows), n_rows, D, ldx,    \
                       first_row, normalize, eps, out, ks)
    if (in_dtype == ISC_F32 && dtype == ISC_F16) ISC_PACK(float, _Float16);
    else if (in_dtype == ISC_F32 && dtype == ISC_F32) ISC_PACK(float, float);
    else if (in_dtype == ISC_F16 && dtype == ISC_F16) ISC_PACK(_Float16, _Float16);
    else ISC_PACK(_Float16, float);
#undef ISC_PACK
    return isc_launch_status();
}

extern "C" int isc_bank_unpack(const void* packed, int dtype, int D, int64_t first_row, int64_t n_rows, void* rows,
                               int64_t ldy, void* stream) {
    ISC_REQUIRE(packed && rows && check_dtype(dtype) && D > 0 && n_rows > 0 && first_row >= 0 && ldy >= D);
    const int64_t blocks = isc_ceil_div<int64_t>(n_rows, 4);
    if (blocks > 0x7fffffff) return ISC_ERR_UNSUPPORTED;
    const unsigned char* in = static_cast<const unsigned char*>(packed);
    if (dtype == ISC_F16)
        hipLaunchKernelGGL(k_bank_unpack<_Float16>, dim3((unsigned)blocks), dim3(256), 0, isc_stream(stream), in, D,
                           isc_ksteps(D, 2), first_row, n_rows, static_cast<_Float16*>(rows), ldy);
    else
        hipLaunchKernelGGL(k_bank_unpack<float>, dim3((unsigned)blocks), dim3(256), 0, isc_stream(stream), in, D,
                           isc_ksteps(D, 4), first_row, n_rows, static_cast<float*>(rows), ldy);
    return isc_launch_status();
}
