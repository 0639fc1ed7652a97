// Packing of embedding rows into the tile-contiguous, row-permuted bank layout (include/imagescry_hip.h:
// isc_bank_pack, isc_bank_unpack, isc_bank_packed_bytes, isc_bank_permutation).
#include "bank_layout.h"
#include "isc_common.h"

namespace {

// One wave per row: optional L2 normalisation (float32, the F.normalize formula), cast, scatter the row's
// 16-byte chunks to their K-step blocks at the row's PERMUTED position; columns past D are zero.  The norm of the
// row AS STORED (after the cast) feeds `norm_bound` (atomic max): the search's rounding-error guard needs an upper
// bound of the stored rows' norms.
template <typename TIN, typename TOUT>
__global__ __launch_bounds__(256) void k_bank_pack(const TIN* __restrict__ x, int64_t n_rows, int d, int64_t ldx,
                                                   int64_t first_row, IscPerm pm, int normalize, float eps,
                                                   unsigned char* __restrict__ packed, int ks,
                                                   unsigned* __restrict__ norm_bound) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const TIN* p = x + r * ldx;
    float denom = 1.f;
    if (normalize) {  // (element order of the sum: lane, lane + 64, ... -- kept as it was: the stored values depend on it)
        float acc = 0.f;
        for (int i = lane; i < d; i += 64) {
            const float v = (float)p[i];
            acc += v * v;
        }
        denom = fmaxf(sqrtf(isc_wave_sum(acc)), eps);
    }
    constexpr int PER_CHUNK = 16 / (int)sizeof(TOUT);
    constexpr int IN_VECS = PER_CHUNK * (int)sizeof(TIN) / 16;  // 16-byte input loads per output chunk: 1, 2, or 0 (f16 -> f32)
    const int chunks = ks * 8;
    const int64_t row = isc_perm_pos(pm, first_row + r);
    // rows whose chunks can be fetched with 16-byte loads (a wave instruction then reads 1 KiB of the row instead of 64
    // scattered 2- or 4-byte elements)
    const bool vec_ok = IN_VECS > 0 && ((reinterpret_cast<uintptr_t>(p) & 15) == 0);
    float stored_sq = 0.f;
    for (int c = lane; c < chunks; c += 64) {
        TOUT v[PER_CHUNK];
        float f[PER_CHUNK];
        const int e0 = c * PER_CHUNK;
        if (vec_ok && e0 + PER_CHUNK <= d) {
            TIN raw[PER_CHUNK];
#pragma unroll
            for (int u = 0; u < (IN_VECS > 0 ? IN_VECS : 1); ++u)
                reinterpret_cast<uint4*>(raw)[u] = reinterpret_cast<const uint4*>(p + e0)[u];
#pragma unroll
            for (int j = 0; j < PER_CHUNK; ++j) f[j] = (float)raw[j];
        } else {
#pragma unroll
            for (int j = 0; j < PER_CHUNK; ++j) f[j] = e0 + j < d ? (float)p[e0 + j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < PER_CHUNK; ++j) {
            if (normalize && e0 + j < d) f[j] = __fdiv_rn(f[j], denom);
            v[j] = (TOUT)f[j];
            const float s = (float)v[j];
            stored_sq = fmaf(s, s, stored_sq);
        }
        unsigned char* dst = packed + isc_packed_offset(row, c >> 3, ks) + (c & 7) * 16;
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(v);
    }
    if (norm_bound) {
        stored_sq = isc_wave_sum(stored_sq);
        // float32 summation of d squares: relative error <= d * 2^-24 -- cover it (and the sqrt) with a factor
        const float nb = sqrtf(stored_sq) * (1.f + 1e-3f);
        // non-negative floats order as uints; a row holding NaN makes the bound +inf: the search then trusts no filter
        // result on this bank and answers through its exhaustive pass.  The atomic goes out only when this row RAISES the
        // bound as this wave sees it (a plain read first): one atomic per row on the one word was the whole kernel --
        // 88 atomics per microsecond on one address, 11.4 ms per 2^20 rows whatever their size.
        if (lane == 0) {
            const unsigned mine = nb == nb ? __float_as_uint(nb) : 0x7f800000u;
            if (mine > __hip_atomic_load(norm_bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(norm_bound, mine);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_bank_unpack(const unsigned char* __restrict__ packed, int d, int ks,
                                                     IscPerm pm, int64_t first_row, int64_t n_rows,
                                                     T* __restrict__ y, int64_t ldy) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const int64_t row = isc_perm_pos(pm, first_row + r);
    for (int e = lane; e < d; e += 64) y[r * ldy + e] = isc_packed_load<T>(packed, row, e, ks);
}

int check_dtype(int dtype) { return dtype == ISC_F16 || dtype == ISC_F32; }

int64_t gcd64(int64_t a, int64_t b) {
    while (b) {
        const int64_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}

// x with (a * x) mod n == 1 (extended Euclid; a and n coprime, n >= 2)
int64_t mod_inverse(int64_t a, int64_t n) {
    int64_t t = 0, nt = 1, r = n, nr = a % n;
    while (nr) {
        const int64_t q = r / nr;
        int64_t tmp = t - q * nt;
        t = nt;
        nt = tmp;
        tmp = r - q * nr;
        r = nr;
        nr = tmp;
    }
    return t < 0 ? t + n : t;
}

}  // namespace

IscPerm isc_make_perm(int64_t n) {
    IscPerm pm;
    pm.n = n;
    if (n <= 2) {  // identity
        pm.mul = pm.mul_inv = 1;
        return pm;
    }
    int64_t m = (int64_t)((double)n * 0.6180339887498949);
    if (m < 1) m = 1;
    while (gcd64(m, n) != 1) ++m;  // terminates: n - 1 is coprime to n
    pm.mul = m % n;
    pm.mul_inv = mod_inverse(pm.mul, n);
    return pm;
}

extern "C" int isc_bank_permutation(int64_t N, int64_t* mul, int64_t* mul_inv) {
    ISC_REQUIRE(N > 0 && N <= 0x7fffffff && mul && mul_inv);
    const IscPerm pm = isc_make_perm(N);
    *mul = pm.mul;
    *mul_inv = pm.mul_inv;
    return ISC_OK;
}

extern "C" int isc_bank_packed_bytes(int dtype, int64_t N, int D, size_t* bytes) {
    ISC_REQUIRE(bytes && check_dtype(dtype) && N > 0 && D > 0);
    const int esz = dtype == ISC_F16 ? 2 : 4;
    const int64_t tiles = isc_ceil_div<int64_t>(N, ISC_TILE_ROWS);
    *bytes = (size_t)tiles * isc_ksteps(D, esz) * ISC_TILE_KSTEP_BYTES;
    return ISC_OK;
}

extern "C" int isc_bank_pack(const void* rows, int in_dtype, int64_t n_rows, int D, int64_t ldx, int64_t first_row,
                             int64_t n_total, int normalize, float eps, void* packed, int dtype, float* norm_bound,
                             void* stream) {
    ISC_REQUIRE(rows && packed && check_dtype(in_dtype) && check_dtype(dtype));
    ISC_REQUIRE(n_rows > 0 && D > 0 && ldx >= D && first_row >= 0);
    ISC_REQUIRE(n_total >= first_row + n_rows && n_total <= 0x7fffffff);
    if (!isc_aligned(packed, 16)) return ISC_ERR_ALIGNMENT;
    const int64_t blocks = isc_ceil_div<int64_t>(n_rows, 4);
    if (blocks > 0x7fffffff) return ISC_ERR_UNSUPPORTED;
    const int ks = isc_ksteps(D, dtype == ISC_F16 ? 2 : 4);
    unsigned char* out = static_cast<unsigned char*>(packed);
    hipStream_t s = isc_stream(stream);
    const IscPerm pm = isc_make_perm(n_total);
    const dim3 grid((unsigned)blocks), block(256);
#define ISC_PACK(TIN, TOUT)                                                                                         \
    hipLaunchKernelGGL((k_bank_pack<TIN, TOUT>), grid, block, 0, s, static_cast<const TIN*>(rows), n_rows, D, ldx, \
                       first_row, pm, normalize, eps, out, ks, reinterpret_cast<unsigned*>(norm_bound))
    if (in_dtype == ISC_F32 && dtype == ISC_F16) ISC_PACK(float, _Float16);
    else if (in_dtype == ISC_F32 && dtype == ISC_F32) ISC_PACK(float, float);
    else if (in_dtype == ISC_F16 && dtype == ISC_F16) ISC_PACK(_Float16, _Float16);
    else ISC_PACK(_Float16, float);
#undef ISC_PACK
    return isc_launch_status();
}

extern "C" int isc_bank_unpack(const void* packed, int dtype, int D, int64_t n_total, int64_t first_row,
                               int64_t n_rows, void* rows, int64_t ldy, void* stream) {
    ISC_REQUIRE(packed && rows && check_dtype(dtype) && D > 0 && n_rows > 0 && first_row >= 0 && ldy >= D);
    ISC_REQUIRE(n_total >= first_row + n_rows && n_total <= 0x7fffffff);
    const int64_t blocks = isc_ceil_div<int64_t>(n_rows, 4);
    if (blocks > 0x7fffffff) return ISC_ERR_UNSUPPORTED;
    const unsigned char* in = static_cast<const unsigned char*>(packed);
    const IscPerm pm = isc_make_perm(n_total);
    if (dtype == ISC_F16)
        hipLaunchKernelGGL(k_bank_unpack<_Float16>, dim3((unsigned)blocks), dim3(256), 0, isc_stream(stream), in, D,
                           isc_ksteps(D, 2), pm, first_row, n_rows, static_cast<_Float16*>(rows), ldy);
    else
        hipLaunchKernelGGL(k_bank_unpack<float>, dim3((unsigned)blocks), dim3(256), 0, isc_stream(stream), in, D,
                           isc_ksteps(D, 4), pm, first_row, n_rows, static_cast<float*>(rows), ldy);
    return isc_launch_status();
}
