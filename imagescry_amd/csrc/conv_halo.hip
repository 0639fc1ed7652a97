// Small-channel R x R convolutions (the RGB stems, EfficientNetV2's 24-channel first stage) as an implicit GEMM whose
// pixel operand is gathered from an input HALO TILE in LDS instead of from the L2 (dispatched by conv_launch, encoder.hip).
//
// k_conv_f32's packed-K mode serves these layers with one 16-byte LDS-DMA piece per (pixel, filter tap, four channels): for
// a 3 x 3 layer on 24 channels that is 54 scattered 16-byte L2 requests per output pixel -- 5.7 GB of 16-byte gathers for
// EfficientNetV2-S's `s1` blocks at batch 512, the 7 x 7 RGB stem of ResNet-50 the same -- and the layers ran at 0.13 - 0.55 of
// the f32 matrix rate, bound by that request stream (profiles/r03_*_layers.txt).  Here a workgroup owns a 16 x 16 tile of
// output pixels of one image: the (15 s + R)^2 input pixels it needs are brought into LDS ONCE, by LDS-DMA in whole
// contiguous pixel rows (out-of-image pixels read a line of zeros), the layer's folded weights sit in LDS for the whole
// kernel, and the K loop reads both MFMA operands from LDS:
//
//     weights  wl[quad][row][g]   16-byte chunks: chunk (4 quad + g) of weight row `row` -- a lane's A fragment of one
//                                 quad is one ds_read_b128, a wave's 64 reads one contiguous KiB;
//     pixels   halo[hy][hx][c4]   the input tile as it lies in memory; lane (pixel, g) reads chunk (tap, c4) = 4 quad + g
//                                 of its pixel's window.
//
// As in k_conv_f32, element j of every lane's 16-byte chunk feeds the j-th v_mfma_f32_16x16x4_f32 of the quad: a
// permutation of the reduction axis applied to both operands alike, so the sum is the same k-ordered float32 fma chain up
// to the order of its terms (the parity tests hold it to 1e-5 of the output scale, as the other convolutions).
// Weight rows are the packed-K rows of isc_conv2d_nhwc ([Cout][ceil(R*S*Cin/32)*32], k = (r*S + s)*Cin + c, zero tail).
// Two workgroups of four waves per CU (<= 78 KiB of LDS each): one stages its next halo while the other computes.
#include <type_traits>

#include "isc_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int HT = 16;  // output tile: HT x HT pixels; four waves, each four rows of 16 pixels
constexpr int HNI = 4;  // 16-pixel blocks (= tile rows) per wave

struct HaloParams {
    const float* x;
    const float* w;
    const float* bias;
    const float* res;
    float* out;
    int B, H, W, Cin, Cout, R, S, stride, pad, Ho, Wo;
    int cin4;     // Cin / 4: 16-byte chunks per pixel
    int nchunks;  // R * S * cin4: chunks of the reduction axis
    int kq;       // quads of four chunks
    int wld;      // floats per weight row in memory (K padded to a multiple of 32)
    int hr, hc;   // halo rows / columns in pixels: (HT - 1) * stride + R (S)
    int halo_chunks;         // hr * hc * cin4
    int tiles_x, tiles_img;  // tiles per image row / per image
    int ntiles;
    int act, res_after_act;
    int wl_bytes;  // LDS bytes of the weight image (the halo image follows)
    unsigned div_row_mul, div_row_sh;  // n / (hc * cin4)
    unsigned div_c4_mul, div_c4_sh;    // n / cin4
    unsigned div_img_mul, div_img_sh;  // n / tiles_img
    unsigned div_tx_mul, div_tx_sh;    // n / tiles_x
};

__device__ __forceinline__ int halo_div(int n, unsigned mul, unsigned sh) {
    return mul ? (int)(__umulhi((unsigned)n, mul) >> sh) : n;
}
void halo_fastdiv(unsigned d, unsigned* mul, unsigned* sh) {  // floor(n / d) = (n * mul) >> (32 + sh) for n < 2^31
    if (d <= 1) { *mul = 0; *sh = 0; return; }
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    const unsigned p = 31 + l;
    *mul = (unsigned)(((1ull << p) + d - 1) / d);
    *sh = p - 32;
}

__device__ __attribute__((aligned(128))) const float g_halo_zero[32] = {0.f};

__device__ __forceinline__ float halo_act(float v, int act) {
    if (act == ISC_ACT_RELU) return fmaxf(v, 0.f);
    if (act == ISC_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    if (act == ISC_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));  // as k_conv_f32 (apply_act)
    if (act == ISC_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.f + __expf(-v));
    return v;
}

// MI = 16-channel blocks of the output (2: Cout <= 32, 4: Cout <= 64)
template <int MI>
__global__ __launch_bounds__(256, 2) void k_conv_halo_f32(const HaloParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const wl = lds;                 // [kq][MI * 16][4] chunks
    unsigned char* const halo = lds + p.wl_bytes;  // [hr][hc][cin4] chunks (+ slack up to a whole staging round)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int prow = lane & 15;  // MFMA row (A: output channel of a block) / column (B: pixel of a block)
    const int g = lane >> 4;     // which of a quad's four chunks this lane supplies

    // ---- the weights, once: chunk (t, row, c) <- w[row][(4 t + c) * 4 .. + 3], zero past Cout and past the row's end
    {
        const int rows = MI * 16;
        const int total = p.kq * rows * 4;
        const int wchunks = p.wld >> 2;
        for (int i = tid; i < total; i += 256) {
            const int c = i & 3;
            const int row = (i >> 2) % rows;
            const int t = (i >> 2) / rows;
            const int q = 4 * t + c;
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (row < p.Cout && q < wchunks) v = *reinterpret_cast<const u32x4*>(p.w + (size_t)row * p.wld + q * 4);
            *reinterpret_cast<u32x4*>(wl + (size_t)i * 16) = v;
        }
    }

    // this lane's bias (four consecutive channels per block: C layout row = 4 g + r)
    f32x4 bv[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int co = mi * 16 + g * 4;
        bv[mi] = (p.bias && co < p.Cout) ? *reinterpret_cast<const f32x4*>(p.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // offsets of this lane's four pixels (tile rows 4 wave .. 4 wave + 3, column prow) inside the halo image
    int pbase[HNI];
#pragma unroll
    for (int ni = 0; ni < HNI; ++ni) pbase[ni] = (((wave * HNI + ni) * p.stride) * p.hc + prow * p.stride) * p.cin4 * 16;
    const int rounds = (p.halo_chunks + 255) >> 8;

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int b = halo_div(tile, p.div_img_mul, p.div_img_sh);
        const int rem = tile - b * p.tiles_img;
        const int ty = halo_div(rem, p.div_tx_mul, p.div_tx_sh);
        const int tx = rem - ty * p.tiles_x;
        const int oy0 = ty * HT, ox0 = tx * HT;
        const int iy0 = oy0 * p.stride - p.pad, ix0 = ox0 * p.stride - p.pad;
        const float* xb = p.x + (size_t)b * p.H * p.W * p.Cin;

        __syncthreads();  // every wave is done with the previous tile's halo (first tile: the weights are written)
        // ---- halo by LDS-DMA: chunk i of the image = (hy, hx, c4); a wave's 64 chunks land in one contiguous KiB
        for (int r = 0; r < rounds; ++r) {
            const int i = (r << 8) + tid;
            const float* src = g_halo_zero + (lane & 7) * 4;
            if (i < p.halo_chunks) {
                const int hy = halo_div(i, p.div_row_mul, p.div_row_sh);
                const int rr = i - hy * p.hc * p.cin4;
                const int hx = halo_div(rr, p.div_c4_mul, p.div_c4_sh);
                const int c4 = rr - hx * p.cin4;
                const int iy = iy0 + hy, ix = ix0 + hx;
                if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                    src = xb + ((size_t)iy * p.W + ix) * p.Cin + c4 * 4;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(halo + (((r << 8) + wave * 64) << 4)),
                                             16, 0, 0);
        }
        // the residual tile is requested now, behind the halo pieces: it lands under the K loop (one wait for everything
        // below; requested inside the epilogue it was eight dependent memory round trips per tile)
        const int ox = ox0 + prow;
        f32x4 rv[MI][HNI];
        if (p.res) {
#pragma unroll
            for (int ni = 0; ni < HNI; ++ni) {
                const int oy = min(oy0 + wave * HNI + ni, p.Ho - 1);  // rows / columns past the image read its last ones
                const size_t pix = ((size_t)b * p.Ho + oy) * p.Wo + min(ox, p.Wo - 1);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    rv[mi][ni] = *reinterpret_cast<const f32x4*>(p.res + pix * p.Cout + min(mi * 16 + g * 4, p.Cout - 4));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        // ---- K loop over quads of four chunks; this lane's chunk of quad t is q = 4 t + g, kept as (tap row, tap column,
        // channel chunk) and advanced by four chunks per quad.  The fragments of quad t + 1 are read before the MFMAs of
        // quad t (two register sets), so no LDS latency stands in front of a quad's first MFMA.
        f32x4 acc[MI][HNI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < HNI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        int cr = 0, cs = 0, cc = g;
        while (cc >= p.cin4) {
            cc -= p.cin4;
            if (++cs == p.S) { cs = 0; ++cr; }
        }
        auto read_quad = [&](int t, u32x4 (&a)[MI], u32x4 (&bq)[HNI]) {
            // chunks past the end of the reduction axis have zero weights; they read the window's first chunk (finite data)
            const int qoff = (4 * t + g < p.nchunks) ? ((cr * p.hc + cs) * p.cin4 + cc) * 16 : 0;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                a[mi] = *reinterpret_cast<const u32x4*>(wl + ((size_t)((t * MI + mi) * 16 + prow) * 4 + g) * 16);
#pragma unroll
            for (int ni = 0; ni < HNI; ++ni) bq[ni] = *reinterpret_cast<const u32x4*>(halo + pbase[ni] + qoff);
            cc += 4;
            while (cc >= p.cin4) {
                cc -= p.cin4;
                if (++cs == p.S) { cs = 0; ++cr; }
            }
        };
        auto mfma_quad = [&](const u32x4 (&a)[MI], const u32x4 (&bq)[HNI]) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < HNI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[mi][j]), __uint_as_float(bq[ni][j]),
                                                                           acc[mi][ni], 0, 0, 0);
        };
        u32x4 a0[MI], b0[HNI], a1[MI], b1[HNI];
        read_quad(0, a0, b0);
        for (int t = 0; t < p.kq; t += 2) {
            if (t + 1 < p.kq) read_quad(t + 1, a1, b1);
            mfma_quad(a0, b0);
            if (t + 1 < p.kq) {
                if (t + 2 < p.kq) read_quad(t + 2, a0, b0);
                mfma_quad(a1, b1);
            }
        }

        // ---- epilogue: a lane holds channels co .. co + 3 of pixel (oy, ox): one 16-byte store.  The activation is chosen
        // ONCE, outside the element loops.
        auto epilogue = [&](auto act_tag) {
            constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
            for (int ni = 0; ni < HNI; ++ni) {
                const int oy = oy0 + wave * HNI + ni;
                const bool pix_ok = oy < p.Ho && ox < p.Wo;
                const size_t pix = ((size_t)b * p.Ho + oy) * p.Wo + ox;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int co = mi * 16 + g * 4;
                    f32x4 v = acc[mi][ni] + bv[mi];
                    if (p.res && !p.res_after_act) v += rv[mi][ni];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = halo_act(v[r], ACT < 0 ? p.act : ACT);
                    if (p.res && p.res_after_act) v += rv[mi][ni];
                    if (pix_ok && co < p.Cout) *reinterpret_cast<f32x4*>(p.out + pix * p.Cout + co) = v;
                }
            }
        };
        if (p.act == ISC_ACT_NONE) epilogue(std::integral_constant<int, ISC_ACT_NONE>{});
        else if (p.act == ISC_ACT_RELU) epilogue(std::integral_constant<int, ISC_ACT_RELU>{});
        else if (p.act == ISC_ACT_SILU) epilogue(std::integral_constant<int, ISC_ACT_SILU>{});
        else epilogue(std::integral_constant<int, -1>{});
    }
}

}  // namespace

// Whether conv_launch hands a layer to the halo kernel: packed-K layers (Cin % 32 != 0) with a square odd filter, "same"
// padding, stride 1 or 2, at most 64 output channels and an LDS image that lets two workgroups share a CU.
bool isc_conv_halo_applies(int Cin, int Cout, int R, int S, int stride, int pad, bool has_sub_or_scale, size_t* lds_bytes) {
    if (has_sub_or_scale || R != S || (R != 3 && R != 5 && R != 7) || pad != R / 2 || stride < 1 || stride > 2) return false;
    if (Cin % 4 != 0 || Cin % 32 == 0 || Cin > 24 || Cout > 64 || Cout % 4 != 0) return false;
    const int cin4 = Cin / 4;
    const int nchunks = R * S * cin4;
    const int kq = (nchunks + 3) / 4;
    const int mi = Cout <= 32 ? 2 : 4;
    const int hr = (HT - 1) * stride + R;
    const size_t wl = (size_t)kq * mi * 16 * 4 * 16;
    const size_t hl = ((size_t)hr * hr * cin4 + 255) / 256 * 256 * 16;
    if (wl + hl > 78 * 1024) return false;
    if (lds_bytes) *lds_bytes = wl + hl;
    return true;
}

int isc_conv_halo_launch(const float* x, int B, int H, int W, int Cin, const float* w, int Cout, int R, int S, int stride,
                         int pad, int Ho, int Wo, const float* bias, const float* residual, int act, int res_after_act,
                         float* out, hipStream_t stream) {
    size_t lds = 0;
    if (!isc_conv_halo_applies(Cin, Cout, R, S, stride, pad, false, &lds)) return ISC_ERR_UNSUPPORTED;
    HaloParams p;
    p.x = x; p.w = w; p.bias = bias; p.res = residual; p.out = out;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.R = R; p.S = S; p.stride = stride; p.pad = pad; p.Ho = Ho; p.Wo = Wo;
    p.cin4 = Cin / 4;
    p.nchunks = R * S * p.cin4;
    p.kq = (p.nchunks + 3) / 4;
    p.wld = (R * S * Cin + 31) / 32 * 32;
    p.hr = (HT - 1) * stride + R;
    p.hc = (HT - 1) * stride + S;
    p.halo_chunks = p.hr * p.hc * p.cin4;
    p.tiles_x = (Wo + HT - 1) / HT;
    p.tiles_img = p.tiles_x * ((Ho + HT - 1) / HT);
    const long long ntiles = (long long)B * p.tiles_img;
    if (ntiles > 0x7fffffffLL) return ISC_ERR_UNSUPPORTED;
    p.ntiles = (int)ntiles;
    p.act = act;
    p.res_after_act = res_after_act;
    const int mi = Cout <= 32 ? 2 : 4;
    p.wl_bytes = p.kq * mi * 16 * 4 * 16;
    halo_fastdiv((unsigned)(p.hc * p.cin4), &p.div_row_mul, &p.div_row_sh);
    halo_fastdiv((unsigned)p.cin4, &p.div_c4_mul, &p.div_c4_sh);
    halo_fastdiv((unsigned)p.tiles_img, &p.div_img_mul, &p.div_img_sh);
    halo_fastdiv((unsigned)p.tiles_x, &p.div_tx_mul, &p.div_tx_sh);
    // more than 64 KiB of dynamic LDS needs the opt-in, once per kernel and device
    static unsigned long long attr_done[2] = {0ull, 0ull};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ISC_ERR_NO_DEVICE;
    const int slot = mi == 2 ? 0 : 1;
    const bool tracked = dev >= 0 && dev < 64;
    if (!tracked || !((__atomic_load_n(&attr_done[slot], __ATOMIC_RELAXED) >> dev) & 1ull)) {
        const void* fn = mi == 2 ? reinterpret_cast<const void*>(&k_conv_halo_f32<2>)
                                 : reinterpret_cast<const void*>(&k_conv_halo_f32<4>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return ISC_ERR_UNSUPPORTED;
        }
        if (tracked) __atomic_fetch_or(&attr_done[slot], 1ull << dev, __ATOMIC_RELAXED);
    }
    const long long resident = 2ll * isc_device_cus();
    const dim3 grid((unsigned)(ntiles < resident ? ntiles : resident)), block(256);
    if (mi == 2) hipLaunchKernelGGL(k_conv_halo_f32<2>, grid, block, lds, stream, p);
    else hipLaunchKernelGGL(k_conv_halo_f32<4>, grid, block, lds, stream, p);
    return isc_launch_status();
}
