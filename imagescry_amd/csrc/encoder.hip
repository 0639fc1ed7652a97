// Encoder blocks in float32 (include/imagescry_hip.h: isc_conv2d_nhwc, isc_conv2d_nhwc_gated, isc_linear_centered,
// isc_dwconv2d_nhwc, isc_dwconv2d_nhwc_pool, isc_se_gate, isc_im2col_nchw, isc_maxpool_nhwc, isc_global_avgpool_nhwc,
// isc_nchw_to_nhwc).
//
// isc_conv2d_nhwc is an implicit GEMM on the f32 matrix cores (v_mfma_f32_16x16x4_f32, exact float32 products
// and a k-ordered fma chain, so results agree with a CPU float32 convolution to rounding):
//
//     out[pixel][cout] = act( sum_{r,s,c} x[b][ho*stride + r - pad][wo*stride + s - pad][c] * w[cout][r][s][c]
//                             + bias[cout] + residual[pixel][cout] )
//
// Layout.  Activations are NHWC and weights KRSC, so for both GEMM operands the reduction axis (r, s, c) is the
// contiguous one; one K step is 32 channels (128 bytes) of one filter tap.  The weight tile is the MFMA "A"
// operand (rows = output channels) and the pixel tile the "B" operand (columns = output pixels): a lane then owns
// four CONSECUTIVE output channels of one pixel, i.e. one 16-byte store into the NHWC output, and bias /
// residual are 16-byte loads.  LDS tiles are [rows][128 B] with the 16-byte chunks XOR-swizzled by (row >> 1) & 7
// (conflict-free ds_read_b128 fragment reads), double buffered, one barrier per K step; the next step is staged by
// LDS-DMA (or, when x needs a per-element transform, prefetched into registers) while the current one is on the matrix
// cores.  The LDS-DMA kernels are persistent (two workgroups per CU walk the tiles); see k_conv_f32 and conv_launch.
//
// Also here: the depthwise row-sweep kernel and the squeeze-excitation gate of the MBConv blocks (k_dwconv3x3_rows,
// k_se_gate), pooling and layout kernels.
#include <stdlib.h>

#include "isc_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct ConvParams {
    const float* x;
    const float* w;
    const float* bias;
    const float* res;
    const float* sub;    // optional per-input-channel vector subtracted from x before the product (PCA centring)
    const float* scale;  // optional [B, Cin] factor applied to x before the product (squeeze-excitation gate)
    // optional SECOND input of a 1 x 1 convolution, K-concatenated: out = act(w[:, :Cin] . x + w[:, Cin:] . x2 + bias):
    // x2 float [B, H2, W2, Cin2], read at (ho * stride2, wo * stride2) -- a ResNet block's 1 x 1 downsample branch folded
    // into its conv3, so that the branch's output is never written and read back as a residual.  K steps >= x2_step0.
    const float* x2;
    int x2_step0, H2, W2, Cin2, stride2;
    float* out;
    int B, H, W, Cin, Cout, R, S, stride, pad, Ho, Wo;
    int M;          // B * Ho * Wo output pixels
    int K;          // R * S * Cin
    int cin_steps;  // Cin / 32
    int cin4;       // packed-K mode: 16-byte chunks per filter tap (Cin / 4)
    int ksteps;     // R * S * cin_steps
    int act;
    int res_after_act;  // out = act(conv + bias) + residual instead of act(conv + bias + residual)
    unsigned div_hw_mul, div_hw_sh;  // n / (Ho * Wo) = umulhi(n, mul) >> sh for n < 2^31 (conv_fastdiv); mul == 0: n itself
    unsigned div_w_mul, div_w_sh;    // n / Wo
    int tile_base;      // this launch's tile 0 in the layer's full-tile numbering
    int tile_split;     // 1, or 2 when the launch computes half tiles (pixel halves) of the full tiles
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ISC_ACT_RELU) return fmaxf(v, 0.f);
    if (act == ISC_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    // hardware exp2 / rcp (about 1 ulp each; 3e-7 relative on the result): the depthwise kernels are VALU-bound on this
    if (act == ISC_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));
    if (act == ISC_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.f + __expf(-v));
    return v;
}

template <int V>
struct IntTag {
    static constexpr int value = V;
};

// Division of n < 2^31 by a run-time constant d without the ~30-instruction integer division sequence (a tile's staging
// rows need 2 * NB of them, and the persistent kernel computes them in front of a tile's last K step): with
// l = ceil(log2 d), p = 31 + l and M = ceil(2^p / d) < 2^32, floor(n / d) = (n * M) >> p exactly for every n < 2^31.
static void conv_fastdiv(unsigned d, unsigned* mul, unsigned* sh) {
    if (d <= 1) { *mul = 0; *sh = 0; return; }
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    const unsigned p = 31 + l;
    *mul = (unsigned)(((1ull << p) + d - 1) / d);
    *sh = p - 32;
}
__device__ __forceinline__ int conv_div(int n, unsigned mul, unsigned sh) {
    return mul ? (int)(__umulhi((unsigned)n, mul) >> sh) : n;
}

// a 128-byte line of zeros: the LDS-DMA source of filter taps outside the image
__device__ __attribute__((aligned(128))) const float g_zero_line[32] = {0.f};

__device__ __forceinline__ void conv_dma16(const void* gsrc, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
#define CONV_DS_READ(dst_, addr_, off_) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "i"(off_))

// TCO output channels x TPIX pixels per workgroup of 4 waves; every wave owns a 64 x 64 sub-tile (32 x 64 in the
// 32-channel tile, for layers with at most 32 output channels: EfficientNetV2's 24-channel stem and first stage).
// TAP4 = packed-K mode, for Cin % 32 != 0 (Cin % 4 == 0): the reduction axis k = (r * S + s) * Cin + c is cut into
// 128-byte K steps regardless of tap boundaries -- every lane's 16-byte chunk (4 channels) finds its own filter tap --
// and weights are [Cout][ceil(R*S*Cin / 32) * 32] with the tail zero.  Cin == 4 (the stem: RGB + one zero channel) is
// the case "8 taps per K step"; 24- and 48-channel stages of EfficientNetV2 run unpadded this way.
// DMA = stage both operand tiles by LDS-DMA (global_load_lds_dwordx4: the per-lane SOURCE address does the im2col gather,
// taps that fall outside the image read a page of zeros) instead of global loads into registers + ds_write_b128: no
// staging registers, no LDS store instructions; the fragment reads then have to be inline asm (a C++ LDS load makes hipcc
// drain the DMA in flight before it).  Used whenever no per-element transform of x is asked for (`sub`, `scale`).
// DUAL = the instantiation that knows the K-concatenated second input (ConvParams::x2); a kernel of its own so that the
// plain ones keep their register allocation (with the branch in every instantiation the 64 x 256 tile spilled 35 registers).
template <int TCO, int TPIX, bool TAP4, bool DMA, bool DUAL = false>
__global__ __launch_bounds__(256, (TCO * TPIX <= 8192 && TCO >= 64 && !TAP4 ? 3 : 2)) void k_conv_f32(const ConvParams p, int ntiles) {
    static_assert(!DUAL || (DMA && !TAP4), "the second input exists in the LDS-DMA form without packed-K only");
    constexpr int WCO = TCO >= 64 ? TCO / 64 : 1;  // waves along the output channels
    constexpr int MI = TCO / WCO / 16;             // 16-channel blocks per wave: 4, or 2 for the 32-channel tile
    constexpr int NA = TCO * 8 / 256;      // 16-byte staging slots per thread, weight tile
    constexpr int NB = TPIX * 8 / 256;     // ... pixel tile
    constexpr int NI = TPIX / (4 / WCO) / 16;      // 16-pixel blocks per wave: 4, or 2 for the 64-pixel half tile
    constexpr int A_BYTES = TCO * 128;
    constexpr int B_BYTES = TPIX * 128;
    constexpr int BUF_BYTES = A_BYTES + B_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wco = wave % WCO;
    const int wpix = wave / WCO;
    // Tiles are numbered output-channel tile fastest (the workgroups that re-read one pixel tile run back to back).  The
    // LDS-DMA instantiations are PERSISTENT: gridDim.x workgroups (two per CU) walk the tiles blockIdx.x, + gridDim.x, ...
    const int n_co_tiles = (p.Cout + TCO - 1) / TCO;

    // ---- staging assignment: slot = tid + 256 * i  ->  row (tid >> 3) + 32 * i, physical chunk tid & 7
    const int srow = tid >> 3;
    const int lchunk = (tid & 7) ^ ((srow >> 1) & 7);  // logical chunk held by this slot ((32 i) >> 1 is 0 mod 8)

    int a_off[NA];   // element offset of the weight row (clamped to the last output channel)
    int b_img[NB];   // image index (for the optional per-image input scale)
    int b_base[NB];  // element offset of image b
    int b_hw0[NB];   // (ho * stride - pad) << 16 | (wo * stride - pad) & 0xffff ; rows past M are pushed out of range
    // packed-K cursor of this lane's chunk (see dma_step): filter tap (pk_r, pk_s), 16-byte piece pk_c4 of its channels
    int pk_r = 0, pk_s = 0, pk_c4 = 0;
    // the staging addresses of one tile (the state dma_step / load_step walk through its K steps)
    // Tile t of this launch is tile p.tile_base + t / p.tile_split of the layer's FULL-tile numbering (pixel tiles of
    // TPIX * p.tile_split), and piece t % p.tile_split of its pixels: the remainder round of a layer is launched as
    // half tiles (conv_launch).
    auto tile_origin = [&](int t, int& oc, int& op) {
        const int full = p.tile_base + t / p.tile_split;
        oc = (full % n_co_tiles) * TCO;
        op = (full / n_co_tiles) * (TPIX * p.tile_split) + (t % p.tile_split) * TPIX;
    };
    auto set_tile = [&](int tile) {
        int tco0, tpix0;
        tile_origin(tile, tco0, tpix0);
#pragma unroll
        for (int i = 0; i < NA; ++i) a_off[i] = min(tco0 + srow + 32 * i, p.Cout - 1) * p.K + lchunk * 4;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int m = tpix0 + srow + 32 * i;
            if (m < p.M) {
                const int b = conv_div(m, p.div_hw_mul, p.div_hw_sh);
                const int rem = m - b * p.Ho * p.Wo;
                const int ho = conv_div(rem, p.div_w_mul, p.div_w_sh);
                const int wo = rem - ho * p.Wo;
                b_img[i] = b;
                b_base[i] = b * p.H * p.W * p.Cin + (TAP4 ? 0 : lchunk * 4);
                b_hw0[i] = ((ho * p.stride - p.pad) << 16) | ((wo * p.stride - p.pad) & 0xffff);
            } else {
                b_img[i] = 0;
                b_base[i] = 0;
                b_hw0[i] = (int)0x80008000;  // hi0 = wi0 = -32768: never inside the image
            }
        }
        if constexpr (TAP4 && DMA) {
            const int tap0 = lchunk / p.cin4;
            pk_c4 = lchunk - tap0 * p.cin4;
            pk_r = tap0 / p.S;
            pk_s = tap0 - pk_r * p.S;
        }
    };
    int tile = blockIdx.x;
    set_tile(tile);
    int co0, pix0;  // of the tile being COMPUTED (set_tile may already describe the next one)
    tile_origin(tile, co0, pix0);

    // ---- fragment read offsets
    const int frow = lane & 15;
    const int fg = lane >> 4;
    const int fsw = (lane >> 1) & 7;
    int foff[2];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) foff[cc] = frow * 128 + (((cc * 4 + fg) ^ fsw) << 4);

    // 16x16 MFMA result layout: column (pixel) = lane & 15, rows (channels) = 4 * (lane >> 4) + 0..3.  In the
    // register-staged instantiations the accumulators start as bias + residual and the epilogue is only the activation and
    // the store; the LDS-DMA ones start from zero and add both in the epilogue, from loads that fly under the last K step
    // (as initial values their latency sat in front of the first MFMA of every tile).
    f32x4 acc[MI][NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int m = pix0 + wpix * (NI * 16) + ni * 16 + (lane & 15);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int co = co0 + wco * (MI * 16) + mi * 16 + (lane >> 4) * 4;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!DMA && m < p.M && co < p.Cout) {
                if (p.bias) v = *reinterpret_cast<const f32x4*>(p.bias + co);
                if (p.res && !p.res_after_act) v += *reinterpret_cast<const f32x4*>(p.res + (size_t)m * p.Cout + co);
            }
            acc[mi][ni] = v;
        }
    }

    u32x4 sa[NA], sb[NB];

    auto load_step = [&](int ks) {
        if constexpr (TAP4) {
            // this thread's 16-byte chunk: chunk kc of the flattened (tap, channel / 4) axis
            const int kc = ks * 8 + lchunk;
            const int tap = kc / p.cin4;
            const int c4 = kc - tap * p.cin4;
            const int r = tap / p.S;
            const int s = tap - r * p.S;
            const bool tap_ok = tap < p.R * p.S;
#pragma unroll
            for (int i = 0; i < NA; ++i) sa[i] = *reinterpret_cast<const u32x4*>(p.w + (size_t)(a_off[i] + ks * 32));
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int hi = (b_hw0[i] >> 16) + r;
                const int wi = (int)(short)(b_hw0[i] & 0xffff) + s;
                const bool ok = tap_ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                u32x4 v = u32x4{0u, 0u, 0u, 0u};
                if (ok) v = *reinterpret_cast<const u32x4*>(p.x + (size_t)(b_base[i] + (hi * p.W + wi) * p.Cin + c4 * 4));
                sb[i] = v;
            }
        } else {
            const int rs = ks / p.cin_steps;
            const int c0 = (ks - rs * p.cin_steps) * 32;
            const int r = rs / p.S;
            const int s = rs - r * p.S;
            const int koff = rs * p.Cin + c0;
#pragma unroll
            for (int i = 0; i < NA; ++i) sa[i] = *reinterpret_cast<const u32x4*>(p.w + (size_t)(a_off[i] + koff));
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int hi = (b_hw0[i] >> 16) + r;
                const int wi = (int)(short)(b_hw0[i] & 0xffff) + s;
                const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                u32x4 v = u32x4{0u, 0u, 0u, 0u};
                if (ok) {
                    v = *reinterpret_cast<const u32x4*>(p.x + (size_t)(b_base[i] + (hi * p.W + wi) * p.Cin + c0));
                    if (p.sub) {  // (x - mean) first, then the product: the order the reference uses
                        const f32x4 mu = *reinterpret_cast<const f32x4*>(p.sub + c0 + lchunk * 4);
                        v = __builtin_bit_cast(u32x4, __builtin_bit_cast(f32x4, v) - mu);
                    }
                    if (p.scale) {
                        const f32x4 g = *reinterpret_cast<const f32x4*>(p.scale + (size_t)b_img[i] * p.Cin + c0 + lchunk * 4);
                        v = __builtin_bit_cast(u32x4, __builtin_bit_cast(f32x4, v) * g);
                    }
                }
                sb[i] = v;
            }
        }
    };
    auto store_step = [&](int buf) {
        unsigned char* a = lds + buf * BUF_BYTES + tid * 16;
        unsigned char* b = a + A_BYTES;
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<u32x4*>(a + 4096 * i) = sa[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<u32x4*>(b + 4096 * i) = sb[i];
    };

    // LDS-DMA form of load_step + store_step: this thread's slots of K step ks straight into buffer buf.  Wave w's 64
    // lanes cover rows 8 w .. 8 w + 7 of a staging round (lane-linear 1 KiB), round i is 32 rows (4 KiB) further.
    auto dma_step = [&](int ks, int buf) {
        unsigned char* a = lds + buf * BUF_BYTES + (tid >> 6) * 1024;
        unsigned char* b = a + A_BYTES;
        if constexpr (TAP4) {
            // this lane's chunk kc = 8 ks + lchunk of the flattened (tap, channel / 4) axis, kept as (r, s, c4) and
            // advanced by 8 chunks per call (dma_step runs for ks = 0, 1, 2, ... in order) instead of divided out
            const int r = pk_r, s = pk_s, c4 = pk_c4;
            const bool tap_ok = r < p.R;
            pk_c4 += 8;
            while (pk_c4 >= p.cin4) {
                pk_c4 -= p.cin4;
                if (++pk_s == p.S) { pk_s = 0; ++pk_r; }
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) conv_dma16(p.w + (size_t)(a_off[i] + ks * 32), a + 4096 * i);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int hi = (b_hw0[i] >> 16) + r;
                const int wi = (int)(short)(b_hw0[i] & 0xffff) + s;
                const bool ok = tap_ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                const float* src = ok ? p.x + (size_t)(b_base[i] + (hi * p.W + wi) * p.Cin + c4 * 4) : g_zero_line + lchunk * 4;
                conv_dma16(src, b + 4096 * i);
            }
        } else if (DUAL && ks >= p.x2_step0) {
            // the K steps of the second input (the main convolution is 1 x 1 / stride 1 / pad 0, so b_hw0 holds the output
            // pixel itself): 32 channels of x2 at (ho * stride2, wo * stride2); the weight rows are [Cin | Cin2]
            const int c0 = (ks - p.x2_step0) * 32;
#pragma unroll
            for (int i = 0; i < NA; ++i) conv_dma16(p.w + (size_t)(a_off[i] + ks * 32), a + 4096 * i);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int ho = b_hw0[i] >> 16;
                const int wo = (int)(short)(b_hw0[i] & 0xffff);
                const float* src = ho >= 0 ? p.x2 + ((size_t)((b_img[i] * p.H2 + ho * p.stride2) * p.W2 + wo * p.stride2) *
                                                         (size_t)p.Cin2 + (size_t)(c0 + lchunk * 4))
                                           : g_zero_line + lchunk * 4;  // rows past M
                conv_dma16(src, b + 4096 * i);
            }
        } else {
            const int rs = ks / p.cin_steps;
            const int c0 = (ks - rs * p.cin_steps) * 32;
            const int r = rs / p.S;
            const int s = rs - r * p.S;
            const int koff = rs * p.Cin + c0;
#pragma unroll
            for (int i = 0; i < NA; ++i) conv_dma16(p.w + (size_t)(a_off[i] + koff), a + 4096 * i);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int hi = (b_hw0[i] >> 16) + r;
                const int wi = (int)(short)(b_hw0[i] & 0xffff) + s;
                const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                const float* src = ok ? p.x + (size_t)(b_base[i] + (hi * p.W + wi) * p.Cin + c0) : g_zero_line + lchunk * 4;
                conv_dma16(src, b + 4096 * i);
            }
        }
    };

    f32x4 rv[MI][NI];  // residual tile and bias of the LDS-DMA path: loaded during the last K step
    f32x4 bv[MI];
    // ---- epilogue: activation and one 16-byte store per (pixel, four channels).  The activation is chosen ONCE, outside
    // the element loops (looked at per element, the switch is a tenth of a short-K tile's time in scalar branches).
    auto epilogue = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int m = pix0 + wpix * (NI * 16) + ni * 16 + frow;
            if (m >= p.M) continue;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int co = co0 + wco * (MI * 16) + mi * 16 + fg * 4;
                if (co >= p.Cout) continue;  // Cout % 4 == 0, so a lane's four channels are in or out together
                f32x4 v = acc[mi][ni];
                if (DMA) v += bv[mi];
                if (DMA && p.res && !p.res_after_act) v += rv[mi][ni];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], ACT < 0 ? p.act : ACT);
                if (p.res && p.res_after_act) {
                    if (DMA) v += rv[mi][ni];
                    else v += *reinterpret_cast<const f32x4*>(p.res + (size_t)m * p.Cout + co);
                }
                *reinterpret_cast<f32x4*>(p.out + (size_t)m * p.Cout + co) = v;
            }
        }
    };
    auto run_epilogue = [&]() {
        if (p.act == ISC_ACT_NONE) epilogue(IntTag<ISC_ACT_NONE>{});
        else if (p.act == ISC_ACT_RELU) epilogue(IntTag<ISC_ACT_RELU>{});
        else if (p.act == ISC_ACT_SILU) epilogue(IntTag<ISC_ACT_SILU>{});
        else epilogue(IntTag<-1>{});
    };

    if constexpr (DMA) {
        dma_step(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
        // one K step of matrix work out of buffer buf (fragment reads in asm, see above)
        auto k_step = [&](int buf) {
            const unsigned a_img = lds_addr + buf * BUF_BYTES + wco * (MI * 16) * 128;
            const unsigned b_img = lds_addr + buf * BUF_BYTES + A_BYTES + wpix * (NI * 16) * 128;
            u32x4 a[2][MI], b[2][NI];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                CONV_DS_READ(a[cc][0], a_img + foff[cc], 0);
                CONV_DS_READ(a[cc][1], a_img + foff[cc], 2048);
                if constexpr (MI == 4) {
                    CONV_DS_READ(a[cc][2], a_img + foff[cc], 4096);
                    CONV_DS_READ(a[cc][3], a_img + foff[cc], 6144);
                }
                CONV_DS_READ(b[cc][0], b_img + foff[cc], 0);
                CONV_DS_READ(b[cc][1], b_img + foff[cc], 2048);
                if constexpr (NI == 4) {
                    CONV_DS_READ(b[cc][2], b_img + foff[cc], 4096);
                    CONV_DS_READ(b[cc][3], b_img + foff[cc], 6144);
                }
            }
            static_assert((MI == 4 && NI == 4) || (MI == 2 && NI == 4) || (MI == 4 && NI == 2), "wave tile");
            if constexpr (NI == 2)
                asm volatile("s_waitcnt lgkmcnt(6)"
                             : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(b[0][0]), "+v"(b[0][1]));
            else if constexpr (MI == 4)
                asm volatile("s_waitcnt lgkmcnt(8)"
                             : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(b[0][0]), "+v"(b[0][1]),
                               "+v"(b[0][2]), "+v"(b[0][3]));
            else
                asm volatile("s_waitcnt lgkmcnt(6)"
                             : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3]));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[0][mi][j]),
                                                                           __uint_as_float(b[0][ni][j]), acc[mi][ni], 0, 0, 0);
            if constexpr (NI == 2)
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]), "+v"(b[1][0]), "+v"(b[1][1]));
            else if constexpr (MI == 4)
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]), "+v"(b[1][0]), "+v"(b[1][1]),
                               "+v"(b[1][2]), "+v"(b[1][3]));
            else
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(a[1][0]), "+v"(a[1][1]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]), "+v"(b[1][3]));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[1][mi][j]),
                                                                           __uint_as_float(b[1][ni][j]), acc[mi][ni], 0, 0, 0);
        };
        auto step_end = [&]() {
            // the next K step has landed (its DMA flew under this step's 128 MFMAs); everyone is done reading this buffer
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        // Bias and residual tile of the tile being computed: fetched under its LAST K step, added in the epilogue.  (As
        // part of the accumulators' initial value the residual's latency sat in front of the first MFMA of every tile:
        // measured on 14 x 14 x 256 -> 1024, B = 512: 676 us with the residual, 546 without -- the whole read,
        // un-overlapped.  Fetching it one step earlier needs more registers than there are: 160-200 bytes of scratch
        // and a slower network.)
        auto fetch_bias_residual = [&]() {
            const int cob = co0 + wco * (MI * 16) + (lane >> 4) * 4;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)  // channel blocks past Cout read the last whole block: never stored
                bv[mi] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + min(cob + mi * 16, p.Cout - 4))
                                : f32x4{0.f, 0.f, 0.f, 0.f};
            if (!p.res) return;
            // 32-bit element offsets from the uniform base (M * Cout < 2^31, checked by the host)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int m = pix0 + wpix * (NI * 16) + ni * 16 + (lane & 15);
                const unsigned row = (unsigned)min(m, p.M - 1) * (unsigned)p.Cout;  // rows past M read row M - 1
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    rv[mi][ni] = *reinterpret_cast<const f32x4*>(p.res + (row + (unsigned)min(cob + mi * 16, p.Cout - 4)));
            }
        };
        int base = 0;  // buffer parity of this tile's K step 0
        for (;;) {
            for (int ks = 0; ks + 1 < p.ksteps; ++ks) {
                const int buf = (ks + base) & 1;
                dma_step(ks + 1, buf ^ 1);
                k_step(buf);
                step_end();
            }
            // last K step of this tile: the other buffer is free (the barrier of the step before says so), so the NEXT
            // tile's K step 0 is staged into it now and no tile after the first starts with an exposed DMA latency
            const int last = (p.ksteps - 1 + base) & 1;
            const int next = tile + (int)gridDim.x;
            if (next < ntiles) {
                set_tile(next);
                dma_step(0, last ^ 1);
            }
            fetch_bias_residual();
            k_step(last);
            // bias, residual and the next tile's first step have landed (this wave's share); nothing younger is in
            // flight, so the stores below drain under the next tile's first K step instead of being waited for here.
            // (The empty asm pins bias and residual HERE: left to itself hipcc re-waits -- vmcnt(0), i.e. for the store
            // just issued -- inside every conditional block of the epilogue.)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                asm volatile("" : "+v"(bv[mi]));
                if (p.res) {
                    asm volatile("" : "+v"(rv[mi][0]), "+v"(rv[mi][1]));
                    if constexpr (NI == 4) asm volatile("" : "+v"(rv[mi][2]), "+v"(rv[mi][3]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            run_epilogue();
            if (next >= ntiles) break;
            __builtin_amdgcn_s_barrier();  // every wave's share has landed; every wave is done with buffer `last`
            __builtin_amdgcn_sched_barrier(0);
            base = last ^ 1;
            tile = next;
            tile_origin(tile, co0, pix0);
            for (int mi = 0; mi < MI; ++mi)
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    } else {
    load_step(0);
    store_step(0);
    __syncthreads();

    for (int ks = 0; ks < p.ksteps; ++ks) {
        const int buf = ks & 1;
        const bool more = ks + 1 < p.ksteps;
        if (more) load_step(ks + 1);

        const unsigned char* a_img = lds + buf * BUF_BYTES + wco * (MI * 16) * 128;
        const unsigned char* b_img = lds + buf * BUF_BYTES + A_BYTES + wpix * (NI * 16) * 128;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            u32x4 a[MI], b[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const u32x4*>(a_img + mi * 2048 + foff[cc]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const u32x4*>(b_img + ni * 2048 + foff[cc]);
            // element j of every lane's chunk feeds the j-th MFMA: a K-axis permutation shared by both operands
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[mi][j]),
                                                                           __uint_as_float(b[ni][j]), acc[mi][ni], 0, 0, 0);
        }
        if (more) store_step(buf ^ 1);
        __syncthreads();
    }
    }
    run_epilogue();
}

__global__ __launch_bounds__(256) void k_nchw_to_nhwc(const float* __restrict__ x, int C, int HW, int Cpad, size_t total,
                                                      float* __restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % Cpad);
        const size_t pix = i / Cpad;  // b * HW + hw
        const size_t b = pix / HW;
        const size_t hw = pix - b * HW;
        y[i] = c < C ? x[(b * C + c) * HW + hw] : 0.f;
    }
}

// patches[m][k], k = (r * S + s) * C + c, four consecutive k per thread (one 16-byte store)
__global__ __launch_bounds__(256) void k_im2col_nchw(const float* __restrict__ x, int C, int H, int W, int R, int S,
                                                     int stride, int pad, int Ho, int Wo, int Kpad, size_t total4,
                                                     float* __restrict__ y) {
    const int kvec = Kpad / 4;
    const int kreal = R * S * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int kv = (int)(i % kvec);
        const size_t m = i / kvec;
        const int b = (int)(m / ((size_t)Ho * Wo));
        const int rem = (int)(m - (size_t)b * Ho * Wo);
        const int ho = rem / Wo, wo = rem - ho * Wo;
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = kv * 4 + j;
            float e = 0.f;
            if (k < kreal) {
                const int rs = k / C, c = k - rs * C;
                const int r = rs / S, s = rs - r * S;
                const int hi = ho * stride + r - pad, wi = wo * stride + s - pad;
                if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W)
                    e = x[(((size_t)b * C + c) * H + hi) * W + wi];
            }
            v[j] = e;
        }
        *reinterpret_cast<f32x4*>(y + i * 4) = v;
    }
}

__global__ __launch_bounds__(256) void k_maxpool_nhwc(const float* __restrict__ x, int H, int W, int C, int R, int stride,
                                                      int pad, int Ho, int Wo, size_t total4, float* __restrict__ y) {
    const int cvec = C / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int cv = (int)(i % cvec);
        const size_t m = i / cvec;
        const int b = (int)(m / ((size_t)Ho * Wo));
        const int rem = (int)(m - (size_t)b * Ho * Wo);
        const int ho = rem / Wo, wo = rem - ho * Wo;
        f32x4 best = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int r = 0; r < R; ++r) {
            const int hi = ho * stride + r - pad;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int s = 0; s < R; ++s) {
                const int wi = wo * stride + s - pad;
                if ((unsigned)wi >= (unsigned)W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)b * H + hi) * W + wi) * C + cv * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) best[j] = fmaxf(best[j], v[j]);
            }
        }
        *reinterpret_cast<f32x4*>(y + i * 4) = best;
    }
}

// depthwise R x R convolution, NHWC, weights [R][R][C]: one thread = four channels of one output pixel
__global__ __launch_bounds__(256) void k_dwconv_nhwc(const float* __restrict__ x, int H, int W, int C, int R, int stride,
                                                     int pad, int Ho, int Wo, const float* __restrict__ w,
                                                     const float* __restrict__ bias, int act, size_t total4,
                                                     float* __restrict__ y) {
    const int cvec = C / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int cv = (int)(i % cvec);
        const size_t m = i / cvec;
        const int b = (int)(m / ((size_t)Ho * Wo));
        const int rem = (int)(m - (size_t)b * Ho * Wo);
        const int ho = rem / Wo, wo = rem - ho * Wo;
        f32x4 acc = bias ? *reinterpret_cast<const f32x4*>(bias + cv * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < R; ++r) {
            const int hi = ho * stride + r - pad;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int s = 0; s < R; ++s) {
                const int wi = wo * stride + s - pad;
                if ((unsigned)wi >= (unsigned)W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)b * H + hi) * W + wi) * C + cv * 4);
                const f32x4 k = *reinterpret_cast<const f32x4*>(w + ((size_t)r * R + s) * C + cv * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaf(v[j], k[j], acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = apply_act(acc[j], act);
        *reinterpret_cast<f32x4*>(y + i * 4) = acc;
    }
}

// Depthwise 3 x 3, stride 1, pad 1 -- the MBConv case -- as a row sweep: one thread = four channels of a strip of TW
// output columns, walking down the image with the three input rows of the window (and the next one, in flight) in registers, so every input element
// is loaded once per strip ((TW + 2) / TW loads per output instead of nine; the nine-fold re-read of k_dwconv_nhwc goes
// through the L2 and is what bounds it).  Lanes = consecutive channel vectors of one pixel: each load / store
// instruction of a wave is one contiguous 1 KiB.  POOL: also the mean over the image of the activated output (the
// squeeze-excitation pooling), written directly (one strip) or with one atomicAdd per strip into a zeroed [B, C]
// (two strips: a + b in either order is the same float).
// WRITE = false is the pooling alone (no y), GATE multiplies the activated output by gate[b, c] before the store: an
// MBConv block runs the kernel twice -- pooled mean, k_se_gate, then y = act(dw) * gate -- so that its 1 x 1 projection is
// a plain convolution (the gate applied while staging the projection's input doubled that convolution's time; the
// second read of x here costs a quarter of that).
// SILU: the activation is known to be SiLU (every MBConv block); otherwise `act` is looked at per element, which costs
// this kernel a quarter of its time in scalar branches.
template <int TW, bool WRITE, bool POOL, bool GATE, bool SILU>
__global__ __launch_bounds__(256) void k_dwconv3x3_rows(const float* __restrict__ x, int H, int W, int C,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        int act, int strips, size_t total, const float* __restrict__ gate,
                                                        float* __restrict__ y, float* __restrict__ pooled, float inv_hw) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int cvec = C / 4;
    const int cv = (int)(i % cvec);
    const size_t t = i / cvec;
    const int strip = (int)(t % strips);
    const size_t b = t / strips;
    const int w0 = strip * TW;

    f32x4 k[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) k[j] = *reinterpret_cast<const f32x4*>(w + (size_t)j * C + cv * 4);
    const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + cv * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xb = x + b * H * W * C + cv * 4;
    float* yb = y + b * H * W * C + cv * 4;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 gv = f32x4{1.f, 1.f, 1.f, 1.f};
    if (GATE) gv = *reinterpret_cast<const f32x4*>(gate + b * C + cv * 4);

    auto load_row = [&](int hi, f32x4 (&dst)[TW + 2]) {
        const bool row_ok = hi < H;  // hi >= 0 always here
#pragma unroll
        for (int j = 0; j < TW + 2; ++j) {
            const int wi = w0 - 1 + j;
            dst[j] = (row_ok && (unsigned)wi < (unsigned)W)
                         ? *reinterpret_cast<const f32x4*>(xb + ((size_t)hi * W + wi) * C) : zero;
        }
    };
    f32x4 sum = zero;
    // rows a, b2, c hold input rows ho - 1, ho, ho + 1; d receives row ho + 2 (in flight under this row's arithmetic);
    // then output row ho
    auto step = [&](f32x4 (&a)[TW + 2], f32x4 (&b2)[TW + 2], f32x4 (&c)[TW + 2], f32x4 (&d)[TW + 2], int ho) {
        load_row(ho + 2, d);
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            if (w0 + j >= W) break;
            f32x4 acc = bv;
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[e] = fmaf(a[j + s2][e], k[s2][e], acc[e]);
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[e] = fmaf(b2[j + s2][e], k[3 + s2][e], acc[e]);
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[e] = fmaf(c[j + s2][e], k[6 + s2][e], acc[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = apply_act(acc[e], SILU ? (int)ISC_ACT_SILU : act);
            if (POOL) sum += acc;
            if (GATE) acc *= gv;
            if (WRITE) *reinterpret_cast<f32x4*>(yb + ((size_t)ho * W + w0 + j) * C) = acc;
        }
    };

    f32x4 r0[TW + 2], r1[TW + 2], r2[TW + 2], r3[TW + 2];
#pragma unroll
    for (int j = 0; j < TW + 2; ++j) r0[j] = zero;  // row -1
    load_row(0, r1);
    load_row(1, r2);
    for (int ho = 0; ho < H; ho += 4) {
        step(r0, r1, r2, r3, ho);
        if (ho + 1 < H) step(r1, r2, r3, r0, ho + 1);
        if (ho + 2 < H) step(r2, r3, r0, r1, ho + 2);
        if (ho + 3 < H) step(r3, r0, r1, r2, ho + 3);
    }
    if (POOL) {
        float* pp = pooled + b * C + cv * 4;
        if (strips == 1) {
            *reinterpret_cast<f32x4*>(pp) = sum * inv_hw;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(pp + e, sum[e] * inv_hw);
        }
    }
}

// grid (ceil(C / 256), B): thread = one channel, loop over the HW positions (coalesced across channels)
__global__ __launch_bounds__(256) void k_global_avgpool_nhwc(const float* __restrict__ x, int HW, int C,
                                                             float* __restrict__ y) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float* p = x + (size_t)blockIdx.y * HW * C + c;
    float acc = 0.f;
    for (int i = 0; i < HW; ++i) acc += p[(size_t)i * C];
    y[(size_t)blockIdx.y * C + c] = acc / (float)HW;
}

// The tail of a pooled encoder in one launch: global average pool -> linear -> (optional) L2 normalisation, HEAD_IMG
// images per workgroup.  As three launches (pool, an M = B-row convolution on 24 workgroups with a serial 64-step K loop,
// the row normalisation) this was 55 + 86 + 5 us at B = 512, C = 2048, E = 768.
//   phase 1  pooled[img][c]: a thread owns channels t, t + 512, ...; sequential float32 sum over the positions, then one
//            division -- the arithmetic of k_global_avgpool_nhwc, bit for bit;
//   phase 2  feat[img][e] = bias[e] + sum_c pooled[img][c] w[e][c]: a wave owns outputs wave, wave + 8, ...; lanes along c
//            with 16-byte weight loads (1 KiB contiguous per wave instruction, the weights stay in the L2), every load
//            serves all the workgroup's images; one wave reduction per (output, image);
//   phase 3  the arithmetic of k_l2norm_rows, bit for bit, on the first 256 threads per image.
#ifndef ISC_HEAD_IMG
#define ISC_HEAD_IMG 2
#endif
constexpr int HEAD_IMG = ISC_HEAD_IMG;
constexpr int HEAD_THREADS = 512;
__global__ __launch_bounds__(HEAD_THREADS) void k_pool_linear_l2norm(const float* __restrict__ x, int B, int HW, int C,
                                                                     const float* __restrict__ w,
                                                                     const float* __restrict__ bias, int E, int normalize,
                                                                     float eps, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float head_lds[];
    float* pooled = head_lds;              // [HEAD_IMG][C]
    float* feat = head_lds + HEAD_IMG * C;  // [HEAD_IMG][E]
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.x * HEAD_IMG;
    const int nimg = min(HEAD_IMG, B - b0);
    // phase 1: four channels per thread (16-byte loads), positions in order, seven loads in flight
    for (int img = 0; img < HEAD_IMG; ++img)
        for (int c = tid * 4; c < C; c += HEAD_THREADS * 4) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (img < nimg) {
                const float* p = x + ((size_t)(b0 + img) * HW) * C + c;
#pragma unroll 7
                for (int i = 0; i < HW; ++i) acc += *reinterpret_cast<const f32x4*>(p + (size_t)i * C);
                acc = acc / (float)HW;
            }
            *reinterpret_cast<f32x4*>(pooled + img * C + c) = acc;
        }
    __syncthreads();
    // phase 2: four outputs per wave and trip -- their weight loads are independent, so 4 - 8 KiB are in flight per wave
    // (one output at a time paid an L2 round trip per KiB: 430 us for B = 512)
    constexpr int EU = 4;
    for (int e0 = wave * EU; e0 < E; e0 += (HEAD_THREADS / 64) * EU) {
        float acc[EU][HEAD_IMG];
#pragma unroll
        for (int u = 0; u < EU; ++u)
#pragma unroll
            for (int img = 0; img < HEAD_IMG; ++img) acc[u][img] = 0.f;
#pragma unroll 8
        for (int c = lane * 4; c < C; c += 256) {
            f32x4 wv[EU];
#pragma unroll
            for (int u = 0; u < EU; ++u)
                wv[u] = *reinterpret_cast<const f32x4*>(w + (size_t)min(e0 + u, E - 1) * C + c);
            f32x4 pv[HEAD_IMG];
#pragma unroll
            for (int img = 0; img < HEAD_IMG; ++img) pv[img] = *reinterpret_cast<const f32x4*>(pooled + img * C + c);
#pragma unroll
            for (int u = 0; u < EU; ++u)
#pragma unroll
                for (int img = 0; img < HEAD_IMG; ++img) {
                    acc[u][img] = fmaf(wv[u][0], pv[img][0], acc[u][img]);
                    acc[u][img] = fmaf(wv[u][1], pv[img][1], acc[u][img]);
                    acc[u][img] = fmaf(wv[u][2], pv[img][2], acc[u][img]);
                    acc[u][img] = fmaf(wv[u][3], pv[img][3], acc[u][img]);
                }
        }
#pragma unroll
        for (int u = 0; u < EU; ++u)
#pragma unroll
            for (int img = 0; img < HEAD_IMG; ++img) {
                const float tot = isc_wave_sum(acc[u][img]);
                if (lane == 0 && e0 + u < E) feat[img * E + e0 + u] = tot + (bias ? bias[e0 + u] : 0.f);
            }
    }
    __syncthreads();
    for (int img = 0; img < nimg; ++img) {
        const float* p = feat + img * E;
        float* o = out + (size_t)(b0 + img) * E;
        if (!normalize) {
            for (int i = tid; i < E; i += HEAD_THREADS) o[i] = p[i];
            continue;
        }
        if (tid < 256) {  // k_l2norm_rows's own order: strided partial sums, wave butterflies, four wave sums in order
            float acc = 0.f;
            for (int i = tid; i < E; i += 256) {
                const float v = p[i];
                acc += v * v;
            }
            acc = isc_wave_sum(acc);
            if (lane == 0) red[wave] = acc;
        }
        __syncthreads();
        const float denom = fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), eps);
        for (int i = tid; i < E; i += HEAD_THREADS) o[i] = __fdiv_rn(p[i], denom);
        __syncthreads();
    }
}

// Squeeze-excitation gate of IMG images per workgroup: gate = sigmoid(w2 . silu(w1 . pooled + b1) + b2).  As two
// launches of the convolution kernel these are M = B-row GEMMs on two or three workgroups with a serial 48-step K loop
// (143 + 34 us for C = 1536, S = 64, B = 512).  This is a latency problem (0.8 MB of L2-resident weights per workgroup),
// so the shape is chosen for loads in flight: 16 waves; fc1 gives every wave four outputs at once with the lanes along C
// (coalesced 16-byte weight loads, one shuffle reduction per output); fc2 puts LPR = 2^k >= S / 4 lanes along a weight
// row, 64 / LPR rows per load instruction, four instructions in flight; the squeezed vector stays in LDS and every
// weight load serves IMG images.
constexpr int SE_THREADS = 1024;

// Sum over aligned groups of `width` lanes (a power of two, wave-uniform), left in every lane of the group.  Within a
// 16-lane row the exchange is a DPP modifier on the add (quad permutes, then the half-row and row mirrors -- any pairing
// of the right sub-groups does for a sum); only widths above 16 go through the LDS permute unit.  With __shfl_xor at
// every step (a ds_bpermute and its latency each) the reductions were three quarters of k_se_gate's fc2 phase.
template <int CTRL>
__device__ __forceinline__ float se_dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float se_group_sum(float v, int width) {
    if (width > 1) v = se_dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    if (width > 2) v = se_dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    if (width > 4) v = se_dpp_add<0x141>(v);  // row_half_mirror
    if (width > 8) v = se_dpp_add<0x140>(v);  // row_mirror
    for (int off = 16; off < width; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <int IMG>
__global__ __launch_bounds__(SE_THREADS) void k_se_gate(const float* __restrict__ pooled, int B, int C,
                                                        const float* __restrict__ w1, int ld1, const float* __restrict__ b1,
                                                        int S, const float* __restrict__ w2, int ld2,
                                                        const float* __restrict__ b2, int lpr, float* __restrict__ gate) {
#ifdef ISC_ABLATION
    const int se_abl = lpr >> 16;  // timing aids (wrong results): 1 = stop after fc1, 2 = skip fc1
    lpr &= 0xffff;
#endif
    extern __shared__ __attribute__((aligned(16))) float se_lds[];
    float* sx = se_lds;            // [IMG][C]
    float* sq = se_lds + IMG * C;  // [IMG][S]
    constexpr int NW = SE_THREADS / 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int img0 = blockIdx.x * IMG;
    const int cvec = C / 4;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < IMG * cvec; i += SE_THREADS) {
        const int im = i / cvec, c4 = i - im * cvec;
        f32x4 v = zero;
        if (img0 + im < B) v = *reinterpret_cast<const f32x4*>(pooled + (size_t)(img0 + im) * C + c4 * 4);
        *reinterpret_cast<f32x4*>(sx + im * C + c4 * 4) = v;
    }
    __syncthreads();
    // fc1: outputs n0 + {0, NW, 2 NW, 3 NW} of this wave together
#ifdef ISC_ABLATION
    if (se_abl == 2) {
        for (int i = tid; i < IMG * S; i += SE_THREADS) sq[i] = 0.5f;
    } else
#endif
    for (int n0 = wave; n0 < S; n0 += 4 * NW) {
        float acc[4][IMG];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int im = 0; im < IMG; ++im) acc[j][im] = 0.f;
        for (int cb = lane; cb < cvec; cb += 64 * 3) {  // twelve weight loads in flight per lane
            f32x4 wv[3][4];
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + j * NW, c4 = cb + 64 * u;
                    wv[u][j] = (n < S && c4 < cvec) ? *reinterpret_cast<const f32x4*>(w1 + (size_t)n * ld1 + c4 * 4) : zero;
                }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int c4 = cb + 64 * u;
                if (c4 >= cvec) break;
#pragma unroll
                for (int im = 0; im < IMG; ++im) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(sx + im * C + c4 * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[j][im] = fmaf(wv[u][j][e], xv[e], acc[j][im]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int im = 0; im < IMG; ++im) acc[j][im] = se_group_sum(acc[j][im], 64);
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + j * NW;
                if (n >= S) continue;
                const float bias = b1 ? b1[n] : 0.f;
#pragma unroll
                for (int im = 0; im < IMG; ++im) sq[im * S + n] = apply_act(acc[j][im] + bias, ISC_ACT_SILU);
            }
        }
    }
    __syncthreads();
#ifdef ISC_ABLATION
    if (se_abl == 1) return;
#endif
    // fc2: lane = (row r of the load, 16-byte piece s4 of the row)
    const int svec = S / 4;
    const int rpl = 64 / lpr;
    const int r = lane / lpr;
    const int s4 = lane - r * lpr;
    const bool s_ok = s4 < svec;
    f32x4 qv[IMG];
#pragma unroll
    for (int im = 0; im < IMG; ++im) qv[im] = s_ok ? *reinterpret_cast<const f32x4*>(sq + im * S + s4 * 4) : zero;
    const int step = NW * rpl;  // rows per round of the workgroup
    constexpr int NR = 8;  // weight loads in flight per lane (12 measured slower)
    for (int c0 = wave * rpl + r; c0 < C + (NR - 1) * step; c0 += NR * step) {
        f32x4 wv[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int c = c0 + j * step;
            wv[j] = (s_ok && c < C) ? *reinterpret_cast<const f32x4*>(w2 + (size_t)c * ld2 + s4 * 4) : zero;
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int c = c0 + j * step;
            float acc[IMG];
#pragma unroll
            for (int im = 0; im < IMG; ++im) {
                acc[im] = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[im] = fmaf(wv[j][e], qv[im][e], acc[im]);
                acc[im] = se_group_sum(acc[im], lpr);
            }
            // the pooled vector is no longer needed: its LDS takes the pre-activation gate, written out below as whole
            // 16-byte pieces by all lanes (four lanes per wave storing 4 bytes each was a quarter of this phase)
            if (s4 == 0 && c < C) {
#pragma unroll
                for (int im = 0; im < IMG; ++im) sx[im * C + c] = acc[im];
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < IMG * cvec; i += SE_THREADS) {
        const int im = i / cvec, c4 = i - im * cvec;
        if (img0 + im >= B) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(sx + im * C + c4 * 4);
        if (b2) v += *reinterpret_cast<const f32x4*>(b2 + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], ISC_ACT_SIGMOID);
        *reinterpret_cast<f32x4*>(gate + (size_t)(img0 + im) * C + c4 * 4) = v;
    }
}

int stream_grid(size_t items) {
    size_t blocks = isc_ceil_div(items, (size_t)256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    return blocks == 0 ? 1 : (int)blocks;
}

}  // namespace

#ifdef ISC_ABLATION
static bool conv_no_dma() {
    static const bool v = getenv("ISC_CONV_NO_DMA") != nullptr;  // A/B aid: register-staged operands everywhere
    return v;
}
#else
static constexpr bool conv_no_dma() { return false; }
#endif

static int conv_cu_count() { return isc_device_cus(); }  // of the current device, cached per device id
#ifdef ISC_ABLATION
static bool conv_one_tile_per_wg() {
    static const bool v = getenv("ISC_CONV_NO_PERSIST") != nullptr;  // A/B aid: one workgroup per tile, as before
    return v;
}
#else
static constexpr bool conv_one_tile_per_wg() { return false; }
#endif

#ifdef ISC_ABLATION
static bool conv_no_halo() {
    static const bool v = getenv("ISC_CONV_NO_HALO") != nullptr;  // A/B aid: small-channel layers on k_conv_f32's packed-K mode
    return v;
}
#else
static constexpr bool conv_no_halo() { return false; }
#endif

#ifdef ISC_ABLATION
static bool conv_no_split() {
    static const bool v = getenv("ISC_CONV_NO_SPLIT") != nullptr;  // A/B aid: no half-tile remainder launch
    return v;
}
#else
static constexpr bool conv_no_split() { return false; }
#endif

// Layers of at most this many K steps run as HALF tiles throughout (three workgroups per CU instead of two), see conv_launch
#ifdef ISC_ABLATION
static int conv_halves_ksteps() {
    static const int v = getenv("ISC_CONV_HALVES_KSTEPS") ? atoi(getenv("ISC_CONV_HALVES_KSTEPS")) : 8;
    return v;
}
#else
static constexpr int conv_halves_ksteps() { return 8; }
#endif

// conv_halo.hip: small-channel R x R layers with the pixel operand gathered from an input halo tile in LDS
bool isc_conv_halo_applies(int Cin, int Cout, int R, int S, int stride, int pad, bool has_sub_or_scale, size_t* lds_bytes);
int isc_conv_halo_launch(const float* x, int B, int H, int W, int Cin, const float* w, int Cout, int R, int S, int stride,
                         int pad, int Ho, int Wo, const float* bias, const float* residual, int act, int res_after_act,
                         float* out, hipStream_t stream);

struct ConvSecondInput {  // see ConvParams::x2
    const float* x2 = nullptr;
    int H2 = 0, W2 = 0, Cin2 = 0, stride2 = 1;
};

static int conv_launch(const float* x, int B, int H, int W, int Cin, const float* w, int Cout, int R, int S, int stride,
                       int pad, const float* bias, const float* residual, const float* sub, const float* scale, int act,
                       float* out, void* stream, const ConvSecondInput second = ConvSecondInput{}) {
    ISC_REQUIRE(x && w && out);
    ISC_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0);
    const int res_after_act = (act & ISC_ACT_RESIDUAL_AFTER) ? 1 : 0;
    act &= ~ISC_ACT_RESIDUAL_AFTER;
    ISC_REQUIRE(act >= ISC_ACT_NONE && act <= ISC_ACT_SIGMOID);
    if (Cin % 4 != 0 || Cout % 4 != 0) return ISC_ERR_UNSUPPORTED;  // pad channels with zeros
    const bool tap4 = Cin % 32 != 0;  // packed-K mode: w is [Cout][ceil(R*S*Cin/32)*32]
    if ((scale || sub) && (tap4 || (scale && !isc_aligned(scale, 16)))) return ISC_ERR_UNSUPPORTED;
    if (H > 16384 || W > 16384 || pad > 8192) return ISC_ERR_UNSUPPORTED;
    const int Ho = (H + 2 * pad - R) / stride + 1;
    const int Wo = (W + 2 * pad - S) / stride + 1;
    ISC_REQUIRE(Ho > 0 && Wo > 0);
    const int64_t M = (int64_t)B * Ho * Wo;
    int ksteps = tap4 ? isc_ceil_div(R * S * Cin, 32) : R * S * (Cin / 32);
    const int x2_step0 = ksteps;
    if (second.x2) {  // K-concatenated second 1 x 1 input: the LDS-DMA form of a plain 1 x 1 / 1 / 0 convolution only
        if (tap4 || sub || scale || R != 1 || S != 1 || stride != 1 || pad != 0) return ISC_ERR_UNSUPPORTED;
        ISC_REQUIRE(second.H2 > 0 && second.W2 > 0 && second.Cin2 > 0 && second.stride2 > 0);
        if (second.Cin2 % 32 != 0) return ISC_ERR_UNSUPPORTED;
        ISC_REQUIRE((second.H2 - 1) / second.stride2 + 1 == H && (second.W2 - 1) / second.stride2 + 1 == W);
        if ((int64_t)B * second.H2 * second.W2 >= (1ll << 31)) return ISC_ERR_UNSUPPORTED;
        if (!isc_aligned(second.x2, 16)) return ISC_ERR_ALIGNMENT;
        ksteps += second.Cin2 / 32;
    }
    const int64_t K = (int64_t)ksteps * 32;
    if ((int64_t)B * H * W * Cin >= (1ll << 31) || M * Cout >= (1ll << 31) || (int64_t)Cout * K >= (1ll << 31))
        return ISC_ERR_UNSUPPORTED;  // 32-bit element offsets inside the kernel
    if (!isc_aligned(x, 16) || !isc_aligned(w, 16) || !isc_aligned(out, 16) || (bias && !isc_aligned(bias, 16)) ||
        (residual && !isc_aligned(residual, 16)))
        return ISC_ERR_ALIGNMENT;
    // the RGB stems and EfficientNetV2's 24-channel stage: pixel operand from an input halo tile in LDS (conv_halo.hip)
    if (tap4 && !second.x2 && !conv_no_halo() && M * Cout < (1ll << 31) &&
        isc_conv_halo_applies(Cin, Cout, R, S, stride, pad, sub != nullptr || scale != nullptr, nullptr)) {
        if (!isc_aligned(x, 16) || !isc_aligned(w, 16) || !isc_aligned(out, 16) || (bias && !isc_aligned(bias, 16)) ||
            (residual && !isc_aligned(residual, 16)))
            return ISC_ERR_ALIGNMENT;
        hipStream_t hs = isc_stream(stream);
        isc_timing_begin(ISC_KERNEL_CONV, hs);
        const int st = isc_conv_halo_launch(x, B, H, W, Cin, w, Cout, R, S, stride, pad, Ho, Wo, bias, residual, act,
                                            res_after_act, out, hs);
        isc_timing_end(ISC_KERNEL_CONV, hs);
        return st;
    }
    ConvParams p;
    p.x = x; p.w = w; p.bias = bias; p.res = residual; p.sub = sub; p.scale = scale; p.out = out;
    p.x2 = second.x2; p.x2_step0 = x2_step0; p.H2 = second.H2; p.W2 = second.W2; p.Cin2 = second.Cin2; p.stride2 = second.stride2;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
    p.Ho = Ho; p.Wo = Wo; p.M = (int)M; p.K = (int)K; p.cin_steps = tap4 ? 1 : Cin / 32; p.cin4 = Cin / 4; p.ksteps = ksteps; p.act = act; p.res_after_act = res_after_act; p.tile_base = 0; p.tile_split = 1;
    conv_fastdiv((unsigned)(Ho * Wo), &p.div_hw_mul, &p.div_hw_sh);
    conv_fastdiv((unsigned)Wo, &p.div_w_mul, &p.div_w_sh);
    hipStream_t s = isc_stream(stream);
    // 64-channel tiles wherever they pad Cout less than 128-channel ones do (Cout <= 64, but also 160 -> 192 instead of
    // 256, 192 -> 192 instead of 256): the wasted quarter of the matrix work is worth more than the extra tile reloads
    const bool narrow = isc_ceil_div(Cout, 64) * 64 < isc_ceil_div(Cout, 128) * 128;
    const bool dma = !sub && !scale && !conv_no_dma();
    const bool tile32 = Cout <= 32 && dma;  // half the matrix work of the 64-channel tile
    const int64_t blocks = tile32   ? isc_ceil_div<int64_t>(M, 256)
                           : narrow ? isc_ceil_div(Cout, 64) * isc_ceil_div<int64_t>(M, 256)
                                    : isc_ceil_div(Cout, 128) * isc_ceil_div<int64_t>(M, 128);
    if (blocks > 0x7fffffff) return ISC_ERR_UNSUPPORTED;
    const int ntiles = (int)blocks;
    // the LDS-DMA kernels are persistent: two workgroups per CU (what their LDS and registers allow) walk the tiles
    const int64_t resident = 2 * (int64_t)conv_cu_count();
    const dim3 grid((unsigned)(dma && !conv_one_tile_per_wg() && blocks > resident ? resident : blocks)), block(256);
    isc_timing_begin(ISC_KERNEL_CONV, s);
    if (second.x2) {  // the K-concatenated form: its own instantiations, one launch
        if (narrow) hipLaunchKernelGGL((k_conv_f32<64, 256, false, true, true>), grid, block, 0, s, p, ntiles);
        else hipLaunchKernelGGL((k_conv_f32<128, 128, false, true, true>), grid, block, 0, s, p, ntiles);
        isc_timing_end(ISC_KERNEL_CONV, s);
        return isc_launch_status();
    }
    // A tile count that leaves a thin last round on the resident workgroups (ResNet-50 layer3 / layer4: 1 568 and 784
    // tiles on 512): whole rounds as one perfectly balanced launch, the remainder as HALF tiles (128 x 64, 64 x 128) in a
    // second one -- its round then costs half a tile time, or spreads over twice the CUs, instead of a whole one.  Worth a
    // second launch only when a tile is long (>= 16 K steps) or there is no whole round at all.
    // Short-K layers (K <= 256: the 1 x 1 expand / project convolutions) are bound by their tiles' prologue and epilogue,
    // not by the matrix pipe: as half tiles (128 x 64, 64 x 128; 48 KiB of LDS, 168 registers) THREE workgroups fit a CU
    // and a third independent stream covers the other two's residual reads and stores.  Same-device A/B of the step
    // (gpurun_out/r4/ab_halves_*.log): EfficientNetV2-S 35.92 -> 35.74 ms, ResNet-50 39.89 -> 39.79 ms.
    if (dma && !tile32 && !tap4 && !conv_one_tile_per_wg() && ksteps <= conv_halves_ksteps() && 2 * blocks <= 0x7fffffff) {
        ConvParams q = p;
        q.tile_split = 2;
        const int halves = (int)(2 * blocks);
        const int64_t res3 = 3 * (int64_t)conv_cu_count();
        const dim3 g((unsigned)(halves > res3 ? res3 : halves));
        if (narrow) hipLaunchKernelGGL((k_conv_f32<64, 128, false, true>), g, block, 0, s, q, halves);
        else hipLaunchKernelGGL((k_conv_f32<128, 64, false, true>), g, block, 0, s, q, halves);
        isc_timing_end(ISC_KERNEL_CONV, s);
        return isc_launch_status();
    }
    if (dma && !tile32 && !conv_one_tile_per_wg() && !conv_no_split()) {
        const int64_t rounds = blocks / resident, rem = blocks - rounds * resident;
        if (rem > 0 && rem * 4 <= resident * 3 && (ksteps >= 16 || rounds == 0)) {
            if (rounds > 0) {
                const dim3 g1((unsigned)resident);
                const int n1 = (int)(rounds * resident);
                if (narrow && tap4) hipLaunchKernelGGL((k_conv_f32<64, 256, true, true>), g1, block, 0, s, p, n1);
                else if (narrow) hipLaunchKernelGGL((k_conv_f32<64, 256, false, true>), g1, block, 0, s, p, n1);
                else if (tap4) hipLaunchKernelGGL((k_conv_f32<128, 128, true, true>), g1, block, 0, s, p, n1);
                else hipLaunchKernelGGL((k_conv_f32<128, 128, false, true>), g1, block, 0, s, p, n1);
            }
            ConvParams q = p;
            q.tile_base = (int)(rounds * resident);
            q.tile_split = 2;
            const int halves = (int)(2 * rem);
            const dim3 g2((unsigned)(halves > 3 * resident / 2 ? 3 * resident / 2 : halves));
            if (narrow && tap4) hipLaunchKernelGGL((k_conv_f32<64, 128, true, true>), g2, block, 0, s, q, halves);
            else if (narrow) hipLaunchKernelGGL((k_conv_f32<64, 128, false, true>), g2, block, 0, s, q, halves);
            else if (tap4) hipLaunchKernelGGL((k_conv_f32<128, 64, true, true>), g2, block, 0, s, q, halves);
            else hipLaunchKernelGGL((k_conv_f32<128, 64, false, true>), g2, block, 0, s, q, halves);
            isc_timing_end(ISC_KERNEL_CONV, s, rounds > 0 ? 2 : 1);
            return isc_launch_status();
        }
    }
    if (tile32) {
        if (tap4) hipLaunchKernelGGL((k_conv_f32<32, 256, true, true>), grid, block, 0, s, p, ntiles);
        else hipLaunchKernelGGL((k_conv_f32<32, 256, false, true>), grid, block, 0, s, p, ntiles);
    } else if (dma) {
        if (narrow && tap4) hipLaunchKernelGGL((k_conv_f32<64, 256, true, true>), grid, block, 0, s, p, ntiles);
        else if (narrow) hipLaunchKernelGGL((k_conv_f32<64, 256, false, true>), grid, block, 0, s, p, ntiles);
        else if (tap4) hipLaunchKernelGGL((k_conv_f32<128, 128, true, true>), grid, block, 0, s, p, ntiles);
        else hipLaunchKernelGGL((k_conv_f32<128, 128, false, true>), grid, block, 0, s, p, ntiles);
    } else {
        if (narrow && tap4) hipLaunchKernelGGL((k_conv_f32<64, 256, true, false>), grid, block, 0, s, p, ntiles);
        else if (narrow) hipLaunchKernelGGL((k_conv_f32<64, 256, false, false>), grid, block, 0, s, p, ntiles);
        else if (tap4) hipLaunchKernelGGL((k_conv_f32<128, 128, true, false>), grid, block, 0, s, p, ntiles);
        else hipLaunchKernelGGL((k_conv_f32<128, 128, false, false>), grid, block, 0, s, p, ntiles);
    }
    isc_timing_end(ISC_KERNEL_CONV, s);
    return isc_launch_status();
}

extern "C" int isc_conv2d_nhwc(const float* x, int B, int H, int W, int Cin, const float* w, int Cout, int R, int S,
                               int stride, int pad, const float* bias, const float* residual, int act, float* out,
                               void* stream) {
    return conv_launch(x, B, H, W, Cin, w, Cout, R, S, stride, pad, bias, residual, nullptr, nullptr, act, out, stream);
}

extern "C" int isc_conv2d_nhwc_dual(const float* x, int B, int H, int W, int Cin, const float* x2, int H2, int W2, int Cin2,
                                    int stride2, const float* w, int Cout, const float* bias, const float* residual,
                                    int act, float* out, void* stream) {
    ISC_REQUIRE(x2);
    ConvSecondInput second;
    second.x2 = x2; second.H2 = H2; second.W2 = W2; second.Cin2 = Cin2; second.stride2 = stride2;
    return conv_launch(x, B, H, W, Cin, w, Cout, 1, 1, 1, 0, bias, residual, nullptr, nullptr, act, out, stream, second);
}

extern "C" int isc_conv2d_nhwc_gated(const float* x, int B, int H, int W, int Cin, const float* gate, const float* w,
                                     int Cout, int R, int S, int stride, int pad, const float* bias,
                                     const float* residual, int act, float* out, void* stream) {
    ISC_REQUIRE(gate);
    return conv_launch(x, B, H, W, Cin, w, Cout, R, S, stride, pad, bias, residual, nullptr, gate, act, out, stream);
}

extern "C" int isc_linear_centered(const float* x, int64_t n, int F, const float* mean, const float* w, int K,
                                   const float* bias, float* out, void* stream) {
    ISC_REQUIRE(n > 0 && n < (1ll << 31));
    if (mean && !isc_aligned(mean, 16)) return ISC_ERR_ALIGNMENT;
    return conv_launch(x, (int)n, 1, 1, F, w, K, 1, 1, 1, 0, bias, nullptr, mean, nullptr, ISC_ACT_NONE, out, stream);
}

// Gram matrix of the ROWS of a [F, n] matrix: gram[f1][f2] = sum_i xt[f1][i] * xt[f2][i].  The implicit-GEMM kernel with
// the same operand on both sides: F "pixels" and F "output channels" of n "input channels" each.
extern "C" int isc_gram_rows(const float* xt, int F, int64_t n, float* gram, void* stream) {
    ISC_REQUIRE(xt && gram && F > 0 && n > 0 && n < (1ll << 31));
    if (n % 32 != 0) return ISC_ERR_UNSUPPORTED;  // whole K steps: pad the sample axis with zero columns
    return conv_launch(xt, F, 1, 1, (int)n, xt, F, 1, 1, 1, 0, nullptr, nullptr, nullptr, nullptr, ISC_ACT_NONE, gram,
                       stream);
}

extern "C" int isc_nchw_to_nhwc(const float* x, int B, int C, int H, int W, int Cpad, float* y, void* stream) {
    ISC_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C);
    const size_t total = (size_t)B * H * W * Cpad;
    hipLaunchKernelGGL(k_nchw_to_nhwc, dim3(stream_grid(total)), dim3(256), 0, isc_stream(stream), x, C, H * W, Cpad,
                       total, y);
    return isc_launch_status();
}

extern "C" int isc_im2col_nchw(const float* x, int B, int C, int H, int W, int R, int S, int stride, int pad, int Kpad,
                               float* y, void* stream) {
    ISC_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0);
    ISC_REQUIRE(Kpad >= R * S * C);
    if (Kpad % 4 != 0 || !isc_aligned(y, 16)) return ISC_ERR_ALIGNMENT;
    const int Ho = (H + 2 * pad - R) / stride + 1;
    const int Wo = (W + 2 * pad - S) / stride + 1;
    ISC_REQUIRE(Ho > 0 && Wo > 0);
    const size_t total4 = (size_t)B * Ho * Wo * (Kpad / 4);
    hipLaunchKernelGGL(k_im2col_nchw, dim3(stream_grid(total4)), dim3(256), 0, isc_stream(stream), x, C, H, W, R, S,
                       stride, pad, Ho, Wo, Kpad, total4, y);
    return isc_launch_status();
}

extern "C" int isc_maxpool_nhwc(const float* x, int B, int H, int W, int C, int R, int stride, int pad, float* y,
                                void* stream) {
    ISC_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && R > 0 && stride > 0 && pad >= 0);
    if (C % 4 != 0) return ISC_ERR_UNSUPPORTED;
    if (!isc_aligned(x, 16) || !isc_aligned(y, 16)) return ISC_ERR_ALIGNMENT;
    const int Ho = (H + 2 * pad - R) / stride + 1;
    const int Wo = (W + 2 * pad - R) / stride + 1;
    ISC_REQUIRE(Ho > 0 && Wo > 0);
    const size_t total4 = (size_t)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(k_maxpool_nhwc, dim3(stream_grid(total4)), dim3(256), 0, isc_stream(stream), x, H, W, C, R, stride,
                       pad, Ho, Wo, total4, y);
    return isc_launch_status();
}

#ifdef ISC_ABLATION
static bool dwconv_no_rows() {
    static const bool v = getenv("ISC_DWCONV_NO_ROWS") != nullptr;  // A/B aid: the one-pixel-per-thread kernel everywhere
    return v;
}
#else
static constexpr bool dwconv_no_rows() { return false; }
#endif

extern "C" int isc_dwconv2d_nhwc(const float* x, int B, int H, int W, int C, const float* w, int R, int stride, int pad,
                                 const float* bias, int act, float* y, void* stream) {
    ISC_REQUIRE(x && w && y && B > 0 && H > 0 && W > 0 && C > 0 && R > 0 && stride > 0 && pad >= 0);
    ISC_REQUIRE(act >= ISC_ACT_NONE && act <= ISC_ACT_SIGMOID);
    if (C % 4 != 0) return ISC_ERR_UNSUPPORTED;
    if (!isc_aligned(x, 16) || !isc_aligned(w, 16) || !isc_aligned(y, 16) || (bias && !isc_aligned(bias, 16)))
        return ISC_ERR_ALIGNMENT;
    if (R == 3 && stride == 1 && pad == 1 && !dwconv_no_rows())  // the row-sweep kernel
        return isc_dwconv2d_nhwc_pool(x, B, H, W, C, w, R, stride, pad, bias, act, nullptr, y, nullptr, stream);
    const int Ho = (H + 2 * pad - R) / stride + 1;
    const int Wo = (W + 2 * pad - R) / stride + 1;
    ISC_REQUIRE(Ho > 0 && Wo > 0);
    const size_t total4 = (size_t)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(k_dwconv_nhwc, dim3(stream_grid(total4)), dim3(256), 0, isc_stream(stream), x, H, W, C, R, stride,
                       pad, Ho, Wo, w, bias, act, total4, y);
    return isc_launch_status();
}

extern "C" int isc_dwconv2d_nhwc_pool(const float* x, int B, int H, int W, int C, const float* w, int R, int stride,
                                      int pad, const float* bias, int act, const float* gate, float* y, float* pooled,
                                      void* stream) {
    ISC_REQUIRE(x && w && (y || pooled) && B > 0 && H > 0 && W > 0 && C > 0 && R > 0 && stride > 0 && pad >= 0);
    ISC_REQUIRE(act >= ISC_ACT_NONE && act <= ISC_ACT_SIGMOID);
    ISC_REQUIRE(!gate || y);
    if (C % 4 != 0) return ISC_ERR_UNSUPPORTED;
    if (!isc_aligned(x, 16) || !isc_aligned(w, 16) || (y && !isc_aligned(y, 16)) || (bias && !isc_aligned(bias, 16)) ||
        (pooled && !isc_aligned(pooled, 16)) || (gate && !isc_aligned(gate, 16)))
        return ISC_ERR_ALIGNMENT;
    hipStream_t s = isc_stream(stream);
    constexpr int TW = 7;
    const int strips = isc_ceil_div(W, TW);
    const bool rows = R == 3 && stride == 1 && pad == 1 && !dwconv_no_rows();
    if (rows && (!pooled || strips <= 2)) {
        const size_t total = (size_t)B * strips * (C / 4);
        const size_t blocks = isc_ceil_div(total, (size_t)256);
        if (blocks > 0x7fffffff) return ISC_ERR_UNSUPPORTED;
        if (pooled && strips > 1 && hipMemsetAsync(pooled, 0, (size_t)B * C * sizeof(float), s) != hipSuccess)
            return ISC_ERR_LAUNCH;
        const dim3 grid((unsigned)blocks), block(256);
        const float inv_hw = 1.f / (float)(H * W);
#define ISC_DW_ROWS(WRITE_, POOL_, GATE_)                                                                            \
    do {                                                                                                             \
        if (act == ISC_ACT_SILU)                                                                                     \
            hipLaunchKernelGGL((k_dwconv3x3_rows<TW, WRITE_, POOL_, GATE_, true>), grid, block, 0, s, x, H, W, C, w, \
                               bias, act, strips, total, gate, y, pooled, inv_hw);                                   \
        else                                                                                                         \
            hipLaunchKernelGGL((k_dwconv3x3_rows<TW, WRITE_, POOL_, GATE_, false>), grid, block, 0, s, x, H, W, C, w, \
                               bias, act, strips, total, gate, y, pooled, inv_hw);                                   \
    } while (0)
        if (!y) ISC_DW_ROWS(false, true, false);
        else if (pooled && gate) ISC_DW_ROWS(true, true, true);
        else if (pooled) ISC_DW_ROWS(true, true, false);
        else if (gate) ISC_DW_ROWS(true, false, true);
        else ISC_DW_ROWS(true, false, false);
#undef ISC_DW_ROWS
        return isc_launch_status();
    }
    if (!y || gate) return ISC_ERR_UNSUPPORTED;  // the pooling alone and the gated output exist in the row-sweep kernel only
    const int st = isc_dwconv2d_nhwc(x, B, H, W, C, w, R, stride, pad, bias, act, y, stream);
    if (st != ISC_OK || !pooled) return st;
    const int Ho = (H + 2 * pad - R) / stride + 1;
    const int Wo = (W + 2 * pad - R) / stride + 1;
    return isc_global_avgpool_nhwc(y, B, Ho, Wo, C, pooled, stream);
}

extern "C" int isc_se_gate(const float* pooled, int B, int C, const float* w1, int ld1, const float* b1, int S,
                           const float* w2, int ld2, const float* b2, float* gate, void* stream) {
    ISC_REQUIRE(pooled && w1 && w2 && gate && B > 0 && C > 0 && S > 0 && ld1 >= C && ld2 >= S);
    if (C % 4 != 0 || S % 4 != 0 || ld1 % 4 != 0 || ld2 % 4 != 0) return ISC_ERR_UNSUPPORTED;
    if (!isc_aligned(pooled, 16) || !isc_aligned(w1, 16) || !isc_aligned(w2, 16) || !isc_aligned(gate, 16) ||
        (b2 && !isc_aligned(b2, 16)))
        return ISC_ERR_ALIGNMENT;
    constexpr int IMG = 2;  // measured: 1 -> 54 us, 2 -> 45 us, 4 -> 72 us per launch at C = 1536, S = 64, B = 512
    const size_t lds = (size_t)IMG * ((size_t)C + S) * sizeof(float);
    if (lds > 64 * 1024 || S > 256) return ISC_ERR_UNSUPPORTED;  // C + S <= 16384, S <= 256
    int lpr = 1;
    while (lpr < S / 4) lpr *= 2;  // lanes along a w2 row
#ifdef ISC_ABLATION
    static const int se_abl = getenv("ISC_SE_ABL") ? atoi(getenv("ISC_SE_ABL")) : 0;
    lpr |= se_abl << 16;
#endif
    hipLaunchKernelGGL((k_se_gate<IMG>), dim3((unsigned)isc_ceil_div(B, IMG)), dim3(SE_THREADS), lds, isc_stream(stream),
                       pooled, B, C, w1, ld1, b1, S, w2, ld2, b2, lpr, gate);
    return isc_launch_status();
}

extern "C" int isc_pool_linear_l2norm(const float* x, int B, int H, int W, int C, const float* w, const float* bias, int E,
                                      int normalize, float eps, float* out, void* stream) {
    ISC_REQUIRE(x && w && out && B > 0 && H > 0 && W > 0 && C > 0 && E > 0);
    if (C % 4 != 0) return ISC_ERR_UNSUPPORTED;
    if (!isc_aligned(w, 16)) return ISC_ERR_ALIGNMENT;
    const size_t lds = (size_t)HEAD_IMG * ((size_t)C + E) * sizeof(float);
    if (lds > 64 * 1024) return ISC_ERR_UNSUPPORTED;  // C + E <= 8192
    hipLaunchKernelGGL(k_pool_linear_l2norm, dim3(isc_ceil_div(B, HEAD_IMG)), dim3(HEAD_THREADS), lds, isc_stream(stream), x,
                       B, H * W, C, w, bias, E, normalize, eps, out);
    return isc_launch_status();
}

extern "C" int isc_global_avgpool_nhwc(const float* x, int B, int H, int W, int C, float* y, void* stream) {
    ISC_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0);
    if (B > 65535) return ISC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_global_avgpool_nhwc, dim3(isc_ceil_div(C, 256), B), dim3(256), 0, isc_stream(stream), x, H * W, C,
                       y);
    return isc_launch_status();
}
