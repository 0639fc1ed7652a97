// Transformer-encoder blocks in float16 on the matrix cores (include/imagescry_hip.h: isc_gemm_f16, isc_layernorm,
// isc_attention_f16, isc_patchify_f16, isc_vit_assemble).  Used by the ViT-B/16 embedder (BASELINE.json configs[4]).
//
// Numerics: every product is fp16 x fp16 accumulated in float32 (v_mfma_f32_16x16x32_f16); the residual stream,
// LayerNorm statistics, softmax statistics, bias and GELU (erf form, erf from a 1.5e-7 polynomial) are float32.  Only
// GEMM operands are rounded to fp16.
#include <stdlib.h>

#include "isc_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

// Direct global -> LDS copy (LDS-DMA): lane l of the wave writes 16 bytes at lds_wave_base + 16 * l.
__device__ __forceinline__ void g2_dma16(const unsigned char* gsrc, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// GELU with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, below float32 rounding of the product for |v| < 4):
// a third of the instructions of erff(), which matters where 16 values per lane pass through it in an epilogue.
__device__ __forceinline__ float gelu_fast(float v) {
    const float x = fabsf(v) * 0.70710678118654752440f;
    const float t = __frcp_rn(fmaf(0.3275911f, x, 1.f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = 1.f - poly * t * __expf(-x * x);  // erf(|v| / sqrt 2)
    return 0.5f * v + 0.5f * fabsf(v) * e;             // 0.5 v (1 + sign(v) erf(|v| / sqrt 2))
}

// ---------------------------------------------------------------------------------------------------------
// out[m][n] = act( sum_k a[m][k] * w[n][k] + bias[n] ) + residual[m][n]
//
// The weight tile is the MFMA "A" operand (rows = output features) and the token tile the "B" operand (columns =
// tokens): a lane then owns four CONSECUTIVE output features of one token, i.e. one 8-byte (fp16) or 16-byte (f32)
// store, and bias / residual are 16-byte loads.  One K step is 64 halves (128 bytes) of every row; LDS tiles are
// [rows][128 B] with the 16-byte chunks XOR-swizzled by (row >> 1) & 7, double buffered, one barrier per K step,
// the next step prefetched into registers while the current one is on the matrix cores.
struct GemmParams {
    const _Float16* a;
    const _Float16* w;
    const float* bias;
    const float* res;
    void* out;
    long long M;
    int N, K, ksteps, act, out_f32;
    int a_packed, w_packed, out_packed;  // operand / fp16 output in the K-step-major tile layout (see pk_offset)
};

// The PACKED fp16 matrix layout (the embedding bank's layout, bank_layout.h): rows in tiles of 256, columns in K steps of
// 64 halves, stored [tile][K step][row][64 halves].  The 256 x 128 B block one K step of one tile needs is 32 KiB of
// contiguous memory and every 1 KiB LDS-DMA instruction reads one contiguous KiB; a head's 64 q / k / v values of a token
// are exactly one 128-byte segment, consecutive tokens 128 bytes apart, which is what the attention kernel reads.
// (Measured on the ViT-B shapes: the GEMMs run at the same speed from row-major operands -- their cost is the
// epilogue traffic and the staging volume, not the access pattern.)
__host__ __device__ __forceinline__ size_t pk_offset(long long row, int col, int cols) {
    return (((size_t)(row >> 8) * (size_t)(cols >> 6) + (size_t)(col >> 6)) * 256 + (size_t)(row & 255)) * 64 + (size_t)(col & 63);
}
// byte offset of the first K step of `row`, and the byte distance between consecutive K steps of a row
__device__ __forceinline__ size_t operand_row_bytes(long long row, int K, int packed) {
    return packed ? pk_offset(row, 0, K) * 2 : (size_t)row * (size_t)K * 2;
}
__device__ __forceinline__ int operand_kstride(int packed) { return packed ? 32768 : 128; }

// Workgroup id -> output tile, XCD-aware.  Workgroup i runs on XCD i % 8 and every XCD has its own 4 MiB L2, so the
// tiles are ordered band-major -- bands of `band` token tiles; inside a band the feature tile is the slow index and the
// token tile the fast one -- and that sequence is cut into 8 contiguous slices, one per XCD (grid = 8 * ceil(T / 8),
// surplus workgroups return).  A band's activation rows (<= ~2 MiB) then stay in ONE L2 while the weight tiles pass
// by once per band; with the plain (token tile, feature tile) order every XCD streams every activation tile and the
// whole weight matrix (3.5 - 4.7 MiB for ViT-B: more than an L2) over and over from the Infinity Cache.
template <int TM_, int TN_>
__device__ __forceinline__ bool gemm_tile(const GemmParams& p, long long& m0, int& n0) {
    const long long mt = (p.M + TM_ - 1) / TM_;
    const int nt = (p.N + TN_ - 1) / TN_;
    const long long total = mt * nt;
    const long long slice = (total + 7) / 8;
    const long long t = (long long)(blockIdx.x & 7) * slice + (blockIdx.x >> 3);
    if ((long long)(blockIdx.x >> 3) >= slice || t >= total) return false;
    int band = (int)((2 << 20) / ((long long)TM_ * p.K * 2));
    band = band < 1 ? 1 : band > 16 ? 16 : band;
    const long long per_band = (long long)band * nt;
    const long long b = t / per_band;
    const long long t_in = t - b * per_band;
    const long long left = mt - b * band;
    const int mb = (int)(left < band ? left : band);
    n0 = (int)(t_in / mb) * TN_;
    m0 = (b * band + t_in % mb) * TM_;
    return true;
}

// ---------------------------------------------------------------------------------------------------------
// 128 x 128 tile, four waves of 64 x 64, two workgroups per CU, operands staged by LDS-DMA
// (`global_load_lds_dwordx4` writes the LDS directly; a register-staged `ds_write_b128` costs 13 LDS cycles per wave
// instruction -- MI355X_MICROARCH.md "LDS" -- about what the matrix cores need for the same K step).
// Two stages of 32 KiB per workgroup: iteration s issues the DMA of step s + 1, computes step s from fragments read
// with inline-asm `ds_read_b128` (a C++ LDS load would make hipcc drain the DMA first), then vmcnt(0) + one barrier;
// the second workgroup on the CU covers the wait and the other's epilogue.
#define G1_DS_READ(dst_, addr_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "i"(off_))

// DBG (ISC_GEMM_DEBUG, bring-up only, wrong results): 1 = no DMA after the prologue, 2 = no fragment reads, 3 = no MFMAs,
// 4 = no epilogue (nothing stored).
template <int DBG>
__global__ __launch_bounds__(256, 2) void k_gemm_f16_dma(const GemmParams p) {
    constexpr int STAGE = 32768;  // [weights 128 rows | activations 128 rows] x 128 B
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1;
    const int wm = wave >> 1;
    long long m0;
    int n0;
    if (!gemm_tile<128, 128>(p, m0, n0)) return;
    const int ksteps = p.ksteps;

    // LDS-DMA assignment: wave w copies rows [32 w, 32 w + 32) of both operand tiles, 8 rows (1 KiB) per instruction
    const unsigned char* wsrc[4];
    const unsigned char* xsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = wave * 32 + j * 8 + (lane >> 3);
        const int lcx = (lane & 7) ^ ((r >> 1) & 7);
        // weight rows use their own swizzle, ((r >> 4) & 3) << 1 | (r >> 1) & 1: conflict-free for the PERMUTED
        // fragment rows read below (rows 16 g + 4 mi + q, not 16 mi + r)
        const int lcw = (lane & 7) ^ ((((r >> 4) & 3) << 1) | ((r >> 1) & 1));
        const long long xr = m0 + r < p.M ? m0 + r : p.M - 1;
        wsrc[j] = reinterpret_cast<const unsigned char*>(p.w) + operand_row_bytes(min(n0 + r, p.N - 1), p.K, p.w_packed) + lcw * 16;
        xsrc[j] = reinterpret_cast<const unsigned char*>(p.a) + operand_row_bytes(xr, p.K, p.a_packed) + lcx * 16;
    }
    const int w_kstride = operand_kstride(p.w_packed), x_kstride = operand_kstride(p.a_packed);
    unsigned char* dma_dst = lds + wave * 4096;  // + stage, + 16 KiB for the activation half, + 1024 j

    auto issue = [&](int ks, int stage) {
        unsigned char* d = dma_dst + stage * STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) g2_dma16(wsrc[j] + (size_t)ks * w_kstride, d + j * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) g2_dma16(xsrc[j] + (size_t)ks * x_kstride, d + 16384 + j * 1024);
    };

    const int frow = lane & 15;
    const int fg = lane >> 4;
    const int fsw = (lane >> 1) & 7;
    const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    // Weight fragment rows are PERMUTED: MFMA row r of block mi is feature 16 (r >> 2) + 4 mi + (r & 3) of the wave's 64.
    // The result registers of a lane (rows 4 fg + q of every block) are then features 16 fg + 4 mi + q: SIXTEEN
    // consecutive features of one token instead of four groups of four, so the epilogue stores 32 (fp16) or 64 (f32)
    // contiguous bytes per lane and a token's four lanes fill whole 128-byte lines.
    const int arow = 16 * (frow >> 2) + (frow & 3);  // + 4 mi
    const int asw = ((frow >> 2) << 1) | ((frow >> 1) & 1);
    const unsigned a_frag0 = lds_addr + wn * 8192 + arow * 128 + (((0 + fg) ^ asw) << 4);
    const unsigned a_frag1 = lds_addr + wn * 8192 + arow * 128 + (((4 + fg) ^ asw) << 4);
    const unsigned b_frag0 = lds_addr + 16384 + wm * 8192 + frow * 128 + (((0 + fg) ^ fsw) << 4);
    const unsigned b_frag1 = lds_addr + 16384 + wm * 8192 + frow * 128 + (((4 + fg) ^ fsw) << 4);

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    for (int s = 0; s < ksteps; ++s) {
        const unsigned st = (unsigned)(s & 1) * STAGE;
        if (DBG != 1 && s + 1 < ksteps) issue(s + 1, (s + 1) & 1);
        u32x4 a0[4], b0[4], a1[4], b1[4];
        const unsigned aa0 = a_frag0 + st, bb0 = b_frag0 + st, aa1 = a_frag1 + st, bb1 = b_frag1 + st;
        if (DBG == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a0[i] = b0[i] = a1[i] = b1[i] = u32x4{(unsigned)s, 1u, (unsigned)lane, 3u};
        } else {
        G1_DS_READ(a0[0], aa0, 0);
        G1_DS_READ(a0[1], aa0, 512);
        G1_DS_READ(a0[2], aa0, 1024);
        G1_DS_READ(a0[3], aa0, 1536);
        G1_DS_READ(b0[0], bb0, 0);
        G1_DS_READ(b0[1], bb0, 2048);
        G1_DS_READ(b0[2], bb0, 4096);
        G1_DS_READ(b0[3], bb0, 6144);
        G1_DS_READ(a1[0], aa1, 0);
        G1_DS_READ(a1[1], aa1, 512);
        G1_DS_READ(a1[2], aa1, 1024);
        G1_DS_READ(a1[3], aa1, 1536);
        G1_DS_READ(b1[0], bb1, 0);
        G1_DS_READ(b1[1], bb1, 2048);
        G1_DS_READ(b1[2], bb1, 4096);
        G1_DS_READ(b1[3], bb1, 6144);
        }
        asm volatile("s_waitcnt lgkmcnt(8)"
                     : "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]), "+v"(b0[0]), "+v"(b0[1]), "+v"(b0[2]), "+v"(b0[3]));
        __builtin_amdgcn_sched_barrier(0);
        if (DBG == 3) {
            acc[0][0][0] += __uint_as_float(a0[0][0] ^ b0[1][1] ^ a1[2][2] ^ b1[3][3]);
        } else {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a0[mi]),
                                                                     __builtin_bit_cast(half8, b0[ni]), acc[mi][ni], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a1[0]), "+v"(a1[1]), "+v"(a1[2]), "+v"(a1[3]), "+v"(b1[0]), "+v"(b1[1]), "+v"(b1[2]), "+v"(b1[3]));
        __builtin_amdgcn_sched_barrier(0);
        if (DBG != 3) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a1[mi]),
                                                                     __builtin_bit_cast(half8, b1[ni]), acc[mi][ni], 0, 0, 0);
        }
        // this wave's DMA of step s + 1 has landed; the barrier publishes every wave's pieces and guarantees that nobody
        // still reads the stage the next iteration refills
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    if (DBG == 4) {  // keep the accumulators alive without the store tail
        float t = 0.f;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) t += acc[mi][ni][0] + acc[mi][ni][1] + acc[mi][ni][2] + acc[mi][ni][3];
        if (t == 123.456f) reinterpret_cast<float*>(p.out)[0] = t;
        return;
    }
    // ---- epilogue: this lane owns features [nb, nb + 16) of tokens m0 + 64 wm + 16 ni + frow
    const int nb = n0 + wn * 64 + fg * 16;
    f32x4 bias[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        bias[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias && nb + 4 * mi < p.N) bias[mi] = *reinterpret_cast<const f32x4*>(p.bias + nb + 4 * mi);
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const long long m = m0 + wm * 64 + ni * 16 + frow;
        if (m >= p.M) continue;
        const size_t o = (size_t)m * p.N + nb;
        f32x4 r[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            r[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.res && nb + 4 * mi < p.N) r[mi] = *reinterpret_cast<const f32x4*>(p.res + o + 4 * mi);
        }
        f32x4 v[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            v[mi] = acc[mi][ni] + bias[mi];
            if (p.act == ISC_ACT_GELU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[mi][q] = gelu_fast(v[mi][q]);
            }
            v[mi] += r[mi];
        }
        if (p.out_f32) {
            float* dst = reinterpret_cast<float*>(p.out) + o;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                if (nb + 4 * mi < p.N) *reinterpret_cast<f32x4*>(dst + 4 * mi) = v[mi];
        } else {
            _Float16* dst = reinterpret_cast<_Float16*>(p.out) + (p.out_packed ? pk_offset(m, nb, p.N) : o);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 lo = v[2 * h], hi = v[2 * h + 1];
                if (nb + 8 * h + 8 <= p.N) {
                    *reinterpret_cast<half8*>(dst + 8 * h) = half8{(_Float16)lo[0], (_Float16)lo[1], (_Float16)lo[2], (_Float16)lo[3],
                                                                   (_Float16)hi[0], (_Float16)hi[1], (_Float16)hi[2], (_Float16)hi[3]};
                } else if (nb + 8 * h + 4 <= p.N) {
                    *reinterpret_cast<half4*>(dst + 8 * h) = half4{(_Float16)lo[0], (_Float16)lo[1], (_Float16)lo[2], (_Float16)lo[3]};
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// The large-problem variant: 256 x 256 tile, FOUR waves (one per SIMD), each owning a 128 x 128 sub-tile.
//
// Why.  With 64 x 64 (or 128 x 64) wave tiles every MFMA needs 0.5 (0.375) fragment reads and the LDS -- 128 B/clk
// per CU -- has to run at 100 % duty for the matrix pipe to do so (the co-limit measured on k_dots_filter).  A
// 128 x 128 wave tile needs 16 + 16 reads per 128 MFMAs: 64 B/clk of reads plus 32 B/clk of DMA writes = 75 % duty.
// The price is 256 accumulator registers (AGPRs), i.e. one wave per SIMD, so nothing but this wave's own instruction
// stream can hide latency: fragment reads for the NEXT half K step and the LDS-DMA of LATER K steps are issued
// between the MFMAs of the current half step, four MFMAs per slot.
//
// LDS = a ring of 2 weight stages and 3 activation stages of 32 KiB (all 160 KiB).  Weights are L2 resident
// (prefetch distance 1 K step), activations stream from HBM (distance 2).  One barrier per K step, placed BETWEEN
// the two half steps: at that point every wave holds the second half's fragments in registers, so the stages of
// step s are free and those of step s + 1 have landed.
//
//   step s:  [64 MFMAs on half 0 | reads of half 1]  wait, barrier  [64 MFMAs on half 1 | reads of step s+1 half 0,
//                                                                    DMA of weights s+2 and activations s+3]
#define G2_DS_READ(dst_, addr_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "i"(off_))


// DBG is a bring-up aid (ISC_GEMM_DEBUG, never set in production; results are wrong for DBG != 0): 1 = no LDS-DMA after
// the prologue, 2 = no fragment reads in the loop, 3 = no MFMAs.
template <int DBG>
__global__ __launch_bounds__(256, 1) void k_gemm_f16_big(const GemmParams p) {
    constexpr int STAGE = 32768;
    __shared__ __attribute__((aligned(16))) unsigned char lds[5 * STAGE];  // [W0 W1 | X0 X1 X2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1;
    const int wm = wave >> 1;
    long long m0;
    int n0;
    if (!gemm_tile<256, 256>(p, m0, n0)) return;
    const int ksteps = p.ksteps;

    // ---- LDS-DMA assignment: wave w copies rows [64 w, 64 w + 64) of both operand tiles, 8 rows (1 KiB) per instruction
    unsigned woff[8], xoff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = wave * 64 + j * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ ((r >> 1) & 7);
        const int wr = min(n0 + r, p.N - 1);
        const long long xr = m0 + r < p.M ? m0 + r : p.M - 1;
        woff[j] = (unsigned)operand_row_bytes(wr, p.K, p.w_packed) + lc * 16;
        xoff[j] = (unsigned)operand_row_bytes(xr, p.K, p.a_packed) + lc * 16;
    }
    const int w_kstride = operand_kstride(p.w_packed), x_kstride = operand_kstride(p.a_packed);
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.w);
    const unsigned char* xsrc = reinterpret_cast<const unsigned char*>(p.a);
    unsigned char* dma_dst = lds + wave * 8192;  // + stage base + 1024 j

    // ---- fragment addresses
    const int frow = lane & 15;
    const int fg = lane >> 4;
    const int fsw = (lane >> 1) & 7;
    const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned a_frag0 = lds_addr + wn * 16384 + frow * 128 + (((0 + fg) ^ fsw) << 4);
    const unsigned a_frag1 = lds_addr + wn * 16384 + frow * 128 + (((4 + fg) ^ fsw) << 4);
    const unsigned b_frag0 = lds_addr + 2 * STAGE + wm * 16384 + frow * 128 + (((0 + fg) ^ fsw) << 4);
    const unsigned b_frag1 = lds_addr + 2 * STAGE + wm * 16384 + frow * 128 + (((4 + fg) ^ fsw) << 4);

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#define G2_DMA_W(ks_, stage_, j_) \
    g2_dma16(wsrc + woff[j_] + (size_t)(ks_) * w_kstride, dma_dst + (stage_) * STAGE + (j_) * 1024)
#define G2_DMA_X(ks_, stage_, j_) \
    g2_dma16(xsrc + xoff[j_] + (size_t)(ks_) * x_kstride, dma_dst + (2 + (stage_)) * STAGE + (j_) * 1024)

    // ---- prologue.  DMA stream order (the counted vmcnt below relies on it): W0 X0 | X1 W1 | X2
#pragma unroll
    for (int j = 0; j < 8; ++j) G2_DMA_W(0, 0, j);
#pragma unroll
    for (int j = 0; j < 8; ++j) G2_DMA_X(0, 0, j);
#pragma unroll
    for (int j = 0; j < 8; ++j) G2_DMA_X(1, 1, j);
#pragma unroll
    for (int j = 0; j < 8; ++j) G2_DMA_W(1, 1, j);
#pragma unroll
    for (int j = 0; j < 8; ++j) G2_DMA_X(2, 2, j);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    u32x4 a0[8], b0[8], a1[8], b1[8];
    G2_DS_READ(a0[0], a_frag0, 0);
    G2_DS_READ(a0[1], a_frag0, 2048);
    G2_DS_READ(a0[2], a_frag0, 4096);
    G2_DS_READ(a0[3], a_frag0, 6144);
    G2_DS_READ(a0[4], a_frag0, 8192);
    G2_DS_READ(a0[5], a_frag0, 10240);
    G2_DS_READ(a0[6], a_frag0, 12288);
    G2_DS_READ(a0[7], a_frag0, 14336);
    G2_DS_READ(b0[0], b_frag0, 0);
    G2_DS_READ(b0[1], b_frag0, 2048);
    G2_DS_READ(b0[2], b_frag0, 4096);
    G2_DS_READ(b0[3], b_frag0, 6144);
    G2_DS_READ(b0[4], b_frag0, 8192);
    G2_DS_READ(b0[5], b_frag0, 10240);
    G2_DS_READ(b0[6], b_frag0, 12288);
    G2_DS_READ(b0[7], b_frag0, 14336);
#define G2_WAIT_LGKM0(a_, b_)                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                           \
                 : "+v"(a_[0]), "+v"(a_[1]), "+v"(a_[2]), "+v"(a_[3]), "+v"(a_[4]), "+v"(a_[5]), "+v"(a_[6]),     \
                   "+v"(a_[7]));                                                                                  \
    asm volatile(""                                                                                               \
                 : "+v"(b_[0]), "+v"(b_[1]), "+v"(b_[2]), "+v"(b_[3]), "+v"(b_[4]), "+v"(b_[5]), "+v"(b_[6]),     \
                   "+v"(b_[7]));                                                                                  \
    __builtin_amdgcn_sched_barrier(0);
    G2_WAIT_LGKM0(a0, b0)

#define G2_MFMA4(a_, b_, i_, jh_)                                                                                   \
    _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) acc[i_][(jh_)*4 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16( \
        __builtin_bit_cast(half8, a_[i_]), __builtin_bit_cast(half8, b_[(jh_)*4 + jj]), acc[i_][(jh_)*4 + jj], 0, 0, 0);

    int ws = 0, xs = 0;  // stages holding step s
    for (int s = 0; s < ksteps; ++s) {
        const unsigned wst = (unsigned)ws * STAGE, xst = (unsigned)xs * STAGE;
        const unsigned a_addr1 = a_frag1 + wst, b_addr1 = b_frag1 + xst;
        // ---- half 0 on the matrix cores; half 1's fragments on their way
#define G2_SLOT1(x_)                                                          \
    if (DBG != 2) {                                                           \
        if ((x_) < 8) { G2_DS_READ(a1[(x_)&7], a_addr1, ((x_)&7) * 2048); }    \
        else { G2_DS_READ(b1[(x_)&7], b_addr1, ((x_)&7) * 2048); }             \
    }                                                                         \
    if (DBG != 3) { G2_MFMA4(a0, b0, (x_) >> 1, (x_)&1) }                      \
    __builtin_amdgcn_sched_barrier(0);
        G2_SLOT1(0) G2_SLOT1(1) G2_SLOT1(2) G2_SLOT1(3) G2_SLOT1(4) G2_SLOT1(5) G2_SLOT1(6) G2_SLOT1(7)
        G2_SLOT1(8) G2_SLOT1(9) G2_SLOT1(10) G2_SLOT1(11) G2_SLOT1(12) G2_SLOT1(13) G2_SLOT1(14) G2_SLOT1(15)
#undef G2_SLOT1
        G2_WAIT_LGKM0(a1, b1)
        // everything of step s + 1 has to be in LDS; the newest DMA group (activations of step s + 2) may stay in flight
        if (s + 2 < ksteps) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);

        // ---- half 1 on the matrix cores; next step's half 0 fragments and the DMA of later steps on their way
        const int ws_n = ws ^ 1, xs_n = xs == 2 ? 0 : xs + 1;
        const unsigned a_addr0 = a_frag0 + (unsigned)ws_n * STAGE, b_addr0 = b_frag0 + (unsigned)xs_n * STAGE;
        const bool do_w = DBG != 1 && s + 2 < ksteps, do_x = DBG != 1 && s + 3 < ksteps;
#define G2_SLOT2(x_)                                                          \
    if ((x_) < 8) {                                                           \
        if (DBG != 2) { G2_DS_READ(a0[(x_)&7], a_addr0, ((x_)&7) * 2048); }    \
        if (do_w) G2_DMA_W(s + 2, ws, (x_)&7);                                 \
    } else {                                                                  \
        if (DBG != 2) { G2_DS_READ(b0[(x_)&7], b_addr0, ((x_)&7) * 2048); }    \
        if (do_x) G2_DMA_X(s + 3, xs, (x_)&7);                                 \
    }                                                                         \
    if (DBG != 3) { G2_MFMA4(a1, b1, (x_) >> 1, (x_)&1) }                      \
    __builtin_amdgcn_sched_barrier(0);
        G2_SLOT2(0) G2_SLOT2(1) G2_SLOT2(2) G2_SLOT2(3) G2_SLOT2(4) G2_SLOT2(5) G2_SLOT2(6) G2_SLOT2(7)
        G2_SLOT2(8) G2_SLOT2(9) G2_SLOT2(10) G2_SLOT2(11) G2_SLOT2(12) G2_SLOT2(13) G2_SLOT2(14) G2_SLOT2(15)
#undef G2_SLOT2
        G2_WAIT_LGKM0(a0, b0)
        ws = ws_n;
        xs = xs_n;
    }
#undef G2_MFMA4
#undef G2_WAIT_LGKM0
#undef G2_DMA_W
#undef G2_DMA_X

    // the bias is added here, not used to initialise the accumulators: a load in front of the DMA prologue would put
    // its vmcnt wait in the middle of it
    f32x4 bias[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = n0 + wn * 128 + i * 16 + fg * 4;
        bias[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias && n < p.N) bias[i] = *reinterpret_cast<const f32x4*>(p.bias + n);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const long long m = m0 + wm * 128 + j * 16 + frow;
        if (m >= p.M) continue;
        f32x4 r[8];  // the eight residual loads of this token go out together, then the eight stores
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = n0 + wn * 128 + i * 16 + fg * 4;
            r[i] = bias[i];
            if (p.res && n < p.N) r[i] += *reinterpret_cast<const f32x4*>(p.res + (size_t)m * p.N + n);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = n0 + wn * 128 + i * 16 + fg * 4;
            if (n >= p.N) continue;
            const f32x4 v = acc[i][j] + r[i];
            const size_t o = (size_t)m * p.N + n;
            if (p.out_f32) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + o) = v;
            } else {
                half4 h = half4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                *reinterpret_cast<half4*>(reinterpret_cast<_Float16*>(p.out) + (p.out_packed ? pk_offset(m, n, p.N) : o)) = h;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm over the last axis, one wave per row: y = (x - mean) / sqrt(var + eps) * gamma + beta, biased variance,
// float32 statistics (two passes over registers).  D % 4 == 0, D <= 2048.
template <bool OUT_F32, int NV>
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, long long rows, int D, long long ldx,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   float eps, void* __restrict__ y, long long ldy, int y_packed) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + (size_t)row * ldx;
    const int nvec = D >> 2;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < nvec) v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * c);
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = isc_wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = v[i][r] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = 1.f / sqrtf(isc_wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c >= nvec) continue;
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + 4 * c);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (v[i][r] - mean) * rstd * g[r] + b[r];
        if (OUT_F32) {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(y) + (size_t)row * ldy + 4 * c) = o;
        } else {
            half4 h = half4{(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
            const size_t at = y_packed ? pk_offset(row, 4 * c, D) : (size_t)row * ldy + 4 * c;
            *reinterpret_cast<half4*>(reinterpret_cast<_Float16*>(y) + at) = h;
        }
    }
}

template <bool OUT_F32>
void launch_layernorm(int nv, dim3 grid, hipStream_t s, const float* x, long long rows, int D, long long ldx,
                      const float* gamma, const float* beta, float eps, void* y, long long ldy, int y_packed) {
    // NV = 16-byte vectors per lane: the smallest instantiation that covers D keeps the register count (and so the
    // number of rows in flight per CU) where this bandwidth-bound kernel needs it
    if (nv <= 1) hipLaunchKernelGGL((k_layernorm<OUT_F32, 1>), grid, dim3(256), 0, s, x, rows, D, ldx, gamma, beta, eps, y, ldy, y_packed);
    else if (nv <= 2) hipLaunchKernelGGL((k_layernorm<OUT_F32, 2>), grid, dim3(256), 0, s, x, rows, D, ldx, gamma, beta, eps, y, ldy, y_packed);
    else if (nv <= 3) hipLaunchKernelGGL((k_layernorm<OUT_F32, 3>), grid, dim3(256), 0, s, x, rows, D, ldx, gamma, beta, eps, y, ldy, y_packed);
    else if (nv <= 4) hipLaunchKernelGGL((k_layernorm<OUT_F32, 4>), grid, dim3(256), 0, s, x, rows, D, ldx, gamma, beta, eps, y, ldy, y_packed);
    else hipLaunchKernelGGL((k_layernorm<OUT_F32, 8>), grid, dim3(256), 0, s, x, rows, D, ldx, gamma, beta, eps, y, ldy, y_packed);
}

// ---------------------------------------------------------------------------------------------------------
// Multi-head self-attention for short sequences (T <= 224, head size 64), one workgroup per (image, head).
//
// qkv [B, T, 3 * D] fp16 with D = heads * 64: query / key / value of head h at columns h*64, D + h*64, 2D + h*64.
// out [B, T, D] fp16.
//
// Keys (row-major) and values (TRANSPOSED) of the head sit in LDS.  Each wave takes 16 queries at a time:
//   S^T[key][query] = K . Q^T / 8      14 key blocks x 2 MFMAs; a lane then holds ONE query (lane & 15) and 56 keys
//   softmax over keys                   registers + two cross-lane steps; statistics in float32
//   O^T[d][query]  = V^T . P^T         the contraction runs over keys, and the order in which a lane's 8 k-slots map
//                                       to keys is free as long as both operands agree: slot (g, j) <-> key
//                                       32 ks + 4 g + j (j < 4) or 32 ks + 16 + 4 g + (j - 4).  With that mapping the
//                                       P^T operand is exactly what the lane already holds after the softmax, so the
//                                       probabilities never leave registers; the values stay row-major in LDS and the
//                                       V^T fragments come out of two transposed reads (ds_read_b64_tr_b16) each.
constexpr int ATT_TMAX = 224;
constexpr int ATT_KSTRIDE = 72;   // halves per key row (64 + 8 pad)
constexpr int ATT_VSTRIDE = 72;   // halves per value row (row-major, like the keys; read transposed by ds_read_b64_tr_b16)
constexpr int ATT_THREADS = 512;
typedef short tr4 __attribute__((__vector_size__(4 * sizeof(short))));  // what ds_read_b64_tr_b16 returns: four 16-bit values
typedef short tr8 __attribute__((__vector_size__(8 * sizeof(short))));

// NKB = key blocks of 16 the kernel computes (T <= 16 NKB).  TAIL: T > 16 (NKB - 1), so only the LAST key block holds
// keys past T and only it is masked -- the shape of every ViT launch (T = 197: NKB = 13, the fourteenth block, which is
// all padding, is never computed); the generic form (NKB = 14, TAIL = false) masks every element and serves any T.
//
// Softmax arithmetic (round 4; the kernel is bound by its vector ALU work): the exponent is ONE fused
// multiply-add on the score -- exp(s - m) = exp2(s * log2(e) - m * log2(e)) with the second product hoisted out of the
// loop -- feeding v_exp_f32 directly, instead of a subtraction, a multiplication and the exponential; the masks shrink
// from two instructions per score to two per score of ONE block.  Per 16-query block: ~230 vector instructions instead
// of ~390 (28 K-fragment reads, 26 + 28 MFMAs).
// One block of 16 queries against the staged keys / values of a head: scores, softmax, P V, output row.  qf = this lane's two
// query fragments, already scaled by 1/8; orow = where query tq's 64 outputs go (tq < T), qi / g = lane & 15 / lane >> 4.
template <int NKB, bool TAIL>
__device__ __forceinline__ void att_query_block(const _Float16* __restrict__ Ks, const _Float16* __restrict__ Vs,
                                                const half8 (&qf)[2], int T, int tq, int qi, int g,
                                                _Float16* __restrict__ orow) {
    constexpr float LOG2E = 1.4426950408889634f;
    f32x4 s[NKB];
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const half8 kf = *reinterpret_cast<const half8*>(&Ks[(kb * 16 + qi) * ATT_KSTRIDE + (kk * 4 + g) * 8]);
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[kk], a, 0, 0, 0);
        }
        if (!TAIL || kb == NKB - 1) {  // (compile-time) the block(s) that can hold keys past T
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (kb * 16 + g * 4 + r >= T) a[r] = -INFINITY;
        }
        s[kb] = a;
        __builtin_amdgcn_sched_barrier(0);  // keep the fragment loads of later key blocks from piling up in VGPRs
    }
    // The row maximum, v_max3_f32 by hand: through fmaxf the compiler first canonicalises every matrix-core result (one
    // v_max_f32 x, x each -- 52 more instructions per query block than the 26 maxima themselves).  hipcc pads no
    // hazards for an asm statement (cdna_hip_programming.md 5.7), and a vector instruction that reads a register a
    // matrix-core instruction has just written needs its wait states: so the maxima run in ONE block behind the whole
    // score loop, fenced from it by a scheduling barrier, and the block opens with those wait states itself (12 >= the 11
    // an 8-pass MFMA result needs; the last MFMA's result is read by the mask code in between at the earliest).
    __builtin_amdgcn_sched_barrier(0);
    // six score blocks (twelve v_max3_f32) per statement: 25 register operands, under the limit of 30
#define ISC_ATT_MAX6(k_, PRE_)                                                                                         \
asm volatile(PRE_ "v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4\n\tv_max3_f32 %0, %0, %5, %6\n\t"         \
             "v_max3_f32 %0, %0, %7, %8\n\tv_max3_f32 %0, %0, %9, %10\n\tv_max3_f32 %0, %0, %11, %12\n\t"           \
             "v_max3_f32 %0, %0, %13, %14\n\tv_max3_f32 %0, %0, %15, %16\n\tv_max3_f32 %0, %0, %17, %18\n\t"        \
             "v_max3_f32 %0, %0, %19, %20\n\tv_max3_f32 %0, %0, %21, %22\n\tv_max3_f32 %0, %0, %23, %24"             \
             : "+v"(mx)                                                                                            \
             : "v"(s[k_][0]), "v"(s[k_][1]), "v"(s[k_][2]), "v"(s[k_][3]), "v"(s[k_ + 1][0]), "v"(s[k_ + 1][1]),   \
               "v"(s[k_ + 1][2]), "v"(s[k_ + 1][3]), "v"(s[k_ + 2][0]), "v"(s[k_ + 2][1]), "v"(s[k_ + 2][2]),      \
               "v"(s[k_ + 2][3]), "v"(s[k_ + 3][0]), "v"(s[k_ + 3][1]), "v"(s[k_ + 3][2]), "v"(s[k_ + 3][3]),      \
               "v"(s[k_ + 4][0]), "v"(s[k_ + 4][1]), "v"(s[k_ + 4][2]), "v"(s[k_ + 4][3]), "v"(s[k_ + 5][0]),      \
               "v"(s[k_ + 5][1]), "v"(s[k_ + 5][2]), "v"(s[k_ + 5][3]))
#define ISC_ATT_MAX1(k_)                                                                  \
asm volatile("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4"                  \
             : "+v"(mx)                                                               \
             : "v"(s[k_][0]), "v"(s[k_][1]), "v"(s[k_][2]), "v"(s[k_][3]))
    static_assert(NKB == 13 || NKB == 14, "the maxima are written out for thirteen or fourteen key blocks");
    ISC_ATT_MAX6(0, "s_nop 7\n\ts_nop 3\n\t");  // the wait states of the matrix-core results, inside the statement
    ISC_ATT_MAX6(6, "");
    ISC_ATT_MAX1(12);
    if constexpr (NKB == 14) ISC_ATT_MAX1(NKB - 1);
#undef ISC_ATT_MAX6
#undef ISC_ATT_MAX1
    __builtin_amdgcn_sched_barrier(0);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mxl = -mx * LOG2E;  // T >= 1: every query has a finite maximum
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(s[kb][r], LOG2E, mxl));  // masked keys: exp2(-inf) = 0
            s[kb][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;

    f32x4 o[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < (NKB + 1) / 2; ++ks) {
        half8 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pf[r] = (_Float16)s[2 * ks][r];
            pf[4 + r] = 2 * ks + 1 < NKB ? (_Float16)s[2 * ks + 1 < NKB ? 2 * ks + 1 : 0][r] : (_Float16)0.f;
        }
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            // V^T fragment: slot j of lane group g is key 32 ks + 4 g + j (j < 4) / 32 ks + 16 + 4 g + (j - 4), of value
            // column 16 db + qi.  One transposed read hands every lane of a 16-lane group ITS column of a 4-row block;
            // lane 4 q + p of the group supplies the address of row q, columns 4 p .. 4 p + 3 (EXEC is all ones here).
            const _Float16* vb = &Vs[(ks * 32 + g * 4 + (qi >> 2)) * ATT_VSTRIDE + db * 16 + (qi & 3) * 4];
            const tr4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr4*)vb);
            const tr4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr4*)(vb + 16 * ATT_VSTRIDE));
            const half8 vf = __builtin_bit_cast(half8, tr8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
            o[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[db], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (tq < T) {
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            half4 hv;
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[r] = (_Float16)(o[db][r] * inv);
            *reinterpret_cast<half4*>(orow + db * 16 + g * 4) = hv;
        }
    }
}

#ifdef ISC_ABLATION
__device__ int g_att_abl = 0;  // ISC_ATT_ABL (timing aid, wrong results): 1 = stage keys / values only, 2 = no staging loads
#endif
template <bool PACKED, int NKB, bool TAIL>
__global__ __launch_bounds__(ATT_THREADS) void k_attention_f16(const _Float16* __restrict__ qkv, int T, int heads,
                                                                _Float16* __restrict__ out) {
    static_assert(NKB >= 1 && NKB <= ATT_TMAX / 16, "key blocks");
    __shared__ __attribute__((aligned(16))) _Float16 Ks[ATT_TMAX * ATT_KSTRIDE];
    __shared__ __attribute__((aligned(16))) _Float16 Vs[ATT_TMAX * ATT_VSTRIDE];
    const int b = blockIdx.x / heads;
    const int h = blockIdx.x - b * heads;
    const int D = heads * 64;
    const size_t row_stride = (size_t)3 * D;
    const _Float16* base = qkv + (size_t)b * T * row_stride + h * 64;
    const int tid = threadIdx.x;
    // element address of (token t of this image, part 0/1/2 = q/k/v, 8-half chunk c of this head).  In the packed
    // layout a head's 64 values are exactly one 128-byte K-step segment of the token's row.
    auto qkv_at = [&](int t, int part, int c) -> const _Float16* {
        if (PACKED) return qkv + pk_offset((long long)b * T + t, part * D + h * 64, 3 * D) + c * 8;
        return base + (size_t)t * row_stride + part * D + c * 8;
    };

    constexpr int TROWS = NKB * 16 < ATT_TMAX ? (NKB + (NKB & 1)) * 16 : ATT_TMAX;  // key rows staged (whole MFMA k-steps of 32)
    // staging: ALL of a thread's key / value chunks are requested before the first one is waited for (as a rolled loop the
    // compiler waits for each pair of loads before it issues the next: four dependent HBM round trips per workgroup)
    constexpr int SROUNDS = (TROWS * 8 + ATT_THREADS - 1) / ATT_THREADS;
    half8 kreg[SROUNDS], vreg[SROUNDS];
#pragma unroll
    for (int r = 0; r < SROUNDS; ++r) {
        const int i = tid + r * ATT_THREADS;
        const int t = i >> 3, c = i & 7;
        kreg[r] = half8{0, 0, 0, 0, 0, 0, 0, 0};
        vreg[r] = kreg[r];
#ifdef ISC_ABLATION
        if (t < T && !(g_att_abl & 2)) {
#else
        if (t < T) {
#endif
            kreg[r] = *reinterpret_cast<const half8*>(qkv_at(t, 1, c));
            vreg[r] = *reinterpret_cast<const half8*>(qkv_at(t, 2, c));
        }
    }
#pragma unroll
    for (int r = 0; r < SROUNDS; ++r) {
        const int i = tid + r * ATT_THREADS;
        const int t = i >> 3, c = i & 7;
        if (i < TROWS * 8) {
            *reinterpret_cast<half8*>(&Ks[t * ATT_KSTRIDE + c * 8]) = kreg[r];
            // values stay ROW-major, one 16-byte LDS store per chunk like the keys: the transposition the V^T operand needs
            // is done by the reads (ds_read_b64_tr_b16).  Round 3 scattered eight 2-byte stores per chunk into a transposed
            // image -- with any 16-byte-aligned row stride all eight land in ONE bank (8 rows x stride = 0 mod 32 dwords).
            *reinterpret_cast<half8*>(&Vs[t * ATT_VSTRIDE + c * 8]) = vreg[r];
        }
    }
    __syncthreads();
#ifdef ISC_ABLATION
    if (g_att_abl & 1) return;
#endif

    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int qi = lane & 15;
    const int g = lane >> 4;
    const int nqb = (T + 15) >> 4;
    constexpr float LOG2E = 1.4426950408889634f;
    for (int qb = wave; qb < nqb; qb += ATT_THREADS / 64) {
        const int tq = qb * 16 + qi;
        half8 qf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            qf[kk] = *reinterpret_cast<const half8*>(qkv_at(min(tq, T - 1), 0, kk * 4 + g));
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[kk][j] = qf[kk][j] * (_Float16)0.125f;  // 1/sqrt(64): exact scaling
        }
        f32x4 s[NKB];
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const half8 kf = *reinterpret_cast<const half8*>(&Ks[(kb * 16 + qi) * ATT_KSTRIDE + (kk * 4 + g) * 8]);
                a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[kk], a, 0, 0, 0);
            }
            if (!TAIL || kb == NKB - 1) {  // (compile-time) the block(s) that can hold keys past T
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kb * 16 + g * 4 + r >= T) a[r] = -INFINITY;
            }
            s[kb] = a;
            __builtin_amdgcn_sched_barrier(0);  // keep the fragment loads of later key blocks from piling up in VGPRs
        }
        // The row maximum, v_max3_f32 by hand: through fmaxf the compiler first canonicalises every matrix-core result (one
        // v_max_f32 x, x each -- 52 more instructions per query block than the 26 maxima themselves).  hipcc pads no
        // hazards for an asm statement (cdna_hip_programming.md 5.7), and a vector instruction that reads a register a
        // matrix-core instruction has just written needs its wait states: so the maxima run in ONE block behind the whole
        // score loop, fenced from it by a scheduling barrier, and the block opens with those wait states itself (12 >= the 11
        // an 8-pass MFMA result needs; the last MFMA's result is read by the mask code in between at the earliest).
        __builtin_amdgcn_sched_barrier(0);
        // six score blocks (twelve v_max3_f32) per statement: 25 register operands, under the limit of 30
#define ISC_ATT_MAX6(k_, PRE_)                                                                                         \
    asm volatile(PRE_ "v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4\n\tv_max3_f32 %0, %0, %5, %6\n\t"         \
                 "v_max3_f32 %0, %0, %7, %8\n\tv_max3_f32 %0, %0, %9, %10\n\tv_max3_f32 %0, %0, %11, %12\n\t"           \
                 "v_max3_f32 %0, %0, %13, %14\n\tv_max3_f32 %0, %0, %15, %16\n\tv_max3_f32 %0, %0, %17, %18\n\t"        \
                 "v_max3_f32 %0, %0, %19, %20\n\tv_max3_f32 %0, %0, %21, %22\n\tv_max3_f32 %0, %0, %23, %24"             \
                 : "+v"(mx)                                                                                            \
                 : "v"(s[k_][0]), "v"(s[k_][1]), "v"(s[k_][2]), "v"(s[k_][3]), "v"(s[k_ + 1][0]), "v"(s[k_ + 1][1]),   \
                   "v"(s[k_ + 1][2]), "v"(s[k_ + 1][3]), "v"(s[k_ + 2][0]), "v"(s[k_ + 2][1]), "v"(s[k_ + 2][2]),      \
                   "v"(s[k_ + 2][3]), "v"(s[k_ + 3][0]), "v"(s[k_ + 3][1]), "v"(s[k_ + 3][2]), "v"(s[k_ + 3][3]),      \
                   "v"(s[k_ + 4][0]), "v"(s[k_ + 4][1]), "v"(s[k_ + 4][2]), "v"(s[k_ + 4][3]), "v"(s[k_ + 5][0]),      \
                   "v"(s[k_ + 5][1]), "v"(s[k_ + 5][2]), "v"(s[k_ + 5][3]))
#define ISC_ATT_MAX1(k_)                                                                  \
    asm volatile("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4"                  \
                 : "+v"(mx)                                                               \
                 : "v"(s[k_][0]), "v"(s[k_][1]), "v"(s[k_][2]), "v"(s[k_][3]))
        static_assert(NKB == 13 || NKB == 14, "the maxima are written out for thirteen or fourteen key blocks");
        ISC_ATT_MAX6(0, "s_nop 7\n\ts_nop 3\n\t");  // the wait states of the matrix-core results, inside the statement
        ISC_ATT_MAX6(6, "");
        ISC_ATT_MAX1(12);
        if constexpr (NKB == 14) ISC_ATT_MAX1(NKB - 1);
#undef ISC_ATT_MAX6
#undef ISC_ATT_MAX1
        __builtin_amdgcn_sched_barrier(0);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mxl = -mx * LOG2E;  // T >= 1: every query has a finite maximum
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(s[kb][r], LOG2E, mxl));  // masked keys: exp2(-inf) = 0
                s[kb][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;

        f32x4 o[4];
#pragma unroll
        for (int db = 0; db < 4; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < (NKB + 1) / 2; ++ks) {
            half8 pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[r] = (_Float16)s[2 * ks][r];
                pf[4 + r] = 2 * ks + 1 < NKB ? (_Float16)s[2 * ks + 1 < NKB ? 2 * ks + 1 : 0][r] : (_Float16)0.f;
            }
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                // V^T fragment: slot j of lane group g is key 32 ks + 4 g + j (j < 4) / 32 ks + 16 + 4 g + (j - 4), of value
                // column 16 db + qi.  One transposed read hands every lane of a 16-lane group ITS column of a 4-row block;
                // lane 4 q + p of the group supplies the address of row q, columns 4 p .. 4 p + 3 (EXEC is all ones here).
                const _Float16* vb = &Vs[(ks * 32 + g * 4 + (qi >> 2)) * ATT_VSTRIDE + db * 16 + (qi & 3) * 4];
                const tr4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr4*)vb);
                const tr4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr4*)(vb + 16 * ATT_VSTRIDE));
                const half8 vf = __builtin_bit_cast(half8, tr8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
                o[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[db], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (tq < T) {
            _Float16* orow = PACKED ? out + pk_offset((long long)b * T + tq, h * 64, D)
                                    : out + ((size_t)b * T + tq) * D + h * 64;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                half4 hv;
#pragma unroll
                for (int r = 0; r < 4; ++r) hv[r] = (_Float16)(o[db][r] * inv);
                *reinterpret_cast<half4*>(orow + db * 16 + g * 4) = hv;
            }
        }
    }
}

// The PERSISTENT form (round 4): one workgroup of SIXTEEN waves per CU walks the (image, head) pairs blockIdx.x,
// + gridDim.x, ...  A head's thirteen (fourteen) query blocks run as ONE round, a wave each, and the keys / values of
// the NEXT head are requested into registers before that round starts and written into the other LDS image after it --
// in k_attention_f16 a workgroup loads, then computes, and the second workgroup of the CU does not fill the gap: the
// two phases add up (staging alone 49 us of a 178 us layer).  Same staging layout, same query-block
// arithmetic (att_query_block), bit-identical results.
constexpr int ATT_P_THREADS = 1024;
constexpr int ATT_P_LDS_BYTES = 2 * ATT_TMAX * (ATT_KSTRIDE + ATT_VSTRIDE) * 2;  // two (K, V) images: 126 KiB
template <bool PACKED, int NKB, bool TAIL>
__global__ __launch_bounds__(ATT_P_THREADS) void k_attention_f16_p(const _Float16* __restrict__ qkv, int T, int heads,
                                                                    _Float16* __restrict__ out, int total) {
    extern __shared__ __attribute__((aligned(16))) unsigned char att_lds[];
    constexpr int IMG = ATT_TMAX * (ATT_KSTRIDE + ATT_VSTRIDE);  // halves per (K, V) image
    _Float16* const kv = reinterpret_cast<_Float16*>(att_lds);
    const int D = heads * 64;
    const size_t row_stride = (size_t)3 * D;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int qi = lane & 15;
    const int g = lane >> 4;
    const int nqb = (T + 15) >> 4;  // <= 14 waves have a query block
    const int tq = wave * 16 + qi;
    auto qkv_at = [&](int b, int h, int t, int part, int c) -> const _Float16* {
        if (PACKED) return qkv + pk_offset((long long)b * T + t, part * D + h * 64, 3 * D) + c * 8;
        return qkv + ((size_t)b * T + t) * row_stride + part * D + h * 64 + c * 8;
    };
    constexpr int TROWS = NKB * 16 < ATT_TMAX ? (NKB + (NKB & 1)) * 16 : ATT_TMAX;
    constexpr int SROUNDS = (TROWS * 8 + ATT_P_THREADS - 1) / ATT_P_THREADS;
    half8 kreg[SROUNDS], vreg[SROUNDS], qn[2];
    auto load_item = [&](int item) {
        const int b = item / heads, h = item - b * heads;
#pragma unroll
        for (int r = 0; r < SROUNDS; ++r) {
            const int i = tid + r * ATT_P_THREADS;
            const int t = i >> 3, c = i & 7;
            kreg[r] = half8{0, 0, 0, 0, 0, 0, 0, 0};
            vreg[r] = kreg[r];
            if (t < T) {
                kreg[r] = *reinterpret_cast<const half8*>(qkv_at(b, h, t, 1, c));
                vreg[r] = *reinterpret_cast<const half8*>(qkv_at(b, h, t, 2, c));
            }
        }
        if (wave < nqb) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) qn[kk] = *reinterpret_cast<const half8*>(qkv_at(b, h, min(tq, T - 1), 0, kk * 4 + g));
        }
    };
    auto store_item = [&](int buf) {
        _Float16* Ks = kv + buf * IMG;
        _Float16* Vs = Ks + ATT_TMAX * ATT_KSTRIDE;
#pragma unroll
        for (int r = 0; r < SROUNDS; ++r) {
            const int i = tid + r * ATT_P_THREADS;
            const int t = i >> 3, c = i & 7;
            if (i < TROWS * 8) {
                *reinterpret_cast<half8*>(&Ks[t * ATT_KSTRIDE + c * 8]) = kreg[r];
                *reinterpret_cast<half8*>(&Vs[t * ATT_VSTRIDE + c * 8]) = vreg[r];
            }
        }
    };
    int item = blockIdx.x;
    if (item >= total) return;
    load_item(item);
    store_item(0);
    half8 qf[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[kk][j] = qn[kk][j] * (_Float16)0.125f;  // 1/sqrt(64): exact scaling
    __syncthreads();
    for (int it = 0;; ++it) {
        const int next = item + (int)gridDim.x;
        const bool more = next < total;
        if (more) load_item(next);  // in flight under this head's round
        if (wave < nqb) {
            const int b = item / heads, h = item - b * heads;
            const _Float16* Ks = kv + (it & 1) * IMG;
            _Float16* orow = PACKED ? out + pk_offset((long long)b * T + min(tq, T - 1), h * 64, D)
                                    : out + ((size_t)b * T + min(tq, T - 1)) * D + h * 64;
            att_query_block<NKB, TAIL>(Ks, Ks + ATT_TMAX * ATT_KSTRIDE, qf, T, tq, qi, g, orow);
        }
        if (!more) break;
        store_item((it + 1) & 1);  // the image the PREVIOUS head was read from: every wave is past that round's barrier
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[kk][j] = qn[kk][j] * (_Float16)0.125f;
        __syncthreads();
        item = next;
    }
}

// ---------------------------------------------------------------------------------------------------------
// patches[b * gh * gw + (ph * gw + pw)][c * P * P + r * P + s] = fp16(x[b][c][ph * P + r][pw * P + s]); 8 values a thread
__global__ __launch_bounds__(256) void k_patchify_f16(const float* __restrict__ x, int C, int H, int W, int P,
                                                      size_t total8, _Float16* __restrict__ y, int packed) {
    const int gw = W / P, gh = H / P;
    const int pv = P / 8;
    const int kdim = C * P * P;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (size_t)gridDim.x * 256) {
        // i enumerates (b, c, row, 8-pixel group) in input order, so loads are coalesced
        const int wg = (int)(i % (W / 8));
        size_t rest = i / (W / 8);
        const int hrow = (int)(rest % H);
        rest /= H;
        const int c = (int)(rest % C);
        const size_t b = rest / C;
        const int ph = hrow / P, r = hrow - ph * P;
        const int pw = wg / pv, s = (wg - pw * pv) * 8;
        const float* src = x + ((b * C + c) * H + hrow) * (size_t)W + wg * 8;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + 4);
        half8 o = half8{(_Float16)v0[0], (_Float16)v0[1], (_Float16)v0[2], (_Float16)v0[3],
                        (_Float16)v1[0], (_Float16)v1[1], (_Float16)v1[2], (_Float16)v1[3]};
        const size_t m = (b * gh + ph) * gw + pw;
        const int k = (c * P + r) * P + s;
        *reinterpret_cast<half8*>(y + (packed ? pk_offset((long long)m, k, kdim) : m * kdim + k)) = o;
    }
}

// tokens[b][0] = cls + pos[0];  tokens[b][t] = patch_embed[b * (T - 1) + t - 1] + pos[t]
__global__ __launch_bounds__(256) void k_vit_assemble(const float* __restrict__ pe, const float* __restrict__ cls,
                                                      const float* __restrict__ pos, int T, int D, size_t total4,
                                                      float* __restrict__ y) {
    const int dv = D / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % dv);
        const size_t bt = i / dv;
        const int t = (int)(bt % T);
        const size_t b = bt / T;
        f32x4 v = t == 0 ? *reinterpret_cast<const f32x4*>(cls + 4 * c)
                         : *reinterpret_cast<const f32x4*>(pe + (b * (T - 1) + t - 1) * (size_t)D + 4 * c);
        v += *reinterpret_cast<const f32x4*>(pos + (size_t)t * D + 4 * c);
        *reinterpret_cast<f32x4*>(y + i * 4) = v;
    }
}

}  // namespace
int isc_gemm_f16_stream_launch(const void* a, long long M, int K, const void* w, int N, const float* bias,
                               const float* residual, int epi, void* out, int out_packed, hipStream_t stream);
namespace {

int grid_for(size_t work_items) {
    const size_t blocks = isc_ceil_div<size_t>(work_items, 256);
    return (int)(blocks < 16384 ? (blocks ? blocks : 1) : 16384);
}

}  // namespace

extern "C" int isc_gemm_f16(const void* a, int64_t M, int K, const void* w, int N, const float* bias,
                            const float* residual, int act, void* out, int out_dtype, int flags, void* stream) {
    ISC_REQUIRE(a && w && out && M > 0 && K > 0 && N > 0);
    ISC_REQUIRE(act == ISC_ACT_NONE || act == ISC_ACT_GELU);
    ISC_REQUIRE(out_dtype == ISC_F16 || out_dtype == ISC_F32);
    ISC_REQUIRE((flags & ~(ISC_GEMM_A_PACKED | ISC_GEMM_W_PACKED | ISC_GEMM_OUT_PACKED | ISC_GEMM_TILE_256 |
                           ISC_GEMM_TILE_128)) == 0);
    if (K % 64 != 0 || N % 4 != 0) return ISC_ERR_UNSUPPORTED;
    if ((flags & ISC_GEMM_OUT_PACKED) && (out_dtype != ISC_F16 || N % 64 != 0)) return ISC_ERR_UNSUPPORTED;
    if (!isc_aligned(a, 16) || !isc_aligned(w, 16) || !isc_aligned(out, 16) || (bias && !isc_aligned(bias, 16)) ||
        (residual && !isc_aligned(residual, 16)))
        return ISC_ERR_ALIGNMENT;
    // Packed operands and a multiple of 256 output features (every GEMM of the ViT encoder): the streaming kernel of
    // gemm_stream.hip -- 256 x 256 tiles on the search kernel's LDS-DMA ring loop.  ISC_GEMM_TILE_128 / _256 select the
    // older kernels below explicitly.
    if ((flags & ISC_GEMM_A_PACKED) && (flags & ISC_GEMM_W_PACKED) && !(flags & (ISC_GEMM_TILE_128 | ISC_GEMM_TILE_256)) &&
        N % 256 == 0 && !(out_dtype == ISC_F32 && act != ISC_ACT_NONE) && !(out_dtype == ISC_F16 && residual)) {
        const int epi = out_dtype == ISC_F32 ? 2 : act == ISC_ACT_GELU ? 1 : 0;
        isc_timing_begin(ISC_KERNEL_GEMM_F16, isc_stream(stream));
        const int st = isc_gemm_f16_stream_launch(a, M, K, w, N, bias, residual, epi, out,
                                                  (flags & ISC_GEMM_OUT_PACKED) != 0, isc_stream(stream));
        isc_timing_end(ISC_KERNEL_GEMM_F16, isc_stream(stream));
        return st;
    }
    GemmParams p;
    p.a = reinterpret_cast<const _Float16*>(a);
    p.w = reinterpret_cast<const _Float16*>(w);
    p.bias = bias;
    p.res = residual;
    p.out = out;
    p.M = M;
    p.N = N;
    p.K = K;
    p.ksteps = K / 64;
    p.act = act;
    p.out_f32 = out_dtype == ISC_F32;
    p.a_packed = (flags & ISC_GEMM_A_PACKED) != 0;
    p.w_packed = (flags & ISC_GEMM_W_PACKED) != 0;
    p.out_packed = (flags & ISC_GEMM_OUT_PACKED) != 0;
    const long long tiles = isc_ceil_div<long long>(M, 128) * isc_ceil_div<long long>(N, 128);
    if (tiles > 0x7ffffff0LL) return ISC_ERR_UNSUPPORTED;
    hipStream_t s = isc_stream(stream);
    // ISC_GEMM_TILE_256 selects the 256 x 256 / one-wave-per-SIMD kernel explicitly (it measures the same as the
    // 128 x 128 kernel on the ViT shapes -- both are bound by operand staging, DESIGN.md 4.3 -- so nothing picks it
    // by default); its 32-bit DMA offsets bound the operand sizes and it has no activation epilogue
    const bool big = (flags & ISC_GEMM_TILE_256) != 0;
    if (big && (p.ksteps < 3 || act != ISC_ACT_NONE || (unsigned long long)(M + 255) * K * 2 >= (1ull << 32) ||
                (unsigned long long)(N + 255) * K * 2 >= (1ull << 32)))
        return ISC_ERR_UNSUPPORTED;
    isc_timing_begin(ISC_KERNEL_GEMM_F16, s);
    if (big) {
        const long long tiles2 = isc_ceil_div<long long>(M, 256) * isc_ceil_div<long long>(N, 256);
        const long long grid2 = isc_ceil_div<long long>(tiles2, 8) * 8;  // gemm_tile: 8 equal slices, one per XCD
#ifdef ISC_ABLATION  // wrong-result bring-up variants: -DISC_ABLATION builds only
        static const int dbg = [] {
            const char* e = getenv("ISC_GEMM_DEBUG");
            return e ? atoi(e) : 0;
        }();
        if (dbg == 1) hipLaunchKernelGGL(k_gemm_f16_big<1>, dim3((unsigned)grid2), dim3(256), 0, s, p);
        else if (dbg == 2) hipLaunchKernelGGL(k_gemm_f16_big<2>, dim3((unsigned)grid2), dim3(256), 0, s, p);
        else if (dbg == 3) hipLaunchKernelGGL(k_gemm_f16_big<3>, dim3((unsigned)grid2), dim3(256), 0, s, p);
        else
#endif
            hipLaunchKernelGGL(k_gemm_f16_big<0>, dim3((unsigned)grid2), dim3(256), 0, s, p);
    } else {
        const long long grid1 = isc_ceil_div<long long>(tiles, 8) * 8;
#ifdef ISC_ABLATION
        static const int dbg1 = [] {
            const char* e = getenv("ISC_GEMM_DEBUG");
            return e ? atoi(e) : 0;
        }();
        if (dbg1 == 1) hipLaunchKernelGGL(k_gemm_f16_dma<1>, dim3((unsigned)grid1), dim3(256), 0, s, p);
        else if (dbg1 == 2) hipLaunchKernelGGL(k_gemm_f16_dma<2>, dim3((unsigned)grid1), dim3(256), 0, s, p);
        else if (dbg1 == 3) hipLaunchKernelGGL(k_gemm_f16_dma<3>, dim3((unsigned)grid1), dim3(256), 0, s, p);
        else if (dbg1 == 4) hipLaunchKernelGGL(k_gemm_f16_dma<4>, dim3((unsigned)grid1), dim3(256), 0, s, p);
        else
#endif
            hipLaunchKernelGGL(k_gemm_f16_dma<0>, dim3((unsigned)grid1), dim3(256), 0, s, p);
    }
    isc_timing_end(ISC_KERNEL_GEMM_F16, s);
    return isc_launch_status();
}

extern "C" int isc_layernorm(const float* x, int64_t rows, int D, int64_t ldx, const float* gamma, const float* beta,
                             float eps, void* y, int y_dtype, int64_t ldy, int y_packed, void* stream) {
    ISC_REQUIRE(x && gamma && beta && y && rows > 0 && D > 0 && eps >= 0.f);
    ISC_REQUIRE(y_dtype == ISC_F16 || y_dtype == ISC_F32);
    if (D % 4 != 0 || D > 2048) return ISC_ERR_UNSUPPORTED;
    if (y_packed && (y_dtype != ISC_F16 || D % 64 != 0)) return ISC_ERR_UNSUPPORTED;
    if (ldx < D || ldy < D || ldx % 4 != 0 || ldy % 4 != 0) return ISC_ERR_ALIGNMENT;
    if (!isc_aligned(x, 16) || !isc_aligned(gamma, 16) || !isc_aligned(beta, 16) || !isc_aligned(y, 16))
        return ISC_ERR_ALIGNMENT;
    const long long blocks = isc_ceil_div<long long>(rows, 4);
    if (blocks > 0x7fffffffLL) return ISC_ERR_UNSUPPORTED;
    hipStream_t s = isc_stream(stream);
    const int nv = (D / 4 + 63) / 64;
    if (y_dtype == ISC_F32)
        launch_layernorm<true>(nv, dim3((unsigned)blocks), s, x, (long long)rows, D, (long long)ldx, gamma, beta, eps, y,
                               (long long)ldy, 0);
    else
        launch_layernorm<false>(nv, dim3((unsigned)blocks), s, x, (long long)rows, D, (long long)ldx, gamma, beta, eps, y,
                                (long long)ldy, y_packed ? 1 : 0);
    return isc_launch_status();
}

// the 126 KiB of dynamic LDS k_attention_f16_p asks for have to be allowed once per (instantiation, device)
static bool att_p_prepare(const void* fn, unsigned long long* done) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const bool tracked = dev >= 0 && dev < 64;
    if (tracked && ((__atomic_load_n(done, __ATOMIC_RELAXED) >> dev) & 1ull)) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_P_LDS_BYTES) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    if (tracked) __atomic_fetch_or(done, 1ull << dev, __ATOMIC_RELAXED);
    return true;
}

extern "C" int isc_attention_f16(const void* qkv, int B, int T, int heads, int head_dim, void* out, int packed,
                                 void* stream) {
    ISC_REQUIRE(qkv && out && B > 0 && T > 0 && heads > 0);
    if (head_dim != 64 || T > ATT_TMAX) return ISC_ERR_UNSUPPORTED;
    if (!isc_aligned(qkv, 16) || !isc_aligned(out, 16)) return ISC_ERR_ALIGNMENT;
    if ((long long)B * heads > 0x7fffffffLL) return ISC_ERR_UNSUPPORTED;
#ifdef ISC_ABLATION
    static const bool att_abl_set = [] {
        const int v = getenv("ISC_ATT_ABL") ? atoi(getenv("ISC_ATT_ABL")) : 0;
        return hipMemcpyToSymbol(HIP_SYMBOL(g_att_abl), &v, sizeof(int)) == hipSuccess;
    }();
    (void)att_abl_set;
#endif
    // the persistent sixteen-wave form wherever there is more than one head per CU to walk (k_attention_f16_p)
    const int total = B * heads;
    const int cus = isc_device_cus();
#ifdef ISC_ABLATION
    static const bool one_shot = getenv("ISC_ATT_ONE_SHOT") != nullptr;  // A/B aid: one workgroup per head, as before
#else
    constexpr bool one_shot = false;
#endif
    const bool persistent = !one_shot && total > cus;
#define ISC_ATT_LAUNCH(PK_, NKB_, TAIL_)                                                                              \
    do {                                                                                                              \
        if (persistent) {                                                                                             \
            auto kern = k_attention_f16_p<PK_, NKB_, TAIL_>;                                                          \
            static unsigned long long attr_done = 0; /* per device: the dynamic-LDS limit of this instantiation is set */ \
            if (!att_p_prepare(reinterpret_cast<const void*>(kern), &attr_done)) return ISC_ERR_UNSUPPORTED;          \
            hipLaunchKernelGGL(kern, dim3((unsigned)cus), dim3(ATT_P_THREADS), ATT_P_LDS_BYTES, isc_stream(stream),   \
                               reinterpret_cast<const _Float16*>(qkv), T, heads, reinterpret_cast<_Float16*>(out),    \
                               total);                                                                                \
        } else {                                                                                                      \
            hipLaunchKernelGGL((k_attention_f16<PK_, NKB_, TAIL_>), dim3((unsigned)total), dim3(ATT_THREADS), 0,      \
                               isc_stream(stream), reinterpret_cast<const _Float16*>(qkv), T, heads,                  \
                               reinterpret_cast<_Float16*>(out));                                                     \
        }                                                                                                             \
    } while (0)
    // 192 < T <= 208 (ViT-B/16: 197 tokens): thirteen key blocks, only the last one masked; 208 < T: fourteen, likewise;
    // anything shorter: the generic form (fourteen blocks, every element masked)
    if (T > 208) {
        if (packed) ISC_ATT_LAUNCH(true, 14, true);
        else ISC_ATT_LAUNCH(false, 14, true);
    } else if (T > 192) {
        if (packed) ISC_ATT_LAUNCH(true, 13, true);
        else ISC_ATT_LAUNCH(false, 13, true);
    } else {
        if (packed) ISC_ATT_LAUNCH(true, 14, false);
        else ISC_ATT_LAUNCH(false, 14, false);
    }
#undef ISC_ATT_LAUNCH
    return isc_launch_status();
}

extern "C" int isc_patchify_f16(const float* x, int B, int C, int H, int W, int patch, void* patches, int packed,
                                void* stream) {
    ISC_REQUIRE(x && patches && B > 0 && C > 0 && H > 0 && W > 0 && patch > 0);
    if (patch % 8 != 0 || H % patch != 0 || W % patch != 0) return ISC_ERR_UNSUPPORTED;
    if (packed && (C * patch * patch) % 64 != 0) return ISC_ERR_UNSUPPORTED;
    if (!isc_aligned(x, 16) || !isc_aligned(patches, 16)) return ISC_ERR_ALIGNMENT;
    const size_t total8 = (size_t)B * C * H * (W / 8);
    hipLaunchKernelGGL(k_patchify_f16, dim3(grid_for(total8)), dim3(256), 0, isc_stream(stream), x, C, H, W, patch, total8,
                       reinterpret_cast<_Float16*>(patches), packed ? 1 : 0);
    return isc_launch_status();
}

extern "C" int isc_vit_assemble(const float* patch_embed, const float* cls_token, const float* pos_embed, int B, int T,
                                int D, float* tokens, void* stream) {
    ISC_REQUIRE(patch_embed && cls_token && pos_embed && tokens && B > 0 && T > 1 && D > 0);
    if (D % 4 != 0) return ISC_ERR_UNSUPPORTED;
    if (!isc_aligned(patch_embed, 16) || !isc_aligned(cls_token, 16) || !isc_aligned(pos_embed, 16) ||
        !isc_aligned(tokens, 16))
        return ISC_ERR_ALIGNMENT;
    const size_t total4 = (size_t)B * T * (D / 4);
    hipLaunchKernelGGL(k_vit_assemble, dim3(grid_for(total4)), dim3(256), 0, isc_stream(stream), patch_embed, cls_token,
                       pos_embed, T, D, total4, tokens);
    return isc_launch_status();
}
