// fp16 GEMM of the transformer encoder on the search kernel's streaming loop (dispatched by isc_gemm_f16, vit.hip).
//
//     out[token][feature] = act( sum_k a[token][k] * w[feature][k] + bias[feature] ) (+ residual[token][feature])
//
// Same machinery as k_dots_filter (cosine_topk.hip): one workgroup of 512 threads owns all 160 KiB of a CU's LDS; the
// K steps of a chunk of consecutive 256-token tiles form ONE stream that is staged by LDS-DMA into a 3-deep ring with
// counted `vmcnt` waits and raw `s_barrier`s, while the workgroup's 256-feature block of the weight matrix cycles
// through a 2-deep ring (it is L2 resident, like the queries of a search).  Both operands come in the packed layout
// ([tile of 256 rows][K step][row][128 B]), so every LDS-DMA instruction moves one contiguous KiB.  Waves 2 x 4, each
// 128 tokens x 64 features; fragments by inline-asm `ds_read_b128` two row blocks ahead of the matrix cores; the two
// waves of a SIMD run copies of the loop shifted by half a row block.
//
// What differs from the search:
//   * MFMA operands are SWAPPED -- the weight fragment is the "A" operand -- so the C layout puts the feature on the
//     register index and the token on the lane: a lane owns features, not tokens.  The weight rows are additionally
//     permuted when they are staged (LDS row 16 n + r of a wave's 64 holds feature 16 (r >> 2) + 4 n + (r & 3); only the
//     LDS-DMA source address changes), so the 16 results of a lane for one token are 16 CONSECUTIVE features: 32-byte
//     (fp16) or 64-byte (f32) runs per lane, whole 128-byte lines per token and store instruction.
//   * the end of a tile is a store epilogue instead of a threshold filter: bias (held in registers for the whole
//     kernel -- the feature block never changes), exact-erf GELU, float32 residual, fp16 (packed) or float32 output.
//     Its stores (and residual loads) share the `vmcnt` counter with the DMA ring, so the wait at the end of a tile's
//     last K step is a full drain; every other K step keeps the ring's counted waits.
//   * workgroup -> (chunk of token tiles, feature block) is XCD-aware (groups of 3 - 4 feature blocks that stream the same
//     token chunk sit on one XCD).
//
// Where it stands (ViT-B/16, M = 100 864 tokens, profiles/r02_gemm_pmc.txt): 0.28 - 0.35 of the fp16 peak, +8 - 12 % over
// the 128 x 128 kernel.  The main loop alone (no stores) reaches 0.37 - 0.43; the matrix pipe is 40 % busy at a held
// clock of 1.72 GHz.  What the store epilogue costs does not depend on where it is issued, on coalescing, on the
// workgroups' relative phase or on the counted waits (all built and measured): a store wave-instruction costs ~88 cycles
// of the CU's vector-store path (~12 B/clk/CU, the guide's figure), so the 128 KiB of a 256 x 256 fp16 tile need ~2.7 K
// steps of that path, and with K = 768 a tile is only 12 K steps; the accumulators are reused from the next step on,
// so the stores cannot be spread further without a second accumulator set.  L2 -> fabric reads are 6 x the token bytes
// (934 MB per qkv GEMM; the partners drift out of each other's L2 window as in the search kernel).
#include <stdlib.h>

#include <type_traits>

#include "isc_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int GT = 256;         // tokens per tile == features per workgroup
constexpr int GTHREADS = 512;   // 8 waves
constexpr int TILE_BYTES = GT * 128;  // 32 KiB: one K step of one operand tile
constexpr int A_ST = 3, B_ST = 2;
constexpr int DA = A_ST - 1, DB = B_ST - 1;
constexpr int NA = 4, NB = 4;   // LDS-DMA instructions per thread per K step and operand

enum { EPI_F16 = 0, EPI_F16_GELU = 1, EPI_F32 = 2 };

struct StreamGemmParams {
    const unsigned char* a;  // tokens, packed [token tile][K step][row][128 B]
    const unsigned char* w;  // weights, packed [feature tile][K step][row][128 B]
    const float* bias;       // [N] or null
    const float* res;        // [M][N] float32 or null (EPI_F32 only)
    void* out;               // fp16 packed / row-major, or float32 row-major [M][N]
    long long M;
    int N, ksteps, ntiles, tiles_per_chunk, out_packed;
    int group, ngroups, npairs;  // XCD-aware mapping, see the kernel
    int seg_off, seg_len;        // this launch covers tiles [seg_off, seg_off + seg_len) of every chunk
};

#define GS_DS_READ(dst_, addr_, off_) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "i"(off_))

__device__ __forceinline__ void gs_dma16(const unsigned char* gsrc, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void gs_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// GELU with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7), as in vit.hip
__device__ __forceinline__ float gs_gelu(float v) {
    const float x = fabsf(v) * 0.70710678118654752440f;
    const float t = __frcp_rn(fmaf(0.3275911f, x, 1.f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = 1.f - poly * t * __expf(-x * x);
    return 0.5f * v + 0.5f * fabsf(v) * e;
}

__device__ __forceinline__ size_t gs_pk_offset(long long row, int col, int cols) {
    return (((size_t)(row >> 8) * (size_t)(cols >> 6) + (size_t)(col >> 6)) * 256 + (size_t)(row & 255)) * 64 +
           (size_t)(col & 63);
}

// 16 row blocks x 8 MFMAs; the weight fragment (b_[n]) is the MFMA "A" operand, the token fragment (a_) the "B" operand
__device__ __forceinline__ void gs_mfma_half(const u32x4& a, const u32x4 (&b)[4], f32x4 (&acc)[4]) {
#pragma unroll
    for (int n = 0; n < 4; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, b[n]), __builtin_bit_cast(half8, a),
                                                        acc[n], 0, 0, 0);
}

// DBG (-DISC_ABLATION builds only, ISC_GEMM_DEBUG, wrong results): 1 = no store epilogue, 2 = the token stream re-reads
// the chunk's first tile (L2 hot), 3 = both, 4 = weights staged once (no re-staging per tile), 5 = fp16 stores to a
// fully coalesced (wrong) address pattern
// SPLIT: only waves 0 - 3 (wm = 0, one per SIMD) issue the LDS-DMA, twice as many pieces each; waves 4 - 7 never touch the
// vector-memory counter for the ring.  Their output stores -- half of every tile -- are then bunched right at the end of
// the tile and nobody ever waits for them; the wm = 0 waves bunch theirs too and leave them in flight with a counted
// wait.  (scripts/microbench/store_wall.hip: stores bunched behind a tile's MFMAs hide 60 % under the next tile, stores
// interleaved with MFMAs cost ~150 cycles of matrix issue each, and what really exposes them is a later vmcnt wait.)
template <int EPI, int DBG, bool SPLIT>
__global__ __launch_bounds__(GTHREADS) void k_gemm_f16_stream(const StreamGemmParams p) {
    constexpr int WN = 4, WM = 2, MB = 8;
    constexpr int LDS_BYTES = (A_ST + B_ST) * TILE_BYTES;
    static_assert(LDS_BYTES == 163840, "the two rings fill the CU's LDS exactly");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;  // which 128 tokens of the tile
    const int wn = wave % WN;  // which 64 features of the block
    // Workgroup -> (chunk of token tiles, feature block), XCD-aware.  Workgroup i runs on XCD i % 8 (its own 4 MiB L2).
    // The `group` feature blocks that stream the SAME token chunk are given consecutive slots of ONE XCD, so a token
    // K-step block is fetched from beyond the L2 once per group instead of once per feature block (measured before
    // this mapping: ~3.7 TB/s of L2 -> fabric traffic with or without the output stores -- the kernel was bound by it).
    // pair = (chunk, group of feature blocks); pairs are dealt round-robin to the XCDs.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = (slot / p.group) * 8 + xcd;
    if (pair >= p.npairs) return;
    const int chunk = pair / p.ngroups;
    const int fb = (pair % p.ngroups) * p.group + slot % p.group;
    const int ksteps = p.ksteps;

    const int tile_begin = chunk * p.tiles_per_chunk + p.seg_off;
    const int tile_end = min(p.ntiles, chunk * p.tiles_per_chunk + min(p.tiles_per_chunk, p.seg_off + p.seg_len));
    const int my_tiles = tile_end - tile_begin;
    if (my_tiles <= 0) return;
    const int total_steps = my_tiles * ksteps;

    const int frow = lane & 15;
    const int fg = lane >> 4;

    // ---- staging.  LDS images are lane-linear [row][128 B]; the XOR swizzle (row >> 1) & 7 is applied to the SOURCE
    // chunk.  Token rows are staged in order; weight rows through the permutation described in the header: staging
    // round i covers LDS rows 64 i + t (t = tid >> 3), i.e. wave column wn = i, and LDS row 16 n + r of it takes feature
    // 16 (r >> 2) + 4 n + (r & 3) of the 64 -- a per-lane constant, so the rounds keep their 8 KiB immediates.
    // SPLIT: the 256 threads of waves 0 - 3 cover 32 rows per round, eight rounds per operand; LDS row 32 i + t of the
    // weight tile takes feature 64 (i >> 1) + 8 (i & 1) + [16 ((t & 15) >> 2) + 4 (t >> 4) + (t & 3)].
    const int srow = SPLIT ? (tid & 255) >> 3 : tid >> 3;
    const int spc = tid & 7;
    const int sw16 = (spc ^ ((srow >> 1) & 7)) << 4;
    // float32 output keeps the NATURAL order (MFMA block n = features 16 n .. 16 n + 15, a lane's four values of block n
    // are features 16 n + 4 fg + j): a store / residual-load instruction then covers 64 contiguous bytes per token row
    // (four lanes x 16 B) instead of four 16-byte pieces 64 bytes apart, which is what the permutation gives it
    constexpr bool NATURAL = EPI == EPI_F32 && DBG == 0;
    const int wperm = NATURAL ? srow : 16 * ((srow & 15) >> 2) + 4 * (srow >> 4) + (srow & 3);
    const unsigned char* a_stream = p.a + (int64_t)tile_begin * ksteps * TILE_BYTES + srow * 128 + sw16;
    const unsigned char* b_stream = p.w + (int64_t)fb * ksteps * TILE_BYTES + wperm * 128 + sw16;
    // byte offset of staging round i inside a weight K-step block (SPLIT): rows 64 (i >> 1) + 8 (i & 1)
    auto wround = [](int i) { return NATURAL ? 32 * i * 128 : (64 * (i >> 1) + 8 * (i & 1)) * 128; };

    unsigned char* const lds_a = lds;
    unsigned char* const lds_b = lds + A_ST * TILE_BYTES;
    const unsigned lds_a_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned lds_b_addr = lds_a_addr + A_ST * TILE_BYTES;
    const int wave_dst = (SPLIT ? (wave & 3) : wave) * 1024;

    auto issue_a = [&](int step) {
        const unsigned char* src = a_stream + (int64_t)step * TILE_BYTES;
        unsigned char* dst = lds_a + (step % A_ST) * TILE_BYTES + wave_dst;
        if constexpr (SPLIT) {
            if (wm == 0) {
#pragma unroll
                for (int i = 0; i < 2 * NA; ++i) gs_dma16(src + 4096 * i, dst + 4096 * i);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) gs_dma16(src + 8192 * i, dst + 8192 * i);
        }
    };
    auto issue_b = [&](int step) {
        const unsigned char* src = b_stream + (int64_t)(step % ksteps) * TILE_BYTES;
        unsigned char* dst = lds_b + (step % B_ST) * TILE_BYTES + wave_dst;
        if constexpr (SPLIT) {
            if (wm == 0) {
#pragma unroll
                for (int i = 0; i < 2 * NB; ++i) gs_dma16(src + wround(i), dst + 4096 * i);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) gs_dma16(src + 8192 * i, dst + 8192 * i);
        }
    };
    auto issue_iter = [&](int it) {
        const int sb = it + DB, sa = it + DA;
        if (sb >= 0 && sb < total_steps) issue_b(sb);
        if (sa >= 0 && sa < total_steps) issue_a(sa);
    };
    // stream order ... B(next) A(next + 1): only A(next + 1) may stay in flight.  SPLIT: the wm = 1 waves have issued
    // nothing for the ring and wait for nothing (the barrier after the wm = 0 waves' wait publishes the stage).
    auto retire_for = [&](int next) {
        if constexpr (SPLIT) {
            if (wm == 0) {
                if (next + 1 < total_steps) gs_wait_vmcnt<2 * NA>();
                else gs_wait_vmcnt<0>();
            }
        } else {
            if (next + 1 < total_steps) gs_wait_vmcnt<NA>();
            else gs_wait_vmcnt<0>();
        }
    };

    const int fsw = (lane >> 1) & 7;
    int foff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) foff[kk] = frow * 128 + (((kk * 4 + fg) ^ fsw) << 4);
    const int a_wave_off = wm * (GT / WM) * 128;
    const int b_wave_off = wn * 64 * 128;

    // this lane's 16 features: nb + NS n + r  (permuted: 16 contiguous features; natural: four groups 16 apart)
    constexpr int NS = NATURAL ? 16 : 4;
    const int nb = fb * GT + wn * 64 + fg * (NATURAL ? 4 : 16);
    f32x4 bias[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
        bias[n] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nb + NS * n) : f32x4{0.f, 0.f, 0.f, 0.f};
    // the bias loads above are ordinary vector-memory operations: drain them before the DMA ring starts counting
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias[0]), "+v"(bias[1]), "+v"(bias[2]), "+v"(bias[3])::"memory");

    f32x4 acc[MB][4];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // static priority for the second-dispatched half of the workgroup, as in k_dots_filter
    if constexpr (!SPLIT) {
        if (__builtin_amdgcn_readfirstlane(tid) >= 256) __builtin_amdgcn_s_setprio(1);
    }

    for (int it = -DA; it < 0; ++it) issue_iter(it);
    retire_for(0);
    __builtin_amdgcn_s_barrier();

    // ---- epilogue of ONE 16-token row block: acc[m][n][r] = out[token trow0 + 16 m + frow][feature nb + 4 n + r].
    // Bias, activation, residual, store, and the accumulators of the block are cleared for the next tile.
    auto epi_block = [&](f32x4 (&c)[4], long long token) {
#pragma unroll
        for (int n = 0; n < 4; ++n) c[n] += bias[n];
        if constexpr (EPI == EPI_F16_GELU) {
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) c[n][r] = gs_gelu(c[n][r]);
        }
        if constexpr (DBG == 1 || DBG == 3) {
            if (c[0][0] + c[1][1] + c[2][2] + c[3][3] == 123.456f) reinterpret_cast<float*>(p.out)[0] = 1.f;
        } else if constexpr (EPI == EPI_F32) {
            if (token < p.M) {
                const size_t o = (size_t)token * p.N + nb;
                if (p.res) {
                    f32x4 rr[4];
#pragma unroll
                    for (int n = 0; n < 4; ++n) rr[n] = *reinterpret_cast<const f32x4*>(p.res + o + NS * n);
#pragma unroll
                    for (int n = 0; n < 4; ++n) c[n] += rr[n];
                }
                float* dst = reinterpret_cast<float*>(p.out) + o;
#pragma unroll
                for (int n = 0; n < 4; ++n) *reinterpret_cast<f32x4*>(dst + NS * n) = c[n];
            }
        } else {
            // packed fp16 output: the buffer holds whole 256-row tiles, rows past M are padding nobody reads -- the two
            // stores are issued unconditionally; row-major output has no such rows
            if (p.out_packed || token < p.M) {
                _Float16* dst = reinterpret_cast<_Float16*>(p.out) +
                                (p.out_packed ? gs_pk_offset(token, nb, p.N) : (size_t)token * p.N + nb);
                if constexpr (DBG == 5) {  // timing aid: the same bytes, but every store instruction one contiguous KiB
                    const long long blk = ((token >> 4) * (p.N >> 6) + (nb >> 6)) * 2;
                    dst = reinterpret_cast<_Float16*>(p.out) + (blk * 64 + lane) * 8;
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 lo = c[2 * h], hi = c[2 * h + 1];
                    *reinterpret_cast<half8*>(dst + (DBG == 5 ? 512 * h : 8 * h)) =
                        half8{(_Float16)lo[0], (_Float16)lo[1], (_Float16)lo[2], (_Float16)lo[3],
                              (_Float16)hi[0], (_Float16)hi[1], (_Float16)hi[2], (_Float16)hi[3]};
                }
            }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) c[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // fp16-output epilogues are DEFERRED: the finished tile's row block m is written out in the next K step of the
    // stream, right before that step's first MFMA on block m -- its conversions, GELU and stores then issue beside the
    // matrix work of the other blocks and of the SIMD's partner wave instead of holding the whole workgroup at the tile
    // boundary (measured without any epilogue: 0.38 - 0.43 of peak; with all of it at the boundary: 0.17 - 0.34).
    // The float32 + residual epilogue needs loads whose waits would drain the DMA ring eight times per tile; it stays
    // at the boundary (one drain).
    constexpr bool DEFER = EPI != EPI_F32 && !SPLIT;

    auto main_loop = [&](auto stagger_tag) {
    constexpr bool STAGGER = decltype(stagger_tag)::value;
    int kt = 0, tile = 0;
    bool pend = false;        // wave-uniform: acc holds a finished tile whose epilogue has not run
    long long pend_row0 = 0;  // ... its first token row for this wave
    for (int step = 0; step < total_steps; ++step) {
        const int sb = step + DB, sa = step + DA;
        const bool do_b = sb < total_steps && !(DBG == 4 && sb >= B_ST);
        const bool do_a = sa < total_steps;
        const unsigned char* bsrc = b_stream + (int64_t)(sb % ksteps) * TILE_BYTES;
        unsigned char* bdst = lds_b + (sb % B_ST) * TILE_BYTES + wave_dst;
        const unsigned char* asrc = a_stream + (int64_t)((DBG == 2 || DBG == 3) ? sa % ksteps : sa) * TILE_BYTES;
        unsigned char* adst = lds_a + (sa % A_ST) * TILE_BYTES + wave_dst;
        const unsigned a_addr = lds_a_addr + (unsigned)((step % A_ST) * TILE_BYTES + a_wave_off);
        const unsigned b_addr = lds_b_addr + (unsigned)((step % B_ST) * TILE_BYTES + b_wave_off);
        const unsigned a_addr0 = a_addr + foff[0], a_addr1 = a_addr + foff[1];
        const unsigned b_addr0 = b_addr + foff[0], b_addr1 = b_addr + foff[1];
        u32x4 b0[4], b1[4], ar[3][2];
        GS_DS_READ(b0[0], b_addr0, 0);
        GS_DS_READ(b0[1], b_addr0, 2048);
        GS_DS_READ(b0[2], b_addr0, 4096);
        GS_DS_READ(b0[3], b_addr0, 6144);
        GS_DS_READ(ar[0][0], a_addr0, 0);
        GS_DS_READ(b1[0], b_addr1, 0);
        GS_DS_READ(b1[1], b_addr1, 2048);
        GS_DS_READ(b1[2], b_addr1, 4096);
        GS_DS_READ(b1[3], b_addr1, 6144);
        GS_DS_READ(ar[0][1], a_addr1, 0);
        GS_DS_READ(ar[1][0], a_addr0, 2048);
        GS_DS_READ(ar[1][1], a_addr1, 2048);
#define GS_EPI(m_) \
    if constexpr (DEFER) { if (pend) epi_block(acc[m_], pend_row0 + (m_) * 16 + frow); }
#define GS_DMA(j_)                                                                                  \
    if constexpr (SPLIT) {                                                                          \
        if constexpr (!STAGGER) { /* the wm = 0 loop: two pieces per row block */                   \
            if ((j_) < 4) {                                                                         \
                if (do_b) {                                                                         \
                    gs_dma16(bsrc + wround(2 * (j_)), bdst + 4096 * (2 * (j_)));                    \
                    gs_dma16(bsrc + wround(2 * (j_) + 1), bdst + 4096 * (2 * (j_) + 1));            \
                }                                                                                   \
            } else if (do_a) {                                                                      \
                gs_dma16(asrc + 4096 * (2 * ((j_)-4)), adst + 4096 * (2 * ((j_)-4)));               \
                gs_dma16(asrc + 4096 * (2 * ((j_)-4) + 1), adst + 4096 * (2 * ((j_)-4) + 1));       \
            }                                                                                       \
        }                                                                                           \
    } else if ((j_) < 4) {                                                                          \
        if (do_b) gs_dma16(bsrc + 8192 * (j_), bdst + 8192 * (j_));                                 \
    } else {                                                                                        \
        if (do_a) gs_dma16(asrc + 8192 * ((j_)-4), adst + 8192 * ((j_)-4));                         \
    }
        if constexpr (STAGGER) {
            // type B: [first half of block m] [reads m + 2, DMA, wait for block m + 1] [second half of block m]
            asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(b0[0]), "+v"(b0[1]), "+v"(b0[2]), "+v"(b0[3]), "+v"(ar[0][0]));
            __builtin_amdgcn_sched_barrier(0);
            GS_EPI(0)
            gs_mfma_half(ar[0][0], b0, acc[0]);
            GS_DS_READ(ar[2][0], a_addr0, 4096);
            GS_DS_READ(ar[2][1], a_addr1, 4096);
            GS_DMA(0)
            asm volatile("s_waitcnt lgkmcnt(2)"
                         : "+v"(b1[0]), "+v"(b1[1]), "+v"(b1[2]), "+v"(b1[3]), "+v"(ar[0][1]), "+v"(ar[1][0]),
                           "+v"(ar[1][1]));
            __builtin_amdgcn_sched_barrier(0);
            gs_mfma_half(ar[0][1], b1, acc[0]);
#define GS_ROW_BLOCK_B(m_, cur_, nxt_, nn_, wait_)                                                          \
    GS_EPI(m_)                                                                                              \
    gs_mfma_half(ar[cur_][0], b0, acc[m_]);                                                                 \
    if constexpr ((m_) + 2 < 8) {                                                                           \
        GS_DS_READ(ar[nn_][0], a_addr0, ((m_) + 2) * 2048);                                                 \
        GS_DS_READ(ar[nn_][1], a_addr1, ((m_) + 2) * 2048);                                                 \
    }                                                                                                       \
    GS_DMA(m_)                                                                                              \
    if constexpr ((m_) + 1 < 8) {                                                                           \
        asm volatile("s_waitcnt lgkmcnt(" wait_ ")" : "+v"(ar[nxt_][0]), "+v"(ar[nxt_][1]), "+v"(ar[cur_][1])); \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    }                                                                                                       \
    gs_mfma_half(ar[cur_][1], b1, acc[m_]);
            GS_ROW_BLOCK_B(1, 1, 2, 0, "2")
            GS_ROW_BLOCK_B(2, 2, 0, 1, "2")
            GS_ROW_BLOCK_B(3, 0, 1, 2, "2")
            GS_ROW_BLOCK_B(4, 1, 2, 0, "2")
            GS_ROW_BLOCK_B(5, 2, 0, 1, "2")
            GS_ROW_BLOCK_B(6, 0, 1, 2, "0")
            GS_ROW_BLOCK_B(7, 1, 2, 0, "0")
#undef GS_ROW_BLOCK_B
        } else {
            // type A
            GS_DS_READ(ar[2][0], a_addr0, 4096);
            GS_DS_READ(ar[2][1], a_addr1, 4096);
            asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(b0[0]), "+v"(b0[1]), "+v"(b0[2]), "+v"(b0[3]), "+v"(ar[0][0]));
            __builtin_amdgcn_sched_barrier(0);
            GS_DMA(0)
            GS_EPI(0)
            gs_mfma_half(ar[0][0], b0, acc[0]);
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(b1[0]), "+v"(b1[1]), "+v"(b1[2]), "+v"(b1[3]), "+v"(ar[0][1]));
            __builtin_amdgcn_sched_barrier(0);
            gs_mfma_half(ar[0][1], b1, acc[0]);
#define GS_ROW_BLOCK(m_, cur_, nxt_, wait_)                                                    \
    if constexpr ((m_) + 2 < 8) {                                                              \
        GS_DS_READ(ar[nxt_][0], a_addr0, ((m_) + 2) * 2048);                                   \
        GS_DS_READ(ar[nxt_][1], a_addr1, ((m_) + 2) * 2048);                                   \
    }                                                                                          \
    asm volatile("s_waitcnt lgkmcnt(" wait_ ")" : "+v"(ar[cur_][0]), "+v"(ar[cur_][1]));       \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    GS_DMA(m_)                                                                                 \
    GS_EPI(m_)                                                                                 \
    gs_mfma_half(ar[cur_][0], b0, acc[m_]);                                                    \
    gs_mfma_half(ar[cur_][1], b1, acc[m_]);
            GS_ROW_BLOCK(1, 1, 0, "4")
            GS_ROW_BLOCK(2, 2, 1, "4")
            GS_ROW_BLOCK(3, 0, 2, "4")
            GS_ROW_BLOCK(4, 1, 0, "4")
            GS_ROW_BLOCK(5, 2, 1, "4")
            GS_ROW_BLOCK(6, 0, 2, "2")
            GS_ROW_BLOCK(7, 1, 0, "0")
#undef GS_ROW_BLOCK
        }
#undef GS_DMA
#undef GS_EPI

        bool drained = false;
        const bool had_epi = DEFER && pend;
        if constexpr (DEFER) pend = false;  // this step wrote the pending tile out block by block
        if (++kt == ksteps) {
            kt = 0;
            const long long trow0 = (long long)(tile_begin + tile) * GT + wm * (GT / WM);
            if constexpr (DEFER) {
                pend = true;  // written out during the next step of the stream (or after the loop)
                pend_row0 = trow0;
            } else {
#pragma unroll
                for (int m = 0; m < MB; ++m) epi_block(acc[m], trow0 + m * 16 + frow);
                drained = true;
            }
            ++tile;
        }

        // retire this wave's DMA for step + 1, then publish.  After a boundary epilogue everything is drained (its loads
        // and stores share the counter).  In a step that carried a deferred epilogue the newest operations are stores
        // and the bank pieces interleaved with them: the counted wait then retires more than it has to, never less.
        if (drained && SPLIT) {
            // the tile's stores were issued just now, behind this step's ring pieces.  wm = 1: nothing to wait for.  wm = 0,
            // packed fp16 output (two unconditional stores per row block): leave the 16 stores and the 8 token pieces
            // of step + 2 in flight; otherwise (float32 output: residual loads, predicated stores) drain.
            if (wm == 0) {
                if (EPI != EPI_F32 && p.out_packed && step + 2 < total_steps) gs_wait_vmcnt<2 * NA + 16>();
                else gs_wait_vmcnt<0>();
            }
        } else if (drained) {
            gs_wait_vmcnt<0>();
        } else if (had_epi && p.out_packed && step + 2 < total_steps) {
            // issue order of this step: [S(0) D0] ... [S(3) D3] [S(4) D4] ... [S(7) D7]  (S = the two stores of a block's
            // epilogue -- unconditional for packed output --, D0-3 the weight pieces of step + 1, D4-7 the token pieces of
            // step + 2).  Only D0-3 and what precedes them must have landed: the newest 12 operations stay in flight,
            // so the stores' acknowledgement latency is not waited for here.
            gs_wait_vmcnt<NA + 8>();
        } else {
            retire_for(step + 1);
        }
        __builtin_amdgcn_s_barrier();
    }
    if constexpr (DEFER) {
        if (pend) {
#pragma unroll
            for (int m = 0; m < MB; ++m) epi_block(acc[m], pend_row0 + m * 16 + frow);
        }
    }
    };
    if (__builtin_amdgcn_readfirstlane(wm) == 1) main_loop(std::true_type{});
    else main_loop(std::false_type{});
}

}  // namespace

// tiles per chunk and launch (0 = whole chunks)
#ifdef ISC_ABLATION
static int gemm_seg_tiles() {
    static const int v = [] {
        const char* e = getenv("ISC_GEMM_SEG");
        return e ? atoi(e) : 0;
    }();
    return v;
}
#else
static constexpr int gemm_seg_tiles() { return 0; }
#endif

// Called by isc_gemm_f16 for packed operands with N a multiple of 256.  epi: 0 = fp16 out, 1 = fp16 out with GELU,
// 2 = float32 out (+ optional float32 residual).
int isc_gemm_f16_stream_launch(const void* a, long long M, int K, const void* w, int N, const float* bias,
                               const float* residual, int epi, void* out, int out_packed, hipStream_t stream) {
    StreamGemmParams p;
    p.a = static_cast<const unsigned char*>(a);
    p.w = static_cast<const unsigned char*>(w);
    p.bias = bias;
    p.res = residual;
    p.out = out;
    p.M = M;
    p.N = N;
    p.ksteps = K / 64;
    p.ntiles = (int)((M + GT - 1) / GT);
    p.out_packed = out_packed;
    const int fbs = N / GT;
    // groups of 4 (or 3) feature blocks share an XCD's L2; 32 CUs per XCD
    p.group = fbs % 4 == 0 ? 4 : fbs % 3 == 0 ? 3 : fbs % 2 == 0 ? 2 : 1;
#ifdef ISC_ABLATION
    static const int force_group = getenv("ISC_GEMM_GROUP") ? atoi(getenv("ISC_GEMM_GROUP")) : 0;  // A/B aid
    if (force_group > 0 && fbs % force_group == 0) p.group = force_group;
#endif
    p.ngroups = fbs / p.group;
    const int pairs_per_xcd = 32 / p.group;
    int want = 8 * pairs_per_xcd / p.ngroups;  // chunks of token tiles
    if (want < 1) want = 1;
    if (want > p.ntiles) want = p.ntiles;
    p.tiles_per_chunk = (p.ntiles + want - 1) / want;
    const int nchunks = (p.ntiles + p.tiles_per_chunk - 1) / p.tiles_per_chunk;
    p.npairs = nchunks * p.ngroups;
    const int slots = ((p.npairs + 7) / 8) * p.group;  // per XCD
    const dim3 grid(8 * slots), block(GTHREADS);
#define GS_LAUNCH_S(DBG_, SPLIT_)                                                                                        \
    do {                                                                                                                \
        if (epi == EPI_F16) hipLaunchKernelGGL((k_gemm_f16_stream<EPI_F16, DBG_, SPLIT_>), grid, block, 0, stream, p);  \
        else if (epi == EPI_F16_GELU)                                                                                   \
            hipLaunchKernelGGL((k_gemm_f16_stream<EPI_F16_GELU, DBG_, SPLIT_>), grid, block, 0, stream, p);             \
        else hipLaunchKernelGGL((k_gemm_f16_stream<EPI_F32, DBG_, SPLIT_>), grid, block, 0, stream, p);                 \
    } while (0)
#ifdef ISC_ABLATION
    static const bool no_split = getenv("ISC_GEMM_NO_SPLIT") != nullptr;  // A/B aid
#define GS_LAUNCH(DBG_)                    \
    do {                                   \
        if (no_split) GS_LAUNCH_S(DBG_, false); \
        else GS_LAUNCH_S(DBG_, true);      \
    } while (0)
#else
#define GS_LAUNCH(DBG_) GS_LAUNCH_S(DBG_, true)
#endif
    // A long chunk runs as several launches over consecutive tile ranges: the feature-block partners that share a
    // token chunk through their XCD's L2 drift apart as a launch goes on, and a kernel boundary realigns them.
    int seg = gemm_seg_tiles();
    if (seg <= 0 || seg > p.tiles_per_chunk) seg = p.tiles_per_chunk;
    for (p.seg_off = 0; p.seg_off < p.tiles_per_chunk; p.seg_off += seg) {
        p.seg_len = seg;
#ifdef ISC_ABLATION
        static const int dbg = [] {
            const char* e = getenv("ISC_GEMM_DEBUG");
            return e ? atoi(e) : 0;
        }();
        if (dbg == 1) GS_LAUNCH(1);
        else if (dbg == 2) GS_LAUNCH(2);
        else if (dbg == 3) GS_LAUNCH(3);
        else if (dbg == 4) GS_LAUNCH(4);
        else if (dbg == 5) GS_LAUNCH(5);
        else
#endif
            GS_LAUNCH(0);
    }
#undef GS_LAUNCH
#undef GS_LAUNCH_S
    return isc_launch_status();
}
