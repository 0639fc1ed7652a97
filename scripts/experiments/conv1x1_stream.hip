// float32 1 x 1 convolution (stride 1, no padding) of the CNN encoders on the search kernel's streaming loop
// (dispatched by isc_conv2d_nhwc, encoder.hip, for Cout % 256 == 0 and Cin % 32 == 0):
//
//     out[pixel][cout] = act( sum_c x[pixel][c] * w[cout][c] + bias[cout] + residual[pixel][cout] )      (NHWC, KRSC)
//
// The f32-input MFMA (v_mfma_f32_16x16x4_f32, an exact k-ordered fma chain) runs at 1/16 of the fp16 rate, so one K step
// (32 channels) of a 256 x 256 tile keeps the matrix pipe busy for ~16 k cycles: the LDS-DMA ring, the fragment reads
// and -- decisive for these layers -- the residual loads and output stores of the PREVIOUS tile all fit beside it.
// ResNet-50's expand convolutions (K = 64 ... 512, 4 K output channels, float32 residual) are two to sixteen K steps per
// tile; the implicit-GEMM kernel of encoder.hip runs them one tile per workgroup, residual load -> MFMAs -> stores one
// after the other (layer1: 0.29 of the f32 MFMA peak, 25 k cycles per tile = 8 k of MFMA + 9.5 k of loads + 5 k of
// stores; scripts/trace_encode_layers.sh).  Here a workgroup streams a chunk of 256-pixel tiles through the ring and
// the epilogue of tile t runs inside the first K step of tile t + 1, one 16-pixel row block at a time, right before
// that block's first MFMA (same structure as k_gemm_f16_stream, gemm_stream.hip; the header there explains the
// swapped MFMA operands and the weight-row permutation that give every lane 16 consecutive output channels of a pixel).
// What it buys (ResNet-50, B = 512): the K <= 128 expand convolutions run 6 % faster; deeper ones do not gain, because
// the wall is not the overlap but the ~3.4 TB/s these layers reach on their mixed residual-read + output-store traffic
// (3.7 GB per layer1 expand) however the epilogue is scheduled -- the same vector-memory limit k_gemm_f16_stream
// documents.  isc_conv2d_nhwc therefore dispatches here only for Cin <= 128 with a residual.
// Operands are read where they lie: activations row-major [M][Cin] and weights [Cout][Cin] (the LDS-DMA source address
// is per lane, so no packing pass is needed).
#include <stdlib.h>

#include <type_traits>

#include "isc_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int GT = 256;         // tokens per tile == features per workgroup
constexpr int GTHREADS = 512;   // 8 waves
constexpr int TILE_BYTES = GT * 128;  // 32 KiB: one K step of one operand tile
constexpr int A_ST = 3, B_ST = 2;
constexpr int DA = A_ST - 1, DB = B_ST - 1;
constexpr int NA = 4, NB = 4;   // LDS-DMA instructions per thread per K step and operand


struct StreamConvParams {
    const unsigned char* a;  // activations, row-major [M][K] float32
    const unsigned char* w;  // weights, row-major [N][K] float32
    const float* bias;       // [N] or null
    const float* res;        // [M][N] float32 or null
    float* out;              // [M][N] float32
    long long M;
    int N, K, ksteps, ntiles, tiles_per_chunk, act, res_after_act;
    int group, ngroups, npairs;  // XCD-aware mapping, see the kernel
};

#define CS_DS_READ(dst_, addr_, off_) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "i"(off_))

__device__ __forceinline__ void cs_dma16(const unsigned char* gsrc, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void cs_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

template <int ACT>
__device__ __forceinline__ float cs_act(float v) {
    if constexpr (ACT == ISC_ACT_RELU) return fmaxf(v, 0.f);
    if constexpr (ACT == ISC_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));  // same form as encoder.hip apply_act
    return v;
}

// one 16-byte chunk = 4 floats: element j of every lane's chunk feeds the j-th v_mfma_f32_16x16x4_f32 (a K-axis
// permutation shared by both operands); the weight fragment (b[n]) is the MFMA "A" operand, the pixel fragment the "B"
__device__ __forceinline__ void cs_mfma_half(const u32x4& a, const u32x4 (&b)[4], f32x4 (&acc)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < 4; ++n)
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(b[n][j]), __uint_as_float(a[j]), acc[n], 0, 0, 0);
}

template <int ACT>
__global__ __launch_bounds__(GTHREADS) void k_conv1x1_f32_stream(const StreamConvParams p) {
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

    const int tile_begin = chunk * p.tiles_per_chunk;
    const int tile_end = min(p.ntiles, tile_begin + p.tiles_per_chunk);
    const int my_tiles = tile_end - tile_begin;
    if (my_tiles <= 0) return;
    const int total_steps = my_tiles * ksteps;

    const int frow = lane & 15;
    const int fg = lane >> 4;

    // ---- staging.  LDS images are lane-linear [row][128 B]; the XOR swizzle (row >> 1) & 7 is applied to the SOURCE
    // chunk.  Operands are row-major with K floats per row: K step s of a row is the 128 bytes at s * 128.  Staging
    // round i covers LDS rows 64 i + t (t = tid >> 3).  Pixel rows are staged in order (rows past M, in the last tile
    // only, re-read row M - 1: their results are never stored); weight rows through the permutation of
    // gemm_stream.hip: LDS row 16 n + r of wave column i takes output channel 64 i + 16 (r >> 2) + 4 n + (r & 3).
    const int srow = tid >> 3;
    const int spc = tid & 7;
    const int sw16 = (spc ^ ((srow >> 1) & 7)) << 4;
    const int wperm = 16 * ((srow & 15) >> 2) + 4 * (srow >> 4) + (srow & 3);
    const int64_t row_bytes = (int64_t)p.K * 4;
    const unsigned char* b_stream = p.w + ((int64_t)fb * GT + wperm) * row_bytes + sw16;
    const int64_t round_bytes = 64 * row_bytes;  // 64 rows further
    // source of pixel row (tile, 64 i + srow), K step ks
    auto a_src = [&](int step, int i) -> const unsigned char* {
        const int t = step / ksteps, ks = step - t * ksteps;
        long long row = (long long)(tile_begin + t) * GT + 64 * i + srow;
        row = row < p.M ? row : p.M - 1;
        return p.a + row * row_bytes + ks * 128 + sw16;
    };

    unsigned char* const lds_a = lds;
    unsigned char* const lds_b = lds + A_ST * TILE_BYTES;
    const unsigned lds_a_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned lds_b_addr = lds_a_addr + A_ST * TILE_BYTES;
    const int wave_dst = wave * 1024;

    auto issue_a = [&](int step) {
        unsigned char* dst = lds_a + (step % A_ST) * TILE_BYTES + wave_dst;
#pragma unroll
        for (int i = 0; i < NA; ++i) cs_dma16(a_src(step, i), dst + 8192 * i);
    };
    auto issue_b = [&](int step) {
        const unsigned char* src = b_stream + (int64_t)(step % ksteps) * 128;
        unsigned char* dst = lds_b + (step % B_ST) * TILE_BYTES + wave_dst;
#pragma unroll
        for (int i = 0; i < NB; ++i) cs_dma16(src + round_bytes * i, dst + 8192 * i);
    };
    auto issue_iter = [&](int it) {
        const int sb = it + DB, sa = it + DA;
        if (sb >= 0 && sb < total_steps) issue_b(sb);
        if (sa >= 0 && sa < total_steps) issue_a(sa);
    };
    // stream order ... B(next) A(next + 1): only A(next + 1) may stay in flight
    auto retire_for = [&](int next) {
        if (next + 1 < total_steps) cs_wait_vmcnt<NA>();
        else cs_wait_vmcnt<0>();
    };

    const int fsw = (lane >> 1) & 7;
    int foff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) foff[kk] = frow * 128 + (((kk * 4 + fg) ^ fsw) << 4);
    const int a_wave_off = wm * (GT / WM) * 128;
    const int b_wave_off = wn * 64 * 128;

    // this lane's 16 features: nb + 4 n + r
    const int nb = fb * GT + wn * 64 + fg * 16;
    f32x4 acc[MB][4];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // static priority for the second-dispatched half of the workgroup, as in k_dots_filter
    if (__builtin_amdgcn_readfirstlane(tid) >= 256) __builtin_amdgcn_s_setprio(1);

    for (int it = -DA; it < 0; ++it) issue_iter(it);
    retire_for(0);
    __builtin_amdgcn_s_barrier();

    // ---- epilogue of ONE 16-pixel row block: acc[m][n][r] = out[pixel trow0 + 16 m + frow][channel nb + 4 n + r].
    // Bias, residual (before or after the activation), activation, four 16-byte stores (64 contiguous bytes per lane,
    // 256 per pixel and store round), and the accumulators of the block are cleared for the next tile.  The residual
    // loads are ordinary loads: hipcc drains the LDS-DMA ring in front of their first use, which costs nothing here --
    // a K step is ~16 k cycles of matrix work and the ring's next stages landed long ago.
    auto epi_block = [&](f32x4 (&c)[4], long long pixel) {
        if (pixel < p.M) {
            const size_t o = (size_t)pixel * p.N + nb;
#pragma unroll
            for (int n = 0; n < 4; ++n) {  // one 16-byte piece at a time: registers are scarce beside 128 accumulators
                f32x4 v = c[n];
                if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + nb + 4 * n);
                f32x4 rr = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.res) rr = *reinterpret_cast<const f32x4*>(p.res + o + 4 * n);
                if (!p.res_after_act) v += rr;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = cs_act<ACT>(v[r]);
                if (p.res_after_act) v += rr;
                *reinterpret_cast<f32x4*>(p.out + o + 4 * n) = v;
            }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) c[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto main_loop = [&](auto stagger_tag) {
    constexpr bool STAGGER = decltype(stagger_tag)::value;
    int kt = 0, tile = 0;
    bool pend = false;        // wave-uniform: acc holds a finished tile whose epilogue has not run
    long long pend_row0 = 0;  // ... its first token row for this wave
    for (int step = 0; step < total_steps; ++step) {
        const int sb = step + DB, sa = step + DA;
        const bool do_b = sb < total_steps;
        const bool do_a = sa < total_steps;
        const unsigned char* bsrc = b_stream + (int64_t)(sb % ksteps) * 128;
        unsigned char* bdst = lds_b + (sb % B_ST) * TILE_BYTES + wave_dst;
        unsigned char* adst = lds_a + (sa % A_ST) * TILE_BYTES + wave_dst;
        const unsigned a_addr = lds_a_addr + (unsigned)((step % A_ST) * TILE_BYTES + a_wave_off);
        const unsigned b_addr = lds_b_addr + (unsigned)((step % B_ST) * TILE_BYTES + b_wave_off);
        const unsigned a_addr0 = a_addr + foff[0], a_addr1 = a_addr + foff[1];
        const unsigned b_addr0 = b_addr + foff[0], b_addr1 = b_addr + foff[1];
        u32x4 b0[4], b1[4], ar[3][2];
        CS_DS_READ(b0[0], b_addr0, 0);
        CS_DS_READ(b0[1], b_addr0, 2048);
        CS_DS_READ(b0[2], b_addr0, 4096);
        CS_DS_READ(b0[3], b_addr0, 6144);
        CS_DS_READ(ar[0][0], a_addr0, 0);
        CS_DS_READ(b1[0], b_addr1, 0);
        CS_DS_READ(b1[1], b_addr1, 2048);
        CS_DS_READ(b1[2], b_addr1, 4096);
        CS_DS_READ(b1[3], b_addr1, 6144);
        CS_DS_READ(ar[0][1], a_addr1, 0);
        CS_DS_READ(ar[1][0], a_addr0, 2048);
        CS_DS_READ(ar[1][1], a_addr1, 2048);
#define CS_EPI(m_) \
    if (pend) epi_block(acc[m_], pend_row0 + (m_) * 16 + frow);
#define CS_DMA(j_)                                                        \
    if ((j_) < 4) {                                                       \
        if (do_b) cs_dma16(bsrc + round_bytes * (j_), bdst + 8192 * (j_)); \
    } else {                                                              \
        if (do_a) cs_dma16(a_src(sa, (j_)-4), adst + 8192 * ((j_)-4));     \
    }
        if constexpr (STAGGER) {
            // type B: [first half of block m] [reads m + 2, DMA, wait for block m + 1] [second half of block m]
            asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(b0[0]), "+v"(b0[1]), "+v"(b0[2]), "+v"(b0[3]), "+v"(ar[0][0]));
            __builtin_amdgcn_sched_barrier(0);
            CS_EPI(0)
            cs_mfma_half(ar[0][0], b0, acc[0]);
            CS_DS_READ(ar[2][0], a_addr0, 4096);
            CS_DS_READ(ar[2][1], a_addr1, 4096);
            CS_DMA(0)
            asm volatile("s_waitcnt lgkmcnt(2)"
                         : "+v"(b1[0]), "+v"(b1[1]), "+v"(b1[2]), "+v"(b1[3]), "+v"(ar[0][1]), "+v"(ar[1][0]),
                           "+v"(ar[1][1]));
            __builtin_amdgcn_sched_barrier(0);
            cs_mfma_half(ar[0][1], b1, acc[0]);
#define CS_ROW_BLOCK_B(m_, cur_, nxt_, nn_, wait_)                                                          \
    CS_EPI(m_)                                                                                              \
    cs_mfma_half(ar[cur_][0], b0, acc[m_]);                                                                 \
    if constexpr ((m_) + 2 < 8) {                                                                           \
        CS_DS_READ(ar[nn_][0], a_addr0, ((m_) + 2) * 2048);                                                 \
        CS_DS_READ(ar[nn_][1], a_addr1, ((m_) + 2) * 2048);                                                 \
    }                                                                                                       \
    CS_DMA(m_)                                                                                              \
    if constexpr ((m_) + 1 < 8) {                                                                           \
        asm volatile("s_waitcnt lgkmcnt(" wait_ ")" : "+v"(ar[nxt_][0]), "+v"(ar[nxt_][1]), "+v"(ar[cur_][1])); \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    }                                                                                                       \
    cs_mfma_half(ar[cur_][1], b1, acc[m_]);
            CS_ROW_BLOCK_B(1, 1, 2, 0, "2")
            CS_ROW_BLOCK_B(2, 2, 0, 1, "2")
            CS_ROW_BLOCK_B(3, 0, 1, 2, "2")
            CS_ROW_BLOCK_B(4, 1, 2, 0, "2")
            CS_ROW_BLOCK_B(5, 2, 0, 1, "2")
            CS_ROW_BLOCK_B(6, 0, 1, 2, "0")
            CS_ROW_BLOCK_B(7, 1, 2, 0, "0")
#undef CS_ROW_BLOCK_B
        } else {
            // type A
            CS_DS_READ(ar[2][0], a_addr0, 4096);
            CS_DS_READ(ar[2][1], a_addr1, 4096);
            asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(b0[0]), "+v"(b0[1]), "+v"(b0[2]), "+v"(b0[3]), "+v"(ar[0][0]));
            __builtin_amdgcn_sched_barrier(0);
            CS_DMA(0)
            CS_EPI(0)
            cs_mfma_half(ar[0][0], b0, acc[0]);
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(b1[0]), "+v"(b1[1]), "+v"(b1[2]), "+v"(b1[3]), "+v"(ar[0][1]));
            __builtin_amdgcn_sched_barrier(0);
            cs_mfma_half(ar[0][1], b1, acc[0]);
#define CS_ROW_BLOCK(m_, cur_, nxt_, wait_)                                                    \
    if constexpr ((m_) + 2 < 8) {                                                              \
        CS_DS_READ(ar[nxt_][0], a_addr0, ((m_) + 2) * 2048);                                   \
        CS_DS_READ(ar[nxt_][1], a_addr1, ((m_) + 2) * 2048);                                   \
    }                                                                                          \
    asm volatile("s_waitcnt lgkmcnt(" wait_ ")" : "+v"(ar[cur_][0]), "+v"(ar[cur_][1]));       \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    CS_DMA(m_)                                                                                 \
    CS_EPI(m_)                                                                                 \
    cs_mfma_half(ar[cur_][0], b0, acc[m_]);                                                    \
    cs_mfma_half(ar[cur_][1], b1, acc[m_]);
            CS_ROW_BLOCK(1, 1, 0, "4")
            CS_ROW_BLOCK(2, 2, 1, "4")
            CS_ROW_BLOCK(3, 0, 2, "4")
            CS_ROW_BLOCK(4, 1, 0, "4")
            CS_ROW_BLOCK(5, 2, 1, "4")
            CS_ROW_BLOCK(6, 0, 2, "2")
            CS_ROW_BLOCK(7, 1, 0, "0")
#undef CS_ROW_BLOCK
        }
#undef CS_DMA
#undef CS_EPI

        pend = false;  // this step wrote the pending tile out block by block
        if (++kt == ksteps) {
            kt = 0;
            pend = true;  // written out during the next step of the stream (or after the loop)
            pend_row0 = (long long)(tile_begin + tile) * GT + wm * (GT / WM);
            ++tile;
        }
        // retire this wave's DMA for step + 1, then publish.  In a step that carried an epilogue the newest operations
        // are its stores: the counted wait then retires more than it has to, never less.
        retire_for(step + 1);
        __builtin_amdgcn_s_barrier();
    }
    if (pend) {
#pragma unroll
        for (int m = 0; m < MB; ++m) epi_block(acc[m], pend_row0 + m * 16 + frow);
    }
    };
    if (__builtin_amdgcn_readfirstlane(wm) == 1) main_loop(std::true_type{});
    else main_loop(std::false_type{});
}

}  // namespace

// Called by isc_conv2d_nhwc for 1 x 1 / stride 1 / no padding with Cin % 32 == 0 and Cout % 256 == 0.
int isc_conv1x1_stream_launch(const float* x, long long M, int K, const float* w, int N, const float* bias,
                              const float* residual, int act, int res_after_act, float* out, hipStream_t stream) {
    StreamConvParams p;
    p.a = reinterpret_cast<const unsigned char*>(x);
    p.w = reinterpret_cast<const unsigned char*>(w);
    p.bias = bias;
    p.res = residual;
    p.out = out;
    p.M = M;
    p.N = N;
    p.K = K;
    p.ksteps = K / 32;
    p.ntiles = (int)((M + GT - 1) / GT);
    p.act = act;
    p.res_after_act = res_after_act;
    const int fbs = N / GT;
    // groups of 4 (or 3, 2) output-channel blocks that stream the same pixels share an XCD's L2; 32 CUs per XCD
    p.group = fbs % 4 == 0 ? 4 : fbs % 3 == 0 ? 3 : fbs % 2 == 0 ? 2 : 1;
    p.ngroups = fbs / p.group;
    const int pairs_per_xcd = 32 / p.group;
    int want = 8 * pairs_per_xcd / p.ngroups;  // chunks of pixel tiles
    if (want < 1) want = 1;
    if (want > p.ntiles) want = p.ntiles;
    p.tiles_per_chunk = (p.ntiles + want - 1) / want;
    const int nchunks = (p.ntiles + p.tiles_per_chunk - 1) / p.tiles_per_chunk;
    p.npairs = nchunks * p.ngroups;
    const int slots = ((p.npairs + 7) / 8) * p.group;  // per XCD
    const dim3 grid(8 * slots), block(GTHREADS);
    if (act == ISC_ACT_RELU) hipLaunchKernelGGL(k_conv1x1_f32_stream<ISC_ACT_RELU>, grid, block, 0, stream, p);
    else if (act == ISC_ACT_SILU) hipLaunchKernelGGL(k_conv1x1_f32_stream<ISC_ACT_SILU>, grid, block, 0, stream, p);
    else if (act == ISC_ACT_NONE) hipLaunchKernelGGL(k_conv1x1_f32_stream<ISC_ACT_NONE>, grid, block, 0, stream, p);
    else return ISC_ERR_UNSUPPORTED;
    return isc_launch_status();
}
