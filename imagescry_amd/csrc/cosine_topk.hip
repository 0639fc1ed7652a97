// Brute-force cosine top-k over an embedding bank (include/imagescry_hip.h: isc_cosine_topk).
//
// Pipeline per call (all on one stream, no host synchronisation):
//
//   k_pack_queries    queries -> the packed K-step-major layout of the bank (1.5 MiB at Q = 1024, L2 resident)
//   for each level L (row ranges [0,4096), [4096,262144), [262144,16.7M), ... -- each 64x the previous):
//     k_dots_filter   S = bank[rows] . queries^T on the matrix cores.  The scores are never written: each lane
//                     compares its accumulators with a per-query threshold tau (the kp-th best score of the rows
//                     seen in the earlier levels) and appends the few survivors (score, row) to a small
//                     per-(segment, query) buffer.  Level 0 runs with tau = -inf.
//                     At its end every lane moves its survivors into a compact per-query list (one atomic per
//                     lane and query block, outside the hot loop).
//     k_select        one wave per query: survivors + the carried list -> the best kp by (score desc, row asc);
//                     tau <- the kp-th score.
//   k_rescore         the kp = k + slack carried candidates are re-scored EXACTLY (float64 dot, float64 query
//                     norm), rounded to float32, ordered by (score desc, row asc); the first k are the result.
//
// The matrix-core pass only has to be a superset filter; ordering and the returned scores come from the exact
// pass, so the result does not depend on tile shape, accumulation order, chunking or sharding.
//
// Data layout.  The bank is PACKED (bank_layout.h): [tile of 256 rows][K step][row][128 B], so the block one K
// step of one tile needs is 32 KiB of contiguous HBM and a workgroup's whole chunk is one linear stream.  The
// queries are packed the same way per call.  Both MFMA operands are "K-major", so the same staging code serves A
// (bank rows, the streamed operand) and B (queries).  One K step is 128 bytes of every row (64 halves or 32
// floats).  LDS tiles are [rows][128 B], the eight 16-byte chunks of a row XOR-swizzled with (row >> 1) & 7 so
// that a ds_read_b128 of an MFMA fragment (16 rows x 4 chunks per wave) is bank-conflict free; the LDS image is
// lane-linear in the staging order (the swizzle is applied to the LDS-DMA source address).
//
// Two tile shapes share the kernel template:
//   TNQ = 256  256 bank rows x 256 queries per workgroup, waves 2 x 4 (128 x 64 each): the MFMA-bound shape for
//              large query batches (arithmetic intensity 128 flop per staged byte);
//   TNQ = 64   256 bank rows x  64 queries, waves 8 x 1 (32 x 64 each), a 4-deep ring for both operands: the
//              HBM-bound shape for small query batches -- three 32 KiB bank blocks are in flight per CU while the
//              matrix cores idle most of the time.
#include <stdlib.h>

#include <type_traits>

#include "bank_layout.h"
#include "isc_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 256;        // bank rows per tile
constexpr int NTHREADS = 512;  // 8 waves
constexpr int CAP = 32;        // candidate slots per (segment, query); segment = (chunk, row-block wave, lane group)
constexpr int64_t LEVEL0_ROWS = 4096;  // level 0: every score is a candidate (4096 per query)
constexpr int LEVEL_RATIO = 64;       // each later level is up to 64x larger: ~kp * 63 survivors per query (level_ratio())
constexpr int TARGET_WGS = 256;  // one workgroup per MI355X CU (the kernel uses all 160 KiB of LDS)
constexpr int MAX_CHUNKS = 256;
constexpr int QCAP = 4096;  // survivors per query per level that the compact list / k_select can hold
constexpr int SLACK = 6;
constexpr int SMALL_Q = 128;  // up to this many queries the 64-query tile shape is used

struct Cand {
    float s;
    int32_t row;
};

struct Plan {
    int tnq;             // queries per tile: 64 or 256
    int segs_per_chunk;  // (8 / (tnq / 64)) row-block waves x 4 lane groups
    int kp;              // candidates carried per query (>= k + SLACK, multiple of 16)
    int qtiles;          // ceil(Q / tnq)
    int qpad;            // qtiles * tnq
    int max_seg;         // segs_per_chunk * max chunks over the levels
};

struct Level {
    int64_t r0, r1;
    int ntiles, tiles_per_chunk, nchunks;
};

int plan_kp(int k) { return (int)isc_align_up((size_t)k + SLACK, 16); }

// How much larger than everything before it a level may be.  A level that is R times the rows seen so far lets about
// kp * (R - 1) rows per query pass the threshold (the kp-th best score of those earlier rows); the per-query list holds
// QCAP of them, so R shrinks with kp to keep a 2x margin: 64 up to kp = 32 (k <= 26), 32 at kp = 64, 16 at kp = 128.
// (With a fixed 64 every search with k > 58 on a multi-level bank overflowed the list and fell back to the exhaustive
// kernel -- found by scripts/fuzz_search.py deep.)
int level_ratio(int kp) {
    int r = QCAP / 2 / kp;
    r = r > LEVEL_RATIO ? LEVEL_RATIO : r;
    return r < 4 ? 4 : r;
}

int64_t level_end(int level, int64_t n, int ratio) {
    int64_t e = LEVEL0_ROWS;
    for (int i = 0; i < level; ++i) {
        if (e > n / ratio + 1) return n;
        e *= ratio;
    }
    return e < n ? e : n;
}

Level make_level(int level, int64_t n, int qtiles, int ratio) {
    Level l;
    l.r0 = level == 0 ? 0 : level_end(level - 1, n, ratio);
    l.r1 = level_end(level, n, ratio);
    l.ntiles = (int)isc_ceil_div<int64_t>(l.r1 - l.r0, TM);
    int want = TARGET_WGS / qtiles;
    if (want < 1) want = 1;
    if (want > MAX_CHUNKS) want = MAX_CHUNKS;
    if (want > l.ntiles) want = l.ntiles;
    l.tiles_per_chunk = isc_ceil_div(l.ntiles, want);
    l.nchunks = isc_ceil_div(l.ntiles, l.tiles_per_chunk);
    return l;
}

int forced_tile() {
    static const int v = [] {
        const char* e = getenv("ISC_FORCE_TILE");  // bring-up / benchmarking aid: 64 or 256
        return e ? atoi(e) : 0;
    }();
    return v;
}

Plan make_plan(int64_t n, int q, int k) {
    Plan p;
    p.tnq = q <= SMALL_Q ? 64 : 256;
    if (forced_tile() == 64 || forced_tile() == 256) p.tnq = forced_tile();
    p.segs_per_chunk = (8 / (p.tnq / 64)) * 4;
    p.kp = plan_kp(k);
    p.qtiles = isc_ceil_div(q, p.tnq);
    p.qpad = p.qtiles * p.tnq;
    p.max_seg = 0;
    for (int level = 0;; ++level) {
        const Level l = make_level(level, n, p.qtiles, level_ratio(p.kp));
        if (p.segs_per_chunk * l.nchunks > p.max_seg) p.max_seg = p.segs_per_chunk * l.nchunks;
        if (l.r1 >= n) break;
    }
    return p;
}

struct Workspace {
    float* tau;              // [qpad]
    float* carry_s;          // [qpad][kp]
    int32_t* carry_r;        // [qpad][kp]
    int32_t* carry_n;        // [qpad]
    Cand* seg_ent;           // [max_seg][qpad][CAP]  lane-private survivor segments (written in the hot loop)
    int32_t* qcount;         // [qpad]                survivors per query of the current level
    Cand* qlist;             // [qpad][QCAP]          ... compacted at the end of k_dots_filter
    unsigned char* qpacked;  // [qtiles][ks][tnq][128 B]
    size_t bytes;
};

Workspace carve(const Plan& p, int ks, void* base) {
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* ptr = base ? static_cast<char*>(base) + off : nullptr;
        off += isc_align_up(bytes, 256);
        return ptr;
    };
    w.tau = static_cast<float*>(take((size_t)p.qpad * 4));
    w.carry_s = static_cast<float*>(take((size_t)p.qpad * p.kp * 4));
    w.carry_r = static_cast<int32_t*>(take((size_t)p.qpad * p.kp * 4));
    w.carry_n = static_cast<int32_t*>(take((size_t)p.qpad * 4));
    w.seg_ent = static_cast<Cand*>(take((size_t)p.max_seg * p.qpad * CAP * sizeof(Cand)));
    w.qcount = static_cast<int32_t*>(take((size_t)p.qpad * 4));
    w.qlist = static_cast<Cand*>(take((size_t)p.qpad * QCAP * sizeof(Cand)));
    w.qpacked = static_cast<unsigned char*>(take((size_t)p.qpad * ks * ISC_KSTEP_BYTES));
    w.bytes = off;
    return w;
}

__global__ void k_init(float* tau, int32_t* carry_n, int32_t* qcount, int q, int qpad, int32_t* status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < qpad) {
        tau[i] = i < q ? -INFINITY : INFINITY;  // padding queries never pass the filter
        carry_n[i] = 0;
        qcount[i] = 0;
    }
    if (i < 4) status[i] = 0;
}

// queries row-major [q][ldq] -> packed [qtile][K step][tnq rows][128 B]; rows >= q and columns >= d are zero.
// One thread per 16-byte chunk.
template <typename T>
__global__ __launch_bounds__(256) void k_pack_queries(const T* __restrict__ queries, int64_t ldq, int q, int d, int ks,
                                                      int qpad, int tnq, unsigned char* __restrict__ packed) {
    constexpr int PER = 16 / (int)sizeof(T);
    const int total = qpad * ks * 8;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = i & 7;
    const int row = (i >> 3) % tnq;
    const int blk = (i >> 3) / tnq;  // qtile * ks + kstep
    const int kstep = blk % ks;
    const int qrow = (blk / ks) * tnq + row;
    T v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int e = (kstep * 8 + c) * PER + j;
        v[j] = (qrow < q && e < d) ? queries[(int64_t)qrow * ldq + e] : (T)0.f;
    }
    *reinterpret_cast<uint4*>(packed + (size_t)i * 16) = *reinterpret_cast<const uint4*>(v);
}

// --- operand traits -------------------------------------------------------------------------------------------
template <typename T>
struct Mma;

template <>
struct Mma<_Float16> {
    // acc[n] += A(16 rows) . B(16 queries x n) over the 64 halves of one K step.  One 16-byte chunk = 8 halves = the
    // k-slice one lane feeds to v_mfma_f32_16x16x32_f16; a0/b[0] hold chunks 0-3, a1/b[1] chunks 4-7.
    static __device__ __forceinline__ void half(const u32x4& a, const u32x4 (&b)[4], f32x4 (&acc)[4]) {
#pragma unroll
        for (int n = 0; n < 4; ++n)
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a),
                                                            __builtin_bit_cast(half8, b[n]), acc[n], 0, 0, 0);
    }
    static __device__ __forceinline__ void row(const u32x4& a0, const u32x4& a1, const u32x4 (&b)[2][4],
                                               f32x4 (&acc)[4]) {
        half(a0, b[0], acc);
        half(a1, b[1], acc);
    }
};

template <>
struct Mma<float> {
    // one 16-byte chunk = 4 floats: element j of every lane's chunk goes to the j-th v_mfma_f32_16x16x4_f32.
    // Lane group g therefore supplies k = 4 * chunk + j instead of k = g: a permutation of the K axis applied
    // identically to both operands, which leaves the dot products unchanged.
    static __device__ __forceinline__ void half(const u32x4& a, const u32x4 (&b)[4], f32x4 (&acc)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[j]), __uint_as_float(b[n][j]), acc[n],
                                                              0, 0, 0);
    }
    static __device__ __forceinline__ void row(const u32x4& a0, const u32x4& a1, const u32x4 (&b)[2][4],
                                               f32x4 (&acc)[4]) {
        half(a0, b[0], acc);
        half(a1, b[1], acc);
    }
};

// LDS fragment read, hidden from the compiler: a C++ load from the staging array would make hipcc drain the
// in-flight LDS-DMA (s_waitcnt vmcnt(0)) in front of it.  The destination is valid only after the counted
// lgkmcnt wait that names it.
#define ISC_DS_READ(dst_, addr_, off_) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "i"(off_))

// Direct global -> LDS copy (LDS-DMA): lane l of the wave writes 16 bytes at lds_wave_base + 16 * l.
__device__ __forceinline__ void glds16(const unsigned char* gsrc, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt range");
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

constexpr int A_TILE_BYTES = TM * 128;  // 32 KiB: one K step of one bank tile

// One workgroup = one query tile x one chunk of consecutive 256-row bank tiles.
//
// Pipeline: the K steps of all the chunk's tiles form one stream.  Iteration s issues the LDS-DMA of query step
// s + DB and bank step s + DA, computes step s, then waits with a COUNTED vmcnt (only what step s + 1 needs is
// retired; later steps stay in flight) and a raw s_barrier -- DA K steps of HBM latency are covered without holding
// a single staging register.  Ordering rules (cdna_hip_programming.md, "Pipelining across barriers"): a slot is
// read one iteration after the vmcnt + barrier that retires its DMA, and refilled one barrier after its last read.
//
// DBG is a bring-up aid (ISC_DEBUG_MODE environment variable, never set in production): 2 = no staging after the
// prologue, 3 = staging but no MFMAs, 7 = like 2 without LDS fragment reads (the filter never fires in these), 11 = production kernel without the
// half-row-block stagger of the wm = 1 waves, 12 = production kernel with every wave issuing its own share of the LDS-DMA (A/B aids, correct results), 15 = DMA issued but never waited for, 17 = DMA and MFMAs but no LDS fragment reads.  Results are wrong for DBG != 0.
template <typename T, int TNQ, int DBG>
__global__ __launch_bounds__(NTHREADS) void k_dots_filter(const unsigned char* __restrict__ bank, int64_t r0,
                                                          int64_t r1, int tiles_per_chunk, int ntiles,
                                                          const unsigned char* __restrict__ qpacked, int ksteps,
                                                          const float* __restrict__ tau, int qpad,
                                                          Cand* __restrict__ seg_ent, int32_t* __restrict__ qcount,
                                                          Cand* __restrict__ qlist, int level0,
                                                          int32_t* __restrict__ status) {
    constexpr int WN = TNQ / 64;              // waves along the queries
    constexpr int WM = 8 / WN;                // waves along the bank rows
    constexpr int MB = TM / WM / 16;          // 16-row blocks per wave: 8 (TNQ 256) or 2 (TNQ 64)
    constexpr int B_TILE_BYTES = TNQ * 128;   // one K step of the query tile
    constexpr int NA = 4;                     // LDS-DMA instructions per thread per bank step (512 x 16 B x 4)
    constexpr int NB = B_TILE_BYTES / 8192;   // ... per query step: 4 or 1
    constexpr int A_ST = TNQ == 256 ? 3 : 4;  // ring depths: together exactly 160 KiB
    constexpr int B_ST = TNQ == 256 ? 2 : 4;
    constexpr int DA = A_ST - 1;  // prefetch distances, in K steps
    constexpr int DB = B_ST - 1;
    constexpr int LDS_BYTES = A_ST * A_TILE_BYTES + B_ST * B_TILE_BYTES;
    static_assert(LDS_BYTES == 163840, "the two rings fill the CU's LDS exactly");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];  // the ONLY LDS object (see the guide)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: LDS-DMA bases stay in SGPRs
    const int wm = wave / WN;  // which TM / WM bank rows of the tile
    const int wn = wave % WN;  // which 64 queries of the tile
    const int chunk = blockIdx.x;
    const int qt = blockIdx.y;
    const int q0 = qt * TNQ;

    const int tile_begin = chunk * tiles_per_chunk;
    const int tile_end = min(ntiles, tile_begin + tiles_per_chunk);
    const int my_tiles = tile_end - tile_begin;
    const int total_steps = my_tiles * ksteps;

    const int frow = lane & 15;
    const int fg = lane >> 4;

    // thresholds of this lane's four query columns, and this lane's private survivor counters
    float thr[4];
    int cnt[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        thr[n] = tau[q0 + wn * 64 + n * 16 + frow];
        if ((DBG != 0 && DBG < 11) || DBG >= 15) thr[n] = fabsf(thr[n]) + 3.0e38f;  // ablations: nothing survives (kept opaque to the optimiser)
        cnt[n] = 0;
    }

    // --- staging: a K-step block is contiguous in memory ([row][128 B]); staging round i moves slots
    // p = tid + 512 * i (row p >> 3, 16-byte chunk p & 7), so every wave instruction reads and writes one contiguous
    // KiB.  The LDS image is lane-linear; the XOR swizzle is applied to the SOURCE chunk index.
    const int srow = tid >> 3;
    const int spc = tid & 7;
    const int slot_src = srow * 128 + ((spc ^ ((srow >> 1) & 7)) << 4);  // (srow + 64 i) >> 1: same low bits for all i
    const unsigned char* a_stream = bank + ((r0 >> 8) + tile_begin) * (int64_t)ksteps * A_TILE_BYTES + slot_src;
    const unsigned char* b_stream = qpacked + (int64_t)qt * ksteps * B_TILE_BYTES + slot_src;

    unsigned char* const lds_a = lds;
    unsigned char* const lds_b = lds + A_ST * A_TILE_BYTES;
    const unsigned lds_a_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned lds_b_addr = lds_a_addr + A_ST * A_TILE_BYTES;
    const int wave_dst = wave * 1024;  // + 8192 * i: this wave's 1 KiB piece of staging round i

    // SPLIT (256-query shape): only the wm = 0 waves issue LDS-DMA, twice as many each (pieces wave + 4 i instead of
    // wave + 8 i), so the wm = 1 wave of every SIMD never waits on a vector-memory issue slot and keeps the matrix
    // pipe fed while its partner is held up by the back-pressure of the L2 -> LDS path (+1 % at Q = 1024, +3.5 % at
    // Q = 256; DBG 12 = every wave issues its own share, the A/B reference)
    constexpr bool SPLIT = TNQ == 256 && DBG != 12;
    auto issue_a = [&](int step) {
        const unsigned char* src = a_stream + (int64_t)step * A_TILE_BYTES;
        unsigned char* dst = lds_a + (step % A_ST) * A_TILE_BYTES + wave_dst;
        if constexpr (SPLIT) {
            if (wm == 0) {
#pragma unroll
                for (int i = 0; i < 2 * NA; ++i) glds16(src + 4096 * i, dst + 4096 * i);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) glds16(src + 8192 * i, dst + 8192 * i);
        }
    };
    auto issue_b = [&](int step) {
        const unsigned char* src = b_stream + (int64_t)(step % ksteps) * B_TILE_BYTES;
        unsigned char* dst = lds_b + (step % B_ST) * B_TILE_BYTES + wave_dst;
        if constexpr (SPLIT) {
            if (wm == 0) {
#pragma unroll
                for (int i = 0; i < 2 * NB; ++i) glds16(src + 4096 * i, dst + 4096 * i);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) glds16(src + 8192 * i, dst + 8192 * i);
        }
    };
    // what iteration `it` issues (it < 0: prologue): first the query step, then the bank step
    auto issue_iter = [&](int it) {
        if ((DBG == 2 || DBG == 7) && it >= 0) return;
        const int sb = it + DB, sa = it + DA;
        if (sb >= 0 && sb < total_steps) issue_b(sb);
        if (sa >= 0 && sa < total_steps) issue_a(sa);
    };
    // after iteration `next - 1` has issued: retire everything step `next` needs, leave the younger DMA in flight
    auto retire_for = [&](int next) {
        if (DBG == 2 || DBG == 7) {
            wait_vmcnt<0>();
            return;
        }
        if (DBG == 15) return;  // ablation: the DMA is issued but never waited for (stale operands, wrong results)
        if (TNQ == 256) {  // stream order ... B(next) A(next + 1): only A(next + 1) may stay in flight
            if constexpr (SPLIT) {
                if (wm == 0) {
                    if (next + 1 < total_steps) wait_vmcnt<2 * NA>();
                    else wait_vmcnt<0>();
                }
            } else {
                if (next + 1 < total_steps) wait_vmcnt<NA>();
                else wait_vmcnt<0>();
            }
        } else {  // stream order ... B(next) A(next) | B(next+1) A(next+1) | B(next+2) A(next+2)
            const int ahead = min(2, total_steps - 1 - next);
            if (ahead >= 2) wait_vmcnt<2 * (NA + NB)>();
            else if (ahead == 1) wait_vmcnt<NA + NB>();
            else wait_vmcnt<0>();
        }
    };

    // --- fragment read offsets (bytes inside a tile image)
    const int fsw = (lane >> 1) & 7;
    int foff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) foff[kk] = frow * 128 + (((kk * 4 + fg) ^ fsw) << 4);
    const int a_wave_off = wm * (TM / WM) * 128;
    const int b_wave_off = wn * 64 * 128;

    f32x4 acc[MB][4];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int seg = (chunk * WM + wm) * 4 + fg;
    Cand* my_ent = seg_ent + ((size_t)seg * qpad + q0 + wn * 64 + frow) * CAP;  // + n * 16 * CAP

    // prologue: what iterations -DA .. -1 would have issued; then publish step 0
    for (int it = -DA; it < 0; ++it) issue_iter(it);
    retire_for(0);
    __builtin_amdgcn_s_barrier();

    // The main loop exists in two instantiations (see "type B" below); the branch is taken once, outside the loop,
    // so neither version pays for the other's registers.
    auto main_loop = [&](auto stagger_tag) {
    constexpr bool STAGGER = decltype(stagger_tag)::value;
    int kt = 0, tile = 0;
    for (int step = 0; step < total_steps; ++step) {
        if constexpr (TNQ == 256 && DBG != 7 && DBG != 17) {
            // ---- 256-query shape.  One K step = 8 row blocks of 8 MFMAs (fp16).  Fragment reads run two row blocks
            // ahead of the matrix cores (LDS returns in order: lgkmcnt(4) = "all but the newest two blocks"), and the
            // eight LDS-DMA instructions of this iteration are issued one per row block, so their issue cost hides
            // behind MFMAs instead of delaying the first one.  DMA stream order (the counted vmcnt relies on it):
            // the query step first, then the bank step.
            const int sb = step + DB, sa = step + DA;
            const bool do_b = DBG != 2 && sb < total_steps;
            const bool do_a = DBG != 2 && sa < total_steps;
            const unsigned char* bsrc = b_stream + (int64_t)(sb % ksteps) * B_TILE_BYTES;
            unsigned char* bdst = lds_b + (sb % B_ST) * B_TILE_BYTES + wave_dst;
            const unsigned char* asrc = a_stream + (int64_t)sa * A_TILE_BYTES;
            unsigned char* adst = lds_a + (sa % A_ST) * A_TILE_BYTES + wave_dst;
            const unsigned a_addr = lds_a_addr + (unsigned)((step % A_ST) * A_TILE_BYTES + a_wave_off);
            const unsigned b_addr = lds_b_addr + (unsigned)((step % B_ST) * B_TILE_BYTES + b_wave_off);
            const unsigned a_addr0 = a_addr + foff[0], a_addr1 = a_addr + foff[1];
            const unsigned b_addr0 = b_addr + foff[0], b_addr1 = b_addr + foff[1];
            u32x4 b0[4], b1[4], ar[3][2];
            ISC_DS_READ(b0[0], b_addr0, 0);  // R0: what the first four MFMAs need
            ISC_DS_READ(b0[1], b_addr0, 2048);
            ISC_DS_READ(b0[2], b_addr0, 4096);
            ISC_DS_READ(b0[3], b_addr0, 6144);
            ISC_DS_READ(ar[0][0], a_addr0, 0);
            ISC_DS_READ(b1[0], b_addr1, 0);  // R1
            ISC_DS_READ(b1[1], b_addr1, 2048);
            ISC_DS_READ(b1[2], b_addr1, 4096);
            ISC_DS_READ(b1[3], b_addr1, 6144);
            ISC_DS_READ(ar[0][1], a_addr1, 0);
            ISC_DS_READ(ar[1][0], a_addr0, 2048);  // R2
            ISC_DS_READ(ar[1][1], a_addr1, 2048);
#define ISC_DMA(j_)                                                                 \
    if constexpr (SPLIT) {                                                          \
        if constexpr (!STAGGER) { /* the wm = 0 loop: two pieces per row block (issuing the whole query step in  \
                                     block 0 instead measured 1 % slower) */        \
            if ((j_) < 4) {                                                         \
                if (do_b) {                                                         \
                    glds16(bsrc + 4096 * (2 * (j_)), bdst + 4096 * (2 * (j_)));     \
                    glds16(bsrc + 4096 * (2 * (j_) + 1), bdst + 4096 * (2 * (j_) + 1)); \
                }                                                                   \
            } else if (do_a) {                                                      \
                glds16(asrc + 4096 * (2 * ((j_)-4)), adst + 4096 * (2 * ((j_)-4)));  \
                glds16(asrc + 4096 * (2 * ((j_)-4) + 1), adst + 4096 * (2 * ((j_)-4) + 1)); \
            }                                                                       \
        }                                                                           \
    } else if ((j_) < 4) {                                                          \
        if (do_b) glds16(bsrc + 8192 * (j_), bdst + 8192 * (j_));                   \
    } else {                                                                        \
        if (do_a) glds16(asrc + 8192 * ((j_)-4), adst + 8192 * ((j_)-4));           \
    }
#define ISC_MFMA_HALF(a_, b_, m_)                                                                         \
    if constexpr (DBG != 3) Mma<T>::half(a_, b_, acc[m_]);                                                 \
    else acc[m_][0][0] += __uint_as_float((a_)[0] ^ (b_)[1][1] ^ (b_)[2][2]);
            // The two waves that share a SIMD (waves w and w + 4, i.e. wm = 0 and wm = 1) run the same program between
            // the same barriers; left alone they reach their read / wait / DMA-issue instructions together and the
            // matrix pipe idles meanwhile.  The wm = 1 waves therefore run a version shifted by half a row block:
            // their overhead instructions sit between the two halves of a row block, the wm = 0 waves' between row
            // blocks, so one partner is always issuing MFMAs.
            if constexpr (STAGGER) {
                // ---- type B: [first half of block m] [reads m + 2, DMA, wait for block m + 1] [second half of block m]
                asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(b0[0]), "+v"(b0[1]), "+v"(b0[2]), "+v"(b0[3]), "+v"(ar[0][0]));
                __builtin_amdgcn_sched_barrier(0);
                ISC_MFMA_HALF(ar[0][0], b0, 0)
                ISC_DS_READ(ar[2][0], a_addr0, 4096);  // R3
                ISC_DS_READ(ar[2][1], a_addr1, 4096);
                ISC_DMA(0)
                asm volatile("s_waitcnt lgkmcnt(2)"
                             : "+v"(b1[0]), "+v"(b1[1]), "+v"(b1[2]), "+v"(b1[3]), "+v"(ar[0][1]), "+v"(ar[1][0]),
                               "+v"(ar[1][1]));
                __builtin_amdgcn_sched_barrier(0);
                ISC_MFMA_HALF(ar[0][1], b1, 0)
#define ISC_ROW_BLOCK_B(m_, cur_, nxt_, nn_, wait_)                                                        \
    ISC_MFMA_HALF(ar[cur_][0], b0, m_)                                                                     \
    if constexpr ((m_) + 2 < 8) {                                                                          \
        ISC_DS_READ(ar[nn_][0], a_addr0, ((m_) + 2) * 2048);                                               \
        ISC_DS_READ(ar[nn_][1], a_addr1, ((m_) + 2) * 2048);                                               \
    }                                                                                                      \
    ISC_DMA(m_)                                                                                            \
    if constexpr ((m_) + 1 < 8) {                                                                          \
        asm volatile("s_waitcnt lgkmcnt(" wait_ ")" : "+v"(ar[nxt_][0]), "+v"(ar[nxt_][1]), "+v"(ar[cur_][1])); \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    }                                                                                                      \
    ISC_MFMA_HALF(ar[cur_][1], b1, m_)
                ISC_ROW_BLOCK_B(1, 1, 2, 0, "2")
                ISC_ROW_BLOCK_B(2, 2, 0, 1, "2")
                ISC_ROW_BLOCK_B(3, 0, 1, 2, "2")
                ISC_ROW_BLOCK_B(4, 1, 2, 0, "2")
                ISC_ROW_BLOCK_B(5, 2, 0, 1, "2")
                ISC_ROW_BLOCK_B(6, 0, 1, 2, "0")
                ISC_ROW_BLOCK_B(7, 1, 2, 0, "0")
#undef ISC_ROW_BLOCK_B
            } else {
            // ---- type A.  row block 0: its two halves arrive separately
            ISC_DS_READ(ar[2][0], a_addr0, 4096);  // R3
            ISC_DS_READ(ar[2][1], a_addr1, 4096);
            asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(b0[0]), "+v"(b0[1]), "+v"(b0[2]), "+v"(b0[3]), "+v"(ar[0][0]));
            __builtin_amdgcn_sched_barrier(0);
            ISC_DMA(0)
            ISC_MFMA_HALF(ar[0][0], b0, 0)
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(b1[0]), "+v"(b1[1]), "+v"(b1[2]), "+v"(b1[3]), "+v"(ar[0][1]));
            __builtin_amdgcn_sched_barrier(0);
            ISC_MFMA_HALF(ar[0][1], b1, 0)
#define ISC_ROW_BLOCK(m_, cur_, nxt_, wait_)                                                   \
    if constexpr ((m_) + 2 < 8) {                                                              \
        ISC_DS_READ(ar[nxt_][0], a_addr0, ((m_) + 2) * 2048);                                  \
        ISC_DS_READ(ar[nxt_][1], a_addr1, ((m_) + 2) * 2048);                                  \
    }                                                                                          \
    asm volatile("s_waitcnt lgkmcnt(" wait_ ")" : "+v"(ar[cur_][0]), "+v"(ar[cur_][1]));       \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    ISC_DMA(m_)                                                                                \
    ISC_MFMA_HALF(ar[cur_][0], b0, m_)                                                         \
    ISC_MFMA_HALF(ar[cur_][1], b1, m_)
            ISC_ROW_BLOCK(1, 1, 0, "4")
            ISC_ROW_BLOCK(2, 2, 1, "4")
            ISC_ROW_BLOCK(3, 0, 2, "4")
            ISC_ROW_BLOCK(4, 1, 0, "4")
            ISC_ROW_BLOCK(5, 2, 1, "4")
            ISC_ROW_BLOCK(6, 0, 2, "2")
            ISC_ROW_BLOCK(7, 1, 0, "0")
            }
#undef ISC_ROW_BLOCK
#undef ISC_MFMA_HALF
#undef ISC_DMA
        } else {
        issue_iter(step);

        // fragment reads and MFMAs, software pipelined per 16-row block: the reads of block m + 1 are in flight
        // while the matrix cores work on block m (LDS returns in order, so lgkmcnt(2) = "all but the newest two")
        {
            const unsigned a_addr = lds_a_addr + (unsigned)((step % A_ST) * A_TILE_BYTES + a_wave_off);
            const unsigned b_addr = lds_b_addr + (unsigned)((step % B_ST) * B_TILE_BYTES + b_wave_off);
            const unsigned a_addr0 = a_addr + foff[0], a_addr1 = a_addr + foff[1];
            const unsigned b_addr0 = b_addr + foff[0], b_addr1 = b_addr + foff[1];
            u32x4 bq[2][4], ar[2][2];
            if constexpr (DBG == 7 || DBG == 17) {  // no LDS traffic: feed the matrix cores from whatever the registers hold
#pragma unroll
                for (int i = 0; i < 8; ++i) bq[i >> 2][i & 3] = u32x4{(unsigned)step, 1u, 2u, (unsigned)lane};
                ar[0][0] = ar[0][1] = ar[1][0] = ar[1][1] = u32x4{(unsigned)lane, 3u, (unsigned)step, 5u};
#pragma unroll
                for (int m = 0; m < MB; ++m) Mma<T>::row(ar[m & 1][0], ar[m & 1][1], bq, acc[m]);
            } else {
            ISC_DS_READ(bq[0][0], b_addr0, 0);
            ISC_DS_READ(bq[0][1], b_addr0, 2048);
            ISC_DS_READ(bq[0][2], b_addr0, 4096);
            ISC_DS_READ(bq[0][3], b_addr0, 6144);
            ISC_DS_READ(bq[1][0], b_addr1, 0);
            ISC_DS_READ(bq[1][1], b_addr1, 2048);
            ISC_DS_READ(bq[1][2], b_addr1, 4096);
            ISC_DS_READ(bq[1][3], b_addr1, 6144);
            ISC_DS_READ(ar[0][0], a_addr0, 0);
            ISC_DS_READ(ar[0][1], a_addr1, 0);
#define ISC_ROW_STEP(m_, cur_, nxt_)                                                                                 \
    if constexpr ((m_) < MB) {                                                                                       \
        if constexpr ((m_) + 1 < MB) {                                                                               \
            ISC_DS_READ(ar[nxt_][0], a_addr0, ((m_) + 1) * 2048);                                                    \
            ISC_DS_READ(ar[nxt_][1], a_addr1, ((m_) + 1) * 2048);                                                    \
        }                                                                                                            \
        if constexpr ((m_) == 0)                                                                                     \
            asm volatile("s_waitcnt lgkmcnt(2)"                                                                      \
                         : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[0][2]), "+v"(bq[0][3]), "+v"(bq[1][0]),           \
                           "+v"(bq[1][1]), "+v"(bq[1][2]), "+v"(bq[1][3]), "+v"(ar[0][0]), "+v"(ar[0][1]));          \
        else if constexpr ((m_) + 1 < MB)                                                                            \
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ar[cur_][0]), "+v"(ar[cur_][1]));                            \
        else                                                                                                         \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ar[cur_][0]), "+v"(ar[cur_][1]));                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        if constexpr (DBG != 3)                                                                                      \
            Mma<T>::row(ar[cur_][0], ar[cur_][1], bq, acc[m_]);                                                      \
        else                                                                                                         \
            acc[m_][0][0] += __uint_as_float(ar[cur_][0][0] ^ ar[cur_][1][1] ^ bq[0][1][0] ^ bq[1][2][1]);           \
    }
            ISC_ROW_STEP(0, 0, 1)
            ISC_ROW_STEP(1, 1, 0)
            ISC_ROW_STEP(2, 0, 1)
            ISC_ROW_STEP(3, 1, 0)
            ISC_ROW_STEP(4, 0, 1)
            ISC_ROW_STEP(5, 1, 0)
            ISC_ROW_STEP(6, 0, 1)
            ISC_ROW_STEP(7, 1, 0)
#undef ISC_ROW_STEP
            }
        }

        }

        if (++kt == ksteps) {
            // ---- tile finished: threshold filter.  C layout of the 16x16 MFMA: column (query) = lane & 15,
            // row (bank row) = 4 * (lane >> 4) + register.  Survivors are rare once tau is warm, so the scan of a
            // query block only runs when some lane of the wave holds one (wave-uniform branch).  The comparison is
            // STRICT: tau is the kp-th best score of the rows of the earlier levels, all of which have smaller row
            // indices, so a row that merely ties it ranks behind those kp rows (ties go to the lower index) and can
            // never enter the list -- while ">=" let every row of a zero query, or every copy of a duplicated row,
            // through and pushed whole calls onto the exhaustive kernel.
            kt = 0;
            const int64_t trow0 = r0 + (int64_t)(tile_begin + tile) * TM + wm * (TM / WM) + fg * 4;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                float mx = -INFINITY;
#pragma unroll
                for (int m = 0; m < MB; ++m)
                    mx = fmaxf(mx, fmaxf(fmaxf(acc[m][n][0], acc[m][n][1]), fmaxf(acc[m][n][2], acc[m][n][3])));
                if (level0) {
                    // level 0 (tau = -inf, at most QCAP rows): every score is a survivor and goes straight to slot
                    // `row` of the query's list -- no counters, no compaction
                    Cand* dst = qlist + (size_t)(q0 + wn * 64 + n * 16 + frow) * QCAP;
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int64_t row = trow0 + m * 16 + r;
                            const float s = acc[m][n][r];
                            if (row < r1) dst[row - r0] = Cand{s == s ? s : -INFINITY, (int32_t)row};
                        }
                } else if (__ballot(mx > thr[n]) != 0ull) {
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float s = acc[m][n][r];
                            const int64_t row = trow0 + m * 16 + r;
                            if (s > thr[n] && row < r1) {
                                const int pos = cnt[n]++;
                                if (pos < CAP) my_ent[(size_t)n * 16 * CAP + pos] = Cand{s, (int32_t)row};
                            }
                        }
                }
            }
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            ++tile;
        }

        // retire this wave's DMA for step + 1; the barrier then publishes every wave's pieces and guarantees nobody
        // still reads the slots refilled next iteration
        retire_for(step + 1);
        __builtin_amdgcn_s_barrier();
    }
    };
    if (TNQ == 256 && DBG != 11 && __builtin_amdgcn_readfirstlane(wm) == 1) main_loop(std::true_type{});  // DBG 12 relies on this split
    else main_loop(std::false_type{});

    // ---- tail: compact this lane's private survivors into the per-query list.  One returning atomic per
    // (lane, query block) with survivors, outside the hot loop; the order inside a list is arbitrary, the
    // selection that follows uses a total order.  Level 0 wrote the lists directly: only the count is set.
    if (level0) {
        if (chunk == 0 && wm == 0 && fg == 0) {
#pragma unroll
            for (int n = 0; n < 4; ++n) qcount[q0 + wn * 64 + n * 16 + frow] = (int)(r1 - r0);
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int c = min(cnt[n], CAP);
        if (cnt[n] > CAP) atomicAdd(&status[0], 1);
        if (c > 0) {
            const int q = q0 + wn * 64 + n * 16 + frow;
            const int off = atomicAdd(&qcount[q], c);
            const Cand* src = my_ent + (size_t)n * 16 * CAP;
            Cand* dst = qlist + (size_t)q * QCAP;
            if (off + c > QCAP) atomicAdd(&status[0], 1);
            for (int i = 0; i < c; i += 4) {  // four independent loads per trip
                Cand e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = src[min(i + j, c - 1)];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i + j < c && off + i + j < QCAP) dst[off + i + j] = e[j];
            }
        }
    }
}

// (score desc, row asc); entries with row < 0 are empty
__device__ __forceinline__ bool better(float sa, int ra, float sb, int rb) {
    if (rb < 0) return ra >= 0;
    if (ra < 0) return false;
    return sa > sb || (sa == sb && ra < rb);
}

// ---- selection ----------------------------------------------------------------------------------------------
// (score, row) as one 64-bit key whose unsigned order is the search order: larger key = better candidate
// (higher score first, then LOWER row).  Keys of distinct rows are distinct.  0 is "empty".
__device__ __forceinline__ unsigned long long make_key(float s, int row) {
    unsigned u = __float_as_uint(s);
    u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;  // monotone float -> unsigned
    return ((unsigned long long)u << 32) | (unsigned)(0x7fffffff - row);
}
__device__ __forceinline__ float key_score(unsigned long long k) {
    unsigned u = (unsigned)(k >> 32);
    u ^= (u >> 31) ? 0x80000000u : 0xffffffffu;
    return __uint_as_float(u);
}
__device__ __forceinline__ int key_row(unsigned long long k) { return 0x7fffffff - (int)(unsigned)(k & 0xffffffffu); }
__device__ __forceinline__ unsigned long long bcast_key(unsigned long long k, int src_lane) {
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)k, src_lane);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(k >> 32), src_lane);
    return ((unsigned long long)hi << 32) | lo;
}

// One WAVE per query: the level's survivors plus the carried list -> the best kp by (score desc, row asc),
// tau <- the kp-th score.  Wave-synchronous, no workgroup barrier.
//   1. every lane takes the maximum key of its strided share of the candidates;
//   2. L = the kp-th largest of the 64 lane maxima: at least kp candidates are >= L, so the best kp all are;
//   3. candidates >= L (typically ~1.3 kp of them) are compacted into LDS;
//   4. if at most 64 remain they are ranked by 64 lane broadcasts, otherwise by repeated arg-max.
constexpr int SEL_WAVES = 2;
constexpr int SEL_SLACK = 8 * 64;  // the unrolled scans read up to this far past the end of the list
__global__ __launch_bounds__(64 * SEL_WAVES) void k_select(int32_t* __restrict__ qcount, const Cand* __restrict__ qlist,
                                                           int n_queries, int kp, float* __restrict__ tau,
                                                           float* __restrict__ carry_s, int32_t* __restrict__ carry_r,
                                                           int32_t* __restrict__ carry_n) {
    __shared__ unsigned long long keys_all[SEL_WAVES][QCAP + 128 + SEL_SLACK];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q = blockIdx.x * SEL_WAVES + wave;
    if (q >= n_queries) return;
    unsigned long long* keys = keys_all[wave];
    // step 3 compacts IN PLACE: a trip loads its 512 keys into registers before it stores, and it only stores below
    // the index it has read up to, so `surv` may alias `keys`
    unsigned long long* surv = keys;

    const int from_list = min(qcount[q], QCAP);
    const int carried = carry_n[q];
    const int total = from_list + carried;
    const Cand* src = qlist + (size_t)q * QCAP;
    for (int i0 = 0; i0 < from_list; i0 += 64 * 8) {  // eight independent coalesced loads per trip
        Cand e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = src[min(i0 + 64 * j + lane, QCAP - 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (i0 + 64 * j + lane < from_list) keys[i0 + 64 * j + lane] = make_key(e[j].s, e[j].row);
    }
    for (int i = lane; i < carried; i += 64)
        keys[from_list + i] = make_key(carry_s[(size_t)q * kp + i], carry_r[(size_t)q * kp + i]);
    for (int i = total + lane; i < total + SEL_SLACK; i += 64) keys[i] = 0ull;  // padding reads as "empty"
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // 1. lane maxima
    // (trip counts are wave-uniform on purpose: `base`, not the lane's own index, bounds the loops -- step 3 counts
    // survivors with ballots, and a lane that left the loop early would keep a stale count; the list is padded with
    // empty keys for SEL_SLACK entries, so the over-read is harmless)
    unsigned long long lmax = 0ull;
    for (int base = 0; base < total; base += 64 * 8) {
        const int i0 = base + lane;
        unsigned long long kk[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) kk[j] = keys[i0 + 64 * j];
#pragma unroll
        for (int j = 0; j < 8; ++j) lmax = kk[j] > lmax ? kk[j] : lmax;
    }
    // 2. threshold key: the kp-th largest lane maximum (0 = keep everything when kp > 64 or few lanes are filled)
    unsigned long long lim = 0ull;
    if (kp <= 64) {
        int rank = 0;
        for (int j = 0; j < 64; ++j) rank += bcast_key(lmax, j) > lmax ? 1 : 0;
        // keys are distinct, except that several lanes may be empty (0): those never reach rank kp - 1 <= 63 ...
        const unsigned long long mine = (rank == kp - 1) ? lmax : 0ull;
        // ... so at most one lane contributes; OR-reduce it to every lane
        unsigned lo = (unsigned)mine, hi = (unsigned)(mine >> 32);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo |= __shfl_xor(lo, off, 64);
            hi |= __shfl_xor(hi, off, 64);
        }
        lim = ((unsigned long long)hi << 32) | lo;
    }
    // 3. compact the candidates >= lim (empty slots are 0 and only pass when lim == 0; they are dropped explicitly)
    int ns = 0;  // wave-uniform
    for (int base = 0; base < total; base += 64 * 8) {
        const int i0 = base + lane;
        unsigned long long kk[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) kk[j] = keys[i0 + 64 * j];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool keep = kk[j] != 0ull && kk[j] >= lim;
            const unsigned long long mask = __ballot(keep);
            if (keep) surv[ns + __popcll(mask & ((1ull << lane) - 1ull))] = kk[j];
            ns += __popcll(mask);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const int rounds = min(kp, ns);
    if (ns <= 64) {
        // 4a. rank the survivors: lane l holds survivor l, its rank is the number of larger keys
        const unsigned long long mine = lane < ns ? surv[lane] : 0ull;
        int rank = 0;
        for (int j = 0; j < ns; ++j) rank += bcast_key(mine, j) > mine ? 1 : 0;
        if (lane < ns && rank < kp) {
            const float sc = key_score(mine);
            carry_s[(size_t)q * kp + rank] = sc;
            carry_r[(size_t)q * kp + rank] = key_row(mine);
            if (rank == kp - 1) tau[q] = sc;
        }
    } else {
        // 4b. many survivors (clustered scores, kp > 64): repeated arg-max below the previous winner
        unsigned long long last = ~0ull;
        for (int r = 0; r < rounds; ++r) {
            unsigned long long best = 0ull;
            for (int i = lane; i < ns; i += 64) {
                const unsigned long long kx = surv[i];
                if (kx < last && kx > best) best = kx;
            }
            unsigned lo = (unsigned)best, hi = (unsigned)(best >> 32);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned olo = __shfl_xor(lo, off, 64), ohi = __shfl_xor(hi, off, 64);
                const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
                const unsigned long long m = ((unsigned long long)hi << 32) | lo;
                if (o > m) {
                    lo = olo;
                    hi = ohi;
                }
            }
            last = ((unsigned long long)hi << 32) | lo;
            if (lane == 0) {
                const float sc = key_score(last);
                carry_s[(size_t)q * kp + r] = sc;
                carry_r[(size_t)q * kp + r] = key_row(last);
                if (r == kp - 1) tau[q] = sc;
            }
        }
    }
    if (lane == 0) {
        carry_n[q] = rounds;
        qcount[q] = 0;  // ready for the next level
    }
}

// One workgroup per query: exact float64 re-score of the carried candidates, final order, output.
template <typename T>
__global__ __launch_bounds__(256) void k_rescore(const unsigned char* __restrict__ bank, int ks,
                                                 const T* __restrict__ queries, int64_t ldq, int d, int kp, int k,
                                                 int64_t index_base, const int32_t* __restrict__ carry_r,
                                                 const int32_t* __restrict__ carry_n, float* __restrict__ out_s,
                                                 int64_t* __restrict__ out_i, int n_rows, int32_t* __restrict__ status) {
    __shared__ float sc[128];
    __shared__ int rw[128];
    __shared__ double qnorm_sh;
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const T* qp = queries + (int64_t)q * ldq;
    const int n = carry_n[q];

    if (wave == 0) {
        double acc = 0.0;
        for (int i = lane; i < d; i += 64) {
            const double x = (double)qp[i];
            acc = fma(x, x, acc);
        }
        acc = isc_wave_sum(acc);
        if (lane == 0) qnorm_sh = fmax(sqrt(acc), 1e-12);
    }
    __syncthreads();
    const double denom = qnorm_sh;
    for (int c = wave; c < n; c += 4) {
        const int row = carry_r[(size_t)q * kp + c];
        if ((unsigned)row >= (unsigned)n_rows) {  // cannot happen; if it ever does, never touch the bank with it:
            if (lane == 0) {                      // drop the entry and make the caller rerun on the exhaustive kernel
                sc[c] = -INFINITY;
                rw[c] = -1;
                atomicAdd(&status[0], 1);
            }
            continue;
        }
        double acc = 0.0;
        for (int i = lane; i < d; i += 64) acc = fma((double)qp[i], (double)isc_packed_load<T>(bank, row, i, ks), acc);
        acc = isc_wave_sum(acc);
        if (lane == 0) {
            sc[c] = (float)(acc / denom);
            rw[c] = row;
        }
    }
    __syncthreads();
    if (tid < n) {
        const float s = sc[tid];
        const int r = rw[tid];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += better(sc[j], rw[j], s, r) ? 1 : 0;
        if (rank < k) {
            out_s[(size_t)q * k + rank] = s;
            out_i[(size_t)q * k + rank] = (int64_t)r + index_base;
        }
    }
}

int debug_mode() {
    static const int mode = [] {
        const char* e = getenv("ISC_DEBUG_MODE");
        return e ? atoi(e) : 0;
    }();
    return mode;
}

template <typename T, int TNQ>
void launch_filter(const Level& l, const Plan& p, const Workspace& w, const unsigned char* bank, int ksteps,
                   int32_t* status, hipStream_t stream) {
#define ISC_LAUNCH_FILTER(DBG_)                                                                                      \
    hipLaunchKernelGGL((k_dots_filter<T, TNQ, DBG_>), dim3(l.nchunks, p.qtiles), dim3(NTHREADS), 0, stream, bank,    \
                       l.r0, l.r1, l.tiles_per_chunk, l.ntiles, w.qpacked, ksteps, w.tau, p.qpad, w.seg_ent,         \
                       w.qcount, w.qlist, l.r0 == 0 ? 1 : 0, status)
    // SPLIT (DBG 0) pays where a chunk has ONE query tile (Q <= 256).  With several query-tile workgroups streaming the
    // same bank rows it makes them drift further apart, and their sharing of those rows through the XCD's L2 drops
    // (measured L2 -> fabric reads per search at Q = 1024: 3.0 x the algorithmic bytes with SPLIT, 1.6 - 2.0 x without,
    // for +1 % speed), so those launches use the every-wave-issues form (DBG 12).
    int mode = debug_mode();
    if (mode == 0 && TNQ == 256 && p.qtiles > 1) mode = 12;
    switch (mode) {
        case 2: ISC_LAUNCH_FILTER(2); break;
        case 3: ISC_LAUNCH_FILTER(3); break;
        case 7: ISC_LAUNCH_FILTER(7); break;
        case 11: ISC_LAUNCH_FILTER(11); break;
        case 12: ISC_LAUNCH_FILTER(12); break;
        case 15: ISC_LAUNCH_FILTER(15); break;
        case 17: ISC_LAUNCH_FILTER(17); break;
        default: ISC_LAUNCH_FILTER(0); break;
    }
#undef ISC_LAUNCH_FILTER
}

template <typename T>
int run(const void* bank, int64_t n, int d, const void* queries, int q, int64_t ldq, int k, int64_t index_base,
        float* out_s, int64_t* out_i, int32_t* status, void* ws_base, hipStream_t stream) {
    const Plan p = make_plan(n, q, k);
    const int ksteps = isc_ksteps(d, (int)sizeof(T));
    const Workspace w = carve(p, ksteps, ws_base);
    const unsigned char* bank_bytes = static_cast<const unsigned char*>(bank);
    hipLaunchKernelGGL(k_init, dim3(isc_ceil_div(p.qpad, 256)), dim3(256), 0, stream, w.tau, w.carry_n, w.qcount, q,
                       p.qpad, status);
    hipLaunchKernelGGL(k_pack_queries<T>, dim3(isc_ceil_div(p.qpad * ksteps * 8, 256)), dim3(256), 0, stream,
                       static_cast<const T*>(queries), ldq, q, d, ksteps, p.qpad, p.tnq, w.qpacked);
    for (int level = 0;; ++level) {
        const Level l = make_level(level, n, p.qtiles, level_ratio(p.kp));
        isc_timing_begin(ISC_KERNEL_DOTS_FILTER, stream);
        if (p.tnq == 256) launch_filter<T, 256>(l, p, w, bank_bytes, ksteps, status, stream);
        else launch_filter<T, 64>(l, p, w, bank_bytes, ksteps, status, stream);
        isc_timing_end(ISC_KERNEL_DOTS_FILTER, stream);
        hipLaunchKernelGGL(k_select, dim3(isc_ceil_div(q, SEL_WAVES)), dim3(64 * SEL_WAVES), 0, stream, w.qcount, w.qlist,
                           q, p.kp, w.tau, w.carry_s, w.carry_r, w.carry_n);
        if (l.r1 >= n) break;
    }
    hipLaunchKernelGGL(k_rescore<T>, dim3(q), dim3(256), 0, stream, bank_bytes, ksteps, static_cast<const T*>(queries),
                       ldq, d, p.kp, k, index_base, w.carry_r, w.carry_n, out_s, out_i, (int)n, status);
    return isc_launch_status();
}

int check_args(int dtype, int64_t n, int d, int q, int k) {
    if (dtype != ISC_F16 && dtype != ISC_F32) return ISC_ERR_INVALID_ARG;
    if (n <= 0 || d <= 0 || q <= 0 || k <= 0 || k > n) return ISC_ERR_INVALID_ARG;
    if (k > ISC_TOPK_MAX_K) return ISC_ERR_UNSUPPORTED;
    if (n > 0x7fffffff) return ISC_ERR_UNSUPPORTED;  // row ids are int32 inside a shard
    if (d > 65536) return ISC_ERR_UNSUPPORTED;
    if (isc_ceil_div(q, 64) > 65535) return ISC_ERR_UNSUPPORTED;
    return ISC_OK;
}

}  // namespace

extern "C" int isc_cosine_topk_workspace_bytes(int dtype, int64_t N, int D, int Q, int k, size_t* bytes) {
    ISC_REQUIRE(bytes);
    const int st = check_args(dtype, N, D, Q, k);
    if (st != ISC_OK) return st;
    *bytes = carve(make_plan(N, Q, k), isc_ksteps(D, dtype == ISC_F16 ? 2 : 4), nullptr).bytes;
    return ISC_OK;
}

extern "C" int isc_cosine_topk(const void* bank, int dtype, int64_t N, int D, const void* queries, int Q, int64_t ldq,
                               int k, int64_t index_base, float* out_scores, int64_t* out_indices, int32_t* status,
                               void* workspace, size_t workspace_bytes, void* stream) {
    ISC_REQUIRE(bank && queries && out_scores && out_indices && status);
    const int st = check_args(dtype, N, D, Q, k);
    if (st != ISC_OK) return st;
    ISC_REQUIRE(ldq >= D);
    if (!isc_aligned(bank, 16) || !isc_aligned(workspace, 256)) return ISC_ERR_ALIGNMENT;
    size_t need = 0;
    isc_cosine_topk_workspace_bytes(dtype, N, D, Q, k, &need);
    if (!workspace || workspace_bytes < need) return ISC_ERR_WORKSPACE;
    if (dtype == ISC_F16)
        return run<_Float16>(bank, N, D, queries, Q, ldq, k, index_base, out_scores, out_indices, status, workspace,
                             isc_stream(stream));
    return run<float>(bank, N, D, queries, Q, ldq, k, index_base, out_scores, out_indices, status, workspace,
                      isc_stream(stream));
}
