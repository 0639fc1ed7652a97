// Brute-force cosine top-k over an embedding bank (include/imagescry_hip.h: isc_cosine_topk).
//
// Pipeline per call (all on one stream, NO host synchronisation and no host-side retry -- the result is final):
//
//   k_prep            per-query state reset + queries -> the packed K-step-major layout of the bank (L2 resident)
//   k_dots_filter<SAMPLE>   level 0: one 256-row tile per workgroup over a prefix of the (permuted, hence evenly
//                     sampled) bank.  Each workgroup bounds its own kp-th best score from below with the kp-th
//                     largest of its per-slot maxima (through LDS) and emits only the scores above that bound
//                     (~1.3 kp per query instead of 256).
//   k_select          one workgroup per query: candidates + the carried list -> the best kp by (score desc, packed
//                     row asc); tau <- the kp-th score.
//   for each later level (each up to ~kp-dependent ratio x everything before it):
//     k_dots_filter   S = bank[rows] . queries^T on the matrix cores.  The scores are never written: each lane
//                     compares its accumulators with the per-query threshold tau and appends the few survivors
//                     (score, row) to a small lane-private buffer; at its end every lane moves them into a compact
//                     per-query list (one atomic per lane and query block, outside the hot loop).
//     k_select        between levels
//   k_final           one workgroup per query: last selection, then the kp = k + slack carried candidates are
//                     re-scored EXACTLY (float64 dot, float64 query norm), rounded to float32, ordered by
//                     (score desc, ORIGINAL row asc); the first k are the result.  A rigorous guard then checks
//                     that no row the float32 filter dropped can belong to the answer:
//                         float32((T + eps) / ||q||) < score_k        T = the kp-th carried filter score,
//                         eps = Dpad * 2^-23 * ||q|| * max row norm    (bound of the fp32 accumulation error)
//                     Queries that fail it, or whose candidate buffers overflowed, are appended to a redo list.
//   k_exact           (search_exact.hip) the listed queries -- normally none: the kernel exits at once -- are
//                     searched again exhaustively in float64 and their output rows overwritten.
//
// The matrix-core pass only has to be a superset filter; ordering and the returned scores come from the exact
// pass, so the result does not depend on tile shape, accumulation order, chunking, level structure or sharding.
//
// Data layout.  The bank is PACKED (bank_layout.h): [tile of 256 rows][K step][row][128 B], rows in a fixed
// pseudo-random permutation of their original order, so the block one K step of one tile needs is 32 KiB of
// contiguous HBM, a workgroup's whole chunk is one linear stream, and every prefix is an even sample of the bank.
// The queries are packed the same way per call.  Both MFMA operands are "K-major", so the same staging code serves
// A (bank rows, the streamed operand) and B (queries).  One K step is 128 bytes of every row (64 halves or 32
// floats).  LDS tiles are [rows][128 B], the eight 16-byte chunks of a row XOR-swizzled with (row >> 1) & 7 so
// that a ds_read_b128 of an MFMA fragment (16 rows x 4 chunks per wave) is bank-conflict free; the LDS image is
// lane-linear in the staging order (the swizzle is applied to the LDS-DMA source address).
//
// Three tile shapes share the kernel template:
//   TNQ = 256  256 bank rows x 256 queries per workgroup, waves 2 x 4 (128 x 64 each): the MFMA-bound shape for
//              large query batches (arithmetic intensity 128 flop per staged byte);
//   TNQ = 64   256 bank rows x  64 queries, waves 8 x 1 (32 x 64 each), a 4-deep ring for both operands: the
//              HBM-bound shape for small query batches -- three 32 KiB bank blocks are in flight per CU while the
//              matrix cores idle most of the time;
//   TNQ = 128  256 bank rows x 128 queries, waves 4 x 2 (64 x 64 each), 3-deep rings (round 4): 64 < Q <= 128.  Two
//              64-query tiles made every CU pull the bank twice through its L2 -> LDS path (3.7 ms at 10 M rows, one and a
//              half times the 64-query search), the 256-query tile runs half empty; this one streams the bank once with
//              the matrix pipe ~45 % busy.
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "bank_layout.h"
#include "isc_common.h"
#include "search_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 256;        // bank rows per tile
constexpr int NTHREADS = 512;  // 8 waves
constexpr int CAP = 32;        // candidate slots per (segment, query); segment = (chunk, row-block wave, lane group)
constexpr int TARGET_WGS = 256;  // one workgroup per MI355X CU (the kernel uses all 160 KiB of LDS)
constexpr int QCAP = 8192;  // candidates per query per level that the compact list / the selection can hold
constexpr int SLACK = 6;
constexpr int SMALL_Q = 128;  // up to this many queries the HBM-bound tile shapes are used: 64 queries (Q <= 64) or 128
constexpr int QBATCH = ISC_SEARCH_PASS_QUERIES;  // queries per pass: larger calls run as several passes over the same workspace
constexpr int MAX_LEVELS = 12;
constexpr int SEL_THREADS = 512;
constexpr int SURV_CAP = 2048;  // candidates the selection's exact ranking step accepts
constexpr int TAIL_LCAP = 256;                       // k_dots_filter tail: LDS list entries per query (= rows of a tile)
constexpr int TAIL_COUNT_OFF = 64 * TAIL_LCAP * 8;   // ... byte offsets inside the (by then free) staging LDS
constexpr int TAIL_LBOUND_OFF = 140 * 1024;  // past the slot maxima of the sample epilogue ([256][128 + 4] floats = 132 KiB)

struct Cand {
    float s;
    int32_t row;
};

struct Level {
    int64_t r0, r1;
    int ntiles, tiles_per_chunk, nchunks;
    int sample;  // level 0
};

struct Plan {
    int tnq;             // queries per tile: 64, 128 or 256
    int segs_per_chunk;  // (8 / (tnq / 64)) row-block waves x 4 lane groups
    int kp;              // candidates carried per query (>= k + SLACK, multiple of 16)
    int nslots;          // level 0: per-query slots whose maxima bound the workgroup's kp-th score (0 = keep every score)
    int qb;              // queries per pass (<= QBATCH)
    int qtiles;          // ceil(qb / tnq)
    int qpad;            // qtiles * tnq
    int max_seg;         // segs_per_chunk * max chunks over the levels
    int nlevels;
    Level levels[MAX_LEVELS];
};

int plan_kp(int k) { return (int)isc_align_up((size_t)k + SLACK, 16); }

#ifdef ISC_ABLATION
int forced_tile() {
    static const int v = [] {
        const char* e = getenv("ISC_FORCE_TILE");  // ablation builds only: 64, 128 or 256
        return e ? atoi(e) : 0;
    }();
    return v;
}
#else
constexpr int forced_tile() { return 0; }
#endif

// Level structure.  Level 0 ("sample") is one tile per workgroup: with kp <= 64 every workgroup emits ~1.35 kp
// candidates per query (see k_dots_filter), so the number of tiles is capped to keep the per-query list at <= ~0.7
// QCAP; with kp > 64 it is 16 tiles whose every score is kept (4096 per query).  A later level that is R times the rows
// seen so far lets about kp * (R - 1) rows per query pass the threshold (the kp-th best score of those earlier rows --
// the bank is stored in a pseudo-random row order, so earlier rows are an even sample); R is chosen so that this stays
// below half the per-query list AND below 1/8 of what the lane-private segments of the level hold together.
#ifdef ISC_ABLATION
__device__ int g_abl_thr_inf = 0;  // set from ISC_THR_INF by run(): every non-sample level filters against +inf (wrong results)
int first_ratio() {
    static const int v = getenv("ISC_FIRST_RATIO") ? atoi(getenv("ISC_FIRST_RATIO")) : 16;  // A/B aid; 1 << 20 = off
    return v;
}
int sample_tiles_cap() {
    static const int v = getenv("ISC_SAMPLE_TILES") ? atoi(getenv("ISC_SAMPLE_TILES")) : 0;  // A/B aid: cap of the sample level's tiles
    return v;
}
#else
constexpr int first_ratio() { return 16; }
constexpr int sample_tiles_cap() { return 0; }
#endif

Plan make_plan(int64_t n, int q, int k) {
    Plan p;
    p.qb = q < QBATCH ? q : QBATCH;
    p.tnq = p.qb <= 64 ? 64 : p.qb <= SMALL_Q ? 128 : 256;
    if (forced_tile() == 64 || forced_tile() == 128 || forced_tile() == 256) p.tnq = forced_tile();
    p.segs_per_chunk = (8 / (p.tnq / 64)) * 4;
    p.kp = plan_kp(k);
    p.qtiles = isc_ceil_div(p.qb, p.tnq);
    p.qpad = p.qtiles * p.tnq;
    p.nslots = p.kp <= 16 ? 32 : p.kp <= 32 ? 64 : p.kp <= 64 ? 128 : 0;
    int wgs = TARGET_WGS / p.qtiles;
    if (wgs < 1) wgs = 1;
    const int64_t ntiles_all = isc_ceil_div<int64_t>(n, TM);

    p.nlevels = 0;
    p.max_seg = 0;
    auto add = [&](int64_t r0, int64_t r1, int sample) {
        Level& l = p.levels[p.nlevels++];
        l.r0 = r0;
        l.r1 = r1;
        l.sample = sample;
        l.ntiles = (int)isc_ceil_div<int64_t>(r1 - r0, TM);
        int want = sample ? l.ntiles : wgs;
        if (want > l.ntiles) want = l.ntiles;
        l.tiles_per_chunk = isc_ceil_div(l.ntiles, want);
        l.nchunks = isc_ceil_div(l.ntiles, l.tiles_per_chunk);
        if (p.segs_per_chunk * l.nchunks > p.max_seg) p.max_seg = p.segs_per_chunk * l.nchunks;
    };
    int64_t stiles = p.nslots ? (int64_t)(QCAP * 7 / 10) / (p.kp * 27 / 20) : 16;
    if (stiles > wgs) stiles = wgs;
    if (sample_tiles_cap() > 0 && stiles > sample_tiles_cap()) stiles = sample_tiles_cap();
    if (stiles < 1) stiles = 1;
    if (stiles > ntiles_all) stiles = ntiles_all;
    int64_t seen = stiles * TM < n ? stiles * TM : n;
    add(0, seen, 1);
    while (seen < n) {
        const int64_t nseg = (int64_t)p.segs_per_chunk * wgs;
        // The level's survivors per query are ~ (ratio - 1) x G with G ~ Gamma(kp): the threshold is the kp-th best of what
        // was seen, so the tail mass above it fluctuates by 1 / sqrt(kp).  Half the list for the MEAN (round 2) overflowed
        // the 8192-entry list of about one query in 500 at kp = 16 (P(G > 32) = 2e-3): with 512 queries per call nearly
        // every search of a bank that uses the full ratio paid a second pass for it.  The list now holds the mean plus
        // eight standard deviations (kp = 16: ratio 171, P < 1e-8).
        int64_t budget = (int64_t)((double)QCAP * p.kp / (p.kp + 8.0 * sqrt((double)p.kp)));
        if (nseg * CAP / 8 < budget) budget = nseg * CAP / 8;
        int64_t ratio = 1 + budget / p.kp;
        // 256-query shape with ONE query tile (128 < Q <= 256): the level after the sample would be the whole bank on
        // the sample's weak threshold -- a wave's ballot over 128 rows x 16 queries trips with probability
        // 1 - exp(-2048 kp / rows seen), 39 % after 65 536 rows, and every trip is a scan of the block.  A short
        // intermediate level buys a threshold that trips 3 % for the price of one k_select: 10 M rows at Q = 256
        // 4.17 -> 4.00 ms (scripts/ab.sh, ab_first_ratio).  With four query tiles (Q = 1024) the same level costs more than it
        // saves (13.34 -> 13.67 ms at ratio 8, no change at 16), so it is not used there.
        if (p.tnq == 256 && p.qtiles == 1 && p.nlevels == 1 && ratio > first_ratio()) ratio = first_ratio();
        if (ratio < 3) ratio = 3;
        int64_t r1 = seen * ratio;  // multiple of TM because `seen` is
        if (r1 > n || p.nlevels == MAX_LEVELS - 1) r1 = n;
        add(seen, r1, 0);
        seen = r1;
    }
    // the redo of unproven queries streams the WHOLE bank as one level (run()): its chunks need segments too
    const int redo_chunks = (int)(wgs < ntiles_all ? wgs : ntiles_all);
    if (p.segs_per_chunk * redo_chunks > p.max_seg) p.max_seg = p.segs_per_chunk * redo_chunks;
    return p;
}

struct Workspace {
    float* tau;              // [qpad]
    float* carry_s;          // [qpad][kp]  filter (float32) scores of the carried candidates
    int32_t* carry_r;        // [qpad][kp]  ... their packed rows
    int32_t* carry_n;        // [qpad]
    int32_t* qflag;          // [qpad]      != 0: a candidate buffer of this query overflowed -> exact redo
    Cand* seg_ent;           // [max_seg][qpad][CAP]  lane-private survivor segments (written in the hot loop)
    int32_t* qcount;         // [qpad]                survivors per query of the current level
    Cand* qlist;             // [qpad][QCAP]          ... compacted at the end of k_dots_filter
    unsigned char* qpacked;  // [qtiles][ks][tnq][128 B]
    // matrix-core redo of the queries whose first answer k_final cannot prove (see k_final / k_final2)
    int32_t* r_count;         // [1]     listed queries ("slots")
    int32_t* r_list;          // [qpad]  slot -> query of this pass
    float* tau2;              // [qpad]  slot -> fixed threshold of the redo filter (+inf: unused slot)
    int32_t* qcount2;         // [qpad]  slot -> survivors of the redo filter
    int32_t* qflag2;          // [qpad]  slot -> a candidate buffer overflowed
    unsigned char* qpacked2;  // [qtiles][ks][tnq][128 B]  packed query rows by slot
    IscExactWs exact;        // list + partial lists of k_exact (the exhaustive float64 pass, the last resort)
    size_t bytes;
};

Workspace carve(const Plan& p, int ks, int64_t n, int k, void* base) {
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
    w.qflag = static_cast<int32_t*>(take((size_t)p.qpad * 4));
    w.seg_ent = static_cast<Cand*>(take((size_t)p.max_seg * p.qpad * CAP * sizeof(Cand)));
    w.qcount = static_cast<int32_t*>(take((size_t)p.qpad * 4));
    w.qlist = static_cast<Cand*>(take((size_t)p.qpad * QCAP * sizeof(Cand)));
    w.qpacked = static_cast<unsigned char*>(take((size_t)p.qpad * ks * ISC_KSTEP_BYTES));
    w.r_count = static_cast<int32_t*>(take(4));
    w.r_list = static_cast<int32_t*>(take((size_t)p.qpad * 4));
    w.tau2 = static_cast<float*>(take((size_t)p.qpad * 4));
    w.qcount2 = static_cast<int32_t*>(take((size_t)p.qpad * 4));
    w.qflag2 = static_cast<int32_t*>(take((size_t)p.qpad * 4));
    w.qpacked2 = static_cast<unsigned char*>(take((size_t)p.qpad * ks * ISC_KSTEP_BYTES));
    const size_t ex = isc_exact_ws_bytes(n, p.qb, k);
    w.exact = isc_exact_ws_carve(take(ex), n, p.qb, k);
    w.bytes = off;
    return w;
}

// Per-query state reset and query packing in one launch.
// queries row-major [q][ldq] -> packed [qtile][K step][tnq rows][128 B]; rows >= q and columns >= d are zero.
// One thread per 16-byte chunk; the first qpad threads also reset the per-query words.
// TQ = element type of the caller's queries: they are rounded to the bank type T while they are packed (float32 -> fp16:
// v_cvt_f16_f32, round to nearest even = `Tensor.to(float16)`; fp16 -> float32 is exact), so no cast kernel runs in front.
template <typename T, typename TQ>
__global__ __launch_bounds__(256) void k_prep(const TQ* __restrict__ queries, int64_t ldq, int q, int d, int ks, int qpad,
                                              int tnq, unsigned char* __restrict__ packed, float* __restrict__ tau,
                                              int32_t* __restrict__ carry_n, int32_t* __restrict__ qcount,
                                              int32_t* __restrict__ qflag, int32_t* __restrict__ redo_count,
                                              int32_t* __restrict__ exact_done, int32_t* __restrict__ r_count,
                                              float* __restrict__ tau2, int32_t* __restrict__ qcount2,
                                              int32_t* __restrict__ qflag2, int32_t* __restrict__ status,
                                              int zero_status) {
    constexpr int PER = 16 / (int)sizeof(T);
    const int total = qpad * ks * 8;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < qpad) {
        tau[i] = i < q ? -INFINITY : INFINITY;  // padding queries never pass the filter
        carry_n[i] = 0;
        qcount[i] = 0;
        qflag[i] = 0;
        tau2[i] = INFINITY;  // redo slots: unused ones never pass the redo filter
        qcount2[i] = 0;
        qflag2[i] = 0;
    }
    if (i == 0) {
        *redo_count = 0;
        *exact_done = 0;
        *r_count = 0;
    }
    if (zero_status && i < 4) status[i] = 0;
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
        v[j] = (qrow < q && e < d) ? (T)queries[(int64_t)qrow * ldq + e] : (T)0.f;
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
    // query blocks N0 .. N1 - 1 of `half` (the same instructions in the same order per accumulator)
    template <int N0, int N1>
    static __device__ __forceinline__ void part(const u32x4& a, const u32x4 (&b)[4], f32x4 (&acc)[4]) {
#pragma unroll
        for (int n = N0; n < N1; ++n)
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a),
                                                            __builtin_bit_cast(half8, b[n]), acc[n], 0, 0, 0);
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
    template <int N0, int N1>
    static __device__ __forceinline__ void part(const u32x4& a, const u32x4 (&b)[4], f32x4 (&acc)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = N0; n < N1; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[j]), __uint_as_float(b[n][j]), acc[n],
                                                              0, 0, 0);
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
// ... with the non-temporal cache policy (aux = 2): for bank bytes that exactly one workgroup reads once per search
template <bool NT>
__device__ __forceinline__ void glds16_bank(const unsigned char* gsrc, unsigned char* lds_wave_base) {
    if constexpr (NT)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                         (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2);
    else
        glds16(gsrc, lds_wave_base);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt range");
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

constexpr int A_TILE_BYTES = TM * 128;  // 32 KiB: one K step of one bank tile

// Tuning constants of the 256-query loop.  They are macros so that `python -m imagescry_amd.build --variant=<name> -D...`
// can build A/B libraries (scripts/ab.sh: ab_asym, ab_hm); the values below are what the measurements kept.
#ifndef ISC_ASYM_MBLO
#define ISC_ASYM_MBLO 6  // row blocks of a wm = 0 wave (of 16 per tile) in launches with several query tiles
#endif
#ifndef ISC_ASYM_SPLIT_MBLO
#define ISC_ASYM_SPLIT_MBLO 8  // ... in the single-query-tile SPLIT form (symmetric)
#endif
#ifndef ISC_HM_RING
#define ISC_HM_RING 4  // half-major loop: bank-fragment registers (reads run RING - 1 units ahead)
#endif
#ifndef ISC_HM_DMA0
#define ISC_HM_DMA0 1  // ... first unit that carries an LDS-DMA piece
#endif
#ifndef ISC_HM_B1U
#define ISC_HM_B1U 2  // ... first unit under which the second half's query fragments are fetched
#endif

// One workgroup = one query tile x one chunk of consecutive 256-row bank tiles.
//
// Pipeline: the K steps of all the chunk's tiles form one stream.  Iteration s issues the LDS-DMA of query step
// s + DB and bank step s + DA, computes step s, then waits with a COUNTED vmcnt (only what step s + 1 needs is
// retired; later steps stay in flight) and a raw s_barrier -- DA K steps of HBM latency are covered without holding
// a single staging register.  Ordering rules (cdna_hip_programming.md, "Pipelining across barriers"): a slot is
// read one iteration after the vmcnt + barrier that retires its DMA, and refilled one barrier after its last read.
//
// SAMPLE = level 0: the workgroup owns exactly ONE tile; nothing is filtered inside the loop, the epilogue after it bounds
// the workgroup's kp-th best score from below and emits what lies above the bound (see there).
//
// DBG: 0 = production, 12 = production with every wave issuing its own share of the LDS-DMA (used when several query-tile
// workgroups stream the same chunk).  The other values exist only in -DISC_ABLATION builds (wrong results by design):
// 2 = no staging after the prologue, 3 = staging but no MFMAs, 41 / 43 = the query operand ablations, 7 = like 2 without LDS fragment reads, 11 = no
// half-row-block stagger of the wm = 1 waves, 15 = DMA issued but never waited for, 17 = DMA and MFMAs but no LDS
// fragment reads.
template <typename T, int TNQ, int DBG, bool SAMPLE>
__global__ __launch_bounds__(NTHREADS) void k_dots_filter(const unsigned char* __restrict__ bank, int64_t r0,
                                                          int64_t r1, int tiles_per_chunk, int ntiles,
                                                          const unsigned char* __restrict__ qpacked, int ksteps,
                                                          const float* __restrict__ tau, int qpad,
                                                          Cand* __restrict__ seg_ent, int32_t* __restrict__ qcount,
                                                          Cand* __restrict__ qlist, int kp, int nslots,
                                                          int32_t* __restrict__ qflag, int32_t* __restrict__ status,
                                                          const int32_t* __restrict__ active) {
    // DBG 20 / 32 / 33 are the REDO instantiations of 0 / 12 / 13 (k_final2's feeder: the whole bank against the fixed
    // thresholds of the listed queries).  They are kernels of their own so that a profile lists them apart from the
    // search's streaming launches; `active` counts the listed query slots -- normally zero, and the launch ends here.
    constexpr bool REDO = DBG == 20 || DBG == 32 || DBG == 33;
    constexpr int MODE = DBG == 20 ? 0 : DBG == 32 ? 12 : DBG == 33 ? 13 : DBG;
    if constexpr (REDO) {
        if (*active <= (int)blockIdx.y * TNQ) return;
    }
    constexpr int WN = TNQ / 64;              // waves along the queries
    constexpr int WM = 8 / WN;                // waves along the bank rows
    constexpr int MB = TM / WM / 16;          // 16-row blocks per wave: 8 (TNQ 256) or 2 (TNQ 64)
    // ASYMMETRIC row split of the 256-query filter loop.  The two waves of a SIMD share its matrix pipe, and the arbiter
    // (priority, then age) serves one of them first: in-kernel stamps (scripts/stamp_search.py) show that wave done with
    // its K step after ~1 940 cycles and idle at the barrier for the last third of the step, while its partner crawls on
    // (2 890 cycles; without the static priority the roles swap, the picture stays).  So the favoured waves own MBHI of the
    // tile's 16 row blocks and the others MBLO.  Sample launches (one tile per workgroup) and the 64-query shape stay
    // symmetric.
    // Ablations of the query operand (what do its LDS bytes cost?  LABLOG.md, scripts/ab.sh: ab_noq): 41 = neither staged nor
    // read (garbage fragments), 43 = staged (LDS-DMA writes) but never read.
    constexpr bool NOBREAD = (MODE == 41 || MODE == 43) && TNQ == 256;
    constexpr bool NOBDMA = MODE == 41 && TNQ == 256;
    // HM: the K step in half-major order -- [all row blocks x first 32 k] [all row blocks x second 32 k] -- so that only
    // the first half's query fragments stand between the barrier and the first MFMA of a step (the second half's are
    // fetched under the first units), and the bank fragments run three units ahead through four registers.
    // It is the form of every launch with several query tiles (MODE 12); MODE 48 (ablation builds) = the row-block-major
    // form it replaced (+1.6 %, scripts/ab.sh: ab_hm), which the stamped build (22) and the operand ablations (41, 43) still run.
    constexpr bool HM = MODE == 12 && TNQ == 256 && !SAMPLE;
    constexpr bool NOSPLIT_FORM = MODE == 12 || MODE == 22 || MODE == 41 || MODE == 43 || MODE == 48;
    constexpr int ASYM_LO = NOSPLIT_FORM ? ISC_ASYM_MBLO : MODE == 0 ? ISC_ASYM_SPLIT_MBLO : 8;
    constexpr bool ASYM = TNQ == 256 && !SAMPLE && ASYM_LO != 8;
    constexpr int MBLO = ASYM ? ASYM_LO : MB;        // row blocks of a wm = 0 wave
    constexpr int MBHI = ASYM ? 2 * MB - MBLO : MB;  // ... of a wm = 1 wave
    constexpr int MBMAX = MBHI > MBLO ? MBHI : MBLO;
    constexpr int B_TILE_BYTES = TNQ * 128;   // one K step of the query tile
    constexpr int NA = 4;                     // LDS-DMA instructions per thread per bank step (512 x 16 B x 4)
    constexpr int NB = B_TILE_BYTES / 8192;   // ... per query step: 4 or 1
    constexpr int A_ST = TNQ == 64 ? 4 : 3;  // ring depths: together exactly 160 KiB (TNQ = 128: 144 KiB)
    constexpr int B_ST = TNQ == 256 ? 2 : TNQ == 128 ? 3 : 4;
    constexpr int DA = A_ST - 1;  // prefetch distances, in K steps
    constexpr int DB = B_ST - 1;
    constexpr int LDS_BYTES = A_ST * A_TILE_BYTES + B_ST * B_TILE_BYTES;
    static_assert(LDS_BYTES == (TNQ == 128 ? 147456 : 163840), "the two rings fill the CU's LDS (TNQ = 128: 144 KiB of it)");
    static_assert(TNQ == 256 || DA == DB, "the HBM-bound shapes stage query and bank steps at one prefetch distance");
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
        if ((MODE != 0 && MODE < 11) || (MODE >= 15 && MODE != 48)) thr[n] = fabsf(thr[n]) + 3.0e38f;  // ablations: nothing survives (kept opaque to the optimiser)
#ifdef ISC_ABLATION
        if (!SAMPLE && g_abl_thr_inf) thr[n] = fabsf(thr[n]) + 3.0e38f;  // ISC_THR_INF: the price of scanning + the tail
#endif
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
    // Q = 256; MODE 12 = every wave issues its own share, the A/B reference)
    constexpr bool SPLIT = TNQ == 256 && !NOSPLIT_FORM;  // 22 = 12 + in-kernel stamps (ablation builds)
    // NT: the launch has ONE query tile, so every bank byte is read by exactly one workgroup, once: stream it with the
    // non-temporal policy (MODE 0 = the single-query-tile form of the 256-query shape, MODE 13 = of the 64-query shape)
    constexpr bool NT = (TNQ == 256 && MODE == 0) || MODE == 13;
    auto issue_a = [&](int step) {
        const unsigned char* src = a_stream + (int64_t)step * A_TILE_BYTES;
        unsigned char* dst = lds_a + (step % A_ST) * A_TILE_BYTES + wave_dst;
        if constexpr (SPLIT) {
            if (wm == 0) {
#pragma unroll
                for (int i = 0; i < 2 * NA; ++i) glds16_bank<NT>(src + 4096 * i, dst + 4096 * i);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) glds16_bank<NT>(src + 8192 * i, dst + 8192 * i);
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
        if ((MODE == 2 || MODE == 7) && it >= 0) return;
        const int sb = it + DB, sa = it + DA;
        if (!NOBDMA && sb >= 0 && sb < total_steps) issue_b(sb);
        if (sa >= 0 && sa < total_steps) issue_a(sa);
    };
    // after iteration `next - 1` has issued: retire everything step `next` needs, leave the younger DMA in flight
    auto retire_for = [&](int next) {
        if (MODE == 2 || MODE == 7) {
            wait_vmcnt<0>();
            return;
        }
        if (MODE == 15) return;  // ablation: the DMA is issued but never waited for (stale operands, wrong results)
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
        } else {  // stream order ... B(next) A(next) | B(next+1) A(next+1) | B(next+2) A(next+2): DA - 1 groups stay in flight
            const int ahead = min(DA - 1, total_steps - 1 - next);
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
    const int wave_row0 = ASYM ? (wm == 0 ? 0 : MBLO * 16) : wm * (TM / WM);  // first tile row of this wave
    const int a_wave_off = wave_row0 * 128;
    const int b_wave_off = wn * 64 * 128;

    f32x4 acc[MBMAX][4];
#pragma unroll
    for (int m = 0; m < MBMAX; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int seg = (chunk * WM + wm) * 4 + fg;
    Cand* my_ent = seg_ent + ((size_t)seg * qpad + q0 + wn * 64 + frow) * CAP;  // + n * 16 * CAP

    // Static priority for the second-dispatched half of the workgroup (waves 4 - 7, the wm = 1 partners of every SIMD in
    // the 256-query shape): that half loses the SIMD's issue arbitration to the older half on every segment
    // (MI355X_MICROARCH.md, "Two waves per SIMD" item 4); one s_setprio before the loop, no flips inside.  +1.0 % at
    // Q = 1024 (scripts/ab.sh: ab_headline, three interleaved rounds: 13.71 -> 13.57 ms); per-cluster flips lost 5 % in round 1.
    // The condition must be provably wave-uniform: s_setprio is a scalar instruction that ignores EXEC.
    // Not with SPLIT (MODE 0: only the older half issues the LDS-DMA there, and prioritising the other half on top of
    // that cost 4 %).
    if constexpr (TNQ == 256 && NOSPLIT_FORM) {
#ifdef ISC_ABLATION
        if (!((nslots >> 18) & 1))
#endif
            if (__builtin_amdgcn_readfirstlane(tid) >= 256) __builtin_amdgcn_s_setprio(1);
    }
    // prologue: what iterations -DA .. -1 would have issued; then publish step 0
    for (int it = -DA; it < 0; ++it) issue_iter(it);
    retire_for(0);
    __builtin_amdgcn_s_barrier();

    // MODE 22 (ablation builds): s_memtime stamps split every K step of a wave into [reads + MFMA issue] [counted vmcnt
    // wait] [barrier]; the sums go to seg_ent (unused: thresholds are +inf in this mode).  Read the SHARES, not the length.
    unsigned long long st_compute = 0, st_vmwait = 0, st_barrier = 0, st_prev = 0;
    auto stamp = [&]() -> unsigned long long {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        return t;
    };
    // The main loop exists in two instantiations (see "type B" below); the branch is taken once, outside the loop,
    // so neither version pays for the other's registers.
    auto main_loop = [&](auto stagger_tag) {
    constexpr bool STAGGER = decltype(stagger_tag)::value;
    int kt = 0, tile = 0;
    if constexpr (MODE == 22) st_prev = stamp();
    for (int step = 0; step < total_steps; ++step) {
        if constexpr (HM) {
            // ---- 256-query shape, half-major.  A K step is 2 MBW units: unit u = (row block u % MBW, half u / MBW), four
            // MFMAs (fp16) each.  LDS read stream of a wave and step (the LDS returns in order; every counted wait below is
            // derived from it): [b0 x 4, a(0), a(1), a(2)] then per unit v: [b1[v] if v < 4] [a(v + 3) if it exists].  Bank
            // fragments run three units ahead through a ring of four registers; the second half's query fragments are
            // fetched under the first four units.  The eight LDS-DMA pieces (query step, then bank step) are spread over
            // the units after the first.
            constexpr int MBW = STAGGER ? MBHI : MBLO;
            constexpr int NU = 2 * MBW;
            static_assert(MBW >= 4, "the second half's query fragments are fetched under four units of the first");
            const int sb = step + DB, sa = step + DA;
            const bool do_b = sb < total_steps;
            const bool do_a = sa < total_steps;
            const unsigned char* bsrc = b_stream + (int64_t)(sb % ksteps) * B_TILE_BYTES;
            unsigned char* bdst = lds_b + (sb % B_ST) * B_TILE_BYTES + wave_dst;
            const unsigned char* asrc = a_stream + (int64_t)sa * A_TILE_BYTES;
            unsigned char* adst = lds_a + (sa % A_ST) * A_TILE_BYTES + wave_dst;
            const unsigned a_addr = lds_a_addr + (unsigned)((step % A_ST) * A_TILE_BYTES + a_wave_off);
            const unsigned b_addr = lds_b_addr + (unsigned)((step % B_ST) * B_TILE_BYTES + b_wave_off);
            const unsigned a_addr0 = a_addr + foff[0], a_addr1 = a_addr + foff[1];
            const unsigned b_addr0 = b_addr + foff[0], b_addr1 = b_addr + foff[1];
            constexpr int RING = ISC_HM_RING;  // bank-fragment registers: reads run RING - 1 units ahead
            u32x4 b0[4], b1[4], ar[RING];
            ISC_DS_READ(b0[0], b_addr0, 0);
            ISC_DS_READ(b0[1], b_addr0, 2048);
            ISC_DS_READ(b0[2], b_addr0, 4096);
            ISC_DS_READ(b0[3], b_addr0, 6144);
            auto read_unit = [&](auto uc) {
                constexpr int u = decltype(uc)::value, m = u % MBW, kk = u / MBW;
                auto& frag = ar;  // (named outside the asm so that the lambda captures them)
                const unsigned addr = kk == 0 ? a_addr0 : a_addr1;
                ISC_DS_READ(frag[u % RING], addr, m * 2048);
            };
            read_unit(std::integral_constant<int, 0>{});
            read_unit(std::integral_constant<int, 1>{});
            read_unit(std::integral_constant<int, 2>{});
            if constexpr (RING > 4) read_unit(std::integral_constant<int, 3>{});
            if constexpr (RING > 5) read_unit(std::integral_constant<int, 4>{});
            static_assert(RING >= 4 && RING <= 6, "prologue reads");
            // reads the overhead of unit v issues, and how many reads are newer than a(w) once the overhead of unit u is out
            constexpr int B1U = ISC_HM_B1U < MBW - 4 ? ISC_HM_B1U : MBW - 4;  // b1[3] is out before unit MBW's fragments are waited for
            constexpr int DMA0 = ISC_HM_DMA0;
            // the read stream: [b0 x 4][a(0 .. RING-2)] then per overhead v: [b1[v - B1U] if B1U <= v < B1U + 4][a(v + RING - 1)]
            auto reads_of = [](int v) constexpr { return (v >= B1U && v < B1U + 4 ? 1 : 0) + (v + RING - 1 < NU ? 1 : 0); };
            auto issued_through = [reads_of](int u) constexpr {
                int c = 4 + RING - 1;
                for (int v = 0; v <= u; ++v) c += reads_of(v);
                return c;
            };
            auto pos_a = [issued_through](int w) constexpr {  // position of a(w) in the stream
                if (w < RING - 1) return 4 + w;
                return issued_through(w - (RING - 1)) - 1;  // the last read of that overhead
            };
            auto pos_b1_last = [issued_through]() constexpr {  // b1[3]: first read of overhead B1U + 3
                return issued_through(B1U + 2);
            };
            // reads that may still be in flight when unit w starts, once overhead u is out
            auto newer_than = [=](int w, int u) constexpr {
                int need = pos_a(w);
                if (w >= MBW && pos_b1_last() > need) need = pos_b1_last();
                return issued_through(u) - 1 - need;
            };
            // LDS-DMA piece j (j < 4: query step, then the bank step) goes out in the overhead of unit 1 + j (NU - 2) / 8
            auto overhead = [&](auto uc, auto wc) {
                constexpr int u = decltype(uc)::value, w = decltype(wc)::value;  // w: the unit whose fragment must be there
                if constexpr (u >= B1U && u < B1U + 4) {
                    auto& q1 = b1;
                    const unsigned addr = b_addr1;
                    ISC_DS_READ(q1[u - B1U], addr, (u - B1U) * 2048);
                }
                if constexpr (u + RING - 1 < NU) read_unit(std::integral_constant<int, u + RING - 1>{});
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (DMA0 + j * (NU - 1 - DMA0) / 8 != u) continue;
                    if (j < 4) {
                        if (do_b) glds16(bsrc + 8192 * j, bdst + 8192 * j);
                    } else {
                        if (do_a) glds16_bank<NT>(asrc + 8192 * (j - 4), adst + 8192 * (j - 4));
                    }
                }
                if constexpr (w < NU) {
                    auto& frag = ar;
                    auto& q0 = b0;
                    auto& q1 = b1;
                    if constexpr (w == 0)
                        asm volatile("s_waitcnt lgkmcnt(%5)"
                                     : "+v"(frag[0]), "+v"(q0[0]), "+v"(q0[1]), "+v"(q0[2]), "+v"(q0[3])
                                     : "i"(newer_than(0, u)));
                    else if constexpr (w == MBW)
                        asm volatile("s_waitcnt lgkmcnt(%5)"
                                     : "+v"(frag[w % RING]), "+v"(q1[0]), "+v"(q1[1]), "+v"(q1[2]), "+v"(q1[3])
                                     : "i"(newer_than(w, u)));
                    else
                        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag[w % RING]) : "i"(newer_than(w, u)));
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if constexpr (STAGGER) {
                // type B: [MFMAs 0-1 of unit u] [overhead, wait for unit u + 1] [MFMAs 2-3 of unit u]
                asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(ar[0]), "+v"(b0[0]), "+v"(b0[1]), "+v"(b0[2]), "+v"(b0[3]) : "i"(RING - 2));
                __builtin_amdgcn_sched_barrier(0);
                auto units = [&](auto self, auto uc) {
                    constexpr int u = decltype(uc)::value, m = u % MBW, kk = u / MBW;
                    if constexpr (kk == 0) Mma<T>::template part<0, 2>(ar[u % RING], b0, acc[m]);
                    else Mma<T>::template part<0, 2>(ar[u % RING], b1, acc[m]);
                    overhead(uc, std::integral_constant<int, u + 1>{});
                    if constexpr (kk == 0) Mma<T>::template part<2, 4>(ar[u % RING], b0, acc[m]);
                    else Mma<T>::template part<2, 4>(ar[u % RING], b1, acc[m]);
                    if constexpr (u + 1 < NU) self(self, std::integral_constant<int, u + 1>{});
                };
                units(units, std::integral_constant<int, 0>{});
            } else {
                // type A: [overhead, wait for unit u] [MFMAs 0-3 of unit u]
                auto units = [&](auto self, auto uc) {
                    constexpr int u = decltype(uc)::value, m = u % MBW, kk = u / MBW;
                    overhead(uc, uc);
                    if constexpr (kk == 0) Mma<T>::template part<0, 4>(ar[u % RING], b0, acc[m]);
                    else Mma<T>::template part<0, 4>(ar[u % RING], b1, acc[m]);
                    if constexpr (u + 1 < NU) self(self, std::integral_constant<int, u + 1>{});
                };
                units(units, std::integral_constant<int, 0>{});
            }
        } else if constexpr (TNQ == 256 && MODE != 7 && MODE != 17) {
            // ---- 256-query shape.  One K step = MBW row blocks of 8 MFMAs (fp16) for this wave.  Fragment reads run two
            // row blocks ahead of the matrix cores (LDS returns in order: lgkmcnt(4) = "all but the newest two blocks"), and
            // the eight LDS-DMA instructions of this iteration are spread over the row blocks, so their issue cost hides
            // behind MFMAs instead of delaying the first one.  DMA stream order (the counted vmcnt relies on it): the query
            // step first, then the bank step.
            constexpr int MBW = STAGGER ? MBHI : MBLO;  // row blocks of this wave (the wm = 1 waves run the STAGGER copy)
            static_assert(MBW >= 3 && MBW <= MBMAX, "row blocks per wave");
            const int sb = step + DB, sa = step + DA;
            const bool do_b = MODE != 2 && !NOBDMA && sb < total_steps;
            const bool do_a = MODE != 2 && sa < total_steps;
            const unsigned char* bsrc = b_stream + (int64_t)(sb % ksteps) * B_TILE_BYTES;
            unsigned char* bdst = lds_b + (sb % B_ST) * B_TILE_BYTES + wave_dst;
            const unsigned char* asrc = a_stream + (int64_t)sa * A_TILE_BYTES;
            unsigned char* adst = lds_a + (sa % A_ST) * A_TILE_BYTES + wave_dst;
            const unsigned a_addr = lds_a_addr + (unsigned)((step % A_ST) * A_TILE_BYTES + a_wave_off);
            const unsigned b_addr = lds_b_addr + (unsigned)((step % B_ST) * B_TILE_BYTES + b_wave_off);
            const unsigned a_addr0 = a_addr + foff[0], a_addr1 = a_addr + foff[1];
            const unsigned b_addr0 = b_addr + foff[0], b_addr1 = b_addr + foff[1];
            u32x4 b0[4], b1[4], ar[3][2];
            constexpr int NBR = NOBREAD ? 0 : 4;  // query fragment reads per half K step
            if constexpr (NOBREAD) {
#pragma unroll
                for (int i = 0; i < 4; ++i) b0[i] = b1[i] = u32x4{(unsigned)step, 1u, 2u, (unsigned)lane};
                ISC_DS_READ(ar[0][0], a_addr0, 0);
                ISC_DS_READ(ar[0][1], a_addr1, 0);
            } else {
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
            }
            ISC_DS_READ(ar[1][0], a_addr0, 2048);  // R2
            ISC_DS_READ(ar[1][1], a_addr1, 2048);
            // LDS-DMA piece j of this iteration (j < 4: the query step, then the bank step).  SPLIT: the wm = 0 loop issues two
            // pieces per j, the wm = 1 loop none (issuing the whole query step in block 0 instead measured 1 % slower).
            auto dma_piece = [&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (SPLIT) {
                    if constexpr (!STAGGER) {
                        if constexpr (j < 4) {
                            if (do_b) {
                                glds16(bsrc + 4096 * (2 * j), bdst + 4096 * (2 * j));
                                glds16(bsrc + 4096 * (2 * j + 1), bdst + 4096 * (2 * j + 1));
                            }
                        } else if (do_a) {
                            glds16_bank<NT>(asrc + 4096 * (2 * (j - 4)), adst + 4096 * (2 * (j - 4)));
                            glds16_bank<NT>(asrc + 4096 * (2 * (j - 4) + 1), adst + 4096 * (2 * (j - 4) + 1));
                        }
                    }
                } else if constexpr (j < 4) {
                    if (do_b) glds16(bsrc + 8192 * j, bdst + 8192 * j);
                } else {
                    if (do_a) glds16_bank<NT>(asrc + 8192 * (j - 4), adst + 8192 * (j - 4));
                }
            };
            // the eight pieces in stream order over the MBW row blocks: the first 8 - MBW blocks take two (MBW < 8), blocks
            // past the eighth none (MBW > 8)
            auto dma_at = [&](auto mc) {
                constexpr int m = decltype(mc)::value;
                constexpr int extra = MBW < 8 ? 8 - MBW : 0;
                if constexpr (m < extra) {
                    dma_piece(std::integral_constant<int, 2 * m>{});
                    dma_piece(std::integral_constant<int, 2 * m + 1>{});
                } else if constexpr (m + extra < 8) {
                    dma_piece(std::integral_constant<int, m + extra>{});
                }
            };
#define ISC_MFMA_HALF(a_, b_, m_)                                                                         \
    if constexpr (MODE != 3) Mma<T>::half(a_, b_, acc[m_]);                                                 \
    else acc[m_][0][0] += __uint_as_float((a_)[0] ^ (b_)[1][1] ^ (b_)[2][2]);
            // The two waves that share a SIMD (waves w and w + 4, i.e. wm = 0 and wm = 1) run the same program between
            // the same barriers; left alone they reach their read / wait / DMA-issue instructions together and the
            // matrix pipe idles meanwhile.  The wm = 1 waves therefore run a version shifted by half a row block:
            // their overhead instructions sit between the two halves of a row block, the wm = 0 waves' between row
            // blocks, so one partner is always issuing MFMAs.
            if constexpr (STAGGER) {
                // ---- type B: [first half of block m] [reads m + 2, DMA, wait for block m + 1] [second half of block m]
                asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(b0[0]), "+v"(b0[1]), "+v"(b0[2]), "+v"(b0[3]), "+v"(ar[0][0]) : "i"(NBR + 3));
                __builtin_amdgcn_sched_barrier(0);
                ISC_MFMA_HALF(ar[0][0], b0, 0)
                ISC_DS_READ(ar[2][0], a_addr0, 4096);  // R3
                ISC_DS_READ(ar[2][1], a_addr1, 4096);
                dma_at(std::integral_constant<int, 0>{});
                asm volatile("s_waitcnt lgkmcnt(2)"
                             : "+v"(b1[0]), "+v"(b1[1]), "+v"(b1[2]), "+v"(b1[3]), "+v"(ar[0][1]), "+v"(ar[1][0]),
                               "+v"(ar[1][1]));
                __builtin_amdgcn_sched_barrier(0);
                ISC_MFMA_HALF(ar[0][1], b1, 0)
                auto blocks = [&](auto self, auto mc) {
                    constexpr int m = decltype(mc)::value;
                    constexpr int cur = m % 3, nxt = (m + 1) % 3, nn = (m + 2) % 3;
                    ISC_MFMA_HALF(ar[cur][0], b0, m)
                    if constexpr (m + 2 < MBW) {
                        ISC_DS_READ(ar[nn][0], a_addr0, (m + 2) * 2048);
                        ISC_DS_READ(ar[nn][1], a_addr1, (m + 2) * 2048);
                    }
                    dma_at(mc);
                    if constexpr (m + 1 < MBW) {
                        asm volatile("s_waitcnt lgkmcnt(%3)"
                                     : "+v"(ar[nxt][0]), "+v"(ar[nxt][1]), "+v"(ar[cur][1])
                                     : "i"(m + 2 < MBW ? 2 : 0));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    ISC_MFMA_HALF(ar[cur][1], b1, m)
                    if constexpr (m + 1 < MBW) self(self, std::integral_constant<int, m + 1>{});
                };
                blocks(blocks, std::integral_constant<int, 1>{});
            } else {
                // ---- type A.  row block 0: its two halves arrive separately
                ISC_DS_READ(ar[2][0], a_addr0, 4096);  // R3
                ISC_DS_READ(ar[2][1], a_addr1, 4096);
                asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(b0[0]), "+v"(b0[1]), "+v"(b0[2]), "+v"(b0[3]), "+v"(ar[0][0]) : "i"(NBR + 5));
                __builtin_amdgcn_sched_barrier(0);
                dma_at(std::integral_constant<int, 0>{});
                ISC_MFMA_HALF(ar[0][0], b0, 0)
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(b1[0]), "+v"(b1[1]), "+v"(b1[2]), "+v"(b1[3]), "+v"(ar[0][1]));
                __builtin_amdgcn_sched_barrier(0);
                ISC_MFMA_HALF(ar[0][1], b1, 0)
                auto blocks = [&](auto self, auto mc) {
                    constexpr int m = decltype(mc)::value;
                    constexpr int cur = m % 3, nn = (m + 2) % 3;
                    if constexpr (m + 2 < MBW) {
                        ISC_DS_READ(ar[nn][0], a_addr0, (m + 2) * 2048);
                        ISC_DS_READ(ar[nn][1], a_addr1, (m + 2) * 2048);
                    }
                    asm volatile("s_waitcnt lgkmcnt(%2)"
                                 : "+v"(ar[cur][0]), "+v"(ar[cur][1])
                                 : "i"(m + 2 < MBW ? 4 : m + 1 < MBW ? 2 : 0));
                    __builtin_amdgcn_sched_barrier(0);
                    dma_at(mc);
                    ISC_MFMA_HALF(ar[cur][0], b0, m)
                    ISC_MFMA_HALF(ar[cur][1], b1, m)
                    if constexpr (m + 1 < MBW) self(self, std::integral_constant<int, m + 1>{});
                };
                blocks(blocks, std::integral_constant<int, 1>{});
            }
#undef ISC_MFMA_HALF
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
            if constexpr (MODE == 7 || MODE == 17) {  // no LDS traffic: feed the matrix cores from whatever the registers hold
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
        if constexpr (MODE != 3)                                                                                      \
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

        if constexpr (!SAMPLE) {
        if (++kt == ksteps) {
            // ---- tile finished: threshold filter.  C layout of the 16x16 MFMA: column (query) = lane & 15,
            // row (bank row) = 4 * (lane >> 4) + register.  Survivors are rare once tau is warm, so the scan of a
            // query block only runs when some lane of the wave holds one (wave-uniform branch).  The comparison is
            // strict; rows that merely tie tau are dropped, which the exact pass's guard accounts for (a dropped row's
            // filter score is <= the kp-th carried one either way) -- while ">=" let every row of a zero query, or every
            // copy of a duplicated row, through and overflowed the buffers.
            kt = 0;
            constexpr int MBE = STAGGER ? MBHI : MBLO;  // this wave's row blocks (MB when the split is symmetric)
            const int64_t trow0 = r0 + (int64_t)(tile_begin + tile) * TM + wave_row0 + fg * 4;
            // rows past the end of the level exist only in its last tile: they become -inf there, once, instead of
            // being tested per element
            if (r0 + (int64_t)(tile_begin + tile + 1) * TM > r1) {
#pragma unroll
                for (int m = 0; m < MBE; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (trow0 + m * 16 + r >= r1) {
#pragma unroll
                            for (int n = 0; n < 4; ++n) acc[m][n][r] = -INFINITY;
                        }
            }
            // Two-stage test per query column block: the maximum over the lane's 32 scores decides (one ballot) whether
            // anything is scanned at all; then the eight row-block maxima decide block by block, so a trip costs eight
            // ballots plus four compares per block that really holds a survivor instead of 32 branchy compares (on the
            // sample's threshold, 86 % of the first level's tiles trip: the scan was 17 % of that level).
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                float bm[MBE];
                float mx = -INFINITY;
#pragma unroll
                for (int m = 0; m < MBE; ++m) {
                    bm[m] = fmaxf(fmaxf(acc[m][n][0], acc[m][n][1]), fmaxf(acc[m][n][2], acc[m][n][3]));
                    mx = fmaxf(mx, bm[m]);
                }
                if (__ballot(mx > thr[n]) != 0ull) {
#pragma unroll
                    for (int m = 0; m < MBE; ++m) {
                        if (__ballot(bm[m] > thr[n]) == 0ull) continue;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float s = acc[m][n][r];
                            if (s > thr[n]) {
                                const int pos = cnt[n]++;
                                if (pos < CAP) my_ent[n * 16 * CAP + pos] = Cand{s, (int32_t)(trow0 + m * 16 + r)};
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < MBE; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            ++tile;
        }
        }

        // retire this wave's DMA for step + 1; the barrier then publishes every wave's pieces and guarantees nobody
        // still reads the slots refilled next iteration
        if constexpr (MODE == 22) {
            const unsigned long long tb = stamp();
            retire_for(step + 1);
            const unsigned long long tc = stamp();
            __builtin_amdgcn_s_barrier();
            const unsigned long long td = stamp();
            st_compute += tb - st_prev;
            st_vmwait += tc - tb;
            st_barrier += td - tc;
            st_prev = td;
        } else {
        retire_for(step + 1);
        __builtin_amdgcn_s_barrier();
        }
    }
    if constexpr (MODE == 22) {
        if (lane == 0) {
            unsigned long long* dbg = reinterpret_cast<unsigned long long*>(seg_ent) +
                                      ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 4;
            dbg[0] = st_compute;
            dbg[1] = st_vmwait;
            dbg[2] = st_barrier;
            dbg[3] = (unsigned long long)total_steps;
        }
    }
    };
    if (TNQ == 256 && MODE != 11 && __builtin_amdgcn_readfirstlane(wm) == 1) main_loop(std::true_type{});  // MODE 12 relies on this split
    else main_loop(std::false_type{});

#ifdef ISC_ABLATION  // timing aids (wrong results): nslots bit 16 = stop after the main loop, bit 17 = stop before the tail
    const int abl = nslots >> 16;
    nslots &= 0xffff;
    if (abl & 1) return;
#endif
    if constexpr (SAMPLE) {
        // ---- level 0 epilogue (one tile per workgroup; every DMA has been retired and every wave is past the last
        // barrier, so the LDS is free).  The 256 scores of a query are cut into `nslots` slots of 256 / nslots scores, each
        // slot living in one lane's registers; the kp-th largest of the slot maxima, L, is a lower bound of the
        // workgroup's kp-th best score (at least kp scores are >= L), so only scores >= L can be among the best kp of
        // the whole bank.  With nslots ~ 2 kp about 1.35 kp scores per query pass.  nslots = 0 (kp > 64): every score
        // is kept.  Rows past the end of the bank count as -inf.
        const int64_t trow0 = r0 + (int64_t)tile_begin * TM + wm * (TM / WM) + fg * 4;
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (trow0 + m * 16 + r >= r1) {
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[m][n][r] = -INFINITY;
                }
        if (nslots > 0) {
            float* smax = reinterpret_cast<float*>(lds);                       // [TNQ][nslots + 4] (<= 132 KiB)
            const int sstride = nslots + 4;  // rows 16 bytes longer than a power of two: the 16 query rows a wave writes at
                                             // once would otherwise all start in the same LDS bank
            float* lbound = reinterpret_cast<float*>(lds + TAIL_LBOUND_OFF);  // [TNQ], past the tail's lists
            const int gsz = TM / nslots;                    // scores per slot: 8, 4 or 2
            const int per_lane = MB * 4 / gsz;              // slots per lane and query column
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                float* dst = smax + (size_t)(wn * 64 + n * 16 + frow) * sstride + (wm * 4 + fg) * per_lane;
                float cur = -INFINITY;
#pragma unroll
                for (int e = 0; e < MB * 4; ++e) {
                    cur = fmaxf(cur, acc[e >> 2][n][e & 3]);
                    if (((e + 1) & (gsz - 1)) == 0) {
                        dst[e / gsz] = cur;
                        cur = -INFINITY;
                    }
                }
            }
            __syncthreads();
            // rank of every slot maximum among its query's nslots (ties by slot index: ranks are a permutation).  The row
            // is read 16 bytes at a time, two reads in flight: a scalar loop pays one LDS latency per element.
            for (int id = tid; id < TNQ * nslots; id += NTHREADS) {
                const int qc = id / nslots, sl = id - qc * nslots;
                const float4* row4 = reinterpret_cast<const float4*>(smax + (size_t)qc * sstride);
                const float v = smax[(size_t)qc * sstride + sl];
                int rank = 0;
#pragma unroll 2
                for (int j4 = 0; j4 < nslots / 4; ++j4) {
                    const float4 x = row4[j4];
                    const int j = j4 * 4;
                    rank += (x.x > v || (x.x == v && j + 0 < sl)) ? 1 : 0;
                    rank += (x.y > v || (x.y == v && j + 1 < sl)) ? 1 : 0;
                    rank += (x.z > v || (x.z == v && j + 2 < sl)) ? 1 : 0;
                    rank += (x.w > v || (x.w == v && j + 3 < sl)) ? 1 : 0;
                }
                if (rank == kp - 1) lbound[qc] = v;
            }
            __syncthreads();
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                // the filter below is "s > thr": step L down by one float so that it reads "s >= L".  Padding queries
                // (tau = +inf, loaded into thr at kernel entry) keep +inf and emit nothing.
                const float lb = lbound[wn * 64 + n * 16 + frow];
                const unsigned u = __float_as_uint(lb);
                const float below = lb == 0.f ? __uint_as_float(0x80000001u)
                                              : __uint_as_float(lb > 0.f ? u - 1u : u + 1u);
                if (thr[n] != INFINITY) thr[n] = lb == -INFINITY ? -INFINITY : below;
            }
        }
    }

#ifdef ISC_ABLATION
    if (abl & 2) return;
#endif
    // ---- tail: this workgroup's survivors -> the per-query lists, aggregated through LDS so that the list counters see
    // ONE global atomic per (workgroup, query) (a lane-level atomic per survivor serialises on the counter's address:
    // 5 k of them on one query's counter took 60 us).  64 queries per pass (the waves of one query column), per query
    // an LDS list of TAIL_LCAP = 256 entries -- all a tile can produce -- filled with LDS atomics, then copied out by 8
    // threads per query.  In the filter levels the survivors come from the lane-private segments written in the hot
    // loop, at level 0 straight from the accumulators.
    {
        // level 0 of the 256-query shape with kp = 16 (k <= 10): all 256 queries in ONE pass with 64-entry lists (a
        // workgroup emits ~1.35 kp = 21 candidates per query there; more than 64 would flag the query for the exact
        // pass).  Four passes of 64 queries with only two of the eight waves active in each cost 69 of that launch's
        // 104 us.  Larger kp keep the four passes (256-entry lists).
        const int PASSES = (SAMPLE && TNQ == 256 && kp <= 16) ? 1 : WN;
        const int QPP = TNQ / PASSES;             // queries per pass
        const int LCAP = 64 * TAIL_LCAP / QPP;     // list entries per query: 256 or 64 (128 KiB of lists either way)
        const int TPQ = NTHREADS / QPP;           // threads per query in the copy-out: 8 or 2
        Cand* lists = reinterpret_cast<Cand*>(lds);                      // [QPP][LCAP]
        int* lcount = reinterpret_cast<int*>(lds + TAIL_COUNT_OFF);     // [QPP]
        const int64_t trow0 = r0 + (int64_t)tile_begin * TM + wm * (TM / WM) + fg * 4;  // level 0: the one tile
#pragma unroll 1
        for (int pass = 0; pass < PASSES; ++pass) {
            __syncthreads();  // the previous pass has been copied out (first pass: every lane has read its lbound)
            if (tid < QPP) lcount[tid] = 0;
            __syncthreads();
            if (PASSES == 1 || wn == pass) {
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int qc = (PASSES == 1 ? wn * 64 : 0) + n * 16 + frow;
                    if constexpr (SAMPLE) {
                        // count this lane's survivors of the query first, then ONE LDS atomic for all of them: one atomic
                        // per survivor was a dependent LDS round trip in nearly every one of the 128 element slots of a
                        // wave (some lane always holds a survivor), 30 us of the 73 us sample launch at Q = 1024
                        int mine = 0;
#pragma unroll
                        for (int m = 0; m < MB; ++m)
#pragma unroll
                            for (int r = 0; r < 4; ++r) mine += acc[m][n][r] > thr[n] ? 1 : 0;  // rows past the end are -inf
                        int pos = mine > 0 ? atomicAdd(&lcount[qc], mine) : 0;
#pragma unroll
                        for (int m = 0; m < MB; ++m)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float sc = acc[m][n][r];
                                if (sc > thr[n]) {
                                    if (pos < LCAP) lists[qc * LCAP + pos] = Cand{sc, (int32_t)(trow0 + m * 16 + r)};
                                    ++pos;
                                }
                            }
                    } else {
                        const int c = min(cnt[n], CAP);
                        if (cnt[n] > CAP) {
                            atomicAdd(&status[0], 1);
                            qflag[q0 + wn * 64 + qc] = 1;
                        }
                        const Cand* src = my_ent + (size_t)n * 16 * CAP;
                        for (int i = 0; i < c; i += 4) {  // four independent loads per trip
                            Cand e[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) e[j] = src[min(i + j, c - 1)];
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (i + j < c) {
                                    const int pos = atomicAdd(&lcount[qc], 1);
                                    if (pos < LCAP) lists[qc * LCAP + pos] = e[j];
                                }
                        }
                    }
                }
            }
            __syncthreads();
            {
                const int qc = tid / TPQ, sub = tid % TPQ;
                const int q = q0 + pass * 64 + qc;
                const int c_all = lcount[qc];
                const int c = min(c_all, LCAP);
                int off = 0;
                if (sub == 0 && c > 0) off = atomicAdd(&qcount[q], c);
                off = __shfl(off, lane & ~(TPQ - 1), 64);
                if (sub == 0 && (c_all > LCAP || off + c > QCAP)) {
                    atomicAdd(&status[0], 1);
                    qflag[q] = 1;
                }
                Cand* dst = qlist + (size_t)q * QCAP + off;
                for (int i = sub; i < c; i += TPQ)
                    if (off + i < QCAP) dst[i] = lists[qc * LCAP + i];
            }
        }
    }
}

// ---- selection ----------------------------------------------------------------------------------------------
// One WORKGROUP (512 threads) per query: the level's survivors plus the carried list -> the best kp by (score
// desc, packed row asc), written to `topk` (LDS, best first).  Returns how many there are (<= kp), or -1 when the
// ranking step's buffer would overflow (adversarial list order; the caller marks the query for the exact redo).
//   1. every thread takes the maximum key of its strided share of the candidates;
//   2. kp <= 32: per wave, L_w = the kp-th largest of the 64 lane maxima -- at least kp candidates of that wave's
//      share are >= L_w -- and lim = max over the waves; kp > 32: lim = the kp-th largest of the 512 thread maxima
//      (at kp = 64 the per-wave form is the MINIMUM lane maximum, which lets a fifth of a long list through).
//      Either way at least kp candidates are >= lim, so the best kp all are;
//   3. candidates >= lim (typically ~1.5 kp of them) are compacted into LDS;
//   4. they are ranked exactly by counting larger keys.
constexpr int SEL_PER = QCAP / SEL_THREADS;  // list entries per thread: the whole list sits in registers
constexpr int SEL_SPEC_MANY = 4;             // speculative entries per thread when a call has more than SMALL_Q queries
struct SelShared {
    unsigned long long surv[SURV_CAP];
    unsigned long long topk[ISC_TOPK_MAX_K + 8];
    unsigned long long wlim[SEL_THREADS];  // kp <= 32: [0..3] per-wave limits; else the thread maxima
    unsigned long long lim;
    int ns;
};

// steps 1 (maxima) - 4 on keys that already sit in registers (0 = empty slot); the caller has set sh.ns = 0 before a
// workgroup barrier or relies on the one inside step 2
__device__ int wg_select_keys(SelShared& sh, const unsigned long long (&key)[SEL_PER + 1], int kp) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    if (tid == 0) sh.ns = 0;
    unsigned long long lmax = 0ull;
#pragma unroll
    for (int j = 0; j <= SEL_PER; ++j) lmax = key[j] > lmax ? key[j] : lmax;

    // 2. threshold key
    if (kp <= 32) {
        int rank = 0;
        for (int j = 0; j < 64; ++j) {
            const unsigned long long o = isc_bcast_key(lmax, j);
            rank += (o > lmax || (o == lmax && j < lane)) ? 1 : 0;  // empty lanes tie at 0: broken by lane
        }
        const unsigned long long mine = (rank == kp - 1) ? lmax : 0ull;  // exactly one lane has this rank
        const unsigned long long lw = isc_wave_max_key(mine);
        if (lane == 0) sh.wlim[wave] = lw;
        __syncthreads();
        unsigned long long lim = sh.wlim[0];
#pragma unroll
        for (int w = 1; w < SEL_THREADS / 64; ++w) lim = sh.wlim[w] > lim ? sh.wlim[w] : lim;
        if (tid == 0) sh.lim = lim;
    } else {
        sh.wlim[tid] = lmax;
        __syncthreads();
        int rank = 0;
        for (int j = 0; j < SEL_THREADS; ++j) {
            const unsigned long long o = sh.wlim[j];
            rank += (o > lmax || (o == lmax && j < tid)) ? 1 : 0;
        }
        if (rank == kp - 1) sh.lim = lmax;
    }
    __syncthreads();
    const unsigned long long lim = sh.lim;

    // 3. compact the candidates >= lim: one LDS atomic per thread that holds any (typically ~1.5 kp in all)
    int mine_n = 0;
#pragma unroll
    for (int j = 0; j <= SEL_PER; ++j) mine_n += (key[j] != 0ull && key[j] >= lim) ? 1 : 0;
    if (mine_n > 0) {
        int pos = atomicAdd(&sh.ns, mine_n);
#pragma unroll
        for (int j = 0; j <= SEL_PER; ++j)
            if (key[j] != 0ull && key[j] >= lim) {
                if (pos < SURV_CAP) sh.surv[pos] = key[j];
                ++pos;
            }
    }
    __syncthreads();
    const int ns = sh.ns;
    if (ns > SURV_CAP) return -1;

    // 4. exact ranks
    for (int e = tid; e < ns; e += SEL_THREADS) {
        const unsigned long long mine = sh.surv[e];
        int rank = 0;
#pragma unroll 8
        for (int j = 0; j < ns; ++j) rank += sh.surv[j] > mine ? 1 : 0;
        if (rank < kp) sh.topk[rank] = mine;
    }
    __syncthreads();
    return min(kp, ns);
}

// SPEC = list entries per thread that are loaded BEFORE the list length is known (the rest only if the list is that long).
// With few queries (SPEC = SEL_PER) the whole 64 KiB buffer of a query is fetched blindly and counter and list travel in
// one memory round trip; with a thousand queries that is 64 MB of list traffic per selection for lists that are
// typically a quarter full, so only the first 2048 slots are speculative there and a longer list pays a second trip.
template <int SPEC>
__device__ int wg_select(SelShared& sh, int q, int kp, const int32_t* __restrict__ qcount,
                         const Cand* __restrict__ qlist, const float* __restrict__ carry_s,
                         const int32_t* __restrict__ carry_r, const int32_t* __restrict__ carry_n) {
    const int tid = threadIdx.x;
    const Cand* src = qlist + (size_t)q * QCAP;
    const int from_list = min(qcount[q], QCAP);
    const int carried = min(carry_n[q], kp);

    // 1. every candidate into registers: SEL_PER independent coalesced 8-byte loads per thread, issued BEFORE the
    // list length is known (the list buffer always holds QCAP slots; what lies past the length is ignored), so the
    // counter and the list travel in one memory round trip; plus one carried entry (kp <= 128 < 512 threads)
    unsigned long long key[SEL_PER + 1];
    {
        const unsigned long long* raw = reinterpret_cast<const unsigned long long*>(src);
#pragma unroll
        for (int j = 0; j < SPEC; ++j) key[j] = raw[tid + SEL_THREADS * j];
        if constexpr (SPEC < SEL_PER) {
            if (from_list > SPEC * SEL_THREADS) {  // workgroup-uniform
#pragma unroll
                for (int j = SPEC; j < SEL_PER; ++j) key[j] = raw[tid + SEL_THREADS * j];
            } else {
#pragma unroll
                for (int j = SPEC; j < SEL_PER; ++j) key[j] = 0ull;
            }
        }
        const int ci = min(tid, kp - 1);
        const float cs = carry_s[(size_t)q * kp + ci];
        const int cr = carry_r[(size_t)q * kp + ci];
#pragma unroll
        for (int j = 0; j < SEL_PER; ++j) {
            const Cand e{__uint_as_float((unsigned)key[j]), (int)(unsigned)(key[j] >> 32)};
            key[j] = tid + SEL_THREADS * j < from_list ? isc_make_key(e.s, e.row) : 0ull;
        }
        key[SEL_PER] = tid < carried ? isc_make_key(cs, cr) : 0ull;
    }
    return wg_select_keys(sh, key, kp);
}

// between two levels: carried list and tau of every query
template <int SPEC>
__global__ __launch_bounds__(SEL_THREADS, 4) void k_select(int32_t* __restrict__ qcount, const Cand* __restrict__ qlist,
                                                        int kp, float* __restrict__ tau, float* __restrict__ carry_s,
                                                        int32_t* __restrict__ carry_r, int32_t* __restrict__ carry_n,
                                                        int32_t* __restrict__ qflag) {
    __shared__ SelShared sh;
    const int q = blockIdx.x;
    const int n = wg_select<SPEC>(sh, q, kp, qcount, qlist, carry_s, carry_r, carry_n);
    const int tid = threadIdx.x;
    if (n < 0) {  // give up on the fast path for this query: nothing more survives, k_final lists it for k_exact
        if (tid == 0) {
            qflag[q] = 1;
            tau[q] = INFINITY;
            carry_n[q] = 0;
            qcount[q] = 0;
        }
        return;
    }
    if (tid < n) {
        const unsigned long long key = sh.topk[tid];
        const float sc = isc_key_score(key);
        carry_s[(size_t)q * kp + tid] = sc;
        carry_r[(size_t)q * kp + tid] = isc_key_row(key);
        if (tid == kp - 1) tau[q] = sc;
    }
    if (tid == 0) {
        carry_n[q] = n;
        qcount[q] = 0;  // ready for the next level
    }
}

// 16 bytes of a packed row as float64 values
template <typename T>
struct Chunk16;
template <>
struct Chunk16<_Float16> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const unsigned char* p, double (&v)[8]) {
        const uint4 raw = *reinterpret_cast<const uint4*>(p);
        const _Float16* h = reinterpret_cast<const _Float16*>(&raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (double)(float)h[j];
    }
};
template <>
struct Chunk16<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const unsigned char* p, double (&v)[8]) {
        const float4 raw = *reinterpret_cast<const float4*>(p);
        v[0] = (double)raw.x;
        v[1] = (double)raw.y;
        v[2] = (double)raw.z;
        v[3] = (double)raw.w;
    }
};

// float64 norm of the packed query row at `qrow_base` (K step s at + s * tnq * 128 B); called by ONE wave, result in
// every lane
template <typename T>
__device__ double wave_query_norm(const unsigned char* qrow_base, int ks, int tnq) {
    const int lane = threadIdx.x & 63;
    const int sub = lane >> 3, ch = lane & 7;
    double acc = 0.0;
    for (int s0 = 0; s0 < ks; s0 += 8) {
        const int s = s0 + sub;
        if (s < ks) {
            double v[8];
            Chunk16<T>::load(qrow_base + (size_t)s * tnq * ISC_KSTEP_BYTES + ch * 16, v);
#pragma unroll
            for (int j = 0; j < Chunk16<T>::N; ++j) acc = fma(v[j], v[j], acc);
        }
    }
    return isc_wave_sum(acc);
}

// Exact float64 dots of `nc` candidates with one query: one wave per candidate, a lane covers the 16-byte chunk `ch` of
// K steps sub, sub + 8, ...; four candidates of a wave are in flight together (their loads are independent).
// row_of(c) -> packed row or -1; store(c, dot) is called by lane 0 of the wave that owns candidate c (dot = 0 for row -1).
template <typename T, typename RowOf, typename Store>
__device__ void exact_dots(const unsigned char* __restrict__ bank, int ks, const unsigned char* qrow_base, int tnq,
                           int nc, int64_t nrows, RowOf row_of, Store store) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 3, ch = lane & 7;
    constexpr int NW = SEL_THREADS / 64;
    for (int c0 = wave; c0 < nc; c0 += 4 * NW) {
        int row[4];
        double acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * NW;
            row[u] = c < nc ? row_of(c) : -1;
            if ((unsigned)row[u] >= (unsigned)nrows) row[u] = -1;
            acc[u] = 0.0;
        }
        for (int s0 = 0; s0 < ks; s0 += 8) {
            const int s = s0 + sub;
            if (s < ks) {
                double a[4][8], b[8];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    Chunk16<T>::load(bank + isc_packed_offset(row[u] < 0 ? 0 : row[u], s, ks) + ch * 16, a[u]);
                Chunk16<T>::load(qrow_base + (size_t)s * tnq * ISC_KSTEP_BYTES + ch * 16, b);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < Chunk16<T>::N; ++j) acc[u] = fma(a[u][j], b[j], acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double tot = isc_wave_sum(acc[u]);
            if (lane == 0 && c0 + u * NW < nc) store(c0 + u * NW, row[u] < 0 ? 0.0 : tot);
        }
    }
}

// What k_final does with a query it cannot prove.
//   -> slot in the REDO list: the matrix-core filter runs over the whole bank once more for the listed queries with a
//      FIXED threshold just below the k-th exact score found so far, every survivor is re-scored in float64 (k_final2);
//   -> the EXACT list (k_exact, an exhaustive float64 sweep) when there is no usable k-th score.
struct RedoLists {
    int32_t* r_count;
    int32_t* r_list;
    float* tau2;
    unsigned char* qpacked2;
    int32_t* x_count;  // k_exact's list
    int32_t* x_list;
};

// One workgroup per query: last selection, exact float64 re-score of the carried candidates, final order, output,
// and the guard that proves the float32 filter lost nothing (see the file header).
template <typename T, int SPEC>
__global__ __launch_bounds__(SEL_THREADS, 4) void k_final(
    const unsigned char* __restrict__ bank, int ks, const unsigned char* __restrict__ qpacked, int tnq, int kp, int k,
    IscPerm pm, int64_t index_base, const float* __restrict__ norm_bound, const int32_t* __restrict__ qcount,
    const Cand* __restrict__ qlist, const float* __restrict__ carry_s, const int32_t* __restrict__ carry_r,
    const int32_t* __restrict__ carry_n, const int32_t* __restrict__ qflag, float* __restrict__ out_s,
    int64_t* __restrict__ out_i, RedoLists rl, int32_t* __restrict__ status) {
    __shared__ SelShared sh;
    __shared__ double exact_dot[ISC_TOPK_MAX_K + 8];
    __shared__ float fsc[ISC_TOPK_MAX_K + 8];
    __shared__ int orig[ISC_TOPK_MAX_K + 8];
    __shared__ double qnorm_sh;
    __shared__ float kth_sh;
    __shared__ int slot_sh;
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = wg_select<SPEC>(sh, q, kp, qcount, qlist, carry_s, carry_r, carry_n);
    bool redo = n < 0 || qflag[q] != 0;
    const int nc = n < 0 ? 0 : n;

    // this query's packed row: K step s at qrow_base + s * tnq * 128
    const unsigned char* qrow_base = qpacked + ((size_t)(q / tnq) * ks * tnq + (q % tnq)) * ISC_KSTEP_BYTES;
    if (tid < 64) {
        const double nn = wave_query_norm<T>(qrow_base, ks, tnq);
        if (tid == 0) qnorm_sh = sqrt(nn);
    }
    exact_dots<T>(bank, ks, qrow_base, tnq, nc, pm.n, [&](int c) { return isc_key_row(sh.topk[c]); },
                  [&](int c, double dot) { exact_dot[c] = dot; });
    __syncthreads();
    const double qnorm = qnorm_sh;
    const double denom = fmax(qnorm, 1e-12);
    const double bmax = norm_bound ? (double)*norm_bound : 1.001;

    // ---- queries whose answer needs no search (and would flood every candidate list: ALL rows tie).  A zero query scores
    // 0 against every row, a query with a non-finite norm NaN (any dot is +-inf or NaN, divided by inf or NaN): the order is
    // by row index alone, i.e. the first k rows.  Only when the bank itself is finite (its norm bound is).
    if ((qnorm == 0.0 || !(qnorm <= 1.7e308)) && bmax <= 1.7e308) {
        if (tid < k) {
            out_s[(size_t)q * k + tid] = qnorm == 0.0 ? 0.f : __uint_as_float(0x7fc00000u);
            out_i[(size_t)q * k + tid] = (int64_t)tid + index_base;
        }
        return;
    }

    if (tid < nc) {
        const int row = isc_key_row(sh.topk[tid]);
        fsc[tid] = (float)(exact_dot[tid] / denom);
        orig[tid] = (unsigned)row < (unsigned)pm.n ? (int)isc_perm_orig(pm, row) : 0x7ffffffe;
        if ((unsigned)row >= (unsigned)pm.n) redo = true;  // cannot happen; never trust such an entry
    }
    if (tid == 0) kth_sh = __uint_as_float(0x7fc00000u);  // NaN until a k-th score exists
    __syncthreads();
    // rounding-error bound of a filter score: Dpad float32 accumulation steps, each off by at most 2^-23 of the running
    // magnitude (<= sum |q_i b_i| <= ||q|| * ||b||): one step of margin over round-to-nearest, whatever the order in
    // which the matrix core adds its 32 products
    const double eps = (double)(ks * (ISC_KSTEP_BYTES / (int)sizeof(T))) * (1.0 / 8388608.0) * qnorm * bmax;
    // The bound is RELATIVE: it assumes every product and partial sum stays in float32's normal range (round to nearest at
    // every accumulation step of the matrix core is assumed too; status[2] < 1 is what the tests assert about it).  A query
    // of denormal scale loses product bits to underflow, one of huge scale can overflow a partial sum to inf / NaN while
    // the float64 score is finite -- and such rows are dropped silently by the filter.  |partial sum| <= ||q|| max||b||, so
    // outside [1e-30, 1e37] (and for a NaN / inf bound) no filter result is trusted: the exhaustive pass answers.
    const bool filter_trusted = qnorm >= 1e-30 && qnorm * bmax <= 1e37;
    if (tid < nc) {
        const unsigned long long mykey = isc_make_key(fsc[tid], orig[tid]);
        int rank = 0;
        for (int j = 0; j < nc; ++j) rank += isc_make_key(fsc[j], orig[j]) > mykey ? 1 : 0;
        if (rank < k) {
            out_s[(size_t)q * k + rank] = fsc[tid];
            out_i[(size_t)q * k + rank] = (int64_t)orig[tid] + index_base;
        }
        if (rank == k - 1) kth_sh = fsc[tid];
        // how far the filter scores really are from the exact ones, in units of the bound (status[2], diagnostics)
        if (eps > 0.0) {
            const float ratio = (float)(fabs((double)isc_key_score(sh.topk[tid]) - exact_dot[tid]) / eps);
            if (ratio == ratio) atomicMax(reinterpret_cast<unsigned*>(&status[2]), __float_as_uint(ratio));
        }
    }
    redo = __syncthreads_or(redo ? 1 : 0) != 0;
    if (tid == 0) {
        slot_sh = -1;
        if (!filter_trusted) redo = true;
        if (!redo && (int64_t)nc < pm.n) {  // with every row of the bank carried the answer is exact as it stands
            if (nc < kp) {
                redo = true;  // fewer candidates than asked for although the bank has more rows (NaN scores)
            } else {
                const float t = isc_key_score(sh.topk[kp - 1]);  // every dropped row's filter score is <= t
                const float bound = (float)(((double)t + eps) / denom);
                redo = !(bound < kth_sh);
            }
        }
        if (redo) {
            atomicAdd(&status[1], 1);
            // Threshold of the redo filter, in the filter's units (raw dot products): a row that belongs to the answer has
            // score >= kth, i.e. E / denom >= the float below kth (float32 rounding is monotone), and its filter score is
            // A >= E - eps.  tau2 is rounded DOWN so that "A > tau2" keeps every such row, ties with kth included.
            const float kth = kth_sh;
            double t2 = (double)__uint_as_float(0x7fc00000u);
            if (filter_trusted && kth == kth && fabsf(kth) <= 3.0e38f) {
                const float kth_dn = nextafterf(kth, -INFINITY);
                const double ed = (double)kth_dn * denom;
                t2 = ed - eps - fabs(ed) * 1e-12;
            }
            if (t2 == t2 && fabs(t2) <= 3.0e38) {
                float tf = (float)t2;
                if ((double)tf >= t2) tf = nextafterf(tf, -INFINITY);
                const int slot = atomicAdd(rl.r_count, 1);
                rl.r_list[slot] = q;
                rl.tau2[slot] = tf;
                slot_sh = slot;
            } else {  // no usable k-th score (fewer than k candidates, NaN / inf scores, untrusted filter)
                const int slot = atomicAdd(rl.x_count, 1);
                rl.x_list[slot] = q;
                atomicAdd(&status[3], 1);
            }
        }
    }
    __syncthreads();
    const int slot = slot_sh;
    if (slot >= 0) {  // this query's packed row -> row `slot` of the redo query tiles
        unsigned char* dst = rl.qpacked2 + ((size_t)(slot / tnq) * ks * tnq + (slot % tnq)) * ISC_KSTEP_BYTES;
        for (int i = tid; i < ks * 8; i += SEL_THREADS) {
            const size_t off = (size_t)(i >> 3) * tnq * ISC_KSTEP_BYTES + (i & 7) * 16;
            *reinterpret_cast<uint4*>(dst + off) = *reinterpret_cast<const uint4*>(qrow_base + off);
        }
    }
}

// One workgroup per redo SLOT (normally none: the launch exits at once).  The redo filter has collected every row whose
// filter score exceeds tau2[slot] -- a superset of the rows that can belong to the answer (k_final) -- into the slot's list.
// ALL of them are re-scored in float64 here, in place (the 8-byte list entry becomes the exact key), the best k by
// (exact score desc, ORIGINAL row asc) are the answer: nothing is left to prove.  A list or candidate buffer that
// overflowed (thousands of rows within rounding noise of the k-th score: a heavily duplicated row) hands the query to
// k_exact.
template <typename T>
__global__ __launch_bounds__(SEL_THREADS, 4) void k_final2(
    const unsigned char* __restrict__ bank, int ks, const unsigned char* __restrict__ qpacked2, int tnq, int k, IscPerm pm,
    int64_t index_base, const int32_t* __restrict__ r_count, const int32_t* __restrict__ r_list,
    const int32_t* __restrict__ qcount2, const int32_t* __restrict__ qflag2, Cand* __restrict__ qlist,
    float* __restrict__ out_s, int64_t* __restrict__ out_i, int32_t* __restrict__ x_count, int32_t* __restrict__ x_list,
    int32_t* __restrict__ status) {
    const int slot = blockIdx.x;
    if (slot >= *r_count) return;
    __shared__ SelShared sh;
    __shared__ double qnorm_sh;
    const int tid = threadIdx.x;
    const int q = r_list[slot];
    const int cnt_all = qcount2[slot];
    const int cnt = min(cnt_all, QCAP);
    bool lost = qflag2[slot] != 0 || cnt_all > QCAP || cnt < k;
    const unsigned char* qrow_base = qpacked2 + ((size_t)(slot / tnq) * ks * tnq + (slot % tnq)) * ISC_KSTEP_BYTES;
    if (tid < 64) {
        const double nn = wave_query_norm<T>(qrow_base, ks, tnq);
        if (tid == 0) qnorm_sh = sqrt(nn);
    }
    __syncthreads();
    const double denom = fmax(qnorm_sh, 1e-12);
    Cand* list = qlist + (size_t)slot * QCAP;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(list);
    int n = -1;
    if (!lost) {
        exact_dots<T>(bank, ks, qrow_base, tnq, cnt, pm.n, [&](int c) { return list[c].row; },
                      [&](int c, double dot) {
                          const int row = list[c].row;
                          keys[c] = (unsigned)row < (unsigned)pm.n
                                        ? isc_make_key((float)(dot / denom), (int)isc_perm_orig(pm, row))
                                        : 0ull;
                      });
        // The keys were written by other waves of THIS workgroup: one CU, one vector L1, through which both the stores and
        // the loads below go -- workgroup scope, which is what __syncthreads() orders (an L1-bypassing load could overtake
        // a store still on its way to the L2).  The volatile access keeps the compiler from reusing the Cand it read before.
        __syncthreads();
        unsigned long long key[SEL_PER + 1];
        const volatile unsigned long long* vkeys = keys;
#pragma unroll
        for (int j = 0; j < SEL_PER; ++j) key[j] = tid + SEL_THREADS * j < cnt ? vkeys[tid + SEL_THREADS * j] : 0ull;
        key[SEL_PER] = 0ull;
        n = wg_select_keys(sh, key, k);
    }
    if (n < k) {  // overflow, or more than SURV_CAP rows tie around the k-th: the exhaustive pass answers
        if (tid == 0) {
            const int xs = atomicAdd(x_count, 1);
            x_list[xs] = q;
            atomicAdd(&status[3], 1);
        }
        return;
    }
    if (tid < k) {
        const unsigned long long key = sh.topk[tid];
        out_s[(size_t)q * k + tid] = isc_key_score(key);
        out_i[(size_t)q * k + tid] = (int64_t)isc_key_row(key) + index_base;
    }
}

#ifdef ISC_ABLATION
int debug_mode() {
    static const int mode = [] {
        const char* e = getenv("ISC_DEBUG_MODE");
        return e ? atoi(e) : 0;
    }();
    return mode;
}
#else
constexpr int debug_mode() { return 0; }
#endif

#ifdef ISC_ABLATION
bool no_nt() {
    static const bool v = getenv("ISC_NO_NT") != nullptr;  // A/B aid: default cache policy on every bank stream
    return v;
}
#else
constexpr bool no_nt() { return false; }
#endif

// tiles per chunk and launch above which a level with several query tiles is cut into more launches
#ifdef ISC_ABLATION
int seg_tiles() {
    static const int v = [] {
        const char* e = getenv("ISC_SEG_TILES");
        return e ? atoi(e) : 128;
    }();
    return v;
}
#else
constexpr int seg_tiles() { return 128; }
#endif

// what a filter launch reads its queries / thresholds from and counts its survivors into: the search proper, or the redo
struct FilterIO {
    const unsigned char* qpacked;
    const float* tau;
    int32_t* qcount;
    int32_t* qflag;
    const int32_t* active;  // nullptr, or the device-side number of listed slots (redo): tiles past it exit at once
};

template <typename T, int TNQ>
void launch_filter(const Level& l, const Plan& p, const Workspace& w, const FilterIO& io, const unsigned char* bank,
                   int ksteps, int32_t* status, hipStream_t stream) {
#define ISC_LAUNCH_FILTER(DBG_, SAMPLE_)                                                                             \
    hipLaunchKernelGGL((k_dots_filter<T, TNQ, DBG_, SAMPLE_>), dim3(l.nchunks, p.qtiles), dim3(NTHREADS), 0, stream, \
                       bank, l.r0, l.r1, l.tiles_per_chunk, l.ntiles, io.qpacked, ksteps, io.tau, p.qpad, w.seg_ent,  \
                       io.qcount, w.qlist, p.kp, nslots_arg, io.qflag, status, io.active)
    int nslots_arg = p.nslots;
    if constexpr (TNQ == 128) {
        // 64 < Q <= 128: always ONE query tile, so the bank stream is non-temporal (13) in the sample level, the filter
        // levels and the redo (33) alike -- the only instantiations of this shape
        if (l.sample) ISC_LAUNCH_FILTER(13, true);
        else if (io.active) ISC_LAUNCH_FILTER(33, false);
        else ISC_LAUNCH_FILTER(13, false);
    } else {
    if (l.sample) {  // one tile per workgroup: the staging variant does not matter
#ifdef ISC_ABLATION
        static const int abl = [] {
            const char* e = getenv("ISC_SAMPLE_ABL");
            return e ? atoi(e) : 0;
        }();
        Plan pa = p;
        pa.nslots |= abl << 16;
        hipLaunchKernelGGL((k_dots_filter<T, TNQ, 12, true>), dim3(l.nchunks, p.qtiles), dim3(NTHREADS), 0, stream, bank,
                           l.r0, l.r1, l.tiles_per_chunk, l.ntiles, io.qpacked, ksteps, io.tau, p.qpad, w.seg_ent, io.qcount,
                           w.qlist, p.kp, pa.nslots, io.qflag, status, io.active);
#else
        if constexpr (TNQ == 64) {
            if (p.qtiles == 1) ISC_LAUNCH_FILTER(13, true);  // one query tile: non-temporal bank stream
            else ISC_LAUNCH_FILTER(12, true);
        } else {
            ISC_LAUNCH_FILTER(12, true);
        }
#endif
        return;
    }
    if (io.active) {  // the redo of unproven queries: its own instantiations (never an ablation variant), see the kernel
        if (TNQ == 256 && p.qtiles > 1) ISC_LAUNCH_FILTER(32, false);
        else if constexpr (TNQ == 256) ISC_LAUNCH_FILTER(20, false);
        else if (p.qtiles == 1) ISC_LAUNCH_FILTER(33, false);
        else ISC_LAUNCH_FILTER(32, false);
        return;
    }
    // SPLIT (DBG 0) pays where a chunk has ONE query tile (Q <= 256).  With several query-tile workgroups streaming the
    // same bank rows it makes them drift further apart, and their sharing of those rows through the XCD's L2 drops
    // (measured L2 -> fabric reads per search at Q = 1024: 3.0 x the algorithmic bytes with SPLIT, 1.6 - 2.0 x without,
    // for +1 % speed), so those launches use the every-wave-issues form (DBG 12).
    int mode = debug_mode();
#ifdef ISC_ABLATION
    static const bool force_split = getenv("ISC_FORCE_SPLIT") != nullptr;
    static const int prio = getenv("ISC_NO_STATIC_PRIO") ? 1 : 0;  // A/B aid: bit 18 of nslots switches the priority off
    nslots_arg |= prio << 18;
    static const int fabl = getenv("ISC_FILTER_ABL") ? atoi(getenv("ISC_FILTER_ABL")) : 0;  // 2: filter launches skip the tail
    nslots_arg |= (fabl & 3) << 16;
    if (mode == 0 && TNQ == 256 && p.qtiles > 1 && !force_split) mode = 12;
#else
    if (mode == 0 && TNQ == 256 && p.qtiles > 1) mode = 12;
#endif
    switch (mode) {
#ifdef ISC_ABLATION
        case 2: ISC_LAUNCH_FILTER(2, false); break;
        case 3: ISC_LAUNCH_FILTER(3, false); break;
        case 7: ISC_LAUNCH_FILTER(7, false); break;
        case 11: ISC_LAUNCH_FILTER(11, false); break;
        case 15: ISC_LAUNCH_FILTER(15, false); break;
        case 17: ISC_LAUNCH_FILTER(17, false); break;
        case 22: ISC_LAUNCH_FILTER(22, false); break;
        case 41: ISC_LAUNCH_FILTER(41, false); break;
        case 43: ISC_LAUNCH_FILTER(43, false); break;
        case 48: ISC_LAUNCH_FILTER(48, false); break;
#endif
        case 12: ISC_LAUNCH_FILTER(12, false); break;
        default:
            if constexpr (TNQ == 256) {
                ISC_LAUNCH_FILTER(0, false);
            } else {  // the 64-query shape has no SPLIT; one query tile -> non-temporal bank stream (13)
                if (p.qtiles == 1 && !no_nt()) ISC_LAUNCH_FILTER(13, false);
                else ISC_LAUNCH_FILTER(12, false);
            }
            break;
    }
    }
#undef ISC_LAUNCH_FILTER
}

// With several query-tile workgroups per chunk, a long level runs as several launches over consecutive row ranges (same
// thresholds, no selection in between): the partner workgroups that share a chunk's bank rows through their XCD's L2
// drift apart as a launch goes on, and a kernel boundary realigns them for free (measured L2 -> fabric reads per search:
// 1.9 x the algorithmic bytes with 594 tiles per chunk and launch, 1.3 x with 127 / 483).
template <typename F>
void for_each_segment(const Level& l, const Plan& p, F&& launch) {
    int nseg = 1;
    if (!l.sample && p.qtiles > 1) nseg = isc_ceil_div(l.tiles_per_chunk, seg_tiles());
    const int64_t seg_rows = isc_ceil_div<int64_t>(isc_ceil_div<int64_t>(l.r1 - l.r0, nseg), TM) * TM;
    for (int sg = 0; sg < nseg; ++sg) {
        Level ls = l;
        ls.r0 = l.r0 + sg * seg_rows;
        ls.r1 = ls.r0 + seg_rows < l.r1 ? ls.r0 + seg_rows : l.r1;
        if (ls.r0 >= ls.r1) break;
        ls.ntiles = (int)isc_ceil_div<int64_t>(ls.r1 - ls.r0, TM);
        const int want = l.nchunks < ls.ntiles ? l.nchunks : ls.ntiles;
        ls.tiles_per_chunk = isc_ceil_div(ls.ntiles, want);
        ls.nchunks = isc_ceil_div(ls.ntiles, ls.tiles_per_chunk);
        launch(ls);
    }
}

template <typename T, typename TQ>
int run(const void* bank, int64_t n, int d, const void* queries, int q_total, int64_t ldq, int k, int64_t index_base,
        const float* norm_bound, float* out_s, int64_t* out_i, int32_t* status, void* ws_base, hipStream_t stream) {
    const Plan p = make_plan(n, q_total, k);
#ifdef ISC_ABLATION
    static const bool thr_inf_set = [] {
        const int v = getenv("ISC_THR_INF") != nullptr;
        return hipMemcpyToSymbol(HIP_SYMBOL(g_abl_thr_inf), &v, sizeof(int)) == hipSuccess;
    }();
    (void)thr_inf_set;
#endif
    const int ksteps = isc_ksteps(d, (int)sizeof(T));
    const Workspace w = carve(p, ksteps, n, k, ws_base);
    const unsigned char* bank_bytes = static_cast<const unsigned char*>(bank);
    const IscPerm pm = isc_make_perm(n);
    for (int q0 = 0; q0 < q_total; q0 += p.qb) {
        const int q = q_total - q0 < p.qb ? q_total - q0 : p.qb;
        const TQ* qptr = static_cast<const TQ*>(queries) + (int64_t)q0 * ldq;
        float* os = out_s + (size_t)q0 * k;
        int64_t* oi = out_i + (size_t)q0 * k;
        const bool spec_all = q <= SMALL_Q;  // few queries: fetch every list blindly (one round trip); see wg_select
        hipLaunchKernelGGL((k_prep<T, TQ>), dim3(isc_ceil_div(p.qpad * ksteps * 8, 256)), dim3(256), 0, stream, qptr, ldq, q,
                           d, ksteps, p.qpad, p.tnq, w.qpacked, w.tau, w.carry_n, w.qcount, w.qflag,
                           w.exact.redo_count, w.exact.done, w.r_count, w.tau2, w.qcount2, w.qflag2, status,
                           q0 == 0 ? 1 : 0);
        const FilterIO io{w.qpacked, w.tau, w.qcount, w.qflag, nullptr};
        for (int li = 0; li < p.nlevels; ++li) {
            const Level& l = p.levels[li];
            // With several query-tile workgroups per chunk, a long level runs as several launches over consecutive row
            // ranges (same thresholds, no selection in between): the partner workgroups that share a chunk's bank rows
            // through their XCD's L2 drift apart as a launch goes on, and a kernel boundary realigns them for free
            // (measured L2 -> fabric reads per search: 1.9 x the algorithmic bytes with 594 tiles per chunk and launch,
            // 1.3 x with 127 / 483).
            for_each_segment(l, p, [&](const Level& ls) {
                isc_timing_begin(ISC_KERNEL_DOTS_FILTER, stream);
                if (p.tnq == 256) launch_filter<T, 256>(ls, p, w, io, bank_bytes, ksteps, status, stream);
                else if (p.tnq == 128) launch_filter<T, 128>(ls, p, w, io, bank_bytes, ksteps, status, stream);
                else launch_filter<T, 64>(ls, p, w, io, bank_bytes, ksteps, status, stream);
                isc_timing_end(ISC_KERNEL_DOTS_FILTER, stream);
            });
            if (li + 1 < p.nlevels) {
                if (spec_all)
                    hipLaunchKernelGGL(k_select<SEL_PER>, dim3(q), dim3(SEL_THREADS), 0, stream, w.qcount, w.qlist, p.kp,
                                       w.tau, w.carry_s, w.carry_r, w.carry_n, w.qflag);
                else
                    hipLaunchKernelGGL(k_select<SEL_SPEC_MANY>, dim3(q), dim3(SEL_THREADS), 0, stream, w.qcount, w.qlist,
                                       p.kp, w.tau, w.carry_s, w.carry_r, w.carry_n, w.qflag);
            }
        }
        const RedoLists rl{w.r_count, w.r_list, w.tau2, w.qpacked2, w.exact.redo_count, w.exact.redo_list};
        if (spec_all)
            hipLaunchKernelGGL((k_final<T, SEL_PER>), dim3(q), dim3(SEL_THREADS), 0, stream, bank_bytes, ksteps, w.qpacked,
                               p.tnq, p.kp, k, pm, index_base, norm_bound, w.qcount, w.qlist, w.carry_s, w.carry_r,
                               w.carry_n, w.qflag, os, oi, rl, status);
        else
            hipLaunchKernelGGL((k_final<T, SEL_SPEC_MANY>), dim3(q), dim3(SEL_THREADS), 0, stream, bank_bytes, ksteps,
                               w.qpacked, p.tnq, p.kp, k, pm, index_base, norm_bound, w.qcount, w.qlist, w.carry_s,
                               w.carry_r, w.carry_n, w.qflag, os, oi, rl, status);
        // ---- matrix-core redo of the listed queries (normally none: every launch below exits at once, ~1.5 us each):
        // the whole bank as ONE level against the fixed thresholds tau2, survivors into the slots' lists, k_final2
        {
            const FilterIO rio{w.qpacked2, w.tau2, w.qcount2, w.qflag2, w.r_count};
            Level all;
            all.r0 = 0;
            all.r1 = n;
            all.sample = 0;
            all.ntiles = (int)isc_ceil_div<int64_t>(n, TM);
            int wgs = TARGET_WGS / p.qtiles;
            if (wgs < 1) wgs = 1;
            const int want = wgs < all.ntiles ? wgs : all.ntiles;
            all.tiles_per_chunk = isc_ceil_div(all.ntiles, want);
            all.nchunks = isc_ceil_div(all.ntiles, all.tiles_per_chunk);
            for_each_segment(all, p, [&](const Level& ls) {
                if (p.tnq == 256) launch_filter<T, 256>(ls, p, w, rio, bank_bytes, ksteps, status, stream);
                else if (p.tnq == 128) launch_filter<T, 128>(ls, p, w, rio, bank_bytes, ksteps, status, stream);
                else launch_filter<T, 64>(ls, p, w, rio, bank_bytes, ksteps, status, stream);
            });
            hipLaunchKernelGGL(k_final2<T>, dim3(q), dim3(SEL_THREADS), 0, stream, bank_bytes, ksteps, w.qpacked2, p.tnq, k,
                               pm, index_base, w.r_count, w.r_list, w.qcount2, w.qflag2, w.qlist, os, oi,
                               w.exact.redo_count, w.exact.redo_list, status);
        }
        const int st = isc_exact_launch(sizeof(T) == 2 ? ISC_F16 : ISC_F32, bank, n, d, qptr,
                                        sizeof(TQ) == 2 ? ISC_F16 : ISC_F32, ldq, k, index_base, w.exact, os, oi, status,
                                        stream);
        if (st != ISC_OK) return st;
    }
    return isc_launch_status();
}

int check_args(int dtype, int64_t n, int d, int q, int k) {
    if (dtype != ISC_F16 && dtype != ISC_F32) return ISC_ERR_INVALID_ARG;
    if (n <= 0 || d <= 0 || q <= 0 || k <= 0 || k > n) return ISC_ERR_INVALID_ARG;
    if (k > ISC_TOPK_MAX_K) return ISC_ERR_UNSUPPORTED;
    if (n > 0x7ffffffe) return ISC_ERR_UNSUPPORTED;  // row ids are int32 inside a shard
    if (d > ISC_SEARCH_MAX_D) return ISC_ERR_UNSUPPORTED;
    if (q > ISC_SEARCH_MAX_Q) return ISC_ERR_UNSUPPORTED;
    return ISC_OK;
}

}  // namespace

extern "C" int isc_cosine_topk_workspace_bytes(int dtype, int64_t N, int D, int Q, int k, size_t* bytes) {
    ISC_REQUIRE(bytes);
    const int st = check_args(dtype, N, D, Q, k);
    if (st != ISC_OK) return st;
    *bytes = carve(make_plan(N, Q, k), isc_ksteps(D, dtype == ISC_F16 ? 2 : 4), N, k, nullptr).bytes;
    return ISC_OK;
}

extern "C" int isc_cosine_topk(const void* bank, int dtype, int64_t N, int D, const void* queries, int q_dtype, int Q,
                               int64_t ldq, int k, int64_t index_base, const float* norm_bound, float* out_scores,
                               int64_t* out_indices, int32_t* status, void* workspace, size_t workspace_bytes,
                               void* stream) {
    ISC_REQUIRE(bank && queries && out_scores && out_indices && status);
    ISC_REQUIRE(q_dtype == ISC_F16 || q_dtype == ISC_F32);
    const int st = check_args(dtype, N, D, Q, k);
    if (st != ISC_OK) return st;
    ISC_REQUIRE(ldq >= D);
    if (!isc_aligned(bank, 16) || !isc_aligned(workspace, 256)) return ISC_ERR_ALIGNMENT;
    size_t need = 0;
    isc_cosine_topk_workspace_bytes(dtype, N, D, Q, k, &need);
    if (!workspace || workspace_bytes < need) return ISC_ERR_WORKSPACE;
#define ISC_RUN(T_, TQ_)                                                                                          \
    return run<T_, TQ_>(bank, N, D, queries, Q, ldq, k, index_base, norm_bound, out_scores, out_indices, status, \
                        workspace, isc_stream(stream))
    if (dtype == ISC_F16) {
        if (q_dtype == ISC_F16) ISC_RUN(_Float16, _Float16);
        ISC_RUN(_Float16, float);
    }
    if (q_dtype == ISC_F16) ISC_RUN(float, _Float16);
    ISC_RUN(float, float);
#undef ISC_RUN
}
