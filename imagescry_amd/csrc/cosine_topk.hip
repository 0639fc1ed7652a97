// Brute-force cosine top-k over an embedding bank (include/imagescry_hip.h: isc_cosine_topk).
//
// Pipeline per call (all on one stream, no host synchronisation):
//
//   for each level L (row ranges [0,4096), [4096,262144), [262144,16.7M), ... -- each 64x the previous):
//     k_dots_filter   S = bank[rows] . queries^T on the matrix cores, 256x256 tiles.  The scores are never
//                     written: each wave compares its accumulators with a per-query threshold tau (the kp-th
//                     best score of the rows seen in the earlier levels) and appends the few survivors
//                     (score, row) to a small per-(segment, query) buffer.  Level 0 runs with tau = -inf.
//     k_select        per query: survivors + the carried list -> the best kp by (score desc, row asc);
//                     tau <- the kp-th score.
//   k_rescore         the kp = k + slack carried candidates are re-scored EXACTLY (float64 dot, float64 query
//                     norm), rounded to float32, ordered by (score desc, row asc); the first k are the result.
//
// The matrix-core pass only has to be a superset filter; ordering and the returned scores come from the exact
// pass, so the result does not depend on tile shape, accumulation order, chunking or sharding.
//
// Data layout.  Bank and queries are row-major [rows][ld] with K (the embedding axis) contiguous -- both MFMA
// operands are "K-major", so the same staging code serves A (bank rows, the streamed operand) and B (queries).
// One K step is 128 bytes of every row (64 halves or 32 floats).  LDS tiles are [256 rows][128 B], the eight
// 16-byte chunks of a row XOR-swizzled with (row >> 1) & 7 so that a ds_read_b128 of an MFMA fragment
// (16 rows x 4 chunks per wave) is bank-conflict free; the image is lane-linear in the staging order.
#include "isc_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 256;        // bank rows per tile
constexpr int TN = 256;        // queries per tile
constexpr int NTHREADS = 512;  // 8 waves: 2 along the bank rows x 4 along the queries, 128 x 64 outputs each
constexpr int CAP = 128;       // candidate slots per (segment, query)
constexpr int64_t LEVEL0_ROWS = 4096;
constexpr int LEVEL_RATIO = 64;
constexpr int TARGET_WGS = 256;  // one workgroup per MI355X CU (the kernel needs 130 KiB of LDS)
constexpr int MAX_CHUNKS = 256;
constexpr int SELECT_CAP = 8192;  // candidates one k_select workgroup can hold in LDS
constexpr int SLACK = 6;

struct Cand {
    float s;
    int32_t row;
};

struct Plan {
    int kp;       // candidates carried per query (>= k + SLACK, multiple of 16)
    int qtiles;   // ceil(Q / 256)
    int qpad;     // qtiles * 256
    int max_seg;  // 2 * max chunks over the levels
};

struct Level {
    int64_t r0, r1;
    int ntiles, tiles_per_chunk, nchunks;
};

int plan_kp(int k) { return (int)isc_align_up((size_t)k + SLACK, 16); }

int64_t level_end(int level, int64_t n) {
    int64_t e = LEVEL0_ROWS;
    for (int i = 0; i < level; ++i) {
        if (e > n / LEVEL_RATIO + 1) return n;
        e *= LEVEL_RATIO;
    }
    return e < n ? e : n;
}

Level make_level(int level, int64_t n, int qtiles) {
    Level l;
    l.r0 = level == 0 ? 0 : level_end(level - 1, n);
    l.r1 = level_end(level, n);
    l.ntiles = (int)isc_ceil_div<int64_t>(l.r1 - l.r0, TM);
    int want = TARGET_WGS / qtiles;
    if (want < 1) want = 1;
    if (want > MAX_CHUNKS) want = MAX_CHUNKS;
    if (want > l.ntiles) want = l.ntiles;
    l.tiles_per_chunk = isc_ceil_div(l.ntiles, want);
    l.nchunks = isc_ceil_div(l.ntiles, l.tiles_per_chunk);
    return l;
}

Plan make_plan(int64_t n, int q, int k) {
    Plan p;
    p.kp = plan_kp(k);
    p.qtiles = isc_ceil_div(q, TN);
    p.qpad = p.qtiles * TN;
    p.max_seg = 0;
    for (int level = 0;; ++level) {
        const Level l = make_level(level, n, p.qtiles);
        if (2 * l.nchunks > p.max_seg) p.max_seg = 2 * l.nchunks;
        if (l.r1 >= n) break;
    }
    return p;
}

struct Workspace {
    float* tau;        // [qpad]
    float* carry_s;    // [qpad][kp]
    int32_t* carry_r;  // [qpad][kp]
    int32_t* carry_n;  // [qpad]
    int32_t* seg_cnt;  // [max_seg][qpad]
    Cand* seg_ent;     // [max_seg][qpad][CAP]
    size_t bytes;
};

Workspace carve(const Plan& p, void* base) {
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
    w.seg_cnt = static_cast<int32_t*>(take((size_t)p.max_seg * p.qpad * 4));
    w.seg_ent = static_cast<Cand*>(take((size_t)p.max_seg * p.qpad * CAP * sizeof(Cand)));
    w.bytes = off;
    return w;
}

__global__ void k_init(float* tau, int32_t* carry_n, int q, int qpad, int32_t* status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < qpad) {
        tau[i] = i < q ? -INFINITY : INFINITY;  // padding queries never pass the filter
        carry_n[i] = 0;
    }
    if (i < 4) status[i] = 0;
}

// --- operand traits -------------------------------------------------------------------------------------------
template <typename T>
struct Mma;

template <>
struct Mma<_Float16> {
    // one 16-byte chunk = 8 halves = the k-slice one lane feeds to v_mfma_f32_16x16x32_f16
    static __device__ __forceinline__ void run(const uint4 (&a)[8], const uint4 (&b)[4], f32x4 (&acc)[8][4]) {
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a[m]),
                                                                   __builtin_bit_cast(half8, b[n]), acc[m][n], 0, 0, 0);
    }
};

template <>
struct Mma<float> {
    // one 16-byte chunk = 4 floats: element j of every lane's chunk goes to the j-th v_mfma_f32_16x16x4_f32.
    // Lane group g therefore supplies k = 4 * chunk + j instead of k = g: a permutation of the K axis applied
    // identically to both operands, which leaves the dot products unchanged.
    static __device__ __forceinline__ void run(const uint4 (&a)[8], const uint4 (&b)[4], f32x4 (&acc)[8][4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const unsigned au = j == 0 ? a[m].x : j == 1 ? a[m].y : j == 2 ? a[m].z : a[m].w;
                    const unsigned bu = j == 0 ? b[n].x : j == 1 ? b[n].y : j == 2 ? b[n].z : b[n].w;
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(au), __uint_as_float(bu),
                                                                     acc[m][n], 0, 0, 0);
                }
    }
};

// LDS map: [A0 | A1 | B0 | B1] 32 KiB each, then the per-wave survivor counters.
constexpr int TILE_BYTES = TM * 128;
constexpr int LDS_CNT_OFF = 4 * TILE_BYTES;
constexpr int LDS_BYTES = LDS_CNT_OFF + 8 * 64 * 4;

template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_dots_filter(const T* __restrict__ bank, int64_t ldb, int64_t n_rows,
                                                          int64_t r0, int64_t r1, int tiles_per_chunk, int ntiles,
                                                          const T* __restrict__ queries, int64_t ldq, int n_queries,
                                                          int ksteps, const float* __restrict__ tau, int qpad,
                                                          int32_t* __restrict__ seg_cnt, Cand* __restrict__ seg_ent,
                                                          int32_t* __restrict__ status) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    int* cnt_all = reinterpret_cast<int*>(lds + LDS_CNT_OFF);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 2;  // 0..1 : which 128 bank rows of the tile
    const int wn = wave & 3;   // 0..3 : which 64 queries of the tile
    const int chunk = blockIdx.x;
    const int qt = blockIdx.y;
    const int q0 = qt * TN;

    int* cnt = cnt_all + wave * 64;
    cnt[lane] = 0;

    const int tile_begin = chunk * tiles_per_chunk;
    const int tile_end = min(ntiles, tile_begin + tiles_per_chunk);
    const int my_tiles = tile_end - tile_begin;

    // thresholds of this lane's four query columns
    float thr[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) thr[n] = tau[q0 + wn * 64 + n * 16 + (lane & 15)];

    // --- staging assignment: slot p = tid + 512 * i  ->  row p >> 3, physical chunk p & 7
    const int srow = tid >> 3;  // + 64 * i
    const int spc = tid & 7;
    const unsigned char* bbase = reinterpret_cast<const unsigned char*>(bank);
    const unsigned char* qbase = reinterpret_cast<const unsigned char*>(queries);
    const int64_t ldb_bytes = ldb * (int64_t)sizeof(T);
    const int64_t ldq_bytes = ldq * (int64_t)sizeof(T);

    const int aoff = (spc ^ ((srow >> 1) & 7)) * 16;  // (row + 64 i) >> 1 has the same low three bits
    // query rows of this tile never change: precompute their byte offsets (clamped to the last valid query)
#define ISC_QOFF(i_) \
    ((int64_t)min(q0 + srow + 64 * (i_), n_queries - 1) * ldq_bytes + ((spc ^ (((srow + 64 * (i_)) >> 1) & 7)) * 16))
    const int64_t qoff0 = ISC_QOFF(0), qoff1 = ISC_QOFF(1), qoff2 = ISC_QOFF(2), qoff3 = ISC_QOFF(3);

    // --- fragment read offsets (bytes inside a tile image)
    const int frow = lane & 15;
    const int fg = lane >> 4;
    const int fsw = (lane >> 1) & 7;
    int foff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) foff[kk] = frow * 128 + (((kk * 4 + fg) ^ fsw) << 4);
    const int a_wave_off = wm * 128 * 128;
    const int b_wave_off = wn * 64 * 128;

    f32x4 acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int total_steps = my_tiles * ksteps;
    uint4 sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3;  // named registers: an array here ends up in scratch

#define ISC_LOAD_ONE(i_, sa_, sb_)                                                                           \
    {                                                                                                        \
        int64_t grow_ = trow0_ + srow + 64 * (i_);                                                           \
        if (grow_ > n_rows - 1) grow_ = n_rows - 1;                                                          \
        sa_ = *reinterpret_cast<const uint4*>(bbase + grow_ * ldb_bytes + (int64_t)kt_ * 128 + aoff);        \
        sb_ = *reinterpret_cast<const uint4*>(qbase + qoff##i_ + (int64_t)kt_ * 128);                        \
    }
#define ISC_LOAD_STEP(step_)                                           \
    do {                                                               \
        const int t_ = (step_) / ksteps;                               \
        const int kt_ = (step_) - t_ * ksteps;                         \
        const int64_t trow0_ = r0 + (int64_t)(tile_begin + t_) * TM;   \
        ISC_LOAD_ONE(0, sa0, sb0)                                      \
        ISC_LOAD_ONE(1, sa1, sb1)                                      \
        ISC_LOAD_ONE(2, sa2, sb2)                                      \
        ISC_LOAD_ONE(3, sa3, sb3)                                      \
    } while (0)
#define ISC_STORE_STEP(buf_)                                                  \
    do {                                                                      \
        unsigned char* a_ = lds + (buf_) * TILE_BYTES + tid * 16;             \
        unsigned char* b_ = lds + (2 + (buf_)) * TILE_BYTES + tid * 16;       \
        *reinterpret_cast<uint4*>(a_) = sa0;                                  \
        *reinterpret_cast<uint4*>(a_ + NTHREADS * 16) = sa1;                  \
        *reinterpret_cast<uint4*>(a_ + NTHREADS * 32) = sa2;                  \
        *reinterpret_cast<uint4*>(a_ + NTHREADS * 48) = sa3;                  \
        *reinterpret_cast<uint4*>(b_) = sb0;                                  \
        *reinterpret_cast<uint4*>(b_ + NTHREADS * 16) = sb1;                  \
        *reinterpret_cast<uint4*>(b_ + NTHREADS * 32) = sb2;                  \
        *reinterpret_cast<uint4*>(b_ + NTHREADS * 48) = sb3;                  \
    } while (0)

    const int seg = chunk * 2 + wm;
    Cand* my_ent = seg_ent + ((size_t)seg * qpad + q0 + wn * 64) * CAP;

    if (total_steps > 0) {
        ISC_LOAD_STEP(0);
        ISC_STORE_STEP(0);
    }
    __syncthreads();

    int kt = 0, tile = 0;
    for (int step = 0; step < total_steps; ++step) {
        const int buf = step & 1;
        const bool more = step + 1 < total_steps;
        if (more) ISC_LOAD_STEP(step + 1);

        const unsigned char* a_img = lds + buf * TILE_BYTES + a_wave_off;
        const unsigned char* b_img = lds + (2 + buf) * TILE_BYTES + b_wave_off;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 a[8], b[4];
#pragma unroll
            for (int m = 0; m < 8; ++m) a[m] = *reinterpret_cast<const uint4*>(a_img + m * 2048 + foff[kk]);
#pragma unroll
            for (int n = 0; n < 4; ++n) b[n] = *reinterpret_cast<const uint4*>(b_img + n * 2048 + foff[kk]);
            Mma<T>::run(a, b, acc);
        }

        if (++kt == ksteps) {
            // ---- tile finished: threshold filter.  C layout of the 16x16 MFMA: column (query) = lane & 15,
            // row (bank row) = 4 * (lane >> 4) + register.
            kt = 0;
            bool any = false;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                float mx = -INFINITY;
#pragma unroll
                for (int m = 0; m < 8; ++m)
                    mx = fmaxf(mx, fmaxf(fmaxf(acc[m][n][0], acc[m][n][1]), fmaxf(acc[m][n][2], acc[m][n][3])));
                any |= (mx >= thr[n]);
            }
            if (__ballot(any) != 0ull) {
                const int64_t trow0 = r0 + (int64_t)(tile_begin + tile) * TM + wm * 128 + fg * 4;
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int ql = n * 16 + frow;
#pragma unroll
                    for (int m = 0; m < 8; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float s = acc[m][n][r];
                            const int64_t row = trow0 + m * 16 + r;
                            if (s >= thr[n] && row < r1) {
                                const int pos = atomicAdd(&cnt[ql], 1);
                                if (pos < CAP) my_ent[(size_t)ql * CAP + pos] = Cand{s, (int32_t)row};
                            }
                        }
                }
            }
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            ++tile;
        }

        if (more) ISC_STORE_STEP(buf ^ 1);
        __syncthreads();
    }

    // publish this wave's survivor counts
    const int c = cnt[lane];
    seg_cnt[(size_t)seg * qpad + q0 + wn * 64 + lane] = min(c, CAP);
    if (c > CAP) atomicAdd(&status[0], 1);
}

// (score desc, row asc); entries with row < 0 are empty
__device__ __forceinline__ bool better(float sa, int ra, float sb, int rb) {
    if (rb < 0) return ra >= 0;
    if (ra < 0) return false;
    return sa > sb || (sa == sb && ra < rb);
}

// One workgroup per query: gather the survivors of every segment plus the carried list, keep the best kp.
__global__ __launch_bounds__(256) void k_select(const int32_t* __restrict__ seg_cnt, const Cand* __restrict__ seg_ent,
                                                int nseg, int qpad, int kp, float* __restrict__ tau,
                                                float* __restrict__ carry_s, int32_t* __restrict__ carry_r,
                                                int32_t* __restrict__ carry_n, int32_t* __restrict__ status) {
    __shared__ Cand cand[SELECT_CAP];
    __shared__ int seg_off[2 * MAX_CHUNKS + 1];
    __shared__ float red_s[4];
    __shared__ int red_r[4], red_p[4];
    __shared__ int total_sh;

    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // exclusive scan of the segment counts (nseg <= 512: two per thread)
    int c0 = 0, c1 = 0;
    if (2 * tid < nseg) c0 = seg_cnt[(size_t)(2 * tid) * qpad + q];
    if (2 * tid + 1 < nseg) c1 = seg_cnt[(size_t)(2 * tid + 1) * qpad + q];
    int v = c0 + c1;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    __shared__ int wave_tot[4];
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wave_tot[w];
    const int excl = wbase + incl - v;
    if (2 * tid < nseg) seg_off[2 * tid] = excl;
    if (2 * tid + 1 < nseg) seg_off[2 * tid + 1] = excl + c0;
    const int carried = carry_n[q];
    if (tid == 255) {
        seg_off[nseg] = excl + v;
        total_sh = excl + v + carried;
    }
    __syncthreads();
    int total = total_sh;
    const int from_segs = seg_off[nseg];
    if (total > SELECT_CAP) {
        if (tid == 0) atomicAdd(&status[0], 1);
        total = SELECT_CAP;
    }
    // copy: each wave takes every 4th segment
    for (int s = wave; s < nseg; s += 4) {
        const int o = seg_off[s];
        const int c = seg_off[s + 1] - o;
        const Cand* src = seg_ent + ((size_t)s * qpad + q) * CAP;
        for (int i = lane; i < c; i += 64)
            if (o + i < SELECT_CAP) cand[o + i] = src[i];
    }
    for (int i = tid; i < carried; i += 256)
        if (from_segs + i < SELECT_CAP) cand[from_segs + i] = Cand{carry_s[(size_t)q * kp + i], carry_r[(size_t)q * kp + i]};
    __syncthreads();

    const int rounds = min(kp, total);
    for (int r = 0; r < rounds; ++r) {
        float bs = 0.f;
        int br = -1, bp = -1;
        for (int i = tid; i < total; i += 256) {
            const Cand c = cand[i];
            if (better(c.s, c.row, bs, br)) {
                bs = c.s;
                br = c.row;
                bp = i;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float os = __shfl_xor(bs, off, 64);
            const int orow = __shfl_xor(br, off, 64);
            const int op = __shfl_xor(bp, off, 64);
            if (better(os, orow, bs, br)) {
                bs = os;
                br = orow;
                bp = op;
            }
        }
        if (lane == 0) {
            red_s[wave] = bs;
            red_r[wave] = br;
            red_p[wave] = bp;
        }
        __syncthreads();
        if (tid == 0) {
            float ws = red_s[0];
            int wr = red_r[0], wp = red_p[0];
            for (int w = 1; w < 4; ++w)
                if (better(red_s[w], red_r[w], ws, wr)) {
                    ws = red_s[w];
                    wr = red_r[w];
                    wp = red_p[w];
                }
            carry_s[(size_t)q * kp + r] = ws;
            carry_r[(size_t)q * kp + r] = wr;
            if (wp >= 0) cand[wp].row = -1;  // taken
            if (r == kp - 1) tau[q] = ws;
        }
        __syncthreads();
    }
    if (tid == 0) carry_n[q] = rounds;
}

// One workgroup per query: exact float64 re-score of the carried candidates, final order, output.
template <typename T>
__global__ __launch_bounds__(256) void k_rescore(const T* __restrict__ bank, int64_t ldb, const T* __restrict__ queries,
                                                 int64_t ldq, int d, int kp, int k, int64_t index_base,
                                                 const int32_t* __restrict__ carry_r,
                                                 const int32_t* __restrict__ carry_n, float* __restrict__ out_s,
                                                 int64_t* __restrict__ out_i) {
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
        const T* bp = bank + (int64_t)row * ldb;
        double acc = 0.0;
        for (int i = lane; i < d; i += 64) acc = fma((double)qp[i], (double)bp[i], acc);
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

template <typename T>
int run(const void* bank, int64_t n, int d, int64_t ldb, const void* queries, int q, int64_t ldq, int k,
        int64_t index_base, float* out_s, int64_t* out_i, int32_t* status, void* ws_base, hipStream_t stream) {
    const Plan p = make_plan(n, q, k);
    const Workspace w = carve(p, ws_base);
    const int ksteps = d * (int)sizeof(T) / 128;
    hipLaunchKernelGGL(k_init, dim3(isc_ceil_div(p.qpad, 256)), dim3(256), 0, stream, w.tau, w.carry_n, q, p.qpad,
                       status);
    for (int level = 0;; ++level) {
        const Level l = make_level(level, n, p.qtiles);
        isc_timing_begin(ISC_KERNEL_DOTS_FILTER, stream);
        hipLaunchKernelGGL(k_dots_filter<T>, dim3(l.nchunks, p.qtiles), dim3(NTHREADS), 0, stream,
                           static_cast<const T*>(bank), ldb, n, l.r0, l.r1, l.tiles_per_chunk, l.ntiles,
                           static_cast<const T*>(queries), ldq, q, ksteps, w.tau, p.qpad, w.seg_cnt, w.seg_ent, status);
        isc_timing_end(ISC_KERNEL_DOTS_FILTER, stream);
        hipLaunchKernelGGL(k_select, dim3(q), dim3(256), 0, stream, w.seg_cnt, w.seg_ent, 2 * l.nchunks, p.qpad, p.kp,
                           w.tau, w.carry_s, w.carry_r, w.carry_n, status);
        if (l.r1 >= n) break;
    }
    hipLaunchKernelGGL(k_rescore<T>, dim3(q), dim3(256), 0, stream, static_cast<const T*>(bank), ldb,
                       static_cast<const T*>(queries), ldq, d, p.kp, k, index_base, w.carry_r, w.carry_n, out_s, out_i);
    return isc_launch_status();
}

int check_args(int dtype, int64_t n, int d, int q, int k) {
    if (dtype != ISC_F16 && dtype != ISC_F32) return ISC_ERR_INVALID_ARG;
    if (n <= 0 || d <= 0 || q <= 0 || k <= 0 || k > n) return ISC_ERR_INVALID_ARG;
    if (k > ISC_TOPK_MAX_K) return ISC_ERR_UNSUPPORTED;
    if (n > 0x7fffffff) return ISC_ERR_UNSUPPORTED;  // row ids are int32 inside a shard
    const int esz = dtype == ISC_F16 ? 2 : 4;
    if ((d * esz) % 128 != 0) return ISC_ERR_UNSUPPORTED;  // pad the embedding axis with zero columns
    if (isc_ceil_div(q, TN) > 65535) return ISC_ERR_UNSUPPORTED;
    return ISC_OK;
}

}  // namespace

extern "C" int isc_cosine_topk_workspace_bytes(int dtype, int64_t N, int D, int Q, int k, size_t* bytes) {
    ISC_REQUIRE(bytes);
    const int st = check_args(dtype, N, D, Q, k);
    if (st != ISC_OK) return st;
    *bytes = carve(make_plan(N, Q, k), nullptr).bytes;
    return ISC_OK;
}

extern "C" int isc_cosine_topk(const void* bank, int dtype, int64_t N, int D, int64_t ldb, const void* queries, int Q,
                               int64_t ldq, int k, int64_t index_base, float* out_scores, int64_t* out_indices,
                               int32_t* status, void* workspace, size_t workspace_bytes, void* stream) {
    ISC_REQUIRE(bank && queries && out_scores && out_indices && status);
    const int st = check_args(dtype, N, D, Q, k);
    if (st != ISC_OK) return st;
    ISC_REQUIRE(ldb >= D && ldq >= D);
    const int esz = dtype == ISC_F16 ? 2 : 4;
    if (!isc_aligned(bank, 16) || !isc_aligned(queries, 16) || (ldb * esz) % 16 != 0 || (ldq * esz) % 16 != 0)
        return ISC_ERR_ALIGNMENT;
    size_t need = 0;
    isc_cosine_topk_workspace_bytes(dtype, N, D, Q, k, &need);
    if (!workspace || workspace_bytes < need) return ISC_ERR_WORKSPACE;
    if (dtype == ISC_F16)
        return run<_Float16>(bank, N, D, ldb, queries, Q, ldq, k, index_base, out_scores, out_indices, status,
                             workspace, isc_stream(stream));
    return run<float>(bank, N, D, ldb, queries, Q, ldq, k, index_base, out_scores, out_indices, status, workspace,
                      isc_stream(stream));
}
