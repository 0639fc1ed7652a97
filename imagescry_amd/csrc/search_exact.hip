// Exact float64 search of a device-side LIST of queries, the merge of partial top-k lists, and the data-independent
// exhaustive search (include/imagescry_hip.h: isc_topk_merge, isc_cosine_topk_exhaustive).
//
// k_exact is the safety net of isc_cosine_topk: k_final lists the queries whose result the float32 filter cannot
// prove (rounding guard, overflowed candidate buffers, NaN scores) and k_exact searches exactly those again, every
// score evaluated in float64, overwriting their output rows.  It is launched after every search and exits at once
// when the list is empty, so the host never has to look at a status word.
#include "bank_layout.h"
#include "isc_common.h"
#include "search_common.h"

namespace {

constexpr int EX_THREADS = 512;
constexpr int EX_WAVES = EX_THREADS / 64;
constexpr int EX_MAX_CHUNKS = 256;
constexpr int EX_PASS = 1024;  // listed queries per pass of isc_cosine_topk_exhaustive

int ex_chunks(int64_t n, int k) {
    const int64_t ntiles = isc_ceil_div<int64_t>(n, ISC_TILE_ROWS);
    int want = 4096 / k;  // keeps the partial lists of a query <= 32 KiB
    if (want < 32) want = 32;
    if (want > EX_MAX_CHUNKS) want = EX_MAX_CHUNKS;
    if (want > ntiles) want = (int)ntiles;
    return want;
}

// 16 bytes of a packed row as float64 values
template <typename T>
struct Chunk;
template <>
struct Chunk<_Float16> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const unsigned char* p, double (&v)[8]) {
        const uint4 raw = *reinterpret_cast<const uint4*>(p);
        const _Float16* h = reinterpret_cast<const _Float16*>(&raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (double)(float)h[j];
    }
};
template <>
struct Chunk<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const unsigned char* p, double (&v)[8]) {
        const float4 raw = *reinterpret_cast<const float4*>(p);
        v[0] = (double)raw.x;
        v[1] = (double)raw.y;
        v[2] = (double)raw.z;
        v[3] = (double)raw.w;
    }
};

__device__ __forceinline__ double group8_sum(double v) {  // over the 8 lanes that share a bank row
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}

// grid = chunks of consecutive bank tiles, 512 threads.  GQ listed queries are scored per sweep of the chunk: the
// queries sit in LDS as float64, a wave takes 8 bank rows at a time (lane = row l >> 3, 16-byte chunk l & 7 of every K
// step: each load instruction reads 1 KiB of contiguous HBM), every product and sum is float64.  Each wave keeps a
// sorted list of its best k keys (exact float32 score, ORIGINAL row) per query; per chunk they are merged into one sorted
// list per query and published; the workgroup that publishes last merges the chunks' lists (k-way merge of sorted
// lists by one wave per query) and writes the result rows.
template <typename T, int GQ>
__global__ __launch_bounds__(EX_THREADS) void k_exact(const unsigned char* __restrict__ bank, int ks, IscPerm pm,
                                                      int ntiles, int tiles_per_chunk, const void* __restrict__ queries,
                                                      int q_f32, int64_t ldq, int d, int k, int64_t index_base,
                                                      const int32_t* __restrict__ redo_count,
                                                      const int32_t* __restrict__ redo_list,
                                                      unsigned long long* __restrict__ part, int32_t* __restrict__ done,
                                                      float* __restrict__ out_s, int64_t* __restrict__ out_i,
                                                      int32_t* __restrict__ status) {
    const int nf = *redo_count;
    if (nf <= 0) return;  // the normal case
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    constexpr int EPK = ISC_KSTEP_BYTES / (int)sizeof(T);  // elements per K step
    constexpr int PER = Chunk<T>::N;
    const int dp = ks * EPK;
    double* qd = reinterpret_cast<double*>(dyn);                                     // [GQ][dp]
    unsigned long long* lists = reinterpret_cast<unsigned long long*>(qd + (size_t)GQ * dp);  // [GQ][EX_WAVES][k]
    __shared__ double denom_sh[GQ];
    __shared__ int last_sh;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = blockIdx.x, chunks = gridDim.x;
    const int tile_begin = chunk * tiles_per_chunk;
    const int tile_end = min(ntiles, tile_begin + tiles_per_chunk);
    const int sub = lane >> 3, ch = lane & 7;

    for (int g0 = 0; g0 < nf; g0 += GQ) {
        const int gn = min(GQ, nf - g0);
        __syncthreads();  // the previous group's lists have been merged
        for (int g = 0; g < GQ; ++g) {
            const int qi = g < gn ? redo_list[g0 + g] : 0;
            // the caller's query elements (fp16 or float32, `q_f32`) rounded to the bank type first, as k_prep packs them
            const float* qp32 = static_cast<const float*>(queries) + (int64_t)qi * ldq;
            const _Float16* qp16 = static_cast<const _Float16*>(queries) + (int64_t)qi * ldq;
            for (int e = tid; e < dp; e += EX_THREADS) {
                double v = 0.0;
                if (g < gn && e < d) v = (double)(float)(T)(q_f32 ? qp32[e] : (float)qp16[e]);
                qd[(size_t)g * dp + e] = v;
            }
        }
        for (int i = tid; i < GQ * EX_WAVES * k; i += EX_THREADS) lists[i] = 0ull;
        __syncthreads();
        if (wave < GQ) {
            double acc = 0.0;
            for (int e = lane; e < dp; e += 64) acc = fma(qd[(size_t)wave * dp + e], qd[(size_t)wave * dp + e], acc);
            acc = isc_wave_sum(acc);
            if (lane == 0) denom_sh[wave] = fmax(sqrt(acc), 1e-12);
        }
        __syncthreads();
        double denom[GQ];
#pragma unroll
        for (int g = 0; g < GQ; ++g) denom[g] = denom_sh[g];

        for (int tile = tile_begin; tile < tile_end; ++tile) {
            for (int grp = wave; grp < ISC_TILE_ROWS / 8; grp += EX_WAVES) {
                const int64_t p = (int64_t)tile * ISC_TILE_ROWS + grp * 8 + sub;
                const unsigned char* src =
                    bank + ((int64_t)tile * ks * ISC_TILE_ROWS + grp * 8 + sub) * ISC_KSTEP_BYTES + ch * 16;
                double acc[GQ];
#pragma unroll
                for (int g = 0; g < GQ; ++g) acc[g] = 0.0;
                for (int s = 0; s < ks; ++s) {
                    double a[8];
                    Chunk<T>::load(src + (size_t)s * ISC_TILE_KSTEP_BYTES, a);
                    const double* qs = qd + s * EPK + ch * PER;
#pragma unroll
                    for (int j = 0; j < PER; ++j)
#pragma unroll
                        for (int g = 0; g < GQ; ++g) acc[g] = fma(qs[(size_t)g * dp + j], a[j], acc[g]);
                }
#pragma unroll
                for (int g = 0; g < GQ; ++g) {
                    if (g >= gn) break;
                    const float sc = (float)(group8_sum(acc[g]) / denom[g]);
                    unsigned long long* wl = lists + ((size_t)g * EX_WAVES + wave) * k;
                    const unsigned long long worst = wl[k - 1];  // 0 while the list is not full
                    const bool cand = ch == 0 && p < pm.n && isc_score_bits(sc) >= (unsigned)(worst >> 32);
                    unsigned long long mask = __ballot(cand);
                    if (mask == 0ull) continue;
                    const unsigned long long key = cand ? isc_make_key(sc, (int)isc_perm_orig(pm, p)) : 0ull;
                    while (mask != 0ull) {
                        const int b = __builtin_ctzll(mask);
                        mask &= mask - 1ull;
                        const unsigned long long kb = isc_bcast_key(key, b);
                        if (kb > wl[k - 1]) {  // wave-uniform
                            if (lane == 0) {
                                int pos = k - 1;
                                while (pos > 0 && kb > wl[pos - 1]) {
                                    wl[pos] = wl[pos - 1];
                                    --pos;
                                }
                                wl[pos] = kb;
                            }
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        }
                    }
                }
            }
        }
        __syncthreads();
        // the chunk's list of every query of the group: rank sort of the EX_WAVES sorted lists (empty slots are 0)
        for (int g = wave; g < gn; g += EX_WAVES) {
            const unsigned long long* all = lists + (size_t)g * EX_WAVES * k;
            unsigned long long* dst = part + ((size_t)(g0 + g) * chunks + chunk) * k;
            const int tot = EX_WAVES * k;
            for (int e = lane; e < tot; e += 64) {
                const unsigned long long mine = all[e];
                int rank = 0;
                for (int j = 0; j < tot; ++j) {
                    const unsigned long long o = all[j];
                    rank += (o > mine || (o == mine && j < e)) ? 1 : 0;
                }
                if (rank < k) dst[rank] = mine;
            }
        }
    }

    // ---- publish; the last workgroup to arrive merges (cdna_hip_programming.md guideline 16: every storing wave drains
    // its stores, workgroup barrier, one agent-scope release, asm wait, then the counter)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int old = atomicAdd(done, 1);
        last_sh = old == chunks - 1 ? 1 : 0;
        if (last_sh) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!last_sh) return;

    // k-way merge: a lane holds the heads of lists lane, lane + 64, ... (chunks <= 256: at most four)
    for (int f = wave; f < nf; f += EX_WAVES) {
        const int qi = redo_list[f];
        const unsigned long long* base = part + (size_t)f * chunks * k;
        int pos[4];
        unsigned long long cur[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane + 64 * j;
            pos[j] = 0;
            cur[j] = c < chunks ? __hip_atomic_load(base + (size_t)c * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
        for (int r = 0; r < k; ++r) {
            unsigned long long best = cur[0];
#pragma unroll
            for (int j = 1; j < 4; ++j) best = cur[j] > best ? cur[j] : best;
            const unsigned long long win = isc_wave_max_key(best);
            if (lane == 0) {
                out_s[(size_t)qi * k + r] = isc_key_score(win);
                out_i[(size_t)qi * k + r] = (int64_t)isc_key_row(win) + index_base;
            }
            if (win != 0ull) {  // keys of distinct rows are distinct: exactly one head matches
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (cur[j] == win) {
                        const int c = lane + 64 * j;
                        ++pos[j];
                        cur[j] = pos[j] < k ? __hip_atomic_load(base + (size_t)c * k + pos[j], __ATOMIC_RELAXED,
                                                                __HIP_MEMORY_SCOPE_AGENT)
                                            : 0ull;
                    }
            }
        }
    }
    if (tid == 0) *done = 0;  // ready for the next launch
    (void)status;
}

__global__ void k_list_all(int32_t* redo_count, int32_t* redo_list, int32_t* done, int q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < q) redo_list[i] = i;
    if (i == 0) {
        *redo_count = q;
        *done = 0;
    }
}

// ---- merge of partial top-k lists ------------------------------------------------------------------------------
constexpr int MERGE_CAP = 4096;

// (score desc with NaN last, index asc)
__device__ __forceinline__ bool merge_better(float sa, int64_t ia, float sb, int64_t ib) {
    const unsigned ua = isc_score_bits(sa), ub = isc_score_bits(sb);
    return ua > ub || (ua == ub && ia < ib);
}

// One workgroup per query.  Rank sort: entry i lands at position #{j better than i}.  Entries are
// distinct (score, index) pairs unless the same row appears in two partial lists; duplicates of an
// identical pair are broken by position so every rank is still unique.
__global__ __launch_bounds__(256) void k_topk_merge(const float* __restrict__ scores,
                                                    const int64_t* __restrict__ indices, int G, int Q, int kin,
                                                    int kout, int64_t gs_scores, int64_t gs_indices,
                                                    float* __restrict__ out_s, int64_t* __restrict__ out_i) {
    __shared__ float s[MERGE_CAP];
    __shared__ int64_t ix[MERGE_CAP];
    const int q = blockIdx.x;
    const int n = G * kin;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int g = e / kin, j = e - g * kin;
        const size_t inner = (size_t)q * kin + j;
        s[e] = scores[(size_t)g * gs_scores + inner];
        ix[e] = indices[(size_t)g * gs_indices + inner];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n; e += 256) {
        const float se = s[e];
        const int64_t ie = ix[e];
        const unsigned ue = isc_score_bits(se);
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float sj = s[j];
            const int64_t ij = ix[j];
            rank += (merge_better(sj, ij, se, ie) || (isc_score_bits(sj) == ue && ij == ie && j < e)) ? 1 : 0;
        }
        if (rank < kout) {
            out_s[(size_t)q * kout + rank] = se;
            out_i[(size_t)q * kout + rank] = ie;
        }
    }
}

int ex_check(int dtype, int64_t n, int d, int q, int k) {
    if (dtype != ISC_F16 && dtype != ISC_F32) return ISC_ERR_INVALID_ARG;
    if (n <= 0 || d <= 0 || q <= 0 || k <= 0 || k > n) return ISC_ERR_INVALID_ARG;
    if (k > ISC_TOPK_MAX_K || q > ISC_SEARCH_MAX_Q || d > ISC_SEARCH_MAX_D || n > 0x7ffffffe) return ISC_ERR_UNSUPPORTED;
    return ISC_OK;
}

template <typename T, int GQ>
int launch_exact(const void* bank, int64_t n, int d, const void* queries, int q_f32, int64_t ldq, int k, int64_t index_base,
                  const IscExactWs& ws, float* out_s, int64_t* out_i, int32_t* status, hipStream_t stream) {
    const int ks = isc_ksteps(d, (int)sizeof(T));
    const int dp = ks * (ISC_KSTEP_BYTES / (int)sizeof(T));
    const size_t lds = (size_t)GQ * dp * 8 + (size_t)GQ * EX_WAVES * k * 8;
    const int ntiles = (int)isc_ceil_div<int64_t>(n, ISC_TILE_ROWS);
    // more than 64 KiB of dynamic LDS needs the opt-in, once per kernel AND device (a process may drive several GPUs):
    // one bit per device id, set only after the call succeeded; a failure is reported, not discarded
    static unsigned long long attr_done = 0ull;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ISC_ERR_NO_DEVICE;
    const bool tracked = dev >= 0 && dev < 64;
    if (!tracked || !((__atomic_load_n(&attr_done, __ATOMIC_RELAXED) >> dev) & 1ull)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_exact<T, GQ>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return ISC_ERR_UNSUPPORTED;
        }
        if (tracked) __atomic_fetch_or(&attr_done, 1ull << dev, __ATOMIC_RELAXED);
    }
    hipLaunchKernelGGL((k_exact<T, GQ>), dim3(ws.chunks), dim3(EX_THREADS), lds, stream,
                       static_cast<const unsigned char*>(bank), ks, isc_make_perm(n), ntiles, ws.tiles_per_chunk,
                       queries, q_f32, ldq, d, k, index_base, ws.redo_count, ws.redo_list, ws.part,
                       ws.done, out_s, out_i, status);
    return ISC_OK;
}

template <typename T>
int launch_exact_t(const void* bank, int64_t n, int d, const void* queries, int q_f32, int64_t ldq, int k, int64_t index_base,
                    const IscExactWs& ws, float* out_s, int64_t* out_i, int32_t* status, hipStream_t stream) {
    const int dp = isc_ksteps(d, (int)sizeof(T)) * (ISC_KSTEP_BYTES / (int)sizeof(T));
    if (dp <= 3072) return launch_exact<T, 4>(bank, n, d, queries, q_f32, ldq, k, index_base, ws, out_s, out_i, status, stream);
    if (dp <= 6144) return launch_exact<T, 2>(bank, n, d, queries, q_f32, ldq, k, index_base, ws, out_s, out_i, status, stream);
    return launch_exact<T, 1>(bank, n, d, queries, q_f32, ldq, k, index_base, ws, out_s, out_i, status, stream);
}

}  // namespace

size_t isc_exact_ws_bytes(int64_t n, int q, int k) {
    return 256 + isc_align_up((size_t)q * 4, 256) + isc_align_up((size_t)q * ex_chunks(n, k) * k * 8, 256);
}

IscExactWs isc_exact_ws_carve(void* base, int64_t n, int q, int k) {
    IscExactWs w;
    char* b = static_cast<char*>(base);
    w.redo_count = base ? reinterpret_cast<int32_t*>(b) : nullptr;
    w.done = base ? reinterpret_cast<int32_t*>(b + 128) : nullptr;
    w.redo_list = base ? reinterpret_cast<int32_t*>(b + 256) : nullptr;
    w.part = base ? reinterpret_cast<unsigned long long*>(b + 256 + isc_align_up((size_t)q * 4, 256)) : nullptr;
    const int ntiles = (int)isc_ceil_div<int64_t>(n, ISC_TILE_ROWS);
    const int want = ex_chunks(n, k);
    w.tiles_per_chunk = isc_ceil_div(ntiles, want);
    w.chunks = isc_ceil_div(ntiles, w.tiles_per_chunk);
    return w;
}

int isc_exact_launch(int dtype, const void* bank, int64_t n, int d, const void* queries, int q_dtype, int64_t ldq, int k,
                     int64_t index_base, const IscExactWs& ws, float* out_s, int64_t* out_i, int32_t* status,
                     hipStream_t stream) {
    const int qf = q_dtype == ISC_F32 ? 1 : 0;
    const int st = dtype == ISC_F16
                       ? launch_exact_t<_Float16>(bank, n, d, queries, qf, ldq, k, index_base, ws, out_s, out_i, status, stream)
                       : launch_exact_t<float>(bank, n, d, queries, qf, ldq, k, index_base, ws, out_s, out_i, status, stream);
    return st != ISC_OK ? st : isc_launch_status();
}

extern "C" int isc_topk_merge(const float* scores, const int64_t* indices, int G, int Q, int kin, int kout,
                              int64_t stride_g_scores, int64_t stride_g_indices, float* out_scores,
                              int64_t* out_indices, void* stream) {
    ISC_REQUIRE(scores && indices && out_scores && out_indices);
    ISC_REQUIRE(G > 0 && Q > 0 && kin > 0 && kout > 0);
    if ((int64_t)G * kin > MERGE_CAP) return ISC_ERR_UNSUPPORTED;
    ISC_REQUIRE(kout <= G * kin);
    const int64_t dense = (int64_t)Q * kin;
    if (stride_g_scores == 0) stride_g_scores = dense;
    if (stride_g_indices == 0) stride_g_indices = dense;
    ISC_REQUIRE(stride_g_scores >= dense && stride_g_indices >= dense);
    hipLaunchKernelGGL(k_topk_merge, dim3(Q), dim3(256), 0, isc_stream(stream), scores, indices, G, Q, kin, kout,
                       stride_g_scores, stride_g_indices, out_scores, out_indices);
    return isc_launch_status();
}

extern "C" int isc_cosine_topk_exhaustive_workspace_bytes(int dtype, int64_t N, int D, int Q, int k, size_t* bytes) {
    ISC_REQUIRE(bytes);
    const int st = ex_check(dtype, N, D, Q, k);
    if (st != ISC_OK) return st;
    *bytes = isc_exact_ws_bytes(N, Q < EX_PASS ? Q : EX_PASS, k);
    return ISC_OK;
}

extern "C" int isc_cosine_topk_exhaustive(const void* bank, int dtype, int64_t N, int D, const void* queries, int q_dtype,
                                          int Q, int64_t ldq, int k, int64_t index_base, float* out_scores,
                                          int64_t* out_indices, void* workspace, size_t workspace_bytes, void* stream) {
    ISC_REQUIRE(bank && queries && out_scores && out_indices);
    ISC_REQUIRE(q_dtype == ISC_F16 || q_dtype == ISC_F32);
    const int st = ex_check(dtype, N, D, Q, k);
    if (st != ISC_OK) return st;
    ISC_REQUIRE(ldq >= D);
    const int qb = Q < EX_PASS ? Q : EX_PASS;
    if (!workspace || workspace_bytes < isc_exact_ws_bytes(N, qb, k)) return ISC_ERR_WORKSPACE;
    if (!isc_aligned(workspace, 256)) return ISC_ERR_ALIGNMENT;
    const IscExactWs ws = isc_exact_ws_carve(workspace, N, qb, k);
    hipStream_t s = isc_stream(stream);
    const size_t esz = q_dtype == ISC_F16 ? 2 : 4;
    for (int q0 = 0; q0 < Q; q0 += qb) {
        const int q = Q - q0 < qb ? Q - q0 : qb;
        hipLaunchKernelGGL(k_list_all, dim3(isc_ceil_div(q, 256)), dim3(256), 0, s, ws.redo_count, ws.redo_list, ws.done,
                           q);
        const int st2 = isc_exact_launch(dtype, bank, N, D, static_cast<const char*>(queries) + (size_t)q0 * ldq * esz,
                                         q_dtype, ldq, k, index_base, ws, out_scores + (size_t)q0 * k,
                                         out_indices + (size_t)q0 * k, nullptr, s);
        if (st2 != ISC_OK) return st2;
    }
    return isc_launch_status();
}
