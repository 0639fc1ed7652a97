// Merge of partial top-k lists and the data-independent exhaustive search
// (include/imagescry_hip.h: isc_topk_merge, isc_cosine_topk_exhaustive).
#include "bank_layout.h"
#include "isc_common.h"

namespace {

constexpr int MERGE_CAP = 4096;

__device__ __forceinline__ bool merge_better(float sa, int64_t ia, float sb, int64_t ib) {
    return sa > sb || (sa == sb && ia < ib);
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
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float sj = s[j];
            const int64_t ij = ix[j];
            rank += (merge_better(sj, ij, se, ie) || (sj == se && ij == ie && j < e)) ? 1 : 0;
        }
        if (rank < kout) {
            out_s[(size_t)q * kout + rank] = se;
            out_i[(size_t)q * kout + rank] = ie;
        }
    }
}

// ---- exhaustive search ----------------------------------------------------------------------------------
// grid (chunks, Q); a workgroup scores one query against one contiguous chunk of bank rows in float64
// (one wave per row, lanes across the embedding axis) and keeps each wave's best k in LDS; the
// per-(chunk, wave) lists are then merged by k_topk_merge.  Cost does not depend on the data.
constexpr int EX_MAXK = ISC_TOPK_MAX_K;

template <typename T>
__global__ __launch_bounds__(256) void k_exhaustive(const unsigned char* __restrict__ bank, int ks, int64_t n_rows,
                                                    int64_t rows_per_chunk, const T* __restrict__ queries, int64_t ldq,
                                                    int d, int k, int64_t index_base, int Q,
                                                    float* __restrict__ part_s, int64_t* __restrict__ part_i) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    double* qd = reinterpret_cast<double*>(dyn);                          // [d]
    int64_t* li = reinterpret_cast<int64_t*>(dyn + (size_t)d * 8);        // [4][EX_MAXK]
    float* ls = reinterpret_cast<float*>(li + 4 * EX_MAXK);               // [4][EX_MAXK] sorted, best first
    __shared__ double denom_sh;

    const int chunk = blockIdx.x, q = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* qp = queries + (int64_t)q * ldq;
    for (int i = tid; i < d; i += 256) qd[i] = (double)qp[i];
    __syncthreads();
    if (wave == 0) {
        double acc = 0.0;
        for (int i = lane; i < d; i += 64) acc = fma(qd[i], qd[i], acc);
        acc = isc_wave_sum(acc);
        if (lane == 0) denom_sh = fmax(sqrt(acc), 1e-12);
    }
    __syncthreads();
    const double denom = denom_sh;

    float* ws = ls + wave * EX_MAXK;
    int64_t* wi = li + wave * EX_MAXK;
    int have = 0;  // wave-uniform
    const int64_t begin = (int64_t)chunk * rows_per_chunk;
    const int64_t end = min(n_rows, begin + rows_per_chunk);
    for (int64_t row = begin + wave; row < end; row += 4) {
        double acc = 0.0;
        for (int i = lane; i < d; i += 64) acc = fma(qd[i], (double)isc_packed_load<T>(bank, row, i, ks), acc);
        acc = isc_wave_sum(acc);
        const float s = (float)(acc / denom);
        // rows arrive in ascending order, so an equal score never displaces an earlier row
        const float worst = have > 0 ? ws[have - 1] : 0.f;
        if (have < k || s > worst) {
            if (lane == 0) {
                int pos = have < k ? have : k - 1;
                while (pos > 0 && s > ws[pos - 1]) {
                    ws[pos] = ws[pos - 1];
                    wi[pos] = wi[pos - 1];
                    --pos;
                }
                ws[pos] = s;
                wi[pos] = row;
            }
            if (have < k) ++have;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    for (int j = lane; j < k; j += 64) {
        const size_t dst = (((size_t)chunk * 4 + wave) * Q + q) * k + j;
        if (j < have) {
            part_s[dst] = ws[j];
            part_i[dst] = wi[j] + index_base;
        } else {  // empty slot: ranks after every real candidate
            part_s[dst] = -INFINITY;
            part_i[dst] = INT64_MAX;
        }
    }
}

struct ExPlan {
    int chunks;
    int64_t rows_per_chunk;
    size_t s_bytes;
    size_t bytes;
};

ExPlan ex_plan(int64_t n, int q, int k) {
    ExPlan p;
    int chunks = 2048 / q;
    if (chunks < 1) chunks = 1;
    if (chunks > 64) chunks = 64;
    if ((int64_t)chunks * 4 > n) chunks = (int)isc_ceil_div<int64_t>(n, 4);
    while ((int64_t)chunks * 4 * k > MERGE_CAP && chunks > 1) --chunks;
    p.rows_per_chunk = isc_ceil_div<int64_t>(n, chunks);
    p.chunks = (int)isc_ceil_div<int64_t>(n, p.rows_per_chunk);
    p.s_bytes = isc_align_up((size_t)p.chunks * 4 * q * k * 4, 256);
    p.bytes = p.s_bytes + isc_align_up((size_t)p.chunks * 4 * q * k * 8, 256);
    return p;
}

int ex_check(int dtype, int64_t n, int d, int q, int k) {
    if (dtype != ISC_F16 && dtype != ISC_F32) return ISC_ERR_INVALID_ARG;
    if (n <= 0 || d <= 0 || q <= 0 || k <= 0 || k > n) return ISC_ERR_INVALID_ARG;
    if (k > ISC_TOPK_MAX_K || q > 65535 || d > 4096) return ISC_ERR_UNSUPPORTED;
    return ISC_OK;
}

}  // namespace

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
    *bytes = ex_plan(N, Q, k).bytes;
    return ISC_OK;
}

extern "C" int isc_cosine_topk_exhaustive(const void* bank, int dtype, int64_t N, int D, const void* queries, int Q,
                                          int64_t ldq, int k, int64_t index_base,
                                          float* out_scores, int64_t* out_indices, void* workspace,
                                          size_t workspace_bytes, void* stream) {
    ISC_REQUIRE(bank && queries && out_scores && out_indices);
    const int st = ex_check(dtype, N, D, Q, k);
    if (st != ISC_OK) return st;
    ISC_REQUIRE(ldq >= D);
    const ExPlan p = ex_plan(N, Q, k);
    if (!workspace || workspace_bytes < p.bytes) return ISC_ERR_WORKSPACE;
    float* part_s = static_cast<float*>(workspace);
    int64_t* part_i = reinterpret_cast<int64_t*>(static_cast<char*>(workspace) + p.s_bytes);
    const size_t lds = (size_t)D * 8 + 4 * EX_MAXK * (4 + 8);
    hipStream_t s = isc_stream(stream);
    if (dtype == ISC_F16)
        hipLaunchKernelGGL(k_exhaustive<_Float16>, dim3(p.chunks, Q), dim3(256), lds, s,
                           static_cast<const unsigned char*>(bank), isc_ksteps(D, 2), N, p.rows_per_chunk,
                           static_cast<const _Float16*>(queries), ldq, D, k, index_base, Q, part_s, part_i);
    else
        hipLaunchKernelGGL(k_exhaustive<float>, dim3(p.chunks, Q), dim3(256), lds, s,
                           static_cast<const unsigned char*>(bank), isc_ksteps(D, 4), N, p.rows_per_chunk, static_cast<const float*>(queries), ldq, D, k, index_base, Q,
                           part_s, part_i);
    hipLaunchKernelGGL(k_topk_merge, dim3(Q), dim3(256), 0, s, part_s, part_i, p.chunks * 4, Q, k, k, (int64_t)Q * k,
                       (int64_t)Q * k, out_scores, out_indices);
    return isc_launch_status();
}
