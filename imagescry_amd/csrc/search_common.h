// Pieces shared by the search kernels (cosine_topk.hip) and the exact float64 search (search_exact.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bank_layout.h"

// ---- (score, row) as one 64-bit key whose unsigned order is the search order ------------------------------------
// larger key = better candidate: higher score first, then LOWER row.  Keys of distinct rows are distinct.  A NaN
// score ranks below every number (the oracle's lexsort puts NaN last), and -0.0 is the same score as +0.0 (as it is to
// the oracle's comparison: a zero query scores -0.0 against a row with no positive entry, and that row ties every
// other).  0 is "empty": a real entry has row <= 2^31 - 2, so its low word is >= 1.
__device__ __forceinline__ unsigned isc_score_bits(float s) {
    if (s != s) return 0u;
    unsigned u = s == 0.f ? 0u : __float_as_uint(s);  // the bits of +0.0 for both zeros
    u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;  // monotone float -> unsigned; -inf -> 0x007fffff
    return u;
}
__device__ __forceinline__ unsigned long long isc_make_key(float s, int row) {
    return ((unsigned long long)isc_score_bits(s) << 32) | (unsigned)(0x7fffffff - row);
}
__device__ __forceinline__ float isc_key_score(unsigned long long k) {
    unsigned u = (unsigned)(k >> 32);
    if (u == 0u) return __uint_as_float(0x7fc00000u);  // NaN
    u ^= (u >> 31) ? 0x80000000u : 0xffffffffu;
    return __uint_as_float(u);
}
__device__ __forceinline__ int isc_key_row(unsigned long long k) {
    return 0x7fffffff - (int)(unsigned)(k & 0xffffffffu);
}
__device__ __forceinline__ unsigned long long isc_bcast_key(unsigned long long k, int src_lane) {
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)k, src_lane);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(k >> 32), src_lane);
    return ((unsigned long long)hi << 32) | lo;
}
// Maximum over the wave, left in every lane.  Inside a 16-lane row the partner's key comes through a DPP move (quad
// permutes, then the half-row and row mirrors: any exchange that pairs the right sub-groups does for a maximum); only
// the two steps across rows go through the LDS permute unit.  k_select / k_final call this kp times per query to peel
// off the kp largest lane maxima: with __shfl_xor at all six steps that was a third of k_select's time at Q <= 64.
template <int CTRL>
__device__ __forceinline__ void isc_key_max_dpp(unsigned& lo, unsigned& hi) {
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, CTRL, 0xf, 0xf, true);
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, CTRL, 0xf, 0xf, true);
    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
    const unsigned long long m = ((unsigned long long)hi << 32) | lo;
    if (o > m) {
        lo = olo;
        hi = ohi;
    }
}
__device__ __forceinline__ unsigned long long isc_wave_max_key(unsigned long long k) {
    unsigned lo = (unsigned)k, hi = (unsigned)(k >> 32);
    isc_key_max_dpp<0xB1>(lo, hi);   // quad_perm [1,0,3,2]
    isc_key_max_dpp<0x4E>(lo, hi);   // quad_perm [2,3,0,1]
    isc_key_max_dpp<0x141>(lo, hi);  // row_half_mirror
    isc_key_max_dpp<0x140>(lo, hi);  // row_mirror
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        const unsigned olo = __shfl_xor(lo, off, 64), ohi = __shfl_xor(hi, off, 64);
        const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
        const unsigned long long m = ((unsigned long long)hi << 32) | lo;
        if (o > m) {
            lo = olo;
            hi = ohi;
        }
    }
    return ((unsigned long long)hi << 32) | lo;
}

// ---- exact float64 search of a LIST of queries (search_exact.hip) -------------------------------------------------
// Workspace of k_exact: the device-side list of queries to search (filled by k_final, or with every query by
// isc_cosine_topk_exhaustive), one sorted partial list of k keys per (listed query, chunk of bank tiles), and the
// arrival counter of the last-workgroup merge.
struct IscExactWs {
    int32_t* redo_count;       // [1]  number of listed queries
    int32_t* redo_list;        // [q]  their indices (into this pass's queries)
    int32_t* done;             // [1]  workgroups that have published their partial lists
    unsigned long long* part;  // [q][chunks][k]  keys (exact float32 score, ORIGINAL row), best first
    int chunks;                // workgroups of k_exact
    int tiles_per_chunk;
};
size_t isc_exact_ws_bytes(int64_t n, int q, int k);
IscExactWs isc_exact_ws_carve(void* base, int64_t n, int q, int k);
// enqueue k_exact: searches queries redo_list[0 .. *redo_count) and writes rows redo_list[i] of out_s / out_i
// (leading dimension k).  Queries of `q_dtype` at `queries` with leading dimension ldq (elements of q_dtype); every
// element is rounded to the bank's `dtype` first, as isc_cosine_topk does when it packs them.
int isc_exact_launch(int dtype, const void* bank, int64_t n, int d, const void* queries, int q_dtype, int64_t ldq, int k,
                     int64_t index_base, const IscExactWs& ws, float* out_s, int64_t* out_i, int32_t* status,
                     hipStream_t stream);
