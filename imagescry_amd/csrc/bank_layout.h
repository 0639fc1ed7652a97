// Packed bank layout shared by the pack / search / rescore kernels (see include/imagescry_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define ISC_TILE_ROWS 256
#define ISC_KSTEP_BYTES 128
#define ISC_TILE_KSTEP_BYTES (ISC_TILE_ROWS * ISC_KSTEP_BYTES)  // 32 KiB

// number of K steps for an embedding of d elements of esz bytes
__host__ __device__ inline int isc_ksteps(int d, int esz) { return (d * esz + ISC_KSTEP_BYTES - 1) / ISC_KSTEP_BYTES; }

// byte offset of the 128-byte segment (row, kstep) inside a packed bank with `ks` K steps per row
__host__ __device__ inline int64_t isc_packed_offset(int64_t row, int kstep, int ks) {
    return (((row >> 8) * ks + kstep) * ISC_TILE_ROWS + (row & 255)) * (int64_t)ISC_KSTEP_BYTES;
}

// element e of row `row` (esz bytes per element)
template <typename T>
__device__ __forceinline__ T isc_packed_load(const void* bank, int64_t row, int e, int ks) {
    const int byte = e * (int)sizeof(T);
    const unsigned char* p = static_cast<const unsigned char*>(bank) + isc_packed_offset(row, byte >> 7, ks) + (byte & 127);
    return *reinterpret_cast<const T*>(p);
}

// ---- row permutation of a packed bank ------------------------------------------------------------------------
// A bank of N rows is stored in a fixed pseudo-random order: packed position p holds ORIGINAL row
//     orig(p) = (mul * p) mod N,          p = (mul_inv * orig) mod N,
// with mul ~ N / golden ratio, coprime to N (isc_bank_permutation).  The first L positions are then the first L
// points of a golden-ratio Kronecker sequence over [0, N): every prefix of the packed bank is an evenly spread
// sample of the original rows, whatever order they arrived in -- banks sorted or clustered by similarity (the
// reference's store returns rows in (record, h, w) order, src/imagescry/storage/operations.py:135-144: all cells of
// one image adjacent) look exchangeable to the search filter, whose thresholds warm up on prefixes.
// N < 2^31, so the products fit 64 bits.
struct IscPerm {
    int64_t n;        // rows
    int64_t mul;      // packed position -> original row
    int64_t mul_inv;  // original row    -> packed position
};
__host__ __device__ inline int64_t isc_perm_orig(const IscPerm& pm, int64_t packed_pos) {
    return (int64_t)(((unsigned long long)pm.mul * (unsigned long long)packed_pos) % (unsigned long long)pm.n);
}
__host__ __device__ inline int64_t isc_perm_pos(const IscPerm& pm, int64_t orig_row) {
    return (int64_t)(((unsigned long long)pm.mul_inv * (unsigned long long)orig_row) % (unsigned long long)pm.n);
}
// host: the permutation of an N-row bank (defined in bank_pack.hip)
IscPerm isc_make_perm(int64_t n);
