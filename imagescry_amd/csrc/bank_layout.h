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
