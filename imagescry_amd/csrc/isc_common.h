// Shared host/device helpers for the gfx950 kernels behind include/imagescry_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/imagescry_hip.h"

#define ISC_WAVE 64

#define ISC_REQUIRE(cond) \
    do {                  \
        if (!(cond)) return ISC_ERR_INVALID_ARG; \
    } while (0)

static inline int isc_launch_status() { return hipGetLastError() == hipSuccess ? ISC_OK : ISC_ERR_LAUNCH; }

static inline hipStream_t isc_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline bool isc_aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// Compute units of the CURRENT device, cached per device id (a process may drive several GPUs; thread-safe: the slots
// are written with the same value by whoever gets there first).  256 when the query fails.
static inline int isc_device_cus() {
    static int cached[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev >= 0 && dev < 64) {
        const int c = __atomic_load_n(&cached[dev], __ATOMIC_RELAXED);
        if (c > 0) return c;
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (dev >= 0 && dev < 64) __atomic_store_n(&cached[dev], cus, __ATOMIC_RELAXED);
    return cus;
}

// device-time bracket around the launch(es) of one instrumented kernel (no-ops unless isc_timing_enable(1); defined in
// capi.hip).  `kernels` = how many launches of that kernel the bracket holds: isc_timing_read reports KERNEL launches, the
// unit a rocprofv3 kernel trace counts in.
void isc_timing_begin(int kernel_id, hipStream_t stream);
void isc_timing_end(int kernel_id, hipStream_t stream, int kernels = 1);

template <typename T>
static inline T isc_ceil_div(T a, T b) {
    return (a + b - 1) / b;
}

static inline size_t isc_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- wave-level reductions (64 lanes) ---------------------------------------------------------
__device__ __forceinline__ float isc_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double isc_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ unsigned long long isc_wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float isc_wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
