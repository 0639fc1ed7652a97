// Library-level entry points of include/imagescry_hip.h.
#include <string.h>

#include <mutex>
#include <utility>
#include <vector>

#include "isc_common.h"

extern "C" int isc_abi_version(void) { return ISC_ABI_VERSION; }

extern "C" int isc_build_flags(void) {
#ifdef ISC_ABLATION
    return ISC_BUILD_ABLATION;
#else
    return 0;
#endif
}

extern "C" const char* isc_strerror(int status) {
    switch (status) {
        case ISC_OK: return "ok";
        case ISC_ERR_INVALID_ARG: return "invalid argument (null pointer, non-positive size or unknown enum)";
        case ISC_ERR_UNSUPPORTED: return "unsupported shape or dtype for this build";
        case ISC_ERR_WORKSPACE: return "workspace missing or too small";
        case ISC_ERR_LAUNCH: return "HIP kernel launch failed";
        case ISC_ERR_NO_DEVICE: return "no usable HIP device";
        case ISC_ERR_ALIGNMENT: return "pointer or leading dimension not 16-byte aligned";
        default: return "unknown status";
    }
}

extern "C" int isc_device_info(int* num_cus, int* lds_bytes_per_cu, char* arch_name, int arch_name_len) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ISC_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return ISC_ERR_NO_DEVICE;
    if (num_cus) *num_cus = prop.multiProcessorCount;
    if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (arch_name && arch_name_len > 0) {
        strncpy(arch_name, prop.gcnArchName, (size_t)arch_name_len - 1);
        arch_name[arch_name_len - 1] = '\0';
    }
    return ISC_OK;
}

// ---- per-kernel device timing ---------------------------------------------------------------------------
namespace {
struct TimingState {
    std::mutex mu;
    bool enabled = false;
    struct Bracket {
        hipEvent_t first, second;
        int kernels;  // kernel launches inside the bracket (a convolution may run as two: whole rounds + half-tile remainder)
    };
    std::vector<Bracket> pending[ISC_KERNEL_COUNT];
    std::vector<hipEvent_t> pool;
    hipEvent_t open_begin[ISC_KERNEL_COUNT] = {};
};
TimingState& timing() {
    static TimingState t;
    return t;
}
hipEvent_t take_event(TimingState& t) {
    if (!t.pool.empty()) {
        hipEvent_t e = t.pool.back();
        t.pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

void isc_timing_begin(int kernel_id, hipStream_t stream) {
    TimingState& t = timing();
    if (!t.enabled) return;
    std::lock_guard<std::mutex> lock(t.mu);
    hipEvent_t e = take_event(t);
    if (!e) return;
    (void)hipEventRecord(e, stream);
    t.open_begin[kernel_id] = e;
}

void isc_timing_end(int kernel_id, hipStream_t stream, int kernels) {
    TimingState& t = timing();
    if (!t.enabled) return;
    std::lock_guard<std::mutex> lock(t.mu);
    hipEvent_t b = t.open_begin[kernel_id];
    if (!b) return;
    t.open_begin[kernel_id] = nullptr;
    hipEvent_t e = take_event(t);
    if (!e) {
        t.pool.push_back(b);
        return;
    }
    (void)hipEventRecord(e, stream);
    t.pending[kernel_id].push_back({b, e, kernels > 0 ? kernels : 1});
}

extern "C" int isc_timing_enable(int enable) {
    TimingState& t = timing();
    std::lock_guard<std::mutex> lock(t.mu);
    t.enabled = enable != 0;
    return ISC_OK;
}

extern "C" int isc_timing_read(int kernel_id, double* total_ms, int* launches) {
    if (kernel_id < 0 || kernel_id >= ISC_KERNEL_COUNT || !total_ms || !launches) return ISC_ERR_INVALID_ARG;
    TimingState& t = timing();
    std::lock_guard<std::mutex> lock(t.mu);
    double sum = 0.0;
    int n = 0;
    for (auto& pr : t.pending[kernel_id]) {
        float ms = 0.f;
        if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
            sum += ms;
            n += pr.kernels;
        }
        t.pool.push_back(pr.first);
        t.pool.push_back(pr.second);
    }
    t.pending[kernel_id].clear();
    *total_ms = sum;
    *launches = n;
    return ISC_OK;
}
