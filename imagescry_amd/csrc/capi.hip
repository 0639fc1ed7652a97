// Library-level entry points of include/imagescry_hip.h.
#include <string.h>

#include "isc_common.h"

extern "C" int isc_abi_version(void) { return ISC_ABI_VERSION; }

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
