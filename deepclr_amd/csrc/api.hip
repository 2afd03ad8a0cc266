// Library identity, error strings, host-mapped memory.
#include "common.h"

extern "C" int dclr_version(void) { return 1000 * 0 + 2; }

extern "C" const char *dclr_error_string(int code) {
    if (code == DCLR_OK) return "ok";
    if (code == DCLR_E_INVALID) return "invalid argument (null pointer, non-positive size or violated size relation)";
    if (code == DCLR_E_UNSUPPORTED) return "configuration not supported by the gfx950 kernels";
    if (code <= -1000) return hipGetErrorString((hipError_t)(-code - 1000));
    return "unknown error";
}

extern "C" int dclr_host_device_pointer(void *host, void **device) {
    DCLR_REQUIRE(host && device);
    const hipError_t e = hipHostGetDevicePointer(device, host, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return -1000 - (int)e;
    }
    return DCLR_OK;
}
