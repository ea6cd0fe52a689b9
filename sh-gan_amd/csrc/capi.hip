// C-ABI plumbing of libshgan_hip.so: version, last-error string, device sanity query.
#include "shg_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void shg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* shg_last_error(void) { return g_err; }

// bump when an exported signature changes (checked by the Python loader)
extern "C" int shg_abi_version(void) { return 36; }

// name of the GCN arch of device `dev` (e.g. "gfx950:sramecc+:xnack-") into buf; returns CU count or <0
extern "C" int shg_device_info(int dev, char* buf, int buflen) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) {
        shg_set_error("hipGetDeviceProperties(%d): %s", dev, hipGetErrorString(e));
        return SHG_ERR_LAUNCH;
    }
    if (buf && buflen > 0) {
        strncpy(buf, prop.gcnArchName, buflen - 1);
        buf[buflen - 1] = 0;
    }
    return prop.multiProcessorCount;
}
