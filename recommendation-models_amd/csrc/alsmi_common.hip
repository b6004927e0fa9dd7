// Error reporting and small host utilities shared by every entry point.
#include "alsmi_common.h"

#include <stdarg.h>

namespace alsmi {

static thread_local char g_error[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

int device_cu_count() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0)
        return 256;
    return cus;
}

}  // namespace alsmi

extern "C" {

int alsmi_version(void) { return 100; }

const char *alsmi_last_error(void) { return alsmi::g_error; }

int alsmi_padded_rank(int rank) {
    if (rank < 1 || rank > 128) return 0;
    if (rank <= 16) return 16;
    if (rank <= 32) return 32;
    if (rank <= 64) return 64;
    return 128;
}

}  // extern "C"
