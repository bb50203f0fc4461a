// Library-level entry points of the C ABI: version, last-error string.
#include "ctn_common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_ctn_error[512] = "";

void ctn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_ctn_error, sizeof(g_ctn_error), fmt, ap);
    va_end(ap);
}

extern "C" {

int ctn_version(void) { return 100; }  // 0.1.0

// Message of the last failing call on this thread ("" if none).  Never NULL.
const char* ctn_last_error(void) { return g_ctn_error; }

// Frames after padding: activations are stored [M, Ch, Kp] with Kp a multiple of 64.
int ctn_padded_frames(int K) { return (K + 63) / 64 * 64; }

}  // extern "C"
