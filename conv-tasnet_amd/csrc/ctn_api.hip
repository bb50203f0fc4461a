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

// Everything enqueued on `from` so far completes before anything enqueued on `to` after this call.
// The event carries no timing and no SYSTEM-scope fence: both streams belong to this device, kernel boundaries already
// order memory at device scope, and the system-scope cache writeback + invalidate of a default event is what makes a
// cross-stream dependency expensive for the kernels that follow it (hip_runtime_api.h, hipEventDisableSystemFence).
int ctn_stream_order(void* from, void* to) {
    static thread_local hipEvent_t ev = nullptr;
    static thread_local int ev_dev = -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { ctn_set_error("ctn_stream_order: no current device"); return CTN_ERR_LAUNCH; }
    if (ev == nullptr || ev_dev != dev) {
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) {
            ev = nullptr;
            ctn_set_error("ctn_stream_order: cannot create an event");
            return CTN_ERR_LAUNCH;
        }
        ev_dev = dev;
    }
    if (hipEventRecord(ev, (hipStream_t)from) != hipSuccess || hipStreamWaitEvent((hipStream_t)to, ev, 0) != hipSuccess) {
        ctn_set_error("ctn_stream_order: %s", hipGetErrorString(hipGetLastError()));
        return CTN_ERR_LAUNCH;
    }
    return CTN_OK;
}

}  // extern "C"
