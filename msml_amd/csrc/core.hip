// Error reporting + version for libmsml_hip.so.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void msml_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int msml_version(void) { return MSML_ABI_VERSION; }
extern "C" const char* msml_last_error(void) { return g_err; }

// 1 when the library was built with -DMSML_EXPERIMENTS (tools/build_variant.py --all MSML_EXPERIMENTS): the measured-slower
// kernel variants of DESIGN section 8 (half-stage weight ring, BatchNorm backward in the backward-data prologue, BatchNorm in
// the weights-stationary kernel's prologue, stride-2 strips of the weight-gradient kernel) are instantiated and their
// opt-in switches live.  The shipped library returns 0 and refuses those paths.
extern "C" int msml_has_experiments(void) {
#ifdef MSML_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}

extern "C" long msml_stream_capture_id(void* stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  if (hipStreamGetCaptureInfo((hipStream_t)stream, &st, &id) != hipSuccess) {
    (void)hipGetLastError();
    msml_set_error("hipStreamGetCaptureInfo failed");
    return -1;
  }
  if (st != hipStreamCaptureStatusActive) return 0;
  return (long)(id & 0x3fffffffffffffffULL) + 1;      // ids start at 0 on some runtimes: keep 0 for "not capturing"
}

// `waiter` waits for everything queued on `src` so far (torch's Stream.wait_stream without the Python-side event object:
// one hipEvent per calling thread, re-recorded each time -- a wait holds on to the record it saw when it was queued, so
// the next record does not disturb it).  The weight-gradient side stream is forked this way ~125 times per training step.
extern "C" int msml_stream_wait_stream(void* waiter, void* src) {
  static thread_local hipEvent_t ev = nullptr;
  if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
    ev = nullptr;
    (void)hipGetLastError();
    msml_set_error("msml_stream_wait_stream: hipEventCreateWithFlags failed");
    return MSML_ERR_LAUNCH;
  }
  hipError_t e = hipEventRecord(ev, (hipStream_t)src);
  if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)waiter, ev, 0);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    msml_set_error("msml_stream_wait_stream: %s", hipGetErrorString(e));
    return MSML_ERR_LAUNCH;
  }
  return 0;
}
