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
