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
