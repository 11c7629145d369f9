// api_misc.hip — version / error plumbing of the C ABI.
#include "common.h"

static thread_local int g_last_hip_error = 0;

extern "C" void mseg_set_hip_error(int e) { g_last_hip_error = e; }
extern "C" int mseg_last_hip_error(void) { return g_last_hip_error; }
extern "C" int mseg_version(void) { return 100; }
extern "C" const char* mseg_strerror(int code) {
  switch (code) {
    case MSEG_OK: return "ok";
    case MSEG_EINVAL: return "invalid argument or unsupported shape";
    case MSEG_ELAUNCH: return "HIP launch/runtime error (see mseg_last_hip_error)";
    case MSEG_EWORKSPACE: return "workspace too small";
    default: return "unknown error";
  }
}
