// api_misc.hip — version / error plumbing of the C ABI.
#include "common.h"
#include <string.h>

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

// ---- dispatch bookkeeping (common.h: MSEG_KL) ---------------------------------------------------------------------------
static thread_local int g_dry = 0;
static thread_local MsegKernelInfo g_call;            // the call being dispatched
static thread_local char g_last_kernel[sizeof(g_call.name)] = "";

extern "C" int mseg_dispatch_dry(void) { return g_dry; }
extern "C" void mseg_dispatch_begin(int dry) {
  g_dry = dry;
  memset(&g_call, 0, sizeof(g_call));
}
// copies what the dispatch recorded; ends a query
extern "C" void mseg_dispatch_end(MsegKernelInfo* info) {
  if (!g_dry && g_call.name[0]) memcpy(g_last_kernel, g_call.name, sizeof(g_last_kernel));
  if (info) *info = g_call;
  g_dry = 0;
}
extern "C" void mseg_dispatch_note(int precision, size_t workspace) {
  g_call.precision = precision;
  g_call.workspace = workspace;
}
extern "C" void mseg_dispatch_note_stats(int rows) { g_call.stats_rows = rows; }
extern "C" void mseg_note_launch(const char* kernel, unsigned grid, unsigned block, int aux) {
  g_call.launches += 1;
  if (aux || g_call.name[0]) return;                  // the first non-helper kernel names the call
  // "(igemm_halo_kernel<128, 1>)" -> "igemm_halo_kernel<128, 1>"
  size_t n = strlen(kernel);
  if (n >= 2 && kernel[0] == '(' && kernel[n - 1] == ')') { ++kernel; n -= 2; }
  if (n >= sizeof(g_call.name)) n = sizeof(g_call.name) - 1;
  memcpy(g_call.name, kernel, n);
  g_call.name[n] = 0;
  g_call.grid = grid;
  g_call.block = block;
}
extern "C" const char* mseg_last_kernel(void) { return g_last_kernel; }
