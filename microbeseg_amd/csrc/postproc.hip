// postproc.hip — placeholder until the HIP post-processing lands (returns MSEG_EINVAL; nothing routes through it yet).
#include "common.h"
extern "C" size_t mseg_postproc_workspace_bytes(int H, int W) { (void)H; (void)W; return 0; }
extern "C" int mseg_distance_postprocess(const float*, const float*, int, int, float, float, int, uint16_t*, int32_t*,
                                         int32_t*, void*, size_t, void*) { return MSEG_EINVAL; }
extern "C" int mseg_boundary_postprocess(const float*, int, int, uint16_t*, int32_t*, int32_t*, void*, size_t, void*) {
  return MSEG_EINVAL;
}
