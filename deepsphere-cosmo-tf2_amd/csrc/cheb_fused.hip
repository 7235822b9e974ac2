// Single-launch fused forward (tile + halo resident in LDS).  Placeholder until the tile
// engine lands: every plan reports "not tileable" and dsph_cheb_forward takes the unfused path.
#include "dsphere_common.h"

namespace dsph {

struct FusedPlan {};

FusedPlan* fused_plan_build(const dsph_plan*, const int32_t*, const float*) { return nullptr; }
void fused_plan_destroy(FusedPlan* fp) { delete fp; }
bool fused_supported(const dsph_plan*, int32_t, int32_t, int32_t) { return false; }
size_t fused_workspace_bytes(const dsph_plan*, int64_t, int32_t, int32_t, int32_t, int32_t) { return 0; }
int launch_cheb_fused(const dsph_plan*, const float*, const float*, const float*, float*, int64_t,
                      int32_t, int32_t, int32_t, int32_t, int32_t, void*, size_t, hipStream_t) {
  set_error("fused kernel not available");
  return DSPH_E_UNSUPPORTED;
}

}  // namespace dsph
