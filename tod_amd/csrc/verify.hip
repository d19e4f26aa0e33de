// placeholder: replaced by the stage C implementation
#include "ctx.h"
void tod_verify_ws_free(todhip_ctx*) {}
extern "C" {
void todhip_rng_seed(todhip_rng* r, uint32_t) { (void)r; }
int todhip_verify(todhip_ctx*, const float*, uint32_t, const float*, uint32_t, uint32_t, const uint32_t*,
                  const todhip_dmatch*, const float*, const float*, uint32_t, const todhip_verify_params*, todhip_rng*,
                  todhip_pose*, uint32_t*, uint32_t*, uint32_t*) { return TODHIP_EINVAL; }
}
