// placeholder: replaced by the stage A implementation
#include "ctx.h"
void tod_orb_ws_free(todhip_ctx*) {}
extern "C" int todhip_orb(todhip_ctx*, const uint8_t*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, float,
                          const int8_t*, float*, float*, uint8_t*, uint32_t*) { return TODHIP_EINVAL; }
