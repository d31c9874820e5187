// TEST INFRASTRUCTURE ONLY -- ubench.hip is inline gfx950 assembly; the interpreter build stubs it.
#include "ctx.h"
namespace og {
int ubench(og_ctx*, int, int, int, float* ms) { *ms = 0.f; set_error("og_ubench: not available in the hipemu interpreter"); return OG_ERR_INVALID; }
}
