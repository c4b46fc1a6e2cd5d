#include "internal.h"
namespace skh {
void screen_pairs(skh_ctx*, const skh_sketch_set*, const skh_sketch_set*, double, int, int, std::vector<uint32_t>&, std::vector<uint32_t>&) { throw Error("screen not built yet"); }
}
