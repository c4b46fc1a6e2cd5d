#include "internal.h"
namespace skh {
void chain_pairs(skh_ctx*, const skh_sketch_set*, const skh_sketch_set*, const uint32_t*, const uint32_t*, uint64_t, const skh_map_params&, skh_ani_result*, skh_chain_stats*) { throw Error("chain not built yet"); }
}
