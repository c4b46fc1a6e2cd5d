// TEST-ONLY (tests/host_shims/build_shims.py build_cli_emu): the kernel simulator build has no RCCL transport; these two entry points fail, which sends
// `skani-hip triangle --gpus N` down its host-collective path.
#include "../../include/skani_hip.h"
extern "C" int skh_comm_unique_id(uint8_t*) { return SKH_ERR_DEVICE; }
extern "C" int skh_comm_create_rccl(skh_ctx*, const uint8_t*, int, int, skh_comm** out) { if (out) *out = nullptr; return SKH_ERR_DEVICE; }
