#include "mgv_common.h"
#include "../../include/mgvae_hip.h"
extern "C" int mgv_abi_version(void) { return 1; }
