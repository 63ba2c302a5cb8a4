#include "common.h"
#include "../../include/vfn_hip.h"
extern "C" int vfn_abi_version(void) { return 1; }
