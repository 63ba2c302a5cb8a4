#include "common.h"
#include "../../include/vfn_hip.h"
// 2: vfn_conv_desc.w_packed, vfn_bankscan_desc.precision, vfn_memread_desc.precision; bf16 / bf16x3 and I/O entry points
extern "C" int vfn_abi_version(void) { return 2; }
