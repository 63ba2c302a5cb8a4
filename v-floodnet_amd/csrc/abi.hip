#include "common.h"
#include "../../include/vfn_hip.h"
// 2: vfn_conv_desc.w_packed, vfn_bankscan_desc.precision, vfn_memread_desc.precision; bf16 / bf16x3 and I/O entry points
// 3: vfn_memread_desc.wide (added at 2 without a bump), vfn_conv_cfg_info, vfn_sizeof_desc
// 4: PNG / JPEG / segment-uncertainty / norm-refresh entry points (round 2)
extern "C" int vfn_abi_version(void) { return VFN_ABI_VERSION; }

// sizeof of every descriptor as THIS library was compiled: a binding whose struct layout drifted fails its
// load-time check (v-floodnet_amd/_lib.py, tests/test_abi.py) instead of reading past the caller's struct
extern "C" int vfn_sizeof_desc(int which) {
    switch (which) {
        case VFN_DESC_CONV: return (int)sizeof(vfn_conv_desc);
        case VFN_DESC_STEM: return (int)sizeof(vfn_stem_desc);
        case VFN_DESC_BANKSCAN: return (int)sizeof(vfn_bankscan_desc);
        case VFN_DESC_MEMREAD: return (int)sizeof(vfn_memread_desc);
        case VFN_DESC_BANK: return (int)sizeof(vfn_bank_desc);
        case VFN_DESC_WGRAD: return (int)sizeof(vfn_wgrad_desc);
        case VFN_DESC_REFRESH_FILTER: return (int)sizeof(vfn_refresh_filter);
        case VFN_DESC_REFRESH_EPILOGUE: return (int)sizeof(vfn_refresh_epilogue);
        case VFN_DESC_GATHER: return (int)sizeof(vfn_gather_entry);
    }
    return -1;
}
