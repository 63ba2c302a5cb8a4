// Internal seams between the convolution translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/vfn_hip.h"

// tile configurations VFN_DIRECT_CFG0 .. VFN_DIRECT_CFG0 + VFN_DIRECT_CFGS - 1 of vfn_conv2d_nhwc_f32 are the
// wave-autonomous kernels of conv_direct.hip (f32 only)
#define VFN_DIRECT_CFG0 38
#define VFN_DIRECT_CFGS 24
// stream-K configurations (conv_streamk_kernel): workspace and counter sizes the caller provides (vfn_conv_desc.partial /
// .tile_counters; include/vfn_hip.h VFN_CONV_SK_*)
#define VFN_SK_WS_FLOATS VFN_CONV_SK_WS_FLOATS
#define VFN_SK_MAX_TILES VFN_CONV_SK_MAX_TILES

int vfn_conv_direct_info(int idx, int* bm, int* bn, int* wk);
int vfn_conv_direct_launch(const vfn_conv_desc& d, int idx, hipStream_t s);
int vfn_conv_direct_is_streamk(int idx);
int vfn_conv_direct_name(int idx, char* buf, int n);
// out = act((sum over the K slices' partial slabs, slice order) * scale + shift + res) for rows >= m_start (conv_igemm.hip)
void vfn_conv_splitk_reduce(const vfn_conv_desc& p, int m_start, hipStream_t s);
