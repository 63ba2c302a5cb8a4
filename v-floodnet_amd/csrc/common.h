// Shared helpers for the gfx950 kernels.  CDNA4 only: 64-wide wavefronts, f32 MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VFN_OK 0
#define VFN_ERR_ARG 1
#define VFN_ERR_LAUNCH 2

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int vfn_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VFN_OK : VFN_ERR_LAUNCH;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// max(v, floor) for the branch-free ReLU-or-nothing epilogues (floor = 0 or -inf) that KEEPS a NaN: fmaxf returns the non-NaN
// operand, so under floor = -inf a NaN would leave as -inf and the next layer's ReLU would turn it into 0 -- a diverging
// training run (the reference's loss.backward() goes NaN) would be kept alive silently.  v_cmp + v_cndmask, no branch.
__device__ __forceinline__ float vfn_floor_nan(float v, float floor_) { return v < floor_ ? floor_ : v; }

// Workgroups are dealt round-robin over the 8 XCDs (a private 4 MiB L2 each).  A kernel whose neighbouring work items read
// overlapping bytes (window / halo reads: Winograd 6x6 patches, 3x3 pooling windows, bilinear taps) wants NEIGHBOURS ON ONE
// XCD, or every overlap is fetched once per L2 (rocprofv3 FETCH_SIZE of winograd_input_kernel: 1.7x its input before this).
// Maps the hardware block index to a logical one such that XCD x owns one contiguous run of logical blocks; bijective on
// [0, nb) for any nb; use it in place of blockIdx.x (grid-stride loops: every pass of gridDim.x blocks is mapped the same way).
__device__ __forceinline__ unsigned vfn_xcd_block(unsigned b, unsigned nb) {
    if (nb < 16) return b;
    const unsigned q = nb >> 3, r = nb & 7, xcd = b & 7, loc = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// ---- the activations' split-bf16 image (vfn_conv_desc.in_lp / out_lp, include/vfn_hip.h)
typedef __attribute__((ext_vector_type(4))) __bf16 vfn_bf16x4;
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi): 16 significant bits in two bf16 (the "bf16x3" operands)
__device__ __forceinline__ void vfn_split_bf16(const f32x4& v, vfn_bf16x4& h, vfn_bf16x4& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = (__bf16)v[e];
        l[e] = (__bf16)(v[e] - (float)h[e]);
    }
}
// image of 4 consecutive channels (col .. col+3) of pixel `row`: hi at byte (col % 32) * 2 of the pixel's 128-byte block
// col / 32, lo 64 bytes behind it; pixel stride ld_floats * 4 bytes (the twin of an f32 tensor with that stride)
__device__ __forceinline__ void vfn_store_lp4(void* base, size_t row, int ld_floats, int col, f32x4 v, bool relu) {
    if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    vfn_bf16x4 h, l;
    vfn_split_bf16(v, h, l);
    char* dst = reinterpret_cast<char*>(base) + (row * (size_t)ld_floats) * 4 + (col >> 5) * 128 + (col & 31) * 2;
    *reinterpret_cast<vfn_bf16x4*>(dst) = h;
    *reinterpret_cast<vfn_bf16x4*>(dst + 64) = l;
}
