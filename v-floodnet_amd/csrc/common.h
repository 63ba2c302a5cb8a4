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
