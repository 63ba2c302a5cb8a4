// Diagnostic (not part of the library): what the f32 matrix pipe of THIS chip sustains.
//   hipcc -O3 --offload-arch=gfx950 scripts/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
// A bare v_mfma_f32_32x32x2_f32 loop (operands in registers, 4 independent accumulators, one or two waves per SIMD) is
// launched back to back for ~2 s; each workgroup stamps s_memtime (shader clock) and s_memrealtime (100 MHz) around its
// loop.  Reported: wall-clock TFLOP/s, and the in-kernel clock = d(memtime) / d(memrealtime) x 100 MHz (median over
// workgroups) -- MI355X_MICROARCH.md, "DVFS give-back" item 6.  Peak at 2.4 GHz = 157.3 TFLOP/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void mfma_loop(const float* __restrict__ in, float* __restrict__ out, unsigned long long* stamps, int iters) {
    f32x16 acc[4];
    const float a0 = in[threadIdx.x], b0 = in[512 + threadIdx.x];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0, b = b0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    const int blocks = 256;
    float *in, *out; unsigned long long* st;
    hipMalloc(&in, 1024 * 4); hipMalloc(&out, blocks * 512 * 4); hipMalloc(&st, blocks * 16);
    std::vector<float> h(1024);
    for (auto& x : h) x = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    for (int threads : {256, 512}) {
        const int iters = 20000;                                  // 32 MFMAs per iteration per wave
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(threads), 0, 0, in, out, st, iters);
        hipDeviceSynchronize();
        const int reps = 40;
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(threads), 0, 0, in, out, st, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)reps * blocks * (threads / 64) * iters * 32.0 * (2.0 * 32 * 32 * 2);
        std::vector<unsigned long long> hs(blocks * 2);
        hipMemcpy(hs.data(), st, blocks * 16, hipMemcpyDeviceToHost);
        std::vector<double> clk;
        for (int b = 0; b < blocks; ++b) clk.push_back((double)hs[2 * b] / (double)hs[2 * b + 1] * 100e6);
        std::sort(clk.begin(), clk.end());
        printf("{\"waves_per_simd\": %d, \"seconds\": %.3f, \"tflops\": %.1f, \"in_kernel_clock_ghz_median\": %.3f, \"min\": %.3f, \"max\": %.3f}\n",
               threads / 256, ms * 1e-3, flops / (ms * 1e-3) / 1e12, clk[blocks / 2] / 1e9, clk.front() / 1e9, clk.back() / 1e9);
    }
    return 0;
}
