// Diagnostic (not part of the library): what the bf16 matrix pipe of THIS chip sustains on random operands.
//   hipcc -O3 --offload-arch=gfx950 scripts/clock_probe_bf16.hip -o /tmp/clock_probe_bf16 && /tmp/clock_probe_bf16
// A bare v_mfma_f32_32x32x16_bf16 loop (operands in registers, 8 independent accumulators, one or two waves per SIMD; random or
// all-zero operands) is launched back to back for ~2 s; each workgroup stamps s_memtime (shader clock) and s_memrealtime
// (100 MHz) around its loop.  Reported: wall-clock TFLOP/s and the in-kernel clock = d(memtime) / d(memrealtime) x 100 MHz
// (median over workgroups) -- MI355X_MICROARCH.md, "DVFS give-back" item 6.  Paper peak at 2.4 GHz = 2516 TFLOP/s; the chip lowers
// its clock under this load, so the sustained rate on random data is the practical roof of the reduced-precision kernels
// (bf16x3 = 3 MFMAs per product: a third of it).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void mfma_loop(const u32x4* __restrict__ in, float* __restrict__ out, unsigned long long* stamps, int iters) {
    f32x16 acc[8];
    u32x4 a[2], b[4];
    for (int i = 0; i < 2; ++i) a[i] = in[(i * 512 + threadIdx.x) % 4096];
    for (int i = 0; i < 4; ++i) b[i] = in[((2 + i) * 512 + threadIdx.x) % 4096];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i & 1]), __builtin_bit_cast(bf16x8, b[i >> 1]), acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void mfma_loop16(const u32x4* __restrict__ in, float* __restrict__ out, unsigned long long* stamps, int iters) {
    f32x4v acc[16];
    u32x4 a[4], b[4];
    for (int i = 0; i < 4; ++i) a[i] = in[(i * 512 + threadIdx.x) % 4096];
    for (int i = 0; i < 4; ++i) b[i] = in[((4 + i) * 512 + threadIdx.x) % 4096];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i)      // 64 MFMAs of half the FLOPs = the 32 of the 32x32x16 loop
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i & 3]), __builtin_bit_cast(bf16x8, b[i >> 2]), acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    const int blocks = 256;
    u32x4* in; float* out; unsigned long long* st;
    hipMalloc(&in, 4096 * 16); hipMalloc(&out, blocks * 512 * 4); hipMalloc(&st, blocks * 16);
    for (int zero : {0, 1}) {
        std::vector<unsigned> h(4096 * 4);
        for (auto& x : h) {
            if (zero) { x = 0; continue; }
            // two random bf16 in [-1, 1): sign, exponent 0x70..0x7e, 7 mantissa bits
            unsigned w = 0;
            for (int k = 0; k < 2; ++k) w |= (((rand() & 1) << 15) | ((0x70 + rand() % 15) << 7) | (rand() & 0x7f)) << (16 * k);
            x = w;
        }
        hipMemcpy(in, h.data(), 4096 * 16, hipMemcpyHostToDevice);
        for (int shape : {32, 16})
        for (int threads : {256, 512}) {
            const int iters = 40000;                                  // 32 MFMAs (32x32x16) or 64 (16x16x32) per iteration per wave
            auto launch = [&]() {
                if (shape == 32) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(threads), 0, 0, in, out, st, iters);
                else hipLaunchKernelGGL(mfma_loop16, dim3(blocks), dim3(threads), 0, 0, in, out, st, iters);
            };
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int w = 0; w < 3; ++w) launch();
            hipDeviceSynchronize();
            const int reps = 60;
            hipEventRecord(e0);
            for (int r = 0; r < reps; ++r) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)reps * blocks * (threads / 64) * iters * 32.0 * (2.0 * 32 * 32 * 16);
            std::vector<unsigned long long> hs(blocks * 2);
            hipMemcpy(hs.data(), st, blocks * 16, hipMemcpyDeviceToHost);
            std::vector<double> clk;
            for (int b = 0; b < blocks; ++b) clk.push_back((double)hs[2 * b] / (double)hs[2 * b + 1] * 100e6);
            std::sort(clk.begin(), clk.end());
            printf("{\"operands\": \"%s\", \"mfma\": \"%s\", \"waves_per_simd\": %d, \"seconds\": %.3f, \"tflops\": %.1f, \"in_kernel_clock_ghz_median\": %.3f, \"min\": %.3f, \"max\": %.3f}\n",
                   zero ? "zero" : "random", shape == 32 ? "32x32x16" : "16x16x32", threads / 256, ms * 1e-3, flops / (ms * 1e-3) / 1e12, clk[blocks / 2] / 1e9, clk.front() / 1e9, clk.back() / 1e9);
        }
    }
    return 0;
}
