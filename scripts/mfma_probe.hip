// Diagnostic: f32 MFMA issue rate under the operand / accumulator patterns the kernels use.
//   hipcc -O3 --offload-arch=gfx950 scripts/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// V: 0 = 4 accumulators, constant operands | 1 = 2 accumulators alternating, 16 different operand registers
//    2 = as 1, operands re-read from LDS (ds_read_b128) every 8 MFMAs, one k-group ahead | 3 = one accumulator (dependent chain)
//    4 = as 2 but reads issued right before use (the schedule hipcc produces)
template <int V>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ in, float* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 128];
    for (int i = threadIdx.x; i < 64 * 128; i += blockDim.x) lds[i] = in[i & 1023];
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = *(const f32x4*)(in + threadIdx.x % 64 * 4 + 16 * i); b[i] = *(const f32x4*)(in + 512 + threadIdx.x % 64 * 4 + 16 * i); }
    const int lane = threadIdx.x & 63;
    const float* base = lds + (lane & 31) * 128 + (lane >> 5) * 4;
    for (int it = 0; it < iters; ++it) {
        if constexpr (V == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][0], b[0][0], acc[i], 0, 0, 0);
        } else if constexpr (V == 1) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][t], b[g][t], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(g + 1) & 3][t], b[g][t], acc[1], 0, 0, 0);
                }
        } else if constexpr (V == 3) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][t], b[g][t], acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(g + 1) & 3][t], b[g][t], acc[0], 0, 0, 0);
                }
        } else {
            f32x4 fa[2][2];
            fa[0][0] = *(const f32x4*)(base); fa[0][1] = *(const f32x4*)(base + 32 * 128);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cur = g & 1;
                if (V == 2) {
                    fa[cur ^ 1][0] = *(const f32x4*)(base + 8 * (g + 1)); fa[cur ^ 1][1] = *(const f32x4*)(base + 32 * 128 + 8 * (g + 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][0][t], b[g][t], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][1][t], b[g][t], acc[1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (V == 4) {
                    fa[cur ^ 1][0] = *(const f32x4*)(base + 8 * (g + 1)); fa[cur ^ 1][1] = *(const f32x4*)(base + 32 * 128 + 8 * (g + 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V> void run(const float* in, float* out, int threads, const char* name) {
    const int blocks = 256, iters = 20000, reps = 10;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)reps * blocks * (threads / 64) * iters * 32.0 * 4096.0;
    printf("%-58s waves/SIMD %d: %6.1f TFLOP/s\n", name, threads / 256, flops / (ms * 1e-3) / 1e12);
}

int main() {
    float *in, *out;
    (void)hipMalloc(&in, 1024 * 4); (void)hipMalloc(&out, 256 * 512 * 4);
    std::vector<float> h(1024);
    for (auto& x : h) x = (float)rand() / RAND_MAX - 0.5f;
    (void)hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    for (int threads : {256, 512}) {
        run<0>(in, out, threads, "4 accumulators, constant operands");
        run<1>(in, out, threads, "2 accumulators alternating, varying operand registers");
        run<3>(in, out, threads, "1 accumulator (dependent chain)");
        run<2>(in, out, threads, "2 acc, A from LDS one k-group ahead (pinned)");
        run<4>(in, out, threads, "2 acc, A from LDS read right before use");
    }
    return 0;
}
