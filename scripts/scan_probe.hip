// Diagnostic: the bank-scan loop rebuilt piece by piece (MFMA + LDS fragment reads | + barrier | + softmax VALU | + LDS-DMA)
// to find which ingredient caps a CU at ~70 % of the f32 matrix rate.  hipcc -O3 --offload-arch=gfx950 scripts/scan_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int DK = 128, CH = 64;
__device__ __forceinline__ int swz(int row, int chunk) { return row * DK + ((chunk ^ (row & 15)) << 2); }

template <int LEVEL>   // 0: mfma+lds reads, 1: +barrier, 2: +softmax, 3: +lds-dma of the next chunk
__global__ __launch_bounds__(256, 2) void scan_like(const float* __restrict__ keys, const float* __restrict__ q, float* __restrict__ out, int chunks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sKb = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 2 * CH * DK; i += 256) sKb[i] = keys[i];
    f32x4 qf[16];
    for (int kk = 0; kk < 16; ++kk) qf[kk] = *(const f32x4*)(q + (size_t)(wave * 32 + li) * DK + 8 * kk + 4 * lh);
    __syncthreads();
    float run_m = -INFINITY, run_l = 0.f;
    const float* K = keys + (size_t)blockIdx.x * 4096;
    for (int c = 0; c < chunks; ++c) {
        const float* sK = sKb + (c & 1) * CH * DK;
        if (LEVEL >= 3) {
            float* dst = sKb + ((c + 1) & 1) * CH * DK;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = (wave * 8 + j) * 2 + (lane >> 5), pc = lane & 31;
                const float* g = K + (size_t)((c * CH + r) & 4095) * DK + ((pc ^ (r & 15)) << 2);
                __builtin_amdgcn_global_load_lds(g, dst + (wave * 8 + j) * 2 * DK, 16, 0, 0);
            }
        }
        f32x16 acc[2];
        for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        f32x4 a[2][2];
        a[0][0] = *(const f32x4*)(sK + swz(li, lh)); a[0][1] = *(const f32x4*)(sK + swz(32 + li, lh));
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int cur = kk & 1;
            if (kk + 1 < 16) { a[cur ^ 1][0] = *(const f32x4*)(sK + swz(li, 2 * (kk + 1) + lh)); a[cur ^ 1][1] = *(const f32x4*)(sK + swz(32 + li, 2 * (kk + 1) + lh)); }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][0][t], qf[kk][t], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][1][t], qf[kk][t], acc[1], 0, 0, 0);
            }
        }
        if (LEVEL >= 2) {
            float mx = acc[0][0];
            for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[i][r]);
            const float mn = fmaxf(run_m, mx * 0.088f);
            float sum = 0.f;
            for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) sum += __expf(acc[i][r] * 0.088f - mn);
            run_l = run_l * expf(run_m - mn) + sum; run_m = mn;
        } else { run_m = fmaxf(run_m, acc[0][3] + acc[1][7]); }
        if (LEVEL >= 1) __syncthreads();
    }
    out[blockIdx.x * 256 + tid] = run_m + run_l;
}

template <int LEVEL> void run(const float* keys, const float* q, float* out, int blocks, const char* name) {
    const int chunks = 400, reps = 5;
    (void)hipFuncSetAttribute((const void*)scan_like<LEVEL>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(scan_like<LEVEL>, dim3(blocks), dim3(256), 65536, 0, keys, q, out, chunks);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(scan_like<LEVEL>, dim3(blocks), dim3(256), 65536, 0, keys, q, out, chunks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)reps * blocks * 4 * chunks * 128.0 * 4096.0;
    printf("%-44s %d WG/CU: %6.1f TFLOP/s  (%.2f us per chunk per CU)\n", name, blocks / 256, flops / (ms * 1e-3) / 1e12, ms * 1e3 / reps / chunks / (blocks / 256));
}

int main() {
    float *keys, *q, *out;
    const size_t nk = (size_t)(512 * 4096 + 8192) * DK;
    (void)hipMalloc(&keys, nk * 4); (void)hipMalloc(&q, 128 * DK * 4); (void)hipMalloc(&out, 512 * 256 * 4);
    std::vector<float> h(nk);
    for (auto& x : h) x = (float)rand() / RAND_MAX - 0.5f;
    (void)hipMemcpy(keys, h.data(), nk * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(q, h.data(), 128 * DK * 4, hipMemcpyHostToDevice);
    for (int blocks : {256, 512}) {
        run<0>(keys, q, out, blocks, "MFMA + LDS fragment reads");
        run<1>(keys, q, out, blocks, "+ one barrier per chunk");
        run<2>(keys, q, out, blocks, "+ softmax statistics (VALU)");
        run<3>(keys, q, out, blocks, "+ LDS-DMA of the next key chunk");
    }
    return 0;
}
