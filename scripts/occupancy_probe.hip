// Diagnostic: how many 256-thread workgroups with X bytes of dynamic LDS really share a CU on this chip.
//   hipcc -O3 --offload-arch=gfx950 scripts/occupancy_probe.hip -o /tmp/occ && /tmp/occ
// Every workgroup spins for a fixed number of clock ticks; a grid of 256 x k workgroups takes ~1 spin if k fit per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256, 2) void spin(float* out, long long ticks) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)__builtin_amdgcn_s_memtime() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); }
    if (threadIdx.x == 0) out[blockIdx.x] = lds[5];
}
int main() {
    float* out; (void)hipMalloc(&out, 4096 * 4);
    (void)hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int kb : {16, 32, 48, 60, 64, 72, 80}) {
        int occ = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin, 256, (size_t)kb * 1024);
        printf("LDS %3d KB: API says %d blocks/CU;", kb, occ);
        for (int k : {1, 2, 3}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(spin, dim3(256 * k), dim3(256), (size_t)kb * 1024, 0, out, 100000LL);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(spin, dim3(256 * k), dim3(256), (size_t)kb * 1024, 0, out, 240000LL);     // ~100 us at 2.4 GHz
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  %d x 256 WGs: %.0f us", k, ms * 1e3);
        }
        printf("\n");
    }
    return 0;
}
