// myutils.postprocessing_pred (myutils/data.py:17-37) on the device: 8-connected component labelling of
// the water mask and "keep the largest component", so that the label map leaves the GPU already
// post-processed (the reference downloads the mask and runs cv2.connectedComponentsWithAlgorithm on the CPU).
//
// Union-find with atomicMin on a parent array (root = smallest pixel index of the component, i.e. its first
// pixel in raster order -- the order in which OpenCV / the reference's `for i in range(label_cnt)` loop meets
// the components, so size ties resolve to the same component):
//   1. init      parent[i] = i for water pixels, -1 for background
//   2. merge     every water pixel unions itself with its W, NW, N, NE water neighbours
//   3. flatten   parent[i] = root(i); count[root] += 1
//   4. pick      arg-max of count (ties: smallest root) ; number of components
//   5. write     out = (root == best), with the reference's special cases:
//                no component -> all ones (labels == 0 everywhere); exactly one component -> labels, or
//                1 - labels when labels[0,0] != pred[0,0] (only possible for pred values > 1)
#include "common.h"
#include "../../include/vfn_hip.h"

namespace {

// parent[] is rewritten by atomics from every CU during the merge: read it past the (never refreshed) L1
__device__ __forceinline__ int uf_load(const int* parent, int x) {
    return __hip_atomic_load(parent + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int uf_find(const int* parent, int x) {
    int p = uf_load(parent, x);
    while (p != x) { x = p; p = uf_load(parent, x); }
    return x;
}

__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
    while (true) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&parent[b], a);         // hang the larger root under the smaller one
        if (old == b) return;
        b = old;                                          // somebody re-parented b meanwhile: retry from there
    }
}

__global__ void ccl_init_kernel(const unsigned char* __restrict__ pred, int* __restrict__ parent, int* __restrict__ count, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        parent[i] = pred[i] ? i : -1;
        count[i] = 0;
    }
}

__global__ void ccl_merge_kernel(const unsigned char* __restrict__ pred, int* __restrict__ parent, int H, int W) {
    const int n = H * W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!pred[i]) continue;
        const int y = i / W, x = i - y * W;
        if (x > 0 && pred[i - 1]) uf_union(parent, i, i - 1);
        if (y > 0) {
            if (pred[i - W]) {
                uf_union(parent, i, i - W);          // N is water: NW / NE, if water, hang on N through its own W links
            } else {
                if (x > 0 && pred[i - W - 1]) uf_union(parent, i, i - W - 1);
                if (x < W - 1 && pred[i - W + 1]) uf_union(parent, i, i - W + 1);
            }
        }
    }
}

__global__ void ccl_flatten_kernel(int* __restrict__ parent, int* __restrict__ count, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (parent[i] < 0) continue;
        const int r = uf_find(parent, i);
        parent[i] = r;                                   // safe: r is a root, roots never change after the merge pass
        atomicAdd(&count[r], 1);
    }
}

// result[0] = best count, result[1] = best root (smallest among ties), result[2] = number of components
__global__ void ccl_pick_kernel(const int* __restrict__ parent, const int* __restrict__ count, int* __restrict__ result, int n) {
    int best_c = 0, best_r = 0x7fffffff, ncomp = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (parent[i] == i) {
            ++ncomp;
            const int c = count[i];
            if (c > best_c || (c == best_c && i < best_r)) { best_c = c; best_r = i; }
        }
    }
    // pack (count, -root) so that one 64-bit max gives "largest count, smallest root"
    unsigned long long key = ((unsigned long long)(unsigned)best_c << 32) | (unsigned)(0x7fffffff - best_r);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o, 64);
        key = other > key ? other : key;
        ncomp += __shfl_xor(ncomp, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(reinterpret_cast<unsigned long long*>(result + 4), key);
        atomicAdd(&result[2], ncomp);
    }
}

__global__ void ccl_write_kernel(const unsigned char* __restrict__ pred, const int* __restrict__ parent,
                                 const int* __restrict__ result, unsigned char* __restrict__ out, int n) {
    const unsigned long long key = *reinterpret_cast<const unsigned long long*>(result + 4);
    const int best_r = 0x7fffffff - (int)(unsigned)(key & 0xffffffffull);
    const int ncomp = result[2];
    const bool flip = (ncomp == 1) && ((pred[0] != 0 ? 1 : 0) != (int)pred[0]);      // labels[0,0] != pred[0,0]
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        unsigned char v;
        if (ncomp == 0) v = 1;                                    // labels == 0 everywhere -> all ones
        else if (ncomp == 1) { const int lab = pred[i] != 0; v = (unsigned char)(flip ? 1 - lab : lab); }
        else v = (unsigned char)(parent[i] == best_r);
        out[i] = v;
    }
}

}  // namespace

extern "C" int vfn_postprocess_pred_device_u8(const unsigned char* pred, unsigned char* out, int* scratch, int H, int W,
                                              void* stream) {
    if (!pred || !out || !scratch || H < 1 || W < 1) return VFN_ERR_ARG;
    const int n = H * W;
    int* parent = scratch;                // [n]
    int* count = scratch + n;             // [n]
    int* result = scratch + 2 * (size_t)n;   // [8]: 0 unused, 2 = components, 4..5 = packed (count, root) key
    hipStream_t s = (hipStream_t)stream;
    const int blocks = cdiv(n, 256) < 2048 ? cdiv(n, 256) : 2048;
    hipMemsetAsync(result, 0, 8 * sizeof(int), s);
    hipLaunchKernelGGL(ccl_init_kernel, dim3(blocks), dim3(256), 0, s, pred, parent, count, n);
    hipLaunchKernelGGL(ccl_merge_kernel, dim3(blocks), dim3(256), 0, s, pred, parent, H, W);
    hipLaunchKernelGGL(ccl_flatten_kernel, dim3(blocks), dim3(256), 0, s, parent, count, n);
    hipLaunchKernelGGL(ccl_pick_kernel, dim3(blocks < 256 ? blocks : 256), dim3(256), 0, s, parent, count, result, n);
    hipLaunchKernelGGL(ccl_write_kernel, dim3(blocks), dim3(256), 0, s, pred, parent, result, out, n);
    return vfn_check_launch();
}
