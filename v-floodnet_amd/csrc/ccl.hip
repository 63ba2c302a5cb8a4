// myutils.postprocessing_pred (myutils/data.py:17-37) on the device: 8-connected component labelling of
// the water mask and "keep the largest component", so that the label map leaves the GPU already
// post-processed (the reference downloads the mask and runs cv2.connectedComponentsWithAlgorithm on the CPU).
//
// Union-find with atomicMin on a parent array (root = smallest pixel index of the component, i.e. its first
// pixel in raster order -- the order in which OpenCV / the reference's `for i in range(label_cnt)` loop meets
// the components, so size ties resolve to the same component):
//   1. init      parent[i] = start of i's horizontal run inside its 64-pixel row segment (one ballot), -1 for background
//   2. merge     runs are united with the runs of the row above that touch them (8-connectivity), once per pair
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

// One wave per 64-pixel segment of a row.  A pixel's first parent is the start of its horizontal run inside the
// segment (found with one ballot), so a run is a depth-1 tree before any atomic is issued.
__global__ void ccl_init_kernel(const unsigned char* __restrict__ pred, int* __restrict__ parent, int* __restrict__ count,
                                int H, int W, int segs) {
    const int lane = threadIdx.x & 63;
    const int seg = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (seg >= H * segs) return;                                        // (whole waves leave together)
    const int y = seg / segs, x0 = (seg - y * segs) * 64, x = x0 + lane;
    const bool in = x < W;
    const int i = y * W + x;
    const bool water = in && pred[i];
    const unsigned long long mask = __ballot(water);
    const unsigned long long below = ~mask & ((1ull << lane) - 1ull);   // non-water lanes below this one
    const int start = below ? 64 - __clzll(below) : 0;
    if (in) {
        parent[i] = water ? (y * W + x0 + start) : -1;
        count[i] = 0;
    }
}

// Unions between runs only: a run R = [xs, xe] of row y is united with every run of row y-1 that overlaps
// [xs-1, xe+1] (8-connectivity), exactly once per pair -- at the above-run's first pixel inside that window.
// (A per-pixel formulation issues ~4 atomics per water pixel; this one a handful per run.)
__global__ void ccl_merge_kernel(const unsigned char* __restrict__ pred, int* __restrict__ parent, int H, int W, int segs) {
    const int lane = threadIdx.x & 63;
    const int seg = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (seg >= H * segs) return;
    const int y = seg / segs, x0 = (seg - y * segs) * 64, x = x0 + lane;
    if (x >= W) return;
    const int i = y * W + x;
    if (!pred[i]) return;
    const bool left = x > 0 && pred[i - 1];
    const bool right = x < W - 1 && pred[i + 1];
    if (lane == 0 && left) uf_union(parent, i, i - 1);                  // the run continues from the previous segment
    if (y > 0) {
        const bool up = pred[i - W];
        const bool upl = x > 0 && pred[i - W - 1];
        const bool upr = x < W - 1 && pred[i - W + 1];
        if (up && !upl) uf_union(parent, i, i - W);                     // an above-run begins here, inside [xs, xe]
        if (!left && upl) uf_union(parent, i, i - W - 1);               // x = xs: the above-run that covers xs-1
        if (!right && upr && !up) uf_union(parent, i, i - W + 1);       // x = xe: an above-run that begins at xe+1
    }
}

// parent[i] = root(i); count[root] += 1.  A frame's water is mostly ONE component, so nearly every count lands on the same address:
// aggregated per wave that was one atomic per 64 pixels -- 14 400 adds to one counter at 720p, which the memory side serialises
// (74 us of a 2.6 ms C3 frame).  Now the 16 waves of a workgroup first pool their (root, count) pairs in LDS and one wave issues
// one atomic per DISTINCT root of the workgroup (900 adds at 720p); a wave with more than 4 distinct roots (component borders)
// sends the rest directly.
__global__ __launch_bounds__(1024)
void ccl_flatten_kernel(int* __restrict__ parent, int* __restrict__ count, int n) {
    __shared__ int s_root[16 * 4], s_cnt[16 * 4], s_n[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int nround = (n + (int)(gridDim.x * blockDim.x) - 1) / (int)(gridDim.x * blockDim.x);
    for (int k = 0; k < nround; ++k) {
        const int i = (k * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        int r = -1;
        if (i < n && parent[i] >= 0) {
            r = uf_find(parent, i);
            parent[i] = r;                               // safe: r is a root, roots never change after the merge pass
        }
        unsigned long long todo = __ballot(r >= 0);
        int slots = 0;
        while (todo) {                                   // one entry per distinct root in the wave
            const int leader = __ffsll((long long)todo) - 1;
            const int rl = __shfl(r, leader, 64);
            const unsigned long long same = __ballot(r == rl);
            if (lane == leader) {
                if (slots < 4) { s_root[wave * 4 + slots] = rl; s_cnt[wave * 4 + slots] = __popcll(same); }
                else atomicAdd(&count[rl], __popcll(same));
            }
            ++slots;                                     // (wave-uniform)
            todo &= ~same;
        }
        if (lane == 0) s_n[wave] = slots < 4 ? slots : 4;
        __syncthreads();
        if (wave == 0) {                                 // lane e = entry e of the pool; equal roots are summed by their first holder
            const int w = lane >> 2, e = lane & 3;
            const bool live = w < nw && e < s_n[w];
            const int my_r = live ? s_root[lane] : -1, my_c = live ? s_cnt[lane] : 0;
            int total = 0;
            bool first = live;
            for (int j = 0; j < 64; ++j) {
                const int rj = __shfl(my_r, j, 64), cj = __shfl(my_c, j, 64);
                if (live && rj == my_r) { total += cj; if (j < lane) first = false; }
            }
            if (first) atomicAdd(&count[my_r], total);
        }
        __syncthreads();
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
    // one pair of atomics per workgroup (a 64-bit atomicMax on one hot address is slow when every wave issues it)
    __shared__ unsigned long long s_key[16];
    __shared__ int s_n[16];
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) { s_key[wave] = key; s_n[wave] = ncomp; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; ++w) { key = s_key[w] > key ? s_key[w] : key; ncomp += s_n[w]; }
        if (key) atomicMax(reinterpret_cast<unsigned long long*>(result + 4), key);
        if (ncomp) atomicAdd(&result[2], ncomp);
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
    const int segs = cdiv(W, 64);                                  // 64-pixel row segments, one wave each
    const int rblocks = cdiv(H * segs, 4);                         // 4 waves per workgroup
    hipLaunchKernelGGL(ccl_init_kernel, dim3(rblocks), dim3(256), 0, s, pred, parent, count, H, W, segs);
    hipLaunchKernelGGL(ccl_merge_kernel, dim3(rblocks), dim3(256), 0, s, pred, parent, H, W, segs);
    hipLaunchKernelGGL(ccl_flatten_kernel, dim3(cdiv(n, 1024) < 1024 ? cdiv(n, 1024) : 1024), dim3(1024), 0, s, parent, count, n);
    hipLaunchKernelGGL(ccl_pick_kernel, dim3(blocks < 128 ? blocks : 128), dim3(1024), 0, s, parent, count, result, n);
    hipLaunchKernelGGL(ccl_write_kernel, dim3(blocks), dim3(256), 0, s, pred, parent, result, out, n);
    return vfn_check_launch();
}
