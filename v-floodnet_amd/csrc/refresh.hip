// Derived-parameter refresh (round 4): after an optimizer step every convolution's packed filters (forward layout, the
// flipped / transposed layout of its data-gradient convolution, the stems' padded-tap layout, the 20-row tap form of the two-filter
// heads) and every folded-BatchNorm epilogue (train_video_seg.py:103-109: the statistics are frozen, weight and bias train) have to
// follow the parameters.  Rebuilding them with tensor operators costs ~1 500 tiny launches per training step (six per layer and
// layout); here they are TWO launches over tables in device memory -- one entry per (parameter tensor, derived tensor) -- that the
// host builds once per model (v-floodnet_amd/refresh.py) and re-uses while the tensors stay where they are.
//
// Pure data movement is bit-identical to the tensor-operator form; the folded scale gamma / sqrt(var + eps) is rounded correctly here
// (the device's tensor-operator division is 1-2 ulp off), so a new engine settles its constants through these kernels too and a
// refreshed engine equals a rebuilt one bit for bit (tests/test_round4_gpu.py).
#include "common.h"
#include "../../include/vfn_hip.h"

namespace {

constexpr int ELEMS_PER_BLOCK = 4096;

__device__ __forceinline__ float bn_scale(const float* gamma, const float* var, float eps, int c) {
    return __fdiv_rn(gamma[c], __fsqrt_rn(__fadd_rn(var[c], eps)));
}

__global__ __launch_bounds__(256)
void refresh_filters_kernel(const vfn_refresh_filter* __restrict__ table, int n) {
    // the entry this block works on: the last one whose first block is <= blockIdx.x
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const vfn_refresh_filter e = table[lo];
    const int T = e.kh * e.kw;
    const long long base = (long long)((int)blockIdx.x - e.block0) * ELEMS_PER_BLOCK;
    if (e.kind >= 4) {                           // Winograd F(4x4, 3x3) filter banks U = G g G^T, one (filter, channel) pair per thread
        const long long pairs = (long long)e.cout * e.cin;
        const double G[6][3] = {{0.25, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
        for (int i = threadIdx.x; i < ELEMS_PER_BLOCK; i += 256) {
            const long long idx = base + i;
            if (idx >= pairs) return;
            int co, ci;
            if (e.kind == 4) { ci = (int)(idx % e.cin); co = (int)(idx / e.cin); }
            else { co = (int)(idx % e.cout); ci = (int)(idx / e.cout); }
            const float* g = e.src + ((size_t)co * e.cin_total + e.cin_off + ci) * 9;
            const float sc = e.gamma ? bn_scale(e.gamma, e.var, e.eps, co) : 1.f;
            double w[3][3];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float v = g[e.kind == 4 ? t : 8 - t];        // (the data-gradient convolution runs over the flipped filter)
                if (e.gamma) v = __fmul_rn(v, sc);
                w[t / 3][t % 3] = (double)v;
            }
            double tmp[6][3];
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) tmp[a][b] = G[a][0] * w[0][b] + G[a][1] * w[1][b] + G[a][2] * w[2][b];
            const size_t row = e.kind == 4 ? (size_t)(e.dst_row0 + co) : (size_t)(e.dst_row0 + ci);
            const size_t col = e.kind == 4 ? (size_t)ci : (size_t)(e.dst_col0 + co);
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    const double u = tmp[a][0] * G[b][0] + tmp[a][1] * G[b][1] + tmp[a][2] * G[b][2];
                    e.dst[((size_t)(a * 6 + b) * e.cout_ld + row) * e.dst_ld + col] = (float)u;
                }
        }
        return;
    }
    const long long total = (long long)e.cout * e.cin * T;
    if (e.kind <= 1 && T == 9) {
        // 3x3 filters, a (filter, channel) pair per thread: its nine taps are 36 contiguous bytes of the source (a thread per
        // destination element read every source sector nine to sixteen times over), and for a fixed tap a wave's stores are contiguous
        const long long pairs = (long long)e.cout * e.cin;
        const int per = ELEMS_PER_BLOCK / 9 + 1;     // (the entry's workgroups were counted over cout * cin * 9 elements: >= pairs / per of them)
        for (int i = threadIdx.x; i < per; i += 256) {
            const long long idx = (long long)((int)blockIdx.x - e.block0) * per + i;
            if (idx >= pairs) return;
            int co, ci;
            if (e.kind == 0) { ci = (int)(idx % e.cin); co = (int)(idx / e.cin); }
            else { co = (int)(idx % e.cout); ci = (int)(idx / e.cout); }
            const float* g = e.src + ((size_t)co * e.cin_total + e.cin_off + ci) * 9;
            const float sc = e.gamma ? bn_scale(e.gamma, e.var, e.eps, co) : 1.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float v = g[t];
                if (e.gamma) v = __fmul_rn(v, sc);
                const size_t dst = e.kind == 0 ? (size_t)(e.dst_row0 + co) * e.dst_ld + (size_t)t * e.cin + ci
                                               : (size_t)(e.dst_row0 + ci) * e.dst_ld + (size_t)(8 - t) * e.cout_ld + e.dst_col0 + co;
                e.dst[dst] = v;
            }
        }
        return;
    }
    for (int i = threadIdx.x; i < ELEMS_PER_BLOCK; i += 256) {
        const long long idx = base + i;
        if (idx >= total) return;
        int co, ci, t;
        size_t dst;
        if (e.kind == 0) {                       // forward: dst[co][(t, ci)]
            ci = (int)(idx % e.cin);
            t = (int)((idx / e.cin) % T);
            co = (int)(idx / ((long long)e.cin * T));
            dst = (size_t)(e.dst_row0 + co) * e.dst_ld + (size_t)t * e.cin + ci;
        } else if (e.kind == 1) {                // data gradient: dst[ci][(T-1-t, co)]
            co = (int)(idx % e.cout);
            const int tf = (int)((idx / e.cout) % T);
            ci = (int)(idx / ((long long)e.cout * T));
            t = T - 1 - tf;
            dst = (size_t)(e.dst_row0 + ci) * e.dst_ld + (size_t)tf * e.cout_ld + e.dst_col0 + co;
        } else if (e.kind == 2) {                // stem: dst[((p0 + ci) * kh + y) * 8 + x][co], 8th tap stays zero
            co = (int)(idx % e.cout);
            t = (int)((idx / e.cout) % T);
            ci = (int)(idx / ((long long)e.cout * T));
            const int y = t / e.kw, x = t % e.kw;
            dst = ((size_t)((e.dst_row0 + ci) * e.kh + y) * 8 + x) * e.dst_ld + co;
        } else {                                 // tap form: dst[t * cout + co][ci]
            ci = (int)(idx % e.cin);
            co = (int)((idx / e.cin) % e.cout);
            t = (int)(idx / ((long long)e.cin * e.cout));
            dst = (size_t)(e.dst_row0 + t * e.cout + co) * e.dst_ld + ci;
        }
        float v = e.src[((size_t)co * e.cin_total + e.cin_off + ci) * T + t];
        if (e.gamma) v = __fmul_rn(v, bn_scale(e.gamma, e.var, e.eps, co));
        e.dst[dst] = v;
    }
}

__global__ __launch_bounds__(256)
void refresh_epilogues_kernel(const vfn_refresh_epilogue* __restrict__ table) {
    const vfn_refresh_epilogue e = table[blockIdx.x];
    for (int c = threadIdx.x; c < e.C; c += 256) {
        if (e.gamma) {
            const float s = bn_scale(e.gamma, e.var, e.eps, c);
            if (e.scale) e.scale[c] = s;
            if (e.shift) e.shift[c] = __fsub_rn(e.beta[c], __fmul_rn(e.mean[c], s));
        } else if (e.shift) {
            e.shift[c] = e.beta ? e.beta[c] : 0.f;
        }
    }
}

}  // namespace

extern "C" int vfn_refresh_elems_per_block(void) { return ELEMS_PER_BLOCK; }

extern "C" int vfn_refresh_filters_f32(const vfn_refresh_filter* table, int n, int total_blocks, void* stream) {
    if (!table || n < 1 || total_blocks < 1) return VFN_ERR_ARG;
    hipLaunchKernelGGL(refresh_filters_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table, n);
    return vfn_check_launch();
}

extern "C" int vfn_refresh_epilogues_f32(const vfn_refresh_epilogue* table, int n, void* stream) {
    if (!table || n < 1) return VFN_ERR_ARG;
    hipLaunchKernelGGL(refresh_epilogues_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, table);
    return vfn_check_launch();
}

// ---- the way back: the step's gradients into the optimizer's flat gradient buffer (train.AdamW.set_grads).  The backward pass
// hands them out as ~300 tensors -- strided [Cout,Cin,kh,kw] views of packed-layout accumulators, bias / BatchNorm vectors, a few
// concatenations -- and a copy launch each was 550 launches at the end of every step; here one launch copies all of them: an entry
// per tensor (<= 4 dimensions, any strides) in a device table that the host re-uses while the tensors keep their addresses.
namespace {
__global__ __launch_bounds__(256)
void gather_strided_kernel(const vfn_gather_entry* __restrict__ table, int n, float* __restrict__ dst) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const vfn_gather_entry e = table[lo];
    const long long total = (long long)e.shape[0] * e.shape[1] * e.shape[2] * e.shape[3];
    const long long base = (long long)((int)blockIdx.x - e.block0) * ELEMS_PER_BLOCK;
    float* out = dst + e.dst_offset;
    for (int i = threadIdx.x; i < ELEMS_PER_BLOCK; i += 256) {
        const long long idx = base + i;
        if (idx >= total) return;
        long long r = idx;
        const int i3 = (int)(r % e.shape[3]); r /= e.shape[3];
        const int i2 = (int)(r % e.shape[2]); r /= e.shape[2];
        const int i1 = (int)(r % e.shape[1]);
        const int i0 = (int)(r / e.shape[1]);
        out[idx] = e.src[i0 * e.stride[0] + i1 * e.stride[1] + i2 * e.stride[2] + i3 * e.stride[3]];
    }
}
}  // namespace

extern "C" int vfn_gather_strided_f32(const vfn_gather_entry* table, int n, int total_blocks, float* dst, void* stream) {
    if (!table || !dst || n < 1 || total_blocks < 1) return VFN_ERR_ARG;
    hipLaunchKernelGGL(gather_strided_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table, n, dst);
    return vfn_check_launch();
}
