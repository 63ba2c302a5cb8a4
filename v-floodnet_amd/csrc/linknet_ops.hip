// The pieces of the first-frame bootstrap model (smp.Linknet over EfficientNet-B4: test_image_seg.py:133 / test_video_seg.py:67-69;
// architecture restated in oracle/linknet_ref.py) that are not 1x1 / 4x4 convolutions -- those run through the implicit-GEMM
// kernel (conv_igemm.hip) on channel-padded NHWC tensors.  Everything here is HBM- or latency-bound elementwise / reduction work:
//   stem        Conv 3x3 / stride 2 / "same" padding (0 before, 1 after), 3 -> 48, BatchNorm, swish; planar input, NHWC output
//   depthwise   k x k (3 or 5), stride 1 / 2, asymmetric zero padding, BatchNorm, swish; optional swish on the input as it is
//               read (the expand convolution's epilogue has no swish: its consumer applies it)
//   SE gate     mean over the pixels (vfn_colsum_f32) -> 1x1 -> swish -> 1x1 -> sigmoid, one workgroup
//   column scale   W'[r][c] = W[r][c] * gate[c]: the gate multiplies the project convolution's INPUT channels, i.e. the columns
//               of its filter matrix -- 0.1..4 MB instead of a pass over the activation tensor
//   add, head (1x1 to one channel + sigmoid)
// The bootstrap runs once per clip; none of this is on the per-frame path.
#include "common.h"
#include "../../include/vfn_hip.h"

namespace {

__device__ __forceinline__ float swishf(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + __expf(-x)); }

inline int blocks_for(size_t total, int per = 256) {
    const size_t b = (total + per - 1) / per;
    return (int)(b < 65535 * 16 ? (b ? b : 1) : 65535 * 16);
}

// x [N][3][H][W] planar -> out [N][Ho][Wo][ld] (channels 48..ld-1 written as 0); w [48][3][3][3]; pad_b rows / columns of zeros
// before the image (the rest of the "same" padding falls after it)
__global__ __launch_bounds__(256)
void ln_stem_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                    const float* __restrict__ shift, float* __restrict__ out, int N, int H, int W, int Ho, int Wo, int ld, int pad_b) {
    __shared__ float sw[48 * 27];
    for (int i = threadIdx.x; i < 48 * 27; i += blockDim.x) sw[i] = w[i];
    __syncthreads();
    const int groups = ld / 16;                                   // 16 output channels per thread
    const size_t total = (size_t)N * Ho * Wo * groups;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % groups);
        const size_t pix = i / groups;
        const int xo = (int)(pix % Wo), yo = (int)((pix / Wo) % Ho), n = (int)(pix / ((size_t)Wo * Ho));
        float v[27];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int yi = yo * 2 - pad_b + ky, xi = xo * 2 - pad_b + kx;
                    v[(c * 3 + ky) * 3 + kx] = ((unsigned)yi < (unsigned)H && (unsigned)xi < (unsigned)W)
                                                   ? x[(((size_t)n * 3 + c) * H + yi) * W + xi] : 0.f;
                }
        float* o = out + pix * ld + g * 16;
#pragma unroll 4
        for (int cc = 0; cc < 16; ++cc) {
            const int co = g * 16 + cc;
            float r = 0.f;
            if (co < 48) {
                float a = 0.f;
#pragma unroll
                for (int t = 0; t < 27; ++t) a = fmaf(v[t], sw[co * 27 + t], a);
                r = swishf(a * scale[co] + shift[co]);
            }
            o[cc] = r;
        }
    }
}

// x [N][H][W][ld_x] -> out [N][Ho][Wo][ld_o], C channels (C % 4 == 0, padded channels carry zero filters / scale / shift);
// w [k*k][C] (tap-major); zero padding pad_b before (after: whatever Ho / Wo imply); act(x) = swish(x) when swish_in
template <int K>
__global__ __launch_bounds__(256)
void ln_dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                      const float* __restrict__ shift, float* __restrict__ out, int N, int H, int W, int C, int ld_x, int ld_o,
                      int stride, int pad_b, int Ho, int Wo, int swish_in) {
    const int c4n = C / 4;
    const size_t total = (size_t)N * Ho * Wo * c4n;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const size_t pix = i / c4n;
        const int xo = (int)(pix % Wo), yo = (int)((pix / Wo) % Ho), n = (int)(pix / ((size_t)Wo * Ho));
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int yi = yo * stride - pad_b + ky;
            if ((unsigned)yi >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int xi = xo * stride - pad_b + kx;
                if ((unsigned)xi >= (unsigned)W) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)n * H + yi) * W + xi) * ld_x + c4 * 4);
                if (swish_in) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = swishf(v[e]);
                }
                const f32x4 f = *reinterpret_cast<const f32x4*>(w + (size_t)(ky * K + kx) * C + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = fmaf(v[e], f[e], a[e]);
            }
        }
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c4 * 4), sh = *reinterpret_cast<const f32x4*>(shift + c4 * 4);
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = swishf(a[e] * sc[e] + sh[e]);
        *reinterpret_cast<f32x4*>(out + pix * ld_o + c4 * 4) = r;
    }
}

// gate[c] = sigmoid(b2[c] + sum_j w2[c][j] * swish(b1[j] + sum_c' w1[j][c'] * sum_px[c'] * inv_hw)); one workgroup of 1024
// (a wave per squeezed channel in turn, lanes over the input channels; then a thread per output channel).  sq <= 256.
__global__ __launch_bounds__(1024)
void ln_se_gate_kernel(const float* __restrict__ sum_px, float inv_hw, const float* __restrict__ w1, const float* __restrict__ b1,
                       const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ gate, int C, int sq, int Cpad) {
    __shared__ float s[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int j = wave; j < sq; j += nw) {
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a = fmaf(w1[(size_t)j * C + c], sum_px[c] * inv_hw, a);
        a = wave_sum(a);
        if (lane == 0) s[j] = swishf(a + b1[j]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < Cpad; c += blockDim.x) {
        float g = 0.f;
        if (c < C) {
            float a = b2[c];
            for (int j = 0; j < sq; ++j) a = fmaf(w2[(size_t)c * sq + j], s[j], a);
            g = sigmoidf(a);
        }
        gate[c] = g;                                               // padded channels: 0
    }
}

__global__ void ln_scale_cols_kernel(const float* __restrict__ w, const float* __restrict__ g, float* __restrict__ out, int rows, int K) {
    const int k4n = K / 4;
    const size_t total = (size_t)rows * k4n;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k4 = (int)(i % k4n);
        f32x4 v = *reinterpret_cast<const f32x4*>(w + i * 4);
        const f32x4 gg = *reinterpret_cast<const f32x4*>(g + k4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= gg[e];
        *reinterpret_cast<f32x4*>(out + i * 4) = v;
    }
}

__global__ void ln_add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(a + i * 4), y = *reinterpret_cast<const f32x4*>(b + i * 4);
        *reinterpret_cast<f32x4*>(out + i * 4) = x + y;
    }
}

// out[m] = sigmoid(bias + sum_c x[m][c] * w[c]) (or the logit when prob == 0); C % 4 == 0
__global__ void ln_head_kernel(const float* __restrict__ x, const float* __restrict__ w, float bias, float* __restrict__ out, size_t M, int C,
                               int ld, int prob) {
    for (size_t m = blockIdx.x * (size_t)blockDim.x + threadIdx.x; m < M; m += (size_t)gridDim.x * blockDim.x) {
        float a = bias;
        for (int c = 0; c < C; c += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + m * ld + c), f = *reinterpret_cast<const f32x4*>(w + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) a = fmaf(v[e], f[e], a);
        }
        out[m] = prob ? sigmoidf(a) : a;
    }
}

}  // namespace

extern "C" int vfn_ln_stem_f32(const float* x, const float* w, const float* scale, const float* shift, float* out, int N, int H, int W,
                               int Ho, int Wo, int ld, int pad_before, void* stream) {
    if (!x || !w || !scale || !shift || !out || N < 1 || H < 3 || W < 3 || Ho < 1 || Wo < 1 || ld < 48 || ld % 16 || pad_before < 0) return VFN_ERR_ARG;
    const size_t total = (size_t)N * Ho * Wo * (ld / 16);
    hipLaunchKernelGGL(ln_stem_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, out, N, H, W, Ho, Wo,
                       ld, pad_before);
    return vfn_check_launch();
}

extern "C" int vfn_ln_dwconv_f32(const float* x, const float* w, const float* scale, const float* shift, float* out, int N, int H, int W,
                                 int C, int ld_x, int ld_out, int k, int stride, int pad_before, int Ho, int Wo, int swish_in, void* stream) {
    if (!x || !w || !scale || !shift || !out || N < 1 || H < 1 || W < 1 || C < 4 || C % 4 || ld_x < C || ld_out < C || ld_x % 4 || ld_out % 4 ||
        (k != 3 && k != 5) || (stride != 1 && stride != 2) || pad_before < 0 || Ho < 1 || Wo < 1)
        return VFN_ERR_ARG;
    const size_t total = (size_t)N * Ho * Wo * (C / 4);
    const dim3 grid(blocks_for(total));
    if (k == 3)
        hipLaunchKernelGGL(ln_dwconv_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, out, N, H, W, C, ld_x, ld_out,
                           stride, pad_before, Ho, Wo, swish_in);
    else
        hipLaunchKernelGGL(ln_dwconv_kernel<5>, grid, dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, out, N, H, W, C, ld_x, ld_out,
                           stride, pad_before, Ho, Wo, swish_in);
    return vfn_check_launch();
}

extern "C" int vfn_ln_se_gate_f32(const float* sum_px, float inv_hw, const float* w1, const float* b1, const float* w2, const float* b2,
                                  float* gate, int C, int sq, int Cpad, void* stream) {
    if (!sum_px || !w1 || !b1 || !w2 || !b2 || !gate || C < 1 || sq < 1 || sq > 256 || Cpad < C) return VFN_ERR_ARG;
    hipLaunchKernelGGL(ln_se_gate_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, sum_px, inv_hw, w1, b1, w2, b2, gate, C, sq, Cpad);
    return vfn_check_launch();
}

extern "C" int vfn_ln_scale_cols_f32(const float* w, const float* g, float* out, int rows, int K, void* stream) {
    if (!w || !g || !out || rows < 1 || K < 4 || K % 4) return VFN_ERR_ARG;
    hipLaunchKernelGGL(ln_scale_cols_kernel, dim3(blocks_for((size_t)rows * (K / 4))), dim3(256), 0, (hipStream_t)stream, w, g, out, rows, K);
    return vfn_check_launch();
}

extern "C" int vfn_ln_add_f32(const float* a, const float* b, float* out, long long n, void* stream) {
    if (!a || !b || !out || n < 4 || n % 4) return VFN_ERR_ARG;
    hipLaunchKernelGGL(ln_add_kernel, dim3(blocks_for((size_t)n / 4)), dim3(256), 0, (hipStream_t)stream, a, b, out, (size_t)n / 4);
    return vfn_check_launch();
}

extern "C" int vfn_ln_head_f32(const float* x, const float* w, float bias, float* out, long long M, int C, int ld, int prob, void* stream) {
    if (!x || !w || !out || M < 1 || C < 4 || C % 4 || ld < C || ld % 4) return VFN_ERR_ARG;
    hipLaunchKernelGGL(ln_head_kernel, dim3(blocks_for((size_t)M)), dim3(256), 0, (hipStream_t)stream, x, w, bias, out, (size_t)M, C, ld, prob);
    return vfn_check_launch();
}
