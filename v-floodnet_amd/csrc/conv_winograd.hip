// Winograd F(4x4, 3x3) around the matrix kernels (round 4): the 3x3 / stride-1 / pad-1 convolutions of the decoder
// (AFB_URR.py:20-30,114-127,191-195: 256 -> 256 filters on 1/4- and 1/8-resolution feature maps, 2/3 of the frame's
// convolution FLOP) as 36 GEMMs in the transform domain -- 36 multiplies per 4x4 output tile and filter pair instead of 144:
//
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A          d: 6x6 input tile (pad 1), g: 3x3 filter, Y: 4x4 outputs
//
//   vfn_winograd_input_f32    V[xi][tile][c]  = (B^T d B)[xi]   (xi = 6 i + j; ReLU on d first for the pre-activation ResBlocks)
//   the 36 GEMMs              M[xi][tile][co] = sum_c V[xi][tile][c] U[xi][co][c]: ONE launch of the convolution kernels with
//                             batched filters (vfn_conv_desc.w_batch_rows = rows per component), U = G g G^T packed by the host
//   vfn_winograd_output_f32   Y = A^T M A per tile, then the convolution's epilogue: * scale + shift (+ residual) (ReLU)
//
// The same algebra cuDNN applies to the reference's convolutions on its own hardware; it is exact in real arithmetic, in f32
// the transforms add ~1e-6 relative rounding (tests/test_conv_gpu.py holds it against F.conv2d like every other configuration).
// Both transforms are HBM-bound: a thread owns 4 channels of one tile, so every load / store of a wave is a contiguous
// 256-byte to 1-KB row; V and M cost 36/16 of the tensor each way (119 MB each for 2 x 120 x 216 x 256).
#include "common.h"
#include "../../include/vfn_hip.h"

namespace {

// t = B^T v (6 values)
__device__ __forceinline__ void bt6(const f32x4 (&v)[6], f32x4 (&t)[6]) {
    t[0] = 4.f * v[0] - 5.f * v[2] + v[4];
    t[1] = -4.f * (v[1] + v[2]) + v[3] + v[4];
    t[2] = 4.f * (v[1] - v[2]) - v[3] + v[4];
    t[3] = -2.f * v[1] - v[2] + 2.f * v[3] + v[4];
    t[4] = 2.f * v[1] - v[2] - 2.f * v[3] + v[4];
    t[5] = 4.f * v[1] - 5.f * v[3] + v[5];
}

__global__ __launch_bounds__(256)
void winograd_input_kernel(const float* __restrict__ x, int N, int H, int W, int C, int ld_x, int relu, float* __restrict__ V,
                           int rows_pad) {
    const int th = (H + 3) / 4, tw = (W + 3) / 4;
    const int c4n = C / 4;
    const long long total = (long long)N * th * tw * c4n;
    for (long long i = vfn_xcd_block(blockIdx.x, gridDim.x) * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const int tile = (int)(i / c4n);
        const int tx = tile % tw, ty = (tile / tw) % th, n = tile / (tw * th);
        f32x4 d[6][6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const int yy = 4 * ty - 1 + a;
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const int xx = 4 * tx - 1 + b;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                    v = *reinterpret_cast<const f32x4*>(x + ((size_t)(n * H + yy) * W + xx) * ld_x + c4 * 4);
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                d[a][b] = v;
            }
        }
        // columns: d <- B^T d, then rows: V = d B
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            f32x4 v[6], t[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) v[a] = d[a][b];
            bt6(v, t);
#pragma unroll
            for (int a = 0; a < 6; ++a) d[a][b] = t[a];
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            f32x4 t[6];
            bt6(d[a], t);
#pragma unroll
            for (int b = 0; b < 6; ++b)
                *reinterpret_cast<f32x4*>(V + ((size_t)(a * 6 + b) * rows_pad + tile) * C + c4 * 4) = t[b];
        }
    }
}

// y = A^T m (6 values -> 4)
__device__ __forceinline__ void at6(const f32x4 (&m)[6], f32x4 (&y)[4]) {
    const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = m[0] + s12 + s34;
    y[1] = d12 + 2.f * d34;
    y[2] = s12 + 4.f * s34;
    y[3] = d12 + 8.f * d34 + m[5];
}

__global__ __launch_bounds__(256)
void winograd_output_kernel(const float* __restrict__ Mb, int rows_pad, int N, int H, int W, int Cout,
                            const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ res, int res_ld,
                            int res_mod, int relu_out, float* __restrict__ out, int out_ld, const float* __restrict__ mask, int mask_ld,
                            int mask_after) {
    const int th = (H + 3) / 4, tw = (W + 3) / 4;
    const int c4n = Cout / 4;
    const long long total = (long long)N * th * tw * c4n;
    for (long long i = vfn_xcd_block(blockIdx.x, gridDim.x) * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const int tile = (int)(i / c4n);
        const int tx = tile % tw, ty = (tile / tw) % th, n = tile / (tw * th);
        f32x4 t[6][4];                            // t = M A  (rows of M through A^T)
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            f32x4 m[6];
#pragma unroll
            for (int b = 0; b < 6; ++b) m[b] = *reinterpret_cast<const f32x4*>(Mb + ((size_t)(a * 6 + b) * rows_pad + tile) * Cout + c4 * 4);
            at6(m, t[a]);
        }
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (scale) sc = *reinterpret_cast<const f32x4*>(scale + c4 * 4);
        if (shift) sh = *reinterpret_cast<const f32x4*>(shift + c4 * 4);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            f32x4 col[6], y[4];
#pragma unroll
            for (int a = 0; a < 6; ++a) col[a] = t[a][b];
            at6(col, y);
            const int xx = 4 * tx + b;
            if (xx >= W) continue;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int yy = 4 * ty + a;
                if (yy >= H) continue;
                const size_t row = (size_t)(n * H + yy) * W + xx;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = y[a][e] * sc[e] + sh[e];
                f32x4 mk = {1.f, 1.f, 1.f, 1.f};               // (the data-gradient form: the ReLU in front of the forward convolution)
                if (mask) mk = *reinterpret_cast<const f32x4*>(mask + row * mask_ld + c4 * 4);
                if (mask && !mask_after) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
                }
                if (res) v += *reinterpret_cast<const f32x4*>(res + (res_mod > 0 ? row % res_mod : row) * res_ld + c4 * 4);
                if (mask && mask_after) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
                }
                if (relu_out) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                *reinterpret_cast<f32x4*>(out + row * out_ld + c4 * 4) = v;
            }
        }
    }
}

// ---- the weight gradient in the transform domain (round 4): dW = G^T [ sum_tiles (B^T d B) (.) (A dY A^T) ] G -- the transposition
// of the forward algorithm: 36 multiplies per (tile, filter, channel) instead of the direct form's 144.
//   vfn_winograd_input_f32   V[xi][tile][ci] = (B^T d B)[xi]           (the forward's input transform, ReLU included)
//   vfn_winograd_gy_f32      Z[xi][tile][co] = (A dY A^T)[xi]          dY: the 4x4 tile of dL/dy (zero outside the image)
//   vfn_conv_wgrad_f32       dU[xi][co][ci]  = sum_tile Z[xi][tile][co] V[xi][tile][ci]     (batch = 36 one-by-one problems)
//   vfn_winograd_dw_f32      dW[co][a][b][ci] (+)= rowscale[co] * sum_ij G[i][a] dU[6i+j][co][ci] G[j][b]
// z = A v (4 values -> 6), A = (A^T)^T
__device__ __forceinline__ void a6(const f32x4 (&v)[4], f32x4 (&z)[6]) {
    const f32x4 e = v[0] + v[2], o = v[1] + v[3];
    const f32x4 e4 = v[0] + 4.f * v[2], o8 = 2.f * v[1] + 8.f * v[3];
    z[0] = v[0];
    z[1] = e + o;
    z[2] = e - o;
    z[3] = e4 + o8;
    z[4] = e4 - o8;
    z[5] = v[3];
}

__global__ __launch_bounds__(256)
void winograd_gy_kernel(const float* __restrict__ gy, int N, int H, int W, int C, int ld, float* __restrict__ Z, int rows_pad) {
    const int th = (H + 3) / 4, tw = (W + 3) / 4;
    const int c4n = C / 4;
    const long long total = (long long)N * th * tw * c4n;
    for (long long i = vfn_xcd_block(blockIdx.x, gridDim.x) * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const int tile = (int)(i / c4n);
        const int tx = tile % tw, ty = (tile / tw) % th, n = tile / (tw * th);
        f32x4 t[6][4];                               // t = A dY (columns of dY through A)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            f32x4 col[4], z[6];
            const int xx = 4 * tx + b;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int yy = 4 * ty + a;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (yy < H && xx < W) v = *reinterpret_cast<const f32x4*>(gy + ((size_t)(n * H + yy) * W + xx) * ld + c4 * 4);
                col[a] = v;
            }
            a6(col, z);
#pragma unroll
            for (int a = 0; a < 6; ++a) t[a][b] = z[a];
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            f32x4 z[6];
            a6(t[a], z);
#pragma unroll
            for (int b = 0; b < 6; ++b)
                *reinterpret_cast<f32x4*>(Z + ((size_t)(a * 6 + b) * rows_pad + tile) * C + c4 * 4) = z[b];
        }
    }
}

// w = G^T u (6 values -> 3)
__device__ __forceinline__ void gt3(const f32x4 (&u)[6], f32x4 (&w)[3]) {
    const f32x4 s12 = u[1] + u[2], d12 = u[2] - u[1], s34 = u[3] + u[4], d34 = u[3] - u[4];
    w[0] = 0.25f * u[0] - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
    w[1] = (1.f / 6.f) * d12 + (1.f / 12.f) * d34;
    w[2] = -(1.f / 6.f) * s12 + (1.f / 6.f) * s34 + u[5];
}

__global__ __launch_bounds__(256)
void winograd_dw_kernel(const float* __restrict__ dU, int Cout, int Cin, const float* __restrict__ rowscale, float* __restrict__ dw,
                        int accumulate) {
    const int c4n = Cin / 4;
    const long long total = (long long)Cout * c4n;
    const size_t bank = (size_t)Cout * Cin;
    for (long long i = vfn_xcd_block(blockIdx.x, gridDim.x) * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n), co = (int)(i / c4n);
        f32x4 t[3][6];                               // t = G^T dU (columns through G^T)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            f32x4 u[6], w[3];
#pragma unroll
            for (int a = 0; a < 6; ++a) u[a] = *reinterpret_cast<const f32x4*>(dU + (size_t)(a * 6 + j) * bank + (size_t)co * Cin + c4 * 4);
            gt3(u, w);
#pragma unroll
            for (int a = 0; a < 3; ++a) t[a][j] = w[a];
        }
        const float sc = rowscale ? rowscale[co] : 1.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            f32x4 w[3];
            gt3(t[a], w);
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                f32x4* o = reinterpret_cast<f32x4*>(dw + ((size_t)co * 9 + a * 3 + b) * Cin + c4 * 4);
                f32x4 v = w[b] * sc;
                if (accumulate) v += *o;
                *o = v;
            }
        }
    }
}

inline int grid_of(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b < 16384 ? (b ? b : 1) : 16384);
}

}  // namespace

extern "C" int vfn_winograd_tiles(int N, int H, int W) { return N * ((H + 3) / 4) * ((W + 3) / 4); }

extern "C" int vfn_winograd_input_f32(const float* x, int N, int H, int W, int C, int ld_x, int relu, float* V, int rows_pad, void* stream) {
    if (!x || !V || N < 1 || H < 1 || W < 1 || C < 4 || C % 4 || ld_x < C || ld_x % 4 || rows_pad < vfn_winograd_tiles(N, H, W)) return VFN_ERR_ARG;
    const long long total = (long long)vfn_winograd_tiles(N, H, W) * (C / 4);
    hipLaunchKernelGGL(winograd_input_kernel, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, x, N, H, W, C, ld_x, relu, V, rows_pad);
    return vfn_check_launch();
}

extern "C" int vfn_winograd_output_f32(const float* Mb, int rows_pad, int N, int H, int W, int Cout, const float* scale, const float* shift,
                                       const float* res, int res_ld, int res_mod, int relu_out, float* out, int out_ld, void* stream) {
    if (!Mb || !out || N < 1 || H < 1 || W < 1 || Cout < 4 || Cout % 4 || out_ld < Cout || out_ld % 4 || (res && res_ld % 4) ||
        rows_pad < vfn_winograd_tiles(N, H, W)) return VFN_ERR_ARG;
    const long long total = (long long)vfn_winograd_tiles(N, H, W) * (Cout / 4);
    hipLaunchKernelGGL(winograd_output_kernel, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, Mb, rows_pad, N, H, W, Cout, scale, shift,
                       res, res_ld, res_mod, relu_out, out, out_ld, (const float*)nullptr, 0, 0);
    return vfn_check_launch();
}

// ... with the epilogue of a data-gradient convolution (vfn_conv_desc.mask / mask_after): the result is zeroed where mask <= 0,
// before (mask_after = 0) or after the residual is added
extern "C" int vfn_winograd_output_masked_f32(const float* Mb, int rows_pad, int N, int H, int W, int Cout, const float* res, int res_ld,
                                              const float* mask, int mask_ld, int mask_after, float* out, int out_ld, void* stream) {
    if (!Mb || !out || N < 1 || H < 1 || W < 1 || Cout < 4 || Cout % 4 || out_ld < Cout || out_ld % 4 || (res && res_ld % 4) ||
        (mask && mask_ld % 4) || rows_pad < vfn_winograd_tiles(N, H, W)) return VFN_ERR_ARG;
    const long long total = (long long)vfn_winograd_tiles(N, H, W) * (Cout / 4);
    hipLaunchKernelGGL(winograd_output_kernel, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, Mb, rows_pad, N, H, W, Cout,
                       (const float*)nullptr, (const float*)nullptr, res, res_ld, 0, 0, out, out_ld, mask, mask_ld, mask_after);
    return vfn_check_launch();
}

extern "C" int vfn_winograd_gy_f32(const float* gy, int N, int H, int W, int C, int ld, float* Z, int rows_pad, void* stream) {
    if (!gy || !Z || N < 1 || H < 1 || W < 1 || C < 4 || C % 4 || ld < C || ld % 4 || rows_pad < vfn_winograd_tiles(N, H, W)) return VFN_ERR_ARG;
    const long long total = (long long)vfn_winograd_tiles(N, H, W) * (C / 4);
    hipLaunchKernelGGL(winograd_gy_kernel, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, gy, N, H, W, C, ld, Z, rows_pad);
    return vfn_check_launch();
}

extern "C" int vfn_winograd_dw_f32(const float* dU, int Cout, int Cin, const float* rowscale, float* dw, int accumulate, void* stream) {
    if (!dU || !dw || Cout < 1 || Cin < 4 || Cin % 4) return VFN_ERR_ARG;
    hipLaunchKernelGGL(winograd_dw_kernel, dim3(grid_of((long long)Cout * (Cin / 4))), dim3(256), 0, (hipStream_t)stream, dU, Cout, Cin, rowscale,
                       dw, accumulate);
    return vfn_check_launch();
}
