// Backward pass, first slice (train_video_seg.py:65-74 runs loss.backward() through the decoder):
// the HBM-bound pieces around the gradient GEMMs.
//
//   data gradient    dX = conv(dY, W flipped and transposed): the forward implicit-GEMM kernel (conv_igemm.hip) over
//                    repacked filters; the ReLU in front of the forward convolution and the skip connection are undone in
//                    its epilogue (vfn_conv_desc.mask / res).  Nothing new here.
//   weight gradient  dW[co][kh][kw][ci] = sum_m dY[m][co] * act(X)[m + (kh,kw)][ci]: a GEMM whose reduction runs over
//                    the PIXELS.  Both operands are pixel-major in HBM (NHWC), i.e. strided along the reduction, so they are
//                    transposed first -- vfn_transpose_taps_f32 writes dY^T [Cout][Mpad] (taps = 1) and the transposed
//                    im2col image act(X)^T [9 * Cin][Mpad] (taps = 9, zero outside the image) -- and the forward kernel
//                    then sees an ordinary 1x1 problem: 'pixels' = Cout rows, 'channels' = Mpad, 'filters' = 9 * Cin rows,
//                    cut along K (= the pixels) over the whole chip (vfn_conv_desc.ksplit).  The result is the packed
//                    filter layout [Cout][kh][kw][Cin] itself.
//   bias gradient    vfn_colsum_f32: column sums of dY, two deterministic stages.
//   Refine           m = s + interpolate(pm, x2): vfn_upsample2x_add_backward_f32 -- ds = sum over the objects that share s
//                    (AFB_URR.py:289-295 only expands r3 / r2), dpm = the adjoint of the bilinear interpolation (a gather
//                    over the <= 4 x 4 fine pixels whose taps touch a coarse pixel: no atomics, fixed order).
#include "common.h"
#include "../../include/vfn_hip.h"

namespace {

struct Lerp { int i0, i1; float l0, l1; };

// bilinear x2, align_corners=False (decoder_ops.hip)
__device__ __forceinline__ Lerp lerp2x(int dst, int in_size) {
    float src = 0.5f * (dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    Lerp L;
    L.i0 = (int)src;
    L.i1 = L.i0 + (L.i0 < in_size - 1 ? 1 : 0);
    L.l1 = src - L.i0;
    L.l0 = 1.f - L.l1;
    return L;
}

inline int grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    return (int)(b < 8192 ? (b ? b : 1) : 8192);
}

// out[(tap * C + c)][m] = act(x[n][y + dy][x + dx][c]) (0 outside the image), m = (n, y, x) flattened; columns M .. Mpad-1
// are written as zeros.  32 x 32 tiles through LDS: reads run along the channels, writes along the pixels.
__global__ __launch_bounds__(256)
void transpose_taps_kernel(const float* __restrict__ x, int N, int H, int W, int C, int ld_x, int relu, int taps,
                           float* __restrict__ out, int Mpad) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int m0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tap = blockIdx.z;
    const int dy = taps == 9 ? tap / 3 - 1 : 0, dx = taps == 9 ? tap % 3 - 1 : 0;
    const int M = N * H * W;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int m = m0 + r;
        float v = 0.f;
        if (m < M && c0 + tx < C) {
            const int n = m / (H * W), rem = m - n * H * W;
            const int yy = rem / W + dy, xx = rem % W + dx;
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
                v = x[((size_t)(n * H + yy) * W + xx) * ld_x + c0 + tx];
                if (relu) v = fmaxf(v, 0.f);
            }
        }
        tile[r][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, m = m0 + tx;
        if (c < C && m < Mpad) out[((size_t)tap * C + c) * Mpad + m] = tile[tx][r];
    }
}

// stage 1: block b sums rows b, b + gridDim.x, ... of x [M][ld] for every column; stage 2 adds the partials in block order
__global__ void colsum_partial_kernel(const float* __restrict__ x, int M, int C, int ld, float* __restrict__ partial) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int m = blockIdx.x; m < M; m += gridDim.x) s += x[(size_t)m * ld + c];
        partial[(size_t)blockIdx.x * C + c] = s;
    }
}
__global__ void colsum_final_kernel(const float* __restrict__ partial, int nb, int C, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int b = 0; b < nb; ++b) s += partial[(size_t)b * C + c];
    out[c] = s;
}

// gs[p][c] = sum_n gm[n][p][c]  (the objects share s)
__global__ void sum_objects_kernel(const float* __restrict__ gm, float* __restrict__ gs, int N, size_t per4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 s = *reinterpret_cast<const f32x4*>(gm + i * 4);
        for (int n = 1; n < N; ++n) s += *reinterpret_cast<const f32x4*>(gm + ((size_t)n * per4 + i) * 4);
        *reinterpret_cast<f32x4*>(gs + i * 4) = s;
    }
}

// gpm[n][y][x][c] = sum over the fine pixels (Y, X) whose interpolation taps include (y, x) of weight * gm[n][Y][X][c];
// fine rows 2y-1 .. 2y+2 are the only candidates (their i0 / i1 are recomputed with lerp2x, so the edge clamps of the forward
// are mirrored exactly); fixed summation order.
__global__ void upsample2x_adjoint_kernel(const float* __restrict__ gm, float* __restrict__ gpm, int N, int h, int w, int C) {
    const int c4n = C / 4;
    const int hi = h / 2, wi = w / 2;
    const size_t total = (size_t)N * hi * wi * c4n;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = i % c4n;
        size_t t = i / c4n;
        const int x = t % wi; t /= wi;
        const int y = t % hi;
        const int n = t / hi;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int Y = 2 * y - 1; Y <= 2 * y + 2; ++Y) {
            if (Y < 0 || Y >= h) continue;
            const Lerp ly = lerp2x(Y, hi);
            const float wy = (ly.i0 == y ? ly.l0 : 0.f) + (ly.i1 == y ? ly.l1 : 0.f);
            if (wy == 0.f) continue;
            for (int X = 2 * x - 1; X <= 2 * x + 2; ++X) {
                if (X < 0 || X >= w) continue;
                const Lerp lx = lerp2x(X, wi);
                const float wx = (lx.i0 == x ? lx.l0 : 0.f) + (lx.i1 == x ? lx.l1 : 0.f);
                if (wx == 0.f) continue;
                const f32x4 g = *reinterpret_cast<const f32x4*>(gm + (((size_t)n * h + Y) * w + X) * C + c4 * 4);
                const float wgt = wy * wx;
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] += wgt * g[k];
            }
        }
        *reinterpret_cast<f32x4*>(gpm + i * 4) = acc;
    }
}

}  // namespace

extern "C" int vfn_transpose_taps_f32(const float* x, int N, int H, int W, int C, int ld_x, int relu, int taps, float* out,
                                      int Mpad, void* stream) {
    if (!x || !out || N < 1 || H < 1 || W < 1 || C < 1 || ld_x < C || (taps != 1 && taps != 9) || Mpad < N * H * W) return VFN_ERR_ARG;
    const dim3 grid(cdiv(Mpad, 32), cdiv(C, 32), taps);
    hipLaunchKernelGGL(transpose_taps_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, N, H, W, C, ld_x, relu, taps, out, Mpad);
    return vfn_check_launch();
}

extern "C" int vfn_colsum_f32(const float* x, int M, int C, int ld, float* partial, int nb, float* out, void* stream) {
    if (!x || !partial || !out || M < 1 || C < 1 || ld < C || nb < 1 || nb > 1024) return VFN_ERR_ARG;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, M, C, ld, partial);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, partial, nb, C, out);
    return vfn_check_launch();
}

extern "C" int vfn_upsample2x_add_backward_f32(const float* gm, float* gs, float* gpm, int N, int h, int w, int C, int s_bcast,
                                               void* stream) {
    if (!gm || !gpm || C % 4 || h % 2 || w % 2 || N < 1) return VFN_ERR_ARG;
    if (s_bcast) {
        if (!gs) return VFN_ERR_ARG;
        const size_t per4 = (size_t)h * w * (C / 4);
        hipLaunchKernelGGL(sum_objects_kernel, dim3(grid_for(per4)), dim3(256), 0, (hipStream_t)stream, gm, gs, N, per4);
    }
    const size_t total = (size_t)N * (h / 2) * (w / 2) * (C / 4);
    hipLaunchKernelGGL(upsample2x_adjoint_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, gm, gpm, N, h, w, C);
    return vfn_check_launch();
}
