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

// out[(tap * C + c)][m] = cs[c] * act(x[n][yo * stride - pad + kh][xo * stride - pad + kw][c]) (0 outside the image), tap = kh * k + kw,
// m = (n, yo, xo) flattened over the OUTPUT pixels of a k x k / stride / pad convolution; columns M .. Mpad-1 are written as
// zeros.  k = 1, stride = 1, pad = 0 is a plain transposition.  32 x 32 tiles through LDS: reads run along the channels,
// writes along the pixels.
__global__ __launch_bounds__(256)
void transpose_taps_kernel(const float* __restrict__ x, int N, int H, int W, int C, int ld_x, int relu, int k, int stride, int pad,
                           int Ho, int Wo, const float* __restrict__ colscale, float* __restrict__ out, int Mpad) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int m0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tap = blockIdx.z;
    const int dy = tap / k - pad, dx = tap % k - pad;
    const int M = N * Ho * Wo;
    const float cs = (colscale && c0 + tx < C) ? colscale[c0 + tx] : 1.f;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int m = m0 + r;
        float v = 0.f;
        if (m < M && c0 + tx < C) {
            const int n = m / (Ho * Wo), rem = m - n * Ho * Wo;
            const int yy = (rem / Wo) * stride + dy, xx = (rem % Wo) * stride + dx;
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
                v = x[((size_t)(n * H + yy) * W + xx) * ld_x + c0 + tx];
                if (relu) v = fmaxf(v, 0.f);
                v *= cs;
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
// Second stage of the two-stage column sums: 32 channels per workgroup, the nb partial rows dealt to 8 row groups (32 independent
// loads per thread instead of a 256-long dependent chain: 55 us -> a few), the groups added in fixed order (deterministic).
__device__ __forceinline__ float final_sum(const float* __restrict__ partial, int nb, size_t row_stride, int c, int C, float* red) {
    const int cx = threadIdx.x & 31, by = threadIdx.x >> 5;
    float s = 0.f;
    if (c < C)
        for (int b = by; b < nb; b += 8) s += partial[(size_t)b * row_stride + c];
    red[by * 32 + cx] = s;
    __syncthreads();
    float t = 0.f;
    if (by == 0)
#pragma unroll
        for (int g = 0; g < 8; ++g) t += red[g * 32 + cx];
    __syncthreads();
    return t;
}
__global__ void colsum_final_kernel(const float* __restrict__ partial, int nb, int C, float* __restrict__ out, int accumulate = 0) {
    __shared__ float red[256];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    const float s = final_sum(partial, nb, C, c, C, red);
    if (threadIdx.x < 32 && c < C) out[c] = accumulate ? out[c] + s : s;
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

// ------------------------------------------------------------------ the decoder's tail, backwards (AFB_URR.py:214-237,300,309-316)
// forward (decoder_ops.hip): p_up = up2(p); rp = softmax_c(p_up)[1]; rough = softmax_k(rp); unc = exp(1 - top1/(top2+1e-8));
//   r1_local = box7(r1 * rough)/49 / (box7(rough)/49 + 1e-8); conf = max7(rough); q = local head(cat[r1, r1_local]);
//   p2 = p_up + unc * conf * q; o = up2(p2); s = softmax_c(o)[1]; score = logit(clamp(s, 1e-7, 1-1e-7)), un-padded.
constexpr int MAX_OBJ = 8;

// T1: g_o[n][Y][X][0..3] (padded frame, channels 2, 3 and the padding stay zero) from G = dL/dscore [n][H0][W0]:
// score = o1 - o0 where the clamp is inactive, constant otherwise
__global__ void tail_grad_o_kernel(const float* __restrict__ G, const float* __restrict__ p_up, const float* __restrict__ unc,
                                   const float* __restrict__ conf, const float* __restrict__ q, float* __restrict__ g_o,
                                   int obj_n, int h, int w, int pad_top, int pad_left, int H0, int W0) {
    const size_t total = (size_t)obj_n * H0 * W0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x0 = i % W0;
        size_t t = i / W0;
        const int y0 = t % H0;
        const int n = t / H0;
        const int y = y0 + pad_top, x = x0 + pad_left;
        const Lerp ly = lerp2x(y, h), lx = lerp2x(x, w);
        float v[2][4];
        const int ys[2] = {ly.i0, ly.i1}, xs[2] = {lx.i0, lx.i1};
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) {
                const size_t pix = (size_t)ys[a] * w + xs[b];
                const size_t pn = (size_t)n * h * w + pix;
                const float u = unc[pix], cf = conf[pn];
                v[0][a * 2 + b] = p_up[pn * 2] + u * (cf * q[pn * 2]);
                v[1][a * 2 + b] = p_up[pn * 2 + 1] + u * (cf * q[pn * 2 + 1]);
            }
        const float o0 = ly.l0 * (lx.l0 * v[0][0] + lx.l1 * v[0][1]) + ly.l1 * (lx.l0 * v[0][2] + lx.l1 * v[0][3]);
        const float o1 = ly.l0 * (lx.l0 * v[1][0] + lx.l1 * v[1][1]) + ly.l1 * (lx.l0 * v[1][2] + lx.l1 * v[1][3]);
        const float m = fmaxf(o0, o1);
        const float e0 = expf(o0 - m), e1 = expf(o1 - m);
        const float sft = e1 / (e0 + e1);
        const float g = (sft > 1e-7f && sft < 1.f - 1e-7f) ? G[i] : 0.f;
        float* dst = g_o + (((size_t)n * 2 * h + y) * (2 * w) + x) * 4;
        dst[0] = -g;
        dst[1] = g;
    }
}

// T2: p2 = p_up + unc * conf * q.  g_p2 [n][pix][4] (2 used) -> g_q [n][pix][32] (2 used, rest zero), g_cf [n][pix], g_u [pix]
__global__ void tail_split_kernel(const float* __restrict__ g_p2, const float* __restrict__ unc, const float* __restrict__ conf,
                                  const float* __restrict__ q, float* __restrict__ g_q, float* __restrict__ g_cf,
                                  float* __restrict__ g_u, int obj_n, int npix) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        const float u = unc[i];
        float gu = 0.f;
        for (int n = 0; n < obj_n; ++n) {
            const size_t pn = (size_t)n * npix + i;
            const float g0 = g_p2[pn * 4], g1 = g_p2[pn * 4 + 1];
            const float cf = conf[pn], q0 = q[pn * 2], q1 = q[pn * 2 + 1];
            g_q[pn * 32] = g0 * u * cf;
            g_q[pn * 32 + 1] = g1 * u * cf;
            const float dot = g0 * q0 + g1 * q1;
            g_cf[pn] = u * dot;
            gu += cf * dot;
        }
        g_u[i] = gu;
    }
}

// arg-max position of the 7x7 window of rough around every pixel (first maximum in row-major order, as MaxPool2d)
__global__ void window_argmax_kernel(const float* __restrict__ rough, int* __restrict__ amax, int obj_n, int h, int w) {
    const int total = obj_n * h * w;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int x = i % w, y = (i / w) % h, n = i / (h * w);
        const float* r = rough + (size_t)n * h * w;
        float best = -INFINITY;
        int arg = y * w + x;
        for (int dy = -3; dy <= 3; ++dy) {
            const int yy = y + dy;
            if ((unsigned)yy >= (unsigned)h) continue;
            for (int dx = -3; dx <= 3; ++dx) {
                const int xx = x + dx;
                if ((unsigned)xx >= (unsigned)w) continue;
                const float v = r[yy * w + xx];
                if (v > best) { best = v; arg = yy * w + xx; }
            }
        }
        amax[i] = arg;
    }
}

// T4a: dA [n][pix][C] = g_lm / Bv,  dBv [n][pix] = -sum_c g_lm * r1_local / Bv, Bv = box7(rough)/49 + 1e-8 (recomputed)
__global__ void local_ratio_bwd_kernel(const float* __restrict__ g_lm, const float* __restrict__ lm, const float* __restrict__ rough,
                                       float* __restrict__ dA, float* __restrict__ dBv, int obj_n, int h, int w, int C) {
    const int total = obj_n * h * w;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int x = i % w, y = (i / w) % h, n = i / (h * w);
        const float* r = rough + (size_t)n * h * w;
        float sr = 0.f;
        for (int dy = -3; dy <= 3; ++dy) {
            const int yy = y + dy;
            if ((unsigned)yy >= (unsigned)h) continue;
            for (int dx = -3; dx <= 3; ++dx) {
                const int xx = x + dx;
                if ((unsigned)xx < (unsigned)w) sr += r[yy * w + xx];
            }
        }
        const float bv = sr / 49.f + 1e-8f;
        float acc = 0.f;
        for (int c = 0; c < C; ++c) {
            const float g = g_lm[(size_t)i * C + c];
            dA[(size_t)i * C + c] = g / bv;
            acc += g * lm[(size_t)i * C + c];
        }
        dBv[i] = -acc / bv;
    }
}

// ... for C = 64 with the channels across lanes (round 4): 16 lanes per pixel, a float4 of channels each -- the loads and stores of
// a wave are 4 pixels x 256 contiguous bytes (the kernel above strides 256 bytes between neighbouring lanes: 182 us per call at
// 2 x 200 x 200); the 49 window taps and the channel sum are dealt to the 16 lanes and folded with four xor-shuffles
__global__ __launch_bounds__(256)
void local_ratio_bwd64_kernel(const float* __restrict__ g_lm, const float* __restrict__ lm, const float* __restrict__ rough,
                              float* __restrict__ dA, float* __restrict__ dBv, int obj_n, int h, int w) {
    const int total = obj_n * h * w;
    const int sub = threadIdx.x & 15;
    for (int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4; i < total; i += (gridDim.x * blockDim.x) >> 4) {
        const int x = i % w, y = (i / w) % h, n = i / (h * w);
        const float* r = rough + (size_t)n * h * w;
        float sr = 0.f;
        for (int t = sub; t < 49; t += 16) {
            const int yy = y + t / 7 - 3, xx = x + t % 7 - 3;
            if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) sr += r[yy * w + xx];
        }
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) sr += __shfl_xor(sr, m, 16);
        const float bv = sr / 49.f + 1e-8f;
        const f32x4 g = *reinterpret_cast<const f32x4*>(g_lm + (size_t)i * 64 + sub * 4);
        const f32x4 l = *reinterpret_cast<const f32x4*>(lm + (size_t)i * 64 + sub * 4);
        f32x4 o;
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = g[e] / bv; acc += g[e] * l[e]; }
        *reinterpret_cast<f32x4*>(dA + (size_t)i * 64 + sub * 4) = o;
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 16);
        if (sub == 0) dBv[i] = -acc / bv;
    }
}

// T4b: everything that reaches rough, then back through the two softmaxes to p_up (out [n][pix][4], 2 used), and the part
// of dL/dr1 that comes through r1 * rough (g_r1 [pix][C], added to what is there)
__global__ void local_stats_bwd_kernel(const float* __restrict__ dA, const float* __restrict__ dBv, const float* __restrict__ g_cf,
                                       const int* __restrict__ amax, const float* __restrict__ g_u, const float* __restrict__ g_p2,
                                       const float* __restrict__ r1, const float* __restrict__ rough, const float* __restrict__ p_up,
                                       float* __restrict__ g_r1, float* __restrict__ g_pup, int obj_n, int h, int w, int C) {
    const int npix = h * w;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        const int x = i % w, y = i / w;
        float g_rough[MAX_OBJ], rg[MAX_OBJ];
        for (int n = 0; n < obj_n; ++n) { g_rough[n] = 0.f; rg[n] = rough[(size_t)n * npix + i]; }
        // box filters of dA (per channel) and dBv; max-pool routing of g_cf
        for (int n = 0; n < obj_n; ++n) {
            float sb = 0.f, gmax = 0.f;
            for (int dy = -3; dy <= 3; ++dy) {
                const int yy = y + dy;
                if ((unsigned)yy >= (unsigned)h) continue;
                for (int dx = -3; dx <= 3; ++dx) {
                    const int xx = x + dx;
                    if ((unsigned)xx >= (unsigned)w) continue;
                    const size_t pn = (size_t)n * npix + yy * w + xx;
                    sb += dBv[pn];
                    if (amax[pn] == i) gmax += g_cf[pn];
                }
            }
            g_rough[n] += sb / 49.f + gmax;
        }
        for (int c = 0; c < C; ++c) {
            const float rv = r1[(size_t)i * C + c];
            float gr1 = 0.f;
            for (int n = 0; n < obj_n; ++n) {
                float sa = 0.f;
                for (int dy = -3; dy <= 3; ++dy) {
                    const int yy = y + dy;
                    if ((unsigned)yy >= (unsigned)h) continue;
                    for (int dx = -3; dx <= 3; ++dx) {
                        const int xx = x + dx;
                        if ((unsigned)xx < (unsigned)w) sa += dA[((size_t)n * npix + yy * w + xx) * C + c];
                    }
                }
                sa /= 49.f;
                gr1 += rg[n] * sa;
                g_rough[n] += rv * sa;
            }
            g_r1[(size_t)i * C + c] += gr1;
        }
        // uncertainty: u = exp(1 - t1 / (t2 + 1e-8)) with (t1, t2) the two largest rough values (first maximum wins ties)
        int k1 = 0, k2 = -1;
        float t1 = -INFINITY, t2 = -INFINITY;
        for (int n = 0; n < obj_n; ++n) {
            if (rg[n] > t1) { t2 = t1; k2 = k1; t1 = rg[n]; k1 = n; }
            else if (rg[n] > t2) { t2 = rg[n]; k2 = n; }
        }
        const float u = expf(1.f - t1 / (t2 + 1e-8f));
        const float gu = g_u[i];
        g_rough[k1] += gu * (-u / (t2 + 1e-8f));
        if (k2 >= 0) g_rough[k2] += gu * (u * t1 / ((t2 + 1e-8f) * (t2 + 1e-8f)));
        // rough = softmax_k(rp); rp = sigmoid(p_up1 - p_up0)
        float dot = 0.f;
        for (int n = 0; n < obj_n; ++n) dot += g_rough[n] * rg[n];
        for (int n = 0; n < obj_n; ++n) {
            const size_t pn = (size_t)n * npix + i;
            const float g_rp = rg[n] * (g_rough[n] - dot);
            const float v0 = p_up[pn * 2], v1 = p_up[pn * 2 + 1];
            const float m = fmaxf(v0, v1);
            const float e0 = expf(v0 - m), e1 = expf(v1 - m);
            const float rp = e1 / (e0 + e1);
            const float d = g_rp * rp * (1.f - rp);
            g_pup[pn * 4] = g_p2[pn * 4] - d;
            g_pup[pn * 4 + 1] = g_p2[pn * 4 + 1] + d;
            g_pup[pn * 4 + 2] = 0.f;
            g_pup[pn * 4 + 3] = 0.f;
        }
    }
}

// The same for C = 64 with the channels across lanes (round 4): 16 lanes per pixel, one float4 of channels each, so the window
// reads of dA are coalesced 256-byte rows (the kernel above walks 64 channels x 49 taps per THREAD with a 256-byte stride between
// neighbouring threads: 2.0 ms per sample at 200 x 200); the 49 scalar taps of dBv / the max-pool routing are dealt to the 16
// lanes as well, and the per-pixel sums meet in a 16-lane butterfly (fixed order).
__global__ __launch_bounds__(256)
void local_stats_bwd64_kernel(const float* __restrict__ dA, const float* __restrict__ dBv, const float* __restrict__ g_cf,
                              const int* __restrict__ amax, const float* __restrict__ g_u, const float* __restrict__ g_p2,
                              const float* __restrict__ r1, const float* __restrict__ rough, const float* __restrict__ p_up,
                              float* __restrict__ g_r1, float* __restrict__ g_pup, int obj_n, int h, int w) {
    constexpr int C = 64;
    const int npix = h * w;
    const int l16 = threadIdx.x & 15;
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = i < npix;
    const int ii = live ? i : npix - 1;
    const int x = ii % w, y = ii / w;
    float g_rough[MAX_OBJ], rg[MAX_OBJ];
    for (int n = 0; n < obj_n; ++n) { g_rough[n] = 0.f; rg[n] = rough[(size_t)n * npix + ii]; }
    // the scalar taps: lane l takes window positions l, l + 16, l + 32, l + 48
    for (int n = 0; n < obj_n; ++n) {
        float part = 0.f;
        for (int t = l16; t < 49; t += 16) {
            const int yy = y + t / 7 - 3, xx = x + t % 7 - 3;
            if ((unsigned)yy >= (unsigned)h || (unsigned)xx >= (unsigned)w) continue;
            const size_t pn = (size_t)n * npix + yy * w + xx;
            part += dBv[pn] * (1.f / 49.f);
            if (amax[pn] == ii) part += g_cf[pn];
        }
        g_rough[n] = part;
    }
    // the channel taps: this lane's four channels
    const f32x4 rv = *reinterpret_cast<const f32x4*>(r1 + (size_t)ii * C + l16 * 4);
    f32x4 gr1 = {0.f, 0.f, 0.f, 0.f};
    for (int n = 0; n < obj_n; ++n) {
        f32x4 sa = {0.f, 0.f, 0.f, 0.f};
        for (int dy = -3; dy <= 3; ++dy) {
            const int yy = y + dy;
            if ((unsigned)yy >= (unsigned)h) continue;
            for (int dx = -3; dx <= 3; ++dx) {
                const int xx = x + dx;
                if ((unsigned)xx < (unsigned)w) sa += *reinterpret_cast<const f32x4*>(dA + ((size_t)n * npix + yy * w + xx) * C + l16 * 4);
            }
        }
        sa *= (1.f / 49.f);
        gr1 += rg[n] * sa;
        g_rough[n] += (rv[0] * sa[0] + rv[1] * sa[1]) + (rv[2] * sa[2] + rv[3] * sa[3]);
    }
    if (live) {
        f32x4* dst = reinterpret_cast<f32x4*>(g_r1 + (size_t)ii * C + l16 * 4);
        *dst = *dst + gr1;
    }
    for (int n = 0; n < obj_n; ++n)
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) g_rough[n] += __shfl_xor(g_rough[n], o, 64);
    if (!live || l16 != 0) return;
    int k1 = 0, k2 = -1;
    float t1 = -INFINITY, t2 = -INFINITY;
    for (int n = 0; n < obj_n; ++n) {
        if (rg[n] > t1) { t2 = t1; k2 = k1; t1 = rg[n]; k1 = n; }
        else if (rg[n] > t2) { t2 = rg[n]; k2 = n; }
    }
    const float u = expf(1.f - t1 / (t2 + 1e-8f));
    const float gu = g_u[ii];
    g_rough[k1] += gu * (-u / (t2 + 1e-8f));
    if (k2 >= 0) g_rough[k2] += gu * (u * t1 / ((t2 + 1e-8f) * (t2 + 1e-8f)));
    float dot = 0.f;
    for (int n = 0; n < obj_n; ++n) dot += g_rough[n] * rg[n];
    for (int n = 0; n < obj_n; ++n) {
        const size_t pn = (size_t)n * npix + ii;
        const float g_rp = rg[n] * (g_rough[n] - dot);
        const float v0 = p_up[pn * 2], v1 = p_up[pn * 2 + 1];
        const float m = fmaxf(v0, v1);
        const float e0 = expf(v0 - m), e1 = expf(v1 - m);
        const float rp = e1 / (e0 + e1);
        const float d = g_rp * rp * (1.f - rp);
        g_pup[pn * 4] = g_p2[pn * 4] - d;
        g_pup[pn * 4 + 1] = g_p2[pn * 4 + 1] + d;
        g_pup[pn * 4 + 2] = 0.f;
        g_pup[pn * 4 + 3] = 0.f;
    }
}

// ------------------------------------------------------------------ the training loss and its gradient
// train_video_seg.py:72-74: loss = CrossEntropyLoss(scores, label) + lu * uncertainty, scores = the logits segment returns
// [bs][obj][n], label [bs][n], uncertainty = AFB_URR.py:302-305 (vfn_segment_uncertainty_f32: mean over the batch of
// ||calc_uncertainty(softmax_obj(s))||_2 / sqrt(n), s = sigmoid(logit) = the decoder's probability).
constexpr int LOSS_BLOCKS = 64;

// stage 1: per (block, sample): sum of -log softmax_obj(logit)[label] and of u^2
__global__ __launch_bounds__(256)
void loss_partial_kernel(const float* __restrict__ logit, const int* __restrict__ label, int obj_n, int n,
                         float* __restrict__ part_ce, float* __restrict__ part_u2) {
    __shared__ float red[2][4];
    const int b = blockIdx.y;
    const float* src = logit + (size_t)b * obj_n * n;
    float ce = 0.f, u2 = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float z[MAX_OBJ], s[MAX_OBJ], zmax = -INFINITY, smax = -INFINITY;
        for (int k = 0; k < obj_n; ++k) {
            z[k] = src[(size_t)k * n + i];
            s[k] = 1.f / (1.f + expf(-z[k]));
            zmax = fmaxf(zmax, z[k]);
            smax = fmaxf(smax, s[k]);
        }
        float zden = 0.f, sden = 0.f;
        for (int k = 0; k < obj_n; ++k) { zden += expf(z[k] - zmax); s[k] = expf(s[k] - smax); sden += s[k]; }
        if (label) {
            const int lab = label[(size_t)b * n + i];
            ce += logf(zden) + zmax - z[lab];
        }
        float t1 = -1.f, t2 = -1.f;
        for (int k = 0; k < obj_n; ++k) {
            const float pk = s[k] / sden;
            if (pk > t1) { t2 = t1; t1 = pk; } else if (pk > t2) t2 = pk;
        }
        const float u = expf(1.f - t1 / (t2 + 1e-8f));
        u2 += u * u;
    }
    ce = wave_sum(ce);
    u2 = wave_sum(u2);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ce; red[1][threadIdx.x >> 6] = u2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part_ce[b * LOSS_BLOCKS + blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        part_u2[b * LOSS_BLOCKS + blockIdx.x] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// stage 2: out[0] = loss, out[1] = CE, out[2] = uncertainty, out[3 + b] = sqrt(sum u^2) of sample b
__global__ void loss_finish_kernel(const float* __restrict__ part_ce, const float* __restrict__ part_u2, int bs, int n, float lu,
                                   float* __restrict__ out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float ce = 0.f, unc = 0.f;
    for (int b = 0; b < bs; ++b) {
        float c = 0.f, q = 0.f;
        for (int j = 0; j < LOSS_BLOCKS; ++j) { c += part_ce[b * LOSS_BLOCKS + j]; q += part_u2[b * LOSS_BLOCKS + j]; }
        ce += c;
        out[3 + b] = sqrtf(q);
        unc += sqrtf(q) / sqrtf((float)n);
    }
    ce /= (float)bs * (float)n;
    unc /= (float)bs;
    out[0] = ce + lu * unc;
    out[1] = ce;
    out[2] = unc;
}

// dloss/dlogit.  The uncertainty term reaches the logit through s = sigmoid(logit) (ds/dlogit = s (1 - s); where the clamp of
// AFB_URR.py:309 is active that factor is below 1e-7 and the term is dropped with the rest of that pixel's gradient).
// label == NULL: no cross-entropy term (vfn_segment_uncertainty_backward_f32: the uncertainty's adjoint alone, for an autograd
// boundary where the criterion is the caller's); lu_dev != NULL: the factor in front of the uncertainty is read from device
// memory (dL/duncertainty as autograd hands it over: no host round trip); add != NULL: added to the result (dL/dscores arriving
// from the caller's criterion).
__global__ void loss_grad_kernel(const float* __restrict__ logit, const int* __restrict__ label, const float* __restrict__ stats,
                                 int bs, int obj_n, int n, float lu, float* __restrict__ grad,
                                 const float* __restrict__ lu_dev = nullptr, const float* __restrict__ add = nullptr) {
    const int b = blockIdx.y;
    if (lu_dev) lu = *lu_dev;
    const float* src = logit + (size_t)b * obj_n * n;
    float* dst = grad + (size_t)b * obj_n * n;
    const float norm = stats[3 + b];
    const float cu = norm > 0.f ? lu / ((float)bs * sqrtf((float)n) * norm) : 0.f;      // dU/du_i = cu * u_i
    const float cce = 1.f / ((float)bs * (float)n);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float z[MAX_OBJ], s[MAX_OBJ], e[MAX_OBJ], zmax = -INFINITY, smax = -INFINITY;
        for (int k = 0; k < obj_n; ++k) {
            z[k] = src[(size_t)k * n + i];
            s[k] = 1.f / (1.f + expf(-z[k]));
            zmax = fmaxf(zmax, z[k]);
            smax = fmaxf(smax, s[k]);
        }
        float zden = 0.f, sden = 0.f;
        for (int k = 0; k < obj_n; ++k) { zden += expf(z[k] - zmax); e[k] = expf(s[k] - smax); sden += e[k]; }
        int k1 = 0, k2 = -1;
        float t1 = -1.f, t2 = -1.f;
        for (int k = 0; k < obj_n; ++k) {
            const float pk = e[k] / sden;
            if (pk > t1) { t2 = t1; k2 = k1; t1 = pk; k1 = k; } else if (pk > t2) { t2 = pk; k2 = k; }
        }
        const float u = expf(1.f - t1 / (t2 + 1e-8f));
        float gP[MAX_OBJ];
        for (int k = 0; k < obj_n; ++k) gP[k] = 0.f;
        gP[k1] = cu * u * (-u / (t2 + 1e-8f));
        if (k2 >= 0) gP[k2] = cu * u * (u * t1 / ((t2 + 1e-8f) * (t2 + 1e-8f)));
        float dot = 0.f;
        for (int k = 0; k < obj_n; ++k) dot += gP[k] * (e[k] / sden);
        const int lab = label ? label[(size_t)b * n + i] : -1;
        for (int k = 0; k < obj_n; ++k) {
            const float P = e[k] / sden;
            const float g_s = P * (gP[k] - dot);
            float ce_g = label ? (expf(z[k] - zmax) / zden - (k == lab ? 1.f : 0.f)) * cce : 0.f;
            if (add) ce_g += add[(size_t)b * obj_n * n + (size_t)k * n + i];
            dst[(size_t)k * n + i] = ce_g + g_s * s[k] * (1.f - s[k]);
        }
    }
}

// ------------------------------------------------------------------ encoder pieces (ResNet trunks with frozen BatchNorm)
// zero insertion: out[n][2y][2x][c] = g[n][y][x][c], 0 elsewhere -- a stride-2 convolution's data gradient is the stride-1
// data-gradient convolution of this
__global__ void dilate2_kernel(const float* __restrict__ g, float* __restrict__ out, int N, int Ho, int Wo, int H, int W, int C) {
    const int c4n = C / 4;
    const size_t total = (size_t)N * H * W * c4n;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = i % c4n;
        size_t t = i / c4n;
        const int x = t % W; t /= W;
        const int y = t % H;
        const int n = t / H;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!(y & 1) && !(x & 1) && (y >> 1) < Ho && (x >> 1) < Wo)
            v = *reinterpret_cast<const f32x4*>(g + (((size_t)n * Ho + (y >> 1)) * Wo + (x >> 1)) * C + c4 * 4);
        *reinterpret_cast<f32x4*>(out + i * 4) = v;
    }
}

// eval-mode BatchNorm y = gamma * xhat + beta (xhat from the frozen running statistics): dbeta = sum_m g, dgamma = sum_m g * xhat
// with xhat = (y - idn - beta) / gamma, where y is what the forward stored (after the residual add `idn` and the ReLU: wherever the
// ReLU clipped, g is zero, so the clipped value is never used).  Stage 1 per block, stage 2 in block order.
__global__ void bn_grads_partial_kernel(const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ idn,
                                        const float* __restrict__ beta, const float* __restrict__ gamma, int M, int C,
                                        float* __restrict__ partial) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float b = beta[c], ig = 1.f / gamma[c];
        float sg = 0.f, sx = 0.f;
        for (int m = blockIdx.x; m < M; m += gridDim.x) {
            const float gv = g[(size_t)m * C + c];
            const float yv = y[(size_t)m * C + c] - (idn ? idn[(size_t)m * C + c] : 0.f);
            sg += gv;
            sx += gv * (yv - b) * ig;
        }
        partial[((size_t)blockIdx.x * 2) * C + c] = sg;
        partial[((size_t)blockIdx.x * 2 + 1) * C + c] = sx;
    }
}
__global__ void bn_grads_final_kernel(const float* __restrict__ partial, int nb, int C, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                      int accumulate = 0) {
    __shared__ float red[256];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    const float sg = final_sum(partial, nb, (size_t)2 * C, c, C, red);
    const float sx = final_sum(partial + C, nb, (size_t)2 * C, c, C, red);
    if (threadIdx.x < 32 && c < C) {
        dbeta[c] = accumulate ? dbeta[c] + sg : sg;
        dgamma[c] = accumulate ? dgamma[c] + sx : sx;
    }
}

// ---- one-launch forms of the two-stage column sums (round 4): every block publishes its partial row write-through (sc1 stores),
// drains, and one lane takes a ticket; the block that draws the last ticket adds the rows IN BLOCK ORDER (sc1 loads) -- the same
// sums in the same order as the two-launch form, without the second launch.  *counter is zero at rest.
__device__ __forceinline__ bool last_block_arrives(int* counter, int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int prev = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev == (int)gridDim.x - 1;
        if (last) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = last;
    }
    __syncthreads();
    return *flag != 0;
}
// write-through store / load past L1 and L2 (sc0 sc1) as ordinary buffer operations: relaxed atomic loads compile to the same
// instruction but are waited for one by one (128 dependent round trips in the last block: 60 us instead of 30 for the two launches)
struct WtBuf {
    __amdgpu_buffer_rsrc_t r;
    const float* base;
    __device__ __forceinline__ WtBuf(float* p) : r(__builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7ffffff0, 0x00020000)), base(p) {}
    __device__ __forceinline__ void store(const float* q, float v) const {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)((q - base) * sizeof(float)), 0, 17);
    }
    __device__ __forceinline__ float load(const float* q) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)((q - base) * sizeof(float)), 0, 17));
    }
};

// Thread layout of the fused column sums: for C < 256 the 256 threads of a block are C channels x (256 / C) row groups (a thread
// per channel alone left 3/4 of the block idle at C = 64); the row groups meet through LDS in group order.
__global__ __launch_bounds__(256)
void colsum_fused_kernel(const float* __restrict__ x, int M, int C, int ld, float* __restrict__ partial, int* counter,
                         float* __restrict__ out, int accumulate) {
    __shared__ int flag;
    __shared__ float red[256];
    const WtBuf wt(partial);
    const int cw = C < 256 ? C : 256, rg = 256 / cw;                  // C is a power of two or a multiple of 256 here; else rg = 1
    const int tx = threadIdx.x % cw, ty = threadIdx.x / cw;
    const bool shaped = (256 % cw) == 0;
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int c = c0 + (shaped ? tx : (int)threadIdx.x);
        float s = 0.f;
        if (c < C) {
            if (shaped) { for (int m = blockIdx.x * rg + ty; m < M; m += gridDim.x * rg) s += x[(size_t)m * ld + c]; }
            else { for (int m = blockIdx.x; m < M; m += gridDim.x) s += x[(size_t)m * ld + c]; }
        }
        if (shaped && rg > 1) {
            red[threadIdx.x] = s;
            __syncthreads();
            if (ty == 0) { for (int g = 1; g < rg; ++g) s += red[g * cw + tx]; }
            __syncthreads();
        }
        if (c < C && (!shaped || ty == 0)) wt.store(partial + (size_t)blockIdx.x * C + c, s);
    }
    if (!last_block_arrives(counter, &flag)) return;
    // the last block: the partial rows dealt to the row groups again (block order inside a group, groups in order)
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int c = c0 + (shaped ? tx : (int)threadIdx.x);
        float s = 0.f;
        if (c < C) {
            if (shaped) { for (int b = ty; b < (int)gridDim.x; b += rg) s += wt.load(partial + (size_t)b * C + c); }
            else { for (int b = 0; b < (int)gridDim.x; ++b) s += wt.load(partial + (size_t)b * C + c); }
        }
        if (shaped && rg > 1) {
            red[threadIdx.x] = s;
            __syncthreads();
            if (ty == 0) { for (int g = 1; g < rg; ++g) s += red[g * cw + tx]; }
            __syncthreads();
        }
        if (c < C && (!shaped || ty == 0)) out[c] = accumulate ? out[c] + s : s;
    }
}
__global__ __launch_bounds__(256)
void bn_grads_fused_kernel(const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ idn,
                           const float* __restrict__ beta, const float* __restrict__ gamma, int M, int C,
                           float* __restrict__ partial, int* counter, float* __restrict__ dgamma, float* __restrict__ dbeta,
                           int accumulate) {
    __shared__ int flag;
    __shared__ float red[2][256];
    const WtBuf wt(partial);
    const int cw = C < 256 ? C : 256, rg = 256 / cw;
    const int tx = threadIdx.x % cw, ty = threadIdx.x / cw;
    const bool shaped = (256 % cw) == 0;
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int c = c0 + (shaped ? tx : (int)threadIdx.x);
        float sg = 0.f, sx = 0.f;
        if (c < C) {
            const float b = beta[c], ig = 1.f / gamma[c];
            const int m0 = shaped ? blockIdx.x * rg + ty : blockIdx.x, dm = shaped ? gridDim.x * rg : gridDim.x;
            for (int m = m0; m < M; m += dm) {
                const float gv = g[(size_t)m * C + c];
                const float yv = y[(size_t)m * C + c] - (idn ? idn[(size_t)m * C + c] : 0.f);
                sg += gv;
                sx += gv * (yv - b) * ig;
            }
        }
        if (shaped && rg > 1) {
            red[0][threadIdx.x] = sg;
            red[1][threadIdx.x] = sx;
            __syncthreads();
            if (ty == 0) { for (int q = 1; q < rg; ++q) { sg += red[0][q * cw + tx]; sx += red[1][q * cw + tx]; } }
            __syncthreads();
        }
        if (c < C && (!shaped || ty == 0)) {
            wt.store(partial + ((size_t)blockIdx.x * 2) * C + c, sg);
            wt.store(partial + ((size_t)blockIdx.x * 2 + 1) * C + c, sx);
        }
    }
    if (!last_block_arrives(counter, &flag)) return;
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int c = c0 + (shaped ? tx : (int)threadIdx.x);
        float sg = 0.f, sx = 0.f;
        if (c < C) {
            const int q0 = shaped ? ty : 0, dq = shaped ? rg : 1;
            for (int q = q0; q < (int)gridDim.x; q += dq) {
                sg += wt.load(partial + ((size_t)q * 2) * C + c);
                sx += wt.load(partial + ((size_t)q * 2 + 1) * C + c);
            }
        }
        if (shaped && rg > 1) {
            red[0][threadIdx.x] = sg;
            red[1][threadIdx.x] = sx;
            __syncthreads();
            if (ty == 0) { for (int q = 1; q < rg; ++q) { sg += red[0][q * cw + tx]; sx += red[1][q * cw + tx]; } }
            __syncthreads();
        }
        if (c < C && (!shaped || ty == 0)) {
            dbeta[c] = accumulate ? dbeta[c] + sg : sg;
            dgamma[c] = accumulate ? dgamma[c] + sx : sx;
        }
    }
}

// ---- the slab form of the one-launch column sums (round 4, second pass).  The block-row form above leaves the tail to ONE block
// that walks nb partial rows of all C channels (the more blocks, the longer the tail: 128 blocks were faster than 256), and reads a
// dword per lane.  Here a block owns a 64-channel slab x a row block: 16 float4 columns x 16 row groups, four independent row loads
// in flight per lane; every slab has its own arrival counter, and the last block OF A SLAB adds that slab's <= 64 partial rows, 16
// row groups x <= 4 rows: one round trip.  Sums in a fixed order (rows of a group ascending, groups ascending, row blocks ascending).
// MODE 0: out[c] = sum_m x[m][c];   MODE 1: dbeta[c] = sum g, dgamma[c] = sum g * ((y - idn) - beta) / gamma
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
struct WtBuf4 {
    __amdgpu_buffer_rsrc_t r;
    const float* base;
    __device__ __forceinline__ WtBuf4(float* p) : r(__builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7ffffff0, 0x00020000)), base(p) {}
    __device__ __forceinline__ void store(const float* q, f32x4 v) const {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, (int)((q - base) * sizeof(float)), 0, 17);
    }
    __device__ __forceinline__ f32x4 load(const float* q) const {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)((q - base) * sizeof(float)), 0, 17));
    }
};
__device__ __forceinline__ bool last_of_slab(int* counter, int n, int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int prev = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev == n - 1;
        if (last) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = last;
    }
    __syncthreads();
    return *flag != 0;
}
template <int MODE>
__global__ __launch_bounds__(256)
void slab_sums_kernel(const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ idn,
                      const float* __restrict__ beta, const float* __restrict__ gamma, int M, int C, int ld,
                      float* __restrict__ partial, int* counters, float* __restrict__ out0, float* __restrict__ out1, int accumulate) {
    constexpr int NS = MODE ? 2 : 1;
    __shared__ int flag;
    __shared__ f32x4 red[NS][256];
    const WtBuf4 wt(partial);
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + tx * 4;
    const int nbr = gridDim.y;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 b = {0.f, 0.f, 0.f, 0.f}, ig = {1.f, 1.f, 1.f, 1.f};
    if (MODE) {
        b = *reinterpret_cast<const f32x4*>(beta + c);
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) ig[e] = 1.f / gm[e];
    }
    const int dm = nbr * 16;
    int m = blockIdx.y * 16 + ty;
    auto term = [&](const f32x4& gv, const f32x4& yv, const f32x4& iv) {
        s0 += gv;
        if (MODE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) s1[e] += gv[e] * ((yv[e] - iv[e]) - b[e]) * ig[e];
        }
    };
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (; m + 3 * dm < M; m += 4 * dm) {
        f32x4 gv[4], yv[4], iv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t o = (size_t)(m + u * dm) * ld + c;
            gv[u] = *reinterpret_cast<const f32x4*>(g + o);
            yv[u] = MODE ? *reinterpret_cast<const f32x4*>(y + o) : z;
            iv[u] = (MODE && idn) ? *reinterpret_cast<const f32x4*>(idn + o) : z;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) term(gv[u], yv[u], iv[u]);
    }
    for (; m < M; m += dm) {
        const size_t o = (size_t)m * ld + c;
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + o);
        const f32x4 yv = MODE ? *reinterpret_cast<const f32x4*>(y + o) : z;
        const f32x4 iv = (MODE && idn) ? *reinterpret_cast<const f32x4*>(idn + o) : z;
        term(gv, yv, iv);
    }
    red[0][threadIdx.x] = s0;
    if (MODE) red[NS - 1][threadIdx.x] = s1;
    __syncthreads();
    if (ty == 0) {
        for (int q = 1; q < 16; ++q) { s0 += red[0][q * 16 + tx]; if (MODE) s1 += red[NS - 1][q * 16 + tx]; }
        wt.store(partial + ((size_t)blockIdx.y * NS) * C + c, s0);
        if (MODE) wt.store(partial + ((size_t)blockIdx.y * NS + 1) * C + c, s1);
    }
    if (!last_of_slab(counters + blockIdx.x, nbr, &flag)) return;
    s0 = z; s1 = z;
    for (int q = ty; q < nbr; q += 16) {
        s0 += wt.load(partial + ((size_t)q * NS) * C + c);
        if (MODE) s1 += wt.load(partial + ((size_t)q * NS + 1) * C + c);
    }
    red[0][threadIdx.x] = s0;
    if (MODE) red[NS - 1][threadIdx.x] = s1;
    __syncthreads();
    if (ty == 0) {
        for (int q = 1; q < 16; ++q) { s0 += red[0][q * 16 + tx]; if (MODE) s1 += red[NS - 1][q * 16 + tx]; }
        f32x4* o0 = reinterpret_cast<f32x4*>(out0 + c);
        *o0 = accumulate ? *o0 + s0 : s0;
        if (MODE) {
            f32x4* o1 = reinterpret_cast<f32x4*>(out1 + c);
            *o1 = accumulate ? *o1 + s1 : s1;
        }
    }
}
// row blocks of the slab form: ~512 blocks in all, <= nb (the caller's partial buffer holds nb rows), <= 64 (one-round tail)
inline int slab_row_blocks(int M, int C, int nb) {
    int r = 512 / (C / 64);
    if (r > nb) r = nb;
    if (r > 64) r = 64;
    const int by_rows = (M + 15) / 16;
    if (r > by_rows) r = by_rows;
    return r < 1 ? 1 : r;
}
inline bool slab_ok(const void* a, const void* b, const void* c, const void* d, const void* e, int C, int ld) {
    auto al = [](const void* p) { return ((size_t)p & 15) == 0; };
    return C % 64 == 0 && C <= 64 * VFN_COLSUM_COUNTERS && ld % 4 == 0 && al(a) && al(b) && al(c) && al(d) && al(e);
}

// MaxPool2d(3, 2, 1) backwards: gx[n][y][x][c] = sum of g[n][yo][xo][c] over the output windows whose FIRST maximum (row-major, as
// PyTorch) is (y, x).  A gather: the <= 2 x 2 windows that contain the input pixel are re-evaluated.
// add (optional): a second gradient arriving at x (the decoder reads r1 too); relu_mask: x is a ReLU's output, the sum is masked
__global__ void maxpool3x3s2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ gx,
                                        int N, int H, int W, int C, int Ho, int Wo, const float* __restrict__ add, int relu_mask) {
    const size_t total = (size_t)N * H * W * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i % C;
        size_t t = i / C;
        const int xx = t % W; t /= W;
        const int yy = t % H;
        const int n = t / H;
        float acc = 0.f;
        for (int yo = (yy >= 1 ? (yy - 1 + 1) / 2 : 0); yo <= (yy + 1) / 2 && yo < Ho; ++yo)
            for (int xo = (xx >= 1 ? (xx - 1 + 1) / 2 : 0); xo <= (xx + 1) / 2 && xo < Wo; ++xo) {
                float best = -INFINITY;
                int by = -1, bx = -1;
                for (int dy = -1; dy <= 1; ++dy) {
                    const int y2 = 2 * yo + dy;
                    if ((unsigned)y2 >= (unsigned)H) continue;
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int x2 = 2 * xo + dx;
                        if ((unsigned)x2 >= (unsigned)W) continue;
                        const float v = x[(((size_t)n * H + y2) * W + x2) * C + c];
                        if (v > best) { best = v; by = y2; bx = x2; }
                    }
                }
                if (by == yy && bx == xx) acc += g[(((size_t)n * Ho + yo) * Wo + xo) * C + c];
            }
        if (add) acc += add[i];
        if (relu_mask && !(x[i] > 0.f)) acc = 0.f;
        gx[i] = acc;
    }
}

// The same, four channels per thread and one thread per 2 x 2 block of input pixels (round 5).  The kernel above re-evaluates up to four
// windows of nine scalar loads for every input ELEMENT (113 us for the query encoder's five 200 x 200 x 64 planes, 0.42 ms per training
// step on the dependent chain).  The four windows that can select one of the block's pixels -- output rows k, k + 1, columns j, j + 1 --
// cover a 5 x 5 input patch: it is loaded once (25 float4, clamped addresses, out-of-image = -inf), the four arg-max scans run on
// registers in PyTorch's order (row-major, first maximum wins), and the four pixels take the windows' gradients in the order of the
// kernel above (bit-identical sums).
__global__ __launch_bounds__(256)
void maxpool3x3s2_bwd4_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ gx,
                              int N, int H, int W, int C, int Ho, int Wo, const float* __restrict__ add, int relu_mask) {
    const int C4 = C >> 2, Hb = (H + 1) >> 1, Wb = (W + 1) >> 1;
    const size_t total = (size_t)N * Hb * Wb * C4;
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c4 = i % C4;
    size_t t = i / C4;
    const int j = t % Wb; t /= Wb;
    const int k = t % Hb;
    const int n = t / Hb;
    f32x4 P[5][5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const int yy = 2 * k - 1 + r;
        const bool rv = (unsigned)yy < (unsigned)H;
        const int yc = min(max(yy, 0), H - 1);
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const int xx = 2 * j - 1 + c;
            const bool ok = rv && (unsigned)xx < (unsigned)W;
            const int xc = min(max(xx, 0), W - 1);
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)n * H + yc) * W + xc) * C + c4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) P[r][c][e] = ok ? v[e] : -INFINITY;
        }
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) acc[u][v] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int yo = k + a, xo = j + b;
            const bool wv = yo < Ho && xo < Wo;
            const int yoc = min(yo, Ho - 1), xoc = min(xo, Wo - 1);
            f32x4 gv = *reinterpret_cast<const f32x4*>(g + (((size_t)n * Ho + yoc) * Wo + xoc) * C + c4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float best = -INFINITY;
                int pos = -1;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float v = P[2 * a + dy][2 * b + dx][e];
                        if (v > best) { best = v; pos = (2 * a + dy) * 5 + 2 * b + dx; }
                    }
                const float ge = wv ? gv[e] : 0.f;
#pragma unroll
                for (int u = a; u < 2; ++u)               // (window row a holds patch rows 2a .. 2a + 2: pixel row 1 + u needs u >= a)
#pragma unroll
                    for (int v = b; v < 2; ++v)
                        if (wv && pos == (1 + u) * 5 + 1 + v) acc[u][v][e] += ge;
            }
        }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int yy = 2 * k + u;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int xx = 2 * j + v;
            if (yy < H && xx < W) {
                const size_t o = (((size_t)n * H + yy) * W + xx) * C + c4 * 4;
                f32x4 r = acc[u][v];
                if (add) r += *reinterpret_cast<const f32x4*>(add + o);
                if (relu_mask) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (!(P[1 + u][1 + v][e] > 0.f)) r[e] = 0.f;
                }
                *reinterpret_cast<f32x4*>(gx + o) = r;
            }
        }
    }
}

// ------------------------------------------------------------------ memory read, backwards (training: the bank is one frame)
// P[b][q] = softmax over b of scale * S[b][q] (AFB_URR.py:144-145).  64 query columns per workgroup, the bank rows dealt to
// SM_R row groups (round 3 walked all B rows three times on ONE thread per column: 340 us at B = Q = 625); the groups' maxima
// and sums meet through LDS in group order (deterministic).
constexpr int SM_R = 16;
__global__ __launch_bounds__(64 * SM_R)
void softmax_cols_kernel(const float* __restrict__ S, int B, int Q, int ld, float scale, float* __restrict__ P) {
    __shared__ float red[SM_R][64];
    const int cx = threadIdx.x & 63, rgp = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + cx;
    const bool ok = q < Q;
    float m = -INFINITY;
    if (ok)
        for (int b = rgp; b < B; b += SM_R) m = fmaxf(m, S[(size_t)b * ld + q] * scale);
    red[rgp][cx] = m;
    __syncthreads();
    m = red[0][cx];
#pragma unroll
    for (int g = 1; g < SM_R; ++g) m = fmaxf(m, red[g][cx]);
    __syncthreads();
    float l = 0.f;
    if (ok)
        for (int b = rgp; b < B; b += SM_R) l += expf(S[(size_t)b * ld + q] * scale - m);
    red[rgp][cx] = l;
    __syncthreads();
    l = 0.f;
#pragma unroll
    for (int g = 0; g < SM_R; ++g) l += red[g][cx];
    if (ok)
        for (int b = rgp; b < B; b += SM_R) P[(size_t)b * ld + q] = expf(S[(size_t)b * ld + q] * scale - m) / l;
}
// dS[b][q] = scale * P (dP - sum_b' P dP)
__global__ __launch_bounds__(64 * SM_R)
void softmax_cols_bwd_kernel(const float* __restrict__ P, const float* __restrict__ dP, int B, int Q, int ld, float scale,
                             float* __restrict__ dS) {
    __shared__ float red[SM_R][64];
    const int cx = threadIdx.x & 63, rgp = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + cx;
    const bool ok = q < Q;
    float dot = 0.f;
    if (ok)
        for (int b = rgp; b < B; b += SM_R) dot += P[(size_t)b * ld + q] * dP[(size_t)b * ld + q];
    red[rgp][cx] = dot;
    __syncthreads();
    dot = 0.f;
#pragma unroll
    for (int g = 0; g < SM_R; ++g) dot += red[g][cx];
    if (ok)
        for (int b = rgp; b < B; b += SM_R) dS[(size_t)b * ld + q] = scale * P[(size_t)b * ld + q] * (dP[(size_t)b * ld + q] - dot);
}

// ------------------------------------------------------------------ AdamW (torch.optim.AdamW defaults' arithmetic)
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                             float lr, float b1, float b2, float omb1, float omb2, float eps, float wd, float bc1, float bc2) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float pv = p[i];
        const float gv = g[i];
        pv *= 1.f - lr * wd;                                           // decoupled weight decay
        const float mv = b1 * m[i] + omb1 * gv;
        const float vv = b2 * v[i] + omb2 * gv * gv;
        m[i] = mv;
        v[i] = vv;
        const float denom = sqrtf(vv) / sqrtf(bc2) + eps;
        p[i] = pv - (lr / bc1) * (mv / denom);
    }
}

}  // namespace

extern "C" int vfn_transpose_taps_f32(const float* x, int N, int H, int W, int C, int ld_x, int relu, int k, int stride, int pad,
                                      int Ho, int Wo, const float* colscale, float* out, int Mpad, void* stream) {
    if (!x || !out || N < 1 || H < 1 || W < 1 || C < 1 || ld_x < C || k < 1 || k > 7 || stride < 1 || pad < 0 || Ho < 1 || Wo < 1 ||
        Mpad < N * Ho * Wo) return VFN_ERR_ARG;
    const dim3 grid(cdiv(Mpad, 32), cdiv(C, 32), k * k);
    hipLaunchKernelGGL(transpose_taps_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, N, H, W, C, ld_x, relu, k, stride, pad, Ho, Wo,
                       colscale, out, Mpad);
    return vfn_check_launch();
}

extern "C" int vfn_colsum_f32(const float* x, int M, int C, int ld, float* partial, int nb, float* out, void* stream) {
    if (!x || !partial || !out || M < 1 || C < 1 || ld < C || nb < 1 || nb > 1024) return VFN_ERR_ARG;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, M, C, ld, partial);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, partial, nb, C, out);
    return vfn_check_launch();
}

// ... accumulating into `out` (a running gradient over the samples of a batch: no separate add launch)
extern "C" int vfn_colsum_acc_f32(const float* x, int M, int C, int ld, float* partial, int nb, float* out, int accumulate, int* counter,
                                  void* stream) {
    if (!x || !partial || !out || M < 1 || C < 1 || ld < C || nb < 1 || nb > 1024) return VFN_ERR_ARG;
    if (counter && slab_ok(x, partial, out, nullptr, nullptr, C, ld)) {
        hipLaunchKernelGGL(slab_sums_kernel<0>, dim3(C / 64, slab_row_blocks(M, C, nb)), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, M, C, ld, partial, counter, out, (float*)nullptr,
                           accumulate);
        return vfn_check_launch();
    }
    if (counter) {                             // one launch: the last block to arrive adds the partial rows
        hipLaunchKernelGGL(colsum_fused_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, M, C, ld, partial, counter, out, accumulate);
        return vfn_check_launch();
    }
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, M, C, ld, partial);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, partial, nb, C, out, accumulate);
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

// Decoder tail backwards, stage by stage (the host runs the local head's convolution gradients between T2 and T4):
//   vfn_tail_grad_o_f32        G [obj][H0][W0] -> g_o [obj][2h][2w][4] (zero-initialised by the caller; channels 0, 1 written)
//   vfn_tail_split_f32         g_p2 [obj][pix][4] -> g_q [obj][pix][32] (zero-initialised; channels 0, 1 written), g_cf, g_u
//   vfn_local_stats_backward_f32   g_lm, g_cf, g_u, g_p2 + forward tensors -> g_r1 (accumulated) and g_pup [obj][pix][4]
//                              scratch: dA [obj][pix][C], dBv [obj][pix], amax int[obj][pix]
extern "C" int vfn_tail_grad_o_f32(const float* G, const float* p_up, const float* unc, const float* conf, const float* q, float* g_o,
                                   int obj_n, int h, int w, int pad_top, int pad_left, int H0, int W0, void* stream) {
    if (!G || !p_up || !unc || !conf || !q || !g_o || obj_n < 1 || obj_n > MAX_OBJ) return VFN_ERR_ARG;
    if (pad_top + H0 > 2 * h || pad_left + W0 > 2 * w) return VFN_ERR_ARG;
    hipLaunchKernelGGL(tail_grad_o_kernel, dim3(grid_for((size_t)obj_n * H0 * W0)), dim3(256), 0, (hipStream_t)stream,
                       G, p_up, unc, conf, q, g_o, obj_n, h, w, pad_top, pad_left, H0, W0);
    return vfn_check_launch();
}

extern "C" int vfn_tail_split_f32(const float* g_p2, const float* unc, const float* conf, const float* q, float* g_q, float* g_cf,
                                  float* g_u, int obj_n, int npix, void* stream) {
    if (!g_p2 || !unc || !conf || !q || !g_q || !g_cf || !g_u || obj_n < 1 || obj_n > MAX_OBJ) return VFN_ERR_ARG;
    hipLaunchKernelGGL(tail_split_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, g_p2, unc, conf, q, g_q, g_cf, g_u,
                       obj_n, npix);
    return vfn_check_launch();
}

extern "C" int vfn_local_stats_backward_f32(const float* g_lm, const float* lm, const float* g_cf, const float* g_u, const float* g_p2,
                                            const float* r1, const float* rough, const float* p_up, float* dA, float* dBv, int* amax,
                                            float* g_r1, float* g_pup, int obj_n, int h, int w, int C, void* stream) {
    if (!g_lm || !lm || !g_cf || !g_u || !g_p2 || !r1 || !rough || !p_up || !dA || !dBv || !amax || !g_r1 || !g_pup) return VFN_ERR_ARG;
    if (obj_n < 1 || obj_n > MAX_OBJ || C < 1) return VFN_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int total = obj_n * h * w;
    hipLaunchKernelGGL(window_argmax_kernel, dim3(grid_for(total)), dim3(256), 0, s, rough, amax, obj_n, h, w);
    if (C == 64 && ((size_t)g_lm & 15) == 0 && ((size_t)lm & 15) == 0 && ((size_t)dA & 15) == 0)
        hipLaunchKernelGGL(local_ratio_bwd64_kernel, dim3(grid_for((size_t)total * 16)), dim3(256), 0, s, g_lm, lm, rough, dA, dBv, obj_n, h, w);
    else
        hipLaunchKernelGGL(local_ratio_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, g_lm, lm, rough, dA, dBv, obj_n, h, w, C);
    if (C == 64)
        hipLaunchKernelGGL(local_stats_bwd64_kernel, dim3(cdiv(h * w, 16)), dim3(256), 0, s, dA, dBv, g_cf, amax, g_u, g_p2, r1, rough, p_up,
                           g_r1, g_pup, obj_n, h, w);
    else
        hipLaunchKernelGGL(local_stats_bwd_kernel, dim3(grid_for(h * w)), dim3(256), 0, s, dA, dBv, g_cf, amax, g_u, g_p2, r1, rough, p_up,
                           g_r1, g_pup, obj_n, h, w, C);
    return vfn_check_launch();
}

// loss = CrossEntropyLoss(logit [bs][obj][n], label int32 [bs][n]) + lu * uncertainty (train_video_seg.py:72-74) and dloss/dlogit.
// partial: scratch 2 * bs * 64 floats; stats: 3 + bs floats (loss, CE, uncertainty, per-sample ||u||); grad (optional): [bs][obj][n]
extern "C" int vfn_segment_loss_f32(const float* logit, const int* label, int bs, int obj_n, int n, float lu, float* partial,
                                    float* stats, float* grad, void* stream) {
    if (!logit || !label || !partial || !stats || bs < 1 || obj_n < 2 || obj_n > MAX_OBJ || n < 1) return VFN_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    float* pce = partial;
    float* pu2 = partial + (size_t)bs * LOSS_BLOCKS;
    hipLaunchKernelGGL(loss_partial_kernel, dim3(LOSS_BLOCKS, bs), dim3(256), 0, s, logit, label, obj_n, n, pce, pu2);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, s, pce, pu2, bs, n, lu, stats);
    if (grad) hipLaunchKernelGGL(loss_grad_kernel, dim3(256, bs), dim3(256), 0, s, logit, label, stats, bs, obj_n, n, lu, grad);
    return vfn_check_launch();
}

// The uncertainty's adjoint on its own (an autograd boundary: ``scores, uncertainty = model.segment(...)``, the criterion is the
// caller's): grad = g_scores (optional, [bs][obj][n]) + *g_unc_dev * d uncertainty / d logit.  partial / stats as above.
extern "C" int vfn_segment_uncertainty_backward_f32(const float* logit, int bs, int obj_n, int n, const float* g_unc_dev,
                                                    const float* g_scores, float* partial, float* stats, float* grad, void* stream) {
    if (!logit || !g_unc_dev || !partial || !stats || !grad || bs < 1 || obj_n < 2 || obj_n > MAX_OBJ || n < 1) return VFN_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    float* pce = partial;
    float* pu2 = partial + (size_t)bs * LOSS_BLOCKS;
    hipLaunchKernelGGL(loss_partial_kernel, dim3(LOSS_BLOCKS, bs), dim3(256), 0, s, logit, (const int*)nullptr, obj_n, n, pce, pu2);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, s, pce, pu2, bs, n, 0.f, stats);
    hipLaunchKernelGGL(loss_grad_kernel, dim3(256, bs), dim3(256), 0, s, logit, (const int*)nullptr, stats, bs, obj_n, n, 0.f, grad,
                       g_unc_dev, g_scores);
    return vfn_check_launch();
}

extern "C" int vfn_dilate2_f32(const float* g, float* out, int N, int Ho, int Wo, int H, int W, int C, void* stream) {
    if (!g || !out || C % 4 || N < 1 || H < 2 * Ho - 1 || W < 2 * Wo - 1) return VFN_ERR_ARG;
    hipLaunchKernelGGL(dilate2_kernel, dim3(grid_for((size_t)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, g, out, N, Ho, Wo, H, W, C);
    return vfn_check_launch();
}

extern "C" int vfn_bn_param_grads_f32(const float* g, const float* y, const float* idn, const float* beta, const float* gamma, int M,
                                      int C, float* partial, int nb, float* dgamma, float* dbeta, void* stream) {
    if (!g || !y || !beta || !gamma || !partial || !dgamma || !dbeta || M < 1 || C < 1 || nb < 1 || nb > 1024) return VFN_ERR_ARG;
    hipLaunchKernelGGL(bn_grads_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, g, y, idn, beta, gamma, M, C, partial);
    hipLaunchKernelGGL(bn_grads_final_kernel, dim3(cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, partial, nb, C, dgamma, dbeta);
    return vfn_check_launch();
}

extern "C" int vfn_bn_param_grads_acc_f32(const float* g, const float* y, const float* idn, const float* beta, const float* gamma, int M,
                                          int C, float* partial, int nb, float* dgamma, float* dbeta, int accumulate, int* counter,
                                          void* stream) {
    if (!g || !y || !beta || !gamma || !partial || !dgamma || !dbeta || M < 1 || C < 1 || nb < 1 || nb > 1024) return VFN_ERR_ARG;
    if (counter && slab_ok(g, y, idn, partial, dgamma, C, C) && ((size_t)dbeta & 15) == 0 && ((size_t)beta & 15) == 0 && ((size_t)gamma & 15) == 0) {
        hipLaunchKernelGGL(slab_sums_kernel<1>, dim3(C / 64, slab_row_blocks(M, C, nb)), dim3(256), 0, (hipStream_t)stream, g, y, idn, beta, gamma,
                           M, C, C, partial, counter, dbeta, dgamma, accumulate);
        return vfn_check_launch();
    }
    if (counter) {
        hipLaunchKernelGGL(bn_grads_fused_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, g, y, idn, beta, gamma, M, C, partial, counter,
                           dgamma, dbeta, accumulate);
        return vfn_check_launch();
    }
    hipLaunchKernelGGL(bn_grads_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, g, y, idn, beta, gamma, M, C, partial);
    hipLaunchKernelGGL(bn_grads_final_kernel, dim3(cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, partial, nb, C, dgamma, dbeta, accumulate);
    return vfn_check_launch();
}

extern "C" int vfn_maxpool3x3s2_backward_f32(const float* x, const float* g, float* gx, int N, int H, int W, int C, const float* add,
                                             int relu_mask, void* stream) {
    if (!x || !g || !gx || N < 1 || H < 1 || W < 1 || C < 1) return VFN_ERR_ARG;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    if (C % 4 == 0 && (((size_t)x | (size_t)g | (size_t)gx | (size_t)add) & 15) == 0) {
        const size_t total = (size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
        hipLaunchKernelGGL(maxpool3x3s2_bwd4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, g, gx, N, H, W, C, Ho, Wo,
                           add, relu_mask);
        return vfn_check_launch();
    }
    hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3(grid_for((size_t)N * H * W * C)), dim3(256), 0, (hipStream_t)stream, x, g, gx, N, H, W, C, Ho, Wo,
                       add, relu_mask);
    return vfn_check_launch();
}

extern "C" int vfn_softmax_cols_f32(const float* S, int B, int Q, int ld, float scale, float* P, void* stream) {
    if (!S || !P || B < 1 || Q < 1 || ld < Q) return VFN_ERR_ARG;
    hipLaunchKernelGGL(softmax_cols_kernel, dim3(cdiv(Q, 64)), dim3(64 * SM_R), 0, (hipStream_t)stream, S, B, Q, ld, scale, P);
    return vfn_check_launch();
}

extern "C" int vfn_softmax_cols_backward_f32(const float* P, const float* dP, int B, int Q, int ld, float scale, float* dS, void* stream) {
    if (!P || !dP || !dS || B < 1 || Q < 1 || ld < Q) return VFN_ERR_ARG;
    hipLaunchKernelGGL(softmax_cols_bwd_kernel, dim3(cdiv(Q, 64)), dim3(64 * SM_R), 0, (hipStream_t)stream, P, dP, B, Q, ld, scale, dS);
    return vfn_check_launch();
}

// one AdamW step on n floats: step >= 1 (bias corrections 1 - beta^step); torch.optim.AdamW arithmetic (decoupled decay first)
extern "C" int vfn_adamw_f32(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2, double eps,
                             double weight_decay, int step, void* stream) {
    if (!p || !g || !m || !v || n < 1 || step < 1) return VFN_ERR_ARG;
    // 1 - beta and the bias corrections are formed in double on the host, as torch.optim forms its python scalars
    const float bc1 = (float)(1.0 - pow(beta1, (double)step)), bc2 = (float)(1.0 - pow(beta2, (double)step));
    const float omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (size_t)n, (float)lr,
                       (float)beta1, (float)beta2, omb1, omb2, (float)eps, (float)weight_decay, bc1, bc2);
    return vfn_check_launch();
}
