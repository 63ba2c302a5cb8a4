// HBM-bound decoder pointwise / window kernels (NHWC fp32).
//
//   vfn_upsample2x_add      Refine: m = s + interpolate(pm, x2, bilinear)          AFB_URR.py:124
//   vfn_rough_uncertainty   interpolate(pred2) -> softmax[:,1] -> object softmax   AFB_URR.py:214-219
//                           -> calc_uncertainty (top-2 ratio)                        myutils/data.py:40-46
//   vfn_local_hpass/vpass   r1*rough, 7x7 avg-pools, divide, 7x7 max-pool            AFB_URR.py:226-229
//   vfn_final_logits        p + unc*(conf*q) -> interpolate x2 -> softmax[:,1]      AFB_URR.py:233-237
//                           -> clamp -> logit -> un-pad crop                         AFB_URR.py:300,309-316
//
// Bilinear x2, align_corners=False (PyTorch): src=(dst+0.5)/2-0.5 clamped at 0,
// i0=floor(src), i1=min(i0+1,in-1), l1=src-i0, l0=1-l1;
// out = lh0*(lw0*a + lw1*b) + lh1*(lw0*c + lw1*d).
#include "common.h"
#include "../../include/vfn_hip.h"

namespace {

struct Lerp { int i0, i1; float l0, l1; };

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

// out_lp (optional): the split-bf16 image of the result (of its ReLU with lp_relu), for a bf16x3 consumer (common.h)
__global__ void upsample2x_add_kernel(const float* __restrict__ s, const float* __restrict__ pm,
                                      float* __restrict__ out, int N, int h, int w, int C, int s_bcast,
                                      void* __restrict__ out_lp, int lp_relu) {
    const int c4n = C / 4;
    const int hi = h / 2, wi = w / 2;
    const size_t total = (size_t)N * h * w * c4n;
    for (size_t i = vfn_xcd_block(blockIdx.x, gridDim.x) * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = i % c4n;
        size_t t = i / c4n;
        const int x = t % w; t /= w;
        const int y = t % h;
        const int n = t / h;
        const Lerp ly = lerp2x(y, hi), lx = lerp2x(x, wi);
        const float* base = pm + (size_t)n * hi * wi * C + c4 * 4;
        const f32x4 a = *reinterpret_cast<const f32x4*>(base + ((size_t)ly.i0 * wi + lx.i0) * C);
        const f32x4 b = *reinterpret_cast<const f32x4*>(base + ((size_t)ly.i0 * wi + lx.i1) * C);
        const f32x4 c = *reinterpret_cast<const f32x4*>(base + ((size_t)ly.i1 * wi + lx.i0) * C);
        const f32x4 d = *reinterpret_cast<const f32x4*>(base + ((size_t)ly.i1 * wi + lx.i1) * C);
        const size_t so = ((size_t)(s_bcast ? 0 : n) * h * w + (size_t)y * w + x) * C + c4 * 4;
        const f32x4 sv = *reinterpret_cast<const f32x4*>(s + so);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o[k] = sv[k] + (ly.l0 * (lx.l0 * a[k] + lx.l1 * b[k]) + ly.l1 * (lx.l0 * c[k] + lx.l1 * d[k]));
        *reinterpret_cast<f32x4*>(out + i * 4) = o;
        if (out_lp) vfn_store_lp4(out_lp, i / c4n, C, c4 * 4, o, lp_relu);
    }
}

constexpr int MAX_OBJ = 8;

// p: [obj][h][w][2] (1/4 res) -> p_up [obj][2h][2w][2], rough [obj][2h][2w], unc [2h][2w]
__global__ void rough_unc_kernel(const float* __restrict__ p, float* __restrict__ p_up, float* __restrict__ rough,
                                 float* __restrict__ unc, int obj_n, int h, int w) {
    const int H = 2 * h, W = 2 * w;
    const int total = H * W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int y = i / W, x = i - y * W;
        const Lerp ly = lerp2x(y, h), lx = lerp2x(x, w);
        float r[MAX_OBJ];
        float rmax = -INFINITY;
        for (int n = 0; n < obj_n; ++n) {
            const float* base = p + (size_t)n * h * w * 2;
            const float2 a = *reinterpret_cast<const float2*>(base + ((size_t)ly.i0 * w + lx.i0) * 2);
            const float2 b = *reinterpret_cast<const float2*>(base + ((size_t)ly.i0 * w + lx.i1) * 2);
            const float2 c = *reinterpret_cast<const float2*>(base + ((size_t)ly.i1 * w + lx.i0) * 2);
            const float2 d = *reinterpret_cast<const float2*>(base + ((size_t)ly.i1 * w + lx.i1) * 2);
            const float v0 = ly.l0 * (lx.l0 * a.x + lx.l1 * b.x) + ly.l1 * (lx.l0 * c.x + lx.l1 * d.x);
            const float v1 = ly.l0 * (lx.l0 * a.y + lx.l1 * b.y) + ly.l1 * (lx.l0 * c.y + lx.l1 * d.y);
            *reinterpret_cast<float2*>(p_up + ((size_t)n * total + i) * 2) = make_float2(v0, v1);
            const float m = fmaxf(v0, v1);
            const float e0 = expf(v0 - m), e1 = expf(v1 - m);
            r[n] = e1 / (e0 + e1);                       // softmax(p, dim=1)[:, 1]
            rmax = fmaxf(rmax, r[n]);
        }
        float sum = 0.f;
        for (int n = 0; n < obj_n; ++n) { r[n] = expf(r[n] - rmax); sum += r[n]; }
        float top1 = -INFINITY, top2 = -INFINITY;
        for (int n = 0; n < obj_n; ++n) {
            const float v = r[n] / sum;                  // object-level softmax
            rough[(size_t)n * total + i] = v;
            if (v > top1) { top2 = top1; top1 = v; } else if (v > top2) top2 = v;
        }
        unc[i] = expf(1.f - top1 / (top2 + 1e-8f));     // calc_uncertainty
    }
}

// horizontal 7-tap pass: hs[obj][y][x][C] = sum_dx r1[y][x+dx][c]*rough[obj][y][x+dx]
//                        hr[obj][y][x]   = sum_dx rough ; hm = max_dx rough (in-bounds only)
__global__ void local_hpass_kernel(const float* __restrict__ r1, const float* __restrict__ rough,
                                   float* __restrict__ hs, float* __restrict__ hr, float* __restrict__ hm,
                                   int obj_n, int h, int w, int C) {
    const int c4n = C / 4;
    const size_t total = (size_t)h * w * c4n;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = i % c4n;
        const size_t pix = i / c4n;
        const int x = pix % w;
        const int y = pix / w;
        f32x4 v[7];
        bool ok[7];
#pragma unroll
        for (int d = 0; d < 7; ++d) {
            const int xx = x + d - 3;
            ok[d] = (unsigned)xx < (unsigned)w;
            v[d] = ok[d] ? *reinterpret_cast<const f32x4*>(r1 + ((size_t)y * w + xx) * C + c4 * 4)
                         : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int n = 0; n < obj_n; ++n) {
            const float* rg = rough + (size_t)n * h * w + (size_t)y * w;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            float sr = 0.f, mr = -INFINITY;
#pragma unroll
            for (int d = 0; d < 7; ++d) {
                if (ok[d]) {
                    const float g = rg[x + d - 3];
                    acc += v[d] * g;
                    sr += g;
                    mr = fmaxf(mr, g);
                }
            }
            *reinterpret_cast<f32x4*>(hs + ((size_t)n * h * w + pix) * C + c4 * 4) = acc;
            if (c4 == 0) { hr[(size_t)n * h * w + pix] = sr; hm[(size_t)n * h * w + pix] = mr; }
        }
    }
}

// vertical pass + divide; writes r1_local [obj][h][w][C] and conf [obj][h][w].  (torch.cat([r1, r1_local]),
// AFB_URR.py:231, is never materialised: local_convFM is applied to the two halves separately.)
__global__ void local_vpass_kernel(const float* __restrict__ hs,
                                   const float* __restrict__ hr, const float* __restrict__ hm,
                                   float* __restrict__ lm, float* __restrict__ conf,
                                   int obj_n, int h, int w, int C) {
    const int c4n = C / 4;
    const size_t total = (size_t)obj_n * h * w * c4n;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = i % c4n;
        size_t t = i / c4n;
        const int x = t % w; t /= w;
        const int y = t % h;
        const int n = t / h;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float sr = 0.f, mr = -INFINITY;
#pragma unroll
        for (int d = 0; d < 7; ++d) {
            const int yy = y + d - 3;
            if ((unsigned)yy < (unsigned)h) {
                const size_t pp = (size_t)n * h * w + (size_t)yy * w + x;
                acc += *reinterpret_cast<const f32x4*>(hs + pp * C + c4 * 4);
                sr += hr[pp];
                mr = fmaxf(mr, hm[pp]);
            }
        }
        const float den = sr / 49.f + 1e-8f;                 // AvgPool2d(7,1,3), count_include_pad
        f32x4 loc;
#pragma unroll
        for (int k = 0; k < 4; ++k) loc[k] = (acc[k] / 49.f) / den;
        const size_t pix = (size_t)y * w + x;
        *reinterpret_cast<f32x4*>(lm + ((size_t)n * h * w + pix) * C + c4 * 4) = loc;
        if (c4 == 0) conf[(size_t)n * h * w + pix] = mr;
    }
}

// Fused window statistics (AFB_URR.py:226-229): r1_local = avg7x7(r1 * rough) / (avg7x7(rough) + 1e-8), r1_conf =
// max7x7(rough), for every object -- one pass over r1, no horizontal-pass scratch in HBM.  A workgroup owns a strip of
// 16 columns x YR rows (thread = column x channel quad) and walks down it: per image row the strip's 22-pixel segment of
// r1 goes through LDS once (loaded a row ahead), every thread forms its 7-tap horizontal sums for each object, and a
// 7-row register ring turns them into the 7x7 sums (added in the order rows y-3 .. y+3, columns x-3 .. x+3, like the
// two-pass kernels above).  HBM traffic: r1 once (+ halo) in, r1_local once out.
constexpr int LS_W = 16;                 // strip width (pixels)
constexpr int LS_C4 = 16;                // channel quads (C = 64)
constexpr int LS_SEG = LS_W + 6;         // pixels of one staged row segment

template <int K>
__global__ __launch_bounds__(256)
void local_stats_fused_kernel(const float* __restrict__ r1, const float* __restrict__ rough, float* __restrict__ lm,
                              float* __restrict__ conf, int h, int w, int yr) {
    __shared__ __attribute__((aligned(16))) float s_r1[2][LS_SEG][LS_C4 * 4];
    __shared__ float s_rg[2][K][LS_SEG + 2];
    const int tid = threadIdx.x;
    const int c4 = tid & 15, x = tid >> 4;                       // 16 x 16
    const int strips = (w + LS_W - 1) / LS_W;
    const int strip = blockIdx.x % strips, ychunk = blockIdx.x / strips;
    const int x0 = strip * LS_W, y0 = ychunk * yr;
    const int y1 = min(h, y0 + yr);                               // output rows [y0, y1)
    const size_t plane = (size_t)h * w;
    // staging role: 22 x 16 float4 of r1 + K x 22 floats of rough per row; thread t loads float4 t and t + 256
    f32x4 st[2];
    float sg[K];
    auto load_row = [&](int yy) {
        const bool row_ok = (unsigned)yy < (unsigned)h;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = tid + 256 * j;
            st[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (e < LS_SEG * LS_C4) {
                const int px = e / LS_C4, cc = e - px * LS_C4;
                const int gx = x0 - 3 + px;
                if (row_ok && (unsigned)gx < (unsigned)w) st[j] = *reinterpret_cast<const f32x4*>(r1 + ((size_t)yy * w + gx) * 64 + cc * 4);
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            sg[k] = -1.f;                                         // marks "outside the image" (rough is a probability >= 0)
            if (tid < LS_SEG) {
                const int gx = x0 - 3 + tid;
                if (row_ok && (unsigned)gx < (unsigned)w) sg[k] = rough[(size_t)k * plane + (size_t)yy * w + gx];
            }
        }
    };
    auto store_row = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = tid + 256 * j;
            if (e < LS_SEG * LS_C4) *reinterpret_cast<f32x4*>(&s_r1[buf][e / LS_C4][(e % LS_C4) * 4]) = st[j];
        }
        if (tid < LS_SEG)
#pragma unroll
            for (int k = 0; k < K; ++k) s_rg[buf][k][tid] = sg[k];
    };
    f32x4 ring[K][7];
    float rsum[K][7], rmax[K][7];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < 7; ++j) { ring[k][j] = f32x4{0.f, 0.f, 0.f, 0.f}; rsum[k][j] = 0.f; rmax[k][j] = -INFINITY; }

    const int ya = y0 - 3, yb = y1 + 3;                           // input rows [ya, yb)
    load_row(ya);
    store_row(0);
    __syncthreads();
    for (int base = ya; base < yb; base += 7) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {                            // statically indexed ring slot j
            const int yy = base + j;
            if (yy >= yb) break;
            const int buf = (yy - ya) & 1;
            if (yy + 1 < yb) load_row(yy + 1);                    // next row: global -> registers behind this row's math
            // horizontal pass of row yy (zero / -inf where the row or the tap is outside the image)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                float sr = 0.f, mr = -INFINITY;
#pragma unroll
                for (int d = 0; d < 7; ++d) {
                    const float g = s_rg[buf][k][x + d];
                    if (g >= 0.f) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(&s_r1[buf][x + d][c4 * 4]);
                        acc += v * g;
                        sr += g;
                        mr = fmaxf(mr, g);
                    }
                }
                ring[k][j] = acc; rsum[k][j] = sr; rmax[k][j] = mr;
            }
            // the window centred on row yo = yy - 3 is complete: rows yo-3 .. yo+3 sit in slots j+1 .. j+7 (mod 7)
            const int yo = yy - 3;
            if (yo >= y0 && yo < y1 && x0 + x < w) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    float sr = 0.f, mr = -INFINITY;
#pragma unroll
                    for (int d = 1; d <= 7; ++d) {
                        const int sl = (j + d) % 7;
                        acc += ring[k][sl]; sr += rsum[k][sl]; mr = fmaxf(mr, rmax[k][sl]);
                    }
                    const float den = sr / 49.f + 1e-8f;         // AvgPool2d(7,1,3), count_include_pad
                    f32x4 loc;
#pragma unroll
                    for (int e = 0; e < 4; ++e) loc[e] = (acc[e] / 49.f) / den;
                    const size_t pix = (size_t)yo * w + x0 + x;
                    *reinterpret_cast<f32x4*>(lm + ((size_t)k * plane + pix) * 64 + c4 * 4) = loc;
                    if (c4 == 0) conf[(size_t)k * plane + pix] = mr;
                }
            }
            if (yy + 1 < yb) {
                store_row(buf ^ 1);
                __syncthreads();
            }
        }
    }
}

// pred2 / local_pred2 (3x3, two filters; AFB_URR.py:195,202,213,234) as a tap GEMM + gather: z[p][tap*2+o] =
// sum_c w[o][c][tap] * relu(x[p][c]) is a 1x1 convolution with 18 (padded to 20) filters that reads x exactly once on the
// matrix cores; this kernel then adds the nine shifted taps: out[p][o] = bias[o] + sum_tap z[p + tap][tap*2+o].
__global__ void pred2_gather_kernel(const float* __restrict__ z, const float* __restrict__ bias, float* __restrict__ out,
                                    int N, int h, int w, int ldz) {
    const size_t total = (size_t)N * h * w;
    const float b0 = bias[0], b1 = bias[1];
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(z), 0, (int)(total * ldz * 4), 0x00020000);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = i % w;
        size_t t = i / w;
        const int y = t % h;
        const int n = t / h;
        // (branch-free since round 5: the nine taps are raw buffer loads -- a tap outside the image has an out-of-range offset and reads 0 --
        // requested together; the first form tested each tap and waited for each load.  Same terms in the same order.)
        float2 v[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + dy - 1;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = x + dx - 1;
                const bool ok = (unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w;
                v[dy * 3 + dx] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(
                    rz, ok ? (int)(((((size_t)n * h + yy) * w + xx) * ldz + (dy * 3 + dx) * 2) * 4) : 0x7ffffff0, 0, 0));
            }
        }
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { a0 += v[k].x; a1 += v[k].y; }
        *reinterpret_cast<float2*>(out + i * 2) = make_float2(a0 + b0, a1 + b1);
    }
}

// score[obj][H0][W0] = logit(clamp(softmax(up2(p_up + unc*(conf*q)))[1]))
__global__ void final_logits_kernel(const float* __restrict__ p_up, const float* __restrict__ unc,
                                    const float* __restrict__ conf, const float* __restrict__ q,
                                    float* __restrict__ score, int obj_n, int h, int w,
                                    int pad_top, int pad_left, int H0, int W0) {
    const size_t total = (size_t)obj_n * H0 * W0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x0 = i % W0;
        size_t t = i / W0;
        const int y0 = t % H0;
        const int n = t / H0;
        const int y = y0 + pad_top, x = x0 + pad_left;       // coordinates in the padded frame (2h x 2w)
        const Lerp ly = lerp2x(y, h), lx = lerp2x(x, w);
        float v[2][4];
        const int ys[2] = {ly.i0, ly.i1}, xs[2] = {lx.i0, lx.i1};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const size_t pix = (size_t)ys[a] * w + xs[b];
                const size_t pn = (size_t)n * h * w + pix;
                const float2 pv = *reinterpret_cast<const float2*>(p_up + pn * 2);
                const float2 qv = *reinterpret_cast<const float2*>(q + pn * 2);
                const float u = unc[pix], cf = conf[pn];
                v[0][a * 2 + b] = pv.x + u * (cf * qv.x);
                v[1][a * 2 + b] = pv.y + u * (cf * qv.y);
            }
        const float o0 = ly.l0 * (lx.l0 * v[0][0] + lx.l1 * v[0][1]) + ly.l1 * (lx.l0 * v[0][2] + lx.l1 * v[0][3]);
        const float o1 = ly.l0 * (lx.l0 * v[1][0] + lx.l1 * v[1][1]) + ly.l1 * (lx.l0 * v[1][2] + lx.l1 * v[1][3]);
        const float m = fmaxf(o0, o1);
        const float e0 = expf(o0 - m), e1 = expf(o1 - m);
        float s = e1 / (e0 + e1);
        s = fminf(fmaxf(s, 1e-7f), 1.f - 1e-7f);
        score[i] = logf(s / (1.f - s));
    }
}

// Training-branch scalar of AFB_URR.segment (AFB_URR.py:302-305): per sample
//   u = calc_uncertainty(softmax_over_objects(score))   (myutils/data.py:40-46: exp(1 - top1 / (top2 + 1e-8)))
//   ||u||_2 / sqrt(H W), then the mean over the batch.
// `logit` holds the logits segment() returns; the probabilities are recovered as sigmoid(logit) (they were clamped to
// [1e-7, 1 - 1e-7] before the logit -- below the f32 resolution of the result).  Two deterministic stages: block partial
// sums of u^2 in a fixed order, then one thread per sample adds the partials in order.
constexpr int UNC_BLOCKS = 64;
__global__ __launch_bounds__(256)
void uncertainty_partial_kernel(const float* __restrict__ logit, int obj_n, int n, float* __restrict__ partial) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const float* src = logit + (size_t)b * obj_n * n;
    float acc = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float s[MAX_OBJ], mx = -INFINITY;
        for (int k = 0; k < obj_n; ++k) { s[k] = 1.f / (1.f + expf(-src[(size_t)k * n + i])); mx = fmaxf(mx, s[k]); }
        float den = 0.f;
        for (int k = 0; k < obj_n; ++k) { s[k] = expf(s[k] - mx); den += s[k]; }
        float t1 = -1.f, t2 = -1.f;
        for (int k = 0; k < obj_n; ++k) {
            const float pk = s[k] / den;
            if (pk > t1) { t2 = t1; t1 = pk; } else if (pk > t2) t2 = pk;
        }
        const float u = expf(1.f - t1 / (t2 + 1e-8f));
        acc += u * u;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[b * UNC_BLOCKS + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void uncertainty_finish_kernel(const float* __restrict__ partial, int bs, int n, float* __restrict__ out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float mean = 0.f;
    for (int b = 0; b < bs; ++b) {
        float ssq = 0.f;
        for (int j = 0; j < UNC_BLOCKS; ++j) ssq += partial[b * UNC_BLOCKS + j];
        mean += sqrtf(ssq) / sqrtf((float)n);
    }
    out[0] = mean / (float)bs;
}

inline int grid_for(size_t total, int block = 256, int cap = 8192) {
    size_t b = (total + block - 1) / block;
    return (int)(b < (size_t)cap ? (b ? b : 1) : cap);
}

}  // namespace

extern "C" int vfn_upsample2x_add_nhwc_f32(const float* s, const float* pm, float* out, int N, int h, int w, int C,
                                           int s_bcast, void* stream) {
    if (!s || !pm || !out || C % 4 || h % 2 || w % 2) return VFN_ERR_ARG;
    const size_t total = (size_t)N * h * w * (C / 4);
    hipLaunchKernelGGL(upsample2x_add_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       s, pm, out, N, h, w, C, s_bcast, (void*)nullptr, 0);
    return vfn_check_launch();
}

extern "C" int vfn_upsample2x_add_lp_nhwc_f32(const float* s, const float* pm, float* out, void* out_lp, int lp_relu,
                                              int N, int h, int w, int C, int s_bcast, void* stream) {
    if (!s || !pm || !out || C % 4 || h % 2 || w % 2 || (out_lp && C % 32)) return VFN_ERR_ARG;
    const size_t total = (size_t)N * h * w * (C / 4);
    hipLaunchKernelGGL(upsample2x_add_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       s, pm, out, N, h, w, C, s_bcast, out_lp, lp_relu);
    return vfn_check_launch();
}

extern "C" int vfn_rough_uncertainty_f32(const float* p, float* p_up, float* rough, float* unc, int obj_n, int h, int w,
                                         void* stream) {
    if (!p || !p_up || !rough || !unc || obj_n < 2 || obj_n > MAX_OBJ) return VFN_ERR_ARG;
    hipLaunchKernelGGL(rough_unc_kernel, dim3(grid_for((size_t)4 * h * w)), dim3(256), 0, (hipStream_t)stream,
                       p, p_up, rough, unc, obj_n, h, w);
    return vfn_check_launch();
}

extern "C" int vfn_local_hpass_f32(const float* r1, const float* rough, float* hs, float* hr, float* hm, int obj_n,
                                   int h, int w, int C, void* stream) {
    if (!r1 || !rough || !hs || !hr || !hm || C % 4) return VFN_ERR_ARG;
    hipLaunchKernelGGL(local_hpass_kernel, dim3(grid_for((size_t)h * w * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       r1, rough, hs, hr, hm, obj_n, h, w, C);
    return vfn_check_launch();
}

extern "C" int vfn_local_vpass_f32(const float* hs, const float* hr, const float* hm, float* lm,
                                   float* conf, int obj_n, int h, int w, int C, void* stream) {
    if (!hs || !hr || !hm || !lm || !conf || C % 4) return VFN_ERR_ARG;
    hipLaunchKernelGGL(local_vpass_kernel, dim3(grid_for((size_t)obj_n * h * w * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, hs, hr, hm, lm, conf, obj_n, h, w, C);
    return vfn_check_launch();
}

extern "C" int vfn_final_logits_f32(const float* p_up, const float* unc, const float* conf, const float* q, float* score,
                                    int obj_n, int h, int w, int pad_top, int pad_left, int H0, int W0, void* stream) {
    if (!p_up || !unc || !conf || !q || !score) return VFN_ERR_ARG;
    if (pad_top + H0 > 2 * h || pad_left + W0 > 2 * w) return VFN_ERR_ARG;
    hipLaunchKernelGGL(final_logits_kernel, dim3(grid_for((size_t)obj_n * H0 * W0)), dim3(256), 0, (hipStream_t)stream,
                       p_up, unc, conf, q, score, obj_n, h, w, pad_top, pad_left, H0, W0);
    return vfn_check_launch();
}

extern "C" int vfn_segment_uncertainty_f32(const float* logit, int bs, int obj_n, int n, float* partial, float* out,
                                           void* stream) {
    if (!logit || !partial || !out || bs < 1 || obj_n < 2 || obj_n > MAX_OBJ || n < 1) return VFN_ERR_ARG;
    hipLaunchKernelGGL(uncertainty_partial_kernel, dim3(UNC_BLOCKS, bs), dim3(256), 0, (hipStream_t)stream, logit, obj_n, n, partial);
    hipLaunchKernelGGL(uncertainty_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, bs, n, out);
    return vfn_check_launch();
}

extern "C" int vfn_local_stats_f32(const float* r1, const float* rough, float* lm, float* conf, int obj_n, int h, int w,
                                   int C, void* stream) {
    if (!r1 || !rough || !lm || !conf || C != 64 || obj_n < 1 || obj_n > 4 || h < 1 || w < 1) return VFN_ERR_ARG;
    const int strips = (w + LS_W - 1) / LS_W;
    // rows per workgroup: the walk down a strip is a dependent chain (one barrier per row), so the kernel wants many
    // short strips in flight -- >= 3 workgroups per CU when the image allows -- at the price of re-reading the 6 halo rows
    // (from L2) per strip
    int yr = 24;
    while (yr > 8 && strips * ((h + yr - 1) / yr) < 768) yr -= 4;
    const dim3 grid(strips * ((h + yr - 1) / yr));
    hipStream_t s = (hipStream_t)stream;
    switch (obj_n) {
        case 1: hipLaunchKernelGGL(local_stats_fused_kernel<1>, grid, dim3(256), 0, s, r1, rough, lm, conf, h, w, yr); break;
        case 2: hipLaunchKernelGGL(local_stats_fused_kernel<2>, grid, dim3(256), 0, s, r1, rough, lm, conf, h, w, yr); break;
        case 3: hipLaunchKernelGGL(local_stats_fused_kernel<3>, grid, dim3(256), 0, s, r1, rough, lm, conf, h, w, yr); break;
        default: hipLaunchKernelGGL(local_stats_fused_kernel<4>, grid, dim3(256), 0, s, r1, rough, lm, conf, h, w, yr); break;
    }
    return vfn_check_launch();
}

extern "C" int vfn_pred2_gather_f32(const float* z, const float* bias, float* out, int N, int h, int w, int ldz, void* stream) {
    if (!z || !bias || !out || ldz < 18 || ldz % 2 || (long long)N * h * w * ldz * 4 >= 0x7fffff00LL) return VFN_ERR_ARG;
    hipLaunchKernelGGL(pred2_gather_kernel, dim3(grid_for((size_t)N * h * w)), dim3(256), 0, (hipStream_t)stream, z, bias, out, N, h, w, ldz);
    return vfn_check_launch();
}
