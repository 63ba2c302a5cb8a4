// Per-frame loop operators of test_video_seg.py:88-116 (planar NCHW fp32 / uint8 labels).
//
//   vfn_resize_bicubic_f32        TF.resize(frame, 480, BICUBIC)                 test_video_seg.py:88,107
//   vfn_resize_nearest_f32        TF.resize(first_mask, 480, NEAREST)            test_video_seg.py:89
//   vfn_softmax_objects_f32       F.softmax(score, dim=1)                        test_video_seg.py:109
//   vfn_resize_argmax_u8          TF.resize(pred_mask, ori_size, BICUBIC) -> argmax(dim=0) -> uint8   :114-115
//   vfn_postprocess_pred_u8       myutils.postprocessing_pred (host: cv2 CCL on the CPU in the reference,
//                                 myutils/data.py:17-37) -- largest 8-connected water blob, with the
//                                 reference's all-background -> all-ones quirk
//
// Bicubic = PyTorch upsample_bicubic2d, align_corners=False, no antialias (torchvision 0.9.2
// tensor path): src = (in/out)*(dst+0.5)-0.5, A=-0.75, 4 taps with border clamping, x first then y.
#include <vector>
#include "common.h"
#include "host/ccl_host.h"
#include "../../include/vfn_hip.h"

namespace {

constexpr int MAX_OBJ = 8;

struct Cubic { int i[4]; float w[4]; };

__device__ __forceinline__ float cc1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cc2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

__device__ __forceinline__ Cubic cubic_taps(int dst, float scale, int in_size) {
    const float A = -0.75f;
    const float real = scale * (dst + 0.5f) - 0.5f;
    const float fl = floorf(real);
    const float t = real - fl;
    const int i0 = (int)fl;
    Cubic c;
#pragma unroll
    for (int j = 0; j < 4; ++j) c.i[j] = max(min(i0 + j - 1, in_size - 1), 0);
    c.w[0] = cc2(t + 1.f, A);
    c.w[1] = cc1(t, A);
    c.w[2] = cc1(1.f - t, A);
    c.w[3] = cc2(2.f - t, A);
    return c;
}

__device__ __forceinline__ float bicubic_at(const float* plane, int Wi, const Cubic& cy, const Cubic& cx) {
    float rows[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float* r = plane + (size_t)cy.i[a] * Wi;
        float t = cx.w[0] * r[cx.i[0]];
        t += cx.w[1] * r[cx.i[1]];
        t += cx.w[2] * r[cx.i[2]];
        t += cx.w[3] * r[cx.i[3]];
        rows[a] = t;
    }
    float o = cy.w[0] * rows[0];
    o += cy.w[1] * rows[1];
    o += cy.w[2] * rows[2];
    o += cy.w[3] * rows[3];
    return o;
}

__global__ void resize_bicubic_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int Hi, int Wi,
                                      int Ho, int Wo) {
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    const size_t total = (size_t)Ho * Wo;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int y = i / Wo, x = i - (size_t)y * Wo;
        const Cubic cy = cubic_taps(y, sy, Hi), cx = cubic_taps(x, sx, Wi);
        for (int c = 0; c < C; ++c) out[(size_t)c * total + i] = bicubic_at(in + (size_t)c * Hi * Wi, Wi, cy, cx);
    }
}

__global__ void resize_nearest_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int Hi, int Wi,
                                      int Ho, int Wo) {
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    const size_t total = (size_t)Ho * Wo;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int y = i / Wo, x = i - (size_t)y * Wo;
        const int yi = min((int)floorf(y * sy), Hi - 1), xi = min((int)floorf(x * sx), Wi - 1);
        for (int c = 0; c < C; ++c) out[(size_t)c * total + i] = in[((size_t)c * Hi + yi) * Wi + xi];
    }
}

__global__ void softmax_objects_kernel(const float* __restrict__ score, float* __restrict__ prob, int obj_n, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float v[MAX_OBJ], m = -INFINITY, s = 0.f;
        for (int o = 0; o < obj_n; ++o) { v[o] = score[(size_t)o * n + i]; m = fmaxf(m, v[o]); }
        for (int o = 0; o < obj_n; ++o) { v[o] = expf(v[o] - m); s += v[o]; }
        for (int o = 0; o < obj_n; ++o) prob[(size_t)o * n + i] = v[o] / s;
    }
}

__global__ void resize_argmax_kernel(const float* __restrict__ prob, unsigned char* __restrict__ label, int obj_n,
                                     int Hi, int Wi, int Ho, int Wo) {
    const bool same = (Hi == Ho && Wi == Wo);          // bicubic at scale 1 is the identity (weights 0,1,0,0)
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    const size_t total = (size_t)Ho * Wo;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int y = i / Wo, x = i - (size_t)y * Wo;
        float best = -INFINITY;
        int arg = 0;
        if (same) {
            for (int o = 0; o < obj_n; ++o) {
                const float v = prob[(size_t)o * total + i];
                if (v > best) { best = v; arg = o; }
            }
        } else {
            const Cubic cy = cubic_taps(y, sy, Hi), cx = cubic_taps(x, sx, Wi);
            for (int o = 0; o < obj_n; ++o) {
                const float v = bicubic_at(prob + (size_t)o * Hi * Wi, Wi, cy, cx);
                if (v > best) { best = v; arg = o; }
            }
        }
        label[i] = (unsigned char)arg;
    }
}

inline int grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    return (int)(b < 8192 ? (b ? b : 1) : 8192);
}

}  // namespace

extern "C" int vfn_resize_bicubic_f32(const float* in, float* out, int C, int Hi, int Wi, int Ho, int Wo, void* stream) {
    if (!in || !out || C < 1) return VFN_ERR_ARG;
    hipLaunchKernelGGL(resize_bicubic_kernel, dim3(grid_for((size_t)Ho * Wo)), dim3(256), 0, (hipStream_t)stream,
                       in, out, C, Hi, Wi, Ho, Wo);
    return vfn_check_launch();
}

extern "C" int vfn_resize_nearest_f32(const float* in, float* out, int C, int Hi, int Wi, int Ho, int Wo, void* stream) {
    if (!in || !out || C < 1) return VFN_ERR_ARG;
    hipLaunchKernelGGL(resize_nearest_kernel, dim3(grid_for((size_t)Ho * Wo)), dim3(256), 0, (hipStream_t)stream,
                       in, out, C, Hi, Wi, Ho, Wo);
    return vfn_check_launch();
}

extern "C" int vfn_softmax_objects_f32(const float* score, float* prob, int obj_n, int n, void* stream) {
    if (!score || !prob || obj_n < 1 || obj_n > MAX_OBJ) return VFN_ERR_ARG;
    hipLaunchKernelGGL(softmax_objects_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, score, prob, obj_n, n);
    return vfn_check_launch();
}

extern "C" int vfn_resize_argmax_u8(const float* prob, unsigned char* label, int obj_n, int Hi, int Wi, int Ho, int Wo,
                                    void* stream) {
    if (!prob || !label || obj_n < 1) return VFN_ERR_ARG;
    hipLaunchKernelGGL(resize_argmax_kernel, dim3(grid_for((size_t)Ho * Wo)), dim3(256), 0, (hipStream_t)stream,
                       prob, label, obj_n, Hi, Wi, Ho, Wo);
    return vfn_check_launch();
}

// (host buffers; documented in host/ccl_host.h)
extern "C" int vfn_postprocess_pred_u8(const unsigned char* pred, int H, int W, unsigned char* out) {
    return vfn_host::postprocess_pred_u8(pred, H, W, out) == 0 ? VFN_OK : VFN_ERR_ARG;
}
