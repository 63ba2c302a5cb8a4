// Input / output side of the loop, so that a frame crosses PCIe as
// uint8 in both directions:
//   vfn_to_tensor_u8     torchvision ToTensor of the decoded frame (Video_DS.__getitem__, Water_DS.py:131-139):
//                        uint8 HWC -> float32 CHW, x / 255 (IEEE division, bit-identical to tensor.div(255))
//   vfn_overlay_u8       myutils.add_overlay + the uint8 conversion of save_overlay (myutils/data.py:56-84) on the
//                        device: RGB uint8 HWC overlay image from the float frame and the label map.
#include "common.h"
#include "../../include/vfn_hip.h"

namespace {

__global__ void to_tensor_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, int H, int W) {
    const int n = H * W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned char* px = src + (size_t)i * 3;
        dst[i] = (float)px[0] / 255.0f;
        dst[n + i] = (float)px[1] / 255.0f;
        dst[2 * (size_t)n + i] = (float)px[2] / 255.0f;
    }
}

__global__ void label_min_kernel(const unsigned char* __restrict__ mask, int* __restrict__ lo, int n) {
    int m = 255;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = min(m, (int)mask[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMin(lo, m);
}

// add_overlay walks the ids present in ascending order and skips the smallest one (`for i in ids[1:]`): pixels of
// id i get img*alpha + (1-alpha)*colour_i, then the 4-neighbourhood contour of id i (dilation minus the region) is
// painted black.  Later ids overwrite earlier ones, which per pixel of label L collapses to:
//   a 4-neighbour with a label > max(L, lowest id)  -> black
//   else L > lowest id                              -> blended colour
//   else                                            -> the frame pixel
// Arithmetic as numpy does it: uint8 frame = trunc(frame*255) (f32 product), blend in f64, truncation to uint8.
__global__ void overlay_kernel(const float* __restrict__ frame, const unsigned char* __restrict__ mask,
                               const unsigned char* __restrict__ palette, const int* __restrict__ lo_p,
                               unsigned char* __restrict__ out, int H, int W, double alpha, double cscale) {
    const int n = H * W;
    const int lo = *lo_p;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = i / W, x = i - y * W;
        const int L = mask[i];
        int hi = L > lo ? L : lo;
        bool black = false;
        if (y > 0 && mask[i - W] > hi) black = true;
        if (y < H - 1 && mask[i + W] > hi) black = true;
        if (x > 0 && mask[i - 1] > hi) black = true;
        if (x < W - 1 && mask[i + 1] > hi) black = true;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned char px = (unsigned char)(frame[(size_t)c * n + i] * 255.0f);
            unsigned char v = px;
            if (black) v = 0;
            else if (L > lo) v = (unsigned char)((double)px * alpha + (1.0 - alpha) * ((double)palette[L * 3 + c] * cscale));
            out[(size_t)i * 3 + c] = v;
        }
    }
}

}  // namespace

extern "C" int vfn_to_tensor_u8(const unsigned char* src, float* dst, int H, int W, void* stream) {
    if (!src || !dst || H < 1 || W < 1) return VFN_ERR_ARG;
    const int n = H * W;
    const int blocks = cdiv(n, 256) < 2048 ? cdiv(n, 256) : 2048;
    hipLaunchKernelGGL(to_tensor_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, H, W);
    return vfn_check_launch();
}

extern "C" int vfn_overlay_u8(const float* frame, const unsigned char* mask, const unsigned char* palette, int* scratch,
                              unsigned char* out, int H, int W, double alpha, double cscale, void* stream) {
    if (!frame || !mask || !palette || !scratch || !out || H < 1 || W < 1) return VFN_ERR_ARG;
    const int n = H * W;
    const int blocks = cdiv(n, 256) < 2048 ? cdiv(n, 256) : 2048;
    hipStream_t s = (hipStream_t)stream;
    hipMemsetAsync(scratch, 0x7f, sizeof(int), s);
    hipLaunchKernelGGL(label_min_kernel, dim3(blocks < 256 ? blocks : 256), dim3(256), 0, s, mask, scratch, n);
    hipLaunchKernelGGL(overlay_kernel, dim3(blocks), dim3(256), 0, s, frame, mask, palette, scratch, out, H, W, alpha, cscale);
    return vfn_check_launch();
}
