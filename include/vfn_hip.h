/* vfn_hip.h -- C ABI of libvfn_hip.so: the gfx950 (MI355X) kernels behind the
 * V-FloodNet video-segmentation hot path.
 *
 * The reference (xmlyqing00/V-FloodNet) has no FFI of its own: its hot path is
 * Python calling torch / torch_scatter / torchvision / cv2 operators.  Each entry
 * point below replaces one such operator call site (cited as file:line relative
 * to the reference root); INTEGRATION.md shows the ctypes binding a maintainer
 * adds on the reference side.
 *
 * Conventions
 *   - plain C: raw device pointers, ints, floats; no torch / C++ types.
 *   - every launcher returns 0 (VFN_OK) or a non-zero status; it enqueues on
 *     `stream` (a hipStream_t passed as void*), never synchronises and never
 *     allocates.  All buffers are caller-owned device memory.
 *   - activations are NHWC fp32 ("pixel-major"); bank entries are entry-major
 *     ([entry][dim]); the NCHW views the reference API exposes are strided views
 *     of these buffers made by the Python host.
 *   - one host thread per process / one process per GPU.
 */
#ifndef VFN_HIP_H
#define VFN_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ version */
int vfn_abi_version(void);

/* ------------------------------------------------------------------ conv (implicit GEMM, f32 MFMA)
 * Replaces nn.Conv2d (+ eval BatchNorm2d, + ReLU, + residual add) at
 *   AFB_URR.py:20-30   ResBlock.conv1/conv2 (ReLU applied to the input: relu_in)
 *   AFB_URR.py:106,109 KeyValue.Key / .Value (one GEMM, 640 filters)
 *   AFB_URR.py:117-126 Refine.convFS / ResFS / ResMM
 *   AFB_URR.py:191-202 Decoder.convFM / pred2 / local_convFM / local_ResMM / local_pred2
 *   torchvision Bottleneck conv1/conv2/conv3/downsample behind AFB_URR.py:43-47,73-77
 * y[m, c] = act( (sum_k x[m,k] w[c,k]) * scale[c] + shift[c] + res[m, c] )
 */
typedef struct vfn_conv_desc {
    const float* in;      /* NHWC [N,H,W,*], pixel stride in_ld floats (>= Cin, multiple of 4) */
    const float* w;       /* [cout_pad][KH*KW*Cin], K ordered (kh,kw,cin); rows >= Cout are zero */
    const float* scale;   /* [Cout] or NULL (=1) */
    const float* shift;   /* [Cout] or NULL (=0) */
    const float* res;     /* optional residual [M, res_ld] or NULL */
    float* out;           /* [M, out_ld] */
    int N, H, W, Cin, in_ld;
    int Ho, Wo, Cout, cout_pad, out_ld, res_ld;
    int KH, KW, stride, pad;
    int relu_in, relu_out;
    int M;                /* N*Ho*Wo */
} vfn_conv_desc;

int vfn_conv_cfg_count(void);
int vfn_conv_cfg_tile(int cfg, int* bm, int* bn);
int vfn_conv2d_nhwc_f32(const vfn_conv_desc* d, int cfg, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VFN_HIP_H */
