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

/* ------------------------------------------------------------------ version
 * vfn_abi_version() == VFN_ABI_VERSION of the header the binding was written against, and
 * vfn_sizeof_desc(which) == sizeof of the binding's own struct: checked when the library is loaded. */
#define VFN_ABI_VERSION 12
enum { VFN_DESC_CONV = 0, VFN_DESC_STEM = 1, VFN_DESC_BANKSCAN = 2, VFN_DESC_MEMREAD = 3, VFN_DESC_BANK = 4, VFN_DESC_WGRAD = 5,
       VFN_DESC_REFRESH_FILTER = 6, VFN_DESC_REFRESH_EPILOGUE = 7, VFN_DESC_GATHER = 8 };
int vfn_abi_version(void);
int vfn_sizeof_desc(int which);

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
    int ksplit;           /* <= 1: single pass.  > 1: output tiles with index >= split_from are cut along K into
                             ksplit slices (one workgroup each); their raw partial sums go to `partial` and a
                             second kernel reduces them in fixed order and applies the epilogue.  split_from = 0
                             cuts every tile (layers with fewer tiles than CUs); split_from = a multiple of 256
                             cuts only the last, partial round of tiles so that all 256 CUs finish together */
    int split_from;       /* tile index (multiple of the number of filter tiles), used when ksplit > 1 */
    int res_mod;          /* > 0: the residual is shared by the images of the batch: row m reads res[m % res_mod]
                             (a term computed once for all objects, e.g. the query-value half of convFM) */
    float* partial;       /* [ksplit][M - m_start][Cout] workspace, m_start = first row of tile split_from */
    int* tile_counters;   /* one int per split tile, zero at rest: the slice workgroup that arrives last reduces the
                             tile inside the same launch (agent-scope release/acquire); NULL: a second kernel reduces */
    int w_packed;         /* vfn_conv2d_nhwc_bf16 / _bf16x3 only: 1 = w holds the filters already converted, 128 bytes
                             per filter row and K tile in (kh,kw,cin) order -- bf16: [cout_pad][K] bf16 (tiles of 64);
                             bf16x3: [cout_pad][K/32][hi 32 bf16 | lo 32 bf16] -- so they are staged without
                             conversion; 0 = w is the f32 [cout_pad][K] array and is converted on the fly */
    /* --- the activations' split-bf16 image (vfn_conv2d_nhwc_bf16x3 only; ABI 8) ---------------------------------------
     * A tensor's image has the SAME size and pixel stride as the f32 tensor: per pixel and 32-channel block 128 bytes =
     * [32 hi bf16 | 32 lo bf16], x = hi + lo with hi = bf16(x), lo = bf16(x - hi) -- exactly one LDS row of the kernel's
     * A tile, so a consumer stages it with plain 16-byte copies and no conversion work.  A producer writes the image of
     * its result (optionally of relu(result): the consumer's "ReLU on the input" moves here) from its epilogue. */
    int in_lp;            /* 1: `in` points at such an image (relu_in must be 0; same in_ld; Cin multiple of 32) */
    int out_lp_relu;      /* the image written to out_lp is that of max(y, 0) */
    void* out_lp;         /* optional: image of y (pixel stride out_ld * 4 bytes; Cout, out_ld multiples of 32 / 4);
                             `out` may then be NULL when no consumer wants the f32 tensor */
    /* --- backward passes (f32; ABI 8): a data-gradient convolution runs this same kernel over the flipped, transposed
     * filters; the ReLU that preceded the forward convolution is undone in the epilogue -------------------------------- */
    const float* mask;    /* optional [M, mask_ld]: y = (mask[m, c] > 0 ? acc * scale + shift : 0) + res -- the gradient of
                             conv(relu(x)) w.r.t. x with mask = x, plus the gradient arriving over the skip connection */
    int mask_ld;
    int mask_after;       /* 1: the mask is applied AFTER the residual add: y = mask > 0 ? acc * scale + shift + res : 0 (a block's
                             input gradient = its two branches summed, then the ReLU that produced that input) */
    /* --- batched filters (f32; ABI 10): the Winograd-domain GEMMs of vfn_conv_winograd_* multiply every transform component
     * with its own filter matrix in ONE launch ---------------------------------------------------------------------------- */
    int w_batch_rows;     /* > 0: output rows [b * w_batch_rows, (b + 1) * w_batch_rows) use filter bank b = w + b * cout_pad * K
                             floats; a multiple of the tile height of the configuration used; 1x1 problems; 0: one filter bank */
    /* --- K rotation (ABI 12; f32, LDS-tiled configurations, 1x1 problems) ------------------------------------------------ */
    int k_rot;            /* 1: the workgroup of output-row tile mt walks its K tiles from tile mt % nk and wraps (a fixed, launch-
                             independent order per tile: results are reproducible, the summation order differs from k_rot = 0).
                             K tile k of a pixel-major operand is the same 128-byte column of every 1-KB row, so workgroups that
                             start together would all pull the same byte column -- the same few memory channels -- at once */
} vfn_conv_desc;

int vfn_conv_cfg_count(void);
int vfn_conv_cfg_tile(int cfg, int* bm, int* bn);
/* tile configuration cfg: workgroup tile bm x bn, wm x wn waves, dma = 0 register-staged / 2 LDS-DMA ring;
 * the kernel it launches is conv_igemm_kernel<bm, bn, wm, wn, MODE> (conv_igemm_dma_kernel<bm, bn, wm, wn, dma>) */
int vfn_conv_cfg_info(int cfg, int* bm, int* bn, int* wm, int* wn, int* dma);
/* K groups per workgroup of configuration cfg: 1 for the plain ones; > 1: split-K inside the workgroup --
 * that many copies of the wm x wn wave grid each take a slice of K of the same output tile and the partial sums are
 * added through LDS in group order (conv_igemm_wk_kernel<bm, bn, wm, wn, wk>); ksplit must be <= 1 with these.  0: no such cfg */
int vfn_conv_cfg_wk(int cfg);
/* K tiles a workgroup multiplies between two barriers: 1, or 2 (configurations with a 4-tile register prefetch and four
 * LDS buffers, conv_igemm_wk_kernel<bm, bn, wm, wn, wk, 4, 2, mode>; ksplit <= 1).  0: no such cfg */
int vfn_conv_cfg_tpb(int cfg);
/* Kind of tile configuration cfg (ABI 10): 0 = LDS-tiled (conv_igemm_kernel / _wk_kernel / _dma_kernel), 1 = wave-autonomous
 * (conv_direct_kernel: operands straight into the MFMA registers, no LDS staging; f32 only; ksplit / split_from as for kind 0,
 * tile_counters must be NULL), 2 = stream-K (conv_streamk_kernel: a persistent launch in which every wave takes an equal share
 * of the layer's (output tile, K tile) pairs and cut tiles are finished inside the launch in K order; f32 only; ksplit <= 1;
 * `partial` must point at VFN_CONV_SK_WS_FLOATS floats and `tile_counters` at VFN_CONV_SK_MAX_TILES ints that are zero at rest,
 * neither shared with a launch that may run concurrently).  -1: no such cfg */
#define VFN_CONV_SK_WS_FLOATS (16 * 1024 * 1024)
#define VFN_CONV_SK_MAX_TILES 16384
int vfn_conv_cfg_kind(int cfg);
/* kernel instantiation behind a configuration of kind 1 / 2 as rocprofv3 prints it, e.g. "conv_direct_kernel<2, 2, 1, 1, 4, 2>"
 * (kind 0: bench.py derives it from vfn_conv_cfg_info); NUL-terminated into buf[0 .. n).  0 ok */
int vfn_conv_cfg_name(int cfg, char* buf, int n);
int vfn_conv2d_nhwc_f32(const vfn_conv_desc* d, int cfg, void* stream);
/* Same convolution with both operands rounded to bf16 (nearest-even) as they are staged into LDS and multiplied
 * on v_mfma_f32_32x32x16_bf16 with f32 accumulation; tensors stay f32 in HBM.  Cin must be a multiple of 64;
 * register-staged tile configurations only (cfg 0-10, 17, 19); split-K slices are 64-channel tiles.
 * For BASELINE.json configs C3 / C5 (the reference itself has no reduced-precision path). */
int vfn_conv2d_nhwc_bf16(const vfn_conv_desc* d, int cfg, void* stream);
/* "bf16x3": every operand is split into two bf16 (x = hi + lo, 16 significant bits) as it is staged and each product
 * is hi*hi + hi*lo + lo*hi on the bf16 matrix cores, f32 accumulation: relative error ~2^-16 per product (bf16: 2^-9,
 * f32: 2^-24).  Cin multiple of 32, split-K slices are 32-channel tiles (as the f32 kernel); cfg as for _bf16. */
int vfn_conv2d_nhwc_bf16x3(const vfn_conv_desc* d, int cfg, void* stream);

/* ------------------------------------------------------------------ Winograd F(4x4, 3x3) around the matrix kernels (ABI 10)
 * A 3x3 / stride-1 / pad-1 nn.Conv2d (the decoder's, AFB_URR.py:20-30,114-127,191-195) as 36 GEMMs in the transform domain:
 *   vfn_winograd_input_f32    V [36][rows_pad][C] = B^T d B per 4x4 output tile (6x6 input patch, zeros outside the image;
 *                             relu: max(d, 0) first); tiles in (n, ty, tx) order, vfn_winograd_tiles of them, rows_pad >= that
 *   the GEMMs                 vfn_conv2d_nhwc_f32 on in = V (one "image" of 36 * rows_pad pixels), w = U [36][cout_pad][C] with
 *                             U[6i+j] = (G g G^T)[i][j], w_batch_rows = rows_pad, out = M [36][rows_pad][Cout], no epilogue
 *   vfn_winograd_output_f32   out[n][y][x][co] = act((A^T M A) * scale + shift + res) -- the convolution's own epilogue
 * (Lavin & Gray's matrices; the same identity cuDNN uses for the reference's convolutions.) */
int vfn_winograd_tiles(int N, int H, int W);
int vfn_winograd_input_f32(const float* x, int N, int H, int W, int C, int ld_x, int relu, float* V, int rows_pad, void* stream);
int vfn_winograd_output_f32(const float* Mb, int rows_pad, int N, int H, int W, int Cout, const float* scale, const float* shift,
                            const float* res, int res_ld, int res_mod, int relu_out, float* out, int out_ld, void* stream);
/* (ABI 11) the output transform with the epilogue of a data-gradient convolution (vfn_conv_desc.mask / mask_after): zero where
 * mask <= 0, before (mask_after = 0) or after the residual is added -- loss.backward() through a 3x3 convolution is the same
 * convolution over the flipped, transposed filters, so it takes the same transform-domain route (backward.py) */
int vfn_winograd_output_masked_f32(const float* Mb, int rows_pad, int N, int H, int W, int Cout, const float* res, int res_ld,
                                   const float* mask, int mask_ld, int mask_after, float* out, int out_ld, void* stream);
/* (ABI 11) the weight gradient of a 3x3 / stride-1 / pad-1 convolution in the transform domain -- the transposition of the forward
 * algorithm, dW = G^T [ sum_tiles (B^T d B) (.) (A dY A^T) ] G, a quarter of the direct form's multiplies:
 *   vfn_winograd_input_f32 (above) V[xi][tile][ci]; vfn_winograd_gy_f32  Z[xi][tile][co] = (A dY A^T)[xi] of the 4x4 tiles of
 *   gy [N,H,W,ld] (C channels used); vfn_conv_wgrad_f32 with k = 1, batch = 36 sums them over the tiles into dU [36][Cout][Cin];
 *   vfn_winograd_dw_f32  dw [Cout][3][3][Cin] (+)= rowscale[co] * (G^T dU G). */
/* (ABI 12) the transform-domain GEMMs as ONE PERSISTENT launch: M [comps][rows_pad][Cout] = V [comps][rows_pad][C] x U [comps][cout_pad][C]^T.
 * A workgroup walks a list of (component, row tile, filter tile) units as one uninterrupted K loop -- the next unit's operand tiles are
 * requested while the current one multiplies, the raw accumulators leave through stores straight from the registers -- instead of one
 * workgroup per unit with a cold prologue and an LDS epilogue around 4-32 K tiles.  Same products in the same order as the batched-filter
 * launch of vfn_conv2d_nhwc_f32 (w_batch_rows) without split-K: bit-identical.  cfg: 0 = 128x128 tiles, 1 = 64x128, 2 = 128x64 (8 waves),
 * 3 = 64x64 (4 waves); + 4: operand tiles requested two K tiles ahead.  wgs: workgroups (0 = 512 = two per CU).  rows_pad % tile height
 * == 0, cout_pad >= Cout rounded up to the tile width, C % 32 == 0, every operand below 2 GiB. */
int vfn_winograd_gemm_f32(const float* V, const float* U, float* Mb, int comps, int rows_pad, int C, int Cout, int cout_pad, int cfg,
                          int wgs, void* stream);
/* (ABI 12) a 1x1 / stride-1 / pad-0 convolution descriptor (+ scale / shift, residual incl. res_mod, ReLU in / out) through the same
 * persistent kernel: one GEMM [M pixels x Cin] x [Cin x Cout] whose workgroups walk their output tiles as one K loop and apply the
 * epilogue from the accumulator registers (the trunk's conv1 / conv3 / downsample have 2-8 K tiles per output tile).  Products and
 * order as vfn_conv2d_nhwc_f32 without split-K.  cfg / wgs as above; taps, strides, masks, operand images, split-K: VFN_ERR_ARG. */
int vfn_conv1x1_persistent_f32(const vfn_conv_desc* d, int cfg, int wgs, void* stream);
/* (ABI 12) Winograd layers of the plain-bf16 mode (BASELINE configs C3 / C5; the reference itself has no reduced-precision path):
 *   vfn_winograd_input_bf16   as vfn_winograd_input_f32 with V [36][rows_pad][C] written as bf16 (f32 transform, one round-to-nearest-even)
 *   vfn_winograd_gemm_bf16    as vfn_winograd_gemm_f32 on bf16 V and bf16 U [36][cout_pad][C] (v_mfma_f32_32x32x16_bf16, f32 accumulate), M in f32
 * C a multiple of 64; the output transform is vfn_winograd_output_f32. */
int vfn_winograd_input_bf16(const float* x, int N, int H, int W, int C, int ld_x, int relu, void* V, int rows_pad, void* stream);
int vfn_winograd_gemm_bf16(const void* V, const void* U, float* Mb, int comps, int rows_pad, int C, int Cout, int cout_pad, int cfg,
                           int wgs, void* stream);
int vfn_winograd_gy_f32(const float* gy, int N, int H, int W, int C, int ld, float* Z, int rows_pad, void* stream);
int vfn_winograd_dw_f32(const float* dU, int Cout, int Cin, const float* rowscale, float* dw, int accumulate, void* stream);

/* ------------------------------------------------------------------ encoder stems
 * vfn_stem_conv7x7_f32: pad_divide_by (myutils/data.py:132-149) + (x-mean)/std + conv1
 * (+ conv1_m(mask) + conv1_o((1-mask).clamp(0,1))) + bn1 + relu, AFB_URR.py:53-57 / :83-87
 * (called from AFB_URR.memorize :259-266 and .segment :279-281).
 * vfn_maxpool3x3s2_nhwc_f32: nn.MaxPool2d(3,2,1), AFB_URR.py:58,88.
 */
typedef struct vfn_stem_desc {
    const float* frame;   /* [3][H0][W0] planar, raw values in [0,1], un-padded */
    const float* mask;    /* [N][H0][W0] (soft or 0/1) or NULL when cin == 3 */
    const float* w;       /* [cin*7*8][64]: k = (plane*7+kh)*8+kw, 8th tap of every row = 0 */
    const float* scale;   /* [64] BN fold */
    const float* shift;   /* [64] */
    float* out;           /* NHWC [N][Ho][Wo][64] */
    float mean[3];
    float std[3];
    int N, cin;           /* cin = 3 (query encoder) or 5 (memory encoder: +mask, +inverse mask) */
    int H0, W0;           /* raw frame size */
    int pad_top, pad_left;/* zero padding applied before normalisation */
    int Hp, Wp;           /* padded size (multiple of 16) */
    int Ho, Wo;           /* Hp/2, Wp/2 */
} vfn_stem_desc;

int vfn_stem_conv7x7_f32(const vfn_stem_desc* d, void* stream);
int vfn_maxpool3x3s2_nhwc_f32(const float* in, float* out, int N, int H, int W, int C, void* stream);

/* ------------------------------------------------------------------ decoder pointwise / window ops
 * vfn_upsample2x_add_nhwc_f32   out = s + bilinear_x2(pm)              AFB_URR.py:124 (Refine)
 *     s: [N or 1][h][w][C] (s_bcast=1: one copy shared by all N), pm: [N][h/2][w/2][C]
 * vfn_rough_uncertainty_f32     AFB_URR.py:214-223 + myutils/data.py:40-46
 *     p: [obj][h][w][2] -> p_up [obj][2h][2w][2], rough [obj][2h][2w], unc [2h][2w]
 * vfn_local_hpass_f32 / vfn_local_vpass_f32   AFB_URR.py:226-231 (r1*rough, AvgPool2d(7,1,3) x2,
 *     divide, MaxPool2d(7,1,3)) as a separable window; cat([r1, r1_local]) (:231) is not materialised.
 *     The path for 5..8 objects (vfn_local_stats_f32 fuses both passes for up to 4) and its test reference.
 *     r1: [h][w][C] (shared by the objects), rough: [obj][h][w]
 *     scratch hs [obj][h][w][C], hr/hm [obj][h][w]; lm = r1_local: [obj][h][w][C], conf: [obj][h][w]
 * vfn_final_logits_f32          AFB_URR.py:233-237,300,309-316
 *     score[obj][H0][W0] = logit(clamp(softmax(bilinear_x2(p_up + unc*(conf*q)))[1], 1e-7, 1-1e-7)), un-padded
 */
int vfn_upsample2x_add_nhwc_f32(const float* s, const float* pm, float* out, int N, int h, int w, int C,
                                int s_bcast, void* stream);
/* the same, also writing the split-bf16 image of the result (of max(result, 0) with lp_relu) for a vfn_conv2d_nhwc_bf16x3
 * consumer with in_lp (vfn_conv_desc); C multiple of 32 */
int vfn_upsample2x_add_lp_nhwc_f32(const float* s, const float* pm, float* out, void* out_lp, int lp_relu,
                                   int N, int h, int w, int C, int s_bcast, void* stream);
int vfn_rough_uncertainty_f32(const float* p, float* p_up, float* rough, float* unc, int obj_n, int h, int w,
                              void* stream);
int vfn_local_hpass_f32(const float* r1, const float* rough, float* hs, float* hr, float* hm, int obj_n,
                        int h, int w, int C, void* stream);
int vfn_local_vpass_f32(const float* hs, const float* hr, const float* hm, float* lm,
                        float* conf, int obj_n, int h, int w, int C, void* stream);
/* vfn_local_stats_f32: vfn_local_hpass_f32 + vfn_local_vpass_f32 in one pass without scratch (AFB_URR.py:226-229): r1 once
 *     in, lm = r1_local [obj][h][w][C] and conf [obj][h][w] out; C = 64, obj_n <= 4 (larger: the two-pass pair).
 * vfn_pred2_gather_f32: second half of pred2 / local_pred2 (AFB_URR.py:195,202) evaluated as a tap GEMM: z [N][h][w][ldz]
 *     holds, per pixel, the 18 products of the 9 filter taps x 2 filters with relu(x) (a 1x1 vfn_conv2d over the
 *     repacked filters, which reads x once); out[n][y][x][o] = bias[o] + sum over the 3x3 neighbourhood (zero padding)
 *     of z[..][tap*2+o] (= the 3x3 convolution with two filters, up to summation order). */
int vfn_local_stats_f32(const float* r1, const float* rough, float* lm, float* conf, int obj_n, int h, int w, int C,
                        void* stream);
int vfn_pred2_gather_f32(const float* z, const float* bias, float* out, int N, int h, int w, int ldz, void* stream);
int vfn_final_logits_f32(const float* p_up, const float* unc, const float* conf, const float* q, float* score,
                         int obj_n, int h, int w, int pad_top, int pad_left, int H0, int W0, void* stream);

/* vfn_segment_uncertainty_f32: the scalar AFB_URR.segment returns in training mode (AFB_URR.py:302-305, consumed by the
 *     loss at train_video_seg.py:73-74): mean over the batch of ||calc_uncertainty(softmax_objects(prob))||_2 / sqrt(n).
 *     logit: [bs][obj][n] as returned by segment (n = H*W); partial: scratch float[bs*64]; out: one float. */
int vfn_segment_uncertainty_f32(const float* logit, int bs, int obj_n, int n, float* partial, float* out, void* stream);

/* ------------------------------------------------------------------ backward pass, first slice (csrc/backward_ops.hip)
 * train_video_seg.py:65-74 runs loss.backward() through AFB_URR.segment; the convolutions' gradients reuse
 * vfn_conv2d_nhwc_f32 (data gradient: flipped / transposed filters + vfn_conv_desc.mask / res; weight gradient: a GEMM over
 * the pixels on transposed operands, cut along K), these entry points are the HBM-bound pieces around them.
 *
 * vfn_transpose_taps_f32   out[(tap*C + c)][m] = colscale[c] * act(x[n][yo*stride - pad + kh][xo*stride - pad + kw][c]), 0 outside
 *     the image; tap = kh*k + kw, m = (n,yo,xo) flattened over the Ho x Wo output pixels of a k x k / stride / pad convolution,
 *     columns N*Ho*Wo .. Mpad-1 zero: the transposed im2col image (k = 1, stride 1, pad 0: a transposition).  relu: act =
 *     max(., 0); colscale optional (a frozen BatchNorm's scale applied to the gradient tensor).  x pixel stride ld_x floats.
 * vfn_colsum_f32           out[c] = sum_m x[m][c] (bias gradient); partial: scratch nb * C floats, nb <= 1024 blocks;
 *     two stages in fixed order (deterministic).
 * vfn_upsample2x_add_backward_f32   adjoint of vfn_upsample2x_add_nhwc_f32 (Refine, AFB_URR.py:124): gm [N][h][w][C] ->
 *     gs = sum over n (s_bcast = 1: the objects share s; gs may be NULL otherwise: ds = dm) and
 *     gpm [N][h/2][w/2][C] = interpolate^T(gm). */
int vfn_transpose_taps_f32(const float* x, int N, int H, int W, int C, int ld_x, int relu, int k, int stride, int pad, int Ho, int Wo,
                           const float* colscale, float* out, int Mpad, void* stream);

/* vfn_conv_wgrad_f32 (ABI 10)   the weight gradient of y = conv_{k x k, stride, pad}(act(x)) as an implicit GEMM over the pixels,
 *     straight from the NHWC tensors (no transposed copies): dw[co][(kh*k + kw)*Cin + ci] (the packed filter layout) =
 *     (accumulate ? dw : 0) + rowscale[co] * sum_m gy[m][co] * act(x)[pixel(m) + (kh, kw)][ci].  loss.backward() of
 *     train_video_seg.py:73 for every nn.Conv2d of AFB_URR.py / the torchvision bottlenecks.  ksplit > 1 cuts the pixels into
 *     4 * ksplit slices (4 per workgroup, summed through LDS; the rest through `partial` and a reduce launch, fixed order). */
typedef struct vfn_wgrad_desc {
    const float* x;          /* NHWC [N,H,W,ld_x]: the convolution's input (before act) */
    const float* gy;         /* [N,Ho,Wo,ld_g]: dL/dy */
    const float* rowscale;   /* optional [Cout]: the frozen BatchNorm scale behind the convolution */
    float* dw;               /* [Cout][k*k*Cin] */
    float* partial;          /* ksplit > 1: [ksplit][Cout][k*k*Cin] floats of workspace */
    int N, H, W, Cin, ld_x;
    int Ho, Wo, Cout, ld_g;
    int k, stride, pad;
    int relu;                /* act = max(., 0) */
    int accumulate;          /* add to dw instead of overwriting it */
    int ksplit;
    int* tile_counters;      /* (ABI 11) optional, ksplit > 1: one int per dW tile (ceil(Cout / 64) * k * k * ceil(Cin / 64 or 32)), zero
                              * at rest -- the slices are finished INSIDE the launch (write-through partial tiles, the wave that
                              * arrives last adds them in slice order: the same sums, no reduce launch) */
    int batch;               /* (ABI 11) > 1: that many independent problems of this shape in one launch -- x, gy advance by
                              * x_bstride / g_bstride floats per problem, dw by Cout * k * k * Cin, partial by ksplit times that, the
                              * counters by the tile count (the 36 components of a Winograd-domain weight gradient) */
    int reserved;
    long long x_bstride, g_bstride;
} vfn_wgrad_desc;
int vfn_conv_wgrad_f32(const vfn_wgrad_desc* d, void* stream);

/* vfn_stem_wgrad_f32 (ABI 12, round 5)   the weight gradient of the encoders' 7x7 / stride-2 / pad-3 stems (AFB_URR.py:44-46,67-69 under
 * train_video_seg.py:73 loss.backward()): x [N,Hp,Wp,C] dense NHWC, C = 3 (query encoder: conv1 over the frame) or 5 (memory encoder:
 * conv1 + conv1_m + conv1_o over frame | mask | other-objects planes), g [N,Ho,Wo,64] dense = dL/d(bn1 output), Ho = (Hp - 1) / 2 + 1,
 * rowscale [64] = the frozen bn1 scale (NULL: 1) -> dw [64][7][7][C] in vfn_conv_wgrad_f32's packed layout (accumulate: added to it).
 * The operand columns are (kw, c) of one filter row -- 7 C contiguous floats of the NHWC planes -- so one pixel walk feeds all seven
 * filter rows (vfn_conv_wgrad_f32 walks the pixels once per tap with 3 of 32 columns in use).  partial: scratch of at least
 * vfn_stem_wgrad_scratch_floats(C) floats.  Fixed summation order: bit-reproducible. */
int vfn_stem_wgrad_scratch_floats(int C);
int vfn_stem_wgrad_f32(const float* x, const float* g, const float* rowscale, float* dw, float* partial, long long partial_floats,
                       int N, int Hp, int Wp, int C, int Ho, int Wo, int accumulate, void* stream);

/* Derived-parameter refresh (ABI 11; csrc/refresh.hip).  After ``optimizer.step()`` (train_video_seg.py:76) everything this library
 * derives from the parameters -- packed filters in the forward layout, in the layout of the data-gradient convolution, the stems'
 * padded taps, the tap form of the two-filter heads; folded BatchNorm scale / shift (the reference freezes the statistics only,
 * train_video_seg.py:103-106) -- follows them in TWO launches over tables that live in device memory, instead of a handful of tensor
 * operators per layer and layout.
 *   src: a torch-layout filter tensor [cout][cin_total][kh][kw]; the entry covers input channels cin_off .. cin_off + cin - 1.
 *   kind 0  dst[(dst_row0 + co) * dst_ld + (t * cin + ci)]                       = src[co][ci][t]            (t = y * kw + x)
 *   kind 1  dst[(dst_row0 + ci) * dst_ld + (T-1-t) * cout_ld + dst_col0 + co]    = src[co][ci][t]            (flipped, transposed)
 *   kind 2  dst[(((dst_row0 + ci) * kh + y) * 8 + x) * dst_ld + co]              = src[co][ci][y][x]         (stem: 8 taps per row)
 *   kind 3  dst[(dst_row0 + t * cout + co) * dst_ld + ci]                        = src[co][ci][t]            (tap form)
 *   kind 4  dst[((xi * cout_ld + dst_row0 + co) * dst_ld + ci]                     = (G g G^T)[xi], g = src[co][ci]   (Winograd
 *           F(4x4, 3x3) filter banks, xi = 6 i + j, computed in float64; cout_ld = rows per bank, kh = kw = 3)
 *   kind 5  dst[((xi * cout_ld + dst_row0 + ci) * dst_ld + dst_col0 + co]          = (G g' G^T)[xi], g' = src[co][ci] flipped  (the
 *           banks of the data-gradient convolution; cout_ld = rows per bank)
 *           -- kinds 4 / 5 take ceil(cout * cin / vfn_refresh_elems_per_block()) workgroups --
 *   gamma / var / eps (optional): the value is multiplied by gamma[co] / sqrt(var[co] + eps) (a frozen BatchNorm's scale).
 *   block0: first workgroup of the entry; an entry takes ceil(cout * cin * kh * kw / vfn_refresh_elems_per_block()) workgroups and
 *   the table is ordered by block0. */
typedef struct vfn_refresh_filter {
    const float* src;
    float* dst;
    const float* gamma;
    const float* var;
    float eps;
    int kind;
    int cout, cin, cin_off, cin_total, kh, kw;
    int dst_ld, dst_row0, dst_col0, cout_ld;
    int block0;
    int reserved;
} vfn_refresh_filter;
/* scale[c] = gamma[c] / sqrt(var[c] + eps), shift[c] = beta[c] - mean[c] * scale[c] (gamma != NULL: a frozen BatchNorm; either
 * destination may be NULL), or shift[c] = beta[c] (gamma == NULL: a plain bias; beta NULL: zero). */
typedef struct vfn_refresh_epilogue {
    const float* gamma;
    const float* beta;
    const float* mean;
    const float* var;
    float* scale;
    float* shift;
    float eps;
    int C;
} vfn_refresh_epilogue;
int vfn_refresh_elems_per_block(void);
int vfn_refresh_filters_f32(const vfn_refresh_filter* table_dev, int n, int total_blocks, void* stream);
int vfn_refresh_epilogues_f32(const vfn_refresh_epilogue* table_dev, int n, void* stream);
/* The way back (ABI 11): the gradients of a step -- tensors of <= 4 dimensions with arbitrary strides (the weight gradients are
 * [Cout,Cin,kh,kw] views of packed-layout accumulators) -- copied into one flat buffer in ONE launch: dst[dst_offset + i] =
 * src[element i in row-major order of `shape`].  block0 / workgroups per entry as vfn_refresh_filter (elements = prod(shape)). */
typedef struct vfn_gather_entry {
    const float* src;
    long long dst_offset;     /* floats */
    long long stride[4];      /* floats */
    int shape[4];             /* leading dimensions padded with 1 */
    int block0;
    int reserved;
} vfn_gather_entry;
int vfn_gather_strided_f32(const vfn_gather_entry* table_dev, int n, int total_blocks, float* dst, void* stream);
/* Encoder pieces (ResNet trunks, BatchNorm frozen as train_video_seg.py:103-106 sets it), memory read, optimiser:
 * vfn_dilate2_f32          out[n][2y][2x][c] = g[n][y][x][c], 0 elsewhere ([N][H][W][C] from [N][Ho][Wo][C]): the data gradient of a
 *     stride-2 convolution is the stride-1 data-gradient convolution of this.
 * vfn_bn_param_grads_f32   dbeta[c] = sum_m g, dgamma[c] = sum_m g * (y - idn - beta) / gamma for y = gamma * xhat + beta (+ idn,
 *     then ReLU) as the forward stored it; g = the gradient w.r.t. the BatchNorm's output (zero wherever the ReLU clipped).
 *     partial: scratch 2 * nb * C floats, nb <= 1024.
 * vfn_maxpool3x3s2_backward_f32   MaxPool2d(3,2,1): gx [N][H][W][C] from g [N][Ho][Wo][C], first maximum of a window wins;
 *     add (optional, shape of x): a second gradient arriving at x; relu_mask: x is a ReLU's output, the sum is zeroed where x <= 0.
 * vfn_softmax_cols_f32 / _backward_f32   P = softmax over the rows of scale * S [B][ld] per column (AFB_URR.py:144-145 with the
 *     bank of one frame materialised, as training has it); dS = scale * P (dP - sum_b P dP).
 * vfn_adamw_f32            one torch.optim.AdamW step on n floats (decoupled decay, bias corrections 1 - beta^step; the hyper-parameters
 *     are doubles: 1 - beta and the corrections are formed in double as torch forms its python scalars, then rounded to f32). */
int vfn_dilate2_f32(const float* g, float* out, int N, int Ho, int Wo, int H, int W, int C, void* stream);
int vfn_bn_param_grads_f32(const float* g, const float* y, const float* idn, const float* beta, const float* gamma, int M, int C,
                           float* partial, int nb, float* dgamma, float* dbeta, void* stream);
int vfn_bn_param_grads_acc_f32(const float* g, const float* y, const float* idn, const float* beta, const float* gamma, int M, int C,
                               float* partial, int nb, float* dgamma, float* dbeta, int accumulate, int* counter,
                               void* stream);   /* (ABI 10; accumulate / counter as vfn_colsum_acc_f32) */
int vfn_maxpool3x3s2_backward_f32(const float* x, const float* g, float* gx, int N, int H, int W, int C, const float* add, int relu_mask,
                                  void* stream);
int vfn_softmax_cols_f32(const float* S, int B, int Q, int ld, float scale, float* P, void* stream);
int vfn_softmax_cols_backward_f32(const float* P, const float* dP, int B, int Q, int ld, float scale, float* dS, void* stream);
int vfn_adamw_f32(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2, double eps,
                  double weight_decay, int step, void* stream);
int vfn_colsum_f32(const float* x, int M, int C, int ld, float* partial, int nb, float* out, void* stream);
/* the same with the result ADDED to `out` when accumulate != 0 (ABI 10: a running gradient over the samples of a batch);
 * counter != NULL (VFN_COLSUM_COUNTERS ints in device memory, zero at rest): ONE launch -- the blocks publish their partial rows
 * write-through and the block that arrives last (of a 64-channel slab, when C % 64 == 0 and everything is 16-byte aligned) adds them
 * in a fixed order (no second launch) */
#define VFN_COLSUM_COUNTERS 64
int vfn_colsum_acc_f32(const float* x, int M, int C, int ld, float* partial, int nb, float* out, int accumulate, int* counter,
                       void* stream);
int vfn_upsample2x_add_backward_f32(const float* gm, float* gs, float* gpm, int N, int h, int w, int C, int s_bcast,
                                    void* stream);
/* The decoder's tail backwards (AFB_URR.py:214-237,300,309-316), in the stages the host threads the local head's convolution
 * gradients through (backward.py::DecoderBackward.run_tail):
 * vfn_tail_grad_o_f32   G = dL/dscore [obj][H0][W0] (the logits segment returns) -> g_o [obj][2h][2w][4]: gradient w.r.t. the
 *     two channels of interpolate(p2) in the PADDED frame (channels 2, 3 and the padding stay as the caller zeroed them);
 *     zero where the clamp of :309 is active.  The adjoint of the interpolation is vfn_upsample2x_add_backward_f32 (C = 4).
 * vfn_tail_split_f32    p2 = p_up + unc * conf * q: g_p2 [obj][pix][4] -> g_q [obj][pix][32] (channels 0, 1 written: the
 *     32-channel gradient tensor the data-gradient convolution of local_pred2 reads), g_cf [obj][pix], g_u [pix].
 * vfn_local_stats_backward_f32   through r1_local = avg7(r1*rough)/(avg7(rough)+1e-8), conf = max7(rough), the uncertainty's
 *     top-2 and the two softmaxes: g_lm = dL/dr1_local [obj][pix][C], g_cf, g_u, g_p2 -> g_r1 [pix][C] (ACCUMULATED) and
 *     g_pup [obj][pix][4] = dL/d interpolate(p).  Scratch: dA [obj][pix][C], dBv [obj][pix], amax int[obj][pix]. */
/* vfn_segment_loss_f32   train_video_seg.py:72-74: loss = CrossEntropyLoss(scores, label) + lu * uncertainty on the logits
 *     segment returns (scores [bs][obj][n], label int32 [bs][n]; uncertainty as vfn_segment_uncertainty_f32) and, with grad
 *     != NULL, dloss/dscores [bs][obj][n] -- the input of the decoder's backward.  partial: scratch 2*bs*64 floats;
 *     stats: 3 + bs floats = (loss, cross entropy, uncertainty, ||u|| per sample).  Deterministic two-stage sums. */
int vfn_segment_loss_f32(const float* logit, const int* label, int bs, int obj_n, int n, float lu, float* partial, float* stats,
                         float* grad, void* stream);
/* vfn_segment_uncertainty_backward_f32   the adjoint of the uncertainty alone (ABI 10), for an autograd boundary where the
 *     criterion is the caller's (``scores, uncertainty = model.segment(frames, fb); loss = criterion(scores, label) + lu *
 *     uncertainty; loss.backward()``, train_video_seg.py:69-74): grad [bs][obj][n] = g_scores (optional: dL/dscores from the
 *     caller's criterion) + *g_unc_dev * d uncertainty / d scores; g_unc_dev = dL/duncertainty as a float IN DEVICE MEMORY (what
 *     autograd hands over: no host round trip).  partial / stats as vfn_segment_loss_f32. */
int vfn_segment_uncertainty_backward_f32(const float* logit, int bs, int obj_n, int n, const float* g_unc_dev,
                                         const float* g_scores, float* partial, float* stats, float* grad, void* stream);
int vfn_tail_grad_o_f32(const float* G, const float* p_up, const float* unc, const float* conf, const float* q, float* g_o,
                        int obj_n, int h, int w, int pad_top, int pad_left, int H0, int W0, void* stream);
int vfn_tail_split_f32(const float* g_p2, const float* unc, const float* conf, const float* q, float* g_q, float* g_cf, float* g_u,
                       int obj_n, int npix, void* stream);
int vfn_local_stats_backward_f32(const float* g_lm, const float* lm, const float* g_cf, const float* g_u, const float* g_p2,
                                 const float* r1, const float* rough, const float* p_up, float* dA, float* dBv, int* amax,
                                 float* g_r1, float* g_pup, int obj_n, int h, int w, int C, void* stream);

/* ------------------------------------------------------------------ feature-bank contractions (f32 MFMA)
 * Bank layout: entry-major, keys [obj][cap][128], values [obj][cap][512], info [obj][cap][2]
 * (birth frame, log-hit accumulator), live length bank_len[obj] in DEVICE memory (kernels never
 * need the host to know it; the host passes only an upper bound through the grid/nsplit choice).
 *
 * vfn_bank_scan (mode 0) + vfn_bank_scan_finish: per query column q, m = max_b s, l = sum_b exp(s-m),
 *     s = scale * <keys[b], q>           -- softmax statistics of Matcher.forward, AFB_URR.py:144-145
 * vfn_bank_scan (mode 1) + vfn_bank_scan_finish: idx = argmax_b <keys[b], q> * rowscale[b],
 *     corr = max * colscale[q]           -- FeatureBank.update cosine match, FeatureBank.py:63-68
 *     (rowscale/colscale = 1/max(||.||, 1e-12) from vfn_row_norms)
 * vfn_memread_apply: p = exp(s-m)/l recomputed per tile, hit counts cnt[b] += sum_q [p > thres],
 *     partial O^T = P^T V per bank split      -- AFB_URR.py:145-146,163-165
 * vfn_memread_finish: out[obj][q][0:512] = sum_split O^T, out[obj][q][512:1024] = query value
 *     (torch.cat([mem, q_out]), AFB_URR.py:159); info[:,1] += log(cnt+1), AFB_URR.py:174
 */
typedef struct vfn_bankscan_desc {
    const float* q;        /* [HW][ldq] (mode 0: one query set; mode 1 with q_per_obj: [obj][HW][ldq]) */
    const float* bank_k;   /* [obj][cap][128] */
    const int* bank_len;   /* [obj], device */
    const float* rowscale; /* mode 1: [obj][stride_rs] */
    float* part;           /* [obj][nsplit][HW][2] scratch */
    long long stride_q, stride_k, stride_rs;   /* elements between objects */
    float scale;           /* mode 0: 1/sqrt(128) */
    int ldq, q_per_obj, HW, obj_n, nsplit, mode;
    int precision;         /* arithmetic of the contraction: 0 exact f32, 1 bf16 operands, 2 bf16x3 (see
                              vfn_conv2d_nhwc_bf16 / _bf16x3); tensors are f32 in every case */
    int* work_counter;     /* one int of device memory: the queue head of the persistent workgroups (zeroed by the
                              launcher on the stream); nsplit is the number of bank slices per (query tile, object) --
                              work items -- and may be far larger than the number of resident workgroups */
    const void* bank_k_lp; /* precision 1 / 2 only, optional: the keys' split-bf16 image kept by vfn_bank_refresh_lp
                              ([obj][cap][128 hi | 128 lo] bf16, object stride = stride_k * 4 bytes).  With it the
                              kernel reads MFMA operands directly; NULL: keys are split on the fly (same results) */
    float* scores;         /* mode 0, optional: the raw scores <keys[b], q> are also WRITTEN here, so that vfn_memread_apply
                              (same field) reads them back instead of multiplying keys and queries a second time.  Per
                              object (stride_scores floats apart) one 32 KB tile per (64-entry chunk c, 128-query tile t) at
                              tile index c * ceil(HW/128) + t, laid out [key half 2][row group 4][lane half 2][query 128][4]
                              (= the MFMA accumulator layout, so both kernels move whole 512-byte runs).  Needs
                              ceil(cap/64) * ceil(HW/128) * 8192 floats per object with cap = stride_k / 128, the capacity
                              of the key slab: vfn_bank_scan and vfn_memread_apply return VFN_ERR_ARG for less. */
    long long stride_scores;
} vfn_bankscan_desc;

typedef struct vfn_memread_desc {
    const float* q;        /* [HW][ldq] query keys */
    const float* qv;       /* [HW][ldqv] query values, copied into out[..][512:1024]; NULL: no copy (the caller
                              applies convFM to the two halves of torch.cat([mem, q_out]) separately) */
    const float* bank_k;   /* [obj][cap][128] */
    const float* bank_v;   /* [obj][cap][512] */
    const int* bank_len;   /* [obj], device */
    const float* ml;       /* [obj][HW][2] from the mode-0 scan */
    float* o_part;         /* [obj][nsplit][HW][512] scratch */
    int* cnt;              /* [obj][stride_cnt] zero-initialised hit counters, or NULL (update_bank=False) */
    float* info;           /* [obj][cap][2] */
    float* out;            /* [obj][HW][ld_out] decoder input, ld_out >= 1024 (>= 512 when qv is NULL) */
    long long stride_k, stride_v, stride_cnt, stride_info;
    float scale, thres;
    int ldq, ldqv, ld_out, HW, obj_n, nsplit;
    int precision;         /* as vfn_bankscan_desc.precision (scores, P and value operands); softmax in f32 */
                           /* (a workgroup owns 128 query columns: choose nsplit for ceil(HW/128) query tiles) */
    const void* bank_k_lp; /* precision 1 / 2: the split-bf16 images of keys and values kept by         */
    const void* bank_v_lp; /* vfn_bank_refresh_lp (both or neither; same results as the on-the-fly split, without
                              the conversion work in the kernel): values [obj][cap / 8 blocks][hi | lo plane][512
                              channels][8 rows] bf16 -- 16 bytes = one lane's B operand of a P^T V MFMA (cap % 8 == 0) */
    const float* scores;   /* precision 0, optional: the scores the mode-0 vfn_bank_scan of this frame stored
                              (vfn_bankscan_desc.scores, same layout and stride).  The kernel then runs no score GEMM and
                              touches neither q nor bank_k: bit-identical results (the scan forms the same sums). */
    long long stride_scores;
} vfn_memread_desc;

/* Round 6, precision 1 (plain bf16) with the kept image (bank_k_lp / bank_v_lp): both entry points select software-pipelined
 * kernels -- vfn_bank_scan: bank_scan_pipe_kernel<mode> (keys through registers two chunks ahead, three workgroups per CU);
 * vfn_memread_apply: memread_apply_pipe_kernel (the softmax of chunk c+1 in the shadow of chunk c's P^T V, value rows a chunk
 * ahead, one barrier per chunk).  Same products in the same order as the kernels they replace: bit-identical outputs
 * (tests/test_round6_gpu.py).  VFN_SCAN_PIPE=0 / VFN_APPLY_PIPE=0 (read at every call) bring the round-5 kernels back. */
int vfn_bank_scan(const vfn_bankscan_desc* d, void* stream);
int vfn_bank_scan_finish(const float* part, int nsplit, int HW, int obj_n, int mode, float* ml, int* idx,
                         float* corr, const float* colscale, void* stream);
int vfn_memread_apply(const vfn_memread_desc* d, void* stream);
int vfn_memread_finish(const vfn_memread_desc* d, void* stream);

/* ------------------------------------------------------------------ feature-bank maintenance
 * vfn_row_norms     ||x_r||_2 per row (bank entries or new features)        FeatureBank.py:63-65,87-89
 * vfn_bank_merge    scatter_mean of normalised new features into their matched entries
 *                   (corr > thres_close) + magnitude-preserving blend         FeatureBank.py:71-97
 * vfn_bank_append   append set (corr <= thres_close) in ascending source order; LFU eviction
 *                   (remove(), FeatureBank.py:117-143) when class_budget < B + n_append;
 *                   order-preserving compaction; new info rows (frame_idx, new_hit_init);
 *                   clamp(info[:,1],0,1e5); new length -> bank_len; stats[obj] =
 *                   {len, peak_n, replace_n, n_append}                        FeatureBank.py:100-115
 * vfn_scatter_mean_f32  torch_scatter.scatter_mean(src[D,S], index, dim=1, out=out[D,B]):
 *                   out[:,t] = (out[:,t] + sum_{index[s]==t} src[:,s]) / max(count_t,1) for every
 *                   t that occurs in index (other columns: out/1 = unchanged)   FeatureBank.py:78,92
 */
#define VFN_BANK_MAX_HW 32768   /* new features per object and frame (1/16-resolution pixels) the bank kernels accept */
typedef struct vfn_bank_desc {
    float* bank_k;             /* [obj][cap][128] */
    float* bank_v;             /* [obj][cap][512] */
    float* info;               /* [obj][cap][2]   */
    float* scratch_k;          /* same shapes: staging for order-preserving compaction */
    float* scratch_v;
    float* scratch_info;
    const int* bank_len;       /* [obj] device */
    int* bank_len_rw;          /* same buffer, written last */
    const float* bank_knorm;   /* [obj][stride_n] ||key_b||   */
    const float* bank_vnorm;   /* [obj][stride_n] ||value_b|| */
    const int* match_idx;      /* [obj][HW] arg-max bank entry per new feature */
    const float* match_corr;   /* [obj][HW] its cosine */
    const float* new_k;        /* [obj][HW][ld_new]: key at +0 (128), value at +voff (512) */
    const float* new_knorm;    /* [obj][HW] */
    const float* new_vnorm;    /* [obj][HW] */
    int* app_pos;              /* [obj][HW] scratch */
    int* keep_dst;             /* [obj][stride_n] scratch */
    int* plan;                 /* [obj][4] scratch */
    int* stats;                /* [obj][4] persistent: len, peak_n, replace_n, last n_append (or, sticky, -1 / -2: the
                                  LFU score of remove() held a NaN / was all-infinite -- the reference raises there) */
    long long stride_k, stride_v, stride_info, stride_n, stride_new;
    double class_budget;       /* FeatureBank.py:20-22: float 0.8*(budget//obj_n) when obj_n == 2 */
    float thres_close, update_rate, new_hit_init;
    int frame_idx, ld_new, voff, HW, obj_n, cap;
    int rm_class;              /* -1 for update(); >= 0: vfn_bank_remove evicts from this object only */
    int rm_request;            /* remove(class_idx, request_n, frame_idx): room to make (FeatureBank.py:117-143) */
} vfn_bank_desc;

/* Split-bf16 image of the bank for the reduced-precision contractions (precision 1 / 2): per entry the keys as
 * [128 hi | 128 lo] bf16; the values in blocks of 8 entries as [hi plane | lo plane][512 channels][8 entries] bf16 (16 KB per
 * block; 16 bytes = 8 consecutive entries of one channel = one lane's B operand of a P^T V MFMA) (hi = RNE bf16 of x, lo = RNE
 * bf16 of x - hi): the same bytes per entry as the f32 rows, object strides stride_k / stride_v * 4 bytes, cap % 8 == 0.
 * all_rows = 0: after vfn_bank_merge + vfn_bank_append of the same descriptor, re-split only the entries that update
 * changed (merged or appended; every entry when it compacted the bank).  all_rows = 1: every live entry. */
int vfn_bank_refresh_lp(const vfn_bank_desc* d, void* bank_k_lp, void* bank_v_lp, int all_rows, void* stream);

int vfn_row_norms(const float* x, long long stride_obj, int ld, int dim, const int* len_dev, int rows,
                  int obj_n, float* nrm, float* inv /* 1/max(nrm,1e-12) or NULL */, long long stride_n, void* stream);
int vfn_bank_merge(const vfn_bank_desc* d, void* stream);
int vfn_bank_append(const vfn_bank_desc* d, void* stream);
/* FeatureBank.remove(class_idx, request_n, frame_idx) on its own (update() fuses it into vfn_bank_append):
 * LFU threshold loop + order-preserving compaction of one object; stats[obj] = {len, peak, replace_n, -}. */
int vfn_bank_remove(const vfn_bank_desc* d, void* stream);
/* Norms carried across frames instead of recomputed over the whole bank (FeatureBank.py:63-65,87-88 recompute them every
 * update): call after vfn_bank_merge + vfn_bank_append of one update with the same descriptor; refreshes ||key||,
 * 1/max(||key||,1e-12) and ||value|| of the entries that update touched (merged and appended rows; every row when it
 * evicted), with the summation of vfn_row_norms -- bit-identical to recomputing them all. */
int vfn_bank_refresh_norms(const vfn_bank_desc* d, float* bank_knorm, float* bank_kinv, float* bank_vnorm, void* stream);
int vfn_scatter_mean_f32(const float* src, long long src_s0, long long src_s1, const long long* index,
                         int S, float* out, long long out_s0, long long out_s1, int D, void* stream);
/* The same with the argument validation torch_scatter performs (a device-side assert there) done on the device, so that the
 * operator never synchronises the host (ABI 10): index row 0 is used; a target outside [0, B) is skipped and sets bit 0 of
 * *status (an int in device memory, sticky, zero at rest); index_s0 != 0 declares a materialised [D][S] index with that row
 * stride (elements), whose rows must all equal row 0 -- a differing element sets bit 1.  The caller reads *status whenever
 * it synchronises anyway (v-floodnet_amd/scatter.py: check_status). */
int vfn_scatter_mean_checked_f32(const float* src, long long src_s0, long long src_s1, const long long* index,
                                 long long index_s0, int S, float* out, long long out_s0, long long out_s1, int D,
                                 long long B, int* status, void* stream);

/* ------------------------------------------------------------------ per-frame loop operators (planar NCHW)
 * vfn_resize_bicubic_f32    TF.resize(frame, 480, BICUBIC)                       test_video_seg.py:88,107
 * vfn_resize_nearest_f32    TF.resize(first_mask, 480, NEAREST)                  test_video_seg.py:89
 * vfn_softmax_objects_f32   F.softmax(score, dim=1)                              test_video_seg.py:109
 * vfn_resize_argmax_u8      TF.resize(pred_mask, ori_size, BICUBIC); argmax(dim=0) -> uint8   test_video_seg.py:114-115
 * vfn_postprocess_pred_u8   myutils.postprocessing_pred (HOST buffers; the reference runs cv2 CCL on the
 *                           CPU as well)                                          myutils/data.py:17-37
 */
int vfn_resize_bicubic_f32(const float* in, float* out, int C, int Hi, int Wi, int Ho, int Wo, void* stream);
int vfn_resize_nearest_f32(const float* in, float* out, int C, int Hi, int Wi, int Ho, int Wo, void* stream);
int vfn_softmax_objects_f32(const float* score, float* prob, int obj_n, int n, void* stream);
int vfn_resize_argmax_u8(const float* prob, unsigned char* label, int obj_n, int Hi, int Wi, int Ho, int Wo,
                         void* stream);
int vfn_postprocess_pred_u8(const unsigned char* pred_host, int H, int W, unsigned char* out_host);
/* The same on DEVICE buffers (union-find CCL with atomicMin, largest component, reference special cases), so the
 * label map leaves the GPU already post-processed.  scratch: int32[2*H*W + 8]. */
int vfn_postprocess_pred_device_u8(const unsigned char* pred, unsigned char* out, int* scratch, int H, int W,
                                   void* stream);

/* ------------------------------------------------------------------ input / output side
 * vfn_to_tensor_u8: torchvision ToTensor of a decoded frame (Video_DS.__getitem__, dataset/Water_DS.py:131-139):
 *     uint8 [H][W][3] -> float32 [3][H][W], x / 255 (IEEE division: bit-identical to tensor.float().div(255)).
 * vfn_overlay_u8: myutils.add_overlay + the uint8 conversion of save_overlay (myutils/data.py:56-84): RGB uint8
 *     [H][W][3] from the float frame [3][H][W] in [0,1] and the label map; palette = 256 RGB triples
 *     (myutils/data.py:14), alpha / cscale as add_overlay; scratch = one int.  Blend in f64, truncating casts, the
 *     smallest label present is the background (`for i in ids[1:]`), 4-neighbour contours black. */
int vfn_to_tensor_u8(const unsigned char* src, float* dst, int H, int W, void* stream);
int vfn_overlay_u8(const float* frame, const unsigned char* mask, const unsigned char* palette, int* scratch,
                   unsigned char* out, int H, int W, double alpha, double cscale, void* stream);

/* ------------------------------------------------------------------ JPEG frames onto the device
 * Replaces PIL's decode + torchvision ToTensor in Video_DS.__getitem__ (video_module/dataset/Water_DS.py:105-109,
 * myutils/data.py:87-90; frames arrive at test_video_seg.py:103-105).
 * vfn_jpeg_entropy_decode   HOST function (no GPU call): marker parsing + Huffman decoding of a baseline / extended
 *     sequential 8-bit JPEG (grey, or YCbCr with 4:4:4 / 4:2:2 / 4:2:0 / 4:4:0 sampling, one interleaved scan, restart
 *     intervals) into quantised coefficient blocks.  info: int[24] (0 width, 1 height, 2 components, 3 hmax, 4 vmax,
 *     5 MCU columns, 6 MCU rows, 7+4c.. per component: h, v, blocks per row, block rows; 22 restart interval; 23 total
 *     shorts of coefficients); qt: unsigned short [3][64] natural order, per component; coef: per component
 *     [block rows][blocks per row][64] natural order, components back to back.  Returns 0, -1 (not a JPEG / truncated),
 *     -2 (unsupported: progressive, arithmetic, 12-bit, CMYK / Adobe RGB, non-interleaved colour scans), -3 (coef_cap
 *     too small: info[23] says how much), -4 (corrupt entropy data).  Call with coef = NULL to size the buffer.
 * vfn_jpeg_idct_u8          dequantise + libjpeg's accurate integer IDCT (jidctint.c) of one component's blocks
 *     (device memory) -> uint8 plane [block_rows*8][pitch], pitch >= blocks_per_row*8, multiple of 8.
 * vfn_jpeg_to_tensor_f32    libjpeg "fancy" chroma upsampling (hs, vs = luma / chroma sampling ratio, 1 or 2) +
 *     YCbCr -> RGB (jdcolor.c fixed point) + ToTensor: out_f32 [3][H][W] = rgb / 255 and / or out_u8 [H][W][3].
 */
int vfn_jpeg_entropy_decode(const unsigned char* data, long long size, short* coef, long long coef_cap,
                            unsigned short* qt, int* info);
int vfn_jpeg_idct_u8(const short* coef, const unsigned short* qt, unsigned char* plane, int blocks_per_row,
                     int block_rows, int pitch, void* stream);
int vfn_jpeg_to_tensor_f32(const unsigned char* y, const unsigned char* cb, const unsigned char* cr, int pitch_y,
                           int pitch_c, int W, int H, int hs, int vs, int ncomp, float* out_f32, unsigned char* out_u8,
                           void* stream);

/* ------------------------------------------------------------------ PNG encoding on the device
 * vfn_png_deflate_u8: the compression inside save_seg_mask (PIL mode-P PNG, myutils/data.py:49-53) and save_overlay
 *     (cv2.imwrite, myutils/data.py:78-84).  raw = uint8 [H][W][bpp] on the device, bpp = 1 (palette indices) or 3 (RGB).
 *     PNG row filter (Up for bpp 1, Paeth for bpp 3) + distance-1 run matches + a Huffman code of the image's own
 *     histogram; out receives ONE final dynamic deflate block (RFC 1951), stats = {deflate bytes, Adler-32 of the
 *     filtered scanlines, deflate bits, 0}.  The host wraps it: 78 01 | deflate | adler32 (big endian) = the IDAT
 *     payload; signature / IHDR / PLTE / IDAT / IEND framing with CRC-32 stays on the host (a few hundred ns per KB).
 * vfn_png_sizes: bytes of `work` and `out` the caller must provide for an H x W x bpp image.
 */
int vfn_png_sizes(int H, int W, int bpp, long long* work_bytes, long long* out_bytes);
int vfn_png_deflate_u8(const unsigned char* raw, int H, int W, int bpp, void* work, unsigned char* out, int* stats,
                       void* stream);

/* ------------------------------------------------------------------ PNG frames on the input side
 * Video_DS opens every frame with PIL (Water_DS.py:105-109 -> myutils/data.py:87-90: Image.open(p).convert('RGB')) and
 * ToTensor divides by 255.  For PNG files a host core only inflates the IDAT stream (zlib); the scanline filters and
 * the pixel conversion run on the device:
 * vfn_png_unfilter_sizes  pitch (bytes per re-pitched row, a multiple of 4*bpp) and the bytes of `work` for an image.
 * vfn_png_unfilter_u8     filtered: height x (1 + width*bpp) bytes as they come out of inflate (filter-type byte first);
 *                         8-bit samples, bpp = 1..4 bytes per pixel, not interlaced, width <= 4096.  raw: height x pitch
 *                         bytes.  *status (device int, zeroed by the caller) becomes 1 on a filter type outside 0..4.
 * vfn_png_to_tensor_f32   colour type 0 grey / 2 RGB / 3 palette (palette: 256 x RGB bytes on the device) / 4 grey+alpha
 *                         / 6 RGBA -> out float[3][H][W] = RGB / 255 exactly as PIL's convert('RGB') + ToTensor (grey
 *                         replicated, alpha dropped), and out_u8 [H][W][3] unless NULL. */
int vfn_png_unfilter_sizes(int width, int height, int bpp, int* pitch, long long* work_bytes);
int vfn_png_unfilter_u8(const unsigned char* filtered, int width, int height, int bpp, void* work, unsigned char* raw,
                        int* status, void* stream);
int vfn_png_to_tensor_f32(const unsigned char* raw, int pitch, int width, int height, int color_type,
                          const unsigned char* palette, float* out, unsigned char* out_u8, void* stream);

/* ------------------------------------------------------------------ first-frame bootstrap model
 * test_video_seg.py:67-69 calls test_image_seg.test_waterseg (:133: a pickled smp.Linknet over EfficientNet-B4) when a clip has
 * no first-frame mask.  Its 1x1 convolutions and the decoder's 4x4 transposed convolutions (stride-1 convolution of the
 * zero-inserted input, vfn_dilate2_f32) run through vfn_conv2d_nhwc_f32 on channel-padded NHWC tensors; these are the rest
 * (the architecture is restated in oracle/linknet_ref.py -- third-party packages, parity unpinned):
 * vfn_ln_stem_f32       x [N][3][H][W] -> out [N][Ho][Wo][ld] = swish(BN(conv 3x3 / stride 2 (x))), 48 filters [48][3][3][3],
 *                       pad_before rows / columns of zeros before the image (TensorFlow-style "same" padding: the rest after),
 *                       channels 48..ld-1 = 0 (ld % 16 == 0)
 * vfn_ln_dwconv_f32     depthwise k x k (3 / 5), stride 1 / 2: out = swish(scale * conv(act(x)) + shift), act = swish when
 *                       swish_in; w [k*k][C] tap-major; C % 4 == 0 (padded channels: zero filters, scale, shift)
 * vfn_ln_se_gate_f32    squeeze-excite gate from the per-channel pixel sums (vfn_colsum_f32): gate[c] = sigmoid(b2 + w2 .
 *                       swish(b1 + w1 . (sum_px * inv_hw))), w1 [sq][C], w2 [C][sq]; gate[C..Cpad-1] = 0; sq <= 256
 * vfn_ln_scale_cols_f32 out[r][k] = w[r][k] * g[k]: the gate applied to the project convolution's packed filter matrix
 * vfn_ln_add_f32        out = a + b (n % 4 == 0): the decoder's skip connections (added AFTER the block's last ReLU)
 * vfn_ln_head_f32       out[m] = sigmoid(bias + sum_c x[m][c] w[c]) (prob = 1) or the logit (prob = 0) */
int vfn_ln_stem_f32(const float* x, const float* w, const float* scale, const float* shift, float* out, int N, int H, int W, int Ho,
                    int Wo, int ld, int pad_before, void* stream);
int vfn_ln_dwconv_f32(const float* x, const float* w, const float* scale, const float* shift, float* out, int N, int H, int W, int C,
                      int ld_x, int ld_out, int k, int stride, int pad_before, int Ho, int Wo, int swish_in, void* stream);
int vfn_ln_se_gate_f32(const float* sum_px, float inv_hw, const float* w1, const float* b1, const float* w2, const float* b2,
                       float* gate, int C, int sq, int Cpad, void* stream);
int vfn_ln_scale_cols_f32(const float* w, const float* g, float* out, int rows, int K, void* stream);
int vfn_ln_add_f32(const float* a, const float* b, float* out, long long n, void* stream);
int vfn_ln_head_f32(const float* x, const float* w, float bias, float* out, long long M, int C, int ld, int prob, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VFN_HIP_H */
