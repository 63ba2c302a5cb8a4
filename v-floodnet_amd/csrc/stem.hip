// Encoder stems: fused zero-pad + ImageNet normalisation + 7x7/s2 conv (+ the two
// 1->64 mask convs of EncoderM) + eval BatchNorm + ReLU, and the 3x3/s2 max-pool.
//
//   AFB_URR.py:53-58  EncoderM: f=(in_f-mean)/std; conv1(f)+conv1_m(m)+conv1_o(o); bn1; relu; maxpool
//   AFB_URR.py:83-88  EncoderQ: same without the mask convs
//   AFB_URR.py:259-264 / :279  the frame/mask are zero-padded to a multiple of 16 *before*
//                     normalisation (myutils/data.py:132-149), and mask_inv=(1-mask).clamp(0,1)
//                     is formed after padding -> padded pixels carry (-mean/std, 0, 1).
//
// The three stem convs are one GEMM over 3 (or 5) input planes.  K is ordered
// (plane, kh, kw) with each 7-tap filter row padded to 8 taps (zero weight) so that
// the two k values of one v_mfma_f32_32x32x2_f32 step are neighbouring pixels and
// every LDS address is base + immediate.  A operand = pixels, B = filters, as in
// conv_igemm.hip.
#include "common.h"
#include "../../include/vfn_hip.h"

namespace {

constexpr int TH = 8, TW = 16;            // output tile (pixels) per workgroup
constexpr int PH = TH * 2 + 5;            // 21 input rows
constexpr int PW = 38;                    // 37 input cols + 1 zero column for the padded 8th tap
// LDS image of the input patch: even and odd columns in separate half-rows, [plane][row][parity][PP].  The stride-2
// convolution makes a lane read column 2*px + 2*q + parity: in a plain row-major image the 32 lanes of a ds_read_b32 group
// touch only every second bank (22 % of the LDS cycles were conflicts); de-interleaved they read px + q -- consecutive
// banks -- and the two 16-lane halves (output rows py, py+1 = image rows +2) sit 4*PP = 80 words = 16 banks apart.
constexpr int PP = 20;                    // columns per parity (19 used)
constexpr int PROW = 2 * PP;              // words per image row

template <int CIN>
__global__ __launch_bounds__(256)
void stem_kernel(const vfn_stem_desc p) {
    constexpr int KP = CIN * 7 * 8;       // padded K
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sW = reinterpret_cast<float*>(smem);        // [KP][64]
    float* sP = sW + KP * 64;                          // [3][PH][2][PP]: the frame planes, then (same space) an object's 2 mask planes

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;

    // One workgroup per output tile, for ALL images of the batch: the frame planes (and with them the first 3 x 7 x 8
    // k-steps of the GEMM) are the same for every object (AFB_URR.py:261 expands the frame), so their products are
    // accumulated once and every object continues from that accumulator with its own two mask planes -- the same
    // k-order as one 5-plane GEMM per object, bit for bit, at 7/10 of the matrix work and one filter load per tile.
    const int tiles_x = (p.Wo + TW - 1) / TW;
    const int b = blockIdx.x;
    const int ty = b / tiles_x, tx = b - ty * tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;

    // filters -> LDS (already packed [KP][64] with zero 8th taps)
    for (int i = tid * 4; i < KP * 64; i += 256 * 4)
        *reinterpret_cast<f32x4*>(sW + i) = *reinterpret_cast<const f32x4*>(p.w + i);

    // input patch -> LDS, with pad + normalisation semantics: planes [c_lo, c_hi) of image n
    const int gy0 = oy0 * 2 - 3, gx0 = ox0 * 2 - 3;     // padded-frame coordinates
    // Round 5: branch-free and batched.  The first form was `for (i = tid; ...; i += 256) { if (inside) v = frame[...]; ... sP[...] = v; }`:
    // one dword load and one full wait per iteration, ten dependent round trips per thread for the frame planes (six per object for the
    // mask planes) -- most of the kernel's 40 / 88 us.  Now a thread requests ALL its elements first (raw buffer loads: an element outside
    // the raw frame gets an out-of-range offset and reads 0, which is exactly the zero padding the reference applies BEFORE it
    // normalises, myutils/data.py:132-149) and converts / stores them afterwards.  Same arithmetic per element.
    const __amdgpu_buffer_rsrc_t rs_frame = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.frame), 0, 3 * p.H0 * p.W0 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_mask = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.mask ? p.mask : p.frame), 0,
                                                                             p.mask ? p.N * p.H0 * p.W0 * 4 : 0, 0x00020000);
    auto load_planes = [&](int c_lo, int c_hi, int n) {
        constexpr int MAXIT = (3 * PH * PW + 255) / 256;
        const int base = c_lo * PH * PW, end = c_hi * PH * PW;
        const bool is_frame = c_lo < 3;
        float raw[MAXIT];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int i = base + tid + it * 256;
            const int c = i / (PH * PW);
            const int r = i - c * PH * PW;
            const int y = r / PW, x = r - y * PW;
            const int ry = gy0 + y - p.pad_top, rx = gx0 + x - p.pad_left;
            const bool inside = i < end && x < PW - 1 && (unsigned)ry < (unsigned)p.H0 && (unsigned)rx < (unsigned)p.W0;
            const int plane = is_frame ? c : n;
            const int off = inside ? ((plane * p.H0 + ry) * p.W0 + rx) * 4 : 0x7ffffff0;
            raw[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(is_frame ? rs_frame : rs_mask, off, 0, 0));
        }
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int i = base + tid + it * 256;      // (no `break` here: it made hipcc index raw[] dynamically, i.e. spill it to scratch)
            const int c = i / (PH * PW);
            const int r = i - c * PH * PW;
            const int y = r / PW, x = r - y * PW;
            const int gy = gy0 + y, gx = gx0 + x;
            float v = 0.f;
            if (x < PW - 1 && (unsigned)gy < (unsigned)p.Hp && (unsigned)gx < (unsigned)p.Wp) {
                if (c < 3) v = (raw[it] - p.mean[c]) / p.std[c];      // (a load from the kernel-argument segment; hoisting the six
                // constants into selects made hipcc build a private array and spill the whole descriptor to scratch: 41 -> 64 us)
                else v = (c == 3) ? raw[it] : fminf(fmaxf(1.f - raw[it], 0.f), 1.f);
            }
            if (i < end) sP[((c < 3 ? c : c - 3) * PH + y) * PROW + (x & 1) * PP + (x >> 1)] = v;
        }
    };
    load_planes(0, 3, 0);
    __syncthreads();

    // this wave: 32 pixels = tile rows 2*wave, 2*wave+1; lane's pixel for the A operand
    const int py = 2 * wave + (li >> 4), px = li & 15;
    const float* pa = sP + (2 * py) * PROW + lh * PP + px;   // + (c*PH+kh)*PROW + q   (column 2*px + 2*q + lh)
    const float* pb = sW + lh * 64 + li;                     // + (2*s)*64 + 32*tn

    f32x16 accF0, accF1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accF0[r] = 0.f; accF1[r] = 0.f; }
    // One filter row (kh) = 4 k-steps = 8 MFMAs.  Its 12 operand words are read a whole row ahead of the MFMAs that use
    // them: left to itself hipcc reads each step's two filter words into the same registers right in front of its two
    // MFMAs -- every 128 cycles of matrix work then waited out an LDS round trip (mfma_util 0.27).
    auto mac_planes = [&](f32x16& a0, f32x16& a1, int c) {
        float av[2][4], b0v[2][4], b1v[2][4];
        auto fetch = [&](int kh, int slot) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = (c * 7 + kh) * 4 + q;          // k-step: k = 2s + lh
                av[slot][q] = pa[((c < 3 ? c : c - 3) * PH + kh) * PROW + q];
                b0v[slot][q] = pb[(2 * s) * 64];
                b1v[slot][q] = pb[(2 * s) * 64 + 32];
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            if (kh + 1 < 7) fetch(kh + 1, (kh + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);               // keep the next row's reads in front of this row's MFMAs
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kh & 1][q], b0v[kh & 1][q], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kh & 1][q], b1v[kh & 1][q], a1, 0, 0, 0);
            }
        }
    };
#pragma unroll
    for (int c = 0; c < 3; ++c) mac_planes(accF0, accF1, c);

    // epilogue buffer: the 128 x 64 tile is transposed through LDS so that a lane owns 4 consecutive channels of one
    // pixel (16-byte stores).  It lives in the image of the frame planes' filters, which are dead from here on.
    constexpr int CP = 68;                                  // padded pitch (floats): 128 * 68 <= 168 * 64
    float* sC = sW;
    const int c4 = tid & 15;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + c4 * 4);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + c4 * 4);
    for (int n = 0; n < p.N; ++n) {
        f32x16 acc0 = accF0, acc1 = accF1;
        if constexpr (CIN == 5) {
            __syncthreads();                                 // previous object's mask planes / epilogue buffer are free
            load_planes(3, 5, n);
            __syncthreads();
            mac_planes(acc0, acc1, 3);
            mac_planes(acc0, acc1, 4);
        } else {
            __syncthreads();
        }
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pix = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;      // 0..127 within the block tile
                sC[pix * CP + tn * 32 + li] = tn == 0 ? acc0[r] : acc1[r];
            }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int pix = (tid >> 4) + 16 * k;            // wave-tile pixel: rows 2*(pix>>5) + ((pix&31)>>4), col pix&15
            const int oy = oy0 + 2 * (pix >> 5) + ((pix & 31) >> 4), ox = ox0 + (pix & 15);
            if (oy < p.Ho && ox < p.Wo) {
                f32x4 v = *reinterpret_cast<const f32x4*>(sC + pix * CP + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] * sc[e] + sh[e], 0.f);
                *reinterpret_cast<f32x4*>(p.out + (((size_t)n * p.Ho + oy) * p.Wo + ox) * 64 + c4 * 4) = v;
            }
        }
    }
}

__global__ void maxpool3x3s2_kernel(const float* __restrict__ in, float* __restrict__ out,
                                    int N, int H, int W, int C, int Ho, int Wo) {
    const int c4n = C / 4;
    const size_t total = (size_t)N * Ho * Wo * c4n;
    for (size_t i = vfn_xcd_block(blockIdx.x, gridDim.x) * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = i % c4n;
        size_t t = i / c4n;
        const int ox = t % Wo; t /= Wo;
        const int oy = t % Ho;
        const int n = t / Ho;
        // Branch-free (round 5): a tap outside the image is CLAMPED onto the border row / column, which lies inside the same 3x3 window
        // (2 oy - 1 = -1 -> 0 = 2 oy; 2 oy + 1 = H -> H - 1 = 2 oy), and a maximum does not change when one of its arguments is repeated:
        // nine unconditional loads in flight instead of nine tested ones with a wait each.
        f32x4 v[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int y = min(max(oy * 2 - 1 + dy, 0), H - 1);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int x = min(max(ox * 2 - 1 + dx, 0), W - 1);
                v[dy * 3 + dx] = *reinterpret_cast<const f32x4*>(in + (((size_t)n * H + y) * W + x) * C + c4 * 4);
            }
        }
        f32x4 m = v[0];
#pragma unroll
        for (int k = 1; k < 9; ++k) { m.x = fmaxf(m.x, v[k].x); m.y = fmaxf(m.y, v[k].y); m.z = fmaxf(m.z, v[k].z); m.w = fmaxf(m.w, v[k].w); }
        *reinterpret_cast<f32x4*>(out + i * 4) = m;
    }
}

template <int CIN>
int launch_stem(const vfn_stem_desc& d, hipStream_t s) {
    constexpr int KP = CIN * 7 * 8;
    // filters + 3 patch planes (the mask planes reuse the frame planes' space): 81 760 B for CIN = 5 -- two workgroups per
    // CU, so that one's load / epilogue phases run under the other's MFMAs (with 5 patch planes it was one per CU)
    const size_t lds = (size_t)(KP * 64 + 3 * PH * PROW) * sizeof(float);
    static bool attr_set = false;
    static_assert(PH * PROW * 3 * 4 + 5 * 7 * 8 * 64 * 4 <= 81920, "two stem workgroups must fit one CU's LDS");
    if (!attr_set && lds > 64 * 1024) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_kernel<CIN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int tiles = cdiv(d.Wo, TW) * cdiv(d.Ho, TH);          // (all images of the batch in one workgroup)
    hipLaunchKernelGGL((stem_kernel<CIN>), dim3(tiles), dim3(256), lds, s, d);
    return vfn_check_launch();
}

}  // namespace

extern "C" int vfn_stem_conv7x7_f32(const vfn_stem_desc* d, void* stream) {
    if (!d || !d->frame || !d->w || !d->out || !d->scale || !d->shift) return VFN_ERR_ARG;
    if (d->Hp % 2 || d->Wp % 2 || d->Ho != d->Hp / 2 || d->Wo != d->Wp / 2) return VFN_ERR_ARG;
    if (d->N < 1) return VFN_ERR_ARG;
    if (d->cin == 3) return launch_stem<3>(*d, (hipStream_t)stream);
    if (d->cin == 5) { if (!d->mask) return VFN_ERR_ARG; return launch_stem<5>(*d, (hipStream_t)stream); }
    return VFN_ERR_ARG;
}

extern "C" int vfn_maxpool3x3s2_nhwc_f32(const float* in, float* out, int N, int H, int W, int C, void* stream) {
    if (!in || !out || C % 4) return VFN_ERR_ARG;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t total = (size_t)N * Ho * Wo * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, N, H, W, C, Ho, Wo);
    return vfn_check_launch();
}
