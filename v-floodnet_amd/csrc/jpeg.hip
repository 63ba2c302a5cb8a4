// JPEG frames onto the device: replaces the PIL decode + torchvision ToTensor of
// Video_DS.__getitem__ (video_module/dataset/Water_DS.py:105-109, myutils/data.py:87-90) and the frame decode of
// test_video_seg.py:74,105.
//
//   host   vfn_jpeg_entropy_decode   marker parsing + Huffman decoding of a baseline (SOF0 / SOF1, 8-bit) JPEG into
//                                    quantised coefficient blocks -- the only inherently serial part of the format
//   device vfn_jpeg_idct_u8          dequantisation + the accurate integer inverse DCT (libjpeg jidctint.c "islow",
//                                    CONST_BITS 13 / PASS1_BITS 2) -> one uint8 plane per component
//          vfn_jpeg_to_tensor_f32    "fancy" (triangle) chroma upsampling for 4:2:0 / 4:2:2 (libjpeg jdsample.c
//                                    h2v2_fancy_upsample / h2v1_fancy_upsample), YCbCr -> RGB with libjpeg's 16-bit
//                                    fixed-point tables (jdcolor.c), and torchvision ToTensor (x / 255) -> f32 [3][H][W]
//
// The device arithmetic is libjpeg's integer arithmetic operation for operation, so the result equals what PIL
// (libjpeg-turbo, whose SIMD paths are bit-exact with these C definitions) hands to ToTensor.
// Byte / integer work: HBM-bound, no matrix cores.
#include "common.h"
#include "../../include/vfn_hip.h"
#include <string.h>

#include "host/jpeg_entropy.h"

// (documented in host/jpeg_entropy.h)
extern "C" int vfn_jpeg_entropy_decode(const unsigned char* data, long long size, short* coef, long long coef_cap,
                                       unsigned short* qt, int* info) {
    return vfn_host::jpeg_entropy_decode(data, size, coef, coef_cap, qt, info);
}

namespace {

// ------------------------------------------------------------------------------------------- device: IDCT
#define VFN_DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

// one 8-point pass of jidctint.c jpeg_idct_islow; `shift` = CONST_BITS - PASS1_BITS (columns) or CONST_BITS + PASS1_BITS + 3 (rows)
__device__ __forceinline__ void idct8(const int* in, int* out, int shift) {
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * 4433;
    int tmp2 = z1 + z3 * (-15137);
    int tmp3 = z1 + z2 * 6270;
    z2 = in[0]; z3 = in[4];
    int tmp0 = (z2 + z3) << 13;
    int tmp1 = (z2 - z3) << 13;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * 9633;
    tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
    z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    out[0] = VFN_DESCALE(tmp10 + tmp3, shift); out[7] = VFN_DESCALE(tmp10 - tmp3, shift);
    out[1] = VFN_DESCALE(tmp11 + tmp2, shift); out[6] = VFN_DESCALE(tmp11 - tmp2, shift);
    out[2] = VFN_DESCALE(tmp12 + tmp1, shift); out[5] = VFN_DESCALE(tmp12 - tmp1, shift);
    out[3] = VFN_DESCALE(tmp13 + tmp0, shift); out[4] = VFN_DESCALE(tmp13 - tmp0, shift);
}

// 8 threads per block of coefficients: thread j does column j (pass 1), then row j (pass 2); the 8x8 workspace goes
// through LDS (pitch 9: conflict-free).  256 threads = 32 blocks per workgroup.
__global__ __launch_bounds__(256)
void jpeg_idct_kernel(const short* __restrict__ coef, const unsigned short* __restrict__ qt, unsigned char* __restrict__ plane,
                      int bpr, int brows, int pitch) {
    __shared__ int ws[32][8][9];
    const int tid = threadIdx.x, j = tid & 7, lb = tid >> 3;
    const long long b = (long long)blockIdx.x * 32 + lb;
    const long long nblocks = (long long)bpr * brows;
    const bool live = b < nblocks;
    int col[8], res[8];
    if (live) {
        const short* src = coef + b * 64;
#pragma unroll
        for (int r = 0; r < 8; ++r) col[r] = (int)src[r * 8 + j] * (int)qt[r * 8 + j];       // DEQUANTIZE
        idct8(col, res, 13 - 2);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[lb][r][j] = res[r];
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int c = 0; c < 8; ++c) col[c] = ws[lb][j][c];
        idct8(col, res, 13 + 2 + 3);
        const int by = (int)(b / bpr), bx = (int)(b - (long long)by * bpr);
        unsigned char* dst = plane + (size_t)(by * 8 + j) * pitch + bx * 8;
        unsigned long long packed = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            int v = res[c] + 128;                                   // range_limit (CENTERJSAMPLE)
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            packed |= (unsigned long long)v << (8 * c);
        }
        *reinterpret_cast<unsigned long long*>(dst) = packed;       // (pitch and bx*8 are multiples of 8)
    }
}

// ------------------------------------------------------------------------------------------- device: upsample + colour
struct JpegPlanes {
    const unsigned char* y; const unsigned char* cb; const unsigned char* cr;
    int pitch_y, pitch_c;
    int W, H;              // image size
    int cw, chh;           // real (downsampled) chroma width / height: ceil(W*h_c/hmax), ceil(H*v_c/vmax)
    int hs, vs;            // luma / chroma sampling ratio: 1 or 2 each
    int ncomp;
};

// chroma sample at full-resolution pixel (x, y): libjpeg's fancy upsampling as closed forms with clamped neighbours
__device__ __forceinline__ int chroma_at(const unsigned char* pl, int pitch, int cw, int chh, int hs, int vs, int x, int y) {
    if (hs == 1 && vs == 1) return pl[(size_t)y * pitch + x];
    if (hs == 2 && cw <= 2) return pl[(size_t)(y / vs) * pitch + (x >> 1)];     // jdsample.c: fancy only if downsampled_width > 2
    if (hs == 2 && vs == 1) {                                       // h2v1_fancy_upsample
        const unsigned char* row = pl + (size_t)y * pitch;
        const int i = x >> 1;
        if (x & 1) { const int n = i + 1 < cw ? i + 1 : cw - 1; return (3 * row[i] + row[n] + 2) >> 2; }
        const int pv = i > 0 ? i - 1 : 0;
        return (3 * row[i] + row[pv] + 1) >> 2;
    }
    if (hs == 2 && vs == 2) {                                       // h2v2_fancy_upsample
        const int r = y >> 1, i = x >> 1;
        int rn = (y & 1) ? r + 1 : r - 1;                           // the nearer neighbouring row
        rn = rn < 0 ? 0 : (rn > chh - 1 ? chh - 1 : rn);
        const unsigned char* r0 = pl + (size_t)r * pitch;
        const unsigned char* r1 = pl + (size_t)rn * pitch;
        const int cur = 3 * r0[i] + r1[i];
        if (x & 1) { const int n = i + 1 < cw ? i + 1 : cw - 1; return (3 * cur + (3 * r0[n] + r1[n]) + 7) >> 4; }
        const int pv = i > 0 ? i - 1 : 0;
        return (3 * cur + (3 * r0[pv] + r1[pv]) + 8) >> 4;
    }
    // hs == 1, vs == 2: h1v2_fancy_upsample (libjpeg-turbo >= 1.5)
    const int r = y >> 1;
    int rn = (y & 1) ? r + 1 : r - 1;
    rn = rn < 0 ? 0 : (rn > chh - 1 ? chh - 1 : rn);
    const int bias = (y & 1) ? 2 : 1;
    return (3 * pl[(size_t)r * pitch + x] + pl[(size_t)rn * pitch + x] + bias) >> 2;
}

__global__ __launch_bounds__(256)
void jpeg_to_tensor_kernel(JpegPlanes p, float* __restrict__ out_f, unsigned char* __restrict__ out_u8) {
    const size_t n = (size_t)p.W * p.H;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / p.W), x = (int)(i - (size_t)y * p.W);
        const int Y = p.y[(size_t)y * p.pitch_y + x];
        int R = Y, G = Y, B = Y;
        if (p.ncomp == 3) {
            const int cb = chroma_at(p.cb, p.pitch_c, p.cw, p.chh, p.hs, p.vs, x, y) - 128;
            const int cr = chroma_at(p.cr, p.pitch_c, p.cw, p.chh, p.hs, p.vs, x, y) - 128;
            // jdcolor.c build_ycc_rgb_table: SCALEBITS 16, ONE_HALF 32768; arithmetic shifts
            R = Y + ((91881 * cr + 32768) >> 16);
            B = Y + ((116130 * cb + 32768) >> 16);
            G = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
            R = R < 0 ? 0 : (R > 255 ? 255 : R);
            G = G < 0 ? 0 : (G > 255 ? 255 : G);
            B = B < 0 ? 0 : (B > 255 ? 255 : B);
        }
        if (out_f) {
            out_f[i] = (float)R / 255.f;                            // ToTensor: IEEE division, as tensor.float().div(255)
            out_f[n + i] = (float)G / 255.f;
            out_f[2 * n + i] = (float)B / 255.f;
        }
        if (out_u8) { out_u8[3 * i] = (unsigned char)R; out_u8[3 * i + 1] = (unsigned char)G; out_u8[3 * i + 2] = (unsigned char)B; }
    }
}

}  // namespace

extern "C" int vfn_jpeg_idct_u8(const short* coef, const unsigned short* qt, unsigned char* plane, int blocks_per_row,
                                int block_rows, int pitch, void* stream) {
    if (!coef || !qt || !plane || blocks_per_row < 1 || block_rows < 1 || pitch < blocks_per_row * 8 || pitch % 8) return VFN_ERR_ARG;
    const long long nblocks = (long long)blocks_per_row * block_rows;
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((nblocks + 31) / 32)), dim3(256), 0, (hipStream_t)stream,
                       coef, qt, plane, blocks_per_row, block_rows, pitch);
    return vfn_check_launch();
}

extern "C" int vfn_jpeg_to_tensor_f32(const unsigned char* y, const unsigned char* cb, const unsigned char* cr, int pitch_y,
                                      int pitch_c, int W, int H, int hs, int vs, int ncomp, float* out_f32,
                                      unsigned char* out_u8, void* stream) {
    if (!y || W < 1 || H < 1 || (ncomp != 1 && ncomp != 3) || (!out_f32 && !out_u8)) return VFN_ERR_ARG;
    if (ncomp == 3 && (!cb || !cr || hs < 1 || hs > 2 || vs < 1 || vs > 2)) return VFN_ERR_ARG;
    JpegPlanes p;
    p.y = y; p.cb = cb; p.cr = cr; p.pitch_y = pitch_y; p.pitch_c = pitch_c; p.W = W; p.H = H;
    p.hs = hs; p.vs = vs; p.ncomp = ncomp;
    p.cw = (W + hs - 1) / hs; p.chh = (H + vs - 1) / vs;
    const size_t n = (size_t)W * H;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(jpeg_to_tensor_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, out_f32, out_u8);
    return vfn_check_launch();
}
