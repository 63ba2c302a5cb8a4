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

namespace {

// ------------------------------------------------------------------------------------------- host: entropy decoding
const unsigned char kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                   15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
    bool present = false;
    unsigned char bits[17];
    unsigned char vals[256];
    // canonical decode tables (ITU T.81 F.2.2.3)
    int mincode[17], maxcode[18], valptr[17];
    // 9-bit lookahead: (length << 8) | symbol, 0 = longer code
    unsigned short look[512];
    void build() {
        int code = 0, k = 0;
        unsigned short huffcode[257];
        unsigned char huffsize[257];
        for (int l = 1; l <= 16; ++l)
            for (int i = 0; i < bits[l]; ++i) huffsize[k++] = (unsigned char)l;
        const int n = k;
        k = 0;
        int si = n ? huffsize[0] : 0;
        while (k < n) {
            while (k < n && huffsize[k] == si) huffcode[k++] = (unsigned short)code++;
            code <<= 1;
            ++si;
        }
        int p = 0;
        for (int l = 1; l <= 16; ++l) {
            if (bits[l]) {
                valptr[l] = p;
                mincode[l] = huffcode[p];
                p += bits[l];
                maxcode[l] = huffcode[p - 1];
            } else {
                maxcode[l] = -1;
                mincode[l] = 0;
                valptr[l] = 0;
            }
        }
        maxcode[17] = 0x7fffffff;
        memset(look, 0, sizeof(look));
        p = 0;
        for (int l = 1; l <= 9; ++l)
            for (int i = 0; i < bits[l]; ++i, ++p) {
                const int first = huffcode[p] << (9 - l);
                for (int j = 0; j < (1 << (9 - l)); ++j) look[first + j] = (unsigned short)((l << 8) | vals[p]);
            }
    }
};

struct BitReader {
    const unsigned char* p;
    const unsigned char* end;
    unsigned long long acc = 0;
    int nbits = 0;
    bool hit_marker = false;
    void fill() {
        while (nbits <= 48) {
            int b = 0;
            if (!hit_marker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;          // stuffed zero
                    else { hit_marker = true; b = 0; }                 // a marker: feed zeros from here on
                } else ++p;
            }
            acc = (acc << 8) | (unsigned)b;
            nbits += 8;
        }
    }
    inline int peek(int n) { if (nbits < n) fill(); return (int)((acc >> (nbits - n)) & ((1u << n) - 1)); }
    inline void skip(int n) { nbits -= n; }
    inline int get(int n) { const int v = peek(n); skip(n); return v; }
    void reset_at(const unsigned char* q) { p = q; acc = 0; nbits = 0; hit_marker = false; }
};

inline int huff_decode(BitReader& br, const HuffTable& t) {
    const int look = t.look[br.peek(9)];
    if (look) { br.skip(look >> 8); return look & 255; }
    int code = br.get(9), l = 9;
    while (true) {
        ++l;
        if (l > 16) return -1;
        code = (code << 1) | br.get(1);
        if (t.maxcode[l] >= 0 && code <= t.maxcode[l] && code >= t.mincode[l]) break;
    }
    return t.vals[t.valptr[l] + code - t.mincode[l]];
}

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

inline int rd16(const unsigned char* p) { return (p[0] << 8) | p[1]; }

}  // namespace

// info[]: 0 width, 1 height, 2 ncomp, 3 hmax, 4 vmax, 5 mcu_cols, 6 mcu_rows,
//         7+4c: h_c, v_c, blocks per row of component c, block rows of component c;  19+c: quant table slot of comp c
// qt: unsigned short [4][64] in NATURAL order.  coef: short, component c at offset coef_off[c] (in shorts), layout
// [block_rows_c][blocks_per_row_c][64] natural order, quantised.  Returns 0, or a negative code:
//   -1 not a JPEG / truncated, -2 unsupported (progressive, arithmetic, 12-bit, CMYK, non-interleaved colour scans,
//   sampling factors other than 1 or 2), -3 coef buffer too small (needed shorts in info[23]), -4 corrupt entropy data
extern "C" int vfn_jpeg_entropy_decode(const unsigned char* data, long long size, short* coef, long long coef_cap,
                                       unsigned short* qt, int* info) {
    if (!data || size < 4 || !qt || !info) return -1;
    if (data[0] != 0xFF || data[1] != 0xD8) return -1;
    HuffTable dc[4], ac[4];
    unsigned short q[4][64];
    bool have_q[4] = {false, false, false, false};
    int width = 0, height = 0, ncomp = 0, restart = 0;
    int cid[4], ch[4], cv[4], ctq[4], ctd[4] = {0, 0, 0, 0}, cta[4] = {0, 0, 0, 0};
    bool have_sof = false;
    const unsigned char* p = data + 2;
    const unsigned char* end = data + size;
    while (p + 4 <= end) {
        if (p[0] != 0xFF) return -1;
        while (p < end && p[1] == 0xFF) ++p;                       // fill bytes
        const int m = p[1];
        p += 2;
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return -1;                                   // EOI before any scan
        if (p + 2 > end) return -1;
        const int len = rd16(p);
        if (len < 2 || p + len > end) return -1;
        const unsigned char* seg = p + 2;
        const unsigned char* seg_end = p + len;
        if (m == 0xDB) {                                            // DQT
            while (seg < seg_end) {
                const int pq = seg[0] >> 4, tq = seg[0] & 15;
                ++seg;
                if (tq > 3 || pq > 1) return -2;
                for (int i = 0; i < 64; ++i) {
                    const int v = pq ? rd16(seg + 2 * i) : seg[i];
                    q[tq][kZigzag[i]] = (unsigned short)v;
                }
                seg += pq ? 128 : 64;
                have_q[tq] = true;
            }
        } else if (m == 0xC4) {                                     // DHT
            while (seg < seg_end) {
                const int tc = seg[0] >> 4, th = seg[0] & 15;
                ++seg;
                if (tc > 1 || th > 3) return -2;
                HuffTable& t = tc ? ac[th] : dc[th];
                int n = 0;
                t.bits[0] = 0;
                for (int i = 1; i <= 16; ++i) { t.bits[i] = seg[i - 1]; n += seg[i - 1]; }
                seg += 16;
                if (n > 256 || seg + n > seg_end) return -1;
                memcpy(t.vals, seg, n);
                seg += n;
                t.present = true;
                t.build();
            }
        } else if (m == 0xC0 || m == 0xC1) {                        // SOF0 / SOF1: sequential DCT, Huffman
            if (seg[0] != 8) return -2;                             // 8-bit samples only
            height = rd16(seg + 1);
            width = rd16(seg + 3);
            ncomp = seg[5];
            if ((ncomp != 1 && ncomp != 3) || width < 1 || height < 1) return -2;
            for (int c = 0; c < ncomp; ++c) {
                cid[c] = seg[6 + 3 * c];
                ch[c] = seg[7 + 3 * c] >> 4;
                cv[c] = seg[7 + 3 * c] & 15;
                ctq[c] = seg[8 + 3 * c];
                if (ch[c] < 1 || ch[c] > 2 || cv[c] < 1 || cv[c] > 2 || ctq[c] > 3) return -2;
            }
            have_sof = true;
        } else if ((m >= 0xC2 && m <= 0xCF) && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return -2;                                              // progressive / lossless / arithmetic
        } else if (m == 0xEE && len >= 14 && memcmp(seg, "Adobe", 5) == 0 && seg[11] != 1) {
            return -2;                                              // Adobe transform 0 / 2: RGB or YCCK, not YCbCr
        } else if (m == 0xDD) {
            restart = rd16(seg);
        } else if (m == 0xDA) {                                     // SOS: the (single) scan of a baseline file
            if (!have_sof) return -1;
            const int ns = seg[0];
            if (ns != ncomp) return -2;                             // non-interleaved colour scans are not produced by the
            for (int i = 0; i < ns; ++i) {                          // encoders in use (libjpeg, OpenCV): unsupported
                const int id = seg[1 + 2 * i];
                int c = -1;
                for (int k = 0; k < ncomp; ++k) if (cid[k] == id) c = k;
                if (c != i) return -2;
                ctd[c] = seg[2 + 2 * i] >> 4;
                cta[c] = seg[2 + 2 * i] & 15;
                if (ctd[c] > 3 || cta[c] > 3 || !dc[ctd[c]].present || !ac[cta[c]].present || !have_q[ctq[c]]) return -1;
            }
            p = seg_end;
            break;
        }
        p = seg_end;
    }
    if (!have_sof || p >= end) return -1;
    int hmax = 1, vmax = 1;
    if (ncomp == 1) { ch[0] = 1; cv[0] = 1; }                      // a single-component scan is never interleaved
    for (int c = 0; c < ncomp; ++c) { hmax = ch[c] > hmax ? ch[c] : hmax; vmax = cv[c] > vmax ? cv[c] : vmax; }
    if (ncomp == 3 && (ch[0] != hmax || cv[0] != vmax || ch[1] != 1 || cv[1] != 1 || ch[2] != 1 || cv[2] != 1)) return -2;
    const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
    const int mcu_cols = (width + mcu_w - 1) / mcu_w, mcu_rows = (height + mcu_h - 1) / mcu_h;
    long long off[4], total = 0;
    int bpr[4], brows[4];
    for (int c = 0; c < ncomp; ++c) {
        bpr[c] = mcu_cols * ch[c];
        brows[c] = mcu_rows * cv[c];
        off[c] = total;
        total += (long long)bpr[c] * brows[c] * 64;
    }
    info[0] = width; info[1] = height; info[2] = ncomp; info[3] = hmax; info[4] = vmax; info[5] = mcu_cols; info[6] = mcu_rows;
    for (int c = 0; c < 3; ++c) {
        const bool ok = c < ncomp;
        info[7 + 4 * c] = ok ? ch[c] : 0; info[8 + 4 * c] = ok ? cv[c] : 0;
        info[9 + 4 * c] = ok ? bpr[c] : 0; info[10 + 4 * c] = ok ? brows[c] : 0;
        info[19 + c] = ok ? c : 0;
    }
    info[22] = restart;
    info[23] = (int)total;
    for (int c = 0; c < ncomp; ++c) memcpy(qt + 64 * c, q[ctq[c]], 64 * sizeof(unsigned short));
    if (!coef || coef_cap < total) return -3;
    memset(coef, 0, (size_t)total * sizeof(short));

    BitReader br;
    br.p = p; br.end = end;
    int pred[4] = {0, 0, 0, 0};
    int until_restart = restart;
    for (int my = 0; my < mcu_rows; ++my)
        for (int mx = 0; mx < mcu_cols; ++mx) {
            if (restart && until_restart == 0) {
                // byte-align, expect RSTn
                const unsigned char* qn = br.p;
                while (qn + 1 < end && !(qn[0] == 0xFF && qn[1] >= 0xD0 && qn[1] <= 0xD7)) ++qn;
                if (qn + 1 >= end) return -4;
                br.reset_at(qn + 2);
                pred[0] = pred[1] = pred[2] = pred[3] = 0;
                until_restart = restart;
            }
            for (int c = 0; c < ncomp; ++c)
                for (int by = 0; by < cv[c]; ++by)
                    for (int bx = 0; bx < ch[c]; ++bx) {
                        short* blk = coef + off[c] + ((long long)(my * cv[c] + by) * bpr[c] + (mx * ch[c] + bx)) * 64;
                        int s = huff_decode(br, dc[ctd[c]]);
                        if (s < 0 || s > 11) return -4;
                        int diff = 0;
                        if (s) diff = extend(br.get(s), s);
                        pred[c] += diff;
                        blk[0] = (short)pred[c];
                        const HuffTable& at = ac[cta[c]];
                        for (int k = 1; k < 64;) {
                            const int rs = huff_decode(br, at);
                            if (rs < 0) return -4;
                            const int r = rs >> 4, sz = rs & 15;
                            if (sz == 0) {
                                if (r == 15) { k += 16; continue; }
                                break;                              // EOB
                            }
                            k += r;
                            if (k > 63) return -4;
                            blk[kZigzag[k]] = (short)extend(br.get(sz), sz);
                            ++k;
                        }
                    }
            if (restart) --until_restart;
        }
    return 0;
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
