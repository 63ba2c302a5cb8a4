// Host-only part of the JPEG input path: marker parsing + Huffman decoding of a baseline JPEG (plain C++, no HIP).
// Included by jpeg.hip (-> libvfn_hip.so) and by host/host_only.cpp (-> the AddressSanitizer / UBSan CPU build,
// `make asan`, tests/test_sanitize.py).  It parses untrusted bytes in DataLoader workers: every read is bounds-checked
// against the segment and the file, and a malformed file yields a negative return code, never an out-of-range access.
#pragma once
#include <string.h>

namespace vfn_host {

// ------------------------------------------------------------------------------------------- host: entropy decoding
const unsigned char kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                   15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
    bool present = false;
    unsigned char bits[17] = {};
    unsigned char vals[256] = {};
    // canonical decode tables (ITU T.81 F.2.2.3)
    int mincode[17] = {}, maxcode[18] = {}, valptr[17] = {};
    // 9-bit lookahead: (length << 8) | symbol, 0 = longer code
    unsigned short look[512] = {};
    // false: the code lengths do not describe a prefix code (more codes of a length than that length has left)
    bool build() {
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
            if (code > (1 << si)) return false;                     // (libjpeg jdhuff.c: "bad code lengths")
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
        return true;
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


// info[]: 0 width, 1 height, 2 ncomp, 3 hmax, 4 vmax, 5 mcu_cols, 6 mcu_rows,
//         7+4c: h_c, v_c, blocks per row of component c, block rows of component c;  19+c: quant table slot of comp c
// qt: unsigned short [4][64] in NATURAL order.  coef: short, component c at offset coef_off[c] (in shorts), layout
// [block_rows_c][blocks_per_row_c][64] natural order, quantised.  Returns 0, or a negative code:
//   -1 not a JPEG / truncated, -2 unsupported (progressive, arithmetic, 12-bit, CMYK, non-interleaved colour scans,
//   sampling factors other than 1 or 2), -3 coef buffer too small (needed shorts in info[23]), -4 corrupt entropy data
inline int jpeg_entropy_decode(const unsigned char* data, long long size, short* coef, long long coef_cap,
                                       unsigned short* qt, int* info) {
    if (!data || size < 4 || !qt || !info) return -1;
    if (data[0] != 0xFF || data[1] != 0xD8) return -1;
    HuffTable dc[4], ac[4];
    unsigned short q[4][64];
    bool have_q[4] = {false, false, false, false};
    int width = 0, height = 0, ncomp = 0, restart = 0;
    int cid[4], ch[4], cv[4], ctq[4], ctd[4] = {0, 0, 0, 0}, cta[4] = {0, 0, 0, 0};
    bool have_sof = false, have_sos = false;
    const unsigned char* p = data + 2;
    const unsigned char* end = data + size;
    while (p + 4 <= end) {
        if (p[0] != 0xFF) return -1;
        while (p + 2 < end && p[1] == 0xFF) ++p;                   // fill bytes (p[1] stays inside the file)
        if (p + 2 > end) return -1;
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
                if (seg + (pq ? 128 : 64) > seg_end) return -1;      // a short table at the end of the segment
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
                if (seg + 16 > seg_end) return -1;
                HuffTable& t = tc ? ac[th] : dc[th];
                int n = 0;
                t.bits[0] = 0;
                for (int i = 1; i <= 16; ++i) { t.bits[i] = seg[i - 1]; n += seg[i - 1]; }
                seg += 16;
                if (n > 256 || seg + n > seg_end) return -1;
                memcpy(t.vals, seg, n);
                seg += n;
                t.present = t.build();
                if (!t.present) return -1;
            }
        } else if (m == 0xC0 || m == 0xC1) {                        // SOF0 / SOF1: sequential DCT, Huffman
            if (len < 8) return -1;
            if (seg[0] != 8) return -2;                             // 8-bit samples only
            height = rd16(seg + 1);
            width = rd16(seg + 3);
            ncomp = seg[5];
            if ((ncomp != 1 && ncomp != 3) || width < 1 || height < 1) return -2;
            if (len < 8 + 3 * ncomp) return -1;
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
            if (len < 4) return -1;
            restart = rd16(seg);
        } else if (m == 0xDA) {                                     // SOS: the (single) scan of a baseline file
            if (!have_sof) return -1;
            if (len < 3) return -1;
            const int ns = seg[0];
            if (ns != ncomp) return -2;                             // non-interleaved colour scans are not produced by the
            if (len < 6 + 2 * ns) return -1;
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
            have_sos = true;
            break;
        }
        p = seg_end;
    }
    if (!have_sof || !have_sos || p >= end) return -1;              // (a file that ends before its scan header)
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
    // A frame, not a decompression bomb: a ~200-byte forged header may declare 65535 x 16000 and have the caller allocate and
    // clear gigabytes of coefficients before the entropy data turns out truncated.  64 Mpx is well past any video frame and
    // in line with PIL's MAX_IMAGE_PIXELS guard (89 Mpx); larger files fall through to PIL (-2 = outside the subset).
    if ((long long)width * height > 64LL * 1024 * 1024) return -2;
    if (total > 0x7fffffffLL) return -2;                           // (info[23] is an int)
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
                        pred[c] = (int)((unsigned)pred[c] + (unsigned)diff);   // (corrupt data may wrap; never UB)
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


}  // namespace vfn_host
