// PNG encoding on the device: the pixel side of save_seg_mask (myutils/data.py:49-53, PIL mode-P PNG) and of
// save_overlay (myutils/data.py:78-84, cv2.imwrite of the BGR overlay).  The reference compresses on a host core
// (zlib inside PIL / OpenCV: 40-50 ms per 480p overlay); here the image never leaves the GPU uncompressed:
//
//   png_scan_kernel     PNG row filter (type 2 "Up" for 1-byte pixels, type 4 "Paeth" for RGB), RLE tokenisation
//                       (literals + distance-1 matches of length 3..258), symbol histogram, Adler-32 partials
//   png_code_kernel     length-limited Huffman code of the image's own histogram (sort + two-queue merge + zlib's
//                       overflow repair), canonical codes, the dynamic-block header (RFC 1951 3.2.7)
//   png_bits_kernel     bits per unit under that code;  png_offsets_kernel: exclusive scan -> bit offsets, Adler-32
//   png_emit_kernel     every unit writes its codes at its bit offset
//
// Unit of parallel work = up to 256 consecutive bytes of one filtered scanline (a run never crosses a unit, which costs
// a few bytes of compression and removes every serial dependency).  The result is ONE final deflate block; the host
// adds the 2-byte zlib header, the Adler-32 and the PNG chunk framing (CRC-32 over the compressed bytes).
// Any PNG decoder reproduces the pixels exactly -- the contract of the consumers (est_waterlevel.py:26-28 reads the
// masks through PIL).  HBM-bound byte work: no matrix cores involved.
#include "common.h"
#include "../../include/vfn_hip.h"
#include <stdlib.h>

namespace {

constexpr int SEG = 256;            // filtered bytes per unit
constexpr int LSTR = 260;           // LDS stride of a unit (65 words: conflict-free across lanes)
constexpr int NSYM = 286;           // literal/length alphabet
constexpr int UPB = 64;             // units per block in the per-unit kernels
constexpr int SCAN_NT = 256;        // threads of the scan kernel (all of them filter; the first UPB tokenise)

struct PngWork {                    // layout of the caller's work buffer
    unsigned* hist;                 // [288]   symbol counts (zeroed by the launcher)
    unsigned* code;                 // [288]   bit-reversed code | length << 16
    unsigned* meta;                 // [8]     0: header bits, 1: total bits, 2: adler
    unsigned* ubits;                // [units] bits per unit, then exclusive offsets
    unsigned* uadl;                 // [units][2] (sum d, sum (n-i) d) per unit
    unsigned char* filt;            // [units][SEG]
};

__host__ __device__ inline int units_per_row(int W, int bpp) { return (1 + W * bpp + SEG - 1) / SEG; }

__host__ __device__ inline PngWork carve(void* work, int units) {
    PngWork w;
    unsigned* p = reinterpret_cast<unsigned*>(work);
    w.hist = p; p += 288;
    w.code = p; p += 288;
    w.meta = p; p += 8;
    w.ubits = p; p += (units + 3) / 4 * 4;
    w.uadl = p; p += 2 * (size_t)units;
    w.filt = reinterpret_cast<unsigned char*>(p);
    return w;
}

// match length 3..258 -> (symbol - 257, extra bits, extra value)
__device__ __forceinline__ void len_symbol(int len, int& idx, int& ebits, int& eval) {
    if (len == 258) { idx = 28; ebits = 0; eval = 0; return; }
    if (len <= 10) { idx = len - 3; ebits = 0; eval = 0; return; }
    const int l = len - 3;                                   // >= 8
    const int e = 31 - __clz(l) - 2;                          // extra bits: 1..5
    idx = 4 * e + 4 + ((l >> e) & 3);
    ebits = e;
    eval = l & ((1 << e) - 1);
}

// Walk the tokens of one unit held in LDS (4-byte aligned, read a word at a time).  Greedy rule: the first byte of a
// run of equal bytes is a literal; the R-1 bytes behind it are ONE distance-1 match when R-1 >= 3 (a unit has at most
// 256 bytes, so R-1 <= 255 < 258) and literals otherwise.  One pass, no inner loop: every lane runs exactly n steps.
template <typename Lit, typename Match>
__device__ __forceinline__ void for_each_token(const unsigned char* d, int n, Lit lit, Match match) {
    const unsigned* dw = reinterpret_cast<const unsigned*>(d);
    int prev = -1, run = 0;
    auto flush = [&]() {
        if (run == 0) return;
        lit(prev);
        const int rest = run - 1;
        if (rest >= 3) match(rest);
        else { if (rest >= 1) lit(prev); if (rest == 2) lit(prev); }
    };
    unsigned w4 = 0;
    for (int i = 0; i < n; ++i) {
        if ((i & 3) == 0) w4 = dw[i >> 2];
        const int b = w4 & 255;
        w4 >>= 8;
        if (b == prev) ++run;
        else { flush(); prev = b; run = 1; }
    }
    flush();
}

__device__ __forceinline__ int paeth(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// unit u -> row, first filtered position, byte count
__device__ __forceinline__ void unit_span(int u, int upr, int rowlen, int& row, int& p0, int& n) {
    row = u / upr;
    p0 = (u - row * upr) * SEG;
    n = min(SEG, rowlen - p0);
}

// ---------------------------------------------------------------- filter + tokenise + histogram + adler partials
__global__ __launch_bounds__(SCAN_NT)
void png_scan_kernel(const unsigned char* __restrict__ raw, int H, int W, int bpp, PngWork wk, int units) {
    __shared__ __attribute__((aligned(16))) unsigned char s_d[UPB * LSTR];
    __shared__ unsigned s_hist[288];
    const int tid = threadIdx.x;
    for (int i = tid; i < 288; i += SCAN_NT) s_hist[i] = 0;
    const int u0 = blockIdx.x * UPB;
    const int upr = units_per_row(W, bpp), rowbytes = W * bpp, rowlen = 1 + rowbytes;
    // phase 1: the block's 64 units are filtered by all threads together, 4 consecutive bytes per thread and step
    // (consecutive threads -> consecutive bytes of a scanline: coalesced loads, independent iterations)
    for (int q = tid; q < UPB * (SEG / 4); q += SCAN_NT) {
        const int ul = q / (SEG / 4), i0 = (q - ul * (SEG / 4)) * 4;
        const int u = u0 + ul;
        if (u >= units) break;
        int row, p0, n;
        unit_span(u, upr, rowlen, row, p0, n);
        if (i0 >= n) continue;
        const unsigned char* cur = raw + (size_t)row * rowbytes;
        const unsigned char* up = row > 0 ? cur - rowbytes : nullptr;
        unsigned w4 = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int p = p0 + i0 + e;
            int f = 0;
            if (i0 + e < n) {
                if (p == 0) f = (bpp == 1) ? 2 : 4;                      // filter type byte
                else {
                    const int x = p - 1;
                    const int v = cur[x];
                    const int b = up ? up[x] : 0;
                    if (bpp == 1) f = (v - b) & 255;
                    else {
                        const int a = x >= bpp ? cur[x - bpp] : 0;
                        const int c = (up && x >= bpp) ? up[x - bpp] : 0;
                        f = (v - paeth(a, b, c)) & 255;
                    }
                }
            }
            w4 |= (unsigned)f << (8 * e);
        }
        *reinterpret_cast<unsigned*>(s_d + ul * LSTR + i0) = w4;
        *reinterpret_cast<unsigned*>(wk.filt + (size_t)u * SEG + i0) = w4;
    }
    __syncthreads();
    // phase 2: one thread per unit
    const int u = u0 + tid;
    if (tid < UPB && u < units) {
        int row, p0, n;
        unit_span(u, upr, rowlen, row, p0, n);
        const unsigned char* d = s_d + tid * LSTR;
        const unsigned* dw = reinterpret_cast<const unsigned*>(d);
        unsigned s1 = 0, s2 = 0, w4 = 0;
        for (int i = 0; i < n; ++i) {
            if ((i & 3) == 0) w4 = dw[i >> 2];
            const unsigned f = w4 & 255;
            w4 >>= 8;
            s1 += f;
            s2 += (unsigned)(n - i) * f;                                 // <= 256 * 255 * 256: fits
        }
        wk.uadl[2 * (size_t)u] = s1;
        wk.uadl[2 * (size_t)u + 1] = s2;
        for_each_token(d, n,
                       [&](int b) { atomicAdd(&s_hist[b], 1u); },
                       [&](int L) { int idx, eb, ev; len_symbol(L, idx, eb, ev); atomicAdd(&s_hist[257 + idx], 1u); });
    }
    __syncthreads();
    for (int i = tid; i < NSYM; i += SCAN_NT)
        if (s_hist[i]) atomicAdd(&wk.hist[i], s_hist[i]);
}

// ---------------------------------------------------------------- Huffman code of the histogram + block header
struct BitW {                                   // serial bit writer (header)
    unsigned* out; unsigned long long acc; int nb; int w;
    __device__ void put(unsigned v, int n) {
        acc |= (unsigned long long)v << nb; nb += n;
        while (nb >= 32) { out[w++] = (unsigned)acc; acc >>= 32; nb -= 32; }
    }
    __device__ int bits() const { return w * 32 + nb; }
    __device__ void flush_partial() { if (nb > 0) out[w] = (unsigned)acc; }
};

__device__ __forceinline__ unsigned rev_bits(unsigned v, int n) { return __brev(v) >> (32 - n); }

__global__ __launch_bounds__(512)
void png_code_kernel(PngWork wk, unsigned* __restrict__ out) {
    __shared__ unsigned s_key[512];
    __shared__ unsigned s_freq[2 * NSYM];
    __shared__ short s_parent[2 * NSYM];
    __shared__ unsigned char s_len[288];
    __shared__ int s_bl[16];
    __shared__ unsigned s_next[16];
    __shared__ int s_n;
    const int tid = threadIdx.x;
    // sort the used symbols by (count, symbol) ascending
    {
        unsigned f = tid < NSYM ? wk.hist[tid] : 0u;
        if (tid == 256) f += 1u;                               // the end-of-block symbol
        s_key[tid] = f ? ((f << 9) | (unsigned)tid) : 0xffffffffu;        // counts < 2^23
        if (tid < 288) s_len[tid] = 0;
        if (tid < 16) s_bl[tid] = 0;
    }
    __syncthreads();
    for (int k = 2; k <= 512; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int ixj = tid ^ j;
            if (ixj > tid) {
                const unsigned a = s_key[tid], b = s_key[ixj];
                const bool up = (tid & k) == 0;
                if ((a > b) == up) { s_key[tid] = b; s_key[ixj] = a; }
            }
            __syncthreads();
        }
    {   // number of used symbols
        const int used = (tid < NSYM && s_key[tid] != 0xffffffffu) ? 1 : 0;
        const int c = __syncthreads_count(used);
        if (tid == 0) s_n = c;
        if (tid < NSYM && used) s_freq[tid] = s_key[tid] >> 9;
    }
    __syncthreads();
    const int n = s_n;                                         // >= 2: a literal and the end-of-block symbol always exist
    if (tid == 0) {
        // two-queue merge: leaves [0,n) ascending, internal nodes [n, 2n-1) come out in ascending order as well.
        // The heads of both queues are kept in registers, so a step costs two LDS round trips, not eight.
        int lq = 0, iq = n, nn = n;
        unsigned lf = s_freq[0], inf_ = 0xffffffffu;           // head values (0xffffffff: queue empty)
        for (int m = 0; m < n - 1; ++m) {
            unsigned fa, fb;
            int a, b;
            if (lf <= inf_) { a = lq++; fa = lf; lf = lq < n ? s_freq[lq] : 0xffffffffu; }
            else { a = iq++; fa = inf_; inf_ = iq < nn ? s_freq[iq] : 0xffffffffu; }
            if (lf <= inf_) { b = lq++; fb = lf; lf = lq < n ? s_freq[lq] : 0xffffffffu; }
            else { b = iq++; fb = inf_; inf_ = iq < nn ? s_freq[iq] : 0xffffffffu; }
            const unsigned fs = fa + fb;
            s_freq[nn] = fs;
            s_parent[a] = (short)nn; s_parent[b] = (short)nn;
            if (inf_ == 0xffffffffu && iq == nn) inf_ = fs;    // the internal queue was empty: the new node is its head
            ++nn;
        }
        s_parent[nn - 1] = -1;
    }
    __syncthreads();
    // depth of every leaf: walk to the root (a few dozen steps at most), in parallel
    int my_len = 0;
    if (tid < n) {
        int j = tid, dpt = 0;
        while (s_parent[j] >= 0) { j = s_parent[j]; ++dpt; }
        my_len = dpt > 15 ? 15 : dpt;
        atomicAdd(&s_bl[my_len], 1);
    }
    __syncthreads();
    if (tid == 0) {
        // leaves deeper than 15 were lifted to 15: the Kraft sum now exceeds 1 by `excess` units of 2^-15.  zlib's
        // repair step (gen_bitlen): a leaf at the deepest level < 15 gets a sibling taken from level 15 -- each
        // step gives back exactly one unit
        int excess = -32768;
        for (int b = 1; b <= 15; ++b) excess += s_bl[b] << (15 - b);
        for (; excess > 0; --excess) {
            int bits = 14;
            while (s_bl[bits] == 0) --bits;
            --s_bl[bits]; s_bl[bits + 1] += 2; --s_bl[15];
        }
        s_bl[0] = 0;
        unsigned c = 0;
        for (int b = 1; b < 16; ++b) { c = (c + s_bl[b - 1]) << 1; s_next[b] = c; }
    }
    __syncthreads();
    // the least frequent symbols get the longest codes: sorted position i -> length
    if (tid < n) {
        int cum = 0, len = 1;
        for (int b = 15; b >= 1; --b) { cum += s_bl[b]; if (tid < cum) { len = b; break; } }
        s_len[s_key[tid] & 511] = (unsigned char)len;
    }
    __syncthreads();
    // canonical codes (RFC 1951 3.2.2): within a length, by symbol value; stored bit-reversed for the LSB-first stream
    if (tid < NSYM) {
        const int l = s_len[tid];
        unsigned v = 0;
        if (l) {
            int rank = 0;
            for (int t = 0; t < tid; ++t) rank += (s_len[t] == l);
            v = rev_bits(s_next[l] + rank, l) | ((unsigned)l << 16);
        }
        wk.code[tid] = v;
    }
    if (tid == 0) {
        // block header: BFINAL=1, BTYPE=10, HLIT=286, HDIST=1, HCLEN=19; code-length code fixed and complete:
        // symbols 0-11,17,18 -> 4 bits, 12-15 -> 5 bits, 16 unused
        BitW bw{out, 0ull, 0, 0};
        bw.put(1, 1); bw.put(2, 2);
        bw.put(NSYM - 257, 5); bw.put(0, 5); bw.put(15, 4);
        const unsigned char order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        for (int k = 0; k < 19; ++k) { const int sy = order[k]; bw.put(sy == 16 ? 0 : (sy >= 12 && sy <= 15 ? 5 : 4), 3); }
        auto cl_put = [&](int sy) {
            unsigned code; int l;
            if (sy <= 11) { code = sy; l = 4; } else if (sy == 17) { code = 12; l = 4; } else if (sy == 18) { code = 13; l = 4; }
            else { code = 28 + (sy - 12); l = 5; }
            bw.put(rev_bits(code, l), l);
        };
        int z = 0;
        auto flush_zeros = [&]() {
            while (z >= 11) { const int r = z > 138 ? 138 : z; cl_put(18); bw.put(r - 11, 7); z -= r; }
            if (z >= 3) { cl_put(17); bw.put(z - 3, 3); z = 0; }
            while (z > 0) { cl_put(0); --z; }
        };
        for (int sy = 0; sy < NSYM; ++sy) {
            const int l = s_len[sy];
            if (l == 0) { ++z; continue; }
            flush_zeros();
            cl_put(l);
        }
        flush_zeros();
        cl_put(1);                                             // the single distance code (distance 1), one bit
        bw.flush_partial();
        wk.meta[0] = bw.bits();
    }
}

// ---------------------------------------------------------------- bits per unit
__device__ __forceinline__ void load_unit(unsigned char* d, const unsigned char* filt, int u, int n) {
    const unsigned* src = reinterpret_cast<const unsigned*>(filt + (size_t)u * SEG);
    unsigned* dst = reinterpret_cast<unsigned*>(d);
    for (int i = 0; i < n; i += 4) dst[i >> 2] = src[i >> 2];
}

__global__ __launch_bounds__(UPB)
void png_bits_kernel(int H, int W, int bpp, PngWork wk, int units) {
    __shared__ __attribute__((aligned(16))) unsigned char s_d[UPB * LSTR];
    __shared__ unsigned s_code[288];
    const int tid = threadIdx.x;
    for (int i = tid; i < 288; i += UPB) s_code[i] = wk.code[i];
    __syncthreads();
    const int u = blockIdx.x * UPB + tid;
    if (u >= units) return;
    const int upr = units_per_row(W, bpp), rowlen = 1 + W * bpp;
    int row, p0, n;
    unit_span(u, upr, rowlen, row, p0, n);
    unsigned char* d = s_d + tid * LSTR;
    load_unit(d, wk.filt, u, n);
    unsigned bits = 0;
    for_each_token(d, n,
                   [&](int b) { bits += s_code[b] >> 16; },
                   [&](int L) { int idx, eb, ev; len_symbol(L, idx, eb, ev); bits += (s_code[257 + idx] >> 16) + eb + 1; });
    wk.ubits[u] = bits;
}

// exclusive scan of the unit bit counts (one block) -> bit offset of every unit behind the header, end-of-block position;
// Adler-32 of the filtered stream from the per-unit partials: appending n bytes with (s1, s2) to a state (A, B) gives
// B += n * A + s2, A += s1 (mod 65521) -- A needs an exclusive scan of s1, B is then a plain sum
__global__ __launch_bounds__(1024)
void png_offsets_kernel(PngWork wk, int units, int W, int bpp) {
    __shared__ unsigned s_part[1024];
    __shared__ unsigned long long s_a[1024];
    const int tid = threadIdx.x;
    const int per = (units + 1023) / 1024;
    const int lo = min(units, tid * per), hi = min(units, lo + per);
    const int upr = units_per_row(W, bpp), rowlen = 1 + W * bpp;
    unsigned sum = 0;
    unsigned long long a_loc = 0;
    for (int u = lo; u < hi; ++u) { sum += wk.ubits[u]; a_loc += wk.uadl[2 * (size_t)u]; }
    s_part[tid] = sum;
    s_a[tid] = a_loc;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                                 // inclusive scans over the 1024 partials
        const unsigned v = tid >= o ? s_part[tid - o] : 0u;
        const unsigned long long va = tid >= o ? s_a[tid - o] : 0ull;
        __syncthreads();
        s_part[tid] += v;
        s_a[tid] += va;
        __syncthreads();
    }
    unsigned off = wk.meta[0] + (tid ? s_part[tid - 1] : 0u);
    unsigned long long A = (1ull + (tid ? s_a[tid - 1] : 0ull)) % 65521ull, Bloc = 0;
    for (int u = lo; u < hi; ++u) {
        const unsigned bts = wk.ubits[u];
        wk.ubits[u] = off;
        off += bts;
        const unsigned n = (unsigned)min(SEG, rowlen - (u % upr) * SEG);
        Bloc += (unsigned long long)n * A + wk.uadl[2 * (size_t)u + 1];       // < 2^63 for any image vfn_png_sizes accepts
        A = (A + wk.uadl[2 * (size_t)u]) % 65521ull;
    }
    const unsigned total_units_bits = s_part[1023];
    const unsigned long long a_total = (1ull + s_a[1023]) % 65521ull;
    __syncthreads();
    s_a[tid] = Bloc % 65521ull;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) s_a[tid] += s_a[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        wk.meta[1] = wk.meta[0] + total_units_bits;                      // where the end-of-block code goes
        wk.meta[2] = (unsigned)(((s_a[0] % 65521ull) << 16) | a_total);
    }
}

// ---------------------------------------------------------------- emit
__global__ __launch_bounds__(UPB)
void png_emit_kernel(int H, int W, int bpp, PngWork wk, int units, unsigned* __restrict__ out, int* __restrict__ stats) {
    __shared__ __attribute__((aligned(16))) unsigned char s_d[UPB * LSTR];
    __shared__ unsigned s_code[288];
    const int tid = threadIdx.x;
    for (int i = tid; i < 288; i += UPB) s_code[i] = wk.code[i];
    __syncthreads();
    const int u = blockIdx.x * UPB + tid;
    if (u >= units) return;
    const int upr = units_per_row(W, bpp), rowlen = 1 + W * bpp;
    int row, p0, n;
    unit_span(u, upr, rowlen, row, p0, n);
    unsigned char* d = s_d + tid * LSTR;
    load_unit(d, wk.filt, u, n);
    const unsigned off = wk.ubits[u];
    int w = off >> 5, nb = off & 31;
    const int w0 = w;
    unsigned long long acc = 0;
    auto put = [&](unsigned v, int nbit) {
        acc |= (unsigned long long)v << nb; nb += nbit;
        while (nb >= 32) {
            if (w == w0) atomicOr(&out[w], (unsigned)acc); else out[w] = (unsigned)acc;     // first word may be shared
            ++w; acc >>= 32; nb -= 32;
        }
    };
    for_each_token(d, n,
                   [&](int b) { const unsigned c = s_code[b]; put(c & 0xffff, c >> 16); },
                   [&](int L) {
                       int idx, eb, ev;
                       len_symbol(L, idx, eb, ev);
                       const unsigned c = s_code[257 + idx];
                       put(c & 0xffff, c >> 16);
                       if (eb) put(ev, eb);
                       put(0, 1);                                        // distance 1: the single 1-bit distance code
                   });
    if (u == units - 1) {                                                // end of block, then the byte count
        const unsigned c = s_code[256];
        put(c & 0xffff, c >> 16);
        const unsigned total = wk.meta[1] + (c >> 16);
        stats[0] = (int)((total + 7) >> 3);
        stats[1] = (int)wk.meta[2];
        stats[2] = (int)total;
    }
    if (nb > 0) atomicOr(&out[w], (unsigned)acc);                        // last word is shared with the next unit
}

}  // namespace

extern "C" int vfn_png_sizes(int H, int W, int bpp, long long* work_bytes, long long* out_bytes) {
    if (H < 1 || W < 1 || (bpp != 1 && bpp != 3)) return VFN_ERR_ARG;
    const long long units = (long long)H * units_per_row(W, bpp);
    if ((long long)H * (1 + (long long)W * bpp) >= (1ll << 23)) return VFN_ERR_ARG;     // symbol counts are sorted as 23-bit keys
    if (work_bytes) *work_bytes = (288 + 288 + 8 + (units + 3) / 4 * 4 + 2 * units) * 4 + units * SEG;
    // a Huffman code of the image's own histogram needs < log2(286) + 1 = 9.2 bits per token, tokens <= bytes; + header
    if (out_bytes) *out_bytes = ((long long)H * (1 + (long long)W * bpp) * 10 / 8 + 4096 + 3) / 4 * 4;
    return VFN_OK;
}

extern "C" int vfn_png_deflate_u8(const unsigned char* raw, int H, int W, int bpp, void* work, unsigned char* out,
                                  int* stats, void* stream) {
    long long wb, ob;
    if (!raw || !work || !out || !stats || vfn_png_sizes(H, W, bpp, &wb, &ob) != VFN_OK) return VFN_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int units = H * units_per_row(W, bpp);
    PngWork wk = carve(work, units);
    if (hipMemsetAsync(wk.hist, 0, (288 + 288 + 8) * sizeof(unsigned), s) != hipSuccess) return VFN_ERR_LAUNCH;
    if (hipMemsetAsync(out, 0, (size_t)ob, s) != hipSuccess) return VFN_ERR_LAUNCH;
    const int blocks = cdiv(units, UPB);
    static const int stages = getenv("VFN_PNG_STAGES") ? atoi(getenv("VFN_PNG_STAGES")) : 5;   // (profiling aid: stop early)
    hipLaunchKernelGGL(png_scan_kernel, dim3(blocks), dim3(SCAN_NT), 0, s, raw, H, W, bpp, wk, units);
    if (stages > 1) hipLaunchKernelGGL(png_code_kernel, dim3(1), dim3(512), 0, s, wk, reinterpret_cast<unsigned*>(out));
    if (stages > 2) hipLaunchKernelGGL(png_bits_kernel, dim3(blocks), dim3(UPB), 0, s, H, W, bpp, wk, units);
    if (stages > 3) hipLaunchKernelGGL(png_offsets_kernel, dim3(1), dim3(1024), 0, s, wk, units, W, bpp);
    if (stages > 4) hipLaunchKernelGGL(png_emit_kernel, dim3(blocks), dim3(UPB), 0, s, H, W, bpp, wk, units, reinterpret_cast<unsigned*>(out), stats);
    return vfn_check_launch();
}
