// Feature-bank contractions on the f32 matrix cores.
//
// (1) Memory read -- Matcher.forward, AFB_URR.py:136-178, per object:
//        p   = softmax_over_bank( keys^T q / sqrt(128) )          [B, HW]
//        mem = values p                                            [512, HW]
//        info[:,1] += log( sum_hw [p > 1e-3] + 1 )                 (update_bank side effect, :161-174)
//     The reference materialises p (648 MB at B = 100k) and makes >= 7 elementwise passes over it,
//     with an out-of-memory -> CPU fallback (:147-157).  Here p is never stored:
//        pass 1  vfn_bank_scan (mode 0) + _finish   per query column: running max m and sum l over the bank;
//                                    f32: the raw scores are also stored (score tiles, one write and one read per frame)
//        pass 2  vfn_memread_apply   scores read back (f32, memread_apply_ss_kernel) or recomputed (reduced precision,
//                                    or no score buffer), p = exp(s-m)/l in registers, hit counts by wave ballots,
//                                    O^T += P^T V on the MFMA
//        finish  vfn_memread_finish  reduce the bank-split partial O, write it (and the query
//                                    value) into the decoder input, apply the log(count+1) bump
//
// (2) Bank match -- FeatureBank.update, FeatureBank.py:63-68:
//        corr = normalize(keys)^T normalize(new_keys); argmax over the bank per new feature
//     Same GEMM with an arg-max epilogue (corr is never stored): vfn_bank_scan (mode 1) + vfn_bank_scan_finish.
//
// Layout: bank entry-major K [cap][128], V [cap][512]; queries / new keys [HW][ld]; reduced precision: the bank's
// split-bf16 image beside it (vfn_bank_refresh_lp).
// Tiling: the bank is walked in chunks of 64 entries.  Scans: persistent workgroups (two per CU) draw (bank slice, 128-query
// tile, object) items from a queue; a wave owns 32 query columns against all 64 rows of a chunk, key chunks arrive by
// LDS-DMA, double-buffered.  Apply: one workgroup = 128 query columns x one slice of the bank, 8 waves; P^T goes through
// LDS, wave w owns value channels 64w..64w+63 of all 128 queries and reads its value rows straight from global memory
// (buffer loads, two k-groups ahead).
#include "common.h"
#include "../../include/vfn_hip.h"
#include <type_traits>
#include <cstdlib>

namespace {

constexpr int DK = 128, DV = 512;
constexpr int CH = 64;      // bank entries per chunk

// [rows][128 floats] LDS image, 16-byte chunk index XOR (row & 15): conflict-free b128 fragment reads
__device__ __forceinline__ int swz(int row, int chunk) { return row * DK + ((chunk ^ (row & 15)) << 2); }

// [rows][64 floats] image (P^T): chunk index 0..15 XOR (row & 15)
__device__ __forceinline__ int swz64(int row, int chunk) { return row * CH + ((chunk ^ (row & 15)) << 2); }

// one 64 x 128 chunk = 2048 float4 = 8 per thread: global -> registers, registers -> LDS
struct ChunkRegs { f32x4 v[8]; };
__device__ __forceinline__ void chunk_load(ChunkRegs& R, const float* src, int valid, int tid) {
    const int c = tid & 31;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = (tid >> 5) + 8 * j;
        const int rr = r < valid ? r : 0;                         // clamp: no branch around the load
        R.v[j] = *reinterpret_cast<const f32x4*>(src + (size_t)rr * DK + c * 4);
        if (r >= valid) R.v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
__device__ __forceinline__ void chunk_store(const ChunkRegs& R, float* dst, int tid) {
    const int c = tid & 31;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = (tid >> 5) + 8 * j;
        *reinterpret_cast<f32x4*>(dst + swz(r, c)) = R.v[j];
    }
}

// One 64 x 128 key chunk straight into its swizzled LDS image with LDS-DMA (no VGPRs): a wave
// instruction writes 1 KB = two image rows linearly, so the XOR swizzle goes on the per-lane SOURCE
// address.  Rows past `valid` re-read the last valid row (finite data; masked by the caller).
__device__ __forceinline__ void chunk_load_async(float* sK, const float* src, int valid, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = (wave * 8 + j) * 2 + (lane >> 5);
        const int pc = lane & 31;
        const int rr = min(r, valid - 1);
        const float* g = src + (size_t)rr * DK + ((pc ^ (r & 15)) << 2);
        __builtin_amdgcn_global_load_lds(g, sK + (wave * 8 + j) * 2 * DK, 16, 0, 0);
    }
}

// stage `rows` x 128 floats (row r from src + r*ld, zero past `valid` rows) into an LDS image
__device__ __forceinline__ void stage_rows(float* dst, const float* src, size_t ld, int rows, int valid, int tid) {
    const int c = tid & 31;
    for (int r = tid >> 5; r < rows; r += 8) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < valid) v = *reinterpret_cast<const f32x4*>(src + (size_t)r * ld + c * 4);
        *reinterpret_cast<f32x4*>(dst + swz(r, c)) = v;
    }
}

// scores for this wave's 32 chunk rows (32*wr..) x 32 query columns (32*wq..): acc[r], row = (r&3)+8*(r>>2)+4*lh
__device__ __forceinline__ void score_tile(const float* sK, const float* sQ, int wr, int wq, int li, int lh, f32x16& acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int ra = wr * 32 + li, rq = wq * 32 + li;
    f32x4 a[2], b[2];
    a[0] = *reinterpret_cast<const f32x4*>(sK + swz(ra, lh));
    b[0] = *reinterpret_cast<const f32x4*>(sQ + swz(rq, lh));
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const int cur = kk & 1;
        if (kk + 1 < 16) {
            a[cur ^ 1] = *reinterpret_cast<const f32x4*>(sK + swz(ra, 2 * (kk + 1) + lh));
            b[cur ^ 1] = *reinterpret_cast<const f32x4*>(sQ + swz(rq, 2 * (kk + 1) + lh));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][t], b[cur][t], acc, 0, 0, 0);
    }
}

// ---- reduced-precision operands (BASELINE configs C3 / C5; precision 1 = bf16, 2 = bf16x3)
// 8 consecutive k of one operand row -> one v_mfma_f32_32x32x16_bf16 fragment.  bf16x3: x = hi + lo (two bf16,
// 16 significant bits); a product is hi*hi + hi*lo + lo*hi, the 2^-16 lo*lo term is dropped.
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ bf16x8 cvt8(const f32x4& a, const f32x4& b) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[e] = (__bf16)a[e]; r[4 + e] = (__bf16)b[e]; }
    return r;
}
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8& h, bf16x8& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 x = (__bf16)a[e], y = (__bf16)b[e];
        h[e] = x; h[4 + e] = y;
        l[e] = (__bf16)(a[e] - (float)x); l[4 + e] = (__bf16)(b[e] - (float)y);
    }
}
// A key chunk that has landed in LDS as f32 ([64 rows][128], 16-byte chunks XOR-swizzled by row & 15) is converted ONCE,
// in place, into the operand image the reduced-precision MFMAs read: per row 128 bf16 "hi" in the first 256 bytes and (bf16x3)
// 128 bf16 "lo" in the second, 16-byte chunks (8 consecutive k) again swizzled by row & 15.  Before, every wave split the
// fragments it read in registers -- the same key row was converted by every wave that used it (4x in the scan and in the
// 128-query apply kernel) and the conversion VALU equalled the MFMA time of the score GEMM.  NT threads, two barriers.
__device__ __forceinline__ int swzk(int row, int half, int chunk) { return row * 512 + half * 256 + ((chunk ^ (row & 15)) << 4); }   // bytes
// rows of 256 B (query image; hi plane of a key chunk): 16-byte chunk index XOR (row & 15)
__device__ __forceinline__ int swzq(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }          // bytes

template <int NT, bool X3>
__device__ __forceinline__ void convert_chunk_inplace(float* sK, int tid) {
    constexpr int TPR = NT / CH;                   // threads per row (4 or 8)
    constexpr int F4 = 32 / TPR;                   // float4 per thread (8 or 4)
    const int r = tid / TPR, part = tid % TPR;
    f32x4 v[F4];
#pragma unroll
    for (int j = 0; j < F4; ++j) v[j] = *reinterpret_cast<const f32x4*>(sK + swz(r, part * F4 + j));
    __syncthreads();                               // every f32 value is in registers before the image is overwritten
    char* base = reinterpret_cast<char*>(sK);
#pragma unroll
    for (int j = 0; j < F4; j += 2) {
        bf16x8 h, l;
        split8(v[j], v[j + 1], h, l);
        const int chunk = (part * F4 + j) >> 1;    // 8 consecutive k = one 16-byte bf16 chunk
        *reinterpret_cast<bf16x8*>(base + swzk(r, 0, chunk)) = h;
        if constexpr (X3) *reinterpret_cast<bf16x8*>(base + swzk(r, 1, chunk)) = l;
    }
    __syncthreads();
}

template <bool X3>
__device__ __forceinline__ void mfma_lp(f32x16& acc, const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl) {
    if constexpr (X3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

// exp for the softmax inner loops: v_exp_f32 on x*log2(e) (1 ulp hardware exp2; the f32 product adds at most
// |x| * 6e-8 relative error, i.e. < 1e-6 where exp(x) still matters) -- the full-range expf is 10x the instructions
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }

__device__ __forceinline__ void chunk_range(int B, int nsplit, int split, int& c_lo, int& c_hi) {
    const int nchunks = (B + CH - 1) / CH;
    const int per = (nchunks + nsplit - 1) / nsplit;
    c_lo = split * per;
    c_hi = min(nchunks, c_lo + per);
}

// ------------------------------------------------------------------ pass 1: softmax statistics
// MODE 0: (max, sum exp) of scale*s per query.  MODE 1: arg-max of s*rowscale[b] per query.
// One workgroup = 128 query columns x one slice of the bank; wave w owns queries 32w..32w+31 against ALL 64 rows of
// every chunk (two 32x32 accumulator tiles), so a key fragment read from LDS feeds eight MFMAs, a key chunk is
// streamed once per 128 queries, and a query's statistics never leave its lane pair (no cross-wave reduction).
// Query fragments live in registers (64 VGPRs), key chunks are double-buffered in LDS by LDS-DMA:
// 64 KB of LDS -> two workgroups per CU, one barrier per chunk, the next chunk lands behind the MFMAs.
constexpr int QTS = 128;    // query columns per scan workgroup

#ifdef VFN_CENSUS
__device__ unsigned long long vfn_census_buf[4096 * 4];
#endif

template <int MODE, int PREC = 0>
__global__ __launch_bounds__(256, 2)
void bank_scan_kernel(const vfn_bankscan_desc p) {
#ifdef VFN_CENSUS
    const unsigned long long census_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sKb = reinterpret_cast<float*>(smem);     // [2][64][128]
    __shared__ int s_item;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    // Persistent workgroups (two per CU) pull work items from a queue: item = (bank slice, query tile, object), slice
    // slowest, so the workgroups that run at the same time walk the same part of the bank (L2).  A census of the former
    // one-workgroup-per-(tile, slice) grid showed why it ran at 0.63 MFMA utilisation: two workgroups sharing a CU keep
    // the matrix pipe 100 % busy, one alone only 71 %, and with 754 (or 494) equal workgroups on 512 slots the last
    // third of the kernel ran half-empty.  With ~2000 small items every slot stays paired until the queue is dry.
    // An item's result depends only on the item, so the output is bit-identical whichever workgroup computes it.
    const int qtiles = (p.HW + QTS - 1) / QTS;
    const int total = p.nsplit * qtiles * p.obj_n;
  for (;;) {
    if (tid == 0) s_item = atomicAdd(p.work_counter, 1);
    __syncthreads();
    const int item = s_item;
    __syncthreads();                                   // (everyone has the item before thread 0 draws the next one)
    if (item >= total) break;
    const int obj = item % p.obj_n;
    const int qt = (item / p.obj_n) % qtiles;
    const int split = item / (p.obj_n * qtiles);
    const int q0 = qt * QTS + wave * 32;             // first query of this wave
    const int B = p.bank_len[obj];
    // reduced precision with a kept split-bf16 image of the keys (vfn_bank_refresh_lp): same bytes per row as f32, the
    // chunk lands in LDS as the operand image itself (hi | lo halves, chunks swizzled like swzk)
    const bool lp_image = PREC != 0 && p.bank_k_lp != nullptr;
    const float* K = (lp_image ? reinterpret_cast<const float*>(p.bank_k_lp) : p.bank_k) + (size_t)obj * p.stride_k;
    const float* Q = p.q + (size_t)(p.q_per_obj ? obj : 0) * p.stride_q;

    int c_lo, c_hi;
    chunk_range(B, p.nsplit, split, c_lo, c_hi);
    if (c_lo < c_hi) chunk_load_async(sKb, K + (size_t)c_lo * CH * DK, min(CH, B - c_lo * CH), wave, lane);

    // B operand: this lane's query column, k = 8kk + 4lh + t (f32) / k = 16g + 8lh + j (bf16 fragments)
    f32x4 qf[PREC == 0 ? 16 : 1];
    bf16x8 qh[PREC == 0 ? 1 : 8], ql[PREC == 2 ? 8 : 1];
    {
        const int q = min(q0 + li, p.HW - 1);                    // columns past HW are never written out
        if constexpr (PREC == 0) {
            const float* qrow = Q + (size_t)q * p.ldq + 4 * lh;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) qf[kk] = *reinterpret_cast<const f32x4*>(qrow + 8 * kk);
        } else {
            const float* qrow = Q + (size_t)q * p.ldq + 8 * lh;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(qrow + 16 * g);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(qrow + 16 * g + 4);
                if constexpr (PREC == 2) split8(x0, x1, qh[g], ql[g]);
                else qh[g] = cvt8(x0, x1);
            }
        }
    }

    float run_m = -INFINITY, run_l = 0.f;
    int run_i = 0x7fffffff;
    __syncthreads();                                   // chunk c_lo landed (the barrier drains the LDS-DMA)

    for (int c = c_lo; c < c_hi; ++c) {
        const int b0 = c * CH;
        const float* sK = sKb + ((c - c_lo) & 1) * CH * DK;
#ifndef VFN_ABLATE_KLOAD
        if (c + 1 < c_hi)
            chunk_load_async(sKb + ((c + 1 - c_lo) & 1) * CH * DK, K + (size_t)(b0 + CH) * DK, min(CH, B - b0 - CH), wave, lane);
#endif
        // MODE 1: the chunk's 32 row scales per lane (1 / |key|), requested BEFORE the score MFMAs (round 5: they were loaded behind
        // them, inside a range test per group, and used at once -- one exposed L2 round trip at the end of every chunk).  Rows past the
        // bank's end lie inside the slab (finite scratch) and are ignored by the comparison below.
        f32x4 scv[MODE == 1 ? 2 : 1][MODE == 1 ? 4 : 1];
        if constexpr (MODE == 1) {
            const float* rs = p.rowscale + (size_t)obj * p.stride_rs;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) scv[i][g] = *reinterpret_cast<const f32x4*>(rs + b0 + 32 * i + 4 * lh + 8 * g);
        }
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        if constexpr (PREC == 0) {
            f32x4 a[2][2];
            a[0][0] = *reinterpret_cast<const f32x4*>(sK + swz(li, lh));
            a[0][1] = *reinterpret_cast<const f32x4*>(sK + swz(32 + li, lh));
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int cur = kk & 1;
                if (kk + 1 < 16) {
                    a[cur ^ 1][0] = *reinterpret_cast<const f32x4*>(sK + swz(li, 2 * (kk + 1) + lh));
                    a[cur ^ 1][1] = *reinterpret_cast<const f32x4*>(sK + swz(32 + li, 2 * (kk + 1) + lh));
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][0][t], qf[kk][t], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][1][t], qf[kk][t], acc[1], 0, 0, 0);
                }
            }
        } else if (PREC == 2 || lp_image) {
            // the chunk as hi | lo operand image: kept beside the bank, or split here ONCE (the four waves read the same
            // 64 rows; splitting in registers per wave cost as many VALU cycles as the score MFMAs).
            // k = 16g + 8h .. +7 is chunk 2g + h of a half.
            if (!lp_image) convert_chunk_inplace<256, true>(const_cast<float*>(sK), tid);
            const char* kb = reinterpret_cast<const char*>(sK);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(kb + swzk(32 * i + li, 0, 2 * g + lh));
                    bf16x8 al = ah;
                    if constexpr (PREC == 2) al = *reinterpret_cast<const bf16x8*>(kb + swzk(32 * i + li, 1, 2 * g + lh));
                    mfma_lp<PREC == 2>(acc[i], ah, al, qh[g], ql[PREC == 2 ? g : 0]);
                }
            }
        } else {
            // bf16: two 16-byte reads (f32 chunks 4g+2h, 4g+2h+1 = k 16g+8h .. +7) per row tile, rounded in registers
            // (one cvt per pair; an in-LDS conversion pass measured slower here: 2 barriers for little VALU)
#pragma unroll
            for (int g = 0; g < 8; ++g) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(sK + swz(32 * i + li, 4 * g + 2 * lh));
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(sK + swz(32 * i + li, 4 * g + 2 * lh + 1));
                    const bf16x8 ah = cvt8(x0, x1);
                    mfma_lp<false>(acc[i], ah, ah, qh[g], ql[0]);
                }
            }
        }
        if (MODE == 0 && p.scores) {
            // the raw scores of this (chunk, query tile), in accumulator order: [key half i][row group g][lane half][query][4
            // rows] -- each store instruction of a wave writes two 512-byte runs; the apply kernel reads them back the
            // same way instead of repeating the GEMM
            float* tile = p.scores + (size_t)obj * p.stride_scores + ((size_t)c * qtiles + qt) * (CH * QTS);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(tile + ((((i * 4 + g) * 2 + lh) * QTS) + wave * 32 + li) * 4));
                }
        }
#ifdef VFN_ABLATE_SOFTMAX
        if (true) { run_m = fmaxf(run_m, acc[0][0] + acc[1][5]); } else
#endif
        if (MODE == 0) {
            if (b0 + CH <= B) {                          // whole chunk inside the bank (all but the last): no row checks
                float mx = acc[0][0];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[i][r]);
                const float mn = fmaxf(run_m, mx * p.scale);     // scale > 0: max commutes with it
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sum += fast_exp(acc[i][r] * p.scale - mn);
                run_l = run_l * expf(run_m - mn) + sum;       // exp(-inf)=0 on the first chunk
                run_m = mn;
            } else {
                float mx = -INFINITY;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = b0 + 32 * i + 4 * lh + (r & 3) + 8 * (r >> 2);
                        const float s_ = acc[i][r] * p.scale;
                        acc[i][r] = s_;
                        if (row < B) mx = fmaxf(mx, s_);
                    }
                const float mn = fmaxf(run_m, mx);
                if (mn > -INFINITY) {
                    float sum = 0.f;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = b0 + 32 * i + 4 * lh + (r & 3) + 8 * (r >> 2);
                            if (row < B) sum += fast_exp(acc[i][r] - mn);
                        }
                    run_l = run_l * expf(run_m - mn) + sum;
                    run_m = mn;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {            // registers 4g..4g+3 = rows +8g .. +3: one 16-byte load
                    const int row0 = b0 + 32 * i + 4 * lh + 8 * g;
                    const f32x4 sc = scv[MODE == 1 ? i : 0][MODE == 1 ? g : 0];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = row0 + j;
                        if (row < B) {
                            const float s_ = acc[i][4 * g + j] * sc[j];
                            if (s_ > run_m) { run_m = s_; run_i = row; }   // (ties: smaller row wins, below)
                        }
                    }
                }
        }
        __syncthreads();                               // next chunk landed; this buffer is free again
    }

    // combine the two lane halves of a query; lanes 0..31 write the slice result
    {
        const float om = __shfl_xor(run_m, 32, 64);
        if (MODE == 0) {
            const float ol = __shfl_xor(run_l, 32, 64);
            const float mn = fmaxf(run_m, om);
            float l = 0.f;
            if (mn > -INFINITY) l = run_l * expf(run_m - mn) + ol * expf(om - mn);
            run_m = mn; run_l = l;
        } else {
            const int oi = __shfl_xor(run_i, 32, 64);
            if (om > run_m || (om == run_m && oi < run_i)) { run_m = om; run_i = oi; }
        }
    }
#ifdef VFN_CENSUS
    if (threadIdx.x == 0) {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x;
        if (lin < 4096) {
            unsigned hw_id, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            vfn_census_buf[lin * 4] = census_t0; vfn_census_buf[lin * 4 + 1] = __builtin_amdgcn_s_memrealtime();
            vfn_census_buf[lin * 4 + 2] = hw_id; vfn_census_buf[lin * 4 + 3] = xcc;
        }
    }
#endif
    const int q = q0 + li;
    if (lh == 0 && q < p.HW) {
        float* dst = p.part + (((size_t)obj * p.nsplit + split) * p.HW + q) * 2;
        dst[0] = run_m;
        dst[1] = (MODE == 0) ? run_l : __int_as_float(run_i);
    }
  }
}

// ------------------------------------------------------------------ pass 1, plain bf16 on the kept key image: register-staged keys (round 6)
// bank_scan_kernel<MODE, 1> brings every key chunk in by LDS-DMA ONE chunk ahead and ends every chunk on a barrier that drains the
// DMA: a chunk is 16 MFMAs + ~230 vector slots per wave (0.7 us), the DMA's round trip to L2 / HBM is longer, so the loop runs at
// the memory latency (rocprofv3, round 5: mfma_util 0.17-0.18 at C5 sizes, where the two scans are a quarter of the frame).
// Here the keys go through registers, requested TWO iterations before their chunk is multiplied: issued at the top of iteration
// c - 1, stored into the other LDS buffer at the end of iteration c, read in iteration c + 1 -- two register sets of 64 bytes per
// thread, no DMA (so hipcc's waits stay counted, also around the row-scale loads of MODE 1), and only the hi plane of the image is
// staged (256-byte rows: 16 KB per chunk instead of 32).  Same fragments, same products, same statistics in the same order as the
// image path of bank_scan_kernel<MODE, 1>: bit-identical.  LDS 32 KB; two workgroups per CU as before.
// (MODE 1: the chunk's 64 row scales travel with its keys -- one 16-byte load on 16 threads, through registers into 256 bytes of LDS
// beside the key buffer, read back as broadcast ds_read_b128.  As register loads of their own, issued behind the key requests, the
// in-order load counter made their wait a wait for the keys two chunks ahead: 0.76-1.04 x the DMA kernel instead of 1.4 x.)
constexpr size_t SCAN_PIPE_LDS = 2 * (size_t)CH * DK * 2 + 2 * CH * sizeof(float);
constexpr unsigned SCAN_PIPE_SC0 = 2 * CH * DK * 2;         // byte offset of the two row-scale buffers

template <int MODE>
__global__ __launch_bounds__(256, 2)
void bank_scan_pipe_kernel(const vfn_bankscan_desc p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 2 x [64 rows][256 B] bf16 hi plane, swzq
    __shared__ int s_item;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int qtiles = (p.HW + QTS - 1) / QTS;
    const int total = p.nsplit * qtiles * p.obj_n;
    const int krow = tid >> 2, kc4 = (tid & 3) * 4;                   // key staging: row, first of four 16-byte chunks
    const unsigned klane_off = (unsigned)krow * 512u + (unsigned)kc4 * 16u;
  for (;;) {
    if (tid == 0) s_item = atomicAdd(p.work_counter, 1);
    __syncthreads();
    const int item = s_item;
    __syncthreads();                                   // (everyone has the item before thread 0 draws the next one)
    if (item >= total) break;
    const int obj = item % p.obj_n;
    const int qt = (item / p.obj_n) % qtiles;
    const int split = item / (p.obj_n * qtiles);
    const int q0 = qt * QTS + wave * 32;             // first query of this wave
    const int B = p.bank_len[obj];
    const char* Kimg = reinterpret_cast<const char*>(p.bank_k_lp) + (size_t)obj * p.stride_k * 4;    // image rows of 512 B: [128 hi | 128 lo]
    const float* Q = p.q + (size_t)(p.q_per_obj ? obj : 0) * p.stride_q;

    int c_lo, c_hi;
    chunk_range(B, p.nsplit, split, c_lo, c_hi);

    u32x4 kst[2][4];                                    // two chunks of keys in flight: 64 bytes per thread each
    f32x4 sst[2];                                       // (MODE 1) ... and their row scales: rows 4 tid .. + 3 on threads 0..15
    const float* rs = MODE == 1 ? p.rowscale + (size_t)obj * p.stride_rs : nullptr;
    auto load_k = [&](int c, auto SET_) {
        constexpr int set = decltype(SET_)::value;
        const int b0 = c * CH;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Kimg + (size_t)b0 * 512), 0, min(CH, B - b0) * 512, 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; ++j) kst[set][j] = __builtin_amdgcn_raw_buffer_load_b128(r, klane_off, 16 * j, 0);
        // (rows past the bank's end lie inside the slab -- finite scratch -- and are ignored by the comparison below)
        if constexpr (MODE == 1) { if (tid < 16) sst[set] = *reinterpret_cast<const f32x4*>(rs + b0 + 4 * tid); }
    };
    auto store_k = [&](int buf, auto SET_) {
        constexpr int set = decltype(SET_)::value;
        char* d = smem + buf * (CH * DK * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(d + swzq(krow, kc4 + j)) = kst[set][j];
        if constexpr (MODE == 1) { if (tid < 16) *reinterpret_cast<f32x4*>(smem + SCAN_PIPE_SC0 + buf * (CH * 4) + tid * 16) = sst[set]; }
    };
    using Z0 = std::integral_constant<int, 0>; using Z1 = std::integral_constant<int, 1>;
    if (c_lo < c_hi) load_k(c_lo, Z0{});
    if (c_lo + 1 < c_hi) load_k(c_lo + 1, Z1{});

    bf16x8 qh[8];                                      // B operand: this lane's query column, k = 16g + 8lh + j
    {
        const int q = min(q0 + li, p.HW - 1);                    // columns past HW are never written out
        const float* qrow = Q + (size_t)q * p.ldq + 8 * lh;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(qrow + 16 * g);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(qrow + 16 * g + 4);
            qh[g] = cvt8(x0, x1);
        }
    }

    float run_m = -INFINITY, run_l = 0.f;
    int run_i = 0x7fffffff;
    if (c_lo < c_hi) store_k(0, Z0{});
    __syncthreads();

    auto body = [&](auto PAR_, int c) {                 // PAR = (c - c_lo) & 1: LDS buffer of chunk c, register set of chunk c + 2
        constexpr int PAR = decltype(PAR_)::value;
        const int b0 = c * CH;
        const char* kb = smem + PAR * (CH * DK * 2);
        if (c + 2 < c_hi) load_k(c + 2, PAR_);          // (set PAR held chunk c: stored to LDS an iteration ago)
        f32x4 scv[MODE == 1 ? 2 : 1][MODE == 1 ? 4 : 1];
        if constexpr (MODE == 1) {
            const float* sS = reinterpret_cast<const float*>(smem + SCAN_PIPE_SC0 + PAR * (CH * 4));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) scv[i][g] = *reinterpret_cast<const f32x4*>(sS + 32 * i + 4 * lh + 8 * g);
        }
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(kb + swzq(32 * i + li, 2 * g + lh));
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qh[g], acc[i], 0, 0, 0);
            }
        }
        if (MODE == 0) {
            if (b0 + CH <= B) {                          // whole chunk inside the bank (all but the last): no row checks
                float mx = acc[0][0];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[i][r]);
                const float mn = fmaxf(run_m, mx * p.scale);     // scale > 0: max commutes with it
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sum += fast_exp(acc[i][r] * p.scale - mn);
                run_l = run_l * expf(run_m - mn) + sum;       // exp(-inf)=0 on the first chunk
                run_m = mn;
            } else {
                float mx = -INFINITY;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = b0 + 32 * i + 4 * lh + (r & 3) + 8 * (r >> 2);
                        const float s_ = acc[i][r] * p.scale;
                        acc[i][r] = s_;
                        if (row < B) mx = fmaxf(mx, s_);
                    }
                const float mn = fmaxf(run_m, mx);
                if (mn > -INFINITY) {
                    float sum = 0.f;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = b0 + 32 * i + 4 * lh + (r & 3) + 8 * (r >> 2);
                            if (row < B) sum += fast_exp(acc[i][r] - mn);
                        }
                    run_l = run_l * expf(run_m - mn) + sum;
                    run_m = mn;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {            // registers 4g..4g+3 = rows +8g .. +3: one 16-byte load
                    const int row0 = b0 + 32 * i + 4 * lh + 8 * g;
                    const f32x4 sc = scv[MODE == 1 ? i : 0][MODE == 1 ? g : 0];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = row0 + j;
                        if (row < B) {
                            const float s_ = acc[i][4 * g + j] * sc[j];
                            if (s_ > run_m) { run_m = s_; run_i = row; }   // (ties: smaller row wins, below)
                        }
                    }
                }
        }
        if (c + 1 < c_hi) store_k(PAR ^ 1, std::integral_constant<int, PAR ^ 1>{});   // chunk c + 1 (requested an iteration ago) into the buffer chunk c - 1 has left
        __syncthreads();
    };
    for (int c = c_lo; c < c_hi; c += 2) {
        body(std::integral_constant<int, 0>{}, c);
        if (c + 1 < c_hi) body(std::integral_constant<int, 1>{}, c + 1);
    }

    // combine the two lane halves of a query; lanes 0..31 write the slice result
    {
        const float om = __shfl_xor(run_m, 32, 64);
        if (MODE == 0) {
            const float ol = __shfl_xor(run_l, 32, 64);
            const float mn = fmaxf(run_m, om);
            float l = 0.f;
            if (mn > -INFINITY) l = run_l * expf(run_m - mn) + ol * expf(om - mn);
            run_m = mn; run_l = l;
        } else {
            const int oi = __shfl_xor(run_i, 32, 64);
            if (om > run_m || (om == run_m && oi < run_i)) { run_m = om; run_i = oi; }
        }
    }
    const int q = q0 + li;
    if (lh == 0 && q < p.HW) {
        float* dst = p.part + (((size_t)obj * p.nsplit + split) * p.HW + q) * 2;
        dst[0] = run_m;
        dst[1] = (MODE == 0) ? run_l : __int_as_float(run_i);
    }
  }
}

// combine bank-split partials.  MODE 0 -> ml[obj][q] = (m, l); MODE 1 -> idx[obj][q], corr[obj][q]
// Eight lanes per query column: lane j folds the slices j, j + 8, ... (a fixed order), then the eight partial results are folded
// in lane order -- 8 + 7 steps instead of a 59-long chain on one thread (the kernel sits between the scan and the apply kernel of
// every frame: 18-22 us -> a few).  Deterministic: the order of the folds does not depend on the schedule.
template <int MODE>
__global__ void bank_scan_finish_kernel(const float* __restrict__ part, int nsplit, int HW, int obj_n,
                                        float* __restrict__ ml, int* __restrict__ idx, float* __restrict__ corr,
                                        const float* __restrict__ colscale) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = min(t >> 3, obj_n * HW - 1), j = t & 7;          // (surplus groups repeat the last column: whole waves shuffle)
    const bool live = (t >> 3) < obj_n * HW;
    const int obj = i / HW, q = i - obj * HW;
    float m = -INFINITY, x = (MODE == 0) ? 0.f : __int_as_float(0x7fffffff);
    auto fold = [&](float om, float ox) {
        if (MODE == 0) {
            const float mn = fmaxf(m, om);
            if (mn > -INFINITY) x = x * expf(m - mn) + ox * expf(om - mn);
            m = mn;
        } else {
            const int i0 = __float_as_int(x), i1 = __float_as_int(ox);
            if (om > m || (om == m && i1 < i0)) { m = om; x = ox; }
        }
    };
    for (int s = j; s < nsplit; s += 8) {
        const float* src = part + (((size_t)obj * nsplit + s) * HW + q) * 2;
        fold(src[0], src[1]);
    }
    const int base = (threadIdx.x & 63) & ~7;
#pragma unroll
    for (int o = 1; o < 8; ++o) {
        const float om = __shfl(m, base + o, 64), ox = __shfl(x, base + o, 64);
        if (j == 0) fold(om, ox);
    }
    if (j != 0 || !live) return;
    if (MODE == 0) { ml[(size_t)i * 2] = m; ml[(size_t)i * 2 + 1] = x; }
    else { idx[i] = __float_as_int(x); corr[i] = m * colscale[i]; }
}

// p = exp(scale*s - m) * (1/l) for the 16 scores of a lane (branch-free: an out-of-range query carries m = +big and
// 1/l = 0, rows past the bank end are masked only in the last chunk), and the hit counts of the wave's 32 bank rows:
// 16 ballots give 32 row masks (lanes 0-31: row (r&3)+8*(r>>2), lanes 32-63: that + 4); their popcounts are dropped
// into lane = row of `cnt` with v_writelane -- no divergent branch anywhere.
__device__ __forceinline__ int softmax_hits(f32x16& acc, float scale, float qm, float qinv, float thres, int nvalid, int rloc) {
    int cnt = 0;
#ifdef VFN_ABLATE_SOFTMAX
    for (int r = 0; r < 16; ++r) acc[r] *= qinv;
    return cnt;
#endif
    const float sm = -qm;
    if (nvalid >= CH) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = fast_exp(fmaf(acc[r], scale, sm)) * qinv;
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = rloc + (r & 3) + 8 * (r >> 2);
            const float pv = fast_exp(fmaf(acc[r], scale, sm)) * qinv;
            acc[r] = rr < nvalid ? pv : 0.f;
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned long long hit = __ballot(acc[r] > thres);
        const int rlo = (r & 3) + 8 * (r >> 2);
        const int c_lo = __popcll(hit & 0xffffffffull), c_hi = __popcll(hit >> 32);      // wave-uniform
        asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(cnt) : "s"(c_lo), "n"(rlo));
        asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(cnt) : "s"(c_hi), "n"(rlo + 4));
    }
    return cnt;                                       // lane i (0..31) holds the hits of chunk-local row 32*wr + i
}

// Workgroup -> (bank slice, query tile, object) of the apply kernels.  Workgroups are dealt round-robin over the 8 XCDs, each with
// a private L2: the query tiles of one (object, slice) pair stream the SAME value rows, so they are given to ONE XCD as a
// contiguous run of workgroup slots (they start together and advance in step: the first reader of a chunk brings it into that
// L2, the other tiles hit it).  With query tiles scattered over the XCDs (the linear order of rounds 1-2) every chunk was
// fetched from beyond L2 by up to 8 XCDs -- at C5 sizes (GBs of values per object) that is the kernel's bandwidth.  Bijective
// for any grid; affects speed only.  VFN_APPLY_XCD=0 (read once by the host) restores the linear order.
__device__ int vfn_apply_linear_order = 0;
__device__ __forceinline__ void apply_item(const vfn_memread_desc& p, int& split, int& qt, int& obj) {
    const int qtiles = (int)gridDim.x / p.nsplit;
    if (vfn_apply_linear_order) { split = blockIdx.x % p.nsplit; qt = blockIdx.x / p.nsplit; obj = blockIdx.y; return; }
    const int total = (int)(gridDim.x * gridDim.y);
    const int id = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    const int q = total >> 3, r = total & 7;
    const int xcd = id & 7, slot = id >> 3;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    const int pair = L / qtiles;
    qt = L - pair * qtiles;
    obj = pair / p.nsplit;
    split = pair - obj * p.nsplit;
}

// ------------------------------------------------------------------ pass 2: P^T V and hit counts
// (The 64-query apply kernels of rounds 1-2 are gone: the 128-query kernels below measured faster at every bank size in
// every precision mode and were the only ones the default path had selected since.)
// LDS swizzles of the reduced-precision kernels (bytes): query image rows of 256 B, P^T rows of 128 B.  k is in natural
// order everywhere: step g of a 32x32x16 MFMA takes k = 16g + 8*(lane>>5) + j.
__device__ __forceinline__ int swzp(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }    // bytes

// ------------------------------------------------------------------ pass 2, bf16 / bf16x3, wide query tile
// For the bandwidth-bound regime (reduced-precision MFMAs are 5-16x faster than the f32 ones, so streaming the bank
// becomes the cost; at the C5 sizes every query tile re-reads a multi-GB bank): 128 query columns per workgroup,
// 8 waves.  A key chunk and a value row are fetched once per 128 queries (half the traffic of the 64-query kernel);
// wave w owns value channels 64w..64w+63 for all 128 queries, so no value row is loaded or converted twice.
// LDS: query image bf16 hi/lo [128][128] (32 + 32 KB), key chunk f32 [64][128] (32 KB), P^T bf16 hi/lo [128 q][64 b]
// (16 + 16 KB): 128 KB for bf16x3, one workgroup per CU (8 waves = two per SIMD, as the other kernels).
constexpr int QTW = 128;

__device__ __forceinline__ void chunk_load_async8(float* sK, const float* src, int valid, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {                    // 8 waves x 4 instructions x 2 rows = 64 rows
        const int r = (wave * 4 + j) * 2 + (lane >> 5);
        const int pc = lane & 31;
        const int rr = min(r, valid - 1);
        const float* g = src + (size_t)rr * DK + ((pc ^ (r & 15)) << 2);
        __builtin_amdgcn_global_load_lds(g, sK + (wave * 4 + j) * 2 * DK, 16, 0, 0);
    }
}

template <bool X3>
__global__ __launch_bounds__(512, 1)
void memread_apply_lpw_kernel(const vfn_memread_desc p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sQh = smem;                                         // [128][128] bf16
    char* sQl = sQh + QTW * DK * 2;                           // (bf16x3 only)
    float* sK = reinterpret_cast<float*>(sQh + (X3 ? 2 : 1) * QTW * DK * 2);   // [64][128] f32
    char* sPh = reinterpret_cast<char*>(sK + CH * DK);        // [128 q][64 b] bf16
    char* sPl = sPh + QTW * CH * 2;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wr = wave >> 2, wq = wave & 3;                  // score tile: key rows 32wr.., query columns 32wq..
    int split, qt, obj;
    apply_item(p, split, qt, obj);
    const int q0 = qt * QTW;
    const int B = p.bank_len[obj];
    const float* K = p.bank_k + (size_t)obj * p.stride_k;
    const float* V = p.bank_v + (size_t)obj * p.stride_v;

    {   // query image
        const int c = tid & 31;
        for (int r = tid >> 5; r < QTW; r += 16) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (q0 + r < p.HW) v = *reinterpret_cast<const f32x4*>(p.q + (size_t)(q0 + r) * p.ldq + c * 4);
            bf16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) { h[e] = (__bf16)v[e]; l[e] = (__bf16)(v[e] - (float)h[e]); }
            const int off = swzq(r, c >> 1) + (c & 1) * 8;
            *reinterpret_cast<bf16x4*>(sQh + off) = h;
            if constexpr (X3) *reinterpret_cast<bf16x4*>(sQl + off) = l;
        }
    }

    int c_lo, c_hi;
    chunk_range(B, p.nsplit, split, c_lo, c_hi);

    f32x16 o[4][2];                                           // O^T tiles: [query tile][channel tile]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[a][b][r] = 0.f;

    // value channels of this wave: 64*wave .. +63; lane li owns channels 2*li, 2*li+1 (one per 32-wide tile tc)
    const float* vcol = V + wave * 64 + li * 2;               // + row*512
    const unsigned vlane_off = (unsigned)((8 * lh) * DV + wave * 64 + li * 2) * 4u;   // bytes, per lane

    if (c_lo < c_hi) chunk_load_async8(sK, K + (size_t)c_lo * CH * DK, min(CH, B - c_lo * CH), wave, lane);
    __syncthreads();

    const int qcol = wq * 32 + li;
    const bool qok = (q0 + qcol) < p.HW;
    float qm = 1e30f, qinv = 0.f;
    if (qok) {
        qm = p.ml[((size_t)obj * p.HW + q0 + qcol) * 2];
        qinv = 1.f / p.ml[((size_t)obj * p.HW + q0 + qcol) * 2 + 1];
    }

    for (int c = c_lo; c < c_hi; ++c) {
        const int b0 = c * CH;
        const bool more = c + 1 < c_hi;

        // value rows of step 0 (16 bank rows; this lane half: rows 8*lh .. +7) -- they land behind the score GEMM
        f32x2 raw[8];
        // buffer descriptor over this chunk's value rows, ending at the bank's last row (rows past it read 0; P is exactly 0 there):
        // one load form for every chunk, no per-load address arithmetic beyond one vector add (round 5: the branch between this
        // form and a clamped plain-load form for the bank's last chunk cost hipcc's wait insertion the count of the loads in flight)
        const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(V + (size_t)b0 * DV), 0, min(CH, B - b0) * DV * 4, 0x00020000);
        auto load_raw = [&](int st) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                raw[j] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(vrsrc, vlane_off + (unsigned)((16 * st + j) * DV * 4), 0, 0));
        };
        load_raw(0);

        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {
            const int ra = wr * 32 + li, rq = wq * 32 + li;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(sK + swz(ra, 4 * g + 2 * lh));
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(sK + swz(ra, 4 * g + 2 * lh + 1));
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(sQh + swzq(rq, 2 * g + lh));
                bf16x8 bl = bh;
                if constexpr (X3) bl = *reinterpret_cast<const bf16x8*>(sQl + swzq(rq, 2 * g + lh));
                bf16x8 ah, al;
                if constexpr (X3) split8(x0, x1, ah, al);
                else { ah = cvt8(x0, x1); al = ah; }
                mfma_lp<X3>(acc, ah, al, bh, bl);
            }
        }

        const int rloc = wr * 32 + 4 * lh;
        const int mycnt = softmax_hits(acc, p.scale, qm, qinv, p.thres, B - b0, rloc);
#pragma unroll
        for (int g = 0; g < 4; ++g) {                // registers 4g..4g+3 = 4 consecutive bank rows of query qcol
            const int brow = rloc + 8 * g;
            bf16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) { h[e] = (__bf16)acc[4 * g + e]; l[e] = (__bf16)(acc[4 * g + e] - (float)h[e]); }
            const int off = swzp(qcol, brow >> 3) + ((brow >> 2) & 1) * 8;
            *reinterpret_cast<bf16x4*>(sPh + off) = h;
            if constexpr (X3) *reinterpret_cast<bf16x4*>(sPl + off) = l;
        }
        // one wave per 32 x 32 score tile: the four query groups add their hits of the same bank rows
        if (p.cnt && lh == 0 && mycnt > 0) {
            const int row = b0 + wr * 32 + li;
            if (row < B) atomicAdd(p.cnt + (size_t)obj * p.stride_cnt + row, mycnt);
        }
        __syncthreads();                             // P^T visible; every wave is done reading sK
        if (more) chunk_load_async8(sK, K + (size_t)(b0 + CH) * DK, min(CH, B - b0 - CH), wave, lane);

        // O^T[q][ch] += sum_b P^T[q][b] V[b][ch]: A = P^T (rows q, 4 tiles), B = value rows (cols = channel 2li + tc)
#pragma unroll
        for (int st = 0; st < CH / 16; ++st) {
            bf16x8 vh[2], vl[2];
#pragma unroll
            for (int tc = 0; tc < 2; ++tc)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = raw[j][tc];
                    const __bf16 hx = (__bf16)x;
                    vh[tc][j] = hx;
                    if constexpr (X3) vl[tc][j] = (__bf16)(x - (float)hx); else vl[tc][j] = hx;
                }
            if (st + 1 < CH / 16) load_raw(st + 1);
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                const bf16x8 ph = *reinterpret_cast<const bf16x8*>(sPh + swzp(tq * 32 + li, 2 * st + lh));
                bf16x8 pl = ph;
                if constexpr (X3) pl = *reinterpret_cast<const bf16x8*>(sPl + swzp(tq * 32 + li, 2 * st + lh));
#pragma unroll
                for (int tc = 0; tc < 2; ++tc) mfma_lp<X3>(o[tq][tc], ph, pl, vh[tc], vl[tc]);
            }
        }
        __syncthreads();
    }

    float* dst = p.o_part + ((size_t)obj * p.nsplit + split) * p.HW * DV;
#pragma unroll
    for (int tq = 0; tq < 4; ++tq)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = q0 + tq * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (q < p.HW) {
                const f32x2 v = {o[tq][0][r], o[tq][1][r]};
                *reinterpret_cast<f32x2*>(dst + (size_t)q * DV + wave * 64 + li * 2) = v;
            }
        }
}


// ------------------------------------------------------------------ pass 2, bf16 / bf16x3, wide tile, kept split image
// As memread_apply_lpw_kernel, for a bank that carries its split-bf16 image (vfn_bank_refresh_lp): no operand is
// converted in the loop except P.  The key chunk lands in LDS as the hi | lo operand image; the value image is stored
// operand-ready (bank.hip: blocks of 8 rows, [hi | lo plane][channel][8 rows]): a lane's B operand of a 16-row step is one
// 16-byte buffer load per channel tile and plane, 32 lanes = 512 contiguous bytes -- no transposition in registers.
// P^T V mapping: wave = (query half qh, channel quarter cq): 2 query tiles x 4 channel tiles (channel = 128cq + 32tc + li),
// so P^T fragments are read from LDS half as often as with 4 query tiles x 2 channel tiles.
constexpr unsigned VBLK = 8 * DV * 4;       // bytes of one 8-row block of the value image

template <bool X3>
__global__ __launch_bounds__(512, 1)
void memread_apply_shw_kernel(const vfn_memread_desc p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sQh = smem;                                         // [128][128] bf16
    char* sQl = sQh + QTW * DK * 2;                           // (bf16x3 only)
    float* sK = reinterpret_cast<float*>(sQh + (X3 ? 2 : 1) * QTW * DK * 2);   // [64][hi 256 B | lo 256 B]
    char* sPh = reinterpret_cast<char*>(sK + CH * DK);        // [128 q][64 b] bf16
    char* sPl = sPh + QTW * CH * 2;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wr = wave >> 2, wq = wave & 3;                  // score tile: key rows 32wr.., query columns 32wq..
    const int qhalf = wave >> 2, cq = wave & 3;               // P^T V: queries 64qhalf.., channels 128cq..
    int split, qt, obj;
    apply_item(p, split, qt, obj);
    const int q0 = qt * QTW;
    const int B = p.bank_len[obj];
    const float* K = reinterpret_cast<const float*>(p.bank_k_lp) + (size_t)obj * p.stride_k;     // image rows are 512 B too
    const char* V = reinterpret_cast<const char*>(p.bank_v_lp) + (size_t)obj * p.stride_v * 4;   // blocks of 8 rows, 16 KB

    {   // query image
        const int c = tid & 31;
        for (int r = tid >> 5; r < QTW; r += 16) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (q0 + r < p.HW) v = *reinterpret_cast<const f32x4*>(p.q + (size_t)(q0 + r) * p.ldq + c * 4);
            bf16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) { h[e] = (__bf16)v[e]; l[e] = (__bf16)(v[e] - (float)h[e]); }
            const int off = swzq(r, c >> 1) + (c & 1) * 8;
            *reinterpret_cast<bf16x4*>(sQh + off) = h;
            if constexpr (X3) *reinterpret_cast<bf16x4*>(sQl + off) = l;
        }
    }

    int c_lo, c_hi;
    chunk_range(B, p.nsplit, split, c_lo, c_hi);

    f32x16 o[2][4];                                           // O^T tiles: [query tile][channel tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[a][b][r] = 0.f;

    const unsigned vlane_off = (unsigned)lh * VBLK + (unsigned)(cq * 128 + li) * 16u;   // bytes, per lane: row block lh of a step

    if (c_lo < c_hi) chunk_load_async8(sK, K + (size_t)c_lo * CH * DK, min(CH, B - c_lo * CH), wave, lane);
    __syncthreads();

    const int qcol = wq * 32 + li;
    const bool qok = (q0 + qcol) < p.HW;
    float qm = 1e30f, qinv = 0.f;
    if (qok) {
        qm = p.ml[((size_t)obj * p.HW + q0 + qcol) * 2];
        qinv = 1.f / p.ml[((size_t)obj * p.HW + q0 + qcol) * 2 + 1];
    }

    for (int c = c_lo; c < c_hi; ++c) {
        const int b0 = c * CH;
        const bool more = c + 1 < c_hi;
        // buffer descriptor over this chunk's live row blocks: blocks past the bank end read 0 through the range check; rows
        // past the end inside the last live block hold zeros or stale (finite) values -- their P is exactly 0
        const int live = min(CH, B - b0);
        const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(V + (size_t)(b0 >> 3) * VBLK), 0, ((live + 7) >> 3) * VBLK, 0x00020000);

        // bf16x3: two steps of operands in flight, [tc][hi, lo].  Plain bf16 (round 5): its steps are 8 MFMAs = 256 cycles, one step of
        // lead left the kernel waiting on memory (PMC: 0.64 of the wave cycles, mfma_util 0.27) -- three of the chunk's four steps
        // are requested here, in front of the score GEMM, the fourth into step 0's slot behind step 0's MFMAs (a fourth slot spills:
        // 52 bytes of scratch at 256 registers)
        constexpr int NSLOT = X3 ? 2 : 3;
        u32x4 vop[NSLOT][8];
        auto load_v = [&](int st, u32x4 (&dst)[8]) {
#pragma unroll
            for (int tc = 0; tc < 4; ++tc) {
#ifdef VFN_ABLATE_V
                dst[2 * tc] = u32x4{(unsigned)lane, (unsigned)st, 0x3f803f80u, 0x3f803f80u};
                dst[2 * tc + 1] = u32x4{0x3f803f80u, (unsigned)tc, 0u, 0u};
#else
                dst[2 * tc] = __builtin_amdgcn_raw_buffer_load_b128(vrsrc, vlane_off, 2 * st * VBLK + tc * 512, 0);
                if constexpr (X3) dst[2 * tc + 1] = __builtin_amdgcn_raw_buffer_load_b128(vrsrc, vlane_off, 2 * st * VBLK + DV * 16 + tc * 512, 0);
#endif
            }
        };
        if constexpr (X3) load_v(0, vop[0]);         // lands behind the score GEMM
        else {
#pragma unroll
            for (int st = 0; st < 3; ++st) load_v(st, vop[st]);
        }

        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {
            const int ra = wr * 32 + li, rq = wq * 32 + li;
            const char* kb = reinterpret_cast<const char*>(sK);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(kb + swzk(ra, 0, 2 * g + lh));
                bf16x8 al = ah;
                if constexpr (X3) al = *reinterpret_cast<const bf16x8*>(kb + swzk(ra, 1, 2 * g + lh));
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(sQh + swzq(rq, 2 * g + lh));
                bf16x8 bl = bh;
                if constexpr (X3) bl = *reinterpret_cast<const bf16x8*>(sQl + swzq(rq, 2 * g + lh));
                mfma_lp<X3>(acc, ah, al, bh, bl);
            }
        }

        const int rloc = wr * 32 + 4 * lh;
        const int mycnt = softmax_hits(acc, p.scale, qm, qinv, p.thres, B - b0, rloc);
#pragma unroll
        for (int g = 0; g < 4; ++g) {                // registers 4g..4g+3 = 4 consecutive bank rows of query qcol
            const int brow = rloc + 8 * g;
            bf16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) { h[e] = (__bf16)acc[4 * g + e]; l[e] = (__bf16)(acc[4 * g + e] - (float)h[e]); }
            const int off = swzp(qcol, brow >> 3) + ((brow >> 2) & 1) * 8;
            *reinterpret_cast<bf16x4*>(sPh + off) = h;
            if constexpr (X3) *reinterpret_cast<bf16x4*>(sPl + off) = l;
        }
        if (p.cnt && lh == 0 && mycnt > 0) {
            const int row = b0 + wr * 32 + li;
            if (row < B) atomicAdd(p.cnt + (size_t)obj * p.stride_cnt + row, mycnt);
        }
        __syncthreads();                             // P^T visible; every wave is done reading sK
        if (more) chunk_load_async8(sK, K + (size_t)(b0 + CH) * DK, min(CH, B - b0 - CH), wave, lane);

        // O^T[q][ch] += sum_b P^T[q][b] V[b][ch]: A = P^T (2 query tiles), B = value rows (4 channel tiles)
#pragma unroll
        for (int st = 0; st < CH / 16; ++st) {
            if constexpr (X3) { if (st + 1 < CH / 16) load_v(st + 1, vop[(st + 1) & 1]); }
            const u32x4 (&vv)[8] = vop[X3 ? (st & 1) : st % 3];
#pragma unroll
            for (int tq = 0; tq < 2; ++tq) {
                const int prow = qhalf * 64 + tq * 32 + li;
                const bf16x8 ph = *reinterpret_cast<const bf16x8*>(sPh + swzp(prow, 2 * st + lh));
                bf16x8 pl = ph;
                if constexpr (X3) pl = *reinterpret_cast<const bf16x8*>(sPl + swzp(prow, 2 * st + lh));
#pragma unroll
                for (int tc = 0; tc < 4; ++tc) {
                    const bf16x8 vh = __builtin_bit_cast(bf16x8, vv[2 * tc]);
                    const bf16x8 vl = X3 ? __builtin_bit_cast(bf16x8, vv[2 * tc + 1]) : vh;
                    mfma_lp<X3>(o[tq][tc], ph, pl, vh, vl);
                }
            }
            if constexpr (!X3) {
                if (st == 0) { __builtin_amdgcn_sched_barrier(0); load_v(3, vop[0]); }
            }
        }
        __syncthreads();
    }

    float* dst = p.o_part + ((size_t)obj * p.nsplit + split) * p.HW * DV;
#pragma unroll
    for (int tq = 0; tq < 2; ++tq)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = q0 + qhalf * 64 + tq * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (q < p.HW) {
#pragma unroll
                for (int tc = 0; tc < 4; ++tc) dst[(size_t)q * DV + cq * 128 + tc * 32 + li] = o[tq][tc][r];
            }
        }
}

// ------------------------------------------------------------------ pass 2, plain bf16, software-pipelined (round 6)
// memread_apply_shw_kernel<false> walks a chunk as  [score GEMM -> softmax -> P^T to LDS] barrier [P^T V] barrier : all eight waves do
// the vector work (exp, threshold ballots, bf16 conversion: ~850 VALU cycles per wave and chunk, v_exp at quarter rate) TOGETHER
// between two barriers, with the matrix pipe idle, then the matrix work together with the vector pipe idle -- and its value rows are
// requested at most one chunk phase ahead (rocprofv3, round 5: mfma_util 0.27, 0.64 of the wave cycles waiting; 0.27 of the bf16 peak
// on the kernel that is 50-70 % of a C5 frame).  Here the loop is skewed by one chunk INSIDE every wave:
//
//     iteration c:   score GEMM of chunk c+1 (8 MFMAs)
//                    4 x { P^T V step of chunk c (8 MFMAs)  ||  a quarter of chunk c+1's softmax (VALU, in the MFMAs' shadow)
//                          -> its quarter of P^T(c+1) into the OTHER P^T buffer;  the value rows of (c+1, step) requested into the
//                          operand registers the step has just consumed: a whole chunk of lead }
//                    hit counts of chunk c+1; keys of chunk c+2: registers -> LDS        ONE barrier per chunk
//
// Keys go through registers (two 16-byte loads per thread at the top of an iteration, stored to LDS in front of its barrier), not by
// LDS-DMA: with a DMA in flight hipcc turns every wait for a register load into a full drain, which would
// put the value rows' latency back on the chain.  Only the hi plane of the key image is staged (256-byte rows: 16 KB per chunk).
// Registers are what the structure costs (128 accumulators + 64 value operands + 16 scores + 8 keys of 256): every LDS address is
// ONE xor away from a per-lane base (the swizzles are xors with lane constants), and the bases are made opaque once per iteration so
// that hipcc does not hoist the 30 derived addresses into registers of their own (a first form spilled 21 registers; each reload of a
// spilled ADDRESS was a scratch load with a full vmcnt(0) drain in front of the LDS access -- the value rows' lead thrown away).
// Same products in the same order as memread_apply_shw_kernel<false> (same fragments, same k order, same chunk order): bit-identical.
// LDS: query image 32 KB + 2 key chunks 32 KB + 2 P^T 32 KB = 96 KB; one workgroup of 8 waves per CU.
constexpr size_t APPLY_PIPE_LDS = (size_t)QTW * DK * 2 + 2 * (size_t)CH * DK * 2 + 2 * (size_t)QTW * CH * 2;
constexpr unsigned PIPE_K0 = QTW * DK * 2;                    // byte offsets inside the dynamic LDS: key buffers
constexpr unsigned PIPE_P0 = PIPE_K0 + 2 * CH * DK * 2;       // P^T buffers
constexpr unsigned PIPE_KB = CH * DK * 2, PIPE_PB = QTW * CH * 2;     // bytes of one key / P^T buffer

__global__ __launch_bounds__(512, 1)
void memread_apply_pipe_kernel(const vfn_memread_desc p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sQh = smem;                                         // [128 q][256 B] bf16, swzq

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wr = wave >> 2, wq = wave & 3;                  // score tile: key rows 32wr.., query columns 32wq..
#ifdef PIPE_2X4
    constexpr int TQN = 2, TCN = 4;                           // P^T V tiles per wave: 2 query tiles x 4 channel tiles (memread_apply_shw_kernel's)
    const int qbase = (wave >> 2) * 64, cbase = (wave & 3) * 128;
#else
    // 4 query tiles x 2 channel tiles: every value row is requested by ONE wave of the workgroup.  With 2 x 4 the two waves that share
    // a channel quarter both request it -- 144 KB per chunk and CU through L1 / L2 instead of 80 KB, 15 TB/s chip-wide at the rate this
    // kernel runs: the L2's limit (an ablation with constant value operands ran 28 % faster).  P^T fragments are read from LDS twice as
    // often instead (128 KB per chunk: 1024 of ~5000 cycles), and the operand registers halve (32 instead of 64).
    constexpr int TQN = 4, TCN = 2;
    const int qbase = 0, cbase = wave * 64;
#endif
    int split, qt, obj;
    apply_item(p, split, qt, obj);
    const int q0 = qt * QTW;
    const int B = p.bank_len[obj];
    const char* Kimg = reinterpret_cast<const char*>(p.bank_k_lp) + (size_t)obj * p.stride_k * 4;   // image rows of 512 B: [128 hi | 128 lo]
    const char* V = reinterpret_cast<const char*>(p.bank_v_lp) + (size_t)obj * p.stride_v * 4;      // blocks of 8 rows, 16 KB

    {   // query image
        const int c = tid & 31;
        for (int r = tid >> 5; r < QTW; r += 16) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (q0 + r < p.HW) v = *reinterpret_cast<const f32x4*>(p.q + (size_t)(q0 + r) * p.ldq + c * 4);
            const bf16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            *reinterpret_cast<bf16x4*>(sQh + swzq(r, c >> 1) + (c & 1) * 8) = h;
        }
    }

    int c_lo, c_hi;
    chunk_range(B, p.nsplit, split, c_lo, c_hi);
    const int n = c_hi - c_lo;

    f32x16 o[TQN][TCN];                                       // O^T tiles: [query tile][channel tile]
#pragma unroll
    for (int a = 0; a < TQN; ++a)
#pragma unroll
        for (int b = 0; b < TCN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[a][b][r] = 0.f;

    const int qcol = wq * 32 + li;
    const bool qok = (q0 + qcol) < p.HW;
    float qm = 1e30f, qinv = 0.f;
    if (qok) {
        qm = p.ml[((size_t)obj * p.HW + q0 + qcol) * 2];
        qinv = 1.f / p.ml[((size_t)obj * p.HW + q0 + qcol) * 2 + 1];
    }
    const float sm = -qm;
    const int rloc = wr * 32 + 4 * lh;                        // this lane's first score row inside a chunk (+ (r & 3) + 8 (r >> 2))

    if (n > 0) {
        const unsigned vlane_off = (unsigned)lh * VBLK + (unsigned)(cbase + li) * 16u;         // bytes: row block lh of a step, channel
        const int krow = tid >> 3, kc2 = (tid & 7) * 2;                                        // key staging: row, first of two 16-byte chunks
        const unsigned klane_off = (unsigned)krow * 512u + (unsigned)kc2 * 16u;
        // LDS byte addresses, each ONE xor away from a per-lane base (swzq / swzp are xors of the 16-byte chunk index with lane constants):
        //   key fragment g of the score GEMM     bK ^ 32 g  (+ buffer)     bK = swzq(32 wr + li, lh)
        //   query fragment g                     bQ ^ 32 g                 bQ = swzq(32 wq + li, lh)
        //   P^T fragment of step st, tile tq     bR ^ 32 st + 4096 tq (+ buffer)      bR = swzp(qbase + li, lh)
        //   P^T store of softmax quarter g       bW ^ 16 g  (+ buffer)     bW = swzp(qcol, 4 wr) + 8 lh   (rows rloc + 8 g: chunk 4 wr + g, half lh)
        //   key staging store                    swzq(krow, kc2), swzq(krow, kc2 + 1)  (+ buffer)
        unsigned bK = PIPE_K0 + (unsigned)swzq(wr * 32 + li, lh);
        unsigned bQ = (unsigned)swzq(wq * 32 + li, lh);
        unsigned bR = PIPE_P0 + (unsigned)swzp(qbase + li, lh);
        unsigned bW = PIPE_P0 + (unsigned)swzp(qcol, 4 * wr) + 8u * lh;
        unsigned bS = PIPE_K0 + (unsigned)swzq(krow, kc2);
        const unsigned bS1 = (unsigned)(swzq(krow, kc2 + 1) - swzq(krow, kc2));                 // (+16 or -16: the pair's second chunk)
        auto lds = [&](unsigned a) { return smem + a; };
        u32x4 vop[4][TCN];                                    // value operands: [step][channel tile] (8 bank rows x 1 channel, bf16)
        u32x4 kreg[2];
        f32x16 acc;
        int mycnt = 0;

        // buffer descriptors over ONE chunk's live rows: key rows / value row blocks past the bank end read zeros through the range check
        auto vres = [&](int c) {
            const int b0 = c * CH;
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(V + (size_t)(b0 >> 3) * VBLK), 0, ((min(CH, B - b0) + 7) >> 3) * (int)VBLK, 0x00020000);
        };
        auto kres = [&](int c) {
            const int b0 = c * CH;
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Kimg + (size_t)b0 * 512), 0, min(CH, B - b0) * 512, 0x00020000);
        };
        auto load_v = [&](const __amdgpu_buffer_rsrc_t& r, int st, auto SLOT_) {
            constexpr int slot = decltype(SLOT_)::value;
#ifdef PIPE_ABLATE_V
            for (int tc = 0; tc < TCN; ++tc) vop[slot][tc] = u32x4{0x3f803f80u + (unsigned)st, 0x3f803f80u, 0x3f803f80u + (unsigned)tc, 0x3f803f80u};
#else
#pragma unroll
            for (int tc = 0; tc < TCN; ++tc) vop[slot][tc] = __builtin_amdgcn_raw_buffer_load_b128(r, vlane_off, 2 * st * (int)VBLK + tc * 512, 0);
#endif
        };
        auto load_k = [&](int c) {
            const __amdgpu_buffer_rsrc_t r = kres(c);
            kreg[0] = __builtin_amdgcn_raw_buffer_load_b128(r, klane_off, 0, 0);
            kreg[1] = __builtin_amdgcn_raw_buffer_load_b128(r, klane_off, 16, 0);
        };
        auto store_k = [&](int buf) {
            const unsigned a = bS + buf * PIPE_KB;
            *reinterpret_cast<u32x4*>(lds(a)) = kreg[0];
            *reinterpret_cast<u32x4*>(lds(a + bS1)) = kreg[1];
        };
        auto score = [&](int buf) {
            const unsigned ak = bK + buf * PIPE_KB;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            // fragments of k-group g + 1 are requested in front of the MFMA of g (a first form read, waited, multiplied: eight exposed LDS
            // round trips with the matrix pipe idle -- both waves of a SIMD are in this section together)
            bf16x8 fa[2], fb[2];
            fa[0] = *reinterpret_cast<const bf16x8*>(lds(ak));
            fb[0] = *reinterpret_cast<const bf16x8*>(lds(bQ));
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (g + 1 < 8) {
                    fa[(g + 1) & 1] = *reinterpret_cast<const bf16x8*>(lds(ak ^ (32u * (g + 1))));
                    fb[(g + 1) & 1] = *reinterpret_cast<const bf16x8*>(lds(bQ ^ (32u * (g + 1))));
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g & 1], fb[g & 1], acc, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // 2 LDS reads, then 1 MFMA: keeps the requests one group ahead
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            }
        };
        // a quarter of a chunk's softmax: score registers 4g .. 4g+3 = bank rows rloc + 8g + (0..3) of query qcol -> p, hit ballots, P^T
        auto softmax_quarter = [&](auto G_, int pbuf) {
            constexpr int g = decltype(G_)::value;
            float pv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) pv[e] = fast_exp(fmaf(acc[4 * g + e], p.scale, sm)) * qinv;
#ifndef PIPE_NO_HITS
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned long long hit = __ballot(pv[e] > p.thres);
                const int h_lo = __popcll(hit & 0xffffffffull), h_hi = __popcll(hit >> 32);      // wave-uniform
                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(mycnt) : "s"(h_lo), "n"(e + 8 * g));
                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(mycnt) : "s"(h_hi), "n"(e + 8 * g + 4));
            }
#endif
            const bf16x4 h = {(__bf16)pv[0], (__bf16)pv[1], (__bf16)pv[2], (__bf16)pv[3]};
            *reinterpret_cast<bf16x4*>(lds((bW + pbuf * PIPE_PB) ^ (16u * g))) = h;
        };
        // the bank's last chunk may be partial: its rows past the end got p = exp(0 - m) / l from the zero keys -- exactly 0 is required
        // (their value rows are stale).  Rare (one chunk per object): a fix-up of the lane's own P^T entries, outside the pipelined stream
        auto zero_dead_rows = [&](int pbuf, int nvalid) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (rloc + 8 * g + e >= nvalid) *reinterpret_cast<unsigned short*>(lds(((bW + pbuf * PIPE_PB) ^ (16u * g)) + e * 2)) = 0;
        };
        auto add_hits = [&](int c) {                          // one wave per 32 x 32 score tile: lane i (0..31) holds the hits of chunk row 32wr + i
            if (p.cnt && lh == 0 && mycnt > 0) {
                const int row = c * CH + wr * 32 + li;
                if (row < B) atomicAdd(p.cnt + (size_t)obj * p.stride_cnt + row, mycnt);
            }
            mycnt = 0;
        };
        auto pv_step = [&](auto ST_, auto SLOT_, int pbuf) {   // O^T[q][ch] += sum_b P^T[q][b] V[b][ch], 16 bank rows: TQN query tiles x TCN channel tiles
            constexpr int st = decltype(ST_)::value, slot = decltype(SLOT_)::value;
            const unsigned ar = (bR + pbuf * PIPE_PB) ^ (32u * st);
#pragma unroll
            for (int tq = 0; tq < TQN; ++tq) {
                const bf16x8 ph = *reinterpret_cast<const bf16x8*>(lds(ar + tq * 4096));
#ifdef PIPE_SETPRIO
                __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                for (int tc = 0; tc < TCN; ++tc)
                    o[tq][tc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph, __builtin_bit_cast(bf16x8, vop[slot][tc]), o[tq][tc], 0, 0, 0);
#ifdef PIPE_SETPRIO
                __builtin_amdgcn_s_setprio(0);
#endif
            }
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        auto opaque = [&]() {                                  // (keeps the derived LDS addresses out of loop-invariant registers)
            asm volatile("" : "+v"(bK), "+v"(bQ), "+v"(bR), "+v"(bW), "+v"(bS));
        };

        // ---- prologue: chunk c_lo's keys, value operands and P^T; chunk c_lo + 1's keys
        {
            const __amdgpu_buffer_rsrc_t vr = vres(c_lo);
            load_k(c_lo);
            load_v(vr, 0, I0{}); load_v(vr, 1, I1{}); load_v(vr, 2, I2{}); load_v(vr, 3, I3{});
            store_k(0);
            if (n > 1) load_k(c_lo + 1);
            __syncthreads();
            score(0);
            softmax_quarter(I0{}, 0); softmax_quarter(I1{}, 0); softmax_quarter(I2{}, 0); softmax_quarter(I3{}, 0);
            if (B - c_lo * CH < CH) zero_dead_rows(0, B - c_lo * CH);
            add_hits(c_lo);
            if (n > 1) store_k(1);
            __syncthreads();
        }

        // one iteration: chunk c = c_lo + i.  NEXT: chunk c + 1 exists (its scores / softmax / value requests ride along);
        // KNEXT: chunk c + 2 exists (its keys are staged)
        auto iter = [&](auto NEXT_, auto KNEXT_, int i) {
            constexpr bool NEXT = decltype(NEXT_)::value, KNEXT = decltype(KNEXT_)::value;
            const int c = c_lo + i;
            const int pc = i & 1, pn = pc ^ 1;
            opaque();
            if constexpr (KNEXT) load_k(c + 2);
            const __amdgpu_buffer_rsrc_t vr1 = vres(NEXT ? c + 1 : c);
#ifdef PIPE_NO_SCHEDBAR
#define PIPE_SB()
#else
#define PIPE_SB() __builtin_amdgcn_sched_barrier(0)
#endif
#ifdef PIPE_SCORE_LATE
            // (experiment) the score GEMM behind step 0's MFMAs, the softmax quarters one step later
            pv_step(I0{}, I0{}, pc);
            if constexpr (NEXT) { load_v(vr1, 0, I0{}); score(pn); }
            PIPE_SB();
            pv_step(I1{}, I1{}, pc);
            if constexpr (NEXT) { load_v(vr1, 1, I1{}); softmax_quarter(I0{}, pn); }
            PIPE_SB();
            pv_step(I2{}, I2{}, pc);
            if constexpr (NEXT) { load_v(vr1, 2, I2{}); softmax_quarter(I1{}, pn); }
            PIPE_SB();
            pv_step(I3{}, I3{}, pc);
            if constexpr (NEXT) { load_v(vr1, 3, I3{}); softmax_quarter(I2{}, pn); softmax_quarter(I3{}, pn); }
#else
            if constexpr (NEXT) score(pn);
            pv_step(I0{}, I0{}, pc);
            if constexpr (NEXT) { load_v(vr1, 0, I0{}); softmax_quarter(I0{}, pn); }
            PIPE_SB();
            pv_step(I1{}, I1{}, pc);
            if constexpr (NEXT) { load_v(vr1, 1, I1{}); softmax_quarter(I1{}, pn); }
            PIPE_SB();
            pv_step(I2{}, I2{}, pc);
            if constexpr (NEXT) { load_v(vr1, 2, I2{}); softmax_quarter(I2{}, pn); }
            PIPE_SB();
            pv_step(I3{}, I3{}, pc);
            if constexpr (NEXT) { load_v(vr1, 3, I3{}); softmax_quarter(I3{}, pn); }
#endif
            if constexpr (NEXT) {
                if (B - (c + 1) * CH < CH) zero_dead_rows(pn, B - (c + 1) * CH);
                add_hits(c + 1);
            }
            if constexpr (KNEXT) store_k(pc);                        // (chunk c's keys, read by the previous iteration's score GEMM, are dead)
            __syncthreads();
        };
        const std::true_type T{};
        const std::false_type F{};
        int i = 0;
        for (; i + 2 < n; ++i) iter(T, T, i);
        if (i + 1 < n) { iter(T, F, i); ++i; }
        iter(F, F, i);
    }

    float* dst = p.o_part + ((size_t)obj * p.nsplit + split) * p.HW * DV;
#pragma unroll
    for (int tq = 0; tq < TQN; ++tq)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = q0 + qbase + tq * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (q < p.HW) {
#pragma unroll
                for (int tc = 0; tc < TCN; ++tc) dst[(size_t)q * DV + cbase + tc * 32 + li] = o[tq][tc][r];
            }
        }
}

#ifdef VFN_CENSUS
#define PH_DECL unsigned long long ph_t = __builtin_amdgcn_s_memtime(), ph_acc[5] = {0, 0, 0, 0, 0}
#define PH_MARK(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_acc[k] += t_ - ph_t; ph_t = t_; } while (0)
#else
#define PH_DECL
#define PH_MARK(k)
#endif
// ------------------------------------------------------------------ pass 2 (f32), wide query tile
// 128 query columns per workgroup, 8 waves: the score tile is 64 x 128 (one 32x32 tile per wave), wave w then owns
// value channels 64w..64w+63 for all 128 queries: a key chunk and a value row are fetched once per 128 queries and a
// chunk costs one barrier pair per 128 queries.
// LDS: queries [128][128] 64 KB + key chunk 32 KB + P^T [128 q][64 b] 32 KB = 128 KB, one workgroup per CU.
__global__ __launch_bounds__(512, 1)
void memread_apply_wide_kernel(const vfn_memread_desc p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sQ = reinterpret_cast<float*>(smem);      // [128][128]
    float* sK = sQ + QTW * DK;                       // [64][128]
    float* sP = sK + CH * DK;                        // [128 q][64 b]  (P^T)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wr = wave >> 2, wq = wave & 3;
    int split, qt, obj;
    apply_item(p, split, qt, obj);
    const int q0 = qt * QTW;
    const int B = p.bank_len[obj];
    const float* K = p.bank_k + (size_t)obj * p.stride_k;
    const float* V = p.bank_v + (size_t)obj * p.stride_v;

    {   // query image (zero past HW)
        const int c = tid & 31;
        for (int r = tid >> 5; r < QTW; r += 16) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (q0 + r < p.HW) v = *reinterpret_cast<const f32x4*>(p.q + (size_t)(q0 + r) * p.ldq + c * 4);
            *reinterpret_cast<f32x4*>(sQ + swz(r, c)) = v;
        }
    }

    int c_lo, c_hi;
    chunk_range(B, p.nsplit, split, c_lo, c_hi);

    f32x16 o[4][2];                                  // O^T tiles: [query tile][channel tile]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[a][b][r] = 0.f;

    const float* vcol = V + wave * 64 + li * 2;      // + row*512; lane li owns channels 2*li, 2*li+1 (tile tc)
    const unsigned vlane_off = (unsigned)((4 * lh) * DV + wave * 64 + li * 2) * 4u;   // bytes, per lane

    if (c_lo < c_hi) chunk_load_async8(sK, K + (size_t)c_lo * CH * DK, min(CH, B - c_lo * CH), wave, lane);
    __syncthreads();

    const int qcol = wq * 32 + li;
    const bool qok = (q0 + qcol) < p.HW;
    float qm = 1e30f, qinv = 0.f;
    if (qok) {
        qm = p.ml[((size_t)obj * p.HW + q0 + qcol) * 2];
        qinv = 1.f / p.ml[((size_t)obj * p.HW + q0 + qcol) * 2 + 1];
    }

    PH_DECL;
    for (int c = c_lo; c < c_hi; ++c) {
        const int b0 = c * CH;
        const bool more = c + 1 < c_hi;

        f32x16 acc;
        score_tile(sK, sQ, wr, wq, li, lh, acc);
        PH_MARK(0);

        const int rloc = wr * 32 + 4 * lh;
        const int mycnt = softmax_hits(acc, p.scale, qm, qinv, p.thres, B - b0, rloc);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int brow = rloc + 8 * g;
            const f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            *reinterpret_cast<f32x4*>(sP + swz64(qcol, brow >> 2)) = v;
        }
        if (p.cnt && lh == 0 && mycnt > 0) {
            const int row = b0 + wr * 32 + li;
            if (row < B) atomicAdd(p.cnt + (size_t)obj * p.stride_cnt + row, mycnt);
        }
        PH_MARK(1);
        __syncthreads();                             // P^T visible; every wave is done reading sK
        PH_MARK(2);
        if (more) chunk_load_async8(sK, K + (size_t)(b0 + CH) * DK, min(CH, B - b0 - CH), wave, lane);

        f32x2 vb[3][4];                              // ring: value rows two k-groups ahead of their MFMAs
        // whole chunks read their value rows through a buffer descriptor (scalar base and row offset, one per-lane byte
        // offset): the 64-bit clamped addresses of the tail form cost ~7 vector-ALU instructions per load, and VALU
        // issue comes straight out of the MFMA pipe's time on this SIMD
        // (round 5: the resource ends at the bank's last row -- one load form for every chunk, see memread_apply_ss_kernel)
        const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(V + (size_t)b0 * DV), 0, min(CH, B - b0) * DV * 4, 0x00020000);
        auto load_v = [&](int kk, int slot) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                vb[slot][t] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(vrsrc, vlane_off + (unsigned)((8 * kk + t) * DV * 4), 0, 0));
        };
        load_v(0, 0);
        load_v(1, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < CH / 8; ++kk) {
            const int cur = kk % 3;
            if (kk + 2 < CH / 8) load_v(kk + 2, (kk + 2) % 3);
            __builtin_amdgcn_sched_barrier(0);
            const int lc = 2 * kk + lh;
            f32x4 a[4];
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) a[tq] = *reinterpret_cast<const f32x4*>(sP + swz64(tq * 32 + li, lc));
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) {
                    o[tq][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tq][t], vb[cur][t][0], o[tq][0], 0, 0, 0);
                    o[tq][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tq][t], vb[cur][t][1], o[tq][1], 0, 0, 0);
                }
        }
        PH_MARK(3);
        __syncthreads();                             // next chunk's keys visible; sP free again
        PH_MARK(4);
    }
#ifdef VFN_CENSUS
    if ((threadIdx.x & 63) == 0) {
        const int lin = (blockIdx.y * gridDim.x + blockIdx.x) * 8 + (threadIdx.x >> 6);
        if (lin < 2048) {
            for (int k = 0; k < 5; ++k) vfn_census_buf[lin * 8 + k] = ph_acc[k];
            vfn_census_buf[lin * 8 + 5] = c_hi - c_lo;
        }
    }
#endif

    float* dst = p.o_part + ((size_t)obj * p.nsplit + split) * p.HW * DV;
#pragma unroll
    for (int tq = 0; tq < 4; ++tq)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = q0 + tq * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (q < p.HW) {
                const f32x2 v = {o[tq][0][r], o[tq][1][r]};
                *reinterpret_cast<f32x2*>(dst + (size_t)q * DV + wave * 64 + li * 2) = v;
            }
        }
}

// ------------------------------------------------------------------ pass 2 (f32), wide tile, scores from the scan
// The statistics scan has already formed every score <key, query> of the frame; with vfn_bankscan_desc.scores it also
// stores them (HW x B floats per object: 363 MB at the C2 mean bank, written once and read once -- the bytes the key
// chunks cost before).  This kernel then has no score GEMM, no query image and no key chunks: it loads the 16 scores of
// its 32 x 32 tile position (a chunk ahead), runs the softmax, and goes on as memread_apply_wide_kernel.  The scan sums
// the same products in the same order, so O^T and the hit counts are bit-identical to the recomputing kernel; the frame
// executes F_min's 1280 B HW FLOP per object for the memory read instead of 1536.
// LDS: two P^T buffers [128 q][64 b], 64 KB.
__global__ __launch_bounds__(512, 1)
void memread_apply_ss_kernel(const vfn_memread_desc p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sP = reinterpret_cast<float*>(smem);      // [128 q][64 b]  (P^T)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wr = wave >> 2, wq = wave & 3;
    int split, qt, obj;
    apply_item(p, split, qt, obj);
    const int q0 = qt * QTW;
    const int qtiles = (p.HW + QTW - 1) / QTW;
    const int B = p.bank_len[obj];
    const float* V = p.bank_v + (size_t)obj * p.stride_v;
    const float* S = p.scores + (size_t)obj * p.stride_scores;

    int c_lo, c_hi;
    chunk_range(B, p.nsplit, split, c_lo, c_hi);

    f32x16 o[4][2];                                  // O^T tiles: [query tile][channel tile]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[a][b][r] = 0.f;

    const float* vcol = V + wave * 64 + li * 2;      // + row*512; lane li owns channels 2*li, 2*li+1 (tile tc)
    const unsigned vlane_off = (unsigned)((4 * lh) * DV + wave * 64 + li * 2) * 4u;   // bytes, per lane
    // this lane's 16 scores of a chunk: key rows 32wr + (r&3) + 8(r>>2) + 4lh, query column 32wq + li
    const unsigned slane_off = (unsigned)(((wr * 4) * 2 + lh) * QTW + wq * 32 + li) * 16u;      // bytes; + g * 2 * QTW * 16
    f32x4 sv[4];
    auto load_scores = [&](int c) {
        const char* tile = reinterpret_cast<const char*>(S + ((size_t)c * qtiles + qt) * (CH * QTW));
#pragma unroll
        for (int g = 0; g < 4; ++g) sv[g] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(tile + slane_off + g * (2 * QTW * 16)));
    };
    const int qcol = wq * 32 + li;
    const bool qok = (q0 + qcol) < p.HW;
    float qm = 1e30f, qinv = 0.f;
    if (qok) {
        qm = p.ml[((size_t)obj * p.HW + q0 + qcol) * 2];
        qinv = 1.f / p.ml[((size_t)obj * p.HW + q0 + qcol) * 2 + 1];
    }
    const int rloc = wr * 32 + 4 * lh;
    // softmax of chunk c from the loaded scores -> P^T buffer `buf`, hit counts
    auto softmax_to = [&](int c, float* sPb) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = sv[r >> 2][r & 3];
        const int b0 = c * CH;
        const int mycnt = softmax_hits(acc, p.scale, qm, qinv, p.thres, B - b0, rloc);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int brow = rloc + 8 * g;
            const f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            *reinterpret_cast<f32x4*>(sPb + swz64(qcol, brow >> 2)) = v;
        }
        if (p.cnt && lh == 0 && mycnt > 0) {
            const int row = b0 + wr * 32 + li;
            if (row < B) atomicAdd(p.cnt + (size_t)obj * p.stride_cnt + row, mycnt);
        }
    };

    // value rows: a ring of four k-groups that runs ACROSS the chunks (round 5).  Within a chunk a group is requested two groups ahead of
    // its MFMAs; the last two groups of a chunk request the first two of the NEXT chunk (8 groups per chunk, 8 % 4 = 0: the slots stay
    // compile-time constants), so a chunk no longer opens with the matrix pipe waiting a full memory latency for its first value rows --
    // both waves of a SIMD leave the chunk barrier together and used to wait there together.  A chunk's buffer resource ends at the
    // bank's last row (rows past it read 0; P is exactly 0 there); behind the slice's last chunk it is empty (loads return 0, no traffic).
    f32x2 vb[4][4];
    auto chunk_rsrc = [&](int c) {
        const int rows = c < c_hi ? min(CH, B - c * CH) : 0;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(V + (size_t)min(c, max(c_hi - 1, 0)) * CH * DV), 0, rows * DV * 4, 0x00020000);
    };
    auto load_v = [&](const __amdgpu_buffer_rsrc_t& rs, int kk, int slot) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            vb[slot][t] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, vlane_off + (unsigned)((8 * kk + t) * DV * 4), 0, 0));
    };
    if (c_lo < c_hi) {
        const __amdgpu_buffer_rsrc_t rs0 = chunk_rsrc(c_lo);
        load_v(rs0, 0, 0);
        load_v(rs0, 1, 1);
        load_scores(c_lo);
        softmax_to(c_lo, sP);
        if (c_lo + 1 < c_hi) load_scores(c_lo + 1);
    }
    __syncthreads();                                 // P^T of the first chunk visible

    // P^T is double-buffered: the softmax of chunk c+1 is issued in the middle of chunk c's P^T V loop (plain VALU work on
    // registers -- the scores were loaded a chunk ahead -- that runs under the queued MFMAs) and written to the other
    // buffer, so a chunk costs ONE barrier and the matrix pipe does not idle through the softmax
    for (int c = c_lo; c < c_hi; ++c) {
        const int b0 = c * CH;
        const int buf = (c - c_lo) & 1;
        const float* sPc = sP + buf * (QTW * CH);
        float* sPn = sP + (buf ^ 1) * (QTW * CH);
        const bool nxt = c + 1 < c_hi;

        const __amdgpu_buffer_rsrc_t vrsrc = chunk_rsrc(c), vrsrc_n = chunk_rsrc(c + 1);
        f32x4 a[2][4];                               // P^T fragments, one k-group ahead of their MFMAs
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) a[0][tq] = *reinterpret_cast<const f32x4*>(sPc + swz64(tq * 32 + li, lh));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < CH / 8; ++kk) {
            const int cur = kk % 4;
            if (kk + 2 < CH / 8) load_v(vrsrc, kk + 2, (kk + 2) % 4);
            else load_v(vrsrc_n, kk + 2 - CH / 8, (kk + 2) % 4);
            if (kk + 1 < CH / 8) {
#pragma unroll
                for (int tq = 0; tq < 4; ++tq)
                    a[(kk + 1) & 1][tq] = *reinterpret_cast<const f32x4*>(sPc + swz64(tq * 32 + li, 2 * (kk + 1) + lh));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) {
                    o[tq][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk & 1][tq][t], vb[cur][t][0], o[tq][0], 0, 0, 0);
                    o[tq][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk & 1][tq][t], vb[cur][t][1], o[tq][1], 0, 0, 0);
                }
            if (kk == 2 && nxt) {
                softmax_to(c + 1, sPn);
                if (c + 2 < c_hi) load_scores(c + 2);
            }
        }
        __syncthreads();                             // P^T of chunk c+1 visible; everyone is done with this buffer
    }

    float* dst = p.o_part + ((size_t)obj * p.nsplit + split) * p.HW * DV;
#pragma unroll
    for (int tq = 0; tq < 4; ++tq)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = q0 + tq * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (q < p.HW) {
                const f32x2 v = {o[tq][0][r], o[tq][1][r]};
                *reinterpret_cast<f32x2*>(dst + (size_t)q * DV + wave * 64 + li * 2) = v;
            }
        }
}

// out[obj][q][0:512] = sum_split o_part ; out[obj][q][512:1024] = query value; then the hit-count bump
__global__ void memread_finish_kernel(const vfn_memread_desc p) {
    const int obj = blockIdx.y;
    const size_t total = (size_t)p.HW * (DV / 4);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = i % (DV / 4);
        const int q = i / (DV / 4);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int sp = 0; sp < p.nsplit; ++sp)
            s += *reinterpret_cast<const f32x4*>(p.o_part + (((size_t)obj * p.nsplit + sp) * p.HW + q) * DV + c4 * 4);
        float* o = p.out + ((size_t)obj * p.HW + q) * p.ld_out;
        *reinterpret_cast<f32x4*>(o + c4 * 4) = s;
        if (p.qv) *reinterpret_cast<f32x4*>(o + DV + c4 * 4) = *reinterpret_cast<const f32x4*>(p.qv + (size_t)q * p.ldqv + c4 * 4);
    }
    if (p.cnt) {
        const int B = p.bank_len[obj];
        int* cnt = p.cnt + (size_t)obj * p.stride_cnt;
        float* info = p.info + (size_t)obj * p.stride_info;
        for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
            info[(size_t)b * 2 + 1] += logf((float)cnt[b] + 1.f);
            cnt[b] = 0;
        }
    }
}

constexpr size_t SCAN_LDS = (size_t)(2 * CH * DK) * sizeof(float);

// The score tiles of one object: one 8192-float tile per (64-entry chunk, 128-query tile); the capacity is what the key
// slab holds per object (stride_k / 128 entries).  An undersized buffer is refused here rather than overrun on the device.
bool scores_fit(long long stride_scores, long long stride_k, int HW) {
    const long long cap = stride_k / DK;
    return stride_scores >= ((cap + CH - 1) / CH) * (long long)((HW + 127) / 128) * 8192;
}

template <typename K>
void allow_lds(K kern, size_t bytes) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

extern "C" int vfn_bank_scan(const vfn_bankscan_desc* d, void* stream) {
    if (!d || !d->q || !d->bank_k || !d->bank_len || !d->part) return VFN_ERR_ARG;
    if (d->nsplit < 1 || d->HW < 1 || d->obj_n < 1 || d->ldq % 4) return VFN_ERR_ARG;
    if (d->mode == 1 && (!d->rowscale || d->stride_rs % 4)) return VFN_ERR_ARG;
    if (d->precision < 0 || d->precision > 2) return VFN_ERR_ARG;
    if (d->scores && (d->mode != 0 || !scores_fit(d->stride_scores, d->stride_k, d->HW))) return VFN_ERR_ARG;
    static bool once = false;
    if (!once) {
        allow_lds(bank_scan_kernel<0, 0>, SCAN_LDS); allow_lds(bank_scan_kernel<1, 0>, SCAN_LDS);
        allow_lds(bank_scan_kernel<0, 1>, SCAN_LDS); allow_lds(bank_scan_kernel<1, 1>, SCAN_LDS);
        allow_lds(bank_scan_kernel<0, 2>, SCAN_LDS); allow_lds(bank_scan_kernel<1, 2>, SCAN_LDS);
        once = true;
    }
    if (!d->work_counter) return VFN_ERR_ARG;
    const int items = cdiv(d->HW, QTS) * d->nsplit * d->obj_n;
    const dim3 grid(items < 512 ? items : 512);        // two resident workgroups per CU (64 KB of LDS each)
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(d->work_counter, 0, sizeof(int), s) != hipSuccess) return VFN_ERR_LAUNCH;
    if (d->precision == 1 && d->bank_k_lp) {
        // plain bf16 on the kept key image (round 6): keys through registers, two chunks ahead; VFN_SCAN_PIPE=0 (read at every
        // call: A/B in one process) restores bank_scan_kernel<MODE, 1>
        const char* ep = getenv("VFN_SCAN_PIPE");
        if (!(ep && atoi(ep) == 0)) {
            // 32 KB of LDS and <= 160 registers: three workgroups fit a CU (the queue feeds any number of them)
            static int wgs = 0;
            if (!wgs) { const char* ew = getenv("VFN_SCAN_WGS"); wgs = ew ? atoi(ew) : 768; if (wgs < 1) wgs = 768; }
            const dim3 gridp(items < wgs ? items : wgs);
            if (d->mode == 0) hipLaunchKernelGGL((bank_scan_pipe_kernel<0>), gridp, dim3(256), SCAN_PIPE_LDS, s, *d);
            else hipLaunchKernelGGL((bank_scan_pipe_kernel<1>), gridp, dim3(256), SCAN_PIPE_LDS, s, *d);
            return vfn_check_launch();
        }
    }
    switch (d->mode * 3 + d->precision) {
        case 0: hipLaunchKernelGGL((bank_scan_kernel<0, 0>), grid, dim3(256), SCAN_LDS, s, *d); break;
        case 1: hipLaunchKernelGGL((bank_scan_kernel<0, 1>), grid, dim3(256), SCAN_LDS, s, *d); break;
        case 2: hipLaunchKernelGGL((bank_scan_kernel<0, 2>), grid, dim3(256), SCAN_LDS, s, *d); break;
        case 3: hipLaunchKernelGGL((bank_scan_kernel<1, 0>), grid, dim3(256), SCAN_LDS, s, *d); break;
        case 4: hipLaunchKernelGGL((bank_scan_kernel<1, 1>), grid, dim3(256), SCAN_LDS, s, *d); break;
        case 5: hipLaunchKernelGGL((bank_scan_kernel<1, 2>), grid, dim3(256), SCAN_LDS, s, *d); break;
        default: return VFN_ERR_ARG;
    }
    return vfn_check_launch();
}

extern "C" int vfn_bank_scan_finish(const float* part, int nsplit, int HW, int obj_n, int mode, float* ml, int* idx,
                                    float* corr, const float* colscale, void* stream) {
    if (!part) return VFN_ERR_ARG;
    const int total = obj_n * HW;
    if (mode == 0) {
        if (!ml) return VFN_ERR_ARG;
        hipLaunchKernelGGL(bank_scan_finish_kernel<0>, dim3(cdiv(total * 8, 256)), dim3(256), 0, (hipStream_t)stream,
                           part, nsplit, HW, obj_n, ml, idx, corr, colscale);
    } else {
        if (!idx || !corr || !colscale) return VFN_ERR_ARG;
        hipLaunchKernelGGL(bank_scan_finish_kernel<1>, dim3(cdiv(total * 8, 256)), dim3(256), 0, (hipStream_t)stream,
                           part, nsplit, HW, obj_n, ml, idx, corr, colscale);
    }
    return vfn_check_launch();
}

extern "C" int vfn_memread_apply(const vfn_memread_desc* d, void* stream) {
    if (!d || !d->q || !d->bank_k || !d->bank_v || !d->bank_len || !d->ml || !d->o_part) return VFN_ERR_ARG;
    if (d->nsplit < 1 || d->ldq % 4) return VFN_ERR_ARG;
    if (d->precision < 0 || d->precision > 2) return VFN_ERR_ARG;
    if (d->scores && !scores_fit(d->stride_scores, d->stride_k, d->HW)) return VFN_ERR_ARG;
    {
        static bool once_x = false;
        if (!once_x) {
            const char* e = getenv("VFN_APPLY_XCD");
            const int lin = (e && atoi(e) == 0) ? 1 : 0;
            if (lin && hipMemcpyToSymbol(HIP_SYMBOL(vfn_apply_linear_order), &lin, sizeof(int)) != hipSuccess) return VFN_ERR_LAUNCH;
            once_x = true;
        }
    }
    if (d->precision == 0) {
        static bool once_f = false;
        constexpr size_t LDS_WF = (size_t)(QTW * DK + CH * DK + QTW * CH) * sizeof(float);       // 128 KB
        if (!once_f) { allow_lds(memread_apply_wide_kernel, LDS_WF); once_f = true; }
        const dim3 gridw(cdiv(d->HW, QTW) * d->nsplit, d->obj_n);
        if (d->scores) hipLaunchKernelGGL(memread_apply_ss_kernel, gridw, dim3(512), (size_t)2 * QTW * CH * sizeof(float), (hipStream_t)stream, *d);
        else hipLaunchKernelGGL(memread_apply_wide_kernel, gridw, dim3(512), LDS_WF, (hipStream_t)stream, *d);
        return vfn_check_launch();
    }
    {
        static bool once_w = false;
        constexpr size_t LDS_W1 = (size_t)QTW * DK * 2 + (size_t)CH * DK * 4 + 2 * (size_t)QTW * CH * 2;         // 80 KB
        constexpr size_t LDS_W2 = LDS_W1 + (size_t)QTW * DK * 2;                                                // 112 KB + 16
        if (!once_w) { allow_lds(memread_apply_lpw_kernel<false>, LDS_W1); allow_lds(memread_apply_lpw_kernel<true>, LDS_W2); once_w = true; }
        const dim3 gridw(cdiv(d->HW, QTW) * d->nsplit, d->obj_n);
        // bf16x3 only: in plain bf16 the image's 8-byte hi halves sit 16 bytes apart and the kernel measured 11 % slower
        // than the f32 rows rounded in registers (9.9 vs 8.9 ms at 1.2M entries)
        // the kept split image: keys land in LDS as the operand image, value operands come straight from the 8-row-blocked
        // image (bank.hip) -- in plain bf16 only its hi plane is read (2 bytes per element instead of the f32 rows' 4)
        static int img_bf16 = 1;
        if (d->bank_k_lp && d->bank_v_lp && (d->precision == 2 || img_bf16)) {
            static bool once_s = false;
            if (!once_s) {
                allow_lds(memread_apply_shw_kernel<true>, LDS_W2);
                allow_lds(memread_apply_shw_kernel<false>, LDS_W1);
                const char* e = getenv("VFN_APPLY_IMG_BF16");
                if (e) img_bf16 = atoi(e);
                once_s = true;
            }
            // plain bf16 (round 6): the software-pipelined kernel; VFN_APPLY_PIPE=0 (read at every call: A/B in one process) restores
            // memread_apply_shw_kernel<false>
            static bool once_p = false;
            if (!once_p) { allow_lds(memread_apply_pipe_kernel, APPLY_PIPE_LDS); once_p = true; }
            const char* ep = getenv("VFN_APPLY_PIPE");
            const bool pipe = !(ep && atoi(ep) == 0);
            if (d->precision == 2) hipLaunchKernelGGL(memread_apply_shw_kernel<true>, gridw, dim3(512), LDS_W2, (hipStream_t)stream, *d);
            else if (img_bf16 && pipe) hipLaunchKernelGGL(memread_apply_pipe_kernel, gridw, dim3(512), APPLY_PIPE_LDS, (hipStream_t)stream, *d);
            else if (img_bf16) hipLaunchKernelGGL(memread_apply_shw_kernel<false>, gridw, dim3(512), LDS_W1, (hipStream_t)stream, *d);
            if (d->precision == 2 || img_bf16) return vfn_check_launch();
        }
        if (d->precision == 1) hipLaunchKernelGGL(memread_apply_lpw_kernel<false>, gridw, dim3(512), LDS_W1, (hipStream_t)stream, *d);
        else hipLaunchKernelGGL(memread_apply_lpw_kernel<true>, gridw, dim3(512), LDS_W2, (hipStream_t)stream, *d);
        return vfn_check_launch();
    }
}

extern "C" int vfn_memread_finish(const vfn_memread_desc* d, void* stream) {
    if (!d || !d->o_part || !d->out || !d->bank_len) return VFN_ERR_ARG;
    if (d->cnt && !d->info) return VFN_ERR_ARG;
    if (d->ld_out % 4 || d->ldqv % 4) return VFN_ERR_ARG;
    const dim3 grid(1024, d->obj_n);
    hipLaunchKernelGGL(memread_finish_kernel, grid, dim3(256), 0, (hipStream_t)stream, *d);
    return vfn_check_launch();
}

#ifdef VFN_CENSUS
extern "C" int vfn_debug_census(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(vfn_census_buf), sizeof(unsigned long long) * 4096 * 4) == hipSuccess ? 0 : 1;
}
#endif
