// Implicit-GEMM convolution on the gfx950 f32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces every nn.Conv2d on the reference hot path except the 7x7 stems
// (AFB_URR.py:20-30,96-111,114-127,191-202 and the torchvision bottlenecks behind
// AFB_URR.py:39-47,69-77; SURVEY.md A.5 lists all 110 launches per frame).
//
//   GEMM view   M = N*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin, K ordered (kh,kw,cin)
//   layout      activations NHWC fp32, weights [CoutPad][K] fp32 (packed at load time)
//   tile        BM x BN per workgroup, K walked 32 channels of one filter tap at a time
//   MFMA        A operand = pixels (rows), B operand = filters (cols): lane l of the
//               32x32 result holds filter column l&31 for 16 pixel rows, so one store
//               instruction writes 2 x 128 contiguous bytes of NHWC output.
//   LDS image   [rows][32 floats] per operand, 16-byte chunks XOR-swizzled with
//               (row>>1)&7 so the ds_read_b128 fragment reads are conflict-free.
//               One b128 read feeds four MFMAs: MFMA t of a k-group takes element t
//               of the float4, i.e. k = 8*kk + 4*h + t for lane half h -- A and B use
//               the same permutation of k, so the products pair correctly.
//   pipeline    register-staged global->LDS double buffer, one barrier per K tile.
//   epilogue    y = acc*scale[c] + shift[c] (+ residual) (ReLU): eval-mode BatchNorm or
//               bias; optional ReLU on the *input* as it is staged (ResBlock applies
//               ReLU before each conv, AFB_URR.py:24-25).
//
// Exact f32: the MFMA is a k-ordered fmaf chain, no reduced precision anywhere.
#include "common.h"
#include "conv_internal.h"
#include "../../include/vfn_hip.h"

namespace {

constexpr int BK = 32;         // floats per LDS image row (128 bytes)
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// x = hi + lo with hi = bf16(x), lo = bf16(x - hi): 16 significant bits in two bf16 (the "bf16x3" operands)
__device__ __forceinline__ void split_bf16(const f32x4& v, bf16x4& h, bf16x4& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = (__bf16)v[e];
        l[e] = (__bf16)(v[e] - (float)h[e]);
    }
}

// In-launch finish of a K-split tile (replaces the separate reduce launch): every slice workgroup stores its raw partial tile
// WRITE-THROUGH (sc0 sc1: the bytes leave the XCD's L2 with the store -- no release fence, whose L2 write-back made the
// round-1 form of this slower than the second launch), every storing wave drains its stores, and behind a workgroup barrier one
// lane takes a ticket on the tile's arrival counter; the workgroup that draws the last ticket loads all slices (sc0 sc1 loads:
// served past the non-coherent L1 / L2) and sums them IN SLICE ORDER -- bit-identical to the reduce launch whichever workgroup
// arrives last -- then applies the epilogue.  (MI355X_MICROARCH.md "Valid forms": sc1 stores + per-wave vmcnt(0) + workgroup
// barrier + agent-scope counter add; the last adder's workgroup loads with sc1 loads behind a barrier that lane joins.)
// `flag` is one int of the (now idle) dynamic LDS; counters are zero at rest.
constexpr int CP_WT = 17;                          // buffer cache policy: sc0 | sc1

__device__ __forceinline__ __amdgpu_buffer_rsrc_t partial_rsrc(const vfn_conv_desc& p) {
    return __builtin_amdgcn_make_buffer_rsrc(p.partial, 0, 0x7ffffff0, 0x00020000);
}

template <int BM, int BN, int NT>
__device__ __forceinline__ void splitk_finish(const vfn_conv_desc& p, int* flag, int tile, int m0, int n0, int n_tiles) {
    const int tid = threadIdx.x;
    const int ksplit = p.ksplit;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's write-through partial stores have left
    __syncthreads();
    if (tid == 0) {
        int* cnt = p.tile_counters + (tile - p.split_from);
        const int prev = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (prev == ksplit - 1);
        if (last) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // at rest again for the next launch
        *flag = last;
    }
    __syncthreads();
    if (!*flag) return;

    const __amdgpu_buffer_rsrc_t rp = partial_rsrc(p);
    const int m_start = (p.split_from / n_tiles) * BM;
    const int slab = (p.M - m_start) * p.Cout * (int)sizeof(float);      // bytes of one slice's slab
    constexpr int C4 = BN / 4;
    for (int i = tid; i < BM * C4; i += NT) {
        const int row = m0 + i / C4;
        const int col = n0 + (i % C4) * 4;
        if (row >= p.M || col >= p.Cout) continue;
        const int off = ((row - m_start) * p.Cout + col) * (int)sizeof(float);
        f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, off, 0, CP_WT));
        for (int sp = 1; sp < ksplit; ++sp) a += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, off + sp * slab, 0, CP_WT));
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + col);
        if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + col);
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = a[k] * sc[k] + sh[k];
        f32x4 mk = {1.f, 1.f, 1.f, 1.f};
        if (p.mask) mk = *reinterpret_cast<const f32x4*>(p.mask + (size_t)row * p.mask_ld + col);
        if (p.mask && !p.mask_after) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = mk[k] > 0.f ? v[k] : 0.f;
        }
        if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + (size_t)(p.res_mod > 0 ? row % p.res_mod : row) * p.res_ld + col);
        if (p.mask && p.mask_after) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = mk[k] > 0.f ? v[k] : 0.f;
        }
        if (p.relu_out) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        if (p.out) *reinterpret_cast<f32x4*>(p.out + (size_t)row * p.out_ld + col) = v;
        if (p.out_lp) vfn_store_lp4(p.out_lp, row, p.out_ld, col, v, p.out_lp_relu);
    }
}

// Wide epilogue shared by the register-staged and the LDS-DMA kernel: the accumulators (lane = filter column,
// registers = pixel rows) are transposed through the now idle staging LDS, so that every lane then owns 4 consecutive
// channels of one pixel: 16-byte residual loads and 16-byte stores (a quarter of the store instructions of the dword
// form -- the thin-K 1x1 layers are bound by exactly those).  One round per row of 32x32 accumulator tiles of the wave.
// Returns false when the shapes do not allow 16-byte accesses (the caller then runs the dword epilogue).
// Precondition: every wave has passed the barrier behind the last K tile (the staging buffers are dead).
template <int BM, int BN, int WM, int WN, int LDS_FLOATS = 2 * (BM + BN) * BK>
__device__ __forceinline__ bool wide_epilogue(const vfn_conv_desc& p, char* smem, f32x16 (&acc)[BM / WM / 32][BN / WN / 32],
                                              bool split_tile, int kz, int m0, int n0, int n_tiles,
                                              int tid_in = -1, bool active = true) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int ROWS = WM * 32;                                   // tile rows handled per round
    constexpr int PITCH = (ROWS * (BN + 4) <= LDS_FLOATS) ? BN + 4 : BN;
    static_assert(ROWS * PITCH <= LDS_FLOATS, "C tile does not fit the staging LDS");
    const bool wide = (p.Cout % 4 == 0) && (p.out_ld % 4 == 0) && (!p.res || p.res_ld % 4 == 0) && (!p.mask || p.mask_ld % 4 == 0);
    if (!wide) return false;
    // (tid_in / active: the in-workgroup split-K variant runs this with its K group 0 only; the other groups keep the
    // barriers company and touch nothing)
    const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    float* sC = reinterpret_cast<float*>(smem);
    constexpr int C4 = BN / 4;
    constexpr int RPP = NT / C4;                                    // rows per pass of the read-back
    const int c4 = tid % C4, rr0 = tid / C4;
    const int col = n0 + c4 * 4;
    const bool col_ok = col < p.Cout;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    float* part = nullptr;
    const bool wt = split_tile && p.tile_counters;                  // in-launch finish: the partial goes out write-through
    const __amdgpu_buffer_rsrc_t rp = partial_rsrc(p);
    if (split_tile) {
        const int m_start = (p.split_from / n_tiles) * BM;
        part = p.partial + ((long long)kz * (p.M - m_start) - m_start) * (long long)p.Cout;
    } else if (col_ok) {
        if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + col);
        if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + col);
    }
    // Round 5: the finished tiles' path of the forward pass is branch-free.  The residual taps of a round are raw buffer loads issued
    // BEFORE the accumulators go through LDS (they land behind the transpose instead of one exposed round trip per row, which is what
    // the per-row `if (row < M) ... v += res[...]` compiled to: load, s_waitcnt vmcnt(0), store, next row), rows past M / columns past
    // Cout get an out-of-range offset (loads return zeros, stores are dropped) and the row loop is unrolled.  Tensors of 2 GiB or
    // more, the backward passes' ReLU masks and the split-bf16 image keep the pointer form below (the mask's registers would cost
    // the 128 x 128 tile its second resident workgroup).
    constexpr int ITERS = (ROWS + RPP - 1) / RPP;
    const long long lim = 0x7fffff00LL;
    const bool fast = !split_tile && !p.out_lp && !p.mask && p.out && (long long)p.M * p.out_ld * 4 < lim &&
                      (!p.res || (long long)(p.res_mod > 0 ? p.res_mod : p.M) * p.res_ld * 4 < lim);
    if (fast) {
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((size_t)p.M * p.out_ld * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rr_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.in), 0,
            p.res ? (int)((size_t)(p.res_mod > 0 ? p.res_mod : p.M) * p.res_ld * 4) : 0, 0x00020000);
        const float floor_ = p.relu_out ? 0.f : -INFINITY;
#pragma unroll
        for (int h = 0; h < TM; ++h) {
            f32x4 rv[ITERS];
            auto row_of = [&](int it) {
                const int rr = rr0 + it * RPP;
                const int row = m0 + (rr >> 5) * (TM * 32) + h * 32 + (rr & 31);
                return (active && rr < ROWS && row < p.M && col_ok) ? row : -1;
            };
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {               // (an empty resource without a residual: zeros, no traffic)
                const int row = row_of(it);
                const int rrow = p.res_mod > 0 ? row % p.res_mod : row;
                rv[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr_, row >= 0 ? (rrow * p.res_ld + col) * 4 : 0x7ffffff0, 0, 0));
            }
            if (h > 0) __syncthreads();                             // the previous round has been read back
            if (active) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        sC[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * PITCH + (wn * TN + j) * 32 + li] = acc[h][j][r];
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int rr = rr0 + it * RPP;
                f32x4 v = *reinterpret_cast<const f32x4*>(sC + (rr < ROWS ? rr : 0) * PITCH + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = vfn_floor_nan(v[e] * sc[e] + sh[e] + rv[it][e], floor_);
                const int row = row_of(it);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), ro,
                                                       row >= 0 ? (row * p.out_ld + col) * 4 : 0x7ffffff0, 0, 0);
            }
        }
        return true;
    }
#pragma unroll
    for (int h = 0; h < TM; ++h) {
        if (h > 0) __syncthreads();                                 // the previous round has been read back
        if (active) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    sC[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * PITCH + (wn * TN + j) * 32 + li] = acc[h][j][r];
        }
        __syncthreads();
        if (!active) continue;
        for (int rr = rr0; rr < ROWS; rr += RPP) {
            const int row = m0 + (rr >> 5) * (TM * 32) + h * 32 + (rr & 31);
            if (row >= p.M || !col_ok) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(sC + rr * PITCH + c4 * 4);
            if (split_tile) {
                if (wt) {
                    const int off = (int)((part + (size_t)row * p.Cout + col) - p.partial) * (int)sizeof(float);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rp, off, 0, CP_WT);
                } else {
                    *reinterpret_cast<f32x4*>(part + (size_t)row * p.Cout + col) = v;
                }
                continue;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * sc[e] + sh[e];
            f32x4 mk = {1.f, 1.f, 1.f, 1.f};
            if (p.mask) mk = *reinterpret_cast<const f32x4*>(p.mask + (size_t)row * p.mask_ld + col);
            if (p.mask && !p.mask_after) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
            }
            if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + (size_t)(p.res_mod > 0 ? row % p.res_mod : row) * p.res_ld + col);
            if (p.mask && p.mask_after) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
            }
            if (p.relu_out) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (p.out) *reinterpret_cast<f32x4*>(p.out + (size_t)row * p.out_ld + col) = v;
            if (p.out_lp) vfn_store_lp4(p.out_lp, row, p.out_ld, col, v, p.out_lp_relu);
        }
    }
    return true;
}

#ifdef VFN_CENSUS
__device__ unsigned long long vfn_conv_census_buf[4096 * 8];
#define CV_MARK(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) vfn_conv_census_buf[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CV_MARK(k)
#endif
// WK > 1: split-K INSIDE the workgroup.  The workgroup holds WK copies of the WM x WN wave grid ("K groups"); group g
// stages and multiplies K tiles [g*kper, (g+1)*kper) of the same output tile in its own LDS buffers, the groups' partial
// accumulators are summed through LDS in group order (bit-reproducible) and group 0 runs the epilogue.  For the layers
// whose output has fewer 32x32 tiles than the chip has SIMDs (M = 1620 .. 6480 at 1/16 and 1/8 resolution): the same
// parallelism as split-K over workgroups, without the partial slabs in HBM and without the reduce launch.
// PD: prefetch distance of the register staging in K tiles.  A layer with about one workgroup per CU has
// nothing else resident to cover a global load, and a 32x64 tile computes a K tile in 0.4 us: with PD = 1 every K tile
// waits out its own load (measured 1.5 us per tile).  PD tiles are kept in flight in registers instead.
// TPB: K tiles per workgroup barrier (2 * TPB LDS buffers).  A 32x64 or 64x64 tile gives a wave 16 MFMAs
// (0.43 us) per K tile, and the LDS store -> barrier -> first fragment read chain behind every barrier costs about as
// much (profiles/r02_census_conv_small_layers.txt); with TPB = 2 the waves run two tiles between barriers.
template <int BM, int BN, int WM, int WN, int MODE, int WK, int PD = 1, int TPB = 1>
__device__ __forceinline__ void conv_igemm_body(const vfn_conv_desc& p) {
    static_assert(TPB == 1 || PD > TPB, "several tiles per barrier need the deep register prefetch");
    constexpr int NBUF = 2 * TPB;          // LDS ring: the tiles being read and the tiles being staged
    CV_MARK(0);
    constexpr int NT = WM * WN * 64;       // threads of one K group
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    // MODE 0: exact f32 (v_mfma_f32_32x32x2_f32).
    // MODE 1: operands rounded to bf16 (RNE) as they are staged, v_mfma_f32_32x32x16_bf16, f32 accumulate; a K tile
    //         is 64 channels, so that an LDS row is 128 bytes in every mode.
    // MODE 2: "bf16x3" -- every operand is split x = hi + lo (two bf16, 16 significant bits together) and the product
    //         is hi*hi + hi*lo + lo*hi on the bf16 matrix cores (lo*lo, relative 2^-16, is dropped); an LDS row holds
    //         the 32 hi values in its first 64 bytes and the 32 lo values in the second.
    constexpr bool BF = (MODE == 1);
    constexpr int BKT = BF ? 64 : 32;  // K elements per tile
    constexpr int CPR = BKT / 4;       // float4 chunks per operand row in global memory
    constexpr int AC = BM * CPR / NT;  // 16-byte global chunks of the A tile per thread
    constexpr int BC = BN * CPR / NT;
    static_assert(AC >= 1 && BC >= 1, "tile too small for the thread count");

    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const int grp = WK > 1 ? (int)threadIdx.x / NT : 0;  // K group (wave-uniform)
    char* smem = smem_all + (size_t)grp * (NBUF * (BM + BN) * BK * sizeof(float));
    float* sA = reinterpret_cast<float*>(smem);          // [NBUF][BM][32]
    float* sB = sA + NBUF * BM * BK;                     // [NBUF][BN][32]

    const int tid = WK > 1 ? (int)threadIdx.x % NT : (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    // tile id -> (m tile, n tile); n fastest so blocks sharing input rows are neighbours
    const int n_tiles = (p.Cout + BN - 1) / BN;
    // blocks [0, split_from) own whole tiles; the rest are (tile, K slice) pairs: the last, partial round
    // of tiles is cut along K so that every CU still gets an equal share (and layers with fewer tiles than
    // CUs are cut entirely: split_from = 0)
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    int tile = blockIdx.x, kz = 0;
    bool split_tile = false;
    {
        // Workgroups are dealt round-robin over the 8 XCDs (private L2 each): give every XCD one contiguous
        // run of tiles so that neighbours (same input rows, overlapping filter taps) share an L2.  Bijective
        // for any tile count; affects speed only.
        const int nfull = (ksplit > 1) ? p.split_from : (int)gridDim.x;
        if (tile < nfull && nfull >= 16) {
            const int q = nfull >> 3, r = nfull & 7;
            const int xcd = tile & 7, loc = tile >> 3;
            tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        }
    }
    if (ksplit > 1 && blockIdx.x >= (unsigned)p.split_from) {
        const int r = (int)blockIdx.x - p.split_from;
        tile = p.split_from + r / ksplit;
        kz = r - (r / ksplit) * ksplit;
        split_tile = true;
    }
    const int mt = tile / n_tiles, nt = tile % n_tiles;
    const int m0 = mt * BM, n0 = nt * BN;

    const int HoWo = p.Ho * p.Wo;
    const int cblks = p.Cin / BKT;
    const int Ktot = p.KH * p.KW * p.Cin;
    const int nk_all = p.KH * p.KW * cblks;
    // a split tile's slice kz owns K tiles [kt_begin, kt_begin + nk)
    const int kper = WK > 1 ? (nk_all + WK - 1) / WK : (split_tile ? (nk_all + ksplit - 1) / ksplit : nk_all);
    const int kt_begin = (WK > 1 ? grp : kz) * kper;
    const int nk = max(0, min(kper, nk_all - kt_begin));
    const int nk_loop = WK > 1 ? kper : nk;               // every K group runs the same number of barriers
    // K rotation (1x1 problems, vfn_conv_desc.k_rot): workgroup (m tile mt) walks its K tiles starting at tile mt % nk and wraps.
    // Every K tile of a pixel-major / filter-major operand is the SAME 128-byte column of 1-KB rows: workgroups that start
    // together would otherwise all read byte column 0 of their rows at the same time, i.e. hammer the same few memory channels.
    const int krot = (p.k_rot && p.KH == 1 && p.KW == 1 && nk > 1) ? (mt % nk) : 0;

    // per-thread staging coordinates
    const int c16 = tid % CPR;             // chunk column (4 floats)
    const int r0 = tid / CPR;              // first row this thread stages
    constexpr int RSTEP = NT / CPR;

    int a_base[AC], a_hi0[AC], a_wi0[AC];
#pragma unroll
    for (int j = 0; j < AC; ++j) {
        const int m = m0 + r0 + j * RSTEP;
        if (m < p.M) {
            const int n = m / HoWo;
            const int rem = m - n * HoWo;
            const int ho = rem / p.Wo;
            const int wo = rem - ho * p.Wo;
            a_base[j] = n * p.H * p.W;
            a_hi0[j] = ho * p.stride - p.pad;
            a_wi0[j] = wo * p.stride - p.pad;
        } else {
            a_base[j] = 0;
            a_hi0[j] = -100000;            // never in range -> zeros
            a_wi0[j] = 0;
        }
    }
    // Operands come in through raw buffer loads: the descriptor carries the tensor size, so a tap that falls
    // outside the image (or a row past M) is given an out-of-range offset and the hardware returns zeros --
    // no select, no branch, and the only per-tile VALU work is two range checks and one add per 16 bytes.
    const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in), 0, (int)((size_t)p.N * p.H * p.W * p.in_ld * sizeof(float)), 0x00020000);
    // w_packed (MODE >= 1): the filters were converted on the host (ops.pack_weights_lp) into the LDS row image,
    // 128 bytes per filter and K tile (bf16: 64 bf16; bf16x3: 32 hi then 32 lo) -- staged with no VALU work
    const bool wpk = (MODE != 0) && p.w_packed;
    // (batched filters, vfn_conv_desc.w_batch_rows: output rows of bank b multiply filter matrix b; f32 only)
    const int w_banks = p.w_batch_rows > 0 ? (p.M + p.w_batch_rows - 1) / p.w_batch_rows : 1;
    const int wb_rows = p.w_batch_rows > 0 ? (m0 / p.w_batch_rows) * p.cout_pad : 0;
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w), 0, wpk ? (int)((size_t)p.cout_pad * nk_all * 128) : (int)((size_t)w_banks * p.cout_pad * Ktot * sizeof(float)),
        0x00020000);
    constexpr int BCP = BN * 8 / NT;       // 16-byte chunks of a packed B tile per thread
    const int pc = tid & 7, pr0 = tid >> 3;
    int wp_off[BCP];
#pragma unroll
    for (int j = 0; j < BCP; ++j) wp_off[j] = ((n0 + pr0 + j * (NT / 8)) * nk_all) * 128 + pc * 16;
    int a_off[AC];                         // byte offset of tap (0,0), channel block 0, this lane's chunk
#pragma unroll
    for (int j = 0; j < AC; ++j)
        a_off[j] = ((a_base[j] + a_hi0[j] * p.W + a_wi0[j]) * p.in_ld + c16 * 4) * (int)sizeof(float);
    int w_off[BC];
#pragma unroll
    for (int j = 0; j < BC; ++j)
        w_off[j] = ((wb_rows + n0 + r0 + j * RSTEP) * Ktot + c16 * 4) * (int)sizeof(float);

    f32x4 ra[PD][AC], rb[PD][BC];
    int kh, kw, cb;                        // tap / channel block of the tile being *loaded*
    {
        const int tap = kt_begin / cblks;
        cb = kt_begin - tap * cblks;
        kh = tap / p.KW;
        kw = tap - kh * p.KW;
    }
    auto rot_tile = [&](int i) { const int a = i + krot; return kt_begin + (a >= nk ? a - nk : a); };   // i-th tile of the walk
    auto load_a = [&](int slot) {
        const int tap_off = ((kh * p.W + kw) * p.in_ld + cb * BKT) * (int)sizeof(float);     // wave-uniform
#pragma unroll
        for (int j = 0; j < AC; ++j) {
            const bool ok = (unsigned)(a_hi0[j] + kh) < (unsigned)p.H && (unsigned)(a_wi0[j] + kw) < (unsigned)p.W;
            const int off = ok ? a_off[j] + tap_off : 0x7ffffff0;
            ra[slot][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, off, 0, 0));
        }
    };
    auto load_b = [&](int kt, int slot) {
        const int k_off = kt * BKT * (int)sizeof(float);
        if (wpk) {
#pragma unroll
            for (int j = 0; j < BCP; ++j)
                rb[slot][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, wp_off[j] + kt * 128, 0, 0));
        } else {
#pragma unroll
            for (int j = 0; j < BC; ++j)
                rb[slot][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_off[j] + k_off, 0, 0));
        }
        if (++cb == cblks) { cb = 0; if (++kw == p.KW) { kw = 0; ++kh; } }
    };
    const float relu_floor = p.relu_in ? 0.f : -INFINITY;
    const bool a_lp = (MODE == 2) && p.in_lp;            // A rows arrive as the split-bf16 image (vfn_conv_desc.in_lp)
    auto store_a = [&](int buf, int slot) {
        float* dA = sA + buf * BM * BK;
#pragma unroll
        for (int j = 0; j < AC; ++j) {
            const int r = r0 + j * RSTEP;
            f32x4 v = ra[slot][j];
            if constexpr (MODE == 2) {
                if (a_lp) {                              // the tensor IS the operand image: 16 bytes in, 16 bytes out
                    *reinterpret_cast<f32x4*>(dA + r * BK + ((c16 ^ ((r >> 1) & 7)) << 2)) = v;
                    continue;
                }
            }
            v.x = fmaxf(v.x, relu_floor); v.y = fmaxf(v.y, relu_floor);
            v.z = fmaxf(v.z, relu_floor); v.w = fmaxf(v.w, relu_floor);
            if constexpr (MODE == 1) {
                const bf16x4 h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
                *reinterpret_cast<bf16x4*>(dA + r * BK + ((((c16 >> 1) ^ ((r >> 1) & 7)) << 2) | ((c16 & 1) << 1))) = h;
            } else if constexpr (MODE == 2) {
                bf16x4 h, l;
                split_bf16(v, h, l);
                const int sw = (r >> 1) & 7, half = (c16 & 1) << 1;
                *reinterpret_cast<bf16x4*>(dA + r * BK + ((((c16 >> 1) ^ sw) << 2) | half)) = h;
                *reinterpret_cast<bf16x4*>(dA + r * BK + ((((4 + (c16 >> 1)) ^ sw) << 2) | half)) = l;
            } else {
                *reinterpret_cast<f32x4*>(dA + r * BK + ((c16 ^ ((r >> 1) & 7)) << 2)) = v;
            }
        }
    };
    auto store_b = [&](int buf, int slot) {
        float* dB = sB + buf * BN * BK;
        if (wpk) {
#pragma unroll
            for (int j = 0; j < BCP; ++j) {
                const int r = pr0 + j * (NT / 8);
                *reinterpret_cast<f32x4*>(dB + r * BK + ((pc ^ ((r >> 1) & 7)) << 2)) = rb[slot][j];
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < BC; ++j) {
            const int r = r0 + j * RSTEP;
            if constexpr (MODE == 1) {
                const f32x4 v = rb[slot][j];
                const bf16x4 h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
                *reinterpret_cast<bf16x4*>(dB + r * BK + ((((c16 >> 1) ^ ((r >> 1) & 7)) << 2) | ((c16 & 1) << 1))) = h;
            } else if constexpr (MODE == 2) {
                bf16x4 h, l;
                split_bf16(rb[slot][j], h, l);
                const int sw = (r >> 1) & 7, half = (c16 & 1) << 1;
                *reinterpret_cast<bf16x4*>(dB + r * BK + ((((c16 >> 1) ^ sw) << 2) | half)) = h;
                *reinterpret_cast<bf16x4*>(dB + r * BK + ((((4 + (c16 >> 1)) ^ sw) << 2) | half)) = l;
            } else {
                *reinterpret_cast<f32x4*>(dB + r * BK + ((c16 ^ ((r >> 1) & 7)) << 2)) = rb[slot][j];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int d = 0; d < PD; ++d)
        if (d < nk) {
            if (krot) { kh = kw = 0; cb = rot_tile(d); load_a(d); load_b(cb, d); }
            else { load_a(d); load_b(kt_begin + d, d); }
        }
#pragma unroll
    for (int d = 0; d < TPB; ++d)
        if (d < nk) { store_a(d, d); store_b(d, d); }
    __syncthreads();
    CV_MARK(1);

    // (unrolled by PD so that the staging slots are compile-time registers: tile kt lives in slot kt % PD)
    for (int kt0 = 0; kt0 < nk_loop; kt0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
        const int kt = kt0 + u;
        if (kt >= nk_loop) break;
        const int buf = kt % NBUF;
        const int buf_st = (kt + TPB) % NBUF;          // where tile kt+TPB is staged during this tile
        const bool more = kt + TPB < nk;               // tile kt+TPB goes to LDS during this tile
        const bool more_load = kt + PD < nk;           // tile kt+PD is requested during this tile
        const bool sync = ((kt + 1) % TPB == 0) || kt + 1 >= nk_loop;
        const float* cA = sA + buf * BM * BK + (wm * TM * 32) * BK;
        const float* cB = sB + buf * BN * BK + (wn * TN * 32) * BK;
        if (WK > 1 && kt >= nk) { if (sync) __syncthreads(); continue; }     // (a K group with a shorter last slice)
        if constexpr (MODE == 2) {
            // two 16-channel steps per tile: hi chunk 2s+h and lo chunk 4+2s+h of every row; three MFMAs per tile pair
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                // (tile kt+PD is requested in the first step, tile kt+TPB goes to LDS in the second: as the f32 path.  In the
                // reduced-precision modes a K tile is a fraction of a microsecond of matrix time, so the K loop of a small
                // layer runs at the speed of its loads: PD tiles in flight, not one)
                if (st == 0) { if (more_load) { if (krot) { kh = kw = 0; cb = rot_tile(kt + PD); load_a(u); load_b(cb, u); } else { load_a(u); load_b(kt_begin + kt + PD, u); } } }
                else if (more) { store_a(buf_st, (u + TPB) % PD); store_b(buf_st, (u + TPB) % PD); }
                const int lc = 2 * st + lh;
                bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int r = i * 32 + li, sw = (r >> 1) & 7;
                    ah[i] = *reinterpret_cast<const bf16x8*>(cA + r * BK + ((lc ^ sw) << 2));
                    al[i] = *reinterpret_cast<const bf16x8*>(cA + r * BK + (((4 + lc) ^ sw) << 2));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int r = j * 32 + li, sw = (r >> 1) & 7;
                    bh[j] = *reinterpret_cast<const bf16x8*>(cB + r * BK + ((lc ^ sw) << 2));
                    bl[j] = *reinterpret_cast<const bf16x8*>(cB + r * BK + (((4 + lc) ^ sw) << 2));
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
        } else {
        // the staging of tile kt+1 is spread over the four k-groups of tile kt (both global loads up front, the LDS
        // writes in the second half), and the fragments of k-group kk+1 are read before the MFMAs of kk are issued,
        // so that neither an LDS round trip nor a late global load sits in front of the matrix pipe
        auto read_frags = [&](int kk, f32x4 (&a)[TM], f32x4 (&b)[TN]) {
            const int lc = 2 * kk + lh;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = i * 32 + li;      // (wm*TM*32) is a multiple of 32: swizzle bits unchanged
                a[i] = *reinterpret_cast<const f32x4*>(cA + r * BK + ((lc ^ ((r >> 1) & 7)) << 2));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = j * 32 + li;
                b[j] = *reinterpret_cast<const f32x4*>(cB + r * BK + ((lc ^ ((r >> 1) & 7)) << 2));
            }
        };
        f32x4 fa[2][TM], fb[2][TN];
        read_frags(0, fa[0], fb[0]);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk == 0 && more_load) {                                                   // slot u: tile kt is in LDS already
                if (krot) { kh = kw = 0; cb = rot_tile(kt + PD); load_a(u); load_b(cb, u); }
                else { load_a(u); load_b(kt_begin + kt + PD, u); }
            }
            if (more) {
                if (kk == 2) store_a(buf_st, (u + TPB) % PD);
                if (kk == 3) store_b(buf_st, (u + TPB) % PD);
            }
            if (kk + 1 < 4) read_frags(kk + 1, fa[(kk + 1) & 1], fb[(kk + 1) & 1]);
            f32x4 (&a)[TM] = fa[kk & 1];
            f32x4 (&b)[TN] = fb[kk & 1];
            if constexpr (BF) {
                // the 16 bytes are 8 bf16 = k 16*kk + 8*h .. +7: one 32x32x16 step per tile pair
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]),
                                                                            acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
            }
        }
        }
        if (sync) __syncthreads();
    }
    }
    CV_MARK(2);
#ifdef VFN_CENSUS
    struct CvEnd { __device__ ~CvEnd() { __builtin_amdgcn_s_waitcnt(0); CV_MARK(3); } } cv_end_;       // after the last store has left
#endif

    if constexpr (WK > 1) {
        // partial accumulators of groups 1.. -> LDS (one float per lane and register: conflict-free), summed by group 0
        float* red = reinterpret_cast<float*>(smem_all);
        if (grp > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        red[((((grp - 1) * (WM * WN) + wave) * (TM * TN) + i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int g = 1; g < WK; ++g)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            acc[i][j][r] += red[((((g - 1) * (WM * WN) + wave) * (TM * TN) + i * TN + j) * 16 + r) * 64 + lane];
        }
        __syncthreads();                                   // the sums are in registers: the LDS is free for the epilogue
        if (wide_epilogue<BM, BN, WM, WN>(p, smem_all, acc, false, 0, m0, n0, n_tiles, tid, grp == 0)) return;
        if (grp > 0) return;
    } else {
        if (wide_epilogue<BM, BN, WM, WN>(p, smem, acc, split_tile, kz, m0, n0, n_tiles)) {
            if (split_tile && p.tile_counters) splitk_finish<BM, BN, WM * WN * 64>(p, reinterpret_cast<int*>(smem), tile, m0, n0, n_tiles);
            return;
        }
    }

    // split-K: raw partial sums to the workspace slab of this split; vfn_conv_splitk_reduce finishes
    if (split_tile) {
        const int m_start = (p.split_from / n_tiles) * BM;          // first row covered by split tiles
        float* part = p.partial + ((long long)kz * (p.M - m_start) - m_start) * (long long)p.Cout;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + li;
            if (col >= p.Cout) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int rbase = m0 + (wm * TM + i) * 32 + 4 * lh;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (row < p.M) part[(size_t)row * p.Cout + col] = acc[i][j][r];
                }
            }
        }
        return;                                    // (split tiles always take the 16-byte epilogue above: the launcher checks)
    }

    // epilogue: lane holds filter column (lane&31) for rows (reg&3)+8*(reg>>2)+4*(lane>>5)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + li;
        if (col >= p.Cout) continue;
        const float sc = p.scale ? p.scale[col] : 1.f;
        const float sh = p.shift ? p.shift[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rbase = m0 + (wm * TM + i) * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < p.M) {
                    float v = acc[i][j][r] * sc + sh;
                    const bool live = !p.mask || p.mask[(size_t)row * p.mask_ld + col] > 0.f;
                    if (!p.mask_after && !live) v = 0.f;
                    if (p.res) v += p.res[(size_t)(p.res_mod > 0 ? row % p.res_mod : row) * p.res_ld + col];
                    if (p.mask_after && !live) v = 0.f;
                    if (p.relu_out) v = fmaxf(v, 0.f);
                    p.out[(size_t)row * p.out_ld + col] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int MODE = 0>
__global__ __launch_bounds__(WM * WN * 64)
void conv_igemm_kernel(const vfn_conv_desc p) { conv_igemm_body<BM, BN, WM, WN, MODE, 1>(p); }

// in-workgroup split-K: WK K groups of WM x WN waves; PD K tiles in flight, TPB K tiles per barrier
template <int BM, int BN, int WM, int WN, int WK, int PD, int TPB, int MODE = 0>
__global__ __launch_bounds__(WM * WN * WK * 64)
void conv_igemm_wk_kernel(const vfn_conv_desc p) { conv_igemm_body<BM, BN, WM, WN, MODE, WK, PD, TPB>(p); }

// 128 bytes of zeros: the LDS-DMA source of every filter tap that falls outside the image
__device__ __attribute__((aligned(128))) float vfn_zero_page[32];

// (device-only helper: the builtin has no host form, and the host pass must still be able to emit the stub)
__device__ __forceinline__ void lds_dma16(const float* src, float* lds_dst) {
    __builtin_amdgcn_global_load_lds(src, lds_dst, 16, 0, 0);
}
__device__ __forceinline__ int wave_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }


// Same GEMM, operands staged by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write, no
// select/ReLU VALU on the load path -- the ablation of the register-staged kernel shows that path costs
// 15-30 % of the MFMA rate.  One wave instruction writes 1 KB = 8 image rows x 128 B linearly, so the XOR
// swizzle is applied to the per-lane SOURCE address; out-of-image taps read vfn_zero_page; ReLU-on-input
// moves to the fragment registers.
template <int BM, int BN, int WM, int WN, int STAGES>
__global__ __launch_bounds__(WM * WN * 64)
void conv_igemm_dma_kernel(const vfn_conv_desc p) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int APW = BM / 8 / NW;       // A wave-instructions per wave per K tile
    constexpr int BPW = BN / 8 / NW;
    static_assert(APW >= 1 && BPW >= 1, "tile too small for the wave count");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sA = reinterpret_cast<float*>(smem);          // [STAGES][BM][32]
    float* sB = sA + STAGES * BM * BK;                   // [STAGES][BN][32]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = wave_uniform(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int lr = lane >> 3, pc = lane & 7;             // row within a wave instruction / physical chunk

    const int n_tiles = (p.Cout + BN - 1) / BN;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    int tile = blockIdx.x, kz = 0;
    bool split_tile = false;
    {
        // Workgroups are dealt round-robin over the 8 XCDs (private L2 each): give every XCD one contiguous
        // run of tiles so that neighbours (same input rows, overlapping filter taps) share an L2.  Bijective
        // for any tile count; affects speed only.
        const int nfull = (ksplit > 1) ? p.split_from : (int)gridDim.x;
        if (tile < nfull && nfull >= 16) {
            const int q = nfull >> 3, r = nfull & 7;
            const int xcd = tile & 7, loc = tile >> 3;
            tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        }
    }
    if (ksplit > 1 && blockIdx.x >= (unsigned)p.split_from) {
        const int r = (int)blockIdx.x - p.split_from;
        tile = p.split_from + r / ksplit;
        kz = r - (r / ksplit) * ksplit;
        split_tile = true;
    }
    const int mt = tile / n_tiles, nt = tile % n_tiles;
    const int m0 = mt * BM, n0 = nt * BN;

    const int HoWo = p.Ho * p.Wo;
    const int cblks = p.Cin / BK;
    const int Ktot = p.KH * p.KW * p.Cin;
    const int nk_all = p.KH * p.KW * cblks;
    const int kper = split_tile ? (nk_all + ksplit - 1) / ksplit : nk_all;
    const int kt_begin = kz * kper;
    const int nk = min(kper, nk_all - kt_begin);

    // per-lane source description of its A rows (one row per wave instruction)
    int a_base[APW], a_hi0[APW], a_wi0[APW], a_c[APW];
#pragma unroll
    for (int j = 0; j < APW; ++j) {
        const int r = (wave * APW + j) * 8 + lr;
        a_c[j] = (pc ^ ((r >> 1) & 7)) << 2;
        const int m = m0 + r;
        if (m < p.M) {
            const int n = m / HoWo;
            const int rem = m - n * HoWo;
            const int ho = rem / p.Wo;
            const int wo = rem - ho * p.Wo;
            a_base[j] = n * p.H * p.W;
            a_hi0[j] = ho * p.stride - p.pad;
            a_wi0[j] = wo * p.stride - p.pad;
        } else {
            a_base[j] = 0;
            a_hi0[j] = -100000;
            a_wi0[j] = 0;
        }
    }
    const float* wsrc[BPW];
#pragma unroll
    for (int j = 0; j < BPW; ++j) {
        const int r = (wave * BPW + j) * 8 + lr;
        wsrc[j] = p.w + (size_t)((p.w_batch_rows > 0 ? (m0 / p.w_batch_rows) * p.cout_pad : 0) + n0 + r) * Ktot + ((pc ^ ((r >> 1) & 7)) << 2);
    }

    int kh, kw, cb;
    {
        const int tap = kt_begin / cblks;
        cb = kt_begin - tap * cblks;
        kh = tap / p.KW;
        kw = tap - kh * p.KW;
    }
    auto issue_tile = [&](int kt, int buf) {
        float* dA = sA + buf * BM * BK + (wave * APW) * 8 * BK;
        float* dB = sB + buf * BN * BK + (wave * BPW) * 8 * BK;
#pragma unroll
        for (int j = 0; j < APW; ++j) {
            const int hi = a_hi0[j] + kh, wi = a_wi0[j] + kw;
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const float* src = ok ? p.in + (size_t)(a_base[j] + hi * p.W + wi) * p.in_ld + cb * BK + a_c[j]
                                  : vfn_zero_page + a_c[j];
            lds_dma16(src, dA + j * 8 * BK);
        }
#pragma unroll
        for (int j = 0; j < BPW; ++j)
            lds_dma16(wsrc[j] + (size_t)kt * BK, dB + j * 8 * BK);
        if (++cb == cblks) { cb = 0; if (++kw == p.KW) { kw = 0; ++kh; } }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float relu_floor = p.relu_in ? 0.f : -INFINITY;
    auto compute_tile = [&](int buf) {
        const float* cA = sA + buf * BM * BK + (wm * TM * 32) * BK;
        const float* cB = sB + buf * BN * BK + (wn * TN * 32) * BK;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int lc = 2 * kk + lh;
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = i * 32 + li;
                a[i] = *reinterpret_cast<const f32x4*>(cA + r * BK + ((lc ^ ((r >> 1) & 7)) << 2));
                a[i].x = fmaxf(a[i].x, relu_floor); a[i].y = fmaxf(a[i].y, relu_floor);
                a[i].z = fmaxf(a[i].z, relu_floor); a[i].w = fmaxf(a[i].w, relu_floor);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = j * 32 + li;
                b[j] = *reinterpret_cast<const f32x4*>(cB + r * BK + ((lc ^ ((r >> 1) & 7)) << 2));
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
        }
    };

    // Two-stage ring.  (A three-stage ring with counted vmcnt + raw s_barrier, i.e. the DMA of tile kt+2 in
    // flight across the barrier, was built and measured: 106 vs 124 TFLOP/s on the balanced 128x128 case --
    // it costs the second resident workgroup per CU and the loads were not the exposed latency.)
    if (nk > 0) issue_tile(kt_begin, 0);
    __syncthreads();                                   // (drains the LDS-DMA)
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) issue_tile(kt_begin + kt + 1, buf ^ 1);     // lands behind this tile's MFMAs
        compute_tile(buf);
        __syncthreads();                               // next tile landed; this buffer may be refilled
    }

    if (wide_epilogue<BM, BN, WM, WN>(p, smem, acc, split_tile, kz, m0, n0, n_tiles)) {
        if (split_tile && p.tile_counters) splitk_finish<BM, BN, WM * WN * 64>(p, reinterpret_cast<int*>(smem), tile, m0, n0, n_tiles);
        return;
    }

    if (split_tile) {
        const int m_start = (p.split_from / n_tiles) * BM;
        float* part = p.partial + ((long long)kz * (p.M - m_start) - m_start) * (long long)p.Cout;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + li;
            if (col >= p.Cout) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int rbase = m0 + (wm * TM + i) * 32 + 4 * lh;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (row < p.M) part[(size_t)row * p.Cout + col] = acc[i][j][r];
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + li;
        if (col >= p.Cout) continue;
        const float sc = p.scale ? p.scale[col] : 1.f;
        const float sh = p.shift ? p.shift[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rbase = m0 + (wm * TM + i) * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < p.M) {
                    float v = acc[i][j][r] * sc + sh;
                    const bool live = !p.mask || p.mask[(size_t)row * p.mask_ld + col] > 0.f;
                    if (!p.mask_after && !live) v = 0.f;
                    if (p.res) v += p.res[(size_t)(p.res_mod > 0 ? row % p.res_mod : row) * p.res_ld + col];
                    if (p.mask_after && !live) v = 0.f;
                    if (p.relu_out) v = fmaxf(v, 0.f);
                    p.out[(size_t)row * p.out_ld + col] = v;
                }
            }
        }
    }
}

// out = act((sum over splits, fixed order) * scale + shift + res)
__global__ void splitk_reduce_kernel(const vfn_conv_desc p, int m_start) {
    const int c4n = p.Cout / 4;
    const size_t rows = (size_t)(p.M - m_start);
    const size_t total = rows * c4n;
    const size_t slab = rows * p.Cout;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = i % c4n;
        const size_t lrow = i / c4n;
        const size_t row = lrow + m_start;
        f32x4 a = *reinterpret_cast<const f32x4*>(p.partial + lrow * p.Cout + c4 * 4);
        for (int sp = 1; sp < p.ksplit; ++sp)
            a += *reinterpret_cast<const f32x4*>(p.partial + sp * slab + lrow * p.Cout + c4 * 4);
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + c4 * 4);
        if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + c4 * 4);
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = a[k] * sc[k] + sh[k];
        f32x4 mk = {1.f, 1.f, 1.f, 1.f};
        if (p.mask) mk = *reinterpret_cast<const f32x4*>(p.mask + row * p.mask_ld + c4 * 4);
        if (p.mask && !p.mask_after) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = mk[k] > 0.f ? v[k] : 0.f;
        }
        if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + (p.res_mod > 0 ? row % p.res_mod : row) * p.res_ld + c4 * 4);
        if (p.mask && p.mask_after) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = mk[k] > 0.f ? v[k] : 0.f;
        }
        if (p.relu_out) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        if (p.out) *reinterpret_cast<f32x4*>(p.out + row * p.out_ld + c4 * 4) = v;
        if (p.out_lp) vfn_store_lp4(p.out_lp, row, p.out_ld, c4 * 4, v, p.out_lp_relu);
    }
}

template <int BM, int BN, int WM, int WN, int DMA = 0, int MODE = 0>     // DMA: 0 = register staged, 2 / 3 = LDS-DMA ring depth
int launch_cfg(const vfn_conv_desc& p, hipStream_t s) {
    constexpr int NT = WM * WN * 64;
    const size_t lds = (DMA == 3 ? 3 : 2) * (size_t)(BM + BN) * BK * sizeof(float);
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        if constexpr (DMA != 0)
            hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_dma_kernel<BM, BN, WM, WN, DMA>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        else
            hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WM, WN, MODE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int m_tiles = cdiv(p.M, BM);
    const int n_tiles = cdiv(p.Cout, BN);
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    const int tiles = m_tiles * n_tiles;
    if (ks > 1 && (p.split_from < 0 || p.split_from > tiles || p.split_from % n_tiles)) return VFN_ERR_ARG;
    const int grid = ks > 1 ? p.split_from + (tiles - p.split_from) * ks : tiles;
    if constexpr (DMA != 0) hipLaunchKernelGGL((conv_igemm_dma_kernel<BM, BN, WM, WN, DMA>), dim3(grid), dim3(NT), lds, s, p);
    else hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, MODE>), dim3(grid), dim3(NT), lds, s, p);
    if (ks > 1 && p.split_from < tiles && !p.tile_counters) {
        vfn_conv_splitk_reduce(p, (p.split_from / n_tiles) * BM, s);
    }
    return vfn_check_launch();
}

template <int BM, int BN, int WM, int WN, int WK, int PD = 3, int TPB = 1, int MODE = 0>
int launch_wk(const vfn_conv_desc& p, hipStream_t s) {
    constexpr int NT = WM * WN * WK * 64;
    constexpr size_t lds = (size_t)WK * 2 * TPB * (BM + BN) * BK * sizeof(float);
    static_assert((size_t)(WK - 1) * BM * BN * sizeof(float) <= lds, "reduce buffer does not fit the staging LDS");
    static_assert(lds <= 160 * 1024 && NT <= 1024, "workgroup too large");
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_wk_kernel<BM, BN, WM, WN, WK, PD, TPB, MODE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (p.ksplit > 1) return VFN_ERR_ARG;                  // one kind of split at a time
    const int tiles = cdiv(p.M, BM) * cdiv(p.Cout, BN);
    hipLaunchKernelGGL((conv_igemm_wk_kernel<BM, BN, WM, WN, WK, PD, TPB, MODE>), dim3(tiles), dim3(NT), lds, s, p);
    return vfn_check_launch();
}

}  // namespace

void vfn_conv_splitk_reduce(const vfn_conv_desc& p, int m_start, hipStream_t s) {
    const size_t total = (size_t)(p.M - m_start) * (p.Cout / 4);
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p, m_start);
}

constexpr int kCfgCount = VFN_DIRECT_CFG0 + VFN_DIRECT_CFGS;
extern "C" int vfn_conv_cfg_count(void) { return kCfgCount; }

// K groups per workgroup of a tile configuration (1 = none): configurations 26.. split K inside the workgroup
extern "C" int vfn_conv_cfg_wk(int cfg) {
    static const int wk[12] = {4, 2, 2, 4, 4, 2, 2, 2, 1, 1, 1, 1};
    if (cfg < 0 || cfg >= kCfgCount) return 0;
    if (cfg >= VFN_DIRECT_CFG0) { int w = 0; vfn_conv_direct_info(cfg - VFN_DIRECT_CFG0, nullptr, nullptr, &w); return w; }
    return cfg < 26 ? 1 : wk[cfg - 26];
}

// K tiles between two workgroup barriers (1 for all but configurations 32..37, which run 2 with a 4-tile register prefetch)
extern "C" int vfn_conv_cfg_tpb(int cfg) {
    if (cfg < 0 || cfg >= kCfgCount) return 0;
    return (cfg < 32 || cfg >= VFN_DIRECT_CFG0) ? 1 : 2;
}

extern "C" int vfn_conv_cfg_info(int cfg, int* bm, int* bn, int* wm, int* wn, int* dma) {
    // 8..10: same tiles as 0 / 0 / 2 with twice the waves (smaller per-wave tiles, 4 waves per SIMD at 2 blocks/CU)
    // 11..16: LDS-DMA staging variants of 8 / 10 / 3 / 7 / 6 / 2
    // 17..19: 256-filter-wide tiles (input tile read once for all 256 filters): 128x256 and 64x256, 8 waves
    // (must match the switch of vfn_conv2d_nhwc_f32 below)
    // 20..25 (f32 only): 32-row tiles for the 1/16-resolution layers (M = 1620: 51 x 32 rows instead of 26 x 64),
    // single-wave 32x32 tiles (most workgroups for the smallest layers), and 8-wave variants of 128x64 / 256x64
    // 26..37 (any arithmetic mode; ksplit must be 1): in-workgroup split-K (vfn_conv_cfg_wk K groups of the WM x WN waves) and,
    // from 32 on, two K tiles per barrier (vfn_conv_cfg_tpb)
    static const int t[38][5] = {{128, 128, 2, 2, 0}, {128, 64, 2, 2, 0}, {64, 128, 2, 2, 0}, {64, 64, 2, 2, 0}, {32, 64, 1, 2, 0},
                                 {64, 32, 2, 1, 0}, {128, 32, 4, 1, 0}, {256, 128, 4, 2, 0},
                                 {128, 128, 2, 4, 0}, {128, 128, 4, 2, 0}, {64, 128, 2, 4, 0},
                                 {128, 128, 2, 4, 2}, {64, 128, 2, 4, 2}, {64, 64, 2, 2, 2}, {256, 128, 4, 2, 2}, {128, 32, 4, 1, 2},
                                 {64, 128, 2, 2, 2},
                                 {128, 256, 2, 4, 0}, {128, 256, 2, 4, 2}, {64, 256, 2, 4, 0},
                                 {32, 128, 1, 4, 0}, {32, 32, 1, 1, 0}, {128, 64, 4, 2, 0}, {256, 64, 4, 2, 0}, {32, 64, 1, 1, 0},
                                 {64, 64, 1, 2, 0},
                                 {32, 64, 1, 2, 0}, {64, 64, 2, 2, 0}, {32, 64, 1, 2, 0}, {64, 64, 2, 2, 0}, {32, 32, 1, 1, 0},
                                 {32, 128, 1, 4, 0},
                                 {32, 64, 1, 2, 0}, {64, 64, 2, 2, 0}, {32, 64, 1, 2, 0}, {64, 64, 2, 2, 0}, {64, 128, 2, 4, 0},
                                 {32, 128, 1, 4, 0}};
    if (cfg >= VFN_DIRECT_CFG0 && cfg < kCfgCount) {
        // wave-autonomous kernels (conv_direct.hip): 4 waves; reported as wm = wn = 0, dma = 9
        if (wm) *wm = 0;
        if (wn) *wn = 0;
        if (dma) *dma = 9;
        return vfn_conv_direct_info(cfg - VFN_DIRECT_CFG0, bm, bn, nullptr);
    }
    if (cfg < 0 || cfg >= 38) return VFN_ERR_ARG;
    if (bm) *bm = t[cfg][0];
    if (bn) *bn = t[cfg][1];
    if (wm) *wm = t[cfg][2];
    if (wn) *wn = t[cfg][3];
    if (dma) *dma = t[cfg][4];
    return VFN_OK;
}

extern "C" int vfn_conv_cfg_kind(int cfg) {
    if (cfg < 0 || cfg >= kCfgCount) return -1;
    if (cfg < VFN_DIRECT_CFG0) return 0;
    return vfn_conv_direct_is_streamk(cfg - VFN_DIRECT_CFG0) ? 2 : 1;
}

extern "C" int vfn_conv_cfg_name(int cfg, char* buf, int n) {
    if (cfg < VFN_DIRECT_CFG0 || cfg >= kCfgCount) return VFN_ERR_ARG;
    return vfn_conv_direct_name(cfg - VFN_DIRECT_CFG0, buf, n);
}

extern "C" int vfn_conv_cfg_tile(int cfg, int* bm, int* bn) { return vfn_conv_cfg_info(cfg, bm, bn, nullptr, nullptr, nullptr); }

// configurations 26..37 (in-workgroup split-K / deep prefetch / two tiles per barrier) in any arithmetic mode
template <int MODE>
int launch_wk_cfg(const vfn_conv_desc& d, int cfg, hipStream_t s) {
    switch (cfg) {
        case 26: return launch_wk<32, 64, 1, 2, 4, 3, 1, MODE>(d, s);
        case 27: return launch_wk<64, 64, 2, 2, 2, 3, 1, MODE>(d, s);
        case 28: return launch_wk<32, 64, 1, 2, 2, 3, 1, MODE>(d, s);
        case 29: return launch_wk<64, 64, 2, 2, 4, 3, 1, MODE>(d, s);
        case 30: return launch_wk<32, 32, 1, 1, 4, 3, 1, MODE>(d, s);
        case 31: return launch_wk<32, 128, 1, 4, 2, 3, 1, MODE>(d, s);
        case 32: return launch_wk<32, 64, 1, 2, 2, 4, 2, MODE>(d, s);
        case 33: return launch_wk<64, 64, 2, 2, 2, 4, 2, MODE>(d, s);
        case 34: return launch_wk<32, 64, 1, 2, 1, 4, 2, MODE>(d, s);
        case 35: return launch_wk<64, 64, 2, 2, 1, 4, 2, MODE>(d, s);
        case 36: return launch_wk<64, 128, 2, 4, 1, 4, 2, MODE>(d, s);
        case 37: return launch_wk<32, 128, 1, 4, 1, 4, 2, MODE>(d, s);
    }
    return VFN_ERR_ARG;
}

extern "C" int vfn_conv2d_nhwc_f32(const vfn_conv_desc* d, int cfg, void* stream) {
    if (!d || !d->in || !d->w || !d->out || d->in_lp || d->out_lp) return VFN_ERR_ARG;
    if (d->Cin % BK != 0 || d->in_ld % 4 != 0 || d->M <= 0) return VFN_ERR_ARG;
    int bm, bn;
    if (vfn_conv_cfg_tile(cfg, &bm, &bn) != VFN_OK) return VFN_ERR_ARG;
    if (d->cout_pad < cdiv(d->Cout, bn) * bn) return VFN_ERR_ARG;
    if (d->w_batch_rows < 0 || (d->w_batch_rows > 0 && (d->w_batch_rows % bm || d->KH != 1 || d->KW != 1))) return VFN_ERR_ARG;
    if (d->ksplit > 1) {
        const int nk_all = d->KH * d->KW * (d->Cin / BK);
        if (!d->partial || d->Cout % 4 || d->out_ld % 4 || (d->res && d->res_ld % 4) || (d->mask && d->mask_ld % 4)) return VFN_ERR_ARG;
        if (d->tile_counters && d->Cout % bn) return VFN_ERR_ARG;      // in-launch finish works on whole filter tiles
        if (cdiv(nk_all, d->ksplit) * (d->ksplit - 1) >= nk_all) return VFN_ERR_ARG;   // every split non-empty
        if (d->tile_counters) {
            // the in-launch finish addresses the partial slabs through ONE buffer resource with 32-bit byte offsets
            // (off + slice * slab): refuse what would wrap instead of dropping stores / loading zeros silently
            const int nt = cdiv(d->Cout, bn);
            if (d->split_from < 0 || d->split_from % nt) return VFN_ERR_ARG;
            const long long m_start = (long long)(d->split_from / nt) * bm;
            if ((long long)d->ksplit * ((long long)d->M - m_start) * d->Cout * (long long)sizeof(float) >= 0x7fffff00LL) return VFN_ERR_ARG;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    switch (cfg) {
        case 0: return launch_cfg<128, 128, 2, 2>(*d, s);
        case 1: return launch_cfg<128, 64, 2, 2>(*d, s);
        case 2: return launch_cfg<64, 128, 2, 2>(*d, s);
        case 3: return launch_cfg<64, 64, 2, 2>(*d, s);
        case 4: return launch_cfg<32, 64, 1, 2>(*d, s);
        case 5: return launch_cfg<64, 32, 2, 1>(*d, s);
        case 6: return launch_cfg<128, 32, 4, 1>(*d, s);
        case 7: return launch_cfg<256, 128, 4, 2>(*d, s);
        case 8: return launch_cfg<128, 128, 2, 4>(*d, s);
        case 9: return launch_cfg<128, 128, 4, 2>(*d, s);
        case 10: return launch_cfg<64, 128, 2, 4>(*d, s);
        case 11: return launch_cfg<128, 128, 2, 4, 2>(*d, s);
        case 12: return launch_cfg<64, 128, 2, 4, 2>(*d, s);
        case 13: return launch_cfg<64, 64, 2, 2, 2>(*d, s);
        case 14: return launch_cfg<256, 128, 4, 2, 2>(*d, s);
        case 15: return launch_cfg<128, 32, 4, 1, 2>(*d, s);
        case 16: return launch_cfg<64, 128, 2, 2, 2>(*d, s);
        case 17: return launch_cfg<128, 256, 2, 4>(*d, s);
        case 18: return launch_cfg<128, 256, 2, 4, 2>(*d, s);
        case 19: return launch_cfg<64, 256, 2, 4>(*d, s);
        case 20: return launch_cfg<32, 128, 1, 4>(*d, s);
        case 21: return launch_cfg<32, 32, 1, 1>(*d, s);
        case 22: return launch_cfg<128, 64, 4, 2>(*d, s);
        case 23: return launch_cfg<256, 64, 4, 2>(*d, s);
        case 24: return launch_cfg<32, 64, 1, 1>(*d, s);
        case 25: return launch_cfg<64, 64, 1, 2>(*d, s);
        default:
            if (cfg >= 26 && cfg <= 37) return launch_wk_cfg<0>(*d, cfg, s);
            if (cfg >= VFN_DIRECT_CFG0 && cfg < kCfgCount) return vfn_conv_direct_launch(*d, cfg - VFN_DIRECT_CFG0, s);
    }
    return VFN_ERR_ARG;
}

// Same convolution with bf16 operands (rounded to nearest-even as they are staged; f32 accumulate, f32
// tensors in HBM): BASELINE configs C3 / C5.  Register-staged tile configurations only (LDS-DMA cannot convert).
extern "C" int vfn_conv2d_nhwc_bf16(const vfn_conv_desc* d, int cfg, void* stream) {
    if (!d || !d->in || !d->w || !d->out || d->in_lp || d->out_lp) return VFN_ERR_ARG;
    // (batched filters -- the transform-domain GEMMs of Winograd layers in this mode, round 5 -- take the f32 banks and convert them as staged)
    if (d->w_batch_rows < 0 || (d->w_batch_rows > 0 && (d->w_packed || d->KH != 1 || d->KW != 1))) return VFN_ERR_ARG;
    if (d->Cin % 64 != 0 || d->in_ld % 4 != 0 || d->M <= 0) return VFN_ERR_ARG;
    int bm, bn;
    if (vfn_conv_cfg_tile(cfg, &bm, &bn) != VFN_OK) return VFN_ERR_ARG;
    if (d->cout_pad < cdiv(d->Cout, bn) * bn) return VFN_ERR_ARG;
    if (d->w_batch_rows > 0 && d->w_batch_rows % bm) return VFN_ERR_ARG;
    if (d->tile_counters) return VFN_ERR_ARG;
    if (d->ksplit > 1) {
        const int nk_all = d->KH * d->KW * (d->Cin / 64);
        if (!d->partial || d->Cout % 4 || d->out_ld % 4 || (d->res && d->res_ld % 4)) return VFN_ERR_ARG;
        if (cdiv(nk_all, d->ksplit) * (d->ksplit - 1) >= nk_all) return VFN_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    switch (cfg) {
        case 0: return launch_cfg<128, 128, 2, 2, 0, 1>(*d, s);
        case 1: return launch_cfg<128, 64, 2, 2, 0, 1>(*d, s);
        case 2: return launch_cfg<64, 128, 2, 2, 0, 1>(*d, s);
        case 3: return launch_cfg<64, 64, 2, 2, 0, 1>(*d, s);
        case 4: return launch_cfg<32, 64, 1, 2, 0, 1>(*d, s);
        case 5: return launch_cfg<64, 32, 2, 1, 0, 1>(*d, s);
        case 6: return launch_cfg<128, 32, 4, 1, 0, 1>(*d, s);
        case 7: return launch_cfg<256, 128, 4, 2, 0, 1>(*d, s);
        case 8: return launch_cfg<128, 128, 2, 4, 0, 1>(*d, s);
        case 9: return launch_cfg<128, 128, 4, 2, 0, 1>(*d, s);
        case 10: return launch_cfg<64, 128, 2, 4, 0, 1>(*d, s);
        case 17: return launch_cfg<128, 256, 2, 4, 0, 1>(*d, s);
        case 19: return launch_cfg<64, 256, 2, 4, 0, 1>(*d, s);
        case 22: return launch_cfg<128, 64, 4, 2, 0, 1>(*d, s);
        case 23: return launch_cfg<256, 64, 4, 2, 0, 1>(*d, s);
        default: if (cfg >= 26 && cfg <= 37) return launch_wk_cfg<1>(*d, cfg, s);
    }
    return VFN_ERR_ARG;
}

// "bf16x3": operands split into hi + lo bf16 halves as they are staged (16 significant bits), three bf16 MFMAs per
// product (hi*hi + hi*lo + lo*hi), f32 accumulate: relative error ~2^-16 per product, against 2^-9 for plain bf16
// and 2^-24 for f32.  Same tile configurations and K tiling (32 channels) as the f32 kernel's register-staged ones.
extern "C" int vfn_conv2d_nhwc_bf16x3(const vfn_conv_desc* d, int cfg, void* stream) {
    if (!d || !d->in || !d->w || (!d->out && !d->out_lp)) return VFN_ERR_ARG;
    if (d->w_batch_rows < 0 || (d->w_batch_rows > 0 && (d->w_packed || d->in_lp || d->out_lp || d->KH != 1 || d->KW != 1))) return VFN_ERR_ARG;
    if (d->in_lp && (d->relu_in || d->in_ld % 32)) return VFN_ERR_ARG;            // ReLU belongs to the image's producer
    // the image is written by the 16-byte epilogue only (4 channels per lane): shapes that fall back to the dword form are refused
    if (d->out_lp && (d->Cout % 32 || d->out_ld % 32 || (d->res && d->res_ld % 4) || d->tile_counters)) return VFN_ERR_ARG;
    if (d->Cin % BK != 0 || d->in_ld % 4 != 0 || d->M <= 0) return VFN_ERR_ARG;
    int bm, bn;
    if (vfn_conv_cfg_tile(cfg, &bm, &bn) != VFN_OK) return VFN_ERR_ARG;
    if (d->cout_pad < cdiv(d->Cout, bn) * bn) return VFN_ERR_ARG;
    if (d->w_batch_rows > 0 && d->w_batch_rows % bm) return VFN_ERR_ARG;
    if (d->tile_counters) return VFN_ERR_ARG;
    if (d->ksplit > 1) {
        const int nk_all = d->KH * d->KW * (d->Cin / BK);
        if (!d->partial || d->Cout % 4 || d->out_ld % 4 || (d->res && d->res_ld % 4)) return VFN_ERR_ARG;
        if (cdiv(nk_all, d->ksplit) * (d->ksplit - 1) >= nk_all) return VFN_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    switch (cfg) {
        case 0: return launch_cfg<128, 128, 2, 2, 0, 2>(*d, s);
        case 1: return launch_cfg<128, 64, 2, 2, 0, 2>(*d, s);
        case 2: return launch_cfg<64, 128, 2, 2, 0, 2>(*d, s);
        case 3: return launch_cfg<64, 64, 2, 2, 0, 2>(*d, s);
        case 4: return launch_cfg<32, 64, 1, 2, 0, 2>(*d, s);
        case 5: return launch_cfg<64, 32, 2, 1, 0, 2>(*d, s);
        case 6: return launch_cfg<128, 32, 4, 1, 0, 2>(*d, s);
        case 7: return launch_cfg<256, 128, 4, 2, 0, 2>(*d, s);
        case 8: return launch_cfg<128, 128, 2, 4, 0, 2>(*d, s);
        case 9: return launch_cfg<128, 128, 4, 2, 0, 2>(*d, s);
        case 10: return launch_cfg<64, 128, 2, 4, 0, 2>(*d, s);
        case 17: return launch_cfg<128, 256, 2, 4, 0, 2>(*d, s);
        case 19: return launch_cfg<64, 256, 2, 4, 0, 2>(*d, s);
        case 22: return launch_cfg<128, 64, 4, 2, 0, 2>(*d, s);      // tall tiles for the 64-filter layers: fewer operand bytes per
        case 23: return launch_cfg<256, 64, 4, 2, 0, 2>(*d, s);      // FLOP from L2 (the reduced-precision K loop is L2 -> LDS bound)
        default: if (cfg >= 26 && cfg <= 37) return launch_wk_cfg<2>(*d, cfg, s);
    }
    return VFN_ERR_ARG;
}

#ifdef VFN_CENSUS
extern "C" int vfn_debug_conv_census(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(vfn_conv_census_buf), sizeof(unsigned long long) * 4096 * 8) == hipSuccess ? 0 : 1;
}
#endif
