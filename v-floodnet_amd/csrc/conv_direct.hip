// Implicit-GEMM convolution with WAVE-AUTONOMOUS operand fetch (round 4): no LDS staging, no barrier in the K loop.
//
// Why this exists.  The f32 matrix instruction is slow by matrix-core standards: v_mfma_f32_32x32x2_f32 occupies a SIMD for
// 64 cycles and consumes 2 x 64 operand floats, i.e. a wave that owns a 64 x 64 output tile needs 16 KB of operands per
// 256 MFMAs = 6.8 us -- 2.4 GB/s per wave, 2.4 TB/s chip-wide if no operand byte were ever shared, a small fraction of what
// the L1s / L2s deliver (MI355X_MICROARCH.md: L2 ~ 34 TB/s).  The LDS-tiled kernel of conv_igemm.hip shares operands
// between the waves of a workgroup, which the f32 pipe does not need, and pays for it with LDS stores, fragment reads and
// one workgroup barrier per K tile -- the K loop of its small tiles runs at half the matrix rate
// (profiles/r02_census_conv_small_layers.txt).  Here every wave loads the rows of its OWN output tile straight into the
// MFMA operand registers (lane l&31 = row, lane half = which 16 of the 32 channels of the K tile: 64 contiguous bytes per
// lane and row, raw buffer loads that return zeros outside the image), one K tile ahead of the MFMAs, and never waits for
// another wave until the epilogue.
//
//   workgroup   4 waves.  GM x GN wave tiles of (32 TM) x (32 TN) outputs, times WK groups along K (GM * GN * WK = 4); the
//               K groups of a tile add their accumulators through LDS in group order (bit-reproducible), as conv_igemm_wk_kernel.
//   k order     MFMA (jj, t) of a K tile multiplies channel 16 * half + 4 * jj + t of both operands: any bijection works as
//               long as A and B agree, and this one makes a lane's four 16-byte loads of a row contiguous.
//   split-K     vfn_conv_desc.ksplit / split_from as in conv_igemm.hip (partial slabs + vfn_conv_splitk_reduce).
//   epilogue    accumulators -> LDS (transposed) -> 16-byte residual loads / stores, all four waves on all GM x GN tiles.
//
// conv_streamk_kernel (below) is the same wave program under a WORK-CENTRIC schedule ("stream-K"): the launch is persistent (every
// wave resident at once), the (output tile, K tile) pairs of the layer are laid out in one line, tile-major, and wave w takes the
// w-th equal share of that line -- so every SIMD gets the same number of MFMAs whatever the tile count (M = 12 960 with 64 x 128
// tiles is 203 workgroups for 256 CUs in the tile-centric kernels).  A tile whose K range is cut between waves is finished inside
// the launch: every wave stores its raw partial tile write-through (sc0 sc1: the bytes leave the XCD's L2 with the store, no
// release fence), drains its stores and takes a ticket on the tile's counter; the wave that draws the last ticket loads the others'
// partials (sc0 sc1 loads: served past the non-coherent L1 / L2) and adds them IN K ORDER -- bit-reproducible whichever wave
// arrives last -- then runs the epilogue.  No wave ever waits for another one, so there is nothing to deadlock on.
// (MI355X_MICROARCH.md, "Valid forms": sc1 stores + per-wave vmcnt(0) + agent-scope counter add; the wave whose add came last
// loads with sc1 loads after its add has returned.)
//
// Same arithmetic as conv_igemm_kernel: an fmaf chain per output in a fixed k order (the order differs from the LDS-tiled
// kernel's only inside a 32-channel K tile).  Reference: every nn.Conv2d behind AFB_URR.py:20-30,96-127,191-202 and the
// torchvision bottlenecks behind AFB_URR.py:39-47,69-77.
#include <stdio.h>
#include "common.h"
#include "conv_internal.h"
#include "../../include/vfn_hip.h"

namespace {

// The wave program: acc = A[rows m0 .. m0 + 32 TM) x B[filters n0 .. n0 + 32 TN) over K tiles [kt_begin, kt_begin + nk) of the
// implicit GEMM (a K tile = 32 channels of one filter tap), operands straight into the MFMA registers.
template <int TM, int TN, bool RELU>
__device__ __forceinline__ void wave_gemm(const vfn_conv_desc& p, const __amdgpu_buffer_rsrc_t rsrc_in,
                                          const __amdgpu_buffer_rsrc_t rsrc_w, int m0, int n0, int kt_begin, int nk,
                                          int li, int lh, f32x16 (&acc)[TM][TN]) {
    const int HoWo = p.Ho * p.Wo;
    const int cblks = p.Cin / 32;
    const int Ktot = p.KH * p.KW * p.Cin;
    int a_off[TM], a_hi0[TM], a_wi0[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + i * 32 + li;
        if (m < p.M) {
            const int n = m / HoWo;
            const int rem = m - n * HoWo;
            const int ho = rem / p.Wo;
            const int wo = rem - ho * p.Wo;
            a_hi0[i] = ho * p.stride - p.pad;
            a_wi0[i] = wo * p.stride - p.pad;
            a_off[i] = ((n * p.H * p.W + a_hi0[i] * p.W + a_wi0[i]) * p.in_ld + lh * 16) * (int)sizeof(float);
        } else {
            a_hi0[i] = -100000;                  // never in range -> zeros
            a_wi0[i] = 0;
            a_off[i] = 0;
        }
    }
    int w_off[TN];
    const int wb_rows = p.w_batch_rows > 0 ? (m0 / p.w_batch_rows) * p.cout_pad : 0;       // filter bank of this tile (batched filters)
#pragma unroll
    for (int j = 0; j < TN; ++j) w_off[j] = ((wb_rows + n0 + j * 32 + li) * Ktot + lh * 16) * (int)sizeof(float);

    int kh, kw, cb;                              // tap / channel block of the K tile being loaded
    {
        const int tap = kt_begin / cblks;
        cb = kt_begin - tap * cblks;
        kh = tap / p.KW;
        kw = tap - kh * p.KW;
    }
    f32x4 ra[2][TM][4], rb[2][TN][4];
    auto load = [&](int slot, int kt) {
        const int tap_off = ((kh * p.W + kw) * p.in_ld + cb * 32) * (int)sizeof(float);      // wave-uniform
        int off[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const bool ok = (unsigned)(a_hi0[i] + kh) < (unsigned)p.H && (unsigned)(a_wi0[i] + kw) < (unsigned)p.W;
            off[i] = ok ? a_off[i] + tap_off : 0x7fffff00;
        }
        const int k_off = kt * 32 * (int)sizeof(float);
        // issued in the order the MFMAs consume them (jj outermost), so that the counted waits release the first k-group
        // as soon as its four-plus-four loads have landed
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                ra[slot][i][jj] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, off[i] + jj * 16, 0, 0));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                rb[slot][j][jj] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_off[j] + k_off + jj * 16, 0, 0));
        }
        if (++cb == cblks) { cb = 0; if (++kw == p.KW) { kw = 0; ++kh; } }
    };

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // RELU (ResBlock: ReLU in front of the convolution, AFB_URR.py:24-25): one integer max per element IN PLACE on the loaded
    // registers (the bits of a float >= +0 are a non-negative int, those of a negative float and of -0 a negative one), for the
    // whole K tile before its first MFMA.  Left to its own schedule hipcc sinks every max next to the MFMA that consumes it
    // and writes a scratch register which the MFMA in flight still names as its operand -- the max then waits for that MFMA
    // (measured: 8-18 % on the whole layer); the scheduling barrier keeps the block where it is written.  Layers without
    // ReLU on the input run the instantiation that has no vector instruction in the matrix stream at all.
    auto compute = [&](int slot) {
        if constexpr (RELU) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float f = ra[slot][i][jj][e];      // (scalar copy: __builtin_bit_cast on a vector element read element 0 for every e)
                        ra[slot][i][jj][e] = __int_as_float(max(__float_as_int(f), 0));
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[slot][i][jj][t], rb[slot][j][jj][t], acc[i][j], 0, 0, 0);
    };

    // One K tile of operands in flight behind the one being multiplied.  The steady-state body has no conditional load, so
    // that the compiler's counted s_waitcnt lets the younger tile's 16 loads stay in flight across the older tile's MFMAs.
    if (nk > 0) load(0, kt_begin);
    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {
        load(1, kt_begin + kt + 1);
        compute(0);
        load(0, kt_begin + kt + 2);
        compute(1);
    }
    if (kt + 1 < nk) {
        load(1, kt_begin + kt + 1);
        compute(0);
        compute(1);
    } else if (kt < nk) {
        compute(0);
    }

}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t conv_rsrc_in(const vfn_conv_desc& p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)((size_t)p.N * p.H * p.W * p.in_ld * sizeof(float)), 0x00020000);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t conv_rsrc_w(const vfn_conv_desc& p) {
    const int banks = p.w_batch_rows > 0 ? (p.M + p.w_batch_rows - 1) / p.w_batch_rows : 1;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)((size_t)banks * p.cout_pad * p.KH * p.KW * p.Cin * sizeof(float)), 0x00020000);
}

template <int TM, int TN, int GM, int GN, int WK, int OCC, bool RELU>
__global__ __launch_bounds__(256, OCC)
void conv_direct_kernel(const vfn_conv_desc p) {
    static_assert(GM * GN * WK == 4, "four waves per workgroup");
    constexpr int NTL = GM * GN;                 // wave tiles per workgroup
    constexpr int BM = 32 * TM * GM, BN = 32 * TN * GN;
    constexpr int PITCH = 32 * TN + 4;           // floats per row of a transposed tile in LDS
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gk = wave / NTL, tl = wave % NTL, gm = tl / GN, gn = tl % GN;
    const int li = lane & 31, lh = lane >> 5;

    const int n_tiles = (p.Cout + BN - 1) / BN;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    int tile = blockIdx.x, kz = 0;
    bool split_tile = false;
    {
        // one contiguous run of tiles per XCD (workgroups are dealt round-robin over the 8 XCDs; speed only, bijective)
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
    const int m0w = mt * BM, n0w = nt * BN;                   // workgroup tile
    const int m0 = m0w + gm * 32 * TM, n0 = n0w + gn * 32 * TN;   // this wave's tile

    const int HoWo = p.Ho * p.Wo;
    const int cblks = p.Cin / 32;
    const int Ktot = p.KH * p.KW * p.Cin;
    const int nk_all = p.KH * p.KW * cblks;
    // K tiles of this workgroup's slice, then of this wave's K group inside it
    const int kper_s = split_tile ? (nk_all + ksplit - 1) / ksplit : nk_all;
    const int s_begin = kz * kper_s;
    const int s_n = max(0, min(kper_s, nk_all - s_begin));
    const int kper_g = (s_n + WK - 1) / WK;
    const int kt_begin = s_begin + gk * kper_g;
    const int nk = max(0, min(kper_g, s_n - gk * kper_g));

    f32x16 acc[TM][TN];
    wave_gemm<TM, TN, RELU>(p, conv_rsrc_in(p), conv_rsrc_w(p), m0, n0, kt_begin, nk, li, lh, acc);

    // ---- K groups of one tile: partial accumulators of groups 1.. -> LDS (lane-major: conflict-free), summed by group 0
    if constexpr (WK > 1) {
        float* red = reinterpret_cast<float*>(smem);
        if (gk > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        red[((((gk - 1) * NTL + tl) * (TM * TN) + i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
        }
        __syncthreads();
        if (gk == 0) {
#pragma unroll
            for (int g = 1; g < WK; ++g)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            acc[i][j][r] += red[((((g - 1) * NTL + tl) * (TM * TN) + i * TN + j) * 16 + r) * 64 + lane];
        }
        __syncthreads();                           // the sums are in registers: the LDS is free for the transposed tiles
    }

    // ---- epilogue: one round per 32-row block of the wave tiles; the whole workgroup reads the rows back
    float* sC = reinterpret_cast<float*>(smem);    // [NTL][32][PITCH]
    const bool wide = (p.Cout % 4 == 0) && (p.out_ld % 4 == 0) && (!p.res || p.res_ld % 4 == 0) && (!p.mask || p.mask_ld % 4 == 0);
    constexpr int C4 = 8 * TN;                     // 16-byte chunks per tile row
    constexpr int RPP = 256 / C4;                  // rows per pass
    const int c4 = tid % C4, rr0 = tid / C4;
    float* part = nullptr;
    if (split_tile) {
        const int m_start = (p.split_from / n_tiles) * BM;
        part = p.partial + ((long long)kz * (p.M - m_start) - m_start) * (long long)p.Cout;
    }
#pragma unroll
    for (int h = 0; h < TM; ++h) {
        if (h > 0) __syncthreads();
        if (gk == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    sC[(tl * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * PITCH + j * 32 + li] = acc[h][j][r];
        }
        __syncthreads();
        for (int rr = rr0; rr < NTL * 32; rr += RPP) {
            const int t_ = rr >> 5;
            const int row = m0w + (t_ / GN) * (32 * TM) + h * 32 + (rr & 31);
            const int col = n0w + (t_ % GN) * (32 * TN) + c4 * 4;
            if (row >= p.M || col >= p.Cout) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(sC + rr * PITCH + c4 * 4);
            if (split_tile) {                      // (split tiles require the 16-byte form: checked by the launcher)
                *reinterpret_cast<f32x4*>(part + (size_t)row * p.Cout + col) = v;
                continue;
            }
            if (wide) {
                f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
                if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + col);
                if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + col);
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
                *reinterpret_cast<f32x4*>(p.out + (size_t)row * p.out_ld + col) = v;
            } else {
                for (int e = 0; e < 4; ++e) {
                    const int c = col + e;
                    if (c >= p.Cout) break;
                    float x = v[e] * (p.scale ? p.scale[c] : 1.f) + (p.shift ? p.shift[c] : 0.f);
                    const bool live = !p.mask || p.mask[(size_t)row * p.mask_ld + c] > 0.f;
                    if (!p.mask_after && !live) x = 0.f;
                    if (p.res) x += p.res[(size_t)(p.res_mod > 0 ? row % p.res_mod : row) * p.res_ld + c];
                    if (p.mask_after && !live) x = 0.f;
                    if (p.relu_out) x = fmaxf(x, 0.f);
                    p.out[(size_t)row * p.out_ld + c] = x;
                }
            }
        }
    }
}

template <int TM, int TN, int GM, int GN, int WK, int OCC = 2>
int launch_direct(const vfn_conv_desc& p, hipStream_t s) {
    constexpr int NTL = GM * GN;
    constexpr int BM = 32 * TM * GM, BN = 32 * TN * GN;
    constexpr size_t lds_red = (size_t)(4 - NTL) * TM * TN * 16 * 64 * sizeof(float);
    constexpr size_t lds_c = (size_t)NTL * 32 * (32 * TN + 4) * sizeof(float);
    constexpr size_t lds = lds_red > lds_c ? lds_red : lds_c;
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_direct_kernel<TM, TN, GM, GN, WK, OCC, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_direct_kernel<TM, TN, GM, GN, WK, OCC, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int m_tiles = cdiv(p.M, BM);
    const int n_tiles = cdiv(p.Cout, BN);
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    const int tiles = m_tiles * n_tiles;
    if (ks > 1 && (p.split_from < 0 || p.split_from > tiles || p.split_from % n_tiles || p.tile_counters)) return VFN_ERR_ARG;
    const int grid = ks > 1 ? p.split_from + (tiles - p.split_from) * ks : tiles;
    if (p.relu_in) hipLaunchKernelGGL((conv_direct_kernel<TM, TN, GM, GN, WK, OCC, true>), dim3(grid), dim3(256), lds, s, p);
    else hipLaunchKernelGGL((conv_direct_kernel<TM, TN, GM, GN, WK, OCC, false>), dim3(grid), dim3(256), lds, s, p);
    if (ks > 1 && p.split_from < tiles) vfn_conv_splitk_reduce(p, (p.split_from / n_tiles) * BM, s);
    return vfn_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------- stream-K
// Epilogue of ONE wave on its own tile, through a wave-private LDS block [32][32 TN + 4] (no workgroup barrier: the waves of a
// stream-K workgroup are at different places of different tiles; the LDS unit serves one wave's requests in order).
template <int TM, int TN>
__device__ __forceinline__ void wave_epilogue(const vfn_conv_desc& p, float* sC, const f32x16 (&acc)[TM][TN], int m0, int n0, int lane) {
    constexpr int PITCH = 32 * TN + 4;
    constexpr int C4 = 8 * TN;                     // 16-byte chunks per tile row
    constexpr int RPP = 64 / C4;                   // rows per pass
    const int li = lane & 31, lh = lane >> 5;
    const bool wide = (p.Cout % 4 == 0) && (p.out_ld % 4 == 0) && (!p.res || p.res_ld % 4 == 0) && (!p.mask || p.mask_ld % 4 == 0);
    const int c4 = lane % C4, rr0 = lane / C4;
    const int col = n0 + c4 * 4;
#pragma unroll
    for (int h = 0; h < TM; ++h) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sC[((r & 3) + 8 * (r >> 2) + 4 * lh) * PITCH + j * 32 + li] = acc[h][j][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int rr = rr0; rr < 32; rr += RPP) {
            const int row = m0 + h * 32 + rr;
            if (row >= p.M || col >= p.Cout) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(sC + rr * PITCH + c4 * 4);
            if (wide) {
                f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
                if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + col);
                if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + col);
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
                *reinterpret_cast<f32x4*>(p.out + (size_t)row * p.out_ld + col) = v;
            } else {
                for (int e = 0; e < 4; ++e) {
                    const int c = col + e;
                    if (c >= p.Cout) break;
                    float x = v[e] * (p.scale ? p.scale[c] : 1.f) + (p.shift ? p.shift[c] : 0.f);
                    const bool live = !p.mask || p.mask[(size_t)row * p.mask_ld + c] > 0.f;
                    if (!p.mask_after && !live) x = 0.f;
                    if (p.res) x += p.res[(size_t)(p.res_mod > 0 ? row % p.res_mod : row) * p.res_ld + c];
                    if (p.mask_after && !live) x = 0.f;
                    if (p.relu_out) x = fmaxf(x, 0.f);
                    p.out[(size_t)row * p.out_ld + c] = x;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the rows are in registers before the next round overwrites them
    }
}

// U units dealt to W waves: the first U % W waves take U / W + 1 units, the others U / W.  First unit of wave w:
__device__ __forceinline__ int sk_start(int w, int q, int r) { return w * q + min(w, r); }

template <int TM, int TN, int OCC, bool RELU>
__global__ __launch_bounds__(256, OCC)
void conv_streamk_kernel(const vfn_conv_desc p) {
    constexpr int PITCH = 32 * TN + 4;
    constexpr int SLOT = TM * TN * 1024;            // floats of one raw partial tile
    constexpr int CP_WT = 17;                       // cache policy sc0 sc1: write-through stores / loads served past L1 and L2
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    float* sC = reinterpret_cast<float*>(smem) + wave * (32 * PITCH);

    // workgroups are dealt round-robin over the 8 XCDs: consecutive waves of the unit line sit on one XCD (shared L2)
    int wg = blockIdx.x;
    {
        const int nwg = (int)gridDim.x;
        if (nwg >= 16) {
            const int q = nwg >> 3, r = nwg & 7;
            const int xcd = wg & 7, loc = wg >> 3;
            wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        }
    }
    const int W = (int)gridDim.x * 4;
    const int w = wg * 4 + wave;

    const int n_tiles = (p.Cout + 32 * TN - 1) / (32 * TN);
    const int m_tiles = (p.M + 32 * TM - 1) / (32 * TM);
    const int nk = p.KH * p.KW * (p.Cin / 32);
    const int U = m_tiles * n_tiles * nk;           // (< 2^31: checked by the launcher)
    const int uq = U / W, ur = U - uq * W;
    const int u_begin = sk_start(w, uq, ur), u_end = sk_start(w + 1, uq, ur);

    const __amdgpu_buffer_rsrc_t rsrc_in = conv_rsrc_in(p), rsrc_w = conv_rsrc_w(p);
    const __amdgpu_buffer_rsrc_t rsrc_part = __builtin_amdgcn_make_buffer_rsrc(p.partial, 0, 0x7ffffff0, 0x00020000);

    for (int u = u_begin; u < u_end;) {
        const int tile = u / nk;
        const int k0 = u - tile * nk;
        const int k1 = min(nk, k0 + (u_end - u));
        const int mt = tile / n_tiles, nt = tile - mt * n_tiles;
        const int m0 = mt * 32 * TM, n0 = nt * 32 * TN;
        f32x16 acc[TM][TN];
        wave_gemm<TM, TN, RELU>(p, rsrc_in, rsrc_w, m0, n0, k0, k1 - k0, li, lh, acc);
        u += k1 - k0;
        if (k0 == 0 && k1 == nk) {                  // the whole K range of the tile: no hand-off
            wave_epilogue<TM, TN>(p, sC, acc, m0, n0, lane);
            continue;
        }
        // ---- a partial tile.  Slot 0 of a wave holds the segment that contains its first unit, slot 1 the other one (a
        // wave's range meets at most two cut tiles: its first and its last)
        const int t_begin = tile * nk;
        const int my_slot = (u_begin >= t_begin) ? 0 : 1;
        {
            const int base = ((w * 2 + my_slot) * SLOT + lane * 4) * (int)sizeof(float);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rsrc_part,
                                                               base + ((i * TN + j) * 4 + q) * 1024, 0, CP_WT);
                    }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's partial has left
        // segments of the tile: the waves whose ranges meet [t_begin, t_begin + nk)
        const int big = ur * (uq + 1);               // units held by the waves that take uq + 1
        const int w_first = t_begin < big ? t_begin / (uq + 1) : ur + (t_begin - big) / max(uq, 1);
        int nseg = 0;
        for (int x = w_first; x < W && sk_start(x, uq, ur) < t_begin + nk; ++x) ++nseg;
        int last = 0;
        if (lane == 0) {
            const int prev = __hip_atomic_fetch_add(p.tile_counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (prev == nseg - 1);
            if (last) __hip_atomic_store(p.tile_counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // at rest again
        }
        last = __builtin_amdgcn_readfirstlane(last);
        if (!last) continue;
        // ---- the last arriver adds the segments in K order (its own from registers) and finishes the tile
        f32x16 tot[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
        for (int x = w_first; x < w_first + nseg; ++x) {
            if (x == w) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) tot[i][j][r] += acc[i][j][r];
                continue;
            }
            const int slot = (sk_start(x, uq, ur) >= t_begin) ? 0 : 1;
            const int base = ((x * 2 + slot) * SLOT + lane * 4) * (int)sizeof(float);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_part, base + ((i * TN + j) * 4 + q) * 1024, 0, CP_WT));
#pragma unroll
                        for (int e = 0; e < 4; ++e) tot[i][j][4 * q + e] += v[e];
                    }
        }
        wave_epilogue<TM, TN>(p, sC, tot, m0, n0, lane);
    }
}

template <int TM, int TN, int WPS>        // WPS: waves per SIMD (workgroups per CU) of the persistent launch
int launch_streamk(const vfn_conv_desc& p, hipStream_t s) {
    constexpr int OCC = WPS > 2 ? 2 : WPS;
    if (!p.partial || !p.tile_counters || p.ksplit > 1) return VFN_ERR_ARG;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipGetDevice(&dev);
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (cus <= 0) cus = 256;
    }
    const long long tiles = (long long)cdiv(p.M, 32 * TM) * cdiv(p.Cout, 32 * TN);
    const int grid = cus * WPS;
    if (tiles > VFN_SK_MAX_TILES || (long long)grid * 4 * 2 * TM * TN * 1024 > VFN_SK_WS_FLOATS) return VFN_ERR_ARG;
    if (tiles * p.KH * p.KW * (p.Cin / 32) >= (1LL << 30)) return VFN_ERR_ARG;
    constexpr size_t lds = (size_t)4 * 32 * (32 * TN + 4) * sizeof(float);
    if (p.relu_in) hipLaunchKernelGGL((conv_streamk_kernel<TM, TN, OCC, true>), dim3(grid), dim3(256), lds, s, p);
    else hipLaunchKernelGGL((conv_streamk_kernel<TM, TN, OCC, false>), dim3(grid), dim3(256), lds, s, p);
    return vfn_check_launch();
}

// configurations VFN_DIRECT_CFG0 ..: {TM, TN, GM, GN, WK}
constexpr int kDirect[VFN_DIRECT_CFGS][5] = {
    {2, 2, 2, 2, 1}, {2, 2, 4, 1, 1}, {2, 2, 2, 1, 2}, {2, 2, 1, 2, 2}, {2, 2, 1, 1, 4},
    {1, 2, 2, 2, 1}, {1, 2, 4, 1, 1}, {1, 2, 2, 1, 2}, {1, 2, 1, 1, 4},
    {2, 1, 2, 2, 1}, {2, 1, 4, 1, 1}, {1, 1, 2, 2, 1}, {1, 1, 4, 1, 1}, {1, 1, 1, 1, 4},
    {2, 4, 2, 1, 2}, {2, 4, 2, 2, 1}, {4, 2, 2, 2, 1}, {2, 4, 1, 1, 4},
    // stream-K (GM = 0): {TM, TN, 0, waves per SIMD, 1}
    {2, 2, 0, 1, 1}, {2, 2, 0, 2, 1}, {1, 2, 0, 2, 1}, {2, 1, 0, 2, 1}, {1, 1, 0, 2, 1}, {1, 2, 0, 1, 1},
};

}  // namespace

int vfn_conv_direct_info(int idx, int* bm, int* bn, int* wk) {
    if (idx < 0 || idx >= VFN_DIRECT_CFGS) return VFN_ERR_ARG;
    const bool sk = kDirect[idx][2] == 0;         // stream-K: the tile is one wave's
    if (bm) *bm = 32 * kDirect[idx][0] * (sk ? 1 : kDirect[idx][2]);
    if (bn) *bn = 32 * kDirect[idx][1] * (sk ? 1 : kDirect[idx][3]);
    if (wk) *wk = kDirect[idx][4];
    return VFN_OK;
}

int vfn_conv_direct_name(int idx, char* buf, int n) {
    if (idx < 0 || idx >= VFN_DIRECT_CFGS || !buf || n < 8) return VFN_ERR_ARG;
    const int* c = kDirect[idx];
    // (the last template argument, ReLU on the input, is left open: "...<2, 2, 2" matches both instantiations as a prefix)
    if (c[2] == 0) snprintf(buf, n, "conv_streamk_kernel<%d, %d, %d", c[0], c[1], c[3] > 2 ? 2 : c[3]);
    else snprintf(buf, n, "conv_direct_kernel<%d, %d, %d, %d, %d, %d", c[0], c[1], c[2], c[3], c[4], (c[0] * c[1] > 4) ? 1 : 2);
    return VFN_OK;
}

int vfn_conv_direct_is_streamk(int idx) { return idx >= 0 && idx < VFN_DIRECT_CFGS && kDirect[idx][2] == 0; }

int vfn_conv_direct_launch(const vfn_conv_desc& d, int idx, hipStream_t s) {
    switch (idx) {
        case 0: return launch_direct<2, 2, 2, 2, 1>(d, s);
        case 1: return launch_direct<2, 2, 4, 1, 1>(d, s);
        case 2: return launch_direct<2, 2, 2, 1, 2>(d, s);
        case 3: return launch_direct<2, 2, 1, 2, 2>(d, s);
        case 4: return launch_direct<2, 2, 1, 1, 4>(d, s);
        case 5: return launch_direct<1, 2, 2, 2, 1>(d, s);
        case 6: return launch_direct<1, 2, 4, 1, 1>(d, s);
        case 7: return launch_direct<1, 2, 2, 1, 2>(d, s);
        case 8: return launch_direct<1, 2, 1, 1, 4>(d, s);
        case 9: return launch_direct<2, 1, 2, 2, 1>(d, s);
        case 10: return launch_direct<2, 1, 4, 1, 1>(d, s);
        case 11: return launch_direct<1, 1, 2, 2, 1>(d, s);
        case 12: return launch_direct<1, 1, 4, 1, 1>(d, s);
        case 13: return launch_direct<1, 1, 1, 1, 4>(d, s);
        case 14: return launch_direct<2, 4, 2, 1, 2, 1>(d, s);
        case 15: return launch_direct<2, 4, 2, 2, 1, 1>(d, s);
        case 16: return launch_direct<4, 2, 2, 2, 1, 1>(d, s);
        case 17: return launch_direct<2, 4, 1, 1, 4, 1>(d, s);
        case 18: return launch_streamk<2, 2, 1>(d, s);
        case 19: return launch_streamk<2, 2, 2>(d, s);
        case 20: return launch_streamk<1, 2, 2>(d, s);
        case 21: return launch_streamk<2, 1, 2>(d, s);
        case 22: return launch_streamk<1, 1, 2>(d, s);
        case 23: return launch_streamk<1, 2, 1>(d, s);
    }
    return VFN_ERR_ARG;
}
