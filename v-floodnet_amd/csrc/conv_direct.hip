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
// Same arithmetic as conv_igemm_kernel: an fmaf chain per output in a fixed k order (the order differs from the LDS-tiled
// kernel's only inside a 32-channel K tile).  Reference: every nn.Conv2d behind AFB_URR.py:20-30,96-127,191-202 and the
// torchvision bottlenecks behind AFB_URR.py:39-47,69-77.
#include "common.h"
#include "conv_internal.h"
#include "../../include/vfn_hip.h"

namespace {

template <int TM, int TN, int GM, int GN, int WK, int OCC>
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
#pragma unroll
    for (int j = 0; j < TN; ++j) w_off[j] = ((n0 + j * 32 + li) * Ktot + lh * 16) * (int)sizeof(float);

    const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in), 0, (int)((size_t)p.N * p.H * p.W * p.in_ld * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w), 0, (int)((size_t)p.cout_pad * Ktot * sizeof(float)), 0x00020000);

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

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ReLU on the staged input as ONE integer max per element: the bits of a float >= +0 are a non-negative int, those of
    // a negative float (and -0) a negative one; floor INT_MIN = no ReLU
    const int relu_floor = p.relu_in ? 0 : (int)0x80000000;
    auto compute = [&](int slot) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                f32x4& v = ra[slot][i][jj];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float f = v[e];      // (a scalar copy first: __builtin_bit_cast on the vector element read element 0 for every e)
                    v[e] = __int_as_float(max(__float_as_int(f), relu_floor));
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[slot][i][jj][t], rb[slot][j][jj][t], acc[i][j], 0, 0, 0);
        }
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
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_direct_kernel<TM, TN, GM, GN, WK, OCC>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int m_tiles = cdiv(p.M, BM);
    const int n_tiles = cdiv(p.Cout, BN);
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    const int tiles = m_tiles * n_tiles;
    if (ks > 1 && (p.split_from < 0 || p.split_from > tiles || p.split_from % n_tiles || p.tile_counters)) return VFN_ERR_ARG;
    const int grid = ks > 1 ? p.split_from + (tiles - p.split_from) * ks : tiles;
    hipLaunchKernelGGL((conv_direct_kernel<TM, TN, GM, GN, WK, OCC>), dim3(grid), dim3(256), lds, s, p);
    if (ks > 1 && p.split_from < tiles) vfn_conv_splitk_reduce(p, (p.split_from / n_tiles) * BM, s);
    return vfn_check_launch();
}

// configurations VFN_DIRECT_CFG0 ..: {TM, TN, GM, GN, WK}
constexpr int kDirect[VFN_DIRECT_CFGS][5] = {
    {2, 2, 2, 2, 1}, {2, 2, 4, 1, 1}, {2, 2, 2, 1, 2}, {2, 2, 1, 2, 2}, {2, 2, 1, 1, 4},
    {1, 2, 2, 2, 1}, {1, 2, 4, 1, 1}, {1, 2, 2, 1, 2}, {1, 2, 1, 1, 4},
    {2, 1, 2, 2, 1}, {2, 1, 4, 1, 1}, {1, 1, 2, 2, 1}, {1, 1, 4, 1, 1}, {1, 1, 1, 1, 4},
    {2, 4, 2, 1, 2}, {2, 4, 2, 2, 1}, {4, 2, 2, 2, 1}, {2, 4, 1, 1, 4},
};

}  // namespace

int vfn_conv_direct_info(int idx, int* bm, int* bn, int* wk) {
    if (idx < 0 || idx >= VFN_DIRECT_CFGS) return VFN_ERR_ARG;
    if (bm) *bm = 32 * kDirect[idx][0] * kDirect[idx][2];
    if (bn) *bn = 32 * kDirect[idx][1] * kDirect[idx][3];
    if (wk) *wk = kDirect[idx][4];
    return VFN_OK;
}

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
    }
    return VFN_ERR_ARG;
}
