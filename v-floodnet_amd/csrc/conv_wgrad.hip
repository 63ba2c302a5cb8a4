// Weight gradient of a convolution as an IMPLICIT GEMM over the pixels (round 4; train_video_seg.py:73 loss.backward()):
//
//     dW[co][kh][kw][ci] = rowscale[co] * sum_m  gy[m][co] * act(x)[pix(m) + (kh, kw)][ci]          (m = output pixel)
//
// The reduction index is the pixel, and both operands are pixel-major in HBM (NHWC): round 3 materialised dY^T and the
// transposed im2col image of every layer input (vfn_transpose_taps_f32: 10 % of the step's device time, 728 launches per
// step) to feed the forward kernel.  Here nothing is transposed: in the f32 MFMA a lane supplies ONE operand element per
// instruction -- lane (l & 31, l >> 5) holds row l & 31 at k = l >> 5 -- so with lane = channel and k = pixel a wave's operand
// load is `buffer_load_dword` over 32 consecutive channels of one pixel (128 contiguous bytes per half wave): the NHWC tensors
// ARE the operand layout.  As in conv_direct.hip there is no LDS staging and no barrier in the loop; a wave owns a
// (32 TM output channels) x (32 TN input channels of one filter tap) tile of dW over a slice of the pixels, 32 pixels of
// operands in flight behind the 32 being multiplied, out-of-image taps and pixels past the slice load zeros (raw buffer loads
// with an out-of-range offset).
//
//   workgroup   4 waves = 4 consecutive pixel slices of one dW tile, summed through LDS in slice order
//   grid        dW tiles x ksplit; ksplit > 1: partial tiles to `partial` [ksplit][Cout][k*k*Cin], wgrad_reduce_kernel adds them
//               in slice order (deterministic), applies rowscale and (accumulate) adds the running gradient
//   rowscale    the frozen BatchNorm scale behind the convolution (train_video_seg.py:103-106): applied to the sum instead of to
//               every gy element
//   ReLU        act = max(., 0) on x for the decoder's pre-activation ResBlocks (AFB_URR.py:24-25)
#include "common.h"
#include "../../include/vfn_hip.h"

namespace {

template <int TM, int TN, bool RELU>
__global__ __launch_bounds__(256, 2)
void conv_wgrad_kernel(const vfn_wgrad_desc p) {
    constexpr int CH = 8;                        // k-steps (pixel pairs) per chunk (16: the loads' 64 offsets spill)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    const int kk = p.k * p.k;
    const int ci_tiles = (p.Cin + 32 * TN - 1) / (32 * TN);
    const int col_tiles = kk * ci_tiles;
    const int tiles = ((p.Cout + 32 * TM - 1) / (32 * TM)) * col_tiles;
    const int tile = blockIdx.x % tiles, kz = blockIdx.x / tiles;
    const int bz = blockIdx.y;                   // batch of independent problems (the 36 components of a Winograd-domain gradient)
    const int ct = tile % col_tiles, rt = tile / col_tiles;
    const int tap = ct / ci_tiles, cit = ct - tap * ci_tiles;
    const int kh = tap / p.k, kw = tap - kh * p.k;
    const int co0 = rt * 32 * TM, ci0 = cit * 32 * TN;

    // pixel slice of this wave: slices of equal, even length
    const int M = p.N * p.Ho * p.Wo;
    const int S = p.ksplit * 4;
    int per = (M + S - 1) / S;
    per += per & 1;
    const int m_begin = (kz * 4 + wave) * per;
    const int m_end = min(M, m_begin + per);

    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.gy) + (size_t)bz * p.g_bstride, 0, (int)((size_t)M * p.ld_g * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x) + (size_t)bz * p.x_bstride, 0, (int)((size_t)p.N * p.H * p.W * p.ld_x * sizeof(float)), 0x00020000);
    constexpr int OOB = 0x7fffff00;

    // this lane's pixel: m = m_begin + 2 * step + lh, tracked incrementally (no multiply or divide in the loop): output
    // coordinates (ho, wo) for the wrap, input coordinates (hi, wi) of the tap for the range check, byte offsets into gy / x
    int m = m_begin + lh;
    int ho, wo, hi, wi, g_off, x_off;
    {
        const int HoWo = p.Ho * p.Wo;
        const int mm = min(m, M - 1);
        const int n = mm / HoWo;
        const int rem = mm - n * HoWo;
        ho = rem / p.Wo;
        wo = rem - ho * p.Wo;
        hi = ho * p.stride - p.pad + kh;
        wi = wo * p.stride - p.pad + kw;
        g_off = (mm * p.ld_g + co0 + li) * (int)sizeof(float);
        x_off = (((n * p.H + hi) * p.W + wi) * p.ld_x + ci0 + li) * (int)sizeof(float);
    }
    // per step (two pixels on): uniform increments; a row / image wrap adds the rest of the way
    const int px = p.ld_x * (int)sizeof(float);
    const int x_step = 2 * p.stride * px;
    const int x_wrap_row = (p.stride * p.W - p.Wo * p.stride) * px;               // column wo - Wo of the next output row
    const int x_wrap_img = (p.H - p.Ho * p.stride) * p.W * px;                    // row 0 of the next image
    const int g_step = 2 * p.ld_g * (int)sizeof(float);
    int a_ok[TM], b_ok[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_ok[i] = co0 + i * 32 + li < p.Cout;
#pragma unroll
    for (int j = 0; j < TN; ++j) b_ok[j] = ci0 + j * 32 + li < p.Cin;

    float ga[2][TM][CH], xb[2][TN][CH];
    auto load = [&](int slot) {
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            // (bitwise, not short-circuit: hipcc turned `a && b ? load(x) : load(OOB)` into exec-masked branches with a
            // full wait inside; offsets are selected first, the loads are unconditional)
            const int in_slice = m < m_end;
            const int in_img = in_slice & ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);
            int og[TM], ox[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) og[i] = (in_slice & a_ok[i]) ? g_off + i * 128 : OOB;
#pragma unroll
            for (int j = 0; j < TN; ++j) ox[j] = (in_img & b_ok[j]) ? x_off + j * 128 : OOB;
#pragma unroll
            for (int i = 0; i < TM; ++i)
                ga[slot][i][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, og[i], 0, 0));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                xb[slot][j][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, ox[j], 0, 0));
            m += 2;
            g_off += g_step;
            wo += 2;
            wi += 2 * p.stride;
            x_off += x_step;
            const int wr = -(int)(wo >= p.Wo);                                 // all ones on a row wrap (Wo >= 2: launcher)
            wo -= wr & p.Wo;
            wi -= wr & (p.Wo * p.stride);
            ho -= wr;
            hi += wr & p.stride;
            x_off += wr & x_wrap_row;
            const int wi2 = -(int)(ho >= p.Ho);                                // ... on an image wrap
            ho -= wi2 & p.Ho;
            hi -= wi2 & (p.Ho * p.stride);
            x_off += wi2 & x_wrap_img;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int slot) {
        if constexpr (RELU) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int s = 0; s < CH; ++s) xb[slot][j][s] = fmaxf(xb[slot][j][s], 0.f);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < CH; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[slot][i][s], xb[slot][j][s], acc[i][j], 0, 0, 0);
    };

    const int nch = (max(0, m_end - m_begin) + 2 * CH - 1) / (2 * CH);
    if (nch > 0) load(0);
    int c = 0;
    for (; c + 2 < nch; c += 2) {
        load(1);
        compute(0);
        load(0);
        compute(1);
    }
    if (c + 1 < nch) {
        load(1);
        compute(0);
        compute(1);
    } else if (c < nch) {
        compute(0);
    }

    // ---- the four pixel slices of the workgroup: slices 1.. -> LDS (lane-major), summed by wave 0 in slice order
    float* red = reinterpret_cast<float*>(smem);
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    red[(((wave - 1) * (TM * TN) + i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int g = 1; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[i][j][r] += red[(((g - 1) * (TM * TN) + i * TN + j) * 16 + r) * 64 + lane];

    // ---- store: a result register is 32 consecutive input channels (lane l & 31) of one output channel (row); buffer
    // stores with 32-bit offsets (64 flat addresses at once spilled registers)
    const int Kc = kk * p.Cin;
    const bool direct = p.ksplit <= 1;
    const bool inlaunch = !direct && p.tile_counters != nullptr;
    const int slab_bytes = p.Cout * Kc * (int)sizeof(float);
    const size_t slab = (size_t)p.Cout * Kc;
    float* const dw = p.dw + (size_t)bz * slab;
    float* const partial = p.partial + (size_t)bz * p.ksplit * slab;
    int* const counters = p.tile_counters ? p.tile_counters + (size_t)bz * tiles : nullptr;
    float* dst = direct ? dw : partial + (size_t)kz * slab;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, slab_bytes, 0x00020000);
    constexpr int WT = 17;                       // sc0 sc1: past L1 and the XCD's L2 (the slices of a tile run on any XCD)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ci = ci0 + j * 32 + li;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int off = (ci < p.Cin && co < p.Cout) ? (co * Kc + tap * p.Cin + ci) * (int)sizeof(float) : OOB;
                float v = acc[i][j][r];
                if (direct) {
                    if (p.rowscale) v *= p.rowscale[min(co, p.Cout - 1)];
                    if (p.accumulate) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, off, 0, 0));
                }
                if (inlaunch) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rd, off, 0, WT);
                else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rd, off, 0, 0);
            }
    }
    if (!inlaunch) return;
    // ---- the slices of this tile meet here (round 4): the partial tile went out write-through; once it has left the wave, one
    // lane draws a ticket, and the wave that draws the last one adds the slices IN SLICE ORDER -- the sums of wgrad_reduce_kernel
    // in the same order, without the second launch.  The counter is zero at rest.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int last = 0;
    if (lane == 0) {
        const int prev = __hip_atomic_fetch_add(counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = prev == p.ksplit - 1;
        if (last) __hip_atomic_store(counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (!last) return;
    // (the range check covers the per-lane offset only: masked lanes carry OOB >= this size; the slice offset rides in the scalar operand)
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(partial, 0, p.ksplit * slab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(dw, 0, slab_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ci = ci0 + j * 32 + li;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            int off[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                off[r] = (ci < p.Cin && co < p.Cout) ? (co * Kc + tap * p.Cin + ci) * (int)sizeof(float) : OOB;
                acc[i][j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, off[r], 0, WT));
            }
            for (int sl = 1; sl < p.ksplit; ++sl) {
                float t[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, off[r], sl * slab_bytes, WT));
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += t[r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[i][j][r];
                if (p.rowscale) v *= p.rowscale[min(co, p.Cout - 1)];
                if (p.accumulate) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, off[r], 0, 0));
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rw, off[r], 0, 0);
            }
        }
    }
}

// dw[co][c] = (accumulate ? dw[co][c] : 0) + rowscale[co] * (partial[0][co][c] + partial[1][co][c] + ...), slice order
__global__ void wgrad_reduce_kernel(vfn_wgrad_desc p) {
    const int Kc = p.k * p.k * p.Cin;
    const size_t slab = (size_t)p.Cout * Kc;
    p.dw += (size_t)blockIdx.y * slab;
    p.partial += (size_t)blockIdx.y * p.ksplit * slab;
    if (Kc % 4 == 0) {
        const size_t total = slab / 4;
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
            f32x4 a = *reinterpret_cast<const f32x4*>(p.partial + i * 4);
            for (int s = 1; s < p.ksplit; ++s) a += *reinterpret_cast<const f32x4*>(p.partial + s * slab + i * 4);
            const int co = (int)((i * 4) / Kc);
            const float sc = p.rowscale ? p.rowscale[co] : 1.f;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = a[e] * sc;
            if (p.accumulate) v += *reinterpret_cast<const f32x4*>(p.dw + i * 4);
            *reinterpret_cast<f32x4*>(p.dw + i * 4) = v;
        }
    } else {                                   // (the 7 x 7 stems: 3 / 5 input planes)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < slab; i += (size_t)gridDim.x * blockDim.x) {
            float a = p.partial[i];
            for (int s = 1; s < p.ksplit; ++s) a += p.partial[s * slab + i];
            float v = a * (p.rowscale ? p.rowscale[i / Kc] : 1.f);
            if (p.accumulate) v += p.dw[i];
            p.dw[i] = v;
        }
    }
}

template <int TM, int TN>
int launch_wgrad(const vfn_wgrad_desc& p, hipStream_t s) {
    const int tiles = cdiv(p.Cout, 32 * TM) * p.k * p.k * cdiv(p.Cin, 32 * TN);
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    constexpr size_t lds = (size_t)3 * TM * TN * 16 * 64 * sizeof(float);
    const int nb = p.batch > 1 ? p.batch : 1;
    if (p.relu) hipLaunchKernelGGL((conv_wgrad_kernel<TM, TN, true>), dim3(tiles * ks, nb), dim3(256), lds, s, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<TM, TN, false>), dim3(tiles * ks, nb), dim3(256), lds, s, p);
    if (ks > 1 && !p.tile_counters) {
        const size_t total = (size_t)p.Cout * p.k * p.k * p.Cin / ((p.k * p.k * p.Cin) % 4 ? 1 : 4);
        const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks, nb), dim3(256), 0, s, p);
    }
    return vfn_check_launch();
}


// ---------------------------------------------------------------------------------------------------------------------------
// The 7x7 / stride-2 stems (AFB_URR.py:44,67-69: conv1 over 3 frame planes; conv1 + conv1_m + conv1_o over 5), round 5.
// conv_wgrad_kernel gives every (tap, 32-channel tile) its own pixel walk; with 3 or 5 input planes that is 49 walks with 3 of
// 32 operand columns in use -- 558 us for the query encoder's five frames (6.7 TFLOP/s), 238 us for the memory encoder, both on the
// END of the step's dependent chain.  Here the columns of the operand are (kw, c) of ONE filter row: the 7 * C values
// x[iy][2 ox - 3 .. 2 ox + 3][0 .. C) are CONTIGUOUS in the NHWC planes, so a half wave loads them with one buffer load (lane =
// column; 21 of 32 in use for C = 3, 35 of 64 for C = 5), the other operand is 32 channels of gy at the same pixel, and one
// pixel walk feeds all 7 (x NT) filter-row tiles of a 32-channel output tile:
//
//   wave        (output-channel tile cot, pixel half ph): 7 * NT accumulators of 32 x 32, units (output row, column segment)
//               dealt round-robin; per pixel pair 1 + 7 NT buffer loads (three pairs in flight) and 7 NT MFMAs (32x32x2)
//   workgroup   2 x 2 waves; the two pixel halves are summed through LDS, the block's tiles go to `partial`
//   stem_wgrad_reduce_kernel   sums the blocks' tiles in block order (deterministic), applies rowscale / accumulate and writes
//               dW [64][7][7][C] (the packed layout of vfn_conv_wgrad_f32)
template <int C, int NT>
__global__ __launch_bounds__(256, NT == 1 ? 2 : 1)
void stem_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ partial, int N, int Hp, int Wp,
                       int Ho, int Wo, int segs, int seg_len) {
    constexpr int NACC = 7 * NT, PD = NT == 1 ? 4 : 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sR = reinterpret_cast<float*>(smem);          // [cot][NACC][16][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int cot = wave & 1, ph = wave >> 1;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)((size_t)N * Hp * Wp * C * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, (int)((size_t)N * Ho * Wo * 64 * sizeof(float)), 0x00020000);
    constexpr int OOB = 0x7fffff00;

    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    // this lane's operand columns: jj = tt * 32 + li -> input column offset jj / C (jj < 7 C)
    int jx[NT];
    bool jok[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) { const int jj = tt * 32 + li; jx[tt] = jj / C; jok[tt] = jj < 7 * C; }

    const int units = N * Ho * segs;
    for (int u = blockIdx.x * 2 + ph; u < units; u += gridDim.x * 2) {
        const int row = u / segs, seg = u - row * segs;
        const int n = row / Ho, oy = row - n * Ho;
        const int ox0 = seg * seg_len, ox1 = min(Wo, ox0 + seg_len);
        const int npairs = (ox1 - ox0 + 1) >> 1;
        int rowoff[7];                                   // byte offset of input row 2 oy + kh - 3 (OOB: a padding row)
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            const int iy = 2 * oy + kh - 3;
            rowoff[kh] = (unsigned)iy < (unsigned)Hp ? ((n * Hp + iy) * Wp) * C * 4 : OOB;
        }
        const int goff = (row * Wo * 64 + cot * 32 + li) * 4;

        float av[PD];
        float bv[PD][NACC];
        auto load = [&](int t, int slot) {
            const int ox = ox0 + 2 * t + lh;
            const bool pv = ox < ox1;
            av[slot] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, pv ? goff + ox * 256 : OOB, 0, 0));
            const int ixb = 2 * ox - 3;
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                const int ix = ixb + jx[tt];
                const bool ok = pv && jok[tt] && (unsigned)ix < (unsigned)Wp;
                // one select per tile, no branch: a column outside the image (or past the segment) gets an offset that stays out of
                // range whatever row offset is added; a padding ROW's offset (OOB) stays out of range plus any in-row offset
                const unsigned coff = ok ? (unsigned)((ixb * C + tt * 32 + li) * 4) : 0x40000000u;
#pragma unroll
                for (int kh = 0; kh < 7; ++kh)
                    bv[slot][kh * NT + tt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (int)((unsigned)rowoff[kh] + coff), 0, 0));
            }
        };
#pragma unroll
        for (int d = 0; d < PD; ++d) { load(d, d); __builtin_amdgcn_sched_barrier(0); }      // (in THIS order: the loop waits for slot 0 first)
        for (int t0 = 0; t0 < npairs; t0 += PD) {
#pragma unroll
            for (int d = 0; d < PD; ++d) {
                // slot d holds pair t0 + d (the oldest loads in flight: the wait leaves the 2 (1 + 7 NT) younger ones outstanding);
                // its registers are refilled for pair t0 + d + PD right behind the MFMAs that read them (pairs past the segment
                // load zeros).  The scheduling barriers keep this order -- without them hipcc gathers the three slots' uses at the
                // loop top behind one vmcnt(0)
#pragma unroll
                for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[d], bv[d][q], acc[q], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                load(t0 + d + PD, d);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // the two pixel halves of an output-channel tile, summed in LDS; then the block's tiles -> partial[block][cot][NACC][16][64]
    if (ph == 1) {
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) sR[((cot * NACC + a) * 16 + r) * 64 + lane] = acc[a][r];
    }
    __syncthreads();
    if (ph == 0) {
        float* dst = partial + ((size_t)blockIdx.x * 2 + cot) * NACC * 1024;
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(a * 16 + r) * 64 + lane] = acc[a][r] + sR[((cot * NACC + a) * 16 + r) * 64 + lane];
    }
}

// dW[co][kh][kw][c] = rowscale[co] * sum_blocks partial[block][cot][kh * NT + tt][r][lane]  (+ dW)    co = 32 cot + (r & 3) + 8 (r >> 2)
// + 4 (lane >> 5), column (lane & 31) + 32 tt = kw * C + c.  256 threads = 64 elements (one r of one tile) x 4 groups of blocks.
template <int C, int NT>
__global__ void stem_wgrad_reduce_kernel(const float* __restrict__ partial, int nblocks, const float* __restrict__ rowscale,
                                         float* __restrict__ dw, int accumulate) {
    constexpr int NACC = 7 * NT;
    __shared__ float sS[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int e = blockIdx.x;                            // (cot, a, r)
    const int r = e & 15, a = (e >> 4) % NACC, cot = e / (16 * NACC);
    const int per = (nblocks + 3) / 4;
    const int b0 = grp * per, b1 = min(nblocks, b0 + per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const float* src = partial + ((size_t)cot * NACC + a) * 1024 + r * 64 + lane;
    const size_t bstride = (size_t)2 * NACC * 1024;
    int b = b0;
    for (; b + 4 <= b1; b += 4) {
        s0 += src[(size_t)b * bstride];
        s1 += src[(size_t)(b + 1) * bstride];
        s2 += src[(size_t)(b + 2) * bstride];
        s3 += src[(size_t)(b + 3) * bstride];
    }
    for (; b < b1; ++b) s0 += src[(size_t)b * bstride];
    sS[grp][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0) {
        const float v = (sS[0][lane] + sS[1][lane]) + (sS[2][lane] + sS[3][lane]);
        const int kh = a / NT, tt = a - kh * NT;
        const int jj = tt * 32 + (lane & 31);
        const int co = cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (jj < 7 * C) {
            float* o = dw + (size_t)co * 49 * C + kh * 7 * C + jj;
            const float w = v * (rowscale ? rowscale[co] : 1.f);
            *o = accumulate ? *o + w : w;
        }
    }
}

template <int C, int NT>
int launch_stem_wgrad(const float* x, const float* g, const float* rowscale, float* dw, float* partial, long long partial_floats,
                      int N, int Hp, int Wp, int Ho, int Wo, int accumulate, hipStream_t s) {
    constexpr int NACC = 7 * NT;
    constexpr size_t lds = (size_t)2 * NACC * 1024 * sizeof(float);
    const int rows = N * Ho;
    int nblocks = 256;
    int segs = cdiv(2048, rows);
    if (segs > Wo / 16) segs = Wo / 16 > 0 ? Wo / 16 : 1;
    int seg_len = cdiv(Wo, segs);
    seg_len += seg_len & 1;
    segs = cdiv(Wo, seg_len);
    if (rows * segs < 2 * nblocks) nblocks = cdiv(rows * segs, 2);
    if (partial_floats < (long long)nblocks * 2 * NACC * 1024) return VFN_ERR_ARG;
    static bool once = false;
    if (!once) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(stem_wgrad_kernel<C, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        once = true;
    }
    hipLaunchKernelGGL((stem_wgrad_kernel<C, NT>), dim3(nblocks), dim3(256), lds, s, x, g, partial, N, Hp, Wp, Ho, Wo, segs, seg_len);
    hipLaunchKernelGGL((stem_wgrad_reduce_kernel<C, NT>), dim3(2 * NACC * 16), dim3(256), 0, s, partial, nblocks, rowscale, dw, accumulate);
    return vfn_check_launch();
}

}  // namespace

extern "C" int vfn_conv_wgrad_f32(const vfn_wgrad_desc* d, void* stream) {
    if (!d || !d->x || !d->gy || !d->dw) return VFN_ERR_ARG;
    if (d->Cin < 1 || d->Cout < 1 || d->k < 1 || d->stride < 1 || d->N < 1 || d->ld_x < d->Cin || d->ld_g < d->Cout) return VFN_ERR_ARG;
    if (d->ksplit > 1 && !d->partial) return VFN_ERR_ARG;
    if (d->ksplit > 1 && d->tile_counters && (long long)d->ksplit * d->Cout * d->k * d->k * d->Cin * 4 >= 0x7fffff00LL) return VFN_ERR_ARG;
    if (d->Wo < 2) return VFN_ERR_ARG;                     // (the pixel walk wraps at most one row per two pixels)
    if ((long long)d->N * d->H * d->W * d->ld_x * 4 >= 0x7fffff00LL || (long long)d->N * d->Ho * d->Wo * d->ld_g * 4 >= 0x7fffff00LL) return VFN_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (d->Cout <= 32) {                                   // (the local head, the two-filter heads in their 32-channel tensors:
        if (d->Cin <= 32) return launch_wgrad<1, 1>(*d, s);   //  a 64-row tile would multiply 32 masked rows)
        return launch_wgrad<1, 2>(*d, s);
    }
    if (d->Cin <= 32) return launch_wgrad<2, 1>(*d, s);
    return launch_wgrad<2, 2>(*d, s);
}

// (ABI 12, round 5) the weight gradient of the encoders' 7x7 / stride-2 / pad-3 stems: x [N,Hp,Wp,C] dense (C = 3: conv1 of the query
// encoder, AFB_URR.py:67; C = 5: conv1 + conv1_m + conv1_o of the memory encoder over the concatenated planes, :44-46), g [N,Ho,Wo,64]
// dense = dL/d(bn1 output), rowscale [64] the frozen bn1 scale (or NULL) -> dw [64][7][7][C] (vfn_conv_wgrad_f32's packed layout);
// partial: scratch of at least vfn_stem_wgrad_scratch_floats(C) floats.  Deterministic (fixed summation order).
extern "C" int vfn_stem_wgrad_scratch_floats(int C) { return C == 3 ? 256 * 2 * 7 * 1024 : C == 5 ? 256 * 2 * 14 * 1024 : -1; }

extern "C" int vfn_stem_wgrad_f32(const float* x, const float* g, const float* rowscale, float* dw, float* partial, long long partial_floats,
                                  int N, int Hp, int Wp, int C, int Ho, int Wo, int accumulate, void* stream) {
    if (!x || !g || !dw || !partial || N < 1 || Hp < 1 || Wp < 1) return VFN_ERR_ARG;
    if (Ho != (Hp - 1) / 2 + 1 || Wo != (Wp - 1) / 2 + 1) return VFN_ERR_ARG;
    // x below 1 GiB: the kernel marks a column outside the image with byte offset 0x40000000, which must lie past the end of the
    // tensor whatever valid row offset is added to it (with a larger x the sentinel would land on real pixels)
    if ((long long)N * Hp * Wp * C * 4 >= 0x40000000LL || (long long)N * Ho * Wo * 64 * 4 >= 0x7fffff00LL) return VFN_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (C == 3) return launch_stem_wgrad<3, 1>(x, g, rowscale, dw, partial, partial_floats, N, Hp, Wp, Ho, Wo, accumulate, s);
    if (C == 5) return launch_stem_wgrad<5, 2>(x, g, rowscale, dw, partial, partial_floats, N, Hp, Wp, Ho, Wo, accumulate, s);
    return VFN_ERR_ARG;
}
