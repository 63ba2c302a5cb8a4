// Winograd F(4x4, 3x3) around the matrix kernels (round 4): the 3x3 / stride-1 / pad-1 convolutions of the decoder
// (AFB_URR.py:20-30,114-127,191-195: 256 -> 256 filters on 1/4- and 1/8-resolution feature maps, 2/3 of the frame's
// convolution FLOP) as 36 GEMMs in the transform domain -- 36 multiplies per 4x4 output tile and filter pair instead of 144:
//
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A          d: 6x6 input tile (pad 1), g: 3x3 filter, Y: 4x4 outputs
//
//   vfn_winograd_input_f32    V[xi][tile][c]  = (B^T d B)[xi]   (xi = 6 i + j; ReLU on d first for the pre-activation ResBlocks)
//   the 36 GEMMs              M[xi][tile][co] = sum_c V[xi][tile][c] U[xi][co][c]: ONE launch of the convolution kernels with
//                             batched filters (vfn_conv_desc.w_batch_rows = rows per component), U = G g G^T packed by the host
//   vfn_winograd_output_f32   Y = A^T M A per tile, then the convolution's epilogue: * scale + shift (+ residual) (ReLU)
//
// The same algebra cuDNN applies to the reference's convolutions on its own hardware; it is exact in real arithmetic, in f32
// the transforms add ~1e-6 relative rounding (tests/test_conv_gpu.py holds it against F.conv2d like every other configuration).
// Both transforms are HBM-bound: a thread owns 4 channels of one tile, so every load / store of a wave is a contiguous
// 256-byte to 1-KB row; V and M cost 36/16 of the tensor each way (119 MB each for 2 x 120 x 216 x 256).
#include "common.h"
#include "../../include/vfn_hip.h"

namespace {

// t = B^T v (6 values)
__device__ __forceinline__ void bt6(const f32x4 (&v)[6], f32x4 (&t)[6]) {
    t[0] = 4.f * v[0] - 5.f * v[2] + v[4];
    t[1] = -4.f * (v[1] + v[2]) + v[3] + v[4];
    t[2] = 4.f * (v[1] - v[2]) - v[3] + v[4];
    t[3] = -2.f * v[1] - v[2] + 2.f * v[3] + v[4];
    t[4] = 2.f * v[1] - v[2] - 2.f * v[3] + v[4];
    t[5] = 4.f * v[1] - 5.f * v[3] + v[5];
}

// Branch-free (round 5): every tap of the 6x6 patch is a raw buffer load whose offset is out of range when the tap falls outside the
// image (the hardware returns zeros), so the 36 loads leave back to back and land together.  (The first form tested the bounds
// per tap: hipcc turned each test into an exec-masked branch with the ReLU behind it -- load, s_waitcnt vmcnt(0), next load: 36
// dependent round trips per thread, 3.6-4.5 TB/s on a kernel that only moves bytes.)
// LPOUT (round 5, the plain-bf16 mode): V is written as bf16 (RNE) -- the operand the reduced-precision GEMM multiplies -- half the bytes
template <bool LPOUT = false>
__global__ __launch_bounds__(256)
void winograd_input_kernel(const float* __restrict__ x, int N, int H, int W, int C, int ld_x, int relu, float* __restrict__ V,
                           int rows_pad) {
    const int th = (H + 3) / 4, tw = (W + 3) / 4;
    const int c4n = C / 4;
    const long long total = (long long)N * th * tw * c4n;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)((size_t)N * H * W * ld_x * 4), 0x00020000);
    const float floor_ = relu ? 0.f : -INFINITY;
    for (long long i = vfn_xcd_block(blockIdx.x, gridDim.x) * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const int tile = (int)(i / c4n);
        const int tx = tile % tw, ty = (tile / tw) % th, n = tile / (tw * th);
        f32x4 d[6][6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const int yy = 4 * ty - 1 + a;
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const int xx = 4 * tx - 1 + b;
                const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
                const int off = ok ? (((n * H + yy) * W + xx) * ld_x + c4 * 4) * 4 : 0x7ffffff0;
                d[a][b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
            }
        }
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e) d[a][b][e] = vfn_floor_nan(d[a][b][e], floor_);
        // columns: d <- B^T d, then rows: V = d B
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            f32x4 v[6], t[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) v[a] = d[a][b];
            bt6(v, t);
#pragma unroll
            for (int a = 0; a < 6; ++a) d[a][b] = t[a];
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            f32x4 t[6];
            bt6(d[a], t);
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                if constexpr (LPOUT) {
                    const vfn_bf16x4 h = {(__bf16)t[b][0], (__bf16)t[b][1], (__bf16)t[b][2], (__bf16)t[b][3]};
                    *reinterpret_cast<vfn_bf16x4*>(reinterpret_cast<__bf16*>(V) + ((size_t)(a * 6 + b) * rows_pad + tile) * C + c4 * 4) = h;
                } else {
                    *reinterpret_cast<f32x4*>(V + ((size_t)(a * 6 + b) * rows_pad + tile) * C + c4 * 4) = t[b];
                }
            }
        }
    }
}

// y = A^T m (6 values -> 4)
__device__ __forceinline__ void at6(const f32x4 (&m)[6], f32x4 (&y)[4]) {
    const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = m[0] + s12 + s34;
    y[1] = d12 + 2.f * d34;
    y[2] = s12 + 4.f * s34;
    y[3] = d12 + 8.f * d34 + m[5];
}

// Branch-free (round 5): the 36 components are loaded back to back, the residual / mask taps of the 16 output pixels are raw buffer
// loads issued together behind the first transform stage, and the stores are buffer stores whose offset is out of range for pixels
// past the image edge (dropped by the hardware).  (The first form tested every pixel: 16 residual loads with a full wait each.)
template <bool MASK>
__global__ __launch_bounds__(256)
void winograd_output_kernel(const float* __restrict__ Mb, int rows_pad, int N, int H, int W, int Cout,
                            const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ res, int res_ld,
                            int res_mod, int relu_out, float* __restrict__ out, int out_ld, const float* __restrict__ mask, int mask_ld,
                            int mask_after) {
    const int th = (H + 3) / 4, tw = (W + 3) / 4;
    const int c4n = Cout / 4;
    const long long total = (long long)N * th * tw * c4n;
    const int npix = N * H * W;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((size_t)npix * out_ld * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(res ? res : Mb), 0,
                                                                        res ? (int)((size_t)(res_mod > 0 ? res_mod : npix) * res_ld * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(MASK ? mask : Mb), 0,
                                                                        MASK ? (int)((size_t)npix * mask_ld * 4) : 0, 0x00020000);
    const float floor_ = relu_out ? 0.f : -INFINITY;
    for (long long i = vfn_xcd_block(blockIdx.x, gridDim.x) * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const int tile = (int)(i / c4n);
        const int tx = tile % tw, ty = (tile / tw) % th, n = tile / (tw * th);
        f32x4 m[6][6];
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) m[a][b] = *reinterpret_cast<const f32x4*>(Mb + ((size_t)(a * 6 + b) * rows_pad + tile) * Cout + c4 * 4);
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (scale) sc = *reinterpret_cast<const f32x4*>(scale + c4 * 4);
        if (shift) sh = *reinterpret_cast<const f32x4*>(shift + c4 * 4);
        f32x4 t[6][4];                            // t = M A  (rows of M through A^T)
#pragma unroll
        for (int a = 0; a < 6; ++a) at6(m[a], t[a]);
        // the 16 pixels of the tile: row index, or -1 past the image edge
        int rowi[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int yy = 4 * ty + a, xx = 4 * tx + b;
                rowi[a][b] = (yy < H && xx < W) ? (n * H + yy) * W + xx : -1;
            }
        f32x4 rv[4][4], mk[MASK ? 4 : 1][MASK ? 4 : 1];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int row = rowi[a][b];
                const int rrow = res_mod > 0 ? row % res_mod : row;
                rv[a][b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, row >= 0 ? (rrow * res_ld + c4 * 4) * 4 : 0x7ffffff0, 0, 0));
                if constexpr (MASK)
                    mk[a][b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, row >= 0 ? (row * mask_ld + c4 * 4) * 4 : 0x7ffffff0, 0, 0));
            }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            f32x4 col[6], y[4];
#pragma unroll
            for (int a = 0; a < 6; ++a) col[a] = t[a][b];
            at6(col, y);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = y[a][e] * sc[e] + sh[e];
                if constexpr (MASK) {
                    if (!mask_after) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = mk[a][b][e] > 0.f ? v[e] : 0.f;
                    }
                }
                v += rv[a][b];                                     // (zeros without a residual: the resource is empty)
                if constexpr (MASK) {
                    if (mask_after) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = mk[a][b][e] > 0.f ? v[e] : 0.f;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = vfn_floor_nan(v[e], floor_);
                const int row = rowi[a][b];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), ro,
                                                       row >= 0 ? (row * out_ld + c4 * 4) * 4 : 0x7ffffff0, 0, 0);
            }
        }
    }
}

// ---- the weight gradient in the transform domain (round 4): dW = G^T [ sum_tiles (B^T d B) (.) (A dY A^T) ] G -- the transposition
// of the forward algorithm: 36 multiplies per (tile, filter, channel) instead of the direct form's 144.
//   vfn_winograd_input_f32   V[xi][tile][ci] = (B^T d B)[xi]           (the forward's input transform, ReLU included)
//   vfn_winograd_gy_f32      Z[xi][tile][co] = (A dY A^T)[xi]          dY: the 4x4 tile of dL/dy (zero outside the image)
//   vfn_conv_wgrad_f32       dU[xi][co][ci]  = sum_tile Z[xi][tile][co] V[xi][tile][ci]     (batch = 36 one-by-one problems)
//   vfn_winograd_dw_f32      dW[co][a][b][ci] (+)= rowscale[co] * sum_ij G[i][a] dU[6i+j][co][ci] G[j][b]
// z = A v (4 values -> 6), A = (A^T)^T
__device__ __forceinline__ void a6(const f32x4 (&v)[4], f32x4 (&z)[6]) {
    const f32x4 e = v[0] + v[2], o = v[1] + v[3];
    const f32x4 e4 = v[0] + 4.f * v[2], o8 = 2.f * v[1] + 8.f * v[3];
    z[0] = v[0];
    z[1] = e + o;
    z[2] = e - o;
    z[3] = e4 + o8;
    z[4] = e4 - o8;
    z[5] = v[3];
}

__global__ __launch_bounds__(256)
void winograd_gy_kernel(const float* __restrict__ gy, int N, int H, int W, int C, int ld, float* __restrict__ Z, int rows_pad) {
    const int th = (H + 3) / 4, tw = (W + 3) / 4;
    const int c4n = C / 4;
    const long long total = (long long)N * th * tw * c4n;
    for (long long i = vfn_xcd_block(blockIdx.x, gridDim.x) * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const int tile = (int)(i / c4n);
        const int tx = tile % tw, ty = (tile / tw) % th, n = tile / (tw * th);
        f32x4 t[6][4];                               // t = A dY (columns of dY through A)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            f32x4 col[4], z[6];
            const int xx = 4 * tx + b;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int yy = 4 * ty + a;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (yy < H && xx < W) v = *reinterpret_cast<const f32x4*>(gy + ((size_t)(n * H + yy) * W + xx) * ld + c4 * 4);
                col[a] = v;
            }
            a6(col, z);
#pragma unroll
            for (int a = 0; a < 6; ++a) t[a][b] = z[a];
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            f32x4 z[6];
            a6(t[a], z);
#pragma unroll
            for (int b = 0; b < 6; ++b)
                *reinterpret_cast<f32x4*>(Z + ((size_t)(a * 6 + b) * rows_pad + tile) * C + c4 * 4) = z[b];
        }
    }
}

// w = G^T u (6 values -> 3)
__device__ __forceinline__ void gt3(const f32x4 (&u)[6], f32x4 (&w)[3]) {
    const f32x4 s12 = u[1] + u[2], d12 = u[2] - u[1], s34 = u[3] + u[4], d34 = u[3] - u[4];
    w[0] = 0.25f * u[0] - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
    w[1] = (1.f / 6.f) * d12 + (1.f / 12.f) * d34;
    w[2] = -(1.f / 6.f) * s12 + (1.f / 6.f) * s34 + u[5];
}

__global__ __launch_bounds__(256)
void winograd_dw_kernel(const float* __restrict__ dU, int Cout, int Cin, const float* __restrict__ rowscale, float* __restrict__ dw,
                        int accumulate) {
    const int c4n = Cin / 4;
    const long long total = (long long)Cout * c4n;
    const size_t bank = (size_t)Cout * Cin;
    for (long long i = vfn_xcd_block(blockIdx.x, gridDim.x) * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n), co = (int)(i / c4n);
        f32x4 t[3][6];                               // t = G^T dU (columns through G^T)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            f32x4 u[6], w[3];
#pragma unroll
            for (int a = 0; a < 6; ++a) u[a] = *reinterpret_cast<const f32x4*>(dU + (size_t)(a * 6 + j) * bank + (size_t)co * Cin + c4 * 4);
            gt3(u, w);
#pragma unroll
            for (int a = 0; a < 3; ++a) t[a][j] = w[a];
        }
        const float sc = rowscale ? rowscale[co] : 1.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            f32x4 w[3];
            gt3(t[a], w);
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                f32x4* o = reinterpret_cast<f32x4*>(dw + ((size_t)co * 9 + a * 3 + b) * Cin + c4 * 4);
                f32x4 v = w[b] * sc;
                if (accumulate) v += *o;
                *o = v;
            }
        }
    }
}


// ---- the transform-domain GEMMs as ONE PERSISTENT launch (round 5) ------------------------------------------------------------
// M[xi][rows][Cout] = V[xi][rows][C] U[xi][cout][C]^T for the 36 components.  Each GEMM has K = C = 128 ... 1024 only, i.e. 4-32 K tiles:
// as separate workgroups (the batched-filter launch of conv_igemm_kernel) a 128 x 128 tile spends 8 K tiles of matrix work between a
// cold prologue (first operand tiles from HBM / L2: 2-14 us in the census, profiles/r05_census_wino_gemm.txt) and an epilogue through
// LDS, and rocprofv3 counts the matrix pipe busy 49 % of the launch.  Here a workgroup walks a LIST of (component, row tile, filter
// tile) units as one uninterrupted K loop: the operand tiles of the next unit are requested PD tiles ahead while the current unit is
// still multiplying, the accumulators leave through dword stores straight from the registers (a raw M tile has no epilogue
// arithmetic) and are zeroed in place -- no drain, no refill, no LDS transpose between units.  Same fragment layout, swizzle and
// k-order as conv_igemm_kernel: every output element is the same fmaf chain, bit for bit.
struct wino_gemm_args {
    const float* V;       // [comps][rows_pad][C]
    const float* U;       // [comps][cout_pad][C]
    float* Mb;            // [comps][rows_pad][Cout]
    int comps, rows_pad, C, Cout, cout_pad;
    int mtiles, ntiles;   // rows_pad / BM, ceil(Cout / BN)
    // CONV form (a 1x1 / stride-1 convolution as ONE such GEMM with the layer's epilogue): V = NHWC input with pixel stride a_ld
    // (rows_pad = M pixels: rows past M read zeros through the buffer resource), Mb = output with pixel stride out_ld
    int a_ld, out_ld, res_ld, res_mod, relu_in, relu_out;
    const float* scale;
    const float* shift;
    const float* res;
};

// LP: both operands are bf16 in memory (V from winograd_input_kernel<true>, U packed by the host): a K tile is 64 channels -- the same
// 128-byte LDS rows, the same staging -- and a 16-byte fragment is one v_mfma_f32_32x32x16_bf16 (k = 16 kk + 8 h .. + 7), f32 accumulate
template <int BM, int BN, int WM, int WN, int PD, bool CONV = false, bool LP = false>
__global__ __launch_bounds__(WM * WN * 64)
void wino_gemm_kernel(const wino_gemm_args p) {
    static_assert(!(CONV && LP), "the reduced-precision form is the transform-domain GEMM only");
    constexpr int BK = 32;                               // floats per LDS row (128 bytes = 32 f32 / 64 bf16)
    constexpr int KT = LP ? 64 : 32;                     // K elements per tile
    constexpr int ES = LP ? 2 : 4;                       // bytes per operand element
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int AC = BM * 8 / NT, BC = BN * 8 / NT, RSTEP = NT / 8;
    static_assert(AC >= 1 && BC >= 1, "tile too small for the thread count");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sA = reinterpret_cast<float*>(smem);          // [2][BM][32]
    float* sB = sA + 2 * BM * BK;                        // [2][BN][32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    // units of this workgroup: XCD x (= blockIdx % 8) owns one contiguous run of the unit list (filter tile fastest, then row
    // tile, then component: the workgroups of one XCD share operand tiles through its L2), dealt round-robin to its workgroups
    const int per = p.mtiles * p.ntiles, total = p.comps * per;
    const int gx = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, loc = (int)blockIdx.x >> 3;
    const int q = total >> 3, r8 = total & 7;
    const int u_begin = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
    const int u_end = u_begin + q + (xcd < r8 ? 1 : 0);
    const int first = u_begin + loc;
    const int n_units = first < u_end ? (u_end - first + gx - 1) / gx : 0;
    const int nk = p.C / KT;
    const int T = n_units * nk;
    if (T == 0) return;

    const int c16 = tid & 7, r0 = tid >> 3;
    const int a_ld = CONV ? p.a_ld : p.C;
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.V), 0, (int)((size_t)p.comps * p.rows_pad * a_ld * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.U), 0, (int)((size_t)p.comps * p.cout_pad * p.C * ES), 0x00020000);
    int a_thr[AC], b_thr[BC];
#pragma unroll
    for (int j = 0; j < AC; ++j) a_thr[j] = (r0 + j * RSTEP) * a_ld * ES + c16 * 16;
#pragma unroll
    for (int j = 0; j < BC; ++j) b_thr[j] = (r0 + j * RSTEP) * p.C * ES + c16 * 16;

    // the tile being requested: (unit, K tile) and its operand bases (wave-uniform)
    int lu = 0, lkt = 0, la_base = 0, lb_base = 0;
    int n0_cur = 0;
    auto unit_bases = [&](int ui, int& a_base, int& b_base, size_t& o_base, int* n0 = nullptr) {
        const int u = first + ui * gx;
        const int xi = u / per, rem = u - xi * per;
        const int mt = rem / p.ntiles, nt = rem - mt * p.ntiles;
        a_base = (xi * p.rows_pad + mt * BM) * a_ld * ES;
        b_base = (xi * p.cout_pad + nt * BN) * p.C * ES;
        o_base = CONV ? (size_t)mt * BM : ((size_t)xi * p.rows_pad + mt * BM) * p.Cout + nt * BN;      // (CONV: the first row of the tile)
        if (n0) *n0 = nt * BN;
    };
    size_t o_dummy;
    unit_bases(0, la_base, lb_base, o_dummy);
    f32x4 ra[PD][AC], rb[PD][BC];
    auto request = [&](int slot) {                         // global loads of the next tile in line into staging slot `slot`
        const int koff = lkt * 128;                        // (a K tile is 128 bytes of every operand row in either arithmetic)
#pragma unroll
        for (int j = 0; j < AC; ++j)
            ra[slot][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsV, a_thr[j], la_base + koff, 0));
#pragma unroll
        for (int j = 0; j < BC; ++j)
            rb[slot][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsU, b_thr[j], lb_base + koff, 0));
        if (++lkt == nk) {
            lkt = 0;
            if (++lu < n_units) unit_bases(lu, la_base, lb_base, o_dummy);
        }
    };
    const float relu_floor = (CONV && p.relu_in) ? 0.f : -INFINITY;
    auto stage = [&](int buf, int slot) {                  // staging slot -> LDS image (XOR-swizzled 16-byte chunks)
        float* dA = sA + buf * BM * BK;
        float* dB = sB + buf * BN * BK;
#pragma unroll
        for (int j = 0; j < AC; ++j) {
            const int r = r0 + j * RSTEP;
            f32x4 v = ra[slot][j];
            if constexpr (CONV) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], relu_floor);
            }
            *reinterpret_cast<f32x4*>(dA + r * BK + ((c16 ^ ((r >> 1) & 7)) << 2)) = v;
        }
#pragma unroll
        for (int j = 0; j < BC; ++j) {
            const int r = r0 + j * RSTEP;
            *reinterpret_cast<f32x4*>(dB + r * BK + ((c16 ^ ((r >> 1) & 7)) << 2)) = rb[slot][j];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // the unit being multiplied
    int cu = 0, ckt = 0, ca_dummy, cb_dummy;
    size_t o_base;
    unit_bases(0, ca_dummy, cb_dummy, o_base, &n0_cur);

#pragma unroll
    for (int d = 0; d < PD; ++d)
        if (d < T) request(d);
    stage(0, 0);
    __syncthreads();

    for (int t0 = 0; t0 < T; t0 += PD) {
#pragma unroll
        for (int uu = 0; uu < PD; ++uu) {
            const int t = t0 + uu;
            if (t >= T) break;
            const int buf = t & 1;
            const bool more = t + 1 < T;                   // tile t+1 goes to LDS during this tile (from slot (uu + 1) % PD)
            const bool more_req = t + PD < T;              // tile t+PD is requested during this tile (into slot uu, free by now)
            const float* cA = sA + buf * BM * BK + (wm * TM * 32) * BK;
            const float* cB = sB + buf * BN * BK + (wn * TN * 32) * BK;
            auto read_frags = [&](int kk, f32x4 (&a)[TM], f32x4 (&b)[TN]) {
                const int lc = 2 * kk + lh;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int r = i * 32 + li;
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
                if (PD > 1 && kk == 0 && more) stage(buf ^ 1, (uu + 1) % PD);       // (requested a whole tile ago: landed)
                if (kk == 1 && more_req) request(uu);
                if (PD == 1 && kk == 3 && more) stage(buf ^ 1, 0);
                if (kk + 1 < 4) read_frags(kk + 1, fa[(kk + 1) & 1], fb[(kk + 1) & 1]);
                if constexpr (LP) {
                    typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_, fa[kk & 1][i]), __builtin_bit_cast(bf16x8_, fb[kk & 1][j]),
                                                                                acc[i][j], 0, 0, 0);
                } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk & 1][i][e], fb[kk & 1][j][e], acc[i][j], 0, 0, 0);
                }
            }
            if (++ckt == nk) {
                // the unit is complete: lane = filter column (lane & 31), registers = rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
                if constexpr (CONV) {
                    // y = act(acc * scale[c] + shift[c] + res): lane = channel, so scale / shift are one value per lane and column tile;
                    // the residual taps of the whole tile are requested first (buffer loads: rows past M / columns past Cout return
                    // zeros), the stores of invalid positions are dropped by the hardware
                    const int m0 = (int)o_base;
                    const int npix = p.rows_pad;
                    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.Mb, 0, (int)((size_t)npix * p.out_ld * 4), 0x00020000);
                    const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.V), 0,
                        p.res ? (int)((size_t)(p.res_mod > 0 ? p.res_mod : npix) * p.res_ld * 4) : 0, 0x00020000);
                    const float floor_ = p.relu_out ? 0.f : -INFINITY;
                    float rv[TM][TN][16];
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int col = n0_cur + (wn * TN + j) * 32 + li;
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int row = m0 + (wm * TM + i) * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
                                const int rrow = p.res_mod > 0 ? row % p.res_mod : row;
                                rv[i][j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                    rsr, (row < npix && col < p.Cout) ? (rrow * p.res_ld + col) * 4 : 0x7ffffff0, 0, 0));
                            }
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int col = n0_cur + (wn * TN + j) * 32 + li;
                        const bool cok = col < p.Cout;
                        const float sc = (p.scale && cok) ? p.scale[col] : 1.f;
                        const float sh = (p.shift && cok) ? p.shift[col] : 0.f;
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int row = m0 + (wm * TM + i) * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
                                const float v = vfn_floor_nan(acc[i][j][r] * sc + sh + rv[i][j][r], floor_);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rso,
                                                                      (row < npix && cok) ? (row * p.out_ld + col) * 4 : 0x7ffffff0, 0, 0);
                                acc[i][j][r] = 0.f;
                            }
                    }
                } else {
                float* o = p.Mb + o_base;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = (wn * TN + j) * 32 + li;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int rbase = (wm * TM + i) * 32 + 4 * lh;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            if (n0_cur + col < p.Cout)
                                o[(size_t)(rbase + (r & 3) + 8 * (r >> 2)) * p.Cout + col] = acc[i][j][r];
                            acc[i][j][r] = 0.f;
                        }
                    }
                }
                }
                ckt = 0;
                if (++cu < n_units) unit_bases(cu, ca_dummy, cb_dummy, o_base, &n0_cur);
            }
            __syncthreads();
        }
    }
}

template <int BM, int BN, int WM, int WN, int PD, bool CONV = false, bool LP = false>
int launch_wino_gemm(const wino_gemm_args& a, int wgs, hipStream_t s) {
    constexpr size_t lds = 2 * (size_t)(BM + BN) * 32 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_gemm_kernel<BM, BN, WM, WN, PD, CONV, LP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    wino_gemm_args p = a;
    p.mtiles = (a.rows_pad + BM - 1) / BM;
    p.ntiles = (a.Cout + BN - 1) / BN;
    const int total = p.comps * p.mtiles * p.ntiles;
    int grid = wgs > 0 ? wgs : 512;
    if (grid > total) grid = total;
    grid = (grid + 7) / 8 * 8;                              // (the kernel deals units to blockIdx & 7 = XCD, blockIdx >> 3 = slot)
    hipLaunchKernelGGL((wino_gemm_kernel<BM, BN, WM, WN, PD, CONV, LP>), dim3(grid), dim3(WM * WN * 64), lds, s, p);
    return vfn_check_launch();
}

// the transforms address activations through buffer resources with 32-bit byte offsets: every tensor must stay below 2 GiB
inline bool tensors_fit_32bit(int N, int H, int W, int ld, int res_ld, int mask_ld, int res_mod) {
    const long long px = (long long)N * H * W;
    return px * ld * 4 < 0x7fffff00LL && (res_mod > 0 ? (long long)res_mod : px) * res_ld * 4 < 0x7fffff00LL && px * mask_ld * 4 < 0x7fffff00LL;
}

inline int grid_of(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b < 16384 ? (b ? b : 1) : 16384);
}

}  // namespace

extern "C" int vfn_winograd_tiles(int N, int H, int W) { return N * ((H + 3) / 4) * ((W + 3) / 4); }

extern "C" int vfn_winograd_input_f32(const float* x, int N, int H, int W, int C, int ld_x, int relu, float* V, int rows_pad, void* stream) {
    if (!x || !V || N < 1 || H < 1 || W < 1 || C < 4 || C % 4 || ld_x < C || ld_x % 4 || rows_pad < vfn_winograd_tiles(N, H, W)) return VFN_ERR_ARG;
    const long long total = (long long)vfn_winograd_tiles(N, H, W) * (C / 4);
    if (!tensors_fit_32bit(N, H, W, ld_x, 0, 0, 0)) return VFN_ERR_ARG;
    hipLaunchKernelGGL(winograd_input_kernel<false>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, x, N, H, W, C, ld_x, relu, V, rows_pad);
    return vfn_check_launch();
}

extern "C" int vfn_winograd_output_f32(const float* Mb, int rows_pad, int N, int H, int W, int Cout, const float* scale, const float* shift,
                                       const float* res, int res_ld, int res_mod, int relu_out, float* out, int out_ld, void* stream) {
    if (!Mb || !out || N < 1 || H < 1 || W < 1 || Cout < 4 || Cout % 4 || out_ld < Cout || out_ld % 4 || (res && res_ld % 4) ||
        rows_pad < vfn_winograd_tiles(N, H, W)) return VFN_ERR_ARG;
    const long long total = (long long)vfn_winograd_tiles(N, H, W) * (Cout / 4);
    if (!tensors_fit_32bit(N, H, W, out_ld, res ? res_ld : 0, 0, res_mod)) return VFN_ERR_ARG;
    hipLaunchKernelGGL(winograd_output_kernel<false>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, Mb, rows_pad, N, H, W, Cout, scale, shift,
                       res, res_ld, res_mod, relu_out, out, out_ld, (const float*)nullptr, 0, 0);
    return vfn_check_launch();
}

// ... with the epilogue of a data-gradient convolution (vfn_conv_desc.mask / mask_after): the result is zeroed where mask <= 0,
// before (mask_after = 0) or after the residual is added
extern "C" int vfn_winograd_output_masked_f32(const float* Mb, int rows_pad, int N, int H, int W, int Cout, const float* res, int res_ld,
                                              const float* mask, int mask_ld, int mask_after, float* out, int out_ld, void* stream) {
    if (!Mb || !out || N < 1 || H < 1 || W < 1 || Cout < 4 || Cout % 4 || out_ld < Cout || out_ld % 4 || (res && res_ld % 4) ||
        (mask && mask_ld % 4) || rows_pad < vfn_winograd_tiles(N, H, W)) return VFN_ERR_ARG;
    const long long total = (long long)vfn_winograd_tiles(N, H, W) * (Cout / 4);
    if (!tensors_fit_32bit(N, H, W, out_ld, res ? res_ld : 0, mask ? mask_ld : 0, 0)) return VFN_ERR_ARG;
    if (mask)
        hipLaunchKernelGGL(winograd_output_kernel<true>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, Mb, rows_pad, N, H, W, Cout,
                           (const float*)nullptr, (const float*)nullptr, res, res_ld, 0, 0, out, out_ld, mask, mask_ld, mask_after);
    else
        hipLaunchKernelGGL(winograd_output_kernel<false>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, Mb, rows_pad, N, H, W, Cout,
                           (const float*)nullptr, (const float*)nullptr, res, res_ld, 0, 0, out, out_ld, (const float*)nullptr, 0, 0);
    return vfn_check_launch();
}

extern "C" int vfn_winograd_gy_f32(const float* gy, int N, int H, int W, int C, int ld, float* Z, int rows_pad, void* stream) {
    if (!gy || !Z || N < 1 || H < 1 || W < 1 || C < 4 || C % 4 || ld < C || ld % 4 || rows_pad < vfn_winograd_tiles(N, H, W)) return VFN_ERR_ARG;
    const long long total = (long long)vfn_winograd_tiles(N, H, W) * (C / 4);
    hipLaunchKernelGGL(winograd_gy_kernel, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, gy, N, H, W, C, ld, Z, rows_pad);
    return vfn_check_launch();
}

extern "C" int vfn_winograd_dw_f32(const float* dU, int Cout, int Cin, const float* rowscale, float* dw, int accumulate, void* stream) {
    if (!dU || !dw || Cout < 1 || Cin < 4 || Cin % 4) return VFN_ERR_ARG;
    hipLaunchKernelGGL(winograd_dw_kernel, dim3(grid_of((long long)Cout * (Cin / 4))), dim3(256), 0, (hipStream_t)stream, dU, Cout, Cin, rowscale,
                       dw, accumulate);
    return vfn_check_launch();
}

// (ABI 12) the 36 transform-domain GEMMs as one persistent launch: M [comps][rows_pad][Cout] = V [comps][rows_pad][C] x U [comps][cout_pad][C]^T.
// cfg: 0 = 128x128 tiles (8 waves), 1 = 64x128 (8 waves), 2 = 128x64 (8 waves), 3 = 64x64 (4 waves); + 4: operand tiles requested two K tiles
// ahead instead of one.  wgs: workgroups to launch (0 = 512, two per CU); rows_pad a multiple of the tile height, cout_pad >= the padded
// filter count, C a multiple of 32; every operand below 2 GiB (32-bit buffer offsets)
extern "C" int vfn_winograd_gemm_f32(const float* V, const float* U, float* Mb, int comps, int rows_pad, int C, int Cout, int cout_pad, int cfg,
                                     int wgs, void* stream) {
    if (!V || !U || !Mb || comps < 1 || rows_pad < 1 || C < 32 || C % 32 || Cout < 1 || cout_pad < Cout || cfg < 0 || cfg > 7) return VFN_ERR_ARG;
    static const int bm[4] = {128, 64, 128, 64}, bn[4] = {128, 128, 64, 64};
    const int tc = cfg & 3;
    if (rows_pad % bm[tc] || cout_pad < (Cout + bn[tc] - 1) / bn[tc] * bn[tc]) return VFN_ERR_ARG;
    if ((long long)comps * rows_pad * C * 4 >= 0x7fffff00LL || (long long)comps * cout_pad * C * 4 >= 0x7fffff00LL) return VFN_ERR_ARG;
    wino_gemm_args a{V, U, Mb, comps, rows_pad, C, Cout, cout_pad, 0, 0, 0, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr};
    hipStream_t s = (hipStream_t)stream;
    switch (cfg) {
        case 0: return launch_wino_gemm<128, 128, 4, 2, 1>(a, wgs, s);
        case 1: return launch_wino_gemm<64, 128, 2, 4, 1>(a, wgs, s);
        case 2: return launch_wino_gemm<128, 64, 4, 2, 1>(a, wgs, s);
        case 3: return launch_wino_gemm<64, 64, 2, 2, 1>(a, wgs, s);
        case 4: return launch_wino_gemm<128, 128, 4, 2, 2>(a, wgs, s);
        case 5: return launch_wino_gemm<64, 128, 2, 4, 2>(a, wgs, s);
        case 6: return launch_wino_gemm<128, 64, 4, 2, 2>(a, wgs, s);
        case 7: return launch_wino_gemm<64, 64, 2, 2, 2>(a, wgs, s);
    }
    return VFN_ERR_ARG;
}

// (ABI 12) a 1x1 / stride-1 convolution (+ eval BatchNorm / bias, residual, ReLU: the trunk's conv1 / conv3 / downsample, AFB_URR.py:59-61,
// 89-91 through torchvision's Bottleneck) through the same persistent kernel: ONE GEMM [M pixels x Cin] x [Cin x Cout] whose workgroups
// walk their list of output tiles as one uninterrupted K loop and apply the epilogue from the accumulator registers.  These layers
// have K = 64 ... 256 -- two to eight K tiles per output tile -- so as one workgroup per tile they are all prologue and epilogue
// (rocprofv3: matrix pipe busy 20-35 %, 3.8 TB/s on layers that only have to move their tensors once).  Same products in the same
// order as vfn_conv2d_nhwc_f32 without split-K.  cfg / wgs as vfn_winograd_gemm_f32.  Refuses what it does not implement (taps, strides,
// masks, operand images, split-K) with VFN_ERR_ARG.
extern "C" int vfn_conv1x1_persistent_f32(const vfn_conv_desc* d, int cfg, int wgs, void* stream) {
    if (!d || !d->in || !d->w || !d->out || cfg < 0 || cfg > 7) return VFN_ERR_ARG;
    if (d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0 || d->mask || d->in_lp || d->out_lp || d->w_packed || d->ksplit > 1 ||
        d->w_batch_rows || d->Cin % 32 || d->in_ld % 4 || d->M < 1 || d->M != d->N * d->H * d->W) return VFN_ERR_ARG;
    static const int bn[4] = {128, 128, 64, 64};
    if (d->cout_pad < (d->Cout + bn[cfg & 3] - 1) / bn[cfg & 3] * bn[cfg & 3]) return VFN_ERR_ARG;
    const long long lim = 0x7fffff00LL;
    if ((long long)d->M * d->in_ld * 4 >= lim || (long long)d->M * d->out_ld * 4 >= lim || (long long)d->cout_pad * d->Cin * 4 >= lim ||
        (d->res && (long long)(d->res_mod > 0 ? d->res_mod : d->M) * d->res_ld * 4 >= lim)) return VFN_ERR_ARG;
    wino_gemm_args a{d->in, d->w, d->out, 1, d->M, d->Cin, d->Cout, d->cout_pad, 0, 0,
                     d->in_ld, d->out_ld, d->res ? d->res_ld : 0, d->res_mod, d->relu_in, d->relu_out, d->scale, d->shift, d->res};
    hipStream_t s = (hipStream_t)stream;
    switch (cfg) {
        case 0: return launch_wino_gemm<128, 128, 4, 2, 1, true>(a, wgs, s);
        case 1: return launch_wino_gemm<64, 128, 2, 4, 1, true>(a, wgs, s);
        case 2: return launch_wino_gemm<128, 64, 4, 2, 1, true>(a, wgs, s);
        case 3: return launch_wino_gemm<64, 64, 2, 2, 1, true>(a, wgs, s);
        case 4: return launch_wino_gemm<128, 128, 4, 2, 2, true>(a, wgs, s);
        case 5: return launch_wino_gemm<64, 128, 2, 4, 2, true>(a, wgs, s);
        case 6: return launch_wino_gemm<128, 64, 4, 2, 2, true>(a, wgs, s);
        case 7: return launch_wino_gemm<64, 64, 2, 2, 2, true>(a, wgs, s);
    }
    return VFN_ERR_ARG;
}

// (ABI 12) the plain-bf16 mode's Winograd layers (BASELINE configs C3 / C5): the input transform writes V as bf16 (computed in f32, rounded to
// nearest-even once -- the rounding the bf16 convolution applies to its operands as it stages them), the filter banks U are bf16 (packed by
// the host from the float64 transform), the persistent GEMM multiplies them on v_mfma_f32_32x32x16_bf16 with f32 accumulation and writes M in
// f32 for vfn_winograd_output_f32.  C a multiple of 64.
extern "C" int vfn_winograd_input_bf16(const float* x, int N, int H, int W, int C, int ld_x, int relu, void* V, int rows_pad, void* stream) {
    if (!x || !V || N < 1 || H < 1 || W < 1 || C < 4 || C % 4 || ld_x < C || ld_x % 4 || rows_pad < vfn_winograd_tiles(N, H, W)) return VFN_ERR_ARG;
    const long long total = (long long)vfn_winograd_tiles(N, H, W) * (C / 4);
    if (!tensors_fit_32bit(N, H, W, ld_x, 0, 0, 0)) return VFN_ERR_ARG;
    hipLaunchKernelGGL(winograd_input_kernel<true>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, x, N, H, W, C, ld_x, relu,
                       reinterpret_cast<float*>(V), rows_pad);
    return vfn_check_launch();
}

extern "C" int vfn_winograd_gemm_bf16(const void* V, const void* U, float* Mb, int comps, int rows_pad, int C, int Cout, int cout_pad, int cfg,
                                      int wgs, void* stream) {
    if (!V || !U || !Mb || comps < 1 || rows_pad < 1 || C < 64 || C % 64 || Cout < 1 || cout_pad < Cout || cfg < 0 || cfg > 7) return VFN_ERR_ARG;
    static const int bm[4] = {128, 64, 128, 64}, bn[4] = {128, 128, 64, 64};
    const int tc = cfg & 3;
    if (rows_pad % bm[tc] || cout_pad < (Cout + bn[tc] - 1) / bn[tc] * bn[tc]) return VFN_ERR_ARG;
    if ((long long)comps * rows_pad * C * 2 >= 0x7fffff00LL || (long long)comps * cout_pad * C * 2 >= 0x7fffff00LL) return VFN_ERR_ARG;
    wino_gemm_args a{reinterpret_cast<const float*>(V), reinterpret_cast<const float*>(U), Mb, comps, rows_pad, C, Cout, cout_pad, 0, 0, 0, 0, 0, 0, 0, 0,
                     nullptr, nullptr, nullptr};
    hipStream_t s = (hipStream_t)stream;
    switch (cfg) {
        case 0: return launch_wino_gemm<128, 128, 4, 2, 1, false, true>(a, wgs, s);
        case 1: return launch_wino_gemm<64, 128, 2, 4, 1, false, true>(a, wgs, s);
        case 2: return launch_wino_gemm<128, 64, 4, 2, 1, false, true>(a, wgs, s);
        case 3: return launch_wino_gemm<64, 64, 2, 2, 1, false, true>(a, wgs, s);
        case 4: return launch_wino_gemm<128, 128, 4, 2, 2, false, true>(a, wgs, s);
        case 5: return launch_wino_gemm<64, 128, 2, 4, 2, false, true>(a, wgs, s);
        case 6: return launch_wino_gemm<128, 64, 4, 2, 2, false, true>(a, wgs, s);
        case 7: return launch_wino_gemm<64, 64, 2, 2, 2, false, true>(a, wgs, s);
    }
    return VFN_ERR_ARG;
}
