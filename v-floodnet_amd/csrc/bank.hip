// Adaptive feature bank maintenance -- FeatureBank.update / remove (FeatureBank.py:53-143) and the
// torch_scatter.scatter_mean call sites (FeatureBank.py:78,92) -- entirely on the device:
// the live bank length stays in device memory and no step needs a host round trip
// (the reference syncs on nonzero() x2, unique(), int(LFU.min()) per object per frame).
//
//   vfn_row_norms      ||x_b|| per bank entry / new feature               FeatureBank.py:63-65,87-89
//   vfn_bank_merge     segmented mean of the normalised new features that matched a bank
//                      entry with corr > thres_close (= scatter_mean into a zero buffer),
//                      blended into that entry with its magnitude kept    FeatureBank.py:71-97
//   vfn_bank_plan      append set (corr <= thres), ascending positions (nonzero order),
//                      and -- when class_budget < B + n_append -- the LFU eviction
//                      threshold loop + order-preserving keep map          FeatureBank.py:100-103,117-143
//   vfn_bank_compact   boolean-mask compaction of keys / values / info     FeatureBank.py:128-131
//   vfn_bank_append    torch.cat of the new columns + info rows, peak_n,
//                      clamp(info[:,1], 0, 1e5), new length                FeatureBank.py:105-115
//   vfn_scatter_mean   the operator itself (dim=1, row-broadcast index, out=)
//
// All sums run in ascending source order, so results are bit-reproducible run to run
// (torch_scatter's CUDA path uses float atomics and is not).
#include "common.h"
#include "../../include/vfn_hip.h"

namespace {

constexpr int DK = 128, DV = 512;

__device__ __forceinline__ float block_reduce_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_reduce_min(float v, float* red) {
    v = wave_min(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = INFINITY;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t = fminf(t, red[i]);
    return t;
}
__device__ __forceinline__ int block_reduce_sum_i(int v, int* red) {
    v = wave_sum_i(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    int t = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}

// exclusive prefix of a 0/1 flag over the block; returns this thread's offset, total in *total
__device__ __forceinline__ int block_excl_scan(int flag, int* red, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    const int within = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) red[w] = __popcll(m);
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { if (i < w) base += red[i]; tot += red[i]; }
    *total = tot;
    return base + within;
}

// one wave per row
__global__ void row_norms_kernel(const float* __restrict__ x, long long stride_obj, int ld, int dim,
                                 const int* __restrict__ len_dev, int rows_fixed,
                                 float* __restrict__ nrm, float* __restrict__ inv, long long stride_n) {
    const int obj = blockIdx.y;
    const int rows = len_dev ? len_dev[obj] : rows_fixed;
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int r = blockIdx.x * wpb + (threadIdx.x >> 6); r < rows; r += gridDim.x * wpb) {
        const float* src = x + (size_t)obj * stride_obj + (size_t)r * ld;
        float s = 0.f;
        for (int d = lane * 4; d < dim; d += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + d);
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        s = wave_sum(s);
        if (lane == 0) {
            const float n = sqrtf(s);
            nrm[(size_t)obj * stride_n + r] = n;
            if (inv) inv[(size_t)obj * stride_n + r] = 1.f / fmaxf(n, 1e-12f);      // F.normalize eps
        }
    }
}

// ---------------------------------------------------------------- merge (scatter_mean + blend)
__global__ __launch_bounds__(256)
void bank_merge_kernel(const vfn_bank_desc p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_any;
    const int obj = blockIdx.y, hw = blockIdx.x, tid = threadIdx.x;
    const int* idx = p.match_idx + (size_t)obj * p.HW;
    const float* corr = p.match_corr + (size_t)obj * p.HW;
    if (!(corr[hw] > p.thres_close)) return;
    const int tgt = idx[hw];
    // leader = smallest source index among the sources merged into tgt
    if (tid == 0) s_any = 0;
    __syncthreads();
    int found = 0;
    for (int j = tid; j < hw; j += 256) found |= (idx[j] == tgt && corr[j] > p.thres_close);
    if (found) s_any = 1;
    __syncthreads();
    if (s_any) return;
    // bitmap of the sources merged into tgt (this block is their leader): set in parallel, walked in ascending order
    unsigned* s_bits = reinterpret_cast<unsigned*>(smem);               // [(HW + 31) / 32]
    const int nwords = (p.HW + 31) >> 5;
    for (int w = tid; w < nwords; w += 256) s_bits[w] = 0u;
    __syncthreads();
    for (int j = hw + tid; j < p.HW; j += 256)
        if (idx[j] == tgt && corr[j] > p.thres_close) atomicOr(&s_bits[j >> 5], 1u << (j & 31));
    __syncthreads();

    const float* nk = p.new_k + (size_t)obj * p.stride_new;       // [HW][ld_new]: key at +0, value at +voff
    const float* nkn = p.new_knorm + (size_t)obj * p.HW;
    const float* nvn = p.new_vnorm + (size_t)obj * p.HW;
    float ak = 0.f, av0 = 0.f, av1 = 0.f;
    int cnt = 0;
    for (int w = hw >> 5; w < nwords; ++w) {                       // ascending order == scatter_add order
        unsigned bits = s_bits[w];
        while (bits) {
            const int j = (w << 5) + __ffs(bits) - 1;
            bits &= bits - 1;
            const float* row = nk + (size_t)j * p.ld_new;
            const float dk = fmaxf(nkn[j], 1e-12f), dv = fmaxf(nvn[j], 1e-12f);
            if (tid < DK) ak += row[tid] / dk;
            av0 += row[p.voff + tid] / dv;
            av1 += row[p.voff + 256 + tid] / dv;
            ++cnt;
        }
    }
    const float c = (float)(cnt < 1 ? 1 : cnt);
    const float r = p.update_rate;
    float* K = p.bank_k + (size_t)obj * p.stride_k + (size_t)tgt * DK;
    float* V = p.bank_v + (size_t)obj * p.stride_v + (size_t)tgt * DV;
    const float magk = p.bank_knorm[(size_t)obj * p.stride_n + tgt];
    const float magv = p.bank_vnorm[(size_t)obj * p.stride_n + tgt];
    const float dk = fmaxf(magk, 1e-12f), dv = fmaxf(magv, 1e-12f);
    if (tid < DK) K[tid] = magk * ((1.f - r) * (K[tid] / dk) + r * (ak / c));
    V[tid] = magv * ((1.f - r) * (V[tid] / dv) + r * (av0 / c));
    V[256 + tid] = magv * ((1.f - r) * (V[256 + tid] / dv) + r * (av1 / c));
}

// ---------------------------------------------------------------- append / evict plan (one block per object)
__global__ __launch_bounds__(1024)
void bank_plan_kernel(const vfn_bank_desc p) {
    __shared__ float redf[16];
    __shared__ int redi[16];
    __shared__ int s_carry;
    const int obj = blockIdx.x, tid = threadIdx.x;
    const float* corr = p.match_corr + (size_t)obj * p.HW;
    int* pos = p.app_pos + (size_t)obj * p.HW;
    int* plan = p.plan + obj * 4;
    const int B = p.bank_len[obj];
    const bool remove_only = p.rm_class >= 0;        // FeatureBank.remove() called on its own
    if (remove_only && obj != p.rm_class) {
        if (tid == 0) { plan[0] = 0; plan[1] = 0; plan[2] = B; plan[3] = 0; }
        return;
    }

    // append set, positions in ascending source order (nonzero())
    int carry = 0;
    for (int base = 0; base < (remove_only ? 0 : p.HW); base += 1024) {
        const int j = base + tid;
        const int f = (j < p.HW) && (corr[j] <= p.thres_close);
        int tot;
        const int off = block_excl_scan(f, redi, &tot);
        if (j < p.HW) pos[j] = f ? carry + off : -1;
        carry += tot;
        __syncthreads();
    }
    const int n_app = remove_only ? p.rm_request : carry;

    int evict = 0, kept = B;
    if (remove_only || (double)p.class_budget < (double)B + (double)n_app) {
        // FeatureBank.remove: LFU = info[:,1] / (frame_idx - info[:,0]); thr = int(min)+1; loop
        evict = 1;
        const float* info = p.info + (size_t)obj * p.stride_info;
        int* dst = p.keep_dst + (size_t)obj * p.stride_n;
        float mn = INFINITY;
        int has_nan = 0;
        for (int b = tid; b < B; b += 1024) {
            const float l = info[(size_t)b * 2 + 1] / ((float)p.frame_idx - info[(size_t)b * 2]);
            has_nan |= (l != l);                    // 0 / 0: an entry born at frame_idx with no hits (fminf would drop it)
            mn = fminf(mn, l);
        }
        mn = block_reduce_min(mn, redf);
        has_nan = block_reduce_sum_i(has_nan, redi);
        // the reference evaluates int(LFU.min()) (FeatureBank.py:123): torch's min propagates NaN -> ValueError, and an
        // all-infinite LFU -> OverflowError, before anything is removed or appended.  Same here: the object is left as
        // it is and the host raises from the statistics block (stats[3] = -1 NaN / -2 infinite).
        if (has_nan || (B > 0 && !(fabsf(mn) < INFINITY))) {
            if (tid == 0) { plan[0] = 0; plan[1] = 0; plan[2] = B; plan[3] = 0; p.stats[obj * 4 + 3] = has_nan ? -1 : -2; }
            return;
        }
        int thr = (int)mn + 1;
        for (int it = 0; it < 100000; ++it) {
            int cnt = 0;
            float mn2 = INFINITY;
            for (int b = tid; b < B; b += 1024) {
                const float l = info[(size_t)b * 2 + 1] / ((float)p.frame_idx - info[(size_t)b * 2]);
                if (l > (float)thr) { ++cnt; mn2 = fminf(mn2, l); }
            }
            kept = block_reduce_sum_i(cnt, redi);
            const double balance = ((double)p.class_budget - (double)kept) - (double)n_app;
            if (balance >= 0.0 || kept == 0) break;
            mn2 = block_reduce_min(mn2, redf);
            thr = (int)mn2 + 1;
        }
        // order-preserving destination map
        int c2 = 0;
        for (int base = 0; base < B; base += 1024) {
            const int b = base + tid;
            int f = 0;
            if (b < B) {
                const float l = info[(size_t)b * 2 + 1] / ((float)p.frame_idx - info[(size_t)b * 2]);
                f = l > (float)thr;
            }
            int tot;
            const int off = block_excl_scan(f, redi, &tot);
            if (b < B) dst[b] = f ? c2 + off : -1;
            c2 += tot;
            __syncthreads();
        }
        kept = c2;
    }
    if (tid == 0) {
        plan[0] = remove_only ? 0 : n_app; plan[1] = evict; plan[2] = kept; plan[3] = B - kept;
        (void)s_carry;
    }
}

// rows that move (dst != src) go to the scratch bank, then back: in-place compaction would race
__global__ void bank_compact_kernel(const vfn_bank_desc p, int phase) {
    const int obj = blockIdx.y;
    const int* plan = p.plan + obj * 4;
    if (!plan[1]) return;
    const int B = p.bank_len[obj];
    const int* dst = p.keep_dst + (size_t)obj * p.stride_n;
    float* K = p.bank_k + (size_t)obj * p.stride_k;
    float* V = p.bank_v + (size_t)obj * p.stride_v;
    float* I = p.info + (size_t)obj * p.stride_info;
    float* Ks = p.scratch_k + (size_t)obj * p.stride_k;
    float* Vs = p.scratch_v + (size_t)obj * p.stride_v;
    float* Is = p.scratch_info + (size_t)obj * p.stride_info;
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int b = blockIdx.x * wpb + (threadIdx.x >> 6); b < B; b += gridDim.x * wpb) {
        const int d = dst[b];
        if (d < 0 || d == b) continue;
        const float* sk = phase == 0 ? K + (size_t)b * DK : Ks + (size_t)d * DK;
        const float* sv = phase == 0 ? V + (size_t)b * DV : Vs + (size_t)d * DV;
        const float* si = phase == 0 ? I + (size_t)b * 2 : Is + (size_t)d * 2;
        float* dk = phase == 0 ? Ks + (size_t)d * DK : K + (size_t)d * DK;
        float* dv = phase == 0 ? Vs + (size_t)d * DV : V + (size_t)d * DV;
        float* di = phase == 0 ? Is + (size_t)d * 2 : I + (size_t)d * 2;
        if (lane < 32) *reinterpret_cast<f32x4*>(dk + lane * 4) = *reinterpret_cast<const f32x4*>(sk + lane * 4);
        *reinterpret_cast<f32x4*>(dv + lane * 4) = *reinterpret_cast<const f32x4*>(sv + lane * 4);
        *reinterpret_cast<f32x4*>(dv + 256 + lane * 4) = *reinterpret_cast<const f32x4*>(sv + 256 + lane * 4);
        if (lane == 0) { di[0] = si[0]; di[1] = si[1]; }
    }
}

__global__ void bank_append_kernel(const vfn_bank_desc p) {
    const int obj = blockIdx.y;
    const int* plan = p.plan + obj * 4;
    const int base = plan[2];                        // length after eviction
    const int* pos = p.app_pos + (size_t)obj * p.HW;
    float* K = p.bank_k + (size_t)obj * p.stride_k;
    float* V = p.bank_v + (size_t)obj * p.stride_v;
    float* I = p.info + (size_t)obj * p.stride_info;
    const float* nk = p.new_k + (size_t)obj * p.stride_new;
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int j = blockIdx.x * wpb + (threadIdx.x >> 6); j < p.HW; j += gridDim.x * wpb) {
        const int q = pos[j];
        if (q < 0) continue;
        const int row = base + q;
        if (row >= p.cap) continue;                  // cannot happen when cap >= budget + HW
        const float* src = nk + (size_t)j * p.ld_new;
        if (lane < 32) *reinterpret_cast<f32x4*>(K + (size_t)row * DK + lane * 4) = *reinterpret_cast<const f32x4*>(src + lane * 4);
        *reinterpret_cast<f32x4*>(V + (size_t)row * DV + lane * 4) = *reinterpret_cast<const f32x4*>(src + p.voff + lane * 4);
        *reinterpret_cast<f32x4*>(V + (size_t)row * DV + 256 + lane * 4) = *reinterpret_cast<const f32x4*>(src + p.voff + 256 + lane * 4);
        if (lane == 0) { I[(size_t)row * 2] = (float)p.frame_idx; I[(size_t)row * 2 + 1] = p.new_hit_init; }
    }
}

// new length, statistics, clamp(info[:,1], 0, 1e5)
__global__ void bank_finalize_kernel(const vfn_bank_desc p) {
    const int obj = blockIdx.y;
    const int* plan = p.plan + obj * 4;
    int newlen = plan[2] + plan[0];
    if (newlen > p.cap) newlen = p.cap;
    float* I = p.info + (size_t)obj * p.stride_info;
    if (p.rm_class < 0)                                // the clamp belongs to update() (FeatureBank.py:115), not to remove()
        for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < newlen; b += gridDim.x * blockDim.x)
            I[(size_t)b * 2 + 1] = fminf(fmaxf(I[(size_t)b * 2 + 1], 0.f), 1e5f);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // stats[obj] = {len, peak_n, replace_n_total, last_n_append}
        int* st = p.stats + obj * 4;
        st[0] = newlen;
        if (newlen > st[1]) st[1] = newlen;
        st[2] += plan[3];
        if (st[3] >= 0 && p.rm_class < 0) st[3] = plan[0];           // (< 0: error flag set by the plan kernel, sticky)
    }
}

__global__ void bank_commit_len_kernel(const vfn_bank_desc p) {
    const int obj = threadIdx.x;
    if (obj < p.obj_n) p.bank_len_rw[obj] = p.stats[obj * 4];
}

// ---------------------------------------------------------------- norms carried across frames
// After an update only the merged entries (new values) and the appended entries (new rows) have norms that differ from
// last frame's; an eviction moves everything.  One wave per row, the same summation as row_norms_kernel, so the carried
// norms equal a full recomputation bit for bit.  Reads plan / app_pos of the update that just ran.
__global__ void bank_refresh_norms_kernel(const vfn_bank_desc p, float* __restrict__ knorm, float* __restrict__ kinv,
                                          float* __restrict__ vnorm) {
    const int obj = blockIdx.y;
    const int* plan = p.plan + obj * 4;
    const bool all = plan[1] != 0;                                // rows were compacted: recompute every norm
    const int newlen = p.stats[obj * 4];
    const int base = plan[2];
    const int items = all ? newlen : p.HW;
    const int* idx = p.match_idx + (size_t)obj * p.HW;
    const float* corr = p.match_corr + (size_t)obj * p.HW;
    const int* pos = p.app_pos + (size_t)obj * p.HW;
    const float* K = p.bank_k + (size_t)obj * p.stride_k;
    const float* V = p.bank_v + (size_t)obj * p.stride_v;
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int it = blockIdx.x * wpb + (threadIdx.x >> 6); it < items; it += gridDim.x * wpb) {
        int row = it;
        if (!all) row = (corr[it] > p.thres_close) ? idx[it] : base + pos[it];
        if (row < 0 || row >= newlen) continue;
        float sk = 0.f, sv = 0.f;
        for (int d = lane * 4; d < DK; d += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(K + (size_t)row * DK + d);
            sk += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        for (int d = lane * 4; d < DV; d += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(V + (size_t)row * DV + d);
            sv += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        sk = wave_sum(sk);
        sv = wave_sum(sv);
        if (lane == 0) {
            const float nk = sqrtf(sk);
            knorm[(size_t)obj * p.stride_n + row] = nk;
            kinv[(size_t)obj * p.stride_n + row] = 1.f / fmaxf(nk, 1e-12f);
            vnorm[(size_t)obj * p.stride_n + row] = sqrtf(sv);
        }
    }
}

// ---------------------------------------------------------------- split-bf16 image of the bank (precision 1 / 2)
// The reduced-precision contractions read every bank entry ~HW/128 times per frame, and splitting an f32 operand into
// bf16 hi + lo in the consuming kernel costs as many vector-ALU cycles as its MFMAs.  The split is kept beside the
// bank instead and re-done only for the entries an update changed (same row walk as bank_refresh_norms_kernel).
// keys:   [row][128 hi | 128 lo] bf16           -- a chunk lands in LDS as the MFMA operand image
// values: blocks of 8 bank rows, [block][hi plane | lo plane][channel 0..511][8 rows] bf16 (a block = 16 KB = its 8 f32 rows'
// bytes): 16 bytes = one channel's 8 consecutive bank rows = one lane's B operand of v_mfma_f32_32x32x16_bf16 in P^T V
// (k = bank rows), 32 lanes = 32 consecutive channels = 512 contiguous bytes.  The apply kernels load their operands straight
// from the image: no transposition in registers, no second register set.
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__global__ void bank_refresh_lp_kernel(const vfn_bank_desc p, char* __restrict__ klp, char* __restrict__ vlp, int all_rows) {
    const int obj = blockIdx.y;
    bool all = all_rows != 0;
    int newlen, base = 0;
    if (all) newlen = p.bank_len[obj];
    else {
        const int* plan = p.plan + obj * 4;
        all = plan[1] != 0;                                       // rows were compacted: every row moved
        newlen = p.stats[obj * 4];
        base = plan[2];
    }
    const int* idx = p.match_idx + (size_t)obj * p.HW;
    const float* corr = p.match_corr + (size_t)obj * p.HW;
    const int* pos = p.app_pos + (size_t)obj * p.HW;
    const float* K = p.bank_k + (size_t)obj * p.stride_k;
    const float* V = p.bank_v + (size_t)obj * p.stride_v;
    char* KL = klp + (size_t)obj * p.stride_k * 4;
    char* VL = vlp + (size_t)obj * p.stride_v * 4;
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    auto split_key_row = [&](int row) {                           // keys: lane l splits k = 2l, 2l+1
        const float2 v = *reinterpret_cast<const float2*>(K + (size_t)row * DK + 2 * lane);
        bf16x2_t h, l;
        h[0] = (__bf16)v.x; h[1] = (__bf16)v.y;
        l[0] = (__bf16)(v.x - (float)h[0]); l[1] = (__bf16)(v.y - (float)h[1]);
        char* dst = KL + (size_t)row * (DK * 4);
        *reinterpret_cast<bf16x2_t*>(dst + 4 * lane) = h;
        *reinterpret_cast<bf16x2_t*>(dst + DK * 2 + 4 * lane) = l;
    };
    if (all) {
        // whole blocks: a wave takes a block of 8 rows, a lane 8 channels (one at a time): 8 row reads of 256 contiguous
        // bytes per wave, 16-byte image stores.  Rows >= newlen of the last block are written as 0.
        const int nblk = (newlen + 7) >> 3;
        for (int blk = blockIdx.x * wpb + (threadIdx.x >> 6); blk < nblk; blk += gridDim.x * wpb) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (blk * 8 + r < newlen) split_key_row(blk * 8 + r);
            char* dst = VL + (size_t)blk * (8 * DV * 4);
            for (int c = lane; c < DV; c += 64) {
                bf16x8_t h, l;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int row = blk * 8 + r;
                    const float x = row < newlen ? V[(size_t)row * DV + c] : 0.f;
                    h[r] = (__bf16)x;
                    l[r] = (__bf16)(x - (float)h[r]);
                }
                *reinterpret_cast<bf16x8_t*>(dst + c * 16) = h;
                *reinterpret_cast<bf16x8_t*>(dst + DV * 16 + c * 16) = l;
            }
        }
        return;
    }
    for (int it = blockIdx.x * wpb + (threadIdx.x >> 6); it < p.HW; it += gridDim.x * wpb) {
        const int row = (corr[it] > p.thres_close) ? idx[it] : base + pos[it];
        if (row < 0 || row >= newlen) continue;
        split_key_row(row);
        char* dst = VL + (size_t)(row >> 3) * (8 * DV * 4) + (row & 7) * 2;
        for (int c = lane; c < DV; c += 64) {                     // one row of a block: 2-byte stores, 16 bytes apart
            const float x = V[(size_t)row * DV + c];
            const __bf16 h = (__bf16)x;
            *reinterpret_cast<__bf16*>(dst + c * 16) = h;
            *reinterpret_cast<__bf16*>(dst + DV * 16 + c * 16) = (__bf16)(x - (float)h);
        }
    }
}

// ---------------------------------------------------------------- scatter_mean operator
// out[d][t] = (out[d][t] + sum_{s: index[s]==t} src[d][s]) / max(count_t, 1), dim = 1, index row-broadcast
// B > 0 (vfn_scatter_mean_checked_f32): the validation torch_scatter does with a device-side assert happens here, without
// a host round trip -- a target outside [0, B) is skipped and sets bit 0 of *status; a materialised [D][S] index
// (index_s0 != 0) whose rows differ sets bit 1.
__global__ __launch_bounds__(256)
void scatter_mean_kernel(const float* __restrict__ src, long long src_s0, long long src_s1,
                         const long long* __restrict__ index, long long index_s0, int S, float* __restrict__ out,
                         long long out_s0, long long out_s1, int D, long long B, int* __restrict__ status) {
    __shared__ int s_any;
    const int s = blockIdx.x, tid = threadIdx.x;
    const long long tgt = index[s];
    if (status) {
        if (index_s0 != 0) {
            int bad = 0;
            for (int d = tid; d < D; d += 256) bad |= (index[d * index_s0 + s] != tgt);
            if (bad) atomicOr(status, 2);
        }
        if (tgt < 0 || tgt >= B) {                       // (uniform over the block)
            if (tid == 0) atomicOr(status, 1);
            return;
        }
    }
    if (tid == 0) s_any = 0;
    __syncthreads();
    int found = 0;
    for (int j = tid; j < s; j += 256) found |= (index[j] == tgt);
    if (found) s_any = 1;
    __syncthreads();
    if (s_any) return;
    for (int d = tid; d < D; d += 256) {
        float acc = out[d * out_s0 + tgt * out_s1];
        int cnt = 0;
        for (int j = s; j < S; ++j)
            if (index[j] == tgt) { acc += src[d * src_s0 + j * src_s1]; ++cnt; }
        out[d * out_s0 + tgt * out_s1] = acc / (float)(cnt < 1 ? 1 : cnt);
    }
}

}  // namespace

extern "C" int vfn_row_norms(const float* x, long long stride_obj, int ld, int dim, const int* len_dev, int rows,
                             int obj_n, float* nrm, float* inv, long long stride_n, void* stream) {
    if (!x || !nrm || dim % 4 || ld % 4 || obj_n < 1) return VFN_ERR_ARG;
    const int blocks = len_dev ? 1024 : (cdiv(rows, 4) < 1024 ? (cdiv(rows, 4) > 0 ? cdiv(rows, 4) : 1) : 1024);
    hipLaunchKernelGGL(row_norms_kernel, dim3(blocks, obj_n), dim3(256), 0, (hipStream_t)stream,
                       x, stride_obj, ld, dim, len_dev, rows, nrm, inv, stride_n);
    return vfn_check_launch();
}

static int bank_desc_ok(const vfn_bank_desc* d) {
    return d && d->bank_k && d->bank_v && d->info && d->bank_len && d->HW > 0 && d->HW <= VFN_BANK_MAX_HW && d->obj_n > 0;
}

extern "C" int vfn_bank_merge(const vfn_bank_desc* d, void* stream) {
    if (!bank_desc_ok(d) || d->rm_class >= 0) return VFN_ERR_ARG;
    if (!d->match_idx || !d->match_corr || !d->new_k || !d->new_knorm || !d->new_vnorm || !d->bank_knorm || !d->bank_vnorm)
        return VFN_ERR_ARG;
    hipLaunchKernelGGL(bank_merge_kernel, dim3(d->HW, d->obj_n), dim3(256), (size_t)((d->HW + 31) / 32) * 4, (hipStream_t)stream, *d);
    return vfn_check_launch();
}

extern "C" int vfn_bank_append(const vfn_bank_desc* d, void* stream) {
    if (!bank_desc_ok(d) || d->rm_class >= 0) return VFN_ERR_ARG;
    if (!d->match_corr || !d->app_pos || !d->plan || !d->keep_dst || !d->new_k || !d->stats || !d->bank_len_rw ||
        !d->scratch_k || !d->scratch_v || !d->scratch_info)
        return VFN_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bank_plan_kernel, dim3(d->obj_n), dim3(1024), 0, s, *d);
    hipLaunchKernelGGL(bank_compact_kernel, dim3(1024, d->obj_n), dim3(256), 0, s, *d, 0);
    hipLaunchKernelGGL(bank_compact_kernel, dim3(1024, d->obj_n), dim3(256), 0, s, *d, 1);
    hipLaunchKernelGGL(bank_append_kernel, dim3(cdiv(d->HW, 4), d->obj_n), dim3(256), 0, s, *d);
    hipLaunchKernelGGL(bank_finalize_kernel, dim3(256, d->obj_n), dim3(256), 0, s, *d);
    hipLaunchKernelGGL(bank_commit_len_kernel, dim3(1), dim3(64), 0, s, *d);
    return vfn_check_launch();
}

extern "C" int vfn_bank_remove(const vfn_bank_desc* d, void* stream) {
    if (!bank_desc_ok(d) || d->rm_class < 0 || d->rm_class >= d->obj_n || d->rm_request < 0) return VFN_ERR_ARG;
    if (!d->app_pos || !d->plan || !d->keep_dst || !d->stats || !d->bank_len_rw || !d->scratch_k || !d->scratch_v ||
        !d->scratch_info || !d->match_corr)
        return VFN_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bank_plan_kernel, dim3(d->obj_n), dim3(1024), 0, s, *d);
    hipLaunchKernelGGL(bank_compact_kernel, dim3(1024, d->obj_n), dim3(256), 0, s, *d, 0);
    hipLaunchKernelGGL(bank_compact_kernel, dim3(1024, d->obj_n), dim3(256), 0, s, *d, 1);
    hipLaunchKernelGGL(bank_finalize_kernel, dim3(256, d->obj_n), dim3(256), 0, s, *d);
    hipLaunchKernelGGL(bank_commit_len_kernel, dim3(1), dim3(64), 0, s, *d);
    return vfn_check_launch();
}

extern "C" int vfn_scatter_mean_f32(const float* src, long long src_s0, long long src_s1, const long long* index,
                                    int S, float* out, long long out_s0, long long out_s1, int D, void* stream) {
    if (!src || !index || !out || S < 0 || D < 1) return VFN_ERR_ARG;
    if (S == 0) return VFN_OK;
    hipLaunchKernelGGL(scatter_mean_kernel, dim3(S), dim3(256), 0, (hipStream_t)stream,
                       src, src_s0, src_s1, index, 0LL, S, out, out_s0, out_s1, D, 0LL, (int*)nullptr);
    return vfn_check_launch();
}

extern "C" int vfn_scatter_mean_checked_f32(const float* src, long long src_s0, long long src_s1, const long long* index,
                                            long long index_s0, int S, float* out, long long out_s0, long long out_s1, int D,
                                            long long B, int* status, void* stream) {
    if (!src || !index || !out || !status || S < 0 || D < 1 || B < 1) return VFN_ERR_ARG;
    if (S == 0) return VFN_OK;
    hipLaunchKernelGGL(scatter_mean_kernel, dim3(S), dim3(256), 0, (hipStream_t)stream,
                       src, src_s0, src_s1, index, index_s0, S, out, out_s0, out_s1, D, B, status);
    return vfn_check_launch();
}

extern "C" int vfn_bank_refresh_norms(const vfn_bank_desc* d, float* bank_knorm, float* bank_kinv, float* bank_vnorm,
                                      void* stream) {
    if (!bank_desc_ok(d) || !d->plan || !d->stats || !d->match_idx || !d->match_corr || !d->app_pos) return VFN_ERR_ARG;
    if (!bank_knorm || !bank_kinv || !bank_vnorm) return VFN_ERR_ARG;
    hipLaunchKernelGGL(bank_refresh_norms_kernel, dim3(1024, d->obj_n), dim3(256), 0, (hipStream_t)stream, *d,
                       bank_knorm, bank_kinv, bank_vnorm);
    return vfn_check_launch();
}

extern "C" int vfn_bank_refresh_lp(const vfn_bank_desc* d, void* bank_k_lp, void* bank_v_lp, int all_rows, void* stream) {
    if (!bank_desc_ok(d) || !bank_k_lp || !bank_v_lp) return VFN_ERR_ARG;
    if (!all_rows && (!d->plan || !d->stats || !d->match_idx || !d->match_corr || !d->app_pos)) return VFN_ERR_ARG;
    hipLaunchKernelGGL(bank_refresh_lp_kernel, dim3(all_rows ? 4096 : 1024, d->obj_n), dim3(256), 0, (hipStream_t)stream, *d,
                       (char*)bank_k_lp, (char*)bank_v_lp, all_rows);
    return vfn_check_launch();
}
