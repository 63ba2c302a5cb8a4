// PNG frames on the input side: what is left of decoding a PNG once a host core has inflated the
// IDAT stream -- undoing the scanline filters and turning the pixels into the tensor Video_DS hands out
// (video_module/dataset/Water_DS.py:105-109: Image.open(path).convert('RGB') -> ToTensor).
//
//   vfn_png_unfilter_u8    PNG filter types 0-4 (None, Sub, Up, Average, Paeth; PNG spec 9.2) of an 8-bit,
//                          non-interlaced image with 1-4 bytes per pixel
//   vfn_png_to_tensor_f32  colour types 0 / 2 / 3 / 4 / 6 -> RGB as PIL's convert('RGB') does (grey replicated, palette
//                          looked up, alpha dropped), then ToTensor: float32 [3][H][W] = u8 / 255
//
// The filters are recurrences: a byte depends on its left, upper and upper-left neighbours (Average and Paeth are not
// linear, so no scan applies).  The dependency front is an anti-diagonal of 4-pixel blocks: thread j owns block column
// j and walks down the rows one step behind thread j-1.  The row above stays in the thread's registers, the only
// exchange is the last pixel of the left neighbour's block (4 bytes through LDS, one barrier per step).  H + W/4 steps;
// 480p: 694 steps of ~300 byte-wise VALU instructions on 4 waves, one CU.
#include "common.h"
#include "../../include/vfn_hip.h"

namespace {

// filtered scanlines [H][1 + rowbytes] (filter-type byte first) -> filter types [H] and rows re-pitched to whole blocks
// (pitch = nblk * 4 * bpp bytes, dword aligned, zero padded), so that the serial kernel moves whole dwords
__global__ void png_realign_kernel(const unsigned char* __restrict__ f, int rowbytes, int H, int pitch,
                                   unsigned char* __restrict__ ftype, unsigned* __restrict__ al) {
    const int pw = pitch >> 2;
    const long long total = (long long)H * pw;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / pw), j = (int)(i - (long long)r * pw);
        const unsigned char* src = f + (size_t)r * (rowbytes + 1) + 1 + 4 * j;
        unsigned v = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * j + e < rowbytes) v |= (unsigned)src[e] << (8 * e);
        al[i] = v;
        if (j == 0) ftype[r] = f[(size_t)r * (rowbytes + 1)];
    }
}

__device__ __forceinline__ int paeth(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

template <int BPP>
__global__ __launch_bounds__(1024)
void png_unfilter_kernel(const unsigned* __restrict__ al, const unsigned char* __restrict__ ftype, int H, int nblk, int pitch,
                         unsigned* __restrict__ raw, int* __restrict__ status) {
    constexpr int BB = 4 * BPP;                      // bytes per block (4 pixels) = BPP dwords
    __shared__ unsigned lastpx[2][1024];             // last pixel of every thread's block, two steps deep
    const int j = threadIdx.x;
    const int pw = pitch >> 2;
    unsigned char above[BB], cur[BB];
#pragma unroll
    for (int i = 0; i < BB; ++i) above[i] = 0;
    unsigned left_prev = 0;                          // last pixel of the left block one row up (= upper-left pixel)
    int bad = 0;
    const int steps = H + nblk - 1;
    // this thread's rows come one per step from step j on; the filtered block and the filter type of the NEXT row are
    // requested a step ahead, so that no step waits for global memory
    unsigned wn[BPP];
    int ftn = 0;
    auto fetch = [&](int r) {
        const bool ok = j < nblk && r >= 0 && r < H;
#pragma unroll
        for (int d = 0; d < BPP; ++d) wn[d] = ok ? al[(size_t)r * pw + j * BPP + d] : 0u;
        ftn = ok ? ftype[r] : 0;
    };
    fetch(0 - j);
    for (int s = 0; s < steps; ++s) {
        const int r = s - j;
        const bool live = j < nblk && r >= 0 && r < H;
        unsigned w[BPP];
#pragma unroll
        for (int d = 0; d < BPP; ++d) w[d] = wn[d];
        const int ft = ftn;
        fetch(r + 1);
        unsigned mine = 0;
        if (live) {
            const unsigned left = j > 0 ? lastpx[(s + 1) & 1][j - 1] : 0u;       // written at step s-1
            if (ft > 4) bad = 1;
            const int m1 = -(ft == 1), m2 = -(ft == 2), m3 = -(ft == 3), m4 = -(ft == 4);
#pragma unroll
            for (int i = 0; i < BB; ++i) {
                const int x = (w[i >> 2] >> (8 * (i & 3))) & 0xff;
                const int a = i < BPP ? (int)((left >> (8 * i)) & 0xff) : (int)cur[i - BPP];
                const int b = above[i];
                const int c = i < BPP ? (int)((left_prev >> (8 * i)) & 0xff) : (int)above[i - BPP];
                // all four predictors, selected with masks: the lanes of a wave sit on different rows, so a branch on
                // the filter type would run every arm one after the other for every byte
                const int pred = (a & m1) | (b & m2) | (((a + b) >> 1) & m3) | (paeth(a, b, c) & m4);
                cur[i] = (unsigned char)(x + pred);
            }
            unsigned o[BPP];
#pragma unroll
            for (int d = 0; d < BPP; ++d) o[d] = 0;
#pragma unroll
            for (int i = 0; i < BB; ++i) o[i >> 2] |= (unsigned)cur[i] << (8 * (i & 3));
#pragma unroll
            for (int d = 0; d < BPP; ++d) raw[(size_t)r * pw + j * BPP + d] = o[d];
#pragma unroll
            for (int i = 0; i < BPP; ++i) mine |= (unsigned)cur[BB - BPP + i] << (8 * i);
#pragma unroll
            for (int i = 0; i < BB; ++i) above[i] = cur[i];
            left_prev = left;
        }
        lastpx[s & 1][j] = mine;
        __syncthreads();
    }
    if (bad) *status = 1;                            // a filter-type byte outside 0..4: corrupt stream
}

// raw pixels [H][pitch] -> float [3][H][W] (= u8 / 255, torchvision ToTensor) and optionally u8 [H][W][3]
__global__ void png_to_tensor_kernel(const unsigned char* __restrict__ raw, int pitch, int W, int H, int bpp, int ctype,
                                     const unsigned char* __restrict__ pal, float* __restrict__ out, unsigned char* __restrict__ out_u8) {
    const long long total = (long long)H * W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long long)y * W);
        const unsigned char* px = raw + (size_t)y * pitch + (size_t)x * bpp;
        int r, g, b;
        if (ctype == 2 || ctype == 6) { r = px[0]; g = px[1]; b = px[2]; }
        else if (ctype == 3) { const unsigned char* e = pal + 3 * px[0]; r = e[0]; g = e[1]; b = e[2]; }
        else { r = g = b = px[0]; }
        out[i] = (float)r / 255.f;
        out[total + i] = (float)g / 255.f;
        out[2 * total + i] = (float)b / 255.f;
        if (out_u8) { out_u8[3 * i] = r; out_u8[3 * i + 1] = g; out_u8[3 * i + 2] = b; }
    }
}

}  // namespace

extern "C" int vfn_png_unfilter_sizes(int width, int height, int bpp, int* pitch, long long* work_bytes) {
    if (width < 1 || height < 1 || bpp < 1 || bpp > 4 || width > 4096) return VFN_ERR_ARG;
    const int nblk = (width + 3) / 4;
    if (pitch) *pitch = nblk * 4 * bpp;
    // work: re-pitched filtered rows + filter types (rounded up to 16 bytes) + one status int
    if (work_bytes) *work_bytes = (long long)height * nblk * 4 * bpp + ((height + 15) / 16) * 16 + 16;
    return VFN_OK;
}

extern "C" int vfn_png_unfilter_u8(const unsigned char* filtered, int width, int height, int bpp, void* work,
                                   unsigned char* raw, int* status, void* stream) {
    int pitch; long long wb;
    if (!filtered || !work || !raw || !status || vfn_png_unfilter_sizes(width, height, bpp, &pitch, &wb) != VFN_OK) return VFN_ERR_ARG;
    const int nblk = pitch / (4 * bpp);
    unsigned* al = reinterpret_cast<unsigned*>(work);
    unsigned char* ftype = reinterpret_cast<unsigned char*>(work) + (size_t)height * pitch;
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)height * (pitch / 4);
    hipLaunchKernelGGL(png_realign_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, s,
                       filtered, width * bpp, height, pitch, ftype, al);
    const int nt = ((nblk + 63) / 64) * 64;
    unsigned* rw = reinterpret_cast<unsigned*>(raw);
    switch (bpp) {
        case 1: hipLaunchKernelGGL(png_unfilter_kernel<1>, dim3(1), dim3(nt), 0, s, al, ftype, height, nblk, pitch, rw, status); break;
        case 2: hipLaunchKernelGGL(png_unfilter_kernel<2>, dim3(1), dim3(nt), 0, s, al, ftype, height, nblk, pitch, rw, status); break;
        case 3: hipLaunchKernelGGL(png_unfilter_kernel<3>, dim3(1), dim3(nt), 0, s, al, ftype, height, nblk, pitch, rw, status); break;
        default: hipLaunchKernelGGL(png_unfilter_kernel<4>, dim3(1), dim3(nt), 0, s, al, ftype, height, nblk, pitch, rw, status); break;
    }
    return vfn_check_launch();
}

extern "C" int vfn_png_to_tensor_f32(const unsigned char* raw, int pitch, int width, int height, int color_type,
                                     const unsigned char* palette, float* out, unsigned char* out_u8, void* stream) {
    if (!raw || !out || width < 1 || height < 1) return VFN_ERR_ARG;
    int bpp;
    switch (color_type) {
        case 0: bpp = 1; break;
        case 2: bpp = 3; break;
        case 3: bpp = 1; if (!palette) return VFN_ERR_ARG; break;
        case 4: bpp = 2; break;
        case 6: bpp = 4; break;
        default: return VFN_ERR_ARG;
    }
    if (pitch < width * bpp) return VFN_ERR_ARG;
    const long long total = (long long)width * height;
    hipLaunchKernelGGL(png_to_tensor_kernel, dim3((unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192)), dim3(256), 0,
                       (hipStream_t)stream, raw, pitch, width, height, bpp, color_type, palette, out, out_u8);
    return vfn_check_launch();
}
