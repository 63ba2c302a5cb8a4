// The host-only C++ of libvfn_hip.so compiled for the CPU alone, with AddressSanitizer + UBSan (`make asan`):
// the JPEG marker / Huffman decoder (runs in DataLoader workers on user-supplied files) and the host connected-component
// filter.  GPU sanitizers are not available on the target pool, and these two are the only parts of the library that walk
// untrusted bytes on the host.  With -DVFN_FUZZ_MAIN the file is a stand-alone fuzz-style driver (tests/test_sanitize.py):
//
//   vfn_host_fuzz FILE.jpg...   for every file: the intact file must decode (rc 0); then every truncation at a marker
//                               boundary and inside every marker segment, every single-bit flip inside the DHT / DQT /
//                               SOF / SOS / DRI segments, and pseudo-random byte flips in the entropy-coded data are
//                               decoded from an exactly-sized heap copy (so that any over-read trips ASan).
//                               Any return code is acceptable for a damaged file; a sanitizer report is not.
#include "jpeg_entropy.h"
#include "ccl_host.h"

extern "C" int vfn_jpeg_entropy_decode(const unsigned char* data, long long size, short* coef, long long coef_cap,
                                       unsigned short* qt, int* info) {
    return vfn_host::jpeg_entropy_decode(data, size, coef, coef_cap, qt, info);
}
extern "C" int vfn_postprocess_pred_u8(const unsigned char* pred, int H, int W, unsigned char* out) {
    return vfn_host::postprocess_pred_u8(pred, H, W, out);
}

#ifdef VFN_FUZZ_MAIN
#include <stdio.h>
#include <stdlib.h>
#include <vector>

namespace {

struct Seg { long long off, len; int marker; };       // marker segment: [off, off + 2 + len) incl. the FF xx

std::vector<unsigned char> read_file(const char* path) {
    std::vector<unsigned char> v;
    FILE* f = fopen(path, "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize(n > 0 ? n : 0);
    if (n > 0 && fread(v.data(), 1, n, f) != (size_t)n) v.clear();
    fclose(f);
    return v;
}

// decode from an exactly-sized heap copy; returns the decoder's code
int decode_copy(const unsigned char* data, long long size, long long* need = nullptr) {
    unsigned char* buf = (unsigned char*)malloc(size > 0 ? size : 1);
    if (size > 0) memcpy(buf, data, size);
    unsigned short qt[4 * 64];
    int info[24];
    memset(info, 0, sizeof(info));
    int rc = vfn_jpeg_entropy_decode(buf, size, nullptr, 0, qt, info);
    if (rc == -3) {
        const long long total = info[23];
        if (need) *need = total;
        short* coef = (short*)malloc((total > 0 ? total : 1) * sizeof(short));
        rc = vfn_jpeg_entropy_decode(buf, size, coef, total, qt, info);
        free(coef);
    }
    free(buf);
    return rc;
}

unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

}  // namespace

int main(int argc, char** argv) {
    long long cases = 0, ok_cases = 0;
    for (int a = 1; a < argc; ++a) {
        const std::vector<unsigned char> f = read_file(argv[a]);
        if (f.size() < 4) { fprintf(stderr, "cannot read %s\n", argv[a]); return 2; }
        const long long n = (long long)f.size();
        if (decode_copy(f.data(), n) != 0) { fprintf(stderr, "%s: intact file does not decode\n", argv[a]); return 3; }
        // marker walk (the file is known good)
        std::vector<Seg> segs;
        long long p = 2, scan = n;
        while (p + 4 <= n && f[p] == 0xFF) {
            const int m = f[p + 1];
            const long long len = (f[p + 2] << 8) | f[p + 3];
            segs.push_back({p, len, m});
            p += 2 + len;
            if (m == 0xDA) { scan = p; break; }
        }
        std::vector<unsigned char> w(f);
        // 1. truncations: at every byte of the header, then 64 cuts through the entropy-coded data
        for (long long cut = 0; cut <= scan && cut <= n; ++cut, ++cases) ok_cases += decode_copy(f.data(), cut) == 0;
        for (int i = 1; i <= 64; ++i, ++cases) ok_cases += decode_copy(f.data(), scan + (n - scan) * i / 65) == 0;
        // 2. every single-bit flip inside the table / frame / scan headers (also the length fields)
        for (const Seg& s : segs) {
            if (!(s.marker == 0xC4 || s.marker == 0xDB || s.marker == 0xC0 || s.marker == 0xC1 || s.marker == 0xDA || s.marker == 0xDD))
                continue;
            for (long long i = s.off; i < s.off + 2 + s.len && i < n; ++i)
                for (int b = 0; b < 8; ++b, ++cases) {
                    w[i] ^= (unsigned char)(1 << b);
                    ok_cases += decode_copy(w.data(), n) == 0;
                    w[i] ^= (unsigned char)(1 << b);
                }
        }
        // 3. pseudo-random damage to the entropy-coded data (1..8 bytes per case)
        unsigned seed = 12345u + (unsigned)a;
        for (int it = 0; it < 400 && scan < n; ++it, ++cases) {
            const int k = 1 + lcg(seed) % 8;
            long long pos[8];
            unsigned char old[8];
            for (int j = 0; j < k; ++j) {
                pos[j] = scan + lcg(seed) % (n - scan);
                old[j] = w[pos[j]];
                w[pos[j]] = (unsigned char)lcg(seed);
            }
            ok_cases += decode_copy(w.data(), n) == 0;
            for (int j = k - 1; j >= 0; --j) w[pos[j]] = old[j];
        }
    }
    // connected components: random masks of awkward shapes, exactly-sized buffers
    unsigned seed = 777u;
    for (int it = 0; it < 200; ++it, ++cases) {
        const int H = 1 + lcg(seed) % 37, W = 1 + lcg(seed) % 41;
        unsigned char* in = (unsigned char*)malloc((size_t)H * W);
        unsigned char* out = (unsigned char*)malloc((size_t)H * W);
        const unsigned dens = lcg(seed) % 4;
        for (int i = 0; i < H * W; ++i) in[i] = (lcg(seed) % 4) <= dens ? (unsigned char)(1 + lcg(seed) % 3) : 0;
        ok_cases += vfn_postprocess_pred_u8(in, H, W, out) == 0;
        free(in);
        free(out);
    }
    printf("vfn_host_fuzz: %lld cases, %lld decoded, no sanitizer report\n", cases, ok_cases);
    return 0;
}
#endif
