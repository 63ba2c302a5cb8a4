// Host-only connected-component filter (myutils/data.py:17-37 runs cv2 on the CPU here too): plain C++, no HIP.
// Included by loop_ops.hip (-> libvfn_hip.so) and by host/host_only.cpp (-> the sanitizer build, `make asan`).
#pragma once
#include <stddef.h>
#include <vector>

namespace vfn_host {

// ---- host-side connected components (the reference runs cv2 on the CPU here too)
inline int uf_find(std::vector<int>& p, int x) {
    while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; }
    return x;
}


// Host buffers.  8-connected labelling of non-zero pixels, labels numbered in raster order of first
// pixel (OpenCV order); output = (labels == largest component) with the reference's special cases.
inline int postprocess_pred_u8(const unsigned char* pred, int H, int W, unsigned char* out) {
    if (!pred || !out || H < 1 || W < 1) return -1;
    const size_t n = (size_t)H * W;
    std::vector<int> lab(n, 0), parent(1, 0);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t i = (size_t)y * W + x;
            if (!pred[i]) continue;
            int best = 0;
            const int nb[4][2] = {{-1, -1}, {-1, 0}, {-1, 1}, {0, -1}};
            for (auto& d : nb) {
                const int yy = y + d[0], xx = x + d[1];
                if (yy < 0 || xx < 0 || xx >= W) continue;
                const int l = lab[(size_t)yy * W + xx];
                if (!l) continue;
                const int r = uf_find(parent, l);
                if (!best) best = r;
                else if (r != best) { const int a = best < r ? best : r, b = best < r ? r : best; parent[b] = a; best = a; }
            }
            if (!best) { best = (int)parent.size(); parent.push_back(best); }
            lab[i] = best;
        }
    // compact roots in raster order of first appearance
    std::vector<int> remap(parent.size(), 0);
    std::vector<long long> count(1, 0);
    int next = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!lab[i]) continue;
        const int r = uf_find(parent, lab[i]);
        if (!remap[r]) { remap[r] = ++next; count.push_back(0); }
        lab[i] = remap[r];
        ++count[lab[i]];
    }
    const int label_cnt = next + 1;
    if (label_cnt == 2) {
        // single component: labels is 0/1; the reference returns labels, or 1-labels when
        // labels[0,0] != pred[0,0] (only possible for pred values > 1)
        const bool same = (lab[0] == (int)pred[0]);
        for (size_t i = 0; i < n; ++i) out[i] = (unsigned char)(same ? lab[i] : 1 - lab[i]);
        return 0;
    }
    long long max_cnt = 0;
    int max_label = 0;
    for (int l = 1; l < label_cnt; ++l)
        if (count[l] > max_cnt) { max_cnt = count[l]; max_label = l; }
    for (size_t i = 0; i < n; ++i) out[i] = (unsigned char)(lab[i] == max_label);   // no component -> all ones
    return 0;
}

}  // namespace vfn_host
