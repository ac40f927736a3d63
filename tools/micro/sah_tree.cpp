// Experiment: a surface-area-heuristic tree over given leaf boxes, written in the layout of pt_bvh_node
// (pre-order, inner link = 2 * second child + 1, leaf link = 2 * surface).  Used to measure how much a better
// tree would be worth to the nearer-child-first walk before building one on the device.
//   g++ -O2 -shared -fPIC -o tools/micro/libsah_tree.so tools/micro/sah_tree.cpp
#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {
struct Box {
    float lo[3], hi[3];
    void clear() { for (int a = 0; a < 3; a++) { lo[a] = FLT_MAX; hi[a] = -FLT_MAX; } }
    void add(const Box& b) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    float area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};
struct Node { float lo[3], hi[3]; uint64_t link; };

struct Builder {
    const Box* leaf;
    Node* out;
    uint64_t next = 0;
    int sweepBelow;
    int B = 64;
    bool boxDomain = false;
    std::vector<float> rightArea;

    uint64_t build(uint32_t* idx, uint32_t n) {
        uint64_t me = next++;
        if (n == 1) {
            std::memcpy(out[me].lo, leaf[idx[0]].lo, 24);
            out[me].link = 2ull * idx[0];
            return me;
        }
        Box all, cen;
        all.clear(); cen.clear();
        for (uint32_t i = 0; i < n; i++) {
            const Box& b = leaf[idx[i]];
            all.add(b);
            Box c;
            for (int a = 0; a < 3; a++) c.lo[a] = c.hi[a] = 0.5f * (b.lo[a] + b.hi[a]);
            cen.add(c);
        }
        uint32_t mid = n / 2;
        int bestAxis = -1;
        if (n <= (uint32_t)sweepBelow) {
            float best = FLT_MAX; uint32_t bestAt = 0;
            for (int a = 0; a < 3; a++) {
                if (cen.hi[a] <= cen.lo[a]) continue;
                std::sort(idx, idx + n, [&](uint32_t x, uint32_t y) {
                    return leaf[x].lo[a] + leaf[x].hi[a] < leaf[y].lo[a] + leaf[y].hi[a]; });
                Box r; r.clear();
                for (uint32_t i = n - 1; i > 0; i--) { r.add(leaf[idx[i]]); rightArea[i] = r.area(); }
                Box l; l.clear();
                for (uint32_t i = 1; i < n; i++) {
                    l.add(leaf[idx[i - 1]]);
                    float c = l.area() * i + rightArea[i] * (n - i);
                    if (c < best) { best = c; bestAt = i; bestAxis = a; }
                }
            }
            if (bestAxis >= 0) {
                int a = bestAxis;
                std::sort(idx, idx + n, [&](uint32_t x, uint32_t y) {
                    return leaf[x].lo[a] + leaf[x].hi[a] < leaf[y].lo[a] + leaf[y].hi[a]; });
                mid = bestAt;
            }
        } else {
            if (boxDomain) cen = all;
            float best = FLT_MAX; int bestBin = 0;
            for (int a = 0; a < 3; a++) {
                float ext = cen.hi[a] - cen.lo[a];
                if (!(ext > 0)) continue;
                Box bins[64]; uint32_t cnt[64];
                for (int b = 0; b < B; b++) { bins[b].clear(); cnt[b] = 0; }
                float k = B / ext;
                for (uint32_t i = 0; i < n; i++) {
                    const Box& bx = leaf[idx[i]];
                    int b = std::min(B - 1, (int)((0.5f * (bx.lo[a] + bx.hi[a]) - cen.lo[a]) * k));
                    bins[b].add(bx); cnt[b]++;
                }
                float ra[64]; uint32_t rc[64];
                Box r; r.clear(); uint32_t c = 0;
                for (int b = B - 1; b > 0; b--) { r.add(bins[b]); c += cnt[b]; ra[b] = r.area(); rc[b] = c; }
                Box l; l.clear(); c = 0;
                for (int b = 1; b < B; b++) {
                    l.add(bins[b - 1]); c += cnt[b - 1];
                    if (c == 0 || rc[b] == 0) continue;
                    float cost = l.area() * c + ra[b] * rc[b];
                    if (cost < best) { best = cost; bestAxis = a; bestBin = b; }
                }
            }
            if (bestAxis >= 0) {
                int a = bestAxis;
                float k = B / (cen.hi[a] - cen.lo[a]);
                uint32_t* m = std::partition(idx, idx + n, [&](uint32_t x) {
                    int b = std::min(B - 1, (int)((0.5f * (leaf[x].lo[a] + leaf[x].hi[a]) - cen.lo[a]) * k));
                    return b < bestBin; });
                mid = (uint32_t)(m - idx);
            }
        }
        if (mid == 0 || mid == n) mid = n / 2;
        build(idx, mid);
        uint64_t second = build(idx + mid, n - mid);
        std::memcpy(out[me].lo, all.lo, 24);
        out[me].link = 2 * second + 1;
        return me;
    }
};
}  // namespace

extern "C" int sah_tree2(const float* leafBoxes, uint32_t n, int sweepBelow, int bins, int boxDomain, void* outNodes);
extern "C" int sah_tree(const float* leafBoxes, uint32_t n, int sweepBelow, void* outNodes) {
    return sah_tree2(leafBoxes, n, sweepBelow, 64, 0, outNodes);
}
extern "C" int sah_tree2(const float* leafBoxes, uint32_t n, int sweepBelow, int bins, int boxDomain, void* outNodes) {
    std::vector<uint32_t> idx(n);
    for (uint32_t i = 0; i < n; i++) idx[i] = i;
    Builder b;
    b.leaf = reinterpret_cast<const Box*>(leafBoxes);
    b.out = reinterpret_cast<Node*>(outNodes);
    b.sweepBelow = sweepBelow;
    b.B = bins;
    b.boxDomain = boxDomain != 0;
    b.rightArea.resize(std::max<uint32_t>(n, 2));
    b.build(idx.data(), n);
    return 0;
}

// Locally-ordered agglomerative clustering (Meister and Bittner 2018) on the host, to measure the quality of its
// trees: clusters in Morton order, every cluster looks `radius` places to either side for the partner with the
// smallest merged box, mutual choices merge, the rest carry on.
namespace {
uint64_t spread21(uint64_t x) {
    x &= 0x1FFFFFull;
    x = (x | (x << 32)) & 0x1F00000000FFFFull;
    x = (x | (x << 16)) & 0x1F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}
struct BNode { Box box; uint32_t child[2]; uint32_t leaves; };  // child[0] == ~0u: leaf, child[1] = surface
uint64_t emit(const std::vector<BNode>& b, uint32_t node, Node* out, uint64_t& next) {
    // iterative pre-order
    struct Item { uint32_t node; uint64_t parentAt; };
    std::vector<Item> stack;
    stack.push_back({ node, ~0ull });
    while (!stack.empty()) {
        Item it = stack.back(); stack.pop_back();
        uint64_t me = next++;
        if (it.parentAt != ~0ull) out[it.parentAt].link = 2 * me + 1;
        const BNode& n = b[it.node];
        std::memcpy(out[me].lo, n.box.lo, 24);
        if (n.child[0] == ~0u) { out[me].link = 2ull * n.child[1]; continue; }
        stack.push_back({ n.child[1], me });   // second child: its place is written into my link when reached
        stack.push_back({ n.child[0], ~0ull }); // first child comes next
    }
    return 0;
}
}  // namespace

extern "C" int ploc_tree(const float* leafBoxes, uint32_t n, int radius, void* outNodes) {
    const Box* leaf = reinterpret_cast<const Box*>(leafBoxes);
    Box cen; cen.clear();
    for (uint32_t i = 0; i < n; i++) {
        Box c;
        for (int a = 0; a < 3; a++) c.lo[a] = c.hi[a] = 0.5f * (leaf[i].lo[a] + leaf[i].hi[a]);
        cen.add(c);
    }
    std::vector<std::pair<uint64_t, uint32_t>> keys(n);
    for (uint32_t i = 0; i < n; i++) {
        uint64_t q[3];
        for (int a = 0; a < 3; a++) {
            float ext = cen.hi[a] - cen.lo[a];
            float f = ext > 0 ? (0.5f * (leaf[i].lo[a] + leaf[i].hi[a]) - cen.lo[a]) / ext : 0.0f;
            f = std::min(std::max(f, 0.0f), 1.0f);
            q[a] = (uint64_t)(f * 2097151.0f);
        }
        keys[i] = { (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]), i };
    }
    std::sort(keys.begin(), keys.end());
    std::vector<BNode> b;
    b.reserve(2 * size_t(n));
    std::vector<uint32_t> cur(n), nxt, nn(n);
    for (uint32_t i = 0; i < n; i++) {
        BNode x; x.box = leaf[keys[i].second]; x.child[0] = ~0u; x.child[1] = keys[i].second; x.leaves = 1;
        b.push_back(x);
        cur[i] = i;
    }
    while (cur.size() > 1) {
        const int m = (int)cur.size();
        for (int i = 0; i < m; i++) {
            float best = FLT_MAX; int bestJ = -1;
            for (int j = std::max(0, i - radius); j <= std::min(m - 1, i + radius); j++) {
                if (j == i) continue;
                Box u = b[cur[i]].box; u.add(b[cur[j]].box);
                float ar = u.area();
                if (ar < best) { best = ar; bestJ = j; }
            }
            nn[i] = bestJ;
        }
        nxt.clear();
        for (int i = 0; i < m; i++) {
            int j = nn[i];
            if ((int)nn[j] == i) {
                if (i < j) {
                    BNode x; x.box = b[cur[i]].box; x.box.add(b[cur[j]].box);
                    x.child[0] = cur[i]; x.child[1] = cur[j]; x.leaves = b[cur[i]].leaves + b[cur[j]].leaves;
                    b.push_back(x);
                    nxt.push_back((uint32_t)b.size() - 1);
                }
            } else {
                nxt.push_back(cur[i]);
            }
        }
        cur.swap(nxt);
    }
    uint64_t next = 0;
    emit(b, cur[0], reinterpret_cast<Node*>(outNodes), next);
    return next == 2ull * n - 1 ? 0 : 1;
}
