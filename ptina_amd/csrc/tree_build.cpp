// tree_build.cpp -- BVH construction behind mpt_build_tree: the reference LBVH on the host, its
// SAH re-partition for the fast build, and the glue around the device build (lbvh_build.hip).

#include "miptina_ctx.h"
#include "tri_records.h"
#include <atomic>
#include <system_error>
#include <thread>

// phase clock of mpt_build_tree (option "build_phases": synchronise at every mark, so a phase's time is its device work too)
struct BuildClock {
    mpt_ctx *c;
    std::chrono::steady_clock::time_point t0, t;
    explicit BuildClock(mpt_ctx *ctx) : c(ctx) {
        for (double &v : c->build_phase_us) v = 0.0;
        t0 = t = std::chrono::steady_clock::now();
    }
    void mark(int k) {
        if (c->build_phases) (void)hipStreamSynchronize(c->stream);
        const auto now = std::chrono::steady_clock::now();
        c->build_phase_us[k] += std::chrono::duration<double, std::micro>(now - t).count();
        c->build_phase_us[5] = std::chrono::duration<double, std::micro>(now - t0).count();
        t = now;
    }
};

// ------------------------------------------------------------------ LBVH build (tree/lbvh.py:169-305)
// Same algorithm as the reference (30-bit Morton codes of centroids, sorted, Karras hierarchy,
// bottom-up boxes) with two robustness changes: the sort key is (code << 32 | index), so equal
// codes cannot corrupt the hierarchy, and the boxes are fitted in one post-order
// pass instead of <=64 level-synchronous launches with a read-back each (lbvh.py:251-261).
// With distinct codes the tree is node-for-node the reference's.

static inline uint32_t expand_bits(uint32_t v) {                               // lbvh.py:13-17
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

static inline int quant1024(float x) {                                         // clamp(ifloor(v * 1024), 0, 1023), lbvh.py:29
    float f = floorf(x * 1024.0f);
    if (!(f == f) || f < 0.f) return 0;
    if (f > 1023.f) return 1023;
    return (int)f;
}

static inline int delta(const std::vector<uint64_t> &key, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __builtin_clzll(key[i] ^ key[j]);
}

// ------------------------------------------------------------------ SAH re-partition (fast build only)
// The image does not depend on the tree's shape (only on which of two equal-depth hits wins), so the
// production traversal is free to walk a better tree over the SAME leaf slots: measured on the
// 978-triangle benchmark scene a full-sweep SAH partition needs 11.0 node fetches per ray where the
// LBVH needs 23.7.  Leaves stay single triangles (the node record is unchanged); nodes are numbered
// in DFS pre-order, root 0.  Exact sweep for ranges <= 8192 leaves, 32-bin SAH above.
struct SahBuild {
    int n = 0;
    std::vector<float> lo, hi, ctr;            // per leaf slot [n][3]
    std::vector<int> idx;                      // leaf slots, partitioned in place
    std::vector<int32_t> child;                // [n-1][2]: >= 0 internal, ~slot leaf
    std::vector<float> blo, bhi;               // per internal node [n-1][3]: its own box
    int depth = 0;
    int exact_max = 8192;                      // ranges up to this many leaves: every split of every axis (exact sweep); above: 32 bins

    static float half_area(const float *l, const float *h) {
        float dx = std::max(h[0] - l[0], 0.f), dy = std::max(h[1] - l[1], 0.f), dz = std::max(h[2] - l[2], 0.f);
        return dx * dy + dy * dz + dz * dx;
    }

    int split(int b, int e) {                  // returns m in (b, e): [b, m) | [m, e)
        const int cnt = e - b;
        float best = INFINITY;
        int best_axis = -1, best_k = -1;
        float best_pos = 0.f;
        bool binned = cnt > exact_max;
        std::vector<std::pair<float, int>> key(binned ? 0 : cnt);
        std::vector<float> rarea(binned ? 0 : cnt);
        float cl[3] = { INFINITY, INFINITY, INFINITY }, ch[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (int t = b; t < e; t++)
            for (int a = 0; a < 3; a++) {
                cl[a] = std::min(cl[a], ctr[(size_t)idx[t] * 3 + a]);
                ch[a] = std::max(ch[a], ctr[(size_t)idx[t] * 3 + a]);
            }
        for (int a = 0; a < 3; a++) {
            if (!(ch[a] > cl[a])) continue;
            if (!binned) {
                for (int t = 0; t < cnt; t++) key[t] = { ctr[(size_t)idx[b + t] * 3 + a], idx[b + t] };
                std::sort(key.begin(), key.end());
                float l[3] = { INFINITY, INFINITY, INFINITY }, h[3] = { -INFINITY, -INFINITY, -INFINITY };
                for (int t = cnt - 1; t > 0; t--) {
                    int s = key[t].second;
                    for (int q = 0; q < 3; q++) { l[q] = std::min(l[q], lo[(size_t)s * 3 + q]); h[q] = std::max(h[q], hi[(size_t)s * 3 + q]); }
                    rarea[t] = half_area(l, h);
                }
                for (int q = 0; q < 3; q++) { l[q] = INFINITY; h[q] = -INFINITY; }
                for (int k = 1; k < cnt; k++) {
                    int s = key[k - 1].second;
                    for (int q = 0; q < 3; q++) { l[q] = std::min(l[q], lo[(size_t)s * 3 + q]); h[q] = std::max(h[q], hi[(size_t)s * 3 + q]); }
                    float cost = half_area(l, h) * k + rarea[k] * (cnt - k);
                    if (cost < best) { best = cost; best_axis = a; best_k = k; }
                }
            } else {
                const int NB = 32;
                float bl[NB][3], bh[NB][3];
                int bc[NB];
                for (int q = 0; q < NB; q++) { bc[q] = 0; for (int r = 0; r < 3; r++) { bl[q][r] = INFINITY; bh[q][r] = -INFINITY; } }
                float scale = NB / (ch[a] - cl[a]);
                for (int t = b; t < e; t++) {
                    int s = idx[t];
                    int q = std::min(NB - 1, std::max(0, (int)((ctr[(size_t)s * 3 + a] - cl[a]) * scale)));
                    bc[q]++;
                    for (int r = 0; r < 3; r++) { bl[q][r] = std::min(bl[q][r], lo[(size_t)s * 3 + r]); bh[q][r] = std::max(bh[q][r], hi[(size_t)s * 3 + r]); }
                }
                float ra[NB]; int rc[NB];
                float l[3] = { INFINITY, INFINITY, INFINITY }, h[3] = { -INFINITY, -INFINITY, -INFINITY };
                int c2 = 0;
                for (int q = NB - 1; q > 0; q--) {
                    c2 += bc[q];
                    for (int r = 0; r < 3; r++) { l[r] = std::min(l[r], bl[q][r]); h[r] = std::max(h[r], bh[q][r]); }
                    ra[q] = half_area(l, h); rc[q] = c2;
                }
                for (int r = 0; r < 3; r++) { l[r] = INFINITY; h[r] = -INFINITY; }
                int c1 = 0;
                for (int q = 1; q < NB; q++) {
                    c1 += bc[q - 1];
                    for (int r = 0; r < 3; r++) { l[r] = std::min(l[r], bl[q - 1][r]); h[r] = std::max(h[r], bh[q - 1][r]); }
                    if (c1 == 0 || rc[q] == 0) continue;
                    float cost = half_area(l, h) * c1 + ra[q] * rc[q];
                    if (cost < best) { best = cost; best_axis = a; best_k = c1; best_pos = cl[a] + q / scale; }
                }
            }
        }
        if (best_axis < 0) return b + cnt / 2;                 // all centroids equal: split the range in half
        if (!binned) {
            for (int t = 0; t < cnt; t++) key[t] = { ctr[(size_t)idx[b + t] * 3 + best_axis], idx[b + t] };
            std::sort(key.begin(), key.end());
            for (int t = 0; t < cnt; t++) idx[b + t] = key[t].second;
            return b + best_k;
        }
        int a = best_axis;
        const int NB = 32;
        float scale = NB / (ch[a] - cl[a]);
        int qsplit = (int)std::lround((best_pos - cl[a]) * scale);
        int m = (int)(std::partition(idx.begin() + b, idx.begin() + e, [&](int s) {
                          int q = std::min(NB - 1, std::max(0, (int)((ctr[(size_t)s * 3 + a] - cl[a]) * scale)));
                          return q < qsplit;
                      }) - idx.begin());
        if (m <= b || m >= e) m = b + cnt / 2;
        return m;
    }

    std::atomic<int> adepth{0};

    // A range [b, e) of k leaf slots becomes a subtree of exactly k - 1 internal nodes, so in DFS pre-order the
    // node of the range is `me`, its left subtree starts at me + 1 and its right subtree at me + (m - b): the
    // numbering needs no shared counter, subtrees touch disjoint parts of idx / child / blo / bhi, and the big
    // ones are built by threads of their own (1 M leaves: 1.5 s on one core).  The result does not depend on
    // how the work was split.
    void build_range(int b, int e, int me, int dep, int spawn_levels) {
        for (;;) {
            int d0 = adepth.load(std::memory_order_relaxed);
            while (dep > d0 && !adepth.compare_exchange_weak(d0, dep, std::memory_order_relaxed)) {}
            float l[3] = { INFINITY, INFINITY, INFINITY }, h[3] = { -INFINITY, -INFINITY, -INFINITY };
            for (int t = b; t < e; t++)
                for (int q = 0; q < 3; q++) { l[q] = std::min(l[q], lo[(size_t)idx[t] * 3 + q]); h[q] = std::max(h[q], hi[(size_t)idx[t] * 3 + q]); }
            for (int q = 0; q < 3; q++) { blo[(size_t)me * 3 + q] = l[q]; bhi[(size_t)me * 3 + q] = h[q]; }
            const int m = split(b, e);
            const int left = me + 1, right = me + (m - b);
            if (m - b == 1) child[(size_t)me * 2 + 0] = ~idx[b]; else child[(size_t)me * 2 + 0] = left;
            if (e - m == 1) child[(size_t)me * 2 + 1] = ~idx[m]; else child[(size_t)me * 2 + 1] = right;
            const bool has_l = m - b > 1, has_r = e - m > 1;
            if (has_l && has_r) {
                if (spawn_levels > 0 && std::max(m - b, e - m) > 16384) {
                    // (a lopsided split keeps its spawn levels for the big half; a thread that cannot be created -- pid /
                    // ulimit limits in a container -- must not take the process down: the half is built here instead)
                    bool spawned = false;
                    std::thread th;
                    if (std::min(m - b, e - m) > 4096) {
                        try { th = std::thread([=] { build_range(b, m, left, dep + 1, spawn_levels - 1); }); spawned = true; }
                        catch (const std::system_error &) { spawned = false; }
                    }
                    if (!spawned) build_range(b, m, left, dep + 1, spawn_levels - 1);
                    build_range(m, e, right, dep + 1, spawn_levels - 1);
                    if (spawned) th.join();
                    return;
                }
                build_range(b, m, left, dep + 1, 0);          // the smaller worlds recurse; depth is bounded by the tree's
                b = m; me = right; dep += 1; spawn_levels = 0;
            } else if (has_l) { e = m; me = left; dep += 1; }
            else if (has_r) { b = m; me = right; dep += 1; }
            else return;
        }
    }

    void run() {
        const int ni = n > 1 ? n - 1 : 0;
        child.assign((size_t)std::max(ni, 1) * 2, 0);
        blo.assign((size_t)std::max(ni, 1) * 3, 0.f);
        bhi.assign((size_t)std::max(ni, 1) * 3, 0.f);
        idx.resize(n);
        for (int i = 0; i < n; i++) idx[i] = i;
        depth = 0;
        if (ni == 0) return;
        adepth.store(0);
        build_range(0, n, 0, 1, 5);                            // up to 2^5 subtrees in flight
        depth = adepth.load();
    }
};

static int build_tree_host(mpt_ctx *c) {
    const int n = c->nfaces;
    const float *V = c->verts.data();
    auto pos = [&](int f, int k) { return V + ((size_t)f * 3 + k) * 8; };

    // genMortonCodes, lbvh.py:169-183
    float bmin[3] = { 1e6f, 1e6f, 1e6f }, bmax[3] = { -1e6f, -1e6f, -1e6f };
    std::vector<float> cen((size_t)n * 3);
    for (int f = 0; f < n; f++)
        for (int a = 0; a < 3; a++) {
            float ctr = ((pos(f, 0)[a] + pos(f, 1)[a]) + pos(f, 2)[a]) / 3.0f;   // lbvh.py:164
            cen[(size_t)f * 3 + a] = ctr;
            bmin[a] = fminf(bmin[a], ctr);
            bmax[a] = fmaxf(bmax[a], ctr);
        }
    std::vector<uint64_t> key(n);
    for (int f = 0; f < n; f++) {
        uint32_t w[3];
        for (int a = 0; a < 3; a++) w[a] = expand_bits((uint32_t)quant1024((cen[(size_t)f * 3 + a] - bmin[a]) / (bmax[a] - bmin[a])));
        uint32_t code = w[0] * 4 + w[1] * 2 + w[2];
        key[f] = ((uint64_t)code << 32) | (uint32_t)f;
    }
    std::sort(key.begin(), key.end());                                          // lbvh.py:204-208

    c->h_leaf.resize(n); c->h_mc.resize(n);
    for (int i = 0; i < n; i++) { c->h_leaf[i] = (int32_t)(key[i] & 0xffffffffu); c->h_mc[i] = (int32_t)(key[i] >> 32); }

    const int ni = n > 1 ? n - 1 : 0;
    c->h_child.assign((size_t)std::max(ni, 1) * 2, 0);
    c->h_bmin.assign((size_t)std::max(ni, 1) * 3, 0.f);
    c->h_bmax.assign((size_t)std::max(ni, 1) * 3, 0.f);

    // genHierarchy, lbvh.py:212-231 (determineRange :93-146, findSplit :62-89)
    for (int i = 0; i < ni; i++) {
        int l, r;
        if (i == 0) { l = 0; r = n - 1; }
        else {
            int d = delta(key, n, i, i + 1) > delta(key, n, i, i - 1) ? 1 : -1;
            int dmin = delta(key, n, i, i - d);
            int lmax = 2;
            while (delta(key, n, i, i + lmax * d) > dmin) lmax <<= 1;
            int s = 0;
            for (int t = lmax >> 1; t > 0; t >>= 1)
                if (delta(key, n, i, i + (s + t) * d) > dmin) s += t;
            l = i; r = i + s * d;
            if (d < 0) std::swap(l, r);
        }
        int cp = delta(key, n, l, r);
        int m = l, s = r - l;
        for (;;) {
            s = (s + 1) >> 1;
            int q = m + s;
            if (q < r && delta(key, n, l, q) > cp) m = q;
            if (s <= 1) break;
        }
        c->h_child[(size_t)i * 2 + 0] = (m == l) ? m : m + n;
        c->h_child[(size_t)i * 2 + 1] = (m + 1 == r) ? m + 1 : m + 1 + n;
    }

    // boxes: iterative post-order from the root (also yields the depth the LDS stack must hold)
    auto leaf_box = [&](int slot, float *lo, float *hi) {                       // lbvh.py:155-158
        int f = c->h_leaf[slot];
        for (int a = 0; a < 3; a++) {
            lo[a] = fminf(fminf(pos(f, 0)[a], pos(f, 1)[a]), pos(f, 2)[a]);
            hi[a] = fmaxf(fmaxf(pos(f, 0)[a], pos(f, 1)[a]), pos(f, 2)[a]);
        }
    };
    int depth = 0;
    if (ni > 0) {
        std::vector<int> order; order.reserve(ni);
        std::vector<std::pair<int, int>> st; st.push_back({ 0, 1 });
        std::vector<char> seen(ni, 0);
        while (!st.empty()) {
            auto [i, dpt] = st.back(); st.pop_back();
            if (i < 0 || i >= ni || seen[i]) return fail("AABB step never stop! hierarchy corrupted?");   // lbvh.py:259
            seen[i] = 1;
            order.push_back(i);
            depth = std::max(depth, dpt);
            for (int k = 0; k < 2; k++) {
                int ch = c->h_child[(size_t)i * 2 + k];
                if (ch >= n) st.push_back({ ch - n, dpt + 1 });
            }
        }
        if ((int)order.size() != ni) return fail("AABB step never stop! hierarchy corrupted?");
        for (int t = ni - 1; t >= 0; t--) {                                     // children before parents
            int i = order[t];
            float lo[2][3], hi[2][3];
            for (int k = 0; k < 2; k++) {
                int ch = c->h_child[(size_t)i * 2 + k];
                if (ch < n) leaf_box(ch, lo[k], hi[k]);
                else for (int a = 0; a < 3; a++) { lo[k][a] = c->h_bmin[(size_t)(ch - n) * 3 + a]; hi[k][a] = c->h_bmax[(size_t)(ch - n) * 3 + a]; }
            }
            for (int a = 0; a < 3; a++) {
                c->h_bmin[(size_t)i * 3 + a] = fminf(lo[0][a], lo[1][a]);
                c->h_bmax[(size_t)i * 3 + a] = fmaxf(hi[0][a], hi[1][a]);
            }
        }
    }
    c->tree_depth = depth;
    if (depth + 2 > 64) return fail("LBVH depth %d exceeds the 64-entry traversal stack", depth);

    // pack device records
    std::vector<MptVec4> snode((size_t)std::max(ni, 1) * 2), fnode((size_t)std::max(ni, 1) * 4);
    std::vector<MptVec4> tgeo((size_t)std::max(n, 1) * 4), tshade((size_t)std::max(n, 1) * 4);
    auto asf = [](int32_t v) { float f; memcpy(&f, &v, 4); return f; };
    for (int i = 0; i < ni; i++) {
        const float *lo = &c->h_bmin[(size_t)i * 3], *hi = &c->h_bmax[(size_t)i * 3];
        int c0 = c->h_child[(size_t)i * 2], c1 = c->h_child[(size_t)i * 2 + 1];
        snode[(size_t)i * 2 + 0] = { lo[0], lo[1], lo[2], asf(c0) };
        snode[(size_t)i * 2 + 1] = { hi[0], hi[1], hi[2], asf(c1) };
    }
    // the tree the fast build walks: child ids >= 0 internal, ~slot leaf; a node record holds its
    // two children's boxes
    std::vector<int32_t> fchild((size_t)std::max(ni, 1) * 2, 0);
    std::vector<float> flo((size_t)std::max(ni, 1) * 3, 0.f), fhi((size_t)std::max(ni, 1) * 3, 0.f);
    if (c->tree_kind == 1 && ni > 0) {
        SahBuild sb;
        sb.n = n;
        sb.exact_max = c->sah_exact_max;
        sb.lo.resize((size_t)n * 3); sb.hi.resize((size_t)n * 3); sb.ctr.resize((size_t)n * 3);
        for (int slot = 0; slot < n; slot++) {
            float l[3], h[3];
            leaf_box(slot, l, h);
            for (int a = 0; a < 3; a++) {
                sb.lo[(size_t)slot * 3 + a] = l[a]; sb.hi[(size_t)slot * 3 + a] = h[a];
                sb.ctr[(size_t)slot * 3 + a] = 0.5f * (l[a] + h[a]);
            }
        }
        sb.run();
        fchild = sb.child; flo = sb.blo; fhi = sb.bhi;
        c->fast_depth = sb.depth;
    } else {
        for (int i = 0; i < ni; i++) {
            for (int k = 0; k < 2; k++) {
                int ch = c->h_child[(size_t)i * 2 + k];
                fchild[(size_t)i * 2 + k] = ch < n ? ~ch : ch - n;
            }
            for (int a = 0; a < 3; a++) { flo[(size_t)i * 3 + a] = c->h_bmin[(size_t)i * 3 + a]; fhi[(size_t)i * 3 + a] = c->h_bmax[(size_t)i * 3 + a]; }
        }
        c->fast_depth = depth;
    }
    if (c->fast_depth + 2 > 64) return fail("BVH depth %d exceeds the 64-entry traversal stack", c->fast_depth);
    for (int i = 0; i < ni; i++) {
        float l[2][3], h[2][3];
        int id[2];
        for (int k = 0; k < 2; k++) {
            id[k] = fchild[(size_t)i * 2 + k];
            if (id[k] < 0) leaf_box(~id[k], l[k], h[k]);
            else for (int a = 0; a < 3; a++) { l[k][a] = flo[(size_t)id[k] * 3 + a]; h[k][a] = fhi[(size_t)id[k] * 3 + a]; }
        }
        for (int a = 0; a < 3; a++) fnode[(size_t)i * 4 + a] = { l[0][a], l[1][a], h[0][a], h[1][a] };
        fnode[(size_t)i * 4 + 3] = { asf(id[0]), asf(id[1]), 0.f, 0.f };
    }
    for (int slot = 0; slot < n; slot++) {
        int f = c->h_leaf[slot];
        const float *p0 = pos(f, 0), *p1 = pos(f, 1), *p2 = pos(f, 2);
        // hoisted terms of Face.intersect, geometries.py:120-122,134-136,140 (same f32 operations: tri_records.h)
        tri_make_tgeo(p0, p1, p2, &tgeo[(size_t)slot * 4]);
        tri_make_tshade(p0, p1, p2, asf(c->mtlids[f]), &tshade[(size_t)slot * 4]);
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if ((size_t)std::max(ni, 1) > c->node_cap) {
        hipFree(c->snode); hipFree(c->fnode); c->snode = c->fnode = nullptr;
        if (dev_alloc(&c->snode, snode.size()) || dev_alloc(&c->fnode, fnode.size())) return 1;
        c->node_cap = std::max(ni, 1);
    }
    if ((size_t)std::max(n, 1) > c->tri_cap) {
        hipFree(c->tgeo); hipFree(c->tshade); c->tgeo = c->tshade = nullptr;
        if (dev_alloc(&c->tgeo, tgeo.size()) || dev_alloc(&c->tshade, tshade.size())) return 1;
        c->tri_cap = std::max(n, 1);
    }
    HIP_TRY(hipMemcpyAsync(c->snode, snode.data(), snode.size() * sizeof(MptVec4), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->fnode, fnode.data(), fnode.size() * sizeof(MptVec4), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->tgeo, tgeo.data(), tgeo.size() * sizeof(MptVec4), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->tshade, tshade.data(), tshade.size() * sizeof(MptVec4), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->tree_valid = true;
    c->host_tree_valid = true;
    return 0;
}

// fnode records for the fast build from a (child, box) description over leaf slots
static void pack_fnode(mpt_ctx *c, int n, const std::vector<int32_t> &fchild, const std::vector<float> &flo,
                       const std::vector<float> &fhi, std::vector<MptVec4> &fnode) {
    const int ni = n > 1 ? n - 1 : 0;
    const float *V = c->verts.data();
    auto asf = [](int32_t v) { float f; memcpy(&f, &v, 4); return f; };
    fnode.assign((size_t)std::max(ni, 1) * 4, MptVec4{ 0, 0, 0, 0 });
    for (int i = 0; i < ni; i++) {
        float l[2][3], h[2][3];
        int id[2];
        for (int k = 0; k < 2; k++) {
            id[k] = fchild[(size_t)i * 2 + k];
            if (id[k] < 0) {
                int f = c->h_leaf[~id[k]];
                for (int a = 0; a < 3; a++) {
                    const float *p0 = V + ((size_t)f * 3) * 8, *p1 = p0 + 8, *p2 = p0 + 16;
                    l[k][a] = fminf(fminf(p0[a], p1[a]), p2[a]);
                    h[k][a] = fmaxf(fmaxf(p0[a], p1[a]), p2[a]);
                }
            } else for (int a = 0; a < 3; a++) { l[k][a] = flo[(size_t)id[k] * 3 + a]; h[k][a] = fhi[(size_t)id[k] * 3 + a]; }
        }
        for (int a = 0; a < 3; a++) fnode[(size_t)i * 4 + a] = { l[0][a], l[1][a], h[0][a], h[1][a] };
        fnode[(size_t)i * 4 + 3] = { asf(id[0]), asf(id[1]), 0.f, 0.f };
    }
}

extern "C" int mpt_sah_workspace(int n, int64_t nseg, int64_t out[4]) {
    if (n < 1 || nseg < 0 || !out) return 1;
    int nb = 0;
    out[0] = (int64_t)mpt_sah_seg_capacity(n);
    out[1] = (int64_t)mpt_sah_part_words(n);
    out[2] = (int64_t)mpt_sah_level_words(n, (size_t)nseg, &nb);
    out[3] = nb;
    return 0;
}

// workspace of mpt_sah_build: one device allocation carved into the arrays of MptSahBuffers
static int build_sah_device(mpt_ctx *c) {
    const int n = c->nfaces;
    const size_t SC = mpt_sah_seg_capacity(n), CC = mpt_sah_chunk_capacity(n), TC = mpt_sah_task_capacity(n), PW = mpt_sah_part_words(n);
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t SW = mpt_sah_segbin_words(n);
    const size_t total = 2 * al((size_t)n * 32) + 2 * al(SC * 64) + al(SC * 32) + al(2 * CC * 4) + al(CC * 4) + al(PW * 4) + al(SW * 4) + al(TC * 32) + al(64);
    if (total > c->sah_ws_bytes) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(c->sah_ws); c->sah_ws = nullptr; c->sah_ws_bytes = 0;
        HIP_TRY(hipMalloc(&c->sah_ws, total));
        c->sah_ws_bytes = total;
    }
    char *q = (char *)c->sah_ws;
    auto take = [&](size_t b) { char *r = q; q += al(b); return r; };
    MptSahBuffers B{};
    B.verts = c->d_verts; B.leaf = c->d_leaf; B.n = n;
    for (int k = 0; k < 2; k++) B.prim[k] = (MptVec4 *)take((size_t)n * 32);
    for (int k = 0; k < 2; k++) B.seg[k] = (int *)take(SC * 64);
    B.seg_cap = SC;
    B.dec = (int *)take(SC * 32);
    B.ch_seg = (int *)take(2 * CC * 4); B.ch_left = (int *)take(CC * 4); B.chunk_cap = CC;
    B.part = (int *)take(PW * 4); B.part_words = PW;
    B.segbins = (int *)take(SW * 4); B.segbin_words = SW;
    B.tasks = (int *)take(TC * 32); B.task_cap = TC;
    B.meta = (int *)take(64);
    B.stats = &c->sah_stats;
    B.mail_host = c->h_sahmeta; B.mail_dev = c->d_sahmeta;
    B.fnode = c->fnode;
    int depth = 0;
    hipError_t e = mpt_sah_build(&B, &depth, c->stream);
    // Neither is fatal: the caller falls back to the host pass, which re-packs c->fnode from the LBVH leaf order (a failed
    // device pass may have overwritten part of it) and has its own depth handling (round-3 ADVICE)
    if (c->sah_inject_fail) e = hipErrorInvalidValue;       // test door: the device pass ran (and wrote c->fnode), then "failed"
    if (e != hipSuccess) { (void)hipGetLastError(); c->sah_fallback = 1; return 2; }
    if (depth + 2 > 64) { c->sah_fallback = 2; return 2; }
    c->fast_depth = depth;
    return 0;
}

// SAH re-partition of the leaves for the fast build: host pass over the leaf order (exact sweep up to sah_exact_max leaves)
static int build_sah_host(mpt_ctx *c) {
    const int n = c->nfaces, ni = n - 1;
    c->h_leaf.resize(n);
    HIP_TRY(hipMemcpy(c->h_leaf.data(), c->d_leaf, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    SahBuild sb;
    sb.n = n;
    sb.exact_max = c->sah_exact_max;
    sb.lo.resize((size_t)n * 3); sb.hi.resize((size_t)n * 3); sb.ctr.resize((size_t)n * 3);
    const float *V = c->verts.data();
    for (int slot = 0; slot < n; slot++) {
        int f = c->h_leaf[slot];
        const float *p0 = V + ((size_t)f * 3) * 8, *p1 = p0 + 8, *p2 = p0 + 16;
        for (int a = 0; a < 3; a++) {
            float l = fminf(fminf(p0[a], p1[a]), p2[a]), h = fmaxf(fmaxf(p0[a], p1[a]), p2[a]);
            sb.lo[(size_t)slot * 3 + a] = l; sb.hi[(size_t)slot * 3 + a] = h; sb.ctr[(size_t)slot * 3 + a] = 0.5f * (l + h);
        }
    }
    sb.run();
    if (sb.depth + 2 > 64) return fail("BVH depth %d exceeds the 64-entry traversal stack", sb.depth);
    std::vector<MptVec4> fnode;
    pack_fnode(c, n, sb.child, sb.blo, sb.bhi, fnode);
    HIP_TRY(hipMemcpy(c->fnode, fnode.data(), (size_t)ni * 4 * sizeof(MptVec4), hipMemcpyHostToDevice));
    c->fast_depth = sb.depth;
    return 0;
}

// lbvh.py:297-305 entirely on the device (lbvh_build.hip); only the depth (4 bytes) comes back,
// plus the leaf order when the fast build wants its SAH re-partition (a host pass today)
static int build_tree_gpu(mpt_ctx *c, BuildClock &clk) {
    const int n = c->nfaces;
    const int ni = n > 1 ? n - 1 : 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if ((size_t)std::max(n, 1) > c->d_model_cap) {
        hipFree(c->d_verts); hipFree(c->d_mtlids); c->d_verts = nullptr; c->d_mtlids = nullptr;
        if (dev_alloc(&c->d_verts, (size_t)std::max(n, 1) * 24) || dev_alloc(&c->d_mtlids, (size_t)std::max(n, 1))) return 1;
        c->d_model_cap = std::max(n, 1);
        c->d_model_stale = true;
    }
    if ((size_t)std::max(n, 1) > c->d_build_cap) {
        hipFree(c->d_cen); hipFree(c->d_bounds); hipFree(c->d_depth); hipFree(c->d_keys_in); hipFree(c->d_keys_out);
        hipFree(c->d_sort_tmp); hipFree(c->d_child); hipFree(c->d_parent); hipFree(c->d_leaf); hipFree(c->d_mc);
        hipFree(c->d_bmin); hipFree(c->d_bmax); hipFree(c->d_arrive);
        c->d_cen = nullptr; c->d_bounds = nullptr; c->d_depth = nullptr; c->d_keys_in = c->d_keys_out = nullptr;
        c->d_sort_tmp = nullptr; c->d_child = c->d_parent = c->d_leaf = c->d_mc = nullptr;
        c->d_bmin = c->d_bmax = nullptr; c->d_arrive = nullptr;
        size_t m = std::max(n, 1);
        HIP_TRY(mpt_lbvh_sort_bytes((int)m, &c->d_sort_bytes));
        if (dev_alloc(&c->d_cen, m * 3) || dev_alloc(&c->d_bounds, 6) || dev_alloc(&c->d_depth, 1) ||
            dev_alloc(&c->d_keys_in, m) || dev_alloc(&c->d_keys_out, m) ||
            dev_alloc((char **)&c->d_sort_tmp, std::max<size_t>(c->d_sort_bytes, 16)) ||
            dev_alloc(&c->d_child, m * 2) || dev_alloc(&c->d_parent, m * 2) || dev_alloc(&c->d_leaf, m) ||
            dev_alloc(&c->d_mc, m) || dev_alloc(&c->d_bmin, m * 3) || dev_alloc(&c->d_bmax, m * 3) ||
            dev_alloc(&c->d_arrive, m)) return 1;
        c->d_build_cap = m;
    }
    if ((size_t)std::max(ni, 1) > c->node_cap) {
        hipFree(c->snode); hipFree(c->fnode); c->snode = c->fnode = nullptr;
        if (dev_alloc(&c->snode, (size_t)std::max(ni, 1) * 2) || dev_alloc(&c->fnode, (size_t)std::max(ni, 1) * 4)) return 1;
        c->node_cap = std::max(ni, 1);
    }
    if ((size_t)std::max(n, 1) > c->tri_cap) {
        hipFree(c->tgeo); hipFree(c->tshade); c->tgeo = c->tshade = nullptr;
        if (dev_alloc(&c->tgeo, (size_t)std::max(n, 1) * 4) || dev_alloc(&c->tshade, (size_t)std::max(n, 1) * 4)) return 1;
        c->tri_cap = std::max(n, 1);
    }
    // the model travels to the device once per ModelPool.load (the reference's load is that copy too, model.py:71-86): a rebuild of
    // the tree -- another option, another tree kind -- finds it there (1.8 of 5.5 ms at a million triangles)
    if (n > 0 && c->d_model_stale) {
        HIP_TRY(hipMemcpyAsync(c->d_verts, c->verts.data(), (size_t)n * 24 * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->d_mtlids, c->mtlids.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    }
    c->d_model_stale = false;
    clk.mark(0);
    MptLbvhBuffers b{};
    b.verts = c->d_verts; b.mtlids = c->d_mtlids; b.n = n;
    b.cen = c->d_cen; b.bounds = c->d_bounds; b.keys_in = c->d_keys_in; b.keys_out = c->d_keys_out;
    b.sort_tmp = c->d_sort_tmp; b.sort_tmp_bytes = c->d_sort_bytes;
    b.child = c->d_child; b.parent = c->d_parent; b.leaf = c->d_leaf; b.mc = c->d_mc;
    b.bmin = c->d_bmin; b.bmax = c->d_bmax; b.arrive = c->d_arrive; b.depth = c->d_depth;
    b.snode = c->snode; b.fnode = c->fnode; b.tgeo = c->tgeo; b.tshade = c->tshade;
    HIP_TRY(mpt_lbvh_build(&b, c->stream));
    int depth = 0;
    if (ni > 0) HIP_TRY(hipMemcpyAsync(&depth, c->d_depth, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (depth + 2 > 64) return fail("LBVH depth %d exceeds the 64-entry traversal stack", depth);
    clk.mark(1);
    c->tree_depth = depth;
    c->fast_depth = depth;
    c->host_tree_valid = false;
    // auto: the host pass up to 8192 faces (all of its splits are exact sweeps there and it costs a millisecond or two: the trees
    // of the small scenes stay what they were), the device pass above (exact sweeps up to 512 triangles, up to 1024 bins per
    // axis above: sah_build.hip)
    const bool sah_on_device = c->sah_build == 1 || (c->sah_build < 0 && n > 8192);
    c->sah_fallback = 0;
    if (c->tree_kind == 1 && ni > 0 && n <= c->sah_max) {
        // SAH re-partition of the leaves: on the device (sah_build.hip: nothing comes back but the depth), or the host pass --
        // also when the device pass gives up (workspace, level or depth limits): both start from the LBVH's leaf order
        int r = (sah_on_device && n >= 2) ? build_sah_device(c) : 2;
        if (r == 1) return 1;
        if (r == 2 && build_sah_host(c)) return 1;
    }
    clk.mark(2);
    c->tree_valid = true;
    return 0;
}

// The binary tree the fast build walks (c->fnode: a node record holds its two children's boxes), collapsed into
// 4-wide nodes for the gather kernel: starting from a node's two children, the internal child with the largest
// surface area is replaced by its own two children until four are held (or only leaves are left).  A ray then
// makes about half as many dependent record fetches, which is what the scenes that do not fit LDS wait for.
// Host pass over the downloaded records (1 M triangles: 64 MB down, ~0.1 s, 64 MB up), off the render path.
// The stack levels a traversal of the 4-wide tree can ask for, exactly (render_kernel_lds4 keeps them all in LDS): a step at a
// node with k children leaves up to k - 1 entries behind and goes on with one child; children come after their parents in the
// array.  ids: the id vector of node w at ids[w * stride]; an unused slot names leaf n
static int wide_stack_levels(const MptVec4 *ids, size_t stride, size_t nw, int n) {
    auto asi = [](float f) { int32_t v; memcpy(&v, &f, 4); return v; };
    std::vector<int> need(nw, 0);
    for (size_t w = nw; w-- > 0;) {
        const MptVec4 &idv = ids[w * stride];
        const int32_t id[4] = { asi(idv.x), asi(idv.y), asi(idv.z), asi(idv.w) };
        int k = 0, deep = 0;
        for (int q = 0; q < 4; q++) {
            if (id[q] != ~n) k++;
            if (id[q] >= 0 && (size_t)id[q] < nw) deep = std::max(deep, need[id[q]]);
        }
        need[w] = std::max(k - 1, 0) + deep;
    }
    return 1 + (nw ? need[0] : 0) + 1;      // the sentinel below, and the level the step's last (unwanted) plain store lands on
}

static int make_wide_host(mpt_ctx *c) {
    c->wide_nodes = 0; c->wide_depth = 0; c->wide_stack = 0;
    const int n = c->nfaces, ni = n > 1 ? n - 1 : 0;
    if (ni < 1) return 0;
    std::vector<MptVec4> fnode((size_t)ni * 4);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(fnode.data(), c->fnode, fnode.size() * sizeof(MptVec4), hipMemcpyDeviceToHost));
    auto asi = [](float f) { int32_t v; memcpy(&v, &f, 4); return v; };
    auto asf = [](int32_t v) { float f; memcpy(&f, &v, 4); return f; };
    struct Child { int32_t id; float lo[3], hi[3]; };
    auto children_of = [&](int b, Child out[2]) {
        const MptVec4 *r = &fnode[(size_t)b * 4];
        const float lox[2] = { r[0].x, r[0].y }, hix[2] = { r[0].z, r[0].w };
        const float loy[2] = { r[1].x, r[1].y }, hiy[2] = { r[1].z, r[1].w };
        const float loz[2] = { r[2].x, r[2].y }, hiz[2] = { r[2].z, r[2].w };
        const int32_t id[2] = { asi(r[3].x), asi(r[3].y) };
        for (int k = 0; k < 2; k++) {
            out[k].id = id[k];
            out[k].lo[0] = lox[k]; out[k].lo[1] = loy[k]; out[k].lo[2] = loz[k];
            out[k].hi[0] = hix[k]; out[k].hi[1] = hiy[k]; out[k].hi[2] = hiz[k];
        }
    };
    auto area = [](const Child &ch) {
        float dx = std::max(ch.hi[0] - ch.lo[0], 0.f), dy = std::max(ch.hi[1] - ch.lo[1], 0.f), dz = std::max(ch.hi[2] - ch.lo[2], 0.f);
        return dx * dy + dy * dz + dz * dx;
    };
    // expected dependent fetches per ray ~ sum of the surface areas of the nodes that are fetched (SAH argument):
    // over all internal nodes for the binary tree, over the nodes wide records are grown from for the collapse
    std::vector<float> node_area((size_t)ni, 0.f);
    {
        Child two[2];
        children_of(0, two);
        Child root = two[0];
        for (int a = 0; a < 3; a++) { root.lo[a] = std::min(two[0].lo[a], two[1].lo[a]); root.hi[a] = std::max(two[0].hi[a], two[1].hi[a]); }
        node_area[0] = area(root);
        for (int b = 0; b < ni; b++) {
            children_of(b, two);
            for (int k = 0; k < 2; k++) if (two[k].id >= 0) node_area[two[k].id] = area(two[k]);
        }
    }
    double area_bin = 0.0, area_wide = 0.0;
    for (int b = 0; b < ni; b++) area_bin += node_area[b];
    std::vector<MptVec4> wnode, qnode;
    wnode.reserve((size_t)ni * 4);
    qnode.reserve((size_t)ni * 2);
    std::vector<int> bin_of;            // wide node -> the binary node it was grown from
    std::vector<int> depth_of;
    bin_of.push_back(0); depth_of.push_back(1);
    int depth = 1;
    for (size_t w = 0; w < bin_of.size(); w++) {
        Child ch[4];
        int cnt = 2;
        children_of(bin_of[w], ch);
        area_wide += node_area[bin_of[w]];
        while (cnt < 4) {
            int best = -1; float ba = -1.f;
            for (int k = 0; k < cnt; k++)
                if (ch[k].id >= 0) { float a = area(ch[k]); if (a > ba) { ba = a; best = k; } }
            if (best < 0) break;
            Child two[2];
            children_of(ch[best].id, two);
            ch[best] = two[0];
            ch[cnt++] = two[1];
        }
        MptVec4 rec[8];
        float *f = &rec[0].x;
        for (int k = 0; k < 32; k++) f[k] = 0.f;
        // an unused slot names leaf slot n: one extra triangle record of NaNs that no ray can hit (round 2 stored 0, the
        // root, there: a ray along (1,1,1) could be sent back to it)
        int32_t ids[4] = { ~n, ~n, ~n, ~n };
        for (int k = 0; k < 4; k++) {
            float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { 1e30f, 1e30f, 1e30f };   // unused child: out of every ray's reach
            if (k < cnt) {
                for (int a = 0; a < 3; a++) { lo[a] = ch[k].lo[a]; hi[a] = ch[k].hi[a]; }
                if (ch[k].id < 0) ids[k] = ch[k].id;
                else {
                    ids[k] = (int32_t)bin_of.size();
                    bin_of.push_back(ch[k].id);
                    depth_of.push_back(depth_of[w] + 1);
                    depth = std::max(depth, depth_of[w] + 1);
                }
            }
            for (int a = 0; a < 3; a++) { (&rec[2 * a].x)[k] = lo[a]; (&rec[2 * a + 1].x)[k] = hi[a]; }
        }
        rec[6] = { asf(ids[0]), asf(ids[1]), asf(ids[2]), asf(ids[3]) };
        for (int k = 0; k < 8; k++) wnode.push_back(rec[k]);
        // the quantised record: child planes as bytes over the node's own box, rounded outwards by a quarter of a step
        // more than needed (the kernel's decode q * (scale * inv) + (origin * inv - o * inv) is off by far less)
        {
            float plo[3] = { INFINITY, INFINITY, INFINITY }, phi[3] = { -INFINITY, -INFINITY, -INFINITY };
            for (int k = 0; k < cnt; k++)
                for (int a = 0; a < 3; a++) { plo[a] = std::min(plo[a], ch[k].lo[a]); phi[a] = std::max(phi[a], ch[k].hi[a]); }
            float scale[3];
            uint32_t qlo[3] = { 0, 0, 0 }, qhi[3] = { 0, 0, 0 };
            for (int a = 0; a < 3; a++) {
                const float e = phi[a] - plo[a];
                float sc = e > 0.f ? e / 255.f : 0.f;
                // 255 steps must reach the far side in f32, and a flat node still needs a positive step
                while (e > 0.f && plo[a] + 255.f * sc < phi[a]) sc = std::nextafter(sc, INFINITY);
                if (!(sc > 0.f)) sc = std::max(std::fabs(plo[a]) * 1e-6f, 1e-30f);
                scale[a] = sc;
                for (int k = 0; k < 4; k++) {
                    uint32_t l = 255, h = 0;                        // unused child: an inverted box
                    if (k < cnt) {
                        const float fl = std::floor((ch[k].lo[a] - plo[a]) / sc - 0.25f), fh = std::ceil((ch[k].hi[a] - plo[a]) / sc + 0.25f);
                        l = (uint32_t)std::min(255.f, std::max(0.f, fl));
                        h = (uint32_t)std::min(255.f, std::max(0.f, fh));
                    }
                    qlo[a] |= l << (8 * k); qhi[a] |= h << (8 * k);
                }
            }
            qnode.push_back({ plo[0], plo[1], plo[2], scale[0] });
            qnode.push_back({ scale[1], scale[2], asf((int32_t)qlo[0]), asf((int32_t)qhi[0]) });
            qnode.push_back({ asf((int32_t)qlo[1]), asf((int32_t)qhi[1]), asf((int32_t)qlo[2]), asf((int32_t)qhi[2]) });
            qnode.push_back(rec[6]);
        }
    }
    // a step pushes up to three entries: 3 x depth + sentinel must fit the LDS levels plus the spill strip
    if (3 * depth + 2 > 128) return 0;      // too deep (40 LDS levels + 88 spilled): the gather kernel keeps walking the binary tree
    const size_t nw = bin_of.size();
    if (nw * 8 * sizeof(MptVec4) >= ((size_t)1 << 31)) return 0;   // the kernel addresses the records with 32-bit byte offsets
    if (nw > c->wnode_cap) {
        hipFree(c->wnode); c->wnode = nullptr; c->wnode_cap = 0;
        if (dev_alloc(&c->wnode, nw * 8)) return 1;
        c->wnode_cap = nw;
    }
    HIP_TRY(hipMemcpy(c->wnode, wnode.data(), nw * 8 * sizeof(MptVec4), hipMemcpyHostToDevice));
    if (nw > c->qnode_cap) {
        hipFree(c->qnode); c->qnode = nullptr; c->qnode_cap = 0;
        if (dev_alloc(&c->qnode, nw * 4)) return 1;
        c->qnode_cap = nw;
    }
    HIP_TRY(hipMemcpy(c->qnode, qnode.data(), nw * 4 * sizeof(MptVec4), hipMemcpyHostToDevice));
    c->wide_nodes = (int)nw; c->wide_depth = depth;
    c->wide_stack = wide_stack_levels(wnode.data() + 6, 8, nw, n);
    c->wide_ratio = area_bin > 0.0 ? (float)(area_wide / area_bin) : 1.f;
    return 0;
}

// The same collapse on the device (wide_build.hip): nothing is downloaded, one integer per level comes back.  Produces the
// host pass's bytes (tests/test_parity_gpu.py::test_device_wide_collapse_equals_the_host_pass).
static int make_wide_device(mpt_ctx *c) {
    c->wide_nodes = 0; c->wide_depth = 0; c->wide_stack = 0;
    const int n = c->nfaces, ni = n > 1 ? n - 1 : 0;
    if (ni < 1) return 0;
    if ((size_t)ni * 8 * sizeof(MptVec4) >= ((size_t)1 << 31)) return 0;   // the kernel addresses the records with 32-bit byte offsets
    if ((size_t)ni > c->wnode_cap) {
        hipFree(c->wnode); c->wnode = nullptr; c->wnode_cap = 0;
        if (dev_alloc(&c->wnode, (size_t)ni * 8)) return 1;
        c->wnode_cap = ni;
    }
    if ((size_t)ni > c->qnode_cap) {
        hipFree(c->qnode); c->qnode = nullptr; c->qnode_cap = 0;
        if (dev_alloc(&c->qnode, (size_t)ni * 4)) return 1;
        c->qnode_cap = ni;
    }
    if ((size_t)ni > c->wb_cap) {
        hipFree(c->wb_bin_of); hipFree(c->wb_ncount); hipFree(c->wb_offset); hipFree(c->wb_scan); hipFree(c->wb_area);
        c->wb_bin_of = c->wb_ncount = c->wb_offset = nullptr; c->wb_scan = nullptr; c->wb_area = nullptr; c->wb_cap = 0;
        HIP_TRY(mpt_wide_scan_bytes(ni, &c->wb_scan_bytes));
        // (wb_offset: not used since round 6 -- the nodes' offsets within their workgroup live in wb_ncount, the workgroups' in wb_scan)
        if (dev_alloc(&c->wb_bin_of, (size_t)ni + 4) || dev_alloc(&c->wb_ncount, (size_t)ni) ||
            dev_alloc((char **)&c->wb_scan, std::max<size_t>(c->wb_scan_bytes, 16)) || dev_alloc(&c->wb_area, 2)) return 1;
        c->wb_cap = ni;
    }
    int nw = 0, depth = 0;
    double area[2] = { 0.0, 0.0 };
    HIP_TRY(mpt_wide_build(c->fnode, n, c->wnode, c->qnode, c->wb_bin_of, c->wb_ncount, c->wb_offset, c->wb_scan, c->wb_scan_bytes,
                           c->wb_area, &nw, &depth, area, c->stream, c->h_sahmeta, c->d_sahmeta));
    // a step pushes up to three entries: 3 x depth + sentinel must fit the LDS levels plus the spill strip
    if (3 * depth + 2 > 128) return 0;      // too deep: the gather kernel keeps walking the binary tree
    c->wide_nodes = nw; c->wide_depth = depth;
    c->wide_stack = 3 * depth + 2;          // the bound ...
    if (nw > 0 && nw <= 4096) {             // ... and for a tree that may be walked out of LDS the exact count, from its id vectors (64 KiB at most)
        std::vector<MptVec4> ids((size_t)nw);
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy2D(ids.data(), sizeof(MptVec4), c->wnode + 6, 8 * sizeof(MptVec4), sizeof(MptVec4), (size_t)nw, hipMemcpyDeviceToHost));
        c->wide_stack = std::min(c->wide_stack, wide_stack_levels(ids.data(), 1, (size_t)nw, n));
    }
    c->wide_ratio = area[1] > 0.0 ? (float)(area[0] / area[1]) : 1.f;
    return 0;
}

static int make_wide(mpt_ctx *c) { return c->wide_build ? make_wide_device(c) : make_wide_host(c); }

// test / inspection: the 4-wide records the gather kernels walk -- wnode [nw][8] float4 (exact boxes), qnode [nw][4] float4
// (8-bit boxes); any pointer may be NULL; *nw = number of wide nodes (0: not built)
extern "C" int mpt_get_wide(mpt_ctx *c, float *wnode, float *qnode, int cap_nodes, int *nw) {
    if (use_ro(c)) return 1;
    if (!c->tree_valid) return fail("BVH not built: call build_tree() after load_model()");
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int k = std::min(cap_nodes, c->wide_nodes);
    if (wnode && k > 0) HIP_TRY(hipMemcpy(wnode, c->wnode, (size_t)k * 8 * sizeof(MptVec4), hipMemcpyDeviceToHost));
    if (qnode && k > 0) HIP_TRY(hipMemcpy(qnode, c->qnode, (size_t)k * 4 * sizeof(MptVec4), hipMemcpyDeviceToHost));
    if (nw) *nw = c->wide_nodes;
    return 0;
}

extern "C" int mpt_build_tree(mpt_ctx *c) {
    if (use(c)) return 1;
    c->fnode_soa_valid = false;
    BuildClock clk(c);
    if (c->gpu_build ? build_tree_gpu(c, clk) : build_tree_host(c)) return 1;
    clk.mark(2);
    // the production kernels' 48-byte triangle records, from the reference-order ones
    // (+ 1: record n is all NaNs -- the leaf an unused slot of a 4-wide node names; no ray can hit it)
    const size_t nt = (size_t)std::max(c->nfaces, 1) + 1;
    if (nt > c->tfast_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(c->tfast); c->tfast = nullptr; c->tfast_cap = 0;
        if (dev_alloc(&c->tfast, nt * 3)) return 1;
        c->tfast_cap = nt;
    }
    HIP_TRY(mpt_launch_derive_tfast(c->tgeo, c->tfast, c->nfaces, c->stream));
    HIP_TRY(hipMemsetAsync(c->tfast + (size_t)c->nfaces * 3, 0xff, 3 * sizeof(MptVec4), c->stream));   // 0xffffffff: a NaN
    clk.mark(3);
    if (make_wide(c)) return 1;
    clk.mark(4);
    c->oct_nodes = 0; c->oct_depth = 0;
    if (c->use_wide8 && c->tree_kind == 1) return make_oct8(c);     // option "wide8": the same tree 8-wide, octant-ordered (oct_build.cpp)
    return 0;
}

static int download_tree(mpt_ctx *c) {
    if (c->host_tree_valid) return 0;
    const int n = c->nfaces, ni = n > 1 ? n - 1 : 0;
    c->h_child.assign((size_t)std::max(ni, 1) * 2, 0); c->h_leaf.assign(std::max(n, 1), 0); c->h_mc.assign(std::max(n, 1), 0);
    c->h_bmin.assign((size_t)std::max(ni, 1) * 3, 0.f); c->h_bmax.assign((size_t)std::max(ni, 1) * 3, 0.f);
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (n > 0) {
        HIP_TRY(hipMemcpy(c->h_leaf.data(), c->d_leaf, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(c->h_mc.data(), c->d_mc, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    if (ni > 0) {
        HIP_TRY(hipMemcpy(c->h_child.data(), c->d_child, (size_t)ni * 2 * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(c->h_bmin.data(), c->d_bmin, (size_t)ni * 3 * sizeof(float), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(c->h_bmax.data(), c->d_bmax, (size_t)ni * 3 * sizeof(float), hipMemcpyDeviceToHost));
    }
    c->host_tree_valid = true;
    return 0;
}

extern "C" int mpt_get_tree(mpt_ctx *c, int32_t *child, int32_t *leaf, float *bmin, float *bmax, int32_t *mc,
                            int32_t *depth) {
    if (!c) return fail("null context");
    if (!c->tree_valid) return fail("BVH not built: call build_tree() after load_model()");
    if (download_tree(c)) return 1;
    int n = c->nfaces, ni = n > 1 ? n - 1 : 0;
    if (child) memcpy(child, c->h_child.data(), (size_t)ni * 2 * sizeof(int32_t));
    if (leaf) memcpy(leaf, c->h_leaf.data(), (size_t)n * sizeof(int32_t));
    if (bmin) memcpy(bmin, c->h_bmin.data(), (size_t)ni * 3 * sizeof(float));
    if (bmax) memcpy(bmax, c->h_bmax.data(), (size_t)ni * 3 * sizeof(float));
    if (mc) memcpy(mc, c->h_mc.data(), (size_t)n * sizeof(int32_t));
    if (depth) *depth = c->tree_depth;
    return 0;
}
