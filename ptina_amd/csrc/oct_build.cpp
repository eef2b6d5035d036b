// oct_build.cpp -- the fast tree collapsed 8-wide, child slots in OCTANT order, child boxes in 8 bits: the records
// render_kernel_oct walks (option "wide8").  No reference counterpart: the reference walks its binary LBVH
// (ptina/tree/lbvh.py:314-347); the hits a ray finds are the same, the order equal-depth ties are met in is not.
//
// Why: the 4-wide gather kernel sorts the four children of every step by entry distance (a five-comparator network on
// (distance, id) pairs) and pushes up to three ids; what it waits for is the number of dependent record gathers per ray.
// Here (after Ylitie, Karras, Laine: "Efficient incoherent ray traversal on GPUs through compressed wide BVHs", HPG 2017):
//   * eight children per node: a ray makes ~0.6 x the node visits of the 4-wide tree, each five 16-byte gathers instead of four;
//   * a child sits in the slot whose octant (sign pattern of its offset from the node's centre) it best matches, so the order in
//     which a ray meets the children is read off the slot number: slot XOR the ray's direction octant, ascending -- NO sort;
//   * a node's internal children are numbered consecutively in slot order and so are the triangles of its leaf children
//     (the triangle records are permuted into that order: tfast8 / tshade8), so a traversal stack entry is ONE pair
//     (first child | mask of slots of that kind, slots still to visit) per node instead of up to three ids.
// Record, 80 bytes = five float4:
//   {origin.x, origin.y, origin.z, scale.x} {scale.y, scale.z, child_base | imask << 24, tri_base | lmask << 24}
//   {lo.x[8], hi.x[8]} {lo.y[8], hi.y[8]} {lo.z[8], hi.z[8]}          (bytes; plane = origin + byte * scale, rounded outwards)
// imask / lmask: the slots holding internal nodes / leaves (one triangle each); an empty slot's box is inverted (never hit).
// Host pass over the downloaded binary records (like the round-2 4-wide collapse): off the render path.

#include "miptina_ctx.h"
#include <cstring>

#if !MPT_WITH_OCT
// The 8-wide octant-ordered tree is an A/B build (make -C ptina_amd/csrc oct -> libmiptina_oct.so; DESIGN.md 3.2: -17 ... -19 % against
// the 4-wide node): the product library keeps the entry point of the C ABI and says so.  (Option "wide8" cannot be set in this build.)
MPT_INTERNAL int make_oct8(mpt_ctx *) { return fail("this library is built without the 8-wide octant-ordered tree (make -C ptina_amd/csrc oct)"); }
extern "C" int mpt_get_oct8(mpt_ctx *, float *, int32_t *, int, int *) {
    return fail("this library is built without the 8-wide octant-ordered tree (an A/B build: make -C ptina_amd/csrc oct)");
}
#else

namespace {
struct OChild { int32_t id; float lo[3], hi[3]; };
inline int32_t asi(float f) { int32_t v; memcpy(&v, &f, 4); return v; }
inline float asf(int32_t v) { float f; memcpy(&f, &v, 4); return f; }
inline float area_of(const OChild &c) {
    const float dx = std::max(c.hi[0] - c.lo[0], 0.f), dy = std::max(c.hi[1] - c.lo[1], 0.f), dz = std::max(c.hi[2] - c.lo[2], 0.f);
    return dx * dy + dy * dz + dz * dx;
}
}

// c->fnode (binary, n - 1 records) -> c->onode [nw][5], c->tfast8 [n + 1][3], c->tshade8 [n][4]; c->oct_nodes = nw (0: not built)
MPT_INTERNAL int make_oct8(mpt_ctx *c) {
    c->oct_nodes = 0; c->oct_depth = 0;
    const int n = c->nfaces, ni = n > 1 ? n - 1 : 0;
    if (ni < 1 || n >= (1 << 24)) return 0;                     // (24-bit child / triangle bases)
    std::vector<MptVec4> fnode((size_t)ni * 4);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(fnode.data(), c->fnode, fnode.size() * sizeof(MptVec4), hipMemcpyDeviceToHost));
    auto children_of = [&](int b, OChild out[2]) {
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
    std::vector<MptVec4> onode;
    onode.reserve((size_t)ni * 2);
    std::vector<int32_t> perm((size_t)n, -1);                   // t8 -> leaf slot of the production records
    std::vector<int> bin_of, depth_of;                          // wide node -> the binary node it grows from, its level
    bin_of.push_back(0); depth_of.push_back(1);
    int depth = 1, ntri = 0;
    for (size_t w = 0; w < bin_of.size(); w++) {
        OChild ch[8];
        int cnt = 2;
        children_of(bin_of[w], ch);
        while (cnt < 8) {                                       // the internal child with the largest surface gives way to its own two
            int best = -1; float ba = -1.f;
            for (int k = 0; k < cnt; k++)
                if (ch[k].id >= 0) { const float a = area_of(ch[k]); if (a > ba) { ba = a; best = k; } }
            if (best < 0) break;
            OChild two[2];
            children_of(ch[best].id, two);
            ch[best] = two[0];
            ch[cnt++] = two[1];
        }
        float plo[3] = { INFINITY, INFINITY, INFINITY }, phi[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (int k = 0; k < cnt; k++)
            for (int a = 0; a < 3; a++) { plo[a] = std::min(plo[a], ch[k].lo[a]); phi[a] = std::max(phi[a], ch[k].hi[a]); }
        // ---- octant slots: slot bit a set = the child lies on the high side of the node's centre along axis a.  Greedy
        // assignment by the largest remaining (child, slot) score = offset of the child's centre from the node's . slot diagonal
        int slot_of[8], child_in[8];
        for (int s = 0; s < 8; s++) child_in[s] = -1;
        for (int k = 0; k < 8; k++) slot_of[k] = -1;
        {
            float off[8][3];
            for (int k = 0; k < cnt; k++)
                for (int a = 0; a < 3; a++) off[k][a] = 0.5f * (ch[k].lo[a] + ch[k].hi[a]) - 0.5f * (plo[a] + phi[a]);
            for (int round = 0; round < cnt; round++) {
                int bk = -1, bs = -1; float bv = -INFINITY;
                for (int k = 0; k < cnt; k++) {
                    if (slot_of[k] >= 0) continue;
                    for (int s = 0; s < 8; s++) {
                        if (child_in[s] >= 0) continue;
                        const float v = ((s & 1) ? off[k][0] : -off[k][0]) + ((s & 2) ? off[k][1] : -off[k][1]) + ((s & 4) ? off[k][2] : -off[k][2]);
                        if (v > bv) { bv = v; bk = k; bs = s; }
                    }
                }
                slot_of[bk] = bs; child_in[bs] = bk;
            }
        }
        // ---- numbering: internal children consecutively in slot order (breadth first), leaf triangles likewise
        uint32_t imask = 0, lmask = 0;
        const uint32_t child_base = (uint32_t)bin_of.size(), tri_base = (uint32_t)ntri;
        for (int s = 0; s < 8; s++) {
            const int k = child_in[s];
            if (k < 0) continue;
            if (ch[k].id >= 0) {
                imask |= 1u << s;
                bin_of.push_back(ch[k].id);
                depth_of.push_back(depth_of[w] + 1);
                depth = std::max(depth, depth_of[w] + 1);
            } else {
                lmask |= 1u << s;
                perm[(size_t)ntri++] = ~ch[k].id;
            }
        }
        if (bin_of.size() >= ((size_t)1 << 24)) return 0;
        // ---- 8-bit planes over the node's own box, rounded outwards by a quarter of a step more than needed (the kernel's decode
        // byte * (scale * inv) + (origin * inv - o * inv) is off by far less): the 4-wide records' rule (tree_build.cpp)
        float scale[3];
        uint32_t qlo[3][2] = { { 0, 0 }, { 0, 0 }, { 0, 0 } }, qhi[3][2] = { { 0, 0 }, { 0, 0 }, { 0, 0 } };
        for (int a = 0; a < 3; a++) {
            const float e = phi[a] - plo[a];
            float sc = e > 0.f ? e / 255.f : 0.f;
            while (e > 0.f && plo[a] + 255.f * sc < phi[a]) sc = std::nextafter(sc, INFINITY);
            if (!(sc > 0.f)) sc = std::max(std::fabs(plo[a]) * 1e-6f, 1e-30f);
            scale[a] = sc;
            for (int s = 0; s < 8; s++) {
                uint32_t l = 255, h = 0;                        // empty slot: an inverted box
                const int k = child_in[s];
                if (k >= 0) {
                    const float fl = std::floor((ch[k].lo[a] - plo[a]) / sc - 0.25f), fh = std::ceil((ch[k].hi[a] - plo[a]) / sc + 0.25f);
                    l = (uint32_t)std::min(255.f, std::max(0.f, fl));
                    h = (uint32_t)std::min(255.f, std::max(0.f, fh));
                }
                qlo[a][s >> 2] |= l << (8 * (s & 3)); qhi[a][s >> 2] |= h << (8 * (s & 3));
            }
        }
        onode.push_back({ plo[0], plo[1], plo[2], scale[0] });
        onode.push_back({ scale[1], scale[2], asf((int32_t)(child_base | (imask << 24))), asf((int32_t)(tri_base | (lmask << 24))) });
        for (int a = 0; a < 3; a++)
            onode.push_back({ asf((int32_t)qlo[a][0]), asf((int32_t)qlo[a][1]), asf((int32_t)qhi[a][0]), asf((int32_t)qhi[a][1]) });
    }
    if (ntri != n) return fail("8-wide collapse: %d of %d triangles placed", ntri, n);
    // a visited node leaves at most two entries behind (its other internal hits, its leaf hits): 2 x depth + sentinel within
    // the LDS levels plus the spill strip
    if (2 * depth + 2 > 120) return 0;
    const size_t nw = bin_of.size();
    if (nw * 5 * sizeof(MptVec4) >= ((size_t)1 << 31)) return 0;                  // 32-bit byte offsets in the kernel
    if (nw > c->onode_cap) {
        hipFree(c->onode); c->onode = nullptr; c->onode_cap = 0;
        if (dev_alloc(&c->onode, nw * 5)) return 1;
        c->onode_cap = nw;
    }
    HIP_TRY(hipMemcpy(c->onode, onode.data(), nw * 5 * sizeof(MptVec4), hipMemcpyHostToDevice));
    // ---- the triangle records in t8 order (+ the NaN record n): gathered on the device from the leaf-order ones
    if ((size_t)n > c->tri8_cap) {
        hipFree(c->tfast8); hipFree(c->tshade8); hipFree(c->d_perm8);
        c->tfast8 = c->tshade8 = nullptr; c->d_perm8 = nullptr; c->tri8_cap = 0;
        if (dev_alloc(&c->tfast8, ((size_t)n + 1) * 3) || dev_alloc(&c->tshade8, (size_t)n * 4) || dev_alloc(&c->d_perm8, (size_t)n)) return 1;
        c->tri8_cap = n;
    }
    HIP_TRY(hipMemcpy(c->d_perm8, perm.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(mpt_launch_permute_tris(c->tfast, c->tshade, c->d_perm8, c->tfast8, c->tshade8, n, c->stream));
    HIP_TRY(hipMemsetAsync(c->tfast8 + (size_t)n * 3, 0xff, 3 * sizeof(MptVec4), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->oct_nodes = (int)nw; c->oct_depth = depth;
    return 0;
}

// test / inspection: the 8-wide records [nw][5][4] f32 and the triangle permutation [n] (t8 -> leaf slot)
extern "C" int mpt_get_oct8(mpt_ctx *c, float *onode, int32_t *perm, int cap_nodes, int *nw) {
    if (!c) return fail("null context");
    HIP_TRY(hipSetDevice(c->device));
    if (!c->tree_valid) return fail("BVH not built: call build_tree() after load_model()");
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int k = std::min(cap_nodes, c->oct_nodes);
    if (onode && k > 0) HIP_TRY(hipMemcpy(onode, c->onode, (size_t)k * 5 * sizeof(MptVec4), hipMemcpyDeviceToHost));
    if (perm && c->oct_nodes > 0) HIP_TRY(hipMemcpy(perm, c->d_perm8, (size_t)c->nfaces * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (nw) *nw = c->oct_nodes;
    return 0;
}

#endif   // MPT_WITH_OCT
