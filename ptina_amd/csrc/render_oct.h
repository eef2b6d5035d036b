// render_oct.h -- the 8-wide, octant-ordered compressed node (Ylitie, Karras, Laine, HPG 2017) for the gather kernel: option "wide8",
// built by oct_build.cpp.  A recorded loss (C4 -19 %, C5 -17 % against the 4-wide 8-bit node: DESIGN.md 3.2, profiles/r04_*oct*), so it is
// NOT part of the product library: `make -C ptina_amd/csrc oct` -> libmiptina_oct.so (-DMPT_WITH_OCT=1), which its test loads by MIPTINA_LIB.
// Included twice by render_kernel.hip: first for the traversal steps (in front of trace_stream), then, with MPT_OCT_KERNELS defined,
// for the kernel and its launchers (behind trace_stream / finalise_tiles).
#ifndef MPT_OCT_KERNELS

// ---- the 8-wide, octant-ordered tree (OctScene / OctStack; oct_build.cpp)
// The ray's direction octant: bit a set = the ray goes DOWN axis a.  A child slot's bit a set = the child lies on the high side of
// the node's centre along a, so slot ^ octant, ascending, is the order the ray meets the children in (nearest first).
DEV unsigned ray_octant(const LaneState &L) {
    return ((unsigned)__float_as_int(L.inv.x) >> 31) | (((unsigned)__float_as_int(L.inv.y) >> 31) << 1) | (((unsigned)__float_as_int(L.inv.z) >> 31) << 2);
}
// bit i of an 8-bit mask to bit i ^ r: three conditional delta swaps
DEV unsigned oct_permute(unsigned m, unsigned r) {
    const unsigned m1 = ((m & 0x55u) << 1) | ((m >> 1) & 0x55u);
    m = (r & 1u) ? m1 : m;
    const unsigned m2 = ((m & 0x33u) << 2) | ((m >> 2) & 0x33u);
    m = (r & 2u) ? m2 : m;
    const unsigned m4 = ((m & 0x0fu) << 4) | ((m >> 4) & 0x0fu);
    return (r & 4u) ? m4 : m;
}
// the next thing to do is on top of the stack: a leaf group (b < 0: the lowest slot left names the next triangle), a group of
// internal children (the lowest MET-ORDER position left names the next node) or the sentinel
template <class STACK>
DEV void oct_next(STACK &stk, LaneState &L, unsigned r) {
    int a, b;
    const int top = L.sp - 1;
    stk.get(top, a, b);
    if (b & STACK::SENTINEL) { L.st = ST_DONE; return; }
    const bool leaf = b < 0;
    const unsigned bits = (unsigned)b & 0xffu;
    const int pos = __builtin_ctz(bits | 0x100u);
    const unsigned slot = leaf ? (unsigned)pos : ((unsigned)pos ^ r);
    const int idx = (a & 0xffffff) + __builtin_popcount(((unsigned)a >> 24) & ((1u << slot) - 1u));
    const unsigned rest = bits & (bits - 1u);
    if (rest) stk.setb(top, (int)(((unsigned)b & 0x80000000u) | rest));
    else L.sp = top;
    L.curr = leaf ? ~idx : idx;
    L.st = leaf ? ST_LEAF : ST_NODE;
}

template <bool COUNT, class SCENE, class STACK>
DEV void stage_node8(const SCENE &sc, STACK &stk, LaneState &L, Cnt &cnt) {
    if (COUNT) { cnt.n_node++; cnt.n_box += 8; }
    MptVec4 h0, h1, px, py, pz;
    sc.node8(L.curr, h0, h1, px, py, pz);
    // plane = origin + byte * scale: its distance along the ray is byte * (scale * inv) + (origin * inv - o * inv)
    const float sx = h0.w * L.inv.x, sy = h1.x * L.inv.y, sz = h1.y * L.inv.z;
    const float bx = __builtin_fmaf(h0.x, L.inv.x, -L.oinv.x), by = __builtin_fmaf(h0.y, L.inv.y, -L.oinv.y),
                bz = __builtin_fmaf(h0.z, L.inv.z, -L.oinv.z);
    const int a_node = __float_as_int(h1.z), a_tri = __float_as_int(h1.w);
    const unsigned imask = (unsigned)a_node >> 24, lmask = (unsigned)a_tri >> 24;
    const unsigned r = ray_octant(L);
    const bool dnx = (r & 1u) != 0, dny = (r & 2u) != 0, dnz = (r & 4u) != 0;
    // entry planes: the low ones for a ray going up the axis, the high ones for one going down ({lo[0..3], lo[4..7], hi[0..3], hi[4..7]})
    const unsigned lx0 = (unsigned)__float_as_int(px.x), lx1 = (unsigned)__float_as_int(px.y), hx0 = (unsigned)__float_as_int(px.z), hx1 = (unsigned)__float_as_int(px.w);
    const unsigned ly0 = (unsigned)__float_as_int(py.x), ly1 = (unsigned)__float_as_int(py.y), hy0 = (unsigned)__float_as_int(py.z), hy1 = (unsigned)__float_as_int(py.w);
    const unsigned lz0 = (unsigned)__float_as_int(pz.x), lz1 = (unsigned)__float_as_int(pz.y), hz0 = (unsigned)__float_as_int(pz.z), hz1 = (unsigned)__float_as_int(pz.w);
    const unsigned nx0 = dnx ? hx0 : lx0, nx1 = dnx ? hx1 : lx1, fx0 = dnx ? lx0 : hx0, fx1 = dnx ? lx1 : hx1;
    const unsigned ny0 = dny ? hy0 : ly0, ny1 = dny ? hy1 : ly1, fy0 = dny ? ly0 : hy0, fy1 = dny ? ly1 : hy1;
    const unsigned nz0 = dnz ? hz0 : lz0, nz1 = dnz ? hz1 : lz1, fz0 = dnz ? lz0 : hz0, fz1 = dnz ? lz1 : hz1;
    unsigned hits = 0u;
#define MPT_UB(w, c) ((float)(((w) >> (8 * (c))) & 0xffu))
#define MPT_OSLAB(c, nxw, nyw, nzw, fxw, fyw, fzw, bit)                                                                \
    {                                                                                                                  \
        const float tn = fmaxf(fmaxf(__builtin_fmaf(MPT_UB(nxw, c), sx, bx), __builtin_fmaf(MPT_UB(nyw, c), sy, by)),  \
                               fmaxf(__builtin_fmaf(MPT_UB(nzw, c), sz, bz), 0.0f));                                   \
        const float tf = fminf(fminf(__builtin_fmaf(MPT_UB(fxw, c), sx, bx), __builtin_fmaf(MPT_UB(fyw, c), sy, by)),  \
                               fminf(__builtin_fmaf(MPT_UB(fzw, c), sz, bz), L.tbest));                                \
        hits |= tn <= tf ? (1u << (bit)) : 0u;                                                                         \
    }
    MPT_OSLAB(0, nx0, ny0, nz0, fx0, fy0, fz0, 0) MPT_OSLAB(1, nx0, ny0, nz0, fx0, fy0, fz0, 1)
    MPT_OSLAB(2, nx0, ny0, nz0, fx0, fy0, fz0, 2) MPT_OSLAB(3, nx0, ny0, nz0, fx0, fy0, fz0, 3)
    MPT_OSLAB(0, nx1, ny1, nz1, fx1, fy1, fz1, 4) MPT_OSLAB(1, nx1, ny1, nz1, fx1, fy1, fz1, 5)
    MPT_OSLAB(2, nx1, ny1, nz1, fx1, fy1, fz1, 6) MPT_OSLAB(3, nx1, ny1, nz1, fx1, fy1, fz1, 7)
#undef MPT_OSLAB
#undef MPT_UB
    // (an empty slot's box is inverted -- lo 255, hi 0 -- and never hit)
    const unsigned lh = hits & lmask;                          // leaf hits, by slot: their order does not matter much, all are tested
    const unsigned pih = oct_permute(hits & imask, r);         // internal hits, by the position the ray meets them in
    // leaves first (they can only shorten the ray), then the nearest internal child; what is left of either kind goes to the stack
    const bool take_leaf = lh != 0u;
    const unsigned sel = take_leaf ? lh : pih;
    const int pos = __builtin_ctz(sel | 0x100u);
    const unsigned rest = sel & (sel - 1u);
    const unsigned slot = take_leaf ? (unsigned)pos : ((unsigned)pos ^ r);
    const int a_sel = take_leaf ? a_tri : a_node;
    const int idx = (a_sel & 0xffffff) + __builtin_popcount(((unsigned)a_sel >> 24) & ((1u << slot) - 1u));
    int sp = L.sp;
    if (__ballot(sp > STACK::CAP - 2) == 0ull) {
        // nobody near the end of the LDS part: plain stores at a running index (a store that is not wanted lands on the level the
        // next one overwrites, or on the free level above the top)
        stk.base[sp * MPT_BLOCK] = a_node; stk.base[(STACK::CAP + sp) * MPT_BLOCK] = (int)pih;
        sp += (take_leaf && pih != 0u) ? 1 : 0;
        stk.base[sp * MPT_BLOCK] = a_sel; stk.base[(STACK::CAP + sp) * MPT_BLOCK] = (int)(rest | (take_leaf ? 0x80000000u : 0u));
        sp += rest != 0u ? 1 : 0;
    } else {
        if (take_leaf && pih != 0u) { stk.put(sp, a_node, (int)pih); sp++; }
        if (rest != 0u) { stk.put(sp, a_sel, (int)(rest | (take_leaf ? 0x80000000u : 0u))); sp++; }
    }
    L.sp = sp;
    if (sel != 0u) {
        L.curr = take_leaf ? ~idx : idx;
        L.st = take_leaf ? ST_LEAF : ST_NODE;
    } else oct_next(stk, L, r);
}

template <bool COUNT, class SCENE, class STACK>
DEV void stage_leaf8(const SCENE &sc, STACK &stk, LaneState &L, Cnt &cnt) {
    const int slot = ~L.curr;                                  // (a t8 index: the triangle's place in the 8-wide tree's leaf order)
    bool stop = false;
    if (COUNT) cnt.n_tri++;
    if (L.curr != L.navoid) {                                  // the triangle the ray left from is never tested (lbvh.py:329)
        MptVec4 g0, g1, g2;
        sc.tri(slot, g0, g1, g2);
        float dd, su, sv;
        if (tri_test_fast(g0, g1, g2, L.to, L.td, &dd, &su, &sv)) {
            if (L.shadow) {
                if (dd <= L.tbest) { L.hidx = slot; stop = true; }              // path.py:51: any occluder within li.dis
            } else if (dd < L.tbest) {                                          // lbvh.py:331
                L.tbest = dd; L.hidx = slot; L.hu = su; L.hv = sv;
            }
        }
    }
    if (stop) L.st = ST_DONE;
    else oct_next(stk, L, ray_octant(L));
}

#else   // MPT_OCT_KERNELS

// ---------------------------------------------------------------- gather kernel over 8-wide octant-ordered nodes (option "wide8")
template <bool COUNT>
__global__ __launch_bounds__(MPT_BLOCK, MPT_WIDE_WAVES) void render_kernel_oct(const MptRenderParams p) {
    __shared__ int s_stack[2 * OctStack::CAP * MPT_BLOCK];
    OctStack stk;
    stk.base = s_stack + threadIdx.x;
    stk.spill = p.stack_spill;
    stk.lane_off = (blockIdx.x * MPT_BLOCK + threadIdx.x) * (unsigned)(2 * OctStack::SPILL);
    stk.sp = 0;
    Cnt cnt = {};
    WorkQueue wq; wq.ctr = p.work_counter; wq.nitems = p.nitems; wq.q0 = blockIdx.x & 7; wq.qoff = 0;
    OctScene sc; sc.onode = p.onode; sc.tgeo = p.tfast;
    trace_stream<COUNT>(p, sc, stk, wq, cnt);
    finalise_tiles<MPT_FIN_GROUP_GATHER>(p);
    flush_counters<COUNT>(p, cnt);
}

MPT_KERNEL_API hipError_t mpt_oct_blocks(int grid, int count, int *blocks) {
    static std::atomic<int> occ_cache[MPT_MAX_DEVICES][2];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MPT_MAX_DEVICES) return hipErrorInvalidDevice;
    int occ = occ_cache[dev][count ? 1 : 0].load(std::memory_order_relaxed);
    if (!occ) {
        occ = count ? blocks_per_cu(render_kernel_oct<true>) : blocks_per_cu(render_kernel_oct<false>);
        occ_cache[dev][count ? 1 : 0].store(occ, std::memory_order_relaxed);
    }
    *blocks = grid * occ;
    return hipSuccess;
}

MPT_KERNEL_API hipError_t mpt_launch_render_oct(const MptRenderParams *p, int blocks, int count, hipStream_t stream) {
    if (count) hipLaunchKernelGGL((render_kernel_oct<true>), dim3(blocks), dim3(MPT_BLOCK), 0, stream, *p);
    else hipLaunchKernelGGL((render_kernel_oct<false>), dim3(blocks), dim3(MPT_BLOCK), 0, stream, *p);
    return hipGetLastError();
}

#endif
