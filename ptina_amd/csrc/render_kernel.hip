// render_kernel.hip -- the per-pixel path-trace megakernels (PathEngine._render + do_render +
// path_trace, engine/path.py:18-93) and the AOV preview kernel (engine/preview.py:23-41).
//
// Built twice from this one source: MPT_STRICT=1 -> symbols mpt_launch_*_strict,
// MPT_STRICT=0 -> mpt_launch_*_fast (see pt_device.h).
//
// Fast build (gfx950, wave64) -- see trace_stream:
//   * persistent workgroups pull (8x8 pixel tile, chunk of frames) work items from eight per-XCD queues
//     with stealing; a wave's item is a pool of samples and lanes are NOT tied to pixels: idle lanes are
//     compacted with ballot + mbcnt and handed the next samples the moment their path ends;
//   * the wave runs an in-wave state machine (NODE / LEAF steps in a tight loop, then one SHADE stage
//     per bounce, then NEW) so that each issued stage has as many ready lanes as possible;
//   * every sample's radiance is stored to a [frame][pixel] slab and a combine pass adds the frames
//     to the film in frame order: the reference's summation order, no float atomics, bit-reproducible
//     and identical for any slab split across GPUs.
//   render_kernel_fast (any scene size): 256-lane workgroups, 4 waves per SIMD, scene records gathered
//     from HBM / L2 / Infinity Cache, per-lane int32 traversal stack in LDS, [level][lane].
//   After the loop a wave that has run out of work finalises finished tiles of the film -- sum of the frames, resolve, the image's
//     write-out -- while the others drain (finalise_tiles): a launch that has the GPU to itself needs no combine pass.
//   render_kernel_wide: 4-wide nodes with 8-bit child boxes (the product path for scenes that do not fit LDS); render_kernel_oct
//     (render_oct.h, -DMPT_WITH_OCT=1: `make oct`): 8-wide octant-ordered nodes, an A/B that lost 17-19 %, kept with its tests
//     outside the product library.
//   render_kernel_lds (scenes whose node + triangle records fit the CU's 160 KiB LDS): measured on
//     MI355X the gather version spends its time in the vector L1 -- a wave's node fetch touches up to 64
//     different cache lines per load instruction, four instructions per node -- so one persistent
//     1024-lane workgroup per CU copies the records into LDS once and every traversal step becomes four
//     ds_read_b128; int16 stacks.
// Strict build: one lane per pixel, frames summed in a register in order, the reference's traversal;
//   one 16x16 tile per workgroup, blockIdx remapped so an XCD's blocks cover a contiguous run of tiles.

#include "pt_device.h"
#include "film_ops.h"
#include <atomic>

#ifndef MPT_SPEC_POP
#define MPT_SPEC_POP 1        // the stack entry a step may pop is read together with the step's node / triangle record
#endif

#ifndef MPT_ONE_START
#define MPT_ONE_START 1       // one ray-start block per shading pass (0: each stage starts its own lanes' rays, as before)
#endif
#if MPT_STRICT
#define MPT_SUFFIX(x) x##_strict
#else
#define MPT_SUFFIX(x) x##_fast
#endif

DEV int xcd_remap(int b, int nb) {
    // blocks are dealt round-robin over the 8 XCDs: give XCD k the k-th contiguous run of work
    int q = nb >> 3, r = nb & 7;
    int xcd = b & 7, k = b >> 3;
    return xcd * q + (xcd < r ? xcd : r) + k;
}

#if MPT_STRICT
struct StrictTracer {
    const MptRenderParams *p;
    int *lds;
    template <bool COUNT>
    DEV Hit closest(V3 ro, V3 rd, int avoid, Cnt &cnt) const { return bvh_closest<COUNT>(*p, lds, ro, rd, avoid, cnt); }
    template <bool COUNT>
    DEV bool occluded(V3 ro, V3 rd, int avoid, float dis, Cnt &cnt) const {
        return bvh_occluded<COUNT>(*p, lds, ro, rd, avoid, dis, cnt);
    }
};
typedef StrictTracer BlockTracer;
DEV BlockTracer make_block_tracer(const MptRenderParams &p, int *lds) {
    BlockTracer t; t.p = &p; t.lds = lds; return t;
}
#else
typedef Tracer<GlobalScene, Stack> BlockTracer;
DEV BlockTracer make_block_tracer(const MptRenderParams &p, int *lds) {
    BlockTracer t;
    t.sc.fnode = p.fnode; t.sc.tgeo = p.tfast; t.sc.soa_n = p.fnode_soa_n;
    t.st.base = lds; t.st.sp = 0;
    t.n = p.n;
    return t;
}
#endif

template <bool COUNT>
DEV void flush_counters(const MptRenderParams &p, const Cnt &c) {
    if (!COUNT) return;
    unsigned v[20] = { c.samples, c.rays, c.n_box, c.n_tri, c.n_shade, c.n_draws, c.bounces, c.n_node,
                       c.it_node, c.it_leaf, c.it_shade, c.it_new,
                       c.pl_local, c.pl_batches, c.pl_batch_lanes, c.pl_prim, c.pl_tidle, c.pl_sidle, c.pl_trips, c.pl_taken };
#pragma unroll
    for (int k = 0; k < 20; k++) {
        unsigned x = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(p.counters + k, (unsigned long long)x);
    }
}

// ---------------------------------------------------------------- one path = do_render + path_trace (path.py:18-93)
struct PathState {
    Rng rng;
    V3 ro, rd, result, throughput;
    float last_brdf_pdf;
    int avoid, depth;
};

// do_render up to the camera ray, path.py:82-90, for pixel (i, j) and batch frame f
template <bool COUNT>
DEV void path_begin(const MptRenderParams &p, PathState &s, int i, int j, int f, Cnt &cnt) {
    s.rng.dim = p.sobol_dim;
    s.rng.P = p.P + (size_t)f * p.sobol_dim;
    s.rng.i = wanghash2(i, j);                                               // path.py:72-73
    float dx = rng_random(s.rng), dy = rng_random(s.rng);
    float x = m_div((float)i + dx, (float)p.nx) * 2.0f - 1.0f;
    float y = m_div((float)j + dy, (float)p.ny) * 2.0f - 1.0f;
    camera_generate(p, x, y, &s.ro, &s.rd);
    s.avoid = -1; s.depth = 0;
    s.result = v3s(0.0f); s.throughput = v3s(1.0f); s.last_brdf_pdf = 0.0f;
    if (COUNT) { cnt.samples++; cnt.n_draws += 2; }
}

// one iteration of the path_trace loop, path.py:25-62; returns true when the path has ended
template <bool COUNT, class TR>
DEV bool path_step(const MptRenderParams &p, const TR &tr, PathState &s, Cnt &cnt) {
    if (!(s.depth < 5 && any_gt0(s.throughput) && any_ne0(s.rd))) return true;   // loop head, path.py:25
    s.depth += 1;
    if (COUNT) cnt.bounces++;

    s.rd = normalized(s.rd);
    Hit hit = tr.template closest<COUNT>(s.ro, s.rd, s.avoid, cnt);

    LightHit lit = lights_hit(p, s.ro, s.rd);
    if (lit.hit && (hit.hit == 0 || lit.dis < hit.depth)) {
        float mis = power_heuristic(s.last_brdf_pdf, lit.pdf);
        s.result = s.result + s.throughput * (lit.color * mis);
    }

    if (hit.hit == 0) {
        s.result = s.result + s.throughput * world_at(p, s.rd);
        return true;                                                         // break, path.py:39
    }
    s.avoid = hit.index;
    V3 hitpos, normal; Disney material;
    get_geometries(p, hit, s.ro, s.rd, &hitpos, &normal, material);
    if (COUNT) { cnt.n_shade++; cnt.n_draws += 6; }

    float sign = -dot(s.rd, normal);                                         // path.py:44-46 (never negative)
    if (sign < 0.0f) normal = -normal;

    LightSample li = lights_sample(p, hitpos, random3(s.rng));
    if (any_gt0(li.color)) {
        // (the candidate is a pure function of the bounce: evaluated before the shadow ray so that option "skip_dark" can
        //  leave out a ray whose candidate is exactly zero; the reference traces first and evaluates if unoccluded -- same values)
        V3 brdf_clr = disney_brdf(material, normal, sign, -s.rd, li.dir);
        float brdf_pdf = vavg(brdf_clr);
        float mis = power_heuristic(li.pdf, brdf_pdf);
        V3 direct_li = li.color * mis * brdf_clr * dot_or_zero(normal, li.dir);
        V3 direct = s.throughput * direct_li;
        if (p.skip_dark == 0 || any_ne0(direct))
            if (!tr.template occluded<COUNT>(hitpos, li.dir, s.avoid, li.dis, cnt))
                s.result = s.result + direct;
    }

    BsdfSample brdf = disney_bounce(material, normal, sign, -s.rd, random3(s.rng));
    s.throughput = s.throughput * brdf.color;
    s.ro = hitpos;
    s.rd = brdf.outdir;
    s.last_brdf_pdf = brdf.pdf;
    return false;
}

// Strict build: a lane owns ONE pixel and walks the batch's frames in order; the film sum is
// kept in a register and grows in exactly the reference's order (filmtable.py:37-39, path.py:93).
template <bool COUNT, class TR>
DEV void trace_pixel(const MptRenderParams &p, const TR &tr, int i, int j, int f, int fend, Cnt &cnt) {
    const int pix = i * p.ny + j;
    MptVec4 acc = p.film0[pix];
    PathState s;
    for (; f < fend; f++) {
        path_begin<COUNT>(p, s, i, j, f, cnt);
        while (!path_step<COUNT>(p, tr, s, cnt)) {}
        acc.x += s.result.x; acc.y += s.result.y; acc.z += s.result.z; acc.w += 1.0f;   // path.py:93
    }
    p.film0[pix] = acc;
}

#if !MPT_STRICT
// Production build: a WAVE owns an 8x8 pixel tile x the frames [f0, f1) = a pool of 64*(f1-f0)
// samples, and runs them as an in-wave state machine.  Measured on MI355X, the straightforward
// "each lane loops over its own path" megakernel is VALU-issue bound at ~14 % lane utilisation
// (SQ_THREAD_CYCLES_VALU / 64 / SQ_ACTIVE_INST_VALU): traversal trip counts, leaf tests and
// shading all diverge.  Here every lane carries a small state and the wave alternates between
//   traversal mode: a tight loop that runs ONE step per iteration for the lanes that are ready for
//       it -- NODE (two child-box tests, near child next, far child pushed) or LEAF (one triangle
//       test), whichever has more lanes -- for as long as most live lanes are traversing;
//   shading mode: lanes whose shadow ray finished add their direct light and start the next bounce;
//       lanes whose closest-hit query finished run SHADE (emitters, miss -> world, material, light
//       sample + BSDF eval, BSDF sample: the whole bounce); then NEW hands the idle lanes the next
//       samples of the pool (ballot + mbcnt compaction) and makes camera rays.
// A bounce issues its shadow ray first and keeps the next ray's direction and the candidate direct
// light C = throughput * mis * li * f * cos in registers; when the shadow traversal ends, C is
// added iff nothing was hit and the closest-hit traversal of the next bounce starts at once, so
// there is one shading stage per bounce and the order of additions into `result` is the
// reference's (path.py:31-56).  Rays, samples and sums do not depend on the schedule: each sample's
// radiance goes to p.partial[frame][column of the share][y] and the combine pass adds frames in order.
enum { ST_NODE = 0, ST_LEAF = 1, ST_DONE = 2, ST_NEW = 3, ST_DEAD = 4,     // DONE: this lane's ray is finished
       // inside one shading pass only (MPT_ONE_START): the lane's next ray starts in the pass's common block, from L.to --
       // a closest-hit ray along L.prd (head of the path_trace loop first) | a shadow ray along L.td up to L.tbest
       ST_BOUNCE = 5, ST_SHADOW = 6 };

// Per-lane state: live across the whole loop, so every word costs a VGPR for the kernel's lifetime.
struct LaneState {
    int st;
    // path, path.py:19-23
    V3 result, throughput;
    float last_brdf_pdf;
    int navoid, depth, rng_i;  // navoid: the id a node record holds for the triangle the ray left from (~slot; 0 = none:
                               // id 0 is the root, which is nobody's child)
    int rng_k;                 // rng_i reduced into [0, dim): the Sobol dimension of the lane's next draw
    int pix, frame;
    V3 prd;                    // closest ray: the path direction r.d; shadow ray: the NEXT bounce direction
    V3 direct;                 // shadow ray in flight: candidate direct light, added if unoccluded
    // ray being traversed (closest: the path ray; shadow: hitpos -> light)
    V3 to, td, inv, oinv;
    int offx, offy, offz;      // byte offset of the entry planes of each axis in a node record (LDS and wide kernels)
    float tbest;               // closest: best depth so far; shadow: li.dis, moved up one float where STACK::ONE_TEST; x t_scale while traversed (T_SCALED)
    int curr, sp, hidx;        // hidx: leaf slot of the hit so far, -1 = none (closest) / any occluder found (shadow); in the 4-wide LDS kernel
                               // curr / hidx hold ids as its LDS node records do (LdsWideScene::ODD_IDS) and sp is the LDS address of the
                               // lane's top stack entry (Stack16W::SP_ADDR), everywhere else a level
    float hu, hv;
    int shadow;                // 1: the ray in flight is a shadow ray.  An int in a VGPR on purpose: as a bool the
                               // compiler keeps it in a scalar lane mask and re-merges that mask (s_andn2 / s_and /
                               // s_or) around every divergent region of the traversal loop
};

DEV Rng lane_rng(const MptRenderParams &p, const LaneState &L) {
    Rng r; r.dim = p.sobol_dim; r.P = p.P + (size_t)L.frame * p.sobol_dim; r.i = L.rng_i; return r;
}

// Python's floor-mod of the proxy counter by the table size (sobol.py:123), without an integer division:
// an estimate of the quotient from the float reciprocal, then the remainder is put right exactly
DEV int reduce_mod_dim(int h, int dim, float inv_dim) {
    // the float estimate is off by |h| / dim * 2^-23 at most: below one for tables of >= 1024 dimensions (the
    // reference's has 21201); smaller ones take the division (wave-uniform branch)
    if (dim < 1024) return pymod(h, dim);
    int q = (int)floorf((float)h * inv_dim);                       // within +-1 of floor(h / dim)
    int r = (int)((unsigned)h - (unsigned)q * (unsigned)dim);      // exact modulo 2^32, and the true remainder is small
    if (r < 0) r += dim;
    if (r < 0) r += dim;
    if (r >= dim) r -= dim;
    if (r >= dim) r -= dim;
    return r;
}

// N consecutive draws of the lane's Sobol proxy (sobol.py:121-125).  The proxy's counter is an i32
// that the reference reduces mod dim (floor-mod) at every draw; unless the counter is about to wrap
// (probability ~N/2^32 per pixel) the N indices are k, k+1, ... with one wrap at dim: the lane carries k
// along with the counter, so a draw costs a load and a compare.  The wrapping case takes the literal path.
template <int N, bool OFF32 = false>
DEV void lane_draws(const MptRenderParams &p, LaneState &L, float *out) {
    // OFF32 (the LDS-resident kernels): the frame's row as a 32-bit word offset from the scalar base (frames x dim stays below 2^30,
    // fill_params checks) instead of 64-bit arithmetic per lane; the gather kernels keep the long form (pt_device.h shade_rec_load)
#define MPT_ROW(k_) (OFF32 ? (const float *)((const char *)p.P + ((__umul24((unsigned)L.frame, (unsigned)p.sobol_dim) + (unsigned)(k_)) << 2)) \
                           : p.P + (size_t)L.frame * p.sobol_dim + (k_))
    const float *P = MPT_ROW(0);
    const int dim = p.sobol_dim;
    if (L.rng_i <= 0x7fffffff - N && L.rng_k + N <= dim) {
        // the N numbers are consecutive words (no wrap at dim inside them): two 16-byte gathers (any 4-byte
        // alignment) instead of six -- a gather instruction costs the big scenes the same whatever its width
        struct __attribute__((packed, aligned(4))) W4 { float a, b, c, d; };
        struct __attribute__((packed, aligned(4))) W2 { float a, b; };
        const float *q = MPT_ROW(L.rng_k);
        static_assert(N == 2 || N == 6, "lane_draws: two (jitter) or six (light + BSDF triples) numbers");
        if constexpr (N == 6) {
            const W4 v = *(const W4 *)q;
            int k2 = L.rng_k + 2;
            asm("" : "+v"(k2));                                              // (or the compiler turns it into two 4-byte gathers)
            const W4 w = *(const W4 *)MPT_ROW(k2);                           // overlaps the first: no read past the six
            out[0] = v.a; out[1] = v.b; out[2] = v.c; out[3] = v.d; out[4] = w.c; out[5] = w.d;
        } else {
            const W2 w = *(const W2 *)q;
            out[0] = w.a; out[1] = w.b;
        }
        const int k = L.rng_k + N;
        L.rng_k = k == dim ? 0 : k;
        L.rng_i += N;
    } else if (L.rng_i <= 0x7fffffff - N) {
        int k = L.rng_k;
#pragma unroll
        for (int t = 0; t < N; t++) {
            out[t] = P[k];
#if MPT_X_DUP_P_LOADS
            // A/B build: every Sobol load issued twice (same film): the slowdown bounds what the loads cost
            float dup = __builtin_nontemporal_load(P + k);
            out[t] = dup == out[t] ? out[t] : dup;
#endif
            k = (k + 1 == dim) ? 0 : k + 1;
        }
        L.rng_k = k;
        L.rng_i += N;
    } else {
        Rng rng = lane_rng(p, L);
#pragma unroll
        for (int t = 0; t < N; t++) out[t] = rng_random(rng);
        L.rng_i = rng.i;
        L.rng_k = pymod(L.rng_i, dim);
    }
#undef MPT_ROW
}

// One sample's radiance into the launch's slab, path.py:93 (the combine pass or the tail finalisation adds the frames in order).
// The entry is two self-validating 8-byte granules (film_ops.h: slab_pack), each written by ONE relaxed agent-scope 64-bit atomic
// store -- single-copy atomic by the language's memory model; on gfx950 a `global_store_dwordx2 ... sc1`, i.e. write-through: it
// leaves this XCD's L2 at once, where a finishing wave of any other XCD can see it.  A half that carries the launch's tag carries
// its data, so the data is the flag: nothing to order, no fence and no read-modify-write in the shading pass (the guide's R2 form:
// cdna_hip_programming.md Guideline 16, Pitfall 8 "ONE aligned 8-B store").  Every launch stores that way, finalising or not: a
// wave-uniform choice between two store flavours in the shading pass cost the whole kernel 4 % (it is short of scalar registers).
// Measured (MI355X, same box, three alternations, profiles/r05_ab_experiments.json): 2.603-2.613 ms per launch against 2.582-2.584
// with the same entry behind ONE 16-byte sc1 store (-DMPT_SC1_STORES=16, round 4's shape, whose halves are only observed to land
// together): the second store instruction costs 0.9 %, and buys a hand-off that rests on nothing but 64-bit atomicity.
// (-DMPT_SC1_STORES=0: plain stores, an A/B build whose tail finalisation must stay off.)
#ifndef MPT_SC1_STORES
#define MPT_SC1_STORES 1
#endif
DEV void store_sample(const MptRenderParams &p, int frame, int pix, V3 radiance) {
    MptVec4 *dst = p.partial + ((size_t)frame * (size_t)p.partial_stride + pix);
    const mpt_u4 v = slab_pack(radiance.x, radiance.y, radiance.z, p.slab_tag);
#if MPT_SC1_STORES == 16
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
#elif MPT_SC1_STORES
    unsigned long long *d64 = (unsigned long long *)dst;
    __hip_atomic_store(d64, ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(d64 + 1, ((unsigned long long)v.w << 32) | v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *(mpt_u4 *)dst = v;
#endif
}

// the bottom entry of every ray's LIFO is a sentinel, so "pop" never needs an emptiness test:
// popping the sentinel means the traversal is over
template <class STACK>
DEV int classify(int v) {      // what a popped / chosen entry means for the lane's state
    if constexpr (STACK::ODD_IDS) return v & 3;      // node ids are multiples of 16 (ST_NODE == 0), leaf ids 16 * slot + 1 (ST_LEAF == 1), the sentinel is 2 (ST_DONE)
    else return v == STACK::SENTINEL ? ST_DONE : (v < 0 ? ST_LEAF : ST_NODE);
}

template <bool COUNT, class STACK>
DEV void lane_start_ray(LaneState &L, STACK &stk, V3 o, V3 d, float tmax, bool shadow, Cnt &cnt) {
    L.to = o; L.td = d;
    L.inv = v3(m_rcp(d.x), m_rcp(d.y), m_rcp(d.z));
    if constexpr (STACK::T_SCALED) { L.inv = L.inv * stk.ts; tmax *= stk.ts; }
    L.oinv = o * L.inv;
    // which of an axis' two planes the ray enters through: offset of that plane in the node record
    if constexpr (STACK::PLANE_OFF != 0) {     // (the 4-wide gather kernels read the signs off L.inv in the step: three registers less to carry)
        L.offx = __float_as_int(L.inv.x) < 0 ? STACK::PLANE_OFF : 0;
        L.offy = __float_as_int(L.inv.y) < 0 ? STACK::PLANE_OFF : 0;
        L.offz = __float_as_int(L.inv.z) < 0 ? STACK::PLANE_OFF : 0;
    }
    // a shadow ray takes any occluder with depth <= li.dis (path.py:51), a closest-hit ray a strictly nearer hit (lbvh.py:331): with
    // the shadow ray's bound moved up to the next float the LEAF step asks both the same question, depth < tbest (STACK::ONE_TEST: the
    // LDS-resident kernels, -0.5 %; the gather kernels lose 1-2 % with it and keep the two tests)
    if constexpr (STACK::ONE_TEST) {
        if (shadow) { const int b = __float_as_int(tmax); tmax = __int_as_float(b + (b < 0x7f800000 ? 1 : 0)); }
    }
    L.tbest = tmax; L.shadow = shadow ? 1 : 0; L.hidx = -1; L.hu = 0.0f; L.hv = 0.0f;
    stk.sp = 0;
    stk.push(STACK::SENTINEL);
    L.curr = 0;
    if constexpr (STACK::SP_ADDR) L.sp = stk.sp_at(1) - STACK::SP_BIAS; else L.sp = 1;
    if (COUNT) cnt.rays++;
    L.st = ST_NODE;
}

// head of the path_trace loop, path.py:25-29: either the path is over or a closest-hit ray
// starts from `ro` along L.prd
template <bool COUNT, class STACK>
DEV void lane_next_bounce(const MptRenderParams &p, LaneState &L, STACK &stk, V3 ro, Cnt &cnt) {
    if (L.depth < 5 && any_gt0(L.throughput) && any_ne0(L.prd)) {
        L.depth += 1;
        if (COUNT) cnt.bounces++;
        L.prd = normalized_unfused(L.prd);
        lane_start_ray<COUNT>(L, stk, ro, L.prd, MPT_INF, false, cnt);
        // lbvh.py:218,319: with fewer than two faces the root box is never written: no hit
        if (p.n < 2) L.st = ST_DONE;
    } else {
        store_sample(p, L.frame, L.pix, L.result);                          // path.py:93, summed by combine
        L.st = ST_NEW;
    }
}

// The loop head alone (path.py:25): a lane about to bounce whose path is over stores its sample and waits for a new one
DEV bool path_continues(const LaneState &L) { return L.depth < 5 && any_gt0(L.throughput) && any_ne0(L.prd); }
DEV void lane_store_sample(const MptRenderParams &p, LaneState &L) {
    store_sample(p, L.frame, L.pix, L.result);                              // path.py:93, summed by combine
    L.st = ST_NEW;
}
// The one place of a shading pass where rays start (MPT_ONE_START): the lanes whose shadow ray just ended, the lanes that
// shaded and the lanes that took a new sample all come here, so the direction set-up (a normalisation, three reciprocals,
// the stack reset) is issued once per pass at the width of all of them, not three times at a third each
template <bool COUNT, class STACK>
DEV void lane_begin_ray(const MptRenderParams &p, LaneState &L, STACK &stk, Cnt &cnt) {
    const bool sh = L.st == ST_SHADOW;
    const V3 n = normalized_unfused(L.prd);
    if (!sh) {
        L.depth += 1;
        if (COUNT) cnt.bounces++;
        L.prd = n;
    }
    const V3 d = sh ? L.td : n;
    lane_start_ray<COUNT>(L, stk, L.to, d, sh ? L.tbest : MPT_INF, sh, cnt);
    // lbvh.py:218,319: with fewer than two faces the root box is never written: no hit
    if (!sh && p.n < 2) L.st = ST_DONE;
}

// Traversal steps touch only (curr, sp, st) and, for leaves, the hit record: everything a finished
// ray triggers happens later, in shading mode, so the traversal loop carries no other live updates.
// They are written with two flat conditionals each (push / pop) instead of nested ones: on this
// code the nested form cost more scalar exec-mask bookkeeping than the box arithmetic itself.
// (Measured in-process A/B on MI355X and not kept: the twelve plane distances as six v_pk_fma_f32 --
//  5 % slower, packed f32 is not double-rate here; filtering the origin triangle in the leaf stage
//  instead of here -- within noise; per-stage instead of ratio scheduler thresholds -- within +-1 %.)
template <bool COUNT, class SCENE, class STACK>
DEV void stage_node(const SCENE &sc, STACK &stk, LaneState &L, Cnt &cnt) {
    int id0, id1;
    float tn0, tn1;
    bool h0, h1;
    if (COUNT) { cnt.n_node++; cnt.n_box += 2; }
#if MPT_SPEC_POP
    int spec = 0;
    if constexpr (STACK::PEEK) spec = stk.peek(L.sp - 1);      // (the sentinel sits at level 0: sp >= 1 while a ray is traversed)
#endif
    if constexpr (SCENE::SIGNED_PLANES) {
        mpt_f2 nx, fx, ny, fy, nz, fz, ids;
        sc.node_planes(L.curr, L.offx, L.offy, L.offz, nx, fx, ny, fy, nz, fz, ids);
        id0 = __float_as_int(ids.x); id1 = __float_as_int(ids.y);
        tn0 = fmaxf(fmaxf(__builtin_fmaf(nx.x, L.inv.x, -L.oinv.x), __builtin_fmaf(ny.x, L.inv.y, -L.oinv.y)),
                    fmaxf(__builtin_fmaf(nz.x, L.inv.z, -L.oinv.z), 0.0f));
        tn1 = fmaxf(fmaxf(__builtin_fmaf(nx.y, L.inv.x, -L.oinv.x), __builtin_fmaf(ny.y, L.inv.y, -L.oinv.y)),
                    fmaxf(__builtin_fmaf(nz.y, L.inv.z, -L.oinv.z), 0.0f));
        float tf0 = fminf(fminf(__builtin_fmaf(fx.x, L.inv.x, -L.oinv.x), __builtin_fmaf(fy.x, L.inv.y, -L.oinv.y)),
                          fminf(__builtin_fmaf(fz.x, L.inv.z, -L.oinv.z), L.tbest));
        float tf1 = fminf(fminf(__builtin_fmaf(fx.y, L.inv.x, -L.oinv.x), __builtin_fmaf(fy.y, L.inv.y, -L.oinv.y)),
                          fminf(__builtin_fmaf(fz.y, L.inv.z, -L.oinv.z), L.tbest));
        h0 = tn0 <= tf0; h1 = tn1 <= tf1;
    } else {
        MptVec4 a, b, c, d;
        sc.node(L.curr, a, b, c, d);
        id0 = __float_as_int(d.x); id1 = __float_as_int(d.y);
        h0 = box_fast(a.x, b.x, c.x, a.z, b.z, c.z, L.inv, L.oinv, L.tbest, &tn0);
        h1 = box_fast(a.y, b.y, c.y, a.w, b.w, c.w, L.inv, L.oinv, L.tbest, &tn1);
    }
    // a leaf that is the triangle the ray left from is never tested (lbvh.py:329)
    h0 = h0 && (id0 != L.navoid);
    h1 = h1 && (id1 != L.navoid);
    bool swap = tn1 < tn0;
    int nearid = swap ? id1 : id0, farid = swap ? id0 : id1;
    int next = h0 ? (h1 ? nearid : id0) : id1;
#if MPT_SPEC_POP
    if constexpr (STACK::PEEK) {
        // the entry a pop would return was asked for with the node record (spec, below the function's head): a step that
        // pops does not wait a second LDS round trip behind the box tests.  Push (both hit) and pop (both missed) exclude
        // each other, and a push goes to level sp, not sp - 1
        int sp = L.sp;
        if (h0 && h1) { stk.sp = sp; stk.push(farid); sp++; }
        if (!(h0 || h1)) { next = spec; sp--; }
        L.sp = sp;
    } else
#endif
    {
        stk.sp = L.sp;
        if (h0 && h1) stk.push(farid);
        if (!(h0 || h1)) next = stk.pop();
        L.sp = stk.sp;
    }
    L.curr = next;
    L.st = classify<STACK>(next);
}

// min(a, b, c, tbest) of a slab test's exit side.  Written as the two instructions themselves: through fminf the compiler first
// quiets a signalling NaN its analysis cannot rule out in tbest (a register carried round the loop) -- one v_max_f32 tbest, tbest
// per step, and min / max issue at half the rate of an FMA on gfx950.  The instructions return the same bits as fminf for every
// input that is not a signalling NaN, and nothing in the kernel makes one.  MI355X, same box, alternated three times
// (profiles/r05_ab_experiments.json): 2.593 / 2.566 / 2.554 ms per launch -> 2.560 / 2.537 / 2.526.  (The 8-bit step of the
// gather kernels, which wait for their gathers as much as for the issue port, did not move with it: C4 1547 / 1543 against 1546 / 1549.)
DEV float exit_min_asm(float a, float b, float c, float tbest) {
    float m, r;
    asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(c), "v"(tbest));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(m));
    return r;
}
DEV float exit_min(float a, float b, float c, float tbest) { return exit_min_asm(a, b, c, tbest); }

// The same step through a 4-wide node: four slab tests (planes picked by the ray's direction signs) on one 128-B
// record, the children that are hit sorted
// by entry distance (a five-comparator network on (distance bits, id) pairs; a miss sorts last), the nearest
// taken next and the others pushed farthest first.
#ifndef MPT_SORT_PACKED
#define MPT_SORT_PACKED 1     // the 4-wide step of the LDS-resident kernel sorts (distance bits | 16-bit id) words (0: (key, id) pairs)
#endif
template <bool COUNT, class SCENE, class STACK>
DEV void stage_node4(const SCENE &sc, STACK &stk, LaneState &L, Cnt &cnt) {
    int id0, id1, id2, id3;
    float t0, t1, t2, t3;
    bool h0, h1, h2, h3;
    if (COUNT) { cnt.n_node++; cnt.n_box += 4; }
    // the entry a step without a hit pops is asked for together with the node record: some lane of the wave pops in nearly every
    // step, and the wave then waited a second LDS round trip behind the sort (pushes go above the top entry, never onto it)
    int spec = 0;
    if constexpr (STACK::SP_ADDR) spec = STACK::ld(L.sp - STACK::SP_STEP + STACK::SP_BIAS);
    if constexpr (SCENE::QUANT) {
        MptVec4 ra, rb, rc, idv;
        sc.node4q(L.curr, ra, rb, rc, idv);
        id0 = __float_as_int(idv.x); id1 = __float_as_int(idv.y); id2 = __float_as_int(idv.z); id3 = __float_as_int(idv.w);
        // plane = origin + q * scale, so its distance along the ray is q * (scale * inv) + (origin * inv - o * inv)
        const float sx = ra.w * L.inv.x, sy = rb.x * L.inv.y, sz = rb.y * L.inv.z;
        const float bx = __builtin_fmaf(ra.x, L.inv.x, -L.oinv.x), by = __builtin_fmaf(ra.y, L.inv.y, -L.oinv.y),
                    bz = __builtin_fmaf(ra.z, L.inv.z, -L.oinv.z);
        const unsigned lox = (unsigned)__float_as_int(rb.z), hix = (unsigned)__float_as_int(rb.w);
        const unsigned loy = (unsigned)__float_as_int(rc.x), hiy = (unsigned)__float_as_int(rc.y);
        const unsigned loz = (unsigned)__float_as_int(rc.z), hiz = (unsigned)__float_as_int(rc.w);
        // entry planes: the low ones for a ray going up the axis, the high ones for one going down (L.off*: per-ray flags)
        const bool dnx = __float_as_int(L.inv.x) < 0, dny = __float_as_int(L.inv.y) < 0, dnz = __float_as_int(L.inv.z) < 0;
        const unsigned nxq = dnx ? hix : lox, fxq = dnx ? lox : hix;
        const unsigned nyq = dny ? hiy : loy, fyq = dny ? loy : hiy;
        const unsigned nzq = dnz ? hiz : loz, fzq = dnz ? loz : hiz;
#define MPT_UB(w, c) ((float)(((w) >> (8 * (c))) & 0xffu))
#define MPT_QSLAB(c, tn, h)                                                                                            \
        tn = fmaxf(fmaxf(__builtin_fmaf(MPT_UB(nxq, c), sx, bx), __builtin_fmaf(MPT_UB(nyq, c), sy, by)),               \
                   fmaxf(__builtin_fmaf(MPT_UB(nzq, c), sz, bz), 0.0f));                                                 \
        h = tn <= fminf(fminf(__builtin_fmaf(MPT_UB(fxq, c), sx, bx), __builtin_fmaf(MPT_UB(fyq, c), sy, by)),          \
                        fminf(__builtin_fmaf(MPT_UB(fzq, c), sz, bz), L.tbest));
        MPT_QSLAB(0, t0, h0) MPT_QSLAB(1, t1, h1) MPT_QSLAB(2, t2, h2) MPT_QSLAB(3, t3, h3)
#undef MPT_QSLAB
#undef MPT_UB
    } else {
        MptVec4 nx, fx, ny, fy, nz, fz, idv;
        if constexpr (STACK::PLANE_OFF != 0) sc.node4(L.curr, L.offx, L.offy, L.offz, nx, fx, ny, fy, nz, fz, idv);     // (per-ray constants)
        else sc.node4(L.curr, __float_as_int(L.inv.x) < 0 ? 16 : 0, __float_as_int(L.inv.y) < 0 ? 16 : 0, __float_as_int(L.inv.z) < 0 ? 16 : 0,
                      nx, fx, ny, fy, nz, fz, idv);
        id0 = __float_as_int(idv.x); id1 = __float_as_int(idv.y); id2 = __float_as_int(idv.z); id3 = __float_as_int(idv.w);
#define MPT_SLAB(c, tn, h)                                                                                              \
        tn = fmaxf(fmaxf(__builtin_fmaf(nx.c, L.inv.x, -L.oinv.x), __builtin_fmaf(ny.c, L.inv.y, -L.oinv.y)),            \
                   STACK::T_SCALED ? __builtin_amdgcn_fmed3f(__builtin_fmaf(nz.c, L.inv.z, -L.oinv.z), 0.0f, 1.0f)       \
                                   : fmaxf(__builtin_fmaf(nz.c, L.inv.z, -L.oinv.z), 0.0f));                             \
        h = tn <= exit_min(__builtin_fmaf(fx.c, L.inv.x, -L.oinv.x), __builtin_fmaf(fy.c, L.inv.y, -L.oinv.y),          \
                           __builtin_fmaf(fz.c, L.inv.z, -L.oinv.z), L.tbest);
        MPT_SLAB(x, t0, h0) MPT_SLAB(y, t1, h1) MPT_SLAB(z, t2, h2) MPT_SLAB(w, t3, h3)
#undef MPT_SLAB
    }
    // entry distances are >= 0, so their bit patterns order like the values; a miss (or the triangle the ray
    // left from, lbvh.py:329) gets the largest key
    const unsigned MISS = 0xffffffffu;
    unsigned k0, k1, k2, k3;
    if constexpr (MPT_SORT_PACKED && sizeof(typename STACK::entry_t) == 2) {
        // 16-bit ids (the LDS-resident kernel): the upper half of the distance's bits over the id is ONE word that sorts with
        // v_min_u32 / v_max_u32 -- ten instructions instead of the 25 of five compare-and-swaps on (key, id) pairs; distances that
        // agree in their first 8 mantissa bits are met in id order, which costs a step now and then and never a hit (the
        // order only decides what is looked at first)
        if constexpr (!SCENE::AVOID_IN_LEAF) { h0 = h0 && id0 != L.navoid; h1 = h1 && id1 != L.navoid; h2 = h2 && id2 != L.navoid; h3 = h3 && id3 != L.navoid; }
        k0 = h0 ? __builtin_amdgcn_perm((unsigned)__float_as_int(t0), (unsigned)id0, 0x07060100u) : MISS;
        k1 = h1 ? __builtin_amdgcn_perm((unsigned)__float_as_int(t1), (unsigned)id1, 0x07060100u) : MISS;
        k2 = h2 ? __builtin_amdgcn_perm((unsigned)__float_as_int(t2), (unsigned)id2, 0x07060100u) : MISS;
        k3 = h3 ? __builtin_amdgcn_perm((unsigned)__float_as_int(t3), (unsigned)id3, 0x07060100u) : MISS;
        const unsigned a0 = min(k0, k1), a1 = max(k0, k1), b0 = min(k2, k3), b1 = max(k2, k3);
        const unsigned m0 = max(a0, b0), m1 = min(a1, b1);
        k0 = min(a0, b0); k3 = max(a1, b1); k1 = min(m0, m1); k2 = max(m0, m1);    // (measured and not kept: without this fifth
        // comparator -- the middle pair in whatever order the network leaves it -- the step is two instructions shorter and the launch 1.8 % longer)
        id0 = STACK::ODD_IDS ? (int)(k0 & 0xffffu) : (int)(short)(k0 & 0xffffu);
        id1 = (int)k1; id2 = (int)k2; id3 = (int)k3;                                         // (the pushes store the low halves)
    } else {
        if constexpr (!SCENE::AVOID_IN_LEAF) { h0 = h0 && id0 != L.navoid; h1 = h1 && id1 != L.navoid; h2 = h2 && id2 != L.navoid; h3 = h3 && id3 != L.navoid; }
        k0 = h0 ? (unsigned)__float_as_int(t0) : MISS;
        k1 = h1 ? (unsigned)__float_as_int(t1) : MISS;
        k2 = h2 ? (unsigned)__float_as_int(t2) : MISS;
        k3 = h3 ? (unsigned)__float_as_int(t3) : MISS;
#define MPT_CSWAP(ka, ia, kb, ib) { bool sw = kb < ka; unsigned tk = sw ? kb : ka; kb = sw ? ka : kb; ka = tk; \
                                    int ti_ = sw ? ib : ia; ib = sw ? ia : ib; ia = ti_; }
        MPT_CSWAP(k0, id0, k1, id1) MPT_CSWAP(k2, id2, k3, id3) MPT_CSWAP(k0, id0, k2, id2) MPT_CSWAP(k1, id1, k3, id3)
        MPT_CSWAP(k1, id1, k2, id2)
#undef MPT_CSWAP
    }
    int next = id0;
    if (STACK::NO_SPILL || __ballot(L.sp > STACK::CAP - 3) == 0ull) {
        // no lane of the wave is within three entries of the LDS part of its stack (the rule, not the exception; the LDS-resident
        // kernel's stack holds every level the tree can ask for): the three pushes are plain stores at a running index -- a store
        // that is not wanted lands on the slot the next one overwrites -- instead of three divergent regions with a spill test each
        typedef typename STACK::entry_t entry_t;
        int sp = L.sp;
        if constexpr (STACK::SP_ADDR) {                                               // (sp: the address of the top entry, Stack16W)
            STACK::st(sp + STACK::SP_BIAS, id3); sp += k3 != MISS ? STACK::SP_STEP : 0;
            STACK::st(sp + STACK::SP_BIAS, id2); sp += k2 != MISS ? STACK::SP_STEP : 0;
            STACK::st(sp + STACK::SP_BIAS, id1); sp += k1 != MISS ? STACK::SP_STEP : 0;
            if (k0 == MISS) { sp -= STACK::SP_STEP; next = spec; }
        } else {
            stk.base[sp * STACK::STRIDE] = (entry_t)id3; sp += k3 != MISS ? 1 : 0;
            stk.base[sp * STACK::STRIDE] = (entry_t)id2; sp += k2 != MISS ? 1 : 0;
            stk.base[sp * STACK::STRIDE] = (entry_t)id1; sp += k1 != MISS ? 1 : 0;
            if (k0 == MISS) { sp--; next = (int)stk.base[sp * STACK::STRIDE]; }       // sorted: then nothing was pushed
        }
        L.sp = sp;
    } else {
        stk.sp = L.sp;
        if (k3 != MISS) stk.push(id3);
        if (k2 != MISS) stk.push(id2);
        if (k1 != MISS) stk.push(id1);
        if (k0 == MISS) next = stk.pop();
        L.sp = stk.sp;
    }
    L.curr = next;
    L.st = classify<STACK>(next);
}

#if MPT_WITH_OCT
#include "render_oct.h"      // stage_node8 / stage_leaf8 / oct_next: the 8-wide octant-ordered tree's steps (A/B build)
#endif

template <bool COUNT, class SCENE, class STACK>
DEV void stage_leaf(const SCENE &sc, STACK &stk, LaneState &L, Cnt &cnt) {
    int slot = SCENE::ODD_IDS ? L.curr : ~L.curr;      // (ODD_IDS: the leaf's id stands for the slot until a shading pass needs it)
    bool stop = false;
    // (the counters count the reference's work: it never tests the triangle a ray left from, lbvh.py:329)
    if (COUNT) cnt.n_tri += (SCENE::AVOID_IN_LEAF && L.curr == L.navoid) ? 0u : 1u;
#if MPT_SPEC_POP
    int spec = 0;
    if constexpr (STACK::SP_ADDR) spec = STACK::ld(L.sp - STACK::SP_STEP + STACK::SP_BIAS);
    else if constexpr (STACK::PEEK) spec = stk.peek(L.sp - 1);      // a leaf step always pops: asked for with the triangle record
#endif
    MptVec4 g0, g1, g2;
    sc.tri(slot, g0, g1, g2);
    float dd, su, sv;
    bool hit = tri_test_fast(g0, g1, g2, L.to, L.td, &dd, &su, &sv);
    if constexpr (STACK::T_SCALED) dd *= stk.ts;                            // (L.tbest is held scaled while the ray is traversed)
    if constexpr (SCENE::AVOID_IN_LEAF) hit = hit && L.curr != L.navoid;    // the triangle the ray left from (lbvh.py:329): the NODE step let it through
    if constexpr (STACK::ONE_TEST) {
        if (hit && dd < L.tbest) {                                          // lbvh.py:331; path.py:51 (lane_start_ray)
            L.tbest = dd; L.hidx = slot; L.hu = su; L.hv = sv;
            stop = L.shadow != 0;
        }
    } else if (hit) {
        if (L.shadow) {
            if (dd <= L.tbest) { L.hidx = slot; stop = true; }              // path.py:51: any occluder within li.dis
        } else if (dd < L.tbest) {                                          // lbvh.py:331
            L.tbest = dd; L.hidx = slot; L.hu = su; L.hv = sv;
        }
    }
    int next;
#if MPT_SPEC_POP
    if constexpr (STACK::PEEK) {
#if MPT_X_LEAFPAIRS      // diagnostic build (counting kernels): how often the entry under a leaf is another leaf (what a two-triangle LEAF step could take along)
        if (COUNT) { cnt.pl_trips++; if (classify<STACK>(spec) == ST_LEAF && !stop) cnt.pl_local++; }
#endif
        next = spec; L.sp = L.sp - (STACK::SP_ADDR ? STACK::SP_STEP : 1);
    } else
#endif
    {
        stk.sp = L.sp;
        next = stk.pop();
        L.sp = stk.sp;
    }
    L.curr = next;
    L.st = stop ? ST_DONE : classify<STACK>(next);
}

// a shadow ray has finished: add the candidate direct light if nothing was hit (path.py:51,56),
// then the next bounce starts from hitpos (= the shadow ray's origin), path.py:60
template <bool COUNT, class STACK>
DEV void stage_shadow_done(const MptRenderParams &p, LaneState &L, STACK &stk, Cnt &cnt) {
    if (L.hidx < 0) L.result = L.result + L.direct;
    lane_next_bounce<COUNT>(p, L, stk, L.to, cnt);
}

// path.py:31-62 for one bounce.  On entry L.to / L.prd are the path ray r.o / r.d and
// (L.hidx >= 0, L.tbest, L.hidx, L.hu, L.hv) the closest hit.  shade_core is the bounce itself; what follows it --
// a shadow ray from hitpos towards the sampled light, or the next bounce from hitpos -- is the caller's: the wave that
// shaded starts it in the same lane (stage_shade), or hands it to another wave through the workgroup's ray pool.
enum { SH_END = 0, SH_BOUNCE = 1, SH_SHADOW = 2 };
// Diagnostic build -DMPT_X_STAMPS=2 (counting kernels): shader-clock cycles (units of 16) of the segments of SHADE, added by the
// first active lane into the pl_* counters: lights hit | geometry + material (waits for the gathers) | light sample |
// BSDF eval + MIS | BSDF sample | ray start
#if MPT_X_STAMPS == 2
#define MPT_SEG_BEGIN unsigned long long seg_t = 0; if (COUNT) { __builtin_amdgcn_sched_barrier(0); seg_t = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define MPT_SEG(field) if (COUNT) { __builtin_amdgcn_sched_barrier(0); const unsigned long long seg_n = __builtin_amdgcn_s_memtime(); \
        const unsigned long long seg_m = __ballot(true); \
        const bool seg_first = __builtin_amdgcn_mbcnt_hi((unsigned)(seg_m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)seg_m, 0u)) == 0; \
        cnt.field += seg_first ? (unsigned)((seg_n - seg_t) >> 4) : 0u; seg_t = seg_n; __builtin_amdgcn_sched_barrier(0); }
#else
#define MPT_SEG_BEGIN
#define MPT_SEG(field)
#endif   // path over (miss: world light added) | next bounce from hitpos | shadow ray first
template <bool COUNT, class SCENE>
DEV int shade_core(const MptRenderParams &p, const SCENE &sc, LaneState &L, Cnt &cnt, V3 &hitpos, V3 &sdir, float &sdis) {
    V3 ro = L.to, rd = L.prd;
    const bool was_hit = L.hidx >= 0;
    float hdepth = was_hit ? (SCENE::T_SCALED ? L.tbest * p.t_unscale : L.tbest) : MPT_INF;
    // everything the stage gathers from L2 is asked for first: the shading record of the triangle and the six
    // Sobol numbers of the bounce (path.py:48,58: light triple, then BSDF triple) -- one round trip, under the
    // light tests, instead of three in a row
    // (the LDS-resident kernels: no initialisers -- both are read by lanes with a hit only, and "= {}" was 21 v_mov_b32 per stage; the
    //  gather kernels keep them: without, their register allocation spills 16 bytes more and loses 2 %)
    ShadeRec rec;
    float u[6];
    if constexpr (!SCENE::LDS_MATS) { rec = ShadeRec{}; for (int k = 0; k < 6; k++) u[k] = 0.0f; }
    MPT_SEG_BEGIN
    const int hslot = SCENE::ODD_IDS ? (L.hidx >> 4) : L.hidx;
    if (was_hit) {
        rec = shade_rec_load<SCENE::LDS_MATS>(p, hslot);
        lane_draws<6, SCENE::LDS_MATS>(p, L, u);
    }
    MPT_SEG(pl_trips)            // (the entry of the stage -- reloads of what the traversal loop had parked -- and the issue of its gathers)
    LightHit lit = lights_hit(p, ro, rd);
    if (lit.hit && (!was_hit || lit.dis < hdepth)) {
        float mis = power_heuristic(L.last_brdf_pdf, lit.pdf);
        L.result = L.result + L.throughput * (lit.color * mis);
    }
    hitpos = ro; sdir = v3s(0.0f); sdis = 0.0f;
    MPT_SEG(pl_local)
    if (!was_hit) {
        L.result = L.result + L.throughput * world_at(p, rd);
        L.depth = 5;                                                         // break, path.py:39
        return SH_END;
    }
    L.navoid = SCENE::ODD_IDS ? L.hidx : ~L.hidx;
    Hit hit; hit.hit = 1; hit.depth = hdepth; hit.index = hslot; hit.u = L.hu; hit.v = L.hv;
    V3 normal; Disney mat;
    get_geometries_rec(p, sc, rec, hit, ro, rd, &hitpos, &normal, mat);
    if (COUNT) { cnt.n_shade++; cnt.n_draws += 6; }
    float sign = -dot(rd, normal);                                           // path.py:44-46 (never negative)
    if (sign < 0.0f) normal = -normal;
    MPT_SEG(pl_batches)

    LightSample li = lights_sample(p, hitpos, v3(u[0], u[1], u[2]));
    bool want_shadow = any_gt0(li.color);
    MPT_SEG(pl_batch_lanes)
    L.direct = v3s(0.0f);
    if (want_shadow) {
        // evaluated before the visibility is known; dropped if the shadow ray hits (path.py:50-56)
        V3 brdf_clr = disney_brdf(mat, normal, sign, -rd, li.dir);
        float brdf_pdf = vavg(brdf_clr);
        float mis = power_heuristic(li.pdf, brdf_pdf);
        V3 direct_li = li.color * mis * brdf_clr * dot_or_zero(normal, li.dir);
        L.direct = L.throughput * direct_li;
    }
    MPT_SEG(pl_prim)
    BsdfSample brdf = disney_bounce(mat, normal, sign, -rd, v3(u[3], u[4], u[5]));
    L.throughput = L.throughput * brdf.color;
    L.prd = brdf.outdir;
    L.last_brdf_pdf = brdf.pdf;
    MPT_SEG(pl_tidle)
    // A shadow ray decides whether `direct` is added (path.py:50-56).  When direct is exactly zero -- the light is behind the
    // surface (cos = 0), a black lobe, a dead throughput -- adding it or not is the same bits, so the ray is not traced:
    // an exact elimination (x + 0 == x; a NaN is != 0 and still takes the ray).  On the benchmark scene that is every
    // surface that faces away from the light: 3.5 % of all rays, 8 % of the node fetches (they are the long ones), -5 % time.
    // Option "skip_dark" = 0 traces them like the reference does.  In the strict build (no contraction) the two settings give the
    // same film bit for bit (tested); in this build a handful of pixels differ in the last bits, because the bounce that follows
    // a skipped ray starts from another inlined copy of lane_next_bounce than the one behind stage_shadow_done, and
    // -ffp-contract=fast fuses normalized()'s multiply-adds differently in the two copies.
    if (want_shadow && p.n >= 2 && (p.skip_dark == 0 || any_ne0(L.direct))) {
        sdir = li.dir; sdis = li.dis;
        return SH_SHADOW;
    }
    if (want_shadow && p.n < 2) { L.result = L.result + L.direct; if (COUNT) cnt.rays++; }   // no geometry to occlude
    return SH_BOUNCE;
}

template <bool COUNT, class SCENE, class STACK>
DEV void stage_shade(const MptRenderParams &p, const SCENE &sc, LaneState &L, STACK &stk, Cnt &cnt) {
    V3 hitpos, sdir;
    float sdis;
    const int next = shade_core<COUNT>(p, sc, L, cnt, hitpos, sdir, sdis);
    MPT_SEG_BEGIN
    if (next == SH_SHADOW) lane_start_ray<COUNT>(L, stk, hitpos, sdir, sdis, true, cnt);
    else lane_next_bounce<COUNT>(p, L, stk, hitpos, cnt);                    // SH_END: depth is 5, the sample is stored
    MPT_SEG(pl_sidle)
}

// do_render up to the camera ray, path.py:82-90, in two halves.  A wave prepares the primary rays of the next 64
// samples of its work item with all lanes on (lane l: sample base + l) and keeps them in eight registers; a lane
// whose path has ended fetches the ray of the sample it is handed with ds_bpermute.  Lanes finish a few at a time
// (a NEW pass found 6 of 64 lanes waiting on average), so the hash, the two Sobol loads and the camera
// transform ran at a tenth of the vector width when every lane prepared its own.
struct PrimaryPool {
    V3 ro, rd;
    int rng_i, rng_k;          // the pixel's proxy after the two jitter draws; rng_k < 0: no such pixel (tile past the edge)
};
DEV void pool_prepare(const MptRenderParams &p, PrimaryPool &pp, bool inside, int i, int j, int frame) {
    pp.ro = v3s(0.0f); pp.rd = v3s(0.0f); pp.rng_i = 0; pp.rng_k = -1;
    if (inside) {
        LaneState T;
        T.frame = frame;
        T.rng_i = wanghash2(i, j);                                           // path.py:72-73
        T.rng_k = reduce_mod_dim(T.rng_i, p.sobol_dim, p.sobol_inv_dim);
        float jit[2];
        lane_draws<2>(p, T, jit);                                            // random2: dx then dy, path.py:87
        float x = m_div((float)i + jit[0], (float)p.nx) * 2.0f - 1.0f;
        float y = m_div((float)j + jit[1], (float)p.ny) * 2.0f - 1.0f;
        camera_generate(p, x, y, &pp.ro, &pp.rd);
        pp.rng_i = T.rng_i; pp.rng_k = T.rng_k;
    }
}
DEV float lane_from(float v, int byte_lane) { return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_lane, __float_as_int(v))); }
DEV int lane_from(int v, int byte_lane) { return __builtin_amdgcn_ds_bpermute(byte_lane, v); }

#ifndef MPT_PREF_NODE
#define MPT_PREF_NODE 1     // a NODE step when nodes * MPT_PREF_NODE >= leaves * MPT_PREF_LEAF, else a LEAF step
#define MPT_PREF_LEAF 1
#endif
#ifndef MPT_LEAVE_A
#define MPT_LEAVE_A 2    // leave traversal mode when traversing * A < waiting * B
#define MPT_LEAVE_B 1
#endif
// Diagnostic build (-DMPT_X_STAMPS=1, counting kernels only): the shader-clock cycles each wave spends in each
// stage, accumulated into the counters named in MPT_STAMP_END instead of their usual meaning (tools/gpu_diag.py stamps)
#if MPT_X_STAMPS
#define MPT_STAMP_BEGIN unsigned long long stamp_t0 = 0; if (COUNT) { __builtin_amdgcn_sched_barrier(0); stamp_t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define MPT_STAMP_END(acc) if (COUNT) { __builtin_amdgcn_sched_barrier(0); acc += __builtin_amdgcn_s_memtime() - stamp_t0; __builtin_amdgcn_sched_barrier(0); }
#else
#define MPT_STAMP_BEGIN
#define MPT_STAMP_END(acc)
#endif
// diagnostics (counting kernels, option "lane_hist"): one issued stage -- how many lanes took part, and whose (depth, ray kind) they were
DEV void lane_hist_add(const MptRenderParams &p, int stage, bool part, int depth, int shadow) {
    unsigned long long *h = p.counters + MPT_HIST_BASE;
    const bool l0 = (threadIdx.x & 63) == 0;
    const int n = (int)__builtin_popcountll(__ballot(part));
    if (l0) atomicAdd(h + stage * 65 + n, 1ull);
    for (int d = 0; d < 6; d++)
        for (int k = 0; k < 2; k++) {
            const int c = (int)__builtin_popcountll(__ballot(part && min(depth, 5) == d && (shadow != 0) == (k != 0)));
            if (l0 && c) atomicAdd(h + 3 * 65 + (stage * 6 + d) * 2 + k, (unsigned long long)c);
        }
}
// diagnostics: a NODE stage's lane-steps by the bucket of the node's number (0 | 1 | 2-3 | 4-7 | ...): the 4-wide nodes are numbered
// breadth first, so "number < N" is "the top of the tree" -- what share of the fetches a cache of the top N records would serve
DEV void node_id_hist_add(const MptRenderParams &p, bool part, int id) {
    unsigned long long *h = p.counters + MPT_HIST_BASE + 3 * 65 + 3 * 6 * 2;
    const int b = id <= 0 ? 0 : 32 - __builtin_clz((unsigned)id);
    for (int k = 0; k < 24; k++) {
        const int c = (int)__builtin_popcountll(__ballot(part && b == k));
        if ((threadIdx.x & 63) == 0 && c) atomicAdd(h + k, (unsigned long long)c);
    }
}
DEV int wave_count(bool pred) { return (int)__builtin_popcountll(__ballot(pred)); }
DEV int wave_count32(bool pred) {            // a count that stays on the scalar unit when compared
    unsigned long long m = __ballot(pred);
    int n;                                   // one s_bcnt1_i32_b64, written out: the compiler's own 64-bit popcount ends up compared on the VALU, and
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(n) : "s"(m) : "scc");     // two 32-bit ones are three scalar instructions in the chain in front of every step
    return n;                                // (MI355X: 2.462 / 2.464 / 2.463 ms per launch -> 2.449 / 2.453 / 2.460; the gather kernels +0.8 %)
}

// (SCENE::OCT is only ever true in the A/B build with the 8-wide kernel: render_oct.h)
template <bool COUNT, class SCENE, class STACK>
DEV void stage_node8_if_built(const SCENE &sc, STACK &stk, LaneState &L, Cnt &cnt) {
#if MPT_WITH_OCT
    stage_node8<COUNT>(sc, stk, L, cnt);
#endif
}
template <bool COUNT, class SCENE, class STACK>
DEV void stage_leaf8_if_built(const SCENE &sc, STACK &stk, LaneState &L, Cnt &cnt) {
#if MPT_WITH_OCT
    stage_leaf8<COUNT>(sc, stk, L, cnt);
#endif
}

// Work items = (8x8 pixel tile, chunk of frames), tile-major, split into 8 contiguous ranges with
// one counter each.  A wave starts on the range of its XCD (blocks b, b+8, ... share an XCD) and
// moves on to the next range when one runs dry, so neighbouring tiles are traced by CUs behind the
// same L2 for as long as there is local work; every wave leaves when all eight ranges are exhausted.
// (Round 4, measured and taken out again -- profiles/r04_ab_experiments.json: a tapered end of launch, the younger waves of a SIMD
//  leaving the last items to the older.  Told by a look at the eight heads it made the launch 2.1-2.7 x slower -- which is how the
//  heads' shared cache line was found, mpt_types.h MPT_QUEUE_STRIDE -- and told by the pull's own result, free of any memory
//  access, 1-2 % slower: the end of a launch wants every wave it can get.  Also: the pull's atomic issued 8 / 16 / 32 samples ahead
//  of need: 2.89 / 2.89 / 2.92 against 2.88 ms -- its round trip is already hidden behind the wave's other lanes.)
struct WorkQueue {
    unsigned int *ctr;
    int nitems, q0, qoff;
    DEV int pull() {           // wave-uniform; -1 = no work left anywhere
        const int lane = threadIdx.x & 63;
        while (qoff < 8) {
            int q = (q0 + qoff) & 7;
            int lo = (int)(((long long)nitems * q) >> 3), hi = (int)(((long long)nitems * (q + 1)) >> 3);
            int k = 0;
            if (lane == 0) k = (int)atomicAdd(ctr + q * MPT_QUEUE_STRIDE, 1u);
            k = __builtin_amdgcn_readfirstlane(k);
            if (lo + k < hi) return lo + k;
            qoff++;
        }
        return -1;
    }
};

template <bool COUNT, class SCENE, class STACK>
DEV void trace_stream(const MptRenderParams &p, const SCENE &sc, STACK stk, WorkQueue wq, Cnt &cnt,
                      unsigned long long *tl = nullptr) {
    // work-item tiles are 2^tw_shift x 2^th_shift pixels (8x8 by default; smaller tiles shorten the
    // end-of-launch skew between waves at the price of primary-ray coherence)
    const int tws = p.tile_w_shift, ths = p.tile_h_shift, tps = tws + ths;
    const int t8y = (p.ny + (1 << ths) - 1) >> ths;
    int S = 0, next = 0;                            // wave-uniform: current pool = 64*frames samples; next unassigned
    int ti = 0, tj = 0, f0 = 0, tx_cur = 0;
    int ndead = 0;                                  // wave-uniform: lanes that have left for good
    int deferred = 0;                               // wave-uniform: lanes whose SHADE the last pass put off (MPT_SHADE_MIN)
    bool more = true;
    PrimaryPool pool;                               // lane l: primary ray of sample pool_base + l of the current item
    pool.ro = v3s(0.0f); pool.rd = v3s(0.0f); pool.rng_i = 0; pool.rng_k = -1;
    int pool_base = -64;
#if MPT_X_TIMELINE2      // diagnostic build: also the time of the last work item pulled, their number, the lanes in flight when the
    int tl_items = 0, tl_passes = 0;     // queues were found empty and the shading passes made after that (timeline words 4..7)
    unsigned long long tl_lastpull = 0;
#endif
#if MPT_X_STAMPS
    unsigned long long acc_node = 0, acc_leaf = 0, acc_sdone = 0, acc_shade = 0, acc_new = 0;
    const unsigned long long stamp_start = __builtin_amdgcn_s_memtime();
#endif
    LaneState L;
    L.st = ST_NEW;
    L.sp = 0; L.curr = 0; L.shadow = 0;
    L.result = v3s(0.0f); L.throughput = v3s(0.0f); L.prd = v3s(0.0f); L.direct = v3s(0.0f);
    L.to = v3s(0.0f); L.td = v3s(0.0f); L.inv = v3s(0.0f); L.oinv = v3s(0.0f);
    L.offx = 0; L.offy = 0; L.offz = 0;
    L.tbest = 0.0f; L.hidx = -1; L.hu = 0.0f; L.hv = 0.0f; L.last_brdf_pdf = 0.0f;
    L.navoid = 0; L.depth = 0; L.rng_i = 0; L.rng_k = 0; L.pix = 0; L.frame = 0;
    // Every pass of this loop retires at least one stage for at least one lane, so it ends when the
    // queues are empty.  The pass counter is a watchdog only: a scheduling bug must not be able to keep
    // a persistent wave (and with it the GPU) spinning -- the host turns the flag into an error.
    for (unsigned guard = 0;; guard++) {
        if (guard > (1u << 26)) {
            if ((threadIdx.x & 63) == 0) __hip_atomic_store(p.watchdog, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        // ---- traversal mode: tight loop while the lanes that are traversing outnumber the waiting ones
        for (;;) {
            // The decision in front of every step is a chain VALU compare -> scalar count -> scalar compare ->
            // branch that a wave cannot overlap with anything of its own (stamped: a fifth of its cycles went
            // there), so it is kept short: two ballots, counts in 32-bit scalar registers (a 64-bit popcount makes
            // the compiler compare on the VALU), the waiting lanes by subtraction, one branch per condition.
            const int cn = wave_count32(L.st == ST_NODE);
            const int cl = wave_count32(L.st == ST_LEAF);
            const int trav = cn + cl;
            if (trav == 0) break;
            // leave when the waiting lanes (DONE or NEW: everything alive that is not traversing) outnumber the
            // traversing ones 2 : 1 (best of the ratios tried on MI355X)
            if (trav * MPT_LEAVE_A < (64 - ndead - trav - deferred) * MPT_LEAVE_B) break;
            MPT_STAMP_BEGIN
#if MPT_X_PAIRS
            // Diagnostic build (counting kernels): what a lane that carried TWO paths could join.  Lanes i and i + 32 stand for the two
            // paths of one such lane: the pairs with at least one path ready for the step, summed per scheduling decision into pl_local
            // (NODE) / pl_batches (LEAF) / pl_batch_lanes (SHADE), the ready lanes into pl_prim / pl_tidle / pl_sidle (tools/pairs.py)
            const unsigned long long mn = __ballot(L.st == ST_NODE), ml = __ballot(L.st == ST_LEAF);      // (every lane votes)
            if (COUNT && (threadIdx.x & 63) == 0) {
                if (cn * MPT_PREF_NODE >= cl * MPT_PREF_LEAF) { cnt.pl_local += (unsigned)__builtin_popcount((unsigned)mn | (unsigned)(mn >> 32)); cnt.pl_prim += (unsigned)__builtin_popcountll(mn); cnt.pl_trips++; }
                else { cnt.pl_batches += (unsigned)__builtin_popcount((unsigned)ml | (unsigned)(ml >> 32)); cnt.pl_tidle += (unsigned)__builtin_popcountll(ml); cnt.pl_taken++; }
            }
#endif
            if (cn * MPT_PREF_NODE >= cl * MPT_PREF_LEAF) {
                if (COUNT && (threadIdx.x & 63) == 0) cnt.it_node++;
                if (COUNT && p.lane_hist) lane_hist_add(p, 0, L.st == ST_NODE, L.depth, L.shadow);
                if constexpr (!STACK::ODD_IDS) { if (COUNT && p.lane_hist) node_id_hist_add(p, L.st == ST_NODE, L.curr); }
                if (L.st == ST_NODE) {
                    if constexpr (SCENE::OCT) stage_node8_if_built<COUNT>(sc, stk, L, cnt);
                    else if constexpr (SCENE::WIDE) stage_node4<COUNT>(sc, stk, L, cnt);
                    else stage_node<COUNT>(sc, stk, L, cnt);
                }
                // further steps for the lanes that are still at a node, without counting again: the three ballots
                // and the decision chain in front of every step cost a wave about as many cycles as half a step.
                // (Measured and not kept, tools/scratch/r05_node_prefetch_attempt.patch: the second step's node record asked for
                //  the moment the first knows where the lane goes, before its pushes and the ballot in between: +1.6 % per launch.)
#pragma unroll
                for (int rep = 0; rep < SCENE::NODE_REP; rep++) {
                    if (__ballot(L.st == ST_NODE) == 0ull) break;
                    if (COUNT && (threadIdx.x & 63) == 0) cnt.it_node++;
                    if (COUNT && p.lane_hist) lane_hist_add(p, 0, L.st == ST_NODE, L.depth, L.shadow);
                    if constexpr (!STACK::ODD_IDS) { if (COUNT && p.lane_hist) node_id_hist_add(p, L.st == ST_NODE, L.curr); }
                    if (L.st == ST_NODE) {
                        if constexpr (SCENE::OCT) stage_node8_if_built<COUNT>(sc, stk, L, cnt);
                        else if constexpr (SCENE::WIDE) stage_node4<COUNT>(sc, stk, L, cnt);
                        else stage_node<COUNT>(sc, stk, L, cnt);
                    }
                }
                MPT_STAMP_END(acc_node)
            } else {
                if (COUNT && (threadIdx.x & 63) == 0) cnt.it_leaf++;
                if (COUNT && p.lane_hist) lane_hist_add(p, 1, L.st == ST_LEAF, L.depth, L.shadow);
                if (L.st == ST_LEAF) {
                    if constexpr (SCENE::OCT) stage_leaf8_if_built<COUNT>(sc, stk, L, cnt);
                    else stage_leaf<COUNT>(sc, stk, L, cnt);
                }
#pragma unroll
                for (int rep = 0; rep < SCENE::LEAF_REP; rep++) {
                    if (__ballot(L.st == ST_LEAF) == 0ull) break;
                    if (COUNT && (threadIdx.x & 63) == 0) cnt.it_leaf++;
                    if (COUNT && p.lane_hist) lane_hist_add(p, 1, L.st == ST_LEAF, L.depth, L.shadow);
                    if (L.st == ST_LEAF) {
                        if constexpr (SCENE::OCT) stage_leaf8_if_built<COUNT>(sc, stk, L, cnt);
                        else stage_leaf<COUNT>(sc, stk, L, cnt);
                    }
                }
                MPT_STAMP_END(acc_leaf)
            }
        }
        // ---- shading mode
#if MPT_ONE_START
        bool shade_now = wave_count(L.st == ST_DONE && !L.shadow) != 0;
        if constexpr (SCENE::SHADE_MIN > 0) {
            // SHADE costs a wave the same whatever the number of lanes in it (8 400 cycles; a NODE step 575): with fewer than
            // SHADE_MIN lanes waiting for it, and other lanes still traversing, the pass serves the cheap stages only and the
            // lanes wait for company (they are left out of the traversal loop's leave test meanwhile).  LDS-resident kernel:
            // SHADE at 36 lanes instead of 26, 3.18 -> 3.06 ms; the gather kernels, where a step costs three times as much and
            // an idle lane with it, lose 5-14 % and keep SHADE_MIN = 0.
            const int ns = wave_count(L.st == ST_DONE && !L.shadow);
            const int ntrav = wave_count(L.st == ST_NODE || L.st == ST_LEAF);
            shade_now = ns != 0 && (ns >= SCENE::SHADE_MIN || ns * 2 >= 64 - ndead || ntrav == 0);
            deferred = shade_now ? 0 : ns;
        }
        if (shade_now) {
#if MPT_X_PAIRS
            const unsigned long long ms = __ballot(L.st == ST_DONE && !L.shadow);
            if (COUNT && (threadIdx.x & 63) == 0) {
                cnt.pl_batch_lanes += (unsigned)__builtin_popcount((unsigned)ms | (unsigned)(ms >> 32)); cnt.pl_sidle += (unsigned)__builtin_popcountll(ms);
            }
#endif
            if (COUNT && (threadIdx.x & 63) == 0) cnt.it_shade++;
            if (COUNT && p.lane_hist) lane_hist_add(p, 2, L.st == ST_DONE && !L.shadow, L.depth, 0);
            MPT_STAMP_BEGIN
            if (L.st == ST_DONE && !L.shadow) {
                V3 hitpos, sdir;
                float sdis;
                const int nk = shade_core<COUNT>(p, sc, L, cnt, hitpos, sdir, sdis);
                L.to = hitpos;
                if (nk == SH_SHADOW) { L.td = sdir; L.tbest = sdis; L.st = ST_SHADOW; }
                else L.st = ST_BOUNCE;                                       // SH_END: depth is 5, the sample is stored below
            }
            MPT_STAMP_END(acc_shade)
        }
        {
            MPT_STAMP_BEGIN
            // a shadow ray has finished: the candidate direct light is added if nothing was hit (path.py:51,56); the next
            // bounce starts from hitpos (= the shadow ray's origin, still in L.to), path.py:60
            if (L.st == ST_DONE && L.shadow) {
                if (L.hidx < 0) L.result = L.result + L.direct;
                L.st = ST_BOUNCE;
            }
            if (L.st == ST_BOUNCE && !path_continues(L)) lane_store_sample(p, L);   // path.py:25,93: these lanes take a new sample below
            MPT_STAMP_END(acc_sdone)
        }
#else
        if (wave_count(L.st == ST_DONE && L.shadow) != 0) {
            MPT_STAMP_BEGIN
            if (L.st == ST_DONE && L.shadow) stage_shadow_done<COUNT>(p, L, stk, cnt);
            MPT_STAMP_END(acc_sdone)
        }
        if (wave_count(L.st == ST_DONE && !L.shadow) != 0) {
            if (COUNT && (threadIdx.x & 63) == 0) cnt.it_shade++;
            MPT_STAMP_BEGIN
            if (L.st == ST_DONE && !L.shadow) stage_shade<COUNT>(p, sc, L, stk, cnt);
            MPT_STAMP_END(acc_shade)
        }
#endif
        MPT_STAMP_BEGIN
        unsigned long long m_new = __ballot(L.st == ST_NEW);
        if (m_new != 0ull) {
            if (next >= S && more) {                // pool drained: fetch the next work item right away,
#if MPT_X_STAMPS == 3       // diagnostic: cycles (units of 16) inside the pull and inside the preparation of 64 primary rays, of NEW's total
                const unsigned long long tp0 = __builtin_amdgcn_s_memtime();
#endif
                int item = wq.pull();               // while the other lanes are still busy (no per-item tail)
#if MPT_X_STAMPS == 3
                if (COUNT && (threadIdx.x & 63) == 0) cnt.pl_local += (unsigned)((__builtin_amdgcn_s_memtime() - tp0) >> 4);
#endif
                if (item < 0) {
                    more = false;
                    if (tl && (threadIdx.x & 63) == 0) {
                        tl[2] = wall_clock64();
#if MPT_X_TIMELINE2
                        tl[4] = tl_lastpull; tl[5] = (unsigned long long)tl_items;
                        tl[6] = (unsigned long long)(64 - (int)__builtin_popcountll(m_new));     // lanes with a path in flight
#endif
                    }
                } else {
#if MPT_X_TIMELINE2
                    tl_items++; tl_lastpull = wall_clock64();
#endif
                    int tile = item / p.nchunks, chunk = item - tile * p.nchunks;
                    int tx = tile / t8y, ty = tile - tx * t8y;
                    int tps_x = p.stripe_w >> tws, st = tx / tps_x;      // stripe of this tile column
                    ti = p.x0 + st * p.stripe_pitch + ((tx - st * tps_x) << tws); tj = ty << ths; tx_cur = tx;
                    f0 = chunk * p.chunk;
                    S = (min(f0 + p.chunk, p.nframes) - f0) << tps;
                    next = 0; pool_base = -64;
                }
            }
            if (next < S) {
                if (COUNT && (threadIdx.x & 63) == 0) cnt.it_new++;
                const int lane = threadIdx.x & 63;
                if (next >= pool_base + 64) {       // wave-uniform: the pool is used up (or belongs to the last item)
                    pool_base = next;
                    const int smp = pool_base + lane;
                    const int q = smp & ((1 << tps) - 1);
                    const int i = ti + (q >> ths), j = tj + (q & ((1 << ths) - 1));
#if MPT_X_STAMPS == 3
                    __builtin_amdgcn_sched_barrier(0);
                    const unsigned long long tq0 = __builtin_amdgcn_s_memtime();
#endif
                    pool_prepare(p, pool, smp < S && i < p.x1 && j < p.ny, i, j, f0 + (smp >> tps));
#if MPT_X_STAMPS == 3
                    {   // (the rays are used right below: make the wait for the Sobol gathers part of this segment)
                        float sink = pool.rd.x + pool.ro.x;
                        asm volatile("" :: "v"(sink));
                        __builtin_amdgcn_sched_barrier(0);
                        if (COUNT && (threadIdx.x & 63) == 0) cnt.pl_batches += (unsigned)((__builtin_amdgcn_s_memtime() - tq0) >> 4);
                    }
#endif
                }
                // idle lanes take the next consecutive samples (neighbouring pixels of one frame)
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m_new >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_new, 0u));
                const int smp = next + rank;
                const int pool_end = min(S, pool_base + 64);
                const int src = ((smp - pool_base) & 63) << 2;                 // every lane fetches: bpermute reads active lanes only
                V3 ro = v3(lane_from(pool.ro.x, src), lane_from(pool.ro.y, src), lane_from(pool.ro.z, src));
                V3 rd = v3(lane_from(pool.rd.x, src), lane_from(pool.rd.y, src), lane_from(pool.rd.z, src));
                const int rng_i = lane_from(pool.rng_i, src), rng_k = lane_from(pool.rng_k, src);
                if (L.st == ST_NEW && smp < pool_end && rng_k >= 0) {
                    const int q = smp & ((1 << tps) - 1);
                    L.frame = f0 + (smp >> tps);
                    // slot in this launch's sample slab: the columns of the share packed side by side
                    L.pix = ((tx_cur << tws) + (q >> ths)) * p.ny + (tj + (q & ((1 << ths) - 1)));
                    L.rng_i = rng_i; L.rng_k = rng_k; L.prd = rd;
                    L.navoid = 0; L.depth = 0;
                    L.result = v3s(0.0f); L.throughput = v3s(1.0f); L.last_brdf_pdf = 0.0f;
                    if (COUNT) { cnt.samples++; cnt.n_draws += 2; }
#if MPT_ONE_START
                    L.to = ro;
                    L.st = ST_BOUNCE;
                    if (!path_continues(L)) lane_store_sample(p, L);          // (a camera ray of zero length: path.py:25)
#else
                    lane_next_bounce<COUNT>(p, L, stk, ro, cnt);
#endif
                }
                // NEW lanes beyond the pool's end keep waiting: the next pass prepares the next 64 samples
                next = min(next + (int)__builtin_popcountll(m_new), pool_end);
            } else if (!more) {
                if (L.st == ST_NEW) L.st = ST_DEAD;  // nothing left anywhere: those lanes are done
                ndead += (int)__builtin_popcountll(m_new);
            }
        }
        MPT_STAMP_END(acc_new)
#if MPT_ONE_START
        {
            MPT_STAMP_BEGIN
            if (L.st == ST_BOUNCE || L.st == ST_SHADOW) lane_begin_ray<COUNT>(p, L, stk, cnt);
            MPT_STAMP_END(acc_sdone)
        }
#endif
#if MPT_X_TIMELINE2
        if (!more) tl_passes++;
        if (ndead == 64 && tl && (threadIdx.x & 63) == 0) tl[7] = (unsigned long long)tl_passes;
#endif
        if (ndead == 64) break;
    }
#if MPT_X_STAMPS
    if (COUNT) {       // the stage cycles (in units of 256) replace the work counters of this diagnostic build
        const bool l0 = (threadIdx.x & 63) == 0;
        const unsigned long long total = __builtin_amdgcn_s_memtime() - stamp_start;
        cnt.n_box = l0 ? (unsigned)(acc_node >> 8) : 0u; cnt.n_tri = l0 ? (unsigned)(acc_leaf >> 8) : 0u;
        cnt.n_draws = l0 ? (unsigned)(acc_sdone >> 8) : 0u; cnt.n_shade = l0 ? (unsigned)(acc_shade >> 8) : 0u;
        cnt.bounces = l0 ? (unsigned)(acc_new >> 8) : 0u; cnt.n_node = l0 ? (unsigned)(total >> 8) : 0u;
    }
#endif
}

// ---------------------------------------------------------------- tail finalisation
// A launch ends with a drain: the queues are dry, waves finish their last paths and leave one by one (a third of a millisecond
// on the benchmark film), and only then could the combine pass, the resolve pass and the read-back start -- 0.14 ms more per
// step.  With p.fin_counter set, a wave that has nothing left to trace turns to the film instead: it takes the next tile of the
// share (tiles finish in the order their items were issued, so all but the last few are complete), waits until every sample of
// it carries this launch's tag, adds the frames to the film in frame order (film_ops.h: the combine pass's arithmetic), and writes
// the resolved pixels to the caller's image as well when the host knows where get_image() will want them.  The slab entries were
// stored write-through (store_sample) and are read here with sc1 loads (L1 bypassed, re-read every pass: R2 of the guide).
// Nothing waits for a finishing wave, and what IT waits for is in the hands of waves that are running (every item has been pulled
// before the first wave gets here), so the loop ends; a bounded spin raises the watchdog instead of hanging if it ever does not.
DEV mpt_u4 slab_load_sc1(const MptVec4 *frame_base, unsigned frame_bytes, unsigned byte_off) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)frame_base, (short)0, (int)frame_bytes, 0x00020000);
#ifndef MPT_FIN_AUX
#define MPT_FIN_AUX 16       // cache bits of the slab loads: 16 = sc1 (A/B: 18 = sc1 nt)
#endif
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, MPT_FIN_AUX);      // aux 16 = sc1
}

#ifndef MPT_FIN_SLEEP
#define MPT_FIN_SLEEP 32     // units of 64 cycles between two looks at a tile that is not complete yet
#endif
#ifndef MPT_FIN_GROUP_LDS
#define MPT_FIN_GROUP_LDS 8       // slab loads in flight per lane: the LDS-resident kernel has 128 VGPRs to lend ...
#define MPT_FIN_GROUP_GATHER 6    // ... the gather kernels 96 (with eight the function needs 102 and they would lose their fifth wave per SIMD)
#endif
#ifndef MPT_FIN_INLINE
// 0: out of line.  Inlined into the render kernels the finalisation moved their register allocation and the traversal loop ran
// 3 % slower (MI355X, same box: 3.21 against 3.13 ms per launch, profiles/r04_ab_experiments.json); as a function of its own it
// leaves them alone, at the price of its registers counting for every kernel that calls it (MPT_FIN_GROUP_*).
#define MPT_FIN_INLINE 0
#endif
// what finalise_tiles reads of the launch parameters.  Out of line, its arguments arrive in vector registers: the ones a buffer
// descriptor is made of are made scalar again (readfirstlane; they are wave-uniform)
struct FinArgs {
    MptVec4 *partial, *film0, *image_out;
    unsigned int *fin_counter, *watchdog;
    int tile_w_shift, tile_h_shift, ny, nitems, nchunks, nframes, partial_stride, stripe_w, stripe_pitch, x0, x1;
    unsigned slab_tag;
};
DEV int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T> DEV T *uniform_p(T *ptr) {
    const unsigned long long v = (unsigned long long)ptr;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}
// (individual parameters, not a struct by value: that one would travel through scratch memory)
template <int GROUP>
#if MPT_FIN_INLINE
DEV int finalise_tiles_impl(
#else
__device__ __attribute__((noinline)) int finalise_tiles_impl(
#endif
        MptVec4 *a_partial, MptVec4 *a_film0, MptVec4 *a_image_out, unsigned int *a_fin_counter, unsigned int *a_watchdog,
        int a_tws, int a_ths, int a_ny, int a_nitems, int a_nchunks, int a_nframes, int a_partial_stride, int a_stripe_w,
        int a_stripe_pitch, int a_x0, int a_x1, unsigned a_slab_tag) {
    FinArgs p;
    p.partial = a_partial; p.film0 = a_film0; p.image_out = a_image_out; p.fin_counter = a_fin_counter; p.watchdog = a_watchdog;
    p.tile_w_shift = a_tws; p.tile_h_shift = a_ths; p.ny = a_ny; p.nitems = a_nitems; p.nchunks = a_nchunks; p.nframes = a_nframes;
    p.partial_stride = a_partial_stride; p.stripe_w = a_stripe_w; p.stripe_pitch = a_stripe_pitch; p.x0 = a_x0; p.x1 = a_x1;
    p.slab_tag = a_slab_tag;
#if !MPT_FIN_INLINE
    p.partial = uniform_p(p.partial); p.partial_stride = uniform_i(p.partial_stride); p.nframes = uniform_i(p.nframes);
    p.tile_w_shift = uniform_i(p.tile_w_shift); p.tile_h_shift = uniform_i(p.tile_h_shift);
#endif
    const int lane = threadIdx.x & 63;
    const int tws = p.tile_w_shift, ths = p.tile_h_shift, tps = tws + ths;
    const int t8y = (p.ny + (1 << ths) - 1) >> ths;
    const int ntile = p.nitems / p.nchunks;                 // items are tile-major: nchunks per tile
    const int B = p.nframes;
    const unsigned frame_bytes = (unsigned)p.partial_stride * 16u;      // (a frame of the slab is far below 4 GiB: the film's cap is 2^26 pixels)
    const unsigned tag = p.slab_tag;
    const mpt_u4 absent = slab_pack(0.0f, 0.0f, 0.0f, tag);     // a frame past the batch's end, a pixel past the film's edge: ready, adds nothing
    const unsigned long long t_begin = wall_clock64();
    int done = 0;
    for (;; done++) {
        int t = 0;
        if (lane == 0) t = (int)atomicAdd(p.fin_counter, 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= ntile) break;
        const int tx = t / t8y, ty = t - tx * t8y;
        const int tps_x = p.stripe_w >> tws, st = tx / tps_x;            // stripe of this tile column (as in trace_stream)
        const int ti = p.x0 + st * p.stripe_pitch + ((tx - st * tps_x) << tws), tj = ty << ths;
        for (int q0 = 0; q0 < (1 << tps); q0 += 64) {                    // (wave-uniform trip count)
            const int q = q0 + lane;
            const int i = ti + (q >> ths), j = tj + (q & ((1 << ths) - 1));
            const bool inside = q < (1 << tps) && i < p.x1 && j < p.ny;  // (pixels of the tile past the film's edge get no samples)
            const unsigned off = (unsigned)(((tx << tws) + (q >> ths)) * p.ny + (tj + (q & ((1 << ths) - 1)))) * 16u;
            const size_t pix = (size_t)i * p.ny + j;
            MptVec4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
            if (inside) acc = p.film0[pix];
            for (int f0 = 0; f0 < B; f0 += GROUP) {
                mpt_u4 v[GROUP];
                for (;;) {
                    bool ready = true;
#pragma unroll
                    for (int k = 0; k < GROUP; k++) {
                        v[k] = absent;
                        if (f0 + k < B && inside) v[k] = slab_load_sc1(p.partial + (size_t)(f0 + k) * (size_t)p.partial_stride, frame_bytes, off);
                        ready = ready && slab_ready(v[k], tag);       // each 8-byte half on its own tag (film_ops.h)
                    }
                    if (__ballot(!ready) == 0ull) break;
                    if (wall_clock64() - t_begin > 400000000ull) {       // 4 s at 100 MHz: some sample never came
                        if (lane == 0) __hip_atomic_store(p.watchdog, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        return done;
                    }
                    __builtin_amdgcn_s_sleep(MPT_FIN_SLEEP);
                }
#pragma unroll
                for (int k = 0; k < GROUP; k++)
                    if (f0 + k < B) film_add_sample(acc, slab_r(v[k]), slab_g(v[k]), slab_b(v[k]));
            }
            if (inside) {
                p.film0[pix] = acc;
                if (p.image_out) p.image_out[pix] = film_resolve(acc);
            }
        }
    }
    return done;
}

// GROUP = slab loads in flight per lane: what the calling kernel's register budget allows (see MPT_FIN_INLINE)
template <int GROUP>
DEV int finalise_tiles(const MptRenderParams &p) {
#if MPT_FIN_INLINE >= 0          // (-1: A/B build without the call: the render kernels as they were before the tail finalisation)
    // In a workgroup of three or four waves per SIMD only the younger two finalise.  The hardware issues the oldest wave of a
    // SIMD first, so the old waves finish tracing first -- and, finalising, stayed in front of the waves still tracing behind
    // them: with all four at it the launch took 3.15 ms, with the younger two 3.12 (the combine pass after the launch: 3.10 + 0.1;
    // MI355X, same box, profiles/r04_ab_experiments.json).  Every tile is still taken by somebody: the loop runs until none is left.
#ifndef MPT_FIN_YOUNG
#define MPT_FIN_YOUNG 2
#endif
    if ((blockDim.x >> 8) >= 3 && (int)((threadIdx.x >> 8) & 3) < MPT_FIN_YOUNG) return 0;
    if (p.fin_counter)
        return finalise_tiles_impl<GROUP>(p.partial, p.film0, p.image_out, p.fin_counter, p.watchdog, p.tile_w_shift, p.tile_h_shift, p.ny,
                                          p.nitems, p.nchunks, p.nframes, p.partial_stride, p.stripe_w, p.stripe_pitch, p.x0, p.x1, p.slab_tag);
#endif
    return 0;
}
#endif

// ---------------------------------------------------------------- gather kernel: 16x16 tile x chunk per workgroup
DEV bool tile_pixel(const MptRenderParams &p, int tile, int *pi, int *pj) {
    int tx = tile / p.tiles_y, ty = tile - tx * p.tiles_y;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int tps_x = p.stripe_w / MPT_TILE, st = tx / tps_x;                       // stripe of this tile column
    int i = p.x0 + st * p.stripe_pitch + (tx - st * tps_x) * MPT_TILE + (wave >> 1) * 8 + (lane >> 3);
    int j = ty * MPT_TILE + (wave & 1) * 8 + (lane & 7);
    *pi = i; *pj = j;
    return i < p.x1 && j < p.ny;
}

#if MPT_STRICT
#define MPT_RENDER_BOUNDS __launch_bounds__(MPT_BLOCK)
#else
// gathers from L2 / Infinity Cache are latency-bound: ask for 4 waves per SIMD (the 32-level kernels then sit at the 128-VGPR
// cap; tools/kernel_resources.py prints registers / scratch of every kernel as built)
#define MPT_RENDER_BOUNDS __launch_bounds__(MPT_BLOCK, 4)
#endif
template <int STACK, bool COUNT>
__global__ MPT_RENDER_BOUNDS void MPT_SUFFIX(render_kernel)(const MptRenderParams p) {
    __shared__ int s_stack[STACK * MPT_BLOCK];
    BlockTracer tr = make_block_tracer(p, s_stack + threadIdx.x);
    Cnt cnt = {};
#if MPT_STRICT
    // one 16x16 tile per workgroup, every frame of the batch; blockIdx remapped for XCD locality
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    int i, j;
    if (tile_pixel(p, tile, &i, &j)) trace_pixel<COUNT>(p, tr, i, j, 0, p.nframes, cnt);
#else
    // persistent workgroups pulling (8x8 tile, chunk) items; see WorkQueue
    WorkQueue wq; wq.ctr = p.work_counter; wq.nitems = p.nitems; wq.q0 = blockIdx.x & 7; wq.qoff = 0;
    trace_stream<COUNT>(p, tr.sc, tr.st, wq, cnt);
    finalise_tiles<MPT_FIN_GROUP_GATHER>(p);
#endif
    flush_counters<COUNT>(p, cnt);
}

#if !MPT_STRICT
// ---------------------------------------------------------------- gather kernel over 4-wide nodes
#ifndef MPT_WIDE_WAVES
#define MPT_WIDE_WAVES 5      // waves per SIMD the 4-wide gather kernel is compiled for (its register budget: 512 / this = 96 VGPRs).
                              // At 96 the allocator parks 27 dwords per lane in scratch, all of them in the shading pass (SHADE's own
                              // temporaries and the prepared primary rays), none in the traversal loop: MI355X C4 1520 -> 1580,
                              // C5 778 -> 825 Msamples/s.  Six waves (80 VGPRs) spill into the steps and lose a third.
#endif
template <bool COUNT, bool QUANT>
__global__ __launch_bounds__(MPT_BLOCK, MPT_WIDE_WAVES) void render_kernel_wide(const MptRenderParams p) {
    __shared__ int s_stack[SpillStack::CAP * MPT_BLOCK];
    SpillStack stk;
    stk.base = s_stack + threadIdx.x;
    stk.spill = p.stack_spill;
    stk.lane_off = (blockIdx.x * MPT_BLOCK + threadIdx.x) * (unsigned)SpillStack::SPILL;   // (grid x 256 x 88 entries: far below 2^32)
    stk.sp = 0;
    Cnt cnt = {};
    WorkQueue wq; wq.ctr = p.work_counter; wq.nitems = p.nitems; wq.q0 = blockIdx.x & 7; wq.qoff = 0;
    if constexpr (QUANT) {
        QuantScene sc; sc.qnode = p.qnode; sc.tgeo = p.tfast;
#if MPT_X_TOPCACHE
        __shared__ __attribute__((aligned(16))) float s_top[MPT_X_TOPCACHE * 16];
        for (int k = threadIdx.x; k < min(p.nwide, MPT_X_TOPCACHE) * 4; k += MPT_BLOCK) ((MptVec4 *)s_top)[k] = p.qnode[k];
        __syncthreads();
        sc.top = (__attribute__((address_space(3))) const QuantScene::top_f4 *)(void *)s_top;
#endif
        trace_stream<COUNT>(p, sc, stk, wq, cnt);
    } else {
        WideScene sc; sc.wnode = p.wnode; sc.tgeo = p.tfast;
        trace_stream<COUNT>(p, sc, stk, wq, cnt);
    }
    finalise_tiles<MPT_FIN_GROUP_GATHER>(p);
    flush_counters<COUNT>(p, cnt);
}

// ---------------------------------------------------------------- LDS-resident persistent kernel
// dynamic LDS: [ (n-1) node records of MPT_LDS_NODE_STRIDE bytes, padded to 16 | n*3 triangle float4 (tfast) | (default_mtl+1)*6 material float4 |
//                n material-record bytes, padded to 16 | lds_stack x 1024 int16 ]     (mpt_lds_scene_bytes)
template <bool COUNT>
__global__ __launch_bounds__(MPT_LDS_BLOCK) void render_kernel_lds(const MptRenderParams p) {
    extern __shared__ __attribute__((aligned(16))) MptVec4 smem[];
    // node records MPT_LDS_NODE_STRIDE bytes apart (bank spreading), the region rounded up to whole float4
    const int nnode4 = ((p.n - 1) * MPT_LDS_NODE_STRIDE + 15) >> 4, ntri4 = p.n * 3, nmat4 = (p.default_mtl + 1) * MPT_LDS_MAT_VEC4;
    const int nmtl4 = (p.n + 15) >> 4;
    unsigned long long *tl = p.timeline ? p.timeline + MPT_TIMELINE_WORDS * (size_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) : nullptr;
    if (tl && (threadIdx.x & 63) == 0) tl[0] = wall_clock64();
    {   // one copy of the scene per CU: coalesced 16-B loads, ds_write_b128
        for (int k = threadIdx.x; k < (p.n - 1) * 4; k += blockDim.x) {          // 8-byte stores: the records are 8-byte aligned
            MptVec4 v = p.fnode[k];
            if (LdsSceneP::PRESCALED_IDS && (k & 3) == 3) {                        // {id0, id1, -, -}: internal ids become byte offset / 8
                const int i0 = __float_as_int(v.x), i1 = __float_as_int(v.y);
                v.x = __int_as_float(i0 >= 0 ? i0 * (MPT_LDS_NODE_STRIDE / 8) : i0);
                v.y = __int_as_float(i1 >= 0 ? i1 * (MPT_LDS_NODE_STRIDE / 8) : i1);
            }
            float *d = (float *)((char *)smem + (k >> 2) * MPT_LDS_NODE_STRIDE + (k & 3) * 16);
            *(float2 *)d = make_float2(v.x, v.y); *(float2 *)(d + 2) = make_float2(v.z, v.w);
        }
        for (int k = threadIdx.x; k < ntri4; k += blockDim.x) smem[nnode4 + k] = p.tfast[k];
        for (int k = threadIdx.x; k < nmat4; k += blockDim.x) {
            const int rec = k / MPT_LDS_MAT_VEC4, w = k - rec * MPT_LDS_MAT_VEC4;
            smem[nnode4 + ntri4 + k] = ((const MptVec4 *)(p.mats + rec))[w < 4 ? w : w + 4];
        }
        unsigned char *mtl = (unsigned char *)(smem + nnode4 + ntri4 + nmat4);
        for (int k = threadIdx.x; k < p.n; k += blockDim.x) {
            const int id = __float_as_int(p.tshade[(size_t)k * 4 + 3].w);
            mtl[k] = (unsigned char)(id == -1 ? p.default_mtl : id);
        }
    }
    __syncthreads();
    if (tl && (threadIdx.x & 63) == 0) tl[1] = wall_clock64();

    LdsSceneP sc;
    sc.fnode = (LdsVec4Ptr)(void *)smem;
    sc.tgeo = (LdsVec4Ptr)(void *)(smem + nnode4);
    sc.mats = (LdsVec4Ptr)(void *)(smem + nnode4 + ntri4);
    sc.mtl = (LdsU8Ptr)(void *)(smem + nnode4 + ntri4 + nmat4);
    sc.nstride = MPT_LDS_NODE_STRIDE;
    sc.mat_last = p.default_mtl; sc.mat_default = p.default_mtl;
    Stack16 stk;
    stk.base = (LdsShortPtr)(void *)(smem + nnode4 + ntri4 + nmat4 + nmtl4) + threadIdx.x;
    stk.sp = 0;
    Cnt cnt = {};
    WorkQueue wq; wq.ctr = p.work_counter; wq.nitems = p.nitems; wq.q0 = blockIdx.x & 7; wq.qoff = 0;
    trace_stream<COUNT>(p, sc, stk, wq, cnt, tl);
    if (tl && (threadIdx.x & 63) == 0) tl[3] = wall_clock64();
    const int fin_tiles = finalise_tiles<MPT_FIN_GROUP_LDS>(p);
#if !MPT_X_TIMELINE2
    if (tl && (threadIdx.x & 63) == 0) { tl[4] = wall_clock64(); tl[5] = (unsigned long long)fin_tiles; }   // left the finalisation; tiles it did
#endif
    flush_counters<COUNT>(p, cnt);
}

// ---------------------------------------------------------------- LDS-resident persistent kernel over the 4-wide nodes
// dynamic LDS: [ nwide node records of MPT_LDS4_NODE_STRIDE bytes | (n+1)*3 triangle float4 (tfast; record n: the unused slots' NaNs) |
//                (lds_nmats+1)*6 material float4 (the records the model uses, then the default one) | n material-record bytes,
//                padded to 16 | lds_stack x 1024 int16 ]
template <bool COUNT>
__global__ __launch_bounds__(MPT_LDS_BLOCK) void render_kernel_lds4(const MptRenderParams p) {
    extern __shared__ __attribute__((aligned(16))) MptVec4 smem[];
    const int nnode4 = p.nwide * (MPT_LDS4_NODE_STRIDE / 16), ntri4 = (p.n + 1) * 3, nmat4 = (p.lds_nmats + 1) * MPT_LDS_MAT_VEC4;
    const int nmtl4 = (p.n + 15) >> 4;
    unsigned long long *tl = p.timeline ? p.timeline + MPT_TIMELINE_WORDS * (size_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) : nullptr;
    if (tl && (threadIdx.x & 63) == 0) tl[0] = wall_clock64();
    {
        for (int k = threadIdx.x; k < p.nwide * 7; k += blockDim.x) {
            const int rec = k / 7, w = k - rec * 7;
            MptVec4 v = p.wnode[rec * 8 + w];
            if (w == 6) {
                // the four ids: internal ones become the record's byte offset (/ 8 without ODD_IDS), leaves (slot << 4) | 1 (~slot)
                const int i0 = __float_as_int(v.x), i1 = __float_as_int(v.y), i2 = __float_as_int(v.z), i3 = __float_as_int(v.w);
                const int scale = LdsWideScene::ODD_IDS ? MPT_LDS4_NODE_STRIDE : MPT_LDS4_NODE_STRIDE / 8;
#define MPT_LDS_ID(i) __int_as_float((i) >= 0 ? (i) * scale : (LdsWideScene::ODD_IDS ? ((~(i)) << 4) | 1 : (i)))
                v.x = MPT_LDS_ID(i0); v.y = MPT_LDS_ID(i1); v.z = MPT_LDS_ID(i2); v.w = MPT_LDS_ID(i3);
#undef MPT_LDS_ID
            }
            smem[k] = v;
        }
        for (int k = threadIdx.x; k < ntri4; k += blockDim.x) smem[nnode4 + k] = p.tfast[k];
        for (int k = threadIdx.x; k < nmat4; k += blockDim.x) {
            const int rec = k / MPT_LDS_MAT_VEC4, w = k - rec * MPT_LDS_MAT_VEC4;
            const int grec = rec == p.lds_nmats ? p.default_mtl : rec;          // the default material's record is kept last
            smem[nnode4 + ntri4 + k] = ((const MptVec4 *)(p.mats + grec))[w < 4 ? w : w + 4];
        }
        unsigned char *mtl = (unsigned char *)(smem + nnode4 + ntri4 + nmat4);
        for (int k = threadIdx.x; k < p.n; k += blockDim.x) {
            const int id = __float_as_int(p.tshade[(size_t)k * 4 + 3].w);
            mtl[k] = (unsigned char)(id == -1 ? p.lds_nmats : id);
        }
    }
    __syncthreads();
    if (tl && (threadIdx.x & 63) == 0) tl[1] = wall_clock64();

    LdsWideScene sc;
    sc.wnode = (LdsVec4Ptr)(void *)smem;
    if (LdsWideScene::ODD_IDS && (unsigned)(unsigned long long)(LdsBytePtr)sc.wnode != 0u) {
        // node ids are LDS addresses counted from 0: the dynamic LDS must be all the LDS this kernel has (it is; a static __shared__
        // object added to it one day would move smem)
        if (threadIdx.x == 0) __hip_atomic_store(p.watchdog, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    sc.tgeo = (LdsVec4Ptr)(void *)(smem + nnode4);
    sc.mats = (LdsVec4Ptr)(void *)(smem + nnode4 + ntri4);
    sc.mtl = (LdsU8Ptr)(void *)(smem + nnode4 + ntri4 + nmat4);
    sc.mat_last = p.lds_nmats; sc.mat_default = p.default_mtl;
    Stack16W stk;
    stk.base = (LdsShortPtr)(void *)(smem + nnode4 + ntri4 + nmat4 + nmtl4) + threadIdx.x;
    stk.sp = 0;
    stk.ts = p.t_scale;
    Cnt cnt = {};
    WorkQueue wq; wq.ctr = p.work_counter; wq.nitems = p.nitems; wq.q0 = blockIdx.x & 7; wq.qoff = 0;
    trace_stream<COUNT>(p, sc, stk, wq, cnt, tl);
    if (tl && (threadIdx.x & 63) == 0) tl[3] = wall_clock64();
    const int fin_tiles = finalise_tiles<MPT_FIN_GROUP_LDS>(p);
#if !MPT_X_TIMELINE2
    if (tl && (threadIdx.x & 63) == 0) { tl[4] = wall_clock64(); tl[5] = (unsigned long long)fin_tiles; }
#endif
    flush_counters<COUNT>(p, cnt);
}
#endif

// PreviewEngine._render, engine/preview.py:23-41
template <int STACK>
__global__ __launch_bounds__(MPT_BLOCK) void MPT_SUFFIX(preview_kernel)(const MptRenderParams p) {
    __shared__ int s_stack[STACK * MPT_BLOCK];
    BlockTracer tr = make_block_tracer(p, s_stack + threadIdx.x);
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    int i, j;
    if (!tile_pixel(p, tile, &i, &j)) return;
    const int pix = i * p.ny + j;
    const int h = wanghash2(i, j);
    Cnt cnt = {};
    MptVec4 a1 = p.film1[pix], a2 = p.film2[pix];
    for (int f = 0; f < p.nframes; f++) {
        Rng rng; rng.dim = p.sobol_dim; rng.P = p.P + (size_t)f * p.sobol_dim; rng.i = h;
        V3 albedo = v3s(0.0f), normal = v3s(0.0f);
        float dx = rng_random(rng), dy = rng_random(rng);
        float x = m_div((float)i + dx, (float)p.nx) * 2.0f - 1.0f;
        float y = m_div((float)j + dy, (float)p.ny) * 2.0f - 1.0f;
        V3 ro, rd;
        camera_generate(p, x, y, &ro, &rd);
        Hit hit = tr.template closest<false>(ro, rd, -1, cnt);
        if (hit.hit == 1) {
            V3 hitpos; Disney material;
            get_geometries(p, hit, ro, rd, &hitpos, &normal, material);
            albedo = material.basecolor;
        }
        a1.x += albedo.x; a1.y += albedo.y; a1.z += albedo.z; a1.w += 1.0f;
        a2.x += normal.x; a2.y += normal.y; a2.z += normal.z; a2.w += 1.0f;
    }
    p.film1[pix] = a1;
    p.film2[pix] = a2;
}

// ---------------------------------------------------------------- host-side launchers
template <class K>
static int blocks_per_cu(K kernel) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, MPT_BLOCK, 0) != hipSuccess || nb < 1) nb = 2;
    return nb;
}

// strict build: grid = number of 16x16 tiles.  fast build: persistent workgroups, `grid` = number of CUs
// (scaled here by the blocks each CU can hold); the work items come from p->work_counter.
MPT_KERNEL_API hipError_t MPT_SUFFIX(mpt_launch_render)(const MptRenderParams *p, int grid, int stack, int count,
                                                     hipStream_t stream) {
#if !MPT_STRICT
    // occupancy answers are per device (the C ABI allows one context per GPU in a process)
    static std::atomic<int> occ_cache[MPT_MAX_DEVICES][4];
    int v = (stack <= 32 ? 0 : 2) + (count ? 1 : 0);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MPT_MAX_DEVICES) return hipErrorInvalidDevice;
    int occ = occ_cache[dev][v].load(std::memory_order_relaxed);
    if (!occ) {
        occ = v == 0 ? blocks_per_cu(MPT_SUFFIX(render_kernel)<32, false>)
            : v == 1 ? blocks_per_cu(MPT_SUFFIX(render_kernel)<32, true>)
            : v == 2 ? blocks_per_cu(MPT_SUFFIX(render_kernel)<64, false>)
                     : blocks_per_cu(MPT_SUFFIX(render_kernel)<64, true>);
        occ_cache[dev][v].store(occ, std::memory_order_relaxed);
    }
    grid *= occ;
#endif
    if (stack <= 32) {
        if (count) hipLaunchKernelGGL((MPT_SUFFIX(render_kernel)<32, true>), dim3(grid), dim3(MPT_BLOCK), 0, stream, *p);
        else hipLaunchKernelGGL((MPT_SUFFIX(render_kernel)<32, false>), dim3(grid), dim3(MPT_BLOCK), 0, stream, *p);
    } else {
        if (count) hipLaunchKernelGGL((MPT_SUFFIX(render_kernel)<64, true>), dim3(grid), dim3(MPT_BLOCK), 0, stream, *p);
        else hipLaunchKernelGGL((MPT_SUFFIX(render_kernel)<64, false>), dim3(grid), dim3(MPT_BLOCK), 0, stream, *p);
    }
    return hipGetLastError();
}

#if !MPT_STRICT
// lds_bytes = scene records + 2 KiB per stack level; grid = one persistent workgroup per CU
template <bool COUNT>
static hipError_t launch_lds(const MptRenderParams *p, int grid, int block, size_t lds_bytes, hipStream_t stream) {
    // function attributes are per device: remember which devices have been told about the 160 KiB
    static std::atomic<bool> configured[MPT_MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MPT_MAX_DEVICES) return hipErrorInvalidDevice;
    if (!configured[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)render_kernel_lds<COUNT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        configured[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((render_kernel_lds<COUNT>), dim3(grid), dim3(block), lds_bytes, stream, *p);
    return hipGetLastError();
}

template <bool COUNT>
static hipError_t launch_lds4(const MptRenderParams *p, int grid, int block, size_t lds_bytes, hipStream_t stream) {
    static std::atomic<bool> configured[MPT_MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MPT_MAX_DEVICES) return hipErrorInvalidDevice;
    if (!configured[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)render_kernel_lds4<COUNT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        configured[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((render_kernel_lds4<COUNT>), dim3(grid), dim3(block), lds_bytes, stream, *p);
    return hipGetLastError();
}

// the same over the 4-wide nodes (p->wnode, p->nwide)
MPT_KERNEL_API hipError_t mpt_launch_render_lds4(const MptRenderParams *p, int grid, int block, size_t lds_bytes, int count,
                                             hipStream_t stream) {
    return count ? launch_lds4<true>(p, grid, block, lds_bytes, stream) : launch_lds4<false>(p, grid, block, lds_bytes, stream);
}

// lds_bytes = scene records + 2 KiB per stack level; grid = one persistent workgroup per CU
MPT_KERNEL_API hipError_t mpt_launch_render_lds(const MptRenderParams *p, int grid, int block, size_t lds_bytes, int count,
                                            hipStream_t stream) {
    return count ? launch_lds<true>(p, grid, block, lds_bytes, stream) : launch_lds<false>(p, grid, block, lds_bytes, stream);
}
#endif

#if !MPT_STRICT && MPT_WITH_OCT
#define MPT_OCT_KERNELS 1
#include "render_oct.h"      // render_kernel_oct + its launchers (second pass over the header)
#endif

#if !MPT_STRICT && MPT_WITH_POOL
// The pooled LDS kernel (waves specialised into tracers and shaders, paths traded through LDS pools) measured 15-50 % slower
// than render_kernel_lds: it is an A/B build (make pool -> libmiptina_pool.so), not part of the product library
#include "render_pool.h"
#endif

#if !MPT_STRICT
// Disney.__init__'s derived terms (disney.py:36-50) of every material record, once per upload: the production
// material fetch then reads them instead of re-deriving them at every hit (same device function: same bits)
__global__ void derive_materials_kernel(MptMaterial *mats, int count) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    MptMaterial *mt = mats + i;
    Disney m;
    m.basecolor = v3(mt->p[0], mt->p[1], mt->p[2]);
    m.metallic = mt->p[3]; m.roughness = mt->p[4]; m.specular = mt->p[5]; m.specularTint = mt->p[6];
    m.subsurface = mt->p[7]; m.sheen = mt->p[8]; m.sheenTint = mt->p[9]; m.clearcoat = mt->p[10];
    m.clearcoatGloss = mt->p[11]; m.transmission = mt->p[12]; m.ior = mt->p[13];
    disney_init(m);
    mt->d[0] = m.speccolor.x; mt->d[1] = m.speccolor.y; mt->d[2] = m.speccolor.z;
    mt->d[3] = m.sheencolor.x; mt->d[4] = m.sheencolor.y; mt->d[5] = m.sheencolor.z;
    mt->d[6] = m.alpha; mt->d[7] = m.clearcoatAlpha;
    mt->p[14] = __int_as_float(mt->any_tex);
}

MPT_KERNEL_API hipError_t mpt_launch_derive_materials(MptMaterial *mats, int count, hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(derive_materials_kernel, dim3((count + 63) / 64), dim3(64), 0, stream, mats, count);
    return hipGetLastError();
}

// persistent workgroups over 4-wide nodes; `grid` = number of CUs (scaled here by the blocks each CU can hold);
// *blocks = workgroups launched (the spill strip must hold blocks x 256 lanes x SpillStack::SPILL entries)
MPT_KERNEL_API hipError_t mpt_wide_blocks(int grid, int count, int quant, int *blocks) {
    static std::atomic<int> occ_cache[MPT_MAX_DEVICES][4];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MPT_MAX_DEVICES) return hipErrorInvalidDevice;
    const int v = (count ? 1 : 0) + (quant ? 2 : 0);
    int occ = occ_cache[dev][v].load(std::memory_order_relaxed);
    if (!occ) {
        occ = v == 0 ? blocks_per_cu(render_kernel_wide<false, false>) : v == 1 ? blocks_per_cu(render_kernel_wide<true, false>)
            : v == 2 ? blocks_per_cu(render_kernel_wide<false, true>) : blocks_per_cu(render_kernel_wide<true, true>);
        occ_cache[dev][v].store(occ, std::memory_order_relaxed);
    }
    *blocks = grid * occ;
    return hipSuccess;
}

MPT_KERNEL_API hipError_t mpt_launch_render_wide(const MptRenderParams *p, int blocks, int count, int quant, hipStream_t stream) {
    if (quant) {
        if (count) hipLaunchKernelGGL((render_kernel_wide<true, true>), dim3(blocks), dim3(MPT_BLOCK), 0, stream, *p);
        else hipLaunchKernelGGL((render_kernel_wide<false, true>), dim3(blocks), dim3(MPT_BLOCK), 0, stream, *p);
    } else {
        if (count) hipLaunchKernelGGL((render_kernel_wide<true, false>), dim3(blocks), dim3(MPT_BLOCK), 0, stream, *p);
        else hipLaunchKernelGGL((render_kernel_wide<false, false>), dim3(blocks), dim3(MPT_BLOCK), 0, stream, *p);
    }
    return hipGetLastError();
}
#endif

MPT_KERNEL_API hipError_t MPT_SUFFIX(mpt_launch_preview)(const MptRenderParams *p, int grid, int stack,
                                                      hipStream_t stream) {
    if (stack <= 32) hipLaunchKernelGGL((MPT_SUFFIX(preview_kernel)<32>), dim3(grid), dim3(MPT_BLOCK), 0, stream, *p);
    else hipLaunchKernelGGL((MPT_SUFFIX(preview_kernel)<64>), dim3(grid), dim3(MPT_BLOCK), 0, stream, *p);
    return hipGetLastError();
}
