// render_pool.h -- the LDS-resident kernel with the waves of a workgroup SPECIALISED and two path pools in LDS between
// them (included by render_kernel.hip, production build only).
//
// render_kernel_lds runs every stage in every wave: a wave's 64 lanes are its scheduling domain, and measured on
// MI355X that domain is too small -- a SHADE stage (the whole Disney bounce, ~900 VALU instructions) found 20 of 64
// lanes ready, a NODE step 33, and the end of a launch is one path latency during which every wave of the chip runs
// nearly empty.  Here the scheduling domain is the WORKGROUP (16 waves, one per CU):
//   * tracer waves only traverse.  A lane whose closest-hit ray has finished writes a 80-byte shade request into the
//     workgroup's shade pool and is free at once; a free lane takes the next ray out of the ray pool.  Shadow rays that
//     finish are settled in place (one add and the next ray: cheap) -- only the expensive stage travels.
//   * shader waves (p.pool_shaders of the 16, one per SIMD first) only shade: 64 requests at a time with every lane on
//     (fewer when nothing else is left to do), each bounce producing a 96-byte ray entry -- the shadow ray towards the
//     sampled light, or the next bounce -- for whichever tracer lane is free first.  They also make the primary rays:
//     work items (8x8 tile x frame = 64 samples) are pulled from the per-XCD queues and go into the ray pool as 64
//     "start a bounce here" entries.
// A path therefore migrates between lanes and waves at every bounce.  Nothing observable depends on where it ran: its
// Sobol proxy, throughput and partial sums travel with it, and its radiance goes to the sample slab slot of its
// (frame, pixel) -- the film is bit for bit the film of render_kernel_lds (tests/test_parity_gpu.py).
//
// Pools: bounded MPMC rings of fixed-size records in LDS, SoA of float4 ([field][slot]: conflict-free ds_*_b128).
//   producer: space -= k (all or nothing; restored on failure) -> first = tail += k -> payload, then flag[slot] = pos + 1
//             -> avail += k
//   consumer: avail -= n (what is there, at most what it wants) -> first = head += n -> wait for flag[slot] == pos + 1
//             (its producer is between "tail +=" and the flag store: a few hundred cycles, and never waits for anybody)
//             -> payload -> space += n
// Every operation is a TRY: a tracer that finds the shade pool full shades its lanes itself (stage_shade, as the
// unspecialised kernel does), a shader reserves the 64 ray slots it may need BEFORE it takes requests, and nobody ever
// waits for space -- so no cycle of waits exists, and paths are finite (depth <= 5): the launch drains.
// End: `feeders` = shader waves that may still find work items, `live` = paths between their primary ray and their
// sample store; everybody leaves when both are zero (live can only grow through a feeder).
#pragma once

#ifndef MPT_POOL_S_CAP
#define MPT_POOL_S_CAP 128      // shade requests (5 float4 each)
#endif
#ifndef MPT_POOL_R_CAP
#define MPT_POOL_R_CAP 128      // rays (6 float4 each)
#endif
#define MPT_POOL_S_VEC4 5
#define MPT_POOL_R_VEC4 6
#ifndef MPT_POOL_LEAVE
#define MPT_POOL_LEAVE 12       // a tracer leaves its traversal loop for the pools when this many lanes are waiting
#endif
#ifndef MPT_POOL_RLOW
#define MPT_POOL_RLOW 48        // shaders make primary rays while fewer rays than this are waiting for a tracer
#endif
#ifndef MPT_POOL_BATCH
#define MPT_POOL_BATCH 48       // a shader takes bounces when this many wait ...
#endif
#ifndef MPT_POOL_PATIENCE
#define MPT_POOL_PATIENCE 6     // ... or fewer after this many polls with nothing else to do (a partial SHADE costs the
#endif                          // SIMD the issue slots of a full one)
#ifndef MPT_POOL_BURST
#define MPT_POOL_BURST 3        // traversal decisions a tracer makes before asking again after it found no ray to take
#endif

// Diagnostic build (-DMPT_X_POOL_STAMPS=1, counting kernels only): shader-clock cycles (in units of 256) per activity, in the
// pl_* counters instead of their usual meaning -- tracer waves: pl_batches = traversal mode, pl_batch_lanes = trips to the
// pools, pl_prim = idle; shader waves: pl_local = SHADE batches, pl_taken = primary rays, pl_sidle = idle; pl_trips = lifetime
#if MPT_X_POOL_STAMPS
#define PL_CNT(x)
#define PL_DECL unsigned long long pl_acc[3] = { 0, 0, 0 }; const unsigned long long pl_start = __builtin_amdgcn_s_memtime();
#define PL_STAMP(v) unsigned long long v = 0; if (COUNT) { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define PL_ACC(k, d) if (COUNT) pl_acc[k] += (d);
#define PL_FLUSH(f0, f1, f2, flife) if (COUNT) { const bool l0 = (threadIdx.x & 63) == 0; cnt.f0 = l0 ? (unsigned)(pl_acc[0] >> 8) : 0u; \
    cnt.f1 = l0 ? (unsigned)(pl_acc[1] >> 8) : 0u; cnt.f2 = l0 ? (unsigned)(pl_acc[2] >> 8) : 0u; \
    cnt.flife = l0 ? (unsigned)((__builtin_amdgcn_s_memtime() - pl_start) >> 8) : 0u; }
#else
#define PL_CNT(x) x
#define PL_DECL
#define PL_STAMP(v)
#define PL_ACC(k, d)
#define PL_FLUSH(f0, f1, f2, flife)
#endif

typedef __attribute__((address_space(3))) int *LdsIntPtr;
typedef __attribute__((address_space(3))) mpt_f4 *LdsVec4W;

// control words (ints in LDS): shade pool {tail, head, avail, space}, ray pool {tail, head, avail, space}, live, feeders,
// abort (a wave that met a protocol fault raised the host's watchdog flag: everybody leaves as soon as its lanes are done)
enum { PC_S = 0, PC_R = 4, PC_LIVE = 8, PC_FEEDERS = 9, PC_ABORT = 10, PC_WORDS = 16 };
enum { PC_TAIL = 0, PC_HEAD = 1, PC_AVAIL = 2, PC_SPACE = 3 };

struct PoolView {
    LdsIntPtr ctl;
    LdsIntPtr sflag, rflag;    // [CAP]: position + 1 of the record a slot holds
    LdsVec4W sq, rq;           // [VEC4][CAP]
};

DEV int lds_add(LdsIntPtr q, int v) { return __hip_atomic_fetch_add(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV int lds_get(LdsIntPtr q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV int lds_get_u(LdsIntPtr q) { return __builtin_amdgcn_readfirstlane(lds_get(q)); }   // wave-uniform (scalar branches)
DEV void lds_put(LdsIntPtr q, int v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// LDS executes one wave's instructions in issue order and there is one LDS per workgroup here, so ordering between
// waves only needs the COMPILER to keep the order of the accesses: a workgroup-scope fence
DEV void lds_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
DEV void lds_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

// wave-uniform: k slots of `space`, all or nothing (lane 0 does the arithmetic)
DEV bool pool_reserve(LdsIntPtr c, int k) {
    int ok = 0;
    if ((threadIdx.x & 63) == 0) {
        const int old = lds_add(c + PC_SPACE, -k);
        if (old >= k) ok = 1;
        else lds_add(c + PC_SPACE, k);
    }
    return __builtin_amdgcn_readfirstlane(ok) != 0;
}
DEV void pool_unreserve(LdsIntPtr c, int k) {
    if (k > 0 && (threadIdx.x & 63) == 0) lds_add(c + PC_SPACE, k);
}
// wave-uniform: first position of k consecutive ones of counter `which` (PC_TAIL for producers, PC_HEAD for consumers)
DEV int pool_positions(LdsIntPtr c, int which, int k) {
    int v = 0;
    if ((threadIdx.x & 63) == 0) v = lds_add(c + which, k);
    return __builtin_amdgcn_readfirstlane(v);
}
DEV void pool_publish(LdsIntPtr c, int k) {
    lds_release();
    if ((threadIdx.x & 63) == 0) lds_add(c + PC_AVAIL, k);
}
// wave-uniform: takes up to `want` published records; returns how many (0: none)
DEV int pool_take(LdsIntPtr c, int want) {
    int n = 0;
    if ((threadIdx.x & 63) == 0) {
        const int a = lds_get(c + PC_AVAIL);
        if (a > 0) {
            const int w = min(want, a);
            const int old = lds_add(c + PC_AVAIL, -w);
            if (old >= w) n = w;
            else { n = max(old, 0); lds_add(c + PC_AVAIL, w - n); }
        }
    }
    return __builtin_amdgcn_readfirstlane(n);
}
// a consumer's lane waits for its record: the producer is between its `tail +=` and its flag store.  Bounded: a
// protocol bug must not hang the GPU (false -> the caller raises the watchdog flag)
DEV bool pool_wait_flag(LdsIntPtr flag, int slot, int pos) {
    for (int t = 0; t < (1 << 22); t++) {
        if (lds_get(flag + slot) == pos + 1) { lds_acquire(); return true; }
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

DEV bool pool_finished(const PoolView &pv) {
    return lds_get_u(pv.ctl + PC_ABORT) != 0 || (lds_get_u(pv.ctl + PC_FEEDERS) == 0 && lds_get_u(pv.ctl + PC_LIVE) == 0);
}
DEV void pool_abort(const MptRenderParams &p, const PoolView &pv) {      // wave-uniform
    if ((threadIdx.x & 63) == 0) {
        __hip_atomic_store(p.watchdog, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        lds_put(pv.ctl + PC_ABORT, 1);
    }
}

DEV mpt_f4 f4(float a, float b, float c, float d) { mpt_f4 v; v.x = a; v.y = b; v.z = c; v.w = d; return v; }
DEV float asf_(int v) { return __int_as_float(v); }
DEV int asi_(float v) { return __float_as_int(v); }

// what travels besides the vectors: rng_k (15 bits) | depth << 16 (3) | frame << 19 (6) | kind << 30
#define MPT_POOL_KIND_SHADOW (1 << 30)
DEV int pool_pack(const LaneState &L) { return L.rng_k | (L.depth << 16) | (L.frame << 19); }
DEV void pool_unpack(int w, LaneState &L) { L.rng_k = w & 0x7fff; L.depth = (w >> 16) & 7; L.frame = (w >> 19) & 63; }

// ------------------------------------------------------------------ tracer waves
template <bool COUNT>
DEV void pool_path_ends(const PoolView &pv, bool ended) {     // wave-aggregated live -= (lanes whose path has just ended)
    const unsigned long long m = __ballot(ended);
    if (m != 0ull && (threadIdx.x & 63) == 0) lds_add(pv.ctl + PC_LIVE, -(int)__builtin_popcountll(m));
}

template <bool COUNT, class SCENE, class STACK>
DEV void pool_tracer(const MptRenderParams &p, const SCENE &sc, STACK stk, const PoolView pv, Cnt &cnt) {
    LaneState L;
    L.st = ST_NEW;                                    // NEW = free: the lane holds no path
    L.sp = 0; L.curr = 0; L.shadow = 0;
    L.result = v3s(0.0f); L.throughput = v3s(0.0f); L.prd = v3s(0.0f); L.direct = v3s(0.0f);
    L.to = v3s(0.0f); L.td = v3s(0.0f); L.inv = v3s(0.0f); L.oinv = v3s(0.0f);
    L.offx = 0; L.offy = 0; L.offz = 0;
    L.tbest = 0.0f; L.hidx = -1; L.hu = 0.0f; L.hv = 0.0f; L.last_brdf_pdf = 0.0f;
    L.navoid = 0; L.depth = 0; L.rng_i = 0; L.rng_k = 0; L.pix = 0; L.frame = 0;
    const int lane = threadIdx.x & 63;
    int burst = 0;                                    // traversal decisions to make before the next trip to the pools
    unsigned idle_polls = 0;
    PL_DECL
    for (unsigned guard = 0;; guard++) {
        if (guard > (1u << 26) || idle_polls > (1u << 21)) { pool_abort(p, pv); break; }
        // ---- traversal mode: until enough lanes wait (finished rays, free lanes) to make a trip to the pools worthwhile
        int trav;
        PL_STAMP(t0)
        for (;;) {
            const int cn = wave_count32(L.st == ST_NODE);
            const int cl = wave_count32(L.st == ST_LEAF);
            trav = cn + cl;
            if (trav == 0) break;
            if (burst > 0) burst--;
            else if (64 - trav >= MPT_POOL_LEAVE) break;
            if (cn * MPT_PREF_NODE >= cl * MPT_PREF_LEAF) {
                if (COUNT && lane == 0) cnt.it_node++;
                if (L.st == ST_NODE) stage_node<COUNT>(sc, stk, L, cnt);
#pragma unroll
                for (int rep = 0; rep < MPT_NODE_REP; rep++) {
                    if (__ballot(L.st == ST_NODE) == 0ull) break;
                    if (COUNT && lane == 0) cnt.it_node++;
                    if (L.st == ST_NODE) stage_node<COUNT>(sc, stk, L, cnt);
                }
            } else {
                if (COUNT && lane == 0) cnt.it_leaf++;
                if (L.st == ST_LEAF) stage_leaf<COUNT>(sc, stk, L, cnt);
#pragma unroll
                for (int rep = 0; rep < MPT_LEAF_REP; rep++) {
                    if (__ballot(L.st == ST_LEAF) == 0ull) break;
                    if (COUNT && lane == 0) cnt.it_leaf++;
                    if (L.st == ST_LEAF) stage_leaf<COUNT>(sc, stk, L, cnt);
                }
            }
        }
        PL_CNT(if (COUNT && lane == 0) cnt.pl_trips++;)
        PL_STAMP(t1)
        PL_ACC(0, t1 - t0)                            // (stamps build: cycles in traversal mode)
        // ---- shadow rays that have finished: settled here (path.py:51,56), the next bounce starts in the same lane
        {
            const bool sd = L.st == ST_DONE && L.shadow;
            if (__ballot(sd) != 0ull) {
                if (sd) stage_shadow_done<COUNT>(p, L, stk, cnt);
                pool_path_ends<COUNT>(pv, sd && L.st == ST_NEW);
            }
        }
        // ---- closest-hit rays that have finished: the bounce is a shader wave's work
        {
            const bool cd = L.st == ST_DONE && !L.shadow;
            const unsigned long long m = __ballot(cd);
            if (m != 0ull) {
                const int k = (int)__builtin_popcountll(m);
                if (pool_reserve(pv.ctl + PC_S, k)) {
                    const int first = pool_positions(pv.ctl + PC_S, PC_TAIL, k);
                    if (cd) {
                        const int pos = first + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        const int slot = pos & (MPT_POOL_S_CAP - 1);
                        pv.sq[0 * MPT_POOL_S_CAP + slot] = f4(L.to.x, L.to.y, L.to.z, L.tbest);
                        pv.sq[1 * MPT_POOL_S_CAP + slot] = f4(L.prd.x, L.prd.y, L.prd.z, asf_(L.hidx));
                        pv.sq[2 * MPT_POOL_S_CAP + slot] = f4(L.hu, L.hv, L.last_brdf_pdf, asf_(L.rng_i));
                        pv.sq[3 * MPT_POOL_S_CAP + slot] = f4(L.result.x, L.result.y, L.result.z, asf_(pool_pack(L)));
                        pv.sq[4 * MPT_POOL_S_CAP + slot] = f4(L.throughput.x, L.throughput.y, L.throughput.z, asf_(L.pix));
                        lds_release();
                        lds_put(pv.sflag + slot, pos + 1);
                        L.st = ST_NEW;
                    }
                    pool_publish(pv.ctl + PC_S, k);
                } else {
                    // the shade pool is full (the shaders are behind): do the bounce here, as the unspecialised kernel does
                    if (COUNT && lane == 0) cnt.it_shade++;
                    PL_CNT(if (COUNT && cd) cnt.pl_local++;)
                    if (cd) stage_shade<COUNT>(p, sc, L, stk, cnt);
                    pool_path_ends<COUNT>(pv, cd && L.st == ST_NEW);
                }
            }
        }
        // ---- free lanes take rays
        bool idle = false;
        {
            const bool fr = L.st == ST_NEW;
            const unsigned long long m = __ballot(fr);
            const int nfree = (int)__builtin_popcountll(m);
            int n = 0;
            if (nfree != 0) n = pool_take(pv.ctl + PC_R, nfree);
            if (n != 0) {
                const int first = pool_positions(pv.ctl + PC_R, PC_HEAD, n);
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                const bool mine = fr && rank < n;
                PL_CNT(if (COUNT && mine) cnt.pl_taken++;)
                bool bad = false;
                if (mine) {
                    const int pos = first + rank, slot = pos & (MPT_POOL_R_CAP - 1);
                    bad = !pool_wait_flag(pv.rflag, slot, pos);
                    const mpt_f4 r0 = pv.rq[0 * MPT_POOL_R_CAP + slot], r1 = pv.rq[1 * MPT_POOL_R_CAP + slot],
                                 r2 = pv.rq[2 * MPT_POOL_R_CAP + slot], r3 = pv.rq[3 * MPT_POOL_R_CAP + slot],
                                 r4 = pv.rq[4 * MPT_POOL_R_CAP + slot], r5 = pv.rq[5 * MPT_POOL_R_CAP + slot];
                    const int w = asi_(r4.w);
                    L.prd = v3(r2.x, r2.y, r2.z); L.last_brdf_pdf = r2.w;
                    L.direct = v3(r3.x, r3.y, r3.z); L.navoid = asi_(r3.w);
                    L.result = v3(r4.x, r4.y, r4.z); pool_unpack(w, L);
                    L.throughput = v3(r5.x, r5.y, r5.z); L.pix = asi_(r5.w);
                    L.rng_i = asi_(r1.w);
                    const V3 o = v3(r0.x, r0.y, r0.z);
                    if (w & MPT_POOL_KIND_SHADOW) lane_start_ray<COUNT>(L, stk, o, v3(r1.x, r1.y, r1.z), r0.w, true, cnt);
                    else lane_next_bounce<COUNT>(p, L, stk, o, cnt);        // head of the path_trace loop: may end the path here
                }
                lds_acquire();
                if (lane == 0) lds_add(pv.ctl + PC_R + PC_SPACE, n);        // the slots are read: theirs to refill
                pool_path_ends<COUNT>(pv, mine && L.st == ST_NEW);
                if (__ballot(bad) != 0ull) { pool_abort(p, pv); break; }
            } else if (nfree == 64) {
                idle = true;                          // no path in this wave and no ray to take
            } else if (nfree != 0) {
                burst = MPT_POOL_BURST;               // nothing to take: keep traversing for a while before asking again
            }
        }
        PL_STAMP(t2)
        PL_ACC(1, t2 - t1)                            // (stamps build: cycles in trips to the pools)
        if (idle) {
            if (pool_finished(pv)) break;
            idle_polls++;
            PL_CNT(if (COUNT && lane == 0) cnt.pl_tidle++;)
            __builtin_amdgcn_s_sleep(8);
            PL_STAMP(t3)
            PL_ACC(2, t3 - t2)                            // (stamps build: cycles idle)
        }
    }
    PL_FLUSH(pl_batches, pl_batch_lanes, pl_prim, pl_trips)
}

// ------------------------------------------------------------------ shader waves
template <bool COUNT, class SCENE>
DEV void pool_shader(const MptRenderParams &p, const SCENE &sc, const PoolView pv, WorkQueue wq, Cnt &cnt, unsigned long long *tl) {
    const int lane = threadIdx.x & 63;
    const int tws = p.tile_w_shift, ths = p.tile_h_shift, tps = tws + ths;
    const int t8y = (p.ny + (1 << ths) - 1) >> ths;
    int S = 0, next = 0, ti = 0, tj = 0, f0 = 0, tx_cur = 0;     // wave-uniform: the work item being turned into primary rays
    bool more = true;
#if MPT_POOL_SHADER_PRIO
    __builtin_amdgcn_s_setprio(MPT_POOL_SHADER_PRIO);  // a bounce is on every path's critical loop: the shader waves issue first
#endif
    int patience = 0;                                 // polls since this wave last had something to do
    unsigned idle_polls = 0;
    PL_DECL
    for (unsigned guard = 0;; guard++) {
        if (guard > (1u << 26) || idle_polls > (1u << 22)) { pool_abort(p, pv); break; }
        // Whatever this pass does produces rays: their slots are reserved BEFORE the requests are taken, so nothing is
        // ever held in registers waiting for space (and nobody waits for space at all)
        const int waiting = lds_get_u(pv.ctl + PC_S + PC_AVAIL);
        const bool starving = more && lds_get_u(pv.ctl + PC_R + PC_AVAIL) < MPT_POOL_RLOW;
        int want = 0;                                 // bounces to take
        if (waiting >= MPT_POOL_BATCH || (waiting > 0 && !starving && patience >= MPT_POOL_PATIENCE)) want = min(waiting, 64);
        const bool primaries = want == 0 && starving;
        int reserved = want != 0 ? want : (primaries ? 64 : 0);
        if (reserved != 0 && !pool_reserve(pv.ctl + PC_R, reserved)) reserved = 0;
        int produced = 0;                             // rays pushed by this pass
        bool did = false;
        PL_STAMP(s0)
        if (reserved != 0) {
            int n = 0;
            if (want != 0) n = pool_take(pv.ctl + PC_S, want);
            if (n != 0) {
                did = true;
                if (COUNT && lane == 0) cnt.it_shade++;
                PL_CNT(if (COUNT && lane == 0) { cnt.pl_batches++; cnt.pl_batch_lanes += n; })
                const int first = pool_positions(pv.ctl + PC_S, PC_HEAD, n);
                const bool mine = lane < n;
                LaneState L;
                L.st = ST_DONE; L.shadow = 0; L.sp = 0; L.curr = 0;
                L.result = v3s(0.0f); L.throughput = v3s(0.0f); L.prd = v3s(0.0f); L.direct = v3s(0.0f);
                L.to = v3s(0.0f); L.td = v3s(0.0f); L.inv = v3s(0.0f); L.oinv = v3s(0.0f);
                L.offx = 0; L.offy = 0; L.offz = 0;
                L.tbest = 0.0f; L.hidx = -1; L.hu = 0.0f; L.hv = 0.0f; L.last_brdf_pdf = 0.0f;
                L.navoid = 0; L.depth = 0; L.rng_i = 0; L.rng_k = 0; L.pix = 0; L.frame = 0;
                bool bad = false;
                if (mine) {
                    const int pos = first + lane, slot = pos & (MPT_POOL_S_CAP - 1);
                    bad = !pool_wait_flag(pv.sflag, slot, pos);
                    const mpt_f4 q0 = pv.sq[0 * MPT_POOL_S_CAP + slot], q1 = pv.sq[1 * MPT_POOL_S_CAP + slot],
                                 q2 = pv.sq[2 * MPT_POOL_S_CAP + slot], q3 = pv.sq[3 * MPT_POOL_S_CAP + slot],
                                 q4 = pv.sq[4 * MPT_POOL_S_CAP + slot];
                    L.to = v3(q0.x, q0.y, q0.z); L.tbest = q0.w;
                    L.prd = v3(q1.x, q1.y, q1.z); L.hidx = asi_(q1.w);
                    L.hu = q2.x; L.hv = q2.y; L.last_brdf_pdf = q2.z; L.rng_i = asi_(q2.w);
                    L.result = v3(q3.x, q3.y, q3.z); pool_unpack(asi_(q3.w), L);
                    L.throughput = v3(q4.x, q4.y, q4.z); L.pix = asi_(q4.w);
                }
                lds_acquire();
                if (lane == 0) lds_add(pv.ctl + PC_S + PC_SPACE, n);
                if (__ballot(bad) != 0ull) { pool_abort(p, pv); break; }
                V3 hitpos = v3s(0.0f), sdir = v3s(0.0f);
                float sdis = 0.0f;
                int nextk = SH_END;
                if (mine) nextk = shade_core<COUNT>(p, sc, L, cnt, hitpos, sdir, sdis);
                // a miss ends the path in the shader: the sample is stored (path.py:93; what lane_next_bounce does at depth 5)
                const bool ended = mine && nextk == SH_END;
                if (ended) {
                    store_sample(p, L.frame, L.pix, L.result);
                }
                pool_path_ends<COUNT>(pv, ended);
                const bool out = mine && nextk != SH_END;
                const unsigned long long mo = __ballot(out);
                produced = (int)__builtin_popcountll(mo);
                if (produced != 0) {
                    const int firstr = pool_positions(pv.ctl + PC_R, PC_TAIL, produced);
                    if (out) {
                        const int pos = firstr + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mo, 0u));
                        const int slot = pos & (MPT_POOL_R_CAP - 1);
                        const int kind = nextk == SH_SHADOW ? MPT_POOL_KIND_SHADOW : 0;
                        pv.rq[0 * MPT_POOL_R_CAP + slot] = f4(hitpos.x, hitpos.y, hitpos.z, sdis);
                        pv.rq[1 * MPT_POOL_R_CAP + slot] = f4(sdir.x, sdir.y, sdir.z, asf_(L.rng_i));
                        pv.rq[2 * MPT_POOL_R_CAP + slot] = f4(L.prd.x, L.prd.y, L.prd.z, L.last_brdf_pdf);
                        pv.rq[3 * MPT_POOL_R_CAP + slot] = f4(L.direct.x, L.direct.y, L.direct.z, asf_(L.navoid));
                        pv.rq[4 * MPT_POOL_R_CAP + slot] = f4(L.result.x, L.result.y, L.result.z, asf_(pool_pack(L) | kind));
                        pv.rq[5 * MPT_POOL_R_CAP + slot] = f4(L.throughput.x, L.throughput.y, L.throughput.z, asf_(L.pix));
                        lds_release();
                        lds_put(pv.rflag + slot, pos + 1);
                    }
                }
            } else if (primaries) {
                // ---- primary rays: the next 64 samples of the current work item (do_render up to the camera ray, path.py:82-90)
                if (next >= S) {
                    const int item = wq.pull();
                    if (item < 0) {
                        more = false;
                        if (lane == 0) lds_add(pv.ctl + PC_FEEDERS, -1);
                        if (tl && lane == 0) tl[2] = wall_clock64();
                    } else {
                        const int tile = item / p.nchunks, chunk = item - tile * p.nchunks;
                        const int tx = tile / t8y, ty = tile - tx * t8y;
                        const int tps_x = p.stripe_w >> tws, st = tx / tps_x;
                        ti = p.x0 + st * p.stripe_pitch + ((tx - st * tps_x) << tws); tj = ty << ths; tx_cur = tx;
                        f0 = chunk * p.chunk;
                        S = (min(f0 + p.chunk, p.nframes) - f0) << tps;
                        next = 0;
                    }
                }
                if (next < S) {
                    did = true;
                    if (COUNT && lane == 0) cnt.it_new++;
                    PL_CNT(if (COUNT && lane == 0) cnt.pl_prim++;)
                    const int smp = next + lane;
                    const int q = smp & ((1 << tps) - 1);
                    const int i = ti + (q >> ths), j = tj + (q & ((1 << ths) - 1));
                    const int frame = f0 + (smp >> tps);
                    const bool inside = smp < S && i < p.x1 && j < p.ny;
                    PrimaryPool pp;
                    pool_prepare(p, pp, inside, i, j, frame);
                    next = min(next + 64, S);
                    const unsigned long long mo = __ballot(inside);
                    produced = (int)__builtin_popcountll(mo);
                    if (produced != 0) {
                        if (lane == 0) lds_add(pv.ctl + PC_LIVE, produced);       // before the rays can be seen
                        const int firstr = pool_positions(pv.ctl + PC_R, PC_TAIL, produced);
                        if (inside) {
                            if (COUNT) { cnt.samples++; cnt.n_draws += 2; }
                            const int pos = firstr + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mo, 0u));
                            const int slot = pos & (MPT_POOL_R_CAP - 1);
                            const int pix = ((tx_cur << tws) + (q >> ths)) * p.ny + (tj + (q & ((1 << ths) - 1)));
                            const int w = pp.rng_k | (0 << 16) | (frame << 19);       // depth 0: lane_next_bounce takes it to 1
                            pv.rq[0 * MPT_POOL_R_CAP + slot] = f4(pp.ro.x, pp.ro.y, pp.ro.z, 0.0f);
                            pv.rq[1 * MPT_POOL_R_CAP + slot] = f4(0.0f, 0.0f, 0.0f, asf_(pp.rng_i));
                            pv.rq[2 * MPT_POOL_R_CAP + slot] = f4(pp.rd.x, pp.rd.y, pp.rd.z, 0.0f);   // prd = r.d, last_brdf_pdf = 0
                            pv.rq[3 * MPT_POOL_R_CAP + slot] = f4(0.0f, 0.0f, 0.0f, asf_(0));         // navoid 0: none
                            pv.rq[4 * MPT_POOL_R_CAP + slot] = f4(0.0f, 0.0f, 0.0f, asf_(w));         // result 0
                            pv.rq[5 * MPT_POOL_R_CAP + slot] = f4(1.0f, 1.0f, 1.0f, asf_(pix));       // throughput 1
                            lds_release();
                            lds_put(pv.rflag + slot, pos + 1);
                        }
                    }
                }
            }
            if (produced != 0) pool_publish(pv.ctl + PC_R, produced);
            pool_unreserve(pv.ctl + PC_R, reserved - produced);
        }
        if (did) {
            patience = 0; idle_polls = 0;
            PL_STAMP(s1)
            PL_ACC(want != 0 ? 0 : 1, s1 - s0)        // (stamps build: cycles in SHADE batches / in primary rays)
        } else {
            if (pool_finished(pv)) break;
            patience++; idle_polls++;
            PL_CNT(if (COUNT && lane == 0) cnt.pl_sidle++;)
            __builtin_amdgcn_s_sleep(4);
            PL_STAMP(s2)
            PL_ACC(2, s2 - s0)
        }
    }
    PL_FLUSH(pl_local, pl_taken, pl_sidle, pl_tidle)
}

// dynamic LDS: [ node records p.lds_node_stride apart, padded to 16 | n*3 triangle float4 (tfast) | (p.lds_nmats + 1) material
//   records of 6 float4, the default material last | n material-record bytes, padded to 16 | 64 B of control words |
//   flags S, flags R | shade pool | ray pool | int16 stacks [levels][tracer lanes] ]        (mpt_pool_lds_bytes)
template <bool COUNT>
__global__ __launch_bounds__(MPT_LDS_BLOCK) void render_kernel_pool(const MptRenderParams p) {
    extern __shared__ __attribute__((aligned(16))) MptVec4 smem[];
    const int nstride = p.lds_node_stride;
    const int nnode4 = ((p.n - 1) * nstride + 15) >> 4, ntri4 = p.n * 3, nmat4 = (p.lds_nmats + 1) * MPT_LDS_MAT_VEC4;
    const int nmtl4 = (p.n + 15) >> 4;
    const int nwaves = blockDim.x >> 6, wave = threadIdx.x >> 6, nshade = p.pool_shaders;
    unsigned long long *tl = p.timeline ? p.timeline + 4 * (size_t)(blockIdx.x * nwaves + wave) : nullptr;
    if (tl && (threadIdx.x & 63) == 0) { tl[0] = wall_clock64(); tl[2] = 0; }
    {   // one copy of the scene per CU
        for (int k = threadIdx.x; k < (p.n - 1) * 4; k += blockDim.x) {
            const MptVec4 v = p.fnode[k];
            float *d = (float *)((char *)smem + (k >> 2) * nstride + (k & 3) * 16);
            *(float2 *)d = make_float2(v.x, v.y); *(float2 *)(d + 2) = make_float2(v.z, v.w);
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
    MptVec4 *q = smem + nnode4 + ntri4 + nmat4 + nmtl4;
    PoolView pv;
    pv.ctl = (LdsIntPtr)(void *)q;                                   q += PC_WORDS / 4;
    pv.sflag = (LdsIntPtr)(void *)q;                                 q += MPT_POOL_S_CAP / 4;
    pv.rflag = (LdsIntPtr)(void *)q;                                 q += MPT_POOL_R_CAP / 4;
    pv.sq = (LdsVec4W)(void *)q;                                     q += MPT_POOL_S_VEC4 * MPT_POOL_S_CAP;
    pv.rq = (LdsVec4W)(void *)q;                                     q += MPT_POOL_R_VEC4 * MPT_POOL_R_CAP;
    for (int k = threadIdx.x; k < MPT_POOL_S_CAP; k += blockDim.x) pv.sflag[k] = 0;
    for (int k = threadIdx.x; k < MPT_POOL_R_CAP; k += blockDim.x) pv.rflag[k] = 0;
    if (threadIdx.x < PC_WORDS) {
        int v = 0;
        if (threadIdx.x == PC_S + PC_SPACE) v = MPT_POOL_S_CAP;
        if (threadIdx.x == PC_R + PC_SPACE) v = MPT_POOL_R_CAP;
        if (threadIdx.x == PC_FEEDERS) v = nshade;
        pv.ctl[threadIdx.x] = v;
    }
    __syncthreads();
    if (tl && (threadIdx.x & 63) == 0) tl[1] = wall_clock64();

    LdsScene sc;
    sc.fnode = (LdsVec4Ptr)(void *)smem;
    sc.tgeo = (LdsVec4Ptr)(void *)(smem + nnode4);
    sc.mats = (LdsVec4Ptr)(void *)(smem + nnode4 + ntri4);
    sc.mtl = (LdsU8Ptr)(void *)(smem + nnode4 + ntri4 + nmat4);
    sc.nstride = nstride;
    sc.mat_last = p.lds_nmats;
    sc.mat_default = p.default_mtl;
    Cnt cnt = {};
    if (wave < nshade) {
        WorkQueue wq; wq.ctr = p.work_counter; wq.nitems = p.nitems; wq.q0 = blockIdx.x & 7; wq.qoff = 0;
        pool_shader<COUNT>(p, sc, pv, wq, cnt, tl);
    } else {
        Stack16V stk;
        stk.stride = (nwaves - nshade) * 64;
        stk.base = (LdsShortPtr)(void *)q + (threadIdx.x - nshade * 64);
        stk.sp = 0;
        pool_tracer<COUNT>(p, sc, stk, pv, cnt);
    }
    if (tl && (threadIdx.x & 63) == 0) tl[3] = wall_clock64();
    flush_counters<COUNT>(p, cnt);
}

template <bool COUNT>
static hipError_t launch_pool(const MptRenderParams *p, int grid, int block, size_t lds_bytes, hipStream_t stream) {
    static std::atomic<bool> configured[MPT_MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MPT_MAX_DEVICES) return hipErrorInvalidDevice;
    if (!configured[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)render_kernel_pool<COUNT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        configured[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((render_kernel_pool<COUNT>), dim3(grid), dim3(block), lds_bytes, stream, *p);
    return hipGetLastError();
}

MPT_KERNEL_API hipError_t mpt_launch_render_pool(const MptRenderParams *p, int grid, int block, size_t lds_bytes, int count,
                                             hipStream_t stream) {
    return count ? launch_pool<true>(p, grid, block, lds_bytes, stream) : launch_pool<false>(p, grid, block, lds_bytes, stream);
}

// bytes of LDS the pooled kernel needs besides the scene records and the stacks
MPT_KERNEL_API size_t mpt_pool_lds_overhead(void) {
    return (size_t)PC_WORDS * 4 + (size_t)(MPT_POOL_S_CAP + MPT_POOL_R_CAP) * 4 +
           (size_t)(MPT_POOL_S_VEC4 * MPT_POOL_S_CAP + MPT_POOL_R_VEC4 * MPT_POOL_R_CAP) * sizeof(MptVec4);
}
