// sah_build.hip -- the SAH re-partition of the fast build's leaves, on the device.
//
// The production traversal walks a better tree than the reference's LBVH over the SAME leaf slots (the image does not
// depend on the tree's shape, only on which of two equal-depth hits wins).  Round 2 built it on the host (a threaded
// recursive pass: 0.22 s at 1 M triangles, plus the leaf order down and 64 MB of node records up).  Here it is built on
// the device, top-down and level by level:
//   prims     per leaf slot: box and box centre of its triangle
//   per level, for the segments (ranges of the slot permutation) with more than 32 triangles:
//     bounds   per segment: bounds of the centres and of the boxes (wave-reduced ordered-int atomics)
//     bin      32 bins per axis over the centre bounds: count and box per bin (atomics)
//     choose   one lane per segment: SAH cost of the 3 x 31 splits (area x count on both sides), best one; the node's
//              number follows from the range sizes (DFS pre-order: left child me + 1, right child me + left size), so no
//              counter is shared; children with <= 32 triangles go to the small list, single triangles become leaves
//     scatter  stable partition of every segment by its split (one global prefix sum of the predicates)
//   small     one lane per small segment: the exact sweep (every split of every axis, sorted by centre) of the host pass,
//              down to the leaves
//   pack      the 64-byte traversal records (both children's boxes + ids)
// Same cost function and tie rules as the host pass (tree_build.cpp SahBuild).  Segments between 33 and 8192 triangles are
// binned here where the host sweeps them exactly; what that costs depends on the scene and on the bin count (see SB_MAXBINS),
// so the bins are spent where they matter: as many per axis as a fixed budget divided by the level's segment count allows.
// Films agree up to equal-depth ties like any two trees (tests).  Deterministic: every
// decision is a function of sums of integers and min / max of floats.

#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cmath>
#include "mpt_types.h"

#define SB_BLOCK 256
#ifndef SB_MINBINS
#define SB_MINBINS 32          // bins per axis at the deep levels (many small segments) ...
#endif
#ifndef SB_MAXBINS
#define SB_MAXBINS 1024        // ... and near the root: as many as SB_BIN_BUDGET / segments allows.  Measured on the 99 382-triangle
#endif                         // scene of BASELINE config 4 (a dense mesh inside ten huge wall triangles): 32 bins everywhere 10.5
#ifndef SB_BIN_BUDGET          // node fetches per ray, 128 everywhere 9.97, the host pass (exact sweep up to 8192 leaves) 9.44
#define SB_BIN_BUDGET (1 << 20)
#endif
#ifndef SB_SMALL
#define SB_SMALL 32            // segments of at most this many triangles are finished by one lane (exact sweep)
#endif
#define SB_LDS_BINS 64         // up to this many bins a segment-uniform workgroup bins into LDS first
// words per segment at nb bins: bounds (12) | counts [3][nb] | boxes [3][nb][6] | suffix areas [3][nb] | suffix counts [3][nb]
#define SB_SEG_WORDS(nb) (12 + 27 * (nb))

__device__ __forceinline__ int sb_f2ord(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float sb_ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__device__ __forceinline__ float sb_half_area(const float *l, const float *h) {
    const float dx = fmaxf(h[0] - l[0], 0.f), dy = fmaxf(h[1] - l[1], 0.f), dz = fmaxf(h[2] - l[2], 0.f);
    return dx * dy + dy * dz + dz * dx;
}

// per leaf slot: box and centre of its triangle (verts [3n][8], leaf: slot -> face)
__global__ __launch_bounds__(SB_BLOCK) void sb_prims_kernel(const float *__restrict__ verts, const int *__restrict__ leaf, int n,
                                                           float *__restrict__ plo, float *__restrict__ phi, float *__restrict__ pct,
                                                           int *__restrict__ idx, int *__restrict__ seg) {
    const int slot = blockIdx.x * SB_BLOCK + threadIdx.x;
    if (slot >= n) return;
    const float *p0 = verts + (size_t)leaf[slot] * 24, *p1 = p0 + 8, *p2 = p0 + 16;
    for (int a = 0; a < 3; a++) {
        const float l = fminf(fminf(p0[a], p1[a]), p2[a]), h = fmaxf(fmaxf(p0[a], p1[a]), p2[a]);
        plo[(size_t)slot * 3 + a] = l; phi[(size_t)slot * 3 + a] = h; pct[(size_t)slot * 3 + a] = 0.5f * (l + h);
    }
    idx[slot] = slot; seg[slot] = 0;
}

// per-segment words: [0..5] centre bounds lo3 hi3, [6..11] box bounds lo3 hi3 (ordered ints), then bin counts [3][32],
// then bin boxes [3][32][6] (ordered ints)
__global__ __launch_bounds__(SB_BLOCK) void sb_reset_kernel(int nseg, int nb, int *__restrict__ sw) {
    const size_t sws = SB_SEG_WORDS(nb);
    const size_t t = (size_t)blockIdx.x * SB_BLOCK + threadIdx.x;
    if (t >= (size_t)nseg * sws) return;
    const int w = (int)(t % sws);
    int v = 0;
    if (w < 12) v = (w % 6) < 3 ? 0x7fffffff : (int)0x80000000;
    else if (w >= 12 + 3 * nb && w < 12 + 21 * nb) v = ((w - 12 - 3 * nb) % 6) < 3 ? 0x7fffffff : (int)0x80000000;
    sw[t] = v;
}

__global__ __launch_bounds__(SB_BLOCK) void sb_bounds_kernel(int n, const int *__restrict__ idx, const int *__restrict__ seg,
                                                            const float *__restrict__ pct, const float *__restrict__ plo,
                                                            const float *__restrict__ phi, int nb, int *__restrict__ sw) {
    const size_t sws = SB_SEG_WORDS(nb);
    const int i = blockIdx.x * SB_BLOCK + threadIdx.x;
    const int s = i < n ? seg[i] : -1;
    int v[12];
    if (s >= 0) {
        const int slot = idx[i];
        for (int a = 0; a < 3; a++) {
            const int c = sb_f2ord(pct[(size_t)slot * 3 + a]);
            v[a] = c; v[3 + a] = c;
            v[6 + a] = sb_f2ord(plo[(size_t)slot * 3 + a]); v[9 + a] = sb_f2ord(phi[(size_t)slot * 3 + a]);
        }
    } else {
        for (int k = 0; k < 12; k++) v[k] = (k % 6) < 3 ? 0x7fffffff : (int)0x80000000;
    }
    // a wave whose lanes all belong to one segment (the rule near the root) reduces first: one atomic per word and wave
    const int s0 = __shfl(s, 0);
    if (__all(s == s0)) {
        if (s0 < 0) return;
        for (int k = 0; k < 12; k++)
            for (int off = 32; off > 0; off >>= 1) {
                const int o = __shfl_xor(v[k], off);
                v[k] = (k % 6) < 3 ? min(v[k], o) : max(v[k], o);
            }
        if ((threadIdx.x & 63) == 0)
            for (int k = 0; k < 12; k++) {
                if ((k % 6) < 3) atomicMin(sw + (size_t)s0 * sws + k, v[k]);
                else atomicMax(sw + (size_t)s0 * sws + k, v[k]);
            }
    } else if (s >= 0) {
        for (int k = 0; k < 12; k++) {
            if ((k % 6) < 3) atomicMin(sw + (size_t)s * sws + k, v[k]);
            else atomicMax(sw + (size_t)s * sws + k, v[k]);
        }
    }
}

__device__ __forceinline__ int sb_bin_of(float c, float cl, float scale, int nb) {
    return min(nb - 1, max(0, (int)((c - cl) * scale)));
}

__global__ __launch_bounds__(SB_BLOCK) void sb_bin_kernel(int n, const int *__restrict__ idx, const int *__restrict__ seg,
                                                         const float *__restrict__ pct, const float *__restrict__ plo,
                                                         const float *__restrict__ phi, int nb, int *__restrict__ sw) {
    // A workgroup whose 256 positions all belong to one segment bins into LDS and merges once (with few bins a segment's
    // lanes would otherwise hammer the same few words: 14 ms for the first level of a million triangles at 32 bins)
    __shared__ int lb[3 * SB_LDS_BINS * 7];           // [axis][bin]{count, lo3, hi3}
    const size_t sws = SB_SEG_WORDS(nb);
    const int i = blockIdx.x * SB_BLOCK + threadIdx.x;
    const int s = i < n ? seg[i] : -2;
    const int sfirst = seg[min((int)(blockIdx.x * SB_BLOCK), n - 1)];
    const bool uniform = nb <= SB_LDS_BINS && __syncthreads_and(s == sfirst || s == -2) != 0;
    if (uniform) {
        if (sfirst < 0) return;
        for (int k = threadIdx.x; k < 3 * nb * 7; k += SB_BLOCK) {
            const int f = k % 7;
            lb[k] = f == 0 ? 0 : (f < 4 ? 0x7fffffff : (int)0x80000000);
        }
        __syncthreads();
    }
    if (s >= 0) {
        const int slot = idx[i];
        int *w = sw + (size_t)s * sws;
        int bl[3], bh[3];
        for (int a = 0; a < 3; a++) { bl[a] = sb_f2ord(plo[(size_t)slot * 3 + a]); bh[a] = sb_f2ord(phi[(size_t)slot * 3 + a]); }
        for (int a = 0; a < 3; a++) {
            const float cl = sb_ord2f(w[a]), ch = sb_ord2f(w[3 + a]);
            if (!(ch > cl)) continue;
            const float scale = nb / (ch - cl);
            const int q = sb_bin_of(pct[(size_t)slot * 3 + a], cl, scale, nb);
            if (uniform) {
                int *e = lb + (a * nb + q) * 7;
                atomicAdd(e, 1);
                for (int r = 0; r < 3; r++) { atomicMin(e + 1 + r, bl[r]); atomicMax(e + 4 + r, bh[r]); }
            } else {
                atomicAdd(w + 12 + a * nb + q, 1);
                int *bb = w + 12 + 3 * nb + (a * nb + q) * 6;
                for (int r = 0; r < 3; r++) { atomicMin(bb + r, bl[r]); atomicMax(bb + 3 + r, bh[r]); }
            }
        }
    }
    if (uniform) {
        __syncthreads();
        int *w = sw + (size_t)sfirst * sws;
        for (int k = threadIdx.x; k < 3 * nb; k += SB_BLOCK) {
            const int *e = lb + k * 7;
            if (e[0] == 0) continue;
            atomicAdd(w + 12 + k, e[0]);
            int *bb = w + 12 + 3 * nb + k * 6;
            for (int r = 0; r < 3; r++) { atomicMin(bb + r, e[1 + r]); atomicMax(bb + 3 + r, e[4 + r]); }
        }
    }
}

__device__ __forceinline__ void sb_emit(int s, int b, int e, int me, const int *w, int best_axis, int best_k, int best_q, int level,
                                        int *__restrict__ child, float *__restrict__ blo, float *__restrict__ bhi, int *__restrict__ dec,
                                        int *__restrict__ flag, int *__restrict__ small, int *__restrict__ counters);

// dec[s] = {axis (-1: split the range in half), first bin of the right side, m (first position of the right side), -}
// flag[2 s + k] = 1: child k is a segment of the next level.  small: {b, e, node, level} per small segment.
__global__ __launch_bounds__(SB_BLOCK) void sb_choose_kernel(int nseg, const int *__restrict__ sb, const int *__restrict__ se,
                                                            const int *__restrict__ snode, int nb, int *__restrict__ sw, int level,
                                                            int *__restrict__ child, float *__restrict__ blo, float *__restrict__ bhi,
                                                            int *__restrict__ dec, int *__restrict__ flag, int *__restrict__ small,
                                                            int *__restrict__ counters) {
    const int s = blockIdx.x * SB_BLOCK + threadIdx.x;
    if (s >= nseg) return;
    const int b = sb[s], e = se[s], me = snode[s];
    int *w = sw + (size_t)s * SB_SEG_WORDS(nb);
    float best = INFINITY;
    int best_axis = -1, best_k = -1, best_q = 0;
    for (int a = 0; a < 3; a++) {
        const float cl = sb_ord2f(w[a]), ch = sb_ord2f(w[3 + a]);
        if (!(ch > cl)) continue;
        const int *bc = w + 12 + a * nb;
        const int *bb = w + 12 + 3 * nb + a * nb * 6;
        float *ra = (float *)(w + 12 + 21 * nb + a * nb);         // suffix areas / counts: the segment's own scratch words
        int *rc = w + 12 + 24 * nb + a * nb;
        float l[3] = { INFINITY, INFINITY, INFINITY }, h[3] = { -INFINITY, -INFINITY, -INFINITY };
        int c2 = 0;
        for (int q = nb - 1; q > 0; q--) {
            c2 += bc[q];
            if (bc[q]) for (int r = 0; r < 3; r++) { l[r] = fminf(l[r], sb_ord2f(bb[q * 6 + r])); h[r] = fmaxf(h[r], sb_ord2f(bb[q * 6 + 3 + r])); }
            ra[q] = sb_half_area(l, h); rc[q] = c2;
        }
        for (int r = 0; r < 3; r++) { l[r] = INFINITY; h[r] = -INFINITY; }
        int c1 = 0;
        for (int q = 1; q < nb; q++) {
            c1 += bc[q - 1];
            if (bc[q - 1]) for (int r = 0; r < 3; r++) { l[r] = fminf(l[r], sb_ord2f(bb[(q - 1) * 6 + r])); h[r] = fmaxf(h[r], sb_ord2f(bb[(q - 1) * 6 + 3 + r])); }
            if (c1 == 0 || rc[q] == 0) continue;
            const float cost = sb_half_area(l, h) * c1 + ra[q] * rc[q];
            if (cost < best) { best = cost; best_axis = a; best_k = c1; best_q = q; }
        }
    }
    sb_emit(s, b, e, me, w, best_axis, best_k, best_q, level, child, blo, bhi, dec, flag, small, counters);
}

// the tail of a segment's decision, shared by both choose kernels: node box, split, children
__device__ __forceinline__ void sb_emit(int s, int b, int e, int me, const int *w, int best_axis, int best_k, int best_q, int level,
                                        int *__restrict__ child, float *__restrict__ blo, float *__restrict__ bhi, int *__restrict__ dec,
                                        int *__restrict__ flag, int *__restrict__ small, int *__restrict__ counters) {
    const int cnt = e - b;
    for (int a = 0; a < 3; a++) { blo[(size_t)me * 3 + a] = sb_ord2f(w[6 + a]); bhi[(size_t)me * 3 + a] = sb_ord2f(w[9 + a]); }
    const int m = best_axis < 0 ? b + cnt / 2 : b + best_k;          // all centres equal: split the range in half
    dec[s * 4 + 0] = best_axis; dec[s * 4 + 1] = best_q; dec[s * 4 + 2] = m; dec[s * 4 + 3] = 0;
    const int node[2] = { me + 1, me + (m - b) };
    const int lo_[2] = { b, m }, hi_[2] = { m, e };
    for (int k = 0; k < 2; k++) {
        const int sz = hi_[k] - lo_[k];
        flag[2 * s + k] = 0;
        if (sz == 1) continue;                                       // a leaf: the scatter pass writes ~slot
        child[(size_t)me * 2 + k] = node[k];
        if (sz <= SB_SMALL) {
            const int at = atomicAdd(counters + 0, 1);
            small[at * 4 + 0] = lo_[k]; small[at * 4 + 1] = hi_[k]; small[at * 4 + 2] = node[k]; small[at * 4 + 3] = level + 1;
        } else {
            flag[2 * s + k] = 1;
        }
    }
    atomicMax(counters + 1, level);
}

// the same decision by one WAVE per segment, for the levels with many bins per axis (few, big segments: one lane walking
// 3 x 1024 bins through global memory took 15 ms per level): lane l owns nb / 64 consecutive bins, the boxes and counts of
// the lanes before / after it come from wave scans, and the lanes' best splits are reduced with the serial loop's tie rule
// (lowest cost, then lowest axis, then lowest bin)
struct SbAgg { int c; float l[3], h[3]; };
__device__ __forceinline__ void sb_agg_add(SbAgg &a, const SbAgg &b) {
    a.c += b.c;
    for (int r = 0; r < 3; r++) { a.l[r] = fminf(a.l[r], b.l[r]); a.h[r] = fmaxf(a.h[r], b.h[r]); }
}
__device__ __forceinline__ SbAgg sb_agg_shfl_up(const SbAgg &a, int d) {
    SbAgg o; o.c = __shfl_up(a.c, d);
    for (int r = 0; r < 3; r++) { o.l[r] = __shfl_up(a.l[r], d); o.h[r] = __shfl_up(a.h[r], d); }
    return o;
}
__device__ __forceinline__ SbAgg sb_agg_shfl_down(const SbAgg &a, int d) {
    SbAgg o; o.c = __shfl_down(a.c, d);
    for (int r = 0; r < 3; r++) { o.l[r] = __shfl_down(a.l[r], d); o.h[r] = __shfl_down(a.h[r], d); }
    return o;
}
__global__ __launch_bounds__(64) void sb_choose_wave_kernel(int nseg, const int *__restrict__ sb, const int *__restrict__ se,
                                                           const int *__restrict__ snode, int nb, const int *__restrict__ sw, int level,
                                                           int *__restrict__ child, float *__restrict__ blo, float *__restrict__ bhi,
                                                           int *__restrict__ dec, int *__restrict__ flag, int *__restrict__ small,
                                                           int *__restrict__ counters) {
    const int s = blockIdx.x, lane = threadIdx.x;
    if (s >= nseg) return;
    const int b = sb[s], e = se[s], me = snode[s];
    const int *w = sw + (size_t)s * SB_SEG_WORDS(nb);
    const int per = nb >> 6;                          // bins per lane (nb is a multiple of 64 here, at most 16 per lane)
    float best = INFINITY;
    int best_axis = -1, best_k = -1, best_q = 0;
    for (int a = 0; a < 3; a++) {
        const float cl = sb_ord2f(w[a]), ch = sb_ord2f(w[3 + a]);
        if (!(ch > cl)) continue;                     // (wave-uniform)
        const int *bc = w + 12 + a * nb;
        const int *bb = w + 12 + 3 * nb + a * nb * 6;
        const int q0 = lane * per;
        SbAgg own; own.c = 0;
        for (int r = 0; r < 3; r++) { own.l[r] = INFINITY; own.h[r] = -INFINITY; }
        for (int t = 0; t < per; t++) {
            const int q = q0 + t, c = bc[q];
            if (c) {
                own.c += c;
                for (int r = 0; r < 3; r++) { own.l[r] = fminf(own.l[r], sb_ord2f(bb[q * 6 + r])); own.h[r] = fmaxf(own.h[r], sb_ord2f(bb[q * 6 + 3 + r])); }
            }
        }
        // what lies in the lanes before this one (exclusive prefix) and after it (exclusive suffix)
        SbAgg pre = own, suf = own;
        for (int d = 1; d < 64; d <<= 1) {
            SbAgg o = sb_agg_shfl_up(pre, d);
            if (lane >= d) sb_agg_add(pre, o);
            SbAgg u = sb_agg_shfl_down(suf, d);
            if (lane + d < 64) sb_agg_add(suf, u);
        }
        SbAgg left = sb_agg_shfl_up(pre, 1), right = sb_agg_shfl_down(suf, 1);
        if (lane == 0) { left.c = 0; for (int r = 0; r < 3; r++) { left.l[r] = INFINITY; left.h[r] = -INFINITY; } }
        if (lane == 63) { right.c = 0; for (int r = 0; r < 3; r++) { right.l[r] = INFINITY; right.h[r] = -INFINITY; } }
        // the right side of a split at the lane's bin t = its bins t .. per-1 + the lanes after it: suffixes inside the chunk
        float ral[16][3], rah[16][3]; int rcn[16];
        {
            SbAgg acc = right;
            for (int t = per - 1; t >= 0; t--) {
                const int q = q0 + t, c = bc[q];
                if (c) {
                    acc.c += c;
                    for (int r = 0; r < 3; r++) { acc.l[r] = fminf(acc.l[r], sb_ord2f(bb[q * 6 + r])); acc.h[r] = fmaxf(acc.h[r], sb_ord2f(bb[q * 6 + 3 + r])); }
                }
                rcn[t] = acc.c;
                for (int r = 0; r < 3; r++) { ral[t][r] = acc.l[r]; rah[t][r] = acc.h[r]; }
            }
        }
        // candidate splits: left = bins 0 .. q-1, right = bins q .. nb-1, for q = q0 + t (q >= 1)
        float lbest = INFINITY; int lk = -1, lq = 0;
        SbAgg acc = left;
        for (int t = 0; t < per; t++) {
            const int q = q0 + t;
            if (q >= 1 && acc.c != 0 && rcn[t] != 0) {
                const float cost = sb_half_area(acc.l, acc.h) * acc.c + sb_half_area(ral[t], rah[t]) * rcn[t];
                if (cost < lbest) { lbest = cost; lk = acc.c; lq = q; }
            }
            const int c = bc[q];
            if (c) {
                acc.c += c;
                for (int r = 0; r < 3; r++) { acc.l[r] = fminf(acc.l[r], sb_ord2f(bb[q * 6 + r])); acc.h[r] = fmaxf(acc.h[r], sb_ord2f(bb[q * 6 + 3 + r])); }
            }
        }
        // wave argmin: lowest cost, ties to the lowest bin (= the serial loop's first strict minimum)
        for (int d = 32; d > 0; d >>= 1) {
            const float oc = __shfl_xor(lbest, d); const int ok = __shfl_xor(lk, d), oq = __shfl_xor(lq, d);
            if (oc < lbest || (oc == lbest && ok >= 0 && (lk < 0 || oq < lq))) { lbest = oc; lk = ok; lq = oq; }
        }
        if (lk >= 0 && lbest < best) { best = lbest; best_axis = a; best_k = lk; best_q = lq; }
    }
    if (lane == 0) sb_emit(s, b, e, me, w, best_axis, best_k, best_q, level, child, blo, bhi, dec, flag, small, counters);
}

__global__ __launch_bounds__(SB_BLOCK) void sb_newseg_kernel(int nseg, const int *__restrict__ sb, const int *__restrict__ se,
                                                            const int *__restrict__ snode, const int *__restrict__ dec,
                                                            const int *__restrict__ flag, const int *__restrict__ foff,
                                                            int *__restrict__ sb2, int *__restrict__ se2, int *__restrict__ snode2) {
    const int s = blockIdx.x * SB_BLOCK + threadIdx.x;
    if (s >= nseg) return;
    const int b = sb[s], e = se[s], me = snode[s], m = dec[s * 4 + 2];
    if (flag[2 * s + 0]) { const int ns = foff[2 * s + 0]; sb2[ns] = b; se2[ns] = m; snode2[ns] = me + 1; }
    if (flag[2 * s + 1]) { const int ns = foff[2 * s + 1]; sb2[ns] = m; se2[ns] = e; snode2[ns] = me + (m - b); }
}

__global__ __launch_bounds__(SB_BLOCK) void sb_pred_kernel(int n, const int *__restrict__ idx, const int *__restrict__ seg,
                                                          const float *__restrict__ pct, int nb, const int *__restrict__ sw,
                                                          const int *__restrict__ dec, int *__restrict__ pred) {
    const int i = blockIdx.x * SB_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int s = seg[i];
    int p = 0;
    if (s >= 0) {
        const int a = dec[s * 4 + 0];
        if (a < 0) p = i < dec[s * 4 + 2];
        else {
            const int *w = sw + (size_t)s * SB_SEG_WORDS(nb);
            const float cl = sb_ord2f(w[a]), ch = sb_ord2f(w[3 + a]);
            const float scale = nb / (ch - cl);
            p = sb_bin_of(pct[(size_t)idx[i] * 3 + a], cl, scale, nb) < dec[s * 4 + 1];
        }
    }
    pred[i] = p;
}

__global__ __launch_bounds__(SB_BLOCK) void sb_scatter_kernel(int n, const int *__restrict__ idx, const int *__restrict__ seg,
                                                             const int *__restrict__ pred, const int *__restrict__ pscan,
                                                             const int *__restrict__ sb, const int *__restrict__ se,
                                                             const int *__restrict__ snode, const int *__restrict__ dec,
                                                             const int *__restrict__ flag, const int *__restrict__ foff,
                                                             int *__restrict__ child, int *__restrict__ idx2, int *__restrict__ seg2) {
    const int i = blockIdx.x * SB_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int s = seg[i], slot = idx[i];
    if (s < 0) { idx2[i] = slot; seg2[i] = -1; return; }
    const int b = sb[s], e = se[s], m = dec[s * 4 + 2], me = snode[s];
    const int left_before = pscan[i] - pscan[b];
    const int k = pred[i] ? 0 : 1;
    const int dest = k == 0 ? b + left_before : m + ((i - b) - left_before);
    idx2[dest] = slot;
    const int sz = k == 0 ? m - b : e - m;
    if (sz == 1) child[(size_t)me * 2 + k] = ~slot;
    seg2[dest] = flag[2 * s + k] ? foff[2 * s + k] : -1;
}

// one lane per small segment: exact sweep SAH (tree_build.cpp SahBuild::split, the path for ranges <= 8192) down to the leaves
__global__ __launch_bounds__(64) void sb_small_kernel(int nsmall, const int *__restrict__ small, int *__restrict__ idx,
                                                     const float *__restrict__ pct, const float *__restrict__ plo,
                                                     const float *__restrict__ phi, int *__restrict__ child, float *__restrict__ blo,
                                                     float *__restrict__ bhi, int *__restrict__ counters) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= nsmall) return;
    const int b0 = small[t * 4 + 0], e0 = small[t * 4 + 1];
    int ids[SB_SMALL];
    for (int k = 0; k < e0 - b0; k++) ids[k] = idx[b0 + k];
    int stb[SB_SMALL], ste[SB_SMALL], stn[SB_SMALL], std_[SB_SMALL];
    int sp = 0, maxdep = 0;
    stb[0] = 0; ste[0] = e0 - b0; stn[0] = small[t * 4 + 2]; std_[0] = small[t * 4 + 3]; sp = 1;
    while (sp > 0) {
        sp--;
        const int b = stb[sp], e = ste[sp], me = stn[sp], dep = std_[sp], cnt = e - b;
        maxdep = max(maxdep, dep);
        float l[3] = { INFINITY, INFINITY, INFINITY }, h[3] = { -INFINITY, -INFINITY, -INFINITY };
        float cl[3] = { INFINITY, INFINITY, INFINITY }, ch[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (int q = b; q < e; q++)
            for (int a = 0; a < 3; a++) {
                l[a] = fminf(l[a], plo[(size_t)ids[q] * 3 + a]); h[a] = fmaxf(h[a], phi[(size_t)ids[q] * 3 + a]);
                cl[a] = fminf(cl[a], pct[(size_t)ids[q] * 3 + a]); ch[a] = fmaxf(ch[a], pct[(size_t)ids[q] * 3 + a]);
            }
        for (int a = 0; a < 3; a++) { blo[(size_t)me * 3 + a] = l[a]; bhi[(size_t)me * 3 + a] = h[a]; }
        float best = INFINITY;
        int best_axis = -1, best_k = -1;
        int ord[SB_SMALL];
        float rarea[SB_SMALL];
        for (int a = 0; a < 3; a++) {
            if (!(ch[a] > cl[a])) continue;
            for (int q = 0; q < cnt; q++) {                        // insertion sort by (centre, slot): std::sort on pairs
                const int s = ids[b + q];
                const float key = pct[(size_t)s * 3 + a];
                int p = q;
                while (p > 0) {
                    const int o = ord[p - 1];
                    const float ko = pct[(size_t)o * 3 + a];
                    if (ko < key || (ko == key && o < s)) break;
                    ord[p] = o; p--;
                }
                ord[p] = s;
            }
            float rl[3] = { INFINITY, INFINITY, INFINITY }, rh[3] = { -INFINITY, -INFINITY, -INFINITY };
            for (int q = cnt - 1; q > 0; q--) {
                const int s = ord[q];
                for (int r = 0; r < 3; r++) { rl[r] = fminf(rl[r], plo[(size_t)s * 3 + r]); rh[r] = fmaxf(rh[r], phi[(size_t)s * 3 + r]); }
                rarea[q] = sb_half_area(rl, rh);
            }
            for (int r = 0; r < 3; r++) { rl[r] = INFINITY; rh[r] = -INFINITY; }
            for (int k = 1; k < cnt; k++) {
                const int s = ord[k - 1];
                for (int r = 0; r < 3; r++) { rl[r] = fminf(rl[r], plo[(size_t)s * 3 + r]); rh[r] = fmaxf(rh[r], phi[(size_t)s * 3 + r]); }
                const float cost = sb_half_area(rl, rh) * k + rarea[k] * (cnt - k);
                if (cost < best) { best = cost; best_axis = a; best_k = k; }
            }
        }
        int m = b + cnt / 2;
        if (best_axis >= 0) {
            const int a = best_axis;
            for (int q = 0; q < cnt; q++) {
                const int s = ids[b + q];
                const float key = pct[(size_t)s * 3 + a];
                int p = q;
                while (p > 0) {
                    const int o = ord[p - 1];
                    const float ko = pct[(size_t)o * 3 + a];
                    if (ko < key || (ko == key && o < s)) break;
                    ord[p] = o; p--;
                }
                ord[p] = s;
            }
            for (int q = 0; q < cnt; q++) ids[b + q] = ord[q];
            m = b + best_k;
        }
        const int left = me + 1, right = me + (m - b);
        if (m - b == 1) child[(size_t)me * 2 + 0] = ~ids[b];
        else { child[(size_t)me * 2 + 0] = left; stb[sp] = b; ste[sp] = m; stn[sp] = left; std_[sp] = dep + 1; sp++; }
        if (e - m == 1) child[(size_t)me * 2 + 1] = ~ids[m];
        else { child[(size_t)me * 2 + 1] = right; stb[sp] = m; ste[sp] = e; stn[sp] = right; std_[sp] = dep + 1; sp++; }
    }
    atomicMax(counters + 1, maxdep);
}

// the 64-byte traversal record: {c0.lo.x, c1.lo.x, c0.hi.x, c1.hi.x} {y} {z} {id0, id1, 0, 0}
__global__ __launch_bounds__(SB_BLOCK) void sb_pack_kernel(int ni, const int *__restrict__ child, const float *__restrict__ blo,
                                                          const float *__restrict__ bhi, const float *__restrict__ plo,
                                                          const float *__restrict__ phi, MptVec4 *__restrict__ fnode) {
    const int i = blockIdx.x * SB_BLOCK + threadIdx.x;
    if (i >= ni) return;
    float l[2][3], h[2][3];
    int id[2];
    for (int k = 0; k < 2; k++) {
        id[k] = child[(size_t)i * 2 + k];
        const float *sl = id[k] < 0 ? plo + (size_t)(~id[k]) * 3 : blo + (size_t)id[k] * 3;
        const float *sh = id[k] < 0 ? phi + (size_t)(~id[k]) * 3 : bhi + (size_t)id[k] * 3;
        for (int a = 0; a < 3; a++) { l[k][a] = sl[a]; h[k][a] = sh[a]; }
    }
    for (int a = 0; a < 3; a++) fnode[(size_t)i * 4 + a] = { l[0][a], l[1][a], h[0][a], h[1][a] };
    fnode[(size_t)i * 4 + 3] = { __int_as_float(id[0]), __int_as_float(id[1]), 0.f, 0.f };
}

MPT_KERNEL_API size_t mpt_sah_seg_capacity(int n) { return (size_t)n / (SB_SMALL + 1) + 2; }
// words of segment workspace: enough for every level's (segments x words at that level's bin count)
// A level with more than SB_MINBINS bins keeps segments x bins within SB_BIN_BUDGET (the rule in mpt_sah_build); once the
// segments outnumber SB_BIN_BUDGET / (2 SB_MINBINS) the bin count stays at SB_MINBINS and the level needs segments x
// SB_SEG_WORDS(SB_MINBINS) words, up to the segment capacity: the larger of the two bounds (round-3 ADVICE: the first alone
// is too small above ~1.08 M faces).
MPT_KERNEL_API size_t mpt_sah_level_words(size_t nseg, int *nb_out) {
    int nb = SB_MINBINS;
    while (nb < SB_MAXBINS && nseg * (size_t)(2 * nb) <= (size_t)SB_BIN_BUDGET) nb *= 2;
    if (nb_out) *nb_out = nb;
    return nseg * SB_SEG_WORDS(nb);
}
MPT_KERNEL_API size_t mpt_sah_seg_words(int n) {
    const size_t sc = mpt_sah_seg_capacity(n);
    const size_t binned = 12 * sc + 27 * (size_t)SB_BIN_BUDGET + SB_SEG_WORDS(SB_MAXBINS);
    const size_t deep = sc * SB_SEG_WORDS(SB_MINBINS);
    return binned > deep ? binned : deep;
}
MPT_KERNEL_API hipError_t mpt_sah_scan_bytes(int n, size_t *bytes) {
    int *p = nullptr;
    return rocprim::exclusive_scan(nullptr, *bytes, p, p, 0, (size_t)std::max(n, 2), rocprim::plus<int>());
}

// verts [3n][8] and leaf [n] on the device (the LBVH build's); writes fnode [n-1][4] and *depth.  Needs n > SB_SMALL.
MPT_KERNEL_API hipError_t mpt_sah_build(const MptSahBuffers *B, int *depth, hipStream_t stream) {
    const int n = B->n, ni = n - 1;
    if (n <= SB_SMALL) return hipErrorInvalidValue;
    hipError_t e;
    const int gp = (n + SB_BLOCK - 1) / SB_BLOCK;
    if ((e = hipMemsetAsync(B->counters, 0, 4 * sizeof(int), stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(sb_prims_kernel, dim3(gp), dim3(SB_BLOCK), 0, stream, B->verts, B->leaf, n, B->plo, B->phi, B->pct, B->idx[0], B->seg[0]);
    int seed[3] = { 0, n, 0 };
    if ((e = hipMemcpyAsync(B->sb[0], &seed[0], sizeof(int), hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(B->se[0], &seed[1], sizeof(int), hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(B->snode[0], &seed[2], sizeof(int), hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    int nseg = 1, cur = 0, level = 1;
    size_t scan_bytes = B->scan_bytes;
    while (nseg > 0) {
        if (level > 62) return hipErrorInvalidValue;
        const int gs = (nseg + SB_BLOCK - 1) / SB_BLOCK;
        // bins per axis at this level: as many as the budget allows for this many segments
        int nb = SB_MINBINS;
        const size_t words = mpt_sah_level_words((size_t)nseg, &nb);
        if (words > B->seg_words) return hipErrorOutOfMemory;     // (cannot happen with a workspace of mpt_sah_seg_words(n))
        hipLaunchKernelGGL(sb_reset_kernel, dim3((unsigned)((words + SB_BLOCK - 1) / SB_BLOCK)), dim3(SB_BLOCK), 0, stream, nseg, nb, B->segw);
        hipLaunchKernelGGL(sb_bounds_kernel, dim3(gp), dim3(SB_BLOCK), 0, stream, n, B->idx[cur], B->seg[cur], B->pct, B->plo, B->phi, nb, B->segw);
        hipLaunchKernelGGL(sb_bin_kernel, dim3(gp), dim3(SB_BLOCK), 0, stream, n, B->idx[cur], B->seg[cur], B->pct, B->plo, B->phi, nb, B->segw);
        if (nb > 64)
            hipLaunchKernelGGL(sb_choose_wave_kernel, dim3(nseg), dim3(64), 0, stream, nseg, B->sb[cur], B->se[cur], B->snode[cur], nb, B->segw,
                               level, B->child, B->blo, B->bhi, B->dec, B->flag, B->small, B->counters);
        else
            hipLaunchKernelGGL(sb_choose_kernel, dim3(gs), dim3(SB_BLOCK), 0, stream, nseg, B->sb[cur], B->se[cur], B->snode[cur], nb, B->segw,
                               level, B->child, B->blo, B->bhi, B->dec, B->flag, B->small, B->counters);
        if ((e = rocprim::exclusive_scan(B->scan_tmp, scan_bytes, B->flag, B->foff, 0, (size_t)(2 * nseg), rocprim::plus<int>(), stream)) != hipSuccess) return e;
        int last[2] = { 0, 0 };
        if ((e = hipMemcpyAsync(&last[0], B->foff + (2 * nseg - 1), sizeof(int), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(&last[1], B->flag + (2 * nseg - 1), sizeof(int), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
        hipLaunchKernelGGL(sb_newseg_kernel, dim3(gs), dim3(SB_BLOCK), 0, stream, nseg, B->sb[cur], B->se[cur], B->snode[cur], B->dec, B->flag,
                           B->foff, B->sb[cur ^ 1], B->se[cur ^ 1], B->snode[cur ^ 1]);
        hipLaunchKernelGGL(sb_pred_kernel, dim3(gp), dim3(SB_BLOCK), 0, stream, n, B->idx[cur], B->seg[cur], B->pct, nb, B->segw, B->dec, B->pred);
        if ((e = rocprim::exclusive_scan(B->scan_tmp, scan_bytes, B->pred, B->pscan, 0, (size_t)n, rocprim::plus<int>(), stream)) != hipSuccess) return e;
        hipLaunchKernelGGL(sb_scatter_kernel, dim3(gp), dim3(SB_BLOCK), 0, stream, n, B->idx[cur], B->seg[cur], B->pred, B->pscan, B->sb[cur],
                           B->se[cur], B->snode[cur], B->dec, B->flag, B->foff, B->child, B->idx[cur ^ 1], B->seg[cur ^ 1]);
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
        nseg = last[0] + last[1];
        if ((size_t)nseg > mpt_sah_seg_capacity(n)) return hipErrorInvalidValue;
        cur ^= 1; level++;
    }
    int cnt[4] = { 0, 0, 0, 0 };
    if ((e = hipMemcpyAsync(cnt, B->counters, sizeof cnt, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    if (cnt[0] > 0)
        hipLaunchKernelGGL(sb_small_kernel, dim3((cnt[0] + 63) / 64), dim3(64), 0, stream, cnt[0], B->small, B->idx[cur], B->pct, B->plo, B->phi,
                           B->child, B->blo, B->bhi, B->counters);
    hipLaunchKernelGGL(sb_pack_kernel, dim3((ni + SB_BLOCK - 1) / SB_BLOCK), dim3(SB_BLOCK), 0, stream, ni, B->child, B->blo, B->bhi, B->plo,
                       B->phi, B->fnode);
    if ((e = hipMemcpyAsync(cnt, B->counters, sizeof cnt, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    *depth = cnt[1];
    return hipGetLastError();
}
