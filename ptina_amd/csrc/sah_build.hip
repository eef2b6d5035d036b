// sah_build.hip -- the SAH re-partition of the fast build's leaves, on the device (round 6: rewritten).
//
// The production traversal walks a better tree than the reference's LBVH over the SAME leaf slots (the image does not
// depend on the tree's shape, only on which of two equal-depth hits wins).  Same cost function and tie rules as the host
// pass (tree_build.cpp SahBuild): area x count on both sides, lowest cost, then lowest axis, then lowest split.
//
// Round 3-5 ran every level of the tree as nine launches over all n positions with global atomics per triangle and
// finished the <= 32-triangle ranges one lane each: 31 ms at a million triangles (profiles/r06_build_kernel_stats_c5_before.csv:
// bin 10.9 ms, small 8.5 ms, bounds 7.5 ms).  Now:
//
//   TOP PHASE, level by level, only for the ranges ("segments") of more than SB_K = 512 triangles.  The triangles' 32-byte
//   records {box, slot} themselves are moved (ping-pong), so every pass streams.  A level's segments are cut into chunks of
//   2048 ... 16384 positions, one workgroup each:
//     bin      a chunk's bins (3 axes x nb x {count, box}) in LDS, written out once, no global atomics; nb = 32 ... 1024 per
//              axis, as many as a budget divided by the level's segment count allows (bins matter near the root)
//     choose   one workgroup per segment: sums its chunks' bins, one wave per axis scans them for the cheapest split; the
//              node's box goes into its PARENT's record (a node record holds its two children's boxes), the chunks' left
//              counts follow from the chunks' own bin counts (no counting pass), children of <= 512 triangles become tasks
//     plan     one workgroup: next level's segment table and chunk list (two prefix sums)
//     scatter  stable partition of every chunk's records into the other buffer; single triangles write themselves into
//              their parent's record; the centre bounds of the children that go on as segments are reduced on the way
//   The plan kernel leaves the next level's grid sizes in a pinned host mailbox the host polls (no copy, no stream wait per level).
//
//   FINISH: one wave per task (<= 512 triangles), everything in 29 KB of LDS: the task's triangles are sorted once on each
//   axis by (centre, slot); then level by level over all of the task's ranges at once: segmented suffix / prefix scans of
//   the boxes along each sorted order give every split of every axis its exact cost (the host pass's exact sweep), a
//   segmented minimum picks each range's split, and a stable partition of all three orders keeps them sorted inside the
//   children.  For n <= 512 the whole tree is this kernel's and is the host pass's tree node for node (tests).
//
// Node numbers follow from range sizes (DFS pre-order: left child me + 1, right child me + left size): nothing is shared
// between ranges.  Deterministic: every decision is a function of sums of integers and min / max of floats, partitions are
// stable, ties go to the lowest slot.  This file is compiled with -ffp-contract=off.

#include <atomic>
#include <chrono>
#include <cstring>
#include <hip/hip_runtime.h>
#include <cmath>
#include "mpt_types.h"

#define SB_BLOCK 256
#ifndef SB_K
#define SB_K 512               // a range of at most this many triangles is finished by one wave in LDS (exact sweep)
#endif
#ifndef SB_K2
#define SB_K2 1024             // ... and one of up to this many too, by the same kernel with twice the positions per lane: exact sweeps up to
#endif                         // 1024 triangles give BASELINE config 4's grid mesh 8.56 node steps per ray, up to 512 give 8.72 (the host pass: 8.55)
#ifndef SB_MINBINS
#define SB_MINBINS 32          // bins per axis at the levels with many segments ...
#endif
#ifndef SB_MAXBINS
#define SB_MAXBINS 1024        // ... and near the root: as many as SB_BIN_BUDGET / (2 x segments) allows
#endif
#ifndef SB_BIN_BUDGET
#define SB_BIN_BUDGET (1 << 17)
#endif
#define SB_CHUNK_MIN 2048      // positions per chunk (workgroup): max(SB_CHUNK_MIN, 16 x bins), so a chunk's bins stay below its records' bytes
#define SB_SEG_INTS 16         // segment record: b, e, node, parent * 2 + side, centre bounds [6] (ordered ints), first chunk, chunks
#define SB_DEC_INTS 8          // decision record: axis (-1: halve the range), first bin of the right side, m, child segments [2] (-1: none)
#define SB_CHUNK_WORDS(nb) (24 * (nb))  // a chunk's bins: [3][nb]{count, lo3, hi3}, then the running bin counts [3][nb]
#define SB_TASK_INTS 8         // task record: b, e, node, parent * 2 + side, level, buffer

enum { SEG_B = 0, SEG_E, SEG_NODE, SEG_PAR, SEG_CB, SEG_CHUNK0 = 10, SEG_NCHUNK = 11 };
enum { DEC_AXIS = 0, DEC_Q, DEC_M, DEC_CHILD0, DEC_CHILD1 };
enum { META_NSEG = 0, META_NCHUNK, META_NB, META_CH, META_NTASK, META_DEPTH, META_BAD, META_NTASK2, META_ELEMS,
       META_T_SORT, META_T_LOOP, META_T_MAX, META_T_LEVELS, META_T_MAXLEVELS, META_INTS = 16 };   // META_T_*: the finish kernels' clocks (units of 1024 shader cycles) and levels

__host__ __device__ __forceinline__ int sb_bins_for(long long nseg) {
    int nb = SB_MINBINS;
    while (nb < SB_MAXBINS && nseg * 2 * nb <= (long long)SB_BIN_BUDGET) nb *= 2;
    return nb;
}
// positions per chunk (workgroup) at a level that streams `elems` positions with nb bins: 16 x bins, so that a chunk's bins stay
// well below its records' bytes -- but no more than 1/256 of the level (a level wants a few hundred workgroups: with 16 384
// positions per chunk the first levels of a 100 000-triangle model ran on 7 CUs), and never less than SB_CHUNK_MIN
__host__ __device__ __forceinline__ int sb_chunk_for(int nb, long long elems) {
    long long ch = 16 * nb;
    const long long spread = ((elems / 256 + 255) / 256) * 256;
    if (ch > spread) ch = spread;
    return ch > SB_CHUNK_MIN ? (int)ch : SB_CHUNK_MIN;
}

__device__ __forceinline__ int sb_f2ord(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float sb_ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }
__device__ __forceinline__ float sb_scale(int nb, float cl, float ch) { return ch > cl ? nb / (ch - cl) : 0.f; }
__device__ __forceinline__ int sb_bin_of(float c, float cl, float scale, int nb) {
    return min(nb - 1, max(0, (int)((c - cl) * scale)));
}
__device__ __forceinline__ float sb_half_area(const float *l, const float *h) {
    const float dx = fmaxf(h[0] - l[0], 0.f), dy = fmaxf(h[1] - l[1], 0.f), dz = fmaxf(h[2] - l[2], 0.f);
    return dx * dy + dy * dz + dz * dx;
}
// child k (0 / 1) of node `par` has this box and id: the 64-byte record is {c0.lo.x, c1.lo.x, c0.hi.x, c1.hi.x} {y} {z} {id0, id1, 0, 0}
__device__ __forceinline__ void sb_write_child(MptVec4 *__restrict__ fnode, int par2, const float *l, const float *h, int id) {
    float *r = (float *)(fnode + (size_t)(par2 >> 1) * 4);
    const int k = par2 & 1;
    for (int a = 0; a < 3; a++) { r[a * 4 + k] = l[a]; r[a * 4 + 2 + k] = h[a]; }
    r[12 + k] = __int_as_float(id);
}

// ------------------------------------------------------------------ records of the leaf slots + the root's centre bounds
// verts [3n][8], leaf: slot -> face.  prim[slot] = {lo.xyz, slot} {hi.xyz, 0}.  The centre bounds: per workgroup into
// partial[block][6], and the workgroup that finishes last reduces those into the root segment's record (6 atomics per WAVE on
// one cache line were 0.27 of this kernel's 0.30 ms at a million triangles).  What crosses between workgroups is written and
// read with agent-scope atomic accesses (memory side, sc1) and ordered by the s_waitcnt in front of the ticket (lbvh_build.hip).
__global__ __launch_bounds__(SB_BLOCK) void sb_prims_kernel(const float *__restrict__ verts, const int *__restrict__ leaf, int n,
                                                           MptVec4 *__restrict__ prim, int *__restrict__ seg0, int *partial,
                                                           int *ticket) {
    __shared__ int red[SB_BLOCK / 64][6];
    __shared__ int is_last;
    int cb[6] = { 0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000 };
    for (int slot = blockIdx.x * SB_BLOCK + threadIdx.x; slot < n; slot += gridDim.x * SB_BLOCK) {
        const float *p0 = verts + (size_t)leaf[slot] * 24, *p1 = p0 + 8, *p2 = p0 + 16;
        float l[3], h[3];
        for (int a = 0; a < 3; a++) {
            l[a] = fminf(fminf(p0[a], p1[a]), p2[a]); h[a] = fmaxf(fmaxf(p0[a], p1[a]), p2[a]);
            const int c = sb_f2ord(0.5f * (l[a] + h[a]));
            cb[a] = min(cb[a], c); cb[3 + a] = max(cb[3 + a], c);
        }
        prim[(size_t)slot * 2 + 0] = { l[0], l[1], l[2], __int_as_float(slot) };
        prim[(size_t)slot * 2 + 1] = { h[0], h[1], h[2], 0.f };
    }
    for (int k = 0; k < 6; k++)
        for (int off = 32; off > 0; off >>= 1) {
            const int o = __shfl_xor(cb[k], off);
            cb[k] = k < 3 ? min(cb[k], o) : max(cb[k], o);
        }
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < 6; k++) red[threadIdx.x >> 6][k] = cb[k];
    __syncthreads();
    if (threadIdx.x < 6) {
        int v = red[0][threadIdx.x];
        for (int w = 1; w < SB_BLOCK / 64; w++) v = threadIdx.x < 3 ? min(v, red[w][threadIdx.x]) : max(v, red[w][threadIdx.x]);
        __hip_atomic_store(partial + blockIdx.x * 6 + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) is_last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __syncthreads();
    if (!is_last) return;
    for (int k = 0; k < 6; k++) cb[k] = k < 3 ? 0x7fffffff : (int)0x80000000;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += SB_BLOCK)
        for (int k = 0; k < 6; k++) {
            const int v = __hip_atomic_load(partial + b * 6 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cb[k] = k < 3 ? min(cb[k], v) : max(cb[k], v);
        }
    for (int k = 0; k < 6; k++)
        for (int off = 32; off > 0; off >>= 1) {
            const int o = __shfl_xor(cb[k], off);
            cb[k] = k < 3 ? min(cb[k], o) : max(cb[k], o);
        }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < 6; k++) red[threadIdx.x >> 6][k] = cb[k];
    __syncthreads();
    if (threadIdx.x < 6) {
        int v = red[0][threadIdx.x];
        for (int w = 1; w < SB_BLOCK / 64; w++) v = threadIdx.x < 3 ? min(v, red[w][threadIdx.x]) : max(v, red[w][threadIdx.x]);
        seg0[SEG_CB + threadIdx.x] = v;
    }
}

// ------------------------------------------------------------------ top phase: bin
// part[chunk][axis][bin]{count, lo3, hi3} (ordered ints); a bin nobody fell into keeps {0, MAX.., MIN..}
__global__ __launch_bounds__(SB_BLOCK) void sb_bin_kernel(const MptVec4 *__restrict__ prim, const int *__restrict__ seg,
                                                         const int *__restrict__ ch_seg, int nb, int CH, int *__restrict__ part) {
    extern __shared__ int lb[];
    const int j = blockIdx.x, s = ch_seg[j];
    const int *S = seg + (size_t)s * SB_SEG_INTS;
    const int b = S[SEG_B], e = S[SEG_E];
    const int start = b + (j - S[SEG_CHUNK0]) * CH, end = min(start + CH, e);
    float cl[3], scale[3];
    for (int a = 0; a < 3; a++) { cl[a] = sb_ord2f(S[SEG_CB + a]); scale[a] = sb_scale(nb, cl[a], sb_ord2f(S[SEG_CB + 3 + a])); }
    const int words = 21 * nb;
    for (int k = threadIdx.x; k < words; k += SB_BLOCK) {
        const int f = k % 7;
        lb[k] = f == 0 ? 0 : (f < 4 ? 0x7fffffff : (int)0x80000000);
    }
    __syncthreads();
    for (int i = start + threadIdx.x; i < end; i += SB_BLOCK) {
        const MptVec4 lo = prim[(size_t)i * 2], hi = prim[(size_t)i * 2 + 1];
        const float l[3] = { lo.x, lo.y, lo.z }, h[3] = { hi.x, hi.y, hi.z };
        int bl[3], bh[3];
        for (int a = 0; a < 3; a++) { bl[a] = sb_f2ord(l[a]); bh[a] = sb_f2ord(h[a]); }
        for (int a = 0; a < 3; a++) {
            const int q = sb_bin_of(0.5f * (l[a] + h[a]), cl[a], scale[a], nb);
            int *w = lb + (a * nb + q) * 7;
            atomicAdd(w, 1);
            for (int r = 0; r < 3; r++) { atomicMin(w + 1 + r, bl[r]); atomicMax(w + 4 + r, bh[r]); }
        }
    }
    __syncthreads();
    int *out = part + (size_t)j * SB_CHUNK_WORDS(nb);
    for (int k = threadIdx.x; k < words; k += SB_BLOCK) out[k] = lb[k];
    // the running counts along every axis (how many of the chunk's positions fall into bins 0 .. q): the choose kernel reads ONE
    // of them per chunk to know how many of the chunk go left of its split
    __shared__ int wtot[SB_BLOCK / 64];
    const int per = nb >= SB_BLOCK ? nb / SB_BLOCK : 1, q0 = threadIdx.x * per, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int a = 0; a < 3; a++) {
        int own = 0;
        if (q0 < nb) for (int t = 0; t < per; t++) own += lb[(a * nb + q0 + t) * 7];
        int inc = own;
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
        __syncthreads();
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        int run = inc - own;
        for (int w = 0; w < wave; w++) run += wtot[w];
        if (q0 < nb)
            for (int t = 0; t < per; t++) { run += lb[(a * nb + q0 + t) * 7]; out[words + a * nb + q0 + t] = run; }
    }
}

// ------------------------------------------------------------------ top phase: choose
struct SbAgg { int c; float l[3], h[3]; };
__device__ __forceinline__ void sb_agg_clear(SbAgg &a) { a.c = 0; for (int r = 0; r < 3; r++) { a.l[r] = INFINITY; a.h[r] = -INFINITY; } }
__device__ __forceinline__ void sb_agg_add(SbAgg &a, const SbAgg &b) {
    a.c += b.c;
    for (int r = 0; r < 3; r++) { a.l[r] = fminf(a.l[r], b.l[r]); a.h[r] = fmaxf(a.h[r], b.h[r]); }
}
__device__ __forceinline__ void sb_agg_add_bin(SbAgg &a, const int *w) {      // w: {count, lo3, hi3}, ignored when empty
    if (w[0] == 0) return;
    a.c += w[0];
    for (int r = 0; r < 3; r++) { a.l[r] = fminf(a.l[r], sb_ord2f(w[1 + r])); a.h[r] = fmaxf(a.h[r], sb_ord2f(w[4 + r])); }
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

// one wave, one axis: the cheapest split of nb bins (in LDS at w0, 7 words each).  Lane l owns `per` consecutive bins; the
// boxes and counts of the lanes before / after it come from wave scans; the lanes' best splits are reduced with the serial
// loop's tie rule (lowest cost, then lowest bin).  Returns in every lane: cost (INFINITY: no split), k (left count), q.
__device__ __forceinline__ void sb_axis_best(const int *w0, int nb, int lane, float &cost_out, int &k_out, int &q_out) {
    const int per = nb >= 64 ? nb >> 6 : 1;           // 1 ... 16 bins per lane
    const bool live = lane * per < nb;
    const int q0 = lane * per;
    SbAgg own; sb_agg_clear(own);
    if (live)
#pragma unroll
        for (int t = 0; t < 16; t++) if (t < per) sb_agg_add_bin(own, w0 + (q0 + t) * 7);
    SbAgg pre = own, suf = own;
    for (int d = 1; d < 64; d <<= 1) {
        SbAgg o = sb_agg_shfl_up(pre, d);
        if (lane >= d) sb_agg_add(pre, o);
        SbAgg u = sb_agg_shfl_down(suf, d);
        if (lane + d < 64) sb_agg_add(suf, u);
    }
    SbAgg left = sb_agg_shfl_up(pre, 1), right = sb_agg_shfl_down(suf, 1);
    if (lane == 0) sb_agg_clear(left);
    if (lane == 63) sb_agg_clear(right);
    // the right side of a split at the lane's bin t = its bins t .. per-1 + the lanes after it
    float rar[16]; int rcn[16];
    {
        SbAgg acc = right;
#pragma unroll
        for (int t = 15; t >= 0; t--) {
            rar[t] = 0.f; rcn[t] = 0;
            if (t < per && live) {
                sb_agg_add_bin(acc, w0 + (q0 + t) * 7);
                rar[t] = sb_half_area(acc.l, acc.h); rcn[t] = acc.c;
            }
        }
    }
    float lbest = INFINITY; int lk = -1, lq = 0;
    {
        SbAgg acc = left;
#pragma unroll
        for (int t = 0; t < 16; t++) {
            if (t < per && live) {
                const int q = q0 + t;
                if (q >= 1 && acc.c != 0 && rcn[t] != 0) {
                    const float cost = sb_half_area(acc.l, acc.h) * acc.c + rar[t] * rcn[t];
                    if (cost < lbest) { lbest = cost; lk = acc.c; lq = q; }
                }
                sb_agg_add_bin(acc, w0 + q * 7);
            }
        }
    }
    for (int d = 32; d > 0; d >>= 1) {
        const float oc = __shfl_xor(lbest, d); const int ok = __shfl_xor(lk, d), oq = __shfl_xor(lq, d);
        if (ok >= 0 && (lk < 0 || oc < lbest || (oc == lbest && oq < lq))) { lbest = oc; lk = ok; lq = oq; }
    }
    cost_out = lk >= 0 ? lbest : INFINITY; k_out = lk; q_out = lq;
}

// a segment's bins = the sum / min / max of its chunks' bins, one lane per word (a workgroup per segment doing this alone took
// 250 us per level near the root, where one segment has 60 chunks of 21 504 words).  Segments of one chunk are skipped: the
// choose kernel reads that chunk's bins directly.
__global__ __launch_bounds__(SB_BLOCK) void sb_reduce_kernel(const int *__restrict__ seg, int nb, const int *__restrict__ part,
                                                            int *__restrict__ segbins) {
    const int s = blockIdx.y, words = 21 * nb, k = blockIdx.x * SB_BLOCK + threadIdx.x;
    const int *S = seg + (size_t)s * SB_SEG_INTS;
    const int chunk0 = S[SEG_CHUNK0], nchunk = S[SEG_NCHUNK];
    if (nchunk < 2 || k >= words) return;
    const int f = k % 7;
    const size_t stride = SB_CHUNK_WORDS(nb);
    const int *p = part + (size_t)chunk0 * stride + k;
    int acc = p[0];
    for (int c = 1; c < nchunk; c++) {
        const int v = p[(size_t)c * stride];
        acc = f == 0 ? acc + v : (f < 4 ? min(acc, v) : max(acc, v));
    }
    segbins[(size_t)s * words + k] = acc;
}

__global__ __launch_bounds__(SB_BLOCK) void sb_choose_kernel(const int *__restrict__ seg, int nb, int CH, const int *__restrict__ part,
                                                            const int *__restrict__ segbins, int level, int nextbuf, int *__restrict__ dec,
                                                            int *__restrict__ ch_left, int *__restrict__ tasks, int task_cap, int *__restrict__ meta,
                                                            MptVec4 *__restrict__ fnode) {
    extern __shared__ int lb[];                       // the segment's bins [3][nb][7], then (same words) the chunks' left counts
    __shared__ float w_cost[3];
    __shared__ int w_k[3], w_q[3], w_dec[3];          // w_dec: axis, q, m
    const int s = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int *S = seg + (size_t)s * SB_SEG_INTS;
    const int b = S[SEG_B], e = S[SEG_E], me = S[SEG_NODE], par = S[SEG_PAR], chunk0 = S[SEG_CHUNK0], nchunk = S[SEG_NCHUNK];
    const int words = 21 * nb;
    {
        const int *src = nchunk == 1 ? part + (size_t)chunk0 * SB_CHUNK_WORDS(nb) : segbins + (size_t)s * words;
        for (int k = tid; k < words; k += SB_BLOCK) lb[k] = src[k];
    }
    __syncthreads();
    if (wave < 3) {
        float cost; int k, q;
        sb_axis_best(lb + wave * nb * 7, nb, lane, cost, k, q);
        if (lane == 0) { w_cost[wave] = cost; w_k[wave] = k; w_q[wave] = q; }
    } else if (par >= 0) {
        // the node's own box (every triangle is in exactly one bin of axis 0) goes into its parent's record
        SbAgg own; sb_agg_clear(own);
        for (int q = lane; q < nb; q += 64) sb_agg_add_bin(own, lb + q * 7);
        for (int d = 32; d > 0; d >>= 1)
            for (int r = 0; r < 3; r++) { own.l[r] = fminf(own.l[r], __shfl_xor(own.l[r], d)); own.h[r] = fmaxf(own.h[r], __shfl_xor(own.h[r], d)); }
        if (lane == 0) sb_write_child(fnode, par, own.l, own.h, me);
    }
    __syncthreads();
    const int cnt = e - b;
    if (tid == 0) {
        float best = INFINITY; int axis = -1, bk = 0, bq = 0;
        for (int a = 0; a < 3; a++)
            if (w_k[a] >= 0 && w_cost[a] < best) { best = w_cost[a]; axis = a; bk = w_k[a]; bq = w_q[a]; }
        const int m = axis < 0 ? b + cnt / 2 : b + bk;          // all centres equal (or no finite cost): halve the range
        w_dec[0] = axis; w_dec[1] = bq; w_dec[2] = m;
        int *D = dec + (size_t)s * SB_DEC_INTS;
        D[DEC_AXIS] = axis; D[DEC_Q] = bq; D[DEC_M] = m;
        const int node[2] = { me + 1, me + (m - b) }, lo_[2] = { b, m }, hi_[2] = { m, e };
        for (int k = 0; k < 2; k++) {
            const int sz = hi_[k] - lo_[k];
            int flag = 0;
            if (sz > SB_K2) flag = 1;                            // a segment of the next level (numbered by the plan kernel)
            else if (sz >= 2) {
                // a range the finish kernels take: the small ones are filed from the front of the task array, the big ones from its end
                int *T = sz <= SB_K ? tasks + (size_t)atomicAdd(meta + META_NTASK, 1) * SB_TASK_INTS
                                    : tasks + ((ptrdiff_t)task_cap - 1 - atomicAdd(meta + META_NTASK2, 1)) * SB_TASK_INTS;
                T[0] = lo_[k]; T[1] = hi_[k]; T[2] = node[k]; T[3] = me * 2 + k; T[4] = level + 1; T[5] = nextbuf; T[6] = 0; T[7] = 0;
            }
            D[DEC_CHILD0 + k] = flag;
        }
        // ids of the children that are nodes; a child that is a single triangle writes ~slot over its id in the scatter pass
        fnode[(size_t)me * 4 + 3] = { __int_as_float(node[0]), __int_as_float(node[1]), 0.f, 0.f };
        atomicMax(meta + META_DEPTH, level);
    }
    __syncthreads();
    // how many of every chunk's positions go left: from the chunk's own bin counts; ch_left = their exclusive prefix sum
    const int axis = w_dec[0], bq = w_dec[1], m = w_dec[2];
    int *lc = lb;                                                // (the bins are done with)
    const bool in_lds = nchunk <= words;
    __syncthreads();
    for (int c = tid; c < nchunk; c += SB_BLOCK) {
        const int cstart = b + c * CH, ccnt = min(CH, e - cstart);
        int left = 0;
        if (axis < 0) left = min(max(m - cstart, 0), ccnt);
        else if (bq > 0) left = part[(size_t)(chunk0 + c) * SB_CHUNK_WORDS(nb) + words + axis * nb + bq - 1];
        if (in_lds) lc[c] = left; else ch_left[chunk0 + c] = left;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int c = 0; c < nchunk; c++) {
            const int v = in_lds ? lc[c] : ch_left[chunk0 + c];
            ch_left[chunk0 + c] = run; run += v;
        }
        if (run != m - b) atomicOr(meta + META_BAD, 1);          // (cannot happen: the bins and the split count the same triangles)
    }
}

// ------------------------------------------------------------------ top phase: plan
// exclusive prefix sum over the 1024 threads of the (only) workgroup; returns the total in every thread
__device__ __forceinline__ int sb_block_scan_1024(int v, int *total, int *wsum /* [16] LDS */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
    __syncthreads();
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < 16; w++) { const int x = wsum[w]; if (w < wave) base += x; tot += x; }
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(1024) void sb_plan_kernel(int nseg, const int *__restrict__ seg, int *__restrict__ dec,
                                                      int *__restrict__ seg_next, int *__restrict__ ch_seg, int *__restrict__ meta,
                                                      int *mail, int seq) {
    __shared__ int wsum[16];
    __shared__ int sh_nb, sh_ch, sh_elems;
    const int tid = threadIdx.x;
    if (tid == 0) sh_elems = 0;
    // 1. number the children that go on as segments: parents in order, left before right
    int carry = 0;
    for (int base = 0; base < nseg; base += 1024) {
        const int s = base + tid;
        int f0 = 0, f1 = 0;
        if (s < nseg) { f0 = dec[(size_t)s * SB_DEC_INTS + DEC_CHILD0]; f1 = dec[(size_t)s * SB_DEC_INTS + DEC_CHILD1]; }
        int tot;
        const int at = carry + sb_block_scan_1024(f0 + f1, &tot, wsum);
        if (s < nseg) {
            const int *S = seg + (size_t)s * SB_SEG_INTS;
            const int b = S[SEG_B], e = S[SEG_E], me = S[SEG_NODE], m = dec[(size_t)s * SB_DEC_INTS + DEC_M];
            dec[(size_t)s * SB_DEC_INTS + DEC_CHILD0] = f0 ? at : -1;
            dec[(size_t)s * SB_DEC_INTS + DEC_CHILD1] = f1 ? at + f0 : -1;
            for (int k = 0; k < 2; k++) {
                if (!(k ? f1 : f0)) continue;
                int *N = seg_next + (size_t)(at + (k ? f0 : 0)) * SB_SEG_INTS;
                N[SEG_B] = k ? m : b; N[SEG_E] = k ? e : m; N[SEG_NODE] = k ? me + (m - b) : me + 1; N[SEG_PAR] = me * 2 + k;
                for (int r = 0; r < 6; r++) N[SEG_CB + r] = r < 3 ? 0x7fffffff : (int)0x80000000;
                atomicAdd(&sh_elems, N[SEG_E] - N[SEG_B]);         // positions the next level streams
            }
        }
        carry += tot;
        __syncthreads();
    }
    const int nnext = carry;
    if (tid == 0) { sh_nb = sb_bins_for(nnext); sh_ch = sb_chunk_for(sh_nb, sh_elems); }
    __syncthreads();
    const int CH = sh_ch;
    // 2. chunks of every new segment
    carry = 0;
    for (int base = 0; base < nnext; base += 1024) {
        const int s = base + tid;
        int nch = 0;
        if (s < nnext) {
            const int sz = seg_next[(size_t)s * SB_SEG_INTS + SEG_E] - seg_next[(size_t)s * SB_SEG_INTS + SEG_B];
            nch = (sz + CH - 1) / CH;
        }
        int tot;
        const int at = carry + sb_block_scan_1024(nch, &tot, wsum);
        if (s < nnext) { seg_next[(size_t)s * SB_SEG_INTS + SEG_CHUNK0] = at; seg_next[(size_t)s * SB_SEG_INTS + SEG_NCHUNK] = nch; }
        carry += tot;
        __syncthreads();
    }
    const int nchunks = carry;
    __threadfence_block();
    __syncthreads();
    // 3. chunk -> segment: the last segment whose first chunk is not after it
    for (int j = tid; j < nchunks; j += 1024) {
        int lo = 0, hi = nnext - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (seg_next[(size_t)mid * SB_SEG_INTS + SEG_CHUNK0] <= j) lo = mid; else hi = mid - 1;
        }
        ch_seg[j] = lo;
    }
    if (tid == 0) {
        meta[META_NSEG] = nnext; meta[META_NCHUNK] = nchunks; meta[META_NB] = sh_nb; meta[META_CH] = CH; meta[META_ELEMS] = sh_elems;
        if (mail) {
            // the level's outcome straight into host memory (pinned, mapped), the sequence word last: the host polls that word instead of
            // enqueueing a copy and waiting for the stream (25 us per level).  What the scatter pass still has to do does not matter to
            // the host: the next level's launches go behind it on the stream
            const int v[8] = { nnext, nchunks, sh_nb, CH, sh_elems, meta[META_NTASK], meta[META_NTASK2], meta[META_BAD] };
            for (int k = 0; k < 8; k++) __hip_atomic_store(mail + k, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(mail + 8, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ------------------------------------------------------------------ top phase: scatter
__global__ __launch_bounds__(SB_BLOCK) void sb_scatter_kernel(const MptVec4 *__restrict__ src, MptVec4 *__restrict__ dst,
                                                             const int *__restrict__ seg, const int *__restrict__ ch_seg,
                                                             const int *__restrict__ dec, const int *__restrict__ ch_left, int nb, int CH,
                                                             int *__restrict__ seg_next, MptVec4 *__restrict__ fnode) {
    __shared__ int wtot[2][SB_BLOCK / 64];
    const int j = blockIdx.x, s = ch_seg[j], tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int *S = seg + (size_t)s * SB_SEG_INTS, *D = dec + (size_t)s * SB_DEC_INTS;
    const int b = S[SEG_B], e = S[SEG_E], me = S[SEG_NODE];
    const int axis = D[DEC_AXIS], bq = D[DEC_Q], m = D[DEC_M];
    const int child[2] = { D[DEC_CHILD0], D[DEC_CHILD1] }, sz[2] = { m - b, e - m };
    const int start = b + (j - S[SEG_CHUNK0]) * CH, end = min(start + CH, e);
    float cl = 0.f, scale = 0.f;
    if (axis >= 0) { cl = sb_ord2f(S[SEG_CB + axis]); scale = sb_scale(nb, cl, sb_ord2f(S[SEG_CB + 3 + axis])); }
    const int lbase = ch_left[j];
    int run = 0, it = 0;
    int cb[2][6];
    for (int k = 0; k < 2; k++) for (int r = 0; r < 6; r++) cb[k][r] = r < 3 ? 0x7fffffff : (int)0x80000000;
    for (int t0 = start; t0 < end; t0 += SB_BLOCK, it ^= 1) {
        const int i = t0 + tid;
        const bool valid = i < end;
        MptVec4 lo = { 0.f, 0.f, 0.f, 0.f }, hi = lo;
        bool left = false;
        if (valid) {
            lo = src[(size_t)i * 2]; hi = src[(size_t)i * 2 + 1];
            if (axis < 0) left = i < m;
            else {
                const float c = axis == 0 ? 0.5f * (lo.x + hi.x) : (axis == 1 ? 0.5f * (lo.y + hi.y) : 0.5f * (lo.z + hi.z));
                left = sb_bin_of(c, cl, scale, nb) < bq;
            }
        }
        const unsigned long long bal = __ballot(left);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wtot[it][wave] = __popcll(bal);
        __syncthreads();
        int wbase = 0, tile = 0;
        for (int w = 0; w < SB_BLOCK / 64; w++) { const int x = wtot[it][w]; if (w < wave) wbase += x; tile += x; }
        if (valid) {
            const int lr = lbase + run + wbase + before;           // positions of this segment before i that go left
            const int k = left ? 0 : 1;
            const int dest = left ? b + lr : m + (i - b) - lr;
            dst[(size_t)dest * 2] = lo; dst[(size_t)dest * 2 + 1] = hi;
            const float l[3] = { lo.x, lo.y, lo.z }, h[3] = { hi.x, hi.y, hi.z };
            if (sz[k] == 1) sb_write_child(fnode, me * 2 + k, l, h, ~__float_as_int(lo.w));
            else if (child[k] >= 0)
                for (int a = 0; a < 3; a++) {
                    const int c = sb_f2ord(0.5f * (l[a] + h[a]));
                    cb[k][a] = min(cb[k][a], c); cb[k][3 + a] = max(cb[k][3 + a], c);
                }
        }
        run += tile;
    }
    // centre bounds of the children that go on as segments (a wave's lanes, then one atomic per word and wave)
    for (int k = 0; k < 2; k++) {
        if (child[k] < 0) continue;                                // (uniform over the workgroup)
        for (int r = 0; r < 6; r++)
            for (int off = 32; off > 0; off >>= 1) {
                const int o = __shfl_xor(cb[k][r], off);
                cb[k][r] = r < 3 ? min(cb[k][r], o) : max(cb[k][r], o);
            }
        if (lane == 0) {
            int *N = seg_next + (size_t)child[k] * SB_SEG_INTS + SEG_CB;
            for (int r = 0; r < 6; r++) {
                if (r < 3) { if (cb[k][r] != 0x7fffffff) atomicMin(N + r, cb[k][r]); }
                else if (cb[k][r] != (int)0x80000000) atomicMax(N + r, cb[k][r]);
            }
        }
    }
}

// ------------------------------------------------------------------ finish: one workgroup of four waves per task, exact sweep in LDS
#define SF_W 4                 // waves per task (one wave per task was latency-bound: every step of a level hangs on LDS round trips)
struct SfBox { float l[3], h[3]; };
__device__ __forceinline__ void sf_clear(SfBox &b) { for (int r = 0; r < 3; r++) { b.l[r] = INFINITY; b.h[r] = -INFINITY; } }
__device__ __forceinline__ void sf_add(SfBox &a, const SfBox &b) {
    for (int r = 0; r < 3; r++) { a.l[r] = fminf(a.l[r], b.l[r]); a.h[r] = fmaxf(a.h[r], b.h[r]); }
}
__device__ __forceinline__ SfBox sf_shfl_up(const SfBox &a, int d) {
    SfBox o; for (int r = 0; r < 3; r++) { o.l[r] = __shfl_up(a.l[r], d); o.h[r] = __shfl_up(a.h[r], d); } return o;
}
__device__ __forceinline__ SfBox sf_shfl_down(const SfBox &a, int d) {
    SfBox o; for (int r = 0; r < 3; r++) { o.l[r] = __shfl_down(a.l[r], d); o.h[r] = __shfl_down(a.h[r], d); } return o;
}
// what the waves hand each other in a scan: per wave a flag and up to six words
struct SfCross { int f[SF_W]; float v[SF_W][6]; };

// Segmented scans over the workgroup's lanes.  A lane's aggregate is (f, v): v = the union over its positions from its near
// end up to the first range boundary inside it, f = there is such a boundary.  The carry a lane receives is what its
// positions in front of its first boundary still miss: the aggregates of the lanes before it, walking away from it until
// one with a boundary has been taken -- first inside its wave (shuffles), then over the waves before (LDS, one barrier).
// The in-wave part runs on DPP moves and v_readlane (VALU), not on ds_bpermute: the finish kernel is bound by the CU's LDS pipe, and
// with shuffles three quarters of what it put through that pipe were the scans' 42 + 7 ds_bpermute each (round 6: a task of 1024
// triangles took 0.35 ms; the LDS pipe is shared by every wave of the CU).  A Kogge-Stone scan inside each row of 16 lanes (row_shr /
// row_shl: 1, 2, 4, 8; a lane without a source gets the identity), then the rows take the aggregates of the rows before them (the
// row's last / first lane, read with v_readlane), then a one-lane shift of the whole wave makes the result exclusive.
template <int CTRL> __device__ __forceinline__ int sf_dpp(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ float sf_dpp(float old, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ int sf_rl(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ float sf_rl(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
// what lane l receives in the six steps of a scan towards the HIGHER lanes (UP: from l - 1, 2, 4, 8 inside its row; rows 1, 3 from
// lane 15 / 47; rows 2, 3 from lane 31) and towards the LOWER ones (DN: from l + 1, 2, 4, 8; rows 0, 2 from lane 16 / 48; rows 0, 1
// from lane 32); I = the identity a lane without a source gets
template <int STEP, bool UP, class T> __device__ __forceinline__ T sf_from(T I, T x, int row) {
    if constexpr (STEP == 0) return UP ? sf_dpp<0x111>(I, x) : sf_dpp<0x101>(I, x);
    else if constexpr (STEP == 1) return UP ? sf_dpp<0x112>(I, x) : sf_dpp<0x102>(I, x);
    else if constexpr (STEP == 2) return UP ? sf_dpp<0x114>(I, x) : sf_dpp<0x104>(I, x);
    else if constexpr (STEP == 3) return UP ? sf_dpp<0x118>(I, x) : sf_dpp<0x108>(I, x);
    else if constexpr (STEP == 4) {
        const T a = sf_rl(x, UP ? 15 : 16), b = sf_rl(x, UP ? 47 : 48);
        return UP ? (row == 1 ? a : (row == 3 ? b : I)) : (row == 0 ? a : (row == 2 ? b : I));
    } else {
        const T a = sf_rl(x, UP ? 31 : 32);
        return UP ? (row >= 2 ? a : I) : (row <= 1 ? a : I);
    }
}
// the neighbour's value across the whole wave (exclusive result): lane l from l - 1 (UP; lane 0 gets I) or l + 1 (lane 63 gets I)
template <bool UP, class T> __device__ __forceinline__ T sf_next(T I, T x) { return UP ? sf_dpp<0x138>(I, x) : sf_dpp<0x130>(I, x); }

template <int STEP, bool UP> __device__ __forceinline__ void sf_box_step(int &f, SfBox &v, int row) {
    SfBox ov;
    for (int r = 0; r < 3; r++) { ov.l[r] = sf_from<STEP, UP>(INFINITY, v.l[r], row); ov.h[r] = sf_from<STEP, UP>(-INFINITY, v.h[r], row); }
    const int of = sf_from<STEP, UP>(0, f, row);
    if (!f) { sf_add(v, ov); f = of; }
}
template <bool UP> __device__ __forceinline__ void sf_box_scan(int &f, SfBox &v, int lane) {     // inclusive, in the wave
    const int row = lane >> 4;
    sf_box_step<0, UP>(f, v, row); sf_box_step<1, UP>(f, v, row); sf_box_step<2, UP>(f, v, row);
    sf_box_step<3, UP>(f, v, row); sf_box_step<4, UP>(f, v, row); sf_box_step<5, UP>(f, v, row);
}

__device__ __forceinline__ SfBox sf_carry_from_below(bool fb, SfBox v, int lane, int wave, SfCross &X) {      // prefix direction
    int f = fb ? 1 : 0;
    sf_box_scan<true>(f, v, lane);
    if (lane == 63) { X.f[wave] = f; for (int r = 0; r < 3; r++) { X.v[wave][r] = v.l[r]; X.v[wave][3 + r] = v.h[r]; } }
    SfBox c;
    for (int r = 0; r < 3; r++) { c.l[r] = sf_next<true>(INFINITY, v.l[r]); c.h[r] = sf_next<true>(-INFINITY, v.h[r]); }
    const int cf = sf_next<true>(0, f);
    __syncthreads();
    if (!cf)
        for (int w = wave - 1; w >= 0; w--) {
            for (int r = 0; r < 3; r++) { c.l[r] = fminf(c.l[r], X.v[w][r]); c.h[r] = fmaxf(c.h[r], X.v[w][3 + r]); }
            if (X.f[w]) break;
        }
    return c;
}
__device__ __forceinline__ SfBox sf_carry_from_above(bool fb, SfBox v, int lane, int wave, SfCross &X) {      // suffix direction
    int f = fb ? 1 : 0;
    sf_box_scan<false>(f, v, lane);
    if (lane == 0) { X.f[wave] = f; for (int r = 0; r < 3; r++) { X.v[wave][r] = v.l[r]; X.v[wave][3 + r] = v.h[r]; } }
    SfBox c;
    for (int r = 0; r < 3; r++) { c.l[r] = sf_next<false>(INFINITY, v.l[r]); c.h[r] = sf_next<false>(-INFINITY, v.h[r]); }
    const int cf = sf_next<false>(0, f);
    __syncthreads();
    if (!cf)
        for (int w = wave + 1; w < SF_W; w++) {
            for (int r = 0; r < 3; r++) { c.l[r] = fminf(c.l[r], X.v[w][r]); c.h[r] = fmaxf(c.h[r], X.v[w][3 + r]); }
            if (X.f[w]) break;
        }
    return c;
}
// (cost, position) minimum, prefix direction; equal costs: the lower position (= the lanes / waves before) wins
template <int STEP> __device__ __forceinline__ void sf_min_step(int &f, float &vc, int &vp, int row) {
    const float oc = sf_from<STEP, true>(INFINITY, vc, row);
    const int op = sf_from<STEP, true>(-1, vp, row), of = sf_from<STEP, true>(0, f, row);
    if (!f) {
        if (op >= 0 && (vp < 0 || oc <= vc)) { vc = oc; vp = op; }
        f = of;
    }
}
__device__ __forceinline__ void sf_min_from_below(bool fb, float &vc, int &vp, int lane, int wave, SfCross &X) {
    int f = fb ? 1 : 0;
    const int row = lane >> 4;
    sf_min_step<0>(f, vc, vp, row); sf_min_step<1>(f, vc, vp, row); sf_min_step<2>(f, vc, vp, row);
    sf_min_step<3>(f, vc, vp, row); sf_min_step<4>(f, vc, vp, row); sf_min_step<5>(f, vc, vp, row);
    if (lane == 63) { X.f[wave] = f; X.v[wave][0] = vc; X.v[wave][1] = __int_as_float(vp); }
    float cc = sf_next<true>(INFINITY, vc); int cp = sf_next<true>(-1, vp);
    const int cf = sf_next<true>(0, f);
    __syncthreads();
    if (!cf)
        for (int w = wave - 1; w >= 0; w--) {
            const float oc = X.v[w][0]; const int op = __float_as_int(X.v[w][1]);
            if (op >= 0 && (cp < 0 || oc <= cc)) { cc = oc; cp = op; }
            if (X.f[w]) break;
        }
    vc = cc; vp = cp;
}
template <int STEP> __device__ __forceinline__ void sf_sum_step(int &f, int &v, int row) {
    const int ov = sf_from<STEP, true>(0, v, row), of = sf_from<STEP, true>(0, f, row);
    if (!f) { v += ov; f = of; }
}
__device__ __forceinline__ int sf_sum_from_below(bool fb, int v, int lane, int wave, SfCross &X) {
    int f = fb ? 1 : 0;
    const int row = lane >> 4;
    sf_sum_step<0>(f, v, row); sf_sum_step<1>(f, v, row); sf_sum_step<2>(f, v, row);
    sf_sum_step<3>(f, v, row); sf_sum_step<4>(f, v, row); sf_sum_step<5>(f, v, row);
    if (lane == 63) { X.f[wave] = f; X.v[wave][0] = __int_as_float(v); }
    int c = sf_next<true>(0, v);
    const int cf = sf_next<true>(0, f);
    __syncthreads();
    if (!cf)
        for (int w = wave - 1; w >= 0; w--) {
            c += __float_as_int(X.v[w][0]);
            if (X.f[w]) break;
        }
    return c;
}

template <int K> struct SfLds {
    float lo[3][K], hi[3][K];                // boxes by local triangle number
    int slot[K];
    unsigned short ord0[3][K];               // per axis: local triangle numbers sorted by (centre, slot), ranges kept contiguous ...
    unsigned short pb[K];                    // per position: first position of its range; | SF_ONE for a range of one (finished)
    // (from here to snode the words hold the 64-bit sort keys until the three orders are made: 8 K bytes)
    unsigned short ord1[3][K];               // ... and the other half of the ping-pong
    // tables of the ranges of two or more positions, by (first position >> 1): two such ranges never start at neighbouring positions
    unsigned short se[K / 2];                // end of the range
    int snode[K / 2];                        // node number
    int spar[K / 2];                         // parent * 2 + side
    int sdec[K / 2];                         // axis << 16 | left count
    unsigned char side[K];                   // per local triangle: 1 = goes left
    SfCross X[12];                           // one per scan of a level
    __device__ __forceinline__ unsigned short *ord(int which, int a) { return which ? ord1[a] : ord0[a]; }
};
#define SF_ONE 0x8000

__device__ __forceinline__ unsigned sf_key(float c) {           // order-preserving, -0 == +0
    const unsigned u = (unsigned)__float_as_int(c + 0.0f);
    return (u & 0x80000000u) ? ~u : u | 0x80000000u;
}

// E positions per lane, 256 lanes: E = 2 for tasks of up to 512 triangles (30 KB of LDS), E = 4 for up to 1024 (58 KB).
// `last` = 1: the tasks are counted down from the end of the task array (the choose kernel files the big ones there).
template <int E>
__global__ __launch_bounds__(64 * SF_W) void sb_finish_kernel(int ntasks, const int *__restrict__ tasks, int last,
                                                             const MptVec4 *__restrict__ prim0, const MptVec4 *__restrict__ prim1,
                                                             MptVec4 *__restrict__ fnode, int *__restrict__ meta) {
    constexpr int NT = 64 * SF_W, K = NT * E;
    __shared__ SfLds<K> L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x >= ntasks) return;
    const int *T = tasks + (last ? -(ptrdiff_t)(blockIdx.x + 1) : (ptrdiff_t)blockIdx.x) * SB_TASK_INTS;
    const int b0 = T[0], c = T[1] - T[0], node0 = T[2], par0 = T[3], level0 = T[4];
    const MptVec4 *prim = T[5] ? prim1 : prim0;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    int kp = 64;                                                  // positions that take part in the sort (a power of two >= c)
    while (kp < c) kp <<= 1;
    for (int i = tid; i < K; i += NT) {
        float l[3] = { INFINITY, INFINITY, INFINITY }, h[3] = { -INFINITY, -INFINITY, -INFINITY };
        int sl = 0x7fffffff;
        if (i < c) {
            const MptVec4 lo = prim[(size_t)(b0 + i) * 2], hi = prim[(size_t)(b0 + i) * 2 + 1];
            l[0] = lo.x; l[1] = lo.y; l[2] = lo.z; h[0] = hi.x; h[1] = hi.y; h[2] = hi.z; sl = __float_as_int(lo.w);
        }
        for (int a = 0; a < 3; a++) { L.lo[a][i] = l[a]; L.hi[a][i] = h[a]; }
        L.slot[i] = sl;
        L.pb[i] = (unsigned short)(i < c ? 0 : (i | SF_ONE));    // positions past the task's triangles: ranges of one, inert
    }
    __syncthreads();
    // ---- the three sorted orders: bitonic over kp positions, keys (centre, slot) in ONE 64-bit word (with the slot looked up through
    // the order array a compare-exchange was three dependent LDS round trips -- order, slot, then the writes -- and the sort a third
    // of a task's time); the padding sorts last.  The keys use the words of ord1 + se + snode, which are not needed before the levels.
    unsigned long long *key = (unsigned long long *)(void *)L.ord1;
    static_assert(K <= 1024, "ten bits of a sort key hold the local number");
    static_assert(offsetof(SfLds<K>, ord1) % 8 == 0 && offsetof(SfLds<K>, spar) - offsetof(SfLds<K>, ord1) >= 8 * K &&
                  offsetof(SfLds<K>, pb) < offsetof(SfLds<K>, ord1), "the sort keys lie over ord1 + se + snode");
    for (int a = 0; a < 3; a++) {
        unsigned short *od = L.ord0[a];
        // key = centre : 32 | slot : 22 | local number : 10 -- the order (centre, slot) and the payload in one word (mpt_sah_build takes
        // models of up to 2^22 triangles: the cap of option sah_max)
        for (int i = tid; i < kp; i += NT)
            key[i] = (i < c ? ((unsigned long long)sf_key(0.5f * (L.lo[a][i] + L.hi[a][i])) << 32) | ((unsigned long long)(unsigned)L.slot[i] << 10)
                            : 0xfffffffffffffc00ull) | (unsigned)i;
        __syncthreads();
        for (int k = 2; k <= kp; k <<= 1)
            for (int jj = k >> 1; jj > 0; jj >>= 1) {
                for (int t = tid; t < (kp >> 1); t += NT) {
                    const int i = ((t & ~(jj - 1)) << 1) | (t & (jj - 1)), l2 = i | jj;
                    const bool up = (i & k) == 0;
                    const unsigned long long ka = key[i], kb = key[l2];
                    if ((ka > kb) == up) { key[i] = kb; key[l2] = ka; }
                }
                // A stage whose partners are at most 64 apart stays inside the 128 positions a wave works on (compare-exchange t of a
                // stage takes positions i(t) and i(t) + jj, and the 64 consecutive t of a wave cover one aligned block of 128 for every
                // jj <= 64): the wave's own LDS accesses are in order, nobody else's are needed.  Only the stages that reach further
                // (6 of the 55 at 1024 positions), and the step from a phase's last stage into such a one, wait for the workgroup:
                // the stamps showed the sort at a third of a task's time, 1 150 cycles per stage, nearly all of it the barrier.
                if (jj >= 128 || (jj == 1 && k >= 128)) __syncthreads();
                else __builtin_amdgcn_wave_barrier();
            }
        __syncthreads();
        for (int i = tid; i < K; i += NT) od[i] = (unsigned short)(i < kp ? (unsigned)key[i] & 1023u : (unsigned)i);
        __syncthreads();
    }
    if (tid == 0) { L.se[0] = (unsigned short)c; L.snode[0] = node0; L.spar[0] = par0; }
    __syncthreads();
    const unsigned long long t_sorted = __builtin_amdgcn_s_memtime();
    // ---- level by level
    int cur = 0, level = level0, deepest = 0;
    const int i0 = tid * E;
    for (;;) {
        // the ranges of this lane's positions
        int pb[E], pe[E];
        unsigned head = 0, tail = 0, act = 0;
#pragma unroll
        for (int j = 0; j < E; j++) {
            const int raw = L.pb[i0 + j];
            pb[j] = raw & (SF_ONE - 1); pe[j] = (raw & SF_ONE) ? pb[j] + 1 : L.se[pb[j] >> 1];
            if (i0 + j == pb[j]) head |= 1u << j;
            if (i0 + j == pe[j] - 1) tail |= 1u << j;
            if (!(raw & SF_ONE)) act |= 1u << j;
        }
        if (!__syncthreads_or(act != 0)) break;
        deepest = level;
        float best[E]; int bestak[E];
#pragma unroll
        for (int j = 0; j < E; j++) { best[j] = INFINITY; bestak[j] = -1; }
        for (int a = 0; a < 3; a++) {
            const unsigned short *od = L.ord(cur, a);
            SfBox bx[E];
#pragma unroll
            for (int j = 0; j < E; j++) {
                const int p = od[i0 + j];
                for (int r = 0; r < 3; r++) { bx[j].l[r] = L.lo[r][p]; bx[j].h[r] = L.hi[r][p]; }
            }
            // suffix: S(i) = union of the boxes at i .. end of the range; its area.  Two sweeps instead of E stored boxes:
            // the first gives the lane's aggregate (up to its first range end), the second starts from the carry
            float sarea[E];
            {
                SfBox run; sf_clear(run);
#pragma unroll
                for (int j = E - 1; j >= 0; j--) { if (tail >> j & 1) run = bx[j]; else sf_add(run, bx[j]); }
                run = sf_carry_from_above(tail != 0, run, lane, wave, L.X[a * 4 + 0]);
#pragma unroll
                for (int j = E - 1; j >= 0; j--) {
                    if (tail >> j & 1) run = bx[j]; else sf_add(run, bx[j]);
                    sarea[j] = sb_half_area(run.l, run.h);
                    // a range's own box (at its first position) goes into its parent's record
                    if (a == 0 && (head >> j & 1) && (act >> j & 1)) {
                        const int par = L.spar[pb[j] >> 1];
                        if (par >= 0) sb_write_child(fnode, par, run.l, run.h, L.snode[pb[j] >> 1]);
                    }
                }
            }
            // prefix (exclusive): P(i) = union of the boxes at start of the range .. i - 1; the cost of the split in front of i
            float cost[E];
            {
                SfBox run; sf_clear(run);
#pragma unroll
                for (int j = 0; j < E; j++) { if (head >> j & 1) sf_clear(run); sf_add(run, bx[j]); }
                run = sf_carry_from_below(head != 0, run, lane, wave, L.X[a * 4 + 1]);
#pragma unroll
                for (int j = 0; j < E; j++) {
                    if (head >> j & 1) sf_clear(run);
                    cost[j] = INFINITY;
                    if ((act >> j & 1) && !(head >> j & 1)) {
                        const int k = i0 + j - pb[j];
                        cost[j] = sb_half_area(run.l, run.h) * k + sarea[j] * (pe[j] - pb[j] - k);
                    }
                    sf_add(run, bx[j]);
                }
            }
            // segmented minimum of (cost, position), in position order, strict: the first of equal costs stays
            {
                float mc = INFINITY; int mp = -1;
#pragma unroll
                for (int j = 0; j < E; j++) {
                    if (head >> j & 1) { mc = INFINITY; mp = -1; }
                    if (cost[j] < mc) { mc = cost[j]; mp = i0 + j; }
                }
                sf_min_from_below(head != 0, mc, mp, lane, wave, L.X[a * 4 + 2]);
#pragma unroll
                for (int j = 0; j < E; j++) {
                    if (head >> j & 1) { mc = INFINITY; mp = -1; }
                    if (cost[j] < mc) { mc = cost[j]; mp = i0 + j; }
                    if ((tail >> j & 1) && (act >> j & 1) && mp >= 0) {
                        // the axis takes part in a range only if the range's centres differ along it (the host pass's rule)
                        const int pf = od[pb[j]], pl = od[pe[j] - 1];
                        const float cf = 0.5f * (L.lo[a][pf] + L.hi[a][pf]), cl_ = 0.5f * (L.lo[a][pl] + L.hi[a][pl]);
                        if (cl_ > cf && mc < best[j]) { best[j] = mc; bestak[j] = (a << 16) | (mp - pb[j]); }
                    }
                }
            }
        }
        // ---- decisions (at the ranges' last positions), then the side of every triangle
#pragma unroll
        for (int j = 0; j < E; j++)
            if ((tail >> j & 1) && (act >> j & 1))
                L.sdec[pb[j] >> 1] = bestak[j] >= 0 ? bestak[j] : (pe[j] - pb[j]) / 2;      // no split found: halve the range in the order of axis 0
        __syncthreads();
        int kk[E];
#pragma unroll
        for (int j = 0; j < E; j++) {
            kk[j] = 0;
            if (act >> j & 1) {
                const int ak = L.sdec[pb[j] >> 1];
                kk[j] = ak & 0xffff;
                L.side[L.ord(cur, ak >> 16)[i0 + j]] = (unsigned char)(i0 + j - pb[j] < kk[j]);
            }
        }
        __syncthreads();
        // ---- stable partition of the three orders
        for (int a = 0; a < 3; a++) {
            const unsigned short *od = L.ord(cur, a);
            unsigned short *on = L.ord(cur ^ 1, a);
            int p[E];
            unsigned sd = 0;
            int run = 0;
#pragma unroll
            for (int j = 0; j < E; j++) {
                p[j] = od[i0 + j];
                if (head >> j & 1) run = 0;
                if ((act >> j & 1) && L.side[p[j]]) { sd |= 1u << j; run++; }
            }
            run = sf_sum_from_below(head != 0, run, lane, wave, L.X[a * 4 + 3]);
#pragma unroll
            for (int j = 0; j < E; j++) {
                if (head >> j & 1) run = 0;
                int dest = i0 + j;
                if (act >> j & 1) dest = (sd >> j & 1) ? pb[j] + run : pb[j] + kk[j] + (i0 + j - pb[j]) - run;
                on[dest] = (unsigned short)p[j];
                if (sd >> j & 1) run++;
            }
        }
        // ---- the children: range tables for the next level, single triangles into their parent's record
        int me[E];
#pragma unroll
        for (int j = 0; j < E; j++) me[j] = (act >> j & 1) ? L.snode[pb[j] >> 1] : 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < E; j++) {
            if (!(act >> j & 1)) continue;
            const int i = i0 + j, b = pb[j], e = pe[j], m = b + kk[j];
            // (a child of one position is finished: marked in pb, no table entry)
            L.pb[i] = (unsigned short)(i < m ? (m - b >= 2 ? b : b | SF_ONE) : (e - m >= 2 ? m : m | SF_ONE));
            if (i == b || i == m) {
                const int k = i == m ? 1 : 0, end = k ? e : m, node = k ? me[j] + kk[j] : me[j] + 1;
                if (end - i >= 2) { L.se[i >> 1] = (unsigned short)end; L.snode[i >> 1] = node; L.spar[i >> 1] = me[j] * 2 + k; }
                else {
                    const int q = L.ord(cur ^ 1, 0)[i];
                    const float l[3] = { L.lo[0][q], L.lo[1][q], L.lo[2][q] }, h[3] = { L.hi[0][q], L.hi[1][q], L.hi[2][q] };
                    sb_write_child(fnode, me[j] * 2 + k, l, h, ~L.slot[q]);
                }
                if (k == 0) {                                      // the pad words of the node's record (ids come from the children)
                    float *r = (float *)(fnode + (size_t)me[j] * 4);
                    r[14] = 0.f; r[15] = 0.f;
                }
            }
        }
        __syncthreads();
        cur ^= 1; level++;
    }
    if (tid == 0 && deepest > 0) atomicMax(meta + META_DEPTH, deepest);
    if (tid == 0) {                 // statistics (MptSahStats): where a task's time goes
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        atomicAdd(meta + META_T_SORT, (int)((t_sorted - t_begin) >> 10));
        atomicAdd(meta + META_T_LOOP, (int)((t_end - t_sorted) >> 10));
        atomicMax(meta + META_T_MAX, (int)((t_end - t_begin) >> 10));
        atomicAdd(meta + META_T_LEVELS, level - level0);
        atomicMax(meta + META_T_MAXLEVELS, level - level0);
    }
}

// ------------------------------------------------------------------ driver
MPT_KERNEL_API size_t mpt_sah_seg_capacity(int n) { return (size_t)n / (SB_K2 + 1) + 2; }
MPT_KERNEL_API size_t mpt_sah_chunk_capacity(int n) { return (size_t)n / SB_CHUNK_MIN + mpt_sah_seg_capacity(n) + 2; }
// words of chunk bins a level can need: a level of nseg segments streaming `elems` positions has at most elems / CH + nseg chunks
// of 24 nb words; nb / CH <= 1 / 16 where CH = 16 nb, elems / CH <= 256 + 1 where CH is the level's 256th part (or the
// 2048 floor under it); nb x nseg <= the budget (or 32 x nseg at the floor)
MPT_KERNEL_API size_t mpt_sah_part_words(int n) {
    const size_t sc = mpt_sah_seg_capacity(n);
    const size_t segbins = std::max((size_t)SB_BIN_BUDGET, (size_t)SB_MINBINS * sc);
    return 24 * ((size_t)n / 16 + (size_t)SB_MAXBINS * 258 + segbins);
}
// words of per-segment bins a level can need: 21 nb nseg
MPT_KERNEL_API size_t mpt_sah_segbin_words(int n) {
    return 21 * std::max((size_t)SB_BIN_BUDGET, (size_t)SB_MINBINS * mpt_sah_seg_capacity(n));
}
// the most a level of nseg segments of a model of n triangles can write (it streams at most all n positions)
MPT_KERNEL_API size_t mpt_sah_level_words(int n, size_t nseg, int *nb_out) {
    const int nb = sb_bins_for((long long)nseg), ch = sb_chunk_for(nb, n);
    if (nb_out) *nb_out = nb;
    return (size_t)SB_CHUNK_WORDS(nb) * ((size_t)n / ch + nseg);
}
MPT_KERNEL_API size_t mpt_sah_task_capacity(int n) { return (size_t)n / 2 + 2; }
MPT_KERNEL_API int mpt_sah_task_max(void) { return SB_K2; }

static hipError_t sb_big_lds(const void *fn) {
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 21 * SB_MAXBINS * (int)sizeof(int));
}

// verts [3n][8] and leaf [n] on the device (the LBVH build's); writes fnode [n-1][4] and *depth.  Needs n >= 2.
MPT_KERNEL_API hipError_t mpt_sah_build(const MptSahBuffers *B, int *depth, hipStream_t stream) {
    const int n = B->n;
    if (n < 2 || n > (1 << 22)) return hipErrorInvalidValue;      // (22 bits of a sort key hold the slot: sb_finish_kernel)
    hipError_t e;
    {   // the two kernels with up to 84 KB of dynamic LDS: the attribute is per device (a process may hold contexts on several)
        static bool attr_done[64] = { false };
        int dev = 0;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
        if (dev < 0 || dev >= 64 || !attr_done[dev]) {
            if ((e = sb_big_lds((const void *)sb_bin_kernel)) != hipSuccess) return e;
            if ((e = sb_big_lds((const void *)sb_choose_kernel)) != hipSuccess) return e;
            if (dev >= 0 && dev < 64) attr_done[dev] = true;
        }
    }
    int meta[META_INTS] = { 0 };
    static std::atomic<int> mail_seq{0};                          // (sequence numbers of the mailbox: never reused within a process)
    if ((e = hipMemsetAsync(B->meta, 0, sizeof meta, stream)) != hipSuccess) return e;
    MptSahStats st{};
    int nb = sb_bins_for(1), CH = sb_chunk_for(nb, n);
    int seg0[SB_SEG_INTS] = { 0, n, 0, -1, 0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000,
                              0, (n + CH - 1) / CH, 0, 0, 0, 0 };
    if ((e = hipMemcpyAsync(B->seg[0], seg0, sizeof seg0, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    const int gp = std::min((n + SB_BLOCK - 1) / SB_BLOCK, 2048);       // (their partial bounds take 6 words each of B->part: far below its size)
    hipLaunchKernelGGL(sb_prims_kernel, dim3(gp), dim3(SB_BLOCK), 0, stream, B->verts, B->leaf, n, B->prim[0], B->seg[0], B->part,
                       B->meta + META_INTS - 1);
    int ntasks = 0, ntasks2 = 0;
    if (n <= SB_K2) {
        // the whole tree is one task
        const int task[SB_TASK_INTS] = { 0, n, 0, -1, 1, 0, 0, 0 };
        int *where = n <= SB_K ? B->tasks : B->tasks + (B->task_cap - 1) * SB_TASK_INTS;
        if ((e = hipMemcpyAsync(where, task, sizeof task, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
        (n <= SB_K ? ntasks : ntasks2) = 1;
    } else {
        int nseg = 1, nchunks = seg0[SEG_NCHUNK], cur = 0, level = 1;
        if ((e = hipMemsetAsync(B->ch_seg, 0, (size_t)nchunks * sizeof(int), stream)) != hipSuccess) return e;
        while (nseg > 0) {
            if (level > 60) return hipErrorInvalidValue;
            if ((size_t)nchunks * SB_CHUNK_WORDS(nb) > B->part_words || (size_t)nseg > B->seg_cap || (size_t)nchunks > B->chunk_cap ||
                (size_t)nseg * 21 * nb > B->segbin_words) return hipErrorOutOfMemory;
            const size_t lds = (size_t)21 * nb * sizeof(int);
            st.levels++; st.elems += level == 1 ? n : meta[META_ELEMS]; st.chunks += nchunks; st.part_words += (long long)nchunks * SB_CHUNK_WORDS(nb);
            st.segments += nseg;
            // (the chunk list is double-buffered like the segment table: the plan kernel writes the next level's while the scatter
            // pass still reads this level's)
            int *chs_cur = B->ch_seg + (size_t)cur * B->chunk_cap, *chs_next = B->ch_seg + (size_t)(cur ^ 1) * B->chunk_cap;
            hipLaunchKernelGGL(sb_bin_kernel, dim3(nchunks), dim3(SB_BLOCK), lds, stream, B->prim[cur], B->seg[cur], chs_cur, nb, CH, B->part);
            if (nchunks > nseg)                          // (some segment has more than one chunk)
                hipLaunchKernelGGL(sb_reduce_kernel, dim3((21 * nb + SB_BLOCK - 1) / SB_BLOCK, nseg), dim3(SB_BLOCK), 0, stream, B->seg[cur], nb, B->part,
                                   B->segbins);
            hipLaunchKernelGGL(sb_choose_kernel, dim3(nseg), dim3(SB_BLOCK), lds, stream, B->seg[cur], nb, CH, B->part, B->segbins, level, cur ^ 1,
                               B->dec, B->ch_left, B->tasks, (int)B->task_cap, B->meta, B->fnode);
            const int seq = ++mail_seq;
            hipLaunchKernelGGL(sb_plan_kernel, dim3(1), dim3(1024), 0, stream, nseg, B->seg[cur], B->dec, B->seg[cur ^ 1], chs_next, B->meta,
                               B->mail_dev, seq);
            hipLaunchKernelGGL(sb_scatter_kernel, dim3(nchunks), dim3(SB_BLOCK), 0, stream, B->prim[cur], B->prim[cur ^ 1], B->seg[cur],
                               chs_cur, B->dec, B->ch_left, nb, CH, B->seg[cur ^ 1], B->fnode);
            bool mailed = false;
            if (B->mail_host) {
                // poll the mailbox (a stream error ends the wait; after 200 ms without the word the ordinary read-back takes over)
                const auto t0 = std::chrono::steady_clock::now();
                for (unsigned spin = 0;; spin++) {
                    if (__atomic_load_n((const int *)(B->mail_host + 8), __ATOMIC_ACQUIRE) == seq) { mailed = true; break; }
                    if ((spin & 1023u) == 1023u) {
                        const hipError_t q = hipStreamQuery(stream);
                        if (q != hipSuccess && q != hipErrorNotReady) return q;
                        if (q == hipSuccess && __atomic_load_n((const int *)(B->mail_host + 8), __ATOMIC_ACQUIRE) != seq) break;   // (the stream is done and nothing came)
                        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
                    }
                }
            }
            if (mailed) {
                meta[META_NSEG] = B->mail_host[0]; meta[META_NCHUNK] = B->mail_host[1]; meta[META_NB] = B->mail_host[2]; meta[META_CH] = B->mail_host[3];
                meta[META_ELEMS] = B->mail_host[4]; meta[META_NTASK] = B->mail_host[5]; meta[META_NTASK2] = B->mail_host[6]; meta[META_BAD] = B->mail_host[7];
            } else {
                if ((e = hipMemcpyAsync(meta, B->meta, sizeof meta, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
                if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
            }
            if (meta[META_BAD]) return hipErrorInvalidValue;
            nseg = meta[META_NSEG]; nchunks = meta[META_NCHUNK]; nb = meta[META_NB]; CH = meta[META_CH];
            cur ^= 1; level++;
        }
        ntasks = meta[META_NTASK]; ntasks2 = meta[META_NTASK2];
        if ((size_t)ntasks + (size_t)ntasks2 > B->task_cap) return hipErrorInvalidValue;
    }
    // (the big tasks first: they take longer)
    if (ntasks2 > 0)
        hipLaunchKernelGGL(sb_finish_kernel<SB_K2 / (64 * SF_W)>, dim3(ntasks2), dim3(64 * SF_W), 0, stream, ntasks2, B->tasks + B->task_cap * SB_TASK_INTS, 1,
                           B->prim[0], B->prim[1], B->fnode, B->meta);
    if (ntasks > 0)
        hipLaunchKernelGGL(sb_finish_kernel<SB_K / (64 * SF_W)>, dim3(ntasks), dim3(64 * SF_W), 0, stream, ntasks, B->tasks, 0, B->prim[0], B->prim[1], B->fnode, B->meta);
    if ((e = hipMemcpyAsync(meta, B->meta, sizeof meta, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    *depth = meta[META_DEPTH];
    st.tasks_small = ntasks; st.tasks_big = ntasks2;
    st.t_sort_k = meta[META_T_SORT]; st.t_loop_k = meta[META_T_LOOP]; st.t_max_k = meta[META_T_MAX]; st.task_levels = meta[META_T_LEVELS];
    st.task_levels_max = meta[META_T_MAXLEVELS];
    if (B->stats) *B->stats = st;
    return hipGetLastError();
}
