// wide_build.hip -- the 4-wide collapse of the fast build's binary tree, on the device.
//
// The gather kernels walk the binary tree of c->fnode (a node record holds its two children's boxes) collapsed into
// 4-wide nodes: starting from a node's two children, the internal child with the largest surface area is replaced by
// its own two children until four are held (or only leaves are left).  Round 2 did this on the host over downloaded
// records (64 MB down, 0.1 s, 96 MB up at 1 M triangles); here it stays on the device:
//   one pass per LEVEL of the wide tree (breadth first, so wide node numbers are the host pass's):
//     expand   one lane per wide node of the level: grows its four children from the binary records, writes the
//              128-byte record with exact boxes (wnode) and the 64-byte record with 8-bit child boxes rounded
//              outwards (qnode), counts its internal children; the workgroup's exclusive prefix of those counts
//     totals   one workgroup: prefix sum of the workgroups' totals; the next level's {first node, count}
//     link     gives the internal children their numbers (level end + prefix + rank) in both records and records
//              which binary node each of them grows from
//   The level's first node and count live in a device-side table (lv[level]); the host enqueues eight levels at a time
//   with grids sized for the most a level can hold (4^level, at most every node) and reads the table back once per group
//   (round 6: the read-back + rocPRIM scan per level were 0.6 of the 0.9 ms this took at 100 000 triangles).
// The arithmetic of the quantisation is the host pass's, operation for operation, without contraction (this file
// is compiled with -ffp-contract=off): both passes produce the same bytes (tests/test_parity_gpu.py).
// An unused child slot holds the id of leaf slot n -- one extra triangle record of NaNs that no ray can hit -- and a
// box no ray enters; round 2 stored id 0 there, the root, which a ray along (1,1,1) could be sent back to.

#include <atomic>
#include <chrono>
#include <cstring>
#include <hip/hip_runtime.h>
#include <cmath>
#include "mpt_types.h"

#define WB_BLOCK 256
#define WB_MAX_LEVELS 64      // levels of the wide tree the level table holds (a binary tree of depth <= 62 collapses into no more)

struct WbChild { int id; float lo[3], hi[3]; };

__device__ __forceinline__ int wb_asi(float f) { return __float_as_int(f); }
__device__ __forceinline__ float wb_asf(int v) { return __int_as_float(v); }

__device__ __forceinline__ void wb_children_of(const MptVec4 *__restrict__ fnode, int b, WbChild out[2]) {
    const MptVec4 r0 = fnode[(size_t)b * 4 + 0], r1 = fnode[(size_t)b * 4 + 1], r2 = fnode[(size_t)b * 4 + 2], r3 = fnode[(size_t)b * 4 + 3];
    out[0].id = wb_asi(r3.x); out[1].id = wb_asi(r3.y);
    out[0].lo[0] = r0.x; out[1].lo[0] = r0.y; out[0].hi[0] = r0.z; out[1].hi[0] = r0.w;
    out[0].lo[1] = r1.x; out[1].lo[1] = r1.y; out[0].hi[1] = r1.z; out[1].hi[1] = r1.w;
    out[0].lo[2] = r2.x; out[1].lo[2] = r2.y; out[0].hi[2] = r2.z; out[1].hi[2] = r2.w;
}

__device__ __forceinline__ float wb_area(const WbChild &c) {
    const float dx = fmaxf(c.hi[0] - c.lo[0], 0.f), dy = fmaxf(c.hi[1] - c.lo[1], 0.f), dz = fmaxf(c.hi[2] - c.lo[2], 0.f);
    return dx * dy + dy * dz + dz * dx;
}

// surface areas are summed as integers in units of 2^-40 of the root's (the sum of <= 2^22 of them stays below 2^63): the
// figure is the same bits run to run, whatever order the lanes arrive in (round-3 ADVICE: it was a double atomicAdd per lane)
__device__ __forceinline__ unsigned long long wb_area_fixed(const MptVec4 *__restrict__ fnode, float a) {
    WbChild ch[2];
    wb_children_of(fnode, 0, ch);
    WbChild root = ch[0];
    for (int k = 0; k < 3; k++) { root.lo[k] = fminf(ch[0].lo[k], ch[1].lo[k]); root.hi[k] = fmaxf(ch[0].hi[k], ch[1].hi[k]); }
    const double ra = (double)wb_area(root);
    if (!(ra > 0.0)) return 0ull;
    const double r = fmin(fmax((double)a / ra, 0.0), 1.0);
    return (unsigned long long)(r * 1099511627776.0);
}

// expand: wide nodes [lo, lo + count) of one level.  ncount[t] = internal children of wide node lo + t (0 beyond the level)
__global__ __launch_bounds__(WB_BLOCK) void wb_expand_kernel(const MptVec4 *__restrict__ fnode, const int *__restrict__ bin_of,
                                                            const int *__restrict__ lv, int empty_id, MptVec4 *__restrict__ wnode,
                                                            MptVec4 *__restrict__ qnode, int *__restrict__ offset, int *__restrict__ btot,
                                                            unsigned long long *__restrict__ area_sum) {
    const int lo = lv[0], count = lv[1];
    if ((int)(blockIdx.x * WB_BLOCK) >= count) return;            // (the grid is sized for the most this level can hold)
    const int t = blockIdx.x * WB_BLOCK + threadIdx.x;
    const bool live = t < count;
    const int w = lo + (live ? t : 0);                        // (lanes past the level stay for the wave reduction below and add 0)
    WbChild ch[4];
    int cnt = 2;
    wb_children_of(fnode, bin_of[w], ch);
    {   // surface area of the node this record is grown from (= the union of its two children): the collapse's share of
        // the expected fetches per ray (informational: option "wide_ratio_permille")
        WbChild own = ch[0];
        for (int a = 0; a < 3; a++) { own.lo[a] = fminf(ch[0].lo[a], ch[1].lo[a]); own.hi[a] = fmaxf(ch[0].hi[a], ch[1].hi[a]); }
        const unsigned long long q = live ? wb_area_fixed(fnode, wb_area(own)) : 0ull;
        unsigned long long tot = q;                           // one atomic per workgroup, and integers: the sum does not depend on the order
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);      // every lane of the wave is active here (no early return)
        __shared__ unsigned long long wtot[WB_BLOCK / 64];
        if ((threadIdx.x & 63) == 0) wtot[threadIdx.x >> 6] = tot;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long t = 0ull;
            for (int w = 0; w < WB_BLOCK / 64; w++) t += wtot[w];
            if (t) atomicAdd(area_sum, t);
        }
    }
    int nint = 0;
    if (live) {
    while (cnt < 4) {
        int best = -1; float ba = -1.f;
        for (int k = 0; k < cnt; k++)
            if (ch[k].id >= 0) { const float a = wb_area(ch[k]); if (a > ba) { ba = a; best = k; } }
        if (best < 0) break;
        WbChild two[2];
        wb_children_of(fnode, ch[best].id, two);
        ch[best] = two[0];
        ch[cnt++] = two[1];
    }
    // ---- the exact record: {lo.x[4]} {hi.x[4]} {lo.y[4]} {hi.y[4]} {lo.z[4]} {hi.z[4]} {id[4]} {-}
    float rec[32];
    for (int k = 0; k < 32; k++) rec[k] = 0.f;
    int ids[4];
    for (int k = 0; k < 4; k++) {
        float l[3] = { 1e30f, 1e30f, 1e30f }, h[3] = { 1e30f, 1e30f, 1e30f };   // unused child: out of every ray's reach
        ids[k] = empty_id;
        if (k < cnt) {
            for (int a = 0; a < 3; a++) { l[a] = ch[k].lo[a]; h[a] = ch[k].hi[a]; }
            ids[k] = ch[k].id;                        // internal children: the BINARY node for now, numbered by wb_link
            if (ch[k].id >= 0) nint++;
        }
        for (int a = 0; a < 3; a++) { rec[(2 * a) * 4 + k] = l[a]; rec[(2 * a + 1) * 4 + k] = h[a]; }
    }
    for (int k = 0; k < 4; k++) rec[24 + k] = wb_asf(ids[k]);
    MptVec4 *wo = wnode + (size_t)w * 8;
    for (int k = 0; k < 8; k++) wo[k] = { rec[4 * k], rec[4 * k + 1], rec[4 * k + 2], rec[4 * k + 3] };
    // ---- the quantised record: child planes as bytes over the node's own box, rounded outwards by a quarter of a step
    // more than needed (the kernel's decode q * (scale * inv) + (origin * inv - o * inv) is off by far less)
    float plo[3] = { INFINITY, INFINITY, INFINITY }, phi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int k = 0; k < cnt; k++)
        for (int a = 0; a < 3; a++) { plo[a] = fminf(plo[a], ch[k].lo[a]); phi[a] = fmaxf(phi[a], ch[k].hi[a]); }
    float scale[3];
    unsigned qlo[3] = { 0, 0, 0 }, qhi[3] = { 0, 0, 0 };
    for (int a = 0; a < 3; a++) {
        const float e = phi[a] - plo[a];
        float sc = e > 0.f ? e / 255.f : 0.f;
        // 255 steps must reach the far side in f32, and a flat node still needs a positive step
        while (e > 0.f && plo[a] + 255.f * sc < phi[a]) sc = nextafterf(sc, INFINITY);
        if (!(sc > 0.f)) sc = fmaxf(fabsf(plo[a]) * 1e-6f, 1e-30f);
        scale[a] = sc;
        for (int k = 0; k < 4; k++) {
            unsigned l = 255, h = 0;                  // unused child: an inverted box
            if (k < cnt) {
                const float fl = floorf((ch[k].lo[a] - plo[a]) / sc - 0.25f), fh = ceilf((ch[k].hi[a] - plo[a]) / sc + 0.25f);
                l = (unsigned)fminf(255.f, fmaxf(0.f, fl));
                h = (unsigned)fminf(255.f, fmaxf(0.f, fh));
            }
            qlo[a] |= l << (8 * k); qhi[a] |= h << (8 * k);
        }
    }
    MptVec4 *qo = qnode + (size_t)w * 4;
    qo[0] = { plo[0], plo[1], plo[2], scale[0] };
    qo[1] = { scale[1], scale[2], wb_asf((int)qlo[0]), wb_asf((int)qhi[0]) };
    qo[2] = { wb_asf((int)qlo[1]), wb_asf((int)qhi[1]), wb_asf((int)qlo[2]), wb_asf((int)qhi[2]) };
    qo[3] = { wb_asf(ids[0]), wb_asf(ids[1]), wb_asf(ids[2]), wb_asf(ids[3]) };
    }
    // where this node's internal children start among the workgroup's (exclusive prefix, node order), and the workgroup's total
    __shared__ int wsum[WB_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = nint;
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < WB_BLOCK / 64; w++) { const int x = wsum[w]; if (w < wave) base += x; tot += x; }
    if (live) offset[t] = base + inc - nint;
    if (threadIdx.x == 0) btot[blockIdx.x] = tot;
}

// the workgroups' totals -> where each workgroup's children start (bbase), and the next level: {first node, count, levels so far, -}
__global__ __launch_bounds__(1024) void wb_totals_kernel(const int *__restrict__ lv, int *__restrict__ lv_next, const int *__restrict__ btot,
                                                        int *__restrict__ bbase, int ni, int *__restrict__ bad, int *mail, int seq) {
    __shared__ int wsum[16];
    const int lo = lv[0], count = lv[1], nblk = (count + WB_BLOCK - 1) / WB_BLOCK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = 0;
    for (int b0 = 0; b0 < nblk; b0 += 1024) {
        const int b = b0 + threadIdx.x;
        const int v = b < nblk ? btot[b] : 0;
        int inc = v;
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
        __syncthreads();
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int base = 0, tot = 0;
        for (int w = 0; w < 16; w++) { const int x = wsum[w]; if (w < wave) base += x; tot += x; }
        if (b < nblk) bbase[b] = carry + base + inc - v;
        carry += tot;
    }
    if (threadIdx.x == 0) {
        if (lo + count + carry > ni) { *bad = 1; carry = 0; }              // (a binary tree has ni internal nodes: cannot happen)
        const int row[4] = { lo + count, carry, lv[2] + (count > 0 ? 1 : 0), 0 };
        for (int k = 0; k < 4; k++) lv_next[k] = row[k];
        if (mail) {     // the last level of a group: the row and the error flag straight into the host's pinned mailbox, the sequence word last
            for (int k = 0; k < 4; k++) __hip_atomic_store(mail + 16 + k, row[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(mail + 20, *bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(mail + 24, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// link: the internal children of wide node lo + t get the numbers next + offset[t] ... in slot order (the host pass's
// queue order), in both records; bin_of of the new nodes = the binary node they grow from
__global__ __launch_bounds__(WB_BLOCK) void wb_link_kernel(const int *__restrict__ lv, const int *__restrict__ offset, const int *__restrict__ bbase,
                                                          MptVec4 *__restrict__ wnode, MptVec4 *__restrict__ qnode, int *__restrict__ bin_of) {
    const int lo = lv[0], count = lv[1], next = lo + count;
    const int t = blockIdx.x * WB_BLOCK + threadIdx.x;
    if (t >= count) return;
    const int w = lo + t;
    MptVec4 idv = wnode[(size_t)w * 8 + 6];
    int ids[4] = { wb_asi(idv.x), wb_asi(idv.y), wb_asi(idv.z), wb_asi(idv.w) };
    int at = next + bbase[blockIdx.x] + offset[t];
    bool any = false;
    for (int k = 0; k < 4; k++)
        if (ids[k] >= 0) { bin_of[at] = ids[k]; ids[k] = at++; any = true; }
    if (any) {
        idv = { wb_asf(ids[0]), wb_asf(ids[1]), wb_asf(ids[2]), wb_asf(ids[3]) };
        wnode[(size_t)w * 8 + 6] = idv;
        qnode[(size_t)w * 4 + 3] = idv;
    }
}

// sum of the surface areas of all binary nodes (the binary tree's expected fetches per ray, same informational figure)
// (at most 512 workgroups stride over the nodes: 16 000 same-address atomics took 0.19 ms at a million triangles)
__global__ __launch_bounds__(WB_BLOCK) void wb_area_kernel(const MptVec4 *__restrict__ fnode, int ni, unsigned long long *__restrict__ area_sum) {
    __shared__ unsigned long long wtot[WB_BLOCK / 64];
    unsigned long long a = 0ull;
    for (int b = blockIdx.x * WB_BLOCK + threadIdx.x; b < ni; b += gridDim.x * WB_BLOCK) {
        WbChild ch[2];
        wb_children_of(fnode, b, ch);
        WbChild own = ch[0];
        for (int k = 0; k < 3; k++) { own.lo[k] = fminf(ch[0].lo[k], ch[1].lo[k]); own.hi[k] = fmaxf(ch[0].hi[k], ch[1].hi[k]); }
        a += wb_area_fixed(fnode, wb_area(own));
    }
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
    if ((threadIdx.x & 63) == 0) wtot[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0ull;
        for (int w = 0; w < WB_BLOCK / 64; w++) t += wtot[w];
        if (t) atomicAdd(area_sum, t);
    }
}

MPT_KERNEL_API hipError_t mpt_wide_scan_bytes(int ni, size_t *bytes) {      // (the level table + the workgroups' totals and bases)
    *bytes = ((size_t)WB_MAX_LEVELS + 2) * 4 * sizeof(int) + 2 * ((size_t)std::max(ni, 1) / WB_BLOCK + 2) * sizeof(int) + 64;
    return hipSuccess;
}

// Builds the wide records of the binary tree in fnode (ni = n - 1 internal nodes) into wnode [ni][8] / qnode [ni][4]
// (capacity: one wide node per binary node, the worst case).  Outputs: *nwide, *depth (levels), area sums [0] over the wide
// nodes' source nodes, [1] over all binary nodes.  `ncount` [ni] holds the nodes' offsets within their workgroup, `scan_tmp`
// (mpt_wide_scan_bytes) the level table and the workgroups' totals; `offset` is not used any more.  One read-back per eight levels.
MPT_KERNEL_API hipError_t mpt_wide_build(const MptVec4 *fnode, int n, MptVec4 *wnode, MptVec4 *qnode, int *bin_of, int *ncount,
                                     int *offset, void *scan_tmp, size_t scan_bytes, double *d_area, int *nwide, int *depth,
                                     double area[2], hipStream_t stream, volatile int *mail_host, int *mail_dev) {
    const int ni = n > 1 ? n - 1 : 0;
    *nwide = 0; *depth = 0; area[0] = area[1] = 0.0;
    if (ni < 1) return hipSuccess;
    (void)offset;
    hipError_t e;
    size_t need = 0;
    mpt_wide_scan_bytes(ni, &need);
    if (scan_bytes < need) return hipErrorInvalidValue;
    int *lv = (int *)scan_tmp;                                           // [WB_MAX_LEVELS + 2][4]
    int *bad = lv + (WB_MAX_LEVELS + 1) * 4;                             // (the last row: error flag)
    int *btot = lv + (WB_MAX_LEVELS + 2) * 4, *bbase = btot + (ni / WB_BLOCK + 2);
    if ((e = hipMemsetAsync(d_area, 0, 2 * sizeof(double), stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(bin_of, 0, sizeof(int), stream)) != hipSuccess) return e;        // wide node 0 grows from the root
    if ((e = hipMemsetAsync(lv, 0, (WB_MAX_LEVELS + 2) * 4 * sizeof(int), stream)) != hipSuccess) return e;
    const int first[4] = { 0, 1, 0, 0 };
    if ((e = hipMemcpyAsync(lv, first, sizeof first, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    unsigned long long *d_fixed = (unsigned long long *)d_area;          // (two 8-byte words either way)
    hipLaunchKernelGGL(wb_area_kernel, dim3(std::min((ni + WB_BLOCK - 1) / WB_BLOCK, 512)), dim3(WB_BLOCK), 0, stream, fnode, ni, d_fixed + 1);
    int level = 0, row[4] = { 0, 1, 0, 0 };
    long long most = 1;                                                  // the most nodes level `level` can hold: 4^level, at most ni
    while (row[1] > 0) {
        if (level >= WB_MAX_LEVELS) return hipErrorInvalidValue;
        const int group = std::min(level + 8, (int)WB_MAX_LEVELS);
        static std::atomic<int> mail_seq{0};
        const int seq = ++mail_seq;
        for (; level < group; level++) {
            const int grid = (int)((std::min(most, (long long)ni) + WB_BLOCK - 1) / WB_BLOCK);
            hipLaunchKernelGGL(wb_expand_kernel, dim3(grid), dim3(WB_BLOCK), 0, stream, fnode, bin_of, lv + level * 4, ~n, wnode, qnode, ncount,
                               btot, d_fixed);
            const bool last_of_group = level + 1 == group && mail_host != nullptr;
            hipLaunchKernelGGL(wb_totals_kernel, dim3(1), dim3(1024), 0, stream, lv + level * 4, lv + (level + 1) * 4, btot, bbase, ni, bad,
                               last_of_group ? mail_dev : (int *)nullptr, seq);
            hipLaunchKernelGGL(wb_link_kernel, dim3(grid), dim3(WB_BLOCK), 0, stream, lv + level * 4, ncount, bbase, wnode, qnode, bin_of);
            most = std::min(most * 4, (long long)ni);
        }
        int isbad = 0;
        bool mailed = false;
        if (mail_host) {        // (the link kernel of the group's last level may still run: what follows goes behind it on the stream)
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spin = 0;; spin++) {
                if (__atomic_load_n((const int *)(mail_host + 24), __ATOMIC_ACQUIRE) == seq) { mailed = true; break; }
                if ((spin & 1023u) == 1023u) {
                    const hipError_t q = hipStreamQuery(stream);
                    if (q != hipSuccess && q != hipErrorNotReady) return q;
                    if (q == hipSuccess && __atomic_load_n((const int *)(mail_host + 24), __ATOMIC_ACQUIRE) != seq) break;
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
                }
            }
        }
        if (mailed) { for (int k = 0; k < 4; k++) row[k] = mail_host[16 + k]; isbad = mail_host[20]; }
        else {
            if ((e = hipMemcpyAsync(row, lv + level * 4, sizeof row, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
            if ((e = hipMemcpyAsync(&isbad, bad, sizeof(int), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
            if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
        }
        if (isbad) return hipErrorInvalidValue;
    }
    unsigned long long fixed[2] = { 0ull, 0ull };
    if ((e = hipMemcpyAsync(fixed, d_fixed, sizeof fixed, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    area[0] = (double)fixed[0]; area[1] = (double)fixed[1];              // in units of 2^-40 of the root's area: only their ratio is used
    *nwide = row[0]; *depth = row[2];
    return hipGetLastError();
}
