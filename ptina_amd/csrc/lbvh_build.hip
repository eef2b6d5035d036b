// lbvh_build.hip -- on-GPU LBVH construction (reference tree/lbvh.py:169-305).
//
// The reference builds in four kernels but round-trips to the host for the sort (np.argsort on
// exported codes, lbvh.py:204-208) and fits the boxes with up to 64 level-synchronous launches, each
// followed by a scalar read-back (lbvh.py:251-261).  Here the whole build stays on the device:
//   centroid_bounds   centroids + their bounding box (wave reduction, ordered-int atomics)
//   morton_keys       30-bit Morton code of the normalised centroid, key = code << 32 | triangle
//                     (unique keys: equal codes cannot corrupt the hierarchy)
//   rocprim radix sort of the 64-bit keys (62 significant bits)
//   hierarchy         one lane per internal node: Karras range + split -> child ids, parents
//   fit_boxes         one lane per leaf walks to the root; the second arrival at a node (agent-scope
//                     atomic counter) merges the two child boxes; also records the tree depth
//   pack              the traversal records of mpt_types.h (snode, fnode, tgeo, tshade) in leaf order
// All arithmetic that decides the tree (centroid, normalisation, quantisation) is the reference's
// f32 sequence with contraction off, so the tree is node-for-node the host build's and the oracle's.

#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "mpt_types.h"
#include "tri_records.h"

#define LB_BLOCK 256

__device__ __forceinline__ int f2ord(float f) {          // order-preserving float -> int
    int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__device__ __forceinline__ const float *vpos(const float *verts, int f, int k) { return verts + ((size_t)f * 3 + k) * 8; }

// lbvh.py:161-165,173-177: centre = (v0 + v1 + v2) / 3; bmin/bmax over centres.  At most 1024 workgroups stride over the
// triangles: six same-line atomics per WAVE of a million triangles took 1.07 ms of the build (profiles/r06_build_kernel_stats_c5_before.csv)
__global__ __launch_bounds__(LB_BLOCK) void centroid_bounds_kernel(const float *__restrict__ verts, int n,
                                                                   float *__restrict__ cen, int *__restrict__ bounds) {
    __shared__ int red[LB_BLOCK / 64][6];
    int lo[3] = { 0x7fffffff, 0x7fffffff, 0x7fffffff }, hi[3] = { (int)0x80000000, (int)0x80000000, (int)0x80000000 };
    for (int f = blockIdx.x * LB_BLOCK + threadIdx.x; f < n; f += gridDim.x * LB_BLOCK) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float c = ((vpos(verts, f, 0)[a] + vpos(verts, f, 1)[a]) + vpos(verts, f, 2)[a]) / 3.0f;
            cen[(size_t)f * 3 + a] = c;
            const int o = f2ord(c);
            lo[a] = min(lo[a], o); hi[a] = max(hi[a], o);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = min(lo[a], __shfl_xor(lo[a], off));
            hi[a] = max(hi[a], __shfl_xor(hi[a], off));
        }
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][a] = lo[a]; red[threadIdx.x >> 6][3 + a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        int v = red[0][threadIdx.x];
        for (int w = 1; w < LB_BLOCK / 64; w++) v = threadIdx.x < 3 ? min(v, red[w][threadIdx.x]) : max(v, red[w][threadIdx.x]);
        if (threadIdx.x < 3) atomicMin(bounds + threadIdx.x, v); else atomicMax(bounds + threadIdx.x, v);
    }
}

__device__ __forceinline__ unsigned expand_bits(unsigned v) {                // lbvh.py:13-17
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__device__ __forceinline__ int quant1024(float x) {                          // clamp(ifloor(v * 1024), 0, 1023), lbvh.py:29
    float f = floorf(x * 1024.0f);
    if (!(f == f) || f < 0.f) return 0;
    if (f > 1023.f) return 1023;
    return (int)f;
}

// lbvh.py:179-183, with the reference's initial bounds +-inf = +-1e6 folded in (lbvh.py:173)
__global__ __launch_bounds__(LB_BLOCK) void morton_keys_kernel(const float *__restrict__ cen, const int *__restrict__ bounds,
                                                               int n, unsigned long long *__restrict__ keys) {
    int f = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (f >= n) return;
    unsigned w[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float bmin = fminf(1e6f, ord2f(bounds[a])), bmax = fmaxf(-1e6f, ord2f(bounds[3 + a]));
        w[a] = expand_bits((unsigned)quant1024((cen[(size_t)f * 3 + a] - bmin) / (bmax - bmin)));
    }
    unsigned code = w[0] * 4 + w[1] * 2 + w[2];
    keys[f] = ((unsigned long long)code << 32) | (unsigned)f;
}

__device__ __forceinline__ int delta(const unsigned long long *key, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(key[i] ^ key[j]));
}

// lbvh.py:212-231 (determineRange :93-146, findSplit :62-89) on unique 64-bit keys
__global__ __launch_bounds__(LB_BLOCK) void hierarchy_kernel(const unsigned long long *__restrict__ key, int n,
                                                             int *__restrict__ child, int *__restrict__ parent,
                                                             int *__restrict__ leaf, int *__restrict__ mc) {
    int i = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (i < n) {
        leaf[i] = (int)(key[i] & 0xffffffffull);
        mc[i] = (int)(key[i] >> 32);
    }
    if (i >= n - 1) return;
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
        if (d < 0) { int tmp = l; l = r; r = tmp; }
    }
    int cp = delta(key, n, l, r);
    int m = l, s = r - l;
    for (;;) {
        s = (s + 1) >> 1;
        int q = m + s;
        if (q < r && delta(key, n, l, q) > cp) m = q;
        if (s <= 1) break;
    }
    int c0 = (m == l) ? m : m + n;
    int c1 = (m + 1 == r) ? m + 1 : m + 1 + n;
    child[(size_t)i * 2 + 0] = c0;
    child[(size_t)i * 2 + 1] = c1;
    parent[c0] = i;                      // parent[] is indexed by node id: leaf slot, or internal + n
    parent[c1] = i;
    if (i == 0) parent[n] = -1;
}

__device__ __forceinline__ void leaf_box(const float *verts, int f, float *lo, float *hi) {   // lbvh.py:155-158
#pragma unroll
    for (int a = 0; a < 3; a++) {
        lo[a] = fminf(fminf(vpos(verts, f, 0)[a], vpos(verts, f, 1)[a]), vpos(verts, f, 2)[a]);
        hi[a] = fmaxf(fmaxf(vpos(verts, f, 0)[a], vpos(verts, f, 1)[a]), vpos(verts, f, 2)[a]);
    }
}

// Bottom-up fit.  Boxes and flags cross workgroups inside one launch, so every word that is handed
// over is written and read with agent-scope atomics and ordered by agent-scope fences around the
// arrival counter (a CU's L1 is never refreshed by another CU's plain stores).
// The arrival word of a node also carries the height of the subtree that arrived first (height << 2 | 1): the second arrival
// knows both children's heights, and the lane that finishes the root knows the tree's depth.  (Rounds 1-5 sent every FIRST
// arrival on up to the root just to count levels -- a chain of ~24 dependent loads per leaf, 3.4 of the build's 39 ms at a
// million triangles.)
__global__ __launch_bounds__(LB_BLOCK) void fit_boxes_kernel(const float *__restrict__ verts, const int *__restrict__ leaf,
                                                             const int *__restrict__ child, const int *__restrict__ parent,
                                                             int n, float *bmin, float *bmax, unsigned *arrive,
                                                             int *__restrict__ depth_out) {
    int slot = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (slot >= n) return;
    int node = slot;
    unsigned height = 0;                 // internal nodes on the longest way down from `node`
    for (;;) {
        int p = parent[node];
        if (p < 0) { atomicMax(depth_out, (int)height); break; }      // `node` is the root: the tree's depth
        // Every word that crosses between lanes here (boxes, arrival words) is written and read with agent-scope ATOMIC
        // accesses, which go to the memory side (sc1) and never sit in a CU's L1 or an XCD's L2.  What the hand-over needs on
        // top is order: this lane's box stores complete before its arrival is counted (s_waitcnt vmcnt(0)), and the loads of
        // the sibling's box are issued after the arrival word came back (they depend on it).  An agent-scope release / acquire
        // FENCE pair gives that too, but on a part with eight L2s it also writes back and invalidates the whole L2 each time,
        // for plain stores this kernel does not have: 3.4 ms at a million triangles against 0.3 (profiles/r06_build_*).
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned prev = __hip_atomic_fetch_add(arrive + p, (height << 2) | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == 0) break;            // first child to arrive: the sibling will finish this node
        __asm__ volatile("" ::: "memory");
        height = max(height, prev >> 2) + 1;
        float lo[3], hi[3];
#pragma unroll
        for (int a = 0; a < 3; a++) { lo[a] = 1e30f; hi[a] = -1e30f; }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            int ch = child[(size_t)p * 2 + k];
            float l[3], h[3];
            if (ch < n) leaf_box(verts, leaf[ch], l, h);
            else {
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    l[a] = __hip_atomic_load(bmin + (size_t)(ch - n) * 3 + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    h[a] = __hip_atomic_load(bmax + (size_t)(ch - n) * 3 + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
            for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], l[a]); hi[a] = fmaxf(hi[a], h[a]); }
        }
#pragma unroll
        for (int a = 0; a < 3; a++) {
            __hip_atomic_store(bmin + (size_t)p * 3 + a, lo[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(bmax + (size_t)p * 3 + a, hi[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        node = p + n;
    }
}

__device__ __forceinline__ float asf(int v) { return __int_as_float(v); }

// node records (mpt_types.h): snode = reference-shaped, fnode = both child boxes of the LBVH itself
__global__ __launch_bounds__(LB_BLOCK) void pack_nodes_kernel(const float *__restrict__ verts, const int *__restrict__ leaf,
                                                              const int *__restrict__ child, const float *__restrict__ bmin,
                                                              const float *__restrict__ bmax, int n,
                                                              MptVec4 *__restrict__ snode, MptVec4 *__restrict__ fnode) {
    int i = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (i >= n - 1) return;
    int c0 = child[(size_t)i * 2], c1 = child[(size_t)i * 2 + 1];
    snode[(size_t)i * 2 + 0] = { bmin[(size_t)i * 3], bmin[(size_t)i * 3 + 1], bmin[(size_t)i * 3 + 2], asf(c0) };
    snode[(size_t)i * 2 + 1] = { bmax[(size_t)i * 3], bmax[(size_t)i * 3 + 1], bmax[(size_t)i * 3 + 2], asf(c1) };
    float l[2][3], h[2][3];
    int id[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        int ch = k ? c1 : c0;
        if (ch < n) { leaf_box(verts, leaf[ch], l[k], h[k]); id[k] = ~ch; }
        else {
#pragma unroll
            for (int a = 0; a < 3; a++) { l[k][a] = bmin[(size_t)(ch - n) * 3 + a]; h[k][a] = bmax[(size_t)(ch - n) * 3 + a]; }
            id[k] = ch - n;
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) fnode[(size_t)i * 4 + a] = { l[0][a], l[1][a], h[0][a], h[1][a] };
    fnode[(size_t)i * 4 + 3] = { asf(id[0]), asf(id[1]), 0.f, 0.f };
}

// per-leaf-slot triangle records: hoisted terms of Face.intersect (geometries.py:120-122,134-136,140)
// in the reference's f32 operation order (this file is compiled with -ffp-contract=off)
__global__ __launch_bounds__(LB_BLOCK) void pack_tris_kernel(const float *__restrict__ verts, const int *__restrict__ mtlids,
                                                             const int *__restrict__ leaf, int n,
                                                             MptVec4 *__restrict__ tgeo, MptVec4 *__restrict__ tshade) {
    int slot = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (slot >= n) return;
    int f = leaf[slot];
    const float *p0 = vpos(verts, f, 0), *p1 = vpos(verts, f, 1), *p2 = vpos(verts, f, 2);
    MptVec4 g[4], sh[4];
    tri_make_tgeo(p0, p1, p2, g);
    tri_make_tshade(p0, p1, p2, asf(mtlids[f]), sh);
#pragma unroll
    for (int k = 0; k < 4; k++) { tgeo[(size_t)slot * 4 + k] = g[k]; tshade[(size_t)slot * 4 + k] = sh[k]; }
}

// ------------------------------------------------------------------ driver
struct MptLbvhBuffers {
    // inputs (device)
    const float *verts;        // [3n][8]
    const int *mtlids;         // [n]
    int n;
    // workspace + outputs (device), all sized by the caller for n
    float *cen;                // [n][3]
    int *bounds;               // [6]
    unsigned long long *keys_in, *keys_out;   // [n]
    void *sort_tmp;
    size_t sort_tmp_bytes;
    int *child;                // [n-1][2]
    int *parent;               // [2n]
    int *leaf, *mc;            // [n]
    float *bmin, *bmax;        // [n-1][3]
    unsigned *arrive;          // [n-1]
    int *depth;                // [1]
    MptVec4 *snode, *fnode, *tgeo, *tshade;
};

MPT_KERNEL_API hipError_t mpt_lbvh_sort_bytes(int n, size_t *bytes) {
    unsigned long long *p = nullptr;
    return rocprim::radix_sort_keys(nullptr, *bytes, p, p, (size_t)n, 0, 62, (hipStream_t)0);
}

MPT_KERNEL_API hipError_t mpt_lbvh_build(const MptLbvhBuffers *b, hipStream_t stream) {
    const int n = b->n;
    if (n <= 0) return hipSuccess;
    const int gl = (n + LB_BLOCK - 1) / LB_BLOCK;
    hipError_t e;
    static const int init_bounds[6] = { 0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000 };
    if ((e = hipMemcpyAsync(b->bounds, init_bounds, sizeof init_bounds, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(centroid_bounds_kernel, dim3(std::min(gl, 1024)), dim3(LB_BLOCK), 0, stream, b->verts, n, b->cen, b->bounds);
    hipLaunchKernelGGL(morton_keys_kernel, dim3(gl), dim3(LB_BLOCK), 0, stream, b->cen, b->bounds, n, b->keys_in);
    size_t tmp = b->sort_tmp_bytes;
    if ((e = rocprim::radix_sort_keys(b->sort_tmp, tmp, b->keys_in, b->keys_out, (size_t)n, 0, 62, stream)) != hipSuccess) return e;
    if (n > 1) {
        if ((e = hipMemsetAsync(b->arrive, 0, (size_t)(n - 1) * sizeof(unsigned), stream)) != hipSuccess) return e;
    }
    if ((e = hipMemsetAsync(b->depth, 0, sizeof(int), stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(hierarchy_kernel, dim3(gl), dim3(LB_BLOCK), 0, stream, b->keys_out, n, b->child, b->parent, b->leaf, b->mc);
    if (n > 1) {
        hipLaunchKernelGGL(fit_boxes_kernel, dim3(gl), dim3(LB_BLOCK), 0, stream, b->verts, b->leaf, b->child, b->parent, n,
                           b->bmin, b->bmax, b->arrive, b->depth);
        hipLaunchKernelGGL(pack_nodes_kernel, dim3(gl), dim3(LB_BLOCK), 0, stream, b->verts, b->leaf, b->child, b->bmin, b->bmax,
                           n, b->snode, b->fnode);
    }
    hipLaunchKernelGGL(pack_tris_kernel, dim3(gl), dim3(LB_BLOCK), 0, stream, b->verts, b->mtlids, b->leaf, n, b->tgeo, b->tshade);
    return hipGetLastError();
}
