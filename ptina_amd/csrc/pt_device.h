// pt_device.h -- gfx950 device functions of the path-trace hot path.
//
// Compiled twice (see Makefile):
//   MPT_STRICT=1, -ffp-contract=off : the reference's traversal order (tree/lbvh.py:314-347),
//       IEEE divide/sqrt, libm-grade sin/cos/pow/log.  Exists to check the logic against the CPU
//       oracle to the last few ulps.
//   MPT_STRICT=0, -ffp-contract=fast : the production path.  Same sampling, shading and hit
//       predicates; the BVH is walked near-child-first with depth culling, shadow rays stop at the
//       first occluder, divisions become v_rcp_f32, sin/cos(2 pi p) become v_sin/v_cos on turns.
//       Differs from strict only by rounding (and by which of two equal-depth hits wins).
//
// Citations are file:line into the reference tree.
#pragma once

#include <hip/hip_runtime.h>
#include "mpt_types.h"

#ifndef MPT_STRICT
#define MPT_STRICT 0
#endif

#define MPT_EPS 1e-6f          // common.py:32
#define MPT_INF 1e6f           // common.py:33
#define MPT_PI 3.14159265358979323846f
#define MPT_TAU 6.28318530717958647692f
#define MPT_INV_PI 0.31830988618379067154f

#define DEV __device__ __forceinline__

// ---------------------------------------------------------------- math shims
#if MPT_STRICT
DEV float m_div(float a, float b) { return a / b; }
DEV float m_rcp(float a) { return 1.0f / a; }
DEV float m_sqrt(float a) { return sqrtf(a); }
DEV float m_rnorm(float sumsq) { return 1.0f / sqrtf(sumsq); }            // Matrix.normalized(): 1 / norm
DEV void m_sincos_turn(float p, float *s, float *c) { *c = cosf(p * MPT_TAU); *s = sinf(p * MPT_TAU); }
DEV float m_pow5(float x) { return powf(x, 5.0f); }
DEV float m_pow(float a, float b) { return powf(a, b); }
DEV float m_log(float a) { return logf(a); }
#else
DEV float m_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
DEV float m_rcp(float a) { return __builtin_amdgcn_rcpf(a); }
DEV float m_sqrt(float a) { return __builtin_amdgcn_sqrtf(a); }
DEV float m_rnorm(float sumsq) { return __builtin_amdgcn_rsqf(sumsq); }
// v_sin_f32 / v_cos_f32 take their argument in turns: spherical()'s p * tau needs no range reduction
DEV void m_sincos_turn(float p, float *s, float *c) { *c = __builtin_amdgcn_cosf(p); *s = __builtin_amdgcn_sinf(p); }
DEV float m_pow5(float x) { float x2 = x * x; return x2 * x2 * x; }
DEV float m_pow(float a, float b) { return __builtin_amdgcn_exp2f(b * __builtin_amdgcn_logf(a)); }
DEV float m_log(float a) { return __builtin_amdgcn_logf(a) * 0.69314718055994530942f; }
#endif

struct V3 { float x, y, z; };
DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV V3 v3s(float s) { return v3(s, s, s); }
DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
DEV V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
DEV V3 vdivs(V3 a, float s) { return v3(m_div(a.x, s), m_div(a.y, s), m_div(a.z, s)); }
DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEV float norm_sqr(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
DEV V3 normalized(V3 a) { return a * m_rnorm(norm_sqr(a)); }
// the same with the sum of squares left unfused in every build: the path direction is normalised where a ray starts, a piece of
// code the production kernels hold in more than one inlined copy, and -ffp-contract=fast would fuse each copy as it pleases --
// the films of two kernels (or of one kernel with and without an option) would then differ in last bits for no arithmetic reason
DEV V3 normalized_unfused(V3 a) {
#pragma clang fp contract(off)
    const float xx = a.x * a.x, yy = a.y * a.y, zz = a.z * a.z;
    const float s = xx + yy + zz;
    return a * m_rnorm(s);
}
#if MPT_STRICT
DEV float vavg(V3 a) { return m_div(a.x + a.y + a.z, 3.0f); }                 // common.py:73-77
#else
DEV float vavg(V3 a) { return (a.x + a.y + a.z) * 0.33333334f; }              // a constant, not a v_rcp_f32 per call
#endif
DEV bool any_gt0(V3 a) { return a.x > 0.0f || a.y > 0.0f || a.z > 0.0f; }
DEV bool any_ne0(V3 a) { return a.x != 0.0f || a.y != 0.0f || a.z != 0.0f; }
DEV float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }   // common.py:163-165
DEV float dot_or_zero(V3 a, V3 b) { return fmaxf(0.0f, dot(a, b)); }          // common.py:178-180
DEV float lerpf(float f, float src, float dst) { return src * (1.0f - f) + dst * f; }   // common.py:269-271
DEV V3 lerpv(float f, V3 src, V3 dst) { return src * (1.0f - f) + dst * f; }
DEV V3 reflectv(V3 I, V3 N) { return I - N * (2.0f * dot(N, I)); }            // common.py:247-249

DEV bool refractv(V3 I, V3 N, float eta, V3 *T) {                            // common.py:252-260
    bool has_r = false;
    *T = v3s(0.0f);
    float NoI = dot(N, I);
    float discr = 1.0f - eta * eta * (1.0f - NoI * NoI);
    if (discr > 0.0f) {
        has_r = true;
        *T = normalized(I * eta - N * (eta * NoI + m_sqrt(discr)));
    }
    return has_r;
}

DEV V3 spherical(float h, float p) {                                         // common.py:221-225
    float s, c;
    m_sincos_turn(p, &s, &c);
    float r = m_sqrt(fmaxf(0.0f, 1.0f - h * h));
    return v3(r * c, r * s, h);
}

// tanspace(nrm) @ v, common.py:213-217.  The basis depends on the normal alone: Disney.bounce builds it once
// for whichever lobe is sampled (the reference's three branches each call tanspace(normal) -- same values)
struct TanSpace { V3 tan, bitan, nrm; };
DEV TanSpace tanspace(V3 nrm) {
    TanSpace t;
    V3 up = v3(233.0f, 666.0f, 512.0f);
    t.bitan = normalized(cross(nrm, up));
    t.tan = cross(t.bitan, nrm);
    t.nrm = nrm;
    return t;
}
DEV V3 tanspace_mul(const TanSpace &t, V3 v) {
    return v3(t.tan.x * v.x + t.bitan.x * v.y + t.nrm.x * v.z,
              t.tan.y * v.x + t.bitan.y * v.y + t.nrm.y * v.z,
              t.tan.z * v.x + t.bitan.z * v.y + t.nrm.z * v.z);
}

// ---------------------------------------------------------------- sampling
DEV int wanghash(int x) {                                                    // sampling/__init__.py:9-16
    unsigned v = (unsigned)x;
    v = (v ^ 61u) ^ (v >> 16);
    v *= 9u;
    v ^= v << 4;
    v *= 0x27d4eb2du;
    v ^= v >> 15;
    return (int)v;
}
DEV int wanghash2(int x, int y) { return wanghash(y ^ wanghash(x)); }        // sampling/__init__.py:20-23

struct Rng { const float *P; int i; int dim; };                              // SobolSampler.Proxy, sobol.py:113-125
DEV float rng_random(Rng &r) {
    int k = r.i % r.dim;
    if (k < 0) k += r.dim;                                                   // Python floor-mod
    float v = r.P[k];
    r.i = (int)((unsigned)r.i + 1u);                                         // i32 wrap
    return v;
}
DEV V3 random3(Rng &r) { float a = rng_random(r), b = rng_random(r), c = rng_random(r); return v3(a, b, c); }

// ---------------------------------------------------------------- counters
struct Cnt { unsigned rays, n_box, n_tri, n_shade, n_draws, bounces, n_node, samples, it_node, it_leaf, it_shade, it_new;
             // pooled kernel (render_pool.h): bounces a tracer did itself because the shade pool was full; shader batches and the
             // requests in them; primary-ray batches; tracer / shader polls with nothing to do; trips of a tracer to the pools;
             // rays taken by tracers
             unsigned pl_local, pl_batches, pl_batch_lanes, pl_prim, pl_tidle, pl_sidle, pl_trips, pl_taken; };

// ---------------------------------------------------------------- geometry
struct Hit { int hit; float depth; int index; float u, v; };

DEV V3 ld3(const MptVec4 &a) { return v3(a.x, a.y, a.z); }

// Face.intersect, geometries.py:118-148, with the ray-independent terms hoisted into tgeo
DEV bool tri_test(MptVec4 g0, MptVec4 g1, MptVec4 g2, MptVec4 g3, V3 ro, V3 rd, float *depth, float *s_, float *t_) {
    V3 v0 = ld3(g0), u = ld3(g1), v = ld3(g2), norm = ld3(g3);
    float D = g0.w, uu = g1.w, uv = g2.w, vv = g3.w;
    float b = dot(norm, rd);
#if !MPT_STRICT
    // Production build: the whole test straight through and the reference's four conditions combined at the end.
    // In a wave the nested form runs every level anyway (some lane always gets that far), each level behind its
    // own exec-mask juggling and its own wait for the record words it needs; the flat form is one wait and no
    // branches.  Lanes that fail an early condition compute garbage (infinities, NaNs) that the flags discard.
    {
        V3 w0 = ro - v0;
        float a = -dot(norm, w0);
        float r = m_div(a, b);
        V3 ip = ro + rd * r;
        V3 w = ip - v0;
        float wu = dot(w, u);
        float wv = dot(w, v);
        float rD = m_rcp(D);
        float s = (uv * wv - vv * wu) * rD;
        float t = (uv * wu - uu * wv) * rD;
        bool ok = fabsf(b) >= MPT_EPS && r > 0.0f && 0.0f <= s && s <= 1.0f && 0.0f <= t && s + t <= 1.0f;
        *depth = r; *s_ = s; *t_ = t;
        return ok;
    }
#else
    bool hit = false;
    if (fabsf(b) >= MPT_EPS) {
        V3 w0 = ro - v0;
        float a = -dot(norm, w0);
        float r = m_div(a, b);
        if (r > 0.0f) {
            V3 ip = ro + rd * r;
            V3 w = ip - v0;
            float wu = dot(w, u);
            float wv = dot(w, v);
            float s = m_div(uv * wv - vv * wu, D);
            float t = m_div(uv * wu - uu * wv, D);
            if (0.0f <= s && s <= 1.0f && 0.0f <= t && s + t <= 1.0f) {
                *depth = r; *s_ = s; *t_ = t;
                hit = true;
            }
        }
    }
    return hit;
#endif
}

#if !MPT_STRICT
// Production build: the same test from a 48-byte record {n, v0.x}{a, v0.y}{c, v0.z} (tfast, derived from tgeo on
// the device).  a = (uv v - vv u) / D and c = (uv u - uu v) / D are the dual edge vectors: the reference's
// s = (uv wv - vv wu) / D is w . a and t = (uv wu - uu wv) / D is w . c, and with w = (o - v0) + r d they are
// (o - v0) . a + r (d . a) -- the hit point, its offset from v0 and the two divisions by D are never formed.
// 26 VALU instructions and one reciprocal instead of 36 and two; three 16-byte reads per triangle instead of four.
DEV bool tri_test_fast(MptVec4 g0, MptVec4 g1, MptVec4 g2, V3 ro, V3 rd, float *depth, float *s_, float *t_) {
    const V3 n = ld3(g0), a = ld3(g1), c = ld3(g2);
    const V3 w0 = ro - v3(g0.w, g1.w, g2.w);
    const float b = dot(n, rd);
    const float r = m_div(-dot(n, w0), b);
    const float s = __builtin_fmaf(r, dot(a, rd), dot(a, w0));
    const float t = __builtin_fmaf(r, dot(c, rd), dot(c, w0));
    *depth = r; *s_ = s; *t_ = t;
    // The reference's conditions (geometries.py:144-146) are 0 <= s <= 1, 0 <= t, s + t <= 1.  "s <= 1" is implied by the others -- t >= 0
    // gives s + t >= s, rounding is monotonic, so fl(s + t) >= s, and fl(s + t) <= 1 then says s <= 1; a NaN fails "0 <= s" either way -- and
    // is left out: the same decisions with one compare fewer.  (Compares are not cheap on this chip: 3.3 cycles per wave64 instruction per
    // SIMD against 2.3 for an FMA, tools/microbench; MI355X, same box, three alternations: 2.617 -> 2.593 ms per launch, -0.9 %.)
    return fabsf(b) >= MPT_EPS && r > 0.0f && 0.0f <= s && 0.0f <= t && s + t <= 1.0f;
}
#endif

// Box.intersect, geometries.py:24-46
DEV bool box_strict(V3 lo, V3 hi, V3 ro, V3 rd, float *near_ = nullptr, float *far_ = nullptr) {
    float nearv = 0.0f, farv = MPT_INF;
    bool hit = true;
    const float lo_[3] = { lo.x, lo.y, lo.z }, hi_[3] = { hi.x, hi.y, hi.z };
    const float o_[3] = { ro.x, ro.y, ro.z }, d_[3] = { rd.x, rd.y, rd.z };
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (fabsf(d_[i]) < MPT_EPS) {
            if (o_[i] < lo_[i] || o_[i] > hi_[i]) hit = false;
        } else {
            float i1 = (lo_[i] - o_[i]) / d_[i];
            float i2 = (hi_[i] - o_[i]) / d_[i];
            if (i1 > i2) { float t = i1; i1 = i2; i2 = t; }
            farv = fminf(farv, i2);
            nearv = fmaxf(nearv, i1);
            if (nearv > farv) hit = false;
        }
    }
    if (near_) *near_ = nearv;
    if (far_) *far_ = farv;
    return hit;
}

// per-lane LIFO in LDS: element [level][thread], so a wave's push/pop touches 64 consecutive
// dwords (conflict-free); replaces GlobalStack's [thread][level] rows in global memory, stack.py:10-60
struct Stack {
    static constexpr int SENTINEL = (int)0x80000000;   // bottom-of-stack marker (never a node or leaf id)
    static constexpr bool ONE_TEST = false;
    static constexpr int PLANE_OFF = 0;                // (the binary gather kernel takes min / max of both planes)
    int *base;                 // &lds[threadIdx.x]
    int sp;
    DEV void push(int v) { base[sp * MPT_BLOCK] = v; sp++; }
    DEV int pop() { sp--; return base[sp * MPT_BLOCK]; }
    static constexpr bool PEEK = true;                 // the entry a pop would return can be read ahead of the decision
    static constexpr bool SP_ADDR = false, ODD_IDS = false, T_SCALED = false;
    static constexpr int SP_STEP = 1;
    DEV int peek(int at) const { return base[at * MPT_BLOCK]; }
};

// extra NODE steps behind one scheduling decision of trace_stream (render_kernel.hip), per kind of scene (SCENE::NODE_REP)
#ifndef MPT_NODE_REP
#define MPT_NODE_REP 2        // extra NODE steps per decision (MI355X: 0 / 1 / 2 / 3 -> 4.03 / 3.85 / 3.72 / 3.72 ms with one extra LEAF step)
#endif
#ifndef MPT_WIDE_REP
#define MPT_WIDE_REP 0        // extra 4-wide NODE steps per decision (gather kernels)
#endif
#ifndef MPT_LEAF_REP
#define MPT_LEAF_REP 1        // extra LEAF steps per decision, LDS-resident kernels (0 / 1 / 2 -> 3.83 / 3.72 / 3.77 ms with two extra NODE steps)
#endif
#ifndef MPT_WIDE_LEAF_REP
#define MPT_WIDE_LEAF_REP 0   // ... gather kernels: MI355X C5 822 -> 853 Msamples/s with none instead of one, C4 1606 -> 1597 (alternated twice)
#endif
#ifndef MPT_LDS4_REP
#define MPT_LDS4_REP 1        // ... of the LDS-resident 4-wide kernel
#endif

// scene records served from HBM/L2 through the vector L1 (any scene size)
struct GlobalScene {
    static constexpr int LEAF_REP = MPT_WIDE_LEAF_REP;
    static constexpr bool AVOID_IN_LEAF = false;
    static constexpr int SHADE_MIN = 0;                // (render_kernel.hip trace_stream: lanes SHADE waits for)
    static constexpr int NODE_REP = MPT_NODE_REP;
    static constexpr bool WIDE = false, QUANT = false, SIGNED_PLANES = false, LDS_MATS = false, OCT = false;
    static constexpr bool ODD_IDS = false, T_SCALED = false;
    const MptVec4 *fnode, *tgeo;
    int soa_n;                 // node count, for the layout A/B build below
    DEV void node(int i, MptVec4 &a, MptVec4 &b, MptVec4 &c, MptVec4 &d) const {
#if MPT_X_NODE_SOA
        // layout A/B build only (option "node_soa" = 1 hands it the transposed arrays): four arrays of float4
        const MptVec4 *nd = fnode + i;
        a = nd[0]; b = nd[soa_n]; c = nd[2 * (size_t)soa_n]; d = nd[3 * (size_t)soa_n];
#else
        const MptVec4 *nd = fnode + (size_t)i * 4;
        a = nd[0]; b = nd[1]; c = nd[2]; d = nd[3];
#endif
    }
    DEV void tri(int slot, MptVec4 &g0, MptVec4 &g1, MptVec4 &g2) const {       // tfast: 48-byte records
        const MptVec4 *g = tgeo + (size_t)slot * 3;
        g0 = g[0]; g1 = g[1]; g2 = g[2];
    }
};

// The triangle a ray left from is filtered by the LEAF step (one compare) instead of by the 4-wide NODE step (four compares and
// four mask merges): its own leaf is then visited once per ray that starts on a surface -- three gathers -- and it is still
// cheaper: MI355X C4 1583 -> 1607, C5 804 -> 824 Msamples/s (alternated twice)
#ifndef MPT_WIDE_AVOID_IN_LEAF
#define MPT_WIDE_AVOID_IN_LEAF 1
#endif
// 4-wide nodes gathered from HBM / L2 / Infinity Cache (scenes that do not fit LDS): a traversal step is one
// 128-B record and four box tests, and a ray makes half as many DEPENDENT fetches as through the binary tree --
// those fetches, not their bytes, are what bounds the big scenes (measured: binary16 boxes at half the bytes
// bought 3-7 %)
struct WideScene {
    static constexpr int LEAF_REP = MPT_WIDE_LEAF_REP;
    static constexpr bool AVOID_IN_LEAF = MPT_WIDE_AVOID_IN_LEAF != 0;
    static constexpr int SHADE_MIN = 0;                // (render_kernel.hip trace_stream: lanes SHADE waits for)
    static constexpr int NODE_REP = MPT_WIDE_REP;
    static constexpr bool WIDE = true, QUANT = false, SIGNED_PLANES = false, LDS_MATS = false, OCT = false;
    static constexpr bool ODD_IDS = false, T_SCALED = false;
    const MptVec4 *wnode, *tgeo;
    // entry (n*) and exit (f*) planes of the four children, picked by the ray's direction signs: o* is 0 for a ray
    // going up the axis and 16 (bytes: the next float4) for one going down -- still seven dwordx4 gathers
    DEV void node4(int i, int ox, int oy, int oz, MptVec4 &nx, MptVec4 &fx, MptVec4 &ny, MptVec4 &fy, MptVec4 &nz,
                   MptVec4 &fz, MptVec4 &id) const {
        // 32-bit byte offsets from the (scalar) array base: one v_add_u32 per gather instead of a 64-bit add
        // (the host keeps the array below 2 GiB: mpt_build_tree)
        const char *base = (const char *)wnode;
        const unsigned o = (unsigned)i << 7;
        nx = *(const MptVec4 *)(base + (o + (unsigned)ox));      fx = *(const MptVec4 *)(base + (o + (unsigned)(ox ^ 16)));
        ny = *(const MptVec4 *)(base + 32 + (o + (unsigned)oy)); fy = *(const MptVec4 *)(base + 32 + (o + (unsigned)(oy ^ 16)));
        nz = *(const MptVec4 *)(base + 64 + (o + (unsigned)oz)); fz = *(const MptVec4 *)(base + 64 + (o + (unsigned)(oz ^ 16)));
        id = *(const MptVec4 *)(base + 96 + o);
#if MPT_X_DUP_NODE_LOADS
        // sensitivity A/B (same film): three of the seven gathers issued twice (ordinary cached loads through an
        // offset the compiler cannot see through) -- what do the gathers themselves cost?
        // (1: as 16-byte gathers; 2: as 4-byte gathers of their first word)
        {
            unsigned o2 = o;
            asm volatile("" : "+v"(o2));
#if MPT_X_DUP_NODE_LOADS == 1
            const MptVec4 a = *(const MptVec4 *)(base + (o2 + (unsigned)(ox ^ 16)));
            const MptVec4 b = *(const MptVec4 *)(base + 32 + (o2 + (unsigned)(oy ^ 16)));
            const MptVec4 c = *(const MptVec4 *)(base + 64 + (o2 + (unsigned)(oz ^ 16)));
            fx.x = a.x == fx.x ? fx.x : a.x; fx.y = a.y == fx.y ? fx.y : a.y; fx.z = a.z == fx.z ? fx.z : a.z; fx.w = a.w == fx.w ? fx.w : a.w;
            fy.x = b.x == fy.x ? fy.x : b.x; fy.y = b.y == fy.y ? fy.y : b.y; fy.z = b.z == fy.z ? fy.z : b.z; fy.w = b.w == fy.w ? fy.w : b.w;
            fz.x = c.x == fz.x ? fz.x : c.x; fz.y = c.y == fz.y ? fz.y : c.y; fz.z = c.z == fz.z ? fz.z : c.z; fz.w = c.w == fz.w ? fz.w : c.w;
#else
            const float a = *(const float *)(base + (o2 + (unsigned)(ox ^ 16)));
            const float b = *(const float *)(base + 32 + (o2 + (unsigned)(oy ^ 16)));
            const float c = *(const float *)(base + 64 + (o2 + (unsigned)(oz ^ 16)));
            fx.x = a == fx.x ? fx.x : a; fy.x = b == fy.x ? fy.x : b; fz.x = c == fz.x ? fz.x : c;
#endif
        }
#endif
    }
    DEV void tri(int slot, MptVec4 &g0, MptVec4 &g1, MptVec4 &g2) const {       // tfast: 48-byte records
        const MptVec4 *g = tgeo + (size_t)slot * 3;
        g0 = g[0]; g1 = g[1]; g2 = g[2];
    }
};

// The same 4-wide nodes in 64 bytes: the child boxes as 8-bit offsets from the node's own box (rounded outwards by
// the builder), so a step is FOUR 16-B gathers instead of seven.  Measured on MI355X with duplicated gathers: every
// extra gather instruction per step costs these kernels 7-9 % whatever its width (4 B or 16 B) -- what they wait
// for is the number of divergent gathers, not bytes -- and the 36 extra VALU instructions of the decode are free
// at 34-43 % issue utilisation.
struct QuantScene {
    static constexpr int LEAF_REP = MPT_WIDE_LEAF_REP;
    static constexpr bool AVOID_IN_LEAF = MPT_WIDE_AVOID_IN_LEAF != 0;
#ifndef MPT_WIDE_SHADE_MIN
#define MPT_WIDE_SHADE_MIN 0
#endif
    static constexpr int SHADE_MIN = MPT_WIDE_SHADE_MIN;   // (render_kernel.hip trace_stream: lanes SHADE waits for; 0: it never waits)
    static constexpr int NODE_REP = MPT_WIDE_REP;      // extra NODE steps per scheduling decision
    static constexpr bool WIDE = true, QUANT = true, SIGNED_PLANES = false, LDS_MATS = false, OCT = false;
    static constexpr bool ODD_IDS = false, T_SCALED = false;
    const MptVec4 *qnode, *tgeo;
#if MPT_X_TOPCACHE
    // A/B build (-DMPT_X_TOPCACHE=N): the first N records (the 4-wide nodes are numbered breadth first: the top of the tree) are kept
    // in the workgroup's LDS; a lane whose node is among them reads it there
    typedef float top_f4 __attribute__((ext_vector_type(4)));
    __attribute__((address_space(3))) const top_f4 *top;
#endif
    DEV void node4q(int i, MptVec4 &a, MptVec4 &b, MptVec4 &c, MptVec4 &id) const {
        const char *base = (const char *)qnode;
        const unsigned o = (unsigned)i << 6;
#if MPT_X_TOPCACHE
        if (i < MPT_X_TOPCACHE) {
            const top_f4 v0 = top[i * 4], v1 = top[i * 4 + 1], v2 = top[i * 4 + 2], v3 = top[i * 4 + 3];
            a = { v0.x, v0.y, v0.z, v0.w }; b = { v1.x, v1.y, v1.z, v1.w }; c = { v2.x, v2.y, v2.z, v2.w }; id = { v3.x, v3.y, v3.z, v3.w };
            return;
        }
#endif
        a = *(const MptVec4 *)(base + o); b = *(const MptVec4 *)(base + 16 + o);
        c = *(const MptVec4 *)(base + 32 + o); id = *(const MptVec4 *)(base + 48 + o);
#if MPT_X_DUP_QNODE
        // sensitivity A/B (same film): MPT_X_DUP_QNODE extra 4-byte gathers per step -- what is one gather more or less worth?
        {
            unsigned o2 = o;
            asm volatile("" : "+v"(o2));
            const float e0 = *(const float *)(base + 4 + o2);
            a.y = e0 == a.y ? a.y : e0;
#if MPT_X_DUP_QNODE > 1
            unsigned o3 = o;
            asm volatile("" : "+v"(o3));
            const float e1 = *(const float *)(base + 20 + o3);
            b.y = e1 == b.y ? b.y : e1;
#endif
        }
#endif
    }
    DEV void tri(int slot, MptVec4 &g0, MptVec4 &g1, MptVec4 &g2) const {       // tfast: 48-byte records
        const MptVec4 *g = tgeo + (size_t)slot * 3;
        g0 = g[0]; g1 = g[1]; g2 = g[2];
    }
};

// LIFO of the wide traversal: a step may push three entries, so the worst case is 3 x depth; the first CAP
// levels live in LDS like Stack's, the (rare) rest in a per-lane strip of global memory
struct SpillStack {
    static constexpr int SENTINEL = (int)0x80000000;
    static constexpr bool ONE_TEST = false;
    static constexpr int PLANE_OFF = 0;                // (the 4-wide steps pick the entry planes by the sign of L.inv themselves: no per-ray offsets to carry)
#ifndef MPT_X_SPILL_CAP
#define MPT_X_SPILL_CAP 24     // (a test build sets it to a handful of levels so that every ray uses the global strip)
#endif
    static constexpr int CAP = MPT_X_SPILL_CAP, SPILL = 128 - CAP;    // 24 levels x 256 lanes x 4 B = 24 KiB of LDS: the five workgroups per CU the registers allow (and a sixth)
    static constexpr int STRIDE = MPT_BLOCK;           // entries from one level of a lane's stack to the next
    typedef int entry_t;
    static constexpr bool NO_SPILL = false;
    int *base;                 // &lds[threadIdx.x]
    int *spill;                // the launch's strips (wave-uniform: stays in scalar registers) ...
    unsigned lane_off;         // ... and this lane's first entry in them: one register instead of a 64-bit pointer per lane
    int sp;
    DEV void push(int v) {
        if (sp < CAP) base[sp * MPT_BLOCK] = v;
        else spill[lane_off + (unsigned)(sp - CAP)] = v;
        sp++;
    }
    DEV int pop() {
        sp--;
        return sp < CAP ? base[sp * MPT_BLOCK] : spill[lane_off + (unsigned)(sp - CAP)];
    }
    static constexpr bool PEEK = false;
    DEV int peek(int) const { return 0; }
    static constexpr bool SP_ADDR = false, ODD_IDS = false, T_SCALED = false;
    static constexpr int SP_STEP = 1;
};

// 8-wide nodes with octant-ordered child slots and 8-bit boxes (oct_build.cpp): 80-byte records, FIVE 16-B gathers per step for
// eight box tests; the order the children are met in is slot XOR the ray's direction octant -- no sort -- and a node leaves at most
// two stack entries behind (its other internal hits, its other leaf hits), each a (base | mask, slots to go) pair
struct OctScene {
    static constexpr int LEAF_REP = MPT_WIDE_LEAF_REP;
    static constexpr bool AVOID_IN_LEAF = false;
    static constexpr int SHADE_MIN = 0;
    static constexpr int NODE_REP = 0;
    static constexpr bool WIDE = true, QUANT = true, SIGNED_PLANES = false, LDS_MATS = false, OCT = true;
    static constexpr bool ODD_IDS = false, T_SCALED = false;
    const MptVec4 *onode, *tgeo;                       // tgeo: tfast8, the 48-byte records in the 8-wide tree's leaf order
    DEV void node8(int i, MptVec4 &h0, MptVec4 &h1, MptVec4 &px, MptVec4 &py, MptVec4 &pz) const {
        const char *base = (const char *)onode;
        const unsigned o = (unsigned)i * 80u;            // (the host keeps the array below 2 GiB)
        h0 = *(const MptVec4 *)(base + o); h1 = *(const MptVec4 *)(base + 16 + o);
        px = *(const MptVec4 *)(base + 32 + o); py = *(const MptVec4 *)(base + 48 + o); pz = *(const MptVec4 *)(base + 64 + o);
    }
    DEV void tri(int slot, MptVec4 &g0, MptVec4 &g1, MptVec4 &g2) const {
        const MptVec4 *g = tgeo + (size_t)slot * 3;
        g0 = g[0]; g1 = g[1]; g2 = g[2];
    }
};

// LIFO of the 8-wide traversal: entries are PAIRS (a, b) -- a = first child or triangle | mask of the node's slots of that kind
// << 24, b = the slots still to visit (internal: in met-order positions; leaves: bit 31 set) -- two words per level in LDS,
// [level][lane] each, the (rare) levels beyond CAP in a per-lane strip of global memory like SpillStack's
struct OctStack {
    static constexpr int SENTINEL = 0x40000000;        // the b word of the bottom entry: traversal over
    static constexpr bool ONE_TEST = false;
    static constexpr int PLANE_OFF = 0;
    static constexpr bool PEEK = false;
    static constexpr bool SP_ADDR = false, ODD_IDS = false, T_SCALED = false;
    static constexpr int SP_STEP = 1;
#ifndef MPT_OCT_CAP
#define MPT_OCT_CAP 14
#endif
    static constexpr int CAP = MPT_OCT_CAP, SPILL = 64 - CAP;   // 2 x 14 levels x 256 lanes x 4 B = 28 KiB of LDS: five workgroups per CU
    int *base;                 // &lds[threadIdx.x]
    int *spill;
    unsigned lane_off;         // this lane's first word in the strips (2 x SPILL words per lane)
    int sp;
    DEV void put(int at, int a, int b) {
        if (at < CAP) { base[at * MPT_BLOCK] = a; base[(CAP + at) * MPT_BLOCK] = b; }
        else { spill[lane_off + 2u * (unsigned)(at - CAP)] = a; spill[lane_off + 2u * (unsigned)(at - CAP) + 1u] = b; }
    }
    DEV void get(int at, int &a, int &b) const {
        if (at < CAP) { a = base[at * MPT_BLOCK]; b = base[(CAP + at) * MPT_BLOCK]; }
        else { a = spill[lane_off + 2u * (unsigned)(at - CAP)]; b = spill[lane_off + 2u * (unsigned)(at - CAP) + 1u]; }
    }
    DEV void setb(int at, int b) {
        if (at < CAP) base[(CAP + at) * MPT_BLOCK] = b;
        else spill[lane_off + 2u * (unsigned)(at - CAP) + 1u] = b;
    }
    DEV void push(int v) { put(sp, 0, v); sp++; }      // (lane_start_ray: the sentinel)
    DEV int pop() { sp--; return 0; }
    DEV int peek(int) const { return 0; }
};

// scene records resident in the CU's LDS (small scenes): ds_read_b128 instead of divergent
// global gathers -- one copy per CU, shared by the 16 waves of a 1024-lane workgroup
typedef float mpt_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const mpt_f4 *LdsVec4Ptr;
typedef __attribute__((address_space(3))) short *LdsShortPtr;
typedef __attribute__((address_space(3))) unsigned short *LdsUShortPtr;

DEV MptVec4 lds_ld(LdsVec4Ptr q) { mpt_f4 v = *q; MptVec4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r; }

typedef float mpt_f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const mpt_f2 *LdsVec2Ptr;
typedef __attribute__((address_space(3))) const char *LdsBytePtr;

typedef __attribute__((address_space(3))) const unsigned char *LdsU8Ptr;
#define MPT_LDS_MAT_VEC4 6      // float4 of a material record kept in LDS: p[0..15] and the derived terms d[0..7]

// PRESCALED: the internal-node ids of the LDS copy (in the records and therefore on the stack) are the node's byte offset / 8, so
// a NODE step forms its record address with one shift instead of a 32-bit integer multiply (quarter rate: four issue slots of the
// ~50 a step has).  render_kernel_lds scales the ids while it copies the records (stride 72 -> id x 9 <= 32767: scenes that fit LDS do).
template <bool PRESCALED>
struct LdsSceneT {
#ifndef MPT_SHADE_MIN_LDS
#define MPT_SHADE_MIN_LDS 24
#endif
    static constexpr bool PRESCALED_IDS = PRESCALED;
    static constexpr int LEAF_REP = MPT_LEAF_REP;
    static constexpr bool AVOID_IN_LEAF = false;
    static constexpr int NODE_REP = MPT_NODE_REP;
    static constexpr int SHADE_MIN = MPT_SHADE_MIN_LDS; // SHADE waits until this many lanes want it (render_kernel.hip trace_stream)
    static constexpr bool WIDE = false, QUANT = false, SIGNED_PLANES = true, LDS_MATS = true, OCT = false;
    static constexpr bool ODD_IDS = false, T_SCALED = false;
    LdsVec4Ptr fnode, tgeo;
    // The material records (parameters + derived terms, 96 B each, the default material last) and one byte per
    // leaf slot naming the record: SHADE reads its material out of LDS while the shading record of the triangle is
    // still on its way from L2, instead of gathering it from L2 after that record has arrived (it holds the id).
    LdsVec4Ptr mats;
    LdsU8Ptr mtl;
    int nstride;               // bytes from one node record to the next (MPT_LDS_NODE_STRIDE; 64 where 72 does not fit)
    int mat_last, mat_default; // LDS record mat_last is the default material = record mat_default of the global table
                               // (the pooled kernel keeps only the records the model uses; else both are default_mtl)
    // The slab planes of both children picked by the ray's direction signs instead of by min / max: a node
    // record holds {lo, lo, hi, hi} (child 0, child 1) per axis, so the entry planes of an axis are the 8 bytes at
    // offset 0 for a ray going up that axis and at offset 8 for one going down, and the exit planes are the
    // other 8 -- the offsets o* (0 or 8) are per-ray constants.  Seven ds_read_b64 instead of four ds_read_b128,
    // and 12 v_min / v_max fewer per step.
    DEV void node_planes(int i, int ox, int oy, int oz, mpt_f2 &nx, mpt_f2 &fx, mpt_f2 &ny, mpt_f2 &fy,
                         mpt_f2 &nz, mpt_f2 &fz, mpt_f2 &ids) const {
        LdsBytePtr nd = (LdsBytePtr)fnode + (PRESCALED ? (i << 3) : i * nstride);
        LdsBytePtr ax = nd + ox, ay = nd + oy, az = nd + oz;
        nx = *(LdsVec2Ptr)ax;        fx = *(LdsVec2Ptr)(nd + (ox ^ 8));
        ny = *(LdsVec2Ptr)(ay + 16); fy = *(LdsVec2Ptr)(nd + 16 + (oy ^ 8));
        nz = *(LdsVec2Ptr)(az + 32); fz = *(LdsVec2Ptr)(nd + 32 + (oz ^ 8));
        ids = *(LdsVec2Ptr)(nd + 48);
    }
    DEV void node(int i, MptVec4 &a, MptVec4 &b, MptVec4 &c, MptVec4 &d) const {
        LdsVec4Ptr nd = fnode + i * 4;
        a = lds_ld(nd); b = lds_ld(nd + 1); c = lds_ld(nd + 2); d = lds_ld(nd + 3);
    }
    DEV void tri(int slot, MptVec4 &g0, MptVec4 &g1, MptVec4 &g2) const {
        // (`tgeo + slot * 3` is compiled into a 64-bit multiply-add, v_mad_u64_u32 -- gfx950 has no 32-bit integer mad -- for a 32-bit LDS
        //  address; the 24-bit multiply is one instruction, v_mad_u32_u24: -0.25 % per launch.  Slots are below 2^15)
        LdsVec4Ptr g = (LdsVec4Ptr)((LdsBytePtr)tgeo + __umul24((unsigned)slot, 48u));
        g0 = lds_ld(g); g1 = lds_ld(g + 1); g2 = lds_ld(g + 2);
    }
};

typedef LdsSceneT<false> LdsScene;       // (the pooled A/B kernel: a run-time stride)
#ifndef MPT_LDS_PRESCALED
#define MPT_LDS_PRESCALED 1
#endif
typedef LdsSceneT<MPT_LDS_PRESCALED != 0> LdsSceneP;

// 16-bit LIFO for the LDS-resident kernel (node ids fit in int16 there), [level][lane of 1024]
#define MPT_LDS_BLOCK 1024
struct Stack16 {
    static constexpr int SENTINEL = -32768;            // leaf ids are ~slot >= -32767 (n < 32768)
    static constexpr bool ONE_TEST = true;             // one depth question in the LEAF step for both kinds of ray (render_kernel.hip lane_start_ray / stage_leaf)
    static constexpr int PLANE_OFF = 8;                // bytes between the {lo, lo} and {hi, hi} pairs of an axis in LDS
    LdsShortPtr base;          // &lds16[threadIdx.x]
    int sp;
    DEV void push(int v) { base[sp * MPT_LDS_BLOCK] = (short)v; sp++; }
    DEV int pop() { sp--; return (int)base[sp * MPT_LDS_BLOCK]; }
    static constexpr bool PEEK = true;
    DEV int peek(int at) const { return (int)base[at * MPT_LDS_BLOCK]; }
    static constexpr bool SP_ADDR = false, ODD_IDS = false, T_SCALED = false;
    static constexpr int SP_STEP = 1;
};

// The 4-wide nodes with exact boxes resident in LDS (render_kernel_lds4): the first seven float4 of a wnode record -- {lo.x[4]}
// {hi.x[4]}{lo.y[4]}{hi.y[4]}{lo.z[4]}{hi.z[4]}{id[4]} -- MPT_LDS4_NODE_STRIDE = 112 bytes apart (a ds_read_b128 is served sixteen
// lanes at a time over sixteen 16-byte bank groups: piece k of record i starts on group (7 i + k) mod 16, sixteen positions), the
// internal ids as byte offset / 8.  The entry planes of an axis are the 16 bytes at offset 0 for a ray going up the axis and at 16
// for one going down, the exit planes the other 16: no decode, no min / max, no select -- what the 8-bit nodes of the gather
// kernels pay 39 VALU instructions a step for, to save gathers this kernel does not make
#ifndef MPT_LDS4_AVOID_IN_LEAF
#define MPT_LDS4_AVOID_IN_LEAF 1   // the triangle a ray left from is filtered by the LEAF step (one compare) instead of by the NODE step (four)
#endif
struct LdsWideScene {
    static constexpr int LEAF_REP = MPT_LEAF_REP;
    static constexpr bool PRESCALED_IDS = true;
    static constexpr bool AVOID_IN_LEAF = MPT_LDS4_AVOID_IN_LEAF != 0;
    static constexpr int NODE_REP = MPT_LDS4_REP;
    static constexpr int SHADE_MIN = MPT_SHADE_MIN_LDS;
    static constexpr bool WIDE = true, QUANT = false, SIGNED_PLANES = false, LDS_MATS = true, OCT = false;
    // ids as the LDS copy of the node records holds them (render_kernel_lds4 rewrites them while it copies): a node's is its record's
    // byte offset in LDS -- the address itself, no shift -- and a leaf's (slot << 4) | 1; records are 16-byte aligned, so bit 0 tells
    // them apart with a full-rate v_and where the sign needed a shift or a sign extension (half rate on gfx950)
    static constexpr bool ODD_IDS = true;
    static constexpr bool T_SCALED = true;                  // (Stack16W::ts)
    LdsVec4Ptr wnode, tgeo, mats;
    LdsU8Ptr mtl;
    int mat_last, mat_default;
    DEV void node4(int i, int ox, int oy, int oz, MptVec4 &nx, MptVec4 &fx, MptVec4 &ny, MptVec4 &fy, MptVec4 &nz, MptVec4 &fz,
                   MptVec4 &id) const {
        // (ODD_IDS: the id is the record's LDS address -- the records start at the workgroup's LDS address 0, which the kernel checks;
        //  written as wnode + i the compiler keeps a v_add_u32 of the constant 0 in every step)
        LdsBytePtr nd = ODD_IDS ? (LdsBytePtr)(unsigned long long)(unsigned)i : (LdsBytePtr)wnode + (i << 3);
        nx = lds_ld((LdsVec4Ptr)(nd + ox));      fx = lds_ld((LdsVec4Ptr)(nd + (ox ^ 16)));
        ny = lds_ld((LdsVec4Ptr)(nd + 32 + oy)); fy = lds_ld((LdsVec4Ptr)(nd + 32 + (oy ^ 16)));
        nz = lds_ld((LdsVec4Ptr)(nd + 64 + oz)); fz = lds_ld((LdsVec4Ptr)(nd + 64 + (oz ^ 16)));
        id = lds_ld((LdsVec4Ptr)(nd + 96));
    }
    DEV void tri(int slot, MptVec4 &g0, MptVec4 &g1, MptVec4 &g2) const {
        // (`tgeo + slot * 3` is compiled into a 64-bit multiply-add, v_mad_u64_u32 -- gfx950 has no 32-bit integer mad -- for a 32-bit LDS
        //  address; the 24-bit multiply is one instruction, v_mad_u32_u24: -0.25 % per launch.  Slots are below 2^15)
        // with ODD_IDS `slot` is the leaf's id, 16 * slot + 1: three times that is the record's offset plus three
        LdsVec4Ptr g = ODD_IDS ? (LdsVec4Ptr)((LdsBytePtr)tgeo - 3 + __umul24((unsigned)slot, 3u))
                               : (LdsVec4Ptr)((LdsBytePtr)tgeo + __umul24((unsigned)slot, 48u));
        g0 = lds_ld(g); g1 = lds_ld(g + 1); g2 = lds_ld(g + 2);
    }
};

// its LIFO: 16-bit entries, [level][lane of 1024], as many levels as the tree can ask for (3 x depth + 2, the host checks): a
// step's three pushes are plain stores, nothing spills
#ifndef MPT_LDS4_PLANE_OFF
#define MPT_LDS4_PLANE_OFF 0       // 16: the ray carries the offsets of its entry planes (three registers); 0: the step reads the signs off 1/d
#endif
struct Stack16W {
    static constexpr bool ODD_IDS = true;              // entries are ids as LdsWideScene holds them (16 bits, unsigned)
    static constexpr bool ONE_TEST = true;             // (render_kernel.hip lane_start_ray / stage_leaf)
    static constexpr int SENTINEL = 2;                 // the two low bits of an entry are the lane's next state: 0 a node (ST_NODE), 1 a leaf
                                                       // (ST_LEAF), 2 -- only this -- the bottom of the stack (ST_DONE)
    static constexpr int PLANE_OFF = MPT_LDS4_PLANE_OFF;
    static constexpr int CAP = 1 << 20, STRIDE = MPT_LDS_BLOCK;
    static constexpr bool NO_SPILL = true;
    typedef short entry_t;
    LdsShortPtr base;          // &lds16[threadIdx.x]
    int sp;
    // T_SCALED: while a ray is traversed its 1/d, o/d and tbest are held multiplied by ts = MptRenderParams::t_scale, a power of two
    // small enough that no box is entered beyond distance 1 / ts: every t of a slab test is the unscaled one times ts bit for bit
    // (so is every comparison between them), and the entry side's max(t, 0) becomes the clamp bit of one of its FMAs -- four
    // half-rate v_max_f32 less per step.  (A t beyond 1 / ts -- a plane nearly parallel to the ray -- clamps to 1: the box test
    // can only pass where it failed, never fail where it passed.)
    static constexpr bool T_SCALED = true;
    float ts;
    DEV void push(int v) { base[sp * MPT_LDS_BLOCK] = (short)v; sp++; }        // (by level: the sentinel at a ray's start)
    DEV int pop() { sp--; return (int)base[sp * MPT_LDS_BLOCK]; }
    static constexpr bool PEEK = true;                 // (the LEAF step reads the entry it will pop together with its triangle)
    DEV int peek(int at) const { return (int)base[at * MPT_LDS_BLOCK]; }
    // SP_ADDR: LaneState::sp is the LDS byte ADDRESS of the lane's TOP entry, not a level: the entry a step may pop is read at the
    // register itself, a push goes to the register + SP_STEP -- the instruction's offset field -- and moving it is a full-rate add of
    // SP_STEP, where level * 2048 + base was a v_lshl_add_u32 per access (shifts and three-operand integer forms issue at half the
    // rate of adds on gfx950: tools/microbench/exec_microbench)
    static constexpr bool SP_ADDR = true;
    static constexpr int SP_STEP = MPT_LDS_BLOCK * 2, SP_BIAS = SP_STEP;       // st / ld address the slot at (sp + SP_BIAS - SP_STEP * k)
    DEV int sp_at(int level) const { return (int)(unsigned)(unsigned long long)(base + level * MPT_LDS_BLOCK); }
    DEV static void st(int sp, int v) { *(LdsShortPtr)(unsigned long long)(unsigned)sp = (short)v; }
    DEV static int ld(int sp) { return (int)*(LdsUShortPtr)(unsigned long long)(unsigned)sp; }
};

// the same LIFO for the tracer waves of the pooled kernel: [level][tracer lane], the lane count a launch parameter
struct Stack16V {
    static constexpr int SENTINEL = -32768;
    static constexpr bool ONE_TEST = true;             // one depth question in the LEAF step for both kinds of ray (render_kernel.hip lane_start_ray / stage_leaf)
    static constexpr int PLANE_OFF = 8;
    LdsShortPtr base;          // &lds16[tracer lane]
    int stride;                // tracer lanes of the workgroup (wave-uniform)
    int sp;
    DEV void push(int v) { base[sp * stride] = (short)v; sp++; }
    DEV int pop() { sp--; return (int)base[sp * stride]; }
    static constexpr bool PEEK = true;
    DEV int peek(int at) const { return (int)base[at * stride]; }
    static constexpr bool SP_ADDR = false, ODD_IDS = false, T_SCALED = false;
    static constexpr int SP_STEP = 1;
};

#if MPT_STRICT
// LinearBVH.intersect, tree/lbvh.py:314-347, operation for operation.  `avoid`/index are leaf slots.
template <bool COUNT>
DEV Hit bvh_closest(const MptRenderParams &p, int *lds, V3 ro, V3 rd, int avoid, Cnt &cnt) {
    const int n = p.n;
    Stack st; st.base = lds; st.sp = 0;
    st.push(n);
    Hit ret; ret.hit = 0; ret.depth = MPT_INF; ret.index = -1; ret.u = 0.0f; ret.v = 0.0f;
    if (COUNT) cnt.rays++;
    int ntimes = 0;
    while (ntimes < n && st.sp != 0) {
        int curr = st.pop();
        if (curr < n) {
            if (curr != avoid) {
                if (COUNT) cnt.n_tri++;
                float d, s, t;
                const MptVec4 *g = p.tgeo + (size_t)curr * 4;
                if (tri_test(g[0], g[1], g[2], g[3], ro, rd, &d, &s, &t) && d < ret.depth) {
                    ret.depth = d; ret.index = curr; ret.u = s; ret.v = t; ret.hit = 1;
                }
            }
            continue;
        }
        int i = curr - n;
        MptVec4 a = p.snode[(size_t)i * 2], b = p.snode[(size_t)i * 2 + 1];
        if (COUNT) { cnt.n_box++; cnt.n_node++; }
        if (!box_strict(ld3(a), ld3(b), ro, rd)) continue;
        ntimes++;
        st.push(__float_as_int(a.w));
        st.push(__float_as_int(b.w));
    }
    return ret;
}

// path.py:50-51: the shadow ray is a full closest-hit query in the reference
template <bool COUNT>
DEV bool bvh_occluded(const MptRenderParams &p, int *lds, V3 ro, V3 rd, int avoid, float dis, Cnt &cnt) {
    Hit occ = bvh_closest<COUNT>(p, lds, ro, rd, avoid, cnt);
    return !(occ.hit == 0 || occ.depth > dis);
}

#else  // ---------------------------------------------------------------- production traversal

// slab test against [0, tmax] with precomputed 1/d and o/d; returns entry distance
DEV bool box_fast(float lox, float loy, float loz, float hix, float hiy, float hiz,
                  V3 inv, V3 oinv, float tmax, float *tnear) {
    float t1x = __builtin_fmaf(lox, inv.x, -oinv.x), t2x = __builtin_fmaf(hix, inv.x, -oinv.x);
    float t1y = __builtin_fmaf(loy, inv.y, -oinv.y), t2y = __builtin_fmaf(hiy, inv.y, -oinv.y);
    float t1z = __builtin_fmaf(loz, inv.z, -oinv.z), t2z = __builtin_fmaf(hiz, inv.z, -oinv.z);
    float tn = fmaxf(fmaxf(fminf(t1x, t2x), fminf(t1y, t2y)), fmaxf(fminf(t1z, t2z), 0.0f));
    float tf = fminf(fminf(fmaxf(t1x, t2x), fmaxf(t1y, t2y)), fminf(fmaxf(t1z, t2z), tmax));
    *tnear = tn;
    return tn <= tf;
}

// Same hit set as lbvh.py:314-347 (every triangle whose own and ancestors' boxes the ray enters
// before the best depth is tested); the order differs: near child first, far child pushed,
// subtrees beyond the best depth skipped.  ANY = stop at the first hit with depth <= tmax
// (path.py:51: occluded iff the closest hit is within li.dis).
template <bool ANY, bool COUNT, class SCENE, class STACK>
DEV Hit bvh_walk(const SCENE &sc, int n, STACK st, V3 ro, V3 rd, int avoid, float tmax, Cnt &cnt) {
    Hit ret; ret.hit = 0; ret.depth = tmax; ret.index = -1; ret.u = 0.0f; ret.v = 0.0f;
    if (COUNT) cnt.rays++;
    if (n < 2) return ret;     // lbvh.py:218,319: with one face the root box is never written
    V3 inv = v3(m_rcp(rd.x), m_rcp(rd.y), m_rcp(rd.z));
    V3 oinv = ro * inv;
    st.sp = 0;
    int curr = 0;
    for (;;) {
        MptVec4 a, b, c, d;
        sc.node(curr, a, b, c, d);
        int id0 = __float_as_int(d.x), id1 = __float_as_int(d.y);
        if (COUNT) { cnt.n_node++; cnt.n_box += 2; }
        float tn0, tn1;
        bool h0 = box_fast(a.x, b.x, c.x, a.z, b.z, c.z, inv, oinv, ret.depth, &tn0);
        bool h1 = box_fast(a.y, b.y, c.y, a.w, b.w, c.w, inv, oinv, ret.depth, &tn1);
        if (h0 && id0 < 0) {
            h0 = false;
            int slot = ~id0;
            if (slot != avoid) {
                if (COUNT) cnt.n_tri++;
                float dd, s, t;
                MptVec4 g0, g1, g2;
                sc.tri(slot, g0, g1, g2);
                if (tri_test_fast(g0, g1, g2, ro, rd, &dd, &s, &t) && (ANY ? dd <= ret.depth : dd < ret.depth)) {
                    ret.depth = dd; ret.index = slot; ret.u = s; ret.v = t; ret.hit = 1;
                    if (ANY) return ret;
                }
            }
        }
        if (h1 && id1 < 0) {
            h1 = false;
            int slot = ~id1;
            if (slot != avoid) {
                if (COUNT) cnt.n_tri++;
                float dd, s, t;
                MptVec4 g0, g1, g2;
                sc.tri(slot, g0, g1, g2);
                if (tri_test_fast(g0, g1, g2, ro, rd, &dd, &s, &t) && (ANY ? dd <= ret.depth : dd < ret.depth)) {
                    ret.depth = dd; ret.index = slot; ret.u = s; ret.v = t; ret.hit = 1;
                    if (ANY) return ret;
                }
            }
        }
        if (h0 && h1) {
            bool swap = tn1 < tn0;
            st.push(swap ? id0 : id1);
            curr = swap ? id1 : id0;
        } else if (h0) {
            curr = id0;
        } else if (h1) {
            curr = id1;
        } else {
            if (st.sp == 0) break;
            curr = st.pop();
        }
    }
    return ret;
}

// what a traversal needs besides the ray: where the scene records and this lane's stack live
template <class SCENE, class STACK>
struct Tracer {
    SCENE sc;
    STACK st;
    int n;
    template <bool COUNT>
    DEV Hit closest(V3 ro, V3 rd, int avoid, Cnt &cnt) const {
        Hit h = bvh_walk<false, COUNT>(sc, n, st, ro, rd, avoid, MPT_INF, cnt);
        if (!h.hit) h.depth = MPT_INF;
        return h;
    }
    template <bool COUNT>
    DEV bool occluded(V3 ro, V3 rd, int avoid, float dis, Cnt &cnt) const {
        return bvh_walk<true, COUNT>(sc, n, st, ro, rd, avoid, dis, cnt).hit != 0;
    }
};
#endif

DEV float sphere_intersect(V3 pos, float rad2, V3 ro, V3 rd) {               // geometries.py:159-177
    float ret = 0.0f;
    V3 op = pos - ro;
    float b = dot(op, rd);
    float det = b * b + rad2 - norm_sqr(op);
    if (det >= 0.0f) {
        det = m_sqrt(det);
        float t = b - det;
        if (t > MPT_EPS) {
            ret = t;
        } else {
            t = b + det;
            if (t > MPT_EPS) ret = t;
        }
    }
    return ret;
}

DEV bool area_intersect(V3 pos, V3 dirx, V3 diry, V3 ro, V3 rd, float *depth, float *u_ = nullptr, float *v_ = nullptr) {   // geometries.py:58-74
    bool hit = false;
    V3 nrm = normalized(cross(dirx, diry));
    float NoD = dot(nrm, rd);
    if (NoD > MPT_EPS) {
        float t = m_div(dot(nrm, pos - ro), NoD);
        V3 hitdisp = ro + rd * t - pos;
        float u = m_div(dot(hitdisp, dirx), norm_sqr(dirx));
        float v = m_div(dot(hitdisp, diry), norm_sqr(diry));
        *depth = t;
        if (u_) *u_ = u;
        if (v_) *v_ = v;
        if (-1.0f < u && u < 1.0f && -1.0f < v && v < 1.0f) hit = true;
    }
    return hit;
}

// ---------------------------------------------------------------- textures (image.py:137-148, common.py:183-192)
DEV int pymod(int a, int b) { int r = a % b; return r < 0 ? r + b : r; }

DEV MptVec4 image_texel(const MptRenderParams &p, MptImage im, int x, int y) {
    x = pymod(x, im.nx);
    y = pymod(y, im.ny);
    return p.texels[(size_t)im.base + (size_t)x * im.ny + y];
}

DEV MptVec4 image_sample(const MptRenderParams &p, int id, float x, float y) {
    MptImage im = p.images[id];
    float px = x * (float)(im.nx - 1), py = y * (float)(im.ny - 1);
    float fx = floorf(px), fy = floorf(py);
    int Ix = (int)fx, Iy = (int)fy;
    float x0 = px - fx, x1 = py - fy;
    float y0 = 1.0f - x0, y1 = 1.0f - x1;
    MptVec4 t11 = image_texel(p, im, Ix + 1, Iy + 1), t10 = image_texel(p, im, Ix + 1, Iy);
    MptVec4 t00 = image_texel(p, im, Ix, Iy), t01 = image_texel(p, im, Ix, Iy + 1);
    MptVec4 r;
    r.x = t11.x * x0 * x1 + t10.x * x0 * y1 + t00.x * y0 * y1 + t01.x * y0 * x1;
    r.y = t11.y * x0 * x1 + t10.y * x0 * y1 + t00.y * y0 * y1 + t01.y * y0 * x1;
    r.z = t11.z * x0 * x1 + t10.z * x0 * y1 + t00.z * y0 * y1 + t01.z * y0 * x1;
    r.w = t11.w * x0 * x1 + t10.w * x0 * y1 + t00.w * y0 * y1 + t01.w * y0 * x1;
    return r;
}

// ---------------------------------------------------------------- microfacet.py
DEV float schlickFresnel(float cost) { return m_pow5(clampf(1.0f - cost, 0.0f, 1.0f)); }   // :9-10

DEV float dielectricFresnel(float etai, float etao, float cosi) {            // :14-27
    float sini = m_sqrt(fmaxf(0.0f, 1.0f - cosi * cosi));
    float sint = m_div(etao, etai) * sini;
    float ret = 1.0f;
    if (sint < 1.0f) {
        float cost = m_sqrt(fmaxf(0.0f, 1.0f - sint * sint));
        float a1 = etai * cosi, a2 = etao * cost;
        float b1 = etao * cosi, b2 = etai * cost;
        float para = m_div(a1 - a2, a1 + a2);
        float perp = m_div(b1 - b2, b1 + b2);
        ret = 0.5f * (para * para + perp * perp);
    }
    return ret;
}

DEV float GTR1(float cosh_, float alpha) {                                   // :31-34
    float alpha2 = alpha * alpha;
    float t = 1.0f + (alpha2 - 1.0f) * (cosh_ * cosh_);
    return m_div(alpha2 - 1.0f, MPT_PI * m_log(alpha2) * t);
}
DEV float GTR2(float cosh_, float alpha) {                                   // :38-41
    float alpha2 = alpha * alpha;
    float t = 1.0f + (alpha2 - 1.0f) * (cosh_ * cosh_);
    return m_div(alpha2, MPT_PI * (t * t));
}
DEV float smithGGX(float cosi, float alpha) {                                // :45-48
    float a = alpha * alpha;
    float b = cosi * cosi;
    return m_rcp(cosi + m_sqrt(a + b - a * b));
}
DEV V3 sample_GTR1(float u, float v, float alpha) {                          // :69-71 (NaN for alpha < 1, like the reference)
    u = m_div(m_sqrt(m_pow(alpha, 2.0f - 2.0f * u) - 1.0f), alpha * alpha - 1.0f);
    return spherical(u, v);
}
DEV V3 sample_GTR2(float u, float v, float alpha) {                          // :75-77
    u = m_sqrt(m_div(1.0f - u, 1.0f - u * (1.0f - alpha * alpha)));
    return spherical(u, v);
}

// ---------------------------------------------------------------- materials/disney.py
struct Disney {
    V3 basecolor;
    float metallic, roughness, specular, specularTint, subsurface, sheen, sheenTint, clearcoat,
        clearcoatGloss, transmission, ior;
    V3 speccolor, sheencolor;
    float alpha, clearcoatAlpha;
};

DEV void disney_init(Disney &m) {                                            // disney.py:14-50
    V3 tint = v3s(1.0f);
    float lum = dot(m.basecolor, v3(0.3f, 0.6f, 0.1f));
    if (lum > MPT_EPS) tint = vdivs(m.basecolor, lum);
    m.speccolor = lerpv(m.metallic, lerpv(m.specularTint, v3s(1.0f), tint) * (m.specular * 0.08f), m.basecolor);
    m.sheencolor = lerpv(m.sheenTint, v3s(1.0f), tint);
    m.alpha = fmaxf(0.001f, m.roughness * m.roughness);
    m.clearcoatAlpha = lerpf(m.clearcoatGloss, 0.1f, 0.001f);
}

// MaterialPool.get + ParameterPair.get, mtllib.py:30-38,79-95
#if !MPT_STRICT
// Production build: every material, the default one included (record default_mtl), is one record; untextured
// ones carry the derived terms of Disney.__init__ ready-made (bit for bit what disney_init computes: the same
// device function filled them in), so a bounce costs six 16-B gathers and no per-hit re-derivation.
// q0..q3, d0, d1: the record's float4 0-3 and 8-9, from wherever the caller keeps them; mt: the record in
// global memory (texture ids, read by textured materials only)
DEV void material_from(const MptRenderParams &p, const MptMaterial *mt, MptVec4 q0, MptVec4 q1, MptVec4 q2, MptVec4 q3,
                       MptVec4 d0, MptVec4 d1, float tu, float tv, Disney &m) {
    float v[14] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y };
    const bool textured = __float_as_int(q3.z) != 0;                           // p[14] mirrors any_tex
    if (textured) {
#pragma unroll 1
        for (int k = 0; k < 12; k++) {
            int texid = mt->tex[k];
            if (texid != -1) {
                MptVec4 t = image_sample(p, texid, tu, tv);
                if (k == 0) { v[0] *= t.x; v[1] *= t.y; v[2] *= t.z; }
                else {
                    switch (k) {                                               // scalar parameters take .x of fac * texel (mtllib.py:83-93)
                    case 1: v[3] *= t.x; break; case 2: v[4] *= t.x; break; case 3: v[5] *= t.x; break;
                    case 4: v[6] *= t.x; break; case 5: v[7] *= t.x; break; case 6: v[8] *= t.x; break;
                    case 7: v[9] *= t.x; break; case 8: v[10] *= t.x; break; case 9: v[11] *= t.x; break;
                    case 10: v[12] *= t.x; break; default: v[13] *= t.x; break;
                    }
                }
            }
        }
    }
    m.basecolor = v3(v[0], v[1], v[2]);
    m.metallic = v[3]; m.roughness = v[4]; m.specular = v[5]; m.specularTint = v[6];
    m.subsurface = v[7]; m.sheen = v[8]; m.sheenTint = v[9]; m.clearcoat = v[10];
    m.clearcoatGloss = v[11]; m.transmission = v[12]; m.ior = v[13];
    if (textured) disney_init(m);
    else {
        m.speccolor = v3(d0.x, d0.y, d0.z); m.sheencolor = v3(d0.w, d1.x, d1.y);
        m.alpha = d1.z; m.clearcoatAlpha = d1.w;
    }
}
DEV void material_get(const MptRenderParams &p, int mtlid, float tu, float tv, Disney &m) {
    const MptMaterial *mt = p.mats + (mtlid == -1 ? p.default_mtl : mtlid);
    const MptVec4 *q = (const MptVec4 *)mt;
    material_from(p, mt, q[0], q[1], q[2], q[3], q[8], q[9], tu, tv, m);
}
#else
DEV void material_get(const MptRenderParams &p, int mtlid, float tu, float tv, Disney &m) {
    if (mtlid == -1) {
        m.basecolor = v3s(0.8f);
        m.metallic = 0.0f; m.roughness = 0.4f; m.specular = 0.5f; m.specularTint = 0.4f;
        m.subsurface = 0.0f; m.sheen = 0.0f; m.sheenTint = 0.4f; m.clearcoat = 0.0f;
        m.clearcoatGloss = 0.5f; m.transmission = 0.0f; m.ior = 1.45f;
    } else {
        const MptVec4 *q = (const MptVec4 *)(p.mats + mtlid);
        MptVec4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        float v[14] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y };
        const MptMaterial *mt = p.mats + mtlid;
        if (mt->any_tex) {
#pragma unroll 1
            for (int k = 0; k < 12; k++) {
                int texid = mt->tex[k];
                if (texid != -1) {
                    MptVec4 t = image_sample(p, texid, tu, tv);
                    if (k == 0) { v[0] *= t.x; v[1] *= t.y; v[2] *= t.z; }
                    else {
                        // scalar parameters take .x of fac * texel (mtllib.py:83-93)
                        switch (k) {
                        case 1: v[3] *= t.x; break; case 2: v[4] *= t.x; break; case 3: v[5] *= t.x; break;
                        case 4: v[6] *= t.x; break; case 5: v[7] *= t.x; break; case 6: v[8] *= t.x; break;
                        case 7: v[9] *= t.x; break; case 8: v[10] *= t.x; break; case 9: v[11] *= t.x; break;
                        case 10: v[12] *= t.x; break; default: v[13] *= t.x; break;
                        }
                    }
                }
            }
        }
        m.basecolor = v3(v[0], v[1], v[2]);
        m.metallic = v[3]; m.roughness = v[4]; m.specular = v[5]; m.specularTint = v[6];
        m.subsurface = v[7]; m.sheen = v[8]; m.sheenTint = v[9]; m.clearcoat = v[10];
        m.clearcoatGloss = v[11]; m.transmission = v[12]; m.ior = v[13];
    }
    disney_init(m);
}
#endif

DEV V3 disney_brdf(const Disney &m, V3 normal, float sign, V3 indir, V3 outdir) {   // disney.py:53-106
    float etai = 1.0f, etao = m.ior;
    if (sign < 0.0f) { etai = m.ior; etao = 1.0f; }

    V3 halfdir = normalized(indir + outdir);
    float cosi = dot(indir, normal);
    float coso = dot(outdir, normal);
    float cosh_ = dot_or_zero(halfdir, normal);
    float cosoh = dot_or_zero(halfdir, outdir);

    V3 result = v3s(0.0f);
    if (coso < 0.0f) {
        if (cosi >= 0.0f) {
            float Ds = GTR2(cosh_, m.alpha);
            float fdf = dielectricFresnel(etao, etai, cosoh);
            V3 transmit = m.basecolor * MPT_INV_PI * (1.0f - fdf) * Ds;
            result = transmit * (1.0f - m.metallic) * m.transmission;
        }
    } else {
        float Fi = schlickFresnel(cosi);
        float Fo = schlickFresnel(coso);
        float Fd90 = 0.5f + 2.0f * (cosoh * cosoh) * m.roughness;
        float Fd = lerpf(Fi, 1.0f, Fd90) * lerpf(Fo, 1.0f, Fd90);

        float Fss90 = (cosoh * cosoh) * m.roughness;
        float Fss = lerpf(Fi, 1.0f, Fss90) * lerpf(Fo, 1.0f, Fss90);
        float ss = 1.25f * (Fss * (m_rcp(cosi + coso) - 0.5f) + 0.5f);

        float Foh = schlickFresnel(cosoh);
        V3 Fsheen = m.sheencolor * (Foh * m.sheen);

        float Ds = GTR2(cosh_, m.alpha);
        V3 Fs = lerpv(Foh, m.speccolor, v3s(1.0f));
        float Gs = smithGGX(cosi, m.alpha) * smithGGX(coso, m.alpha);
        V3 diffuse = m.basecolor * (MPT_INV_PI * lerpf(m.subsurface, Fd, ss)) + Fsheen;
#if MPT_STRICT
        float fdf = dielectricFresnel(etao, etai, cosoh);
        float Dr = GTR1(cosh_, m.clearcoatAlpha);
        float Gr = smithGGX(cosi, 0.25f) * smithGGX(coso, 0.25f);
        float Fr = lerpf(Foh, 0.04f, 1.0f);
        V3 specular = Fs * Gs * Ds + v3s(0.25f * m.clearcoat * Gr * Fr * Dr);
        V3 transmit = m.basecolor * (MPT_INV_PI * fdf * Ds);
#else
        // the clearcoat and transmission lobes are multiplied by their parameter: when that is
        // exactly zero the term is skipped (identical result whenever the skipped factor is finite)
        float coat = 0.0f;
        if (m.clearcoat != 0.0f) {
            float Dr = GTR1(cosh_, m.clearcoatAlpha);
            float Gr = smithGGX(cosi, 0.25f) * smithGGX(coso, 0.25f);
            float Fr = lerpf(Foh, 0.04f, 1.0f);
            coat = 0.25f * m.clearcoat * Gr * Fr * Dr;
        }
        V3 specular = Fs * Gs * Ds + v3s(coat);
        V3 transmit = v3s(0.0f);
        if (m.transmission != 0.0f) transmit = m.basecolor * (MPT_INV_PI * dielectricFresnel(etao, etai, cosoh) * Ds);
#endif

        result = diffuse * (1.0f - m.metallic) * (1.0f - m.transmission);
        result = result + transmit * (1.0f - m.metallic) * m.transmission;
        result = result + specular * (1.0f - m.transmission);
    }
    return result;
}

struct BsdfSample { V3 outdir; float pdf; V3 color; };

struct Choice {                                                               // materials/__init__.py:22-48
    float pdf, w;
    DEV bool operator()(float r) {
        bool ret;
        if (w < r) {
            w = m_div(w, r);
            pdf *= r;
            ret = true;
        } else {
            w = m_div(w - r, 1.0f - r);
            pdf *= 1.0f - r;
            ret = false;
        }
        return ret;
    }
};

DEV BsdfSample disney_bounce(const Disney &m, V3 normal, float sign, V3 indir, V3 samp) {   // disney.py:115-233
    BsdfSample result;
    result.outdir = v3s(0.0f); result.pdf = 0.0f; result.color = v3s(0.0f);

    float etai = 1.0f, etao = m.ior;
    if (sign < 0.0f) { etai = m.ior; etao = 1.0f; }
#if MPT_STRICT
    float eta = m_div(etai, etao);
#endif

    float cosi = dot(indir, normal);
    float Fi = schlickFresnel(cosi);
    V3 Fs = lerpv(Fi, m.speccolor, v3s(1.0f));

    const TanSpace ts = tanspace(normal);
    Choice choice; choice.pdf = 1.0f; choice.w = samp.z;
    float specrate = lerpf(m.transmission, lerpf(m.metallic, vavg(Fs), 1.0f), 1.0f);
    float coatrate = 0.04f * m.clearcoat;

    specrate = lerpf(specrate, 0.1f, 1.0f);
    if (coatrate != 0.0f) coatrate = lerpf(coatrate, 0.1f, 1.0f);

#if MPT_STRICT
    const bool coat = choice(coatrate);
    if (coat) {
        float alpha = m.clearcoatAlpha;
        V3 halfdir = tanspace_mul(ts, sample_GTR1(samp.x, samp.y, alpha));
        V3 outdir = reflectv(-indir, halfdir);

        float coso = dot(outdir, normal);
        float cosh_ = dot_or_zero(halfdir, normal);
        float cosoh = dot_or_zero(halfdir, outdir);
        if (cosoh > 0.0f) {
            float Dr = GTR1(cosh_, alpha);
            float Foh = schlickFresnel(cosoh);
            float Fr = lerpf(Foh, 0.04f, 1.0f);

            result.outdir = outdir;
            float partial = m_div(m.clearcoat * Fr * coso, cosoh);
            result.pdf = Dr * partial;
            result.color = v3s(m_div(partial, choice.pdf));
        }
    } else if (choice(specrate)) {
        float alpha = m.alpha;
        V3 halfdir = tanspace_mul(ts, sample_GTR2(samp.x, samp.y, alpha));
        V3 outdir = reflectv(-indir, halfdir);

        float coso = dot_or_zero(outdir, normal);
        float cosh_ = dot_or_zero(halfdir, normal);
        float cosoh = dot_or_zero(halfdir, outdir);
        if (cosoh > 0.0f && coso > 0.0f && cosh_ > 0.0f) {
            float Ds = GTR2(cosh_, alpha);

            if (choice(m.transmission)) {
                float fdf = dielectricFresnel(etao, etai, cosoh);
                float reflrate = lerpf(fdf, 0.2f, 1.0f);

                if (choice(reflrate)) {
                    result.outdir = outdir;
                    result.pdf = Ds * fdf;
                    result.color = vdivs(m.basecolor * fdf * m.transmission, choice.pdf);
                } else {
                    V3 T;
                    if (refractv(-indir, halfdir, eta, &T)) {
                        result.outdir = T;
                        result.pdf = Ds * (1.0f - fdf);
                        result.color = vdivs(m.basecolor * (1.0f - fdf) * m.transmission, choice.pdf);
                    }
                }
            } else {
                float Foh = schlickFresnel(cosoh);
                V3 Fs2 = lerpv(Foh, m.speccolor, v3s(1.0f));

                result.outdir = outdir;
                float partial = m_div(0.5f, cosoh * smithGGX(coso, alpha));
                result.pdf = Ds * vavg(Fs2) * partial;
                result.color = vdivs(Fs2 * partial * (1.0f - m.transmission), choice.pdf);
            }
        }
    } else {
        V3 outdir = tanspace_mul(ts, spherical(m_sqrt(samp.x), samp.y));

        V3 halfdir = normalized(indir + outdir);
        float cosi2 = cosi;                                                  // disney.py:207 recomputes the same dot product
        float coso = dot(outdir, normal);
        float cosoh = dot_or_zero(halfdir, outdir);

        float Fi2 = Fi;                                                      // and the same schlickFresnel(cosi), :212
        float Fo = schlickFresnel(coso);
        float Fd90 = 0.5f + 2.0f * (cosoh * cosoh) * m.roughness;
        float Fd = lerpf(Fi2, 1.0f, Fd90) * lerpf(Fo, 1.0f, Fd90);

        float Fss90 = (cosoh * cosoh) * m.roughness;
        float Fss = lerpf(Fi2, 1.0f, Fss90) * lerpf(Fo, 1.0f, Fss90);
        float ss = 1.25f * (Fss * (m_rcp(cosi2 + coso) - 0.5f) + 0.5f);

        float Foh = schlickFresnel(cosoh);
        V3 Fsheen = m.sheencolor * (Foh * m.sheen);

        V3 diffuse = m.basecolor * (MPT_INV_PI * lerpf(m.subsurface, Fd, ss)) + Fsheen;

        result.outdir = outdir;
        result.pdf = MPT_INV_PI;
        result.color = vdivs(diffuse * MPT_PI * (1.0f - m.metallic) * (1.0f - m.transmission), choice.pdf);
    }
#else
    // Production build: the three lobes of disney.py:136-231 share what they have in common.  Each lane first
    // picks its lobe (same Choice sequence), then ONE tangent-space direction is sampled with the lobe's own
    // polar cosine (the branches' sample_GTR1 / sample_GTR2 / cosine sampling differ in nothing else), ONE set of
    // cosines and Schlick terms is formed, and only the lobe-specific weights diverge.  Per lane the operations
    // and their order are those of the branches above; a wave whose lanes sit in two lobes issues the shared part
    // once instead of twice.
    // without a clearcoat the first Choice is the identity (w < 0 never holds, w = (w - 0) / (1 - 0), pdf *= 1)
    const bool coat = coatrate != 0.0f && choice(coatrate);
    const bool spec = !coat && choice(specrate);
    float hz;                                                                 // cosine of the sampled direction to the normal
    if (coat) {                                                               // sample_GTR1, microfacet.py:69-71
        float a = m.clearcoatAlpha;
        hz = m_div(m_sqrt(m_pow(a, 2.0f - 2.0f * samp.x) - 1.0f), a * a - 1.0f);
    } else if (spec) {                                                        // sample_GTR2, :75-77
        hz = m_sqrt(m_div(1.0f - samp.x, 1.0f - samp.x * (1.0f - m.alpha * m.alpha)));
    } else {
        hz = m_sqrt(samp.x);                                                  // cosine-weighted hemisphere, disney.py:203
    }
    const V3 sdir = tanspace_mul(ts, spherical(hz, samp.y));
    V3 halfdir, outdir;
    if (coat || spec) { halfdir = sdir; outdir = reflectv(-indir, halfdir); }
    else { outdir = sdir; halfdir = normalized(indir + outdir); }
    const float coso_raw = dot(outdir, normal);
    const float cosh_ = dot_or_zero(halfdir, normal);
    const float cosoh = dot_or_zero(halfdir, outdir);
    const float Foh = schlickFresnel(cosoh);
    if (coat) {
        if (cosoh > 0.0f) {
            float Dr = GTR1(cosh_, m.clearcoatAlpha);
            float Fr = lerpf(Foh, 0.04f, 1.0f);
            result.outdir = outdir;
            float partial = m_div(m.clearcoat * Fr * coso_raw, cosoh);
            result.pdf = Dr * partial;
            result.color = v3s(m_div(partial, choice.pdf));
        }
    } else if (spec) {
        const float coso = fmaxf(0.0f, coso_raw);
        if (cosoh > 0.0f && coso > 0.0f && cosh_ > 0.0f) {
            float Ds = GTR2(cosh_, m.alpha);
            if (choice(m.transmission)) {
                float fdf = dielectricFresnel(etao, etai, cosoh);
                float reflrate = lerpf(fdf, 0.2f, 1.0f);
                if (choice(reflrate)) {
                    result.outdir = outdir;
                    result.pdf = Ds * fdf;
                    result.color = vdivs(m.basecolor * fdf * m.transmission, choice.pdf);
                } else {
                    V3 T;
                    const float eta = m_div(etai, etao);                     // only this branch needs it
                    if (refractv(-indir, halfdir, eta, &T)) {
                        result.outdir = T;
                        result.pdf = Ds * (1.0f - fdf);
                        result.color = vdivs(m.basecolor * (1.0f - fdf) * m.transmission, choice.pdf);
                    }
                }
            } else {
                V3 Fs2 = lerpv(Foh, m.speccolor, v3s(1.0f));
                result.outdir = outdir;
                float partial = m_div(0.5f, cosoh * smithGGX(coso, m.alpha));
                result.pdf = Ds * vavg(Fs2) * partial;
                result.color = vdivs(Fs2 * partial * (1.0f - m.transmission), choice.pdf);
            }
        }
    } else {
        const float coso = coso_raw;
        float Fo = schlickFresnel(coso);
        float Fd90 = 0.5f + 2.0f * (cosoh * cosoh) * m.roughness;
        float Fd = lerpf(Fi, 1.0f, Fd90) * lerpf(Fo, 1.0f, Fd90);
        float Fss90 = (cosoh * cosoh) * m.roughness;
        float Fss = lerpf(Fi, 1.0f, Fss90) * lerpf(Fo, 1.0f, Fss90);
        float ss = 1.25f * (Fss * (m_rcp(cosi + coso) - 0.5f) + 0.5f);
        V3 Fsheen = m.sheencolor * (Foh * m.sheen);
        V3 diffuse = m.basecolor * (MPT_INV_PI * lerpf(m.subsurface, Fd, ss)) + Fsheen;
        result.outdir = outdir;
        result.pdf = MPT_INV_PI;
        result.color = vdivs(diffuse * MPT_PI * (1.0f - m.metallic) * (1.0f - m.transmission), choice.pdf);
    }
#endif
    return result;
}

DEV float power_heuristic(float a, float b) {                                // path.py:11-15
    a = clampf(a, MPT_EPS, MPT_INF); a = a * a;
    b = clampf(b, MPT_EPS, MPT_INF); b = b * b;
    return m_div(a, a + b);
}

// ---------------------------------------------------------------- lights
#if !MPT_STRICT && defined(__HIP_DEVICE_COMPILE__)     // (the host pass only parses this file)
// Production build: the light records are read through the constant address space, so a wave-uniform index (the
// loop of lights_hit; the one light of a one-light scene in lights_sample) becomes scalar loads through the
// scalar cache instead of a 64-lane gather of one address that the stage then waits a full L2 round trip for
typedef const __attribute__((address_space(4))) MptLight *LightPtr;
DEV LightPtr lights_const(const MptLight *g) {
    LightPtr q = (LightPtr)(const void *)g;
    asm("" : "+s"(q));         // forget that this was a global pointer (or the compiler turns the reads back into global loads)
    return q;
}
#define MPT_LIGHTS(p) lights_const((p).lights)
#else
#define MPT_LIGHTS(p) ((p).lights)
#endif
DEV V3 axes_mul(const MptLight &L, V3 v) {
    return v3(L.ax0.x * v.x + L.ax0.y * v.y + L.ax0.z * v.z,
              L.ax1.x * v.x + L.ax1.y * v.y + L.ax1.z * v.z,
              L.ax2.x * v.x + L.ax2.y * v.y + L.ax2.z * v.z);
}

struct LightHit { bool hit; float dis, pdf; V3 color; };

DEV LightHit lights_hit(const MptRenderParams &p, V3 ro, V3 rd) {            // light/__init__.py:51-81
    LightHit ret; ret.hit = false; ret.dis = MPT_INF; ret.pdf = 0.0f; ret.color = v3s(0.0f);
    for (int i = 0; i < p.nlights; i++) {
        MptLight L = MPT_LIGHTS(p)[i];
        int type = __float_as_int(L.pos_type.w);
        V3 pos = ld3(L.pos_type);
        float size = L.color_size.w;
        float t = 0.0f, area = 0.0f;
        if (type == 1) {
            t = sphere_intersect(pos, size * size, ro, rd);
            area = MPT_PI * (size * size);
        } else if (type == 2) {
            V3 dirx = axes_mul(L, v3(size, 0.0f, 0.0f));
            V3 diry = axes_mul(L, v3(0.0f, size, 0.0f));
            float d;
            if (area_intersect(pos, dirx, diry, ro, rd, &d)) {
                t = d;
                area = 4.0f * (size * size);
            }
        }
        if (0.0f < t && t < ret.dis) {
            ret.dis = t;
            ret.pdf = m_div(ret.dis * ret.dis, area);
            ret.color = ld3(L.color_size);
            ret.hit = true;
            break;
        }
    }
    return ret;
}

struct LightSample { float dis; V3 dir; float pdf; V3 color; };

DEV LightSample light_sample_one(const MptLight &L, V3 hitpos, V3 samp) {      // light/__init__.py:90-121
    LightSample ret;
    int type = __float_as_int(L.pos_type.w);
    V3 color = ld3(L.color_size);
    V3 pos = ld3(L.pos_type);
    float size = L.color_size.w;

    V3 litpos = v3s(MPT_INF);
    V3 norm = v3s(0.0f);
    float area = 0.0f;
    if (type == 1) {
        V3 disp = spherical(samp.x, samp.y);
        litpos = pos + disp * size;
        area = MPT_PI * (size * size);
    } else if (type == 2) {
        V3 disp = axes_mul(L, v3(samp.x * 2.0f - 1.0f, samp.y * 2.0f - 1.0f, 0.0f));
        norm = axes_mul(L, v3(0.0f, 0.0f, 1.0f));
        litpos = pos + disp * size;
        area = 4.0f * (size * size);
    }
    V3 toli = litpos - hitpos;
    float d2 = norm_sqr(toli);
#if MPT_STRICT
    float dis = sqrtf(d2);
    V3 dir = vdivs(toli, dis);
    float pdf = dis * dis / area;
#else
    float rdis = __builtin_amdgcn_rsqf(d2);
    float dis = d2 * rdis;
    V3 dir = toli * rdis;
    float pdf = dis * dis * __builtin_amdgcn_rcpf(area);
#endif
    color = vdivs(color, pdf);
    if (any_ne0(norm)) color = color * dot_or_zero(norm, dir);
    ret.dis = dis; ret.dir = dir; ret.pdf = pdf; ret.color = color;
    return ret;
}

DEV LightSample lights_sample(const MptRenderParams &p, V3 hitpos, V3 samp) {   // light/__init__.py:83-121
    LightSample ret; ret.dis = MPT_INF; ret.dir = v3s(0.0f); ret.pdf = 0.0f; ret.color = v3s(0.0f);
#if !MPT_STRICT
    // one light: samp.z < 1 (a Sobol point), so floor(samp.z * 1) is 0 -- the record is read with scalar loads and
    // its fields are scalar operands of the arithmetic (own copy of the code, so that nothing merges the two reads)
    if (p.nlights == 1) {
        const MptLight L = MPT_LIGHTS(p)[0];
        return light_sample_one(L, hitpos, samp);
    }
#endif
    if (p.nlights != 0) {
        int i = (int)floorf(samp.z * (float)p.nlights);
        i = min(max(i, 0), min(p.nlights, MPT_MAX_LIGHTS - 1));
        const MptLight L = MPT_LIGHTS(p)[i];
        ret = light_sample_one(L, hitpos, samp);
    }
    return ret;
}

DEV void dir2tex(V3 dir, float *s, float *t) {                               // common.py:234-239
    V3 dn = normalized(dir);
    *s = atan2f(dn.z, dn.x) / MPT_PI * 0.5f + 0.5f;
    *t = atan2f(dn.y, sqrtf(dn.x * dn.x + dn.z * dn.z)) / MPT_PI + 0.5f;
}

DEV V3 world_at(const MptRenderParams &p, V3 dir) {                          // light/world.py:22-29
    V3 fac = v3(p.world_fac[0], p.world_fac[1], p.world_fac[2]);
    if (p.world_tex != -1) {
        V3 d2 = v3(dir.x, dir.z, -dir.y);                                    // dir.y, dir.z = dir.z, -dir.y
        float s, t;
        dir2tex(d2, &s, &t);
        MptVec4 tx = image_sample(p, p.world_tex, s, t);
        fac = fac * v3(tx.x, tx.y, tx.z);
    }
    return fac;
}

// ---------------------------------------------------------------- camera (camera.py:34-39)
DEV void camera_generate(const MptRenderParams &p, float x, float y, V3 *ro, V3 *rd) {
    const float *M = p.v2w;
    float a0 = M[0] * x + M[1] * y + M[2] * -1.0f + M[3] * 1.0f;
    float a1 = M[4] * x + M[5] * y + M[6] * -1.0f + M[7] * 1.0f;
    float a2 = M[8] * x + M[9] * y + M[10] * -1.0f + M[11] * 1.0f;
    float a3 = M[12] * x + M[13] * y + M[14] * -1.0f + M[15] * 1.0f;
    float b0 = M[0] * x + M[1] * y + M[2] * 1.0f + M[3] * 1.0f;
    float b1 = M[4] * x + M[5] * y + M[6] * 1.0f + M[7] * 1.0f;
    float b2 = M[8] * x + M[9] * y + M[10] * 1.0f + M[11] * 1.0f;
    float b3 = M[12] * x + M[13] * y + M[14] * 1.0f + M[15] * 1.0f;
    V3 o = v3(m_div(a0, a3), m_div(a1, a3), m_div(a2, a3));
    V3 o1 = v3(m_div(b0, b3), m_div(b1, b3), m_div(b2, b3));
    *ro = o;
    *rd = normalized(o1 - o);
}

// ---------------------------------------------------------------- shading geometry (model.py:88-101, geometries.py:96-108)
// Face.normal / Face.texcoord, geometries.py:96-108, from a tshade record and the hit's barycentrics
DEV void face_shading(MptVec4 s0, MptVec4 s1, MptVec4 s2, MptVec4 s3, float u, float v, V3 *nrm, float *tu, float *tv) {
    float wx = 1.0f - u - v, wy = u, wz = v;
    V3 vn0 = v3(s0.x, s0.y, s0.z), vn1 = v3(s0.w, s1.x, s1.y), vn2 = v3(s1.z, s1.w, s2.x);
    *nrm = normalized(vn0 * wx + vn1 * wy + vn2 * wz);
    *tu = wx * s2.y + wy * s2.w + wz * s3.y;
    *tv = wx * s2.z + wy * s3.x + wz * s3.z;
}

DEV void get_geometries(const MptRenderParams &p, const Hit &hit, V3 ro, V3 rd, V3 *hitpos, V3 *normal, Disney &mat) {
    const MptVec4 *s = p.tshade + (size_t)hit.index * 4;
    MptVec4 s0 = s[0], s1 = s[1], s2 = s[2], s3 = s[3];
    V3 nrm; float tu, tv;
    face_shading(s0, s1, s2, s3, hit.u, hit.v, &nrm, &tu, &tv);
    *hitpos = ro + rd * hit.depth;
    float sign = -dot(rd, nrm);
    if (sign < 0.0f) nrm = -nrm;
    *normal = nrm;
    material_get(p, __float_as_int(s3.w), tu, tv, mat);
}

#if !MPT_STRICT
// Production SHADE: the shading record is fetched by the caller ahead of everything else in the stage (its
// round trip to L2 then overlaps the light tests instead of following them), the material comes from the scene
// (LDS-resident kernel) or from the record's material id
struct ShadeRec { MptVec4 s0, s1, s2, s3; };
// OFF32 (the LDS-resident kernels): a 32-bit byte offset from the (scalar) base -- one shift and the instruction's own address add, where
// base + (size_t)slot * 4 is a 64-bit shift and a 64-bit add per lane (slots are below 2^26: fill_params refuses more faces).  The
// gather kernels keep the 64-bit form: with the short one their 96-register allocation comes out 5-13 % slower (profiles/r05_ab_experiments.json)
template <bool OFF32>
DEV ShadeRec shade_rec_load(const MptRenderParams &p, int slot) {
    const MptVec4 *s = OFF32 ? (const MptVec4 *)((const char *)p.tshade + ((unsigned)slot << 6)) : p.tshade + (size_t)slot * 4;
    ShadeRec r; r.s0 = s[0]; r.s1 = s[1]; r.s2 = s[2]; r.s3 = s[3];
    return r;
}
template <class SCENE>
DEV void get_geometries_rec(const MptRenderParams &p, const SCENE &sc, const ShadeRec &r, const Hit &hit, V3 ro, V3 rd,
                            V3 *hitpos, V3 *normal, Disney &mat) {
    const MptVec4 s0 = r.s0, s1 = r.s1, s2 = r.s2, s3 = r.s3;
    V3 nrm; float tu, tv;
    face_shading(s0, s1, s2, s3, hit.u, hit.v, &nrm, &tu, &tv);
    *hitpos = ro + rd * hit.depth;
    float sign = -dot(rd, nrm);
    if (sign < 0.0f) nrm = -nrm;
    *normal = nrm;
    if constexpr (SCENE::LDS_MATS) {
        const int rec = sc.mtl[hit.index];
        LdsVec4Ptr q = sc.mats + rec * MPT_LDS_MAT_VEC4;
        MptVec4 q0 = lds_ld(q), q1 = lds_ld(q + 1), q2 = lds_ld(q + 2), q3 = lds_ld(q + 3), d0 = lds_ld(q + 4), d1 = lds_ld(q + 5);
        material_from(p, p.mats + (rec == sc.mat_last ? sc.mat_default : rec), q0, q1, q2, q3, d0, d1, tu, tv, mat);
    } else {
        material_get(p, __float_as_int(s3.w), tu, tv, mat);
    }
}
#endif
