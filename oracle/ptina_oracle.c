/*
 * ptina_oracle.c -- TEST INFRASTRUCTURE ONLY (see ptina_oracle.h).
 *
 * Line-faithful CPU restatement of the PTina hot path.  Every function cites the
 * reference lines it follows (file:line relative to /root/reference).  Arithmetic is
 * orc_real (f32 by default, Taichi's default_fp; build with -DORC_F64 for the f64
 * calibration variant), evaluated in source order, compiled with -ffp-contract=off
 * and without fast-math so results are reproducible.
 *
 * PARITY UNPINNED against real PTina output (Taichi is not installable here, the
 * reference ships no golden vectors).  The Sobol sampler alone is pinned, against
 * scipy (tests/golden/sobol_points.npz).
 *
 * Documented deviations from the reference:
 *   Q6  unset texture ids are treated as -1 (reference default 0 reads an
 *       unallocated image with `% 0`, undefined).
 *   Q14 Morton codes are sorted with a STABLE sort on (code, original index);
 *       the reference uses numpy's unstable argsort (tree/lbvh.py:207), whose order
 *       among equal codes is implementation-defined.
 *   Q9  traversal stack holds 256 entries instead of 32; the deepest use is
 *       reported in the counters (the reference overflows silently).
 */
#include "ptina_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef orc_real real;

#ifdef ORC_F64
#define R_SQRT sqrt
#define R_SIN sin
#define R_COS cos
#define R_POW pow
#define R_LOG log
#define R_ATAN2 atan2
#define R_FLOOR floor
#define R_FABS fabs
#define R_FMAX fmax
#define R_FMIN fmin
#else
#define R_SQRT sqrtf
#define R_SIN sinf
#define R_COS cosf
#define R_POW powf
#define R_LOG logf
#define R_ATAN2 atan2f
#define R_FLOOR floorf
#define R_FABS fabsf
#define R_FMAX fmaxf
#define R_FMIN fminf
#endif

/* common.py:32-33 */
#define EPS ((real)1e-6)
#define INF ((real)1e6)
#define PI ((real)3.141592653589793)
#define TAU ((real)6.283185307179586)
#define INV_PI ((real)(1.0 / 3.141592653589793))   /* python evaluates `1 / ti.pi` in f64 first */

#define STACK_CAP 256
#define MAX_LIGHTS 64
#define LIGHT_POINT 1
#define LIGHT_AREA 2

typedef struct { real x, y, z; } v3;
typedef struct { real x, y, z, w; } v4;

static inline v3 V3(real x, real y, real z) { v3 r = {x, y, z}; return r; }
static inline v3 V3s(real s) { v3 r = {s, s, s}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, real s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, real s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline real vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 vcross(v3 a, v3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline real vnorm_sqr(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
static inline real vnorm(v3 a) { return R_SQRT(vnorm_sqr(a)); }
/* Taichi Matrix.normalized(): invlen = 1 / norm; invlen * v */
static inline v3 vnormalized(v3 a) { real inv = (real)1 / vnorm(a); return vscale(a, inv); }
static inline v3 vmin3(v3 a, v3 b) { return V3(R_FMIN(a.x, b.x), R_FMIN(a.y, b.y), R_FMIN(a.z, b.z)); }
static inline v3 vmax3(v3 a, v3 b) { return V3(R_FMAX(a.x, b.x), R_FMAX(a.y, b.y), R_FMAX(a.z, b.z)); }
static inline real vavg(v3 a) { return (a.x + a.y + a.z) / (real)3; }     /* common.py:73-77 */
static inline int vany_gt0(v3 a) { return a.x > 0 || a.y > 0 || a.z > 0; }
static inline int vany_ne0(v3 a) { return a.x != 0 || a.y != 0 || a.z != 0; }

/* common.py:163-165 */
static inline real clampr(real x, real lo, real hi) { return R_FMIN(hi, R_FMAX(lo, x)); }
static inline int clampi(int x, int lo, int hi) { int t = x > lo ? x : lo; return t < hi ? t : hi; }
/* common.py:178-180 */
static inline real dot_or_zero(v3 a, v3 b) { return R_FMAX((real)0, vdot(a, b)); }
/* common.py:269-271 */
static inline real lerpr(real f, real src, real dst) { return src * ((real)1 - f) + dst * f; }
static inline v3 lerpv(real f, v3 src, v3 dst) { return vadd(vscale(src, (real)1 - f), vscale(dst, f)); }
/* common.py:247-249 */
static inline v3 reflectv(v3 I, v3 N) { return vsub(I, vscale(N, (real)2 * vdot(N, I))); }
/* common.py:252-260 */
static inline int refractv(v3 I, v3 N, real eta, v3 *T) {
    int has_r = 0;
    *T = vscale(I, 0);
    real NoI = vdot(N, I);
    real discr = (real)1 - eta * eta * ((real)1 - NoI * NoI);
    if (discr > 0) {
        has_r = 1;
        *T = vnormalized(vsub(vscale(I, eta), vscale(N, eta * NoI + R_SQRT(discr))));
    }
    return has_r;
}
/* common.py:221-225 */
static inline v3 spherical(real h, real p) {
    real ux = R_COS(p * TAU), uy = R_SIN(p * TAU);
    real r = R_SQRT(R_FMAX((real)0, (real)1 - h * h));
    return V3(r * ux, r * uy, h);
}
/* common.py:213-217 : columns (tan, bitan, nrm); returns M @ v */
static inline v3 tanspace_mul(v3 nrm, v3 v) {
    v3 up = V3((real)233., (real)666., (real)512.);
    v3 bitan = vnormalized(vcross(nrm, up));
    v3 tan = vcross(bitan, nrm);
    return V3(tan.x * v.x + bitan.x * v.y + nrm.x * v.z,
              tan.y * v.x + bitan.y * v.y + nrm.y * v.z,
              tan.z * v.x + bitan.z * v.y + nrm.z * v.z);
}

/* ------------------------------------------------------------------ */
/* integer helpers                                                      */

int32_t orc_wanghash(int32_t x) {                     /* sampling/__init__.py:9-16 */
    uint32_t value = (uint32_t)x;
    value = (value ^ 61u) ^ (value >> 16);
    value *= 9u;
    value ^= value << 4;
    value *= 0x27d4eb2du;
    value ^= value >> 15;
    return (int32_t)value;
}

int32_t orc_wanghash2(int32_t x, int32_t y) {         /* sampling/__init__.py:20-23 */
    int32_t value = orc_wanghash(x);
    value = orc_wanghash(y ^ value);
    return value;
}

int32_t orc_expand_bits(int32_t vi) {                 /* tree/lbvh.py:13-17 */
    uint32_t v = (uint32_t)vi;
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return (int32_t)v;
}

static inline int ifloor_clamped_1024(real x) {
    /* clamp(ifloor(v * 1024), 0, 1023), lbvh.py:29; NaN (degenerate extent) maps to 0 */
    real f = R_FLOOR(x * (real)1024);
    if (!(f == f)) return 0;
    if (f < 0) return 0;
    if (f > 1023) return 1023;
    return (int)f;
}

int32_t orc_morton3d(const real v[3]) {               /* tree/lbvh.py:28-30 */
    int32_t wx = orc_expand_bits(ifloor_clamped_1024(v[0]));
    int32_t wy = orc_expand_bits(ifloor_clamped_1024(v[1]));
    int32_t wz = orc_expand_bits(ifloor_clamped_1024(v[2]));
    return wx * 4 + wy * 2 + wz * 1;
}

int32_t orc_clz(int32_t x) {                          /* tree/lbvh.py:34-42 (true clz + 1; 32 for 0 AND 1) */
    int r = 0;
    for (;;) {
        int32_t f = x >> (31 - r);                    /* arithmetic shift on i32 */
        if (f == 1 || r == 31) { r += 1; break; }
        r += 1;
    }
    return r;
}

int32_t orc_count_low_bits(int32_t i) {               /* sampling/sobol.py:11-17 */
    int bits = 1;
    int32_t value = i;
    while (value & 1) { value >>= 1; bits += 1; }
    return bits;
}

real orc_construct_float(int32_t i) {                 /* sampling/sobol.py:20-29 */
    real ret = 0;
    uint32_t value = (uint32_t)i;
    real term = (real)0.5;
    while (value) {
        if (value & 0x80000000u) ret += term;
        value <<= 1;
        term *= (real)0.5;
    }
    return ret;
}

void orc_sobol_vgrid(const uint8_t *s_, const uint32_t *a_, const uint32_t *m_, int D, int L,
                     int32_t *Vout) {                 /* sampling/sobol.py:32-70 */
    /* the reference computes in int64 numpy and stores to an i32 field (bit pattern) */
    int64_t *V = (int64_t *)calloc((size_t)(L + 1) * D, sizeof(int64_t));
    int64_t *m = (int64_t *)calloc((size_t)(L + 32), sizeof(int64_t));
    for (int j = 0; j < D; j++) {
        int s; int64_t a = 0;
        if (j != 0) {
            s = s_[j]; a = a_[j];
            m[0] = 0;
            for (int i = 0; i < s; i++) m[i + 1] = m_[(size_t)j * 18 + i];
        } else {
            for (int i = 0; i <= L; i++) m[i] = 1;
            s = L;
        }
        if (L <= s) {
            for (int i = 0; i <= L; i++) V[(size_t)i * D + j] = m[i] << (32 - i);
        } else {
            for (int i = 0; i <= s; i++) V[(size_t)i * D + j] = m[i] << (32 - i);
            for (int i = s + 1; i <= L; i++) {
                int64_t vv = V[(size_t)(i - s) * D + j] ^ (V[(size_t)(i - s) * D + j] >> s);
                for (int k = 1; k < s; k++)
                    vv ^= ((a >> (s - 1 - k)) & 1) * V[(size_t)(i - k) * D + j];
                V[(size_t)i * D + j] = vv;
            }
        }
    }
    for (size_t t = 0; t < (size_t)(L + 1) * D; t++) Vout[t] = (int32_t)(uint32_t)(uint64_t)V[t];
    free(V); free(m);
}

/* ------------------------------------------------------------------ */
/* geometry                                                             */

int orc_box_intersect(const real lo[3], const real hi[3], const real o[3], const real d[3],
                      real *near_, real *far_) {      /* geometries.py:24-46 */
    real near = 0, far = INF;
    int hit = 1;
    for (int i = 0; i < 3; i++) {
        if (R_FABS(d[i]) < EPS) {
            if (o[i] < lo[i] || o[i] > hi[i]) hit = 0;
        } else {
            real i1 = (lo[i] - o[i]) / d[i];
            real i2 = (hi[i] - o[i]) / d[i];
            if (i1 > i2) { real t = i1; i1 = i2; i2 = t; }
            far = R_FMIN(far, i2);
            near = R_FMAX(near, i1);
            if (near > far) hit = 0;
        }
    }
    if (near_) *near_ = near;
    if (far_) *far_ = far;
    return hit;
}

typedef struct { int hit; real depth; real s, t; } facehit;

static inline facehit face_intersect(v3 v0, v3 v1, v3 v2, v3 ro, v3 rd) {  /* geometries.py:118-148 */
    facehit h; h.hit = 0; h.depth = INF * 2; h.s = 0; h.t = 0;
    v3 u = vsub(v1, v0);
    v3 v = vsub(v2, v0);
    v3 norm = vcross(u, v);
    real b = vdot(norm, rd);
    if (R_FABS(b) >= EPS) {
        v3 w0 = vsub(ro, v0);
        real a = -vdot(norm, w0);
        real r = a / b;
        if (r > 0) {
            v3 ip = vadd(ro, vscale(rd, r));
            real uu = vdot(u, u);
            real uv = vdot(u, v);
            real vv = vdot(v, v);
            v3 w = vsub(ip, v0);
            real wu = vdot(w, u);
            real wv = vdot(w, v);
            real D = uv * uv - uu * vv;
            h.s = (uv * wv - vv * wu) / D;
            h.t = (uv * wu - uu * wv) / D;
            if (0 <= h.s && h.s <= 1) {
                if (0 <= h.t && h.s + h.t <= 1) {
                    h.depth = r;
                    h.hit = 1;
                }
            }
        }
    }
    return h;
}

int orc_face_intersect(const real v[9], const real o[3], const real d[3], real *depth, real *s,
                       real *t) {
    facehit h = face_intersect(V3(v[0], v[1], v[2]), V3(v[3], v[4], v[5]), V3(v[6], v[7], v[8]),
                               V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]));
    *depth = h.depth; *s = h.s; *t = h.t;
    return h.hit;
}

static inline real sphere_intersect(v3 pos, real rad2, v3 ro, v3 rd) {      /* geometries.py:159-177 */
    real ret = 0;
    v3 op = vsub(pos, ro);
    real b = vdot(op, rd);
    real det = b * b + rad2 - vnorm_sqr(op);
    if (det < 0) {
        ret = 0;
    } else {
        det = R_SQRT(det);
        real t = b - det;
        if (t > EPS) {
            ret = t;
        } else {
            t = b + det;
            if (t > EPS) ret = t; else ret = 0;
        }
    }
    return ret;
}

real orc_sphere_intersect(const real pos[3], real rad2, const real o[3], const real d[3]) {
    return sphere_intersect(V3(pos[0], pos[1], pos[2]), rad2, V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]));
}

typedef struct { int hit; real depth; real u, v; } areahit;

static inline areahit area_intersect(v3 pos, v3 dirx, v3 diry, v3 ro, v3 rd) {  /* geometries.py:58-74 */
    areahit h; h.hit = 0; h.depth = INF; h.u = 0; h.v = 0;
    v3 nrm = vnormalized(vcross(dirx, diry));
    real NoD = vdot(nrm, rd);
    if (NoD > EPS) {
        h.depth = vdot(nrm, vsub(pos, ro)) / NoD;
        v3 hitdisp = vsub(vadd(ro, vscale(rd, h.depth)), pos);
        h.u = vdot(hitdisp, dirx) / vnorm_sqr(dirx);
        h.v = vdot(hitdisp, diry) / vnorm_sqr(diry);
        if (-1 < h.u && h.u < 1 && -1 < h.v && h.v < 1) h.hit = 1;
    }
    return h;
}

int orc_area_intersect(const real pos[3], const real dirx[3], const real diry[3], const real o[3],
                       const real d[3], real *depth, real uv[2]) {
    areahit h = area_intersect(V3(pos[0], pos[1], pos[2]), V3(dirx[0], dirx[1], dirx[2]),
                               V3(diry[0], diry[1], diry[2]), V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]));
    *depth = h.depth; uv[0] = h.u; uv[1] = h.v;
    return h.hit;
}

/* ------------------------------------------------------------------ */
/* microfacet.py                                                        */

static inline real pow5(real x) { return R_POW(x, (real)5); }
static inline real schlickFresnel(real cost) { return pow5(clampr((real)1 - cost, 0, 1)); }  /* :9-10 */

static inline real dielectricFresnel(real etai, real etao, real cosi) {     /* :14-27 */
    real sini = R_SQRT(R_FMAX((real)0, (real)1 - cosi * cosi));
    real sint = etao / etai * sini;
    real ret = 1;
    if (sint < 1) {
        real cost = R_SQRT(R_FMAX((real)0, (real)1 - sint * sint));
        real a1 = etai * cosi, a2 = etao * cost;
        real b1 = etao * cosi, b2 = etai * cost;
        real para = (a1 - a2) / (a1 + a2);
        real perp = (b1 - b2) / (b1 + b2);
        ret = (real)0.5 * (para * para + perp * perp);
    }
    return ret;
}

static inline real GTR1(real cosh_, real alpha) {                           /* :31-34 */
    real alpha2 = alpha * alpha;
    real t = (real)1 + (alpha2 - (real)1) * (cosh_ * cosh_);
    return (alpha2 - (real)1) / (PI * R_LOG(alpha2) * t);
}

static inline real GTR2(real cosh_, real alpha) {                           /* :38-41 */
    real alpha2 = alpha * alpha;
    real t = (real)1 + (alpha2 - (real)1) * (cosh_ * cosh_);
    return alpha2 / (PI * (t * t));
}

static inline real smithGGX(real cosi, real alpha) {                        /* :45-48 */
    real a = alpha * alpha;
    real b = cosi * cosi;
    return (real)1 / (cosi + R_SQRT(a + b - a * b));
}

static inline v3 sample_GTR1(real u, real v, real alpha) {                  /* :69-71 (NaN for alpha<1, as in the reference) */
    u = R_SQRT(R_POW(alpha, (real)2 - (real)2 * u) - (real)1) / (alpha * alpha - (real)1);
    return spherical(u, v);
}

static inline v3 sample_GTR2(real u, real v, real alpha) {                  /* :75-77 */
    u = R_SQRT(((real)1 - u) / ((real)1 - u * ((real)1 - alpha * alpha)));
    return spherical(u, v);
}

/* ------------------------------------------------------------------ */
/* materials/disney.py                                                  */

typedef struct {
    v3 basecolor;
    real metallic, roughness, specular, specularTint, subsurface, sheen, sheenTint, clearcoat,
        clearcoatGloss, transmission, ior;
    v3 tintcolor, speccolor, sheencolor;
    real alpha, clearcoatAlpha;
} disney;

static void disney_init(disney *m) {                                        /* disney.py:14-50 */
    m->tintcolor = V3s(1);
    real luminance = vdot(m->basecolor, V3((real)0.3, (real)0.6, (real)0.1));
    if (luminance > EPS) m->tintcolor = vdivs(m->basecolor, luminance);
    m->speccolor = lerpv(m->metallic,
                         vscale(lerpv(m->specularTint, V3s(1), m->tintcolor), m->specular * (real)0.08),
                         m->basecolor);
    m->sheencolor = lerpv(m->sheenTint, V3s(1), m->tintcolor);
    m->alpha = R_FMAX((real)0.001, m->roughness * m->roughness);
    m->clearcoatAlpha = lerpr(m->clearcoatGloss, (real)0.1, (real)0.001);
}

static void disney_from_params(disney *m, const real p[14]) {
    m->basecolor = V3(p[0], p[1], p[2]);
    m->metallic = p[3]; m->roughness = p[4]; m->specular = p[5]; m->specularTint = p[6];
    m->subsurface = p[7]; m->sheen = p[8]; m->sheenTint = p[9]; m->clearcoat = p[10];
    m->clearcoatGloss = p[11]; m->transmission = p[12]; m->ior = p[13];
    disney_init(m);
}

typedef struct { v3 outdir; real pdf; v3 color; } bsdfsample;

/* Choice.__call__, materials/__init__.py:37-48 */
typedef struct { real pdf, w; } choice_t;
static inline int choice_call(choice_t *c, real r) {
    int ret;
    if (c->w < r) {
        c->w /= r;
        c->pdf *= r;
        ret = 1;
    } else {
        c->w = (c->w - r) / ((real)1 - r);
        c->pdf *= (real)1 - r;
        ret = 0;
    }
    return ret;
}

static v3 disney_brdf(const disney *m, v3 normal, real sign, v3 indir, v3 outdir) {  /* disney.py:53-106 */
    real etai = 1, etao = m->ior;
    if (sign < 0) { etai = m->ior; etao = 1; }
    /* eta = etai / etao  (unused in brdf) */

    v3 halfdir = vnormalized(vadd(indir, outdir));
    real cosi = vdot(indir, normal);
    real coso = vdot(outdir, normal);
    real cosh_ = dot_or_zero(halfdir, normal);
    real cosoh = dot_or_zero(halfdir, outdir);

    v3 result = V3s(0);
    if (coso < 0) {
        if (cosi >= 0) {
            real Ds = GTR2(cosh_, m->alpha);
            real fdf = dielectricFresnel(etao, etai, cosoh);
            /* 1 / pi * basecolor * (1 - fdf) * Ds */
            v3 transmit = vscale(vscale(vscale(m->basecolor, INV_PI), (real)1 - fdf), Ds);
            result = vscale(vscale(transmit, (real)1 - m->metallic), m->transmission);
        }
    } else {
        real Fi = schlickFresnel(cosi);
        real Fo = schlickFresnel(coso);
        real Fd90 = (real)0.5 + (real)2 * (cosoh * cosoh) * m->roughness;
        real Fd = lerpr(Fi, 1, Fd90) * lerpr(Fo, 1, Fd90);

        real Fss90 = (cosoh * cosoh) * m->roughness;
        real Fss = lerpr(Fi, 1, Fss90) * lerpr(Fo, 1, Fss90);
        real ss = (real)1.25 * (Fss * ((real)1 / (cosi + coso) - (real)0.5) + (real)0.5);

        real Foh = schlickFresnel(cosoh);
        v3 Fsheen = vscale(m->sheencolor, Foh * m->sheen);

        real fdf = dielectricFresnel(etao, etai, cosoh);

        real Ds = GTR2(cosh_, m->alpha);
        v3 Fs = lerpv(Foh, m->speccolor, V3s(1));
        real Gs = smithGGX(cosi, m->alpha) * smithGGX(coso, m->alpha);

        real Dr = GTR1(cosh_, m->clearcoatAlpha);
        real Gr = smithGGX(cosi, (real)0.25) * smithGGX(coso, (real)0.25);
        real Fr = lerpr(Foh, (real)0.04, 1);

        v3 diffuse = vadd(vscale(m->basecolor, INV_PI * lerpr(m->subsurface, Fd, ss)), Fsheen);
        v3 specular = vadd(vscale(vscale(Fs, Gs), Ds), V3s((real)0.25 * m->clearcoat * Gr * Fr * Dr));
        v3 transmit = vscale(m->basecolor, INV_PI * fdf * Ds);

        result = vscale(vscale(diffuse, (real)1 - m->metallic), (real)1 - m->transmission);
        result = vadd(result, vscale(vscale(transmit, (real)1 - m->metallic), m->transmission));
        result = vadd(result, vscale(specular, (real)1 - m->transmission));
    }
    return result;
}

static bsdfsample disney_bounce(const disney *m, v3 normal, real sign, v3 indir, v3 samp) {  /* disney.py:115-233 */
    bsdfsample result;                                 /* BSDFSample.invalid() */
    result.outdir = V3s(0); result.pdf = 0; result.color = V3s(0);

    real etai = 1, etao = m->ior;
    if (sign < 0) { etai = m->ior; etao = 1; }
    real eta = etai / etao;

    real cosi = vdot(indir, normal);
    real Fi = schlickFresnel(cosi);
    v3 Fs = lerpv(Fi, m->speccolor, V3s(1));

    choice_t choice; choice.pdf = 1; choice.w = samp.z;
    real specrate = lerpr(m->transmission, lerpr(m->metallic, vavg(Fs), 1), 1);
    real coatrate = (real)0.04 * m->clearcoat;

    specrate = lerpr(specrate, (real)0.1, 1);
    if (coatrate != 0) coatrate = lerpr(coatrate, (real)0.1, 1);

    if (choice_call(&choice, coatrate)) {
        real alpha = m->clearcoatAlpha;
        v3 halfdir = tanspace_mul(normal, sample_GTR1(samp.x, samp.y, alpha));
        v3 outdir = reflectv(vneg(indir), halfdir);

        real coso = vdot(outdir, normal);
        real cosh_ = dot_or_zero(halfdir, normal);
        real cosoh = dot_or_zero(halfdir, outdir);
        if (cosoh > 0) {
            real Dr = GTR1(cosh_, alpha);
            real Foh = schlickFresnel(cosoh);
            real Fr = lerpr(Foh, (real)0.04, 1);

            result.outdir = outdir;
            real partial = m->clearcoat * Fr * coso / cosoh;
            result.pdf = Dr * partial;
            result.color = V3s(partial / choice.pdf);
        } else {
            result.pdf = 0;
            result.color = V3s(0);
        }
    } else if (choice_call(&choice, specrate)) {
        real alpha = m->alpha;
        v3 halfdir = tanspace_mul(normal, sample_GTR2(samp.x, samp.y, alpha));
        v3 outdir = reflectv(vneg(indir), halfdir);

        real coso = dot_or_zero(outdir, normal);
        real cosh_ = dot_or_zero(halfdir, normal);
        real cosoh = dot_or_zero(halfdir, outdir);
        if (cosoh > 0 && coso > 0 && cosh_ > 0) {
            real Ds = GTR2(cosh_, alpha);

            if (choice_call(&choice, m->transmission)) {
                real fdf = dielectricFresnel(etao, etai, cosoh);
                real reflrate = lerpr(fdf, (real)0.2, 1);

                if (choice_call(&choice, reflrate)) {
                    result.outdir = outdir;
                    result.pdf = Ds * fdf;
                    result.color = vdivs(vscale(vscale(m->basecolor, fdf), m->transmission), choice.pdf);
                } else {
                    v3 T;
                    int has_r = refractv(vneg(indir), halfdir, eta, &T);
                    if (has_r) {
                        result.outdir = T;
                        result.pdf = Ds * ((real)1 - fdf);
                        result.color = vdivs(vscale(vscale(m->basecolor, (real)1 - fdf), m->transmission), choice.pdf);
                    }
                }
            } else {
                real Foh = schlickFresnel(cosoh);
                v3 Fs2 = lerpv(Foh, m->speccolor, V3s(1));

                result.outdir = outdir;
                real partial = (real)0.5 / (cosoh * smithGGX(coso, alpha));
                result.pdf = Ds * vavg(Fs2) * partial;
                result.color = vdivs(vscale(vscale(Fs2, partial), (real)1 - m->transmission), choice.pdf);
            }
        } else {
            result.pdf = 0;
            result.color = V3s(0);
        }
    } else {
        v3 outdir = tanspace_mul(normal, spherical(R_SQRT(samp.x), samp.y));

        v3 halfdir = vnormalized(vadd(indir, outdir));
        real cosi2 = vdot(indir, normal);
        real coso = vdot(outdir, normal);
        real cosoh = dot_or_zero(halfdir, outdir);

        real Fi2 = schlickFresnel(cosi2);
        real Fo = schlickFresnel(coso);
        real Fd90 = (real)0.5 + (real)2 * (cosoh * cosoh) * m->roughness;
        real Fd = lerpr(Fi2, 1, Fd90) * lerpr(Fo, 1, Fd90);

        real Fss90 = (cosoh * cosoh) * m->roughness;
        real Fss = lerpr(Fi2, 1, Fss90) * lerpr(Fo, 1, Fss90);
        real ss = (real)1.25 * (Fss * ((real)1 / (cosi2 + coso) - (real)0.5) + (real)0.5);

        real Foh = schlickFresnel(cosoh);
        v3 Fsheen = vscale(m->sheencolor, Foh * m->sheen);

        v3 diffuse = vadd(vscale(m->basecolor, INV_PI * lerpr(m->subsurface, Fd, ss)), Fsheen);

        result.outdir = outdir;
        result.pdf = INV_PI;
        result.color = vdivs(vscale(vscale(vscale(diffuse, PI), (real)1 - m->metallic), (real)1 - m->transmission), choice.pdf);
    }
    return result;
}

void orc_disney_brdf(const real params[14], const real normal[3], real sign, const real indir[3],
                     const real outdir[3], real out_rgb[3]) {
    disney m; disney_from_params(&m, params);
    v3 r = disney_brdf(&m, V3(normal[0], normal[1], normal[2]), sign, V3(indir[0], indir[1], indir[2]),
                       V3(outdir[0], outdir[1], outdir[2]));
    out_rgb[0] = r.x; out_rgb[1] = r.y; out_rgb[2] = r.z;
}

void orc_disney_bounce(const real params[14], const real normal[3], real sign, const real indir[3],
                       const real samp[3], real out[7]) {
    disney m; disney_from_params(&m, params);
    bsdfsample b = disney_bounce(&m, V3(normal[0], normal[1], normal[2]), sign,
                                 V3(indir[0], indir[1], indir[2]), V3(samp[0], samp[1], samp[2]));
    out[0] = b.outdir.x; out[1] = b.outdir.y; out[2] = b.outdir.z; out[3] = b.pdf;
    out[4] = b.color.x; out[5] = b.color.y; out[6] = b.color.z;
}

real orc_power_heuristic(real a, real b) {                                   /* path.py:11-15 */
    a = clampr(a, EPS, INF); a = a * a;
    b = clampr(b, EPS, INF); b = b * b;
    return a / (a + b);
}

/* ------------------------------------------------------------------ */
/* context                                                              */

typedef struct { int nx, ny, base; } imginfo;

struct orc_ctx {
    int nthreads;
    /* film, filmtable.py:12-14 : 3 passes of float4[nx*ny], element x*ny + y */
    int nx, ny, wx0, wx1;
    int sw, spitch;             /* striped window: columns x >= wx0 with (x - wx0) % spitch < sw; sw = 0: all */
    v4 *film[3];
    /* model, model.py:11-14 */
    int nfaces;
    real *vertices;            /* [3n][8] */
    int32_t *mtlids;
    /* materials, mtllib.py:44-57 */
    int nmat;
    real *mfac;                /* [m][12][4] */
    int32_t *mtex;             /* [m][12]    */
    /* images, image.py:10-17 */
    int nimg; imginfo img[64];
    v4 *texels; size_t ntexels;
    /* tree, tree/lbvh.py:47-58 */
    int n;
    real *bmin, *bmax;         /* [n][3] */
    int32_t *bready, *child /*[n][2]*/, *leaf, *mc, *id;
    /* camera */
    real v2w[16];
    /* lights, light/__init__.py:13-19 */
    int nlights;
    v3 lcolor[MAX_LIGHTS], lpos[MAX_LIGHTS];
    real laxes[MAX_LIGHTS][9], lsize[MAX_LIGHTS];
    int ltype[MAX_LIGHTS];
    /* world, light/world.py:10-16 */
    v4 wfac; int wtex;
    /* sobol, sampling/sobol.py:75-90 */
    int sdim, srows;
    int32_t *sV, *sX; real *sP; int32_t stime;
    orc_counters cnt;
};

orc_ctx *orc_create(void) {
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
    c->nthreads = 1;
    /* default light, light/__init__.py:22-28 */
    c->lcolor[0] = V3(32, 32, 32);
    c->lpos[0] = V3(1, 2, 3);
    c->lsize[0] = (real)0.5;
    c->ltype[0] = LIGHT_POINT;
    c->nlights = 1;
    /* world default fac 0.1, light/world.py:14-16; tex: documented deviation Q6 (-1, not 0) */
    c->wfac.x = c->wfac.y = c->wfac.z = c->wfac.w = (real)0.1;
    c->wtex = -1;
    /* identity camera until set */
    for (int i = 0; i < 4; i++) c->v2w[i * 4 + i] = 1;
    return c;
}

void orc_destroy(orc_ctx *c) {
    if (!c) return;
    for (int p = 0; p < 3; p++) free(c->film[p]);
    free(c->vertices); free(c->mtlids); free(c->mfac); free(c->mtex); free(c->texels);
    free(c->bmin); free(c->bmax); free(c->bready); free(c->child); free(c->leaf); free(c->mc); free(c->id);
    free(c->sV); free(c->sX); free(c->sP);
    free(c);
}

void orc_set_threads(orc_ctx *c, int n) { c->nthreads = n > 0 ? n : 1; }

void orc_set_size(orc_ctx *c, int nx, int ny) {
    if (nx != c->nx || ny != c->ny) {
        for (int p = 0; p < 3; p++) {
            free(c->film[p]);
            c->film[p] = (v4 *)calloc((size_t)nx * ny, sizeof(v4));
        }
    }
    c->nx = nx; c->ny = ny; c->wx0 = 0; c->wx1 = nx; c->sw = 0; c->spitch = 1;
}

void orc_set_window(orc_ctx *c, int x0, int x1) { c->wx0 = x0; c->wx1 = x1; c->sw = 0; c->spitch = 1; }
/* the share of rank `index` of `modulo` when the film is dealt out in stripes of `width` columns */
void orc_set_stripes(orc_ctx *c, int width, int index, int modulo) {
    c->wx0 = index * width; c->wx1 = c->nx; c->sw = width; c->spitch = width * modulo;
}
static int window_has(const orc_ctx *c, int i) { return c->sw == 0 || (i - c->wx0) % c->spitch < c->sw; }

int orc_load_model(orc_ctx *c, const float *verts, const int32_t *mtlids, int n) {  /* model.py:54-60 */
    free(c->vertices); free(c->mtlids);
    c->vertices = (real *)malloc((size_t)n * 24 * sizeof(real));
    c->mtlids = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    for (size_t i = 0; i < (size_t)n * 24; i++) c->vertices[i] = (real)verts[i];
    for (int i = 0; i < n; i++) c->mtlids[i] = mtlids ? mtlids[i] : -1;
    c->nfaces = n;
    return 0;
}

int orc_load_materials(orc_ctx *c, const float *fac, const int32_t *tex, int m) {   /* mtllib.py:58-77 */
    free(c->mfac); free(c->mtex);
    c->mfac = (real *)malloc((size_t)m * 48 * sizeof(real));
    c->mtex = (int32_t *)malloc((size_t)m * 12 * sizeof(int32_t));
    for (size_t i = 0; i < (size_t)m * 48; i++) c->mfac[i] = (real)fac[i];
    memcpy(c->mtex, tex, (size_t)m * 12 * sizeof(int32_t));
    c->nmat = m;
    return 0;
}

void orc_reset_images(orc_ctx *c) { c->nimg = 0; c->ntexels = 0; }          /* image.py:90-92 */

int orc_add_image(orc_ctx *c, const float *rgba, int nx, int ny) {           /* image.py:51-88 */
    if (c->nimg >= 64) return -1;
    int id = c->nimg++;
    size_t base = c->ntexels;
    c->ntexels += (size_t)nx * ny;
    c->texels = (v4 *)realloc(c->texels, c->ntexels * sizeof(v4));
    for (size_t t = 0; t < (size_t)nx * ny; t++) {
        c->texels[base + t].x = (real)rgba[t * 4 + 0];
        c->texels[base + t].y = (real)rgba[t * 4 + 1];
        c->texels[base + t].z = (real)rgba[t * 4 + 2];
        c->texels[base + t].w = (real)rgba[t * 4 + 3];
    }
    c->img[id].nx = nx; c->img[id].ny = ny; c->img[id].base = (int)base;
    return id;
}

void orc_set_camera_v2w(orc_ctx *c, const float v2w[16]) {
    for (int i = 0; i < 16; i++) c->v2w[i] = (real)v2w[i];
}

void orc_clear_lights(orc_ctx *c) { c->nlights = 0; }                        /* light/__init__.py:31-32 */

int orc_add_light(orc_ctx *c, int type, const float color[3], const float pos[3],
                  const float axes[9], float size) {                         /* light/__init__.py:34-49 */
    int i = c->nlights;
    if (i >= MAX_LIGHTS) return -1;
    c->ltype[i] = type;
    c->lcolor[i] = V3(color[0], color[1], color[2]);
    c->lpos[i] = V3(pos[0], pos[1], pos[2]);
    for (int k = 0; k < 9; k++) c->laxes[i][k] = (real)axes[k];
    c->lsize[i] = (real)size;
    c->nlights = i + 1;
    return i;
}

void orc_set_world(orc_ctx *c, const float fac[4], int tex) {                /* light/world.py:18-20 */
    c->wfac.x = fac[0]; c->wfac.y = fac[1]; c->wfac.z = fac[2]; c->wfac.w = fac[3];
    c->wtex = tex;
}

/* ------------------------------------------------------------------ */
/* sobol                                                                */

void orc_sobol_init(orc_ctx *c, const int32_t *V, int rows, int D) {
    free(c->sV); free(c->sX); free(c->sP);
    c->sV = (int32_t *)malloc((size_t)rows * D * sizeof(int32_t));
    memcpy(c->sV, V, (size_t)rows * D * sizeof(int32_t));
    c->sX = (int32_t *)calloc(D, sizeof(int32_t));
    c->sP = (real *)calloc(D, sizeof(real));
    c->sdim = D; c->srows = rows; c->stime = 0;
}

void orc_sobol_update(orc_ctx *c) {                                          /* sobol.py:99-105 */
    int i = orc_count_low_bits(c->stime);
    c->stime += 1;
    for (int j = 0; j < c->sdim; j++) {
        c->sX[j] ^= c->sV[(size_t)i * c->sdim + j];
        c->sP[j] = orc_construct_float(c->sX[j]);
    }
}

void orc_sobol_reset(orc_ctx *c, int skip) {                                 /* sobol.py:92-97 */
    c->stime = 0;
    memset(c->sX, 0, (size_t)c->sdim * sizeof(int32_t));
    for (int i = 0; i < skip; i++) orc_sobol_update(c);
}

int orc_sobol_get(orc_ctx *c, int32_t *X, real *P) {
    if (X) memcpy(X, c->sX, (size_t)c->sdim * sizeof(int32_t));
    if (P) memcpy(P, c->sP, (size_t)c->sdim * sizeof(real));
    return c->stime;
}

/* SobolSampler.Proxy, sobol.py:107-125 : i32 counter that wraps, floor-mod indexing */
typedef struct { const orc_ctx *c; int32_t i; uint64_t draws; } rng_t;
static inline real rng_random(rng_t *r) {
    int32_t dim = r->c->sdim;
    int32_t k = r->i % dim;
    if (k < 0) k += dim;                                /* Python/Taichi floor-mod */
    real ret = r->c->sP[k];
    r->i = (int32_t)((uint32_t)r->i + 1u);
    r->draws++;
    return ret;
}
/* common.py:303-309 : left-to-right */
static inline v3 random3(rng_t *r) { real a = rng_random(r), b = rng_random(r), cc = rng_random(r); return V3(a, b, cc); }

/* ------------------------------------------------------------------ */
/* model / tree                                                         */

static inline v3 vert_pos(const orc_ctx *c, int v) { const real *p = c->vertices + (size_t)v * 8; return V3(p[0], p[1], p[2]); }
static inline v3 vert_nrm(const orc_ctx *c, int v) { const real *p = c->vertices + (size_t)v * 8; return V3(p[3], p[4], p[5]); }

typedef struct { int32_t code, id; } mcpair;
static int mc_cmp(const void *a, const void *b) {
    const mcpair *x = (const mcpair *)a, *y = (const mcpair *)b;
    if (x->code != y->code) return x->code < y->code ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id);        /* stable tie-break (deviation Q14) */
}

static int findSplit(const orc_ctx *c, int l, int r) {                       /* lbvh.py:62-89 */
    int m = 0;
    int32_t lc = c->mc[l], rc = c->mc[r];
    if (lc == rc) {
        m = (l + r) >> 1;
    } else {
        int cp = orc_clz(lc ^ rc);
        m = l;
        int s = r - l;
        for (;;) {
            s += 1;
            s >>= 1;
            int n = m + s;
            if (n < r) {
                int32_t nc = c->mc[n];
                int sp = orc_clz(lc ^ nc);
                if (sp > cp) m = n;
            }
            if (s <= 1) break;
        }
    }
    return m;
}

static void determineRange(const orc_ctx *c, int n, int i, int *lo, int *ro) {  /* lbvh.py:93-146 */
    int l = 0, r = n - 1;
    if (i != 0) {
        int32_t ic = c->mc[i];
        int32_t lc = c->mc[i - 1];
        int32_t rc = c->mc[i + 1];
        if (lc == ic && ic == rc) {
            l = i;
            while (i < n - 1) {
                i += 1;
                if (i >= n - 1) break;
                if (c->mc[i] != c->mc[i + 1]) break;
            }
            r = i;
        } else {
            int ld = orc_clz(ic ^ lc);
            int rd = orc_clz(ic ^ rc);
            int d = -1;
            if (rd > ld) d = 1;
            int delta_min = ld < rd ? ld : rd;
            int lmax = 2;
            int delta = -1;
            int itmp = i + d * lmax;
            if (0 <= itmp && itmp < n) delta = orc_clz(ic ^ c->mc[itmp]);
            while (delta > delta_min) {
                lmax <<= 1;
                itmp = i + d * lmax;
                delta = -1;
                if (0 <= itmp && itmp < n) delta = orc_clz(ic ^ c->mc[itmp]);
            }
            int s = 0;
            int t = lmax >> 1;
            while (t > 0) {
                itmp = i + (s + t) * d;
                delta = -1;
                if (0 <= itmp && itmp < n) delta = orc_clz(ic ^ c->mc[itmp]);
                if (delta > delta_min) s += t;
                t >>= 1;
            }
            l = i; r = i + s * d;
            if (d < 0) { int tmp = l; l = r; r = tmp; }
        }
    }
    *lo = l; *ro = r;
}

static void face_bbox(const orc_ctx *c, int f, v3 *lo, v3 *hi) {             /* lbvh.py:155-158 */
    v3 v0 = vert_pos(c, f * 3), v1 = vert_pos(c, f * 3 + 1), v2 = vert_pos(c, f * 3 + 2);
    *lo = vmin3(vmin3(v0, v1), v2);
    *hi = vmax3(vmax3(v0, v1), v2);
}

static v3 face_center(const orc_ctx *c, int f) {                             /* lbvh.py:161-165 */
    v3 v0 = vert_pos(c, f * 3), v1 = vert_pos(c, f * 3 + 1), v2 = vert_pos(c, f * 3 + 2);
    return vdivs(vadd(vadd(v0, v1), v2), (real)3);
}

static int node_bbox(const orc_ctx *c, int n, int i, v3 *lo, v3 *hi) {       /* lbvh.py:234-248 */
    if (i < n) { face_bbox(c, c->leaf[i], lo, hi); return 1; }
    i -= n;
    *lo = V3(c->bmin[i * 3], c->bmin[i * 3 + 1], c->bmin[i * 3 + 2]);
    *hi = V3(c->bmax[i * 3], c->bmax[i * 3 + 1], c->bmax[i * 3 + 2]);
    return c->bready[i];
}

int orc_build_tree(orc_ctx *c) {                                             /* lbvh.py:297-305 */
    int n = c->nfaces;
    free(c->bmin); free(c->bmax); free(c->bready); free(c->child); free(c->leaf); free(c->mc); free(c->id);
    c->n = n;
    size_t cap = (size_t)(n > 0 ? n : 1);
    c->bmin = (real *)calloc(cap * 3, sizeof(real));
    c->bmax = (real *)calloc(cap * 3, sizeof(real));
    c->bready = (int32_t *)calloc(cap, sizeof(int32_t));
    c->child = (int32_t *)calloc(cap * 2, sizeof(int32_t));
    c->leaf = (int32_t *)calloc(cap, sizeof(int32_t));
    c->mc = (int32_t *)calloc(cap, sizeof(int32_t));
    c->id = (int32_t *)calloc(cap, sizeof(int32_t));

    /* genMortonCodes, lbvh.py:169-183 */
    v3 bmin = V3s(INF), bmax = V3s(-INF);
    for (int i = 0; i < n; i++) {
        v3 center = face_center(c, i);
        bmax = vmax3(bmax, center);
        bmin = vmin3(bmin, center);
    }
    mcpair *arr = (mcpair *)malloc(cap * sizeof(mcpair));
    for (int i = 0; i < n; i++) {
        v3 center = face_center(c, i);
        v3 ext = vsub(bmax, bmin);
        v3 d = vsub(center, bmin);
        real coord[3] = { d.x / ext.x, d.y / ext.y, d.z / ext.z };
        arr[i].code = orc_morton3d(coord);
        arr[i].id = i;
    }
    /* sortMortonCodes, lbvh.py:204-208 */
    qsort(arr, (size_t)n, sizeof(mcpair), mc_cmp);
    for (int i = 0; i < n; i++) { c->mc[i] = arr[i].code; c->id[i] = arr[i].id; }
    free(arr);

    /* genHierarchy, lbvh.py:212-231 */
    for (int i = 0; i < n; i++) c->leaf[i] = c->id[i];
    for (int i = 0; i < n - 1; i++) {
        int l, r;
        determineRange(c, n, i, &l, &r);
        int split = findSplit(c, l, r);
        int lhs = split;
        if (lhs != l) lhs += n;
        int rhs = split + 1;
        if (rhs != r) rhs += n;
        c->child[i * 2 + 0] = lhs;
        c->child[i * 2 + 1] = rhs;
    }

    /* genAABBs, lbvh.py:251-294 (Jacobi-style: a substep only sees the previous substep's
     * ready flags would be the strict level-synchronous reading; the reference's kernel is a
     * parallel for with no ordering guarantee, so in-place sequential sweeps -- which converge
     * to the same boxes since min/max are exact -- are used here) */
    int count = 1;
    for (int i = 0; i < n; i++) c->bready[i] = 0;
    for (;;) {
        for (int i = 0; i < n - 1; i++) {
            if (c->bready[i]) continue;
            /* bound-check corrupted hierarchies instead of reading out of range */
            int c0 = c->child[i * 2], c1 = c->child[i * 2 + 1];
            if (c0 < 0 || c0 >= 2 * n - 1 || c1 < 0 || c1 >= 2 * n - 1) return -1;
            v3 lo1, hi1, lo2, hi2;
            int r1 = node_bbox(c, n, c0, &lo1, &hi1);
            int r2 = node_bbox(c, n, c1, &lo2, &hi2);
            if (r1 == 1 && r2 == 1) {
                v3 lo = vmin3(lo1, lo2), hi = vmax3(hi1, hi2);
                c->bmin[i * 3] = lo.x; c->bmin[i * 3 + 1] = lo.y; c->bmin[i * 3 + 2] = lo.z;
                c->bmax[i * 3] = hi.x; c->bmax[i * 3 + 1] = hi.y; c->bmax[i * 3 + 2] = hi.z;
                c->bready[i] = 1;
            }
        }
        int all_ready = 1;
        for (int i = 0; i < n - 1; i++) if (c->bready[i] == 0) all_ready = 0;
        if (all_ready) break;
        count += 1;
        if (count > 64) return -1;                      /* 'AABB step never stop! hierarchy corrupted?' */
    }
    return 0;
}

int orc_get_tree(orc_ctx *c, int32_t *child, int32_t *leaf, float *bmin, float *bmax, int32_t *mc) {
    int n = c->n;
    if (child) memcpy(child, c->child, (size_t)(n - 1) * 2 * sizeof(int32_t));
    if (leaf) memcpy(leaf, c->leaf, (size_t)n * sizeof(int32_t));
    for (int i = 0; i < (n - 1) * 3; i++) { if (bmin) bmin[i] = (float)c->bmin[i]; if (bmax) bmax[i] = (float)c->bmax[i]; }
    if (mc) memcpy(mc, c->mc, (size_t)n * sizeof(int32_t));
    return n;
}

typedef struct { int hit; real depth; int index; real u, v; } bvhhit;

static bvhhit bvh_intersect(const orc_ctx *c, v3 ro, v3 rd, int avoid, orc_counters *cnt) {  /* lbvh.py:314-347 */
    int n = c->n;
    int32_t stack[STACK_CAP];
    int sp = 0;
    stack[sp++] = n;

    bvhhit ret; ret.hit = 0; ret.depth = INF; ret.index = -1; ret.u = 0; ret.v = 0;
    real o[3] = { ro.x, ro.y, ro.z }, d[3] = { rd.x, rd.y, rd.z };
    cnt->rays++;

    int ntimes = 0;
    while (ntimes < n && sp != 0) {
        int curr = stack[--sp];

        if (curr < n) {
            int index = c->leaf[curr];
            if (index != avoid) {
                cnt->n_leaf++;
                facehit h = face_intersect(vert_pos(c, index * 3), vert_pos(c, index * 3 + 1),
                                           vert_pos(c, index * 3 + 2), ro, rd);
                if (h.hit != 0 && h.depth < ret.depth) {
                    ret.depth = h.depth;
                    ret.index = index;
                    ret.u = h.s; ret.v = h.t;
                    ret.hit = 1;
                }
            }
            continue;
        }

        int i = curr - n;
        cnt->n_int++;
        if (orc_box_intersect(c->bmin + i * 3, c->bmax + i * 3, o, d, 0, 0) == 0) continue;

        ntimes += 1;
        if (sp + 2 > STACK_CAP) break;                  /* never reached for sane trees */
        stack[sp++] = c->child[i * 2 + 0];
        stack[sp++] = c->child[i * 2 + 1];
        if ((uint64_t)sp > cnt->max_stack) cnt->max_stack = (uint64_t)sp;
    }
    return ret;
}

int orc_intersect(orc_ctx *c, const real o[3], const real d[3], int avoid, real out[4]) {
    orc_counters tmp; memset(&tmp, 0, sizeof tmp);
    bvhhit h = bvh_intersect(c, V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), avoid, &tmp);
    out[0] = h.depth; out[1] = (real)h.index; out[2] = h.u; out[3] = h.v;
    return h.hit;
}

/* ------------------------------------------------------------------ */
/* textures: image.py:137-148, common.py:183-192                        */

static inline int pymod(int a, int b) { int r = a % b; if (r < 0) r += b; return r; }

static inline v4 image_texel(const orc_ctx *c, int id, int x, int y) {       /* image.py:138-143, :18-20 */
    const imginfo *im = &c->img[id];
    x = pymod(x, im->nx);
    y = pymod(y, im->ny);
    return c->texels[(size_t)im->base + (size_t)x * im->ny + y];
}

static inline v4 v4scale(v4 a, real s) { v4 r = { a.x * s, a.y * s, a.z * s, a.w * s }; return r; }
static inline v4 v4add(v4 a, v4 b) { v4 r = { a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w }; return r; }
static inline v4 v4mul(v4 a, v4 b) { v4 r = { a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w }; return r; }

static v4 image_sample(const orc_ctx *c, int id, real x, real y) {           /* image.py:145-148 + bilerp */
    const imginfo *im = &c->img[id];
    real px = x * (real)(im->nx - 1), py = y * (real)(im->ny - 1);
    int Ix = (int)R_FLOOR(px), Iy = (int)R_FLOOR(py);
    real x0 = px - (real)Ix, x1 = py - (real)Iy;         /* x = p - I */
    real y0 = (real)1 - x0, y1 = (real)1 - x1;           /* y = 1 - x */
    v4 r = v4scale(v4scale(image_texel(c, id, Ix + 1, Iy + 1), x0), x1);
    r = v4add(r, v4scale(v4scale(image_texel(c, id, Ix + 1, Iy), x0), y1));
    r = v4add(r, v4scale(v4scale(image_texel(c, id, Ix, Iy), y0), y1));
    r = v4add(r, v4scale(v4scale(image_texel(c, id, Ix, Iy + 1), y0), x1));
    return r;
}

/* ParameterPair.get, mtllib.py:30-38 */
static v4 param_get(const orc_ctx *c, int k, int mtlid, real tu, real tv, real dflt) {
    v4 fac = { dflt, dflt, dflt, dflt };
    if (mtlid != -1) {
        const real *f = c->mfac + ((size_t)mtlid * 12 + k) * 4;
        fac.x = f[0]; fac.y = f[1]; fac.z = f[2]; fac.w = f[3];
        int texid = c->mtex[(size_t)mtlid * 12 + k];
        if (texid != -1) fac = v4mul(fac, image_sample(c, texid, tu, tv));
    }
    return fac;
}

static void material_get(const orc_ctx *c, int mtlid, real tu, real tv, disney *m) {  /* mtllib.py:79-95 */
    v4 b = param_get(c, 0, mtlid, tu, tv, (real)0.8);
    m->basecolor = V3(b.x, b.y, b.z);
    m->metallic = param_get(c, 1, mtlid, tu, tv, (real)0.0).x;
    m->roughness = param_get(c, 2, mtlid, tu, tv, (real)0.4).x;
    m->specular = param_get(c, 3, mtlid, tu, tv, (real)0.5).x;
    m->specularTint = param_get(c, 4, mtlid, tu, tv, (real)0.4).x;
    m->subsurface = param_get(c, 5, mtlid, tu, tv, (real)0.0).x;
    m->sheen = param_get(c, 6, mtlid, tu, tv, (real)0.0).x;
    m->sheenTint = param_get(c, 7, mtlid, tu, tv, (real)0.4).x;
    m->clearcoat = param_get(c, 8, mtlid, tu, tv, (real)0.0).x;
    m->clearcoatGloss = param_get(c, 9, mtlid, tu, tv, (real)0.5).x;
    m->transmission = param_get(c, 10, mtlid, tu, tv, (real)0.0).x;
    m->ior = param_get(c, 11, mtlid, tu, tv, (real)1.45).x;
    disney_init(m);
}

/* ------------------------------------------------------------------ */
/* lights                                                               */

typedef struct { int hit; real dis, pdf; v3 color; } lighthit;

static inline v3 axes_mul(const real *A, v3 v) {        /* 3x3 row-major @ v */
    return V3(A[0] * v.x + A[1] * v.y + A[2] * v.z,
              A[3] * v.x + A[4] * v.y + A[5] * v.z,
              A[6] * v.x + A[7] * v.y + A[8] * v.z);
}

static lighthit lights_hit(const orc_ctx *c, v3 ro, v3 rd) {                 /* light/__init__.py:51-81 */
    lighthit ret; ret.hit = 0; ret.dis = INF; ret.pdf = 0; ret.color = V3s(0);
    for (int i = 0; i < c->nlights; i++) {
        int type = c->ltype[i];
        v3 color = c->lcolor[i];
        v3 pos = c->lpos[i];
        real size = c->lsize[i];
        const real *axes = c->laxes[i];

        real t = 0, area = 0;
        if (type == LIGHT_POINT) {
            t = sphere_intersect(pos, size * size, ro, rd);
            area = PI * (size * size);
        } else if (type == LIGHT_AREA) {
            v3 dirx = axes_mul(axes, V3(size, 0, 0));
            v3 diry = axes_mul(axes, V3(0, size, 0));
            areahit h = area_intersect(pos, dirx, diry, ro, rd);
            if (h.hit) {
                t = h.depth;
                area = (real)4 * (size * size);
            }
        }
        if (0 < t && t < ret.dis) {
            ret.dis = t;
            ret.pdf = ret.dis * ret.dis / area;
            ret.color = color;
            ret.hit = 1;
            break;
        }
    }
    return ret;
}

typedef struct { real dis; v3 dir; real pdf; v3 color; } lightsample;

static lightsample lights_sample(const orc_ctx *c, v3 hitpos, v3 samp) {     /* light/__init__.py:83-121 */
    lightsample ret; ret.dis = INF; ret.dir = V3s(0); ret.pdf = 0; ret.color = V3s(0);
    if (c->nlights != 0) {
        int i = clampi((int)R_FLOOR(samp.z * (real)c->nlights), 0, c->nlights);
        if (i >= MAX_LIGHTS) i = MAX_LIGHTS - 1;        /* unreachable: samp.z < 1 */
        int type = c->ltype[i];
        v3 color = c->lcolor[i];
        v3 pos = c->lpos[i];
        real size = c->lsize[i];
        const real *axes = c->laxes[i];

        v3 litpos = V3s(INF);
        v3 norm = V3s(0);
        real area = 0;

        if (type == LIGHT_POINT) {
            v3 disp = spherical(samp.x, samp.y);
            litpos = vadd(pos, vscale(disp, size));
            area = PI * (size * size);
        } else if (type == LIGHT_AREA) {
            v3 disp = axes_mul(axes, V3(samp.x * (real)2 - (real)1, samp.y * (real)2 - (real)1, 0));
            norm = axes_mul(axes, V3(0, 0, 1));
            litpos = vadd(pos, vscale(disp, size));
            area = (real)4 * (size * size);
        }

        v3 toli = vsub(litpos, hitpos);
        real dis = vnorm(toli);
        v3 dir = vdivs(toli, dis);
        real pdf = dis * dis / area;
        color = vdivs(color, pdf);
        if (vany_ne0(norm)) color = vscale(color, dot_or_zero(norm, dir));
        ret.dis = dis; ret.dir = dir; ret.pdf = pdf; ret.color = color;
    }
    return ret;
}

static inline void dir2tex(v3 dir, real *s, real *t) {                       /* common.py:234-239 */
    v3 dn = vnormalized(dir);
    *s = R_ATAN2(dn.z, dn.x) / PI * (real)0.5 + (real)0.5;
    *t = R_ATAN2(dn.y, R_SQRT(dn.x * dn.x + dn.z * dn.z)) / PI + (real)0.5;
}

static v3 world_at(const orc_ctx *c, v3 dir) {                               /* light/world.py:22-29 */
    v4 fac = c->wfac;
    int texid = c->wtex;
    if (texid != -1) {
        real ny = dir.z, nz = -dir.y;                   /* dir.y, dir.z = dir.z, -dir.y */
        dir.y = ny; dir.z = nz;
        real s, t;
        dir2tex(dir, &s, &t);
        fac = v4mul(fac, image_sample(c, texid, s, t));
    }
    return V3(fac.x, fac.y, fac.z);
}

/* ------------------------------------------------------------------ */
/* camera                                                               */

static inline void mat4_mulv(const real *M, real x, real y, real z, real w, real out[4]) {
    for (int i = 0; i < 4; i++) out[i] = M[i * 4] * x + M[i * 4 + 1] * y + M[i * 4 + 2] * z + M[i * 4 + 3] * w;
}

static void camera_generate(const orc_ctx *c, real x, real y, v3 *ro, v3 *rd) {  /* camera.py:34-39 */
    real a[4], b[4];
    mat4_mulv(c->v2w, x, y, (real)-1.0, (real)1.0, a);
    mat4_mulv(c->v2w, x, y, (real)1.0, (real)1.0, b);
    v3 o = V3(a[0] / a[3], a[1] / a[3], a[2] / a[3]);   /* V43, common.py:48-49 */
    v3 o1 = V3(b[0] / b[3], b[1] / b[3], b[2] / b[3]);
    *ro = o;
    *rd = vnormalized(vsub(o1, o));
}

void orc_camera_generate(orc_ctx *c, real x, real y, real o[3], real d[3]) {
    v3 ro, rd; camera_generate(c, x, y, &ro, &rd);
    o[0] = ro.x; o[1] = ro.y; o[2] = ro.z; d[0] = rd.x; d[1] = rd.y; d[2] = rd.z;
}

/* ------------------------------------------------------------------ */
/* shading geometry, model.py:88-101 + geometries.py:96-108             */

/* Face.normal / Face.texcoord, geometries.py:96-108 */
static inline void face_shading(v3 vn0, v3 vn1, v3 vn2, const real *t0, const real *t1, const real *t2, real u, real v,
                                v3 *nrm, real *tu, real *tv) {
    real wx = (real)1 - u - v, wy = u, wz = v;          /* w = V(1 - u - v, u, v) */
    *nrm = vnormalized(vadd(vadd(vscale(vn0, wx), vscale(vn1, wy)), vscale(vn2, wz)));
    *tu = wx * t0[0] + wy * t1[0] + wz * t2[0];
    *tv = wx * t0[1] + wy * t1[1] + wz * t2[1];
}

static void get_geometries(const orc_ctx *c, const bvhhit *hit, v3 ro, v3 rd, v3 *hitpos, v3 *normal,
                           disney *material) {
    int f = hit->index;
    real u = hit->u, v = hit->v;
    v3 vn0 = vert_nrm(c, f * 3), vn1 = vert_nrm(c, f * 3 + 1), vn2 = vert_nrm(c, f * 3 + 2);
    const real *t0 = c->vertices + (size_t)(f * 3) * 8 + 6;
    const real *t1 = c->vertices + (size_t)(f * 3 + 1) * 8 + 6;
    const real *t2 = c->vertices + (size_t)(f * 3 + 2) * 8 + 6;
    v3 nrm;
    real tu, tv;
    face_shading(vn0, vn1, vn2, t0, t1, t2, u, v, &nrm, &tu, &tv);
    *hitpos = vadd(ro, vscale(rd, hit->depth));

    real sign = -vdot(rd, nrm);
    if (sign < 0) nrm = vneg(nrm);
    *normal = nrm;
    material_get(c, c->mtlids[f], tu, tv, material);
}

/* ------------------------------------------------------------------ */
/* engine/path.py:18-64                                                 */

static v3 path_trace(const orc_ctx *c, v3 ro, v3 rd, rng_t *rng, orc_counters *cnt) {
    int avoid = -1;
    int depth = 0;
    v3 result = V3s(0);
    v3 throughput = V3s(1);
    real last_brdf_pdf = 0;

    while (depth < 5 && vany_gt0(throughput) && vany_ne0(rd)) {
        depth += 1;
        cnt->bounces++;

        rd = vnormalized(rd);
        bvhhit hit = bvh_intersect(c, ro, rd, avoid, cnt);

        lighthit lit = lights_hit(c, ro, rd);
        if (lit.hit != 0 && (hit.hit == 0 || lit.dis < hit.depth)) {
            real mis = orc_power_heuristic(last_brdf_pdf, lit.pdf);
            v3 direct_li = vscale(lit.color, mis);
            result = vadd(result, vmul(throughput, direct_li));
        }

        if (hit.hit == 0) {
            result = vadd(result, vmul(throughput, world_at(c, rd)));
            break;
        }

        avoid = hit.index;
        v3 hitpos, normal; disney material;
        get_geometries(c, &hit, ro, rd, &hitpos, &normal, &material);
        cnt->n_shade++;

        real sign = -vdot(rd, normal);                  /* path.py:44 : always >= 0 (Q1) */
        if (sign < 0) normal = vneg(normal);

        lightsample li = lights_sample(c, hitpos, random3(rng));
        if (vany_gt0(li.color)) {
            bvhhit occ = bvh_intersect(c, hitpos, li.dir, avoid, cnt);
            if (occ.hit == 0 || occ.depth > li.dis) {
                v3 brdf_clr = disney_brdf(&material, normal, sign, vneg(rd), li.dir);
                real brdf_pdf = vavg(brdf_clr);
                real mis = orc_power_heuristic(li.pdf, brdf_pdf);
                v3 direct_li = vscale(vmul(vscale(li.color, mis), brdf_clr), dot_or_zero(normal, li.dir));
                result = vadd(result, vmul(throughput, direct_li));
            }
        }

        bsdfsample brdf = disney_bounce(&material, normal, sign, vneg(rd), random3(rng));
        throughput = vmul(throughput, brdf.color);
        ro = hitpos;
        rd = brdf.outdir;
        last_brdf_pdf = brdf.pdf;
    }
    return result;
}

static v3 do_render_pixel(const orc_ctx *c, int i, int j, orc_counters *cnt) {  /* path.py:82-92 */
    rng_t rng; rng.c = c; rng.i = orc_wanghash2(i, j); rng.draws = 0;
    real dx = rng_random(&rng), dy = rng_random(&rng);  /* random2: left to right */
    real x = ((real)i + dx) / (real)c->nx * (real)2 - (real)1;
    real y = ((real)j + dy) / (real)c->ny * (real)2 - (real)1;
    v3 ro, rd;
    camera_generate(c, x, y, &ro, &rd);
    v3 clr = path_trace(c, ro, rd, &rng, cnt);
    cnt->samples++;
    cnt->n_draws += rng.draws;
    return clr;
}

void orc_trace_pixel(orc_ctx *c, int i, int j, real rgb[3]) {
    orc_counters tmp; memset(&tmp, 0, sizeof tmp);
    v3 r = do_render_pixel(c, i, j, &tmp);
    rgb[0] = r.x; rgb[1] = r.y; rgb[2] = r.z;
}

static void counters_add(orc_counters *a, const orc_counters *b) {
    a->samples += b->samples; a->rays += b->rays; a->n_int += b->n_int; a->n_leaf += b->n_leaf;
    a->n_shade += b->n_shade; a->n_draws += b->n_draws; a->bounces += b->bounces;
    if (b->max_stack > a->max_stack) a->max_stack = b->max_stack;
}

void orc_render(orc_ctx *c) {                                                /* path.py:75-83 */
    orc_sobol_update(c);
    int nx = c->nx, ny = c->ny;
    orc_counters total; memset(&total, 0, sizeof total);
#ifdef _OPENMP
#pragma omp parallel num_threads(c->nthreads)
#endif
    {
        orc_counters local; memset(&local, 0, sizeof local);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1) nowait
#endif
        for (int i = c->wx0; i < c->wx1; i++) {
            if (!window_has(c, i)) continue;
            for (int j = 0; j < ny; j++) {
                v3 clr = do_render_pixel(c, i, j, &local);
                v4 *px = &c->film[0][(size_t)i * ny + j];   /* filmtable.py:37-39 */
                px->x += clr.x; px->y += clr.y; px->z += clr.z; px->w += (real)1.0;
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        counters_add(&total, &local);
    }
    (void)nx;
    counters_add(&c->cnt, &total);
}

void orc_render_preview(orc_ctx *c) {                                        /* preview.py:18-41 */
    orc_sobol_update(c);
    int ny = c->ny;
    orc_counters dummy; memset(&dummy, 0, sizeof dummy);
    for (int i = c->wx0; i < c->wx1; i++) {
        if (!window_has(c, i)) continue;
        for (int j = 0; j < ny; j++) {
            rng_t rng; rng.c = c; rng.i = orc_wanghash2(i, j); rng.draws = 0;
            v3 albedo = V3s(0), normal = V3s(0);
            real dx = rng_random(&rng), dy = rng_random(&rng);
            real x = ((real)i + dx) / (real)c->nx * (real)2 - (real)1;
            real y = ((real)j + dy) / (real)c->ny * (real)2 - (real)1;
            v3 ro, rd;
            camera_generate(c, x, y, &ro, &rd);
            bvhhit hit = bvh_intersect(c, ro, rd, -1, &dummy);
            if (hit.hit == 1) {
                v3 hitpos; disney material;
                get_geometries(c, &hit, ro, rd, &hitpos, &normal, &material);
                albedo = material.basecolor;
            }
            v4 *p1 = &c->film[1][(size_t)i * ny + j];
            p1->x += albedo.x; p1->y += albedo.y; p1->z += albedo.z; p1->w += (real)1.0;
            v4 *p2 = &c->film[2][(size_t)i * ny + j];
            p2->x += normal.x; p2->y += normal.y; p2->z += normal.z; p2->w += (real)1.0;
        }
    }
}

void orc_clear(orc_ctx *c) {                                                 /* filmtable.py:44-45 */
    for (int p = 0; p < 3; p++) memset(c->film[p], 0, (size_t)c->nx * c->ny * sizeof(v4));
}

void orc_get_image(orc_ctx *c, int pass, float *out) {                       /* filmtable.py:47-63 */
    for (int x = 0; x < c->nx; x++) {
        for (int y = 0; y < c->ny; y++) {
            v4 val = c->film[pass][(size_t)x * c->ny + y];
            if (val.w != 0) {
                val.x /= val.w; val.y /= val.w; val.z /= val.w;
                val.w = 1;
            } else {
                val.x = (real)0.9; val.y = (real)0.4; val.z = (real)0.9; val.w = 0;
            }
            float *o = out + ((size_t)x * c->ny + y) * 4;
            o[0] = (float)val.x; o[1] = (float)val.y; o[2] = (float)val.z; o[3] = (float)val.w;
        }
    }
}

void orc_fast_export_image(orc_ctx *c, int pass, float *out) {               /* filmtable.py:66-79 */
    for (int x = 0; x < c->nx; x++) {
        for (int y = 0; y < c->ny; y++) {
            size_t base = ((size_t)y * c->nx + x) * 3;
            v4 val = c->film[pass][(size_t)x * c->ny + y];
            if (val.w != 0) {
                val.x /= val.w; val.y /= val.w; val.z /= val.w;
            } else {
                val.x = (real)0.9; val.y = (real)0.4; val.z = (real)0.9;
            }
            out[base + 0] = (float)val.x; out[base + 1] = (float)val.y; out[base + 2] = (float)val.z;
        }
    }
}

void orc_get_film_raw(orc_ctx *c, int pass, float *out) {
    for (size_t t = 0; t < (size_t)c->nx * c->ny; t++) {
        out[t * 4 + 0] = (float)c->film[pass][t].x; out[t * 4 + 1] = (float)c->film[pass][t].y;
        out[t * 4 + 2] = (float)c->film[pass][t].z; out[t * 4 + 3] = (float)c->film[pass][t].w;
    }
}

/* the sums in the build's own precision (the f64 build's film is not rounded to f32 on the way out) */
void orc_get_film_real(orc_ctx *c, int pass, real *out) {
    for (size_t t = 0; t < (size_t)c->nx * c->ny; t++) {
        out[t * 4 + 0] = c->film[pass][t].x; out[t * 4 + 1] = c->film[pass][t].y;
        out[t * 4 + 2] = c->film[pass][t].z; out[t * 4 + 3] = c->film[pass][t].w;
    }
}

void orc_get_counters(orc_ctx *c, orc_counters *out) { *out = c->cnt; }
void orc_reset_counters(orc_ctx *c) { memset(&c->cnt, 0, sizeof c->cnt); }

/* ------------------------------------------------------------------ */
/* unit exports: the small functions one by one, for tests/test_reference_l1_cpu.py          */

void orc_unit_microfacet(int which, const real in[3], real out[3]) {          /* microfacet.py:9-78 */
    out[0] = out[1] = out[2] = 0;
    v3 r;
    switch (which) {
    case 0: out[0] = schlickFresnel(in[0]); break;
    case 1: out[0] = dielectricFresnel(in[0], in[1], in[2]); break;
    case 2: out[0] = GTR1(in[0], in[1]); break;
    case 3: out[0] = GTR2(in[0], in[1]); break;
    case 4: out[0] = smithGGX(in[0], in[1]); break;
    case 5: r = sample_GTR1(in[0], in[1], in[2]); out[0] = r.x; out[1] = r.y; out[2] = r.z; break;
    case 6: r = sample_GTR2(in[0], in[1], in[2]); out[0] = r.x; out[1] = r.y; out[2] = r.z; break;
    default: break;
    }
}

void orc_unit_common(int which, const real in[7], real out[4]) {              /* common.py:213-260 */
    out[0] = out[1] = out[2] = out[3] = 0;
    v3 a = V3(in[0], in[1], in[2]), b = V3(in[3], in[4], in[5]), r;
    switch (which) {
    case 0: r = tanspace_mul(a, b); out[0] = r.x; out[1] = r.y; out[2] = r.z; break;   /* tanspace(a) @ b */
    case 1: r = spherical(in[0], in[1]); out[0] = r.x; out[1] = r.y; out[2] = r.z; break;
    case 2: dir2tex(a, &out[0], &out[1]); break;
    case 3: r = reflectv(a, b); out[0] = r.x; out[1] = r.y; out[2] = r.z; break;
    case 4: out[0] = (real)refractv(a, b, in[6], &r); out[1] = r.x; out[2] = r.y; out[3] = r.z; break;
    default: break;
    }
}

void orc_unit_face_shading(const real vn[9], const real vt[6], real u, real v, real nrm[3], real tex[2]) {
    v3 n;
    face_shading(V3(vn[0], vn[1], vn[2]), V3(vn[3], vn[4], vn[5]), V3(vn[6], vn[7], vn[8]), vt, vt + 2, vt + 4, u, v,
                 &n, &tex[0], &tex[1]);
    nrm[0] = n.x; nrm[1] = n.y; nrm[2] = n.z;
}
