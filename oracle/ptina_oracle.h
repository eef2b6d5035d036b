/*
 * ptina_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of archibate/ptina's per-pixel path-trace hot path, used as the
 * parity checker for the HIP implementation in ptina_amd/csrc.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (ptina_amd) never links, imports or calls it.
 *
 * PARITY: formally UNPINNED against real PTina output -- the reference is Python + Taichi, Taichi is
 * not installed in the build container (ordinary ModuleNotFoundError) and the reference's own tests hold
 * no golden vectors for this path.  What IS pinned:
 *   - the restatement's LOGIC, to the reference's own source executed as plain Python on numpy scalars
 *     (tests/golden/taichi_standin + make_reference_l1_golden.py / make_reference_path_golden.py; the
 *     f64 build of this file reproduces the reference's functions and its end-to-end films to 1e-12);
 *   - the Sobol sampler, against scipy's unscrambled Sobol points (tests/golden/sobol_points.npz).
 * Not reproduced by anything here: Taichi's own code generation (fast-math, backend intrinsics).
 *
 * All citations are file:line into /root/reference.
 */
#ifndef PTINA_ORACLE_H
#define PTINA_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifdef ORC_F64
typedef double orc_real;
#else
typedef float orc_real;
#endif

typedef struct orc_ctx orc_ctx;

typedef struct {
    uint64_t samples;      /* camera samples traced                                  */
    uint64_t rays;         /* BVH traversals (closest-hit + shadow)                  */
    uint64_t n_int;        /* internal nodes popped (box tests), lbvh.py:338-340     */
    uint64_t n_leaf;       /* triangle tests executed, lbvh.py:328-330               */
    uint64_t n_shade;      /* shaded hits, path.py:42                                */
    uint64_t n_draws;      /* Sobol draws, sobol.py:121-125                          */
    uint64_t max_stack;    /* deepest traversal stack seen                           */
    uint64_t bounces;      /* loop iterations of path.py:25                          */
} orc_counters;

/* ---- scalar helpers (known-answer tests) ---- */
int32_t  orc_wanghash(int32_t x);                      /* sampling/__init__.py:9-16  */
int32_t  orc_wanghash2(int32_t x, int32_t y);          /* sampling/__init__.py:20-23 */
int32_t  orc_expand_bits(int32_t v);                   /* tree/lbvh.py:13-24         */
int32_t  orc_morton3d(const orc_real v[3]);            /* tree/lbvh.py:28-30         */
int32_t  orc_clz(int32_t x);                           /* tree/lbvh.py:34-42         */
int32_t  orc_count_low_bits(int32_t i);                /* sampling/sobol.py:11-17    */
orc_real orc_construct_float(int32_t i);               /* sampling/sobol.py:20-29    */

/* geometries.py:24-46; returns hit, writes near/far */
int orc_box_intersect(const orc_real lo[3], const orc_real hi[3], const orc_real o[3],
                      const orc_real d[3], orc_real *near_, orc_real *far_);
/* geometries.py:118-148; v = 3 positions; returns hit, writes depth,s,t */
int orc_face_intersect(const orc_real v[9], const orc_real o[3], const orc_real d[3],
                       orc_real *depth, orc_real *s, orc_real *t);
/* geometries.py:159-177 */
orc_real orc_sphere_intersect(const orc_real pos[3], orc_real rad2, const orc_real o[3],
                              const orc_real d[3]);
/* geometries.py:58-74 */
int orc_area_intersect(const orc_real pos[3], const orc_real dirx[3], const orc_real diry[3],
                       const orc_real o[3], const orc_real d[3], orc_real *depth, orc_real uv[2]);

/* materials/disney.py:14-50,53-106: params = the 12 Disney parameters in mtllib.py order,
 * basecolor expanded: [r,g,b, metallic, roughness, specular, specularTint, subsurface,
 * sheen, sheenTint, clearcoat, clearcoatGloss, transmission, ior] (14 values) */
void orc_disney_brdf(const orc_real params[14], const orc_real normal[3], orc_real sign,
                     const orc_real indir[3], const orc_real outdir[3], orc_real out_rgb[3]);
/* materials/disney.py:115-233: out = [outdir(3), pdf, color(3)] */
void orc_disney_bounce(const orc_real params[14], const orc_real normal[3], orc_real sign,
                       const orc_real indir[3], const orc_real samp[3], orc_real out[7]);
/* engine/path.py:11-15 */
orc_real orc_power_heuristic(orc_real a, orc_real b);
/* the small functions one by one (tests/test_reference_l1_cpu.py).
 * microfacet.py:9-78, which = 0 schlickFresnel(cost) | 1 dielectricFresnel(etai, etao, cosi) | 2 GTR1(cosh, alpha) |
 *   3 GTR2(cosh, alpha) | 4 smithGGX(cosi, alpha) | 5 sample_GTR1(u, v, alpha) -> xyz | 6 sample_GTR2(u, v, alpha) -> xyz */
void orc_unit_microfacet(int which, const orc_real in[3], orc_real out[3]);
/* common.py:213-260, which = 0 tanspace(in[0:3]) @ in[3:6] | 1 spherical(h, p) | 2 dir2tex(dir) -> (s, t) |
 *   3 reflect(I, N) | 4 refract(I, N, eta = in[6]) -> (has_r, T) */
void orc_unit_common(int which, const orc_real in[7], orc_real out[4]);
/* Face.normal / Face.texcoord, geometries.py:96-108 */
void orc_unit_face_shading(const orc_real vn[9], const orc_real vt[6], orc_real u, orc_real v, orc_real nrm[3],
                           orc_real tex[2]);

/* sampling/sobol.py:32-70 : V has (L+1) rows of D, L = ceil(log2(nsamples)).
 * s[j], a[j], m[j][18] for j>=1 are the Joe-Kuo triplets; dimension 0 is van der Corput. */
void orc_sobol_vgrid(const uint8_t *s, const uint32_t *a, const uint32_t *m, int D, int L,
                     int32_t *V);

/* ---- scene context ---- */
orc_ctx *orc_create(void);
void orc_destroy(orc_ctx *c);
void orc_set_threads(orc_ctx *c, int nthreads);        /* OpenMP threads for render */

void orc_set_size(orc_ctx *c, int nx, int ny);                         /* filmtable.py:41 */
void orc_set_window(orc_ctx *c, int x0, int x1);       /* render only x in [x0,x1) (slab tests) */
void orc_set_stripes(orc_ctx *c, int width, int index, int modulo);   /* ... only stripes index, index+modulo, ... */
int  orc_load_model(orc_ctx *c, const float *verts /*[3n][8]*/, const int32_t *mtlids, int n);
int  orc_load_materials(orc_ctx *c, const float *fac /*[m][12][4]*/, const int32_t *tex /*[m][12]*/, int m);
void orc_reset_images(orc_ctx *c);
int  orc_add_image(orc_ctx *c, const float *rgba /*[nx][ny][4]*/, int nx, int ny);
int  orc_build_tree(orc_ctx *c);                       /* 0 ok, -1 'hierarchy corrupted' */
int  orc_get_tree(orc_ctx *c, int32_t *child /*[n-1][2]*/, int32_t *leaf /*[n]*/,
                  float *bmin /*[n-1][3]*/, float *bmax /*[n-1][3]*/, int32_t *mc /*[n]*/);
void orc_set_camera_v2w(orc_ctx *c, const float v2w[16]);              /* camera.py:19-22 */
void orc_clear_lights(orc_ctx *c);
int  orc_add_light(orc_ctx *c, int type, const float color[3], const float pos[3],
                   const float axes[9], float size);                   /* light/__init__.py:34-49 */
void orc_set_world(orc_ctx *c, const float fac[4], int tex);           /* light/world.py:18-20 */

void orc_sobol_init(orc_ctx *c, const int32_t *V, int rows, int D);
void orc_sobol_reset(orc_ctx *c, int skip);                            /* sobol.py:92-97 */
void orc_sobol_update(orc_ctx *c);                                     /* sobol.py:99-105 */
int  orc_sobol_get(orc_ctx *c, int32_t *X, orc_real *P);               /* returns time */

void orc_render(orc_ctx *c);                           /* path.py:75-77 : update + one frame */
void orc_render_preview(orc_ctx *c);                   /* preview.py:18-41 */
void orc_clear(orc_ctx *c);                            /* filmtable.py:44-45 (all passes) */
void orc_get_image(orc_ctx *c, int pass, float *out /*[nx][ny][4]*/);  /* filmtable.py:47-63 */
void orc_fast_export_image(orc_ctx *c, int pass, float *out /*[ny*nx*3]*/); /* filmtable.py:66-79 */
void orc_get_film_raw(orc_ctx *c, int pass, float *out /*[nx*ny][4]*/);
void orc_get_film_real(orc_ctx *c, int pass, orc_real *out /*[nx*ny][4]*/);   /* in the build's own precision */
void orc_get_counters(orc_ctx *c, orc_counters *out);
void orc_reset_counters(orc_ctx *c);

/* debugging / unit parity: trace one camera sample of pixel (i,j) with the CURRENT Sobol
 * point; does not touch the film. */
void orc_trace_pixel(orc_ctx *c, int i, int j, orc_real rgb[3]);
/* camera.py:34-39 */
void orc_camera_generate(orc_ctx *c, orc_real x, orc_real y, orc_real o[3], orc_real d[3]);
/* lbvh.py:314-347: returns hit; out = [depth, index, u, v] */
int orc_intersect(orc_ctx *c, const orc_real o[3], const orc_real d[3], int avoid, orc_real out[4]);

#ifdef __cplusplus
}
#endif
#endif
