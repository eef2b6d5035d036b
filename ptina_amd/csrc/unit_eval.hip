// unit_eval.hip -- test door: ONE device function of the hot path evaluated on rows of inputs.
//
// Built twice like render_kernel.hip (MPT_STRICT=1 -> mpt_launch_unit_eval_strict, =0 -> _fast), from the same
// pt_device.h the render kernels inline, so what the rows exercise is the code the megakernels run: the strict
// build's reference-order IEEE functions, the production build's v_rcp / v_rsq / v_sin / exp2-log2 / FMA forms,
// its 48-byte triangle records and its shared-lobe Disney.bounce.  tests/test_reference_units_gpu.py feeds it the
// inputs of tests/golden/reference_l1.npz -- vectors computed by the reference's own function bodies -- so the HIP
// code is held to the reference's source directly, not through the CPU oracle.
//
// Rows are 4-byte words: f32, except the two hash kinds (i32).  Kinds and columns: include/miptina.h.

#include "pt_device.h"
#include "tri_records.h"
#include "../../include/miptina.h"

#if MPT_STRICT
#define MPT_SUFFIX(x) x##_strict
#else
#define MPT_SUFFIX(x) x##_fast
#endif

DEV V3 ld(const float *r, int k) { return v3(r[k], r[k + 1], r[k + 2]); }
DEV void st(float *o, int k, V3 v) { o[k] = v.x; o[k + 1] = v.y; o[k + 2] = v.z; }

DEV Disney disney_from14(const float *p) {                                   // Disney.__init__, disney.py:13-50
    Disney m;
    m.basecolor = v3(p[0], p[1], p[2]);
    m.metallic = p[3]; m.roughness = p[4]; m.specular = p[5]; m.specularTint = p[6];
    m.subsurface = p[7]; m.sheen = p[8]; m.sheenTint = p[9]; m.clearcoat = p[10];
    m.clearcoatGloss = p[11]; m.transmission = p[12]; m.ior = p[13];
    disney_init(m);
    return m;
}

__global__ __launch_bounds__(64) void MPT_SUFFIX(unit_eval_kernel)(int kind, const float *__restrict__ in, int in_cols,
                                                                   float *__restrict__ out, int out_cols, int n) {
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= n) return;
    const float *r = in + (size_t)row * in_cols;
    float *o = out + (size_t)row * out_cols;
    switch (kind) {
    case MPT_UNIT_SCHLICK: o[0] = schlickFresnel(r[0]); break;                               // microfacet.py:9-10
    case MPT_UNIT_DIELECTRIC: o[0] = dielectricFresnel(r[0], r[1], r[2]); break;             // :14-27
    case MPT_UNIT_GTR1: o[0] = GTR1(r[0], r[1]); break;                                      // :31-34
    case MPT_UNIT_GTR2: o[0] = GTR2(r[0], r[1]); break;                                      // :38-41
    case MPT_UNIT_SMITHGGX: o[0] = smithGGX(r[0], r[1]); break;                              // :45-48
    case MPT_UNIT_SAMPLE_GTR1: st(o, 0, sample_GTR1(r[0], r[1], r[2])); break;               // :69-71
    case MPT_UNIT_SAMPLE_GTR2: st(o, 0, sample_GTR2(r[0], r[1], r[2])); break;               // :75-77
    case MPT_UNIT_TANSPACE: st(o, 0, tanspace_mul(tanspace(ld(r, 0)), ld(r, 3))); break;     // common.py:213-217
    case MPT_UNIT_SPHERICAL: st(o, 0, spherical(r[0], r[1])); break;                         // common.py:221-225
    case MPT_UNIT_DIR2TEX: dir2tex(ld(r, 0), &o[0], &o[1]); break;                           // common.py:234-239
    case MPT_UNIT_REFLECT: st(o, 0, reflectv(ld(r, 0), ld(r, 3))); break;                    // common.py:247-249
    case MPT_UNIT_REFRACT: {                                                                 // common.py:252-260
        V3 T;
        o[0] = refractv(ld(r, 0), ld(r, 3), r[6], &T) ? 1.0f : 0.0f;
        st(o, 1, T);
        break;
    }
    case MPT_UNIT_BOX: {                                                                     // geometries.py:24-46
#if MPT_STRICT
        float nearv, farv;
        o[0] = box_strict(ld(r, 0), ld(r, 3), ld(r, 6), ld(r, 9), &nearv, &farv) ? 1.0f : 0.0f;
        o[1] = nearv; o[2] = farv;
#else
        // the production slab test: 1/d and o/d per ray, entry distance out; the exit distance is not formed
        const V3 ro = ld(r, 6), rd = ld(r, 9);
        const V3 inv = v3(m_rcp(rd.x), m_rcp(rd.y), m_rcp(rd.z)), oinv = ro * inv;
        float tn;
        o[0] = box_fast(r[0], r[1], r[2], r[3], r[4], r[5], inv, oinv, MPT_INF, &tn) ? 1.0f : 0.0f;
        o[1] = tn; o[2] = -1.0f;
#endif
        break;
    }
    case MPT_UNIT_FACE: {                                   // geometries.py:96-148: intersect, normal, texcoord
        // in: v0 v1 v2 (9), ro (3), rd (3), vn0 vn1 vn2 (9), vt0 vt1 vt2 (6); out: hit, depth, s, t, normal (3), texcoord (2)
        MptVec4 g[4];
        tri_make_tgeo(r, r + 3, r + 6, g);
        float d = 2.0f * MPT_INF, s = 0.0f, t = 0.0f;
#if MPT_STRICT
        const bool hit = tri_test(g[0], g[1], g[2], g[3], ld(r, 9), ld(r, 12), &d, &s, &t);
#else
        MptVec4 f[3];
        tri_make_tfast(g, f);
        const bool hit = tri_test_fast(f[0], f[1], f[2], ld(r, 9), ld(r, 12), &d, &s, &t);
#endif
        o[0] = hit ? 1.0f : 0.0f; o[1] = d; o[2] = s; o[3] = t;
        const float *vn = r + 15, *vt = r + 24;
        const MptVec4 s0 = { vn[0], vn[1], vn[2], vn[3] }, s1 = { vn[4], vn[5], vn[6], vn[7] },
                      s2 = { vn[8], vt[0], vt[1], vt[2] }, s3 = { vt[3], vt[4], vt[5], 0.0f };
        V3 nrm; float tu, tv;
        face_shading(s0, s1, s2, s3, s, t, &nrm, &tu, &tv);
        st(o, 4, nrm); o[7] = tu; o[8] = tv;
        break;
    }
    case MPT_UNIT_SPHERE: o[0] = sphere_intersect(ld(r, 0), r[3], ld(r, 4), ld(r, 7)); break;   // geometries.py:159-177
    case MPT_UNIT_AREA: {                                                                    // geometries.py:58-74
        float d = MPT_INF, u = 0.0f, v = 0.0f;
        o[0] = area_intersect(ld(r, 0), ld(r, 3), ld(r, 6), ld(r, 9), ld(r, 12), &d, &u, &v) ? 1.0f : 0.0f;
        o[1] = d; o[2] = u; o[3] = v;
        break;
    }
    case MPT_UNIT_DISNEY_BRDF: {                            // disney.py:53-106; in: 14 parameters, normal, sign, indir, outdir
        const Disney m = disney_from14(r);
        st(o, 0, disney_brdf(m, ld(r, 14), r[17], ld(r, 18), ld(r, 21)));
        break;
    }
    case MPT_UNIT_DISNEY_BOUNCE: {                          // disney.py:115-233; in: 14 parameters, normal, sign, indir, samp
        const Disney m = disney_from14(r);
        const BsdfSample b = disney_bounce(m, ld(r, 14), r[17], ld(r, 18), ld(r, 21));
        st(o, 0, b.outdir); o[3] = b.pdf; st(o, 4, b.color);
        break;
    }
    case MPT_UNIT_POWER_HEURISTIC: o[0] = power_heuristic(r[0], r[1]); break;                // path.py:11-15
    case MPT_UNIT_WANGHASH: o[0] = __int_as_float(wanghash(__float_as_int(r[0]))); break;    // sampling/__init__.py:9-16
    case MPT_UNIT_WANGHASH2: o[0] = __int_as_float(wanghash2(__float_as_int(r[0]), __float_as_int(r[1]))); break;   // :20-23
    default: break;
    }
}

MPT_KERNEL_API hipError_t MPT_SUFFIX(mpt_launch_unit_eval)(int kind, const float *in, int in_cols, float *out, int out_cols,
                                                         int n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(MPT_SUFFIX(unit_eval_kernel), dim3((n + 63) / 64), dim3(64), 0, stream, kind, in, in_cols, out,
                       out_cols, n);
    return hipGetLastError();
}
