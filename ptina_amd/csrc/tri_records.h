// tri_records.h -- the triangle records of mpt_types.h derived from three vertex positions.
//
// One definition for every place that makes them (the device build's pack kernel, lbvh_build.hip; the host
// build, tree_build.cpp; derive_tfast_kernel, aux_kernels.hip; the unit-evaluation kernel, unit_eval.hip), in IEEE
// arithmetic without contraction whatever the flags of the including file, so the records are the same bits
// everywhere: the ray-independent terms of Face.intersect, geometries.py:118-143.
#pragma once

#include "mpt_types.h"

#if defined(__HIPCC__)
#define MPT_HD __host__ __device__ inline
#else
#define MPT_HD inline
#endif

// tgeo: {v0, D} {u, uu} {v, uv} {n, vv} with u = v1 - v0, v = v2 - v0, n = u x v, D = uv^2 - uu vv
MPT_HD void tri_make_tgeo(const float *p0, const float *p1, const float *p2, MptVec4 g[4]) {
#pragma clang fp contract(off)
    float u[3], v[3], nn[3];
    for (int a = 0; a < 3; a++) { u[a] = p1[a] - p0[a]; v[a] = p2[a] - p0[a]; }
    nn[0] = u[1] * v[2] - u[2] * v[1];
    nn[1] = u[2] * v[0] - u[0] * v[2];
    nn[2] = u[0] * v[1] - u[1] * v[0];
    float uu = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
    float uv = u[0] * v[0] + u[1] * v[1] + u[2] * v[2];
    float vv = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    float D = uv * uv - uu * vv;
    g[0] = { p0[0], p0[1], p0[2], D };
    g[1] = { u[0], u[1], u[2], uu };
    g[2] = { v[0], v[1], v[2], uv };
    g[3] = { nn[0], nn[1], nn[2], vv };
}

// tfast: {n, v0.x} {a, v0.y} {c, v0.z} with the dual edge vectors a = (uv v - vv u) / D, c = (uv u - uu v) / D of the
// reference's barycentric solve (geometries.py:134-143); a degenerate triangle (D = 0) gets infinities / NaNs, which
// fail every comparison of the test like the reference's own division by zero
MPT_HD void tri_make_tfast(const MptVec4 g[4], MptVec4 f[3]) {
#pragma clang fp contract(off)
    const float D = g[0].w, uu = g[1].w, uv = g[2].w, vv = g[3].w;
    const float u[3] = { g[1].x, g[1].y, g[1].z }, v[3] = { g[2].x, g[2].y, g[2].z };
    float a[3], c[3];
    for (int k = 0; k < 3; k++) {
        a[k] = (uv * v[k] - vv * u[k]) / D;
        c[k] = (uv * u[k] - uu * v[k]) / D;
    }
    f[0] = { g[3].x, g[3].y, g[3].z, g[0].x };
    f[1] = { a[0], a[1], a[2], g[0].y };
    f[2] = { c[0], c[1], c[2], g[0].z };
}

// tshade: three vertex normals, three uv pairs and the material id of face f (verts: [3n][8] = pos3 nrm3 uv2)
MPT_HD void tri_make_tshade(const float *p0, const float *p1, const float *p2, float mtlid_bits, MptVec4 s[4]) {
    s[0] = { p0[3], p0[4], p0[5], p1[3] };
    s[1] = { p1[4], p1[5], p2[3], p2[4] };
    s[2] = { p2[5], p0[6], p0[7], p1[6] };
    s[3] = { p1[7], p2[6], p2[7], mtlid_bits };
}
