// film_ops.h -- the per-pixel film operations and the sample slab's entry format, shared by the small kernels (aux_kernels.hip) and
// the render kernel's tail finalisation (render_kernel.hip): ONE definition each, so that a film finalised inside the render launch
// is bit for bit the film the combine + resolve passes produce.  Plain IEEE adds and divides (no contraction can touch them).
#pragma once
#include "mpt_types.h"

// film[pix] += sample: one frame of FilmTable's running sum, reference filmtable.py:37-39 / path.py:93 ((r, g, b, 1) per sample)
__device__ __forceinline__ void film_add_sample(MptVec4 &a, float r, float g, float b) {
    a.x += r; a.y += g; a.z += b; a.w += 1.0f;
}

// FilmTable._get_image, filmtable.py:53-63 : rgb / w, w -> 1; empty -> (0.9, 0.4, 0.9, 0)
__device__ __forceinline__ MptVec4 film_resolve(MptVec4 v) {
    if (v.w != 0.0f) {
        v.x /= v.w; v.y /= v.w; v.z /= v.w;
        v.w = 1.0f;
    } else {
        v.x = 0.9f; v.y = 0.4f; v.z = 0.9f; v.w = 0.0f;
    }
    return v;
}

// ---- one sample of a launch's slab (path.py:93 before the sum): 16 bytes = TWO self-validating 8-byte granules
//        words 0-1: { bits(r),  bits(g) & 0xffff0000 | tag16 }
//        words 2-3: { bits(b),  bits(g) << 16        | tag16 }
// All 96 bits of (r, g, b) are kept (g travels in two halves).  tag16 is the launch's tag (mpt_flush: 2 + launch number mod 65534; 1
// for a launch that keeps the combine pass; 0 = freshly zeroed memory).  Each half is written by ONE naturally aligned 8-byte store
// (store_sample, render_kernel.hip), which is single-copy atomic: a reader that finds the launch's tag in a half has that half's data,
// whatever it sees of the other half -- so the tail finalisation's "the data is its own ready flag" needs no property of 16-byte
// stores (rounds 1-4 used {r, g, b, tag} behind ONE 16-byte store, whose halves are only OBSERVED to land together).
// What is left as an assumption (round-5 ADVICE): the READER takes an entry with one 16-byte load (raw_buffer_load_b128), and relies
// on each naturally aligned 8-byte half of that load being read as a unit -- the hardware's memory path moves at least 8-byte
// aligned units (its granule is 32 bytes), the language's memory model does not spell that out for a vector load.  Two 8-byte
// relaxed atomic loads per entry would remove the assumption for 0.9 % of the launch; the soaks (profiles/r05_soak_*.log: 120 360
// launches compared on data) have never seen a torn half.
typedef unsigned int mpt_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mpt_u4 slab_pack(float r, float g, float b, unsigned tag16) {
    const unsigned gb = (unsigned)__float_as_int(g);
    mpt_u4 v;
    v.x = (unsigned)__float_as_int(r); v.y = (gb & 0xffff0000u) | tag16;
    v.z = (unsigned)__float_as_int(b); v.w = (gb << 16) | tag16;
    return v;
}
__device__ __forceinline__ bool slab_ready(mpt_u4 v, unsigned tag16) {          // BOTH halves carry the launch's tag
    return ((v.y & 0xffffu) | (v.w << 16)) == (tag16 | (tag16 << 16));
}
__device__ __forceinline__ float slab_r(mpt_u4 v) { return __int_as_float((int)v.x); }
__device__ __forceinline__ float slab_g(mpt_u4 v) { return __int_as_float((int)((v.y & 0xffff0000u) | (v.w >> 16))); }
__device__ __forceinline__ float slab_b(mpt_u4 v) { return __int_as_float((int)v.z); }
