// film_ops.h -- the two per-pixel film operations, shared by the small kernels (aux_kernels.hip) and the render kernel's tail
// finalisation (render_kernel.hip): ONE definition each, so that a film finalised inside the render launch is bit for bit the
// film the combine + resolve passes produce.  Plain IEEE adds and divides (no contraction can touch them).
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
