// render_kernel.hip -- the per-pixel path-trace megakernel (PathEngine._render + do_render +
// path_trace, engine/path.py:18-93) and the AOV preview kernel (engine/preview.py:23-41).
//
// Built twice from this one source: MPT_STRICT=1 -> symbols mpt_launch_*_strict,
// MPT_STRICT=0 -> mpt_launch_*_fast (see pt_device.h).
//
// Work decomposition (gfx950, wave64):
//   * one workgroup = 256 lanes = a 16x16 pixel tile x a chunk of consecutive frames (spp);
//     each wave owns an 8x8 sub-tile so primary rays are coherent; within a row of 8 lanes
//     consecutive lanes are consecutive y = consecutive film addresses (index x*ny + y).
//   * a lane owns ONE pixel and walks its chunk's frames in order, regenerating a new camera
//     path the moment the previous one ends (no lane idles while its neighbours finish a
//     5-bounce path); the per-pixel sum is kept in registers in frame order.
//   * grid = tiles x chunks, so the hardware dispatcher load-balances thousands of work items
//     over the 256 CUs; chunk partial sums go to a scratch slab and a deterministic combine
//     adds them in chunk order (no float atomics: results are bit-reproducible and identical
//     for any slab split across GPUs).
//   * blockIdx is remapped so that the blocks an XCD receives (b, b+8, ...) cover a contiguous
//     run of tiles: neighbouring tiles share BVH subtrees in that XCD's private 4 MiB L2.
//   * traversal stack: per-lane LIFO in LDS, [level][lane] so push/pop are conflict-free.

#include "pt_device.h"

#if MPT_STRICT
#define MPT_SUFFIX(x) x##_strict
#else
#define MPT_SUFFIX(x) x##_fast
#endif

DEV int xcd_remap(int b, int nb) {
    // blocks are dealt round-robin over the 8 XCDs: give XCD k the k-th contiguous run of work
    int q = nb >> 3, r = nb & 7;
    int xcd = b & 7, k = b >> 3;
    return xcd * q + (xcd < r ? xcd : r) + k;
}

DEV bool tile_pixel(const MptRenderParams &p, int tile, int *pi, int *pj) {
    int tx = tile / p.tiles_y, ty = tile - tx * p.tiles_y;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int i = p.x0 + tx * MPT_TILE + (wave >> 1) * 8 + (lane >> 3);
    int j = ty * MPT_TILE + (wave & 1) * 8 + (lane & 7);
    *pi = i; *pj = j;
    return i < p.x1 && j < p.ny;
}

template <bool COUNT>
DEV void flush_counters(const MptRenderParams &p, const Cnt &c) {
    if (!COUNT) return;
    unsigned v[8] = { c.samples, c.rays, c.n_box, c.n_tri, c.n_shade, c.n_draws, c.bounces, c.n_node };
#pragma unroll
    for (int k = 0; k < 8; k++) {
        unsigned x = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(p.counters + k, (unsigned long long)x);
    }
}

template <int STACK, bool COUNT>
__global__ __launch_bounds__(MPT_BLOCK) void MPT_SUFFIX(render_kernel)(const MptRenderParams p) {
    __shared__ int s_stack[STACK * MPT_BLOCK];
    int *lds = s_stack + threadIdx.x;

    int item = xcd_remap(blockIdx.x, gridDim.x);
    int tile = item / p.nchunks, chunk = item - tile * p.nchunks;
    int i, j;
    bool valid = tile_pixel(p, tile, &i, &j);
    Cnt cnt = {};
    if (valid) {
        const int pix = i * p.ny + j;
        const int h = wanghash2(i, j);                                       // path.py:72-73
        int f = chunk * p.chunk;
        const int fend = min(f + p.chunk, p.nframes);

        MptVec4 acc;
        if (p.nchunks == 1) acc = p.film0[pix];                              // film += in frame order, filmtable.py:37-39
        else { acc.x = acc.y = acc.z = acc.w = 0.0f; }

        bool alive = false;
        Rng rng; rng.dim = p.sobol_dim; rng.P = p.P; rng.i = h;
        V3 ro = v3s(0.0f), rd = v3s(0.0f), result = v3s(0.0f), throughput = v3s(0.0f);
        float last_brdf_pdf = 0.0f;
        int avoid = -1, depth = 0;

        while (true) {
            if (!alive) {
                if (f >= fend) break;
                // do_render, path.py:86-92
                rng.P = p.P + (size_t)f * p.sobol_dim;
                rng.i = h;
                float dx = rng_random(rng), dy = rng_random(rng);
                float x = m_div((float)i + dx, (float)p.nx) * 2.0f - 1.0f;
                float y = m_div((float)j + dy, (float)p.ny) * 2.0f - 1.0f;
                camera_generate(p, x, y, &ro, &rd);
                avoid = -1; depth = 0;
                result = v3s(0.0f); throughput = v3s(1.0f); last_brdf_pdf = 0.0f;
                alive = true;
                if (COUNT) { cnt.samples++; cnt.n_draws += 2; }
            }

            bool done = true;
            // path_trace loop head, path.py:25
            if (depth < 5 && any_gt0(throughput) && any_ne0(rd)) {
                done = false;
                depth += 1;
                if (COUNT) cnt.bounces++;

                rd = normalized(rd);
                Hit hit = bvh_closest<COUNT>(p, lds, ro, rd, avoid, cnt);

                LightHit lit = lights_hit(p, ro, rd);
                if (lit.hit && (hit.hit == 0 || lit.dis < hit.depth)) {
                    float mis = power_heuristic(last_brdf_pdf, lit.pdf);
                    result = result + throughput * (lit.color * mis);
                }

                if (hit.hit == 0) {
                    result = result + throughput * world_at(p, rd);
                    done = true;                                             // break, path.py:39
                } else {
                    avoid = hit.index;
                    V3 hitpos, normal; Disney material;
                    get_geometries(p, hit, ro, rd, &hitpos, &normal, material);
                    if (COUNT) { cnt.n_shade++; cnt.n_draws += 6; }

                    float sign = -dot(rd, normal);                           // path.py:44-46 (never negative)
                    if (sign < 0.0f) normal = -normal;

                    LightSample li = lights_sample(p, hitpos, random3(rng));
                    if (any_gt0(li.color)) {
                        if (!bvh_occluded<COUNT>(p, lds, hitpos, li.dir, avoid, li.dis, cnt)) {
                            V3 brdf_clr = disney_brdf(material, normal, sign, -rd, li.dir);
                            float brdf_pdf = vavg(brdf_clr);
                            float mis = power_heuristic(li.pdf, brdf_pdf);
                            V3 direct_li = li.color * mis * brdf_clr * dot_or_zero(normal, li.dir);
                            result = result + throughput * direct_li;
                        }
                    }

                    BsdfSample brdf = disney_bounce(material, normal, sign, -rd, random3(rng));
                    throughput = throughput * brdf.color;
                    ro = hitpos;
                    rd = brdf.outdir;
                    last_brdf_pdf = brdf.pdf;
                }
            }

            if (done) {
                acc.x += result.x; acc.y += result.y; acc.z += result.z; acc.w += 1.0f;   // path.py:93
                f++;
                alive = false;
            }
        }

        if (p.nchunks == 1) p.film0[pix] = acc;
        else p.partial[(size_t)chunk * ((size_t)p.nx * p.ny) + pix] = acc;
    }
    flush_counters<COUNT>(p, cnt);
}

// PreviewEngine._render, engine/preview.py:23-41
template <int STACK>
__global__ __launch_bounds__(MPT_BLOCK) void MPT_SUFFIX(preview_kernel)(const MptRenderParams p) {
    __shared__ int s_stack[STACK * MPT_BLOCK];
    int *lds = s_stack + threadIdx.x;
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    int i, j;
    if (!tile_pixel(p, tile, &i, &j)) return;
    const int pix = i * p.ny + j;
    const int h = wanghash2(i, j);
    Cnt cnt = {};
    MptVec4 a1 = p.film1[pix], a2 = p.film2[pix];
    for (int f = 0; f < p.nframes; f++) {
        Rng rng; rng.dim = p.sobol_dim; rng.P = p.P + (size_t)f * p.sobol_dim; rng.i = h;
        V3 albedo = v3s(0.0f), normal = v3s(0.0f);
        float dx = rng_random(rng), dy = rng_random(rng);
        float x = m_div((float)i + dx, (float)p.nx) * 2.0f - 1.0f;
        float y = m_div((float)j + dy, (float)p.ny) * 2.0f - 1.0f;
        V3 ro, rd;
        camera_generate(p, x, y, &ro, &rd);
        Hit hit = bvh_closest<false>(p, lds, ro, rd, -1, cnt);
        if (hit.hit == 1) {
            V3 hitpos; Disney material;
            get_geometries(p, hit, ro, rd, &hitpos, &normal, material);
            albedo = material.basecolor;
        }
        a1.x += albedo.x; a1.y += albedo.y; a1.z += albedo.z; a1.w += 1.0f;
        a2.x += normal.x; a2.y += normal.y; a2.z += normal.z; a2.w += 1.0f;
    }
    p.film1[pix] = a1;
    p.film2[pix] = a2;
}

// ---------------------------------------------------------------- host-side launchers
extern "C" hipError_t MPT_SUFFIX(mpt_launch_render)(const MptRenderParams *p, int grid, int stack, int count,
                                                     hipStream_t stream) {
    if (stack <= 32) {
        if (count) hipLaunchKernelGGL((MPT_SUFFIX(render_kernel)<32, true>), dim3(grid), dim3(MPT_BLOCK), 0, stream, *p);
        else hipLaunchKernelGGL((MPT_SUFFIX(render_kernel)<32, false>), dim3(grid), dim3(MPT_BLOCK), 0, stream, *p);
    } else {
        if (count) hipLaunchKernelGGL((MPT_SUFFIX(render_kernel)<64, true>), dim3(grid), dim3(MPT_BLOCK), 0, stream, *p);
        else hipLaunchKernelGGL((MPT_SUFFIX(render_kernel)<64, false>), dim3(grid), dim3(MPT_BLOCK), 0, stream, *p);
    }
    return hipGetLastError();
}

extern "C" hipError_t MPT_SUFFIX(mpt_launch_preview)(const MptRenderParams *p, int grid, int stack,
                                                      hipStream_t stream) {
    if (stack <= 32) hipLaunchKernelGGL((MPT_SUFFIX(preview_kernel)<32>), dim3(grid), dim3(MPT_BLOCK), 0, stream, *p);
    else hipLaunchKernelGGL((MPT_SUFFIX(preview_kernel)<64>), dim3(grid), dim3(MPT_BLOCK), 0, stream, *p);
    return hipGetLastError();
}
