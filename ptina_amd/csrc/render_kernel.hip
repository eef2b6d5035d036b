// render_kernel.hip -- the per-pixel path-trace megakernels (PathEngine._render + do_render +
// path_trace, engine/path.py:18-93) and the AOV preview kernel (engine/preview.py:23-41).
//
// Built twice from this one source: MPT_STRICT=1 -> symbols mpt_launch_*_strict,
// MPT_STRICT=0 -> mpt_launch_*_fast (see pt_device.h).
//
// Common to every kernel here (gfx950, wave64):
//   * a lane owns ONE pixel and walks a chunk of consecutive frames (spp) in order, regenerating
//     a new camera path the moment the previous one ends (no lane idles while its neighbours
//     finish a 5-bounce path); the per-pixel sum is kept in registers in frame order;
//   * a wave owns an 8x8 pixel tile so primary rays are coherent; within a row of 8 lanes
//     consecutive lanes are consecutive y = consecutive film addresses (index x*ny + y);
//   * chunk partial sums go to a scratch slab and a deterministic combine adds them in chunk
//     order (no float atomics: results are bit-reproducible and identical for any slab split
//     across GPUs).
//
// render_kernel (any scene size): scene records are gathered from HBM/L2; one 256-lane
//   workgroup = a 16x16 tile x one chunk; grid = tiles x chunks so the hardware dispatcher
//   balances thousands of items over 256 CUs; blockIdx is remapped so that the blocks an XCD
//   receives cover a contiguous run of tiles (its private 4 MiB L2 then holds that region's
//   subtrees); per-lane traversal stack in LDS, [level][lane].
//
// render_kernel_lds (fast build, scenes whose nodes + triangles fit the 160 KiB LDS): measured
//   on MI355X the gather version spends its time in the vector L1 -- a wave's node fetch touches
//   up to 64 different cache lines per load instruction, four instructions per node -- so for
//   small scenes the node and triangle records are copied ONCE per CU into LDS by a persistent
//   1024-lane workgroup (one per CU), and every traversal step becomes four ds_read_b128.  Waves
//   then pull (8x8 tile, chunk) work items from a global counter until it runs out.

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

#if MPT_STRICT
struct StrictTracer {
    const MptRenderParams *p;
    int *lds;
    template <bool COUNT>
    DEV Hit closest(V3 ro, V3 rd, int avoid, Cnt &cnt) const { return bvh_closest<COUNT>(*p, lds, ro, rd, avoid, cnt); }
    template <bool COUNT>
    DEV bool occluded(V3 ro, V3 rd, int avoid, float dis, Cnt &cnt) const {
        return bvh_occluded<COUNT>(*p, lds, ro, rd, avoid, dis, cnt);
    }
};
typedef StrictTracer BlockTracer;
DEV BlockTracer make_block_tracer(const MptRenderParams &p, int *lds) {
    BlockTracer t; t.p = &p; t.lds = lds; return t;
}
#else
typedef Tracer<GlobalScene, Stack> BlockTracer;
DEV BlockTracer make_block_tracer(const MptRenderParams &p, int *lds) {
    BlockTracer t;
    t.sc.fnode = p.fnode; t.sc.tgeo = p.tgeo;
    t.st.base = lds; t.st.sp = 0;
    t.n = p.n;
    return t;
}
#endif

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

// One pixel, frames [f, fend): do_render + path_trace (path.py:18-93) with path regeneration.
template <bool COUNT, class TR>
DEV void trace_pixel(const MptRenderParams &p, const TR &tr, int i, int j, int f, int fend, int chunk_id, Cnt &cnt) {
    const int pix = i * p.ny + j;
    const int h = wanghash2(i, j);                                           // path.py:72-73

    MptVec4 acc;
    if (p.nchunks == 1) acc = p.film0[pix];                                  // film += in frame order, filmtable.py:37-39
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
            Hit hit = tr.template closest<COUNT>(ro, rd, avoid, cnt);

            LightHit lit = lights_hit(p, ro, rd);
            if (lit.hit && (hit.hit == 0 || lit.dis < hit.depth)) {
                float mis = power_heuristic(last_brdf_pdf, lit.pdf);
                result = result + throughput * (lit.color * mis);
            }

            if (hit.hit == 0) {
                result = result + throughput * world_at(p, rd);
                done = true;                                                 // break, path.py:39
            } else {
                avoid = hit.index;
                V3 hitpos, normal; Disney material;
                get_geometries(p, hit, ro, rd, &hitpos, &normal, material);
                if (COUNT) { cnt.n_shade++; cnt.n_draws += 6; }

                float sign = -dot(rd, normal);                               // path.py:44-46 (never negative)
                if (sign < 0.0f) normal = -normal;

                LightSample li = lights_sample(p, hitpos, random3(rng));
                if (any_gt0(li.color)) {
                    if (!tr.template occluded<COUNT>(hitpos, li.dir, avoid, li.dis, cnt)) {
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
    else p.partial[(size_t)chunk_id * ((size_t)p.nx * p.ny) + pix] = acc;
}

// ---------------------------------------------------------------- gather kernel: 16x16 tile x chunk per workgroup
DEV bool tile_pixel(const MptRenderParams &p, int tile, int *pi, int *pj) {
    int tx = tile / p.tiles_y, ty = tile - tx * p.tiles_y;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int i = p.x0 + tx * MPT_TILE + (wave >> 1) * 8 + (lane >> 3);
    int j = ty * MPT_TILE + (wave & 1) * 8 + (lane & 7);
    *pi = i; *pj = j;
    return i < p.x1 && j < p.ny;
}

template <int STACK, bool COUNT>
__global__ __launch_bounds__(MPT_BLOCK) void MPT_SUFFIX(render_kernel)(const MptRenderParams p) {
    __shared__ int s_stack[STACK * MPT_BLOCK];
    BlockTracer tr = make_block_tracer(p, s_stack + threadIdx.x);

    int item = xcd_remap(blockIdx.x, gridDim.x);
    int tile = item / p.nchunks, chunk = item - tile * p.nchunks;
    int i, j;
    Cnt cnt = {};
    if (tile_pixel(p, tile, &i, &j)) {
        int f = chunk * p.chunk;
        trace_pixel<COUNT>(p, tr, i, j, f, min(f + p.chunk, p.nframes), chunk, cnt);
    }
    flush_counters<COUNT>(p, cnt);
}

#if !MPT_STRICT
// ---------------------------------------------------------------- LDS-resident persistent kernel
// dynamic LDS: [ (n-1)*4 node float4 | n*4 triangle float4 | lds_stack x 1024 int16 ]
template <bool COUNT>
__global__ __launch_bounds__(MPT_LDS_BLOCK) void render_kernel_lds(const MptRenderParams p) {
    extern __shared__ __attribute__((aligned(16))) MptVec4 smem[];
    const int nnode4 = (p.n - 1) * 4, ntri4 = p.n * 4;
    {   // one copy of the scene per CU: coalesced 16-B loads, ds_write_b128
        for (int k = threadIdx.x; k < nnode4; k += MPT_LDS_BLOCK) smem[k] = p.fnode[k];
        for (int k = threadIdx.x; k < ntri4; k += MPT_LDS_BLOCK) smem[nnode4 + k] = p.tgeo[k];
    }
    __syncthreads();

    Tracer<LdsScene, Stack16> tr;
    tr.sc.fnode = (LdsVec4Ptr)(void *)smem;
    tr.sc.tgeo = (LdsVec4Ptr)(void *)(smem + nnode4);
    tr.st.base = (LdsShortPtr)(void *)(smem + nnode4 + ntri4) + threadIdx.x;
    tr.st.sp = 0;
    tr.n = p.n;

    const int lane = threadIdx.x & 63;
    const int t8y = (p.ny + 7) >> 3;
    const int t8x = (p.x1 - p.x0 + 7) >> 3;
    const int nitems = t8x * t8y * p.nchunks;
    Cnt cnt = {};
    // every wave pulls work until the counter runs past the last item: all waves leave the loop
    for (;;) {
        int item = 0;
        if (lane == 0) item = (int)atomicAdd(p.work_counter, 1u);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= nitems) break;
        int tile = item / p.nchunks, chunk = item - tile * p.nchunks;
        int tx = tile / t8y, ty = tile - tx * t8y;
        int i = p.x0 + tx * 8 + (lane >> 3);
        int j = ty * 8 + (lane & 7);
        if (i < p.x1 && j < p.ny) {
            int f = chunk * p.chunk;
            trace_pixel<COUNT>(p, tr, i, j, f, min(f + p.chunk, p.nframes), chunk, cnt);
        }
    }
    flush_counters<COUNT>(p, cnt);
}
#endif

// PreviewEngine._render, engine/preview.py:23-41
template <int STACK>
__global__ __launch_bounds__(MPT_BLOCK) void MPT_SUFFIX(preview_kernel)(const MptRenderParams p) {
    __shared__ int s_stack[STACK * MPT_BLOCK];
    BlockTracer tr = make_block_tracer(p, s_stack + threadIdx.x);
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
        Hit hit = tr.template closest<false>(ro, rd, -1, cnt);
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

#if !MPT_STRICT
// lds_bytes = scene records + 2 KiB per stack level; grid = one persistent workgroup per CU
extern "C" hipError_t mpt_launch_render_lds(const MptRenderParams *p, int grid, size_t lds_bytes, int count,
                                            hipStream_t stream) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void *)render_kernel_lds<false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void *)render_kernel_lds<true>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        configured = true;
    }
    if (count) hipLaunchKernelGGL(render_kernel_lds<true>, dim3(grid), dim3(MPT_LDS_BLOCK), lds_bytes, stream, *p);
    else hipLaunchKernelGGL(render_kernel_lds<false>, dim3(grid), dim3(MPT_LDS_BLOCK), lds_bytes, stream, *p);
    return hipGetLastError();
}
#endif

extern "C" hipError_t MPT_SUFFIX(mpt_launch_preview)(const MptRenderParams *p, int grid, int stack,
                                                      hipStream_t stream) {
    if (stack <= 32) hipLaunchKernelGGL((MPT_SUFFIX(preview_kernel)<32>), dim3(grid), dim3(MPT_BLOCK), 0, stream, *p);
    else hipLaunchKernelGGL((MPT_SUFFIX(preview_kernel)<64>), dim3(grid), dim3(MPT_BLOCK), 0, stream, *p);
    return hipGetLastError();
}
