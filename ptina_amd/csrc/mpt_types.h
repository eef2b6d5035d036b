// mpt_types.h -- structures shared by the host runtime (miptina.cpp) and the gfx950 kernels.
//
// HBM layout (all SoA-of-records, 16-byte aligned so every fetch is a dwordx4):
//   snode  float4[(n-1)*2]  reference-shaped node: {bmin.xyz, child0}, {bmax.xyz, child1}
//                           (tree/lbvh.py:48-53 keeps four separate arrays; one 32-B record here)
//   fnode  float4[(n-1)*4]  traversal node: both CHILD boxes + child ids in one 64-B record, axis by axis
//                           with the two children side by side so that a dwordx4 fetch lands in the
//                           register pairs v_pk_fma_f32 wants:
//                           {c0.lo.x, c1.lo.x, c0.hi.x, c1.hi.x} {..y..} {..z..} {id0, id1, -, -}
//                           id >= 0: internal node index; id < 0: leaf, slot = ~id
//   qnode  float4[nw*4]     quantised 4-wide node (the production gather kernel): {origin.xyz, scale.x} {scale.yz, lo_x, hi_x}
//                           {lo_y, hi_y, lo_z, hi_z} {ids}: a plane of child c is origin + byte c of the word * scale, boxes
//                           rounded outwards; four 16-B gathers per step instead of seven (the gathers are what the big
//                           scenes wait for, whatever their width)
//   wnode  float4[nw*8]     4-wide traversal node (gather kernel): the binary tree collapsed so that one 128-B
//                           record holds four child boxes, component by component with the four children side
//                           by side: {lo.x[4]} {hi.x[4]} {lo.y[4]} {hi.y[4]} {lo.z[4]} {hi.z[4]} {id[4]} {-};
//                           id >= 0: wide node index, id < 0: leaf, slot = ~id; an unused child is a point box at
//                           1e30 that no ray reaches
//   tgeo   float4[n*4]      per-leaf-slot triangle, ray-independent terms of geometries.py:118-148
//                           hoisted: {v0.xyz, D} {u.xyz, uu} {v.xyz, uv} {n.xyz, vv}
//   tfast  float4[n*3]      the production kernels' triangle record: plane normal n = u x v, the dual edge vectors
//                           a = (uv v - vv u) / D, c = (uv u - uu v) / D, and v0 in the .w lanes (derive_tfast_kernel)
//   tshade float4[n*4]      per-leaf-slot shading data: {n0.xyz, n1.x} {n1.yz, n2.xy} {n2.z, uv0.xy, uv1.x}
//                           {uv1.y, uv2.xy, mtlid}
//   mats   MptMaterial[m+1] the 12 Disney parameters of mtllib.py:44-56 + the terms derived from them, 160 B; the
//                           last record is the default material of mtllib.py:82-93 (mtlid -1)
//   P      float[B][dim]    Sobol points of the B frames of a batch (sobol.py:82-83 keeps one)
//   film   float4[passes][nx*ny], element x*ny + y (filmtable.py:14,37-39)
#pragma once

#include <stdint.h>

#define MPT_BLOCK 256          // 4 waves of 64; one 16x16 pixel tile, an 8x8 sub-tile per wave
#define MPT_TILE 16
#define MPT_MAX_BATCH 64       // frames per launch
// The eight per-XCD queue heads of a persistent launch sit MPT_QUEUE_STRIDE words apart -- each on a cache line of its own: device-
// scope atomics on words of ONE line are served one after the other (a single head saturates near 88 pulls per microsecond; the
// benchmark launch makes 52), so eight heads in one 32-byte block were still one queue to the memory system.  The tail
// finalisation's tile counter follows them on its own line.  MPT_QUEUE_WORDS = what a launch's block holds (zeroed per launch).
#ifndef MPT_QUEUE_STRIDE
#define MPT_QUEUE_STRIDE 32
#endif
#define MPT_QUEUE_WORDS (10 * MPT_QUEUE_STRIDE)
#define MPT_TIMELINE_WORDS 8    // diagnostics: 64-bit words per wave of the launch timeline (option "timeline")
#define MPT_MAX_LIGHTS 64
#define MPT_MAX_DEVICES 64     // device ids a process may hold contexts on (per-device launch caches)

struct MptVec4 { float x, y, z, w; };

// kernel launchers: C linkage between the objects of libmiptina.so, not part of its ABI
#define MPT_KERNEL_API extern "C" __attribute__((visibility("hidden")))

struct MptMaterial {
    float p[16];               // [0..2] basecolor, [3] metallic, [4] roughness, [5] specular, [6] specularTint,
                               // [7] subsurface, [8] sheen, [9] sheenTint, [10] clearcoat, [11] clearcoatGloss,
                               // [12] transmission, [13] ior
    int32_t tex[12];           // per parameter texture id, -1 = none
    int32_t any_tex;           // 1 if any tex != -1
    int32_t pad[3];
    float d[8];                // fast build, untextured materials: what Disney.__init__ derives from the parameters
                               // (disney.py:36-50) -- speccolor[3], sheencolor[3], alpha, clearcoatAlpha -- computed
                               // once per material by derive_materials_kernel with the same device code
};

struct MptLight {              // light/__init__.py:14-18
    MptVec4 color_size;        // rgb, size
    MptVec4 pos_type;          // xyz, type (as int bits)
    MptVec4 ax0, ax1, ax2;     // rows of the 3x3 axes matrix
};

// one contiguous float4 range of the film gather's pack / unpack (comm.cpp): count elements from src to dst
struct MptPiece { long long src, dst, count; };

// what the last SAH re-partition did (diagnostics / the traffic model of profiles/r06_build_table.json)
struct MptSahStats {
    int levels;                      // binned levels
    long long elems;                 // positions streamed, summed over the binned levels
    long long chunks, segments;      // summed over the binned levels
    long long part_words;            // words of chunk bins written, summed over the levels
    int tasks_small, tasks_big;      // ranges finished in LDS: <= 512 triangles, 513 ... 1024
    int t_sort_k, t_loop_k, t_max_k; // finish kernels, units of 1024 ticks of s_memtime: summed over the tasks in the sort / in the level loop, the longest task
    int task_levels, task_levels_max; // levels the tasks ran, summed / the most of one task
};

// device workspace of the SAH re-partition (sah_build.hip); capacities from mpt_sah_*_capacity(n)
struct MptSahBuffers {
    const float *verts; const int *leaf; int n;          // the LBVH build's inputs / leaf order, on the device
    MptVec4 *prim[2];                                    // [n][2]: {box lo, slot} {box hi, -} per position (ping-pong)
    int *seg[2]; size_t seg_cap;                         // [seg_cap][16] segment tables (this level / the next)
    int *dec;                                            // [seg_cap][8] this level's decisions
    int *ch_seg, *ch_left; size_t chunk_cap;             // [2][chunk_cap] chunk -> segment (this level / the next), [chunk_cap] left counts
    int *part; size_t part_words;                        // chunk bins of one level
    int *segbins; size_t segbin_words;                   // the bins of the segments that have several chunks
    int *tasks; size_t task_cap;                         // [task_cap][8] ranges the finish kernel takes
    int *meta;                                           // [16]
    MptSahStats *stats;                                  // host, optional
    volatile int *mail_host; int *mail_dev;              // host-pinned, device-mapped [32]: where the plan kernel leaves a level's outcome, or null
    MptVec4 *fnode;                                      // out: [n-1][4]
};

struct MptImage { int32_t nx, ny, base, pad; };   // image.py:14-16

// LDS-resident kernel: bytes from one node record to the next in LDS.  72, not 64: a ds_read_b64 is served in two
// groups of 32 lanes over 64 banks of 4 bytes, and with 64-byte records every lane's read of a given plane lands
// on one of FOUR bank pairs (16 i mod 64); with 72-byte records on one of 32 (18 i mod 64) -- reads stay 8-byte aligned
#ifndef MPT_LDS_NODE_STRIDE
#define MPT_LDS_NODE_STRIDE 72
#endif
#ifndef MPT_LDS4_NODE_STRIDE
#define MPT_LDS4_NODE_STRIDE 112     // bytes between the 4-wide node records in LDS (render_kernel_lds4: seven float4 of a wnode record)
#endif

// 16-bit tags of a launch's sample entries (film_ops.h): 0 = zeroed memory, 1 = a launch that keeps the combine pass, 2 ... 65535 =
// finalising launches in turn
enum { MPT_TAG_COMBINE = 1, MPT_TAG_FIRST = 2, MPT_TAG_PERIOD = 65534 };
enum { MPT_HIST_BASE = 32, MPT_HIST_WORDS = 3 * 65 + 3 * 6 * 2 + 24, MPT_COUNTER_WORDS = MPT_HIST_BASE + MPT_HIST_WORDS };   // + 24: node steps by log2 of the node's number (gather kernels)

struct MptRenderParams {
    int32_t nx, ny, x0, x1;                 // film size and the slab [x0,x1) this context renders
    int32_t nframes, chunk, nchunks, n;     // batch frames; frames per work item; items per tile; #triangles
    int32_t sobol_dim, nlights, world_tex, tiles_x;   // tiles_* : 16x16 tiles of the slab (strict build)
    int32_t tiles_y, ntiles;
    int32_t fnode_soa_n;                    // 0, or the node count when fnode holds the SoA transpose (layout A/B)
    float sobol_inv_dim;                    // 1 / sobol_dim (quotient estimate of the draw index reduction)
    int32_t nitems, tile_w_shift, tile_h_shift;   // fast build: (2^w x 2^h tile, chunk) work items of this launch
    int32_t nwide;                          // records of qnode (render_kernel_lds4 copies them into LDS)
    // columns rendered: x = x0 + s*stripe_pitch + w, w < stripe_w, x < x1 (one contiguous slab: stripe_w = 2^30)
    int32_t stripe_w, stripe_pitch;
    int32_t partial_stride;
    // pooled LDS kernel (render_pool.h): bytes between node records in LDS, material records kept in LDS besides the
    // default one, shader waves of the 16
    int32_t lds_node_stride, lds_nmats, pool_shaders;
    int32_t skip_dark;                      // 1: shadow rays whose candidate direct light is exactly zero are not traced (production build)
    int32_t default_mtl;                    // index of the default material's record in mats (fast build)           // float4 per frame of the sample slab = (tile-padded columns of the share) * ny
    float world_fac[4];
    float v2w[16];
    const MptVec4 *snode;
    const MptVec4 *fnode;
    const MptVec4 *wnode;                    // 4-wide traversal nodes (scenes that do not fit LDS), or null
    const MptVec4 *qnode;                    // the same nodes with the child boxes quantised to 8 bits (64-B records), or null
    const MptVec4 *onode;                    // 8-wide octant-ordered nodes (80-B records, oct_build.cpp), or null; tfast / tshade then hold the records in ITS leaf order
    const MptVec4 *tgeo;
    const MptVec4 *tshade;
    const MptVec4 *tfast;                    // production build: 48-byte triangle records {n, v0.x}{a, v0.y}{c, v0.z} derived from tgeo
    const MptMaterial *mats;
    const MptLight *lights;
    const MptImage *images;
    const MptVec4 *texels;
    const float *P;                          // [nframes][sobol_dim]
    MptVec4 *film0;                          // pass 0 (path) / unused by preview
    MptVec4 *film1;                          // pass 1 (albedo)
    MptVec4 *film2;                          // pass 2 (normal)
    MptVec4 *partial;                        // fast build: per-sample radiance [nframes][columns of the share][ny]
    unsigned long long *counters;            // mpt_counters when counting, else unused
    unsigned int *work_counter;              // persistent kernels: 8 per-range item counters
    int *stack_spill;                        // wide kernel: the overflow stack entries (128 - the LDS levels) of every lane of the grid
    unsigned int *watchdog;                  // host-pinned flag a persistent wave raises when it gives up
    // Tail finalisation (fast build): a wave that has run out of work sums, resolves and writes out the tiles
    // whose samples are all in the slab, while other waves still drain their last paths.  fin_counter: the next tile to
    // finalise (zeroed with the queue heads), null = the combine pass does it after the launch; slab_tag: the 16-bit tag both
    // 8-byte halves of every sample entry of THIS launch carry (their ready flags, film_ops.h; 1 when fin_counter is null);
    // image_out: where the resolved pixel goes as well (FilmTable.get_image's array, device-visible), or null
    unsigned int *fin_counter;
    MptVec4 *image_out;
    uint32_t slab_tag;                       // MPT_TAG_COMBINE, or MPT_TAG_FIRST + launch number mod MPT_TAG_PERIOD
    // Diagnostics (option "lane_hist", counting kernels only): counters + MPT_HIST_BASE holds, per issued NODE / LEAF / SHADE stage, how
    // many of the wave's 64 lanes took part ([3][65] stage counts) and whose lanes they were ([3][6 depths][2 ray kinds] lane-steps)
    int32_t lane_hist;
    // render_kernel_lds4 measures distances along a ray in units of 1 / t_scale (a power of two, so every comparison comes out as it
    // would unscaled): no box of the scene is entered farther than 1 / t_scale from any ray origin, which lets the clamp bit of an
    // FMA stand for max(t, 0) (pt_device.h Stack16W::T_SCALED).  t_unscale = 1 / t_scale
    float t_scale, t_unscale;
    int32_t pad3;
    unsigned long long *timeline;            // diagnostics: per wave {start, scene ready, queue empty, exit} in
                                             // 100 MHz ticks, or null
};
