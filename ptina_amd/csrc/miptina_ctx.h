// miptina_ctx.h -- what the translation units of the host runtime share: the context, the error
// plumbing and the launchers of the kernels.  Internal; the public surface is include/miptina.h.
#pragma once

#include "../../include/miptina.h"
#include "mpt_types.h"
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

// kernel launchers (render_kernel.hip x2, aux_kernels.hip)
MPT_KERNEL_API hipError_t mpt_launch_render_fast(const MptRenderParams *, int grid, int stack, int count, hipStream_t);
MPT_KERNEL_API hipError_t mpt_launch_render_strict(const MptRenderParams *, int grid, int stack, int count, hipStream_t);
MPT_KERNEL_API hipError_t mpt_launch_derive_materials(MptMaterial *mats, int count, hipStream_t);
MPT_KERNEL_API hipError_t mpt_launch_derive_tfast(const MptVec4 *tgeo, MptVec4 *tfast, int n, hipStream_t);
MPT_KERNEL_API hipError_t mpt_wide_blocks(int grid, int count, int quant, int *blocks);
MPT_KERNEL_API hipError_t mpt_launch_render_wide(const MptRenderParams *, int blocks, int count, int quant, hipStream_t);
MPT_KERNEL_API hipError_t mpt_launch_render_lds(const MptRenderParams *, int grid, int block, size_t lds_bytes, int count, hipStream_t);
MPT_KERNEL_API hipError_t mpt_launch_render_lds4(const MptRenderParams *, int grid, int block, size_t lds_bytes, int count, hipStream_t);
MPT_KERNEL_API hipError_t mpt_launch_render_pool(const MptRenderParams *, int grid, int block, size_t lds_bytes, int count, hipStream_t);
MPT_KERNEL_API size_t mpt_pool_lds_overhead(void);
MPT_KERNEL_API hipError_t mpt_launch_preview_fast(const MptRenderParams *, int grid, int stack, hipStream_t);
MPT_KERNEL_API hipError_t mpt_launch_preview_strict(const MptRenderParams *, int grid, int stack, hipStream_t);
MPT_KERNEL_API hipError_t mpt_launch_sobol_update(const int *X, int *Xout, const int *V, float *P, int dim, int rows, int time0,
                                              int count, int keep, int write_x, hipStream_t stream);
MPT_KERNEL_API hipError_t mpt_launch_combine(MptVec4 *film, const MptVec4 *partial, int ny, int x0, int x1,
                                         int stripe_w, int stripe_pitch, int ccols, int nframes, hipStream_t);
MPT_KERNEL_API hipError_t mpt_launch_resolve(const MptVec4 *film, MptVec4 *out, size_t npix, hipStream_t);
MPT_KERNEL_API hipError_t mpt_launch_transpose_nodes(const MptVec4 *in, MptVec4 *out, int ni, hipStream_t);
MPT_KERNEL_API hipError_t mpt_launch_probe(double *out, int threads, size_t lds_bytes, hipStream_t);
MPT_KERNEL_API hipError_t mpt_launch_unit_eval_fast(int kind, const float *in, int in_cols, float *out, int out_cols, int n, hipStream_t);
MPT_KERNEL_API hipError_t mpt_launch_unit_eval_strict(int kind, const float *in, int in_cols, float *out, int out_cols, int n, hipStream_t);
MPT_KERNEL_API hipError_t mpt_launch_copy_pieces(const MptVec4 *src, MptVec4 *dst, const MptPiece *tab, int npieces,
                                             long long max_count, hipStream_t);
MPT_KERNEL_API hipError_t mpt_launch_export(const MptVec4 *film, float *out, int nx, int ny, hipStream_t);

// on-GPU LBVH build (lbvh_build.hip)
struct MptLbvhBuffers {
    const float *verts; const int *mtlids; int n;
    float *cen; int *bounds;
    unsigned long long *keys_in, *keys_out;
    void *sort_tmp; size_t sort_tmp_bytes;
    int *child, *parent, *leaf, *mc;
    float *bmin, *bmax;
    unsigned *arrive;
    int *depth;
    MptVec4 *snode, *fnode, *tgeo, *tshade;
};
MPT_KERNEL_API size_t mpt_sah_seg_capacity(int n);
MPT_KERNEL_API size_t mpt_sah_chunk_capacity(int n);
MPT_KERNEL_API size_t mpt_sah_task_capacity(int n);
MPT_KERNEL_API size_t mpt_sah_part_words(int n);
MPT_KERNEL_API size_t mpt_sah_segbin_words(int n);
MPT_KERNEL_API size_t mpt_sah_level_words(int n, size_t nseg, int *nb_out);
MPT_KERNEL_API int mpt_sah_task_max(void);
MPT_KERNEL_API hipError_t mpt_sah_build(const MptSahBuffers *B, int *depth, hipStream_t stream);
MPT_KERNEL_API hipError_t mpt_wide_scan_bytes(int ni, size_t *bytes);
MPT_KERNEL_API hipError_t mpt_wide_build(const MptVec4 *fnode, int n, MptVec4 *wnode, MptVec4 *qnode, int *bin_of, int *ncount,
                                     int *offset, void *scan_tmp, size_t scan_bytes, double *d_area, int *nwide, int *depth,
                                     double area[2], hipStream_t stream, volatile int *mail_host, int *mail_dev);
MPT_KERNEL_API hipError_t mpt_launch_permute_tris(const MptVec4 *tfast, const MptVec4 *tshade, const int32_t *perm, MptVec4 *tfast8,
                                              MptVec4 *tshade8, int n, hipStream_t stream);
MPT_KERNEL_API hipError_t mpt_oct_blocks(int grid, int count, int *blocks);
MPT_KERNEL_API hipError_t mpt_launch_render_oct(const MptRenderParams *p, int blocks, int count, hipStream_t stream);
MPT_KERNEL_API hipError_t mpt_lbvh_sort_bytes(int n, size_t *bytes);
MPT_KERNEL_API hipError_t mpt_lbvh_build(const MptLbvhBuffers *b, hipStream_t stream);

// ------------------------------------------------------------------ errors (miptina.cpp)
#ifndef MPT_WITH_POOL
#define MPT_WITH_POOL 0          // 1: the pooled LDS kernel (render_pool.h) is compiled in (A/B build `make pool`)
#endif
#define MPT_INTERNAL __attribute__((visibility("hidden")))   // shared between the .cpp files, not exported
MPT_INTERNAL int fail(const char *fmt, ...);
struct mpt_ctx;
MPT_INTERNAL int make_oct8(mpt_ctx *c);          // oct_build.cpp

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) return fail("%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------ context
enum { MPT_MAX_PIPE = 6 };

struct mpt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    mpt_caps caps{};

    // options
    int mode = MPT_MODE_FAST, batch = 32, chunk = 0, count = 0, use_lds = 1, lds_block = 0, zero_copy = 1;
    int lds_wide = 1;                                 // 1: scenes that fit LDS beside them walk the 4-wide nodes there (render_kernel_lds4), 0: the binary ones (render_kernel_lds)
    int skip_dark = -1;                  // -1 auto (production build on, strict build off), 0 / 1 as set: do not trace shadow rays whose candidate direct light is exactly zero (production build: default;
                                         // the strict build traces them like the reference unless the option is set explicitly to 1 there)
    int use_pool = 0, pool_shaders = 3;  // LDS kernel with specialised waves and path pools (render_pool.h)
    int max_mtlid = -1;                  // largest material id of the model (-1: only the default material)
    int num_cus = 256;
    int clock_khz = 0;                   // hipDeviceProp_t.clockRate: peak shader clock (roofline peaks in bench.py)
    int tile_w_shift = 3, tile_h_shift = 3;   // work-item tile 2^w x 2^h pixels
    int last_div = 1;                    // share of the chip the last launch took: 1/last_div of the CUs
    int last_kernel = 0;                 // 0 gather kernel, 1 LDS-resident kernel (what the last flush launched)

    // film
    int nx = 0, ny = 0, x0 = 0, x1 = 0;
    int stripe_w = 0, stripe_idx = 0, stripe_mod = 1;   // stripe_w > 0: columns dealt out in stripes (mpt_set_stripes)
    MptVec4 *film[3] = { nullptr, nullptr, nullptr };
    size_t film_cap = 0;                 // pixels allocated per pass
    MptVec4 *resolved = nullptr;         // nx*ny float4 (get_image staging on device)
    float *exported = nullptr;           // nx*ny*3

    // model (host copy kept for the tree build)
    int nfaces = 0;
    std::vector<float> verts;            // [3n][8]
    double scene_cen[3] = { 0, 0, 0 }, scene_rad = 0;   // bounding sphere of the model's box (mpt_load_model)
    std::vector<int32_t> mtlids;
    bool tree_valid = false;
    int tree_depth = 0;                  // reference LBVH (strict build)
    int fast_depth = 0;                  // tree the fast build walks (SAH or LBVH)
    int tree_kind = 1;                   // fast build: 1 = SAH re-partition of the LBVH's leaves, 0 = the LBVH itself
    int gpu_build = 1;                   // 1 = LBVH built on the device (lbvh_build.hip), 0 = host build
    int sah_max = 1 << 22;               // above this many faces the fast build walks the LBVH itself (the threaded host
                                         // SAH pass takes ~0.2 s at 1 M faces; it was 1.5 s on one core, hence 2^18 in round 1)
    bool host_tree_valid = false;        // h_child/h_leaf/... mirror the device tree (lazily downloaded)
    // device-side build workspace
    float *d_verts = nullptr; int *d_mtlids = nullptr; size_t d_model_cap = 0;
    float *d_cen = nullptr; int *d_bounds = nullptr; int *d_depth = nullptr;
    unsigned long long *d_keys_in = nullptr, *d_keys_out = nullptr;
    void *d_sort_tmp = nullptr; size_t d_sort_bytes = 0;
    int *d_child = nullptr, *d_parent = nullptr, *d_leaf = nullptr, *d_mc = nullptr;
    float *d_bmin = nullptr, *d_bmax = nullptr;
    unsigned *d_arrive = nullptr;
    size_t d_build_cap = 0;
    std::vector<int32_t> h_child, h_leaf, h_mc;
    std::vector<float> h_bmin, h_bmax;
    MptVec4 *snode = nullptr, *fnode = nullptr, *tgeo = nullptr, *tshade = nullptr;
    MptVec4 *tfast = nullptr; size_t tfast_cap = 0;   // production triangle records, 3 float4 each (derived from tgeo)
    MptVec4 *qnode = nullptr; size_t qnode_cap = 0;   // the same nodes, child boxes quantised to 8 bits, 4 float4 each
    int use_quant = 1;
    MptVec4 *wnode = nullptr; size_t wnode_cap = 0;   // 4-wide nodes of the fast tree (gather kernel), 8 float4 each
    int wide_nodes = 0, wide_depth = 0;               // 0 nodes: not built (too deep)
    int wide_stack = 0;                               // stack levels a traversal of the 4-wide tree can ask for (exact from the host pass, 3 x depth + 2 from the device pass)
    float wide_ratio = 1.f;                           // expected fetches per ray, wide / binary (surface-area sums)
    int sah_exact_max = 8192;                         // host SAH pass: ranges up to this size are swept exactly (diagnostics)
    int sah_inject_fail = 0;                          // test door: treat the device SAH pass as failed after it ran
    int *h_sahmeta = nullptr, *d_sahmeta = nullptr;   // host-pinned, device-mapped [32]: the SAH pass's per-level hand-back (sah_build.hip plan kernel)
    MptSahStats sah_stats{};                          // what the last device SAH pass did
    bool d_model_stale = true;                        // the device copy of the model (d_verts, d_mtlids) is behind the host's: the next device build uploads
    int lane_hist = 0;                                // diagnostics: counting kernels fill the lane histogram (mpt_get_lane_hist)
    int build_phases = 0;                             // diagnostics: synchronise at the end of every phase of mpt_build_tree and time it
    double build_phase_us[6] = { 0, 0, 0, 0, 0, 0 };   // upload | LBVH | SAH pass | triangle records | 4-wide collapse | total (host clock)
    int sah_fallback = 0;                             // last build: the device SAH pass gave up (1: error, 2: depth) and the host pass ran
    int sah_build = -1;                               // SAH re-partition: 1 on the device (sah_build.hip), 0 host pass, -1 auto
                                                      // (device above 8192 faces; below, all of the host pass's splits are exact
                                                      // sweeps and it costs a millisecond)
    void *sah_ws = nullptr; size_t sah_ws_bytes = 0;  // one allocation, carved up in build_sah_device
    int wide_build = 1;                               // 1: the 4-wide collapse runs on the device (wide_build.hip), 0: host pass
    int *wb_bin_of = nullptr, *wb_ncount = nullptr, *wb_offset = nullptr; void *wb_scan = nullptr; size_t wb_scan_bytes = 0;
    double *wb_area = nullptr; size_t wb_cap = 0;
    int use_wide = 1;                                 // option "wide": 1 walk the 4-wide nodes when the scene does not fit LDS
                                                      // (default), 0 the binary tree
    // overflow strips of the wide kernel's per-lane stacks: one per ring slot, because launches of different slots
    // are resident together and index their strips by block and lane only
    int *stack_spill2[MPT_MAX_PIPE] = {}; size_t stack_spill2_cap[MPT_MAX_PIPE] = {};
    int node_soa = 0;                                 // option "node_soa" (layout A/B): binary gather kernel reads an SoA transpose
    MptVec4 *fnode_soa = nullptr; size_t fnode_soa_cap = 0; bool fnode_soa_valid = false;
    size_t node_cap = 0, tri_cap = 0;

    // materials / images / lights / world / camera
    MptMaterial *mats = nullptr;
    MptImage *images = nullptr;
    std::vector<MptImage> h_images;
    MptVec4 *texels = nullptr;
    size_t texels_used = 0;
    MptLight *lights = nullptr;
    std::vector<MptLight> h_lights;
    float world_fac[4] = { 0.1f, 0.1f, 0.1f, 0.1f };   // light/world.py:14-16
    int world_tex = -1;                                 // documented deviation Q6 (reference default 0)
    float v2w[16], w2v[16];

    // sobol
    int sdim = 0, srows = 0;
    int32_t stime = 0;
    int *sV = nullptr, *sX = nullptr;
    int *sX_spec = nullptr;              // the state the batch whose points were computed ahead of time will leave behind (swapped with sX when it is launched)
    float *sP = nullptr;                 // [MPT_MAX_BATCH][sdim]

    // command batching
    int pending = 0;

    // launch pipelining (fast build): batch i renders on rstream[i & 1] into partial[i & 1] while the main
    // stream still combines / gathers / resolves batch i-1, so one launch's tail overlaps the next one's head
    hipStream_t rstream[MPT_MAX_PIPE] = {};
    hipEvent_t ev_render[MPT_MAX_PIPE] = {};          // render of the batch on rstream[k] finished
    hipEvent_t ev_free[MPT_MAX_PIPE] = {};            // combine has consumed partial[k]
    hipStream_t probe_stream = nullptr;               // mpt_probe_kernel
    hipStream_t stress_stream = nullptr;              // mpt_stress_copies: a stream of device-to-device copies beside the render
    char *stress_buf = nullptr;                       // 2 x stress_bytes
    size_t stress_bytes = 0;
    hipStream_t aux = nullptr;                        // Sobol advances + queue resets of the pipelined batches
    hipEvent_t ev_sobol2[MPT_MAX_PIPE] = {};          // Sobol points + zeroed queue heads of the batch on rstream[k] ready
    int pipe_depth = 0;                               // batches in flight (slots of P / partial / queue heads); 0 = auto
    int reserve_cus = 0;                              // CUs no persistent workgroup claims (experiments: see mpt_flush)
    int grid_div = 0;                                 // each launch takes 1/grid_div of the CUs; 0 = auto
    int cur_depth = 2, cur_div = 1;                   // what the last launch used
    hipEvent_t ev_film = nullptr;                     // main-stream work on the film (combine, clear, gather) a finalising launch must see
    // tail finalisation (render_kernel.hip finalise_tiles): option "finalise" (1 = launches that find the ring idle sum, resolve and
    // write out their tiles themselves; 0 = always the combine pass); launch_seq numbers the launches (slab tags);
    // film_version counts the changes of pass 0; hint_image = where the next mpt_get_image(0) wants the image (mpt_hint_image);
    // early_* = the image a finalising launch has written (or is writing) and the film version it shows
    int finalise = 1;
    int spin_us = 20000;                              // mpt_get_image polls a finalising launch for this long before it blocks
    unsigned launch_seq = 0;
    unsigned tag_epoch = 0;                           // launch_seq / MPT_TAG_PERIOD when the slabs were last zeroed
    unsigned tag_wraps = 0;                           // times the slab tags came round (every slab zeroed): a test reads it
    unsigned long long film_version = 0;
    float *hint_image = nullptr;
    float *early_ptr = nullptr;
    hipStream_t early_stream = nullptr;               // the stream of the launch that writes early_ptr
    unsigned long long early_version = 0;
    int last_finalised = 0;                           // the last launch finalised its tiles itself (diagnostics, option "last_finalised")
    hipEvent_t ev_main = nullptr;                     // main-stream work a render must see (uploads, resets, ...)
    bool main_dirty = true;
    int flip = 0;
    // Sobol points of the NEXT batch, computed ahead of time into the ring slot it will use (the sequence is
    // deterministic): valid while nothing has touched the sampler or the ring since
    bool spec_valid = false;
    int spec_slot = -1, spec_B = 0;
    int32_t spec_time = 0;
    MptVec4 *partial2[MPT_MAX_PIPE] = {};
    size_t partial2_cap[MPT_MAX_PIPE] = {};           // float4 elements per buffer
    float *sP2[MPT_MAX_PIPE] = {};
    unsigned int *d_work2[MPT_MAX_PIPE] = {};

    // 8-wide octant-ordered tree (oct_build.cpp, render_kernel_oct): option "wide8" = 1 builds and walks it for scenes that do
    // not fit LDS; onode [oct_nodes][5], the triangle records in its leaf order (tfast8 / tshade8), d_perm8 [n]: t8 -> leaf slot
    int use_wide8 = 0;
    MptVec4 *onode = nullptr; size_t onode_cap = 0;
    MptVec4 *tfast8 = nullptr, *tshade8 = nullptr; int32_t *d_perm8 = nullptr; size_t tri8_cap = 0;
    int oct_nodes = 0, oct_depth = 0;

    // measurement
    int timeline = 0;                    // 1: the LDS kernel records per-wave timestamps of its last launch
    unsigned long long *d_timeline = nullptr;
    int timeline_waves = 0;
    unsigned long long *d_counters = nullptr;
    unsigned int *d_work = nullptr;
    unsigned int *h_watchdog = nullptr, *d_watchdog = nullptr;   // host-pinned, device-mapped: raised by a render kernel's watchdog
    void *h_stage = nullptr; size_t h_stage_bytes = 0;           // page-locked staging for read-backs into pageable buffers
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<hipEvent_t> event_pool;

    // comm
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    double *d_scratch = nullptr;
    // film gather of a striped split (comm.cpp): the share packed into one message per peer
    MptVec4 *gather_buf = nullptr; size_t gather_cap = 0;   // sender: its packed share; root: every peer's, back to back
    MptPiece *d_pieces = nullptr; size_t pieces_cap = 0;     // the plan's piece table on the device
    int plan_key[6] = { -1, -1, -1, -1, -1, -1 };            // (nx, ny, stripe_w, R, rank, root) the table was made for
    int plan_npieces = 0; long long plan_max_count = 0;
};

// entry checks of the API calls (miptina.cpp)
MPT_INTERNAL int use_ro(mpt_ctx *c);   // calls that only read results
MPT_INTERNAL int use(mpt_ctx *c);      // calls that may change what the next render launch reads
MPT_INTERNAL int check_pass(mpt_ctx *c, int pass);
MPT_INTERNAL int check_watchdog(mpt_ctx *c);   // after a synchronise: did a persistent kernel give up?

template <class T>
static int dev_alloc(T **p, size_t count) {
    HIP_TRY(hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(T)));
    return 0;
}

// comm.cpp
MPT_INTERNAL void mpt_comm_release(mpt_ctx *c);   // mpt_destroy: drop the communicator, if any

