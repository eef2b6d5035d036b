/*
 * miptina.h -- C ABI of libmiptina.so, the MI355X (gfx950) path-trace hot path that sits
 * underneath PTina's Python object API.
 *
 * The reference (archibate/ptina) has no FFI of its own: the path is reached through Python
 * singletons whose flattest statement is ptina/worker.py:11-87.  Each entry point below names
 * the reference call it replaces (file:line relative to the reference tree); the Python
 * package ptina_amd binds them with ctypes and keeps the reference's class and method names.
 *
 * Conventions
 *   - plain C types only; every pointer is a host pointer borrowed for the duration of the
 *     call (inputs are copied to the device, outputs are written into caller buffers);
 *   - functions return 0 on success, non-zero on error; mpt_last_error() gives the message
 *     (the Python layer raises RuntimeError with it, as the reference raises);
 *   - a context is NOT re-entrant: one caller thread per context (the reference funnels
 *     every call through one thread, ptina/tools/mtworker.py:22-42);
 *   - mpt_render* only ENQUEUE work (the reference's kernel launches are asynchronous too);
 *     the read-backs and mpt_synchronize block.  Consecutive mpt_render calls are batched
 *     into one launch at the next flush point.
 */
#ifndef MIPTINA_H
#define MIPTINA_H

#include <stddef.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpt_ctx mpt_ctx;

/* capacities, ptina/things.py:12-19 (init_things keyword arguments) */
typedef struct {
    int32_t max_faces;       /* 2^21 */
    int32_t max_texels;      /* 2^22 */
    int32_t max_materials;   /* 2^6  */
    int32_t max_textures;    /* 2^6  */
    int32_t max_lights;      /* 2^6  */
    int32_t max_filmsize;    /* 2^21 */
    int32_t max_filmpasses;  /* 3    */
} mpt_caps;

/* work counters of the traversal actually run (for the roofline's algorithmic bytes) */
typedef struct {
    uint64_t samples;        /* camera samples traced                    */
    uint64_t rays;           /* BVH traversals (closest-hit + shadow)    */
    uint64_t n_box;          /* box tests                                */
    uint64_t n_tri;          /* triangle tests                           */
    uint64_t n_shade;        /* shaded hits                              */
    uint64_t n_draws;        /* Sobol draws                              */
    uint64_t bounces;        /* path-loop iterations                     */
    uint64_t n_node;         /* internal-node records fetched            */
    uint64_t it_node;        /* wave-level NODE steps issued (n_node / (64 it_node) = lane utilisation) */
    uint64_t it_leaf;        /* wave-level LEAF steps issued             */
    uint64_t it_shade;       /* wave-level SHADE stages issued           */
    uint64_t it_new;         /* wave-level NEW stages issued             */
    /* the pooled LDS kernel (option "pool"): how its two kinds of waves spent the launch */
    uint64_t pl_local;       /* bounces a tracer wave did itself because the shade pool was full (lanes)        */
    uint64_t pl_batches;     /* SHADE batches of the shader waves                                               */
    uint64_t pl_batch_lanes; /* requests in those batches (pl_batch_lanes / (64 pl_batches) = SHADE lane use)   */
    uint64_t pl_prim;        /* batches of primary rays made by the shader waves                                */
    uint64_t pl_tidle;       /* polls of a tracer wave that held no path and found no ray                       */
    uint64_t pl_sidle;       /* polls of a shader wave that found nothing to do                                 */
    uint64_t pl_trips;       /* trips of a tracer wave from its traversal loop to the pools                     */
    uint64_t pl_taken;       /* rays taken out of the ray pool by tracer lanes                                  */
} mpt_counters;

#define MPT_LIGHT_POINT 1    /* LightPool.TYPES, ptina/light/__init__.py:11 */
#define MPT_LIGHT_AREA  2

/* render modes (mpt_set_option "mode") */
#define MPT_MODE_FAST   0    /* ordered, depth-culled traversal; any-hit shadow rays; FMA + native rcp/sin/cos */
#define MPT_MODE_STRICT 1    /* the reference's traversal order and IEEE arithmetic without contraction        */

const char *mpt_last_error(void);
int  mpt_device_count(void);
int  mpt_version(void);

/* init_things(), ptina/things.py:12-28 */
mpt_ctx *mpt_create(const mpt_caps *caps, int device);
void mpt_destroy(mpt_ctx *ctx);

/* "mode" (MPT_MODE_*), "batch" (max frames per launch, 1..64), "chunk" (frames per work item,
 * 0 = auto), "count" (1 = accumulate mpt_counters, slower), "lds" (1 = use the LDS-resident
 * persistent kernel when the scene fits a CU's 160 KiB LDS, default; 0 = always gather from HBM/L2),
 * "lds_wide" (what that kernel walks: 1 = the 4-wide nodes with exact boxes, 112 bytes apart in LDS, default -- needs "wide" = 1;
 * 0 = the binary nodes; the same film up to ties between equally distant hits).
 * "tree" (fast build: 1 = SAH re-partition of the LBVH's leaves, default; 0 = walk the LBVH itself;
 * takes effect at the next mpt_build_tree), "tile_w_shift"/"tile_h_shift" (work-item tile 2^w x 2^h pixels),
 * "wide" (scenes that do not fit LDS: 1 = walk the fast tree collapsed into 4-wide nodes, default; 0 = the
 * binary tree), "wide_quant" (1 = the 4-wide nodes as 64-byte records with 8-bit child boxes rounded outwards: four
 * gathers per step, default; 0 = 128-byte records with the exact boxes: seven), "wide_build" (1 = that collapse runs on the
 * device, default; 0 = host pass over downloaded records: same bytes), "sah_build" (where the SAH
 * re-partition runs: 1 = on the device, binned above 32 triangles and exact below; 0 = host pass, exact up to 8192; -1 = auto: the
 * device above 131072 faces, default), "gpu_build" (1 = LBVH built on the device, default), "sah_max" (faces above which the fast build
 * walks the LBVH itself; default 2^22), "grid_div" (each launch takes 1/G of the CUs so that G launches are resident
 * in different phases; 0 = choose by samples per lane, and the whole chip for a launch that finds nothing else in flight: default), "pipe_depth" (batches in flight,
 * 2..6; 0 = auto), "lds_block" (lanes per persistent workgroup of the LDS kernel, diagnostics),
 * "zero_copy" (1, default: mpt_get_image into an mpt_host_alloc array has the resolve pass write the image straight into it over
 * PCIe; 0: device buffer + DMA -- same image, 20 us more per call),
 * "skip_dark" (1 = a shadow ray whose candidate direct light is exactly zero -- the light behind the surface -- is not traced:
 * adding zero or not is the same sum; 0 = traced like the reference does; -1 = on in the production build, off in the strict build: default), "pool" /
 * "pool_shaders" (the LDS kernel with its waves specialised into tracers and shaders and two path pools in LDS between them:
 * measured slower, default off),
 * "spin_us" (how long mpt_get_image polls a finalising launch before it blocks; default 20000, 0 = block at once),
 * "wide8" (1: scenes that do not fit LDS walk the 8-wide octant-ordered tree instead of the 4-wide one; only in the A/B build `make oct` ->
 * libmiptina_oct.so since round 5 -- the product library refuses it; takes effect at the
 * next mpt_build_tree),
 * "finalise" (1, default: a render launch that finds no other launch in flight adds its frames to the film, resolves and
 * writes out finished tiles itself while its last paths drain; 0: always the combine pass after the launch; same film bit for bit),
 * "timeline" (1 = record mpt_get_timeline data), "reserve_cus" (CUs every persistent render launch leaves
 * unclaimed, default 0; measured to be of no use to foreign kernels while launches overlap, kept for experiments).
 * read-only: "tree_depth", "fast_depth", "wide_nodes", "wide_depth", "wide_stack" (stack levels a traversal of the 4-wide tree can
 * ask for), "wide_ratio_permille", "pending", "last_kernel" (0 = gather over the binary tree, 1 = LDS-resident over the binary
 * nodes, 2 = gather over 4-wide nodes, 3 = pooled LDS kernel (A/B library), 4 = gather over 8-wide nodes, 5 = LDS-resident over
 * the 4-wide nodes), "num_cus",
 * "cur_div", "cur_depth" (the ring the last launch belonged to: G launches of 1/G of the CUs, that many batches in
 * flight), "last_div" (what the last launch really took: 1 when it found the ring idle, else cur_div), "hw_queues"
 * (GPU_MAX_HW_QUEUES as the HIP runtime was asked for it -- the library requests 12 at load time unless the variable is
 * set; NEGATIVE when that request came after a preloaded profiler tool may already have initialised HIP) */
int mpt_set_option(mpt_ctx *ctx, const char *key, int value);
int mpt_get_option(mpt_ctx *ctx, const char *key, int *value);

/* FilmTable.set_size / nx,ny, ptina/filmtable.py:41-42,16-24; worker.set_size/get_size, worker.py:29-34 */
int mpt_set_size(mpt_ctx *ctx, int nx, int ny);
int mpt_get_size(mpt_ctx *ctx, int *nx, int *ny);
/* multi-GPU: this context renders film columns x in [x0, x1) only (no reference counterpart) */
int mpt_set_slab(mpt_ctx *ctx, int x0, int x1);
/* Deal the film out in stripes of `width` columns (a multiple of 16): this context renders stripes index,
 * index + modulo, ...  The multi-GPU split with even load; mpt_comm_gather_film follows it when every
 * rank r of R called mpt_set_stripes(width, r, R).  mpt_set_slab / mpt_set_size return to one slab. */
int mpt_set_stripes(mpt_ctx *ctx, int width, int index, int modulo);

/* ModelPool.from_numpy, ptina/model.py:54-60: verts [3n][8] = pos3 nrm3 uv2, mtlids [n] or NULL (= -1) */
int mpt_load_model(mpt_ctx *ctx, const float *verts, const int32_t *mtlids, int n);
/* MaterialPool.load, ptina/mtllib.py:58-77: fac [m][12][4], tex [m][12] (-1 = none) */
int mpt_load_materials(mpt_ctx *ctx, const float *fac, const int32_t *tex, int m);
/* ImagePool.load / load_one, ptina/image.py:69-96: rgba [nx][ny][4] f32; returns image id via *id */
int mpt_reset_images(mpt_ctx *ctx);
int mpt_load_image(mpt_ctx *ctx, const float *rgba, int nx, int ny, int *id);
/* BVHTree.build, ptina/tree/lbvh.py:297-305 */
int mpt_build_tree(mpt_ctx *ctx);
/* test/inspection: reference-layout tree arrays (tree/lbvh.py:48-56); any pointer may be NULL */
int mpt_get_tree(mpt_ctx *ctx, int32_t *child /*[n-1][2]*/, int32_t *leaf /*[n]*/,
                 float *bmin /*[n-1][3]*/, float *bmax /*[n-1][3]*/, int32_t *mc /*[n]*/, int32_t *depth);

/* test/inspection: the 4-wide records the gather kernels walk (no reference counterpart): wnode [nw][8][4] f32 with the exact
 * child boxes, qnode [nw][4][4] with 8-bit boxes; any pointer may be NULL; *nw = wide nodes built (0: none) */
int mpt_get_wide(mpt_ctx *ctx, float *wnode, float *qnode, int cap_nodes, int *nw);

/* test/inspection: the 8-wide octant-ordered records the option "wide8" kernel walks (no reference counterpart; layout in
 * ptina_amd/csrc/oct_build.cpp): onode [nw][5][4] f32, perm [n] = the leaf slot of the triangle at place t of that tree's leaf
 * order; any pointer may be NULL; *nw = 8-wide nodes built (0: none) */
int mpt_get_oct8(mpt_ctx *ctx, float *onode, int32_t *perm, int cap_nodes, int *nw);

/* Pure sizing rule of the on-device SAH re-partition's workspace (no context, no GPU; the reference has no counterpart: its
 * tree is the LBVH of ptina/tree/lbvh.py:297-305).  For a model of n faces: out[0] = segments a level can hold, out[1] = words
 * of per-segment workspace the build allocates, out[2] = words a level of `nseg` segments writes, out[3] = bins per axis at
 * that level.  The build is sound iff out[2] <= out[1] for every nseg <= out[0]; returns 0, or 1 for n < 1 / nseg < 0. */
int mpt_sah_workspace(int n, int64_t nseg, int64_t out[4]);

/* Camera.set_perspective, ptina/camera.py:19-22: the two f32 matrices the reference stores
 * (V2W = inv(pers) computed by the caller in f64 exactly as the reference does) */
int mpt_set_camera(mpt_ctx *ctx, const float v2w[16], const float w2v[16]);

/* LightPool.clear / add, ptina/light/__init__.py:31-49 (pos/axes already extracted from the world matrix) */
int mpt_clear_lights(mpt_ctx *ctx);
int mpt_add_light(mpt_ctx *ctx, int type, const float color[3], const float pos[3],
                  const float axes[9], float size, int *index);
/* WorldLight.set, ptina/light/world.py:18-20 */
int mpt_set_world_light(mpt_ctx *ctx, const float fac[4], int tex);

/* SobolSampler.__init__ / reset / update, ptina/sampling/sobol.py:75-105.
 * V = direction-number grid [rows][dim] (bit pattern, i32) from calc_sobol_vgrid */
int mpt_sobol_init(mpt_ctx *ctx, const int32_t *V, int rows, int dim);
int mpt_sobol_reset(mpt_ctx *ctx, int skip);
int mpt_sobol_update(mpt_ctx *ctx, int count);
int mpt_sobol_get(mpt_ctx *ctx, int32_t *X, float *P, int32_t *time);

/* PathEngine.render, ptina/engine/path.py:75-77 : nframes x (Sobol update + one sample per pixel) */
int mpt_render(mpt_ctx *ctx, int nframes);
/* PreviewEngine.render, ptina/engine/preview.py:18-41 : albedo -> pass 1, normal -> pass 2 */
int mpt_render_preview(mpt_ctx *ctx, int nframes);
/* launch everything enqueued so far (does not wait) */
int mpt_flush(mpt_ctx *ctx);
/* worker.synchronize, ptina/worker.py:17-18 */
int mpt_synchronize(mpt_ctx *ctx);

/* FilmTable.clear, ptina/filmtable.py:44-45 (zeroes every pass, whatever `pass` says, as the reference does) */
int mpt_clear(mpt_ctx *ctx, int pass);
/* FilmTable.get_image, ptina/filmtable.py:47-63 : out [nx][ny][4] */
int mpt_get_image(mpt_ctx *ctx, int pass, float *out);
/* Advice, optional (the reference's get_image allocates its array inside the call, filmtable.py:48; this says beforehand where
 * that array will be): the next mpt_get_image(pass, out) will be given THIS `out` (a buffer of mpt_host_alloc, [nx][ny][4]).
 * A render launch that finalises its own tiles (option "finalise") then also writes the resolved image there while it drains,
 * and that mpt_get_image only waits for the launch.  Results never depend on it: a different pointer, a film that has changed
 * since, or no hint at all take the resolve pass.  `out` must stay allocated until mpt_get_image(pass, out) has returned or
 * another hint (NULL = none) has replaced it; only pass 0 is used. */
int mpt_hint_image(mpt_ctx *ctx, int pass, float *out);
/* FilmTable.fast_export_image, ptina/filmtable.py:66-79 : out [ny*nx*3] */
int mpt_fast_export_image(mpt_ctx *ctx, int pass, float *out);
/* raw accumulators [nx*ny][4] (rgb sums, sample count) */
int mpt_get_film_raw(mpt_ctx *ctx, int pass, float *out);
/* device-side resolve only (no read-back): what get_image does before the copy */
int mpt_resolve(mpt_ctx *ctx, int pass);

/* Page-locked host buffers for the read-backs above: into such a buffer mpt_get_image /
 * mpt_fast_export_image / mpt_get_film_raw are one DMA; any other buffer is served through a
 * page-locked staging copy.  (The reference's get_image returns a fresh numpy array,
 * ptina/filmtable.py:48; FilmTable.get_image here builds that array on a recycled buffer of these.) */
void *mpt_host_alloc(size_t bytes);
void  mpt_host_free(void *p);

/* measurement */
int mpt_get_counters(mpt_ctx *ctx, mpt_counters *out);
/* Diagnostics (option "timeline" = 1): per wave of the last LDS-kernel launch, eight words: four 100 MHz timestamps
 * {start, scene copied to LDS, work queues found empty, exit} and, in a -DMPT_X_TIMELINE2 build only (else 0), {time of the
 * last work item pulled, items pulled, lanes in flight when the queues were found empty, shading passes after that}.
 * *nwaves = waves recorded. */
int mpt_get_timeline(mpt_ctx *ctx, unsigned long long *out /* [cap_waves][8] */, int cap_waves, int *nwaves);
int mpt_reset_counters(mpt_ctx *ctx);
/* Diagnostics (options "lane_hist" = 1 and "count" = 1; no counterpart in the reference): out[0 .. 195) = how many NODE / LEAF / SHADE
 * stages the waves issued with k of their 64 lanes taking part ([3][65]); out[195 .. 231) = the lane-steps of those stages by bounce
 * depth and ray kind ([3][6][closest, shadow]); out[231 .. 255) = the gather kernels' NODE lane-steps by log2 bucket of the node's
 * (breadth-first) number.  n = words `out` holds (>= 255). */
int mpt_get_lane_hist(mpt_ctx *ctx, unsigned long long *out, int n);
/* Diagnostics: launch a one-workgroup kernel (`threads` lanes, `lds_bytes` of LDS) on a stream of its own
 * while the enqueued render launches keep running, and return the wall time until it has completed --
 * what a collective's kernel would wait for a CU beside the persistent render workgroups. */
int mpt_probe_kernel(mpt_ctx *ctx, int threads, int lds_bytes, double *usec);
/* Test door (the tail finalisation's soak: tools/soak.py, tests/test_parity_gpu.py): enqueue `count` device-to-device copies of
 * `mbytes` MiB on a stream of their own and return at once -- HBM / L2 traffic beside the render launches; count = 0 waits for
 * the copies enqueued so far.  Nothing in the reference corresponds (its film add is one kernel's `+=`, ptina/engine/path.py:93). */
int mpt_stress_copies(mpt_ctx *ctx, int mbytes, int count);
/* HIP-event time of the render kernels launched since the last call (ms) and their count */
int mpt_kernel_time(mpt_ctx *ctx, double *ms, int *launches);

/* Test door: ONE device function of the hot path evaluated on n rows of inputs by the build the context's "mode"
 * selects (the strict build's reference-order IEEE code or the production build's fast forms) -- the same inlined
 * functions the render kernels run.  Rows are 4-byte words (f32; i32 for the two hash kinds); the column counts
 * per kind are fixed (checked).  Holds the HIP code directly to vectors computed by the reference's own function
 * bodies (tests/golden/reference_l1.npz); each kind names the reference function it evaluates. */
enum {
    MPT_UNIT_SCHLICK = 0,         /* materials/microfacet.py:9-10   in: cos                                  out: 1 */
    MPT_UNIT_DIELECTRIC = 1,      /* materials/microfacet.py:14-27  in: etai, etao, cosi                     out: 1 */
    MPT_UNIT_GTR1 = 2,            /* materials/microfacet.py:31-34  in: cosh, alpha                          out: 1 */
    MPT_UNIT_GTR2 = 3,            /* materials/microfacet.py:38-41  in: cosh, alpha                          out: 1 */
    MPT_UNIT_SMITHGGX = 4,        /* materials/microfacet.py:45-48  in: cos, alpha                           out: 1 */
    MPT_UNIT_SAMPLE_GTR1 = 5,     /* materials/microfacet.py:69-71  in: u, v, alpha                          out: 3 */
    MPT_UNIT_SAMPLE_GTR2 = 6,     /* materials/microfacet.py:75-77  in: u, v, alpha                          out: 3 */
    MPT_UNIT_TANSPACE = 7,        /* common.py:213-217              in: normal3, v3                          out: tanspace(normal) @ v */
    MPT_UNIT_SPHERICAL = 8,       /* common.py:221-225              in: h, p                                 out: 3 */
    MPT_UNIT_DIR2TEX = 9,         /* common.py:234-239              in: dir3                                 out: 2 */
    MPT_UNIT_REFLECT = 10,        /* common.py:247-249              in: I3, N3                               out: 3 */
    MPT_UNIT_REFRACT = 11,        /* common.py:252-260              in: I3, N3, eta                          out: has_r, T3 */
    MPT_UNIT_BOX = 12,            /* geometries.py:24-46            in: lo3, hi3, o3, d3                     out: hit, near, far (fast build: far = -1) */
    MPT_UNIT_FACE = 13,           /* geometries.py:96-148           in: v0 v1 v2, o3, d3, vn0 vn1 vn2, vt0 vt1 vt2 (30)  out: hit, depth, s, t, normal3, texcoord2 */
    MPT_UNIT_SPHERE = 14,         /* geometries.py:159-177          in: pos3, rad2, o3, d3                   out: t */
    MPT_UNIT_AREA = 15,           /* geometries.py:58-74            in: pos3, dirx3, diry3, o3, d3           out: hit, depth, u, v */
    MPT_UNIT_DISNEY_BRDF = 16,    /* materials/disney.py:13-106     in: 14 parameters, normal3, sign, indir3, outdir3 (24)  out: rgb */
    MPT_UNIT_DISNEY_BOUNCE = 17,  /* materials/disney.py:13-50,115-233  in: 14 parameters, normal3, sign, indir3, samp3 (24) out: outdir3, pdf, color3 */
    MPT_UNIT_POWER_HEURISTIC = 18,/* engine/path.py:11-15           in: a, b                                 out: 1 */
    MPT_UNIT_WANGHASH = 19,       /* sampling/__init__.py:9-16      in: i32                                  out: i32 */
    MPT_UNIT_WANGHASH2 = 20,      /* sampling/__init__.py:20-23     in: i32, i32                             out: i32 */
    MPT_UNIT_KINDS = 21
};
int mpt_unit_eval(mpt_ctx *ctx, int kind, const void *in, int in_cols, void *out, int out_cols, int n);

/* multi-GPU film gather over RCCL (one process per GPU).  uid = ncclUniqueId bytes (128). */
int mpt_comm_unique_id(char uid[128]);
int mpt_comm_init(mpt_ctx *ctx, const char uid[128], int nranks, int rank);
/* The split of the film between R ranks as a pure function (no context, no GPU): the float4 ranges (film index
 * x*ny + y) rank r owns, in ascending x = the order they are packed into its one message of the gather.
 * stripe_w == 0: one slab [r*nx/R, (r+1)*nx/R); stripe_w > 0: stripes r, r+R, ... of stripe_w columns.  Writes the
 * first `cap` pieces (offsets / counts in float4 elements; either may be NULL) and returns the number of pieces,
 * -1 for arguments that name no split. */
int mpt_comm_plan(int nx, int ny, int stripe_w, int r, int R, int64_t *offsets, int64_t *counts, int cap);
/* One message per peer: a share of several stripes is packed side by side, sent as one range and scattered into
 * the root's film by one kernel in front of the resolve; a one-range share travels film to film. */
int mpt_comm_gather_film(mpt_ctx *ctx, int pass, int root);
/* Test door, one GPU, no communicator: the pack and scatter halves of the gather for rank `as_rank` of `nranks`
 * sending to `root`, with a device copy standing in for the message.  film_in: the sender's film [nx*ny][4];
 * film_out: the root's film, updated in place.  Uses the context's film size and stripe width. */
int mpt_comm_selftest(mpt_ctx *ctx, int as_rank, int nranks, int root, const float *film_in, float *film_out);
int mpt_comm_barrier(mpt_ctx *ctx);
int mpt_comm_allreduce_max(mpt_ctx *ctx, double *value);
int mpt_comm_destroy(mpt_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
