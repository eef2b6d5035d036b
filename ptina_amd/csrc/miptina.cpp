// miptina.cpp -- host runtime behind the C ABI of include/miptina.h.
//
// Owns the device state of one PTina "scene" (the singletons of ptina/things.py:20-28 collapsed
// into one context), builds the LBVH, batches enqueued frames into single launches, and gathers
// film slabs across GPUs with RCCL.  No PyTorch, no Python: plain HIP runtime calls.

#include "../../include/miptina.h"
#include "mpt_types.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

// kernel launchers (render_kernel.hip x2, aux_kernels.hip)
extern "C" hipError_t mpt_launch_render_fast(const MptRenderParams *, int grid, int stack, int count, hipStream_t);
extern "C" hipError_t mpt_launch_render_strict(const MptRenderParams *, int grid, int stack, int count, hipStream_t);
extern "C" hipError_t mpt_launch_render_lds(const MptRenderParams *, int grid, int block, size_t lds_bytes, int count, hipStream_t);
extern "C" hipError_t mpt_launch_preview_fast(const MptRenderParams *, int grid, int stack, hipStream_t);
extern "C" hipError_t mpt_launch_preview_strict(const MptRenderParams *, int grid, int stack, hipStream_t);
extern "C" hipError_t mpt_launch_sobol_update(int *X, const int *V, float *P, int dim, int rows, int time0, int count,
                                              int keep, hipStream_t);
extern "C" hipError_t mpt_launch_combine(MptVec4 *film, const MptVec4 *partial, int nx, int ny, int x0, int x1,
                                         int stripe_w, int stripe_pitch,
                                         int nchunks, hipStream_t);
extern "C" hipError_t mpt_launch_resolve(const MptVec4 *film, MptVec4 *out, size_t npix, hipStream_t);
extern "C" hipError_t mpt_launch_export(const MptVec4 *film, float *out, int nx, int ny, hipStream_t);

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
extern "C" hipError_t mpt_lbvh_sort_bytes(int n, size_t *bytes);
extern "C" hipError_t mpt_lbvh_build(const MptLbvhBuffers *b, hipStream_t stream);

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;

static int fail(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) return fail("%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" const char *mpt_last_error(void) { return g_err.c_str(); }
extern "C" int mpt_version(void) { return 100; }

extern "C" int mpt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------ RCCL, bound lazily
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl g_rccl;

static int rccl_load() {
    if (g_rccl.h) return 0;
    const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    void *h = nullptr;
    for (const char *nm : names) {
        h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return fail("cannot load librccl: %s", dlerror());
#define SYM(field, name)                                                        \
    *(void **)(&g_rccl.field) = dlsym(h, name);                                 \
    if (!g_rccl.field) return fail("librccl lacks symbol %s", name);
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(AllReduce, "ncclAllReduce");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.h = h;
    return 0;
}

#define NCCL_TRY(expr)                                                                          \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess) return fail("%s failed: %s", #expr, g_rccl.GetErrorString(r_));  \
    } while (0)

// Up to MPT_MAX_PIPE render streams, the main stream and the aux stream carry work at the same time.
// The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless told
// otherwise) and streams that share one are serialised, so ask for more before the runtime starts --
// unless the user has chosen a value.
__attribute__((constructor)) static void mpt_want_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

// ------------------------------------------------------------------ context
enum { MPT_MAX_PIPE = 6 };

struct mpt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    mpt_caps caps{};

    // options
    int mode = MPT_MODE_FAST, batch = 32, chunk = 0, count = 0, use_lds = 1, lds_block = 0;
    int num_cus = 256;
    int tile_w_shift = 3, tile_h_shift = 3;   // work-item tile 2^w x 2^h pixels
    int sched_num = 2, sched_den = 1;    // scheduler: stay in traversal mode while traversing*num >= waiting*den (tuned on MI355X)
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
    std::vector<int32_t> mtlids;
    bool tree_valid = false;
    int tree_depth = 0;                  // reference LBVH (strict build)
    int fast_depth = 0;                  // tree the fast build walks (SAH or LBVH)
    int tree_kind = 1;                   // fast build: 1 = SAH re-partition of the LBVH's leaves, 0 = the LBVH itself
    int gpu_build = 1;                   // 1 = LBVH built on the device (lbvh_build.hip), 0 = host build
    int sah_max = 1 << 18;               // above this many faces the fast build walks the LBVH itself
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
    float *sP = nullptr;                 // [MPT_MAX_BATCH][sdim]

    // command batching
    int pending = 0;

    // launch pipelining (fast build): batch i renders on rstream[i & 1] into partial[i & 1] while the main
    // stream still combines / gathers / resolves batch i-1, so one launch's tail overlaps the next one's head
    hipStream_t rstream[MPT_MAX_PIPE] = {};
    hipEvent_t ev_render[MPT_MAX_PIPE] = {};          // render of the batch on rstream[k] finished
    hipEvent_t ev_free[MPT_MAX_PIPE] = {};            // combine has consumed partial[k]
    hipStream_t aux = nullptr;                        // Sobol advances + queue resets of the pipelined batches
    hipEvent_t ev_sobol2[MPT_MAX_PIPE] = {};          // Sobol points + zeroed queue heads of the batch on rstream[k] ready
    int pipe_depth = 0;                               // batches in flight (slots of P / partial / queue heads); 0 = auto
    int grid_div = 0;                                 // each launch takes 1/grid_div of the CUs; 0 = auto
    int cur_depth = 2, cur_div = 1;                   // what the last launch used
    hipEvent_t ev_main = nullptr;                     // main-stream work a render must see (uploads, resets, ...)
    bool main_dirty = true;
    int flip = 0;
    MptVec4 *partial2[MPT_MAX_PIPE] = {};
    size_t partial2_cap[MPT_MAX_PIPE] = {};           // float4 elements per buffer
    float *sP2[MPT_MAX_PIPE] = {};
    unsigned int *d_work2[MPT_MAX_PIPE] = {};

    // measurement
    int timeline = 0;                    // 1: the LDS kernel records per-wave timestamps of its last launch
    unsigned long long *d_timeline = nullptr;
    int timeline_waves = 0;
    unsigned long long *d_counters = nullptr;
    unsigned int *d_work = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<hipEvent_t> event_pool;

    // comm
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    double *d_scratch = nullptr;
};

static int use_ro(mpt_ctx *c) {   // entry of calls that only read results
    if (!c) return fail("null context");
    HIP_TRY(hipSetDevice(c->device));
    return 0;
}

static int use(mpt_ctx *c) {      // entry of calls that may change what the next render launch reads
    if (use_ro(c)) return 1;
    c->main_dirty = true;
    return 0;
}

template <class T>
static int dev_alloc(T **p, size_t count) {
    HIP_TRY(hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(T)));
    return 0;
}

static void default_light(mpt_ctx *c) {
    // light/__init__.py:22-28: one POINT light at (1,2,3), radius 0.5, colour 32
    MptLight L{};
    L.color_size = { 32.f, 32.f, 32.f, 0.5f };
    L.pos_type = { 1.f, 2.f, 3.f, 0.f };
    int t = MPT_LIGHT_POINT;
    memcpy(&L.pos_type.w, &t, 4);
    c->h_lights.assign(1, L);
}

static int upload_lights(mpt_ctx *c) {
    if (!c->h_lights.empty())
        HIP_TRY(hipMemcpyAsync(c->lights, c->h_lights.data(), c->h_lights.size() * sizeof(MptLight),
                               hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

static int make_render_streams(mpt_ctx *c) {
    for (int k = 0; k < MPT_MAX_PIPE; k++)
        if (!c->rstream[k]) HIP_TRY(hipStreamCreateWithFlags(&c->rstream[k], hipStreamNonBlocking));
    return 0;
}

extern "C" mpt_ctx *mpt_create(const mpt_caps *caps, int device) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        fail("no HIP device available (%s): the MI355X path has no CPU fallback",
             e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= ndev) { fail("device %d out of range (%d devices)", device, ndev); return nullptr; }
    mpt_ctx *c = new mpt_ctx();
    c->device = device;
    mpt_caps d = { 1 << 21, 1 << 22, 1 << 6, 1 << 6, 1 << 6, 1 << 21, 3 };
    c->caps = caps ? *caps : d;
    if (c->caps.max_lights > MPT_MAX_LIGHTS) c->caps.max_lights = MPT_MAX_LIGHTS;
    if (c->caps.max_filmpasses < 3) c->caps.max_filmpasses = 3;
    auto bail = [&](const char *what) -> mpt_ctx * {
        std::string m = g_err;
        fail("mpt_create: %s: %s", what, m.c_str());
        delete c;
        return nullptr;
    };
    if (hipSetDevice(device) != hipSuccess) return bail("hipSetDevice");
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail("stream");
    if (dev_alloc(&c->mats, c->caps.max_materials)) return bail("materials");
    if (dev_alloc(&c->images, c->caps.max_textures)) return bail("images");
    if (dev_alloc(&c->lights, MPT_MAX_LIGHTS)) return bail("lights");
    if (dev_alloc(&c->d_counters, 12)) return bail("counters");
    if (dev_alloc(&c->d_scratch, 2)) return bail("scratch");
    if (dev_alloc(&c->d_work, 16)) return bail("work counters");   // 8 queue heads + [8] watchdog flag
    for (int k = 0; k < MPT_MAX_PIPE; k++) {
        if (hipEventCreateWithFlags(&c->ev_sobol2[k], hipEventDisableTiming) != hipSuccess) return bail("event");
        if (hipEventCreateWithFlags(&c->ev_render[k], hipEventDisableTiming) != hipSuccess) return bail("event");
        if (hipEventCreateWithFlags(&c->ev_free[k], hipEventDisableTiming) != hipSuccess) return bail("event");
        if (dev_alloc(&c->d_work2[k], 16)) return bail("work counters");
        hipMemsetAsync(c->d_work2[k], 0, 16 * sizeof(unsigned int), c->stream);
    }
    if (hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking) != hipSuccess) return bail("aux stream");
    if (hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming) != hipSuccess) return bail("event");
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            c->num_cus = prop.multiProcessorCount;
    }
    if (make_render_streams(c)) return bail("render streams");
    hipMemsetAsync(c->d_counters, 0, 12 * sizeof(unsigned long long), c->stream);
    hipMemsetAsync(c->d_work, 0, 16 * sizeof(unsigned int), c->stream);
    {   // unset materials: factor 0 (field-zero, mtllib.py:12-13), texture -1 (deviation Q6)
        std::vector<MptMaterial> z(c->caps.max_materials);
        for (auto &m : z) { memset(&m, 0, sizeof m); for (int k = 0; k < 12; k++) m.tex[k] = -1; }
        hipMemcpyAsync(c->mats, z.data(), z.size() * sizeof(MptMaterial), hipMemcpyHostToDevice, c->stream);
        hipStreamSynchronize(c->stream);
    }
    for (int i = 0; i < 16; i++) c->v2w[i] = c->w2v[i] = (i % 5 == 0) ? 1.f : 0.f;
    default_light(c);
    if (upload_lights(c)) return bail("lights upload");
    return c;
}

extern "C" void mpt_destroy(mpt_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    for (int k = 0; k < MPT_MAX_PIPE; k++) if (c->rstream[k]) hipStreamSynchronize(c->rstream[k]);
    if (c->aux) { hipStreamSynchronize(c->aux); }
    hipStreamSynchronize(c->stream);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    if (c->aux) hipStreamDestroy(c->aux);
    for (int k = 0; k < MPT_MAX_PIPE; k++) {
        if (c->rstream[k]) hipStreamDestroy(c->rstream[k]);
        if (c->ev_render[k]) hipEventDestroy(c->ev_render[k]);
        if (c->ev_free[k]) hipEventDestroy(c->ev_free[k]);
        if (c->ev_sobol2[k]) hipEventDestroy(c->ev_sobol2[k]);
        hipFree(c->partial2[k]); hipFree(c->sP2[k]); hipFree(c->d_work2[k]);
    }
    if (c->ev_main) hipEventDestroy(c->ev_main);
    for (auto &pr : c->events) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    for (auto &ev : c->event_pool) hipEventDestroy(ev);
    for (int p = 0; p < 3; p++) hipFree(c->film[p]);
    hipFree(c->resolved); hipFree(c->exported);
    hipFree(c->snode); hipFree(c->fnode); hipFree(c->tgeo); hipFree(c->tshade);
    hipFree(c->mats); hipFree(c->images); hipFree(c->texels); hipFree(c->lights);
    hipFree(c->sV); hipFree(c->sX); hipFree(c->sP);
    hipFree(c->d_counters); hipFree(c->d_scratch); hipFree(c->d_work); hipFree(c->d_timeline);
    hipFree(c->d_verts); hipFree(c->d_mtlids); hipFree(c->d_cen); hipFree(c->d_bounds); hipFree(c->d_depth);
    hipFree(c->d_keys_in); hipFree(c->d_keys_out); hipFree(c->d_sort_tmp);
    hipFree(c->d_child); hipFree(c->d_parent); hipFree(c->d_leaf); hipFree(c->d_mc);
    hipFree(c->d_bmin); hipFree(c->d_bmax); hipFree(c->d_arrive);
    hipStreamDestroy(c->stream);
    delete c;
}

// ------------------------------------------------------------------ options
extern "C" int mpt_set_option(mpt_ctx *c, const char *key, int value) {
    if (use(c)) return 1;
    if (mpt_flush(c)) return 1;
    std::string k = key ? key : "";
    if (k == "mode") {
        if (value != MPT_MODE_FAST && value != MPT_MODE_STRICT) return fail("mode must be 0 (fast) or 1 (strict)");
        c->mode = value;
    } else if (k == "batch") {
        if (value < 1 || value > MPT_MAX_BATCH) return fail("batch must be in 1..%d", MPT_MAX_BATCH);
        c->batch = value;
    } else if (k == "chunk") {
        if (value < 0) return fail("chunk must be >= 0");
        c->chunk = value;
    } else if (k == "count") {
        c->count = value ? 1 : 0;
    } else if (k == "lds") {
        c->use_lds = value ? 1 : 0;
    } else if (k == "timeline") {
        c->timeline = value ? 1 : 0;
    } else if (k == "pipe_depth") {
        if (value != 0 && (value < 2 || value > MPT_MAX_PIPE)) return fail("pipe_depth must be 0 (auto) or 2..%d", MPT_MAX_PIPE);
        c->pipe_depth = value;
    } else if (k == "grid_div") {
        if (value < 0 || value > 8) return fail("grid_div must be 0 (auto) or 1..8");
        c->grid_div = value;
    } else if (k == "lds_block") {
        if (value != 0 && value != 256 && value != 512 && value != 768 && value != 1024)
            return fail("lds_block must be 0 (auto), 256, 512, 768 or 1024");
        c->lds_block = value;
    } else if (k == "tree") {
        if (value != 0 && value != 1) return fail("tree must be 0 (LBVH) or 1 (SAH)");
        if (value != c->tree_kind) { c->tree_kind = value; c->tree_valid = false; }
    } else if (k == "tile_w_shift" || k == "tile_h_shift") {
        if (value < 0 || value > 3) return fail("%s must be in 0..3", k.c_str());
        (k == "tile_w_shift" ? c->tile_w_shift : c->tile_h_shift) = value;
    } else if (k == "gpu_build") {
        if ((value ? 1 : 0) != c->gpu_build) { c->gpu_build = value ? 1 : 0; c->tree_valid = false; }
    } else if (k == "sah_max") {
        c->sah_max = value; c->tree_valid = false;
    } else if (k == "sched_num") {
        if (value < 1) return fail("sched_num must be >= 1");
        c->sched_num = value;
    } else if (k == "sched_den") {
        if (value < 0) return fail("sched_den must be >= 0");
        c->sched_den = value;
    } else {
        return fail("unknown option '%s'", k.c_str());
    }
    return 0;
}

extern "C" int mpt_get_option(mpt_ctx *c, const char *key, int *value) {
    if (!c || !value) return fail("null argument");
    std::string k = key ? key : "";
    if (k == "mode") *value = c->mode;
    else if (k == "batch") *value = c->batch;
    else if (k == "chunk") *value = c->chunk;
    else if (k == "count") *value = c->count;
    else if (k == "tree_depth") *value = c->tree_depth;
    else if (k == "fast_depth") *value = c->fast_depth;
    else if (k == "tree") *value = c->tree_kind;
    else if (k == "gpu_build") *value = c->gpu_build;
    else if (k == "pending") *value = c->pending;
    else if (k == "lds") *value = c->use_lds;
    else if (k == "lds_block") *value = c->lds_block;
    else if (k == "pipe_depth") *value = c->pipe_depth;
    else if (k == "grid_div") *value = c->grid_div;
    else if (k == "cur_depth") *value = c->cur_depth;
    else if (k == "cur_div") *value = c->cur_div;
    else if (k == "last_kernel") *value = c->last_kernel;
    else if (k == "num_cus") *value = c->num_cus;
    else return fail("unknown option '%s'", k.c_str());
    return 0;
}

// ------------------------------------------------------------------ film
extern "C" int mpt_set_size(mpt_ctx *c, int nx, int ny) {
    if (use(c)) return 1;
    if (mpt_flush(c)) return 1;
    if (nx <= 0 || ny <= 0) return fail("film size must be positive, got %dx%d", nx, ny);
    size_t npix = (size_t)nx * ny;
    if (npix > (size_t)c->caps.max_filmsize)
        return fail("film %dx%d exceeds max_filmsize=%d (init_things(max_filmsize=...))", nx, ny, c->caps.max_filmsize);
    if (npix > c->film_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int p = 0; p < 3; p++) { hipFree(c->film[p]); c->film[p] = nullptr; }
        hipFree(c->resolved); c->resolved = nullptr;
        hipFree(c->exported); c->exported = nullptr;
        for (int p = 0; p < 3; p++) {
            if (dev_alloc(&c->film[p], npix)) return 1;
            HIP_TRY(hipMemsetAsync(c->film[p], 0, npix * sizeof(MptVec4), c->stream));
        }
        if (dev_alloc(&c->resolved, npix)) return 1;
        if (dev_alloc(&c->exported, npix * 3)) return 1;
        c->film_cap = npix;
    }
    // the reference keeps one flat buffer and only changes `res` (filmtable.py:41-42): stale sums
    // of another resolution are the caller's to clear(); same here.
    c->nx = nx; c->ny = ny; c->x0 = 0; c->x1 = nx; c->stripe_w = 0; c->stripe_idx = 0; c->stripe_mod = 1;
    return 0;
}

extern "C" int mpt_get_size(mpt_ctx *c, int *nx, int *ny) {
    if (!c) return fail("null context");
    if (nx) *nx = c->nx;
    if (ny) *ny = c->ny;
    return 0;
}

extern "C" int mpt_set_slab(mpt_ctx *c, int x0, int x1) {
    if (use(c)) return 1;
    if (mpt_flush(c)) return 1;
    if (x0 < 0 || x1 > c->nx || x0 > x1) return fail("slab [%d,%d) outside film width %d", x0, x1, c->nx);
    c->x0 = x0; c->x1 = x1;
    c->stripe_w = 0; c->stripe_idx = 0; c->stripe_mod = 1;
    return 0;
}

// The film dealt out in stripes of `width` columns: this context renders stripes index, index + modulo,
// ...  A contiguous slab per GPU leaves the GPUs unevenly loaded (the centre columns of a Cornell view
// cost 25 % more than the outer ones); interleaved stripes even that out.
extern "C" int mpt_set_stripes(mpt_ctx *c, int width, int index, int modulo) {
    if (use(c)) return 1;
    if (mpt_flush(c)) return 1;
    if (c->nx <= 0) return fail("film size not set: call set_size() first");
    if (width <= 0 || width % MPT_TILE != 0) return fail("stripe width must be a positive multiple of %d", MPT_TILE);
    if (modulo < 1 || index < 0 || index >= modulo) return fail("stripe index %d outside [0, %d)", index, modulo);
    if ((long long)width * modulo > (1 << 30)) return fail("stripe pitch too large");
    c->stripe_w = width; c->stripe_idx = index; c->stripe_mod = modulo;
    c->x0 = std::min((long long)index * width, (long long)c->nx); c->x1 = c->nx;
    return 0;
}

// columns of this context's share, and its tile columns for tiles `tile` pixels wide
static void share_extent(const mpt_ctx *c, int tile, long long *cols, int *tile_cols) {
    long long n = 0; int t = 0;
    if (c->stripe_w == 0) {
        n = c->x1 - c->x0; t = (c->x1 - c->x0 + tile - 1) / tile;
    } else {
        for (long long x = (long long)c->stripe_idx * c->stripe_w; x < c->nx; x += (long long)c->stripe_w * c->stripe_mod) {
            int w = (int)std::min<long long>(c->stripe_w, c->nx - x);
            n += w; t += (w + tile - 1) / tile;
        }
    }
    if (cols) *cols = n;
    if (tile_cols) *tile_cols = t;
}

// ------------------------------------------------------------------ scene upload
extern "C" int mpt_load_model(mpt_ctx *c, const float *verts, const int32_t *mtlids, int n) {
    if (use(c)) return 1;
    if (mpt_flush(c)) return 1;
    if (n < 0 || (n > 0 && !verts)) return fail("bad model arguments");
    if (n >= c->caps.max_faces) return fail("too many faces");                // model.py:84
    c->nfaces = n;
    c->verts.assign(verts, verts + (size_t)n * 24);
    if (mtlids) c->mtlids.assign(mtlids, mtlids + n);
    else c->mtlids.assign(n, -1);                                             // model.py:80-81
    for (int i = 0; i < n; i++)
        if (c->mtlids[i] < -1 || c->mtlids[i] >= c->caps.max_materials)
            return fail("material id %d of face %d outside [-1, %d)", c->mtlids[i], i, c->caps.max_materials);
    c->tree_valid = false;
    return 0;
}

extern "C" int mpt_load_materials(mpt_ctx *c, const float *fac, const int32_t *tex, int m) {
    if (use(c)) return 1;
    if (mpt_flush(c)) return 1;
    if (m < 0 || m > c->caps.max_materials) return fail("%d materials exceed max_materials=%d", m, c->caps.max_materials);
    std::vector<MptMaterial> h(std::max(m, 1));
    for (int i = 0; i < m; i++) {
        MptMaterial &M = h[i];
        memset(&M, 0, sizeof M);
        const float *f = fac + (size_t)i * 48;
        M.p[0] = f[0]; M.p[1] = f[1]; M.p[2] = f[2];                          // basecolor .xyz, mtllib.py:82
        for (int k = 1; k < 12; k++) M.p[2 + k] = f[k * 4];                   // scalars take .x, mtllib.py:83-93
        M.any_tex = 0;
        for (int k = 0; k < 12; k++) {
            int t = tex ? tex[(size_t)i * 12 + k] : -1;
            if (t < -1 || t >= c->caps.max_textures) return fail("texture id %d outside [-1, %d)", t, c->caps.max_textures);
            M.tex[k] = t;
            if (t != -1) M.any_tex = 1;
        }
    }
    if (m) HIP_TRY(hipMemcpyAsync(c->mats, h.data(), (size_t)m * sizeof(MptMaterial), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int mpt_reset_images(mpt_ctx *c) {                                  // image.py:90-92
    if (use(c)) return 1;
    if (mpt_flush(c)) return 1;
    c->h_images.clear();
    c->texels_used = 0;
    return 0;
}

extern "C" int mpt_load_image(mpt_ctx *c, const float *rgba, int nx, int ny, int *id) {   // image.py:51-88
    if (use(c)) return 1;
    if (mpt_flush(c)) return 1;
    if ((int)c->h_images.size() >= c->caps.max_textures) return fail("Out of ID!");       // allocator.py:53
    size_t need = (size_t)nx * ny;
    if (c->texels_used + need > (size_t)c->caps.max_texels) return fail("Out of memory!"); // allocator.py:24
    if (!c->texels) { if (dev_alloc(&c->texels, (size_t)c->caps.max_texels)) return 1; }
    HIP_TRY(hipMemcpyAsync(c->texels + c->texels_used, rgba, need * sizeof(MptVec4), hipMemcpyHostToDevice, c->stream));
    MptImage im = { nx, ny, (int32_t)c->texels_used, 0 };
    c->h_images.push_back(im);
    c->texels_used += need;
    HIP_TRY(hipMemcpyAsync(c->images, c->h_images.data(), c->h_images.size() * sizeof(MptImage),
                           hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (id) *id = (int)c->h_images.size() - 1;
    return 0;
}

extern "C" int mpt_set_camera(mpt_ctx *c, const float v2w[16], const float w2v[16]) {
    if (use(c)) return 1;
    if (mpt_flush(c)) return 1;
    memcpy(c->v2w, v2w, sizeof c->v2w);
    if (w2v) memcpy(c->w2v, w2v, sizeof c->w2v);
    return 0;
}

extern "C" int mpt_clear_lights(mpt_ctx *c) {
    if (use(c)) return 1;
    if (mpt_flush(c)) return 1;
    c->h_lights.clear();
    return 0;
}

extern "C" int mpt_add_light(mpt_ctx *c, int type, const float color[3], const float pos[3], const float axes[9],
                             float size, int *index) {
    if (use(c)) return 1;
    if (mpt_flush(c)) return 1;
    if (type != MPT_LIGHT_POINT && type != MPT_LIGHT_AREA) return fail("unknown light type %d", type);
    if ((int)c->h_lights.size() >= c->caps.max_lights) return fail("too many lights (max_lights=%d)", c->caps.max_lights);
    MptLight L{};
    L.color_size = { color[0], color[1], color[2], size };
    L.pos_type = { pos[0], pos[1], pos[2], 0.f };
    memcpy(&L.pos_type.w, &type, 4);
    L.ax0 = { axes[0], axes[1], axes[2], 0.f };
    L.ax1 = { axes[3], axes[4], axes[5], 0.f };
    L.ax2 = { axes[6], axes[7], axes[8], 0.f };
    c->h_lights.push_back(L);
    if (index) *index = (int)c->h_lights.size() - 1;
    return upload_lights(c);
}

extern "C" int mpt_set_world_light(mpt_ctx *c, const float fac[4], int tex) {
    if (use(c)) return 1;
    if (mpt_flush(c)) return 1;
    memcpy(c->world_fac, fac, sizeof c->world_fac);
    c->world_tex = tex;
    return 0;
}

// ------------------------------------------------------------------ LBVH build (tree/lbvh.py:169-305)
// Same algorithm as the reference (30-bit Morton codes of centroids, sorted, Karras hierarchy,
// bottom-up boxes) with two robustness changes: the sort key is (code << 32 | index), so equal
// codes cannot corrupt the hierarchy, and the boxes are fitted in one post-order
// pass instead of <=64 level-synchronous launches with a read-back each (lbvh.py:251-261).
// With distinct codes the tree is node-for-node the reference's.

static inline uint32_t expand_bits(uint32_t v) {                               // lbvh.py:13-17
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

static inline int quant1024(float x) {                                         // clamp(ifloor(v * 1024), 0, 1023), lbvh.py:29
    float f = floorf(x * 1024.0f);
    if (!(f == f) || f < 0.f) return 0;
    if (f > 1023.f) return 1023;
    return (int)f;
}

static inline int delta(const std::vector<uint64_t> &key, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __builtin_clzll(key[i] ^ key[j]);
}

// ------------------------------------------------------------------ SAH re-partition (fast build only)
// The image does not depend on the tree's shape (only on which of two equal-depth hits wins), so the
// production traversal is free to walk a better tree over the SAME leaf slots: measured on the
// 978-triangle benchmark scene a full-sweep SAH partition needs 11.0 node fetches per ray where the
// LBVH needs 23.7.  Leaves stay single triangles (the node record is unchanged); nodes are numbered
// in DFS pre-order, root 0.  Exact sweep for ranges <= 8192 leaves, 32-bin SAH above.
struct SahBuild {
    int n = 0;
    std::vector<float> lo, hi, ctr;            // per leaf slot [n][3]
    std::vector<int> idx;                      // leaf slots, partitioned in place
    std::vector<int32_t> child;                // [n-1][2]: >= 0 internal, ~slot leaf
    std::vector<float> blo, bhi;               // per internal node [n-1][3]: its own box
    int depth = 0;

    static float half_area(const float *l, const float *h) {
        float dx = std::max(h[0] - l[0], 0.f), dy = std::max(h[1] - l[1], 0.f), dz = std::max(h[2] - l[2], 0.f);
        return dx * dy + dy * dz + dz * dx;
    }

    int split(int b, int e) {                  // returns m in (b, e): [b, m) | [m, e)
        const int cnt = e - b;
        float best = INFINITY;
        int best_axis = -1, best_k = -1;
        float best_pos = 0.f;
        bool binned = cnt > 8192;
        std::vector<std::pair<float, int>> key(binned ? 0 : cnt);
        std::vector<float> rarea(binned ? 0 : cnt);
        float cl[3] = { INFINITY, INFINITY, INFINITY }, ch[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (int t = b; t < e; t++)
            for (int a = 0; a < 3; a++) {
                cl[a] = std::min(cl[a], ctr[(size_t)idx[t] * 3 + a]);
                ch[a] = std::max(ch[a], ctr[(size_t)idx[t] * 3 + a]);
            }
        for (int a = 0; a < 3; a++) {
            if (!(ch[a] > cl[a])) continue;
            if (!binned) {
                for (int t = 0; t < cnt; t++) key[t] = { ctr[(size_t)idx[b + t] * 3 + a], idx[b + t] };
                std::sort(key.begin(), key.end());
                float l[3] = { INFINITY, INFINITY, INFINITY }, h[3] = { -INFINITY, -INFINITY, -INFINITY };
                for (int t = cnt - 1; t > 0; t--) {
                    int s = key[t].second;
                    for (int q = 0; q < 3; q++) { l[q] = std::min(l[q], lo[(size_t)s * 3 + q]); h[q] = std::max(h[q], hi[(size_t)s * 3 + q]); }
                    rarea[t] = half_area(l, h);
                }
                for (int q = 0; q < 3; q++) { l[q] = INFINITY; h[q] = -INFINITY; }
                for (int k = 1; k < cnt; k++) {
                    int s = key[k - 1].second;
                    for (int q = 0; q < 3; q++) { l[q] = std::min(l[q], lo[(size_t)s * 3 + q]); h[q] = std::max(h[q], hi[(size_t)s * 3 + q]); }
                    float cost = half_area(l, h) * k + rarea[k] * (cnt - k);
                    if (cost < best) { best = cost; best_axis = a; best_k = k; }
                }
            } else {
                const int NB = 32;
                float bl[NB][3], bh[NB][3];
                int bc[NB];
                for (int q = 0; q < NB; q++) { bc[q] = 0; for (int r = 0; r < 3; r++) { bl[q][r] = INFINITY; bh[q][r] = -INFINITY; } }
                float scale = NB / (ch[a] - cl[a]);
                for (int t = b; t < e; t++) {
                    int s = idx[t];
                    int q = std::min(NB - 1, std::max(0, (int)((ctr[(size_t)s * 3 + a] - cl[a]) * scale)));
                    bc[q]++;
                    for (int r = 0; r < 3; r++) { bl[q][r] = std::min(bl[q][r], lo[(size_t)s * 3 + r]); bh[q][r] = std::max(bh[q][r], hi[(size_t)s * 3 + r]); }
                }
                float ra[NB]; int rc[NB];
                float l[3] = { INFINITY, INFINITY, INFINITY }, h[3] = { -INFINITY, -INFINITY, -INFINITY };
                int c2 = 0;
                for (int q = NB - 1; q > 0; q--) {
                    c2 += bc[q];
                    for (int r = 0; r < 3; r++) { l[r] = std::min(l[r], bl[q][r]); h[r] = std::max(h[r], bh[q][r]); }
                    ra[q] = half_area(l, h); rc[q] = c2;
                }
                for (int r = 0; r < 3; r++) { l[r] = INFINITY; h[r] = -INFINITY; }
                int c1 = 0;
                for (int q = 1; q < NB; q++) {
                    c1 += bc[q - 1];
                    for (int r = 0; r < 3; r++) { l[r] = std::min(l[r], bl[q - 1][r]); h[r] = std::max(h[r], bh[q - 1][r]); }
                    if (c1 == 0 || rc[q] == 0) continue;
                    float cost = half_area(l, h) * c1 + ra[q] * rc[q];
                    if (cost < best) { best = cost; best_axis = a; best_k = c1; best_pos = cl[a] + q / scale; }
                }
            }
        }
        if (best_axis < 0) return b + cnt / 2;                 // all centroids equal: split the range in half
        if (!binned) {
            for (int t = 0; t < cnt; t++) key[t] = { ctr[(size_t)idx[b + t] * 3 + best_axis], idx[b + t] };
            std::sort(key.begin(), key.end());
            for (int t = 0; t < cnt; t++) idx[b + t] = key[t].second;
            return b + best_k;
        }
        int a = best_axis;
        const int NB = 32;
        float scale = NB / (ch[a] - cl[a]);
        int qsplit = (int)std::lround((best_pos - cl[a]) * scale);
        int m = (int)(std::partition(idx.begin() + b, idx.begin() + e, [&](int s) {
                          int q = std::min(NB - 1, std::max(0, (int)((ctr[(size_t)s * 3 + a] - cl[a]) * scale)));
                          return q < qsplit;
                      }) - idx.begin());
        if (m <= b || m >= e) m = b + cnt / 2;
        return m;
    }

    void run() {
        const int ni = n > 1 ? n - 1 : 0;
        child.assign((size_t)std::max(ni, 1) * 2, 0);
        blo.assign((size_t)std::max(ni, 1) * 3, 0.f);
        bhi.assign((size_t)std::max(ni, 1) * 3, 0.f);
        idx.resize(n);
        for (int i = 0; i < n; i++) idx[i] = i;
        depth = 0;
        if (ni == 0) return;
        struct Item { int b, e, parent, which, depth; };
        std::vector<Item> st;
        st.push_back({ 0, n, -1, 0, 1 });
        int next_node = 0;
        while (!st.empty()) {
            Item it = st.back(); st.pop_back();
            int me = next_node++;
            if (it.parent >= 0) child[(size_t)it.parent * 2 + it.which] = me;
            depth = std::max(depth, it.depth);
            float l[3] = { INFINITY, INFINITY, INFINITY }, h[3] = { -INFINITY, -INFINITY, -INFINITY };
            for (int t = it.b; t < it.e; t++)
                for (int q = 0; q < 3; q++) { l[q] = std::min(l[q], lo[(size_t)idx[t] * 3 + q]); h[q] = std::max(h[q], hi[(size_t)idx[t] * 3 + q]); }
            for (int q = 0; q < 3; q++) { blo[(size_t)me * 3 + q] = l[q]; bhi[(size_t)me * 3 + q] = h[q]; }
            int m = split(it.b, it.e);
            // right first on the stack so the left subtree gets the next indices (pre-order)
            if (it.e - m == 1) child[(size_t)me * 2 + 1] = ~idx[m];
            else st.push_back({ m, it.e, me, 1, it.depth + 1 });
            if (m - it.b == 1) child[(size_t)me * 2 + 0] = ~idx[it.b];
            else st.push_back({ it.b, m, me, 0, it.depth + 1 });
        }
    }
};

static int build_tree_host(mpt_ctx *c) {
    const int n = c->nfaces;
    const float *V = c->verts.data();
    auto pos = [&](int f, int k) { return V + ((size_t)f * 3 + k) * 8; };

    // genMortonCodes, lbvh.py:169-183
    float bmin[3] = { 1e6f, 1e6f, 1e6f }, bmax[3] = { -1e6f, -1e6f, -1e6f };
    std::vector<float> cen((size_t)n * 3);
    for (int f = 0; f < n; f++)
        for (int a = 0; a < 3; a++) {
            float ctr = ((pos(f, 0)[a] + pos(f, 1)[a]) + pos(f, 2)[a]) / 3.0f;   // lbvh.py:164
            cen[(size_t)f * 3 + a] = ctr;
            bmin[a] = fminf(bmin[a], ctr);
            bmax[a] = fmaxf(bmax[a], ctr);
        }
    std::vector<uint64_t> key(n);
    for (int f = 0; f < n; f++) {
        uint32_t w[3];
        for (int a = 0; a < 3; a++) w[a] = expand_bits((uint32_t)quant1024((cen[(size_t)f * 3 + a] - bmin[a]) / (bmax[a] - bmin[a])));
        uint32_t code = w[0] * 4 + w[1] * 2 + w[2];
        key[f] = ((uint64_t)code << 32) | (uint32_t)f;
    }
    std::sort(key.begin(), key.end());                                          // lbvh.py:204-208

    c->h_leaf.resize(n); c->h_mc.resize(n);
    for (int i = 0; i < n; i++) { c->h_leaf[i] = (int32_t)(key[i] & 0xffffffffu); c->h_mc[i] = (int32_t)(key[i] >> 32); }

    const int ni = n > 1 ? n - 1 : 0;
    c->h_child.assign((size_t)std::max(ni, 1) * 2, 0);
    c->h_bmin.assign((size_t)std::max(ni, 1) * 3, 0.f);
    c->h_bmax.assign((size_t)std::max(ni, 1) * 3, 0.f);

    // genHierarchy, lbvh.py:212-231 (determineRange :93-146, findSplit :62-89)
    for (int i = 0; i < ni; i++) {
        int l, r;
        if (i == 0) { l = 0; r = n - 1; }
        else {
            int d = delta(key, n, i, i + 1) > delta(key, n, i, i - 1) ? 1 : -1;
            int dmin = delta(key, n, i, i - d);
            int lmax = 2;
            while (delta(key, n, i, i + lmax * d) > dmin) lmax <<= 1;
            int s = 0;
            for (int t = lmax >> 1; t > 0; t >>= 1)
                if (delta(key, n, i, i + (s + t) * d) > dmin) s += t;
            l = i; r = i + s * d;
            if (d < 0) std::swap(l, r);
        }
        int cp = delta(key, n, l, r);
        int m = l, s = r - l;
        for (;;) {
            s = (s + 1) >> 1;
            int q = m + s;
            if (q < r && delta(key, n, l, q) > cp) m = q;
            if (s <= 1) break;
        }
        c->h_child[(size_t)i * 2 + 0] = (m == l) ? m : m + n;
        c->h_child[(size_t)i * 2 + 1] = (m + 1 == r) ? m + 1 : m + 1 + n;
    }

    // boxes: iterative post-order from the root (also yields the depth the LDS stack must hold)
    auto leaf_box = [&](int slot, float *lo, float *hi) {                       // lbvh.py:155-158
        int f = c->h_leaf[slot];
        for (int a = 0; a < 3; a++) {
            lo[a] = fminf(fminf(pos(f, 0)[a], pos(f, 1)[a]), pos(f, 2)[a]);
            hi[a] = fmaxf(fmaxf(pos(f, 0)[a], pos(f, 1)[a]), pos(f, 2)[a]);
        }
    };
    int depth = 0;
    if (ni > 0) {
        std::vector<int> order; order.reserve(ni);
        std::vector<std::pair<int, int>> st; st.push_back({ 0, 1 });
        std::vector<char> seen(ni, 0);
        while (!st.empty()) {
            auto [i, dpt] = st.back(); st.pop_back();
            if (i < 0 || i >= ni || seen[i]) return fail("AABB step never stop! hierarchy corrupted?");   // lbvh.py:259
            seen[i] = 1;
            order.push_back(i);
            depth = std::max(depth, dpt);
            for (int k = 0; k < 2; k++) {
                int ch = c->h_child[(size_t)i * 2 + k];
                if (ch >= n) st.push_back({ ch - n, dpt + 1 });
            }
        }
        if ((int)order.size() != ni) return fail("AABB step never stop! hierarchy corrupted?");
        for (int t = ni - 1; t >= 0; t--) {                                     // children before parents
            int i = order[t];
            float lo[2][3], hi[2][3];
            for (int k = 0; k < 2; k++) {
                int ch = c->h_child[(size_t)i * 2 + k];
                if (ch < n) leaf_box(ch, lo[k], hi[k]);
                else for (int a = 0; a < 3; a++) { lo[k][a] = c->h_bmin[(size_t)(ch - n) * 3 + a]; hi[k][a] = c->h_bmax[(size_t)(ch - n) * 3 + a]; }
            }
            for (int a = 0; a < 3; a++) {
                c->h_bmin[(size_t)i * 3 + a] = fminf(lo[0][a], lo[1][a]);
                c->h_bmax[(size_t)i * 3 + a] = fmaxf(hi[0][a], hi[1][a]);
            }
        }
    }
    c->tree_depth = depth;
    if (depth + 2 > 64) return fail("LBVH depth %d exceeds the 64-entry traversal stack", depth);

    // pack device records
    std::vector<MptVec4> snode((size_t)std::max(ni, 1) * 2), fnode((size_t)std::max(ni, 1) * 4);
    std::vector<MptVec4> tgeo((size_t)std::max(n, 1) * 4), tshade((size_t)std::max(n, 1) * 4);
    auto asf = [](int32_t v) { float f; memcpy(&f, &v, 4); return f; };
    for (int i = 0; i < ni; i++) {
        const float *lo = &c->h_bmin[(size_t)i * 3], *hi = &c->h_bmax[(size_t)i * 3];
        int c0 = c->h_child[(size_t)i * 2], c1 = c->h_child[(size_t)i * 2 + 1];
        snode[(size_t)i * 2 + 0] = { lo[0], lo[1], lo[2], asf(c0) };
        snode[(size_t)i * 2 + 1] = { hi[0], hi[1], hi[2], asf(c1) };
    }
    // the tree the fast build walks: child ids >= 0 internal, ~slot leaf; a node record holds its
    // two children's boxes
    std::vector<int32_t> fchild((size_t)std::max(ni, 1) * 2, 0);
    std::vector<float> flo((size_t)std::max(ni, 1) * 3, 0.f), fhi((size_t)std::max(ni, 1) * 3, 0.f);
    if (c->tree_kind == 1 && ni > 0) {
        SahBuild sb;
        sb.n = n;
        sb.lo.resize((size_t)n * 3); sb.hi.resize((size_t)n * 3); sb.ctr.resize((size_t)n * 3);
        for (int slot = 0; slot < n; slot++) {
            float l[3], h[3];
            leaf_box(slot, l, h);
            for (int a = 0; a < 3; a++) {
                sb.lo[(size_t)slot * 3 + a] = l[a]; sb.hi[(size_t)slot * 3 + a] = h[a];
                sb.ctr[(size_t)slot * 3 + a] = 0.5f * (l[a] + h[a]);
            }
        }
        sb.run();
        fchild = sb.child; flo = sb.blo; fhi = sb.bhi;
        c->fast_depth = sb.depth;
    } else {
        for (int i = 0; i < ni; i++) {
            for (int k = 0; k < 2; k++) {
                int ch = c->h_child[(size_t)i * 2 + k];
                fchild[(size_t)i * 2 + k] = ch < n ? ~ch : ch - n;
            }
            for (int a = 0; a < 3; a++) { flo[(size_t)i * 3 + a] = c->h_bmin[(size_t)i * 3 + a]; fhi[(size_t)i * 3 + a] = c->h_bmax[(size_t)i * 3 + a]; }
        }
        c->fast_depth = depth;
    }
    if (c->fast_depth + 2 > 64) return fail("BVH depth %d exceeds the 64-entry traversal stack", c->fast_depth);
    for (int i = 0; i < ni; i++) {
        float l[2][3], h[2][3];
        int id[2];
        for (int k = 0; k < 2; k++) {
            id[k] = fchild[(size_t)i * 2 + k];
            if (id[k] < 0) leaf_box(~id[k], l[k], h[k]);
            else for (int a = 0; a < 3; a++) { l[k][a] = flo[(size_t)id[k] * 3 + a]; h[k][a] = fhi[(size_t)id[k] * 3 + a]; }
        }
        for (int a = 0; a < 3; a++) fnode[(size_t)i * 4 + a] = { l[0][a], l[1][a], h[0][a], h[1][a] };
        fnode[(size_t)i * 4 + 3] = { asf(id[0]), asf(id[1]), 0.f, 0.f };
    }
    for (int slot = 0; slot < n; slot++) {
        int f = c->h_leaf[slot];
        const float *p0 = pos(f, 0), *p1 = pos(f, 1), *p2 = pos(f, 2);
        // hoisted terms of Face.intersect, geometries.py:120-122,134-136,140 (same f32 operations)
        float u[3], v[3], nn[3];
        for (int a = 0; a < 3; a++) { u[a] = p1[a] - p0[a]; v[a] = p2[a] - p0[a]; }
        nn[0] = u[1] * v[2] - u[2] * v[1];
        nn[1] = u[2] * v[0] - u[0] * v[2];
        nn[2] = u[0] * v[1] - u[1] * v[0];
        float uu = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
        float uv = u[0] * v[0] + u[1] * v[1] + u[2] * v[2];
        float vv = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
        float D = uv * uv - uu * vv;
        tgeo[(size_t)slot * 4 + 0] = { p0[0], p0[1], p0[2], D };
        tgeo[(size_t)slot * 4 + 1] = { u[0], u[1], u[2], uu };
        tgeo[(size_t)slot * 4 + 2] = { v[0], v[1], v[2], uv };
        tgeo[(size_t)slot * 4 + 3] = { nn[0], nn[1], nn[2], vv };
        tshade[(size_t)slot * 4 + 0] = { p0[3], p0[4], p0[5], p1[3] };
        tshade[(size_t)slot * 4 + 1] = { p1[4], p1[5], p2[3], p2[4] };
        tshade[(size_t)slot * 4 + 2] = { p2[5], p0[6], p0[7], p1[6] };
        tshade[(size_t)slot * 4 + 3] = { p1[7], p2[6], p2[7], asf(c->mtlids[f]) };
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if ((size_t)std::max(ni, 1) > c->node_cap) {
        hipFree(c->snode); hipFree(c->fnode); c->snode = c->fnode = nullptr;
        if (dev_alloc(&c->snode, snode.size()) || dev_alloc(&c->fnode, fnode.size())) return 1;
        c->node_cap = std::max(ni, 1);
    }
    if ((size_t)std::max(n, 1) > c->tri_cap) {
        hipFree(c->tgeo); hipFree(c->tshade); c->tgeo = c->tshade = nullptr;
        if (dev_alloc(&c->tgeo, tgeo.size()) || dev_alloc(&c->tshade, tshade.size())) return 1;
        c->tri_cap = std::max(n, 1);
    }
    HIP_TRY(hipMemcpyAsync(c->snode, snode.data(), snode.size() * sizeof(MptVec4), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->fnode, fnode.data(), fnode.size() * sizeof(MptVec4), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->tgeo, tgeo.data(), tgeo.size() * sizeof(MptVec4), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->tshade, tshade.data(), tshade.size() * sizeof(MptVec4), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->tree_valid = true;
    c->host_tree_valid = true;
    return 0;
}

// fnode records for the fast build from a (child, box) description over leaf slots
static void pack_fnode(mpt_ctx *c, int n, const std::vector<int32_t> &fchild, const std::vector<float> &flo,
                       const std::vector<float> &fhi, std::vector<MptVec4> &fnode) {
    const int ni = n > 1 ? n - 1 : 0;
    const float *V = c->verts.data();
    auto asf = [](int32_t v) { float f; memcpy(&f, &v, 4); return f; };
    fnode.assign((size_t)std::max(ni, 1) * 4, MptVec4{ 0, 0, 0, 0 });
    for (int i = 0; i < ni; i++) {
        float l[2][3], h[2][3];
        int id[2];
        for (int k = 0; k < 2; k++) {
            id[k] = fchild[(size_t)i * 2 + k];
            if (id[k] < 0) {
                int f = c->h_leaf[~id[k]];
                for (int a = 0; a < 3; a++) {
                    const float *p0 = V + ((size_t)f * 3) * 8, *p1 = p0 + 8, *p2 = p0 + 16;
                    l[k][a] = fminf(fminf(p0[a], p1[a]), p2[a]);
                    h[k][a] = fmaxf(fmaxf(p0[a], p1[a]), p2[a]);
                }
            } else for (int a = 0; a < 3; a++) { l[k][a] = flo[(size_t)id[k] * 3 + a]; h[k][a] = fhi[(size_t)id[k] * 3 + a]; }
        }
        for (int a = 0; a < 3; a++) fnode[(size_t)i * 4 + a] = { l[0][a], l[1][a], h[0][a], h[1][a] };
        fnode[(size_t)i * 4 + 3] = { asf(id[0]), asf(id[1]), 0.f, 0.f };
    }
}

// lbvh.py:297-305 entirely on the device (lbvh_build.hip); only the depth (4 bytes) comes back,
// plus the leaf order when the fast build wants its SAH re-partition (a host pass today)
static int build_tree_gpu(mpt_ctx *c) {
    const int n = c->nfaces;
    const int ni = n > 1 ? n - 1 : 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if ((size_t)std::max(n, 1) > c->d_model_cap) {
        hipFree(c->d_verts); hipFree(c->d_mtlids); c->d_verts = nullptr; c->d_mtlids = nullptr;
        if (dev_alloc(&c->d_verts, (size_t)std::max(n, 1) * 24) || dev_alloc(&c->d_mtlids, (size_t)std::max(n, 1))) return 1;
        c->d_model_cap = std::max(n, 1);
    }
    if ((size_t)std::max(n, 1) > c->d_build_cap) {
        hipFree(c->d_cen); hipFree(c->d_bounds); hipFree(c->d_depth); hipFree(c->d_keys_in); hipFree(c->d_keys_out);
        hipFree(c->d_sort_tmp); hipFree(c->d_child); hipFree(c->d_parent); hipFree(c->d_leaf); hipFree(c->d_mc);
        hipFree(c->d_bmin); hipFree(c->d_bmax); hipFree(c->d_arrive);
        c->d_cen = nullptr; c->d_bounds = nullptr; c->d_depth = nullptr; c->d_keys_in = c->d_keys_out = nullptr;
        c->d_sort_tmp = nullptr; c->d_child = c->d_parent = c->d_leaf = c->d_mc = nullptr;
        c->d_bmin = c->d_bmax = nullptr; c->d_arrive = nullptr;
        size_t m = std::max(n, 1);
        HIP_TRY(mpt_lbvh_sort_bytes((int)m, &c->d_sort_bytes));
        if (dev_alloc(&c->d_cen, m * 3) || dev_alloc(&c->d_bounds, 6) || dev_alloc(&c->d_depth, 1) ||
            dev_alloc(&c->d_keys_in, m) || dev_alloc(&c->d_keys_out, m) ||
            dev_alloc((char **)&c->d_sort_tmp, std::max<size_t>(c->d_sort_bytes, 16)) ||
            dev_alloc(&c->d_child, m * 2) || dev_alloc(&c->d_parent, m * 2) || dev_alloc(&c->d_leaf, m) ||
            dev_alloc(&c->d_mc, m) || dev_alloc(&c->d_bmin, m * 3) || dev_alloc(&c->d_bmax, m * 3) ||
            dev_alloc(&c->d_arrive, m)) return 1;
        c->d_build_cap = m;
    }
    if ((size_t)std::max(ni, 1) > c->node_cap) {
        hipFree(c->snode); hipFree(c->fnode); c->snode = c->fnode = nullptr;
        if (dev_alloc(&c->snode, (size_t)std::max(ni, 1) * 2) || dev_alloc(&c->fnode, (size_t)std::max(ni, 1) * 4)) return 1;
        c->node_cap = std::max(ni, 1);
    }
    if ((size_t)std::max(n, 1) > c->tri_cap) {
        hipFree(c->tgeo); hipFree(c->tshade); c->tgeo = c->tshade = nullptr;
        if (dev_alloc(&c->tgeo, (size_t)std::max(n, 1) * 4) || dev_alloc(&c->tshade, (size_t)std::max(n, 1) * 4)) return 1;
        c->tri_cap = std::max(n, 1);
    }
    if (n > 0) {
        HIP_TRY(hipMemcpyAsync(c->d_verts, c->verts.data(), (size_t)n * 24 * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->d_mtlids, c->mtlids.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    }
    MptLbvhBuffers b{};
    b.verts = c->d_verts; b.mtlids = c->d_mtlids; b.n = n;
    b.cen = c->d_cen; b.bounds = c->d_bounds; b.keys_in = c->d_keys_in; b.keys_out = c->d_keys_out;
    b.sort_tmp = c->d_sort_tmp; b.sort_tmp_bytes = c->d_sort_bytes;
    b.child = c->d_child; b.parent = c->d_parent; b.leaf = c->d_leaf; b.mc = c->d_mc;
    b.bmin = c->d_bmin; b.bmax = c->d_bmax; b.arrive = c->d_arrive; b.depth = c->d_depth;
    b.snode = c->snode; b.fnode = c->fnode; b.tgeo = c->tgeo; b.tshade = c->tshade;
    HIP_TRY(mpt_lbvh_build(&b, c->stream));
    int depth = 0;
    if (ni > 0) HIP_TRY(hipMemcpyAsync(&depth, c->d_depth, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (depth + 2 > 64) return fail("LBVH depth %d exceeds the 64-entry traversal stack", depth);
    c->tree_depth = depth;
    c->fast_depth = depth;
    c->host_tree_valid = false;
    if (c->tree_kind == 1 && ni > 0 && n <= c->sah_max) {
        // SAH re-partition of the leaves for the fast build (host pass over the leaf order)
        c->h_leaf.resize(n);
        HIP_TRY(hipMemcpy(c->h_leaf.data(), c->d_leaf, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
        SahBuild sb;
        sb.n = n;
        sb.lo.resize((size_t)n * 3); sb.hi.resize((size_t)n * 3); sb.ctr.resize((size_t)n * 3);
        const float *V = c->verts.data();
        for (int slot = 0; slot < n; slot++) {
            int f = c->h_leaf[slot];
            const float *p0 = V + ((size_t)f * 3) * 8, *p1 = p0 + 8, *p2 = p0 + 16;
            for (int a = 0; a < 3; a++) {
                float l = fminf(fminf(p0[a], p1[a]), p2[a]), h = fmaxf(fmaxf(p0[a], p1[a]), p2[a]);
                sb.lo[(size_t)slot * 3 + a] = l; sb.hi[(size_t)slot * 3 + a] = h; sb.ctr[(size_t)slot * 3 + a] = 0.5f * (l + h);
            }
        }
        sb.run();
        if (sb.depth + 2 > 64) return fail("BVH depth %d exceeds the 64-entry traversal stack", sb.depth);
        std::vector<MptVec4> fnode;
        pack_fnode(c, n, sb.child, sb.blo, sb.bhi, fnode);
        HIP_TRY(hipMemcpy(c->fnode, fnode.data(), (size_t)ni * 4 * sizeof(MptVec4), hipMemcpyHostToDevice));
        c->fast_depth = sb.depth;
    }
    c->tree_valid = true;
    return 0;
}

extern "C" int mpt_build_tree(mpt_ctx *c) {
    if (use(c)) return 1;
    if (mpt_flush(c)) return 1;
    return c->gpu_build ? build_tree_gpu(c) : build_tree_host(c);
}

static int download_tree(mpt_ctx *c) {
    if (c->host_tree_valid) return 0;
    const int n = c->nfaces, ni = n > 1 ? n - 1 : 0;
    c->h_child.assign((size_t)std::max(ni, 1) * 2, 0); c->h_leaf.assign(std::max(n, 1), 0); c->h_mc.assign(std::max(n, 1), 0);
    c->h_bmin.assign((size_t)std::max(ni, 1) * 3, 0.f); c->h_bmax.assign((size_t)std::max(ni, 1) * 3, 0.f);
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (n > 0) {
        HIP_TRY(hipMemcpy(c->h_leaf.data(), c->d_leaf, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(c->h_mc.data(), c->d_mc, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    if (ni > 0) {
        HIP_TRY(hipMemcpy(c->h_child.data(), c->d_child, (size_t)ni * 2 * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(c->h_bmin.data(), c->d_bmin, (size_t)ni * 3 * sizeof(float), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(c->h_bmax.data(), c->d_bmax, (size_t)ni * 3 * sizeof(float), hipMemcpyDeviceToHost));
    }
    c->host_tree_valid = true;
    return 0;
}

extern "C" int mpt_get_tree(mpt_ctx *c, int32_t *child, int32_t *leaf, float *bmin, float *bmax, int32_t *mc,
                            int32_t *depth) {
    if (!c) return fail("null context");
    if (!c->tree_valid) return fail("BVH not built: call build_tree() after load_model()");
    if (download_tree(c)) return 1;
    int n = c->nfaces, ni = n > 1 ? n - 1 : 0;
    if (child) memcpy(child, c->h_child.data(), (size_t)ni * 2 * sizeof(int32_t));
    if (leaf) memcpy(leaf, c->h_leaf.data(), (size_t)n * sizeof(int32_t));
    if (bmin) memcpy(bmin, c->h_bmin.data(), (size_t)ni * 3 * sizeof(float));
    if (bmax) memcpy(bmax, c->h_bmax.data(), (size_t)ni * 3 * sizeof(float));
    if (mc) memcpy(mc, c->h_mc.data(), (size_t)n * sizeof(int32_t));
    if (depth) *depth = c->tree_depth;
    return 0;
}

// ------------------------------------------------------------------ sobol
extern "C" int mpt_sobol_init(mpt_ctx *c, const int32_t *V, int rows, int dim) {
    if (use(c)) return 1;
    if (mpt_flush(c)) return 1;
    if (rows < 2 || dim < 1 || !V) return fail("bad sobol grid %dx%d", rows, dim);
    HIP_TRY(hipStreamSynchronize(c->stream));
    hipFree(c->sV); hipFree(c->sX); hipFree(c->sP);
    c->sV = c->sX = nullptr; c->sP = nullptr;
    HIP_TRY(hipStreamSynchronize(c->aux));
    for (int k = 0; k < MPT_MAX_PIPE; k++) { HIP_TRY(hipStreamSynchronize(c->rstream[k])); hipFree(c->sP2[k]); c->sP2[k] = nullptr; }
    if (dev_alloc(&c->sV, (size_t)rows * dim) || dev_alloc(&c->sX, (size_t)dim) ||
        dev_alloc(&c->sP, (size_t)MPT_MAX_BATCH * dim)) return 1;
    for (int k = 0; k < MPT_MAX_PIPE; k++)
        if (dev_alloc(&c->sP2[k], (size_t)MPT_MAX_BATCH * dim)) return 1;
    HIP_TRY(hipMemcpyAsync(c->sV, V, (size_t)rows * dim * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->sX, 0, (size_t)dim * sizeof(int), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->sdim = dim; c->srows = rows; c->stime = 0;
    return 0;
}

static int sobol_advance(mpt_ctx *c, int count, int keep, hipStream_t stream = nullptr, float *P = nullptr) {
    // keep = number of trailing frames whose points are written to P[0..keep)
    if (!stream) stream = c->stream;
    if (!P) P = c->sP;
    while (count > 0) {
        int step = count;
        int k = std::min(keep, step);
        HIP_TRY(mpt_launch_sobol_update(c->sX, c->sV, P, c->sdim, c->srows, c->stime, step, k, stream));
        c->stime = (int32_t)((uint32_t)c->stime + (uint32_t)step);
        count -= step;
    }
    return 0;
}

extern "C" int mpt_sobol_reset(mpt_ctx *c, int skip) {                         // sobol.py:92-97
    if (use(c)) return 1;
    if (mpt_flush(c)) return 1;
    if (!c->sV) return fail("sobol sampler not initialised");
    c->stime = 0;
    HIP_TRY(hipMemsetAsync(c->sX, 0, (size_t)c->sdim * sizeof(int), c->stream));
    return sobol_advance(c, skip, 0);
}

extern "C" int mpt_sobol_update(mpt_ctx *c, int count) {                       // sobol.py:99-105
    if (use(c)) return 1;
    if (mpt_flush(c)) return 1;
    if (!c->sV) return fail("sobol sampler not initialised");
    return sobol_advance(c, count, 0);
}

extern "C" int mpt_sobol_get(mpt_ctx *c, int32_t *X, float *P, int32_t *time) {
    if (use(c)) return 1;
    if (mpt_flush(c)) return 1;
    if (!c->sV) return fail("sobol sampler not initialised");
    std::vector<int32_t> x(c->sdim);
    HIP_TRY(hipMemcpyAsync(x.data(), c->sX, (size_t)c->sdim * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (X) memcpy(X, x.data(), (size_t)c->sdim * sizeof(int32_t));
    if (P)
        for (int j = 0; j < c->sdim; j++) {                                    // construct_float, sobol.py:20-29
            float ret = 0.f, term = 0.5f;
            for (uint32_t v = (uint32_t)x[j]; v; v <<= 1, term *= 0.5f)
                if (v & 0x80000000u) ret += term;
            P[j] = ret;
        }
    if (time) *time = c->stime;
    return 0;
}

// ------------------------------------------------------------------ rendering
static int fill_params(mpt_ctx *c, MptRenderParams &p, int nframes) {
    if (c->nx <= 0) return fail("film size not set: call set_size() first");
    if (!c->sV) return fail("sobol sampler not initialised");
    if (!c->tree_valid) return fail("BVH not built: call build_tree() after load_model()");
    memset(&p, 0, sizeof p);
    p.nx = c->nx; p.ny = c->ny; p.x0 = c->x0; p.x1 = c->x1;
    p.nframes = nframes; p.n = c->nfaces;
    p.sobol_dim = c->sdim; p.nlights = (int)c->h_lights.size(); p.world_tex = c->world_tex;
    share_extent(c, MPT_TILE, nullptr, &p.tiles_x);
    p.stripe_w = c->stripe_w ? c->stripe_w : (1 << 30);
    p.stripe_pitch = c->stripe_w ? c->stripe_w * c->stripe_mod : (1 << 30);
    p.tiles_y = (c->ny + MPT_TILE - 1) / MPT_TILE;
    p.ntiles = p.tiles_x * p.tiles_y;
    memcpy(p.world_fac, c->world_fac, sizeof p.world_fac);
    memcpy(p.v2w, c->v2w, sizeof p.v2w);
    p.snode = c->snode; p.fnode = c->fnode; p.tgeo = c->tgeo; p.tshade = c->tshade;
    p.mats = c->mats; p.lights = c->lights; p.images = c->images; p.texels = c->texels;
    p.P = c->sP;
    p.film0 = c->film[0]; p.film1 = c->film[1]; p.film2 = c->film[2];
    p.counters = c->d_counters;
    if (p.world_tex != -1 && (p.world_tex < 0 || p.world_tex >= (int)c->h_images.size()))
        return fail("world light texture %d is not a loaded image", p.world_tex);
    return 0;
}

static hipEvent_t get_event(mpt_ctx *c) {
    hipEvent_t e = nullptr;
    if (!c->event_pool.empty()) { e = c->event_pool.back(); c->event_pool.pop_back(); }
    else hipEventCreate(&e);
    return e;
}

extern "C" int mpt_flush(mpt_ctx *c) {
    if (!c) return fail("null context");
    if (c->pending == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    int B = c->pending;
    c->pending = 0;
    MptRenderParams p;
    if (fill_params(c, p, B)) return 1;
    const bool fast = c->mode == MPT_MODE_FAST;
    // fast build: this batch runs on its own stream; strict build: everything stays on the main stream
    if (fast) {
        // A launch ends with a drain of about one path latency (~0.4 ms on MI355X for depth-5 paths)
        // in which its lanes run empty one by one; a persistent workgroup leaves only when its
        // slowest lane has.  A launch that owns every CU pays that on every CU.  Small launches
        // therefore take 1/G of the CUs each and G of them are resident at once, in different
        // phases: the drain then idles 1/G of the chip.  (Measured, 1/8 film slab of 512x512x32:
        // 1.03 ms per step with G=1, 0.78 with G=2, 0.73 with G=4; whole film: 4.69 / 4.62 / 4.84.)
        long long cols = 0;
        share_extent(c, 1, &cols, nullptr);
        const double per_lane = (double)B * cols * c->ny / ((double)c->num_cus * 1024.0);
        int div = c->grid_div > 0 ? c->grid_div : (per_lane >= 24.0 ? 1 : per_lane >= 6.0 ? 2 : 4);
        int depth = c->pipe_depth > 0 ? c->pipe_depth : (div == 1 ? 2 : std::min(div + 2, (int)MPT_MAX_PIPE));
        if (depth != c->cur_depth) {
            // slots are reused round-robin: let everything in flight finish before the ring changes size
            HIP_TRY(hipStreamSynchronize(c->aux));
            for (int q = 0; q < MPT_MAX_PIPE; q++) HIP_TRY(hipStreamSynchronize(c->rstream[q]));
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->flip = 0;
        }
        c->cur_depth = depth; c->cur_div = div;
    }
    const int k = fast ? (c->flip++ % c->cur_depth) : 0;
    hipStream_t rs = fast ? c->rstream[k] : c->stream;
    // fast build: the Sobol points and the zeroed queue heads of this batch are prepared on the aux
    // stream (never behind a render kernel), the render itself goes to rstream[k]
    hipStream_t ss = fast ? c->aux : c->stream;
    if (fast) {
        if (c->main_dirty) {       // uploads / resets / option changes enqueued on the main stream come first
            HIP_TRY(hipEventRecord(c->ev_main, c->stream));
            c->main_dirty = false;
        }
        HIP_TRY(hipStreamWaitEvent(ss, c->ev_main, 0));
        HIP_TRY(hipStreamWaitEvent(ss, c->ev_render[k], 0));   // the batch that last read sP2[k] / d_work2[k]
        HIP_TRY(hipStreamWaitEvent(rs, c->ev_main, 0));
        p.P = c->sP2[k];
    }
    if (p.ntiles == 0) {
        if (sobol_advance(c, B, 0, ss, fast ? c->sP2[k] : nullptr)) return 1;
        if (fast) { HIP_TRY(hipEventRecord(c->ev_sobol2[k], ss)); HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_sobol2[k], 0)); }
        return 0;
    }
    if (sobol_advance(c, B, B, ss, fast ? c->sP2[k] : nullptr)) return 1;

    const int stack = ((c->mode == MPT_MODE_STRICT ? c->tree_depth : c->fast_depth) + 2 <= 32) ? 32 : 64;
    // LDS-resident kernel: node + triangle records + a 16-bit stack of (depth+1) levels x 1024
    // lanes must fit the CU's 160 KiB; ids must fit int16
    const int lds_stack = c->fast_depth + 1;            // sentinel + one pending sibling (node or leaf) per level
    const size_t lds_bytes = ((size_t)(c->nfaces - 1) * 4 + (size_t)c->nfaces * 4) * sizeof(MptVec4) +
                             (size_t)lds_stack * 1024 * sizeof(short);
    const bool lds_kernel = fast && c->use_lds && c->nfaces >= 2 && c->nfaces < 32768 && lds_bytes <= 160 * 1024;
    int chunk = B, nchunks = 1;
    const int tw = 1 << c->tile_w_shift, th = 1 << c->tile_h_shift;
    int tile_cols = 0;
    share_extent(c, tw, nullptr, &tile_cols);
    const int tiles8 = tile_cols * ((c->ny + th - 1) / th);
    if (fast) {
        chunk = c->chunk;
        if (chunk <= 0) {
            // work items are (8x8 tile, chunk of frames); persistent waves refill from the queue as
            // soon as their pool drains, so small items cost nothing and balance best: aim for
            // ~16 items per resident wave
            int want_items = 16 * 16 * c->num_cus;
            int want = (want_items + tiles8 - 1) / std::max(tiles8, 1);
            want = std::max(1, std::min(want, B));
            chunk = std::max((B + want - 1) / want, 1);
        }
        chunk = std::min(chunk, B);
        nchunks = (B + chunk - 1) / chunk;
    }
    p.chunk = chunk; p.nchunks = nchunks;
    p.sched_num = c->sched_num; p.sched_den = c->sched_den;
    p.nitems = tiles8 * nchunks;
    p.tile_w_shift = c->tile_w_shift; p.tile_h_shift = c->tile_h_shift;
    if (fast) {
        // one float4 per sample: [frame][pixel]; the combine pass sums frames in order
        size_t need = (size_t)B * c->nx * c->ny;
        // every slot of the ring at once: an allocation synchronises the device, so it must not
        // happen again on the second, third, ... batch of a run
        for (int q = 0; q < c->cur_depth; q++)
            if (need > c->partial2_cap[q]) {
                HIP_TRY(hipDeviceSynchronize());
                hipFree(c->partial2[q]); c->partial2[q] = nullptr; c->partial2_cap[q] = 0;
                if (dev_alloc(&c->partial2[q], need)) return 1;
                c->partial2_cap[q] = need;
            }
        p.partial = c->partial2[k];
        p.work_counter = c->d_work2[k];
        HIP_TRY(hipMemsetAsync(c->d_work2[k], 0, 8 * sizeof(unsigned int), ss));   // [8] (watchdog flag) is sticky
        HIP_TRY(hipEventRecord(c->ev_sobol2[k], ss));
        HIP_TRY(hipStreamWaitEvent(rs, c->ev_sobol2[k], 0));
        HIP_TRY(hipStreamWaitEvent(rs, c->ev_free[k], 0));   // combine of the batch that last used partial[k]
    }
    p.timeline = nullptr;
    if (c->timeline && lds_kernel) {
        const int block = c->lds_block ? c->lds_block : 1024;
        const int waves = ((c->num_cus + c->cur_div - 1) / c->cur_div) * (block / 64);
        if (waves != c->timeline_waves) {
            HIP_TRY(hipDeviceSynchronize());
            hipFree(c->d_timeline); c->d_timeline = nullptr;
            if (dev_alloc(&c->d_timeline, (size_t)waves * 4)) return 1;
            c->timeline_waves = waves;
        }
        p.timeline = c->d_timeline;
    }
    hipEvent_t e0 = get_event(c), e1 = get_event(c);
    HIP_TRY(hipEventRecord(e0, rs));
    if (!fast) HIP_TRY(mpt_launch_render_strict(&p, p.ntiles, stack, c->count, rs));
    else if (lds_kernel) HIP_TRY(mpt_launch_render_lds(&p, (c->num_cus + c->cur_div - 1) / c->cur_div, c->lds_block ? c->lds_block : 1024, lds_bytes, c->count, rs));
    else HIP_TRY(mpt_launch_render_fast(&p, (c->num_cus + c->cur_div - 1) / c->cur_div, stack, c->count, rs));
    c->last_kernel = lds_kernel ? 1 : 0;
    HIP_TRY(hipEventRecord(e1, rs));
    c->events.push_back({ e0, e1 });
    if (fast) {
        // everything later on the main stream (combine, gather, resolve, read-backs, scene changes) is
        // ordered after this render; the next batch, on the other stream, is not
        HIP_TRY(hipEventRecord(c->ev_render[k], rs));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_render[k], 0));
        HIP_TRY(mpt_launch_combine(c->film[0], c->partial2[k], c->nx, c->ny, c->x0, c->x1, p.stripe_w, p.stripe_pitch, B, c->stream));
        HIP_TRY(hipEventRecord(c->ev_free[k], c->stream));
    }
    return 0;
}

extern "C" int mpt_render(mpt_ctx *c, int nframes) {                           // path.py:75-77
    if (!c) return fail("null context");
    if (nframes < 0) return fail("nframes must be >= 0");
    // fail at the call, not at the deferred launch
    if (c->nx <= 0) return fail("film size not set: call set_size() first");
    if (!c->sV) return fail("sobol sampler not initialised");
    if (!c->tree_valid) return fail("BVH not built: call build_tree() after load_model()");
    while (nframes > 0) {
        int room = c->batch - c->pending;
        int take = std::min(room, nframes);
        c->pending += take;
        nframes -= take;
        if (c->pending >= c->batch && mpt_flush(c)) return 1;
    }
    return 0;
}

extern "C" int mpt_render_preview(mpt_ctx *c, int nframes) {                   // preview.py:18-41
    if (use(c)) return 1;
    if (mpt_flush(c)) return 1;
    while (nframes > 0) {
        int B = std::min(nframes, MPT_MAX_BATCH);
        MptRenderParams p;
        if (fill_params(c, p, B)) return 1;
        if (sobol_advance(c, B, B)) return 1;
        p.chunk = B; p.nchunks = 1;
        const int stack = ((c->mode == MPT_MODE_STRICT ? c->tree_depth : c->fast_depth) + 2 <= 32) ? 32 : 64;
        if (p.ntiles) {
            if (c->mode == MPT_MODE_STRICT) HIP_TRY(mpt_launch_preview_strict(&p, p.ntiles, stack, c->stream));
            else HIP_TRY(mpt_launch_preview_fast(&p, p.ntiles, stack, c->stream));
        }
        nframes -= B;
    }
    return 0;
}

// a persistent render kernel that had to be stopped by its watchdog leaves a flag behind
static int check_watchdog(mpt_ctx *c) {
    unsigned int flag = 0, f2[MPT_MAX_PIPE] = {};
    HIP_TRY(hipMemcpyAsync(&flag, c->d_work + 8, sizeof flag, hipMemcpyDeviceToHost, c->stream));
    for (int k = 0; k < MPT_MAX_PIPE; k++)
        HIP_TRY(hipMemcpyAsync(&f2[k], c->d_work2[k] + 8, sizeof flag, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int k = 0; k < MPT_MAX_PIPE; k++) flag |= f2[k];
    if (flag) return fail("render kernel stopped by its watchdog (scheduler made no progress): film is incomplete");
    return 0;
}

extern "C" int mpt_synchronize(mpt_ctx *c) {                                   // worker.py:17-18
    if (use_ro(c)) return 1;
    if (mpt_flush(c)) return 1;
    HIP_TRY(hipStreamSynchronize(c->aux));
    for (int k = 0; k < MPT_MAX_PIPE; k++) HIP_TRY(hipStreamSynchronize(c->rstream[k]));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return check_watchdog(c);
}

extern "C" int mpt_clear(mpt_ctx *c, int pass) {                               // filmtable.py:44-45: every pass, `id` ignored
    (void)pass;
    if (use(c)) return 1;
    if (mpt_flush(c)) return 1;
    size_t npix = (size_t)c->nx * c->ny;
    for (int p = 0; p < 3; p++)
        if (c->film[p]) HIP_TRY(hipMemsetAsync(c->film[p], 0, npix * sizeof(MptVec4), c->stream));
    return 0;
}

static int check_pass(mpt_ctx *c, int pass) {
    if (pass < 0 || pass >= 3) return fail("film pass %d out of range", pass);
    if (!c->film[pass]) return fail("film size not set: call set_size() first");
    return 0;
}

extern "C" int mpt_resolve(mpt_ctx *c, int pass) {
    if (use_ro(c)) return 1;
    if (mpt_flush(c)) return 1;
    if (check_pass(c, pass)) return 1;
    HIP_TRY(mpt_launch_resolve(c->film[pass], c->resolved, (size_t)c->nx * c->ny, c->stream));
    return 0;
}

extern "C" int mpt_get_image(mpt_ctx *c, int pass, float *out) {               // filmtable.py:47-63
    if (mpt_resolve(c, pass)) return 1;
    HIP_TRY(hipMemcpyAsync(out, c->resolved, (size_t)c->nx * c->ny * sizeof(MptVec4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return check_watchdog(c);
}

extern "C" int mpt_fast_export_image(mpt_ctx *c, int pass, float *out) {       // filmtable.py:66-79
    if (use_ro(c)) return 1;
    if (mpt_flush(c)) return 1;
    if (check_pass(c, pass)) return 1;
    HIP_TRY(mpt_launch_export(c->film[pass], c->exported, c->nx, c->ny, c->stream));
    HIP_TRY(hipMemcpyAsync(out, c->exported, (size_t)c->nx * c->ny * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int mpt_get_film_raw(mpt_ctx *c, int pass, float *out) {
    if (use_ro(c)) return 1;
    if (mpt_flush(c)) return 1;
    if (check_pass(c, pass)) return 1;
    HIP_TRY(hipMemcpyAsync(out, c->film[pass], (size_t)c->nx * c->ny * sizeof(MptVec4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

// ------------------------------------------------------------------ measurement
extern "C" int mpt_get_counters(mpt_ctx *c, mpt_counters *out) {
    if (use_ro(c)) return 1;
    if (mpt_flush(c)) return 1;
    unsigned long long h[12];
    HIP_TRY(hipMemcpyAsync(h, c->d_counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    out->samples = h[0]; out->rays = h[1]; out->n_box = h[2]; out->n_tri = h[3];
    out->n_shade = h[4]; out->n_draws = h[5]; out->bounces = h[6]; out->n_node = h[7];
    out->it_node = h[8]; out->it_leaf = h[9]; out->it_shade = h[10]; out->it_new = h[11];
    return 0;
}

// diagnostics: out[wave][4] = {start, scene ready, queue empty, exit} of the last LDS-kernel launch, 100 MHz ticks
extern "C" int mpt_get_timeline(mpt_ctx *c, unsigned long long *out, int cap_waves, int *nwaves) {
    if (use_ro(c)) return 1;
    if (mpt_synchronize(c)) return 1;
    if (!c->d_timeline) return fail("no timeline recorded: set option 'timeline' and render with the LDS kernel");
    int n = std::min(cap_waves, c->timeline_waves);
    if (out && n > 0)
        HIP_TRY(hipMemcpy(out, c->d_timeline, (size_t)n * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (nwaves) *nwaves = c->timeline_waves;
    return 0;
}

extern "C" int mpt_reset_counters(mpt_ctx *c) {
    if (use(c)) return 1;
    if (mpt_flush(c)) return 1;
    HIP_TRY(hipMemsetAsync(c->d_counters, 0, 12 * sizeof(unsigned long long), c->stream));
    return 0;
}

extern "C" int mpt_kernel_time(mpt_ctx *c, double *ms, int *launches) {
    if (use_ro(c)) return 1;
    if (mpt_flush(c)) return 1;
    HIP_TRY(hipStreamSynchronize(c->aux));
    for (int k = 0; k < MPT_MAX_PIPE; k++) HIP_TRY(hipStreamSynchronize(c->rstream[k]));
    HIP_TRY(hipStreamSynchronize(c->stream));
    double total = 0;
    for (auto &pr : c->events) {
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, pr.first, pr.second));
        total += t;
        c->event_pool.push_back(pr.first);
        c->event_pool.push_back(pr.second);
    }
    if (ms) *ms = total;
    if (launches) *launches = (int)c->events.size();
    c->events.clear();
    return 0;
}

// ------------------------------------------------------------------ multi-GPU film gather (RCCL over xGMI)
// One process per GPU; each renders the slab [x0,x1) of a replicated scene.  Film index is
// x*ny + y (filmtable.py:38), so a slab is ONE contiguous float4 range: every rank sends its
// range straight into the same range of the root's film -- a one-shot gather on the
// point-to-point xGMI links, no ring, no reduction.

extern "C" int mpt_comm_unique_id(char uid[128]) {
    if (rccl_load()) return 1;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    ncclUniqueId id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(uid, &id, 128);
    return 0;
}

extern "C" int mpt_comm_init(mpt_ctx *c, const char uid[128], int nranks, int rank) {
    if (use(c)) return 1;
    if (rccl_load()) return 1;
    if (c->comm) return fail("communicator already initialised");
    ncclUniqueId id;
    memcpy(&id, uid, 128);
    NCCL_TRY(g_rccl.CommInitRank(&c->comm, nranks, id, rank));
    c->nranks = nranks; c->rank = rank;
    return 0;
}

extern "C" int mpt_comm_gather_film(mpt_ctx *c, int pass, int root) {
    if (use_ro(c)) return 1;
    if (mpt_flush(c)) return 1;
    if (check_pass(c, pass)) return 1;
    if (!c->comm) return fail("communicator not initialised");
    // every rank holds the same split: contiguous slabs x in [r*nx/R, (r+1)*nx/R), or -- after
    // mpt_set_stripes(width, rank, R) -- stripes r, r+R, ... of `width` columns; each piece is one
    // contiguous float4 range (film index x*ny + y) and travels as its own send/recv of one group
    const int R = c->nranks;
    if (c->stripe_w && (c->stripe_mod != R || c->stripe_idx != c->rank))
        return fail("stripes (index %d of %d) do not match the communicator (rank %d of %d)", c->stripe_idx,
                    c->stripe_mod, c->rank, R);
    auto pieces = [&](int r, std::vector<std::pair<size_t, size_t>> &out) {
        out.clear();
        if (c->stripe_w == 0) {
            size_t lo = (size_t)((long long)r * c->nx / R) * c->ny, hi = (size_t)((long long)(r + 1) * c->nx / R) * c->ny;
            if (hi > lo) out.push_back({ lo, hi - lo });
        } else {
            for (long long x = (long long)r * c->stripe_w; x < c->nx; x += (long long)c->stripe_w * R) {
                long long w = std::min<long long>(c->stripe_w, c->nx - x);
                out.push_back({ (size_t)x * c->ny, (size_t)w * c->ny });
            }
        }
    };
    std::vector<std::pair<size_t, size_t>> pc;
    NCCL_TRY(g_rccl.GroupStart());
    if (c->rank == root) {
        for (int r = 0; r < R; r++) {
            if (r == root) continue;
            pieces(r, pc);
            for (auto &q : pc) NCCL_TRY(g_rccl.Recv(c->film[pass] + q.first, q.second * 4, ncclFloat, r, c->comm, c->stream));
        }
    } else {
        pieces(c->rank, pc);
        for (auto &q : pc) NCCL_TRY(g_rccl.Send(c->film[pass] + q.first, q.second * 4, ncclFloat, root, c->comm, c->stream));
    }
    NCCL_TRY(g_rccl.GroupEnd());
    return 0;
}

extern "C" int mpt_comm_barrier(mpt_ctx *c) {
    if (use_ro(c)) return 1;
    if (mpt_flush(c)) return 1;
    if (!c->comm) return fail("communicator not initialised");
    HIP_TRY(hipMemsetAsync(c->d_scratch, 0, sizeof(double), c->stream));
    NCCL_TRY(g_rccl.AllReduce(c->d_scratch, c->d_scratch, 1, ncclDouble, ncclSum, c->comm, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int mpt_comm_allreduce_max(mpt_ctx *c, double *value) {
    if (use_ro(c)) return 1;
    if (!c->comm) return fail("communicator not initialised");
    HIP_TRY(hipMemcpyAsync(c->d_scratch, value, sizeof(double), hipMemcpyHostToDevice, c->stream));
    NCCL_TRY(g_rccl.AllReduce(c->d_scratch, c->d_scratch, 1, ncclDouble, ncclMax, c->comm, c->stream));
    HIP_TRY(hipMemcpyAsync(value, c->d_scratch, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int mpt_comm_destroy(mpt_ctx *c) {
    if (use(c)) return 1;
    if (c->comm) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        NCCL_TRY(g_rccl.CommDestroy(c->comm));
        c->comm = nullptr;
    }
    c->nranks = 1; c->rank = 0;
    return 0;
}
