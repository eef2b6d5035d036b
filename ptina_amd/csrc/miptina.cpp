// miptina.cpp -- host runtime behind the C ABI of include/miptina.h.
//
// Owns the device state of one PTina "scene" (the singletons of ptina/things.py:20-28 collapsed
// into one context), uploads the scene, advances the Sobol sampler, batches enqueued frames into
// single launches and reads the film back.  The BVH builders are in tree_build.cpp, the RCCL film
// gather in comm.cpp.  No PyTorch, no Python: plain HIP runtime calls.

#include "miptina_ctx.h"
#include <chrono>

// Up to MPT_MAX_PIPE render streams, the main stream and the aux stream carry work at the same time.
// The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless told
// otherwise) and streams that share one are serialised, so ask for more before the runtime starts --
// eight of our own plus room for what RCCL creates -- unless the user has chosen a value.
// The runtime reads the variable when it initialises: this constructor runs at dlopen, which is early enough in
// every process that loads libmiptina before touching HIP (the Python package does).  Where HIP is already up --
// another HIP user in the process, or rocprofv3 --pmc, whose preloaded tool initialises the GPU first -- the
// request has no effect and streams fold onto 4 hardware queues: launches of a multi-GPU share then overlap
// less (G = 4 is slower than G = 1 there), results do not change.  Export GPU_MAX_HW_QUEUES
// yourself in such a process.
// Where the request comes too late it is SAID (stderr, once) and option "hw_queues" reads negative; bench.py hands
// the variable to its rocprofv3 passes itself, so that counters and timings come from the same queue configuration.
static int g_hwq = 0;            // GPU_MAX_HW_QUEUES after the constructor (0: unparsable)
static bool g_hwq_late = false;  // set here while a preloaded profiler tool may already have initialised HIP
__attribute__((constructor)) static void mpt_want_hw_queues() {
    if (!getenv("GPU_MAX_HW_QUEUES")) {
        setenv("GPU_MAX_HW_QUEUES", "12", 0);
        const char *pre = getenv("LD_PRELOAD");
        if ((pre && strstr(pre, "rocprof")) || getenv("ROCP_TOOL_LIBRARIES")) {
            g_hwq_late = true;
            fprintf(stderr, "libmiptina: GPU_MAX_HW_QUEUES=12 requested after a preloaded profiler tool may have initialised HIP: "
                            "streams may fold onto the default 4 hardware queues (results unchanged; export the variable "
                            "before starting the profiler)\n");
        }
    }
    const char *v = getenv("GPU_MAX_HW_QUEUES");
    g_hwq = v ? atoi(v) : 0;
}

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;

int fail(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}


extern "C" const char *mpt_last_error(void) { return g_err.c_str(); }
extern "C" int mpt_version(void) { return 100; }

extern "C" int mpt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int use_ro(mpt_ctx *c) {   // entry of calls that only read results
    if (!c) return fail("null context");
    HIP_TRY(hipSetDevice(c->device));
    return 0;
}

int use(mpt_ctx *c) {      // entry of calls that may change what the next render launch reads
    if (use_ro(c)) return 1;
    // frames enqueued so far render with the state they were enqueued under
    if (mpt_flush(c)) return 1;
    // Marked AFTER that flush: the flush records ev_main and clears the mark, and what this call is
    // about to enqueue on the main stream (a Sobol reset, an upload, a film clear) must be behind the
    // ev_main the NEXT batch's aux / render streams wait for.
    c->main_dirty = true;
    c->spec_valid = false;     // whatever this call changes, the points computed ahead are not trusted across it
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
    if (dev_alloc(&c->mats, (size_t)c->caps.max_materials + 1)) return bail("materials");   // + the default material's record
    if (dev_alloc(&c->images, c->caps.max_textures)) return bail("images");
    if (dev_alloc(&c->lights, MPT_MAX_LIGHTS)) return bail("lights");
    if (dev_alloc(&c->d_counters, MPT_COUNTER_WORDS)) return bail("counters");
    if (dev_alloc(&c->d_scratch, 2)) return bail("scratch");
    if (dev_alloc(&c->d_work, MPT_QUEUE_WORDS)) return bail("work counters");   // 8 queue heads, a cache line each
    for (int k = 0; k < MPT_MAX_PIPE; k++) {
        if (hipEventCreateWithFlags(&c->ev_sobol2[k], hipEventDisableTiming) != hipSuccess) return bail("event");
        if (hipEventCreateWithFlags(&c->ev_render[k], hipEventDisableTiming) != hipSuccess) return bail("event");
        if (hipEventCreateWithFlags(&c->ev_free[k], hipEventDisableTiming) != hipSuccess) return bail("event");
        if (dev_alloc(&c->d_work2[k], MPT_QUEUE_WORDS)) return bail("work counters");
        hipMemsetAsync(c->d_work2[k], 0, MPT_QUEUE_WORDS * sizeof(unsigned int), c->stream);
    }
    if (hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking) != hipSuccess) return bail("aux stream");
    if (hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming) != hipSuccess) return bail("event");
    if (hipEventCreateWithFlags(&c->ev_film, hipEventDisableTiming) != hipSuccess) return bail("event");
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
            if (prop.multiProcessorCount > 0) c->num_cus = prop.multiProcessorCount;
            c->clock_khz = prop.clockRate;
        }
    }
    if (make_render_streams(c)) return bail("render streams");
    {
        void *dp = nullptr;
        if (hipHostMalloc((void **)&c->h_watchdog, sizeof(unsigned int), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer(&dp, c->h_watchdog, 0) != hipSuccess) return bail("pinned watchdog flag");
        *c->h_watchdog = 0;
        c->d_watchdog = (unsigned int *)dp;
        void *dm = nullptr;
        if (hipHostMalloc((void **)&c->h_sahmeta, 32 * sizeof(int), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer(&dm, c->h_sahmeta, 0) != hipSuccess) return bail("pinned build mailbox");
        c->d_sahmeta = (int *)dm;
        memset(c->h_sahmeta, 0, 32 * sizeof(int));
    }
    hipMemsetAsync(c->d_counters, 0, MPT_COUNTER_WORDS * sizeof(unsigned long long), c->stream);
    hipMemsetAsync(c->d_work, 0, MPT_QUEUE_WORDS * sizeof(unsigned int), c->stream);
    {   // unset materials: factor 0 (field-zero, mtllib.py:12-13), texture -1 (deviation Q6)
        std::vector<MptMaterial> z((size_t)c->caps.max_materials + 1);
        for (auto &m : z) { memset(&m, 0, sizeof m); for (int k = 0; k < 12; k++) m.tex[k] = -1; }
        // record max_materials: the default material of mtllib.py:82-93 (mtlid -1)
        const float dflt[14] = { 0.8f, 0.8f, 0.8f, 0.0f, 0.4f, 0.5f, 0.4f, 0.0f, 0.0f, 0.4f, 0.0f, 0.5f, 0.0f, 1.45f };
        memcpy(z.back().p, dflt, sizeof dflt);
        hipMemcpyAsync(c->mats, z.data(), z.size() * sizeof(MptMaterial), hipMemcpyHostToDevice, c->stream);
        mpt_launch_derive_materials(c->mats, (int)z.size(), c->stream);
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
    mpt_comm_release(c);
    if (c->aux) hipStreamDestroy(c->aux);
    if (c->probe_stream) hipStreamDestroy(c->probe_stream);
    if (c->stress_stream) { hipStreamSynchronize(c->stress_stream); hipStreamDestroy(c->stress_stream); }
    hipFree(c->stress_buf);
    for (int k = 0; k < MPT_MAX_PIPE; k++) {
        if (c->rstream[k]) hipStreamDestroy(c->rstream[k]);
        if (c->ev_render[k]) hipEventDestroy(c->ev_render[k]);
        if (c->ev_free[k]) hipEventDestroy(c->ev_free[k]);
        if (c->ev_sobol2[k]) hipEventDestroy(c->ev_sobol2[k]);
        hipFree(c->partial2[k]); hipFree(c->sP2[k]); hipFree(c->d_work2[k]);
    }
    if (c->ev_main) hipEventDestroy(c->ev_main);
    if (c->ev_film) hipEventDestroy(c->ev_film);
    hipFree(c->onode); hipFree(c->tfast8); hipFree(c->tshade8); hipFree(c->d_perm8);
    for (auto &pr : c->events) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    for (auto &ev : c->event_pool) hipEventDestroy(ev);
    for (int p = 0; p < 3; p++) hipFree(c->film[p]);
    hipFree(c->resolved); hipFree(c->exported);
    hipFree(c->snode); hipFree(c->fnode); hipFree(c->tgeo); hipFree(c->tshade); hipFree(c->wnode); hipFree(c->qnode); hipFree(c->tfast); hipFree(c->fnode_soa);
    for (int k = 0; k < MPT_MAX_PIPE; k++) hipFree(c->stack_spill2[k]);
    hipFree(c->mats); hipFree(c->images); hipFree(c->texels); hipFree(c->lights);
    hipFree(c->sV); hipFree(c->sX); hipFree(c->sX_spec); hipFree(c->sP);
    hipFree(c->gather_buf); hipFree(c->d_pieces); hipFree(c->sah_ws);
    hipFree(c->wb_bin_of); hipFree(c->wb_ncount); hipFree(c->wb_offset); hipFree(c->wb_scan); hipFree(c->wb_area);
    hipFree(c->d_counters); hipFree(c->d_scratch); hipFree(c->d_work); hipFree(c->d_timeline);
    if (c->h_watchdog) hipHostFree(c->h_watchdog);
    if (c->h_sahmeta) hipHostFree(c->h_sahmeta);
    if (c->h_stage) hipHostFree(c->h_stage);
    hipFree(c->d_verts); hipFree(c->d_mtlids); hipFree(c->d_cen); hipFree(c->d_bounds); hipFree(c->d_depth);
    hipFree(c->d_keys_in); hipFree(c->d_keys_out); hipFree(c->d_sort_tmp);
    hipFree(c->d_child); hipFree(c->d_parent); hipFree(c->d_leaf); hipFree(c->d_mc);
    hipFree(c->d_bmin); hipFree(c->d_bmax); hipFree(c->d_arrive);
    hipStreamDestroy(c->stream);
    delete c;
}

// ------------------------------------------------------------------ page-locked host buffers
// Read-backs into page-locked memory are one DMA (4 MiB film: ~80 us); into pageable memory the runtime
// stages them through its own bounce buffers at a fraction of that rate.  mpt_host_alloc hands out
// page-locked buffers (FilmTable.get_image builds its arrays on them); buffers of any other origin go
// through one page-locked staging buffer of the context and a memcpy.
static std::mutex g_host_mu;
static std::map<uintptr_t, size_t> g_host_bufs;   // base -> bytes

extern "C" void *mpt_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, std::max<size_t>(bytes, 1), hipHostMallocDefault) != hipSuccess) {
        fail("hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_host_mu);
    g_host_bufs[(uintptr_t)p] = bytes;
    return p;
}

extern "C" void mpt_host_free(void *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        g_host_bufs.erase((uintptr_t)p);
    }
    hipHostFree(p);
}

static bool is_locked_range(const void *p, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_host_mu);
    auto it = g_host_bufs.upper_bound((uintptr_t)p);
    if (it == g_host_bufs.begin()) return false;
    --it;
    return (uintptr_t)p + bytes <= it->first + it->second;
}

// device -> caller buffer on the main stream, blocking
static int read_back(mpt_ctx *c, void *out, const void *dev, size_t bytes) {
    if (is_locked_range(out, bytes)) {
        HIP_TRY(hipMemcpyAsync(out, dev, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return 0;
    }
    if (bytes > c->h_stage_bytes) {
        if (c->h_stage) hipHostFree(c->h_stage);
        c->h_stage = nullptr; c->h_stage_bytes = 0;
        HIP_TRY(hipHostMalloc(&c->h_stage, bytes, hipHostMallocDefault));
        c->h_stage_bytes = bytes;
    }
    HIP_TRY(hipMemcpyAsync(c->h_stage, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(out, c->h_stage, bytes);
    return 0;
}

// ------------------------------------------------------------------ options
extern "C" int mpt_set_option(mpt_ctx *c, const char *key, int value) {
    if (use(c)) return 1;
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
    } else if (k == "lds_wide") {
        if (value != 0 && value != 1) return fail("lds_wide must be 0 or 1");
        c->lds_wide = value;
    } else if (k == "zero_copy") {
        c->zero_copy = value ? 1 : 0;
    } else if (k == "skip_dark") {
        if (value < -1 || value > 1) return fail("skip_dark must be -1 (auto), 0 or 1");
        c->skip_dark = value;
    } else if (k == "wide8") {
        if (value != 0 && value != 1) return fail("wide8 must be 0 or 1");
#if !MPT_WITH_OCT
        if (value) return fail("this library is built without the 8-wide octant-ordered kernel (an A/B build: make -C ptina_amd/csrc oct)");
#endif
        if (value != c->use_wide8) { c->use_wide8 = value; c->tree_valid = false; }    // (built by the next mpt_build_tree)
    } else if (k == "spin_us") {
        if (value < 0) return fail("spin_us must be >= 0");
        c->spin_us = value;
    } else if (k == "finalise") {
        if (value < 0 || value > 2) return fail("finalise must be 0 (combine pass), 1 (tail finalisation) or 2 (the same without the early image: A/B)");
        c->finalise = value;
    } else if (k == "launch_seq") {
        // test door: the number of the next launch minus one (its slab tag is 2 + number mod 65534), so that a test can walk
        // the launches across the point where the tags come round and every slab is zeroed
        if (value < 0) return fail("launch_seq must be >= 0");
        if (mpt_flush(c)) return 1;
        c->launch_seq = (unsigned)value;
        // the slabs may hold entries with any tag of the old numbering: an epoch no launch number maps to makes the next
        // finalising launch zero them all (round-5 ADVICE: a jump inside an epoch, or to one whose tags overlap, met stale entries)
        c->tag_epoch = 0xffffffffu;
    } else if (k == "pool") {
#if !MPT_WITH_POOL
        if (value) return fail("this library is built without the pooled LDS kernel (an A/B build: make -C ptina_amd/csrc pool)");
#endif
        c->use_pool = value ? 1 : 0;
    } else if (k == "pool_shaders") {
        if (value < 1 || value > 8) return fail("pool_shaders must be in 1..8");
        c->pool_shaders = value;
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
    } else if (k == "reserve_cus") {
        if (value < 0 || value >= c->num_cus) return fail("reserve_cus must be in 0..%d", c->num_cus - 1);
        c->reserve_cus = value;
    } else if (k == "node_soa") {
        c->node_soa = value ? 1 : 0;
    } else if (k == "wide") {
        c->use_wide = value ? 1 : 0;
    } else if (k == "sah_exact_max") {
        if (value < 2) return fail("sah_exact_max must be >= 2");
        c->sah_exact_max = value; c->tree_valid = false;
    } else if (k == "sah_build") {
        if (value < -1 || value > 1) return fail("sah_build must be -1 (auto), 0 (host) or 1 (device)");
        if (value != c->sah_build) { c->sah_build = value; c->tree_valid = false; }
    } else if (k == "wide_build") {
        if ((value ? 1 : 0) != c->wide_build) { c->wide_build = value ? 1 : 0; c->tree_valid = false; }
    } else if (k == "wide_quant") {
        c->use_quant = value ? 1 : 0;
    } else if (k == "tree") {
        if (value != 0 && value != 1) return fail("tree must be 0 (LBVH) or 1 (SAH)");
        if (value != c->tree_kind) { c->tree_kind = value; c->tree_valid = false; }
    } else if (k == "tile_w_shift" || k == "tile_h_shift") {
        if (value < 0 || value > 3) return fail("%s must be in 0..3", k.c_str());
        (k == "tile_w_shift" ? c->tile_w_shift : c->tile_h_shift) = value;
    } else if (k == "gpu_build") {
        if ((value ? 1 : 0) != c->gpu_build) { c->gpu_build = value ? 1 : 0; c->tree_valid = false; }
    } else if (k == "sah_inject_fail") {
        c->sah_inject_fail = value != 0; c->tree_valid = false;
    } else if (k == "build_phases") {
        c->build_phases = value ? 1 : 0;
    } else if (k == "lane_hist") {
        c->lane_hist = value ? 1 : 0;
    } else if (k == "sah_max") {
        c->sah_max = value; c->tree_valid = false;
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
    else if (k == "lds_wide") *value = c->lds_wide;
    else if (k == "lds_block") *value = c->lds_block;
    else if (k == "zero_copy") *value = c->zero_copy;
    else if (k == "wide8") *value = c->use_wide8;
    else if (k == "oct_nodes") *value = c->oct_nodes;
    else if (k == "oct_depth") *value = c->oct_depth;
    else if (k == "spin_us") *value = c->spin_us;
    else if (k == "finalise") *value = c->finalise;
    else if (k == "last_finalised") *value = c->last_finalised;
    else if (k == "launch_seq") *value = (int)(c->launch_seq & 0x7fffffffu);
    else if (k == "tag_wraps") *value = (int)c->tag_wraps;
    else if (k == "pool") *value = c->use_pool;
    else if (k == "skip_dark") *value = c->skip_dark;
    else if (k == "pool_shaders") *value = c->pool_shaders;
    else if (k == "pipe_depth") *value = c->pipe_depth;
    else if (k == "grid_div") *value = c->grid_div;
    else if (k == "cur_depth") *value = c->cur_depth;
    else if (k == "cur_div") *value = c->cur_div;
    else if (k == "last_div") *value = c->last_div;
    else if (k == "last_kernel") *value = c->last_kernel;
    else if (k == "num_cus") *value = c->num_cus;
    else if (k == "reserve_cus") *value = c->reserve_cus;
    else if (k == "wide") *value = c->use_wide;
    else if (k == "wide_quant") *value = c->use_quant;
    else if (k == "wide_build") *value = c->wide_build;
    else if (k == "sah_build") *value = c->sah_build;
    else if (k == "sah_fallback") *value = c->sah_fallback;
    else if (k == "build_phases") *value = c->build_phases;
    else if (k == "lane_hist") *value = c->lane_hist;
    else if (k == "sah_levels") *value = c->sah_stats.levels;
    else if (k == "sah_kelems") *value = (int)(c->sah_stats.elems / 1000);
    else if (k == "sah_chunks") *value = (int)c->sah_stats.chunks;
    else if (k == "sah_segments") *value = (int)c->sah_stats.segments;
    else if (k == "sah_part_kwords") *value = (int)(c->sah_stats.part_words / 1000);
    else if (k == "sah_tasks_small") *value = c->sah_stats.tasks_small;
    else if (k == "sah_tasks_big") *value = c->sah_stats.tasks_big;
    else if (k == "sah_t_sort_k") *value = c->sah_stats.t_sort_k;
    else if (k == "sah_t_loop_k") *value = c->sah_stats.t_loop_k;
    else if (k == "sah_t_max_k") *value = c->sah_stats.t_max_k;
    else if (k == "sah_task_levels") *value = c->sah_stats.task_levels;
    else if (k == "sah_task_levels_max") *value = c->sah_stats.task_levels_max;
    else if (k.rfind("build_phase_us_", 0) == 0 && k.size() == 16 && k[15] >= '0' && k[15] <= '5') *value = (int)(c->build_phase_us[k[15] - '0'] + 0.5);
    else if (k == "wide_nodes") *value = c->wide_nodes;
    else if (k == "wide_stack") *value = c->wide_stack;
    else if (k == "wide_ratio_permille") *value = (int)(c->wide_ratio * 1000.f + 0.5f);
    else if (k == "wide_depth") *value = c->wide_depth;
    else if (k == "nranks") *value = c->nranks;
    else if (k == "rank") *value = c->rank;
    else if (k == "device") *value = c->device;
    else if (k == "clock_khz") *value = c->clock_khz;
    else if (k == "hw_queues") *value = g_hwq_late ? -g_hwq : g_hwq;
    else return fail("unknown option '%s'", k.c_str());
    return 0;
}

// ------------------------------------------------------------------ film
extern "C" int mpt_set_size(mpt_ctx *c, int nx, int ny) {
    if (use(c)) return 1;
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
    c->film_version++;
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
    if (n < 0 || (n > 0 && !verts)) return fail("bad model arguments");
    if (n >= c->caps.max_faces) return fail("too many faces");                // model.py:84
    c->nfaces = n;
    c->verts.assign(verts, verts + (size_t)n * 24);
    if (mtlids) c->mtlids.assign(mtlids, mtlids + n);
    else c->mtlids.assign(n, -1);                                             // model.py:80-81
    c->max_mtlid = -1;
    for (int i = 0; i < n; i++) {
        if (c->mtlids[i] < -1 || c->mtlids[i] >= c->caps.max_materials)
            return fail("material id %d of face %d outside [-1, %d)", c->mtlids[i], i, c->caps.max_materials);
        c->max_mtlid = std::max(c->max_mtlid, (int)c->mtlids[i]);
    }
    {   // the bounding sphere of the model's box (MptRenderParams::t_scale)
        double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
        for (size_t v = 0; v < (size_t)n * 3; v++)
            for (int k = 0; k < 3; k++) {
                const double x = c->verts[v * 8 + k];
                if (x < lo[k]) lo[k] = x;
                if (x > hi[k]) hi[k] = x;
            }
        double r2 = 0;
        for (int k = 0; k < 3; k++) { c->scene_cen[k] = n ? 0.5 * (lo[k] + hi[k]) : 0.0; r2 += n ? 0.25 * (hi[k] - lo[k]) * (hi[k] - lo[k]) : 0.0; }
        c->scene_rad = std::sqrt(r2);
        if (!std::isfinite(c->scene_rad)) c->scene_rad = 1e30;
    }
    c->tree_valid = false;
    c->d_model_stale = true;
    return 0;
}

extern "C" int mpt_load_materials(mpt_ctx *c, const float *fac, const int32_t *tex, int m) {
    if (use(c)) return 1;
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
    HIP_TRY(mpt_launch_derive_materials(c->mats, m, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int mpt_reset_images(mpt_ctx *c) {                                  // image.py:90-92
    if (use(c)) return 1;
    c->h_images.clear();
    c->texels_used = 0;
    return 0;
}

extern "C" int mpt_load_image(mpt_ctx *c, const float *rgba, int nx, int ny, int *id) {   // image.py:51-88
    if (use(c)) return 1;
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
    memcpy(c->v2w, v2w, sizeof c->v2w);
    if (w2v) memcpy(c->w2v, w2v, sizeof c->w2v);
    return 0;
}

extern "C" int mpt_clear_lights(mpt_ctx *c) {
    if (use(c)) return 1;
    c->h_lights.clear();
    return 0;
}

extern "C" int mpt_add_light(mpt_ctx *c, int type, const float color[3], const float pos[3], const float axes[9],
                             float size, int *index) {
    if (use(c)) return 1;
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
    memcpy(c->world_fac, fac, sizeof c->world_fac);
    c->world_tex = tex;
    return 0;
}

// ------------------------------------------------------------------ sobol
extern "C" int mpt_sobol_init(mpt_ctx *c, const int32_t *V, int rows, int dim) {
    if (use(c)) return 1;
    if (rows < 2 || dim < 1 || !V) return fail("bad sobol grid %dx%d", rows, dim);
    HIP_TRY(hipStreamSynchronize(c->stream));
    hipFree(c->sV); hipFree(c->sX); hipFree(c->sX_spec); hipFree(c->sP);
    c->sV = c->sX = c->sX_spec = nullptr; c->sP = nullptr;
    HIP_TRY(hipStreamSynchronize(c->aux));
    for (int k = 0; k < MPT_MAX_PIPE; k++) { HIP_TRY(hipStreamSynchronize(c->rstream[k])); hipFree(c->sP2[k]); c->sP2[k] = nullptr; }
    if (dev_alloc(&c->sV, (size_t)rows * dim) || dev_alloc(&c->sX, (size_t)dim) || dev_alloc(&c->sX_spec, (size_t)dim) ||
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
        HIP_TRY(mpt_launch_sobol_update(c->sX, c->sX, c->sV, P, c->sdim, c->srows, c->stime, step, k, 1, stream));
        c->stime = (int32_t)((uint32_t)c->stime + (uint32_t)step);
        count -= step;
    }
    return 0;
}

extern "C" int mpt_sobol_reset(mpt_ctx *c, int skip) {                         // sobol.py:92-97
    if (use(c)) return 1;
    if (!c->sV) return fail("sobol sampler not initialised");
    c->stime = 0;
    HIP_TRY(hipMemsetAsync(c->sX, 0, (size_t)c->sdim * sizeof(int), c->stream));
    return sobol_advance(c, skip, 0);
}

extern "C" int mpt_sobol_update(mpt_ctx *c, int count) {                       // sobol.py:99-105
    if (use(c)) return 1;
    if (!c->sV) return fail("sobol sampler not initialised");
    return sobol_advance(c, count, 0);
}

extern "C" int mpt_sobol_get(mpt_ctx *c, int32_t *X, float *P, int32_t *time) {
    if (use(c)) return 1;
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
    // the production kernels address a frame's Sobol row and a triangle's shading record by 32-bit offsets from a scalar base, and
    // multiply frame x dimension in 24 bits (lane_draws, shade_rec_load)
    if ((long long)nframes * c->sdim >= (1ll << 30) || nframes >= (1 << 24) || c->sdim >= (1 << 24))
        return fail("%d frames of %d Sobol dimensions in one launch exceed the kernels' 32-bit row offsets", nframes, c->sdim);
    if (c->nfaces >= (1 << 26)) return fail("%d faces exceed the kernels' 32-bit record offsets", c->nfaces);
    memset(&p, 0, sizeof p);
    p.nx = c->nx; p.ny = c->ny; p.x0 = c->x0; p.x1 = c->x1;
    p.nframes = nframes; p.n = c->nfaces;
    p.sobol_inv_dim = 1.0f / (float)c->sdim;
    p.sobol_dim = c->sdim; p.nlights = (int)c->h_lights.size(); p.world_tex = c->world_tex;
    share_extent(c, MPT_TILE, nullptr, &p.tiles_x);
    p.stripe_w = c->stripe_w ? c->stripe_w : (1 << 30);
    p.stripe_pitch = c->stripe_w ? c->stripe_w * c->stripe_mod : (1 << 30);
    p.tiles_y = (c->ny + MPT_TILE - 1) / MPT_TILE;
    p.ntiles = p.tiles_x * p.tiles_y;
    memcpy(p.world_fac, c->world_fac, sizeof p.world_fac);
    memcpy(p.v2w, c->v2w, sizeof p.v2w);
    {   // t_scale: 2^-e with 2^e no less than the farthest a ray origin -- a point of the camera's near plane or of a surface -- can be
        // from any point of the scene's bounding sphere (mpt_load_model)
        double reach = c->nfaces > 0 ? 2.0 * c->scene_rad : 1.0;
        for (int corner = 0; corner < 4 && c->nfaces > 0; corner++) {
            const double x = (corner & 1) ? 1.0 : -1.0, y = (corner & 2) ? 1.0 : -1.0;
            const float *M = c->v2w;
            const double a3 = M[12] * x + M[13] * y - M[14] + M[15];
            double d2 = 0;
            for (int k = 0; k < 3; k++) {
                const double o = (M[4 * k] * x + M[4 * k + 1] * y - M[4 * k + 2] + M[4 * k + 3]) / a3;
                d2 += (o - c->scene_cen[k]) * (o - c->scene_cen[k]);
            }
            if (std::isfinite(d2)) reach = std::max(reach, std::sqrt(d2) + c->scene_rad);
        }
        int e = 0;
        std::frexp(std::min(std::max(reach * 1.001, 1e-30), 1e30), &e);       // reach * 1.001 <= 2^e
        p.t_scale = (float)std::ldexp(1.0, -e);
        p.t_unscale = (float)std::ldexp(1.0, e);
    }
    p.wnode = c->wnode; p.qnode = c->qnode; p.nwide = c->wide_nodes; p.onode = nullptr; p.stack_spill = nullptr;
    p.snode = c->snode; p.fnode = c->fnode; p.tgeo = c->tgeo; p.tshade = c->tshade; p.tfast = c->tfast;
    p.default_mtl = c->caps.max_materials;
    p.skip_dark = c->skip_dark >= 0 ? c->skip_dark : (c->mode == MPT_MODE_FAST ? 1 : 0);
    p.mats = c->mats; p.lights = c->lights; p.images = c->images; p.texels = c->texels;
    p.P = c->sP;
    p.film0 = c->film[0]; p.film1 = c->film[1]; p.film2 = c->film[2];
    p.counters = c->d_counters;
    p.lane_hist = c->lane_hist;
    p.watchdog = c->d_watchdog;
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
    // the points (and zeroed queue heads) of this batch may have been prepared when the previous one was launched
    const bool use_spec = fast && p.ntiles != 0 && c->spec_valid && c->spec_slot == k && c->spec_B == B &&
                          c->spec_time == c->stime;
    c->spec_valid = false;
    if (p.ntiles == 0) {
        if (sobol_advance(c, B, 0, ss, fast ? c->sP2[k] : nullptr)) return 1;
        if (fast) { HIP_TRY(hipEventRecord(c->ev_sobol2[k], ss)); HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_sobol2[k], 0)); }
        return 0;
    }
    if (use_spec) {
        // the points AND the state this batch leaves behind were computed when the previous batch was launched: the sampler moves by
        // a swap of two pointers -- no kernel at launch time (round 3 ran an X-only update here, whose 83 small workgroups took
        // CUs just when the persistent render workgroups wanted them: half the workgroups of a 1/8-share launch started 35 us late)
        std::swap(c->sX, c->sX_spec);
        c->stime = (int32_t)((uint32_t)c->stime + (uint32_t)B);
    } else if (sobol_advance(c, B, B, ss, fast ? c->sP2[k] : nullptr)) return 1;

    const int stack = ((c->mode == MPT_MODE_STRICT ? c->tree_depth : c->fast_depth) + 2 <= 32) ? 32 : 64;
    // LDS-resident kernel: node (64 B, MPT_LDS_NODE_STRIDE apart) + triangle (48 B) records + a 16-bit stack of (depth+1) levels x 1024
    // lanes must fit the CU's 160 KiB; ids must fit int16
    const int lds_stack = c->fast_depth + 1;            // sentinel + one pending sibling (node or leaf) per level
    // + the material records (96 B each, the default one last) and one byte per triangle naming its record
    const size_t lds_bytes = ((((size_t)(c->nfaces - 1) * MPT_LDS_NODE_STRIDE + 15) >> 4) + (size_t)c->nfaces * 3 + (size_t)(c->caps.max_materials + 1) * 6) * sizeof(MptVec4) +
                             (((size_t)c->nfaces + 15) & ~(size_t)15) + (size_t)lds_stack * 1024 * sizeof(short);
    // ... or the 4-wide nodes with exact boxes (option "lds_wide"): seven float4 of a wnode record MPT_LDS4_NODE_STRIDE apart, one
    // triangle record more (the unused slots' leaf), only the material records the model uses (+ the default one), and every
    // stack level the tree can ask for (a step leaves up to three entries behind)
    const int lds4_nmats = c->max_mtlid + 1;
    const size_t lds4_bytes = ((size_t)c->wide_nodes * (MPT_LDS4_NODE_STRIDE / 16) + ((size_t)c->nfaces + 1) * 3 + (size_t)(lds4_nmats + 1) * 6) * sizeof(MptVec4) +
                              (((size_t)c->nfaces + 15) & ~(size_t)15) + (size_t)c->wide_stack * 1024 * sizeof(short);
    const bool lds4_kernel = fast && c->use_lds && c->lds_wide && c->use_wide && !c->use_pool && c->wide_nodes > 0 && c->wide_stack > 0 && c->wnode &&
                             c->nfaces >= 2 && c->nfaces < 4095 && lds4_nmats < 255 && lds4_bytes <= 160 * 1024 &&
                             (size_t)c->wide_nodes * MPT_LDS4_NODE_STRIDE < 65536;   // (16-bit ids: a node's is its record's LDS address, a leaf's 16 * slot + 1)
    const bool lds_kernel = lds4_kernel ||
                            (fast && c->use_lds && c->nfaces >= 2 && c->nfaces < 32768 && c->caps.max_materials < 256 &&
                             lds_bytes <= 160 * 1024 &&
                             (size_t)(c->nfaces - 1) * (MPT_LDS_NODE_STRIDE / 8) < 32768);   // (the LDS copy's node ids are byte offsets / 8 in an int16 stack)
    // the same scene with the waves of the workgroup specialised and two path pools in LDS (render_pool.h): only the material
    // records the model uses, stacks for the tracer waves only, node records 72 bytes apart where that fits and 64 where not
    size_t pool_bytes = 0;
    int pool_stride = 0;
    const int pool_nmats = c->max_mtlid + 1;
#if MPT_WITH_POOL
    if (lds_kernel && c->use_pool && c->sdim <= 32768 && c->lds_block == 0) {
        for (int stride : { MPT_LDS_NODE_STRIDE, 64 }) {
            const size_t b = ((((size_t)(c->nfaces - 1) * stride + 15) >> 4) + (size_t)c->nfaces * 3 + (size_t)(pool_nmats + 1) * 6) * sizeof(MptVec4) +
                             (((size_t)c->nfaces + 15) & ~(size_t)15) + mpt_pool_lds_overhead() +
                             (size_t)lds_stack * (size_t)(16 - c->pool_shaders) * 64 * sizeof(short);
            if (b <= 160 * 1024) { pool_bytes = b; pool_stride = stride; break; }
        }
    }
#endif
    const bool pool_kernel = pool_bytes != 0;
    p.lds_node_stride = pool_stride; p.lds_nmats = pool_nmats; p.pool_shaders = c->pool_shaders;
    int chunk = B, nchunks = 1;
    const int tw = 1 << c->tile_w_shift, th = 1 << c->tile_h_shift;
    int tile_cols = 0;
    share_extent(c, tw, nullptr, &tile_cols);
    const int tiles8 = tile_cols * ((c->ny + th - 1) / th);
    if (fast) {
        chunk = c->chunk;
        if (chunk <= 0) {
            // Work items are (8x8 tile, chunk of frames).  Persistent waves refill from the queue the moment their
            // pool drains, so small items cost one returning atomic each and nothing else, and the launch ends
            // with what the last items hold: one frame of a tile (64 samples = one generation of paths per wave)
            // measured best on MI355X -- 512x512x32: 4.56 ms per launch against 4.67 (2 frames) and 5.07 (4);
            // 32-sample items (8x4 tiles) lose again (4.70) because a wave then starts half empty.
            chunk = 1;
        }
        chunk = std::min(chunk, B);
        nchunks = (B + chunk - 1) / chunk;
    }
    p.chunk = chunk; p.nchunks = nchunks;
    p.nitems = tiles8 * nchunks;
    p.tile_w_shift = c->tile_w_shift; p.tile_h_shift = c->tile_h_shift;
    if (fast) {
        // one float4 per sample: [frame][pixel]; the combine pass sums frames in order
        // ... of this context's share only: its columns packed side by side, padded to whole work-item tiles
        const int ccols = tile_cols * tw;
        p.partial_stride = ccols * c->ny;
        size_t need = (size_t)B * (size_t)ccols * c->ny;
        // every slot of the ring at once: an allocation synchronises the device, so it must not
        // happen again on the second, third, ... batch of a run
        bool zeroed = false;
        for (int q = 0; q < c->cur_depth; q++)
            if (need > c->partial2_cap[q]) {
                HIP_TRY(hipDeviceSynchronize());
                hipFree(c->partial2[q]); c->partial2[q] = nullptr; c->partial2_cap[q] = 0;
                if (dev_alloc(&c->partial2[q], need)) return 1;
                c->partial2_cap[q] = need;
                // the w of a slab entry is its ready flag for the tail finalisation (the launch's tag): fresh memory must not
                // hold one by accident -- and it does: the allocator hands back the slab of an earlier context, whose first
                // finalising launch used the very tag this context's first one will use
                HIP_TRY(hipMemsetAsync(c->partial2[q], 0, need * sizeof(MptVec4), c->stream));
                zeroed = true;
            }
        // ... and the zeros must be there before any launch of the ring writes or reads the slab: the render streams are not
        // ordered behind the stream that zeroes, and a memset may return before it is done.  (Seen in the full GPU test run as a
        // 2048 x 2048 film with single samples of an earlier test's render in it: its 2 GiB slabs take a millisecond to zero.)
        if (zeroed) HIP_TRY(hipStreamSynchronize(c->stream));
        p.partial = c->partial2[k];
        p.work_counter = c->d_work2[k];
        if (!use_spec) {
            HIP_TRY(hipMemsetAsync(c->d_work2[k], 0, MPT_QUEUE_WORDS * sizeof(unsigned int), ss));   // 8 queue heads + the finalisation's tile counter
            HIP_TRY(hipEventRecord(c->ev_sobol2[k], ss));
        }
        HIP_TRY(hipStreamWaitEvent(rs, c->ev_sobol2[k], 0));
        HIP_TRY(hipStreamWaitEvent(rs, c->ev_free[k], 0));   // combine of the batch that last used partial[k]
    }
    // Persistent workgroups hold their CUs for a whole launch, and the launches of the ring keep every CU taken
    // all the time: a foreign kernel -- RCCL's send/recv kernel of the film gather -- waits for a launch to end
    // (measured with mpt_probe_kernel on MI355X: median 0.13-0.47 ms, up to 2.2 ms beside a 1/8 share's launches,
    // 4.2 ms beside whole-film launches, against 12 us on an idle GPU).  Leaving CUs out of every launch does NOT
    // change that while launches overlap -- the workgroups of the next launch in the ring take any free CU at
    // once (same waits measured with 2 CUs left out) -- so "reserve_cus" defaults to 0 and the gather of batch i
    // simply completes one launch late; nothing waits for it but the read-back, and a step that ends with a
    // read-back (bench.py's `value`) has no other launch in flight when its gather starts.
    const int reserve = std::max(c->reserve_cus, 0);
    const int usable_cus = std::max(c->num_cus - reserve, 1);
    // A launch that finds the ring idle has no other launch to share the chip with -- a step that ends with a
    // read-back (bench.py's `value`, an interactive frame) is one launch at a time -- so it takes every CU; 1/G of
    // them would only make it G times longer.  Launches issued while others are still in flight take 1/G each.
    int launch_div = std::max(c->cur_div, 1);
    bool ring_idle = fast;
    if (fast && ((launch_div > 1 && c->grid_div <= 0) || c->finalise)) {
        for (int q = 0; q < c->cur_depth && ring_idle; q++) {
            hipError_t st = hipEventQuery(c->ev_render[q]);
            if (st == hipErrorNotReady) ring_idle = false;
            else if (st != hipSuccess) HIP_TRY(st);
        }
    }
    if (fast && launch_div > 1 && c->grid_div <= 0 && ring_idle) launch_div = 1;
    c->last_div = launch_div;
    const int launch_cus = std::max(usable_cus / launch_div, 1);   // G launches never claim more than usable_cus
    // scenes that do not fit LDS walk the 4-wide nodes (option "wide"; built by mpt_build_tree unless too deep)
    // A wide step costs ~2x the VALU instructions of a binary one (four slab tests and a sorting network) and makes
    // half the dependent fetches; with the planes picked by direction sign it wins on both big configurations
    // (MI355X: C4 963 -> 1135 Msamples/s, C5 494 -> 520), so it is the default wherever the collapse was built.
    const bool wide_pays = c->use_wide != 0;
    // ... or the 8-wide octant-ordered nodes (option "wide8", oct_build.cpp) with the triangle records in that tree's leaf order
    const bool oct_kernel = fast && !lds_kernel && c->use_wide8 && c->oct_nodes > 0;
    const bool wide_kernel = oct_kernel || (fast && !lds_kernel && wide_pays && c->wide_nodes > 0);
    int wide_blocks = 0;
    if (oct_kernel) { p.onode = c->onode; p.tfast = c->tfast8; p.tshade = c->tshade8; }
    if (wide_kernel) {
#if MPT_WITH_OCT
        if (oct_kernel) HIP_TRY(mpt_oct_blocks(launch_cus, c->count, &wide_blocks));
        else
#endif
        HIP_TRY(mpt_wide_blocks(launch_cus, c->count, c->use_quant, &wide_blocks));
        const size_t need_spill = (size_t)wide_blocks * MPT_BLOCK * 128;   // >= SpillStack::SPILL entries per lane (128 - LDS levels)
        // every slot of the ring at once (an allocation synchronises the device), and one strip PER SLOT: the launches of
        // different slots overlap, and a strip is indexed by block and lane only
        for (int q = 0; q < c->cur_depth; q++)
            if (need_spill > c->stack_spill2_cap[q]) {
                HIP_TRY(hipDeviceSynchronize());
                hipFree(c->stack_spill2[q]); c->stack_spill2[q] = nullptr; c->stack_spill2_cap[q] = 0;
                if (dev_alloc(&c->stack_spill2[q], need_spill)) return 1;
                c->stack_spill2_cap[q] = need_spill;
            }
        p.stack_spill = c->stack_spill2[k];
    }
    if (fast && !lds_kernel && !wide_kernel && c->node_soa && c->nfaces >= 2) {
        // layout A/B: the same records as four arrays (one 16-B gather per array instead of one 64-B record)
        const size_t ni = (size_t)c->nfaces - 1;
        if (!c->fnode_soa_valid) {
            if (ni > c->fnode_soa_cap) {
                HIP_TRY(hipDeviceSynchronize());
                hipFree(c->fnode_soa); c->fnode_soa = nullptr; c->fnode_soa_cap = 0;
                if (dev_alloc(&c->fnode_soa, ni * 4)) return 1;
                c->fnode_soa_cap = ni;
            }
            HIP_TRY(mpt_launch_transpose_nodes(c->fnode, c->fnode_soa, (int)ni, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->fnode_soa_valid = true;
        }
        p.fnode = c->fnode_soa; p.fnode_soa_n = (int)ni;
    }
    // Lanes per persistent workgroup of the LDS-resident kernel.  A launch that gives a lane fewer than six samples (an eighth
    // of the benchmark film: four) is over before a steady state forms: its length is a few path latencies, and a path is
    // faster with three waves per SIMD than with four (MI355X, 1/8 share: 0.78 ms at 768 lanes against 0.84 at 1024; from a
    // quarter share upwards 1024 is as good or better).  Same film either way.
    // (The kernel over the 4-wide nodes is fastest with 1024 lanes at every size: solo launch of a 1/8 share 0.49 ms against 0.53
    // at 768, 0.64 at 512; whole film 2.59 / 2.97 / 3.94.)
    int lds_block_used = c->lds_block ? c->lds_block : 1024;
    if (lds_kernel && !lds4_kernel && !pool_kernel && c->lds_block == 0 &&
        ((long long)p.nitems * chunk << (c->tile_w_shift + c->tile_h_shift)) < 6ll * launch_cus * 1024) lds_block_used = 768;
    p.timeline = nullptr;
    if (c->timeline && lds_kernel) {
        const int block = lds_block_used;
        const int waves = launch_cus * (block / 64);
        if (waves != c->timeline_waves) {
            HIP_TRY(hipDeviceSynchronize());
            hipFree(c->d_timeline); c->d_timeline = nullptr;
            if (dev_alloc(&c->d_timeline, (size_t)waves * MPT_TIMELINE_WORDS)) return 1;
            HIP_TRY(hipMemsetAsync(c->d_timeline, 0, (size_t)waves * MPT_TIMELINE_WORDS * sizeof(unsigned long long), c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->timeline_waves = waves;
        }
        p.timeline = c->d_timeline;
    }
    // Tail finalisation: a launch that has the chip to itself -- a step that ends with a read-back, an interactive frame -- adds
    // its frames to the film, resolves and writes the image out by itself, in the shadow of its own drain (finalise_tiles).  A
    // launch issued while another is in flight keeps the combine pass on the main stream: two launches must not add to the same
    // film pixels at the same time, and the main stream is what orders them.
    const bool fin = fast && c->finalise && ring_idle && !pool_kernel && p.nitems > 0;
    c->launch_seq++;
    p.fin_counter = nullptr; p.image_out = nullptr; p.slab_tag = MPT_TAG_COMBINE;
    c->film_version++;                           // this batch changes pass 0, whoever adds it
    if (fin) {
        // the 16-bit tag both halves of every sample entry of this launch carry (film_ops.h): never 0 (fresh memory) or 1 (a launch
        // that keeps the combine pass).  When the tags come round again no entry of any slab may still carry an old one: every slab
        // is zeroed -- behind whatever still reads one (the ring is idle here, so only a combine pass of an earlier batch on the main
        // stream can be pending: round-4 ADVICE), and in front of this launch AND of every later one: the next pipelined launch runs
        // on another stream and writes another slab, which nothing would order behind that slab's memset, so the host waits for the
        // memsets here (round-5 ADVICE; a wrap comes once in 65 534 launches)
        const unsigned phase = c->launch_seq % (unsigned)MPT_TAG_PERIOD, epoch = c->launch_seq / (unsigned)MPT_TAG_PERIOD;
        if (epoch != c->tag_epoch) {             // (told by the epoch, not by phase == 0: the launch with that number may have kept the combine pass)
            c->tag_epoch = epoch;
            HIP_TRY(hipStreamSynchronize(c->stream));
            for (int q = 0; q < MPT_MAX_PIPE; q++)
                if (c->partial2[q]) HIP_TRY(hipMemsetAsync(c->partial2[q], 0, c->partial2_cap[q] * sizeof(MptVec4), rs));
            HIP_TRY(hipStreamSynchronize(rs));
            c->tag_wraps++;
        }
        p.fin_counter = c->d_work2[k] + 8 * MPT_QUEUE_STRIDE;
        p.slab_tag = (unsigned)MPT_TAG_FIRST + phase;
        // everything the main stream still has to do to the film (an earlier batch's combine, a clear, a gather) comes first --
        // nothing, when the stream is idle (a step that ended with a read-back): then the event pair (8 us of API calls in front of
        // the launch, HIP trace of bench.py) is left out
        const hipError_t main_busy = hipStreamQuery(c->stream);
        if (main_busy == hipErrorNotReady) {
            HIP_TRY(hipEventRecord(c->ev_film, c->stream));
            HIP_TRY(hipStreamWaitEvent(rs, c->ev_film, 0));
        } else if (main_busy != hipSuccess) HIP_TRY(main_busy);
        // the resolved image as well, if the caller has said where get_image() will want it and this share is the whole film
        const size_t img_bytes = (size_t)c->nx * c->ny * sizeof(MptVec4);
        void *mapped = nullptr;
        if (c->hint_image && c->finalise == 1 && c->zero_copy && c->x0 == 0 && c->x1 == c->nx && c->stripe_w == 0 && is_locked_range(c->hint_image, img_bytes) &&
            hipHostGetDevicePointer(&mapped, c->hint_image, 0) == hipSuccess && mapped) {
            p.image_out = (MptVec4 *)mapped;
            c->early_ptr = c->hint_image; c->early_version = c->film_version; c->early_stream = rs;
        }
    }
    c->last_finalised = fin ? 1 : 0;
    hipEvent_t e0 = get_event(c), e1 = get_event(c);
    HIP_TRY(hipEventRecord(e0, rs));
    if (!fast) HIP_TRY(mpt_launch_render_strict(&p, p.ntiles, stack, c->count, rs));
#if MPT_WITH_POOL
    else if (pool_kernel) HIP_TRY(mpt_launch_render_pool(&p, launch_cus, 1024, pool_bytes, c->count, rs));
#endif
    else if (lds4_kernel) HIP_TRY(mpt_launch_render_lds4(&p, launch_cus, lds_block_used, lds4_bytes, c->count, rs));
    else if (lds_kernel) HIP_TRY(mpt_launch_render_lds(&p, launch_cus, lds_block_used, lds_bytes, c->count, rs));
#if MPT_WITH_OCT
    else if (oct_kernel) HIP_TRY(mpt_launch_render_oct(&p, wide_blocks, c->count, rs));
#endif
    else if (wide_kernel) HIP_TRY(mpt_launch_render_wide(&p, wide_blocks, c->count, c->use_quant, rs));
    else HIP_TRY(mpt_launch_render_fast(&p, launch_cus, stack, c->count, rs));
    c->last_kernel = pool_kernel ? 3 : lds4_kernel ? 5 : lds_kernel ? 1 : oct_kernel ? 4 : wide_kernel ? 2 : 0;
    HIP_TRY(hipEventRecord(e1, rs));
    c->events.push_back({ e0, e1 });
    if (c->events.size() > 4096) {
        // nobody has asked for kernel times for a long while (an interactive session renders frame after
        // frame): keep the newest half.  The dropped pairs were recorded at least 2048 launches ago.
        for (size_t q = 0; q < 2048; q++) { c->event_pool.push_back(c->events[q].first); c->event_pool.push_back(c->events[q].second); }
        c->events.erase(c->events.begin(), c->events.begin() + 2048);
    }
    if (fast) {
        // everything later on the main stream (combine, gather, resolve, read-backs, scene changes) is
        // ordered after this render; the next batch, on the other stream, is not
        HIP_TRY(hipEventRecord(c->ev_render[k], rs));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_render[k], 0));
        if (!fin)
            HIP_TRY(mpt_launch_combine(c->film[0], c->partial2[k], c->ny, c->x0, c->x1, p.stripe_w, p.stripe_pitch,
                                       p.partial_stride / std::max(c->ny, 1), B, c->stream));
        HIP_TRY(hipEventRecord(c->ev_free[k], c->stream));      // (a finalising launch has consumed its slab itself)
        // Ahead of time, on the aux stream: the Sobol points and zeroed queue heads of the NEXT batch, assuming it
        // has as many frames as this one and uses the next ring slot (the sampler's future is deterministic; its
        // state X moves only when that batch is really launched).  Takes 25 us of Sobol kernel + memset off the
        // head of every step.
        const int k2 = c->flip % c->cur_depth;
        if (k2 != k) {
            HIP_TRY(hipStreamWaitEvent(ss, c->ev_render[k2], 0));   // the batch that last read sP2[k2] / d_work2[k2]
            HIP_TRY(mpt_launch_sobol_update(c->sX, c->sX_spec, c->sV, c->sP2[k2], c->sdim, c->srows, c->stime, B, B, 1, ss));
            HIP_TRY(hipMemsetAsync(c->d_work2[k2], 0, MPT_QUEUE_WORDS * sizeof(unsigned int), ss));
            HIP_TRY(hipEventRecord(c->ev_sobol2[k2], ss));
            c->spec_valid = true; c->spec_slot = k2; c->spec_B = B; c->spec_time = c->stime;
        }
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

// a persistent render kernel that had to be stopped by its watchdog leaves a flag behind (in host-pinned
// memory: no copy to read it); call after the streams have been synchronised
int check_watchdog(mpt_ctx *c) {
    // read and re-arm: the failure is reported once, by the first read-back after it (the film holds
    // incomplete sums until the caller clears it)
    if (__atomic_exchange_n(c->h_watchdog, 0u, __ATOMIC_ACQ_REL))
        return fail("render kernel stopped by its watchdog (scheduler made no progress): film is incomplete");
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
    size_t npix = (size_t)c->nx * c->ny;
    for (int p = 0; p < 3; p++)
        if (c->film[p]) HIP_TRY(hipMemsetAsync(c->film[p], 0, npix * sizeof(MptVec4), c->stream));
    c->film_version++;
    return 0;
}

int check_pass(mpt_ctx *c, int pass) {
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

extern "C" int mpt_hint_image(mpt_ctx *c, int pass, float *out) {
    if (use_ro(c)) return 1;
    if (pass != 0) return 0;                   // only the path pass is finalised inside the render launch
    if (c->hint_image == out) return 0;
    // the old array may be the one a launch in flight is writing its image into: the caller is free to let go of it after this call
    if (c->early_ptr && c->early_ptr == c->hint_image) HIP_TRY(hipStreamSynchronize(c->stream));
    c->hint_image = out;
    c->early_ptr = nullptr;
    return 0;
}

extern "C" int mpt_get_image(mpt_ctx *c, int pass, float *out) {               // filmtable.py:47-63
    if (use_ro(c)) return 1;                   // validates the handle and makes the context's device current (round-3 ADVICE)
    const size_t bytes = (size_t)c->nx * c->ny * sizeof(MptVec4);
    void *mapped = nullptr;
    if (pass == 0) {
        // The launch that rendered the frames may have written this very image already (tail finalisation, mpt_hint_image): then
        // there is nothing left to do but wait for it.  The hint is spent either way: the array is the caller's from here on.
        if (mpt_flush(c)) return 1;
        const bool early = out && c->early_ptr == out && c->early_version == c->film_version;
        if (c->hint_image == out || c->early_ptr == out) {
            if (!early && c->early_ptr == out) HIP_TRY(hipStreamSynchronize(c->stream));    // (a stale image still being written)
            c->hint_image = nullptr; c->early_ptr = nullptr;
        }
        if (early) {
            if (check_pass(c, pass)) return 1;
            // the launch's own stream: one completion signal (the main stream, which waits for the same launch before anything else
            // it is given, would add a cross-stream hop in front of the host's wake-up)
            // ... and polled, for a while, instead of slept on: the runtime's blocking wait returned 18 us after the kernel's end
            // (HIP trace), a query loop sees the completion signal within a few; a launch that runs longer than the poll window
            // falls back to the blocking wait (option "spin_us", 0 = always block)
            if (c->spin_us > 0) {
                const auto t0 = std::chrono::steady_clock::now();
                for (;;) {
                    const hipError_t q = hipStreamQuery(c->early_stream);
                    if (q == hipSuccess) return check_watchdog(c);
                    if (q != hipErrorNotReady) HIP_TRY(q);
                    if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > (double)c->spin_us) break;
                }
            }
            HIP_TRY(hipStreamSynchronize(c->early_stream));
            return check_watchdog(c);
        }
    }
    if (c->zero_copy && is_locked_range(out, bytes) && hipHostGetDevicePointer(&mapped, out, 0) == hipSuccess && mapped) {
        // the caller's array is page-locked memory of ours: the resolve pass writes the image straight into it over PCIe,
        // instead of into a device buffer that a DMA then copies (one dependent hop and the copy engine's start-up less)
        if (mpt_flush(c)) return 1;
        if (check_pass(c, pass)) return 1;
        HIP_TRY(mpt_launch_resolve(c->film[pass], (MptVec4 *)mapped, (size_t)c->nx * c->ny, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return check_watchdog(c);
    }
    if (mpt_resolve(c, pass)) return 1;
    if (read_back(c, out, c->resolved, bytes)) return 1;
    return check_watchdog(c);
}

extern "C" int mpt_fast_export_image(mpt_ctx *c, int pass, float *out) {       // filmtable.py:66-79
    if (use_ro(c)) return 1;
    if (mpt_flush(c)) return 1;
    if (check_pass(c, pass)) return 1;
    HIP_TRY(mpt_launch_export(c->film[pass], c->exported, c->nx, c->ny, c->stream));
    if (read_back(c, out, c->exported, (size_t)c->nx * c->ny * 3 * sizeof(float))) return 1;
    return check_watchdog(c);
}

extern "C" int mpt_get_film_raw(mpt_ctx *c, int pass, float *out) {
    if (use_ro(c)) return 1;
    if (mpt_flush(c)) return 1;
    if (check_pass(c, pass)) return 1;
    if (read_back(c, out, c->film[pass], (size_t)c->nx * c->ny * sizeof(MptVec4))) return 1;
    return check_watchdog(c);
}

// ------------------------------------------------------------------ measurement
extern "C" int mpt_get_counters(mpt_ctx *c, mpt_counters *out) {
    if (use_ro(c)) return 1;
    if (mpt_flush(c)) return 1;
    unsigned long long h[20];
    HIP_TRY(hipMemcpyAsync(h, c->d_counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    out->samples = h[0]; out->rays = h[1]; out->n_box = h[2]; out->n_tri = h[3];
    out->n_shade = h[4]; out->n_draws = h[5]; out->bounces = h[6]; out->n_node = h[7];
    out->it_node = h[8]; out->it_leaf = h[9]; out->it_shade = h[10]; out->it_new = h[11];
    out->pl_local = h[12]; out->pl_batches = h[13]; out->pl_batch_lanes = h[14]; out->pl_prim = h[15];
    out->pl_tidle = h[16]; out->pl_sidle = h[17]; out->pl_trips = h[18]; out->pl_taken = h[19];
    return 0;
}

// diagnostics (option "lane_hist" = 1 and "count" = 1): out[0 .. 3 x 65) = issued NODE / LEAF / SHADE stages by the number of lanes
// that took part; out[195 ..) = [stage][depth 0 .. 5][closest, shadow] lane-steps; out[231 .. 255) = the gather kernels' NODE
// lane-steps by the bucket of the node's number (0 | 1 | 2-3 | 4-7 | ...).  Zeroed by mpt_reset_counters
extern "C" int mpt_get_lane_hist(mpt_ctx *c, unsigned long long *out, int n) {
    if (use_ro(c)) return 1;
    if (!out || n < MPT_HIST_WORDS) return fail("mpt_get_lane_hist: the buffer must hold %d words", (int)MPT_HIST_WORDS);
    if (mpt_flush(c)) return 1;
    HIP_TRY(hipMemcpyAsync(out, c->d_counters + MPT_HIST_BASE, MPT_HIST_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

// diagnostics: out[wave][4] = {start, scene ready, queue empty, exit} of the last LDS-kernel launch, 100 MHz ticks
extern "C" int mpt_get_timeline(mpt_ctx *c, unsigned long long *out, int cap_waves, int *nwaves) {
    if (use_ro(c)) return 1;
    if (mpt_synchronize(c)) return 1;
    if (!c->d_timeline) return fail("no timeline recorded: set option 'timeline' and render with the LDS kernel");
    int n = std::min(cap_waves, c->timeline_waves);
    if (out && n > 0)
        HIP_TRY(hipMemcpy(out, c->d_timeline, (size_t)n * MPT_TIMELINE_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (nwaves) *nwaves = c->timeline_waves;
    return 0;
}

// diagnostics: wall time from the launch of a one-workgroup kernel (threads lanes, lds_bytes of LDS) on a
// stream of its own to its completion, with whatever render launches are in flight left running -- how long
// a small foreign kernel (RCCL's) waits for a CU next to the persistent workgroups
extern "C" int mpt_probe_kernel(mpt_ctx *c, int threads, int lds_bytes, double *usec) {
    if (use_ro(c)) return 1;
    if (threads < 64 || threads > 1024 || lds_bytes < 4 * threads || lds_bytes > 64 * 1024)
        return fail("probe: threads in 64..1024, lds_bytes in 4*threads..65536");
    if (!c->probe_stream) HIP_TRY(hipStreamCreateWithFlags(&c->probe_stream, hipStreamNonBlocking));
    auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(mpt_launch_probe(c->d_scratch + 1, threads, (size_t)lds_bytes, c->probe_stream));
    HIP_TRY(hipStreamSynchronize(c->probe_stream));
    auto t1 = std::chrono::steady_clock::now();
    if (usec) *usec = std::chrono::duration<double, std::micro>(t1 - t0).count();
    return 0;
}

// test door (tools/soak.py, the tail finalisation's soak test): `count` device-to-device copies of `mbytes` MiB enqueued on a stream
// of their own -- HBM and L2 traffic beside the render launches, whose hand-off of samples between XCDs must not care.  Returns
// at once; count = 0 waits for the copies enqueued so far.
extern "C" int mpt_stress_copies(mpt_ctx *c, int mbytes, int count) {
    if (use_ro(c)) return 1;
    if (mbytes < 1 || mbytes > 4096 || count < 0 || count > 100000) return fail("stress_copies: mbytes in 1..4096, count in 0..100000");
    if (!c->stress_stream) HIP_TRY(hipStreamCreateWithFlags(&c->stress_stream, hipStreamNonBlocking));
    if (count == 0) { HIP_TRY(hipStreamSynchronize(c->stress_stream)); return 0; }
    const size_t bytes = (size_t)mbytes << 20;
    if (bytes != c->stress_bytes) {
        HIP_TRY(hipStreamSynchronize(c->stress_stream));
        hipFree(c->stress_buf); c->stress_buf = nullptr; c->stress_bytes = 0;
        HIP_TRY(hipMalloc((void **)&c->stress_buf, 2 * bytes));
        HIP_TRY(hipMemsetAsync(c->stress_buf, 0x5a, 2 * bytes, c->stress_stream));
        c->stress_bytes = bytes;
    }
    for (int k = 0; k < count; k++)
        HIP_TRY(hipMemcpyAsync(c->stress_buf + ((k & 1) ? 0 : bytes), c->stress_buf + ((k & 1) ? bytes : 0), bytes, hipMemcpyDeviceToDevice, c->stress_stream));
    return 0;
}

// test door: one device function of the hot path on rows of inputs (include/miptina.h, unit_eval.hip)
extern "C" int mpt_unit_eval(mpt_ctx *c, int kind, const void *in, int in_cols, void *out, int out_cols, int n) {
    if (use_ro(c)) return 1;
    static const int cols[MPT_UNIT_KINDS][2] = {
        { 1, 1 }, { 3, 1 }, { 2, 1 }, { 2, 1 }, { 2, 1 }, { 3, 3 }, { 3, 3 }, { 6, 3 }, { 2, 3 }, { 3, 2 }, { 6, 3 }, { 7, 4 },
        { 12, 3 }, { 30, 9 }, { 10, 1 }, { 15, 4 }, { 24, 3 }, { 24, 7 }, { 2, 1 }, { 1, 1 }, { 2, 1 } };
    if (kind < 0 || kind >= MPT_UNIT_KINDS) return fail("unit kind %d outside [0, %d)", kind, (int)MPT_UNIT_KINDS);
    if (in_cols != cols[kind][0] || out_cols != cols[kind][1])
        return fail("unit kind %d takes %d input and %d output columns, got %d and %d", kind, cols[kind][0], cols[kind][1],
                    in_cols, out_cols);
    if (n < 0 || (n > 0 && (!in || !out))) return fail("bad unit_eval arguments");
    if (n == 0) return 0;
    float *d_in = nullptr, *d_out = nullptr;
    if (dev_alloc(&d_in, (size_t)n * in_cols)) return 1;
    if (dev_alloc(&d_out, (size_t)n * out_cols)) { hipFree(d_in); return 1; }
    hipError_t e = hipMemcpyAsync(d_in, in, (size_t)n * in_cols * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_out, 0, (size_t)n * out_cols * 4, c->stream);
    if (e == hipSuccess)
        e = c->mode == MPT_MODE_STRICT ? mpt_launch_unit_eval_strict(kind, d_in, in_cols, d_out, out_cols, n, c->stream)
                                       : mpt_launch_unit_eval_fast(kind, d_in, in_cols, d_out, out_cols, n, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, (size_t)n * out_cols * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d_in); hipFree(d_out);
    if (e != hipSuccess) return fail("mpt_unit_eval: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int mpt_reset_counters(mpt_ctx *c) {
    if (use(c)) return 1;
    HIP_TRY(hipMemsetAsync(c->d_counters, 0, MPT_COUNTER_WORDS * sizeof(unsigned long long), c->stream));
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
