// comm.cpp -- RCCL (bound lazily with dlopen) and the film gather of the multi-GPU split.

#include "miptina_ctx.h"

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
    if (root < 0 || root >= R) return fail("gather root %d outside [0, %d)", root, R);
    if (c->stripe_w && (c->stripe_mod != R || c->stripe_idx != c->rank))
        return fail("stripes (index %d of %d) do not match the communicator (rank %d of %d)", c->stripe_idx,
                    c->stripe_mod, c->rank, R);
    if (!c->stripe_w && R > 1) {
        // slab mode: what this rank rendered must be exactly the columns the others expect from it
        const int lo = (int)((long long)c->rank * c->nx / R), hi = (int)((long long)(c->rank + 1) * c->nx / R);
        if (c->x0 != lo || c->x1 != hi)
            return fail("slab [%d,%d) of rank %d does not match the communicator's split [%d,%d) of %d ranks: "
                        "set it with mpt_set_slab(rank*nx/R, (rank+1)*nx/R) or use mpt_set_stripes",
                        c->x0, c->x1, c->rank, lo, hi, R);
    }
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
    ncclResult_t bad = ncclSuccess;                // a failing call must not leave the group open
    if (c->rank == root) {
        for (int r = 0; r < R && bad == ncclSuccess; r++) {
            if (r == root) continue;
            pieces(r, pc);
            for (auto &q : pc) {
                bad = g_rccl.Recv(c->film[pass] + q.first, q.second * 4, ncclFloat, r, c->comm, c->stream);
                if (bad != ncclSuccess) break;
            }
        }
    } else {
        pieces(c->rank, pc);
        for (auto &q : pc) {
            bad = g_rccl.Send(c->film[pass] + q.first, q.second * 4, ncclFloat, root, c->comm, c->stream);
            if (bad != ncclSuccess) break;
        }
    }
    ncclResult_t ended = g_rccl.GroupEnd();
    if (bad != ncclSuccess) return fail("ncclSend/ncclRecv of the film gather failed: %s", g_rccl.GetErrorString(bad));
    NCCL_TRY(ended);
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


void mpt_comm_release(mpt_ctx *c) {
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    c->comm = nullptr;
}
