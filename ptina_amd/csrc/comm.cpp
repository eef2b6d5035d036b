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

// The split, as a pure function (no context, no GPU; unit-tested on the CPU against ptina_amd.dist): the float4
// ranges of the film (index x*ny + y) that rank r of R owns, in ascending x -- the order they are packed in.
//   stripe_w == 0 : one contiguous slab, columns [r*nx/R, (r+1)*nx/R)
//   stripe_w  > 0 : stripes r, r+R, ... of stripe_w columns (mpt_set_stripes(stripe_w, r, R)), the last one ragged
// Writes the first `cap` pieces to offsets[] / counts[] (either may be NULL) and returns the number of pieces
// (0 for an empty share), or -1 for arguments that name no split.
extern "C" int mpt_comm_plan(int nx, int ny, int stripe_w, int r, int R, int64_t *offsets, int64_t *counts, int cap) {
    if (nx < 0 || ny < 0 || stripe_w < 0 || R < 1 || r < 0 || r >= R) return -1;
    int np = 0;
    auto put = [&](long long x, long long w) {
        if (w <= 0 || ny == 0) return;
        if (np < cap) {
            if (offsets) offsets[np] = (int64_t)(x * ny);
            if (counts) counts[np] = (int64_t)(w * ny);
        }
        np++;
    };
    if (stripe_w == 0) {
        const long long lo = (long long)r * nx / R, hi = (long long)(r + 1) * nx / R;
        put(lo, hi - lo);
    } else {
        for (long long x = (long long)r * stripe_w; x < nx; x += (long long)stripe_w * R)
            put(x, std::min<long long>(stripe_w, nx - x));
    }
    return np;
}

// the plan of one rank as (film offset, count) pairs
static void plan_of(const mpt_ctx *c, int r, int R, std::vector<int64_t> &off, std::vector<int64_t> &cnt) {
    int np = mpt_comm_plan(c->nx, c->ny, c->stripe_w, r, R, nullptr, nullptr, 0);
    off.assign(std::max(np, 0), 0); cnt.assign(std::max(np, 0), 0);
    if (np > 0) mpt_comm_plan(c->nx, c->ny, c->stripe_w, r, R, off.data(), cnt.data(), np);
}

// Gather = ONE message per peer (one contiguous buffer per GPU).  A share that is one range (slab
// split, or a single stripe) travels straight from film to film; a share of several stripes is packed side by side
// into gather_buf by copy_pieces, sent as one range, received by the root into its own gather_buf (every peer's
// message back to back) and scattered into the film by one copy_pieces launch in front of the resolve.  The root
// posts R - 1 receives in one group whatever the film size (the round-2 code posted one per stripe: 112 at 2048^2).
extern "C" int mpt_comm_gather_film(mpt_ctx *c, int pass, int root) {
    if (use_ro(c)) return 1;
    if (mpt_flush(c)) return 1;
    if (check_pass(c, pass)) return 1;
    if (!c->comm) return fail("communicator not initialised");
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
    if (R == 1) return 0;
    c->film_version++;                          // the root's film is about to change under any image written early (mpt_hint_image)
    // ---- the piece table of this rank's side of the gather, rebuilt when the split changes
    //   sender: its ranges film -> packed;  root: every peer's ranges packed -> film
    const bool is_root = c->rank == root;
    std::vector<int64_t> off, cnt;
    std::vector<long long> msg_elems(R, 0), msg_base(R, 0);   // per peer: float4 in its message, where it starts in gather_buf
    std::vector<int> msg_pieces(R, 0);
    std::vector<long long> first_off(R, 0);
    std::vector<MptPiece> tab;
    long long total = 0, max_count = 0;
    for (int r = 0; r < R; r++) {
        if (r == root || (!is_root && r != c->rank)) continue;
        plan_of(c, r, R, off, cnt);
        msg_pieces[r] = (int)off.size();
        first_off[r] = off.empty() ? 0 : off[0];
        long long n = 0;
        for (size_t i = 0; i < off.size(); i++) n += cnt[i];
        msg_elems[r] = n;
        if (off.size() > 1) {                       // packed: its message lives in gather_buf
            msg_base[r] = total;
            long long at = total;
            for (size_t i = 0; i < off.size(); i++) {
                MptPiece pc;
                if (is_root) { pc.src = at; pc.dst = off[i]; } else { pc.src = off[i]; pc.dst = at; }
                pc.count = cnt[i];
                tab.push_back(pc);
                at += cnt[i];
                max_count = std::max<long long>(max_count, cnt[i]);
            }
            total += n;
        }
    }
    const int key[6] = { c->nx, c->ny, c->stripe_w, R, c->rank, root };
    if (memcmp(key, c->plan_key, sizeof key) != 0) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        // the cached plan is dead from here on: a failure below must not leave a key that matches a later call (round-3 ADVICE)
        for (int k = 0; k < 6; k++) c->plan_key[k] = -1;
        c->plan_npieces = 0; c->plan_max_count = 0;
        if (tab.size() > 65535) return fail("film gather: %zu pieces exceed the 65535 the pack kernel's grid takes (wider stripes)", tab.size());
        if ((size_t)total > c->gather_cap) {
            hipFree(c->gather_buf); c->gather_buf = nullptr; c->gather_cap = 0;
            if (dev_alloc(&c->gather_buf, (size_t)total)) return 1;
            c->gather_cap = (size_t)total;
        }
        if (tab.size() > c->pieces_cap) {
            hipFree(c->d_pieces); c->d_pieces = nullptr; c->pieces_cap = 0;
            if (dev_alloc(&c->d_pieces, tab.size())) return 1;
            c->pieces_cap = tab.size();
        }
        if (!tab.empty()) HIP_TRY(hipMemcpy(c->d_pieces, tab.data(), tab.size() * sizeof(MptPiece), hipMemcpyHostToDevice));
        memcpy(c->plan_key, key, sizeof key);
        c->plan_npieces = (int)tab.size(); c->plan_max_count = max_count;
    }
    // ---- sender: pack, then one send
    if (!is_root && msg_pieces[c->rank] > 1)
        HIP_TRY(mpt_launch_copy_pieces(c->film[pass], c->gather_buf, c->d_pieces, c->plan_npieces, c->plan_max_count, c->stream));
    NCCL_TRY(g_rccl.GroupStart());
    ncclResult_t bad = ncclSuccess;                // a failing call must not leave the group open
    if (is_root) {
        for (int r = 0; r < R && bad == ncclSuccess; r++) {
            if (r == root || msg_elems[r] == 0) continue;
            MptVec4 *dst = msg_pieces[r] > 1 ? c->gather_buf + msg_base[r] : c->film[pass] + first_off[r];
            bad = g_rccl.Recv(dst, (size_t)msg_elems[r] * 4, ncclFloat, r, c->comm, c->stream);
        }
    } else if (msg_elems[c->rank] > 0) {
        const MptVec4 *src = msg_pieces[c->rank] > 1 ? c->gather_buf : c->film[pass] + first_off[c->rank];
        bad = g_rccl.Send(src, (size_t)msg_elems[c->rank] * 4, ncclFloat, root, c->comm, c->stream);
    }
    ncclResult_t ended = g_rccl.GroupEnd();
    if (bad != ncclSuccess) return fail("ncclSend/ncclRecv of the film gather failed: %s", g_rccl.GetErrorString(bad));
    NCCL_TRY(ended);
    // ---- root: scatter every packed message into the film
    if (is_root && c->plan_npieces > 0)
        HIP_TRY(mpt_launch_copy_pieces(c->gather_buf, c->film[pass], c->d_pieces, c->plan_npieces, c->plan_max_count, c->stream));
    return 0;
}

// Test door for the pack / scatter half of the gather on ONE GPU (no communicator needed): plays rank `as_rank` of
// `nranks` sending to `root` and the root receiving, with a device-to-device copy standing in for the RCCL message.
// `film_in` [nx*ny*4] is the sender's film (only its share is read); `film_out` the root's film before (in) and
// after (out) the scatter.  Uses the context's film size and stripe width; leaves the context's own film alone.
extern "C" int mpt_comm_selftest(mpt_ctx *c, int as_rank, int nranks, int root, const float *film_in, float *film_out) {
    if (use_ro(c)) return 1;
    if (c->nx <= 0) return fail("film size not set");
    if (nranks < 1 || as_rank < 0 || as_rank >= nranks || root < 0 || root >= nranks || as_rank == root)
        return fail("selftest: need 0 <= as_rank != root < nranks");
    const size_t npix = (size_t)c->nx * c->ny;
    std::vector<int64_t> off, cnt;
    int np = mpt_comm_plan(c->nx, c->ny, c->stripe_w, as_rank, nranks, nullptr, nullptr, 0);
    off.assign(std::max(np, 0), 0); cnt.assign(std::max(np, 0), 0);
    if (np > 0) mpt_comm_plan(c->nx, c->ny, c->stripe_w, as_rank, nranks, off.data(), cnt.data(), np);
    std::vector<MptPiece> pack, scatter;
    long long at = 0, max_count = 0;
    for (int i = 0; i < np; i++) {
        pack.push_back({ off[i], at, cnt[i] });
        scatter.push_back({ at, off[i], cnt[i] });
        at += cnt[i]; max_count = std::max<long long>(max_count, cnt[i]);
    }
    MptVec4 *d_in = nullptr, *d_out = nullptr, *d_msg = nullptr, *d_msg2 = nullptr;
    MptPiece *d_tab = nullptr;
    int rc = 1;
    do {
        if (dev_alloc(&d_in, npix) || dev_alloc(&d_out, npix) || dev_alloc(&d_msg, (size_t)at) || dev_alloc(&d_msg2, (size_t)at) ||
            dev_alloc(&d_tab, (size_t)std::max(np, 1) * 2)) break;
        // every copy on the context's own stream, the one the pack / scatter kernels run on: ordered by the stream itself and
        // not by what a blocking copy on the null stream happens to have finished when it returns
        if (hipMemcpyAsync(d_in, film_in, npix * sizeof(MptVec4), hipMemcpyHostToDevice, c->stream) != hipSuccess) { fail("selftest upload"); break; }
        if (hipMemcpyAsync(d_out, film_out, npix * sizeof(MptVec4), hipMemcpyHostToDevice, c->stream) != hipSuccess) { fail("selftest upload"); break; }
        if (np > 0) {
            if (hipMemcpyAsync(d_tab, pack.data(), (size_t)np * sizeof(MptPiece), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                hipMemcpyAsync(d_tab + np, scatter.data(), (size_t)np * sizeof(MptPiece), hipMemcpyHostToDevice, c->stream) != hipSuccess) {
                fail("selftest upload"); break;
            }
            if (mpt_launch_copy_pieces(d_in, d_msg, d_tab, np, max_count, c->stream) != hipSuccess) { fail("selftest pack"); break; }
            hipMemcpyAsync(d_msg2, d_msg, (size_t)at * sizeof(MptVec4), hipMemcpyDeviceToDevice, c->stream);   // the "message"
            if (mpt_launch_copy_pieces(d_msg2, d_out, d_tab + np, np, max_count, c->stream) != hipSuccess) { fail("selftest scatter"); break; }
        }
        if (hipStreamSynchronize(c->stream) != hipSuccess) { fail("selftest sync"); break; }
        if (hipMemcpyAsync(film_out, d_out, npix * sizeof(MptVec4), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) { fail("selftest download"); break; }
        rc = 0;
    } while (0);
    hipFree(d_in); hipFree(d_out); hipFree(d_msg); hipFree(d_msg2); hipFree(d_tab);
    return rc;
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
