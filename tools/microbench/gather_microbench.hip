// gather_microbench.hip -- what a 64-byte record costs a gfx950 CU when every lane of a wave wants a different one.
//
// The gather kernels' traversal step reads one 64-byte node per lane as four 16-byte loads (what they wait for
// is the NUMBER of gather instructions).  This program times a dependent chain of such reads -- the next record's index comes
// out of the bytes just read, as in a traversal -- in three forms:
//     own     every lane issues four 16-B loads from its own record: 4 instructions x 64 different cache lines
//     quad    the four lanes of a quad fetch the record of quad member i together, i = 0..3 (lane p takes bytes 16p..16p+15):
//             4 instructions x 16 different cache lines, each quad one aligned 64-byte piece; every lane then holds piece p of
//             the four records of its quad and the 4 x 4 transposition that gives each lane its own record is done with DPP
//     quadraw the same loads without the transposition (what the fetch alone costs)
// with W workgroups of 256 lanes resident per CU (LDS allocation of 160 KiB / W), a fraction `active` of the lanes taking part
// in a step (the kernels run their NODE steps at 55-60 % of the lanes), over tables of several sizes (L2, Infinity Cache, HBM).
// Reported: records per second per CU and chip-wide, and ns per wave-step.
//
// build: hipcc -O2 --offload-arch=gfx950 gather_microbench.hip -o gather_microbench

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

enum { M_OWN = 0, M_QUAD = 1, M_QUADRAW = 2 };
static const char *mname[] = { "own", "quad", "quadraw" };

__device__ inline unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int CTRL> __device__ inline unsigned dpp(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

// out_p(lane j of the quad) = r_j(lane p): two rounds of conditional exchanges with the lane one, then two, away
#define TRANSPOSE4(r0, r1, r2, r3) {                                                             \
    const unsigned n0 = b0 ? dpp<0xB1>(r1) : r0, n1 = b0 ? r1 : dpp<0xB1>(r0);                    \
    const unsigned n2 = b0 ? dpp<0xB1>(r3) : r2, n3 = b0 ? r3 : dpp<0xB1>(r2);                    \
    r0 = b1 ? dpp<0x4E>(n2) : n0; r2 = b1 ? n2 : dpp<0x4E>(n0);                                   \
    r1 = b1 ? dpp<0x4E>(n3) : n1; r3 = b1 ? n3 : dpp<0x4E>(n1); }

template <int MODE>
__global__ __launch_bounds__(256) void gather_loop(const u4 *__restrict__ table, unsigned nrec, unsigned *out, int iters, unsigned active_256) {
    extern __shared__ int pad_lds[];           // residency control only
    if (threadIdx.x == 0 && iters < 0) pad_lds[0] = 1;
    const unsigned lane = threadIdx.x & 63u, p = lane & 3u;
    unsigned idx = mix(blockIdx.x * 256u + threadIdx.x) % nrec;
    unsigned acc = 0, seed = mix(idx + 77u);
    for (int it = 0; it < iters; it++) {
        seed = mix(seed + (unsigned)it);
        const bool on = (seed & 255u) < active_256;
        u4 a = { 0, 0, 0, 0 }, b = a, c = a, d = a;
        if (MODE == M_OWN) {
            if (on) {
                const u4 *r = table + (size_t)idx * 4;
                a = r[0]; b = r[1];
                c = r[2]; d = r[3];
            }
        } else {
            const unsigned want = on ? idx : 0xffffffffu;
            const unsigned w0 = dpp<0x00>(want), w1 = dpp<0x55>(want), w2 = dpp<0xAA>(want), w3 = dpp<0xFF>(want);
            if (w0 != 0xffffffffu) a = table[(size_t)w0 * 4 + p];
            if (w1 != 0xffffffffu) b = table[(size_t)w1 * 4 + p];
            if (w2 != 0xffffffffu) c = table[(size_t)w2 * 4 + p];
            if (w3 != 0xffffffffu) d = table[(size_t)w3 * 4 + p];
            if (MODE == M_QUAD) {
                const bool b0 = lane & 1u, b1 = lane & 2u;
                TRANSPOSE4(a.x, b.x, c.x, d.x) TRANSPOSE4(a.y, b.y, c.y, d.y)
                TRANSPOSE4(a.z, b.z, c.z, d.z) TRANSPOSE4(a.w, b.w, c.w, d.w)
            }
        }
        // the next record comes out of the bytes read (every word takes part, so that no load can be dropped)
        const unsigned h = (a.x ^ b.y ^ c.z ^ d.w) + (a.y ^ b.z ^ c.w ^ d.x) + (a.z ^ b.w ^ c.x ^ d.y) + (a.w ^ b.x ^ c.y ^ d.z);
        acc += h;
        if (on) idx = mix(h + idx + seed) % nrec;     // (seed: a pure function of idx would fall into a short cycle of cached records)
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc + idx;
}

int main(int argc, char **argv) {
    int iters = 4000;
    for (int i = 1; i < argc; i++) if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t max_rec = (size_t)1 << 22;                   // 256 MiB of 64-byte records
    u4 *table; unsigned *out;
    CHECK(hipMalloc(&table, max_rec * 64));
    std::vector<unsigned> h(max_rec * 16);
    unsigned s = 12345u;
    for (size_t i = 0; i < h.size(); i++) { s = s * 1664525u + 1013904223u; h[i] = s >> 3; }
    CHECK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const unsigned sizes[] = { 1u << 14, 1u << 18, 333333u, 1u << 20, 1u << 22 };          // 1 MiB, 16 MiB, 21 MB, 64 MiB, 256 MiB
    const int waves[] = { 4, 5 };
    const unsigned actives[] = { 256, 148 };
    printf("%-8s %9s %3s %6s %12s %12s %10s\n", "mode", "table", "W", "active", "Grec/s", "Mrec/s/CU", "ns/step");
    for (unsigned nrec : sizes) for (int W : waves) for (unsigned act : actives) for (int mode = 0; mode < 3; mode++) {
        const int lds = 160 * 1024 / W - 512;
        const int grid = cus * W;
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            if (mode == M_OWN) hipLaunchKernelGGL(gather_loop<M_OWN>, dim3(grid), dim3(256), lds, 0, table, nrec, out, iters, act);
            else if (mode == M_QUAD) hipLaunchKernelGGL(gather_loop<M_QUAD>, dim3(grid), dim3(256), lds, 0, table, nrec, out, iters, act);
            else hipLaunchKernelGGL(gather_loop<M_QUADRAW>, dim3(grid), dim3(256), lds, 0, table, nrec, out, iters, act);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        const double recs = (double)grid * 256 * iters * (act / 256.0);
        printf("%-8s %7.1fMB %3d %6.2f %12.2f %12.1f %10.1f\n", mname[mode], nrec * 64 / 1e6, W, act / 256.0,
               recs / best / 1e6, recs / best / 1e3 / cus, best * 1e6 / iters);
        fflush(stdout);
    }
    return 0;
}
