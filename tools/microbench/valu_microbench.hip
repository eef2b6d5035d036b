// valu_microbench.hip -- what a wave64 VALU instruction costs a gfx950 SIMD, measured.
//
// DESIGN.md prices the render kernel against the f32 vector lane rate (256 CUs x 4 SIMD-32 x clock).
// That needs one hardware fact: how many cycles a SIMD is busy per wave64 VALU instruction when it
// has 1, 2, 4 or 8 waves to pick from.  This program times dependency-free instruction streams
// (16 independent accumulators per lane, inline asm so nothing is packed, fused or hoisted):
//     fma     v_fma_f32            (the kernel's bread and butter)
//     pkfma   v_pk_fma_f32         (what the SLP vectoriser emits; 2 FMAs per lane per instruction)
//     rcp     v_rcp_f32            (transcendental unit)
//     cndmask v_cndmask_b32 + v_cmp_lt_f32  (selects and compares: the state machine's bookkeeping)
// with exactly W workgroups of 256 lanes resident per CU (W waves per SIMD; enforced with an LDS
// allocation of 160 KiB / W per workgroup) and reports, per case,
//     wave-instructions/s per SIMD, and cycles per wave-instruction at the in-kernel clock
//     (s_memtime ticks over s_memrealtime's 100 MHz ticks, median over workgroups).
// Run it bare for the table, or under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
// SQ_BUSY_CYCLES SQ_WAVE_CYCLES` to see what the counters the roofline uses say about the same streams.
//
// build: hipcc -O2 --offload-arch=gfx950 valu_microbench.hip -o valu_microbench

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { K_FMA = 0, K_PKFMA = 1, K_RCP = 2, K_CNDMASK = 3, K_PKFMA16 = 4, K_PKMIN16 = 5, K_PERM = 6, K_MAXF = 7, K_MINU = 8, K_ADDU = 9, K_MAX3 = 10, K_MUL = 11, K_ANDOR = 12, K_BFI = 13, K_LSHLOR = 14, K_BFE = 15, K_CNDMASK1 = 16, K_CMP1 = 17, K_MOV = 18, K_LSHLADD = 19 };
static const char *kname[] = { "v_fma_f32", "v_pk_fma_f32", "v_rcp_f32", "v_cmp_lt_f32+v_cndmask_b32", "v_pk_fma_f16", "v_pk_min_f16", "v_perm_b32", "v_max_f32", "v_min_u32", "v_add_u32", "v_max3_f32", "v_mul_f32", "v_and_or_b32", "v_bfi_b32", "v_lshl_or_b32", "v_bfe_i32", "v_cndmask_b32 (vcc set once)", "v_cmp_lt_f32 alone", "v_mov_b32", "v_lshl_add_u32" };
static const int vinsts_per_iter[] = { 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64 };

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

template <int KIND>
__global__ __launch_bounds__(256) void valu_loop(float *out, unsigned long long *stamps, int iters, float x, float y) {
    extern __shared__ int pad_lds[];           // residency control only
    if (threadIdx.x == 0 && iters < 0) pad_lds[0] = 1;
    float a[16];
    f2 p[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { a[k] = (float)(threadIdx.x + k); p[k] = (f2){ a[k], a[k] + 1.0f }; }
    f2 xx = (f2){ x, x }, yy = (f2){ y, y };
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (KIND == K_FMA) {
#define M(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(x), "v"(y));
                REP16(M)
#undef M
            } else if (KIND == K_PKFMA) {
#define M(k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[k]) : "v"(xx), "v"(yy));
                REP16(M)
#undef M
            } else if (KIND == K_PKFMA16) {
                // round 5: two f16 FMAs per lane per instruction -- is THAT double rate? (the 4-wide NODE step's 24 FMAs and 16 min / max as 12 + 12)
#define M(k) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(a[k]) : "v"(x), "v"(y));
                REP16(M)
#undef M
            } else if (KIND == K_PKMIN16) {
#define M(k) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(a[k]) : "v"(x));
                REP16(M)
#undef M
            } else if (KIND == K_MAXF) {
#define M(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(x));
                REP16(M)
#undef M
            } else if (KIND == K_MINU) {
#define M(k) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[k]) : "v"(x));
                REP16(M)
#undef M
            } else if (KIND == K_ADDU) {
#define M(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(x));
                REP16(M)
#undef M
            } else if (KIND == K_MAX3) {
#define M(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));
                REP16(M)
#undef M
            } else if (KIND == K_MUL) {
#define M(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(x));
                REP16(M)
#undef M
            } else if (KIND == K_ANDOR) {
#define M(k) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));
                REP16(M)
#undef M
            } else if (KIND == K_BFI) {
#define M(k) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[k]) : "v"(x), "v"(y));
                REP16(M)
#undef M
            } else if (KIND == K_LSHLOR) {
#define M(k) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(a[k]) : "v"(x));
                REP16(M)
#undef M
            } else if (KIND == K_BFE) {
#define M(k) asm volatile("v_bfe_i32 %0, %0, 0, 16" : "+v"(a[k]));
                REP16(M)
#undef M
            } else if (KIND == K_CNDMASK1) {
                asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x), "v"(y) : "vcc");
#define M(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(x) : "vcc");
                REP16(M)
#undef M
            } else if (KIND == K_CMP1) {
#define M(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(x) : "vcc");
                REP16(M)
#undef M
            } else if (KIND == K_MOV) {
#define M(k) asm volatile("v_mov_b32 %0, %1" : "=v"(a[k]) : "v"(x));
                REP16(M)
#undef M
            } else if (KIND == K_LSHLADD) {
#define M(k) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[k]) : "v"(x));
                REP16(M)
#undef M
            } else if (KIND == K_PERM) {
#define M(k) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));
                REP16(M)
#undef M
            } else if (KIND == K_RCP) {
#define M(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
                REP16(M)
#undef M
            } else {
                // 8 x (compare into vcc, select on vcc): 16 VALU instructions
#define M(k) if ((k) < 8) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(x), "v"(a[(k) + 8]) : "vcc");
                REP16(M)
#undef M
            }
        }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; k++) s += a[k] + p[k].x + p[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int KIND>
static void run(int ncu, int waves_per_simd, int iters, float *d_out, unsigned long long *d_st, bool json, bool first) {
    const int grid = ncu * waves_per_simd;
    // 160 KiB per CU: an allocation of 160/W KiB (minus slack) admits exactly W workgroups per CU
    size_t lds = (160 * 1024) / waves_per_simd - 512;
    if (lds > 64 * 1024) CHECK(hipFuncSetAttribute((const void *)valu_loop<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(valu_loop<KIND>, dim3(grid), dim3(256), lds, 0, d_out, d_st, iters / 8, 1.0001f, 0.5f);   // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(valu_loop<KIND>, dim3(grid), dim3(256), lds, 0, d_out, d_st, iters, 1.0001f, 0.5f);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st(2 * grid);
    CHECK(hipMemcpy(st.data(), d_st, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> clk, cyc;
    for (int b = 0; b < grid; b++) {
        if (st[2 * b + 1]) clk.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 100e6);
        cyc.push_back((double)st[2 * b]);
    }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    const double clock_hz = clk.empty() ? 0 : clk[clk.size() / 2];
    const double loop_cycles = cyc[cyc.size() / 2];                         // one workgroup's loop, shader cycles
    const double insts_per_wave = (double)iters * vinsts_per_iter[KIND == K_CNDMASK ? 3 : KIND];
    const double insts_per_simd = insts_per_wave * waves_per_simd;         // 4 waves of a workgroup -> 4 SIMDs
    const double cyc_per_inst_simd = loop_cycles / insts_per_simd;         // SIMD cycles per wave-instruction
    const double cyc_per_inst_wave = loop_cycles / insts_per_wave;         // what ONE wave sees
    const double wall_rate = insts_per_simd / (ms * 1e-3);                 // wave-instructions / s / SIMD, by hipEvents
    if (json)
        printf("%s{\"inst\": \"%s\", \"waves_per_simd\": %d, \"wave_insts_per_wave\": %.0f, \"kernel_ms\": %.4f, "
               "\"clock_ghz\": %.4f, \"cycles_per_wave_inst_per_simd\": %.4f, \"cycles_per_wave_inst_seen_by_one_wave\": %.4f, "
               "\"wave_insts_per_s_per_simd\": %.4e, \"lane_ops_per_s_chip\": %.4e}",
               first ? "" : ",\n ", kname[KIND], waves_per_simd, insts_per_wave, ms, clock_hz / 1e9, cyc_per_inst_simd,
               cyc_per_inst_wave, wall_rate, wall_rate * 64.0 * ((KIND == K_PKFMA || KIND == K_PKFMA16) ? 2 : 1) * ncu * 4);
    else
        printf("%-28s W=%d  %.3f ms  clock %.3f GHz  %.3f cycles/wave-inst/SIMD  (one wave sees %.2f)  %.3e lane-ops/s chip\n",
               kname[KIND], waves_per_simd, ms, clock_hz / 1e9, cyc_per_inst_simd, cyc_per_inst_wave,
               wall_rate * 64.0 * ((KIND == K_PKFMA || KIND == K_PKFMA16) ? 2 : 1) * ncu * 4);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main(int argc, char **argv) {
    bool json = argc > 1 && !strcmp(argv[1], "--json");
    int iters = 20000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    float *d_out; unsigned long long *d_st;
    CHECK(hipMalloc(&d_out, (size_t)ncu * 8 * 256 * sizeof(float)));
    CHECK(hipMalloc(&d_st, (size_t)ncu * 8 * 2 * sizeof(unsigned long long)));
    if (json) printf("{\"device\": \"%s\", \"cus\": %d, \"clockRate_khz\": %d, \"cases\": [\n ", prop.gcnArchName, ncu, prop.clockRate);
    else printf("%s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, ncu, prop.clockRate);
    bool first = true;
    const int ws[] = { 1, 2, 4, 8 };
    for (int w : ws) { run<K_FMA>(ncu, w, iters, d_out, d_st, json, first); first = false; }
    for (int w : ws) run<K_PKFMA>(ncu, w, iters, d_out, d_st, json, false);
    for (int w : ws) run<K_RCP>(ncu, w, iters / 2, d_out, d_st, json, false);
    for (int w : ws) run<K_CNDMASK>(ncu, w, iters, d_out, d_st, json, false);
    for (int w : ws) run<K_PKFMA16>(ncu, w, iters, d_out, d_st, json, false);
    for (int w : ws) run<K_PKMIN16>(ncu, w, iters, d_out, d_st, json, false);
    for (int w : ws) run<K_PERM>(ncu, w, iters, d_out, d_st, json, false);
    for (int w : ws) run<K_MAXF>(ncu, w, iters, d_out, d_st, json, false);
    for (int w : ws) run<K_MINU>(ncu, w, iters, d_out, d_st, json, false);
    for (int w : ws) run<K_ADDU>(ncu, w, iters, d_out, d_st, json, false);
    for (int w : ws) run<K_MAX3>(ncu, w, iters, d_out, d_st, json, false);
    for (int w : ws) run<K_MUL>(ncu, w, iters, d_out, d_st, json, false);
    for (int w : ws) run<K_ANDOR>(ncu, w, iters, d_out, d_st, json, false);
    for (int w : ws) run<K_BFI>(ncu, w, iters, d_out, d_st, json, false);
    for (int w : ws) run<K_LSHLOR>(ncu, w, iters, d_out, d_st, json, false);
    for (int w : ws) run<K_BFE>(ncu, w, iters, d_out, d_st, json, false);
    for (int w : ws) run<K_CNDMASK1>(ncu, w, iters, d_out, d_st, json, false);
    for (int w : ws) run<K_CMP1>(ncu, w, iters, d_out, d_st, json, false);
    for (int w : ws) run<K_MOV>(ncu, w, iters, d_out, d_st, json, false);
    for (int w : ws) run<K_LSHLADD>(ncu, w, iters, d_out, d_st, json, false);
    if (json) printf("\n]}\n");
    return 0;
}
