// exec_microbench.hip -- does a gfx950 SIMD spend less time on a wave64 VALU instruction when part of the wave is
// switched off, and what do the conversion / logic instructions of the 8-bit node decode cost?
// (cycles = the launch's wall time by hipEvents x the in-loop shader clock / wave-instructions per SIMD)
//
// The render kernels run at 0.41-0.50 lane occupancy: every issued vector instruction carries switched-off lanes.
// If the SIMD skipped a half (or quarter) of a wave whose EXEC bits are all zero, keeping the active lanes of a
// wave together would pay; if it does not, only fewer instructions do.  This program times dependency-free
// streams of one instruction (16 independent accumulators, inline asm) with EXEC forced to a pattern for the
// whole loop: all 64 lanes | the low 32 | the low 16 | every other lane | lanes 0-15 and 32-47 | one lane.
// W workgroups of 256 lanes resident per CU (W waves per SIMD) as in valu_microbench.hip.
//
// build: hipcc -O2 --offload-arch=gfx950 exec_microbench.hip -o exec_microbench

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { K_FMA, K_MAX, K_CVTUB, K_OR, K_AND, K_LSHL, K_SUB, K_MED3, K_CVTU, K_XOR, K_MADU24, K_MULU24, K_ORSDWA, K_CVTSDWA, K_MIN3, K_ADD3, K_FMAMIX, K_CVTF16, K_CVTFP8, K_MAD64, K_LSHLADD64, K_LSHL64, K_MULLO, K_MULHI, K_ASHR, K_MIXFM, K_MIXFC, K_LSHR, K_NKINDS };
static const char *kname[] = { "v_fma_f32", "v_max_f32", "v_cvt_f32_ubyte1", "v_or_b32", "v_and_b32", "v_lshlrev_b32", "v_sub_f32", "v_med3_f32",
                               "v_cvt_f32_u32", "v_xor_b32", "v_mad_u32_u24", "v_mul_u32_u24", "v_or_b32_sdwa (BYTE_1)", "v_cvt_f32_u32_sdwa (BYTE_1)",
                               "v_min3_f32", "v_add3_u32", "v_fma_mix_f32 (f16 hi, f32, f32)", "v_cvt_f32_f16", "v_cvt_pk_f32_fp8 (2 values)", "v_mad_i64_i32", "v_lshl_add_u64", "v_lshlrev_b64", "v_mul_lo_u32", "v_mul_hi_u32", "v_ashrrev_i32", "v_fma_f32 / v_max_f32 alternating", "v_fma_f32 / v_cvt_f32_ubyte1 alternating", "v_lshrrev_b32" };

#define REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

template <int KIND>
__global__ __launch_bounds__(256) void exec_loop(float *out, unsigned long long *stamps, int iters, float x, float y, unsigned long long mask) {
    extern __shared__ int pad_lds[];           // residency control only
    if (threadIdx.x == 0 && iters < 0) pad_lds[0] = 1;
    float a[16];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 q[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { a[k] = (float)(threadIdx.x + k); q[k] = (f2){ a[k], a[k] }; }
    unsigned long long saved;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1" : "=s"(saved) : "s"(mask));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#define ONE(txt, ...) asm volatile(txt : "+v"(a[k_]) : __VA_ARGS__);
            if (KIND == K_FMA) {
#define M(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(x), "v"(y));
                REP16(M)
#undef M
            } else if (KIND == K_MAX) {
#define M(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(x));
                REP16(M)
#undef M
            } else if (KIND == K_CVTUB) {
#define M(k) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[k]));
                REP16(M)
#undef M
            } else if (KIND == K_OR) {
#define M(k) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[k]) : "v"(x));
                REP16(M)
#undef M
            } else if (KIND == K_AND) {
#define M(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(x));
                REP16(M)
#undef M
            } else if (KIND == K_LSHL) {
#define M(k) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[k]));
                REP16(M)
#undef M
            } else if (KIND == K_SUB) {
#define M(k) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[k]) : "v"(x));
                REP16(M)
#undef M
            } else if (KIND == K_MED3) {
#define M(k) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));
                REP16(M)
#undef M
            } else if (KIND == K_CVTU) {
#define M(k) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[k]));
                REP16(M)
#undef M
            } else if (KIND == K_XOR) {
#define M(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[k]) : "v"(x));
                REP16(M)
#undef M
            } else if (KIND == K_MADU24) {
#define M(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));
                REP16(M)
#undef M
            } else if (KIND == K_MULU24) {
#define M(k) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[k]) : "v"(x));
                REP16(M)
#undef M
            } else if (KIND == K_ORSDWA) {
#define M(k) asm volatile("v_or_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(a[k]) : "v"(x));
                REP16(M)
#undef M
            } else if (KIND == K_CVTSDWA) {
#define M(k) asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "+v"(a[k]));
                REP16(M)
#undef M
            } else if (KIND == K_MIN3) {
#define M(k) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));
                REP16(M)
#undef M
            } else if (KIND == K_FMAMIX) {
                // src0 read as the f16 in the upper half of the register (op_sel_hi bit 0 = f16, op_sel bit 0 = high half), src1 / src2 as f32
#define M(k) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[k]) : "v"(x), "v"(y));
                REP16(M)
#undef M
            } else if (KIND == K_CVTF16) {
#define M(k) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a[k]));
                REP16(M)
#undef M
            } else if (KIND == K_MAD64) {
#define M(k) asm volatile("v_mad_i64_i32 %0, s[10:11], %1, %2, %0" : "+v"(q[k]) : "v"(x), "v"(y) : "s10", "s11");
                REP16(M)
#undef M
            } else if (KIND == K_LSHLADD64) {
#define M(k) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(q[k]) : "v"(q[(k + 1) & 15]));
                REP16(M)
#undef M
            } else if (KIND == K_LSHL64) {
#define M(k) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(q[k]));
                REP16(M)
#undef M
            } else if (KIND == K_MULLO) {
#define M(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(x));
                REP16(M)
#undef M
            } else if (KIND == K_MULHI) {
#define M(k) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[k]) : "v"(x));
                REP16(M)
#undef M
            } else if (KIND == K_ASHR) {
#define M(k) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(a[k]));
                REP16(M)
#undef M
            } else if (KIND == K_MIXFM) {
                // one issue port or two units side by side?  32 FMAs and 32 max per 64: 3.25 cycles per instruction if their times add, 2.1 if they overlap
#define M(k) if ((k) & 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(x)); else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(x), "v"(y));
                REP16(M)
#undef M
            } else if (KIND == K_MIXFC) {
#define M(k) if ((k) & 1) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[k])); else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(x), "v"(y));
                REP16(M)
#undef M
            } else if (KIND == K_LSHR) {
#define M(k) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[k]));
                REP16(M)
#undef M
            } else if (KIND == K_CVTFP8) {
#define M(k) asm volatile("v_cvt_pk_f32_fp8 %0, %1" : "+v"(q[k]) : "v"(x));
                REP16(M)
#undef M
            } else {
#define M(k) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));
                REP16(M)
#undef M
            }
        }
    }
    asm volatile("s_mov_b64 exec, %0" : : "s"(saved));
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; k++) s += a[k] + q[k].x + q[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int KIND>
static void run(int ncu, int waves_per_simd, int iters, unsigned long long mask, const char *mname, float *d_out, unsigned long long *d_st) {
    const int grid = ncu * waves_per_simd;
    size_t lds = (160 * 1024) / waves_per_simd - 512;
    if (lds > 64 * 1024) CHECK(hipFuncSetAttribute((const void *)exec_loop<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(exec_loop<KIND>, dim3(grid), dim3(256), lds, 0, d_out, d_st, iters / 8, 1.0001f, 0.5f, mask);   // warm-up
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(exec_loop<KIND>, dim3(grid), dim3(256), lds, 0, d_out, d_st, iters, 1.0001f, 0.5f, mask);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    std::vector<unsigned long long> st(2 * grid);
    CHECK(hipMemcpy(st.data(), d_st, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> clk;
    for (int b = 0; b < grid; b++) if (st[2 * b + 1]) clk.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 100e6);
    std::sort(clk.begin(), clk.end());
    const double clock_hz = clk.empty() ? 0 : clk[clk.size() / 2];        // shader clock inside the loop (s_memtime over the 100 MHz real-time counter)
    const double insts_per_simd = (double)iters * 64.0 * waves_per_simd;
    const double rate = insts_per_simd / (ms * 1e-3);                      // wave-instructions / s / SIMD by hipEvents (the whole launch)
    printf("%-30s W=%d  exec %-18s %8.3f ms  clock %.3f GHz  %.3f cycles/wave-inst/SIMD\n", kname[KIND], waves_per_simd, mname, ms, clock_hz / 1e9,
           clock_hz / rate);
}

int main() {
    const int iters = 10000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    float *d_out; unsigned long long *d_st;
    CHECK(hipMalloc(&d_out, (size_t)ncu * 8 * 256 * sizeof(float)));
    CHECK(hipMalloc(&d_st, (size_t)ncu * 8 * 2 * sizeof(unsigned long long)));
    printf("%s, %d CUs\n", prop.gcnArchName, ncu);
    struct { unsigned long long m; const char *n; } masks[] = {
        { ~0ull, "all 64" }, { 0xffffffffull, "low 32" }, { 0xffffull, "low 16" }, { 0x5555555555555555ull, "every other lane" },
        { 0x0000ffff0000ffffull, "0-15 and 32-47" }, { 1ull, "one lane" } };
    for (int w : { 4, 8 })
        for (auto &mk : masks) {
            run<K_FMA>(ncu, w, iters, mk.m, mk.n, d_out, d_st);
            run<K_MAX>(ncu, w, iters, mk.m, mk.n, d_out, d_st);
        }
    for (int w : { 4, 8 }) {
        run<K_CVTUB>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_CVTSDWA>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_CVTU>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_OR>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_ORSDWA>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_AND>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_XOR>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_LSHL>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_SUB>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_MED3>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_MIN3>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_ADD3>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_MADU24>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_MULU24>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_FMAMIX>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_CVTF16>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_CVTFP8>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_MAD64>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_LSHLADD64>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_LSHL64>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_MULLO>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_MULHI>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_ASHR>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_LSHR>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_MIXFM>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
        run<K_MIXFC>(ncu, w, iters, ~0ull, "all 64", d_out, d_st);
    }
    return 0;
}
