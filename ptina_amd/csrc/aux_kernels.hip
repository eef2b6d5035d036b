// aux_kernels.hip -- the small kernels around the megakernel: Sobol state advance for a whole
// batch of frames, film combine / resolve / export.  All are streaming, HBM-bound passes over
// float4 records (dwordx4 per lane, consecutive lanes on consecutive addresses).

#include <hip/hip_runtime.h>
#include <algorithm>
#include "mpt_types.h"
#include "tri_records.h"
#include "film_ops.h"

// SobolSampler.update, sampling/sobol.py:99-105, for `count` consecutive frames in one launch:
// thread j owns dimension j, keeps X[j] in a register, and emits P[f][j] for every frame.
// count_low_bits(time) (sobol.py:11-17) is uniform, so it stays on the scalar unit.
__device__ __forceinline__ int count_low_bits(int i) {
    int bits = 1;
    int value = i;
    while (value & 1) { value >>= 1; bits += 1; }
    return bits;
}

// construct_float, sobol.py:20-29: MSB-first accumulation in f32 (exact for <= 24 significant
// bits; kept literal so that wider direction grids round exactly as the reference does)
__device__ __forceinline__ float construct_float(int i) {
    float ret = 0.0f;
    unsigned value = (unsigned)i;
    float term = 0.5f;
    while (value) {
        if (value & 0x80000000u) ret += term;
        value <<= 1;
        term *= 0.5f;
    }
    return ret;
}

__global__ __launch_bounds__(256) void sobol_update_kernel(const int *X, int *Xout, const int *__restrict__ V,
                                                           float *__restrict__ P, int dim, int rows, int time0,
                                                           int count, int pstride_frames, int write_x) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= dim) return;
    int x = X[j];
    // eight frames at a time: which row of V a frame takes depends on the frame number only, so the eight loads are issued
    // together and the kernel waits for memory four times per batch of 32 frames instead of 32 times (it runs beside the start of
    // a render launch, whose persistent workgroups need whole CUs: 35 us of it delayed a third of them -- DESIGN.md 3.6)
    for (int f0 = 0; f0 < count; f0 += 8) {
        int v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = count_low_bits(time0 + f0 + k);
            v[k] = (f0 + k < count && i < rows) ? V[(size_t)i * dim + j] : 0;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int f = f0 + k;
            if (f < count) {
                x ^= v[k];
                // only the last `pstride_frames` frames are kept when count exceeds the P capacity
                // (reset's skipped updates never read P)
                if (f >= count - pstride_frames) P[(size_t)(f - (count - pstride_frames)) * dim + j] = construct_float(x);
            }
        }
    }
    // Xout = X: the sampler's state moves.  Xout = a second buffer: the points of the NEXT batch computed ahead of time, and the
    // state that batch will leave behind with them -- the host swaps the two buffers when that batch is launched
    if (write_x) Xout[j] = x;
}

// film[pix] += sample[0][pix], then sample[1][pix], ... : the reference's frame-by-frame
// accumulation order (filmtable.py:37-39, path.py:93), one sample slab per frame of the batch.
// The slab holds only the columns of this context's share, packed: slab column cc is film column
// x0 + (cc / stripe_w) * stripe_pitch + cc % stripe_w (one slab: stripe_w = 2^30, so x0 + cc).
__global__ __launch_bounds__(256) void combine_kernel(MptVec4 *__restrict__ film, const MptVec4 *__restrict__ partial,
                                                      int ny, int x0, int x1, int stripe_w, int stripe_pitch,
                                                      int ccols, int nframes) {
    const size_t stride = (size_t)ccols * ny;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= stride) return;
    int cc = (int)(t / ny), y = (int)(t - (size_t)cc * ny);
    int s = cc / stripe_w;
    int x = x0 + s * stripe_pitch + (cc - s * stripe_w);
    if (x >= x1) return;                               // tile padding past the film's last column
    const size_t pix = (size_t)x * ny + y;
    MptVec4 a = film[pix];
    for (int c = 0; c < nframes; c++) {
        const mpt_u4 b = ((const mpt_u4 *)partial)[(size_t)c * stride + t];       // film_ops.h: {r, g.hi | tag}{b, g.lo | tag}
        film_add_sample(a, slab_r(b), slab_g(b), slab_b(b));
    }
    film[pix] = a;
}

// FilmTable._get_image, filmtable.py:53-63 : out[x][y] = rgb / w, w -> 1; empty -> (0.9, 0.4, 0.9, 0)
__global__ __launch_bounds__(256) void resolve_kernel(const MptVec4 *__restrict__ film, MptVec4 *__restrict__ out,
                                                      size_t npix) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npix) return;
    out[t] = film_resolve(film[t]);
}

// FilmTable.fast_export_image, filmtable.py:66-79 : flat RGB at (y * nx + x) * 3.
// One thread per OUTPUT pixel (consecutive x) so the 12-byte stores are contiguous across the wave.
__global__ __launch_bounds__(256) void export_kernel(const MptVec4 *__restrict__ film, float *__restrict__ out,
                                                     int nx, int ny) {
    size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= (size_t)nx * ny) return;
    int y = (int)(o / nx), x = (int)(o - (size_t)y * nx);
    MptVec4 v = film[(size_t)x * ny + y];
    if (v.w != 0.0f) {
        v.x /= v.w; v.y /= v.w; v.z /= v.w;
    } else {
        v.x = 0.9f; v.y = 0.4f; v.z = 0.9f;
    }
    out[o * 3 + 0] = v.x; out[o * 3 + 1] = v.y; out[o * 3 + 2] = v.z;
}

// diagnostics: a one-workgroup kernel with a chosen LDS footprint -- a stand-in for a collective's
// kernel when measuring how long such a kernel waits for a CU while persistent render workgroups
// hold all of them (mpt_probe_kernel)
__global__ void probe_kernel(double *out) {
    extern __shared__ int probe_lds[];
    probe_lds[threadIdx.x] = (int)threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (double)probe_lds[blockDim.x - 1];
}

// layout A/B (option "node_soa"): the 64-B node records transposed into four arrays of float4
__global__ __launch_bounds__(256) void transpose_nodes_kernel(const MptVec4 *__restrict__ in, MptVec4 *__restrict__ out, int ni) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)ni * 4) return;
    size_t i = t >> 2, k = t & 3;
    out[k * (size_t)ni + i] = in[t];
}

MPT_KERNEL_API hipError_t mpt_launch_transpose_nodes(const MptVec4 *in, MptVec4 *out, int ni, hipStream_t stream) {
    if (ni <= 0) return hipSuccess;
    hipLaunchKernelGGL(transpose_nodes_kernel, dim3((unsigned)(((size_t)ni * 4 + 255) / 256)), dim3(256), 0, stream, in, out, ni);
    return hipGetLastError();
}

MPT_KERNEL_API hipError_t mpt_launch_probe(double *out, int threads, size_t lds_bytes, hipStream_t stream) {
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(threads), lds_bytes, stream, out);
    return hipGetLastError();
}

MPT_KERNEL_API hipError_t mpt_launch_sobol_update(const int *X, int *Xout, const int *V, float *P, int dim, int rows, int time0,
                                              int count, int keep, int write_x, hipStream_t stream) {
    int grid = (dim + 255) / 256;
    hipLaunchKernelGGL(sobol_update_kernel, dim3(grid), dim3(256), 0, stream, X, Xout, V, P, dim, rows, time0, count, keep, write_x);
    return hipGetLastError();
}

MPT_KERNEL_API hipError_t mpt_launch_combine(MptVec4 *film, const MptVec4 *partial, int ny, int x0, int x1,
                                         int stripe_w, int stripe_pitch, int ccols, int nframes, hipStream_t stream) {
    size_t n = (size_t)ccols * ny;
    if (n == 0) return hipSuccess;
    int grid = (int)((n + 255) / 256);
    hipLaunchKernelGGL(combine_kernel, dim3(grid), dim3(256), 0, stream, film, partial, ny, x0, x1, stripe_w, stripe_pitch, ccols, nframes);
    return hipGetLastError();
}

MPT_KERNEL_API hipError_t mpt_launch_resolve(const MptVec4 *film, MptVec4 *out, size_t npix, hipStream_t stream) {
    if (npix == 0) return hipSuccess;
    int grid = (int)((npix + 255) / 256);
    hipLaunchKernelGGL(resolve_kernel, dim3(grid), dim3(256), 0, stream, film, out, npix);
    return hipGetLastError();
}

MPT_KERNEL_API hipError_t mpt_launch_export(const MptVec4 *film, float *out, int nx, int ny, hipStream_t stream) {
    size_t n = (size_t)nx * ny;
    if (n == 0) return hipSuccess;
    int grid = (int)((n + 255) / 256);
    hipLaunchKernelGGL(export_kernel, dim3(grid), dim3(256), 0, stream, film, out, nx, ny);
    return hipGetLastError();
}

// ---------------------------------------------------------------- film gather: pack / unpack by the comm plan
// A rank's share of a striped film is several column ranges (mpt_comm_plan); it travels as ONE contiguous
// message: copy_pieces packs the ranges side by side before the send, and on the root scatters every peer's
// received message back into the film -- the plan's piece table drives both (blockIdx.y = piece).
__global__ __launch_bounds__(256) void copy_pieces_kernel(const MptVec4 *__restrict__ src, MptVec4 *__restrict__ dst,
                                                          const MptPiece *__restrict__ tab) {
    const MptPiece pc = tab[blockIdx.y];
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < pc.count; t += (long long)gridDim.x * 256)
        dst[pc.dst + t] = src[pc.src + t];
}

MPT_KERNEL_API hipError_t mpt_launch_copy_pieces(const MptVec4 *src, MptVec4 *dst, const MptPiece *tab, int npieces,
                                             long long max_count, hipStream_t stream) {
    if (npieces <= 0 || max_count <= 0) return hipSuccess;
    const int gx = (int)std::min<long long>((max_count + 255) / 256, 1024);
    hipLaunchKernelGGL(copy_pieces_kernel, dim3(gx, npieces), dim3(256), 0, stream, src, dst, tab);
    return hipGetLastError();
}

// ---------------------------------------------------------------- production triangle records
// tfast from tgeo (mpt_types.h): the dual edge vectors of the reference's barycentric solve, geometries.py:134-143,
// in IEEE arithmetic (this file is compiled with -ffp-contract=off); a degenerate triangle (D = 0) gets
// infinities / NaNs, which fail every comparison of the test like the reference's own division by zero
__global__ __launch_bounds__(256) void derive_tfast_kernel(const MptVec4 *__restrict__ tgeo, MptVec4 *__restrict__ tfast, int n) {
    int slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= n) return;
    const MptVec4 g[4] = { tgeo[(size_t)slot * 4 + 0], tgeo[(size_t)slot * 4 + 1], tgeo[(size_t)slot * 4 + 2], tgeo[(size_t)slot * 4 + 3] };
    MptVec4 f[3];
    tri_make_tfast(g, f);
    tfast[(size_t)slot * 3 + 0] = f[0]; tfast[(size_t)slot * 3 + 1] = f[1]; tfast[(size_t)slot * 3 + 2] = f[2];
}

MPT_KERNEL_API hipError_t mpt_launch_derive_tfast(const MptVec4 *tgeo, MptVec4 *tfast, int n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(derive_tfast_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, tgeo, tfast, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------- triangle records in the 8-wide tree's leaf order
// tfast8[t] = tfast[perm[t]] (48 B), tshade8[t] = tshade[perm[t]] (64 B): the leaf children of an 8-wide node name their
// triangles by one base index (oct_build.cpp)
__global__ __launch_bounds__(256) void permute_tris_kernel(const MptVec4 *__restrict__ tfast, const MptVec4 *__restrict__ tshade,
                                                           const int32_t *__restrict__ perm, MptVec4 *__restrict__ tfast8,
                                                           MptVec4 *__restrict__ tshade8, int n) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const size_t s = (size_t)perm[t];
    for (int k = 0; k < 3; k++) tfast8[(size_t)t * 3 + k] = tfast[s * 3 + k];
    for (int k = 0; k < 4; k++) tshade8[(size_t)t * 4 + k] = tshade[s * 4 + k];
}

MPT_KERNEL_API hipError_t mpt_launch_permute_tris(const MptVec4 *tfast, const MptVec4 *tshade, const int32_t *perm, MptVec4 *tfast8,
                                              MptVec4 *tshade8, int n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(permute_tris_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, tfast, tshade, perm, tfast8, tshade8, n);
    return hipGetLastError();
}
