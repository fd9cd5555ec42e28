// Shared declarations for the gfx950 denoiser kernels (internal; the public C ABI is
// include/mi355_denoiser.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MI355_OK 0
#define MI355_ERR_ARG (-1)
#define MI355_ERR_LAUNCH (-2)
#define MI355_ERR_UNSUPPORTED (-3)

// Records a message retrievable through mi355_last_error() (thread-local) and returns `code`.
// run-time tuning switches of the launchers (mi355_tune_set; measurement aid, all zero by default)
extern int mi355_g_tune[16];
int mi355_fail(int code, const char* fmt, ...);
// Checks hipGetLastError() after a launch.
int mi355_check_launch(const char* what);

// Optional per-family launch timing (HIP events recorded on the launch stream).
enum { FAM_GEMM = 0, FAM_WGRAD = 1, FAM_ATTN = 2, FAM_NORM = 3, FAM_ELEM = 4, FAM_COUNT = 5 };
void mi355_prof_begin(int fam, hipStream_t s, double work);  // work = flops (contractions) or bytes
void mi355_prof_end(int fam, hipStream_t s);

struct ProfScope {
    int fam; hipStream_t s;
    ProfScope(int f, hipStream_t st, double work) : fam(f), s(st) { mi355_prof_begin(f, st, work); }
    ~ProfScope() { mi355_prof_end(fam, s); }
};

__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }

// erf to ~2e-7 absolute (Abramowitz & Stegun 7.1.26) in ~12 VALU operations plus one reciprocal and one exponential;
// the library erff costs several times that, which made the GELU gate the dominant part of the fused GEGLU epilogue.
__device__ __forceinline__ float fast_erff(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(ax * ax * -1.4426950408889634f);
    return copysignf(fmaf(-p * t, e, 1.0f), x);
}
// erf-form GELU (torch.nn.functional.gelu default) and its derivative
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.f + fast_erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad_f(float x) {
    return 0.5f * (1.f + fast_erff(x * 0.70710678118654752f)) +
           x * 0.3989422804014327f * __builtin_amdgcn_exp2f(x * x * -0.7213475204444817f);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Sum over the whole workgroup (blockDim.x multiple of 64, <= 1024). `red` holds >= 16 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// out[s][n] = sum_r partial[s][r][n]; launch with grid (ceil(N/64), nseg), 256 threads (64 columns x 4 row phases).
// (static: every translation unit that launches it carries its own copy; no relocatable device code needed)
// Zero fill by a kernel.  hipMemsetAsync must not be used on a path that can be captured into a hipGraph:
// on this stack a captured memset node zeroes correctly on the first replay only (later replays leave garbage),
// which silently corrupts anything that accumulates into the buffer.
static __global__ void zero_fill_kernel(char* __restrict__ p, size_t bytes) {
    const size_t n16 = bytes >> 4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        *(f32x4*)(p + (i << 4)) = z;
    if (blockIdx.x == 0 && threadIdx.x < (bytes & 15)) p[(n16 << 4) + threadIdx.x] = 0;
}
// Zero many ranges in one launch: table[2i] = device address (16-byte aligned), table[2i+1] = bytes (multiple of 16).
// grid = (blocks per range, ranges).
static __global__ void zero_ranges_kernel(const long long* __restrict__ table) {
    char* p = (char*)table[2 * blockIdx.y];
    const size_t n16 = (size_t)table[2 * blockIdx.y + 1] >> 4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        *(f32x4*)(p + (i << 4)) = z;
}
// p must be 16-byte aligned (every buffer of this library is).
static inline void mi355_zero_async(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return;
    size_t blocks = ((bytes >> 4) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (char*)p, bytes);
}

// out[s][col] = sum over r < R of partial[s][r][col].  With out_hi != nullptr, columns >= split are written to
// out_hi[col - split] instead (two destinations from one planar [R][2][C] partial buffer; single segment).
static __global__ void reduce_rows_kernel(const float* __restrict__ partial, float* __restrict__ out, int R, int N,
                                          float* __restrict__ out_hi = nullptr, int split = 0) {
    // 256 threads = 16 groups of 4 columns (one 16-byte load each) x 16 row phases; N (and split) multiples of 4
    __shared__ f32x4 red[16][16];
    const int cg = threadIdx.x & 15, ph = threadIdx.x >> 4, s = blockIdx.y;
    const int col = blockIdx.x * 64 + cg * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (col < N) {
        const float* src = partial + (size_t)s * R * N + col;
        int r = ph;
        for (; r + 48 < R; r += 64) {
            const f32x4 a = *(const f32x4*)(src + (size_t)r * N), b = *(const f32x4*)(src + (size_t)(r + 16) * N);
            const f32x4 c = *(const f32x4*)(src + (size_t)(r + 32) * N), d = *(const f32x4*)(src + (size_t)(r + 48) * N);
            acc += (a + b) + (c + d);
        }
        for (; r < R; r += 16) acc += *(const f32x4*)(src + (size_t)r * N);
    }
    red[ph][cg] = acc;
    __syncthreads();
    if (ph == 0 && col < N) {
        f32x4 t = red[0][cg];
#pragma unroll
        for (int q = 1; q < 16; ++q) t += red[q][cg];
        if (out_hi != nullptr && col >= split) *(f32x4*)(out_hi + col - split) = t;
        else *(f32x4*)(out + (size_t)s * N + col) = t;
    }
}

// Many row reductions in one launch: record e = {partial, out, out_hi, R, N, split} (int64), blockIdx.y picks the
// record, blockIdx.x the 64-column slab (slabs past N exit).  Same arithmetic as reduce_rows_kernel.
static __global__ void reduce_rows_batched_kernel(const long long* __restrict__ table) {
    const long long* e = table + 6 * blockIdx.y;
    const float* partial = (const float*)e[0];
    float* out = (float*)e[1];
    float* out_hi = (float*)e[2];
    const int R = (int)e[3], N = (int)e[4], split = (int)e[5];
    if ((int)blockIdx.x * 64 >= N) return;
    __shared__ f32x4 red[16][16];
    const int cg = threadIdx.x & 15, ph = threadIdx.x >> 4;
    const int col = blockIdx.x * 64 + cg * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (col < N) {
        const float* src = partial + col;
        int r = ph;
        for (; r + 48 < R; r += 64) {
            const f32x4 a = *(const f32x4*)(src + (size_t)r * N), b = *(const f32x4*)(src + (size_t)(r + 16) * N);
            const f32x4 c = *(const f32x4*)(src + (size_t)(r + 32) * N), d = *(const f32x4*)(src + (size_t)(r + 48) * N);
            acc += (a + b) + (c + d);
        }
        for (; r < R; r += 16) acc += *(const f32x4*)(src + (size_t)r * N);
    }
    red[ph][cg] = acc;
    __syncthreads();
    if (ph == 0 && col < N) {
        f32x4 t = red[0][cg];
#pragma unroll
        for (int q = 1; q < 16; ++q) t += red[q][cg];
        if (out_hi != nullptr && col >= split) *(f32x4*)(out_hi + col - split) = t;
        else *(f32x4*)(out + col) = t;
    }
}

// Blocks that share an XCD (ids congruent mod 8) get one contiguous chunk of the logical grid.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}
