// What this GPU sustains on the two resources the path's kernels are priced against: a register-only loop of
// v_mfma_f32_16x16x32_bf16 (no memory traffic) and a 16-byte-per-lane streaming copy.  Measurement aids for the
// roofline figures; nothing in the product calls them.
#include "common.h"

// Each wave issues `iters` rounds of eight independent MFMAs (eight accumulator tiles hide the 8-pass latency).
// `pattern` 0: all-zero operands (the least the multipliers can toggle); 1: pseudo-random signs, mantissas and
// exponents, one operand element re-signed every round (the clock the chip holds depends on the data).
template <int PATTERN>
__global__ __launch_bounds__(256) void peak_mfma_kernel(float* __restrict__ sink, int iters) {
    bf16x8 a, b;
    const int lane = threadIdx.x & 63;
    // PATTERN 1: hashed sign, seven hashed mantissa bits and exponents 2^-4 .. 2^3 - the bit activity of random data
    unsigned h = 0x9E3779B9u * (unsigned)(threadIdx.x + 256 * blockIdx.x + 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        h = h * 1664525u + 1013904223u;
        const unsigned short ua = (unsigned short)(((h >> 16) & 0x807Fu) | ((123u + ((h >> 9) & 7u)) << 7));
        h = h * 1664525u + 1013904223u;
        const unsigned short ub = (unsigned short)(((h >> 16) & 0x807Fu) | ((123u + ((h >> 9) & 7u)) << 7));
        a[e] = PATTERN ? __builtin_bit_cast(bf16_t, ua) : (bf16_t)0.f;
        b[e] = PATTERN ? __builtin_bit_cast(bf16_t, ub) : (bf16_t)0.f;
    }
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it += 8) {          // eight rounds (64 MFMAs) per loop trip
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
            if (PATTERN) a[u] = (bf16_t)((float)a[u] * -1.f);        // new operand bits every round (one VALU op per 8 MFMAs)
        }
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += acc[i];
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) sink[0] = s[0];     // keeps the loop alive; practically never true
}

__global__ __launch_bounds__(256) void peak_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + u * 256 < n16) v[u] = __builtin_nontemporal_load(src + i + u * 256);
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + u * 256 < n16) __builtin_nontemporal_store(v[u], dst + i + u * 256);
    }
}

extern "C" {

// Launches the MFMA loop on `blocks` workgroups of four waves; flops = blocks * 4 * iters * 8 * 2*16*16*32.
int mi355_peak_mfma_bf16(float* sink, int blocks, int iters, int pattern, void* stream) {
    if (!sink || blocks <= 0 || iters <= 0 || (iters & 7)) return mi355_fail(MI355_ERR_ARG, "peak_mfma: iters must be a positive multiple of 8");
    if (pattern) hipLaunchKernelGGL(peak_mfma_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sink, iters);
    else hipLaunchKernelGGL(peak_mfma_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sink, iters);
    return mi355_check_launch("peak_mfma");
}

// Streaming copy of `bytes` (multiple of 16) from src to dst; traffic = 2 * bytes.
int mi355_peak_copy(const void* src, void* dst, long long bytes, void* stream) {
    if (!src || !dst || bytes <= 0 || (bytes & 15) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15))
        return mi355_fail(MI355_ERR_ARG, "peak_copy: pointers and size must be multiples of 16 bytes");
    const size_t n16 = (size_t)bytes >> 4;
    size_t grid = (n16 + 1023) / 1024;
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(peak_copy_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src, (f32x4*)dst, n16);
    return mi355_check_launch("peak_copy");
}

}  // extern "C"
