// GroupNorm(+SiLU) and LayerNorm, forward and backward, for NHWC / token-major bf16 activations.
//
// All kernels are HBM-bound; every global access is a 16-byte (8 x bf16) vector on a contiguous
// channel run, statistics are fp32.  GroupNorm works on the [B][HW][C] view: a thread owns one
// 8-channel chunk for a strip of pixels, so reads are fully coalesced rows regardless of the group
// width (C/32 is 10..80 channels in this network and is not a multiple of the vector width).
//
//   forward :  gn_partial<0> (per (b, pixel strip): sum x, sum x^2 per group)
//              ->  gn_apply  (prologue: strips -> mean, rstd per group;  y = silu?(x*scale + shift))
//   backward:  gn_partial<1> (per (b, strip): sum dz, sum dz*xhat per channel, and gamma-weighted per group)
//              ->  gn_bwd_apply (prologue: strips -> group means;  dx)   +   reduce_rows (dgamma, dbeta)
// No atomics and no stand-alone finalize kernels: the few hundred partial sums a block needs are re-reduced
// in its prologue, which is cheaper than a dependent launch.
#include "common.h"
#include <mutex>

__device__ __forceinline__ float silu_f(float z) { return z / (1.f + __expf(-z)); }
__device__ __forceinline__ float silu_grad_f(float z) {
    const float s = 1.f / (1.f + __expf(-z));
    return s * (1.f + z * (1.f - s));
}

// Group index of each of the 8 consecutive channels c0 .. c0 + 7: one integer division instead of eight (a group of at
// least 8 channels is crossed at most once inside the chunk).
__device__ __forceinline__ void gn_groups8(int c0, int cg, int (&gi)[8]) {
    const int g0 = c0 / cg, rem = c0 - g0 * cg;
    if (cg >= 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) gi[e] = g0 + (rem + e >= cg ? 1 : 0);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) gi[e] = (c0 + e) / cg;
    }
}
// The 8 consecutive fp32 parameters of a chunk as two 16-byte loads.
__device__ __forceinline__ void gn_load8(const float* __restrict__ p, float (&v)[8]) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}

// MODE 0: acc0 += x, acc1 += x^2.   MODE 1: dz = dy * act'(z); acc0 += dz, acc1 += dz * xhat.
// grid = (strips, B); blockDim.x = nc * rows_par (nc = C/8).
// gpart [B][strips][G][2]: per-group sums of this strip (MODE 1: weighted by gamma).
// cpart [B][strips][2][C]: per-channel sums (MODE 1 only; feeds dbeta / dgamma).
template <int MODE>
// x2 (optional, MODE 0): the input is the channel concatenation [x | x2] of two tensors, x holding the first C1 channels
// (rows of C1) and x2 the other C - C1 (rows of C - C1): the concatenated tensor is never materialised.
__global__ void gn_partial_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                  const float* __restrict__ stats, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, float* __restrict__ gpart, float* __restrict__ cpart,
                                  int HW, int C, int G, int rows_per_block, int silu,
                                  const bf16_t* __restrict__ x2 = nullptr, int C1 = 0) {
    const int nc = C >> 3;
    const int b = blockIdx.y;
    const int c8 = threadIdx.x % nc, rp = threadIdx.x / nc, rpar = blockDim.x / nc;
    const int r_begin = blockIdx.x * rows_per_block;
    int r_end = r_begin + rows_per_block;
    if (r_end > HW) r_end = HW;
    const int cg = C / G;
    float sc[8], sh[8], mu[8], rs[8];
    if (MODE == 1) {
        int gi[8];
        gn_groups8(c8 * 8, cg, gi);
        gn_load8(gamma + c8 * 8, sc);
        gn_load8(beta + c8 * 8, sh);
        if (cg >= 8) {                       // at most two groups in the chunk: two pairs of statistics
            const f32x2 sa = *(const f32x2*)(stats + ((size_t)b * G + gi[0]) * 2), sb = *(const f32x2*)(stats + ((size_t)b * G + gi[7]) * 2);
#pragma unroll
            for (int e = 0; e < 8; ++e) { mu[e] = gi[e] == gi[0] ? sa[0] : sb[0]; rs[e] = gi[e] == gi[0] ? sa[1] : sb[1]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { mu[e] = stats[((size_t)b * G + gi[e]) * 2]; rs[e] = stats[((size_t)b * G + gi[e]) * 2 + 1]; }
        }
    }
    float a0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const size_t base = (size_t)b * HW * C + c8 * 8;
    // this thread's chunk of every row: xs + (b * HW + r) * ldx
    int ldx = C;
    const bf16_t* xs = x + c8 * 8;
    if (x2 != nullptr) {
        if (c8 * 8 < C1) { ldx = C1; } else { xs = x2 + (c8 * 8 - C1); ldx = C - C1; }
    }
    xs += (size_t)b * HW * ldx;
    // four rows in flight per thread: the loads of a row are otherwise exposed in full on every iteration
    auto accumulate = [&](const bf16x8& xv, const bf16x8& dv) {
        if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float v = (float)xv[e]; a0[e] += v; a1[e] += v * v; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = ((float)xv[e] - mu[e]) * rs[e];
                float dz = (float)dv[e];
                if (silu) dz *= silu_grad_f(xh * sc[e] + sh[e]);
                a0[e] += dz; a1[e] += dz * xh;
            }
        }
    };
    int r = r_begin + rp;
    for (; r + 3 * rpar < r_end; r += 4 * rpar) {
        bf16x8 xv[4], dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xv[u] = *(const bf16x8*)(xs + (size_t)(r + u * rpar) * ldx);
            if (MODE == 1) dv[u] = *(const bf16x8*)(dy + base + (size_t)(r + u * rpar) * C);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) accumulate(xv[u], dv[u]);
    }
    for (; r < r_end; r += rpar) {
        const bf16x8 xv = *(const bf16x8*)(xs + (size_t)r * ldx);
        bf16x8 dv = xv;
        if (MODE == 1) dv = *(const bf16x8*)(dy + base + (size_t)r * C);
        accumulate(xv, dv);
    }
    // combine the row phases of this block through LDS
    extern __shared__ float red[];   // [rpar][nc][16]
    float* mine = red + ((size_t)rp * nc + c8) * 16;
#pragma unroll
    for (int e = 0; e < 8; ++e) { mine[e] = a0[e]; mine[8 + e] = a1[e]; }
    __syncthreads();
    const size_t slot = (size_t)b * gridDim.x + blockIdx.x;
    if (rp == 0) {
        for (int q = 1; q < rpar; ++q) {
            const float* o = red + ((size_t)q * nc + c8) * 16;
#pragma unroll
            for (int e = 0; e < 8; ++e) { a0[e] += o[e]; a1[e] += o[8 + e]; }
        }
        if (MODE == 1) {
            float* dst = cpart + slot * C * 2 + c8 * 8;          // planar: [C] sums of dz, then [C] sums of dz*xhat
            *(f32x4*)dst = f32x4{a0[0], a0[1], a0[2], a0[3]};
            *(f32x4*)(dst + 4) = f32x4{a0[4], a0[5], a0[6], a0[7]};
            *(f32x4*)(dst + C) = f32x4{a1[0], a1[1], a1[2], a1[3]};
            *(f32x4*)(dst + C + 4) = f32x4{a1[4], a1[5], a1[6], a1[7]};
#pragma unroll
            for (int e = 0; e < 8; ++e) { a0[e] *= sc[e]; a1[e] *= sc[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { mine[e] = a0[e]; mine[8 + e] = a1[e]; }     // rp == 0: mine = red[c8][16]
    }
    __syncthreads();
    // channels -> groups (a group is C/G channels, not a multiple of the 8-channel chunk in general)
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        float s0 = 0.f, s1 = 0.f;
        for (int c = g * cg; c < (g + 1) * cg; ++c) { s0 += red[(c >> 3) * 16 + (c & 7)]; s1 += red[(c >> 3) * 16 + 8 + (c & 7)]; }
        gpart[(slot * G + g) * 2] = s0;
        gpart[(slot * G + g) * 2 + 1] = s1;
    }
}

// Sums the strips' group partials of sample b into LDS: gs[g] = {sum0, sum1}.  Every thread of the block calls it.
__device__ __forceinline__ void gn_group_sums(const float* __restrict__ gpart, float* gs, int b, int strips, int G) {
    for (int i = threadIdx.x; i < G * 2; i += blockDim.x) {
        const float* src = gpart + (size_t)b * strips * G * 2 + i;
        float s = 0.f;
        // 16 independent loads per round trip: left as a plain loop the compiler waits for every load before the next
        // (one L2 round trip per strip, ~10 us of prologue in every block of a 32-strip map)
        for (int t0 = 0; t0 < strips; t0 += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = t0 + u < strips ? src[(size_t)(t0 + u) * G * 2] : 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
        gs[i] = s;
    }
    __syncthreads();
}

// Statistics that the producer(s) of x emitted from their own epilogues (mi355_conv2d_gnstats_bf16 /
// mi355_gemm_gnstats_bf16): per row tile and run of `unit` adjacent channels the sum and the sum of squares.  p1 covers
// the first n1 runs of the channel axis (all of it, or the first source of a two-source input), p2 the other n2;
// a sample has t1 (t2) row tiles: p[(b * t + tile) * n + run][2].
struct GnPre {
    const float* p1;
    const float* p2;
    int t1, t2, n1, n2, unit;
};
// gs[g] = {sum, sum of squares} of group g of sample b from those records (a group is cg / unit whole runs).
__device__ __forceinline__ void gn_group_sums_pre(const GnPre& q, float* gs, int b, int G, int cg) {
    const int per = cg / q.unit;
    for (int i = threadIdx.x; i < G * 2; i += blockDim.x) {
        const int g = i >> 1, k = i & 1;
        float s = 0.f;
        for (int r = g * per; r < (g + 1) * per; ++r) {
            const bool first = r < q.n1;
            const float* src = first ? q.p1 + ((size_t)b * q.t1 * q.n1 + r) * 2 + k : q.p2 + ((size_t)b * q.t2 * q.n2 + (r - q.n1)) * 2 + k;
            const int t = first ? q.t1 : q.t2, n = first ? q.n1 : q.n2;
            for (int t0 = 0; t0 < t; t0 += 8) {            // eight independent loads per round trip
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = t0 + u < t ? src[(size_t)(t0 + u) * n * 2] : 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
        }
        gs[i] = s;
    }
    __syncthreads();
}

__global__ void gn_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gpart, float* __restrict__ stats,
                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                bf16_t* __restrict__ y, int HW, int C, int G, int rows_per_block, int silu, float eps,
                                const bf16_t* __restrict__ x2 = nullptr, int C1 = 0, const GnPre pre = GnPre{}) {
    extern __shared__ float gs[];            // [G][2] -> mean, rstd
    const int nc = C >> 3;
    const int b = blockIdx.y;
    const int cg = C / G;
    if (pre.p1 != nullptr) gn_group_sums_pre(pre, gs, b, G, cg);
    else gn_group_sums(gpart, gs, b, gridDim.x, G);
    const float inv_n = 1.f / ((float)HW * cg);
    float sc[8], sh[8];
    const int c8 = threadIdx.x % nc, rp = threadIdx.x / nc, rpar = blockDim.x / nc;
    {
        int gi[8];
        float ga[8], be[8];
        gn_groups8(c8 * 8, cg, gi);
        gn_load8(gamma + c8 * 8, ga);
        gn_load8(beta + c8 * 8, be);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = gi[e];
            const float mean = gs[g * 2] * inv_n;
            float var = gs[g * 2 + 1] * inv_n - mean * mean;
            if (var < 0.f) var = 0.f;
            const float rstd = rsqrtf(var + eps);
            sc[e] = rstd * ga[e];
            sh[e] = be[e] - mean * sc[e];
        }
    }
    if (blockIdx.x == 0) {                   // one block per sample publishes the statistics for the backward pass
        for (int g = threadIdx.x; g < G; g += blockDim.x) {
            const float mean = gs[g * 2] * inv_n;
            float var = gs[g * 2 + 1] * inv_n - mean * mean;
            if (var < 0.f) var = 0.f;
            stats[((size_t)b * G + g) * 2] = mean;
            stats[((size_t)b * G + g) * 2 + 1] = rsqrtf(var + eps);
        }
    }
    const int r_begin = blockIdx.x * rows_per_block;
    int r_end = r_begin + rows_per_block;
    if (r_end > HW) r_end = HW;
    const size_t base = (size_t)b * HW * C + c8 * 8;
    int ldx = C;                             // two-source input: see gn_partial_kernel
    const bf16_t* xs = x + c8 * 8;
    if (x2 != nullptr) {
        if (c8 * 8 < C1) { ldx = C1; } else { xs = x2 + (c8 * 8 - C1); ldx = C - C1; }
    }
    xs += (size_t)b * HW * ldx;
    auto apply = [&](const bf16x8& xv, int r) {
        bf16x8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float z = (float)xv[e] * sc[e] + sh[e];
            if (silu) z = silu_f(z);
            ov[e] = (bf16_t)z;
        }
        *(bf16x8*)(y + base + (size_t)r * C) = ov;
    };
    int r = r_begin + rp;
    for (; r + 3 * rpar < r_end; r += 4 * rpar) {
        bf16x8 xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[u] = *(const bf16x8*)(xs + (size_t)(r + u * rpar) * ldx);
#pragma unroll
        for (int u = 0; u < 4; ++u) apply(xv[u], r + u * rpar);
    }
    for (; r < r_end; r += rpar) apply(*(const bf16x8*)(xs + (size_t)r * ldx), r);
}

// SUMS: the block also leaves the column sums of the dx rows it wrote in dxsum [B][strips][C] (the values as stored, i.e.
// rounded to bf16): per sample they are the gradient of a per-sample bias that the producer of x added (the
// timestep-embedding projection in the first convolution of a residual block), which then needs no pass of its own.
template <bool SUMS = false>
__global__ void gn_bwd_apply_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                    const float* __restrict__ stats, const float* __restrict__ gpart,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    bf16_t* __restrict__ dx, int HW, int C, int G, int rows_per_block, int silu,
                                    const bf16_t* __restrict__ dres, const bf16_t* __restrict__ x2 = nullptr, int C1 = 0,
                                    bf16_t* __restrict__ dx2 = nullptr, const bf16_t* __restrict__ dres2 = nullptr,
                                    float* __restrict__ dxsum = nullptr) {
    extern __shared__ float gs[];            // [G][2] = {sum_c gamma*sum dz, sum_c gamma*sum dz*xhat} (+ SUMS: [threads][8])
    const int nc = C >> 3;
    const int b = blockIdx.y;
    gn_group_sums(gpart, gs, b, gridDim.x, G);
    const int c8 = threadIdx.x % nc, rp = threadIdx.x / nc, rpar = blockDim.x / nc;
    const int r_begin = blockIdx.x * rows_per_block;
    int r_end = r_begin + rows_per_block;
    if (r_end > HW) r_end = HW;
    const int cg = C / G;
    const float inv_n = 1.f / ((float)HW * cg);
    float mu[8], rs[8], ga[8], be[8], m1[8], m2[8];
    {
        int gi[8];
        gn_groups8(c8 * 8, cg, gi);
        gn_load8(gamma + c8 * 8, ga);
        gn_load8(beta + c8 * 8, be);
        if (cg >= 8) {                       // at most two groups in the chunk: two pairs of statistics
            const f32x2 sa = *(const f32x2*)(stats + ((size_t)b * G + gi[0]) * 2), sb = *(const f32x2*)(stats + ((size_t)b * G + gi[7]) * 2);
#pragma unroll
            for (int e = 0; e < 8; ++e) { mu[e] = gi[e] == gi[0] ? sa[0] : sb[0]; rs[e] = gi[e] == gi[0] ? sa[1] : sb[1]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { mu[e] = stats[((size_t)b * G + gi[e]) * 2]; rs[e] = stats[((size_t)b * G + gi[e]) * 2 + 1]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { m1[e] = gs[gi[e] * 2] * inv_n; m2[e] = gs[gi[e] * 2 + 1] * inv_n; }
    }
    const size_t base = (size_t)b * HW * C + c8 * 8;
    // this thread's chunk of every row of x / dres / dx: p + (b * HW + r) * ldx.  Two-source input (see gn_partial_kernel):
    // the chunk lies in x (rows of C1) or in x2 (rows of C - C1), and so do its gradient and its bypass gradient.
    int ldx = C;
    size_t off = (size_t)c8 * 8;
    const bf16_t* xs = x;
    const bf16_t* rsrc = dres;
    bf16_t* ds = dx;
    if (x2 != nullptr) {
        if (c8 * 8 < C1) { ldx = C1; } else { xs = x2; rsrc = dres2; ds = dx2; off = (size_t)(c8 * 8 - C1); ldx = C - C1; }
    }
    off += (size_t)b * HW * ldx;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto apply = [&](const bf16x8& xv, const bf16x8& dv, int r) {
        bf16x8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xh = ((float)xv[e] - mu[e]) * rs[e];
            float dz = (float)dv[e];
            if (silu) dz *= silu_grad_f(xh * ga[e] + be[e]);
            ov[e] = (bf16_t)(rs[e] * (ga[e] * dz - m1[e] - xh * m2[e]));
        }
        if (rsrc) {          // gradient of the branch that bypasses the normalisation (residual / shortcut input)
            const bf16x8 rv = *(const bf16x8*)(rsrc + off + (size_t)r * ldx);
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = (bf16_t)((float)ov[e] + (float)rv[e]);
        }
        if (SUMS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) cs[e] += (float)ov[e];
        }
        *(bf16x8*)(ds + off + (size_t)r * ldx) = ov;
    };
    int r = r_begin + rp;
    for (; r + 3 * rpar < r_end; r += 4 * rpar) {
        bf16x8 xv[4], dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xv[u] = *(const bf16x8*)(xs + off + (size_t)(r + u * rpar) * ldx);
            dv[u] = *(const bf16x8*)(dy + base + (size_t)(r + u * rpar) * C);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) apply(xv[u], dv[u], r + u * rpar);
    }
    for (; r < r_end; r += rpar)
        apply(*(const bf16x8*)(xs + off + (size_t)r * ldx), *(const bf16x8*)(dy + base + (size_t)r * C), r);
    if (SUMS) {                              // row phases of the block combined through LDS, one partial row per block
        float* red = gs + 2 * G;
        float* mine = red + ((size_t)rp * nc + c8) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) mine[e] = cs[e];
        __syncthreads();
        if (rp == 0) {
            for (int q = 1; q < rpar; ++q) {
                const float* o = red + ((size_t)q * nc + c8) * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) cs[e] += o[e];
            }
            float* dst = dxsum + ((size_t)b * gridDim.x + blockIdx.x) * C + c8 * 8;
            *(f32x4*)dst = f32x4{cs[0], cs[1], cs[2], cs[3]};
            *(f32x4*)(dst + 4) = f32x4{cs[4], cs[5], cs[6], cs[7]};
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Small maps (8x8 and 4x4 levels, group width a multiple of 8 channels): one workgroup per (group, sample) holds its
// whole slab - at most 768 chunks of 8 channels - in registers, so the forward is ONE kernel that reads x once and
// the backward ONE kernel that reads x and dy once.  The strip kernels above need two launches each way and are
// latency-bound on these 1-5 MB maps (7 + 6 us forward, 12 + 9 us backward against ~5 and ~7 us here).
// ---------------------------------------------------------------------------------------------
constexpr int GNS_MAXCH = 3;        // backward: slab <= 768 chunks (registers: dz, xhat, gamma per element)
constexpr int GNS_MAXCH_FWD = 5;    // forward only keeps x: slabs up to 1280 chunks (1280 channels at 16x16)
static bool gn_small(int HW, int C, int G) {
    const int cg = C / G;
    return C % G == 0 && cg % 8 == 0 && HW <= 64 && HW * (cg / 8) <= 256 * GNS_MAXCH;
}
static bool gn_small_fwd(int HW, int C, int G) {
    const int cg = C / G;
    return C % G == 0 && cg % 8 == 0 && HW <= 256 && HW * (cg / 8) <= 256 * GNS_MAXCH_FWD;
}

__global__ __launch_bounds__(256) void gn_small_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                           float* __restrict__ stats, int HW, int C, int G, int silu, float eps,
                                                           const bf16_t* __restrict__ x2 = nullptr, int C1 = 0) {
    __shared__ float red[2][16];
    const int g = blockIdx.x, b = blockIdx.y, cg = C / G, ncc = cg >> 3, n = HW * ncc;
    const size_t base = (size_t)b * HW * C + (size_t)g * cg;
    float v[GNS_MAXCH_FWD][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < GNS_MAXCH_FWD; ++k) {
        const int idx = threadIdx.x + 256 * k;
        if (idx < n) {
            const int row = idx / ncc, cc = idx - row * ncc;
            const int c = g * cg + cc * 8;
            const bf16_t* src = x + base + (size_t)row * C + cc * 8;
            if (x2 != nullptr)                   // two-source input: see gn_partial_kernel
                src = c < C1 ? x + ((size_t)b * HW + row) * C1 + c : x2 + ((size_t)b * HW + row) * (C - C1) + (c - C1);
            const bf16x8 t = *(const bf16x8*)src;
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[k][e] = (float)t[e]; s += v[k][e]; }
        }
    }
    const float inv_n = 1.f / (float)(HW * cg);
    const float mean = block_sum(s, red[0]) * inv_n;
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < GNS_MAXCH_FWD; ++k) {
        if (threadIdx.x + 256 * k < n) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[k][e] - mean; ss += d * d; }
        }
    }
    const float rstd = rsqrtf(block_sum(ss, red[1]) * inv_n + eps);
    if (threadIdx.x == 0) {
        stats[((size_t)b * G + g) * 2] = mean;
        stats[((size_t)b * G + g) * 2 + 1] = rstd;
    }
#pragma unroll
    for (int k = 0; k < GNS_MAXCH_FWD; ++k) {
        const int idx = threadIdx.x + 256 * k;
        if (idx < n) {
            const int row = idx / ncc, cc = idx - row * ncc;
            const float* ga = gamma + g * cg + cc * 8;
            const float* be = beta + g * cg + cc * 8;
            const f32x4 g0 = *(const f32x4*)ga, g1 = *(const f32x4*)(ga + 4), b0 = *(const f32x4*)be, b1 = *(const f32x4*)(be + 4);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float z = (v[k][e] - mean) * rstd * (e < 4 ? g0[e] : g1[e - 4]) + (e < 4 ? b0[e] : b1[e - 4]);
                if (silu) z = silu_f(z);
                o[e] = (bf16_t)z;
            }
            *(bf16x8*)(y + base + (size_t)row * C + cc * 8) = o;
        }
    }
}

// cpart (optional): [B][2][C] per-sample channel sums (first plane sum dz, second sum dz * xhat) for dbeta / dgamma.
__global__ __launch_bounds__(256) void gn_small_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                           const float* __restrict__ stats, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, bf16_t* __restrict__ dx,
                                                           float* __restrict__ cpart, const bf16_t* __restrict__ dres,
                                                           int HW, int C, int G, int silu,
                                                           const bf16_t* __restrict__ x2 = nullptr, int C1 = 0,
                                                           bf16_t* __restrict__ dx2 = nullptr,
                                                           const bf16_t* __restrict__ dres2 = nullptr,
                                                           float* __restrict__ dxsum = nullptr) {
    // dxsum (optional): [B][C] column sums of the sample's dx rows as stored (see gn_bwd_apply_kernel<true>)
    extern __shared__ float pc[];            // [chunk][16]: 8 x dz, 8 x dz*xhat per chunk (parameter gradients only);
                                             // then [chunk][8]: the dx values (dxsum only)
    __shared__ float red[2][16];
    const int g = blockIdx.x, b = blockIdx.y, cg = C / G, ncc = cg >> 3, n = HW * ncc;
    const size_t base = (size_t)b * HW * C + (size_t)g * cg;
    const float mean = stats[((size_t)b * G + g) * 2], rstd = stats[((size_t)b * G + g) * 2 + 1];
    float dz[GNS_MAXCH][8], xh[GNS_MAXCH][8], gw[GNS_MAXCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < GNS_MAXCH; ++k) {
        const int idx = threadIdx.x + 256 * k;
        if (idx < n) {
            const int row = idx / ncc, cc = idx - row * ncc;
            const bf16_t* xsrc = x + base + (size_t)row * C + cc * 8;
            if (x2 != nullptr) {                 // two-source input: see gn_partial_kernel
                const int c = g * cg + cc * 8;
                xsrc = c < C1 ? x + ((size_t)b * HW + row) * C1 + c : x2 + ((size_t)b * HW + row) * (C - C1) + (c - C1);
            }
            const bf16x8 xv = *(const bf16x8*)xsrc;
            const bf16x8 dv = *(const bf16x8*)(dy + base + (size_t)row * C + cc * 8);
            const float* ga = gamma + g * cg + cc * 8;
            const float* be = beta + g * cg + cc * 8;
            const f32x4 g0 = *(const f32x4*)ga, g1 = *(const f32x4*)(ga + 4), b0 = *(const f32x4*)be, b1 = *(const f32x4*)(be + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float gam = e < 4 ? g0[e] : g1[e - 4];
                const float h = ((float)xv[e] - mean) * rstd;
                float d = (float)dv[e];
                if (silu) d *= silu_grad_f(h * gam + (e < 4 ? b0[e] : b1[e - 4]));
                dz[k][e] = d; xh[k][e] = h; gw[k][e] = gam;
                s1 += gam * d;
                s2 += gam * d * h;
            }
            if (cpart) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { pc[idx * 16 + e] = dz[k][e]; pc[idx * 16 + 8 + e] = dz[k][e] * xh[k][e]; }
            }
        }
    }
    const float inv_n = 1.f / (float)(HW * cg);
    const float m1 = block_sum(s1, red[0]) * inv_n;       // (its barriers also publish pc)
    const float m2 = block_sum(s2, red[1]) * inv_n;
#pragma unroll
    for (int k = 0; k < GNS_MAXCH; ++k) {
        const int idx = threadIdx.x + 256 * k;
        if (idx < n) {
            const int row = idx / ncc, cc = idx - row * ncc;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(rstd * (gw[k][e] * dz[k][e] - m1 - xh[k][e] * m2));
            size_t doff = base + (size_t)row * C + cc * 8;
            const bf16_t* rsrc = dres;
            bf16_t* ddst = dx;
            if (x2 != nullptr) {
                const int c = g * cg + cc * 8;
                if (c < C1) doff = ((size_t)b * HW + row) * C1 + c;
                else { doff = ((size_t)b * HW + row) * (C - C1) + (c - C1); rsrc = dres2; ddst = dx2; }
            }
            if (rsrc) {
                const bf16x8 rv = *(const bf16x8*)(rsrc + doff);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)o[e] + (float)rv[e]);
            }
            if (dxsum) {
                float* q = pc + (size_t)n * 16 + (size_t)idx * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) q[e] = (float)o[e];
            }
            *(bf16x8*)(ddst + doff) = o;
        }
    }
    if (dxsum) {
        __syncthreads();
        for (int t = threadIdx.x; t < cg; t += 256) {
            const int cc = t >> 3, e = t & 7;
            float acc = 0.f;
            for (int r = 0; r < HW; ++r) acc += pc[(size_t)n * 16 + (size_t)(r * ncc + cc) * 8 + e];
            dxsum[(size_t)b * C + g * cg + t] = acc;
        }
    }
    if (cpart) {
        for (int t = threadIdx.x; t < 2 * cg; t += 256) {
            const int which = t / cg, ch = t - which * cg, cc = ch >> 3, e = ch & 7;
            float acc = 0.f;
            for (int r = 0; r < HW; ++r) acc += pc[(r * ncc + cc) * 16 + which * 8 + e];
            cpart[(size_t)b * 2 * C + (size_t)which * C + g * cg + ch] = acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row kept in registers (C <= 8*64*MAXCH).
// ---------------------------------------------------------------------------------------------
constexpr int LN_MAXCH = 4;   // chunks of 8 per lane -> C <= 2048

__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                     float* __restrict__ stats, int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int nc = C >> 3;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < M; row += gridDim.x * wpb) {
        float v[LN_MAXCH][8];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < LN_MAXCH; ++k) {
            const int c8 = lane + 64 * k;
            if (c8 < nc) {
                const bf16x8 t = *(const bf16x8*)(x + (size_t)row * C + c8 * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[k][e] = (float)t[e]; s += v[k][e]; }
            }
        }
        const float mean = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < LN_MAXCH; ++k) {
            if (lane + 64 * k < nc) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[k][e] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / C + eps);
        if (lane == 0 && stats) { stats[(size_t)row * 2] = mean; stats[(size_t)row * 2 + 1] = rstd; }
#pragma unroll
        for (int k = 0; k < LN_MAXCH; ++k) {
            const int c8 = lane + 64 * k;
            if (c8 < nc) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    o[e] = (bf16_t)((v[k][e] - mean) * rstd * gamma[c8 * 8 + e] + beta[c8 * 8 + e]);
                *(bf16x8*)(y + (size_t)row * C + c8 * 8) = o;
            }
        }
    }
}

// dx per row; dgamma/dbeta partials kept in registers across the rows a wave visits, combined per block
// through LDS, one partial row per block.  NCH = 8-channel chunks per lane (C <= 512 * NCH).  1024-thread
// blocks (512 for the wide rows, whose per-lane state needs more registers): 8-16 waves per CU keep enough
// rows in flight to cover the load -> reduce -> store chain of each row.
template <int NCH>
__global__ __launch_bounds__(NCH <= 2 ? 1024 : 512) void ln_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                      const float* __restrict__ stats, const float* __restrict__ gamma,
                                                      bf16_t* __restrict__ dx, float* __restrict__ partial,
                                                      int M, int C, const bf16_t* __restrict__ dres) {
    extern __shared__ float lred[];      // [waves][2][C] per-wave sums (dgamma, dbeta), then [C] gamma
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    float* lgam = lred + (size_t)wpb * 2 * C;
    for (int i = threadIdx.x; i < C; i += blockDim.x) lgam[i] = gamma[i];
    __syncthreads();
    const int nc = C >> 3;
    float pg[NCH][8], pb[NCH][8];
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) { pg[k][e] = 0.f; pb[k][e] = 0.f; }
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < M; row += gridDim.x * wpb) {
        const float mean = stats[(size_t)row * 2], rstd = stats[(size_t)row * 2 + 1];
        bf16x8 xv[NCH], dv[NCH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c8 = lane + 64 * k;
            if (c8 < nc) {
                xv[k] = *(const bf16x8*)(x + (size_t)row * C + c8 * 8);
                dv[k] = *(const bf16x8*)(dy + (size_t)row * C + c8 * 8);
            }
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c8 = lane + 64 * k;
            if (c8 < nc) {
                const f32x4 g0 = *(const f32x4*)(lgam + c8 * 8), g1 = *(const f32x4*)(lgam + c8 * 8 + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = (float)dv[k][e];
                    const float xh = ((float)xv[k][e] - mean) * rstd;
                    const float dg = d * (e < 4 ? g0[e] : g1[e - 4]);
                    s1 += dg;
                    s2 += dg * xh;
                    pg[k][e] += d * xh;
                    pb[k][e] += d;
                }
            }
        }
        s1 = wave_sum(s1) / C;
        s2 = wave_sum(s2) / C;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c8 = lane + 64 * k;
            if (c8 < nc) {
                const f32x4 g0 = *(const f32x4*)(lgam + c8 * 8), g1 = *(const f32x4*)(lgam + c8 * 8 + 4);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = ((float)xv[k][e] - mean) * rstd;
                    const float dg = (float)dv[k][e] * (e < 4 ? g0[e] : g1[e - 4]);
                    o[e] = (bf16_t)(rstd * (dg - s1 - xh * s2));
                }
                if (dres) {     // gradient of the residual branch that bypasses the normalisation
                    const bf16x8 rv = *(const bf16x8*)(dres + (size_t)row * C + c8 * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)o[e] + (float)rv[e]);
                }
                *(bf16x8*)(dx + (size_t)row * C + c8 * 8) = o;
            }
        }
    }
    float* mine = lred + (size_t)(threadIdx.x >> 6) * 2 * C;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c8 = lane + 64 * k;
        if (c8 < nc) {
            *(f32x4*)(mine + c8 * 8) = f32x4{pg[k][0], pg[k][1], pg[k][2], pg[k][3]};
            *(f32x4*)(mine + c8 * 8 + 4) = f32x4{pg[k][4], pg[k][5], pg[k][6], pg[k][7]};
            *(f32x4*)(mine + C + c8 * 8) = f32x4{pb[k][0], pb[k][1], pb[k][2], pb[k][3]};
            *(f32x4*)(mine + C + c8 * 8 + 4) = f32x4{pb[k][4], pb[k][5], pb[k][6], pb[k][7]};
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        float sacc = 0.f;
        for (int q = 0; q < wpb; ++q) sacc += lred[(size_t)q * 2 * C + i];
        partial[(size_t)blockIdx.x * 2 * C + i] = sacc;
    }
}

static int gn_geometry(int HW, int C, int G, int* threads, int* rpar, int* rows_per_block, int* strips) {
    if (C % 8 || C % G) return mi355_fail(MI355_ERR_ARG, "groupnorm: C=%d must be a multiple of 8 and of G=%d", C, G);
    const int nc = C / 8;
    if (nc > 1024) return mi355_fail(MI355_ERR_UNSUPPORTED, "groupnorm: C=%d too large", C);
    int rp = 256 / nc;
    if (rp < 1) rp = 1;
    if (rp > HW) rp = HW;
    *rpar = rp;
    *threads = nc * rp;
    int rows = rp * 8;
    if (rows > HW) rows = HW;
    // every consumer block re-reduces the strips' group partials in its prologue: keep them few on large maps
    constexpr int MAX_STRIPS = 32;
    if ((HW + rows - 1) / rows > MAX_STRIPS) {
        rows = (HW + MAX_STRIPS - 1) / MAX_STRIPS;
        rows = (rows + rp - 1) / rp * rp;
    }
    *rows_per_block = rows;
    *strips = (HW + rows - 1) / rows;
    if (gn_small(HW, C, G)) { *rows_per_block = HW; *strips = 1; }      // one workgroup per (group, sample) holds the slab
    return 0;
}

// LayerNorm folded into the product behind it: W' = bf16(scale gamma (.) W), colsum[n] = sum_k W'[n][k] (of the rounded
// values the matrix pipe will multiply), bias'[n] = scale (sum_k beta[k] W[n][k] + bias[n]).  One workgroup per output row
// n.  scale: a factor on the layer's output folded in as well (the softmax scale of an attention's query projection).
__global__ __launch_bounds__(256) void ln_fold_weight_kernel(const float* __restrict__ W, int ldw, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ bias,
                                                             bf16_t* __restrict__ Wp, float* __restrict__ colsum,
                                                             float* __restrict__ bias_out, int K, float scale) {
    __shared__ float red[16];
    const int n = blockIdx.x;
    float cs = 0.f, bb = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float w = W[(size_t)n * ldw + k];
        const bf16_t wp = (bf16_t)(scale * gamma[k] * w);
        Wp[(size_t)n * K + k] = wp;
        cs += (float)wp;
        bb += beta[k] * w;
    }
    cs = block_sum(cs, red);
    __syncthreads();
    bb = block_sum(bb, red);
    if (threadIdx.x == 0) {
        colsum[n] = cs;
        bias_out[n] = scale * (bb + (bias ? bias[n] : 0.f));
    }
}

extern "C" {

// scratch floats needed by the two GroupNorm calls for a [B][HW][C] tensor with G groups.
long long mi355_groupnorm_scratch_floats(int B, int HW, int C, int G) {
    int threads, rpar, rows, strips;
    if (gn_geometry(HW, C, G, &threads, &rpar, &rows, &strips)) return -1;
    return (long long)B * strips * C * 2 + (long long)B * strips * G * 2;
}

static int gn_fwd_launch(const void* x, const void* x2, int C1, const float* gamma, const float* beta, void* y, float* stats,
                         float* scratch, int B, int HW, int C, int G, float eps, int silu, void* stream,
                         const GnPre* pre = nullptr) {
    int threads, rpar, rows, strips;
    if (int rc = gn_geometry(HW, C, G, &threads, &rpar, &rows, &strips)) return rc;
    hipStream_t st = (hipStream_t)stream;
    float* gpart = scratch;
    ProfScope prof(FAM_NORM, st, (double)B * HW * C * 2 * (pre ? 2 : 3));
    if (gn_small_fwd(HW, C, G)) {
        hipLaunchKernelGGL(gn_small_fwd_kernel, dim3(G, B), dim3(256), 0, st, (const bf16_t*)x, gamma, beta, (bf16_t*)y, stats,
                           HW, C, G, silu, eps, (const bf16_t*)x2, C1);
        return mi355_check_launch("groupnorm_fwd(small)");
    }
    if (pre) {          // the statistics exist already: the normalising pass alone
        hipLaunchKernelGGL(gn_apply_kernel, dim3(strips, B), dim3(threads), (size_t)G * 2 * sizeof(float), st, (const bf16_t*)x,
                           nullptr, stats, gamma, beta, (bf16_t*)y, HW, C, G, rows, silu, eps, (const bf16_t*)x2, C1, *pre);
        return mi355_check_launch("groupnorm_fwd(pre)");
    }
    const size_t sh = (size_t)threads * 16 * sizeof(float);
    hipLaunchKernelGGL(gn_partial_kernel<0>, dim3(strips, B), dim3(threads), sh, st, (const bf16_t*)x, nullptr,
                       nullptr, nullptr, nullptr, gpart, nullptr, HW, C, G, rows, 0, (const bf16_t*)x2, C1);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(strips, B), dim3(threads), (size_t)G * 2 * sizeof(float), st, (const bf16_t*)x,
                       gpart, stats, gamma, beta, (bf16_t*)y, HW, C, G, rows, silu, eps, (const bf16_t*)x2, C1, GnPre{});
    return mi355_check_launch("groupnorm_fwd");
}

// scratch: mi355_groupnorm_scratch_floats().  stats out: [B][G][2] = mean, rstd.
int mi355_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, float* scratch,
                        int B, int HW, int C, int G, float eps, int silu, void* stream) {
    if (!x || !gamma || !beta || !y || !stats || !scratch) return mi355_fail(MI355_ERR_ARG, "groupnorm_fwd: null pointer");
    return gn_fwd_launch(x, nullptr, 0, gamma, beta, y, stats, scratch, B, HW, C, G, eps, silu, stream);
}

// The same on the channel concatenation [x1 | x2] (x1: [B][HW][C1], x2: [B][HW][C2]) without materialising it: the input
// of the up-path residual blocks (hidden state | skip connection).  y: [B][HW][C1 + C2].  scratch:
// mi355_groupnorm_scratch_floats(B, HW, C1 + C2, G).
int mi355_groupnorm_fwd_cat(const void* x1, int C1, const void* x2, int C2, const float* gamma, const float* beta, void* y,
                            float* stats, float* scratch, int B, int HW, int G, float eps, int silu, void* stream) {
    if (!x1 || !x2 || !gamma || !beta || !y || !stats || !scratch) return mi355_fail(MI355_ERR_ARG, "groupnorm_fwd_cat: null pointer");
    if (C1 < 8 || C2 < 8 || (C1 & 7) || (C2 & 7)) return mi355_fail(MI355_ERR_ARG, "groupnorm_fwd_cat: C1 = %d, C2 = %d must be multiples of 8", C1, C2);
    return gn_fwd_launch(x1, x2, C1, gamma, beta, y, stats, scratch, B, HW, C1 + C2, G, eps, silu, stream);
}

// 1 when mi355_groupnorm_fwd of this geometry runs the statistics pass + the normalising pass (and
// mi355_groupnorm_fwd_pre can save the first), 0 when one workgroup per (group, sample) does both from registers.
int mi355_groupnorm_fwd_two_pass(int HW, int C, int G) { return (C % 8 == 0 && G > 0 && C % G == 0 && !gn_small_fwd(HW, C, G)) ? 1 : 0; }

// GroupNorm(+SiLU) forward from statistics that the producer(s) of the input emitted in their epilogues: no pass over x
// for the sums.  x2 / C1 as in mi355_groupnorm_fwd_cat (x2 NULL: one source).  pre1 [B * tiles1][n1][2] with n1 = C1 / unit
// (one source: C / unit) runs per row tile and tiles1 row tiles per sample; pre2 likewise for the second source.  Every
// group must be whole runs: (C / G) % unit == 0 and C1 % unit == 0.  Small maps (one workgroup per group and sample holds
// the slab in registers) have no use for them: MI355_ERR_UNSUPPORTED, call mi355_groupnorm_fwd.
int mi355_groupnorm_fwd_pre(const void* x, const void* x2, int C1, const float* gamma, const float* beta, void* y, float* stats,
                            const float* pre1, int tiles1, const float* pre2, int tiles2, int unit, int B, int HW, int C, int G,
                            float eps, int silu, void* stream) {
    if (!x || !gamma || !beta || !y || !stats || !pre1 || (x2 && !pre2)) return mi355_fail(MI355_ERR_ARG, "groupnorm_fwd_pre: null pointer");
    if (unit < 1 || tiles1 < 1 || (x2 && tiles2 < 1) || C % G || (C / G) % unit || (x2 && (C1 % unit || C1 < 8 || C1 >= C || (C1 & 7) || ((C - C1) & 7))))
        return mi355_fail(MI355_ERR_ARG, "groupnorm_fwd_pre: groups must be whole runs of %d channels (C=%d C1=%d G=%d)", unit, C, C1, G);
    if (gn_small_fwd(HW, C, G)) return mi355_fail(MI355_ERR_UNSUPPORTED, "groupnorm_fwd_pre: small map, use mi355_groupnorm_fwd");
    GnPre q;
    q.p1 = pre1; q.p2 = pre2; q.t1 = tiles1; q.t2 = x2 ? tiles2 : 0; q.unit = unit;
    q.n1 = (x2 ? C1 : C) / unit; q.n2 = x2 ? (C - C1) / unit : 0;
    return gn_fwd_launch(x, x2, x2 ? C1 : 0, gamma, beta, y, stats, nullptr, B, HW, C, G, eps, silu, stream, &q);
}

// GroupNorm forward for callers that cannot allocate (HIP graph capture, serving loops that forbid hipMalloc): the
// strips' group partials go to a file-scope device array instead of a caller-provided scratch buffer.  Each GPU holds its
// own copy of the array at its own address, looked up once per device; the kernels get that address as an ordinary
// argument.  Calls that overlap on one GPU (two streams) race on the array; the partials are overwritten, never
// accumulated, so nothing has to be cleared between calls.
constexpr long long GN_STATIC_FLOATS = 256 * 1024;              // 1 MiB per GPU, held until the process ends
static __device__ __align__(256) float gn_static_scratch[GN_STATIC_FLOATS];
constexpr int GN_MAX_DEVICES = 64;
static float* gn_static_ptr[GN_MAX_DEVICES];
static std::mutex gn_static_mutex;

// Device address of gn_static_scratch on the current device (nullptr + error message on failure).
static float* gn_static_address() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= GN_MAX_DEVICES) {
        mi355_fail(MI355_ERR_LAUNCH, "groupnorm_fwd_static: no current device (or device index >= %d)", GN_MAX_DEVICES);
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(gn_static_mutex);
    if (!gn_static_ptr[dev]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(gn_static_scratch));
        if (e != hipSuccess || !p) {
            mi355_fail(MI355_ERR_LAUNCH, "groupnorm_fwd_static: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
            return nullptr;
        }
        gn_static_ptr[dev] = (float*)p;
    }
    return gn_static_ptr[dev];
}

int mi355_groupnorm_fwd_static(const void* x, const float* gamma, const float* beta, void* y, float* stats, int B, int HW,
                               int C, int G, float eps, int silu, void* stream) {
    if (!x || !gamma || !beta || !y || !stats) return mi355_fail(MI355_ERR_ARG, "groupnorm_fwd_static: null pointer");
    if (B < 1 || HW < 1) return mi355_fail(MI355_ERR_ARG, "groupnorm_fwd_static: B=%d HW=%d", B, HW);
    int threads, rpar, rows, strips;
    if (int rc = gn_geometry(HW, C, G, &threads, &rpar, &rows, &strips)) return rc;
    const long long need = (long long)B * strips * G * 2;       // the forward's part of mi355_groupnorm_scratch_floats()
    if (need > GN_STATIC_FLOATS)
        return mi355_fail(MI355_ERR_UNSUPPORTED, "groupnorm_fwd_static: B=%d HW=%d G=%d needs %lld scratch floats, the built-in "
                          "buffer holds %lld (use mi355_groupnorm_fwd with a caller-provided scratch)", B, HW, G, need, GN_STATIC_FLOATS);
    float* scratch = gn_static_address();
    if (!scratch) return MI355_ERR_LAUNCH;
    return mi355_groupnorm_fwd(x, gamma, beta, y, stats, scratch, B, HW, C, G, eps, silu, stream);
}

// dx_colsum (optional, with colsum_scratch of mi355_groupnorm_scratch_floats() floats): [B][C] per-sample column sums of dx
static int gn_bwd_launch(const void* x, const void* x2, int C1, const void* dy, const float* stats, const float* gamma,
                         const float* beta, void* dx, void* dx2, float* dgamma, float* dbeta, float* scratch, int B, int HW, int C,
                         int G, int silu, const void* dres, const void* dres2, void* stream, float* dx_colsum = nullptr,
                         float* colsum_scratch = nullptr) {
    int threads, rpar, rows, strips;
    if (int rc = gn_geometry(HW, C, G, &threads, &rpar, &rows, &strips)) return rc;
    hipStream_t st = (hipStream_t)stream;
    float* gpart = scratch;
    float* cpart = scratch + (size_t)B * strips * G * 2;
    ProfScope prof(FAM_NORM, st, (double)B * HW * C * 2 * 5);
    const size_t shs_small = (size_t)HW * (C / G / 8) * (dx_colsum ? 24 : 16) * sizeof(float);
    if (gn_small(HW, C, G) && shs_small <= 64 * 1024) {
        const size_t shs = shs_small;
        hipLaunchKernelGGL(gn_small_bwd_kernel, dim3(G, B), dim3(256), shs, st, (const bf16_t*)x, (const bf16_t*)dy, stats, gamma,
                           beta, (bf16_t*)dx, cpart, (const bf16_t*)dres, HW, C, G, silu, (const bf16_t*)x2, C1, (bf16_t*)dx2,
                           (const bf16_t*)dres2, dx_colsum);
        if (dgamma)
            hipLaunchKernelGGL(reduce_rows_kernel, dim3((2 * C + 63) / 64, 1), dim3(256), 0, st, cpart, dbeta, B, 2 * C, dgamma, C);
        return mi355_check_launch("groupnorm_bwd(small)");
    }
    const size_t sh = (size_t)threads * 16 * sizeof(float);
    hipLaunchKernelGGL(gn_partial_kernel<1>, dim3(strips, B), dim3(threads), sh, st, (const bf16_t*)x,
                       (const bf16_t*)dy, stats, gamma, beta, gpart, cpart, HW, C, G, rows, silu, (const bf16_t*)x2, C1);
    if (dx_colsum) {
        hipLaunchKernelGGL(gn_bwd_apply_kernel<true>, dim3(strips, B), dim3(threads),
                           ((size_t)G * 2 + (size_t)threads * 8) * sizeof(float), st, (const bf16_t*)x, (const bf16_t*)dy, stats,
                           gpart, gamma, beta, (bf16_t*)dx, HW, C, G, rows, silu, (const bf16_t*)dres, (const bf16_t*)x2, C1,
                           (bf16_t*)dx2, (const bf16_t*)dres2, colsum_scratch);
        hipLaunchKernelGGL(reduce_rows_kernel, dim3((C + 63) / 64, B), dim3(256), 0, st, colsum_scratch, dx_colsum, strips, C);
    } else {
        hipLaunchKernelGGL(gn_bwd_apply_kernel<false>, dim3(strips, B), dim3(threads), (size_t)G * 2 * sizeof(float), st,
                           (const bf16_t*)x, (const bf16_t*)dy, stats, gpart, gamma, beta, (bf16_t*)dx, HW, C, G, rows, silu,
                           (const bf16_t*)dres, (const bf16_t*)x2, C1, (bf16_t*)dx2, (const bf16_t*)dres2);
    }
    // dbeta = column sums of the first plane, dgamma of the second (skipped for frozen affine parameters)
    if (dgamma)
        hipLaunchKernelGGL(reduce_rows_kernel, dim3((2 * C + 63) / 64, 1), dim3(256), 0, st, cpart, dbeta, B * strips, 2 * C, dgamma, C);
    return mi355_check_launch("groupnorm_bwd");
}

// scratch: mi355_groupnorm_scratch_floats().
int mi355_groupnorm_bwd(const void* x, const void* dy, const float* stats, const float* gamma, const float* beta,
                        void* dx, float* dgamma, float* dbeta, float* scratch, int B, int HW, int C, int G, int silu,
                        const void* dres, void* stream) {
    if (!x || !dy || !stats || !gamma || !beta || !dx || !scratch || (!dgamma != !dbeta))
        return mi355_fail(MI355_ERR_ARG, "groupnorm_bwd: null pointer");
    return gn_bwd_launch(x, nullptr, 0, dy, stats, gamma, beta, dx, nullptr, dgamma, dbeta, scratch, B, HW, C, G, silu, dres,
                         nullptr, stream);
}

// mi355_groupnorm_bwd that also returns dx_colsum [B][C] = per-sample column sums of the dx it stores (the gradient of a
// per-sample bias added by the producer of x); colsum_scratch: another mi355_groupnorm_scratch_floats() floats.
int mi355_groupnorm_bwd_colsum(const void* x, const void* dy, const float* stats, const float* gamma, const float* beta,
                               void* dx, float* dgamma, float* dbeta, float* scratch, int B, int HW, int C, int G, int silu,
                               const void* dres, float* dx_colsum, float* colsum_scratch, void* stream) {
    if (!x || !dy || !stats || !gamma || !beta || !dx || !scratch || (!dgamma != !dbeta) || !dx_colsum || !colsum_scratch)
        return mi355_fail(MI355_ERR_ARG, "groupnorm_bwd_colsum: null pointer");
    return gn_bwd_launch(x, nullptr, 0, dy, stats, gamma, beta, dx, nullptr, dgamma, dbeta, scratch, B, HW, C, G, silu, dres,
                         nullptr, stream, dx_colsum, colsum_scratch);
}

// Backward of mi355_groupnorm_fwd_cat: the input gradient leaves as two tensors shaped like x1 and x2.
int mi355_groupnorm_bwd_cat(const void* x1, int C1, const void* x2, int C2, const void* dy, const float* stats,
                            const float* gamma, const float* beta, void* dx1, void* dx2, float* dgamma, float* dbeta,
                            float* scratch, int B, int HW, int G, int silu, const void* dres1, const void* dres2, void* stream) {
    if (!x1 || !x2 || !dy || !stats || !gamma || !beta || !dx1 || !dx2 || !scratch || (!dgamma != !dbeta) || (!dres1 != !dres2))
        return mi355_fail(MI355_ERR_ARG, "groupnorm_bwd_cat: null pointer");
    if (C1 < 8 || C2 < 8 || (C1 & 7) || (C2 & 7)) return mi355_fail(MI355_ERR_ARG, "groupnorm_bwd_cat: C1 = %d, C2 = %d must be multiples of 8", C1, C2);
    return gn_bwd_launch(x1, x2, C1, dy, stats, gamma, beta, dx1, dx2, dgamma, dbeta, scratch, B, HW, C1 + C2, G, silu, dres1,
                         dres2, stream);
}

int mi355_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, int M, int C,
                        float eps, void* stream) {
    if (!x || !gamma || !beta || !y) return mi355_fail(MI355_ERR_ARG, "layernorm_fwd: null pointer");
    if (C % 8 || C > 8 * 64 * LN_MAXCH) return mi355_fail(MI355_ERR_UNSUPPORTED, "layernorm: C=%d (multiple of 8, <= %d)", C, 8 * 64 * LN_MAXCH);
    hipStream_t st = (hipStream_t)stream;
    int grid = (M + 3) / 4;
    if (grid > 4096) grid = 4096;
    ProfScope prof(FAM_NORM, st, (double)M * C * 2 * 2);
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, gamma, beta, (bf16_t*)y, stats, M, C, eps);
    return mi355_check_launch("layernorm_fwd");
}

static int ln_bwd_grid(int M) {
    int grid = (M + 7) / 8;              // 8 or 16 waves per block, one row per wave at a time
    return grid > 256 ? 256 : grid;
}
long long mi355_layernorm_bwd_scratch_floats(int M, int C) { return (long long)ln_bwd_grid(M) * 2 * C; }

// dgamma_dbeta: [2][C] floats (dgamma row, then dbeta row), overwritten.
// scratch: mi355_layernorm_bwd_scratch_floats().
int mi355_layernorm_bwd(const void* x, const void* dy, const float* stats, const float* gamma, void* dx,
                        float* dgamma_dbeta, float* scratch, int M, int C, const void* dres, void* stream) {
    if (!x || !dy || !stats || !gamma || !dx || !scratch)
        return mi355_fail(MI355_ERR_ARG, "layernorm_bwd: null pointer");
    if (C % 8 || C > 8 * 64 * LN_MAXCH) return mi355_fail(MI355_ERR_UNSUPPORTED, "layernorm: C=%d unsupported", C);
    hipStream_t st = (hipStream_t)stream;
    const int grid = ln_bwd_grid(M);
    ProfScope prof(FAM_NORM, st, (double)M * C * 2 * 3);
#define LN_BWD(NCH)                                                                                            \
    do {                                                                                                       \
        constexpr int threads = NCH <= 2 ? 1024 : 512;                                                         \
        const size_t sh = ((size_t)(threads / 64) * 2 + 1) * C * sizeof(float);                                \
        static std::once_flag attr_set;                                                                          \
        std::call_once(attr_set, [] {                                                                                       \
            (void)hipFuncSetAttribute((const void*)ln_bwd_kernel<NCH>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                (int)(((size_t)(threads / 64) * 2 + 1) * 512 * NCH * sizeof(float)));                                                                                   \
        });                                                                                                      \
        hipLaunchKernelGGL(ln_bwd_kernel<NCH>, dim3(grid), dim3(threads), sh, st, (const bf16_t*)x,            \
                           (const bf16_t*)dy, stats, gamma, (bf16_t*)dx, scratch, M, C, (const bf16_t*)dres); \
    } while (0)
    if (C <= 512) LN_BWD(1);
    else if (C <= 1024) LN_BWD(2);
    else if (C <= 1536) LN_BWD(3);
    else LN_BWD(4);
#undef LN_BWD
    if (dgamma_dbeta)       // null: frozen affine parameters, only dx is wanted
        hipLaunchKernelGGL(reduce_rows_kernel, dim3((2 * C + 63) / 64, 1), dim3(256), 0, st, scratch, dgamma_dbeta, grid, 2 * C);
    return mi355_check_launch("layernorm_bwd");
}

// Deferred parameter-gradient reductions of many norm layers in one launch.  table: n records of 6 int64 in device
// memory {partial rows, out, out_hi (or 0), rows, N, split}: out[c] = sum_r partial[r][c] for c < split (or all c when
// out_hi is 0), out_hi[c - split] otherwise.  max_cols: the largest N.  N and split multiples of 4.
int mi355_reduce_rows_batched(const long long* table, int n, int max_cols, void* stream) {
    if (!table || n < 1 || max_cols < 4) return mi355_fail(MI355_ERR_ARG, "reduce_rows_batched: bad arguments");
    hipLaunchKernelGGL(reduce_rows_batched_kernel, dim3((max_cols + 63) / 64, n), dim3(256), 0, (hipStream_t)stream, table);
    return mi355_check_launch("reduce_rows_batched");
}


int mi355_ln_fold_weight(const float* W, int ldw, const float* gamma, const float* beta, const float* bias, void* Wp,
                         float* colsum, float* bias_out, int N, int K, float scale, void* stream) {
    if (!W || !gamma || !beta || !Wp || !colsum || !bias_out || N < 1 || K < 1 || ldw < K)
        return mi355_fail(MI355_ERR_ARG, "ln_fold_weight: bad arguments");
    hipLaunchKernelGGL(ln_fold_weight_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, W, ldw, gamma, beta, bias, (bf16_t*)Wp,
                       colsum, bias_out, K, scale);
    return mi355_check_launch("ln_fold_weight");
}
}  // extern "C"
