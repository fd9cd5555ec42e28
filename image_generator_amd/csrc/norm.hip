// GroupNorm(+SiLU) and LayerNorm, forward and backward, for NHWC / token-major bf16 activations.
//
// All kernels are HBM-bound; every global access is a 16-byte (8 x bf16) vector on a contiguous
// channel run, statistics are fp32.  GroupNorm works on the [B][HW][C] view: a thread owns one
// 8-channel chunk for a strip of pixels, so reads are fully coalesced rows regardless of the group
// width (C/32 is 10..80 channels in this network and is not a multiple of the vector width).
//
//   forward :  chan_sums (sum x, sum x^2 per (b,c), atomics)  ->  gn_finalize (mean, rstd per (b,g))
//              ->  gn_apply  (y = silu?(x*scale + shift))
//   backward:  chan_sums (sum dz, sum dz*xhat per (b,c))      ->  gn_bwd_finalize (group sums, dgamma, dbeta)
//              ->  gn_bwd_apply (dx)
#include "common.h"

__device__ __forceinline__ float silu_f(float z) { return z / (1.f + __expf(-z)); }
__device__ __forceinline__ float silu_grad_f(float z) {
    const float s = 1.f / (1.f + __expf(-z));
    return s * (1.f + z * (1.f - s));
}

// MODE 0: acc0 += x, acc1 += x^2.   MODE 1: dz = dy * act'(z); acc0 += dz, acc1 += dz * xhat.
// grid = (strips, B); blockDim.x = nc * rows_par (nc = C/8).  sums: per-strip partials [B][strips][C][2] fp32.
template <int MODE>
__global__ void gn_chan_sums_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                    const float* __restrict__ stats, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, float* __restrict__ sums, int HW, int C, int G,
                                    int rows_per_block, int silu) {
    const int nc = C >> 3;
    const int b = blockIdx.y;
    const int c8 = threadIdx.x % nc, rp = threadIdx.x / nc, rpar = blockDim.x / nc;
    const int r_begin = blockIdx.x * rows_per_block;
    int r_end = r_begin + rows_per_block;
    if (r_end > HW) r_end = HW;
    const int cg = C / G;
    float sc[8], sh[8], mu[8], rs[8];
    if (MODE == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c8 * 8 + e, g = c / cg;
            mu[e] = stats[((size_t)b * G + g) * 2];
            rs[e] = stats[((size_t)b * G + g) * 2 + 1];
            sc[e] = gamma[c];
            sh[e] = beta[c];
        }
    }
    float a0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const size_t base = (size_t)b * HW * C + c8 * 8;
    for (int r = r_begin + rp; r < r_end; r += rpar) {
        const bf16x8 xv = *(const bf16x8*)(x + base + (size_t)r * C);
        if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float v = (float)xv[e]; a0[e] += v; a1[e] += v * v; }
        } else {
            const bf16x8 dv = *(const bf16x8*)(dy + base + (size_t)r * C);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = ((float)xv[e] - mu[e]) * rs[e];
                float dz = (float)dv[e];
                if (silu) dz *= silu_grad_f(xh * sc[e] + sh[e]);
                a0[e] += dz; a1[e] += dz * xh;
            }
        }
    }
    // combine the row phases of this block through LDS, then one atomic per (channel, stat)
    extern __shared__ float red[];   // [rpar][nc][16]
    float* mine = red + ((size_t)rp * nc + c8) * 16;
#pragma unroll
    for (int e = 0; e < 8; ++e) { mine[e] = a0[e]; mine[8 + e] = a1[e]; }
    __syncthreads();
    if (rp == 0) {
        for (int q = 1; q < rpar; ++q) {
            const float* o = red + ((size_t)q * nc + c8) * 16;
#pragma unroll
            for (int e = 0; e < 8; ++e) { a0[e] += o[e]; a1[e] += o[8 + e]; }
        }
        // per-strip partial (plain stores; a second kernel sums the strips - no contended atomics)
        float* dst = sums + (((size_t)b * gridDim.x + blockIdx.x) * C + c8 * 8) * 2;
#pragma unroll
        for (int e = 0; e < 8; e += 2)
            *(f32x4*)(dst + 2 * e) = f32x4{a0[e], a1[e], a0[e + 1], a1[e + 1]};
    }
}

// sums[b][c][k] = sum over strips of partial[b][strip][c][k]   (one thread per (b, c, k))
__global__ void gn_reduce_strips_kernel(const float* __restrict__ partial, float* __restrict__ sums, int B, int C,
                                        int strips) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C * 2) return;
    const int b = i / (C * 2), r = i - b * C * 2;
    float s = 0.f;
    for (int t = 0; t < strips; ++t) s += partial[((size_t)b * strips + t) * C * 2 + r];
    sums[i] = s;
}

// one thread per (b, g)
__global__ void gn_finalize_kernel(const float* __restrict__ sums, float* __restrict__ stats, int B, int C, int G,
                                   int HW, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * G) return;
    const int b = i / G, g = i - b * G, cg = C / G;
    float s = 0.f, ss = 0.f;
    for (int c = g * cg; c < (g + 1) * cg; ++c) { s += sums[((size_t)b * C + c) * 2]; ss += sums[((size_t)b * C + c) * 2 + 1]; }
    const float n = (float)HW * cg;
    const float mean = s / n;
    float var = ss / n - mean * mean;
    if (var < 0.f) var = 0.f;
    stats[(size_t)i * 2] = mean;
    stats[(size_t)i * 2 + 1] = rsqrtf(var + eps);
}

__global__ void gn_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ stats,
                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                bf16_t* __restrict__ y, int HW, int C, int G, int rows_per_block, int silu) {
    const int nc = C >> 3;
    const int b = blockIdx.y;
    const int c8 = threadIdx.x % nc, rp = threadIdx.x / nc, rpar = blockDim.x / nc;
    const int r_begin = blockIdx.x * rows_per_block;
    int r_end = r_begin + rows_per_block;
    if (r_end > HW) r_end = HW;
    const int cg = C / G;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c8 * 8 + e, g = c / cg;
        const float mean = stats[((size_t)b * G + g) * 2], rstd = stats[((size_t)b * G + g) * 2 + 1];
        sc[e] = rstd * gamma[c];
        sh[e] = beta[c] - mean * sc[e];
    }
    const size_t base = (size_t)b * HW * C + c8 * 8;
    for (int r = r_begin + rp; r < r_end; r += rpar) {
        const bf16x8 xv = *(const bf16x8*)(x + base + (size_t)r * C);
        bf16x8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float z = (float)xv[e] * sc[e] + sh[e];
            if (silu) z = silu_f(z);
            ov[e] = (bf16_t)z;
        }
        *(bf16x8*)(y + base + (size_t)r * C) = ov;
    }
}

// one thread per (b,g): gsum[b][g] = {sum_c gamma*dbeta_bc, sum_c gamma*dgamma_bc}; block (0) threads also
// reduce dgamma/dbeta over the batch.  sums: [B][C][2] = {sum dz, sum dz*xhat}.
__global__ void gn_bwd_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ gamma,
                                       float* __restrict__ gsum, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, int B, int C, int G) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int cg = C / G;
    if (i < B * G) {
        const int b = i / G, g = i - b * G;
        float s1 = 0.f, s2 = 0.f;
        for (int c = g * cg; c < (g + 1) * cg; ++c) {
            s1 += gamma[c] * sums[((size_t)b * C + c) * 2];
            s2 += gamma[c] * sums[((size_t)b * C + c) * 2 + 1];
        }
        gsum[(size_t)i * 2] = s1;
        gsum[(size_t)i * 2 + 1] = s2;
    }
    if (i < C) {
        float db = 0.f, dg = 0.f;
        for (int b = 0; b < B; ++b) { db += sums[((size_t)b * C + i) * 2]; dg += sums[((size_t)b * C + i) * 2 + 1]; }
        dbeta[i] = db;
        dgamma[i] = dg;
    }
}

__global__ void gn_bwd_apply_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                    const float* __restrict__ stats, const float* __restrict__ gsum,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    bf16_t* __restrict__ dx, int HW, int C, int G, int rows_per_block, int silu) {
    const int nc = C >> 3;
    const int b = blockIdx.y;
    const int c8 = threadIdx.x % nc, rp = threadIdx.x / nc, rpar = blockDim.x / nc;
    const int r_begin = blockIdx.x * rows_per_block;
    int r_end = r_begin + rows_per_block;
    if (r_end > HW) r_end = HW;
    const int cg = C / G;
    const float inv_n = 1.f / ((float)HW * cg);
    float mu[8], rs[8], ga[8], be[8], m1[8], m2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c8 * 8 + e, g = c / cg;
        mu[e] = stats[((size_t)b * G + g) * 2];
        rs[e] = stats[((size_t)b * G + g) * 2 + 1];
        ga[e] = gamma[c];
        be[e] = beta[c];
        m1[e] = gsum[((size_t)b * G + g) * 2] * inv_n;
        m2[e] = gsum[((size_t)b * G + g) * 2 + 1] * inv_n;
    }
    const size_t base = (size_t)b * HW * C + c8 * 8;
    for (int r = r_begin + rp; r < r_end; r += rpar) {
        const bf16x8 xv = *(const bf16x8*)(x + base + (size_t)r * C);
        const bf16x8 dv = *(const bf16x8*)(dy + base + (size_t)r * C);
        bf16x8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xh = ((float)xv[e] - mu[e]) * rs[e];
            float dz = (float)dv[e];
            if (silu) dz *= silu_grad_f(xh * ga[e] + be[e]);
            ov[e] = (bf16_t)(rs[e] * (ga[e] * dz - m1[e] - xh * m2[e]));
        }
        *(bf16x8*)(dx + base + (size_t)r * C) = ov;
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

// dx per row; dgamma/dbeta partials kept in registers across the rows a wave visits, combined by atomics.
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     bf16_t* __restrict__ dx, float* __restrict__ partial,
                                                     int M, int C) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int nc = C >> 3;
    float pg[LN_MAXCH][8], pb[LN_MAXCH][8], ga[LN_MAXCH][8];
#pragma unroll
    for (int k = 0; k < LN_MAXCH; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            pg[k][e] = 0.f; pb[k][e] = 0.f;
            const int c8 = lane + 64 * k;
            ga[k][e] = c8 < nc ? gamma[c8 * 8 + e] : 0.f;
        }
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < M; row += gridDim.x * wpb) {
        const float mean = stats[(size_t)row * 2], rstd = stats[(size_t)row * 2 + 1];
        float xh[LN_MAXCH][8], dg[LN_MAXCH][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < LN_MAXCH; ++k) {
            const int c8 = lane + 64 * k;
            if (c8 < nc) {
                const bf16x8 xv = *(const bf16x8*)(x + (size_t)row * C + c8 * 8);
                const bf16x8 dv = *(const bf16x8*)(dy + (size_t)row * C + c8 * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = (float)dv[e];
                    xh[k][e] = ((float)xv[e] - mean) * rstd;
                    dg[k][e] = d * ga[k][e];
                    s1 += dg[k][e];
                    s2 += dg[k][e] * xh[k][e];
                    pg[k][e] += d * xh[k][e];
                    pb[k][e] += d;
                }
            }
        }
        s1 = wave_sum(s1) / C;
        s2 = wave_sum(s2) / C;
#pragma unroll
        for (int k = 0; k < LN_MAXCH; ++k) {
            const int c8 = lane + 64 * k;
            if (c8 < nc) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(rstd * (dg[k][e] - s1 - xh[k][e] * s2));
                *(bf16x8*)(dx + (size_t)row * C + c8 * 8) = o;
            }
        }
    }
    // combine the block's waves through LDS, then one partial row per block: partial[block][2][C]
    extern __shared__ float lred[];      // [wpb][2][C]
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < LN_MAXCH; ++k) {
        const int c8 = lane + 64 * k;
        if (c8 < nc) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                lred[((size_t)w * 2) * C + c8 * 8 + e] = pg[k][e];
                lred[((size_t)w * 2 + 1) * C + c8 * 8 + e] = pb[k][e];
            }
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
    *rows_per_block = rows;
    *strips = (HW + rows - 1) / rows;
    return 0;
}

extern "C" {

// scratch floats needed by the two GroupNorm calls for a [B][HW][C] tensor with G groups.
long long mi355_groupnorm_scratch_floats(int B, int HW, int C, int G) {
    int threads, rpar, rows, strips;
    if (gn_geometry(HW, C, G, &threads, &rpar, &rows, &strips)) return -1;
    return (long long)B * strips * C * 2 + (long long)B * C * 2 + (long long)B * G * 2;
}

// scratch: mi355_groupnorm_scratch_floats().  stats out: [B][G][2] = mean, rstd.
int mi355_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, float* scratch,
                        int B, int HW, int C, int G, float eps, int silu, void* stream) {
    if (!x || !gamma || !beta || !y || !stats || !scratch) return mi355_fail(MI355_ERR_ARG, "groupnorm_fwd: null pointer");
    int threads, rpar, rows, strips;
    if (int rc = gn_geometry(HW, C, G, &threads, &rpar, &rows, &strips)) return rc;
    hipStream_t st = (hipStream_t)stream;
    float* partial = scratch;
    float* sums = scratch + (size_t)B * strips * C * 2;
    ProfScope prof(FAM_NORM, st, (double)B * HW * C * 2 * 3);
    const size_t sh = (size_t)threads * 16 * sizeof(float);
    hipLaunchKernelGGL(gn_chan_sums_kernel<0>, dim3(strips, B), dim3(threads), sh, st, (const bf16_t*)x, nullptr,
                       nullptr, nullptr, nullptr, partial, HW, C, G, rows, 0);
    hipLaunchKernelGGL(gn_reduce_strips_kernel, dim3((B * C * 2 + 255) / 256), dim3(256), 0, st, partial, sums, B, C, strips);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((B * G + 255) / 256), dim3(256), 0, st, sums, stats, B, C, G, HW, eps);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(strips, B), dim3(threads), 0, st, (const bf16_t*)x, stats, gamma, beta,
                       (bf16_t*)y, HW, C, G, rows, silu);
    return mi355_check_launch("groupnorm_fwd");
}

// scratch: mi355_groupnorm_scratch_floats().
int mi355_groupnorm_bwd(const void* x, const void* dy, const float* stats, const float* gamma, const float* beta,
                        void* dx, float* dgamma, float* dbeta, float* scratch, int B, int HW, int C, int G, int silu,
                        void* stream) {
    if (!x || !dy || !stats || !gamma || !beta || !dx || !dgamma || !dbeta || !scratch)
        return mi355_fail(MI355_ERR_ARG, "groupnorm_bwd: null pointer");
    int threads, rpar, rows, strips;
    if (int rc = gn_geometry(HW, C, G, &threads, &rpar, &rows, &strips)) return rc;
    hipStream_t st = (hipStream_t)stream;
    float* partial = scratch;
    float* sums = scratch + (size_t)B * strips * C * 2;
    float* gsum = sums + (size_t)B * C * 2;
    ProfScope prof(FAM_NORM, st, (double)B * HW * C * 2 * 5);
    const size_t sh = (size_t)threads * 16 * sizeof(float);
    hipLaunchKernelGGL(gn_chan_sums_kernel<1>, dim3(strips, B), dim3(threads), sh, st, (const bf16_t*)x,
                       (const bf16_t*)dy, stats, gamma, beta, partial, HW, C, G, rows, silu);
    hipLaunchKernelGGL(gn_reduce_strips_kernel, dim3((B * C * 2 + 255) / 256), dim3(256), 0, st, partial, sums, B, C, strips);
    const int nfin = B * G > C ? B * G : C;
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3((nfin + 255) / 256), dim3(256), 0, st, sums, gamma, gsum, dgamma,
                       dbeta, B, C, G);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(strips, B), dim3(threads), 0, st, (const bf16_t*)x, (const bf16_t*)dy,
                       stats, gsum, gamma, beta, (bf16_t*)dx, HW, C, G, rows, silu);
    return mi355_check_launch("groupnorm_bwd");
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
    int grid = (M + 3) / 4;
    return grid > 256 ? 256 : grid;
}
long long mi355_layernorm_bwd_scratch_floats(int M, int C) { return (long long)ln_bwd_grid(M) * 2 * C; }

// dgamma_dbeta: [2][C] floats (dgamma row, then dbeta row), overwritten.
// scratch: mi355_layernorm_bwd_scratch_floats().
int mi355_layernorm_bwd(const void* x, const void* dy, const float* stats, const float* gamma, void* dx,
                        float* dgamma_dbeta, float* scratch, int M, int C, void* stream) {
    if (!x || !dy || !stats || !gamma || !dx || !dgamma_dbeta || !scratch)
        return mi355_fail(MI355_ERR_ARG, "layernorm_bwd: null pointer");
    if (C % 8 || C > 8 * 64 * LN_MAXCH) return mi355_fail(MI355_ERR_UNSUPPORTED, "layernorm: C=%d unsupported", C);
    hipStream_t st = (hipStream_t)stream;
    const int grid = ln_bwd_grid(M);
    ProfScope prof(FAM_NORM, st, (double)M * C * 2 * 3);
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(grid), dim3(256), (size_t)4 * 2 * C * sizeof(float), st, (const bf16_t*)x,
                       (const bf16_t*)dy, stats, gamma, (bf16_t*)dx, scratch, M, C);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((2 * C + 63) / 64, 1), dim3(256), 0, st, scratch, dgamma_dbeta, grid, 2 * C);
    return mi355_check_launch("layernorm_bwd");
}

}  // extern "C"
