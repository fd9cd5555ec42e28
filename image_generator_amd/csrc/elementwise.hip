// HBM-bound elementwise / layout / loss / optimizer kernels (16-byte vector accesses, fp32 math).
#include "common.h"
#include <math.h>

#define GRID_FOR(n, per) ({ size_t _g = ((size_t)(n) + (per) - 1) / (per); if (_g > 4096) _g = 4096; if (_g < 1) _g = 1; (int)_g; })

__device__ __forceinline__ float gelu_f(float x) { return gelu_erf_f(x); }
__device__ __forceinline__ float gelu_grad_f(float x) { return gelu_erf_grad_f(x); }

// ---- row softmax of a bf16 [M][N] score matrix (single-head attention of the auto-encoder's mid block, whose
// head dim 512 is beyond the flash kernel): out = softmax(scale * in) per row, one wave per row, fp32 statistics
__global__ __launch_bounds__(256) void softmax_rows_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int M,
                                                           int N, float scale_log2) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const bf16_t* src = in + (size_t)row * N;
    bf16_t* dst = out + (size_t)row * N;
    const int nc = N >> 3;
    float mx = -INFINITY;
    for (int c = lane; c < nc; c += 64) {
        const bf16x8 v = *(const bf16x8*)(src + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)v[e]);
    }
    mx = wave_max(mx) * scale_log2;
    float sum = 0.f;
    for (int c = lane; c < nc; c += 64) {
        const bf16x8 v = *(const bf16x8*)(src + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += exp2f(fmaf((float)v[e], scale_log2, -mx));
    }
    const float inv = 1.f / wave_sum(sum);
    for (int c = lane; c < nc; c += 64) {
        const bf16x8 v = *(const bf16x8*)(src + c * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(exp2f(fmaf((float)v[e], scale_log2, -mx)) * inv);
        *(bf16x8*)(dst + c * 8) = o;
    }
}

// ---- GELU (erf form), bf16 -> bf16 (text-encoder feed-forward activation)
__global__ void gelu_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const bf16x8 v = *(const bf16x8*)(in + i * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)gelu_f((float)v[e]);
        *(bf16x8*)(out + i * 8) = o;
    }
}

// ---- GEGLU: in [M][2F] (hidden | gate) -> out [M][F] = hidden * gelu(gate)
__global__ void geglu_fwd_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, size_t M, int F) {
    const int nc = F >> 3;
    const size_t total = M * nc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / nc; const int c = (int)(i - m * nc) * 8;
        const bf16x8 h = *(const bf16x8*)(in + m * 2 * F + c);
        const bf16x8 g = *(const bf16x8*)(in + m * 2 * F + F + c);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)h[e] * gelu_f((float)g[e]));
        *(bf16x8*)(out + m * F + c) = o;
    }
}
__global__ void geglu_bwd_kernel(const bf16_t* __restrict__ in, const bf16_t* __restrict__ dout,
                                 bf16_t* __restrict__ din, size_t M, int F) {
    const int nc = F >> 3;
    const size_t total = M * nc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / nc; const int c = (int)(i - m * nc) * 8;
        const bf16x8 h = *(const bf16x8*)(in + m * 2 * F + c);
        const bf16x8 g = *(const bf16x8*)(in + m * 2 * F + F + c);
        const bf16x8 d = *(const bf16x8*)(dout + m * F + c);
        bf16x8 dh, dg;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gv = (float)g[e], dv = (float)d[e];
            dh[e] = (bf16_t)(dv * gelu_f(gv));
            dg[e] = (bf16_t)(dv * (float)h[e] * gelu_grad_f(gv));
        }
        *(bf16x8*)(din + m * 2 * F + c) = dh;
        *(bf16x8*)(din + m * 2 * F + F + c) = dg;
    }
}

// ---- SiLU on small fp32 matrices (timestep-embedding MLP), output bf16 for the next GEMM
__global__ void silu_f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        y[i] = (bf16_t)(v / (1.f + __expf(-v)));
    }
}
// dx = dy * silu'(x), fp32
__global__ void silu_bwd_f32_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                    size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        const float s = 1.f / (1.f + __expf(-v));
        dx[i] = dy[i] * s * (1.f + v * (1.f - s));
    }
}

// ---- sinusoidal timestep embedding: out[b][0:half] = cos(t*f_i), out[b][half:] = sin(t*f_i)
__global__ void timestep_embed_kernel(const float* __restrict__ t, bf16_t* __restrict__ out, int B, int dim) {
    const int half = dim >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, j = i - b * half;
    const float f = __expf(-9.210340371976184f * (float)j / (float)half);
    const float a = t[b] * f;
    out[(size_t)b * dim + j] = (bf16_t)cosf(a);
    out[(size_t)b * dim + half + j] = (bf16_t)sinf(a);
}

// ---- casts and layout
__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 a = *(const f32x4*)(x + i * 8), b = *(const f32x4*)(x + i * 8 + 4);
        bf16x8 o = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
        *(bf16x8*)(y + i * 8) = o;
    }
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = (float)x[i];
}
// in: [G][R][C] -> out: [G][C][R]  (32x32 LDS tile transpose), bf16
__global__ void transpose_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int R, int C) {
    __shared__ bf16_t tile[32][33];
    const size_t g = blockIdx.z;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < R && c < C) ? in[(g * R + r) * C + c] : (bf16_t)0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (r < R && c < C) out[(g * C + c) * R + r] = tile[tx][k];
    }
}
// conv weight for the data gradient: w [Co][T][Ci] -> wt [Ci][T][Co] with the taps reversed (T = 9 or 1)
__global__ void conv_weight_transpose_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ wt, int Co, int T,
                                             int Ci) {
    __shared__ bf16_t tile[32][33];
    const int tap = blockIdx.z;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int co = co0 + k, ci = ci0 + tx;
        tile[k][tx] = (co < Co && ci < Ci) ? w[((size_t)co * T + tap) * Ci + ci] : (bf16_t)0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int ci = ci0 + k, co = co0 + tx;
        if (co < Co && ci < Ci) wt[((size_t)ci * T + (T - 1 - tap)) * Co + co] = tile[tx][k];
    }
}
// All per-step weight transposes in one launch: job j covers tiles [first_j, first_{j+1}); each tile is a
// 64x64 block of one tap plane.  src [R][T][C] -> dst [C][T-1-t][R].
__global__ __launch_bounds__(256) void transpose_batched_kernel(const long long* __restrict__ jobs, int njobs) {
    __shared__ bf16_t tile[64][72];
    __shared__ int sjob;
    const int bid = blockIdx.x;
    if (threadIdx.x == 0) {
        int lo = 0, hi = njobs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[(size_t)mid * 6 + 5] <= bid) lo = mid; else hi = mid - 1;
        }
        sjob = lo;
    }
    __syncthreads();
    const long long* J = jobs + (size_t)sjob * 6;
    const bf16_t* src = (const bf16_t*)J[0];
    bf16_t* dst = (bf16_t*)J[1];
    const int R = (int)J[2], C = (int)J[3], T = (int)J[4];
    int t = bid - (int)J[5];
    const int tc = (C + 63) / 64, tr = (R + 63) / 64;
    const int c0 = (t % tc) * 64; t /= tc;
    const int r0 = (t % tr) * 64;
    const int tap = t / tr;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int lr = threadIdx.x >> 3, lc = (threadIdx.x & 7) * 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = r0 + lr + 32 * h, c = c0 + lc;
        bf16x8 v = zero8;
        if (r < R && c < C) v = *(const bf16x8*)(src + ((size_t)r * T + tap) * C + c);   // C % 8 == 0
        *(bf16x8*)(&tile[lr + 32 * h][lc]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = c0 + lr + 32 * h, r = r0 + lc;
        if (c < C && r < R) {
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[lc + e][lr + 32 * h];
            *(bf16x8*)(dst + ((size_t)c * T + (T - 1 - tap)) * R + r) = v;                 // R % 8 == 0
        }
    }
}

// NCHW fp32 [B][Cs][HW] -> NHWC bf16 [B][HW][Cd] (channels >= Cs zero padded)
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int B, int Cs, int Cd, int HW) {
    const size_t total = (size_t)B * HW * Cd;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cd); const size_t p = i / Cd; const int hw = (int)(p % HW); const size_t b = p / HW;
        y[i] = c < Cs ? (bf16_t)x[(b * Cs + c) * HW + hw] : (bf16_t)0.f;
    }
}
// NHWC fp32 [B][HW][Cs] -> NCHW fp32 [B][Cd][HW] (first Cd channels)
__global__ void nhwc_to_nchw_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int Cs, int Cd, int HW) {
    const size_t total = (size_t)B * Cd * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int hw = (int)(i % HW); const size_t p = i / HW; const int c = (int)(p % Cd); const size_t b = p / Cd;
        y[i] = x[(b * HW + hw) * Cs + c];
    }
}
// gradient of a nearest 2x upsample: dx[b][y][x][c] = sum of the 2x2 block of dy (NHWC bf16, dy is [B][2H][2W][C])
__global__ void sumpool2x2_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int B, int H, int W, int C) {
    const int nc = C >> 3;
    const size_t total = (size_t)B * H * W * nc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % nc) * 8; size_t p = i / nc;
        const int x = (int)(p % W); p /= W; const int y = (int)(p % H); const size_t b = p / H;
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dyy = 0; dyy < 2; ++dyy)
#pragma unroll
            for (int dxx = 0; dxx < 2; ++dxx) {
                const bf16x8 v = *(const bf16x8*)(dy + ((b * 2 * H + 2 * y + dyy) * 2 * W + 2 * x + dxx) * C + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] += (float)v[e];
            }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)a[e];
        *(bf16x8*)(dx + ((b * H + y) * W + x) * C + c) = o;
    }
}

// ---- MSE objective, fused forward + gradient: loss += sum((p-t)^2)/n ; grad = 2(p-t)/n
__global__ __launch_bounds__(256) void mse_fused_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                        float* __restrict__ grad, float* __restrict__ loss, size_t n) {
    __shared__ float red[16];
    const float inv = 1.f / (float)n;
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = pred[i] - target[i];
        s += d * d;
        if (grad) grad[i] = 2.f * d * inv;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(loss, s * inv);
}

// ---- diffusion schedule elementwise ops (fp32, NCHW contiguous, per-sample coefficients)
__global__ void ddpm_add_noise_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                      const long long* __restrict__ t, const float* __restrict__ acp,
                                      float* __restrict__ out, int B, size_t per) {
    const size_t total = (size_t)B * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float a = acp[t[i / per]];
        out[i] = sqrtf(a) * x0[i] + sqrtf(1.f - a) * noise[i];
    }
}
// eps = eps_u + g (eps_c - eps_u);  x_prev = sqrt(a_prev) * (x - sqrt(1-a_t) eps)/sqrt(a_t) + sqrt(1-a_prev) eps
// coef (device): {a_t, a_prev, guidance}.  eps holds [2][n] (uncond first) when cfg != 0, else [n].
__global__ void ddim_step_cfg_kernel(const float* __restrict__ eps, const float* __restrict__ x, float* __restrict__ out,
                                     const float* __restrict__ coef, size_t n, int cfg) {
    const float a_t = coef[0], a_p = coef[1], g = coef[2];
    const float st = sqrtf(1.f - a_t), it = rsqrtf(a_t), sp = sqrtf(a_p), np_ = sqrtf(1.f - a_p);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float e = eps[i];
        if (cfg) e = e + g * (eps[n + i] - e);
        const float x0 = (x[i] - st * e) * it;
        out[i] = sp * x0 + np_ * e;
    }
}

// ---- optimizer: sum of squares, clip coefficient, fused AdamW with bf16 weight refresh
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, float* __restrict__ out, size_t n) {
    __shared__ float red[16];
    float s = 0.f;
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = *(const f32x4*)(g + i * 4);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t i = n4 * 4; i < n; ++i) s += g[i] * g[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(out, s);
}
// table[2r], table[2r+1]: address and float count of range r; a NEGATIVE count marks values that are already partial sums
// (added as they are).  grid (blocks per range, ranges).
__global__ __launch_bounds__(256) void sumsq_ranges_kernel(const long long* __restrict__ table, float* __restrict__ out) {
    __shared__ float red[16];
    const float* g = (const float*)table[2 * blockIdx.y];
    long long n = table[2 * blockIdx.y + 1];
    const bool plain = n < 0;
    if (plain) n = -n;
    float s = 0.f;
    const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
    if (plain) {
        for (long long i = i0; i < n; i += stride) s += g[i];
    } else if ((((uintptr_t)g) & 15) == 0) {
        const long long n4 = n >> 2;
        for (long long i = i0; i < n4; i += stride) {
            const f32x4 v = *(const f32x4*)(g + i * 4);
            s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
        if (i0 < (n & 3)) { const float v = g[n4 * 4 + i0]; s += v * v; }
    } else {
        for (long long i = i0; i < n; i += stride) s += g[i] * g[i];
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0 && s != 0.f) atomicAdd(out, s);
}
__global__ __launch_bounds__(256) void sumsq_bf16_kernel(const bf16_t* __restrict__ g, float* __restrict__ out, size_t n8) {
    __shared__ float red[16];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const bf16x8 v = *(const bf16x8*)(g + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)v[e] * (float)v[e];
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(out, s);
}
// The same sum with a result that depends only on the data: every block stores its partial, one block adds the partials
// in index order.  Data-parallel replicas compute their clip factor from identical reduced gradients; with the atomic
// version the order of the additions differs from rank to rank, the factors differ in the last bit and the replicas'
// weights drift apart by an ulp per step.
__global__ __launch_bounds__(256) void sumsq_bf16_partial_kernel(const bf16_t* __restrict__ g, float* __restrict__ partial, size_t n8) {
    __shared__ float red[16];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const bf16x8 v = *(const bf16x8*)(g + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)v[e] * (float)v[e];
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] += s;
}
// hyper (device floats): [0]=lr [1]=beta1 [2]=beta2 [3]=eps [4]=weight_decay [5]=bias_corr1 [6]=bias_corr2
//                        [7]=max_grad_norm (<=0: no clipping) [8]=grad pre-scale (e.g. 1/world)
// state (device floats): [0]=sum of squared grads (input) ; [1] += 1 when the update is skipped (non-finite norm)
// G16: the gradients are bf16 (the all-reduced staging buffer of a data-parallel step, read without unpacking)
template <int U, bool NT, bool G16 = false>
__global__ void __launch_bounds__(256) adamw_kernel(float* __restrict__ p, const void* __restrict__ gptr, float* __restrict__ m,
                             float* __restrict__ v, bf16_t* __restrict__ p_bf16, const float* __restrict__ hyper,
                             float* __restrict__ state, size_t n) {
    const float* g = (const float*)gptr;
    const bf16_t* g16 = (const bf16_t*)gptr;
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], bc1 = hyper[5], bc2 = hyper[6];
    const float maxn = hyper[7], pre = hyper[8];
    const float gnorm = sqrtf(state[0]) * pre;
    if (!(gnorm == gnorm) || gnorm > 3.0e38f) {                 // non-finite gradients: skip the update, leave a trace
        if (blockIdx.x == 0 && threadIdx.x == 0) state[1] += 1.f;
        return;
    }
    float scale = pre;
    if (maxn > 0.f) { const float c = maxn / (gnorm + 1e-6f); if (c < 1.f) scale *= c; }
    const float decay = 1.f - lr * wd, rbc2 = 1.f / sqrtf(bc2), step = lr / bc1, omb1 = 1.f - b1, omb2 = 1.f - b2;
    const size_t n4 = n >> 2;
    const size_t stride = 256;                                 // one block owns U consecutive 4 KB runs of each array
    {
        const size_t i0 = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
        f32x4 pv[U], gv[U], mv[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = i0 + u * stride;
            if (i < n4) {
                if (G16) {
                    const bf16x4 t = __builtin_nontemporal_load((const bf16x4*)(g16 + i * 4));
                    gv[u] = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
                }
                if (NT) {
                    pv[u] = __builtin_nontemporal_load((const f32x4*)(p + i * 4));
                    if (!G16) gv[u] = __builtin_nontemporal_load((const f32x4*)(g + i * 4));
                    mv[u] = __builtin_nontemporal_load((const f32x4*)(m + i * 4));
                    vv[u] = __builtin_nontemporal_load((const f32x4*)(v + i * 4));
                } else {
                    pv[u] = *(const f32x4*)(p + i * 4);
                    if (!G16) gv[u] = *(const f32x4*)(g + i * 4);
                    mv[u] = *(const f32x4*)(m + i * 4); vv[u] = *(const f32x4*)(v + i * 4);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = i0 + u * stride;
            if (i >= n4) break;
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gg = gv[u][e] * scale;
                float pe = pv[u][e] * decay;
                const float me = b1 * mv[u][e] + omb1 * gg;
                const float ve = b2 * vv[u][e] + omb2 * gg * gg;
                const float denom = sqrtf(ve) * rbc2 + eps;
                pe -= step * (me / denom);
                pv[u][e] = pe; mv[u][e] = me; vv[u][e] = ve;
                o[e] = (bf16_t)pe;
            }
            if (NT) {
                __builtin_nontemporal_store(pv[u], (f32x4*)(p + i * 4));
                __builtin_nontemporal_store(mv[u], (f32x4*)(m + i * 4));
                __builtin_nontemporal_store(vv[u], (f32x4*)(v + i * 4));
            } else {
                *(f32x4*)(p + i * 4) = pv[u]; *(f32x4*)(m + i * 4) = mv[u]; *(f32x4*)(v + i * 4) = vv[u];
            }
            if (p_bf16) *(bf16x4*)(p_bf16 + i * 4) = o;
        }
    }
}

// ---- symmetric InfoNCE (CLIP-style) on L2-normalised embeddings, one workgroup, B <= 1024
//   logits = (t_hat . i_hat^T) / clamp(exp(log_temp), 1e-4, 1e4); loss = (CE_rows + CE_cols)/2
// Outputs loss, d_text, d_image, d_log_temp.  scratch: 4*B*B floats? no: works row-by-row from global logits.
// grid 2B: blocks [0, B) normalise the text rows, [B, 2B) the image rows; block 0 also clears the two scalars that the
// later kernels of the same call accumulate into with atomics (no separate zeroing launches)
__global__ __launch_bounds__(256) void l2norm_pair_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                          float* __restrict__ xh0, float* __restrict__ xh1,
                                                          float* __restrict__ nrm0, float* __restrict__ nrm1,
                                                          float* __restrict__ loss, float* __restrict__ dlogtemp, int B, int D) {
    __shared__ float red[16];
    if (blockIdx.x == 0 && threadIdx.x == 0) { loss[0] = 0.f; dlogtemp[0] = 0.f; }
    const bool second = (int)blockIdx.x >= B;
    const float* x = second ? x1 : x0;
    float* xh = second ? xh1 : xh0;
    float* nrm = second ? nrm1 : nrm0;
    const int b = second ? blockIdx.x - B : blockIdx.x;
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += blockDim.x) { const float v = x[(size_t)b * D + d]; s += v * v; }
    s = block_sum(s, red);
    const float n = fmaxf(sqrtf(s), 1e-12f);
    if (threadIdx.x == 0) nrm[b] = n;
    for (int d = threadIdx.x; d < D; d += blockDim.x) xh[(size_t)b * D + d] = x[(size_t)b * D + d] / n;
}
// logits[i][j] = dot(th[i], ih[j]) * inv_temp     (grid = B x B/… small; fp32 exact dot)
__global__ __launch_bounds__(256) void clip_logits_kernel(const float* __restrict__ th, const float* __restrict__ ih,
                                                          const float* __restrict__ log_temp, float* __restrict__ logits,
                                                          int B, int D) {
    const int i = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float temp = fminf(fmaxf(__expf(log_temp[0]), 1e-4f), 1e4f);
    for (int j = blockIdx.x * 4 + w; j < B; j += gridDim.x * 4) {
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += th[(size_t)i * D + d] * ih[(size_t)j * D + d];
        s = wave_sum(s);
        if (lane == 0) logits[(size_t)i * B + j] = s / temp;
    }
}
// row (dir=0) / column (dir=1) log-sum-exp of logits
__global__ __launch_bounds__(256) void clip_lse_kernel(const float* __restrict__ logits, float* __restrict__ lse, int B) {
    __shared__ float red[16];
    const int r = blockIdx.x, dir = blockIdx.y;
    float mx = -3.0e38f;
    for (int j = threadIdx.x; j < B; j += blockDim.x) mx = fmaxf(mx, dir ? logits[(size_t)j * B + r] : logits[(size_t)r * B + j]);
    mx = wave_max(mx);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = red[0];
    for (int k = 1; k < (int)(blockDim.x >> 6); ++k) mx = fmaxf(mx, red[k]);
    float s = 0.f;
    for (int j = threadIdx.x; j < B; j += blockDim.x) s += __expf((dir ? logits[(size_t)j * B + r] : logits[(size_t)r * B + j]) - mx);
    s = block_sum(s, red);
    if (threadIdx.x == 0) lse[(size_t)dir * B + r] = mx + logf(s);
}
// loss and dlogits (in place):  dL/dlogits[i][j] = ( softmax_row + softmax_col - 2*delta_ij ) / (2B)
__global__ __launch_bounds__(256) void clip_dlogits_kernel(float* __restrict__ logits, const float* __restrict__ lse,
                                                           float* __restrict__ loss, float* __restrict__ dlogtemp,
                                                           int B) {
    __shared__ float red[16];
    const int i = blockIdx.x;
    float acc_t = 0.f, l = 0.f;
    for (int j = threadIdx.x; j < B; j += blockDim.x) {
        const float z = logits[(size_t)i * B + j];
        float d = (__expf(z - lse[i]) + __expf(z - lse[B + j])) * (0.5f / B);
        if (i == j) { d -= 1.0f / B; l = 0.5f * ((lse[i] - z) + (lse[B + i] - z)) / B; }
        logits[(size_t)i * B + j] = d;
        acc_t += d * z;          // d logits / d log_temp = -logits (when the clamp is inactive)
    }
    acc_t = block_sum(acc_t, red);
    l = block_sum(l, red);
    if (threadIdx.x == 0) { atomicAdd(loss, l); atomicAdd(dlogtemp, -acc_t); }
}
// d_xhat[i][:] = sum_j dlogits[i][j] (or [j][i]) * other_hat[j][:] / temp ; then through the L2 normalisation:
// dx = (d_xhat - xhat * dot(xhat, d_xhat)) / norm
// grid (B, 2): blockIdx.y = 0 text rows (d_text from the rows of dlogits and the image embeddings), 1 image rows (columns of
// dlogits, text embeddings).  Block (0, 0) also applies the clamp of the temperature to its gradient: every atomic add to
// it comes from the previous kernel of the call.
__global__ __launch_bounds__(256) void clip_dembed_kernel(const float* __restrict__ dlogits, const float* __restrict__ th,
                                                          const float* __restrict__ ih, const float* __restrict__ tn,
                                                          const float* __restrict__ in_, const float* __restrict__ log_temp,
                                                          float* __restrict__ d_text, float* __restrict__ d_image,
                                                          float* __restrict__ dlogtemp, int B, int D) {
    __shared__ float red[16];
    extern __shared__ float dxh[];
    const int i = blockIdx.x, transpose = blockIdx.y;
    const float* other = transpose ? th : ih;
    const float* xh = transpose ? ih : th;
    const float* nrm = transpose ? in_ : tn;
    float* dx = transpose ? d_image : d_text;
    const float e_t = __expf(log_temp[0]);
    if (i == 0 && transpose == 0 && threadIdx.x == 0 && (e_t < 1e-4f || e_t > 1e4f)) dlogtemp[0] = 0.f;   // clamp saturated
    const float temp = fminf(fmaxf(e_t, 1e-4f), 1e4f);
    float dot = 0.f;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float s = 0.f;
        for (int j = 0; j < B; ++j) s += (transpose ? dlogits[(size_t)j * B + i] : dlogits[(size_t)i * B + j]) * other[(size_t)j * D + d];
        s /= temp;
        dxh[d] = s;
        dot += s * xh[(size_t)i * D + d];
    }
    dot = block_sum(dot, red);
    for (int d = threadIdx.x; d < D; d += blockDim.x) dx[(size_t)i * D + d] = (dxh[d] - xh[(size_t)i * D + d] * dot) / nrm[i];
}

// ---- LoRA skinny products
// t[m][r] = sum_k x[m][k] * A[r][k]     (x bf16 [M][K], A fp32 [R][K], R <= 16), one wave per row
// t16 (optional): bf16 [M][8] copy scale16 * t, columns >= R zero - the operand the MFMA weight-gradient kernel reads.
__global__ __launch_bounds__(256) void lora_down_kernel(const bf16_t* __restrict__ x, const float* __restrict__ A,
                                                        float* __restrict__ t, int M, int K, int R,
                                                        bf16_t* __restrict__ t16, float scale16) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        const bf16x8 xv = *(const bf16x8*)(x + (size_t)row * K + k);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (r < R) {
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) s += (float)xv[e] * A[(size_t)r * K + k + e];
                acc[r] += s;
            }
        }
    }
    bf16x8 h[2] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (r < R) {
            const float s = wave_sum(acc[r]);
            if (lane == 0) t[(size_t)row * R + r] = s;
            h[r >> 3][r & 7] = (bf16_t)(s * scale16);
        }
    }
    if (t16 && lane == 0) {
        if (R <= 8) *(bf16x8*)(t16 + (size_t)row * 8) = h[0];
        else { *(bf16x8*)(t16 + (size_t)row * 16) = h[0]; *(bf16x8*)(t16 + (size_t)row * 16 + 8) = h[1]; }
    }
}
// dt[m][s*R + r] = alpha * sum_{n in segment s} dy[m][n] * Bm[n][r]   (dy bf16 [M][N], Bm fp32 [N][R], N = S * seg:
// S row-stacked adapters of one fused projection; S = 1 is a single adapter).  One wave per row.
// R4: rank 4 (S <= 4), each column's four factors are one 16-byte load.  dt16 (optional): bf16 copy, 8 columns when
// S*R <= 8, else 16, zero padded.
template <bool R4>
__global__ __launch_bounds__(256) void lora_dt_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ Bm,
                                                      float* __restrict__ dt, int M, int N, int R, int S, float alpha,
                                                      bf16_t* __restrict__ dt16) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (R4) {
        const int seg = N / S;
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
            if (sg < S) {
                for (int n = sg * seg + lane * 8; n < (sg + 1) * seg; n += 512) {
                    const bf16x8 dv = *(const bf16x8*)(dy + (size_t)row * N + n);
                    const f32x4* bp = (const f32x4*)(Bm + (size_t)n * 4);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const f32x4 bv = bp[e];
                        const float d = (float)dv[e];
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[sg * 4 + r] += d * bv[r];
                    }
                }
            }
        }
    } else {
        for (int n = lane * 8; n < N; n += 512) {
            const bf16x8 dv = *(const bf16x8*)(dy + (size_t)row * N + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = (float)dv[e];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (r < R) acc[r] += d * Bm[(size_t)(n + e) * R + r];
            }
        }
    }
    const int RT = R * S;
    bf16x8 h[2] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (r < RT) {
            const float s = wave_sum(acc[r]) * alpha;
            if (lane == 0) dt[(size_t)row * RT + r] = s;
            h[r >> 3][r & 7] = (bf16_t)s;
        }
    }
    if (dt16 && lane == 0) {
        if (RT <= 8) *(bf16x8*)(dt16 + (size_t)row * 8) = h[0];
        else { *(bf16x8*)(dt16 + (size_t)row * 16) = h[0]; *(bf16x8*)(dt16 + (size_t)row * 16 + 8) = h[1]; }
    }
}
// out[c][r] += scale * sum_m y[m][c] * t[m][r]   (y bf16 [M][C], t fp32 [M][R]); out fp32 [C][R] (or [R][C] when
// `transposed`), zeroed beforehand.  Serves dB (y = dy, t = x A^T, scale = alpha) and dA (y = x, t = dt, scale = 1).
// A workgroup is NC = C/8 column chunks x RL row lanes; it walks `rows_per_block` rows, combines the row lanes
// through LDS and adds its [C][R] partial to the result with atomics.  fp32 atomics retire at ~15 G/s on this
// part, so the grid is kept small (C*R atomics per workgroup); a training step uses the MFMA weight-gradient
// queue on bf16 copies of t instead and keeps this kernel for stand-alone adapters.
template <bool R4>
__global__ __launch_bounds__(256) void lora_outer_kernel(const bf16_t* __restrict__ y, const float* __restrict__ t,
                                                         float* __restrict__ out, int M, int C, int R, float scale,
                                                         int rows_per_block, int NC, int RL, int transposed) {
    constexpr int RA = R4 ? 4 : 16;
    extern __shared__ float red[];          // [RL-1][NC][8*RA] partials of row lanes 1..RL-1
    const int tid = threadIdx.x;
    const int c8 = tid % NC, rl = tid / NC;
    const int c0 = blockIdx.y * NC * 8 + c8 * 8;
    const bool active = rl < RL && c0 < C;
    const int m_begin = blockIdx.x * rows_per_block;
    const int m_end = min(M, m_begin + rows_per_block);
    float acc[8][RA];
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int r = 0; r < RA; ++r) acc[e][r] = 0.f;
    if (active) {
        int m = m_begin + rl;
        for (; m + RL < m_end; m += 2 * RL) {        // two rows in flight
            const bf16x8 y0 = *(const bf16x8*)(y + (size_t)m * C + c0);
            const bf16x8 y1 = *(const bf16x8*)(y + (size_t)(m + RL) * C + c0);
            if (R4) {
                const f32x4 t0 = *(const f32x4*)(t + (size_t)m * 4), t1 = *(const f32x4*)(t + (size_t)(m + RL) * 4);
#pragma unroll
                for (int e = 0; e < 8; ++e)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[e][r] += (float)y0[e] * t0[r] + (float)y1[e] * t1[r];
            } else {
#pragma unroll
                for (int r = 0; r < RA; ++r)
                    if (r < R) {
                        const float t0 = t[(size_t)m * R + r], t1 = t[(size_t)(m + RL) * R + r];
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[e][r] += (float)y0[e] * t0 + (float)y1[e] * t1;
                    }
            }
        }
        for (; m < m_end; m += RL) {
            const bf16x8 y0 = *(const bf16x8*)(y + (size_t)m * C + c0);
#pragma unroll
            for (int r = 0; r < RA; ++r)
                if (r < R) {
                    const float t0 = t[(size_t)m * R + r];
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e][r] += (float)y0[e] * t0;
                }
        }
    }
    if (RL > 1) {
        if (active && rl > 0) {
            float* dst = red + ((size_t)(rl - 1) * NC + c8) * (8 * RA);
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int r = 0; r < RA; ++r) dst[e * RA + r] = acc[e][r];
        }
        __syncthreads();
        if (active && rl == 0) {
            for (int j = 0; j < RL - 1; ++j) {
                const float* src = red + ((size_t)j * NC + c8) * (8 * RA);
#pragma unroll
                for (int e = 0; e < 8; ++e)
#pragma unroll
                    for (int r = 0; r < RA; ++r) acc[e][r] += src[e * RA + r];
            }
        }
    }
    if (active && rl == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int r = 0; r < RA; ++r)
                if (r < R) {
                    float* dst = transposed ? out + (size_t)r * C + c0 + e : out + (size_t)(c0 + e) * R + r;
                    atomicAdd(dst, acc[e][r] * scale);
                }
    }
}

__global__ __launch_bounds__(256) void copy2d_batched_kernel(const long long* __restrict__ table) {
    const long long* e = table + 6 * blockIdx.y;
    const float* src = (const float*)e[0];
    float* dst = (float*)e[1];
    const long long rows = e[2], cols = e[3], sld = e[4], dld = e[5];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < rows * cols; i += (long long)gridDim.x * 256) {
        const long long r = i / cols, c = i - r * cols;
        dst[r * dld + c] = src[r * sld + c];
    }
}

// Per-step bf16 operands of the adapters' skinny GEMMs, one launch for all of them.  Record = 6 int64:
//   {0, src fp32, dst bf16, n, -, -}          dst[i] = bf16(src[i])                      (stacked A factors, row-major)
//   {1, src fp32 [C][r], dst bf16, C, r, ld}   dst[q * ld + c] = bf16(src[c * r + q])      (one adapter's B, transposed into
//                                                                                         its block of the [W][N] operand)
__global__ __launch_bounds__(256) void lora_pack_kernel(const long long* __restrict__ table) {
    const long long* e = table + 6 * blockIdx.y;
    const float* src = (const float*)e[1];
    bf16_t* dst = (bf16_t*)e[2];
    if (e[0] == 0) {
        const long long n = e[3];
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = (bf16_t)src[i];
    } else {
        const long long C = e[3], r = e[4], ld = e[5];
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < C * r; i += (long long)gridDim.x * 256) {
            const long long q = i / C, c = i - q * C;
            dst[q * ld + c] = (bf16_t)src[c * r + q];
        }
    }
}

// ---- dropout for the text encoder's training mode (encoder/text_encoder.py:16-23, dropout 0.2).  The mask of element
// i is a hash of (seed, i): forward and backward regenerate it, nothing is stored.
__device__ __forceinline__ bool drop_keep(unsigned long long i, unsigned seed, unsigned thresh) {
    unsigned x = (unsigned)i * 0x9E3779B1u + (unsigned)(i >> 32) * 0x85EBCA6Bu + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x >= thresh;
}
// y = res + dropout(x)   (res may be null; x, res, y bf16, 8 elements per thread)
__global__ __launch_bounds__(256) void dropout_add_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ res,
                                                          bf16_t* __restrict__ y, size_t n8, unsigned thresh, float scale,
                                                          unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const bf16x8 v = *(const bf16x8*)(x + i * 8);
        bf16x8 r = {0, 0, 0, 0, 0, 0, 0, 0};
        if (res) r = *(const bf16x8*)(res + i * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = drop_keep(i * 8 + e, seed, thresh) ? (float)v[e] * scale : 0.f;
            o[e] = (bf16_t)((float)r[e] + d);
        }
        *(bf16x8*)(y + i * 8) = o;
    }
}
// forward: y = dropout(gelu(x));  backward (dy given): dx = dropout(dy) * gelu'(x)
template <bool BWD>
__global__ __launch_bounds__(256) void gelu_dropout_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                           bf16_t* __restrict__ out, size_t n8, unsigned thresh, float scale,
                                                           unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const bf16x8 v = *(const bf16x8*)(x + i * 8);
        bf16x8 g = v, o;
        if (BWD) g = *(const bf16x8*)(dy + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float keep = drop_keep(i * 8 + e, seed, thresh) ? scale : 0.f;
            o[e] = (bf16_t)(BWD ? (float)g[e] * keep * gelu_grad_f((float)v[e]) : gelu_f((float)v[e]) * keep);
        }
        *(bf16x8*)(out + i * 8) = o;
    }
}
// ---- L2 normalisation of fp32 rows (the pooled output, encoder/text_encoder.py:37-40: F.normalize(x, dim=-1))
// forward: y = x / max(|x|, 1e-12), nrm[b] = that denominator;  backward: dx = (dy - y (y . dy)) / nrm
template <bool BWD>
__global__ __launch_bounds__(256) void l2norm_f32_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ out, float* __restrict__ nrm, int D) {
    __shared__ float red[16];
    const size_t b = blockIdx.x;
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) { const float v = x[b * D + d]; s += BWD ? v * dy[b * D + d] : v * v; }
    s = block_sum(s, red);
    if (BWD) {      // x is y here
        const float inv = 1.f / nrm[b];
        for (int d = threadIdx.x; d < D; d += 256) out[b * D + d] = (dy[b * D + d] - x[b * D + d] * s) * inv;
    } else {
        const float n = fmaxf(sqrtf(s), 1e-12f);
        if (threadIdx.x == 0) nrm[b] = n;
        for (int d = threadIdx.x; d < D; d += 256) out[b * D + d] = x[b * D + d] / n;
    }
}

// dst[i] = bf16(src[i]) over many ranges in one launch: record e = {src fp32 address, dst bf16 address, count} (int64),
// count a multiple of 4, src 16-byte and dst 8-byte aligned; blockIdx.y picks the record.
__global__ __launch_bounds__(256) void cast_ranges_kernel(const long long* __restrict__ table) {
    const long long* e = table + 3 * blockIdx.y;
    const f32x4* src = (const f32x4*)e[0];
    bf16x4* dst = (bf16x4*)e[1];
    const long long n4 = e[2] >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v = __builtin_nontemporal_load(src + i);
        dst[i] = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    }
}

// dst <- src over n ranges in one launch: table (device) = n records of 3 int64 {source address, destination address,
// bytes}, addresses 16-byte aligned, bytes a multiple of 16.  The packed operand of a product over several weights that
// are not neighbours in memory (the text key / value projections of every cross-attention layer run as one product).
__global__ __launch_bounds__(256) void copy_ranges_kernel(const long long* __restrict__ table) {
    const char* src = (const char*)table[3 * blockIdx.y];
    char* dst = (char*)table[3 * blockIdx.y + 1];
    const size_t n16 = (size_t)table[3 * blockIdx.y + 2] >> 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        *(f32x4*)(dst + (i << 4)) = *(const f32x4*)(src + (i << 4));
}

extern "C" {

#define ST ((hipStream_t)stream)

int mi355_softmax_rows_bf16(const void* in, void* out, int M, int N, float scale, void* stream) {
    if (!in || !out || M <= 0 || N <= 0 || (N & 7) || !(scale > 0.f)) return mi355_fail(MI355_ERR_ARG, "softmax_rows: bad arguments (N multiple of 8, scale > 0)");
    ProfScope prof(FAM_ELEM, ST, (double)M * N * 8);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, ST, (const bf16_t*)in, (bf16_t*)out, M, N,
                       scale * 1.4426950408889634f);
    return mi355_check_launch("softmax_rows");
}
int mi355_gelu_bf16(const void* in, void* out, long long n, void* stream) {
    if (!in || !out || (n & 7)) return mi355_fail(MI355_ERR_ARG, "gelu: bad arguments (n must be a multiple of 8)");
    ProfScope prof(FAM_ELEM, ST, (double)n * 4);
    hipLaunchKernelGGL(gelu_kernel, dim3(GRID_FOR((size_t)n / 8, 256)), dim3(256), 0, ST, (const bf16_t*)in, (bf16_t*)out, (size_t)n / 8);
    return mi355_check_launch("gelu");
}
int mi355_geglu_fwd(const void* in, void* out, long long M, int F, void* stream) {
    if (!in || !out || (F & 7)) return mi355_fail(MI355_ERR_ARG, "geglu_fwd: bad arguments");
    ProfScope prof(FAM_ELEM, ST, (double)M * F * 2 * 3);
    hipLaunchKernelGGL(geglu_fwd_kernel, dim3(GRID_FOR((size_t)M * (F / 8), 256)), dim3(256), 0, ST, (const bf16_t*)in, (bf16_t*)out, (size_t)M, F);
    return mi355_check_launch("geglu_fwd");
}
int mi355_geglu_bwd(const void* in, const void* dout, void* din, long long M, int F, void* stream) {
    if (!in || !dout || !din || (F & 7)) return mi355_fail(MI355_ERR_ARG, "geglu_bwd: bad arguments");
    ProfScope prof(FAM_ELEM, ST, (double)M * F * 2 * 5);
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3(GRID_FOR((size_t)M * (F / 8), 256)), dim3(256), 0, ST, (const bf16_t*)in, (const bf16_t*)dout, (bf16_t*)din, (size_t)M, F);
    return mi355_check_launch("geglu_bwd");
}
int mi355_silu_f32_to_bf16(const float* x, void* y, long long n, void* stream) {
    if (!x || !y) return mi355_fail(MI355_ERR_ARG, "silu: null pointer");
    hipLaunchKernelGGL(silu_f32_to_bf16_kernel, dim3(GRID_FOR(n, 256)), dim3(256), 0, ST, x, (bf16_t*)y, (size_t)n);
    return mi355_check_launch("silu_f32_to_bf16");
}
int mi355_silu_bwd_f32(const float* x, const float* dy, float* dx, long long n, void* stream) {
    if (!x || !dy || !dx) return mi355_fail(MI355_ERR_ARG, "silu_bwd: null pointer");
    hipLaunchKernelGGL(silu_bwd_f32_kernel, dim3(GRID_FOR(n, 256)), dim3(256), 0, ST, x, dy, dx, (size_t)n);
    return mi355_check_launch("silu_bwd_f32");
}
int mi355_timestep_embed(const float* t, void* out, int B, int dim, void* stream) {
    if (!t || !out || (dim & 1)) return mi355_fail(MI355_ERR_ARG, "timestep_embed: bad arguments");
    hipLaunchKernelGGL(timestep_embed_kernel, dim3((B * (dim / 2) + 255) / 256), dim3(256), 0, ST, t, (bf16_t*)out, B, dim);
    return mi355_check_launch("timestep_embed");
}
int mi355_cast_f32_to_bf16(const float* x, void* y, long long n, void* stream) {
    if (!x || !y || (n & 7)) return mi355_fail(MI355_ERR_ARG, "cast_f32_to_bf16: n must be a multiple of 8");
    ProfScope prof(FAM_ELEM, ST, (double)n * 6);
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(GRID_FOR(n / 8, 256)), dim3(256), 0, ST, x, (bf16_t*)y, (size_t)n / 8);
    return mi355_check_launch("cast_f32_to_bf16");
}
int mi355_cast_bf16_to_f32(const void* x, float* y, long long n, void* stream) {
    if (!x || !y) return mi355_fail(MI355_ERR_ARG, "cast_bf16_to_f32: null pointer");
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(GRID_FOR(n, 256)), dim3(256), 0, ST, (const bf16_t*)x, y, (size_t)n);
    return mi355_check_launch("cast_bf16_to_f32");
}
int mi355_transpose_bf16(const void* in, void* out, int G, int R, int C, void* stream) {
    if (!in || !out) return mi355_fail(MI355_ERR_ARG, "transpose: null pointer");
    ProfScope prof(FAM_ELEM, ST, (double)G * R * C * 4);
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((C + 31) / 32, (R + 31) / 32, G), dim3(256), 0, ST, (const bf16_t*)in, (bf16_t*)out, R, C);
    return mi355_check_launch("transpose_bf16");
}
int mi355_conv_weight_transpose(const void* w, void* wt, int Co, int taps, int Ci, void* stream) {
    if (!w || !wt) return mi355_fail(MI355_ERR_ARG, "conv_weight_transpose: null pointer");
    ProfScope prof(FAM_ELEM, ST, (double)Co * taps * Ci * 4);
    hipLaunchKernelGGL(conv_weight_transpose_kernel, dim3((Ci + 31) / 32, (Co + 31) / 32, taps), dim3(256), 0, ST, (const bf16_t*)w, (bf16_t*)wt, Co, taps, Ci);
    return mi355_check_launch("conv_weight_transpose");
}
int mi355_transpose_batched(const long long* jobs, int njobs, int total_tiles, void* stream) {
    if (!jobs || njobs <= 0 || total_tiles <= 0) return mi355_fail(MI355_ERR_ARG, "transpose_batched: bad arguments");
    ProfScope prof(FAM_ELEM, ST, (double)total_tiles * 64 * 64 * 4);
    hipLaunchKernelGGL(transpose_batched_kernel, dim3(total_tiles), dim3(256), 0, ST, jobs, njobs);
    return mi355_check_launch("transpose_batched");
}
int mi355_nchw_f32_to_nhwc_bf16(const float* x, void* y, int B, int Cs, int Cd, int HW, void* stream) {
    if (!x || !y || Cd < Cs) return mi355_fail(MI355_ERR_ARG, "nchw_to_nhwc: bad arguments");
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(GRID_FOR((size_t)B * HW * Cd, 256)), dim3(256), 0, ST, x, (bf16_t*)y, B, Cs, Cd, HW);
    return mi355_check_launch("nchw_to_nhwc");
}
int mi355_nhwc_f32_to_nchw_f32(const float* x, float* y, int B, int Cs, int Cd, int HW, void* stream) {
    if (!x || !y || Cd > Cs) return mi355_fail(MI355_ERR_ARG, "nhwc_to_nchw: bad arguments");
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3(GRID_FOR((size_t)B * HW * Cd, 256)), dim3(256), 0, ST, x, y, B, Cs, Cd, HW);
    return mi355_check_launch("nhwc_to_nchw");
}
int mi355_sumpool2x2_bf16(const void* dy, void* dx, int B, int H, int W, int C, void* stream) {
    if (!dy || !dx || (C & 7)) return mi355_fail(MI355_ERR_ARG, "sumpool2x2: bad arguments");
    ProfScope prof(FAM_ELEM, ST, (double)B * H * W * C * 2 * 5);
    hipLaunchKernelGGL(sumpool2x2_kernel, dim3(GRID_FOR((size_t)B * H * W * (C / 8), 256)), dim3(256), 0, ST, (const bf16_t*)dy, (bf16_t*)dx, B, H, W, C);
    return mi355_check_launch("sumpool2x2");
}
// loss must point at one float; it is zeroed here.
int mi355_mse_loss_fused(const float* pred, const float* target, float* grad, float* loss, long long n, void* stream) {
    if (!pred || !target || !loss || n <= 0) return mi355_fail(MI355_ERR_ARG, "mse_loss: bad arguments");
    mi355_zero_async(loss, sizeof(float), ST);
    ProfScope prof(FAM_ELEM, ST, (double)n * 12);
    hipLaunchKernelGGL(mse_fused_kernel, dim3(GRID_FOR(n, 1024)), dim3(256), 0, ST, pred, target, grad, loss, (size_t)n);
    return mi355_check_launch("mse_loss_fused");
}
int mi355_ddpm_add_noise(const float* x0, const float* noise, const long long* t, const float* alphas_cumprod,
                         float* out, int B, long long per_sample, void* stream) {
    if (!x0 || !noise || !t || !alphas_cumprod || !out) return mi355_fail(MI355_ERR_ARG, "ddpm_add_noise: null pointer");
    hipLaunchKernelGGL(ddpm_add_noise_kernel, dim3(GRID_FOR((size_t)B * per_sample, 256)), dim3(256), 0, ST, x0, noise, t, alphas_cumprod, out, B, (size_t)per_sample);
    return mi355_check_launch("ddpm_add_noise");
}
int mi355_ddim_step_cfg(const float* eps, const float* x, float* out, const float* coef, long long n, int cfg, void* stream) {
    if (!eps || !x || !out || !coef) return mi355_fail(MI355_ERR_ARG, "ddim_step: null pointer");
    hipLaunchKernelGGL(ddim_step_cfg_kernel, dim3(GRID_FOR(n, 256)), dim3(256), 0, ST, eps, x, out, coef, (size_t)n, cfg);
    return mi355_check_launch("ddim_step_cfg");
}
// out (one float) is NOT zeroed here so several buffers can be accumulated; zero it first with mi355_zero_f32.
int mi355_sumsq_f32(const float* g, float* out, long long n, void* stream) {
    if (!g || !out) return mi355_fail(MI355_ERR_ARG, "sumsq: null pointer");
    if (((uintptr_t)g) & 15) return mi355_fail(MI355_ERR_ARG, "sumsq: buffer must be 16-byte aligned");
    ProfScope prof(FAM_ELEM, ST, (double)n * 4);
    hipLaunchKernelGGL(sumsq_kernel, dim3(GRID_FOR(n / 4 + 1, 1024)), dim3(256), 0, ST, g, out, (size_t)n);
    return mi355_check_launch("sumsq");
}
// out += sum over the table's ranges (see sumsq_ranges_kernel): the parts of a gradient buffer that no producer summed.
int mi355_sumsq_ranges(const long long* table, int n, float* out, void* stream) {
    if (!table || !out || n <= 0) return mi355_fail(MI355_ERR_ARG, "sumsq_ranges: bad arguments");
    hipLaunchKernelGGL(sumsq_ranges_kernel, dim3(128, n), dim3(256), 0, ST, table, out);
    return mi355_check_launch("sumsq_ranges");
}
int mi355_sumsq_bf16(const void* g, float* out, long long n, void* stream) {
    if (!g || !out || (n & 7)) return mi355_fail(MI355_ERR_ARG, "sumsq_bf16: null pointer or n not a multiple of 8");
    if (((uintptr_t)g) & 15) return mi355_fail(MI355_ERR_ARG, "sumsq_bf16: buffer must be 16-byte aligned");
    ProfScope prof(FAM_ELEM, ST, (double)n * 2);
    hipLaunchKernelGGL(sumsq_bf16_kernel, dim3(GRID_FOR(n / 8 + 1, 1024)), dim3(256), 0, ST, (const bf16_t*)g, out, (size_t)n / 8);
    return mi355_check_launch("sumsq_bf16");
}
int mi355_sumsq_bf16_ordered(const void* g, float* out, float* partials, int max_partials, long long n, void* stream) {
    if (!g || !out || !partials || (n & 7)) return mi355_fail(MI355_ERR_ARG, "sumsq_bf16_ordered: null pointer or n not a multiple of 8");
    if (((uintptr_t)g) & 15) return mi355_fail(MI355_ERR_ARG, "sumsq_bf16_ordered: buffer must be 16-byte aligned");
    if (max_partials < 1) return mi355_fail(MI355_ERR_ARG, "sumsq_bf16_ordered: no room for partial sums");
    ProfScope prof(FAM_ELEM, ST, (double)n * 2);
    int grid = GRID_FOR(n / 8 + 1, 1024);
    if (grid > max_partials) grid = max_partials;
    hipLaunchKernelGGL(sumsq_bf16_partial_kernel, dim3(grid), dim3(256), 0, ST, (const bf16_t*)g, partials, (size_t)n / 8);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, ST, (const float*)partials, grid, out);
    return mi355_check_launch("sumsq_bf16_ordered");
}
int mi355_zero_f32(float* p, long long n, void* stream) {
    if (!p) return mi355_fail(MI355_ERR_ARG, "zero: null pointer");
    if (((uintptr_t)p) & 15) return mi355_fail(MI355_ERR_ARG, "zero: buffer must be 16-byte aligned");
    mi355_zero_async(p, (size_t)n * sizeof(float), ST);
    return mi355_check_launch("zero");
}
int mi355_zero_ranges(const long long* table, int n, void* stream) {
    if (!table || n <= 0) return mi355_fail(MI355_ERR_ARG, "zero_ranges: bad arguments");
    hipLaunchKernelGGL(zero_ranges_kernel, dim3(64, n), dim3(256), 0, ST, table);
    return mi355_check_launch("zero_ranges");
}
int mi355_copy_ranges(const long long* table, int n, int blocks_per_range, void* stream) {
    if (!table || n <= 0 || blocks_per_range <= 0) return mi355_fail(MI355_ERR_ARG, "copy_ranges: bad arguments");
    hipLaunchKernelGGL(copy_ranges_kernel, dim3(blocks_per_range, n), dim3(256), 0, ST, table);
    return mi355_check_launch("copy_ranges");
}
int mi355_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, const float* hyper, float* state,
                     long long n, void* stream) {
    if (!p || !g || !m || !v || !hyper || !state) return mi355_fail(MI355_ERR_ARG, "adamw: null pointer");
    if (n & 3) return mi355_fail(MI355_ERR_ARG, "adamw: n must be a multiple of 4 (pad the flat buffer)");
    ProfScope prof(FAM_ELEM, ST, (double)n * 30);
    // 4 consecutive 4 KB runs per block and non-temporal accesses: 5.5 TB/s over the 30 B/parameter of traffic
    constexpr int U = 4;
    hipLaunchKernelGGL((adamw_kernel<U, true>), dim3((unsigned)(((size_t)n / 4 + 256 * U - 1) / (256 * U))), dim3(256), 0, ST,
                       p, (const void*)g, m, v, (bf16_t*)p_bf16, hyper, state, (size_t)n);
    return mi355_check_launch("adamw");
}
// Same step with bf16 gradients (8-byte aligned, n a multiple of 4).
int mi355_adamw_step_g16(float* p, const void* g16, float* m, float* v, void* p_bf16, const float* hyper, float* state,
                         long long n, void* stream) {
    if (!p || !g16 || !m || !v || !hyper || !state) return mi355_fail(MI355_ERR_ARG, "adamw_g16: null pointer");
    if ((n & 3) || ((uintptr_t)g16 & 7)) return mi355_fail(MI355_ERR_ARG, "adamw_g16: n must be a multiple of 4, gradients 8-byte aligned");
    ProfScope prof(FAM_ELEM, ST, (double)n * 28);
    constexpr int U = 4;
    hipLaunchKernelGGL((adamw_kernel<U, true, true>), dim3((unsigned)(((size_t)n / 4 + 256 * U - 1) / (256 * U))), dim3(256), 0, ST,
                       p, g16, m, v, (bf16_t*)p_bf16, hyper, state, (size_t)n);
    return mi355_check_launch("adamw_g16");
}

// scratch floats: 2*B*D (normalised copies) + 2*B (norms) + B*B (logits/dlogits) + 2*B (lse)
long long mi355_contrastive_scratch_floats(int B, int D) { return 2LL * B * D + 2LL * B + 1LL * B * B + 2LL * B; }
int mi355_contrastive_loss_fused(const float* text, const float* image, const float* log_temp, float* loss,
                                 float* d_text, float* d_image, float* d_log_temp, float* scratch, int B, int D,
                                 void* stream) {
    if (!text || !image || !log_temp || !loss || !d_text || !d_image || !d_log_temp || !scratch)
        return mi355_fail(MI355_ERR_ARG, "contrastive: null pointer");
    if (B <= 0 || D <= 0 || D > 12288) return mi355_fail(MI355_ERR_ARG, "contrastive: bad shape");
    float* th = scratch; float* ih = th + (size_t)B * D; float* tn = ih + (size_t)B * D; float* in_ = tn + B;
    float* logits = in_ + B; float* lse = logits + (size_t)B * B;
    // five launches: normalise both inputs (and clear the accumulators), logits, both log-sum-exps, loss + dlogits,
    // both embedding gradients (and the clamp of the temperature gradient)
    ProfScope prof(FAM_ELEM, ST, 6.0 * B * B * D);
    hipLaunchKernelGGL(l2norm_pair_kernel, dim3(2 * B), dim3(256), 0, ST, text, image, th, ih, tn, in_, loss, d_log_temp, B, D);
    int gx = (B + 3) / 4; if (gx > 64) gx = 64;
    hipLaunchKernelGGL(clip_logits_kernel, dim3(gx, B), dim3(256), 0, ST, th, ih, log_temp, logits, B, D);
    hipLaunchKernelGGL(clip_lse_kernel, dim3(B, 2), dim3(256), 0, ST, logits, lse, B);
    hipLaunchKernelGGL(clip_dlogits_kernel, dim3(B), dim3(256), 0, ST, logits, lse, loss, d_log_temp, B);
    hipLaunchKernelGGL(clip_dembed_kernel, dim3(B, 2), dim3(256), D * sizeof(float), ST, logits, th, ih, tn, in_, log_temp,
                       d_text, d_image, d_log_temp, B, D);
    return mi355_check_launch("contrastive_loss_fused");
}

int mi355_lora_down(const void* x, const float* A, float* t, int M, int K, int R, void* t16, float scale16, void* stream) {
    if (!x || !A || !t || (K & 7) || R < 1 || R > 16) return mi355_fail(MI355_ERR_ARG, "lora_down: bad arguments (K%%8, 1<=R<=16)");
    hipLaunchKernelGGL(lora_down_kernel, dim3((M + 3) / 4), dim3(256), 0, ST, (const bf16_t*)x, A, t, M, K, R, (bf16_t*)t16, scale16);
    return mi355_check_launch("lora_down");
}
int mi355_lora_dt(const void* dy, const float* Bm, float* dt, int M, int N, int R, int S, float alpha, void* dt16, void* stream) {
    if (!dy || !Bm || !dt || (N & 7) || R < 1 || S < 1 || R * S > 16 || N % S || ((N / S) & 7) || (S > 1 && R != 4))
        return mi355_fail(MI355_ERR_ARG, "lora_dt: bad arguments (R*S <= 16; stacked adapters need rank 4)");
    if (R == 4) hipLaunchKernelGGL(lora_dt_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, ST, (const bf16_t*)dy, Bm, dt, M, N, R, S, alpha, (bf16_t*)dt16);
    else hipLaunchKernelGGL(lora_dt_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, ST, (const bf16_t*)dy, Bm, dt, M, N, R, S, alpha, (bf16_t*)dt16);
    return mi355_check_launch("lora_dt");
}
// flags: bit 0 = result laid out [R][C] instead of [C][R]; bit 1 = accumulate into `out` (the caller zeroed it, or
// wants the sum with its previous contents) instead of overwriting it
int mi355_lora_outer(const void* y, const float* t, float* out, int M, int C, int R, float scale, int flags,
                     void* stream) {
    if (!y || !t || !out || (C & 7) || R < 1 || R > 16 || M < 1) return mi355_fail(MI355_ERR_ARG, "lora_outer: bad arguments");
    if (!(flags & 2)) mi355_zero_async(out, (size_t)C * R * sizeof(float), ST);
    const int nc_all = C / 8;
    const int NC = nc_all < 256 ? nc_all : 256;               // column chunks per workgroup
    const int RA = R == 4 ? 4 : 16;
    int RL = 256 / NC;                                         // row lanes, as many as 48 KB of LDS partials allow
    const int rl_cap = 1 + (48 * 1024) / (NC * 8 * RA * (int)sizeof(float));
    if (RL > rl_cap) RL = rl_cap;
    const int gy = (nc_all + NC - 1) / NC;
    int rows = (M + 63) / 64;                                  // ~64 workgroups, at least 16 rows per row lane
    if (rows < 16 * RL) rows = 16 * RL;
    const int gx = (M + rows - 1) / rows;
    const size_t smem = (size_t)(RL - 1) * NC * 8 * RA * sizeof(float);
    if (R == 4)
        hipLaunchKernelGGL(lora_outer_kernel<true>, dim3(gx, gy), dim3(256), smem, ST, (const bf16_t*)y, t, out, M, C, R, scale,
                           rows, NC, RL, flags & 1);
    else
        hipLaunchKernelGGL(lora_outer_kernel<false>, dim3(gx, gy), dim3(256), smem, ST, (const bf16_t*)y, t, out, M, C, R, scale,
                           rows, NC, RL, flags & 1);
    return mi355_check_launch("lora_outer");
}
static int drop_params(float p, unsigned* thresh, float* scale, const char* what) {
    if (!(p >= 0.f) || p >= 1.f) return mi355_fail(MI355_ERR_ARG, "%s: dropout probability %g", what, (double)p);
    *thresh = (unsigned)((double)p * 4294967296.0);
    *scale = 1.f / (1.f - p);
    return 0;
}
int mi355_dropout_add_bf16(const void* x, const void* residual, void* y, long long n, float p, unsigned seed, void* stream) {
    if (!x || !y || n < 8 || (n & 7)) return mi355_fail(MI355_ERR_ARG, "dropout_add: bad arguments (n a multiple of 8)");
    unsigned th; float sc;
    if (int rc = drop_params(p, &th, &sc, "dropout_add")) return rc;
    ProfScope prof(FAM_ELEM, ST, (double)n * (residual ? 6 : 4));
    hipLaunchKernelGGL(dropout_add_kernel, dim3(GRID_FOR((size_t)n / 8, 256)), dim3(256), 0, ST, (const bf16_t*)x,
                       (const bf16_t*)residual, (bf16_t*)y, (size_t)n / 8, th, sc, seed);
    return mi355_check_launch("dropout_add");
}
int mi355_gelu_dropout_fwd(const void* x, void* y, long long n, float p, unsigned seed, void* stream) {
    if (!x || !y || n < 8 || (n & 7)) return mi355_fail(MI355_ERR_ARG, "gelu_dropout_fwd: bad arguments (n a multiple of 8)");
    unsigned th; float sc;
    if (int rc = drop_params(p, &th, &sc, "gelu_dropout")) return rc;
    ProfScope prof(FAM_ELEM, ST, (double)n * 4);
    hipLaunchKernelGGL(gelu_dropout_kernel<false>, dim3(GRID_FOR((size_t)n / 8, 256)), dim3(256), 0, ST, (const bf16_t*)x,
                       (const bf16_t*)nullptr, (bf16_t*)y, (size_t)n / 8, th, sc, seed);
    return mi355_check_launch("gelu_dropout_fwd");
}
int mi355_gelu_dropout_bwd(const void* x, const void* dy, void* dx, long long n, float p, unsigned seed, void* stream) {
    if (!x || !dy || !dx || n < 8 || (n & 7)) return mi355_fail(MI355_ERR_ARG, "gelu_dropout_bwd: bad arguments (n a multiple of 8)");
    unsigned th; float sc;
    if (int rc = drop_params(p, &th, &sc, "gelu_dropout")) return rc;
    ProfScope prof(FAM_ELEM, ST, (double)n * 6);
    hipLaunchKernelGGL(gelu_dropout_kernel<true>, dim3(GRID_FOR((size_t)n / 8, 256)), dim3(256), 0, ST, (const bf16_t*)x,
                       (const bf16_t*)dy, (bf16_t*)dx, (size_t)n / 8, th, sc, seed);
    return mi355_check_launch("gelu_dropout_bwd");
}
int mi355_l2norm_rows_f32(const float* x, float* y, float* norms, int B, int D, void* stream) {
    if (!x || !y || !norms || B < 1 || D < 1) return mi355_fail(MI355_ERR_ARG, "l2norm_rows: bad arguments");
    hipLaunchKernelGGL(l2norm_f32_kernel<false>, dim3(B), dim3(256), 0, ST, x, (const float*)nullptr, y, norms, D);
    return mi355_check_launch("l2norm_rows");
}
int mi355_l2norm_rows_bwd_f32(const float* y, const float* norms, const float* dy, float* dx, int B, int D, void* stream) {
    if (!y || !norms || !dy || !dx || B < 1 || D < 1) return mi355_fail(MI355_ERR_ARG, "l2norm_rows_bwd: bad arguments");
    hipLaunchKernelGGL(l2norm_f32_kernel<true>, dim3(B), dim3(256), 0, ST, y, dy, dx, (float*)norms, D);
    return mi355_check_launch("l2norm_rows_bwd");
}
int mi355_cast_ranges_f32_to_bf16(const long long* table, int n, int blocks_per_range, void* stream) {
    if (!table || n < 1 || blocks_per_range < 1 || blocks_per_range > 4096)
        return mi355_fail(MI355_ERR_ARG, "cast_ranges: bad arguments");
    hipLaunchKernelGGL(cast_ranges_kernel, dim3(blocks_per_range, n), dim3(256), 0, ST, table);
    return mi355_check_launch("cast_ranges");
}
int mi355_lora_pack(const long long* table, int n, void* stream) {
    if (!table || n < 1) return mi355_fail(MI355_ERR_ARG, "lora_pack: bad arguments");
    hipLaunchKernelGGL(lora_pack_kernel, dim3(8, n), dim3(256), 0, ST, table);
    return mi355_check_launch("lora_pack");
}
// Batched strided 2-D copy of fp32 blocks: table holds n records {src, dst, rows, cols, src_ld, dst_ld} (int64).
int mi355_copy2d_f32_batched(const long long* table, int n, void* stream) {
    if (!table || n < 1) return mi355_fail(MI355_ERR_ARG, "copy2d_batched: bad arguments");
    hipLaunchKernelGGL(copy2d_batched_kernel, dim3(16, n), dim3(256), 0, ST, table);
    return mi355_check_launch("copy2d_batched");
}

}  // extern "C"
