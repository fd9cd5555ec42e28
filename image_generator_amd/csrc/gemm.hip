// bf16 MFMA GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
//   out[M,N] = epilogue( sum_k A(m,k) * Bw[n,k] )
//
// A is either a dense row-major matrix (linear layers, 1x1 convolutions) or the *virtual* im2col
// matrix of an NHWC activation (3x3 convolutions, stride 1/2, optional nearest-2x upsampled or
// zero-dilated input), gathered on the fly by the tile loader - the im2col matrix never exists in
// HBM.  Bw is always row-major [N][K] (K contiguous), which is how linear weights are stored and
// how conv weights are stored as [Cout][ky][kx][Cin].
//
// Tile: 128 x BN x 64 per 256-thread workgroup (4 waves as 2x2, v_mfma_f32_16x16x32_bf16),
// register-staged double-buffered LDS with an XOR swizzle so ds_read_b128 fragment reads are
// conflict-free.  BN=160 divides every channel count of the SD-v1 denoiser (320/640/960/1280/...).
// The epilogue stages fp32 accumulators through LDS so that bias / broadcast row bias (timestep
// embedding) / LoRA rank-r update / residual are applied with 16-byte coalesced accesses.
#include "common.h"
#include <mutex>
#include <stdlib.h>

struct ConvGeom {
    int B, H, W, C;   // stored NHWC input
    int Ho, Wo;       // output spatial size (M = B*Ho*Wo)
    int stride;       // 1 or 2
    int ups;          // 1: input is read through a nearest 2x upsample
    int dil;          // 1: input is read through a zero-dilation by 2 (data-gradient of a stride-2 conv)
    int ksize;        // 1 or 3
    int pad;          // zero padding on the top/left (ksize/2, or 0 for the bottom/right-only padding of the VAE downsampler)
};

struct GemmArgs {
    const bf16_t* A;
    const bf16_t* Bw;
    int M, N, K;
    int lda, ldb;
    void* out;
    int ldo;
    int out_fp32;
    const float* bias;       // [N] or null
    const float* rowbias;    // [M / rowbias_div][ld_rowbias] or null
    int rowbias_div, ld_rowbias;
    int rowbias_shift;       // log2(rowbias_div) when it is a power of two (no integer division per chunk), else -1
    const bf16_t* residual;  // [M][ldr] or null
    int ldr;
    float alpha;
    const float* lora_t;     // [M][lora_r] fp32 or null  (x A^T)
    const float* lora_b;     // [N][lora_r] fp32          (B)
    int lora_r;
    int lora_ldt;            // row stride of lora_t (floats)
    int lora_seg;            // > 0: outputs per row-stacked adapter (each reads its own r columns of lora_t)
    int lora_rmajor;         // lora_b is [r][N] instead of [N][r]
    int lora_t_bf16;         // lora_t points at bf16 [M][lora_ldt], lora_ldt 8 or 16
    float lora_alpha;
    float* partial;          // split-K slabs [splitk][M][N] fp32
    int splitk, kt_per_split;
    int bm;                  // rows per tile (128, or 256 for large grids on the LDS-DMA kernel)
    int geglu;               // fused GEGLU: N = 2F weight rows (hidden | gate), out [M][F] = h * gelu(g)
    bf16_t* preact;          // geglu only, optional: [M][2F] pre-activation (hidden | gate) kept for the backward pass
    int tiles_m, tiles_n;
    ConvGeom g;
    // AMODE_DENSE_CAT: A is the column concatenation [A | A2] read in place - A holds the first K1 columns (row stride
    // lda), A2 the other K - K1 (row stride lda2); K1 is a multiple of the 64-deep k-tile
    const bf16_t* A2;
    int K1, lda2;
    unsigned a2_bytes;
    // two-destination output (any kernel): columns n >= N1 go to out2 [M][N - N1] instead; N1 is a multiple of every tile
    // width, so a tile never straddles the two
    void* out2;
    int N1;
    // GEGLU backward in the epilogue (any kernel): the product is the gradient dy [M][N = F] of the gated output
    // y = h * gelu(g); instead of dy the epilogue writes gb_out [M][2F] = (dy * gelu(g) | dy * h * gelu'(g)) from the
    // pre-activation gb_pre [M][2F] = (h | g) the forward pass kept
    const bf16_t* gb_pre;
    bf16_t* gb_out;
    // k-major B (LDS-DMA kernel, BN = 160 only): Bw holds B[k][n] with n contiguous - a weight [K][N] used untransposed
    // by the data-gradient product dx = dy W.  bt_ld: elements between consecutive k.  For a convolution the product runs
    // over (tap, channel) of the stored weight [C][taps][N] (C = g.C output channels of the layer, N its input
    // channels) with the taps in reverse order: k = tap * C + c reads row c, tap taps-1-tap.
    int bt, bt_ld;
    // AMODE_DENSE_LN - LayerNorm folded into the products on either side of it (no pass over the activations):
    //   producer: rowstat_out [M][tiles_n][2] receives, per row and column tile, the sum and the sum of squares of the
    //             bf16 values this launch stores (the rows of the residual stream x it writes);
    //   consumer: A is that x, un-normalised, Bw holds gamma-scaled weights W' = bf16(gamma (.) W); with the row's
    //             mean and rstd from ln_part [M][ln_nparts][2] (the producer's partial sums, C = K values per row)
    //             out = rstd * (A W'^T - mean * ln_cs) + bias,  ln_cs[n] = sum_k W'[n][k],  bias = W beta + b,
    //             which is LayerNorm(x) W^T + b.
    // GroupNorm statistics out of the producing epilogue (AMODE_DENSE_LN / AMODE_CONV_GS): gstat_out [tiles_m][N / gs_unit][2]
    // receives, per row tile and per run of gs_unit (8 or 10) adjacent output channels, the sum and the sum of squares of
    // the bf16 values this launch stores.  A row tile must lie inside one sample; the consumer's groups are whole runs.
    float* gstat_out;
    int gs_unit;
    float* rowstat_out;
    const float* ln_part;
    const float* ln_cs;
    int ln_nparts;
    float ln_invc, ln_eps;
};

// AMODE_DENSE_LORA: dense A with the low-rank term in the epilogue - a kernel variant of its own so that every other
// GEMM keeps the short epilogue (carrying the low-rank code path cost them 0.3 ms per training step)
// AMODE_DENSE_LN: dense A with the row statistics of a LayerNorm emitted and / or applied in the epilogue (see
// GemmArgs) - again a kernel variant of its own, so that the other GEMMs' epilogues do not carry it
// AMODE_CONV_GS: the LDS-DMA convolution whose epilogue also emits GroupNorm statistics (gstat_out); dense products that
// emit them run as AMODE_DENSE_LN
enum { AMODE_DENSE = 0, AMODE_CONV_FAST = 1, AMODE_CONV_GENERIC = 2, AMODE_DENSE_LORA = 3, AMODE_DENSE_CAT = 4, AMODE_DENSE_LN = 5,
       AMODE_CONV_GS = 6 };
__host__ __device__ constexpr bool amode_stats(int amode) { return amode == AMODE_DENSE_LN || amode == AMODE_CONV_GS; }
__host__ __device__ constexpr bool amode_dense(int amode) {
    return amode == AMODE_DENSE || amode == AMODE_DENSE_LORA || amode == AMODE_DENSE_CAT || amode == AMODE_DENSE_LN;
}

__device__ __forceinline__ float gemm_gelu_erf(float x) { return gelu_erf_f(x); }

// Shared tail of both GEMM kernels: split-K slab store, or the LDS-staged vectorised epilogue.
// Low-rank term of 8 adjacent outputs: v[e] += lora_alpha * sum_r t[m][r] * b(n + e, r).  `lora_seg` > 0: the columns
// are row-stacked adapters of `lora_seg` outputs each and adapter s reads t[m][s*r .. s*r + r) (fused projections).
// `fb` / `fld`: where the factors of column n are read from - global memory (fb = lora_b + n * r resp. lora_b + n,
// fld = N) or the per-tile copy the epilogue staged in LDS (fld = tile width).  `trow`: the row's 16 fp32 low-rank
// activations staged in LDS, or null to read them from global memory.
__device__ __forceinline__ void lora_apply(const GemmArgs& p, int m, int n, float (&v)[8], const float* fb, int fld,
                                           const float* trow) {
    // adapter index of column n (at most 4 row-stacked adapters) without an integer division
    const int toff = p.lora_seg ? ((n >= p.lora_seg) + (n >= 2 * p.lora_seg) + (n >= 3 * p.lora_seg)) * p.lora_r : 0;
    float tv[16];
    if (trow) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 t4 = *(const f32x4*)(trow + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) tv[4 * q + r] = t4[r];
        }
    } else if (p.lora_t_bf16) {                 // bf16 [M][ldt] (ldt = 8 or 16), e.g. the output of a skinny GEMM
        const bf16_t* t = (const bf16_t*)p.lora_t + (size_t)m * p.lora_ldt;
        const bf16x8 lo = *(const bf16x8*)t;
        const bf16x8 hi = p.lora_ldt > 8 ? *(const bf16x8*)(t + 8) : lo;
#pragma unroll
        for (int r = 0; r < 8; ++r) { tv[r] = (float)lo[r]; tv[8 + r] = (float)hi[r]; }
    } else {
        const float* t = p.lora_t + (size_t)m * p.lora_ldt;
#pragma unroll
        for (int r = 0; r < 16; ++r) tv[r] = r < p.lora_ldt ? t[r] : 0.f;
    }
    if (trow && !p.lora_rmajor) {               // LDS copy of [N][4] factors, staged as [4][tile width]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = toff == 0 ? tv[r] : toff == 4 ? tv[4 + r] : toff == 8 ? tv[8 + r] : tv[12 + r];
            const float tr = a * p.lora_alpha;
            const float* bp = fb + r * fld;
            const f32x4 b0 = *(const f32x4*)bp, b1 = *(const f32x4*)(bp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += tr * b0[e]; v[4 + e] += tr * b1[e]; }
        }
    } else if (p.lora_rmajor) {                 // factors [r][N]: eight adjacent columns are two 16-byte loads per rank
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (r < p.lora_r) {
                const float tr = tv[r] * p.lora_alpha;
                const float* bp = fb + (size_t)r * fld;
                const f32x4 b0 = *(const f32x4*)bp, b1 = *(const f32x4*)(bp + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += tr * b0[e]; v[4 + e] += tr * b1[e]; }
            }
        }
    } else if (p.lora_r == 4) {                 // factors [N][4]: one 16-byte load per column
        float t4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // toff is 0, 4, 8 or 12: pick without dynamic register indexing
            const float a = toff == 0 ? tv[r] : toff == 4 ? tv[4 + r] : toff == 8 ? tv[8 + r] : tv[12 + r];
            t4[r] = a * p.lora_alpha;
        }
        const f32x4* bp = (const f32x4*)fb;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const f32x4 bv = bp[e];
            v[e] += t4[0] * bv[0] + t4[1] * bv[1] + t4[2] * bv[2] + t4[3] * bv[3];
        }
    } else {
        const float* t = p.lora_t + (size_t)m * p.lora_ldt + toff;     // generic rank: fp32 factors only
        for (int r = 0; r < p.lora_r; ++r) {
            const float tr = t[r] * p.lora_alpha;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += tr * p.lora_b[(size_t)(n + e) * p.lora_r + r];
        }
    }
}

// LayerNorm fold, consumer side (see GemmArgs): mean and rstd of the tile's BM rows from the producer's partial sums,
// into LDS beyond the operand stages.  Called at kernel start - the loads' round trip hides behind the first k-tiles
// (done in the epilogue it was exposed once per tile: +8 us per tile on the one-workgroup-per-CU kernel).
template <int BM, int NTHREADS>
__device__ __forceinline__ void ln_row_stats(const GemmArgs& p, int m0, float* rstat, int tid) {
    if (p.ln_part == nullptr) return;
    for (int r = tid; r < BM; r += NTHREADS) {
        const int m = m0 + r;
        float s0 = 0.f, s1 = 0.f;
        if (m < p.M) {
            const float* src = p.ln_part + (size_t)m * p.ln_nparts * 2;
            for (int q = 0; q < p.ln_nparts; ++q) { const f32x2 pv = *(const f32x2*)(src + 2 * q); s0 += pv[0]; s1 += pv[1]; }
        }
        const float mean = s0 * p.ln_invc;
        float var = s1 * p.ln_invc - mean * mean;
        if (var < 0.f) var = 0.f;
        rstat[2 * r] = mean;
        rstat[2 * r + 1] = rsqrtf(var + p.ln_eps);
    }
}

// Bias (every kernel) and, for the LayerNorm fold's consumer side, column sum of tile column `tid` (threads below BN),
// fetched at kernel start and kept in two registers through the k-loop; the epilogue stages them in LDS.  Gated tiles: column c < BN/2 is hidden
// unit f0 + c, column c >= BN/2 the gate of unit f0 + c - BN/2 (weight row F + that).
template <int BN>
__device__ __forceinline__ void ln_fold_column(const GemmArgs& p, int n0, int tid, float& ln_c, float& ln_b) {
    ln_c = 0.f; ln_b = 0.f;
    if ((p.ln_part == nullptr && p.bias == nullptr) || tid >= BN) return;
    const int F = p.N >> 1, f0 = (n0 / BN) * (BN / 2);
    const int n = p.geglu ? (tid < BN / 2 ? f0 + tid : F + f0 + tid - BN / 2) : n0 + tid;
    const bool ok = p.geglu ? (tid < BN / 2 ? f0 + tid < F : f0 + tid - BN / 2 < F) : n < p.N;
    if (ok) { if (p.ln_part) ln_c = p.ln_cs[n]; if (p.bias) ln_b = p.bias[n]; }
}

// Convolutions contract one 64-channel slice through all of its taps at a time: k-tile kt of a 3x3 layer is (slice
// kt / 9, tap kt % 9) - computed from kt with a multiply and a shift, no state carried from tile to tile (counters
// advanced inside the request lambdas ended up in scratch memory: +1.9 ms per train step).  1x1: slice kt, tap 0.
// Tap-major order (all channels of tap 0, then tap 1, ...) re-reads a pixel's neighbourhood a whole pass over the
// channels apart: at 640 channels and more the rows a tile needs no longer stay in the XCD's L2 between two taps
// (measured 8.5 x the operand bytes fetched at 640 -> 320 channels on a 32 x 32 map, 1.6 x in this order).
__device__ __forceinline__ void conv_ktile(int kt, int ksize, int& slice, int& ky, int& kx) {
    if (ksize == 3) {
        slice = (kt * 7282) >> 16;              // kt / 9 for kt < 16384 (K = 9 * 64 * 16384 channels: far beyond any layer)
        const int tap = kt - slice * 9;
        ky = (tap * 11) >> 5;                   // tap / 3 for tap < 9
        kx = tap - ky * 3;
    } else {
        slice = kt; ky = 0; kx = 0;
    }
}

// The same in two halves for the kernels with LDS to spare (one request burst at kernel start, no register and no
// wait until the k-loop is over): the tile's rows of ln_part are ONE contiguous run of [BM][nparts][2] floats, fetched
// by the LDS-DMA as lane-linear 1 KiB pieces (rows past M come back as zeros); ln_stats_finish turns them into mean and
// rstd behind the k-loop's closing vmcnt(0) + barrier.  The requests are older than every operand request of the
// wave, so the counted waits of the k-loops still retire what they mean to retire.
template <int BM, int NW>
__device__ __forceinline__ void ln_stats_request(const GemmArgs& p, int m0, char* raw, int wave, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (p.ln_part == nullptr) return;
    const int rows = p.M - m0 < BM ? p.M - m0 : BM;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.ln_part + (size_t)m0 * p.ln_nparts * 2), 0, (unsigned)(rows * p.ln_nparts * 8), 0x00020000);
    const int pieces = (BM * p.ln_nparts * 8 + 1023) >> 10;
    for (int pc = wave; pc < pieces; pc += NW)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(raw + pc * 1024), 16,
                                                 pc * 1024 + lane * 16, 0, 0, 0);
#endif
}
// ALIAS: rstat overlays the fetched partial sums (the one-workgroup-per-CU kernel has 16 KiB beside its slabs and
// [256][8][2] partial sums fill them): every thread sums its row first, a barrier, then the results are written.
template <int BM, int NTHREADS, bool ALIAS = false>
__device__ __forceinline__ void ln_stats_finish(const GemmArgs& p, const char* raw, float* rstat, int tid) {
    static_assert(!ALIAS || BM <= NTHREADS, "aliased statistics: one row per thread");
    if (p.ln_part == nullptr) return;           // (uniform over the workgroup)
    for (int r = tid; r < (ALIAS ? NTHREADS : BM); r += NTHREADS) {
        float s0 = 0.f, s1 = 0.f;
        if (r < BM) {
            const float* src = (const float*)raw + (size_t)r * p.ln_nparts * 2;
            for (int q = 0; q < p.ln_nparts; ++q) { const f32x2 pv = *(const f32x2*)(src + 2 * q); s0 += pv[0]; s1 += pv[1]; }
        }
        if (ALIAS) __syncthreads();
        if (r < BM) {
            const float mean = s0 * p.ln_invc;
            float var = s1 * p.ln_invc - mean * mean;
            if (var < 0.f) var = 0.f;
            rstat[2 * r] = mean;
            rstat[2 * r + 1] = rsqrtf(var + p.ln_eps);
        }
    }
}

template <int BN, int TM, int TN, int NTHREADS = 256, int BM = 128, bool LORA = false, bool LN = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[TM][TN], char* smem, int m0, int n0,
                                              int split, int wm0, int wn0, int wave, int tid, int fr, int fq,
                                              const float* rstat = nullptr, float ln_c = 0.f, float ln_b = 0.f) {
    constexpr int EP_LD = BN + 4;
    // ---- split-K: raw fp32 partial slab, the reduce kernel applies the epilogue
    if (p.splitk > 1) {
        float* dst = p.partial + (size_t)split * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm0 + i * 16 + fq * 4 + r;
                    const int n = n0 + wn0 + j * 16 + fr;
                    if (m < p.M && n < p.N) dst[(size_t)m * p.N + n] = acc[i][j][r];
                }
        return;
    }

    // ---- epilogue through LDS, 64 rows at a time
    char* outp = (char*)p.out;               // destination of this tile's columns (two-destination output: see GemmArgs)
    int ldo = p.ldo, nsh = 0;
    if (p.out2 != nullptr && n0 >= p.N1) { outp = (char*)p.out2; ldo = p.N - p.N1; nsh = p.N1; }
    float* ep = (float*)smem;
    // Low-rank factors of this tile's BN columns, staged once behind the staging rows: read from global memory per
    // 8-column chunk they cost 128 B of L1 traffic per 16 B of output (measured +87 us on a 38 us GEMM).
    float* lf = ep + 64 * EP_LD;
    float* lt = lf + 16 * BN;                  // [64][16] low-rank activations of the current 64 rows (fp32)
    const bool lora_lds = LORA && p.lora_t && (p.lora_rmajor || p.lora_r == 4);
    if (lora_lds) {
        if (p.lora_rmajor) {
            for (int i = tid; i < p.lora_r * BN; i += NTHREADS) {
                const int r = i / BN, c = i - r * BN;
                lf[i] = n0 + c < p.N ? p.lora_b[(size_t)r * p.N + n0 + c] : 0.f;
            }
        } else {
            // [N][4] factors are staged transposed ([4][BN], like the [r][N] form): read per column they would sit
            // 128 B apart, two bank groups for a whole wave
            for (int i = tid; i < BN * 4; i += NTHREADS)
                lf[(i & 3) * BN + (i >> 2)] = n0 + (i >> 2) < p.N ? p.lora_b[(size_t)n0 * 4 + i] : 0.f;
        }
    }
    // LayerNorm fold: the tile's column sums and biases staged once (lf is free in this variant) - loaded per 8-column
    // chunk from global memory they put one more L1 round trip into every iteration of the chunk walk.
    // Layout: cs [BN] | bias [BN]; gated tiles hold hidden units in columns [0, BN/2) and their gates behind them
    const bool ln_fold = LN && p.ln_part != nullptr;
    // the bias of the tile's columns is staged in LDS too (every kernel but the low-rank variant, whose factors live
    // there): fetched per 8-column chunk it is one more L1 round trip in every iteration of the chunk walk - sampling
    // step 33.9 -> 33.4 ms (its K = 320 products are all epilogue), train step +-0.  A per-sample bias staged the same
    // way where a tile lies inside one sample: +0.1 ms on both, not kept.
    const bool bias_lds = !LORA && p.bias != nullptr;
    if ((ln_fold || bias_lds) && tid < BN) {      // (fetched by ln_fold_column() at kernel start: no round trip here)
        lf[tid] = ln_c;
        lf[BN + tid] = ln_b;
    }
    // low-rank activations of all BM rows, four per thread and 64-row group (zero past lora_ldt), fetched up front so
    // that their global-memory latency is paid once and not inside every group's barrier interval
    f32x4 tpre[BM / 64];
    if (lora_lds && tid < 256) {
#pragma unroll
        for (int h = 0; h < BM / 64; ++h) {
            const int q4 = (tid & 3) * 4, m = m0 + h * 64 + (tid >> 2);
            f32x4 t4 = {0.f, 0.f, 0.f, 0.f};
            if (m < p.M && q4 < p.lora_ldt) {
                if (p.lora_t_bf16) {
                    const bf16x4 tb = *(const bf16x4*)((const bf16_t*)p.lora_t + (size_t)m * p.lora_ldt + q4);
                    t4 = f32x4{(float)tb[0], (float)tb[1], (float)tb[2], (float)tb[3]};
                } else {
                    t4 = *(const f32x4*)(p.lora_t + (size_t)m * p.lora_ldt + q4);
                }
            }
            tpre[h] = t4;
        }
    }
    constexpr int TPR = NTHREADS / 64;                          // lanes per row of the row-owned chunk walk (LN variants)
    // GroupNorm statistics: after each 64-row group the values as stored (written back into the staging rows) are added
    // up by column - a thread owns four adjacent columns of every RP-th row - so eight registers hold its sums over the
    // tile (per-chunk sums kept by the lanes that store the chunks took 20 registers and spilled in the 256-register kernels)
    f32x4 gq0 = {0.f, 0.f, 0.f, 0.f}, gq1 = {0.f, 0.f, 0.f, 0.f};
    const bool gs_emit = LN && p.gstat_out != nullptr;
    for (int h = 0; h < BM / 64; ++h) {
        __syncthreads();
        if (lora_lds && tid < 256) *(f32x4*)(lt + (tid >> 2) * 16 + (tid & 3) * 4) = tpre[h];
        // the wave stages those of its 16-row blocks that lie inside this 64-row group
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int wrow = wm0 + i * 16 - h * 64;
            if (wrow >= 0 && wrow < 64) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        ep[(wrow + fq * 4 + r) * EP_LD + wn0 + j * 16 + fr] = acc[i][j][r];
            }
        }
        __syncthreads();
        constexpr int NC = BN / 8;
        if (p.geglu) {
            // tile columns [0, BN/2) are hidden units f0 .. f0 + BN/2, columns [BN/2, BN) their gates
            const int F = p.N >> 1, f0 = (n0 / BN) * (BN / 2);
            for (int idx = tid; idx < 64 * (NC / 2); idx += NTHREADS) {
                const int row = idx / (NC / 2), c8 = idx - row * (NC / 2);
                const int m = m0 + h * 64 + row, f = f0 + c8 * 8;
                if (m >= p.M || f >= F) continue;
                const float* er = ep + row * EP_LD + c8 * 8;
                f32x4 hv[2] = {*(const f32x4*)er, *(const f32x4*)(er + 4)};
                f32x4 gv[2] = {*(const f32x4*)(er + BN / 2), *(const f32x4*)(er + BN / 2 + 4)};
                if (p.alpha != 1.f) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) { hv[q] *= p.alpha; gv[q] *= p.alpha; }
                }
                if (ln_fold) {
                    const float mu = rstat[2 * (h * 64 + row)], rs = rstat[2 * (h * 64 + row) + 1];
                    const f32x4* ch = (const f32x4*)(lf + c8 * 8);
                    const f32x4* cg = (const f32x4*)(lf + BN / 2 + c8 * 8);
                    const f32x4* bh = (const f32x4*)(lf + BN + c8 * 8);
                    const f32x4* bg = (const f32x4*)(lf + BN + BN / 2 + c8 * 8);
#pragma unroll
                    for (int q = 0; q < 2; ++q) { hv[q] = (hv[q] - ch[q] * mu) * rs + bh[q]; gv[q] = (gv[q] - cg[q] * mu) * rs + bg[q]; }
                } else if (bias_lds) {
                    const f32x4* bh = (const f32x4*)(lf + BN + c8 * 8);
                    const f32x4* bg = (const f32x4*)(lf + BN + BN / 2 + c8 * 8);
#pragma unroll
                    for (int q = 0; q < 2; ++q) { hv[q] += bh[q]; gv[q] += bg[q]; }
                } else if (p.bias) {
                    const f32x4* bh = (const f32x4*)(p.bias + f);
                    const f32x4* bg = (const f32x4*)(p.bias + F + f);
#pragma unroll
                    for (int q = 0; q < 2; ++q) { hv[q] += bh[q]; gv[q] += bg[q]; }
                }
                bf16x8 ov;
                if (p.preact) {     // training: the gate is applied to the bf16-rounded values the backward pass will read
                    bf16x8 hb, gb;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { hb[e] = (bf16_t)hv[e >> 2][e & 3]; gb[e] = (bf16_t)gv[e >> 2][e & 3]; }
                    *(bf16x8*)(p.preact + (size_t)m * p.N + f) = hb;
                    *(bf16x8*)(p.preact + (size_t)m * p.N + F + f) = gb;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ov[e] = (bf16_t)((float)hb[e] * gemm_gelu_erf((float)gb[e]));
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ov[e] = (bf16_t)(hv[e >> 2][e & 3] * gemm_gelu_erf(gv[e >> 2][e & 3]));
                }
                *(bf16x8*)((bf16_t*)p.out + (size_t)m * p.ldo + f) = ov;
            }
            continue;
        }
        // chunk walk.  Plain: consecutive lanes run along a row's chunks and on into the next row.  LN (the variants that
        // emit statistics): NTHREADS / 64 adjacent lanes own a row and take every TPR-th chunk of it, so the row's sums
        // - and, over the tile's row groups, each chunk's column sums - stay in their registers
        // (the row-owned walk in every kernel: 33.83 against 33.67 ms per train step, sampling step +-0 - kept to LN)
        const int lrow = tid / TPR;
        float rs0 = 0.f, rs1 = 0.f;
        for (int it = LN ? tid % TPR : tid; it < (LN ? NC : 64 * NC); it += LN ? TPR : NTHREADS) {
            const int row = LN ? lrow : it / NC, c8 = LN ? it : it - row * NC;
            const int m = m0 + h * 64 + row, n = n0 + c8 * 8;
            if (m >= p.M || n >= p.N) continue;
            float v[8];
            const f32x4 v0 = *(const f32x4*)(ep + row * EP_LD + c8 * 8);
            const f32x4 v1 = *(const f32x4*)(ep + row * EP_LD + c8 * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = v0[e] * p.alpha; v[4 + e] = v1[e] * p.alpha; }
            if (ln_fold) {
                const float mu = rstat[2 * (h * 64 + row)], rs = rstat[2 * (h * 64 + row) + 1];
                const f32x4 c0 = *(const f32x4*)(lf + c8 * 8), c1 = *(const f32x4*)(lf + c8 * 8 + 4);
                const f32x4 b0 = *(const f32x4*)(lf + BN + c8 * 8), b1 = *(const f32x4*)(lf + BN + c8 * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = (v[e] - mu * c0[e]) * rs + b0[e]; v[4 + e] = (v[4 + e] - mu * c1[e]) * rs + b1[e]; }
            } else if (bias_lds) {
                const f32x4 b0 = *(const f32x4*)(lf + BN + c8 * 8), b1 = *(const f32x4*)(lf + BN + c8 * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
            } else if (p.bias) {
                const f32x4 b0 = *(const f32x4*)(p.bias + n), b1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
            }
            if (p.rowbias) {
                const float* rb = p.rowbias + (size_t)(p.rowbias_shift >= 0 ? m >> p.rowbias_shift : m / p.rowbias_div) * p.ld_rowbias + n;
                const f32x4 b0 = *(const f32x4*)rb, b1 = *(const f32x4*)(rb + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
            }
            if (LORA && p.lora_t) {
                if (lora_lds) lora_apply(p, m, n, v, lf + c8 * 8, BN, lt + row * 16);
                else lora_apply(p, m, n, v, p.lora_b + (size_t)n * p.lora_r, p.N, nullptr);
            }
            if (p.residual) {
                const bf16x8 rv = *(const bf16x8*)(p.residual + (size_t)m * p.ldr + n);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
            }
            if (p.gb_pre != nullptr) {
                const size_t row = (size_t)m * (2 * p.N);
                const bf16x8 hb = *(const bf16x8*)(p.gb_pre + row + n), gb = *(const bf16x8*)(p.gb_pre + row + p.N + n);
                bf16x8 dh, dg;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float gv = (float)gb[e];
                    dh[e] = (bf16_t)(v[e] * gelu_erf_f(gv));
                    dg[e] = (bf16_t)(v[e] * (float)hb[e] * gelu_erf_grad_f(gv));
                }
                *(bf16x8*)(p.gb_out + row + n) = dh;
                *(bf16x8*)(p.gb_out + row + p.N + n) = dg;
                continue;
            }
            if (p.out_fp32) {
                float* o = (float*)outp + (size_t)m * ldo + n - nsh;
                *(f32x4*)o = f32x4{v[0], v[1], v[2], v[3]};
                *(f32x4*)(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
            } else {
                bf16x8 ov;
#pragma unroll
                for (int e = 0; e < 8; ++e) ov[e] = (bf16_t)v[e];
                *(bf16x8*)((bf16_t*)outp + (size_t)m * ldo + n - nsh) = ov;
                if (LN && p.rowstat_out != nullptr) {      // sums of the values as stored
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float f = (float)ov[e]; rs0 += f; rs1 += f * f; }
                }
                if (gs_emit) {
                    float* er = ep + row * EP_LD + c8 * 8;
                    *(f32x4*)er = f32x4{(float)ov[0], (float)ov[1], (float)ov[2], (float)ov[3]};
                    *(f32x4*)(er + 4) = f32x4{(float)ov[4], (float)ov[5], (float)ov[6], (float)ov[7]};
                }
            }
        }
        if (LN && p.rowstat_out != nullptr) {
            // (LDS atomics per chunk cost more than the whole epilogue - up to 20 lanes on one address; a second pass
            // over the staged rows cost a write-back, a barrier and a read)
            static_assert((TPR & (TPR - 1)) == 0, "row sums: a power of two of lanes per row");
#pragma unroll
            for (int o = TPR / 2; o > 0; o >>= 1) { rs0 += __shfl_xor(rs0, o); rs1 += __shfl_xor(rs1, o); }
            const int m = m0 + h * 64 + lrow;
            if (tid % TPR == 0 && m < p.M) *(f32x2*)(p.rowstat_out + ((size_t)m * p.tiles_n + n0 / BN) * 2) = f32x2{rs0, rs1};
        }
        if (gs_emit) {
            // thread (q, rp): columns 4q .. 4q + 3 of rows rp, rp + RP, ...: one 16-byte LDS read per row
            constexpr int NQ = BN / 4, RP = NTHREADS / NQ >= 4 ? 4 : (NTHREADS / NQ >= 2 ? 2 : 1);
            __syncthreads();
            const int q = tid % NQ, rp = tid / NQ;
            if (rp < RP) {
                const int nrow = p.M - m0 - h * 64 < 64 ? p.M - m0 - h * 64 : 64;
#pragma unroll 4
                for (int r = rp; r < nrow; r += RP) {
                    const f32x4 f = *(const f32x4*)(ep + r * EP_LD + 4 * q);
                    gq0 += f;
                    gq1 += f * f;
                }
            }
        }
    }
    if (gs_emit) {
        // columns -> runs of gs_unit columns through LDS, one lane per run
        constexpr int NQ = BN / 4, RP = NTHREADS / NQ >= 4 ? 4 : (NTHREADS / NQ >= 2 ? 2 : 1);
        __syncthreads();
        if (tid < NQ * RP) {            // [RP][BN] sums, then [RP][BN] sums of squares
            *(f32x4*)(ep + (tid / NQ) * BN + 4 * (tid % NQ)) = gq0;
            *(f32x4*)(ep + (RP + tid / NQ) * BN + 4 * (tid % NQ)) = gq1;
        }
        __syncthreads();
        const int u = p.gs_unit;
        if (tid < BN / u && n0 + tid * u < p.N) {
            float s0 = 0.f, s1 = 0.f;
            for (int c = tid * u; c < tid * u + u; ++c)
                for (int r = 0; r < RP; ++r) { s0 += ep[r * BN + c]; s1 += ep[(RP + r) * BN + c]; }
            *(f32x2*)(p.gstat_out + ((size_t)(m0 / BM) * (p.N / u) + (n0 / u) + tid) * 2) = f32x2{s0, s1};
        }
    }
}

template <int BN, int AMODE>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs p) {
    constexpr int BM = 128, BK = 64;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr int A_CH = BM * 8 / 256, B_CH = BN * 8 / 256;
    constexpr int STAGE = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;

    // ---- block -> (tile_m, tile_n, split) with XCD-contiguous chunks of the logical grid
    const int nwg = gridDim.x;
    int lid = xcd_remap(blockIdx.x, nwg);
    const int tile_n = lid % p.tiles_n; lid /= p.tiles_n;
    const int tile_m = lid % p.tiles_m;
    const int split = lid / p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nkt_total = (p.K + BK - 1) / BK;
    const int kt_begin = split * p.kt_per_split;
    int kt_end = kt_begin + p.kt_per_split;
    if (kt_end > nkt_total) kt_end = nkt_total;

    // ---- per-thread loader state
    const int cc = tid & 7;          // 16-byte chunk inside the 64-wide k slice
    const int r0 = tid >> 3;         // first tile row handled (then +32 per step)
    const bf16_t* a_base[A_CH];      // dense: row pointer; conv: pointer to pixel (b,0,0,0)
    int a_iy0[A_CH], a_ix0[A_CH];
    bool a_ok[A_CH];
    const int pad = p.g.pad;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int m = m0 + r0 + 32 * i;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        if constexpr (amode_dense(AMODE)) {
            a_base[i] = p.A + (size_t)mm * p.lda;
            a_iy0[i] = a_ix0[i] = 0;
        } else {
            const int ox = mm % p.g.Wo;
            const int t = mm / p.g.Wo;
            const int oy = t % p.g.Ho;
            const int b = t / p.g.Ho;
            a_base[i] = p.A + (size_t)b * p.g.H * p.g.W * p.g.C;
            a_iy0[i] = oy * p.g.stride - pad;
            a_ix0[i] = ox * p.g.stride - pad;
        }
    }
    const bf16_t* b_base[B_CH];
    bool b_ok[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        const int n = n0 + r0 + 32 * i;
        b_ok[i] = n < p.N;
        b_base[i] = p.Bw + (size_t)(b_ok[i] ? n : 0) * p.ldb;
    }

    bf16x8 a_reg[A_CH], b_reg[B_CH];
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    auto load_tile = [&](int kt) {
        const int k = kt * BK + cc * 8;
        const bool k_ok = k < p.K;
        if constexpr (amode_dense(AMODE)) {
#pragma unroll
            for (int i = 0; i < A_CH; ++i)
                a_reg[i] = (a_ok[i] && k_ok) ? *(const bf16x8*)(a_base[i] + k) : zero8;
        } else {
            int tap, c;
            if constexpr (AMODE == AMODE_CONV_FAST) {
                const int kk = kt * BK;
                tap = kk / p.g.C;              // wave-uniform
                c = kk - tap * p.g.C + cc * 8;
            } else {
                tap = k / p.g.C;
                c = k - tap * p.g.C;
            }
            int ky = 0, kx = 0;
            if (p.g.ksize == 3) { ky = tap / 3; kx = tap - ky * 3; }
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                int sy = a_iy0[i] + ky, sx = a_ix0[i] + kx;
                bool ok = a_ok[i] && k_ok && sy >= 0 && sx >= 0;
                if (p.g.dil) {
                    ok = ok && !((sy | sx) & 1);
                    sy >>= 1; sx >>= 1;
                } else if (p.g.ups) {
                    sy >>= 1; sx >>= 1;
                }
                ok = ok && sy < p.g.H && sx < p.g.W;
                a_reg[i] = ok ? *(const bf16x8*)(a_base[i] + ((size_t)(sy * p.g.W + sx) * p.g.C + c)) : zero8;
            }
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i)
            b_reg[i] = (b_ok[i] && k_ok) ? *(const bf16x8*)(b_base[i] + k) : zero8;
    };
    auto store_tile = [&](int stage) {
        char* As = smem + stage * STAGE;
        char* Bs = As + BM * 128;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int r = r0 + 32 * i;
            *(bf16x8*)(As + r * 128 + ((cc ^ (r & 7)) << 4)) = a_reg[i];
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            const int r = r0 + 32 * i;
            *(bf16x8*)(Bs + r * 128 + ((cc ^ (r & 7)) << 4)) = b_reg[i];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    if (kt_begin < kt_end) {
        load_tile(kt_begin);
        store_tile(0);
    }
    __syncthreads();
    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = kt + 1 < kt_end;
        if (more) load_tile(kt + 1);
        const char* As = smem + cur * STAGE;
        const char* Bs = As + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TM], bfr[TN];
            const int ch = ks * 4 + fq;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm0 + i * 16 + fr;
                af[i] = *(const bf16x8*)(As + r * 128 + ((ch ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn0 + j * 16 + fr;
                bfr[j] = *(const bf16x8*)(Bs + r * 128 + ((ch ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    float ln_c = 0.f, ln_b = 0.f;
    if constexpr (AMODE != AMODE_DENSE_LORA) ln_fold_column<BN>(p, n0, tid, ln_c, ln_b);
    gemm_epilogue<BN, TM, TN, 256, 128, AMODE == AMODE_DENSE_LORA, amode_stats(AMODE)>(p, acc, smem, m0, n0, split, wm0, wn0, wave, tid, fr, fq,
                                                                                           nullptr, ln_c, ln_b);
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA variant: both operand tiles are written into LDS by `buffer_load ... lds` (no VGPR staging,
// no ds_write).  One wave instruction fills a 1 KiB piece = 8 tile rows x 128 B in lane order, so the
// XOR swizzle is applied on the SOURCE side: lane l of a piece fetches k-chunk (l&7)^(l>>3) of row l>>3.
// Rows outside the problem and padding taps of the implicit im2col get an out-of-range buffer offset:
// the hardware bounds check returns zeros, which is exactly the zero padding the convolution needs.
// Pipeline: two LDS stages, ONE barrier per k-tile; the loads of tile t+1 are issued right after the
// barrier and stay in flight under the MFMAs of tile t.
// Requirements (checked by the launcher): K % 64 == 0, conv C % 64 == 0, operand sizes < 2 GiB.
// ---------------------------------------------------------------------------------------------
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
constexpr int OOB_OFF = (int)0x80000000;

// Transposing LDS read as inline asm: an LDS read the compiler knows about, other than a plain load, is made to wait for
// EVERY outstanding `buffer_load ... lds` (vmcnt(0)) - the prefetch of the next tiles would never overlap the products.
// lds_tr_wait() before the first use of the results.
template <int OFF>
__device__ __forceinline__ bf16x4 lds_tr_asm(unsigned addr) {
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void lds_tr_wait(bf16x4 (&lo)[N], bf16x4 (&hi)[N]) {
    static_assert(N == 5 || N == 1, "one register list per fragment count");
    if constexpr (N == 5)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(lo[4]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]), "+v"(hi[4])
                     :: "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]) :: "memory");
}

// NW = 4: 2x2 waves of 64 x BN/2, two workgroups per CU.  NW = 8: 4x2 waves of 32 x BN/2 on the same tile, for
// grids of at most one workgroup per CU: a lone 4-wave workgroup has one wave per SIMD, whose DMA issue, LDS
// reads and MFMAs serialise; eight waves give every SIMD a second wave to overlap them with.
// BM: rows per tile.  (256 rows with 8 waves as ONE workgroup per CU measured 15 % slower than two 128-row
// workgroups: its waves run barrier-locked, so DMA issue and MFMA phases never overlap.  Large grids use
// gemm_wide_kernel instead.)
// NST: LDS stages.  3 (grids of at most one workgroup per CU, which have the LDS to spare): tile kt + 2 is requested
// while tile kt is multiplied, so a request has two iterations to land instead of one - with one workgroup per CU the
// 2-stage loop waits on every tile's round trip (~1 600 cycles per k-tile against 640 of MFMA).
// BT: B is k-major (GemmArgs::bt).  A k-tile of B is then 64 rows of BN contiguous values, kept in LDS as [k][BN] (row
// = 320 B for BN = 160: consecutive rows 64 B apart modulo the 256-B bank row) and read with ds_read_b64_tr_b16, two
// per fragment: the 16 lanes of a group fetch 4 rows x 16 columns and each receives one column.  The two groups that
// share an LDS pass read rows 8 apart - the same banks - so the rows of every odd 8-row block are rotated by 32 B
// (on the source side of the DMA, like the XOR swizzle of the row-major tiles).
template <int BN, int AMODE, int NW = 4, int BM = 128, int NST = 2, bool BT = false>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void gemm_dma_kernel(const GemmArgs p, unsigned a_bytes, unsigned b_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)      // the body uses device-only buffer-resource builtins
    constexpr int BK = 64;
    constexpr int WM = BM / (NW / 2), WN = BN / 2;
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr int A_P = BM / 8 / NW;                     // 1 KiB pieces (8 rows x 128 B) per wave
    constexpr int B_PIECES = BN / 8, B_P = (B_PIECES + NW - 1) / NW;
    constexpr int STAGE = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;

    const int nwg = gridDim.x;
    int lid = xcd_remap(blockIdx.x, nwg);
    const int tile_n = lid % p.tiles_n; lid /= p.tiles_n;
    const int tile_m = lid % p.tiles_m;
    const int split = lid / p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    float ln_c = 0.f, ln_b = 0.f;
    if constexpr (AMODE != AMODE_DENSE_LORA) ln_fold_column<BN>(p, n0, tid, ln_c, ln_b);
    if constexpr (AMODE == AMODE_DENSE_LN) ln_row_stats<BM, NW * 64>(p, m0, (float*)(smem + NST * STAGE), tid);
    const int nkt_total = p.K / BK;
    const int kt_begin = split * p.kt_per_split;
    int kt_end = kt_begin + p.kt_per_split;
    if (kt_end > nkt_total) kt_end = nkt_total;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.Bw, 0, b_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(AMODE == AMODE_DENSE_CAT ? p.A2 : p.A), 0, AMODE == AMODE_DENSE_CAT ? p.a2_bytes : 0u, 0x00020000);

    const int sr = lane >> 3;               // row inside the 8-row piece
    const int kc = (lane & 7) ^ sr;         // k-chunk fetched by this lane (source-side swizzle)
    int a_voff[A_P];                         // dense: byte offset of (row, kc); conv: pixel index of (b,0,0) or -1
    int a_voff2[AMODE == AMODE_DENSE_CAT ? A_P : 1];     // the same in the second source
    int a_iy0[A_P], a_ix0[A_P];
    const int pad = p.g.pad;
#pragma unroll
    for (int j = 0; j < A_P; ++j) {
        const int m = m0 + (wave + NW * j) * 8 + sr;
        const bool ok = m < p.M;
        if constexpr (amode_dense(AMODE)) {
            a_voff[j] = ok ? (m * p.lda + kc * 8) * 2 : OOB_OFF;
            if constexpr (AMODE == AMODE_DENSE_CAT) a_voff2[j] = ok ? (m * p.lda2 + kc * 8) * 2 : OOB_OFF;
            a_iy0[j] = a_ix0[j] = 0;
        } else {
            const int mm = ok ? m : 0;
            const int ox = mm % p.g.Wo;
            const int t = mm / p.g.Wo;
            const int oy = t % p.g.Ho;
            const int b = t / p.g.Ho;
            a_voff[j] = ok ? b * p.g.H * p.g.W : -1;
            a_iy0[j] = oy * p.g.stride - pad;
            a_ix0[j] = ox * p.g.stride - pad;
        }
    }
    int b_voff[B_P];
    if constexpr (BT) {
        static_assert(!BT || BN % 16 == 0, "rows of whole 32-byte blocks");
        constexpr int RCH = BN / 8;             // 16-byte chunks per k row
#pragma unroll
        for (int j = 0; j < B_P; ++j) {
            const int slot = ((wave + NW * j) * 64 + lane) % (64 * RCH);      // (pieces past the tile are never issued)
            const int k = slot / RCH;
            int c = slot - k * RCH - 2 * ((k >> 3) & 1);
            if (c < 0) c += RCH;
            const int n = n0 + c * 8;
            b_voff[j] = n < p.N ? (k * p.bt_ld + n) * 2 : OOB_OFF;
        }
    } else
#pragma unroll
    for (int j = 0; j < B_P; ++j) {
        int n = n0 + (wave + NW * j) * 8 + sr;
        if (p.geglu) {                       // first half of the tile: hidden rows, second half: the matching gate rows
            const int nl = (wave + NW * j) * 8 + sr, F = p.N >> 1, f0 = tile_n * (BN / 2);
            n = nl < BN / 2 ? f0 + nl : F + f0 + nl - BN / 2;
            if (f0 + (nl < BN / 2 ? nl : nl - BN / 2) >= F) n = p.N;
        }
        b_voff[j] = n < p.N ? (n * p.ldb + kc * 8) * 2 : OOB_OFF;
    }
    const int nbp = (B_PIECES - wave + NW - 1) / NW;     // B pieces of this wave (uneven when NW does not divide them)

    // Plain 3x3 / 1x1 convolution (no up-sampling, no dilation): a piece's source address is affine in the tap, so a
    // request costs one add and one select per piece (`a_mask`: which of the 9 taps fall inside the image).  Tiles are
    // requested in increasing order: the tap and the channel offset advance by increments, no division per tile.
    const bool conv_affine = !amode_dense(AMODE) && !p.g.ups && !p.g.dil;
    int a_base[A_P], a_mask[A_P];
    if constexpr (!amode_dense(AMODE)) {
#pragma unroll
        for (int j = 0; j < A_P; ++j) {
            a_base[j] = ((a_voff[j] + a_iy0[j] * p.g.W + a_ix0[j]) * p.g.C + kc * 8) * 2;
            int mask = 0;
            if (conv_affine) {
                const int nt = p.g.ksize == 3 ? 3 : 1;
                for (int ky = 0; ky < nt; ++ky)
                    for (int kx = 0; kx < nt; ++kx) {
                        const int sy = a_iy0[j] + ky, sx = a_ix0[j] + kx;
                        if (a_voff[j] >= 0 && sy >= 0 && sx >= 0 && sy < p.g.H && sx < p.g.W) mask |= 1 << (ky * 3 + kx);
                    }
            }
            a_mask[j] = mask;
        }
    }

    auto issue = [&](int kt, int stage) {
        char* As = smem + stage * STAGE;
        char* Bs = As + BM * 128;
        int soffb = kt * (BK * 2);
        int cur_c = 0, cur_ky = 0, cur_kx = 0;
        if constexpr (!amode_dense(AMODE)) {
            int slice;
            conv_ktile(kt, p.g.ksize, slice, cur_ky, cur_kx);
            cur_c = slice * BK;
            soffb = ((cur_ky * 3 + cur_kx) * p.g.C + cur_c) * 2;                       // weight row: [tap][channel]
        }
        if constexpr (BT) {
            if constexpr (amode_dense(AMODE)) soffb = kt * BK * p.bt_ld * 2;
            else soffb = (cur_c * p.bt_ld + (p.g.ksize == 3 ? 8 - (cur_ky * 3 + cur_kx) : 0) * p.N) * 2;
        }
        if constexpr (AMODE == AMODE_DENSE_CAT) {
            if (kt * BK < p.K1) {            // (uniform) the k-tile lies in the first source ...
                const int soff = kt * (BK * 2);
#pragma unroll
                for (int j = 0; j < A_P; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(As + (wave + NW * j) * 1024), 16, a_voff[j], soff, 0, 0);
            } else {                         // ... or in the second
                const int soff = (kt * BK - p.K1) * 2;
#pragma unroll
                for (int j = 0; j < A_P; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, LDS_PTR(As + (wave + NW * j) * 1024), 16, a_voff2[j], soff, 0, 0);
            }
        } else if constexpr (amode_dense(AMODE)) {
            const int soff = kt * (BK * 2);
#pragma unroll
            for (int j = 0; j < A_P; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(As + (wave + NW * j) * 1024), 16, a_voff[j], soff, 0, 0);
        } else {
            const int soff = cur_c * 2;
            const int ky = cur_ky, kx = cur_kx;
            const int tap_off = (ky * p.g.W + kx) * p.g.C * 2, tap_bit = ky * 3 + kx;
#pragma unroll
            for (int j = 0; j < A_P; ++j) {
                int voff;
                if (conv_affine) {
                    voff = (a_mask[j] >> tap_bit) & 1 ? a_base[j] + tap_off : OOB_OFF;
                } else {
                    int sy = a_iy0[j] + ky, sx = a_ix0[j] + kx;
                    bool ok = a_voff[j] >= 0 && sy >= 0 && sx >= 0;
                    if (p.g.dil) {
                        ok = ok && !((sy | sx) & 1);
                        sy >>= 1; sx >>= 1;
                    } else if (p.g.ups) {
                        sy >>= 1; sx >>= 1;
                    }
                    ok = ok && sy < p.g.H && sx < p.g.W;
                    voff = ok ? ((a_voff[j] + sy * p.g.W + sx) * p.g.C + kc * 8) * 2 : OOB_OFF;
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(As + (wave + NW * j) * 1024), 16, voff, soff, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < B_P; ++j)
            if (B_PIECES % NW == 0 || j < nbp)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(Bs + (wave + NW * j) * 1024), 16, b_voff[j], soffb, 0, 0);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    // BT: byte offset of this lane's transposing reads inside a k-step's 32 rows, per 16-column block of the wave
    int bt_off[BT ? TN : 1];
    if constexpr (BT) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            int col = (wn0 + j * 16 + 4 * (fr & 3)) * 2 + 32 * (fq & 1);
            if (col >= BN * 2) col -= BN * 2;
            bt_off[j] = (fq * 8 + (fr >> 2)) * (BN * 2) + col;
        }
    }
    // NST stages: tiles kt + 1 .. kt + NST - 2 stay in flight while tile kt is multiplied
#pragma unroll
    for (int d = 0; d < NST - 1; ++d)
        if (kt_begin + d < kt_end) issue(kt_begin + d, d);
    int cur = 0;
    // requests complete in order: wait until only the newer tiles' can be outstanding (the waves with the fewest
    // pieces per tile have PER_TILE of them)
    constexpr int PER_TILE = A_P + (B_PIECES % NW == 0 ? B_P : B_P - 1);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        if (NST >= 3 && kt + NST - 2 < kt_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * PER_TILE) : "memory");
        else if (NST == 4 && kt + 1 < kt_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's pieces of tile kt have landed
        __builtin_amdgcn_s_barrier();                         // everyone's have; everyone left tile kt-1
        __builtin_amdgcn_sched_barrier(0);
        // the next request goes into the stage tile kt-1 was read from
        if (kt + NST - 1 < kt_end) issue(kt + NST - 1, cur == 0 ? NST - 1 : cur - 1);
        const char* As = smem + cur * STAGE;
        const char* Bs = As + BM * 128;
        if constexpr (BT) {
            // both k-steps' fragments are requested up front, the second step's while the first step multiplies
            static_assert(!BT || TN == 5, "k-major B: 160-column tiles");
            bf16x8 af[2][TM];
            bf16x4 lo[2][BT ? TN : 1], hi[2][BT ? TN : 1];
            const unsigned bs = (unsigned)(size_t)(__attribute__((address_space(3))) char*)Bs;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int ch = ks * 4 + fq;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int r = wm0 + i * 16 + fr;
                    af[ks][i] = *(const bf16x8*)(As + r * 128 + ((ch ^ (r & 7)) << 4));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const unsigned b0 = bs + bt_off[BT ? j : 0];
                    if (ks == 0) { lo[0][BT ? j : 0] = lds_tr_asm<0>(b0); hi[0][BT ? j : 0] = lds_tr_asm<4 * BN * 2>(b0); }
                    else { lo[1][BT ? j : 0] = lds_tr_asm<32 * BN * 2>(b0); hi[1][BT ? j : 0] = lds_tr_asm<36 * BN * 2>(b0); }
                }
                if (ks == 0) lds_tr_wait(lo[0], hi[0]);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks == 1) lds_tr_wait(lo[1], hi[1]);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const bf16x4 l = lo[ks][BT ? j : 0], h = hi[ks][BT ? j : 0];
                    const bf16x8 bf = bf16x8{l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]};
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][i], bf, acc[i][j], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TM], bfr[TN];
            const int ch = ks * 4 + fq;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm0 + i * 16 + fr;
                af[i] = *(const bf16x8*)(As + r * 128 + ((ch ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn0 + j * 16 + fr;
                bfr[j] = *(const bf16x8*)(Bs + r * 128 + ((ch ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        }
        cur = cur == NST - 1 ? 0 : cur + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    gemm_epilogue<BN, TM, TN, NW * 64, BM, AMODE == AMODE_DENSE_LORA, amode_stats(AMODE)>(
        p, acc, smem, m0, n0, split, wm0, wn0, wave, tid, fr, fq, (const float*)(smem + NST * STAGE), ln_c, ln_b);
#endif
}

// ---------------------------------------------------------------------------------------------
// Wide-tile variant for large grids: 256 x 160 output tile, 32-deep k-steps, 4 waves of 128 x 80, two
// workgroups per CU.  Every operand byte enters the CU through the vector-memory path (64 B/clk), and the 128 x 160
// tile keeps that path ~90 % busy at 40 % MFMA utilisation (72 KB per 1280 MFMA cycles); this tile moves 28 % fewer
// bytes per FLOP through it and through LDS (13 instead of 18 fragment reads per 40 MFMAs).
// A 1 KiB DMA piece is 16 rows x 32 k = exactly one 16x16x32 MFMA operand fragment: lane l fetches row l/4,
// 16-byte k-chunk (l%4) ^ F[l/16], and the fragment read of lane (row fr, chunk fq) is slot fr*4 + (fq ^ F[fr/4]);
// F = {0,2,3,1} makes the four 16-lane groups of ds_read_b128 hit 16 distinct 16-byte bank groups each.
// ---------------------------------------------------------------------------------------------
template <int AMODE>
__global__ __launch_bounds__(256, 2) void gemm_wide_kernel(const GemmArgs p, unsigned a_bytes, unsigned b_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 256, BN = 160, BK = 32, NW = 4;
    constexpr int TM = 8, TN = 5;
    constexpr int A_PIECES = BM / 16, B_PIECES = BN / 16, PIECES = A_PIECES + B_PIECES;     // 16 + 10
    constexpr int NP = (PIECES + NW - 1) / NW;                                               // 7 per wave (waves 2,3: 6)
    constexpr int STAGE = PIECES * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 128, wn0 = (wave & 1) * 80;

    int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % p.tiles_n;
    const int tile_m = lid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    constexpr int LN_STAT_OFF = (64 * 164 + 16 * 160 + 64 * 16) * 4;      // behind the stages and the epilogue staging
    constexpr int LN_RAW_OFF = LN_STAT_OFF + 2 * BM * 4;                  // up to 16 KiB of partial sums ([256][8][2])
    float ln_c = 0.f, ln_b = 0.f;
    if constexpr (AMODE != AMODE_DENSE_LORA) ln_fold_column<BN>(p, n0, tid, ln_c, ln_b);
    if constexpr (AMODE == AMODE_DENSE_LN) ln_stats_request<BM, NW>(p, m0, smem + LN_RAW_OFF, wave, lane);
    const int nkt = p.K / BK;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.Bw, 0, b_bytes, 0x00020000);

    const int prow = lane >> 2;                                            // row inside the 16-row piece
    const int kc = (lane & 3) ^ ((0x78 >> (2 * (lane >> 4))) & 3);         // k-chunk fetched by this lane
    // piece q of this wave is tile piece wave + 4*q: an A piece while < 16, else B piece (index - 16)
    int voff[NP], iy0[NP], ix0[NP];
    const int pad = p.g.pad;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int pc = wave + NW * q;
        iy0[q] = ix0[q] = 0;
        if (pc < A_PIECES) {
            const int m = m0 + pc * 16 + prow;
            const bool ok = m < p.M;
            if constexpr (amode_dense(AMODE)) {
                voff[q] = ok ? (m * p.lda + kc * 8) * 2 : OOB_OFF;
            } else {
                const int mm = ok ? m : 0;
                const int ox = mm % p.g.Wo;
                const int t = mm / p.g.Wo;
                const int oy = t % p.g.Ho;
                const int b = t / p.g.Ho;
                voff[q] = ok ? b * p.g.H * p.g.W : -1;
                iy0[q] = oy * p.g.stride - pad;
                ix0[q] = ox * p.g.stride - pad;
            }
        } else if (pc < PIECES) {
            const int nl = (pc - A_PIECES) * 16 + prow;
            int n = n0 + nl;
            if (p.geglu) {                   // first half of the tile: hidden rows, second half: the matching gate rows
                const int F = p.N >> 1, f0 = tile_n * (BN / 2);
                n = nl < BN / 2 ? f0 + nl : F + f0 + nl - BN / 2;
                if (f0 + (nl < BN / 2 ? nl : nl - BN / 2) >= F) n = p.N;
            }
            voff[q] = n < p.N ? (n * p.ldb + kc * 8) * 2 : OOB_OFF;
        } else {
            voff[q] = OOB_OFF;
        }
    }

    // plain convolutions: affine per-tap addresses and an inside-the-image mask per piece, taps advanced by increments
    // (tiles are requested in increasing order) - see gemm_dma_kernel
    const bool conv_affine = !amode_dense(AMODE) && !p.g.ups && !p.g.dil;
    int a_base[NP], a_mask[NP];
    if constexpr (!amode_dense(AMODE)) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            a_base[q] = ((voff[q] + iy0[q] * p.g.W + ix0[q]) * p.g.C + kc * 8) * 2;
            int mask = 0;
            if (conv_affine && wave + NW * q < A_PIECES) {
                const int nt = p.g.ksize == 3 ? 3 : 1;
                for (int ky = 0; ky < nt; ++ky)
                    for (int kx = 0; kx < nt; ++kx) {
                        const int sy = iy0[q] + ky, sx = ix0[q] + kx;
                        if (voff[q] >= 0 && sy >= 0 && sx >= 0 && sy < p.g.H && sx < p.g.W) mask |= 1 << (ky * 3 + kx);
                    }
            }
            a_mask[q] = mask;
        }
    }

    // convolutions: 32-deep slab kt = (64-channel slice, tap, 32-channel half) - the order of the 64-deep kernels, k-step by k-step
    auto issue = [&](int kt, int stage) {
        char* base = smem + stage * STAGE;
        int soffb = kt * (BK * 2);
        int soffa = soffb, ky = 0, kx = 0;
        if constexpr (!amode_dense(AMODE)) {
            int slice;
            conv_ktile(kt >> 1, p.g.ksize, slice, ky, kx);
            const int c = slice * (2 * BK) + (kt & 1) * BK;
            soffa = c * 2;
            soffb = ((ky * 3 + kx) * p.g.C + c) * 2;
        }
        const int tap_off = (ky * p.g.W + kx) * p.g.C * 2, tap_bit = ky * 3 + kx;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int pc = wave + NW * q;
            if (pc >= PIECES) continue;
            if (pc < A_PIECES) {
                int v = voff[q];
                if constexpr (!amode_dense(AMODE)) {
                    if (conv_affine) {
                        v = (a_mask[q] >> tap_bit) & 1 ? a_base[q] + tap_off : OOB_OFF;
                    } else {
                        int sy = iy0[q] + ky, sx = ix0[q] + kx;
                        bool ok = voff[q] >= 0 && sy >= 0 && sx >= 0;
                        if (p.g.dil) {
                            ok = ok && !((sy | sx) & 1);
                            sy >>= 1; sx >>= 1;
                        } else if (p.g.ups) {
                            sy >>= 1; sx >>= 1;
                        }
                        ok = ok && sy < p.g.H && sx < p.g.W;
                        v = ok ? ((voff[q] + sy * p.g.W + sx) * p.g.C + kc * 8) * 2 : OOB_OFF;
                    }
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(base + pc * 1024), 16, v, soffa, 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(base + pc * 1024), 16, voff[q], soffb, 0, 0);
            }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    const int rd = (fr * 4 + (fq ^ ((0x78 >> (2 * (fr >> 2))) & 3))) * 16;       // this lane's slot inside a fragment piece
    issue(0, 0);
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nkt) issue(kt + 1, cur ^ 1);
        const char* As = smem + cur * STAGE + (wm0 >> 4) * 1024 + rd;
        const char* Bs = smem + cur * STAGE + (A_PIECES + wn0 / 16) * 1024 + rd;
        bf16x8 af[TM], bfr[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = *(const bf16x8*)(Bs + j * 1024);
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *(const bf16x8*)(As + i * 1024);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        cur ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (AMODE == AMODE_DENSE_LN) ln_stats_finish<BM, 256>(p, smem + LN_RAW_OFF, (float*)(smem + LN_STAT_OFF), tid);
    gemm_epilogue<BN, TM, TN, 256, BM, AMODE == AMODE_DENSE_LORA, amode_stats(AMODE)>(
        p, acc, smem, m0, n0, 0, wm0, wn0, wave, tid, fr, fq, (const float*)(smem + LN_STAT_OFF), ln_c, ln_b);
#endif
}

// ---------------------------------------------------------------------------------------------
// Ping-pong variant for grids that fill the chip with 256 x 320 tiles (the 512^2 sampling step, the widest layers of
// the train step): 8 waves as 2 (M) x 4 (N), each 128 x 80, ONE workgroup per CU.  The k-loop runs on 32-deep slabs in
// a ring of four LDS buffers (36 fragment pieces = 36 KiB each, the piece layout of gemm_wide_kernel).  A slab is
// multiplied in two phases (the wave's upper, then lower 64 rows: 20 MFMAs each); a phase is
//     fragment reads + DMA requests -> s_barrier -> lgkmcnt(0) -> 20 MFMAs -> s_barrier.
// The four waves of the lower tile half run ONE barrier behind the upper four, so while one wave of a SIMD multiplies,
// its partner reads fragments and issues requests: neither the LDS reads nor the request bursts stop the matrix pipe.
// Requests run three slabs ahead and are retired by a counted vmcnt (never 0 inside the loop):
//     A pieces of slab n+3 are requested in the second phase of slab n, its B pieces in the first phase of slab n+1
//     (two barriers after any wave last read that buffer, slab n-1's), and the second phase of slab n waits until
//     slab n+1 has landed - one barrier before any wave reads it.
// Waves 0-3 carry 5 pieces per slab (2 A + 3 B), waves 4-7 carry 4 (2 A + 2 B): the counts in the waits differ per half.
// ---------------------------------------------------------------------------------------------
template <int AMODE>
__global__ __launch_bounds__(512, 1) void gemm_pp_kernel(const GemmArgs p, unsigned a_bytes, unsigned b_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 256, BN = 320, BK = 32, NW = 8;
    constexpr int TM = 8, TN = 5;
    constexpr int A_PIECES = BM / 16, B_PIECES = BN / 16, PIECES = A_PIECES + B_PIECES;     // 16 + 20
    constexpr int STAGE = PIECES * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int wm0 = wr * 128, wn0 = wc * 80;
    const int grp = wr;       // waves w and w + 4 share a SIMD: grouping by wave parity measured 11-19 % slower

    int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % p.tiles_n;
    const int tile_m = lid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    constexpr int LN_STAT_OFF = 4 * 36 * 1024;                            // behind the four slabs
    constexpr int LN_RAW_OFF = LN_STAT_OFF;                               // up to 16 KiB of partial sums ([256][8][2]), overlaid
    float ln_c = 0.f, ln_b = 0.f;
    if constexpr (AMODE != AMODE_DENSE_LORA) ln_fold_column<BN>(p, n0, tid, ln_c, ln_b);
    if constexpr (AMODE == AMODE_DENSE_LN) ln_stats_request<BM, NW>(p, m0, smem + LN_RAW_OFF, wave, lane);
    const int nslab = p.K / BK;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.Bw, 0, b_bytes, 0x00020000);

    const int prow = lane >> 2;                                            // row inside the 16-row piece
    const int kc = (lane & 3) ^ ((0x78 >> (2 * (lane >> 4))) & 3);         // k-chunk fetched by this lane
    // A pieces of this wave: wave and wave + 8
    int a_voff[2], a_iy0[2], a_ix0[2];
    const int pad = p.g.pad;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int m = m0 + (wave + NW * q) * 16 + prow;
        const bool ok = m < p.M;
        a_iy0[q] = a_ix0[q] = 0;
        if constexpr (amode_dense(AMODE)) {
            a_voff[q] = ok ? (m * p.lda + kc * 8) * 2 : OOB_OFF;
        } else {
            const int mm = ok ? m : 0;
            const int ox = mm % p.g.Wo;
            const int t = mm / p.g.Wo;
            const int oy = t % p.g.Ho;
            const int b = t / p.g.Ho;
            a_voff[q] = ok ? b * p.g.H * p.g.W : -1;
            a_iy0[q] = oy * p.g.stride - pad;
            a_ix0[q] = ox * p.g.stride - pad;
        }
    }
    // B pieces of this wave: wave, wave + 8 and (waves 0-3) wave + 16
    int b_voff[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int pc = wave + NW * q;
        const int nl = pc * 16 + prow;
        int n = n0 + nl;
        if (p.geglu) {                       // first half of the tile: hidden rows, second half: the matching gate rows
            const int F = p.N >> 1, f0 = tile_n * (BN / 2);
            n = nl < BN / 2 ? f0 + nl : F + f0 + nl - BN / 2;
            if (f0 + (nl < BN / 2 ? nl : nl - BN / 2) >= F) n = p.N;
        }
        b_voff[q] = (pc < B_PIECES && n < p.N) ? (n * p.ldb + kc * 8) * 2 : OOB_OFF;
    }

    // Plain 3x3 / 1x1 convolution (no up-sampling, no dilation): a piece's source address is affine in the tap, so the
    // per-slab work is one add and one select per piece - `a_mask` holds which of the 9 taps fall inside the image.
    // Slabs are requested in increasing order: the tap and the channel offset advance by increments, no division.
    const bool conv_affine = !amode_dense(AMODE) && !p.g.ups && !p.g.dil;
    int a_base[2] = {0, 0}, a_mask[2] = {0, 0};
    if constexpr (!amode_dense(AMODE)) {
        if (conv_affine) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                a_base[q] = ((a_voff[q] + a_iy0[q] * p.g.W + a_ix0[q]) * p.g.C + kc * 8) * 2;
                int mask = 0;
                const int nt = p.g.ksize == 3 ? 3 : 1;
                for (int ky = 0; ky < nt; ++ky)
                    for (int kx = 0; kx < nt; ++kx) {
                        const int sy = a_iy0[q] + ky, sx = a_ix0[q] + kx;
                        if (a_voff[q] >= 0 && sy >= 0 && sx >= 0 && sy < p.g.H && sx < p.g.W) mask |= 1 << (ky * 3 + kx);
                    }
                a_mask[q] = mask;
            }
        }
    }
    // convolutions: 32-deep slab = (64-channel slice, tap, 32-channel half), the order of the 64-deep kernels
    auto issue_a = [&](int slab) {
        char* base = smem + (slab & 3) * STAGE;
        int soffa = slab * (BK * 2), ky = 0, kx = 0;
        if constexpr (!amode_dense(AMODE)) {
            int slice;
            conv_ktile(slab >> 1, p.g.ksize, slice, ky, kx);
            soffa = (slice * (2 * BK) + (slab & 1) * BK) * 2;
        }
        const int tap_off = (ky * p.g.W + kx) * p.g.C * 2, tap_bit = ky * 3 + kx;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int v = a_voff[q];
            if constexpr (!amode_dense(AMODE)) {
                if (conv_affine) {
                    v = (a_mask[q] >> tap_bit) & 1 ? a_base[q] + tap_off : OOB_OFF;
                } else {
                    int sy = a_iy0[q] + ky, sx = a_ix0[q] + kx;
                    bool ok = a_voff[q] >= 0 && sy >= 0 && sx >= 0;
                    if (p.g.dil) {
                        ok = ok && !((sy | sx) & 1);
                        sy >>= 1; sx >>= 1;
                    } else if (p.g.ups) {
                        sy >>= 1; sx >>= 1;
                    }
                    ok = ok && sy < p.g.H && sx < p.g.W;
                    v = ok ? ((a_voff[q] + sy * p.g.W + sx) * p.g.C + kc * 8) * 2 : OOB_OFF;
                }
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(base + (wave + NW * q) * 1024), 16, v, soffa, 0, 0);
        }
    };
    auto issue_b = [&](int slab) {
        char* base = smem + (slab & 3) * STAGE + A_PIECES * 1024;
        int soffb = slab * (BK * 2);
        if constexpr (!amode_dense(AMODE)) {
            int slice, ky, kx;
            conv_ktile(slab >> 1, p.g.ksize, slice, ky, kx);
            soffb = ((ky * 3 + kx) * p.g.C + slice * (2 * BK) + (slab & 1) * BK) * 2;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(base + wave * 1024), 16, b_voff[0], soffb, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(base + (wave + NW) * 1024), 16, b_voff[1], soffb, 0, 0);
        if (wave < 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(base + (wave + 2 * NW) * 1024), 16, b_voff[2], soffb, 0, 0);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    const int rd = (fr * 4 + (fq ^ ((0x78 >> (2 * (fr >> 2))) & 3))) * 16;       // this lane's slot inside a fragment piece

    // prologue: slabs 0, 1, 2 requested; slab 0 landed before the first read
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (d < nslab) { issue_a(d); issue_b(d); }
    if (nslab >= 3) {
        if (wave < 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (nslab == 2) {
        if (wave < 4) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) {                           // the second group runs one barrier behind
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    for (int n = 0; n < nslab; ++n) {
        const char* st = smem + (n & 3) * STAGE;
        const char* As = st + (wm0 >> 4) * 1024 + rd;
        const char* Bs = st + (A_PIECES + wn0 / 16) * 1024 + rd;
        bf16x8 af[4], bfr[TN];
        // ---- first phase: rows 0-63 of the wave's tile
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = *(const bf16x8*)(Bs + j * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(As + i * 1024);
        if (n >= 1 && n + 2 < nslab) issue_b(n + 2);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- second phase: rows 64-127
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(As + (4 + i) * 1024);
        if (n + 3 < nslab) issue_a(n + 3);
        // slab n + 1 landed: what may still be in flight is slab n + 2 and the A pieces of slab n + 3
        if (n + 3 < nslab) {
            if (wave < 4) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else if (n + 2 < nslab) {
            if (wave < 4) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[4 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 0) {                           // every wave passes the same number of barriers
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (AMODE == AMODE_DENSE_LN) ln_stats_finish<BM, 512, true>(p, smem + LN_RAW_OFF, (float*)(smem + LN_STAT_OFF), tid);
    gemm_epilogue<BN, TM, TN, 512, BM, false, amode_stats(AMODE)>(
        p, acc, smem, m0, n0, 0, wm0, wn0, wave, tid, fr, fq, (const float*)(smem + LN_STAT_OFF), ln_c, ln_b);
#endif
}

// ---------------------------------------------------------------------------------------------
// Skinny GEMM: out[M][W] = alpha * A[M][K] . Bw[W][K]^T with W = 8 or 16 (the rank-padded products of the LoRA
// adapters: x A^T and dy B).  The tile kernels would spend a 128 x 64 tile, split-K slabs and a reduce launch on it;
// here a workgroup owns 16 rows, its four waves take every fourth 32-deep k-step with both operands loaded straight
// from global memory in MFMA layout (Bw is a few KB and stays in L1/L2), and the MFMA is issued with the operands
// swapped so that a lane ends up with four adjacent outputs of one row - an 8-byte store, no LDS transpose.
// ---------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ Bw,
                                                          bf16_t* __restrict__ out, int M, int K, float alpha) {
    __shared__ f32x4 red[3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int row = blockIdx.x * 16 + fr;
    const bool row_ok = row < M, col_ok = fr < W;
    const bf16_t* ap = A + (size_t)(row_ok ? row : 0) * lda + fq * 8;
    const bf16_t* bp = Bw + (size_t)(col_ok ? fr : 0) * K + fq * 8;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int nks = K >> 5;
    int ks = wave;
    for (; ks + 12 < nks; ks += 16) {               // four k-steps in flight per wave
        bf16x8 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = row_ok ? *(const bf16x8*)(ap + (ks + 4 * u) * 32) : zero8;
            b[u] = col_ok ? *(const bf16x8*)(bp + (ks + 4 * u) * 32) : zero8;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[u], a[u], acc, 0, 0, 0);
    }
    for (; ks < nks; ks += 4) {
        const bf16x8 a = row_ok ? *(const bf16x8*)(ap + ks * 32) : zero8;
        const bf16x8 b = col_ok ? *(const bf16x8*)(bp + ks * 32) : zero8;
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc, 0, 0, 0);
    }
    // acc[r] = sum_k A[row fr][k] * Bw[4 fq + r][k] over this wave's k-steps
    if (wave > 0) red[wave - 1][lane] = acc;
    __syncthreads();
    if (wave == 0 && row_ok && 4 * fq < W) {
        acc += red[0][lane] + red[1][lane] + red[2][lane];
        bf16x4 o = {(bf16_t)(acc[0] * alpha), (bf16_t)(acc[1] * alpha), (bf16_t)(acc[2] * alpha), (bf16_t)(acc[3] * alpha)};
        *(bf16x4*)(out + (size_t)row * W + 4 * fq) = o;
    }
}

// Sums the split-K slabs and applies the same epilogue as the fused path.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs p) {
    const int nc = p.N / 8;
    const size_t total = (size_t)p.M * nc;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int m = (int)(idx / nc), n = (int)(idx - (size_t)m * nc) * 8;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int s = 0; s < p.splitk; ++s) {
            const float* src = p.partial + ((size_t)s * p.M + m) * p.N + n;
            const f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += a[e]; v[4 + e] += b[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
        if (p.bias) {
            const f32x4 b0 = *(const f32x4*)(p.bias + n), b1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
        }
        if (p.rowbias) {
            const float* rb = p.rowbias + (size_t)(p.rowbias_shift >= 0 ? m >> p.rowbias_shift : m / p.rowbias_div) * p.ld_rowbias + n;
            const f32x4 b0 = *(const f32x4*)rb, b1 = *(const f32x4*)(rb + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
        }
        if (p.lora_t) lora_apply(p, m, n, v, p.lora_rmajor ? p.lora_b + n : p.lora_b + (size_t)n * p.lora_r, p.N, nullptr);
        if (p.residual) {
            const bf16x8 rv = *(const bf16x8*)(p.residual + (size_t)m * p.ldr + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
        }
        if (p.gb_pre != nullptr) {           // GEGLU backward in the epilogue: see GemmArgs
            const size_t row = (size_t)m * (2 * p.N);
            const bf16x8 hb = *(const bf16x8*)(p.gb_pre + row + n), gb = *(const bf16x8*)(p.gb_pre + row + p.N + n);
            bf16x8 dh, dg;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float gv = (float)gb[e];
                dh[e] = (bf16_t)(v[e] * gelu_erf_f(gv));
                dg[e] = (bf16_t)(v[e] * (float)hb[e] * gelu_erf_grad_f(gv));
            }
            *(bf16x8*)(p.gb_out + row + n) = dh;
            *(bf16x8*)(p.gb_out + row + p.N + n) = dg;
            continue;
        }
        char* outp = (char*)p.out;
        int ldo = p.ldo, nn = n;
        if (p.out2 != nullptr && n >= p.N1) { outp = (char*)p.out2; ldo = p.N - p.N1; nn = n - p.N1; }
        if (p.out_fp32) {
            float* o = (float*)outp + (size_t)m * ldo + nn;
            *(f32x4*)o = f32x4{v[0], v[1], v[2], v[3]};
            *(f32x4*)(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
            bf16x8 ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = (bf16_t)v[e];
            *(bf16x8*)((bf16_t*)outp + (size_t)m * ldo + nn) = ov;
        }
    }
}

// Tuning switches settable at run time (A/B measurements of kernel variants inside one process): mi355_tune_set().
// No key is read by the committed launchers; an experiment reads g_tune[key] where it picks its variant.
#define g_tune mi355_g_tune

template <int BN, int AMODE>
static int launch_gemm(const GemmArgs& a, hipStream_t s) {
    constexpr int smem = 2 * (128 + BN) * 128;
    static std::once_flag attr_set;
    std::call_once(attr_set, [] {
        (void)hipFuncSetAttribute((const void*)gemm_kernel<BN, AMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    });
    const int grid = a.tiles_m * a.tiles_n * a.splitk;
    hipLaunchKernelGGL((gemm_kernel<BN, AMODE>), dim3(grid), dim3(256), smem, s, a);
    return mi355_check_launch("gemm_kernel");
}

template <int AMODE>
static int launch_gemm_wide(const GemmArgs& a, unsigned a_bytes, unsigned b_bytes, hipStream_t s) {
    // two 26 KiB stages; the epilogue re-uses them for 64 staging rows + low-rank factors + low-rank activations
    // (+ the LayerNorm fold's row statistics: [256][2] + [64][2] floats)
    constexpr int smem = (64 * 164 + 16 * 160 + 64 * 16 + (AMODE == AMODE_DENSE_LN ? 2 * 256 + 256 * 8 * 2 : 0)) * 4;
    static_assert(smem >= 2 * 26 * 1024, "stages must fit");
    static std::once_flag attr_set;
    std::call_once(attr_set, [] {
        (void)hipFuncSetAttribute((const void*)gemm_wide_kernel<AMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    });
    hipLaunchKernelGGL((gemm_wide_kernel<AMODE>), dim3(a.tiles_m * a.tiles_n), dim3(256), smem, s, a, a_bytes, b_bytes);
    return mi355_check_launch("gemm_wide_kernel");
}

template <int AMODE>
static int launch_gemm_pp(const GemmArgs& a, unsigned a_bytes, unsigned b_bytes, hipStream_t s) {
    // four slabs; the epilogue re-uses them (64 staging rows of 324 floats); + [256][2] row statistics (LayerNorm fold)
    // overlaid on the up to [256][8][2] partial sums fetched for them: all of the CU's 160 KiB
    constexpr int smem = 4 * 36 * 1024 + (AMODE == AMODE_DENSE_LN ? 256 * 8 * 8 : 0);
    static_assert(4 * 36 * 1024 >= (64 * 324 + 16 * 320 + 64 * 16) * 4, "epilogue staging must fit");
    static std::once_flag attr_set;
    std::call_once(attr_set, [] {
        (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<AMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    });
    hipLaunchKernelGGL((gemm_pp_kernel<AMODE>), dim3(a.tiles_m * a.tiles_n), dim3(512), smem, s, a, a_bytes, b_bytes);
    return mi355_check_launch("gemm_pp_kernel");
}

// Split-K choice by a small cost model (all in microseconds): the grid runs in waves of `capacity`
// co-resident workgroups (2 per CU); a workgroup spends ~t_tile per 64-deep k-tile; splitting adds the
// slab write + reduce pass.  Picks the split with the smallest estimated makespan.
template <int BN, int AMODE, bool BT = false>
static int launch_gemm_dma(const GemmArgs& a, unsigned a_bytes, unsigned b_bytes, hipStream_t s) {
    constexpr int ln_extra = AMODE == AMODE_DENSE_LN ? 2 * 128 * 4 : 0;      // [128][2] row statistics behind the stages
    constexpr int smem = 2 * (128 + BN) * 128 + ln_extra;
    static std::once_flag attr_set;
    std::call_once(attr_set, [] {
        (void)hipFuncSetAttribute((const void*)gemm_dma_kernel<BN, AMODE, 4, 128, 2, BT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    });
    const int grid = a.tiles_m * a.tiles_n * a.splitk;
    // measured on the denoiser step: 8 waves win up to two workgroups per CU, 4 waves beyond
    static const int nw8_max = getenv("MI355_GEMM_NW8") ? atoi(getenv("MI355_GEMM_NW8")) : 512;
    // at most one workgroup per CU: the LDS for a third stage is free (110 KB of the CU's 160 KB); a fourth (147 KB)
    // measured 0.1 ms/step slower than three
    static const int nst3_max = getenv("MI355_GEMM_NST3") ? atoi(getenv("MI355_GEMM_NST3")) : 256;
    // k-major B costs two LDS reads per fragment: on long k loops four waves of 64 x 80 (a B fragment feeds four products
    // instead of two) beat the eight-wave forms (convolutions 1.05-1.07x -> 0.99-1.04x the time of the row-major product);
    // short loops keep the eight waves, whose second wave per SIMD hides the prologue and epilogue
    if (BT && a.K >= 2560) {
        hipLaunchKernelGGL((gemm_dma_kernel<BN, AMODE, 4, 128, 2, BT>), dim3(grid), dim3(256), smem, s, a, a_bytes, b_bytes);
        return mi355_check_launch("gemm_dma_kernel");
    }
    if (grid <= nst3_max && grid <= nw8_max) {
        constexpr int smem3 = 3 * (128 + BN) * 128 + ln_extra;
        static std::once_flag attr83_set;
        std::call_once(attr83_set, [] {
            (void)hipFuncSetAttribute((const void*)gemm_dma_kernel<BN, AMODE, 8, 128, 3, BT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem3);
        });
        hipLaunchKernelGGL((gemm_dma_kernel<BN, AMODE, 8, 128, 3, BT>), dim3(grid), dim3(512), smem3, s, a, a_bytes, b_bytes);
        return mi355_check_launch("gemm_dma_kernel<8 waves, 3 stages>");
    }
    if (grid <= nw8_max) {
        static std::once_flag attr8_set;
        std::call_once(attr8_set, [] {
            (void)hipFuncSetAttribute((const void*)gemm_dma_kernel<BN, AMODE, 8, 128, 2, BT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        });
        hipLaunchKernelGGL((gemm_dma_kernel<BN, AMODE, 8, 128, 2, BT>), dim3(grid), dim3(512), smem, s, a, a_bytes, b_bytes);
        return mi355_check_launch("gemm_dma_kernel<8 waves>");
    }
    hipLaunchKernelGGL((gemm_dma_kernel<BN, AMODE, 4, 128, 2, BT>), dim3(grid), dim3(256), smem, s, a, a_bytes, b_bytes);
    return mi355_check_launch("gemm_dma_kernel");
}

static int pick_splitk(int tiles, int nkt, double out_elems, int bn) {
    static const int capacity = getenv("MI355_GEMM_SK_CAP") ? atoi(getenv("MI355_GEMM_SK_CAP")) : 512;
    static const double sk_pen = getenv("MI355_GEMM_SK_PEN") ? atof(getenv("MI355_GEMM_SK_PEN")) : 2.5;
    static const double tt_scale = getenv("MI355_GEMM_SK_TT") ? atof(getenv("MI355_GEMM_SK_TT")) : 1.0;
    const double t_tile = tt_scale * (bn == 160 ? 1.0 : bn == 128 ? 0.8 : 0.45);
    int best = 1;
    double best_t = 1e30;
    const int smax = nkt / 2 < 16 ? (nkt / 2 < 1 ? 1 : nkt / 2) : 16;
    for (int s = 1; s <= smax; ++s) {
        const int kps = (nkt + s - 1) / s;
        const int se = (nkt + kps - 1) / kps;          // effective splits
        if (se != s) continue;
        const double waves = (double)((tiles * s + capacity - 1) / capacity);
        double t = waves * kps * t_tile + 2.0;
        if (s > 1) t += sk_pen + 2.0 * s * out_elems * 4.0 / 4.0e6;   // slab write + read at ~4 TB/s, + launch
        if (t < best_t) { best_t = t; best = s; }
    }
    return best;
}

// tiles_n_out: the column tiles taken (the partial sums per row of a LayerNorm-statistics producer).
// gs_rows_out: GroupNorm statistics were asked for (gstat_out) - receives the rows per tile they were emitted for, or 0
// when this launch could not emit them (split contraction, a tile that would span two samples, a fallback kernel): the
// product then ran as its plain form and the caller takes the statistics pass.
static int run_gemm(GemmArgs a, int amode, void* ws, size_t ws_bytes, hipStream_t s, int* tiles_n_out = nullptr,
                    int* gs_rows_out = nullptr) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return mi355_fail(MI355_ERR_ARG, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    if ((a.N & 7) || (a.K & 7)) return mi355_fail(MI355_ERR_ARG, "gemm: N (%d) and K (%d) must be multiples of 8", a.N, a.K);
    if ((a.ldb & 7) || (a.ldo & 7) || (amode_dense(amode) && (a.lda & 7)))
        return mi355_fail(MI355_ERR_ARG, "gemm: leading dimensions must be multiples of 8 elements");
    if (a.residual && (a.ldr & 7)) return mi355_fail(MI355_ERR_ARG, "gemm: residual stride must be a multiple of 8");
    if (a.rowbias && a.rowbias_div <= 0) return mi355_fail(MI355_ERR_ARG, "gemm: rowbias_div must be positive");
    if (((uintptr_t)a.bias & 15) || ((uintptr_t)a.rowbias & 15) || (a.rowbias && (a.ld_rowbias & 3)))
        return mi355_fail(MI355_ERR_ARG, "gemm: bias / rowbias must be 16-byte aligned (rowbias stride a multiple of 4 floats)");
    a.rowbias_shift = -1;
    if (a.rowbias_div > 0 && !(a.rowbias_div & (a.rowbias_div - 1))) {
        a.rowbias_shift = 0;
        while ((1 << a.rowbias_shift) < a.rowbias_div) ++a.rowbias_shift;
    }
    int bn;
    if (a.N % 160 == 0) bn = 160;
    else if (a.N <= 64) bn = 64;
    else bn = 128;
    a.tiles_m = (a.M + 127) / 128;
    a.tiles_n = (a.N + bn - 1) / bn;
    if (a.geglu) {
        if (bn != 160 || (a.N / 2) % 8) return mi355_fail(MI355_ERR_UNSUPPORTED, "gemm_geglu: 2F = %d must be a multiple of 160", a.N);
        a.tiles_n = (a.N / 2 + 79) / 80;
    }
    const int nkt = (a.K + 63) / 64;
    int sk = a.geglu ? 1 : pick_splitk(a.tiles_m * a.tiles_n, nkt, (double)a.M * a.N, bn);
    static const int force_sk = getenv("MI355_GEMM_SK") ? atoi(getenv("MI355_GEMM_SK")) : 0;     // tuning knob
    if (force_sk > 0 && !a.geglu) sk = force_sk < nkt ? force_sk : nkt;
    if (amode == AMODE_DENSE_LN && (a.rowstat_out || a.ln_part)) sk = 1;   // row statistics: emitted / applied by the tile epilogue only
    while (sk > 1 && (size_t)sk * a.M * a.N * sizeof(float) > ws_bytes) --sk;
    a.kt_per_split = (nkt + sk - 1) / sk;
    sk = (nkt + a.kt_per_split - 1) / a.kt_per_split;
    a.splitk = sk;
    a.partial = (float*)ws;
    ProfScope prof(FAM_GEMM, s, 2.0 * a.M * a.N * a.K);
    int rc;
    // LDS-DMA kernel when the shape allows it (every large layer of the denoiser does)
    static const bool force_v1 = getenv("MI355_GEMM_V1") != nullptr;
    const double a_sz = amode_dense(amode) ? ((double)(a.M - 1) * a.lda + a.K) * 2.0
                                             : (double)a.g.B * a.g.H * a.g.W * a.g.C * 2.0;
    const double b_sz = a.bt ? (amode_dense(amode) ? ((double)(a.K - 1) * a.bt_ld + a.N) * 2.0 : (double)a.K * a.N * 2.0)
                             : ((double)(a.N - 1) * a.ldb + a.K) * 2.0;
    const bool dma_ok = !force_v1 && a.K % 64 == 0 && amode != AMODE_CONV_GENERIC && a_sz < 2.0e9 && b_sz < 2.0e9 &&
                        (double)a.M * (amode_dense(amode) ? a.lda : 1) < 1.0e9;
    if (a.geglu && !dma_ok) return mi355_fail(MI355_ERR_UNSUPPORTED, "gemm_geglu: needs K %% 64 == 0 and operands < 2 GB");
    if (amode == AMODE_DENSE_LN && (a.rowstat_out || a.ln_part) && (!dma_ok || bn != 160))
        return mi355_fail(MI355_ERR_UNSUPPORTED, "gemm (LayerNorm fold): needs N %% 160 == 0, K %% 64 == 0 and operands < 2 GB");
    if (a.bt) {         // k-major B: the 128 x 160 LDS-DMA kernel only
        if (!dma_ok || bn != 160 || (amode != AMODE_DENSE && amode != AMODE_CONV_FAST) || a.geglu || (a.bt_ld & 7) ||
            (amode == AMODE_CONV_FAST && (a.g.C % 64 || a.g.ups)))
            return mi355_fail(MI355_ERR_UNSUPPORTED, "gemm (k-major B): needs N %% 160 == 0, K (conv: channels) %% 64 == 0, operands < 2 GB");
        a.bm = 128;
        const unsigned ab = (unsigned)a_sz, bb = (unsigned)b_sz;
        rc = amode == AMODE_DENSE ? launch_gemm_dma<160, AMODE_DENSE, true>(a, ab, bb, s)
                                  : launch_gemm_dma<160, AMODE_CONV_FAST, true>(a, ab, bb, s);
        if (rc) return rc;
        if (sk > 1) {
            const size_t total = (size_t)a.M * (a.N / 8);
            int grid = (int)((total + 255) / 256);
            if (grid > 2048) grid = 2048;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, s, a);
            rc = mi355_check_launch("splitk_reduce_kernel");
        }
        return rc;
    }
    a.bm = 128;
    {   // 256-row tiles when the grid still gives every CU its two workgroups
        static const int bm256 = getenv("MI355_GEMM_BM256") ? atoi(getenv("MI355_GEMM_BM256")) : 1;
        const int t256 = ((a.M + 255) / 256) * a.tiles_n;
        static const int wide_min = getenv("MI355_GEMM_WIDE_MIN") ? atoi(getenv("MI355_GEMM_WIDE_MIN")) : 512;
        if (bm256 && dma_ok && bn == 160 && sk == 1 && t256 >= wide_min && amode != AMODE_DENSE_CAT) {       // at least two workgroups per CU
            a.bm = 256;
            a.tiles_m = (a.M + 255) / 256;
        }
    }
    // 256 x 320 ping-pong tiles when they fill the 256 CUs in whole rounds (one workgroup per CU)
    bool use_pp = false;
    {
        static const int pp_on = getenv("MI355_GEMM_PP") ? atoi(getenv("MI355_GEMM_PP")) : 1;
        static const int pp_fill = getenv("MI355_GEMM_PP_FILL") ? atoi(getenv("MI355_GEMM_PP_FILL")) : 80;    // percent
        const int tn = a.geglu ? (a.N / 2 + 159) / 160 : a.N / 320;
        const int tpp = ((a.M + 255) / 256) * tn;
        const int rounds = (tpp + 255) / 256;
        if (pp_on && g_tune[1] == 0 && dma_ok && sk == 1 && amode != AMODE_DENSE_LORA && amode != AMODE_DENSE_CAT &&
            a.K % 32 == 0 && a.K >= 128 &&
            (a.K >= 512 || tpp >= 1024 || g_tune[3]) &&      // short loops: one workgroup per CU cannot hide its epilogue
            (a.geglu ? (a.N / 2) % 160 == 0 : a.N % 320 == 0) && tpp * 100 >= rounds * 256 * pp_fill) {
            use_pp = true;
            a.bm = 256;
            a.tiles_m = (a.M + 255) / 256;
            a.tiles_n = tn;
        }
    }
    if (a.gstat_out) {
        // statistics per (row tile, run of channels): only from an unsplit tile epilogue whose rows lie in one sample
        const bool ok = dma_ok && bn == 160 && sk == 1 && a.rowbias_div % a.bm == 0 && a.M % a.bm == 0 &&
                        (a.gs_unit == 8 || a.gs_unit == 10) && !a.geglu;
        if (!ok) {
            a.gstat_out = nullptr;
            if (amode == AMODE_CONV_GS) amode = AMODE_CONV_FAST;
            else if (amode == AMODE_DENSE_LN && !a.rowstat_out && !a.ln_part) amode = AMODE_DENSE;
        }
        if (gs_rows_out) *gs_rows_out = ok ? a.bm : 0;
    }
    if (tiles_n_out) *tiles_n_out = a.tiles_n;
    if (use_pp) {
        const unsigned ab = (unsigned)a_sz, bb = (unsigned)b_sz;
        rc = amode == AMODE_DENSE ? launch_gemm_pp<AMODE_DENSE>(a, ab, bb, s)
             : amode == AMODE_DENSE_LN ? launch_gemm_pp<AMODE_DENSE_LN>(a, ab, bb, s)
             : amode == AMODE_CONV_GS ? launch_gemm_pp<AMODE_CONV_GS>(a, ab, bb, s) : launch_gemm_pp<AMODE_CONV_FAST>(a, ab, bb, s);
    } else if (dma_ok && a.bm == 256) {
        const unsigned ab = (unsigned)a_sz, bb = (unsigned)b_sz;
        rc = amode == AMODE_DENSE ? launch_gemm_wide<AMODE_DENSE>(a, ab, bb, s)
             : amode == AMODE_DENSE_LORA ? launch_gemm_wide<AMODE_DENSE_LORA>(a, ab, bb, s)
             : amode == AMODE_DENSE_LN ? launch_gemm_wide<AMODE_DENSE_LN>(a, ab, bb, s)
             : amode == AMODE_CONV_GS ? launch_gemm_wide<AMODE_CONV_GS>(a, ab, bb, s)
                                      : launch_gemm_wide<AMODE_CONV_FAST>(a, ab, bb, s);
    } else if (amode == AMODE_DENSE_CAT) {
        if (!dma_ok || bn != 160)
            return mi355_fail(MI355_ERR_UNSUPPORTED, "gemm_cat: needs N %% 160 == 0, K1, K2 %% 64 == 0 and operands < 2 GB");
        rc = launch_gemm_dma<160, AMODE_DENSE_CAT>(a, (unsigned)(((double)(a.M - 1) * a.lda + a.K1) * 2.0), (unsigned)b_sz, s);
    } else if (dma_ok) {
        const unsigned ab = (unsigned)a_sz, bb = (unsigned)b_sz;
#define DISPATCH_DMA(BN) (amode == AMODE_DENSE ? launch_gemm_dma<BN, AMODE_DENSE>(a, ab, bb, s) \
                          : amode == AMODE_DENSE_LORA ? launch_gemm_dma<BN, AMODE_DENSE_LORA>(a, ab, bb, s) \
                                               : launch_gemm_dma<BN, AMODE_CONV_FAST>(a, ab, bb, s))
        if (amode == AMODE_DENSE_LN) rc = launch_gemm_dma<160, AMODE_DENSE_LN>(a, ab, bb, s);
        else if (amode == AMODE_CONV_GS) rc = launch_gemm_dma<160, AMODE_CONV_GS>(a, ab, bb, s);
        else if (bn == 160) rc = DISPATCH_DMA(160);
        else if (bn == 64) rc = DISPATCH_DMA(64);
        else rc = DISPATCH_DMA(128);
#undef DISPATCH_DMA
    } else {
#define DISPATCH(BN)                                                                       \
    (amode == AMODE_DENSE ? launch_gemm<BN, AMODE_DENSE>(a, s)                             \
     : amode == AMODE_DENSE_LORA ? launch_gemm<BN, AMODE_DENSE_LORA>(a, s)                 \
     : amode == AMODE_CONV_FAST ? launch_gemm<BN, AMODE_CONV_FAST>(a, s)                   \
                                : launch_gemm<BN, AMODE_CONV_GENERIC>(a, s))
    if (bn == 160) rc = DISPATCH(160);
    else if (bn == 64) rc = DISPATCH(64);
    else rc = DISPATCH(128);
#undef DISPATCH
    }
    if (rc) return rc;
    if (sk > 1) {
        const size_t total = (size_t)a.M * (a.N / 8);
        int grid = (int)((total + 255) / 256);
        if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, s, a);
        rc = mi355_check_launch("splitk_reduce_kernel");
    }
    return rc;
}

extern "C" {

int mi355_tune_set(int key, int value) {
    if (key < 0 || key >= 16) return mi355_fail(MI355_ERR_ARG, "tune_set: key %d", key);
    g_tune[key] = value;
    return 0;
}

size_t mi355_gemm_workspace_bytes(int M, int N) {
    // room for up to 16 split-K slabs, capped at 256 MiB (the launcher lowers the split to fit)
    size_t b = (size_t)16 * M * N * sizeof(float);
    const size_t cap = (size_t)256 << 20;
    return b > cap ? cap : b;
}

// out [M][N] = [A1 | A2] Bw^T + bias with the column concatenation read in place (A1 [M][K1], A2 [M][K2], both
// contiguous; Bw [N][K1 + K2], row stride ldb).  K1 and K2 must be multiples of 64, N of 160.
int mi355_gemm_cat_bf16(const void* A1, int K1, const void* A2, int K2, const void* Bw, int ldb, int M, int N, void* out,
                        int ldo, const float* bias, void* workspace, size_t workspace_bytes, void* stream) {
    if (!A1 || !A2 || !Bw || !out) return mi355_fail(MI355_ERR_ARG, "gemm_cat: null pointer");
    if (K1 < 64 || K2 < 64 || (K1 & 63) || (K2 & 63) || N % 160)
        return mi355_fail(MI355_ERR_UNSUPPORTED, "gemm_cat: K1 = %d, K2 = %d must be multiples of 64 and N = %d of 160", K1, K2, N);
    const double a2_sz = (double)M * K2 * 2.0;
    if (a2_sz >= 2.0e9) return mi355_fail(MI355_ERR_UNSUPPORTED, "gemm_cat: operand of 2 GB or more");
    GemmArgs a = {};
    a.A = (const bf16_t*)A1; a.A2 = (const bf16_t*)A2; a.K1 = K1; a.lda = K1; a.lda2 = K2; a.a2_bytes = (unsigned)a2_sz;
    a.Bw = (const bf16_t*)Bw; a.M = M; a.N = N; a.K = K1 + K2; a.ldb = ldb;
    a.out = out; a.ldo = ldo; a.bias = bias; a.alpha = 1.f;
    a.g = ConvGeom{0, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0};
    return run_gemm(a, AMODE_DENSE_CAT, workspace, workspace_bytes, (hipStream_t)stream);
}

// [out1 | out2] = A Bw^T written as two tensors: out1 [M][N1] takes the first N1 columns, out2 [M][N2] the rest
// (A [M][K] with row stride lda, Bw [N1 + N2][K] with row stride ldb).  N1 must be a multiple of 320 and N1 + N2 of
// 160: the kernels pick the destination once per tile, so no tile may straddle N1 - with N1 + N2 a multiple of 160 the
// launcher takes 160- or 320-column tiles, which divide N1; any other width would get 128-column tiles, which do not.
int mi355_gemm_split_bf16(const void* A, int lda, const void* Bw, int ldb, int M, int N1, int N2, int K, void* out1,
                          void* out2, void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !Bw || !out1 || !out2) return mi355_fail(MI355_ERR_ARG, "gemm_split: null pointer");
    if (N1 < 320 || N1 % 320 || N2 < 8 || (N2 & 7) || (N1 + N2) % 160)
        return mi355_fail(MI355_ERR_UNSUPPORTED, "gemm_split: N1 = %d must be a multiple of 320 and N1 + N2 = %d of 160", N1, N1 + N2);
    GemmArgs a = {};
    a.A = (const bf16_t*)A; a.Bw = (const bf16_t*)Bw; a.M = M; a.N = N1 + N2; a.K = K; a.lda = lda; a.ldb = ldb;
    a.out = out1; a.ldo = N1; a.out2 = out2; a.N1 = N1; a.alpha = 1.f;
    a.g = ConvGeom{0, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0};
    return run_gemm(a, AMODE_DENSE, workspace, workspace_bytes, (hipStream_t)stream);
}

// dpre [M][2F] = GEGLU backward of (dy = A Bw^T [M][F], pre [M][2F]) without writing dy: the data gradient of the
// feed-forward output projection with the gate's backward in its epilogue.  A [M][K] (row stride lda), Bw [F][K].
int mi355_gemm_geglu_bwd_bf16(const void* A, int lda, const void* Bw, int ldb, int M, int F, int K, const void* pre,
                              void* dpre, void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !Bw || !pre || !dpre) return mi355_fail(MI355_ERR_ARG, "gemm_geglu_bwd: null pointer");
    GemmArgs a = {};
    a.A = (const bf16_t*)A; a.Bw = (const bf16_t*)Bw; a.M = M; a.N = F; a.K = K; a.lda = lda; a.ldb = ldb;
    a.out = dpre; a.ldo = F; a.alpha = 1.f;      // (out is never written: the epilogue stores through gb_out)
    a.gb_pre = (const bf16_t*)pre; a.gb_out = (bf16_t*)dpre;
    a.g = ConvGeom{0, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0};
    return run_gemm(a, AMODE_DENSE, workspace, workspace_bytes, (hipStream_t)stream);
}

// Data gradient of a linear layer from the weight as stored: out [M][N] = A [M][K] W [K][N] (row stride ldw), i.e. the
// product with the transpose of what mi355_gemm_bf16 takes, without a transposed copy of W.  Optional, as in the
// entries above: out2 / N1 (columns n >= N1 go to out2 [M][N - N1]), pre / dpre (GEGLU backward in the epilogue: out is
// then not written).  N must be a multiple of 160, K of 64.
int mi355_gemm_bt_bf16(const void* A, int lda, const void* W, int ldw, int M, int N, int K, void* out, int ldo,
                       void* out2, int N1, const void* pre, void* dpre, void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !W || (!out && !dpre)) return mi355_fail(MI355_ERR_ARG, "gemm_bt: null pointer");
    if (out2 && (N1 < 160 || N1 % 160 || N1 >= N)) return mi355_fail(MI355_ERR_UNSUPPORTED, "gemm_bt: N1 = %d must be a multiple of 160 below N", N1);
    if ((pre != nullptr) != (dpre != nullptr)) return mi355_fail(MI355_ERR_ARG, "gemm_bt: pre and dpre go together");
    GemmArgs a = {};
    a.A = (const bf16_t*)A; a.Bw = (const bf16_t*)W; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = K;
    a.bt = 1; a.bt_ld = ldw;
    a.out = out ? out : dpre; a.ldo = out2 ? N1 : (out ? ldo : N); a.out2 = out2; a.N1 = out2 ? N1 : 0; a.alpha = 1.f;
    a.gb_pre = (const bf16_t*)pre; a.gb_out = (bf16_t*)dpre;
    a.g = ConvGeom{0, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0};
    return run_gemm(a, AMODE_DENSE, workspace, workspace_bytes, (hipStream_t)stream);
}

// Data gradient of a 3x3 / 1x1 convolution (padding ksize / 2) from the weight as stored, w [Cout][k][k][Cin]:
// dx [B][Ho][Wo][Cin] = the stride-1 convolution of dy [B][H][W][Cout] (dil: read through a zero-dilation by 2, the
// gradient of a stride-2 layer) with the transposed, tap-reversed weight - which is never built.  Cin must be a
// multiple of 160, Cout of 64.
int mi355_conv2d_dgrad_bt_bf16(const void* dy, int B, int H, int W, int Cout, const void* w, int Cin, int ksize, int dil,
                               int Ho, int Wo, void* dx, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dy || !w || !dx) return mi355_fail(MI355_ERR_ARG, "conv2d_dgrad_bt: null pointer");
    if (ksize != 1 && ksize != 3) return mi355_fail(MI355_ERR_UNSUPPORTED, "conv2d_dgrad_bt: ksize %d (1 or 3 supported)", ksize);
    if (Cout % 64 || Cin % 160) return mi355_fail(MI355_ERR_UNSUPPORTED, "conv2d_dgrad_bt: Cout = %d must be a multiple of 64, Cin = %d of 160", Cout, Cin);
    const int Hv = dil ? 2 * H : H, Wv = dil ? 2 * W : W;
    if (!dil ? (Ho != H || Wo != W) : (Ho > Hv || Wo > Wv || Ho < Hv - 1 || Wo < Wv - 1))
        return mi355_fail(MI355_ERR_ARG, "conv2d_dgrad_bt: output size %dx%d inconsistent with input %dx%d", Ho, Wo, H, W);
    const int T = ksize * ksize;
    GemmArgs a = {};
    a.A = (const bf16_t*)dy; a.Bw = (const bf16_t*)w;
    a.M = B * Ho * Wo; a.N = Cin; a.K = T * Cout; a.lda = Cout; a.ldb = a.K;
    a.bt = 1; a.bt_ld = T * Cin;
    a.out = dx; a.ldo = Cin; a.rowbias_div = Ho * Wo; a.ldr = Cin; a.alpha = 1.f;
    a.g = ConvGeom{B, H, W, Cout, Ho, Wo, 1, 0, dil, ksize, ksize / 2};
    return run_gemm(a, AMODE_CONV_FAST, workspace, workspace_bytes, (hipStream_t)stream);
}

int mi355_gemm_bf16(const void* A, int lda, const void* Bw, int ldb, int M, int N, int K, void* out, int ldo,
                    int out_fp32, const float* bias, const float* rowbias, int rowbias_div, int ld_rowbias,
                    const void* residual, int ldr, float alpha, const float* lora_t, int lora_ldt, const float* lora_b, int lora_r,
                    int lora_seg, int lora_rmajor, int lora_t_bf16, float lora_alpha, void* workspace, size_t workspace_bytes,
                    void* stream) {
    if (!A || !Bw || !out) return mi355_fail(MI355_ERR_ARG, "gemm: null pointer");
    if (lora_t && (!lora_b || lora_r < 1 || lora_ldt < lora_r || lora_ldt > 16 || (lora_seg && (lora_seg & 7)) ||
                   ((lora_r == 4 && !lora_rmajor) && (lora_ldt & 3)) ||
                   (lora_t_bf16 && ((lora_ldt != 8 && lora_ldt != 16) || (lora_r != 4 && !lora_rmajor))) ||
                   (lora_seg && (lora_r != 4 || lora_rmajor || (N / lora_seg) * lora_r > lora_ldt))))
        return mi355_fail(MI355_ERR_ARG, "gemm: bad low-rank operands (r=%d ldt=%d seg=%d)", lora_r, lora_ldt, lora_seg);
    GemmArgs a = {};
    a.A = (const bf16_t*)A; a.Bw = (const bf16_t*)Bw; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb;
    a.out = out; a.ldo = ldo; a.out_fp32 = out_fp32; a.bias = bias; a.rowbias = rowbias; a.rowbias_div = rowbias_div;
    a.ld_rowbias = ld_rowbias; a.residual = (const bf16_t*)residual; a.ldr = ldr; a.alpha = alpha;
    a.lora_t = lora_t; a.lora_b = lora_b; a.lora_r = lora_r; a.lora_alpha = lora_alpha;
    a.lora_ldt = lora_ldt; a.lora_seg = lora_seg; a.lora_rmajor = lora_rmajor; a.lora_t_bf16 = lora_t_bf16;
    a.g = ConvGeom{0, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0};
    return run_gemm(a, lora_t ? AMODE_DENSE_LORA : AMODE_DENSE, workspace, workspace_bytes, (hipStream_t)stream);
}

// LayerNorm folded into the two products around it (inference: no pass over the activations in between).
//   mi355_gemm_rowstats_bf16: out [M][N] = A Bw^T + bias + residual as mi355_gemm_bf16, and rowstat [M][parts][2] =
//       per row and column tile the sum and sum of squares of the stored bf16 values; *parts_out = the number of column
//       tiles (N / 160 or N / 320, whichever tile the launcher took).  rowstat must hold M * (N / 160) * 2 floats.
//   mi355_gemm_ln_bf16: out = LayerNorm(A) W^T + b from the un-normalised A [M][K] (K = the normalised width) and those
//       partial sums: Bw = bf16(gamma (.) W) [N][K], colsum[n] = sum_k Bw[n][k] (fp32), bias = W beta + b
//       (mi355_ln_fold_weight builds the three).  geglu != 0: N = 2F rows (hidden | gate), out [M][F] = h * gelu(g).
int mi355_gemm_rowstats_bf16(const void* A, int lda, const void* Bw, int ldb, int M, int N, int K, void* out, int ldo,
                             const float* bias, const void* residual, int ldr, float* rowstat, int* parts_out,
                             void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !Bw || !out || !rowstat || !parts_out) return mi355_fail(MI355_ERR_ARG, "gemm_rowstats: null pointer");
    GemmArgs a = {};
    a.A = (const bf16_t*)A; a.Bw = (const bf16_t*)Bw; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb;
    a.out = out; a.ldo = ldo; a.bias = bias; a.rowbias_div = 1; a.residual = (const bf16_t*)residual; a.ldr = ldr; a.alpha = 1.f;
    a.rowstat_out = rowstat;
    a.g = ConvGeom{0, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0};
    return run_gemm(a, AMODE_DENSE_LN, workspace, workspace_bytes, (hipStream_t)stream, parts_out);
}

int mi355_gemm_ln_bf16(const void* A, int lda, const void* Bw, int ldb, int M, int N, int K, void* out, int ldo,
                       const float* bias, const float* ln_part, int ln_nparts, const float* colsum, float eps, int geglu,
                       void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !Bw || !out || !ln_part || !colsum) return mi355_fail(MI355_ERR_ARG, "gemm_ln: null pointer");
    if (ln_nparts < 1 || ln_nparts > 8 || ((uintptr_t)colsum & 15) || ((uintptr_t)ln_part & 15))
        return mi355_fail(MI355_ERR_ARG, "gemm_ln: needs 1 to 8 partial sums per row, colsum and ln_part 16-byte aligned");
    GemmArgs a = {};
    a.A = (const bf16_t*)A; a.Bw = (const bf16_t*)Bw; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb;
    a.out = out; a.ldo = ldo; a.bias = bias; a.rowbias_div = 1; a.alpha = 1.f; a.geglu = geglu ? 1 : 0;
    a.ln_part = ln_part; a.ln_nparts = ln_nparts; a.ln_cs = colsum; a.ln_invc = 1.f / (float)K; a.ln_eps = eps;
    a.g = ConvGeom{0, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0};
    return run_gemm(a, AMODE_DENSE_LN, workspace, workspace_bytes, (hipStream_t)stream);
}

int mi355_gemm_skinny_bf16(const void* A, int lda, const void* Bw, int M, int W, int K, void* out, float alpha, void* stream) {
    if (!A || !Bw || !out) return mi355_fail(MI355_ERR_ARG, "gemm_skinny: null pointer");
    if (M < 1 || (W != 8 && W != 16) || K < 32 || (K & 31) || (lda & 7) || lda < K)
        return mi355_fail(MI355_ERR_ARG, "gemm_skinny: W must be 8 or 16, K a multiple of 32 (M=%d W=%d K=%d lda=%d)", M, W, K, lda);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(FAM_GEMM, s, 2.0 * M * W * K);
    if (W == 8)
        hipLaunchKernelGGL(gemm_skinny_kernel<8>, dim3((M + 15) / 16), dim3(256), 0, s, (const bf16_t*)A, lda, (const bf16_t*)Bw,
                           (bf16_t*)out, M, K, alpha);
    else
        hipLaunchKernelGGL(gemm_skinny_kernel<16>, dim3((M + 15) / 16), dim3(256), 0, s, (const bf16_t*)A, lda, (const bf16_t*)Bw,
                           (bf16_t*)out, M, K, alpha);
    return mi355_check_launch("gemm_skinny");
}

int mi355_gemm_geglu_bf16(const void* A, int lda, const void* Bw, int ldb, int M, int F, int K, void* out, int ldo,
                          const float* bias, void* preact, void* stream) {
    if (!A || !Bw || !out) return mi355_fail(MI355_ERR_ARG, "gemm_geglu: null pointer");
    GemmArgs a = {};
    a.A = (const bf16_t*)A; a.Bw = (const bf16_t*)Bw; a.M = M; a.N = 2 * F; a.K = K; a.lda = lda; a.ldb = ldb;
    a.out = out; a.ldo = ldo; a.out_fp32 = 0; a.bias = bias; a.rowbias_div = 1; a.alpha = 1.f; a.geglu = 1;
    a.preact = (bf16_t*)preact;
    a.g = ConvGeom{0, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0};
    return run_gemm(a, AMODE_DENSE, nullptr, 0, (hipStream_t)stream);
}

static int conv2d_run(const void* x, int B, int H, int W, int C, const void* w, int Cout, int ksize, int stride,
                      int ups, int dil, int Ho, int Wo, void* out, int out_fp32, const float* bias,
                      const float* rowbias, int ld_rowbias, const void* residual, float alpha, float* gstat, int gs_unit,
                      int* gs_rows_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !w || !out) return mi355_fail(MI355_ERR_ARG, "conv2d: null pointer");
    if (ksize != 1 && ksize != 3) return mi355_fail(MI355_ERR_UNSUPPORTED, "conv2d: ksize %d (1 or 3 supported)", ksize);
    if (stride != 1 && stride != 2) return mi355_fail(MI355_ERR_UNSUPPORTED, "conv2d: stride %d", stride);
    if ((C & 7) || (Cout & 7)) return mi355_fail(MI355_ERR_ARG, "conv2d: channels must be multiples of 8 (C=%d Cout=%d)", C, Cout);
    if (ups && dil) return mi355_fail(MI355_ERR_ARG, "conv2d: ups and dil are exclusive");
    const int pad = ksize / 2;
    const int Hv = ups ? 2 * H : (dil ? 2 * H : H), Wv = ups ? 2 * W : (dil ? 2 * W : W);
    // dil: the virtual input is the zero-dilated map of size (2H-1 .. 2H); any Ho in that range is legal
    if (!dil) {
        if (Ho != (Hv + 2 * pad - ksize) / stride + 1 || Wo != (Wv + 2 * pad - ksize) / stride + 1)
            return mi355_fail(MI355_ERR_ARG, "conv2d: output size %dx%d inconsistent with input %dx%d", Ho, Wo, Hv, Wv);
    } else if (Ho > Hv || Wo > Wv || Ho < Hv - 1 || Wo < Wv - 1) {
        return mi355_fail(MI355_ERR_ARG, "conv2d(dil): output size %dx%d inconsistent with input %dx%d", Ho, Wo, H, W);
    }
    GemmArgs a = {};
    a.A = (const bf16_t*)x; a.Bw = (const bf16_t*)w;
    a.M = B * Ho * Wo; a.N = Cout; a.K = ksize * ksize * C; a.lda = C; a.ldb = a.K;
    a.out = out; a.ldo = Cout; a.out_fp32 = out_fp32; a.bias = bias; a.rowbias = rowbias; a.rowbias_div = Ho * Wo;
    a.ld_rowbias = ld_rowbias; a.residual = (const bf16_t*)residual; a.ldr = Cout; a.alpha = alpha;
    a.g = ConvGeom{B, H, W, C, Ho, Wo, stride, ups, dil, ksize, ksize / 2};
    int amode = (C % 64 == 0) ? AMODE_CONV_FAST : AMODE_CONV_GENERIC;
    if (gs_rows_out) *gs_rows_out = 0;
    if (gstat && amode == AMODE_CONV_FAST && !out_fp32 && Cout % gs_unit == 0) {
        a.gstat_out = gstat; a.gs_unit = gs_unit;
        amode = AMODE_CONV_GS;
    }
    return run_gemm(a, amode, workspace, workspace_bytes, (hipStream_t)stream, nullptr, gs_rows_out);
}

int mi355_conv2d_bf16(const void* x, int B, int H, int W, int C, const void* w, int Cout, int ksize, int stride,
                      int ups, int dil, int Ho, int Wo, void* out, int out_fp32, const float* bias,
                      const float* rowbias, int ld_rowbias, const void* residual, float alpha, void* workspace,
                      size_t workspace_bytes, void* stream) {
    return conv2d_run(x, B, H, W, C, w, Cout, ksize, stride, ups, dil, Ho, Wo, out, out_fp32, bias, rowbias, ld_rowbias, residual,
                      alpha, nullptr, 0, nullptr, workspace, workspace_bytes, stream);
}

// The same convolution / product with the statistics of a following GroupNorm out of its epilogue: gstat
// [M / rows][Cout / unit][2] = per row tile (rows = *rows_out output rows, all of one sample) and run of `unit` (8 or 10)
// adjacent output channels the sum and sum of squares of the bf16 values stored.  *rows_out = 0: this launch could not
// emit them (it ran as the plain product; gstat is untouched).  gstat holds (M / 128) * (Cout / unit) * 2 floats.
int mi355_conv2d_gnstats_bf16(const void* x, int B, int H, int W, int C, const void* w, int Cout, int ksize, int stride,
                              int ups, int Ho, int Wo, void* out, const float* bias, const float* rowbias, int ld_rowbias,
                              const void* residual, float* gstat, int unit, int* rows_out, void* workspace,
                              size_t workspace_bytes, void* stream) {
    if (!gstat || !rows_out || (unit != 8 && unit != 10)) return mi355_fail(MI355_ERR_ARG, "conv2d_gnstats: gstat, rows_out, unit 8 or 10");
    return conv2d_run(x, B, H, W, C, w, Cout, ksize, stride, ups, 0, Ho, Wo, out, 0, bias, rowbias, ld_rowbias, residual, 1.f, gstat,
                      unit, rows_out, workspace, workspace_bytes, stream);
}

int mi355_gemm_gnstats_bf16(const void* A, int lda, const void* Bw, int ldb, int M, int N, int K, void* out, int ldo,
                            const float* bias, const void* residual, int ldr, int rows_per_sample, float* gstat, int unit,
                            int* rows_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !Bw || !out || !gstat || !rows_out || (unit != 8 && unit != 10) || rows_per_sample < 1)
        return mi355_fail(MI355_ERR_ARG, "gemm_gnstats: null pointer, or unit not 8 / 10");
    GemmArgs a = {};
    a.A = (const bf16_t*)A; a.Bw = (const bf16_t*)Bw; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb;
    a.out = out; a.ldo = ldo; a.bias = bias; a.rowbias_div = rows_per_sample; a.residual = (const bf16_t*)residual; a.ldr = ldr;
    a.alpha = 1.f;
    a.g = ConvGeom{0, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0};
    *rows_out = 0;
    int amode = AMODE_DENSE;
    if (N % unit == 0) { a.gstat_out = gstat; a.gs_unit = unit; amode = AMODE_DENSE_LN; }
    return run_gemm(a, amode, workspace, workspace_bytes, (hipStream_t)stream, nullptr, rows_out);
}

// 3x3 stride-2 convolution whose zero padding is one pixel on the bottom and right only: Ho = H / 2 for even H.
int mi355_conv2d_down_asym_bf16(const void* x, int B, int H, int W, int C, const void* w, int Cout, void* out,
                                const float* bias, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !w || !out) return mi355_fail(MI355_ERR_ARG, "conv2d_down_asym: null pointer");
    if ((C & 7) || (Cout & 7)) return mi355_fail(MI355_ERR_ARG, "conv2d_down_asym: channels must be multiples of 8");
    if (H < 2 || W < 2) return mi355_fail(MI355_ERR_ARG, "conv2d_down_asym: input smaller than 2x2");
    const int Ho = (H + 1 - 3) / 2 + 1, Wo = (W + 1 - 3) / 2 + 1;
    GemmArgs a = {};
    a.A = (const bf16_t*)x; a.Bw = (const bf16_t*)w;
    a.M = B * Ho * Wo; a.N = Cout; a.K = 9 * C; a.lda = C; a.ldb = a.K;
    a.out = out; a.ldo = Cout; a.bias = bias; a.rowbias_div = Ho * Wo; a.ldr = Cout; a.alpha = 1.f;
    a.g = ConvGeom{B, H, W, C, Ho, Wo, 2, 0, 0, 3, 0};
    const int amode = (C % 64 == 0) ? AMODE_CONV_FAST : AMODE_CONV_GENERIC;
    return run_gemm(a, amode, workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
