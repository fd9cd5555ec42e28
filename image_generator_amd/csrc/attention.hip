// Fused (flash-style) multi-head attention for gfx950: forward, dK/dV backward, dQ backward.
//
// Layout: q/k/v/o and their gradients are token-major [B][N][H*D] bf16 (exactly what the projection
// GEMMs write), addressed with a row stride, so no head permute is ever materialised.  The N x Nk
// score matrix never reaches HBM: K/V tiles are staged in LDS and scores live in MFMA accumulators.
//
// MFMA orientation (v_mfma_f32_32x32x16_bf16): the forward and the dQ pass compute S^T = K Q^T so a
// lane owns one query column - the softmax row statistics are lane-local plus one exchange with
// lane^32 - and the accumulator registers 8s..8s+7 are directly the B operand of the following
// O^T += V^T P^T product (V^T fragments come from ds_read_b64_tr_b16 on the row-major V tile).
// The dK/dV pass computes S = Q K^T (key on the lane) so P and dS are directly the B operands of
// dV^T += dO^T P and dK^T += Q^T dS.  Head dims 40/80/160 are zero-padded to 48/80/160 in LDS only.
// Row statistics (log-sum-exp) are stored in log2 units: lse2 = log2(sum_k exp(scale * q.k)).
#include "common.h"

struct AttnArgs {
    const bf16_t *q, *k, *v, *o, *dout;
    bf16_t *out, *dq, *dk, *dv;
    float* lse;            // [B][H][Nq]
    float* delta;          // [B][H][Nq]
    const int* kv_len;     // optional [B]: keys >= kv_len[b] are masked (key-padding mask); null = all Nk
    unsigned drop_thresh;  // dropout on the attention probabilities: an entry is dropped when its hash < drop_thresh
    unsigned drop_seed;    //   (= p * 2^32; kernels built with DROP only); kept entries are scaled by drop_scale = 1/(1-p)
    float drop_scale;
    int B, H, Nq, Nk, D;
    int ldq, ldk, ldv, ldo;
    int lddk, lddv;        // row strides of dk / dv (backward): the gradients need not be laid out like k / v themselves
    float scale_log2;      // softmax scale * log2(e)
    float scale;
};

// Dropout mask of attention entry (query q, key k) of head bh: a counter-based hash, so the forward pass and both backward
// kernels regenerate the same mask from (seed, position) and nothing is stored.
__device__ __forceinline__ bool attn_keep(const AttnArgs& p, unsigned bh, unsigned q, unsigned k) {
    unsigned x = (q * (unsigned)p.Nk + k) * 0x9E3779B1u + bh * 0x85EBCA6Bu + p.drop_seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x >= p.drop_thresh;
}

// raw v_exp_f32 (arguments here are <= 0 or -inf; no denormal range handling needed)
#define EXP2(x) __builtin_amdgcn_exp2f(x)
#define LDS_TR(ptr) __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(ptr))

__device__ __forceinline__ bf16x8 cat4(bf16x4 a, bf16x4 b) { return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }

// Row stride (bytes) of a tile that is read with transposing reads of 32-column blocks: the 4 rows a
// 16-lane group touches must sit 64 B apart modulo the 256 B bank row.
__host__ __device__ constexpr int tr_ld(int no) { return (no * 64) % 128 == 64 ? no * 64 : no * 64 + 64; }

// Cooperative copy of `nrows` x `ncols8` 16-byte chunks into LDS (rows past the tensor and columns >= D are zero
// filled), split in two so the global loads of the NEXT tile are in flight while the current tile is being multiplied:
// TileFetch::load() fills NCH registers per thread (chunk idx = threadIdx.x + 256*i), tile_store() writes them to LDS
// after the barrier that retires the current tile.
// ONES: column D of every row (the first padding column) holds 1.0 instead of 0 - a product with this tile then
// carries the row sums of the other operand in that column for free.
// SWZ: 16-byte chunk c of row r is stored at chunk c ^ ((r >> 2) & 3).  Tiles whose row stride is tr_ld() (64 or 192 B
// modulo the 256-B bank row, what the transposing reads need) put a column of 16 consecutive rows on only four 16-B bank
// slots, a 4-way conflict for the row-wise ds_read_b128 of a 16-lane group; with the swizzle the 16 rows cover all 16
// slots, and a transposing read still sees each row's 64-B block permuted within itself (its four rows share one XOR).
template <int NCH, bool SWZ = false>
__device__ __forceinline__ void tile_store(const bf16x8 (&reg)[NCH], char* dst, int dst_ld, int nrows, int ncols8) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int r = idx / ncols8, c = idx - r * ncols8;
        if (idx < nrows * ncols8) *(bf16x8*)(dst + r * dst_ld + (SWZ ? c ^ ((r >> 2) & 3) : c) * 16) = reg[i];
    }
}
// Byte offsets inside a row of a swizzled tile for the two transposing reads of a fragment (rows tr_row and tr_row + 8 of
// a 16-row block: XOR hh and hh + 2), without the 64-byte column block.
__device__ __forceinline__ void tr_cols_swz(int lane, int& c0, int& c1) {
    const int hh = lane >> 5, chunk = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1), b = (lane & 1) * 8;
    c0 = (chunk ^ hh) * 16 + b;
    c1 = (chunk ^ (hh + 2)) * 16 + b;
}

// The fetch goes through a buffer descriptor: a thread's chunks are the same (row, column) slots of every tile, so their
// byte offsets are computed once; a tile is then NCH loads at (per-thread offset, per-tile scalar offset) with no vector
// arithmetic, and the rows past the tensor come back as zeros from the descriptor's bounds check (the forward loop is
// bound by vector issue: per-tile address and bounds arithmetic was 20 of its 157 vector instructions per key tile).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int NCH>
struct TileFetch {
    __amdgpu_buffer_rsrc_t rs;
    int voff[NCH];
    unsigned one[NCH];      // 0x3F80 (bf16 1.0 in the low half of the first dword) for the ones chunk, else 0
    int row_bytes;
    template <bool ONES>
    __device__ __forceinline__ void init(const bf16_t* base, int ld, int rows_valid, int D, int nrows, int ncols8) {
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned bytes = rows_valid > 0 ? (unsigned)(((size_t)(rows_valid - 1) * ld + D) * 2) : 0u;
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
#endif
        row_bytes = ld * 2;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = threadIdx.x + 256 * i;
            const int r = idx / ncols8, c = idx - r * ncols8;
            const bool in = idx < nrows * ncols8;
            voff[i] = (in && c * 8 < D) ? (r * ld + c * 8) * 2 : (int)0x80000000;
            one[i] = (ONES && in && c * 8 == D) ? 0x3F80u : 0u;
        }
    }
    template <bool ONES>
    __device__ __forceinline__ void load(bf16x8 (&reg)[NCH], int row0) const {
#if defined(__HIP_DEVICE_COMPILE__)
        const int soff = row0 * row_bytes;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], soff, 0);
            if (ONES) v[0] |= one[i];
            reg[i] = __builtin_bit_cast(bf16x8, v);
        }
#endif
    }
};

// =============================================================================================
// forward
// =============================================================================================
// ONES (needs D < 32 * NO): the softmax denominators are accumulated by the P V product itself, through a column of
// ones in the V tile's padding - 32 fewer additions per lane and key tile in a loop that is VALU-bound at small D.
// PS (with ONES, without DROP): the caller folded scale * log2(e) into q, so a score leaves the matrix pipe as the exponent
// of its probability, and the key loop takes NO running maximum: p = exp2(s) against the fixed reference 0.  Floating
// point does not care about the common factor as long as nothing leaves the normal range, and bf16 probabilities have
// fp32's exponent range; a row whose sum (read from the ones column after the loop) is not within 2^-100 .. 2^100 - scores
// beyond about +-100 in log2 units, or a NaN - sends its workgroup through the loop again with the running maximum.  Per
// key tile and lane that is 32 v_exp + 16 conversions instead of 32 v_fma + 16 v_max3 + 32 v_exp + 16 conversions and the
// rescaling test: -13 % at 4 096 tokens, -15 % at 1 024 (d = 40).
template <int ND, int NO, bool ONES, bool DROP = false, bool PS = false>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs p) {
    static_assert(!(ONES && DROP), "the ones column would sum the dropped probabilities");
    static_assert(!PS || (ONES && !DROP), "the prescaled form reads its row sums from the ones column");
    constexpr int DP = 16 * ND;
    constexpr int KLD = DP * 2 + 16;
    constexpr int VLD = tr_ld(NO);
    __shared__ __attribute__((aligned(16))) char smem[64 * KLD + 64 * VLD];
    char* Ks = smem;
    char* Vs = smem + 64 * KLD;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    // Hardware hands consecutive workgroup ids to the 8 XCDs in turn: with the plain (tile, head) grid and at most 8
    // tiles per head, every tile of ONE head lands on a different L2 and each fetches that head's whole K / V (or
    // Q / dO) again.  The remapped id gives an XCD a contiguous run of heads with all their tiles (attention of the
    // train step 3.53 -> 3.37 ms).  With more tiles per head (4 096 tokens: 32) an XCD already runs several tiles of a
    // head at once and the remapped order measured 0.7 % slower on the sampling step, so those grids keep the plain order.
    int wg_x = blockIdx.x, wg_y = blockIdx.y;
    if (gridDim.x <= 8) {
        const int wg = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
        wg_x = wg % gridDim.x; wg_y = wg / gridDim.x;
    }
    const int b = wg_y / p.H, h = wg_y % p.H;
    const int q0 = wg_x * 128 + wave * 32;
    const bf16_t* Qb = p.q + (size_t)b * p.Nq * p.ldq + h * p.D;
    const bf16_t* Kb = p.k + (size_t)b * p.Nk * p.ldk + h * p.D;
    const bf16_t* Vb = p.v + (size_t)b * p.Nk * p.ldv + h * p.D;

    const int qrow = q0 + l31;
    const bool q_ok = qrow < p.Nq;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 qf[ND];
#pragma unroll
    for (int ks = 0; ks < ND; ++ks) {
        const int d = ks * 16 + 8 * hh;
        qf[ks] = (q_ok && d < p.D) ? *(const bf16x8*)(Qb + (size_t)qrow * p.ldq + d) : zero8;
    }
    f32x16 o[NO];
#pragma unroll
    for (int i = 0; i < NO; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    const int Nk = p.kv_len ? min(p.kv_len[b], p.Nk) : p.Nk;     // valid keys of this sample
    const int nkt = (Nk + 63) / 64;
    constexpr int NCK = (64 * (DP / 8) + 255) / 256, NCV = (64 * NO * 4 + 255) / 256;
    bf16x8 kreg[NCK], vreg[NCV];
    TileFetch<NCK> kfetch;
    TileFetch<NCV> vfetch;
    kfetch.template init<false>(Kb, p.ldk, Nk, p.D, 64, DP / 8);
    vfetch.template init<ONES>(Vb, p.ldv, Nk, p.D, 64, NO * 4);
    const float sl2 = PS ? 1.f : p.scale_log2;
    // FAST: no running maximum (PS only, see above)
    auto run_tiles = [&](auto fast_c) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fast_c)::value;
    kfetch.template load<false>(kreg, 0);
    vfetch.template load<ONES>(vreg, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * 64;
        __syncthreads();
        tile_store(kreg, Ks, KLD, 64, DP / 8);
        tile_store(vreg, Vs, VLD, 64, NO * 4);
        __syncthreads();
        if (kt + 1 < nkt) {                       // next tile's loads fly under this tile's MFMAs and softmax
            kfetch.template load<false>(kreg, k0 + 64);
            vfetch.template load<ONES>(vreg, k0 + 64);
        }

        // All K fragments of the tile are requested before the first product and the two 32-key blocks alternate, so no
        // MFMA waits for an LDS round trip or for the previous MFMA on its own accumulator (left to the compiler, every
        // product sat behind its own read and a dependent predecessor: the loop was latency-bound).
        // Small heads only (d <= 48): at d = 80 the extra live registers cost more than the waits (+7 % at 1 024 tokens).
        f32x16 s[2];
        constexpr bool VPRE = ND <= 3 && NO <= 2;
        bf16x8 kf[2][VPRE ? ND : 1];
        if (VPRE) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int ks = 0; ks < ND; ++ks)
                    kf[kb][VPRE ? ks : 0] = *(const bf16x8*)(Ks + (kb * 32 + l31) * KLD + (ks * 16 + 8 * hh) * 2);
        }
        // ... and the V^T fragments (they do not depend on the scores) while the QK^T products and the softmax run
        bf16x8 vf[VPRE ? NO : 1][2][2];
        const int tr_row = 4 * hh + ((lane & 15) >> 2);
        const int tr_col = (((lane >> 4) & 1) * 16 + 4 * (lane & 3)) * 2;
        if (VPRE) {
#pragma unroll
            for (int db = 0; db < NO; ++db)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        const char* a0 = Vs + (kb * 32 + 16 * st + tr_row) * VLD + db * 64 + tr_col;
                        vf[VPRE ? db : 0][kb][st] = cat4(LDS_TR(a0), LDS_TR(a0 + 8 * VLD));
                    }
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
        if (VPRE) {
#pragma unroll
            for (int ks = 0; ks < ND; ++ks)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
                    s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][VPRE ? ks : 0], qf[ks], s[kb], 0, 0, 0);
        } else {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int ks = 0; ks < ND; ++ks) {
                    const bf16x8 a = *(const bf16x8*)(Ks + (kb * 32 + l31) * KLD + (ks * 16 + 8 * hh) * 2);
                    s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], s[kb], 0, 0, 0);
                }
        }
        if (k0 + 64 > Nk) {            // only the last key tile can hold padded keys (wave-uniform branch)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (key >= Nk) s[kb][r] = -INFINITY;
                }
        }
        if constexpr (FAST) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = EXP2(s[kb][r]);
        } else {
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * sl2;      // scale > 0: max commutes with the scaling
        const float m_new = fmaxf(m, mx);
        const float alpha = EXP2(m - m_new);
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = EXP2(fmaf(s[kb][r], sl2, -m_new));
                s[kb][r] = e;
                if (!ONES) sum += e;
                if (DROP) {     // the denominator sums every probability, the product with V only the kept ones
                    const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    s[kb][r] = attn_keep(p, b * p.H + h, qrow, key) ? e * p.drop_scale : 0.f;
                }
            }
        if (!ONES) {
            sum += __shfl_xor(sum, 32, 64);
            l = l * alpha + sum;
        }
        // the running maximum settles after a few tiles: rescale only when some query of the wave moved it
        if (__builtin_amdgcn_ballot_w64(m_new != m) != 0) {
#pragma unroll
            for (int i = 0; i < NO; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        }
        m = m_new;
        }

        bf16x8 pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[kb][st][j] = (bf16_t)s[kb][8 * st + j];

        if (VPRE) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int db = 0; db < NO; ++db)         // output blocks alternate: no back-to-back dependent products
                        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[VPRE ? db : 0][kb][st], pf[kb][st], o[db], 0, 0, 0);
        } else {
#pragma unroll
            for (int db = 0; db < NO; ++db)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        const char* a0 = Vs + (kb * 32 + 16 * st + tr_row) * VLD + db * 64 + tr_col;
                        const bf16x8 a = cat4(LDS_TR(a0), LDS_TR(a0 + 8 * VLD));
                        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf[kb][st], o[db], 0, 0, 0);
                    }
        }
    }

    };
    // the ones column is row D % 32 of the last 32-row block of O^T: register (D % 32) / 2 of the lanes with hh = 0
    auto ones_sum = [&]() {
        const int c = p.D & 31;
        const float lsum = c == 8 ? o[NO - 1][4] : c == 16 ? o[NO - 1][8] : o[NO - 1][12];
        return __shfl(lsum, l31, 64);
    };
    if constexpr (PS) {
        m = 0.f;
        run_tiles(std::true_type{});
        l = ones_sum();
        if (__syncthreads_or(!(l > 0x1p-100f && l < 0x1p100f))) {      // (uniform over the workgroup: the loop has barriers)
#pragma unroll
            for (int i = 0; i < NO; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
            m = -INFINITY;
            run_tiles(std::false_type{});
            l = ones_sum();
        }
    } else {
        run_tiles(std::false_type{});
        if (ONES) l = ones_sum();
    }
    if (q_ok) {
        const float inv = 1.f / l;
        bf16_t* Ob = p.out + ((size_t)b * p.Nq + qrow) * p.ldo + h * p.D;
#pragma unroll
        for (int db = 0; db < NO; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = db * 32 + 8 * g + 4 * hh;
                if (d < p.D) {
                    bf16x4 ov = {(bf16_t)(o[db][4 * g] * inv), (bf16_t)(o[db][4 * g + 1] * inv),
                                 (bf16_t)(o[db][4 * g + 2] * inv), (bf16_t)(o[db][4 * g + 3] * inv)};
                    *(bf16x4*)(Ob + d) = ov;
                }
            }
        if (hh == 0 && p.lse) p.lse[((size_t)b * p.H + h) * p.Nq + qrow] = m + log2f(l);
    }
}

// The prescaled forward for long sequences: a wave owns TWO 32-query blocks (a workgroup 256 queries), so every staged K / V
// tile and every K / V^T fragment read from LDS serves twice the queries, and a wave has four independent accumulator
// chains.  With the running maximum in the loop this form gained 1 % (202 registers, two waves per SIMD: the loop was bound
// by vector issue); without it the fragment reads and the staging are what is left, and it gains 9 % at 4 096 tokens
// (1 128 -> 1 029 us per launch at d = 40) - and loses 6 % at 1 024, where it halves an already small grid.  Same
// fall-back as the one-block form: a workgroup whose row sums leave 2^-100 .. 2^100 repeats its loop with the maximum.
template <int ND, int NO>
__global__ __launch_bounds__(256, 2) void attn_fwd_ps2_kernel(const AttnArgs p) {
    constexpr int DP = 16 * ND;
    constexpr int KLD = DP * 2 + 16;
    constexpr int VLD = tr_ld(NO);
    __shared__ __attribute__((aligned(16))) char smem[64 * KLD + 64 * VLD];
    char* Ks = smem;
    char* Vs = smem + 64 * KLD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wg_x = blockIdx.x, wg_y = blockIdx.y;
    const int b = wg_y / p.H, h = wg_y % p.H;
    const int q0 = wg_x * 256 + wave * 64;
    const bf16_t* Qb = p.q + (size_t)b * p.Nq * p.ldq + h * p.D;
    const bf16_t* Kb = p.k + (size_t)b * p.Nk * p.ldk + h * p.D;
    const bf16_t* Vb = p.v + (size_t)b * p.Nk * p.ldv + h * p.D;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 qf[2][ND];
    f32x16 o[2][NO];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qrow = q0 + qb * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < ND; ++ks) {
            const int d = ks * 16 + 8 * hh;
            qf[qb][ks] = (qrow < p.Nq && d < p.D) ? *(const bf16x8*)(Qb + (size_t)qrow * p.ldq + d) : zero8;
        }
#pragma unroll
        for (int i = 0; i < NO; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][i][r] = 0.f;
    }
    const int Nk = p.kv_len ? min(p.kv_len[b], p.Nk) : p.Nk;
    const int nkt = (Nk + 63) / 64;
    constexpr int NCK = (64 * (DP / 8) + 255) / 256, NCV = (64 * NO * 4 + 255) / 256;
    bf16x8 kreg[NCK], vreg[NCV];
    TileFetch<NCK> kfetch;
    TileFetch<NCV> vfetch;
    kfetch.template init<false>(Kb, p.ldk, Nk, p.D, 64, DP / 8);
    vfetch.template init<true>(Vb, p.ldv, Nk, p.D, 64, NO * 4);
    const int tr_row = 4 * hh + ((lane & 15) >> 2);
    const int tr_col = (((lane >> 4) & 1) * 16 + 4 * (lane & 3)) * 2;
    float m[2] = {0.f, 0.f};
    auto run_tiles = [&](auto fast_c) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fast_c)::value;
    kfetch.template load<false>(kreg, 0);
    vfetch.template load<true>(vreg, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * 64;
        __syncthreads();
        tile_store(kreg, Ks, KLD, 64, DP / 8);
        tile_store(vreg, Vs, VLD, 64, NO * 4);
        __syncthreads();
        if (kt + 1 < nkt) {
            kfetch.template load<false>(kreg, k0 + 64);
            vfetch.template load<true>(vreg, k0 + 64);
        }
        bf16x8 kf[2][ND];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < ND; ++ks)
                kf[kb][ks] = *(const bf16x8*)(Ks + (kb * 32 + l31) * KLD + (ks * 16 + 8 * hh) * 2);
        f32x16 s[2][2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[qb][kb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < ND; ++ks)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
                    s[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][ks], qf[qb][ks], s[qb][kb], 0, 0, 0);
        if (k0 + 64 > Nk) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (key >= Nk) { s[0][kb][r] = -INFINITY; s[1][kb][r] = -INFINITY; }
                }
        }
        bf16x8 pf[2][2][2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float sub = 0.f;
            if constexpr (!FAST) {
                float mx = -INFINITY;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qb][kb][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float m_new = fmaxf(m[qb], mx);
                const float alpha = EXP2(m[qb] - m_new);
                if (__builtin_amdgcn_ballot_w64(m_new != m[qb]) != 0) {
#pragma unroll
                    for (int i = 0; i < NO; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[qb][i][r] *= alpha;
                }
                m[qb] = m_new;
                sub = m_new;
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        pf[qb][kb][st][j] = (bf16_t)EXP2(FAST ? s[qb][kb][8 * st + j] : s[qb][kb][8 * st + j] - sub);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int db = 0; db < NO; ++db) {
                    const char* a0 = Vs + (kb * 32 + 16 * st + tr_row) * VLD + db * 64 + tr_col;
                    const bf16x8 a = cat4(LDS_TR(a0), LDS_TR(a0 + 8 * VLD));
                    o[0][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf[0][kb][st], o[0][db], 0, 0, 0);
                    o[1][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf[1][kb][st], o[1][db], 0, 0, 0);
                }
    }
    };
    const int c = p.D & 31;
    auto ones_sum = [&](int qb) {
        const float lsum = c == 8 ? o[qb][NO - 1][4] : c == 16 ? o[qb][NO - 1][8] : o[qb][NO - 1][12];
        return __shfl(lsum, l31, 64);
    };
    run_tiles(std::true_type{});
    float lq[2] = {ones_sum(0), ones_sum(1)};
    if (__syncthreads_or(!(lq[0] > 0x1p-100f && lq[0] < 0x1p100f && lq[1] > 0x1p-100f && lq[1] < 0x1p100f))) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
            for (int i = 0; i < NO; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[qb][i][r] = 0.f;
            m[qb] = -INFINITY;
        }
        run_tiles(std::false_type{});
        lq[0] = ones_sum(0); lq[1] = ones_sum(1);
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qrow = q0 + qb * 32 + l31;
        const float l = lq[qb];
        if (qrow < p.Nq) {
            const float inv = 1.f / l;
            bf16_t* Ob = p.out + ((size_t)b * p.Nq + qrow) * p.ldo + h * p.D;
#pragma unroll
            for (int db = 0; db < NO; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = db * 32 + 8 * g + 4 * hh;
                    if (d < p.D) {
                        bf16x4 ov = {(bf16_t)(o[qb][db][4 * g] * inv), (bf16_t)(o[qb][db][4 * g + 1] * inv),
                                     (bf16_t)(o[qb][db][4 * g + 2] * inv), (bf16_t)(o[qb][db][4 * g + 3] * inv)};
                        *(bf16x4*)(Ob + d) = ov;
                    }
                }
            if (hh == 0 && p.lse) p.lse[((size_t)b * p.H + h) * p.Nq + qrow] = m[qb] + log2f(l);
        }
    }
}

// =============================================================================================
// backward: dK, dV.  Workgroup = 128 keys of one (b,h) (32 per wave); loops over 64-query tiles.
// =============================================================================================
// FQ (at most 128 keys: the workgroup holds every key of its head): the whole backward in this one launch.  dQ contracts
// over the keys, which sit on the lanes here, so dS crosses the LDS once ([key][32 queries] per 32-query sub-tile, written
// as four 8-byte runs per lane, read back by transposing reads as the B operand of dQ^T = K^T dS^T, K staged once as a
// [key][d] tile); the (sub-tile, 32-wide d block) units of a staged tile are dealt to the four waves.  delta = rowsum(dO o)
// is computed while a tile is staged.  Replaces a dQ launch that recomputes S and dP (the 64- and 16-token levels and all
// the text cross-attentions of the denoiser: 77 keys).
template <int ND, int NO, bool DROP = false, bool FQ = false>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkdv_kernel(const AttnArgs p) {
    constexpr int TLD = tr_ld(NO);      // Q / dO tiles [QT][NO*32], row + transposed reads
    constexpr int QT = NO <= 2 ? 128 : 64;   // queries per staged tile (32-query MFMA sub-tiles between one barrier pair)
    constexpr int DLD = 64 + 16;        // dS rows: 32 queries + padding (80 B: transposing reads of rows 4 apart hit distinct banks)
    constexpr int FQ_BYTES = FQ ? 128 * TLD + (QT / 32) * 128 * DLD : 0;
    __shared__ __attribute__((aligned(16))) char smem[2 * QT * TLD + 2 * QT * 4 + FQ_BYTES];
    char* Qs = smem;
    char* Gs = smem + QT * TLD;
    float* Ls = (float*)(smem + 2 * QT * TLD);   // lse2[QT], delta[QT]
    char* Kt = smem + 2 * QT * TLD + 2 * QT * 4;  // FQ: K [128 keys][NO*32] (swizzled as the dQ kernel's K tile)
    char* Ds = Kt + 128 * TLD;                    // FQ: dS [QT / 32][128 keys][32 queries]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    // Hardware hands consecutive workgroup ids to the 8 XCDs in turn: with the plain (tile, head) grid and at most 8
    // tiles per head, every tile of ONE head lands on a different L2 and each fetches that head's whole K / V (or
    // Q / dO) again.  The remapped id gives an XCD a contiguous run of heads with all their tiles (attention of the
    // train step 3.53 -> 3.37 ms).  With more tiles per head (4 096 tokens: 32) an XCD already runs several tiles of a
    // head at once and the remapped order measured 0.7 % slower on the sampling step, so those grids keep the plain order.
    int wg_x = blockIdx.x, wg_y = blockIdx.y;
    if (gridDim.x <= 8) {
        const int wg = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
        wg_x = wg % gridDim.x; wg_y = wg / gridDim.x;
    }
    const int b = wg_y / p.H, h = wg_y % p.H;
    const int key = wg_x * 128 + wave * 32 + l31;
    const bool k_ok = key < p.Nk;                                           // a row of dK / dV exists
    const bool k_valid = key < (p.kv_len ? min(p.kv_len[b], p.Nk) : p.Nk);  // and the key is not masked (else its gradients are 0)
    const bf16_t* Qb = p.q + (size_t)b * p.Nq * p.ldq + h * p.D;
    const bf16_t* Gb = p.dout + (size_t)b * p.Nq * p.ldo + h * p.D;
    const bf16_t* Kb = p.k + (size_t)b * p.Nk * p.ldk + h * p.D;
    const bf16_t* Vb = p.v + (size_t)b * p.Nk * p.ldv + h * p.D;
    const float* lse = p.lse + ((size_t)b * p.H + h) * p.Nq;
    const float* dlt = p.delta + ((size_t)b * p.H + h) * p.Nq;

    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 kf[ND], vf[ND];
#pragma unroll
    for (int ks = 0; ks < ND; ++ks) {
        const int d = ks * 16 + 8 * hh;
        const bool ok = k_ok && d < p.D;
        kf[ks] = ok ? *(const bf16x8*)(Kb + (size_t)key * p.ldk + d) : zero8;
        vf[ks] = ok ? *(const bf16x8*)(Vb + (size_t)key * p.ldv + d) : zero8;
    }
    f32x16 dkT[NO], dvT[NO];
#pragma unroll
    for (int i = 0; i < NO; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkT[i][r] = 0.f; dvT[i][r] = 0.f; }

    const int tr_row = 4 * hh + ((lane & 15) >> 2);
    int tr_c0, tr_c1;
    tr_cols_swz(lane, tr_c0, tr_c1);
    const int nqt = (p.Nq + QT - 1) / QT;
    constexpr int NCQ = (QT * NO * 4 + 255) / 256;
    bf16x8 qreg[NCQ], greg[NCQ];
    float lreg = 0.f;
    float dreg = 0.f;       // FQ: this thread's row sum of dO * o (256 / QT threads per query row)
    auto load_stats = [&](int q0) {
        if (FQ) {
            constexpr int TPR = 256 / QT;
            const int qi = q0 + threadIdx.x / TPR;
            dreg = 0.f;
            if (qi < p.Nq) {
                const bf16_t* Or = p.o + ((size_t)b * p.Nq + qi) * p.ldo + h * p.D;
                const bf16_t* Gr = Gb + (size_t)qi * p.ldo;
                for (int c = threadIdx.x % TPR; c * 8 < p.D; c += TPR) {
                    const bf16x8 ov = *(const bf16x8*)(Or + c * 8), gv = *(const bf16x8*)(Gr + c * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) dreg += (float)ov[e] * (float)gv[e];
                }
            }
#pragma unroll
            for (int off = 1; off < TPR; off <<= 1) dreg += __shfl_xor(dreg, off, 64);
            if (threadIdx.x < QT) {
                const int ql = q0 + threadIdx.x;
                lreg = ql < p.Nq ? lse[ql] : INFINITY;
            }
        } else if (threadIdx.x < 2 * QT) {
            const int qi = q0 + (threadIdx.x & (QT - 1));
            lreg = 0.f;
            if (qi < p.Nq) lreg = threadIdx.x < QT ? lse[qi] : dlt[qi];
            else if (threadIdx.x < QT) lreg = INFINITY;     // padded query rows: p = exp2(-inf) = 0
        }
    };
    if constexpr (FQ) {      // every key of the head, once: [key][d], rows past the valid keys zero (they add nothing to dQ)
        constexpr int NCK = (128 * NO * 4 + 255) / 256;
        bf16x8 kreg[NCK];
        TileFetch<NCK> kfetch;
        kfetch.template init<false>(Kb, p.ldk, p.kv_len ? min(p.kv_len[b], p.Nk) : p.Nk, p.D, 128, NO * 4);
        kfetch.template load<false>(kreg, 0);
        tile_store<NCK, true>(kreg, Kt, TLD, 128, NO * 4);
    }
    TileFetch<NCQ> qfetch, gfetch;
    qfetch.template init<false>(Qb, p.ldq, p.Nq, p.D, QT, NO * 4);
    gfetch.template init<false>(Gb, p.ldo, p.Nq, p.D, QT, NO * 4);
    qfetch.template load<false>(qreg, 0);
    gfetch.template load<false>(greg, 0);
    load_stats(0);
    for (int qt = 0; qt < nqt; ++qt) {
        const int q0 = qt * QT;
        __syncthreads();
        tile_store<NCQ, true>(qreg, Qs, TLD, QT, NO * 4);
        tile_store<NCQ, true>(greg, Gs, TLD, QT, NO * 4);
        if (FQ) {
            if (threadIdx.x < QT) Ls[threadIdx.x] = lreg;
            if (threadIdx.x % (256 / QT) == 0) Ls[QT + threadIdx.x / (256 / QT)] = dreg;
        } else if (threadIdx.x < 2 * QT) Ls[threadIdx.x] = lreg;
        __syncthreads();
        if (qt + 1 < nqt) {
            qfetch.template load<false>(qreg, q0 + QT);
            gfetch.template load<false>(greg, q0 + QT);
            load_stats(q0 + QT);
        }
#pragma unroll
        for (int qs = 0; qs < QT / 32; ++qs) {
        if (q0 + qs * 32 >= p.Nq) break;         // (uniform) the rest of the staged tile is padding
        const char* Qt = Qs + qs * 32 * TLD;
        const char* Gt = Gs + qs * 32 * TLD;
        const float* Lt = Ls + qs * 32;

        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < ND; ++ks) {
            const int off = l31 * TLD + ((ks * 2 + hh) ^ ((l31 >> 2) & 3)) * 16;
            const bf16x8 aq = *(const bf16x8*)(Qt + off);
            const bf16x8 ag = *(const bf16x8*)(Gt + off);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, kf[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ag, vf[ks], dp, 0, 0, 0);
        }
        bf16x8 pf[2], dsf[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qi = (r & 3) + 8 * (r >> 2) + 4 * hh;
            // a masked key (lane) is not zeroed here: its column of P / dS only reaches its own rows of dK / dV, which
            // the store below replaces with zeros (one select per score less in a loop that is bound by vector issue)
            const float pv = EXP2(fmaf(s[r], p.scale_log2, -Lt[qi]));
            float pd = pv, dpv = dp[r];
            if (DROP) {
                const bool keep = attn_keep(p, b * p.H + h, q0 + qs * 32 + qi, key);
                pd = keep ? pv * p.drop_scale : 0.f;
                dpv = keep ? dpv * p.drop_scale : 0.f;
            }
            const float dsv = pv * (dpv - Lt[QT + qi]);
            pf[r >> 3][r & 7] = (bf16_t)pd;
            dsf[r >> 3][r & 7] = (bf16_t)dsv;
        }
        if constexpr (FQ) {     // dS of this wave's 32 keys x 32 queries -> LDS [key][query]; a masked key contributes nothing
            char* row = Ds + (qs * 128 + wave * 32 + l31) * DLD + hh * 8;
            const bf16x4 z4 = {0, 0, 0, 0};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 v4 = {dsf[g >> 1][(g & 1) * 4], dsf[g >> 1][(g & 1) * 4 + 1], dsf[g >> 1][(g & 1) * 4 + 2],
                             dsf[g >> 1][(g & 1) * 4 + 3]};
                *(bf16x4*)(row + g * 16) = k_valid ? v4 : z4;
            }
        }
#pragma unroll
        for (int db = 0; db < NO; ++db)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int off = (16 * st + tr_row) * TLD + db * 64;
                const bf16x8 ag = cat4(LDS_TR(Gt + off + tr_c0), LDS_TR(Gt + off + 8 * TLD + tr_c1));
                dvT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ag, pf[st], dvT[db], 0, 0, 0);
                const bf16x8 aq = cat4(LDS_TR(Qt + off + tr_c0), LDS_TR(Qt + off + 8 * TLD + tr_c1));
                dkT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, dsf[st], dkT[db], 0, 0, 0);
            }
        }
        if constexpr (FQ) {
            __syncthreads();                    // every wave's dS of this staged tile is in LDS
            const int nkb = (p.Nk + 31) / 32;   // 32-key blocks that hold keys
            for (int u = wave; u < (QT / 32) * NO; u += 4) {
                const int qs = u / NO, db = u - qs * NO;
                const int qrow = q0 + qs * 32 + l31;
                if (q0 + qs * 32 >= p.Nq) break;        // (uniform) units are ordered by sub-tile
                f32x16 dq;
#pragma unroll
                for (int r = 0; r < 16; ++r) dq[r] = 0.f;
                for (int kb = 0; kb < nkb; ++kb)
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        const int krow = kb * 32 + 16 * st + tr_row;
                        const bf16x8 ak = cat4(LDS_TR(Kt + krow * TLD + db * 64 + tr_c0), LDS_TR(Kt + (krow + 8) * TLD + db * 64 + tr_c1));
                        const char* d0 = Ds + (qs * 128 + krow) * DLD + (((lane >> 4) & 1) * 16 + 4 * (lane & 3)) * 2;
                        const bf16x8 bd = cat4(LDS_TR(d0), LDS_TR(d0 + 8 * DLD));
                        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ak, bd, dq, 0, 0, 0);
                    }
                if (qrow < p.Nq) {
                    bf16_t* dQ = p.dq + ((size_t)b * p.Nq + qrow) * p.ldq + h * p.D;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int d = db * 32 + 8 * g + 4 * hh;
                        if (d < p.D) {
                            bf16x4 a = {(bf16_t)(dq[4 * g] * p.scale), (bf16_t)(dq[4 * g + 1] * p.scale),
                                        (bf16_t)(dq[4 * g + 2] * p.scale), (bf16_t)(dq[4 * g + 3] * p.scale)};
                            *(bf16x4*)(dQ + d) = a;
                        }
                    }
                }
            }
        }
    }
    if (k_ok) {
        bf16_t* dK = p.dk + ((size_t)b * p.Nk + key) * p.lddk + h * p.D;
        bf16_t* dV = p.dv + ((size_t)b * p.Nk + key) * p.lddv + h * p.D;
#pragma unroll
        for (int db = 0; db < NO; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = db * 32 + 8 * g + 4 * hh;
                if (d < p.D) {
                    bf16x4 a = {(bf16_t)(dkT[db][4 * g] * p.scale), (bf16_t)(dkT[db][4 * g + 1] * p.scale),
                                (bf16_t)(dkT[db][4 * g + 2] * p.scale), (bf16_t)(dkT[db][4 * g + 3] * p.scale)};
                    bf16x4 c = {(bf16_t)dvT[db][4 * g], (bf16_t)dvT[db][4 * g + 1], (bf16_t)dvT[db][4 * g + 2],
                                (bf16_t)dvT[db][4 * g + 3]};
                    if (!k_valid) { a = bf16x4{0, 0, 0, 0}; c = a; }        // masked key: no gradient
                    *(bf16x4*)(dK + d) = a;
                    *(bf16x4*)(dV + d) = c;
                }
            }
    }
}

// =============================================================================================
// backward: dQ.  Workgroup = 128 queries of one (b,h) (32 per wave); loops over 64-key tiles.
// =============================================================================================
template <int ND, int NO, bool DROP = false>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq_kernel(const AttnArgs p) {
    constexpr int DP = 16 * ND;
    constexpr int KLD = tr_ld(NO);        // K tile: row reads (S^T) + transposed reads (dQ^T)
    constexpr int VLD = DP * 2 + 16;      // V tile: row reads only
    __shared__ __attribute__((aligned(16))) char smem[64 * KLD + 64 * VLD];
    char* Ks = smem;
    char* Vs = smem + 64 * KLD;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    // Hardware hands consecutive workgroup ids to the 8 XCDs in turn: with the plain (tile, head) grid and at most 8
    // tiles per head, every tile of ONE head lands on a different L2 and each fetches that head's whole K / V (or
    // Q / dO) again.  The remapped id gives an XCD a contiguous run of heads with all their tiles (attention of the
    // train step 3.53 -> 3.37 ms).  With more tiles per head (4 096 tokens: 32) an XCD already runs several tiles of a
    // head at once and the remapped order measured 0.7 % slower on the sampling step, so those grids keep the plain order.
    int wg_x = blockIdx.x, wg_y = blockIdx.y;
    if (gridDim.x <= 8) {
        const int wg = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
        wg_x = wg % gridDim.x; wg_y = wg / gridDim.x;
    }
    const int b = wg_y / p.H, h = wg_y % p.H;
    const int qrow = wg_x * 128 + wave * 32 + l31;
    const bool q_ok = qrow < p.Nq;
    const bf16_t* Qb = p.q + (size_t)b * p.Nq * p.ldq + h * p.D;
    const bf16_t* Gb = p.dout + (size_t)b * p.Nq * p.ldo + h * p.D;
    const bf16_t* Kb = p.k + (size_t)b * p.Nk * p.ldk + h * p.D;
    const bf16_t* Vb = p.v + (size_t)b * p.Nk * p.ldv + h * p.D;

    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 qf[ND], gf[ND];
#pragma unroll
    for (int ks = 0; ks < ND; ++ks) {
        const int d = ks * 16 + 8 * hh;
        const bool ok = q_ok && d < p.D;
        qf[ks] = ok ? *(const bf16x8*)(Qb + (size_t)qrow * p.ldq + d) : zero8;
        gf[ks] = ok ? *(const bf16x8*)(Gb + (size_t)qrow * p.ldo + d) : zero8;
    }
    const float lse = q_ok ? p.lse[((size_t)b * p.H + h) * p.Nq + qrow] : INFINITY;
    // delta[q] = sum_d dO[q][d] * O[q][d]: this pass owns every query row once, so it also produces the row
    // sums the dK/dV pass (launched after it) reads
    float dlt = 0.f;
    {
        const bf16_t* Ob = p.o + (size_t)b * p.Nq * p.ldo + h * p.D;
#pragma unroll
        for (int ks = 0; ks < ND; ++ks) {
            const int d = ks * 16 + 8 * hh;
            if (q_ok && d < p.D) {
                const bf16x8 ov = *(const bf16x8*)(Ob + (size_t)qrow * p.ldo + d);
#pragma unroll
                for (int e = 0; e < 8; ++e) dlt += (float)ov[e] * (float)gf[ks][e];
            }
        }
        dlt += __shfl_xor(dlt, 32, 64);
        if (q_ok && hh == 0) p.delta[((size_t)b * p.H + h) * p.Nq + qrow] = dlt;
    }
    f32x16 dqT[NO];
#pragma unroll
    for (int i = 0; i < NO; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqT[i][r] = 0.f;

    const int tr_row = 4 * hh + ((lane & 15) >> 2);
    int tr_c0, tr_c1;
    tr_cols_swz(lane, tr_c0, tr_c1);
    const int Nk = p.kv_len ? min(p.kv_len[b], p.Nk) : p.Nk;     // valid keys of this sample
    const int nkt = (Nk + 63) / 64;
    constexpr int NCK = (64 * NO * 4 + 255) / 256, NCV = (64 * (DP / 8) + 255) / 256;
    bf16x8 kreg[NCK], vreg[NCV];
    TileFetch<NCK> kfetch;
    TileFetch<NCV> vfetch;
    kfetch.template init<false>(Kb, p.ldk, Nk, p.D, 64, NO * 4);
    vfetch.template init<false>(Vb, p.ldv, Nk, p.D, 64, DP / 8);
    kfetch.template load<false>(kreg, 0);
    vfetch.template load<false>(vreg, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * 64;
        __syncthreads();
        tile_store<NCK, true>(kreg, Ks, KLD, 64, NO * 4);
        tile_store(vreg, Vs, VLD, 64, DP / 8);
        __syncthreads();
        if (kt + 1 < nkt) {
            kfetch.template load<false>(kreg, k0 + 64);
            vfetch.template load<false>(vreg, k0 + 64);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < ND; ++ks) {
                const bf16x8 ak = *(const bf16x8*)(Ks + (kb * 32 + l31) * KLD + ((ks * 2 + hh) ^ ((l31 >> 2) & 3)) * 16);
                const bf16x8 av = *(const bf16x8*)(Vs + (kb * 32 + l31) * VLD + (ks * 16 + 8 * hh) * 2);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ak, qf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, gf[ks], dp, 0, 0, 0);
            }
            bf16x8 dsf[2];
            if (k0 + 64 > Nk) {            // only the last key tile can hold padded keys (wave-uniform branch)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= Nk) s[r] = -INFINITY;     // p = exp2(-inf) = 0
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const float pv = EXP2(fmaf(s[r], p.scale_log2, -lse));
                float dpv = dp[r];
                if (DROP) dpv = attn_keep(p, b * p.H + h, qrow, key) ? dpv * p.drop_scale : 0.f;
                dsf[r >> 3][r & 7] = (bf16_t)(pv * (dpv - dlt));
            }
#pragma unroll
            for (int db = 0; db < NO; ++db)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const int off = (kb * 32 + 16 * st + tr_row) * KLD + db * 64;
                    const bf16x8 a = cat4(LDS_TR(Ks + off + tr_c0), LDS_TR(Ks + off + 8 * KLD + tr_c1));
                    dqT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, dsf[st], dqT[db], 0, 0, 0);
                }
        }
    }
    if (q_ok) {
        bf16_t* dQ = p.dq + ((size_t)b * p.Nq + qrow) * p.ldq + h * p.D;
#pragma unroll
        for (int db = 0; db < NO; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = db * 32 + 8 * g + 4 * hh;
                if (d < p.D) {
                    bf16x4 a = {(bf16_t)(dqT[db][4 * g] * p.scale), (bf16_t)(dqT[db][4 * g + 1] * p.scale),
                                (bf16_t)(dqT[db][4 * g + 2] * p.scale), (bf16_t)(dqT[db][4 * g + 3] * p.scale)};
                    *(bf16x4*)(dQ + d) = a;
                }
            }
    }
}

template <int ND, int NO>
static int launch_fwd(const AttnArgs& a, hipStream_t s) {
    // a spare padding column exists: row sums ride the P V product (measured: +4 % at D = 40, N = 4096; a loss at
    // D = 80, where the extra column costs a third 32-wide block nothing but the additions it saves are fewer per MFMA)
    if (NO <= 2 && a.D < 32 * NO)
        hipLaunchKernelGGL((attn_fwd_kernel<ND, NO, true>), dim3((a.Nq + 127) / 128, a.B * a.H), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((attn_fwd_kernel<ND, NO, false>), dim3((a.Nq + 127) / 128, a.B * a.H), dim3(256), 0, s, a);
    return mi355_check_launch("attn_fwd");
}
// q already carries scale * log2(e) (a.scale_log2 == 1): heads with a spare padding column take the loop without a running
// maximum, the others the plain kernel.
template <int ND, int NO>
static int launch_fwd_ps(const AttnArgs& a, hipStream_t s) {
    if constexpr (NO <= 2) {      // (d = 80 in three 32-wide blocks: measured, +-0 in the sampling step)
        // long sequences: 64 queries per wave.  mi355_tune_set(0, 2): never, (0, 3): whenever the kernel applies
        if (a.D < 32 * NO && mi355_g_tune[0] != 1 && mi355_g_tune[0] != 2 && ((a.Nq >= 2048 && a.Nk >= 1024) || mi355_g_tune[0] == 3)) {
            hipLaunchKernelGGL((attn_fwd_ps2_kernel<ND, NO>), dim3((a.Nq + 255) / 256, a.B * a.H), dim3(256), 0, s, a);
            return mi355_check_launch("attn_fwd(prescaled, 64 queries per wave)");
        }
        if (a.D < 32 * NO && mi355_g_tune[0] != 1) {      // mi355_tune_set(0, 1): plain kernel (A/B inside one process)
            hipLaunchKernelGGL((attn_fwd_kernel<ND, NO, true, false, true>), dim3((a.Nq + 127) / 128, a.B * a.H), dim3(256), 0, s, a);
            return mi355_check_launch("attn_fwd(prescaled)");
        }
    }
    return launch_fwd<ND, NO>(a, s);
}
template <int ND, int NO>
static int launch_bwd(const AttnArgs& a, hipStream_t s) {
    if (a.Nk <= 128 && mi355_g_tune[1] != 1) {       // one launch (mi355_tune_set(1, 1): two, for A/B inside one process)
        hipLaunchKernelGGL((attn_bwd_dkdv_kernel<ND, NO, false, true>), dim3(1, a.B * a.H), dim3(256), 0, s, a);
        return mi355_check_launch("attn_bwd(one launch)");
    }
    // dQ first: it writes delta (row sums of dO * O), which the dK/dV pass consumes
    hipLaunchKernelGGL((attn_bwd_dq_kernel<ND, NO>), dim3((a.Nq + 127) / 128, a.B * a.H), dim3(256), 0, s, a);
    hipLaunchKernelGGL((attn_bwd_dkdv_kernel<ND, NO>), dim3((a.Nk + 127) / 128, a.B * a.H), dim3(256), 0, s, a);
    return mi355_check_launch("attn_bwd");
}

// The variants with dropout on the attention probabilities (encoder pre-training, encoder/text_encoder.py:16-23): head
// dims up to 96.
template <int ND, int NO>
static int launch_fwd_drop(const AttnArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((attn_fwd_kernel<ND, NO, false, true>), dim3((a.Nq + 127) / 128, a.B * a.H), dim3(256), 0, s, a);
    return mi355_check_launch("attn_fwd(dropout)");
}
template <int ND, int NO>
static int launch_bwd_drop(const AttnArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<ND, NO, true>), dim3((a.Nq + 127) / 128, a.B * a.H), dim3(256), 0, s, a);
    hipLaunchKernelGGL((attn_bwd_dkdv_kernel<ND, NO, true>), dim3((a.Nk + 127) / 128, a.B * a.H), dim3(256), 0, s, a);
    return mi355_check_launch("attn_bwd(dropout)");
}
#define ATTN_DISPATCH_DROP(FN, a, s)                                     \
    ((a).D <= 16 ? FN<1, 1>(a, s) : (a).D <= 32 ? FN<2, 1>(a, s)         \
     : (a).D <= 64 ? FN<4, 2>(a, s) : FN<6, 3>(a, s))

static int set_dropout(AttnArgs& a, float drop_p, unsigned seed) {
    if (!(drop_p >= 0.f) || drop_p >= 1.f) return mi355_fail(MI355_ERR_ARG, "attention: dropout probability %g", (double)drop_p);
    if (drop_p > 0.f && a.D > 96) return mi355_fail(MI355_ERR_UNSUPPORTED, "attention: dropout needs head dim <= 96 (got %d)", a.D);
    a.drop_thresh = (unsigned)((double)drop_p * 4294967296.0);
    a.drop_scale = 1.f / (1.f - drop_p);
    a.drop_seed = seed;
    return 0;
}

static int check_attn(const AttnArgs& a) {
    if (a.D % 8 || a.D < 8 || a.D > 160) return mi355_fail(MI355_ERR_UNSUPPORTED, "attention: head dim %d (multiple of 8, <= 160)", a.D);
    if ((a.ldq | a.ldk | a.ldv | a.ldo) & 7) return mi355_fail(MI355_ERR_ARG, "attention: row strides must be multiples of 8");
    if (a.Nq <= 0 || a.Nk <= 0 || a.B <= 0 || a.H <= 0) return mi355_fail(MI355_ERR_ARG, "attention: empty problem");
    if ((long long)a.B * a.H > 65535) return mi355_fail(MI355_ERR_UNSUPPORTED, "attention: B*H > 65535");
    return 0;
}

#define ATTN_DISPATCH(FN, a, s)                                          \
    ((a).D <= 16 ? FN<1, 1>(a, s) : (a).D <= 32 ? FN<2, 1>(a, s)         \
     : (a).D <= 48 ? FN<3, 2>(a, s) : (a).D <= 64 ? FN<4, 2>(a, s)       \
     : (a).D <= 80 ? FN<5, 3>(a, s) : (a).D <= 96 ? FN<6, 3>(a, s)       \
     : (a).D <= 128 ? FN<8, 4>(a, s) : FN<10, 5>(a, s))

extern "C" {

int mi355_attention_fwd(const void* q, const void* k, const void* v, void* out, float* lse, int B, int H, int Nq,
                        int Nk, int D, int ldq, int ldk, int ldv, int ldo, float scale, void* stream) {
    if (!q || !k || !v || !out) return mi355_fail(MI355_ERR_ARG, "attention_fwd: null pointer");
    AttnArgs a = {};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.out = (bf16_t*)out; a.lse = lse;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.D = D; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.scale = scale; a.scale_log2 = scale * 1.4426950408889634f;
    if (int rc = check_attn(a)) return rc;
    ProfScope prof(FAM_ATTN, (hipStream_t)stream, 4.0 * B * H * (double)Nq * Nk * D);
    return ATTN_DISPATCH(launch_fwd, a, (hipStream_t)stream);
}

// Forward on queries that already carry the softmax scale and the change of base: scores = q k^T are exponents of 2
// (the caller folded scale * log2(e) into q, e.g. into the weights of the projection that produced it).  Same result
// as mi355_attention_fwd on the unscaled q up to the rounding of q; lse in log2 units as everywhere.
int mi355_attention_fwd_prescaled(const void* q, const void* k, const void* v, void* out, float* lse, int B, int H, int Nq,
                                  int Nk, int D, int ldq, int ldk, int ldv, int ldo, void* stream) {
    if (!q || !k || !v || !out) return mi355_fail(MI355_ERR_ARG, "attention_fwd_prescaled: null pointer");
    AttnArgs a = {};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.out = (bf16_t*)out; a.lse = lse;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.D = D; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.scale = 0.6931471805599453f; a.scale_log2 = 1.f;
    if (int rc = check_attn(a)) return rc;
    ProfScope prof(FAM_ATTN, (hipStream_t)stream, 4.0 * B * H * (double)Nq * Nk * D);
    return ATTN_DISPATCH(launch_fwd_ps, a, (hipStream_t)stream);
}

// Forward with a key-padding mask given as valid key counts per sample (1 <= kv_len[b] <= Nk, device int32).
int mi355_attention_fwd_masked(const void* q, const void* k, const void* v, void* out, float* lse, const int* kv_len,
                               int B, int H, int Nq, int Nk, int D, int ldq, int ldk, int ldv, int ldo, float scale,
                               void* stream) {
    if (!q || !k || !v || !out || !kv_len) return mi355_fail(MI355_ERR_ARG, "attention_fwd_masked: null pointer");
    AttnArgs a = {};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.out = (bf16_t*)out; a.lse = lse;
    a.kv_len = kv_len;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.D = D; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.scale = scale; a.scale_log2 = scale * 1.4426950408889634f;
    if (int rc = check_attn(a)) return rc;
    ProfScope prof(FAM_ATTN, (hipStream_t)stream, 4.0 * B * H * (double)Nq * Nk * D);
    return ATTN_DISPATCH(launch_fwd, a, (hipStream_t)stream);
}

// Forward / backward of the encoder's self-attention in training mode: key-padding mask as valid key counts (kv_len, may
// be null) and dropout with probability drop_p on the attention probabilities; the mask of an entry is a hash of
// (seed, head, query, key), so the backward call regenerates it from the same seed.
int mi355_attention_fwd_train(const void* q, const void* k, const void* v, void* out, float* lse, const int* kv_len,
                              float drop_p, unsigned seed, int B, int H, int Nq, int Nk, int D, int ldq, int ldk, int ldv,
                              int ldo, float scale, void* stream) {
    if (!q || !k || !v || !out || !lse) return mi355_fail(MI355_ERR_ARG, "attention_fwd_train: null pointer");
    AttnArgs a = {};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.out = (bf16_t*)out; a.lse = lse;
    a.kv_len = kv_len;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.D = D; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.scale = scale; a.scale_log2 = scale * 1.4426950408889634f;
    if (int rc = check_attn(a)) return rc;
    if (int rc = set_dropout(a, drop_p, seed)) return rc;
    ProfScope prof(FAM_ATTN, (hipStream_t)stream, 4.0 * B * H * (double)Nq * Nk * D);
    if (a.drop_thresh == 0) return ATTN_DISPATCH(launch_fwd, a, (hipStream_t)stream);
    return ATTN_DISPATCH_DROP(launch_fwd_drop, a, (hipStream_t)stream);
}
int mi355_attention_bwd_train(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                              float* delta, void* dq, void* dk, void* dv, const int* kv_len, float drop_p, unsigned seed,
                              int B, int H, int Nq, int Nk, int D, int ldq, int ldk, int ldv, int ldo, float scale,
                              void* stream) {
    if (!q || !k || !v || !o || !dout || !lse || !delta || !dq || !dk || !dv)
        return mi355_fail(MI355_ERR_ARG, "attention_bwd_train: null pointer");
    AttnArgs a = {};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (const bf16_t*)o;
    a.dout = (const bf16_t*)dout; a.lse = (float*)lse; a.delta = delta; a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk;
    a.dv = (bf16_t*)dv; a.kv_len = kv_len;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.D = D; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.lddk = ldk; a.lddv = ldv;
    a.scale = scale; a.scale_log2 = scale * 1.4426950408889634f;
    if (int rc = check_attn(a)) return rc;
    if (int rc = set_dropout(a, drop_p, seed)) return rc;
    // algorithmic work: dV, dP, dQ, dK are four products of 2 B H Nq Nk D (the recomputed scores are not counted)
    ProfScope prof(FAM_ATTN, (hipStream_t)stream, 8.0 * B * H * (double)Nq * Nk * D);
    if (a.drop_thresh == 0) return ATTN_DISPATCH(launch_bwd, a, (hipStream_t)stream);
    return ATTN_DISPATCH_DROP(launch_bwd_drop, a, (hipStream_t)stream);
}

// delta: scratch [B][H][Nq] floats.  dq uses ldq, dout/o use ldo; dk / dv use lddk / lddv (0: ldk / ldv, the layout of
// k / v themselves - keys and values that are column slices of a wider matrix can have contiguous gradients).
int mi355_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                        float* delta, void* dq, void* dk, void* dv, int B, int H, int Nq, int Nk, int D, int ldq,
                        int ldk, int ldv, int ldo, int lddk, int lddv, float scale, void* stream) {
    if (!q || !k || !v || !o || !dout || !lse || !delta || !dq || !dk || !dv)
        return mi355_fail(MI355_ERR_ARG, "attention_bwd: null pointer");
    AttnArgs a = {};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (const bf16_t*)o;
    a.dout = (const bf16_t*)dout; a.lse = (float*)lse; a.delta = delta; a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk;
    a.dv = (bf16_t*)dv; a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.D = D; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.lddk = lddk > 0 ? lddk : ldk; a.lddv = lddv > 0 ? lddv : ldv;
    if ((a.lddk & 7) || (a.lddv & 7) || a.lddk < H * D || a.lddv < H * D)
        return mi355_fail(MI355_ERR_ARG, "attention_bwd: dk / dv row strides must be multiples of 8 and hold H * D values");
    a.scale = scale; a.scale_log2 = scale * 1.4426950408889634f;
    if (int rc = check_attn(a)) return rc;
    ProfScope prof(FAM_ATTN, (hipStream_t)stream, 8.0 * B * H * (double)Nq * Nk * D);
    return ATTN_DISPATCH(launch_bwd, a, (hipStream_t)stream);
}

}  // extern "C"
