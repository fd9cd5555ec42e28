// Weight-gradient GEMM for gfx950:  dW[co][tap][ci] (+)= sum_m dy[m][co] * xcol[m][tap][ci]
//
// The contraction runs over pixels/tokens m, which is the SLOW dimension of both operands
// (dy is [M][Cout], x is NHWC).  Tiles are therefore staged in LDS in their natural
// [pixel][channel] order with 16-byte coalesced copies and the MFMA fragments are fetched with the
// gfx950 transposing LDS read (ds_read_b64_tr_b16), so no transposed copy of an activation is ever
// written to HBM.  The im2col view of x (3x3 taps, stride 1/2, optional nearest-2x upsampled input)
// is gathered by the loader; linear layers are the 1x1 / dense case.
//
// Tile: 160(co) x 160(ci) x 64 pixels per 256-thread workgroup (2x2 waves, 5x5 MFMA 16x16x32 each).
// LDS row stride 352 B = 11 x 32 B keeps the 8 rows touched by one half-wave transposed read on
// distinct banks.  The pixel range is split across workgroups (split-K); partial sums are combined
// with fp32 atomics into a zero-initialised (or accumulating) fp32 gradient buffer.
#include "common.h"
#include <algorithm>
#include <stdlib.h>

struct WgradArgs {
    const bf16_t* dy;   // [M][Cout]
    const bf16_t* x;    // NHWC [B][H][W][Cin]  (or dense [M][Cin])
    float* dw;          // [Cout][taps][Cin] fp32
    int M, Cout, Cin;
    int B, H, W, Ho, Wo, stride, ups, ksize;
    int tiles_co, tiles_ci, taps, splits, kt_per_split;
    int use_atomic;
    int ldx;            // dense: row stride of x
    float* dbias;       // optional [Cout] fp32, pre-zeroed: column sums of dy (bias gradient), or null
    unsigned dy_bytes, x_bytes;   // extents of dy / x for the LDS-DMA loader's buffer descriptors (< 2 GiB)
    int lw, lh;         // log2(Wo), log2(Ho) when both are powers of two, else -1
    int nmain;          // LDS-DMA launch: number of workgroups that have work (a grouped launch pads each job)
    float* sumsq;       // optional WG_SQ_SLOTS fp32 slots: unsplit jobs add the sum of squares of what they store (the global
                        // gradient norm then needs no second pass over these gradients)
    int gci;            // block order: ci tiles are walked in groups of gci (see wgrad_dma_blocks)
    bf16_t* dw16;       // optional: unsplit jobs store bf16 here (same [Cout][taps][Cin] layout) INSTEAD of fp32 into dw
    const bf16_t* x2;   // dense jobs only, optional: x is the channel concatenation [x | x2] read in place - x holds the
    int C1;             //   first C1 input channels (rows of C1), x2 the other Cin - C1 (rows of Cin - C1); C1 is a multiple
    unsigned x2_bytes;  //   of the 160-channel tile, so every ci tile lies in one of the two tensors
};

constexpr int WG_SQ_SLOTS = 1024;   // partial sums of squares are spread over this many addresses (atomics to one address serialise)
constexpr int WG_BT = 160;          // channels per tile side
constexpr int WG_BK = 64;           // pixels per k tile
constexpr int WG_LD = 352;          // LDS row stride in bytes
constexpr int WG_CH = 5;            // 16-byte chunks per thread per operand

__device__ __forceinline__ void wgrad_body(const WgradArgs& p, const int block_id, const int grid_blocks, char* smem) {
    char* As = smem;                       // dy tile  [64][160]
    char* Bs = smem + WG_BK * WG_LD;       // x tile   [64][160]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * 80, wn0 = (wave & 1) * 80;

    // block order: taps fastest, then ci/co tiles, pixel split slowest - workgroups that run together (one
    // XCD-contiguous chunk) read the SAME pixel range, so the dy tile and the (tap-shifted) x rows are
    // served from that XCD's L2 instead of the fabric
    int lid = xcd_remap(block_id, grid_blocks);
    const int tap = lid % p.taps; lid /= p.taps;
    const int tile_ci = lid % p.tiles_ci; lid /= p.tiles_ci;
    const int tile_co = lid % p.tiles_co;
    const int split = lid / p.tiles_co;
    if (split >= p.splits) return;
    const int co0 = tile_co * WG_BT, ci0 = tile_ci * WG_BT;
    int ky = 0, kx = 0;
    if (p.ksize == 3) { ky = tap / 3; kx = tap - ky * 3; }
    const int pad = p.ksize >> 1;

    const int nkt_total = (p.M + WG_BK - 1) / WG_BK;
    const int kt_begin = split * p.kt_per_split;
    int kt_end = kt_begin + p.kt_per_split;
    if (kt_end > nkt_total) kt_end = nkt_total;

    // loader: thread -> one pixel row of the tile, chunks (tid&3) + 4*i
    const int lp = tid >> 2, lc = tid & 3;
    bf16x8 a_reg[WG_CH], b_reg[WG_CH];
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    auto load_tile = [&](int kt) {
        const int m = kt * WG_BK + lp;
        const bool m_ok = m < p.M;
        const bf16_t* arow = p.dy + (size_t)(m_ok ? m : 0) * p.Cout;
        const bf16_t* brow;
        bool b_ok = m_ok;
        if (p.ksize == 1 && p.stride == 1 && !p.ups) {
            brow = p.x + (size_t)(m_ok ? m : 0) * p.ldx;
        } else {
            const int mm = m_ok ? m : 0;
            const int ox = mm % p.Wo;
            const int t = mm / p.Wo;
            const int oy = t % p.Ho;
            const int b = t / p.Ho;
            int sy = oy * p.stride - pad + ky, sx = ox * p.stride - pad + kx;
            b_ok = b_ok && sy >= 0 && sx >= 0;
            if (p.ups) { sy >>= 1; sx >>= 1; }
            b_ok = b_ok && sy < p.H && sx < p.W;
            brow = p.x + (b_ok ? ((size_t)(b * p.H + sy) * p.W + sx) * p.Cin : 0);
        }
#pragma unroll
        for (int i = 0; i < WG_CH; ++i) {
            const int ch = (lc + 4 * i) * 8;
            a_reg[i] = (m_ok && co0 + ch < p.Cout) ? *(const bf16x8*)(arow + co0 + ch) : zero8;
            b_reg[i] = (b_ok && ci0 + ch < p.Cin) ? *(const bf16x8*)(brow + ci0 + ch) : zero8;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < WG_CH; ++i) {
            const int off = lp * WG_LD + (lc + 4 * i) * 16;
            *(bf16x8*)(As + off) = a_reg[i];
            *(bf16x8*)(Bs + off) = b_reg[i];
        }
    };

    f32x4 acc[5][5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias gradient rides along in the workgroups that own (ci tile 0, tap 0): they see every dy row once
    const bool do_bias = p.dbias != nullptr && tile_ci == 0 && tap == 0;
    float bsum[WG_CH][8];
#pragma unroll
    for (int i = 0; i < WG_CH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[i][e] = 0.f;

    // transposed-read addressing: 16-lane group g reads pixels kbase + 4g + q (then +16), q = (lane&15)>>2,
    // channels c0 + 4*(lane&3) .. +3; lane i of the group receives channel c0 + i for those 4 pixels.
    const int g = lane >> 4, li = lane & 15;
    const int tr_off = (4 * g + (li >> 2)) * WG_LD + (li & 3) * 8;

    if (kt_begin < kt_end) load_tile(kt_begin);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        __syncthreads();           // previous tile fully consumed
        store_tile();
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < WG_CH; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[i][e] += (float)a_reg[i][e];
        }
        __syncthreads();
        if (kt + 1 < kt_end) load_tile(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[5], bfr[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const char* pa = As + ks * 32 * WG_LD + tr_off + (wm0 + i * 16) * 2;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(pa));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(pa + 16 * WG_LD));
                af[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const char* pb = Bs + ks * 32 * WG_LD + tr_off + (wn0 + i * 16) * 2;
                const bf16x4 lo2 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(pb));
                const bf16x4 hi2 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(pb + 16 * WG_LD));
                bfr[i] = bf16x8{lo2[0], lo2[1], lo2[2], lo2[3], hi2[0], hi2[1], hi2[2], hi2[3]};
            }
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }

    if (do_bias) {
        __syncthreads();
        float* red = (float*)smem;               // [64 pixel rows][160 channels]
#pragma unroll
        for (int i = 0; i < WG_CH; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[lp * WG_BT + (lc + 4 * i) * 8 + e] = bsum[i][e];
        __syncthreads();
        if (tid < WG_BT && co0 + tid < p.Cout) {
            float v = 0.f;
            for (int r = 0; r < WG_BK; ++r) v += red[r * WG_BT + tid];
            atomicAdd(p.dbias + co0 + tid, v);
        }
    }

    const int ktot = p.taps * p.Cin;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm0 + i * 16 + g * 4 + r;
                const int ci = ci0 + wn0 + j * 16 + li;
                if (co < p.Cout && ci < p.Cin) {
                    float* dst = p.dw + (size_t)co * ktot + (size_t)tap * p.Cin + ci;
                    if (p.use_atomic) unsafeAtomicAdd(dst, acc[i][j][r]);
                    else *dst = acc[i][j][r];
                }
            }
}

__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * WG_BK * WG_LD];
    wgrad_body(p, blockIdx.x, gridDim.x, smem);
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA weight-gradient body (the default).  Tiles of 32 pixels x 160 channels per operand are written
// to LDS by `buffer_load ... lds` in a linear [pixel][176-channel] image (the 16 spare channels per row
// are fetched out of range = zeros; they give the 352-byte row stride the transposing reads need).  Three
// LDS stages of 22 KB, one barrier per k-tile, the next two tiles' DMA in flight under the MFMAs.
// Out-of-range pixels / padding taps / channels beyond the tensor use an out-of-range buffer offset.
//
// Addressing is incremental: every lane owns the same (pixel row, channel chunk) slot of each of its
// pieces for the whole contraction, and for stride-1 taps the source offset is LINEAR in the pixel
// index (pixel m reads pixel m + (ky-1)*W + (kx-1)), so a k-tile costs one add per piece plus the border
// test - the divisions by Wo/Ho of a naive im2col gather would otherwise outweigh the 25 MFMAs a wave
// issues per k-tile.  Strided and upsampled taps keep the generic gather.
// ---------------------------------------------------------------------------------------------
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
constexpr int WD_BK = 32;                       // pixels per k-tile
constexpr int WD_CHUNKS = 22;                   // 16-byte chunks per LDS row (20 real + 2 zero)
constexpr int WD_TILE = WD_BK * WG_LD;          // bytes per operand tile (11 KiB pieces)
constexpr int WD_PIECES = WD_BK * WD_CHUNKS / 64;   // 11 per operand
constexpr int OOB_OFF = (int)0x80000000;

// Transposing LDS read issued as inline asm: the compiler makes an LDS read it knows about wait for EVERY
// outstanding `buffer_load ... lds` (vmcnt(0)), which would serialise the DMA prefetch with the MFMAs.  The
// caller waits with lds_wait() before the first use of the results.
template <int OFF>
__device__ __forceinline__ bf16x4 lds_tr(unsigned addr) {
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void lds_wait(bf16x4 (&a)[5], bf16x4 (&b)[5]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4])
                 :: "memory");
}

__device__ __forceinline__ int fast_div(int n, int d, float rd) {
    int q = (int)((float)n * rd);
    int r = n - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) ++q;
    return q;
}

// MODE 0: dense (linear layer / 1x1 conv), 1: stride-1 taps on power-of-two Ho x Wo, 2: generic gather
template <int MODE, bool BIAS>
__device__ __forceinline__ void wgrad_dma_body(const WgradArgs& p, const int block_id, const int grid_blocks, char* smem) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 80, wn0 = (wave & 1) * 80;

    // block order: taps fastest, then ci/co tiles, pixel split slowest - workgroups that run together (one
    // XCD-contiguous chunk) read the SAME pixel range, so the dy tile and the (tap-shifted) x rows are
    // served from that XCD's L2 instead of the fabric
    int lid = xcd_remap(block_id, grid_blocks);
    if (lid >= p.nmain) return;              // padding block of a grouped launch
    const int tap = lid % p.taps; lid /= p.taps;
    // inside one pixel split the (co, ci) tile plane is walked ci-group by ci-group (gci tiles wide), co tiles inside a
    // group, ci tiles innermost: an XCD's contiguous chunk is then a near-square block of the plane and fetches
    // (rows + columns) operand slices instead of (one row + every column)
    const int plane = p.tiles_co * p.tiles_ci;
    const int split = lid / plane;
    const int t = lid - split * plane;
    const int gsize = p.gci * p.tiles_co;
    const int cg = t / gsize, r = t - cg * gsize;
    int gw = p.tiles_ci - cg * p.gci;
    if (gw > p.gci) gw = p.gci;
    const int tile_co = r / gw;
    const int tile_ci = cg * p.gci + (r - tile_co * gw);
    const int co0 = tile_co * WG_BT, ci0 = tile_ci * WG_BT;
    int ky = 0, kx = 0;
    if (p.ksize == 3) { ky = tap / 3; kx = tap - ky * 3; }
    const int pad = p.ksize >> 1;
    const int dyy = ky - pad, dxx = kx - pad;
    const float rWo = 1.0f / (float)p.Wo, rHo = 1.0f / (float)p.Ho;

    const int nkt_total = (p.M + WD_BK - 1) / WD_BK;
    const int kt_begin = split * p.kt_per_split;
    int kt_end = kt_begin + p.kt_per_split;
    if (kt_end > nkt_total) kt_end = nkt_total;

    // x operand of this ci tile: the tensor itself, or the half of a two-source concatenation the tile lies in
    const bf16_t* xb = p.x;
    unsigned xb_bytes = p.x_bytes;
    int ldx = p.ldx, cib = ci0, cin_lim = p.Cin;
    if (MODE == 0 && p.x2 != nullptr) {
        if (ci0 >= p.C1) { xb = p.x2; xb_bytes = p.x2_bytes; ldx = p.Cin - p.C1; cib = ci0 - p.C1; cin_lim = ldx; }
        else { ldx = p.C1; cin_lim = p.C1; }
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, xb_bytes, 0x00020000);

    // Wave w moves pieces w, w+4, w+8 (< 11) of BOTH operand tiles; a piece is 64 consecutive 16-byte slots
    // of the [32][22-chunk] tile image, so a lane owns the same (pixel row, channel chunk) of its pieces for
    // the whole contraction.  Per piece it keeps the next pixel (pm) and the byte offsets of the next dy / x
    // fetch; slots that never hold data (spare chunks, channels past the tensor) keep an out-of-range offset
    // and a zero step.
    int pm[3], offA[3], offB[3], chB[3];
    const int stepA = WD_BK * p.Cout * 2;
    const int stepB = MODE == 0 ? WD_BK * ldx * 2 : WD_BK * p.Cin * 2;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int q = (wave + 4 * j) * 64 + lane;
        const int prow = q / WD_CHUNKS;
        const int pch = q - prow * WD_CHUNKS;
        const int ch = pch * 8;
        pm[j] = kt_begin * WD_BK + prow;
        chB[j] = ch;
        // an out-of-range offset stays out of range under the per-tile steps (operands are < 2 GB)
        const bool okA = pch < 20 && co0 + ch < p.Cout;
        offA[j] = okA ? (pm[j] * p.Cout + co0 + ch) * 2 : OOB_OFF;
        const bool okB = pch < 20 && (MODE == 0 ? cib + ch < cin_lim : ci0 + ch < p.Cin);
        if (MODE == 0) offB[j] = okB ? (pm[j] * ldx + cib + ch) * 2 : OOB_OFF;
        else if (MODE == 1) offB[j] = okB ? ((pm[j] + dyy * p.W + dxx) * p.Cin + ci0 + ch) * 2 : OOB_OFF;   // used only when the tap is in range
        else offB[j] = okB ? 0 : OOB_OFF;
    }
    const int npieces = wave == 3 ? 2 : 3;      // piece 11 does not exist

    // pixels at or beyond m_end fetch nothing: rows past the tensor, and the tiles the pipeline requests past
    // this workgroup's range (every iteration issues its pieces, so the vmcnt bookkeeping is uniform)
    const int m_end = kt_end * WD_BK < p.M ? kt_end * WD_BK : p.M;
    auto issue_piece = [&](int j, int stage) {
        char* As = smem + stage * 2 * WD_TILE;
        const int idx = wave + 4 * j;
        const int m = pm[j];
        const bool in_m = m < m_end;
        const int va = (in_m && offA[j] >= 0) ? offA[j] : OOB_OFF;
        int vb;
        if (MODE == 0) {
            vb = (in_m && offB[j] >= 0) ? offB[j] : OOB_OFF;
        } else if (MODE == 1) {
            const int ox = m & (p.Wo - 1);
            const int oy = (m >> p.lw) & (p.Ho - 1);
            const bool ok = in_m & ((unsigned)(oy + dyy) < (unsigned)p.H) & ((unsigned)(ox + dxx) < (unsigned)p.W);
            vb = (ok && offB[j] >= 0) ? offB[j] : OOB_OFF;
        } else {
            const int t = fast_div(m, p.Wo, rWo);
            const int ox = m - t * p.Wo;
            const int b = fast_div(t, p.Ho, rHo);
            const int oy = t - b * p.Ho;
            int sy = oy * p.stride + dyy, sx = ox * p.stride + dxx;
            bool ok = in_m & (sy >= 0) & (sx >= 0);
            if (p.ups) { sy >>= 1; sx >>= 1; }
            ok = ok & (sy < p.H) & (sx < p.W) & (offB[j] == 0);
            vb = ok ? (((b * p.H + sy) * p.W + sx) * p.Cin + ci0 + chB[j]) * 2 : OOB_OFF;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(As + idx * 1024), 16, va, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(As + WD_TILE + idx * 1024), 16, vb, 0, 0, 0);
        pm[j] = m + WD_BK;
        offA[j] += stepA;
        if (MODE != 2) offB[j] += stepB;
    };
    auto issue_tile = [&](int stage) {
        issue_piece(0, stage);
        issue_piece(1, stage);
        if (npieces == 3) issue_piece(2, stage);
    };

    f32x4 acc[5][5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, li = lane & 15;
    const int tr_off = (4 * g + (li >> 2)) * WG_LD + (li & 3) * 8;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned a_rd = lds_base + tr_off + wm0 * 2;
    const unsigned b_rd = lds_base + WD_TILE + tr_off + wn0 * 2;

    // Software pipeline over 32-pixel tiles with three LDS stages:
    //   iteration t:  wait until tile t+1 has landed (tile t+2 may still be in flight) -> barrier ->
    //                 MFMAs on the fragments of tile t, interleaved with the transposing LDS reads of tile t+1
    //                 and with the DMA requests for tile t+3, which overwrite the stage of tile t (every wave
    //                 finished reading it before the barrier)
    // so neither the LDS reads nor the DMA issue sit between a barrier and the matrix instructions.  The dy
    // fragment of row block i is dead after its five MFMAs and is refilled in place; the x fragments are
    // used by every row block, so they alternate between two register sets.
    bf16x4 alo[5], ahi[5], blo0[5], bhi0[5], blo1[5], bhi1[5];
    auto read_a = [&](int i, unsigned a_addr) {     // i is a compile-time constant after unrolling
        switch (i) {
            case 0: alo[0] = lds_tr<0>(a_addr);   ahi[0] = lds_tr<16 * WG_LD>(a_addr); break;
            case 1: alo[1] = lds_tr<32>(a_addr);  ahi[1] = lds_tr<32 + 16 * WG_LD>(a_addr); break;
            case 2: alo[2] = lds_tr<64>(a_addr);  ahi[2] = lds_tr<64 + 16 * WG_LD>(a_addr); break;
            case 3: alo[3] = lds_tr<96>(a_addr);  ahi[3] = lds_tr<96 + 16 * WG_LD>(a_addr); break;
            default: alo[4] = lds_tr<128>(a_addr); ahi[4] = lds_tr<128 + 16 * WG_LD>(a_addr); break;
        }
    };
    auto read_b = [&](bf16x4 (&lo)[5], bf16x4 (&hi)[5], int i, unsigned b_addr) {
        switch (i) {
            case 0: lo[0] = lds_tr<0>(b_addr);   hi[0] = lds_tr<16 * WG_LD>(b_addr); break;
            case 1: lo[1] = lds_tr<32>(b_addr);  hi[1] = lds_tr<32 + 16 * WG_LD>(b_addr); break;
            case 2: lo[2] = lds_tr<64>(b_addr);  hi[2] = lds_tr<64 + 16 * WG_LD>(b_addr); break;
            case 3: lo[3] = lds_tr<96>(b_addr);  hi[3] = lds_tr<96 + 16 * WG_LD>(b_addr); break;
            default: lo[4] = lds_tr<128>(b_addr); hi[4] = lds_tr<128 + 16 * WG_LD>(b_addr); break;
        }
    };
    auto frag = [](const bf16x4& lo, const bf16x4& hi) { return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; };
    // Bias gradient (BIAS instances: the workgroups of ci tile 0, tap 0, which see every dy row once): after the
    // transposing read a lane holds 8 pixels of ONE channel, so four packed dot products against ones add them to
    // a per-lane partial column sum; the two waves that share a dy row block split its five 16-channel groups.
    float bsum[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const bf16_t one = (bf16_t)1.0f;
    const bf16x2 ones2 = {one, one};
    const bool bias_lo = (wave & 1) == 0;
    auto mfma_row = [&](const bf16x4 (&lo)[5], const bf16x4 (&hi)[5], int i) {
        const bf16x8 a = frag(alo[i], ahi[i]);
#pragma unroll
        for (int j = 0; j < 5; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, frag(lo[j], hi[j]), acc[i][j], 0, 0, 0);
        if (BIAS && ((i < 3) == bias_lo)) {
            float t = bsum[i];
            t = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{alo[i][0], alo[i][1]}, ones2, t, false);
            t = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{alo[i][2], alo[i][3]}, ones2, t, false);
            t = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{ahi[i][0], ahi[i][1]}, ones2, t, false);
            t = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{ahi[i][2], ahi[i][3]}, ones2, t, false);
            bsum[i] = t;
        }
    };
    auto wait_tile = [&]() {        // all but this wave's youngest tile request have landed
        if (npieces == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    };
    // one pipeline iteration: x fragments of the current tile in (clo, chi), the next tile's go to (nlo, nhi)
    auto iteration = [&](const bf16x4 (&clo)[5], const bf16x4 (&chi)[5], bf16x4 (&nlo)[5], bf16x4 (&nhi)[5],
                         int rd_stage, int wr_stage) {
        wait_tile();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned a_addr = a_rd + rd_stage * 2 * WD_TILE, b_addr = b_rd + rd_stage * 2 * WD_TILE;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            mfma_row(clo, chi, i);
            read_a(i, a_addr);
            read_b(nlo, nhi, i, b_addr);
            if (i < 2) issue_piece(i, wr_stage);
            else if (i == 2 && npieces == 3) issue_piece(2, wr_stage);
        }
        lds_wait(alo, ahi);
        lds_wait(nlo, nhi);
    };

    if (kt_begin >= kt_end) return;
    issue_tile(0); issue_tile(1); issue_tile(2);
    if (npieces == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 5; ++i) { read_a(i, a_rd); read_b(blo0, bhi0, i, b_rd); }
    lds_wait(alo, ahi);
    lds_wait(blo0, bhi0);

    int st = 0;                                    // LDS stage of the tile whose fragments are in registers
    int kt = kt_begin;
    for (; kt + 2 < kt_end; kt += 2) {
        const int s1 = st == 2 ? 0 : st + 1, s2 = s1 == 2 ? 0 : s1 + 1;
        iteration(blo0, bhi0, blo1, bhi1, s1, st);
        iteration(blo1, bhi1, blo0, bhi0, s2, s1);
        st = s2;
    }
    if (kt + 1 < kt_end) {
        iteration(blo0, bhi0, blo1, bhi1, st == 2 ? 0 : st + 1, st);
#pragma unroll
        for (int i = 0; i < 5; ++i) mfma_row(blo1, bhi1, i);
    } else {
#pragma unroll
        for (int i = 0; i < 5; ++i) mfma_row(blo0, bhi0, i);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no DMA write may outlive the workgroup's LDS

    if (BIAS) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if ((i < 3) != bias_lo) continue;
            float t = bsum[i];
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            const int co = co0 + wm0 + i * 16 + li;
            if (g == 0 && co < p.Cout) atomicAdd(p.dbias + co, t);
        }
    }

    const int ktot = p.taps * p.Cin;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm0 + i * 16 + g * 4 + r;
                const int ci = ci0 + wn0 + j * 16 + li;
                if (co < p.Cout && ci < p.Cin) {
                    const size_t idx = (size_t)co * ktot + (size_t)tap * p.Cin + ci;
                    if (p.use_atomic) unsafeAtomicAdd(p.dw + idx, acc[i][j][r]);
                    else {
                        if (p.dw16 != nullptr) p.dw16[idx] = f2bf(acc[i][j][r]);
                        else p.dw[idx] = acc[i][j][r];
                        sq += acc[i][j][r] * acc[i][j][r];
                    }
                }
            }
    if (p.sumsq != nullptr && !p.use_atomic) {
        sq = wave_sum(sq);
        if (lane == 0) atomicAdd(p.sumsq + ((block_id * 4 + wave) & (WG_SQ_SLOTS - 1)), sq);
    }
#endif
}

__device__ __forceinline__ void wgrad_dma_dispatch(const WgradArgs& p, const int block_id, const int grid_blocks, char* smem) {
    const bool linear = p.stride == 1 && !p.ups;
    const int mode = (p.ksize == 1 && linear) ? 0 : (linear && p.lw >= 0) ? 1 : 2;
    bool bias = false;
    if (p.dbias != nullptr) {                    // same decode as the body: the workgroups of (tap 0, ci tile 0)
        const int lid = xcd_remap(block_id, grid_blocks);
        const int t = (lid / p.taps) % (p.tiles_co * p.tiles_ci);
        const int gsize = p.gci * p.tiles_co;
        const int cg = t / gsize, r = t - cg * gsize;
        int gw = p.tiles_ci - cg * p.gci;
        if (gw > p.gci) gw = p.gci;
        bias = lid % p.taps == 0 && cg == 0 && r % gw == 0;
    }
    if (bias) {
        if (mode == 0) wgrad_dma_body<0, true>(p, block_id, grid_blocks, smem);
        else if (mode == 1) wgrad_dma_body<1, true>(p, block_id, grid_blocks, smem);
        else wgrad_dma_body<2, true>(p, block_id, grid_blocks, smem);
    } else {
        if (mode == 0) wgrad_dma_body<0, false>(p, block_id, grid_blocks, smem);
        else if (mode == 1) wgrad_dma_body<1, false>(p, block_id, grid_blocks, smem);
        else wgrad_dma_body<2, false>(p, block_id, grid_blocks, smem);
    }
}

__global__ __launch_bounds__(256, 2) void wgrad_dma_kernel(const WgradArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[6 * WD_TILE];     // [stage][A | B]
    wgrad_dma_dispatch(p, blockIdx.x, gridDim.x, smem);
}

// Many independent weight-gradient problems in ONE launch (deferred from the backward pass): job j owns
// blocks [first_block_j, first_block_{j+1}), padded to a multiple of 8 so the XCD-aware block order of
// each job is preserved.  With thousands of tiles in flight no job needs an aggressive pixel split, so
// atomic traffic drops and small layers no longer run alone on a mostly idle chip.
struct WgradJob {
    WgradArgs a;
    int first_block;
    int nblocks;
};

__global__ __launch_bounds__(256, 2) void wgrad_grouped_kernel(const WgradJob* __restrict__ jobs, int njobs) {
    __shared__ __attribute__((aligned(16))) char smem[6 * WD_TILE];
    int lo = 0, hi = njobs - 1;
    const int bid = blockIdx.x;
    while (lo < hi) {                           // wave-uniform binary search over the block prefix
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= bid) lo = mid; else hi = mid - 1;
    }
    const WgradJob& j = jobs[lo];
    WgradArgs p = j.a;
    wgrad_dma_dispatch(p, bid - j.first_block, j.nblocks, smem);
}

// Column sums of a bf16 matrix in segments of `seg` rows: out[s][n] = sum_{m in segment s} y[m][n].
// Used for bias gradients (one segment) and the timestep-embedding broadcast gradient (one segment
// per sample).  Two stages without atomics: each block reduces a strip of rows to one partial row
// (row phases combined through LDS), a second kernel sums the partial rows of every segment.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16_t* __restrict__ y, float* __restrict__ partial,
                                                             int M, int N, int seg, int rows_per_block,
                                                             int chunks_per_seg) {
    extern __shared__ float cred[];      // [rphase][N]
    const int nc = N / 8;
    const int s = blockIdx.x / chunks_per_seg, ck = blockIdx.x - s * chunks_per_seg;
    const int r_begin = s * seg + ck * rows_per_block;
    int r_end = r_begin + rows_per_block;
    if (r_end > (s + 1) * seg) r_end = (s + 1) * seg;
    if (r_end > M) r_end = M;
    const int cpb = nc < 256 ? nc : 256;
    const int rphase = 256 / cpb;
    const int rp = threadIdx.x / cpb;
    for (int c0 = 0; c0 < nc; c0 += cpb) {
        const int c = c0 + (threadIdx.x % cpb);
        if (c < nc && rp < rphase) {
            float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int r = r_begin + rp; r < r_end; r += rphase) {
                const bf16x8 t = *(const bf16x8*)(y + (size_t)r * N + c * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)t[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) cred[(size_t)rp * N + c * 8 + e] = v[e];
        }
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += 256) {
        float a = 0.f;
        for (int q = 0; q < rphase; ++q) a += cred[(size_t)q * N + n];
        partial[(size_t)blockIdx.x * N + n] = a;
    }
}
static void colsum_geometry(int M, int N, int seg, int* rows_per_block, int* chunks) {
    const int nseg = M / seg;
    int target = 256 / nseg;            // about one block per CU in total
    if (target < 1) target = 1;
    int rows = (seg + target - 1) / target;
    if (rows < 16) rows = 16;
    if (rows > seg) rows = seg;
    *rows_per_block = rows;
    *chunks = (seg + rows - 1) / rows;
}

extern "C" {

static int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

// Geometry shared by the single and grouped launches; bk = pixels per k-tile of the kernel that will run.
static void wgrad_geometry(WgradArgs& a, int B, int H, int W, int Cin, int Cout, int ksize, int stride, int ups, int Ho, int Wo) {
    a.M = B * Ho * Wo; a.Cout = Cout; a.Cin = Cin; a.B = B; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
    a.stride = stride; a.ups = ups; a.ksize = ksize; a.ldx = Cin;
    a.taps = ksize * ksize;
    a.tiles_co = (Cout + WG_BT - 1) / WG_BT;
    a.tiles_ci = (Cin + WG_BT - 1) / WG_BT;
    a.lw = ilog2_exact(Wo); a.lh = ilog2_exact(Ho);
    if (a.lw < 0 || a.lh < 0) a.lw = a.lh = -1;
}
static bool wgrad_dma_extents(WgradArgs& a) {
    const bool dense = a.ksize == 1 && a.stride == 1 && !a.ups;
    const double dy_sz = (double)a.M * a.Cout * 2.0;
    const double x_sz = dense ? (double)a.M * a.Cin * 2.0 : (double)a.B * a.H * a.W * a.Cin * 2.0;
    if (dy_sz >= 2.0e9 || x_sz >= 2.0e9) return false;
    a.dy_bytes = (unsigned)dy_sz; a.x_bytes = (unsigned)x_sz;
    return true;
}
static int wgrad_dma_blocks(WgradArgs& a) {
    a.nmain = a.tiles_co * a.tiles_ci * a.taps * a.splits;
    // An XCD runs a contiguous eighth of the block order: `per_xcd` tiles of the (co, ci) plane (all taps of a tile are
    // adjacent).  When that is less than a plane, a block gci = sqrt(per_xcd) tiles wide reads the fewest operand slices
    // through that XCD's L2; otherwise the XCD owns whole planes and reads each operand once anyway.
    static const int force = getenv("MI355_WGRAD_GCI") ? atoi(getenv("MI355_WGRAD_GCI")) : -1;
    const int plane = a.tiles_co * a.tiles_ci;
    const int per_xcd = (plane * a.splits + 7) / 8;
    int g = a.tiles_ci;
    if (force > 0) g = force;
    else if (force < 0 && per_xcd < plane) {
        g = 1;
        while ((g + 1) * (g + 1) <= per_xcd) ++g;       // floor(sqrt(per_xcd))
        if (g * (g + 1) <= per_xcd) ++g;                // round to nearest
    }
    a.gci = g < 1 ? 1 : (g > a.tiles_ci ? a.tiles_ci : g);
    return a.nmain;
}

int mi355_wgrad_job_bytes(void) { return (int)sizeof(WgradJob); }
int mi355_wgrad_sumsq_slots(void) { return WG_SQ_SLOTS; }

// Fills one job record in HOST memory (mi355_wgrad_job_bytes() bytes).  dw (and dbias) must have been
// zeroed by the caller when the job is split (the record says so through *splits_out).  Returns the
// number of blocks the job occupies in the grouped launch (a multiple of 8), or a negative error.
int mi355_wgrad_job_fill(void* host_job, const void* dy, const void* x, float* dw, float* dbias, int B, int H, int W,
                         int Cin, int Cout, int ksize, int stride, int ups, int Ho, int Wo, int first_block,
                         int* splits_out, float* sumsq_slots, void* dw_bf16) {
    if (!host_job || !dy || !x || !dw) return mi355_fail(MI355_ERR_ARG, "wgrad_job_fill: null pointer");
    if (ksize != 1 && ksize != 3) return mi355_fail(MI355_ERR_UNSUPPORTED, "wgrad_job_fill: ksize %d", ksize);
    if ((Cin & 7) || (Cout & 7)) return mi355_fail(MI355_ERR_ARG, "wgrad_job_fill: channels must be multiples of 8");
    WgradJob* j = (WgradJob*)host_job;
    WgradArgs& a = j->a;
    a = WgradArgs{};
    a.dy = (const bf16_t*)dy; a.x = (const bf16_t*)x; a.dw = dw; a.dbias = dbias;
    wgrad_geometry(a, B, H, W, Cin, Cout, ksize, stride, ups, Ho, Wo);
    if (!wgrad_dma_extents(a)) return mi355_fail(MI355_ERR_UNSUPPORTED, "wgrad_job_fill: operand of 2 GB or more");
    {   // pixel split: about 8192 pixels (256 k-tiles of 32) per workgroup; measured best of 64..512
        const int nkt = (a.M + WD_BK - 1) / WD_BK;
        static const int target = getenv("MI355_WGRAD_KT") ? atoi(getenv("MI355_WGRAD_KT")) : 256;
        int sp = (nkt + target - 1) / target;
        if (sp < 1) sp = 1;
        a.kt_per_split = (nkt + sp - 1) / sp;
        a.splits = (nkt + a.kt_per_split - 1) / a.kt_per_split;
    }
    a.use_atomic = a.splits > 1 ? 1 : 0;
    a.sumsq = a.use_atomic ? nullptr : sumsq_slots;
    a.dw16 = a.use_atomic ? nullptr : (bf16_t*)dw_bf16;
    const int blocks = wgrad_dma_blocks(a);
    j->first_block = first_block;
    j->nblocks = (blocks + 7) & ~7;
    if (splits_out) *splits_out = a.splits;
    return j->nblocks;
}

// The dense job whose x operand is the channel concatenation [x1 | x2] (x1: [M][C1], x2: [M][C2]) read in place:
// dw is [Cout][C1 + C2] as if the concatenation had been built.  C1 must be a multiple of 160 (the channel tile).
int mi355_wgrad_job_fill_cat(void* host_job, const void* dy, const void* x1, int C1, const void* x2, int C2, float* dw,
                             float* dbias, int M, int Cout, int first_block, int* splits_out, float* sumsq_slots, void* dw_bf16) {
    if (!x1 || !x2) return mi355_fail(MI355_ERR_ARG, "wgrad_job_fill_cat: null pointer");
    if (C1 < WG_BT || C1 % WG_BT || C2 < 8 || (C2 & 7))
        return mi355_fail(MI355_ERR_UNSUPPORTED, "wgrad_job_fill_cat: C1 = %d must be a multiple of %d, C2 = %d of 8", C1, WG_BT, C2);
    const int nb = mi355_wgrad_job_fill(host_job, dy, x1, dw, dbias, M, 1, 1, C1 + C2, Cout, 1, 1, 0, 1, 1, first_block, splits_out,
                                        sumsq_slots, dw_bf16);
    if (nb < 0) return nb;
    WgradArgs& a = ((WgradJob*)host_job)->a;
    const double x1_sz = (double)M * C1 * 2.0, x2_sz = (double)M * C2 * 2.0;
    a.x2 = (const bf16_t*)x2; a.C1 = C1;
    a.x_bytes = (unsigned)x1_sz; a.x2_bytes = (unsigned)x2_sz;        // (both below the 2 GB the concatenation was checked against)
    return nb;
}

// Reorders njobs records in HOST memory so that the jobs whose workgroups walk the most k-tiles come first (ties keep
// their order) and renumbers their blocks: the launch then drains on its shortest workgroups instead of on whichever
// layer was queued last.  Returns the total block count (unchanged).
int mi355_wgrad_jobs_sort(void* host_jobs, int njobs) {
    if (!host_jobs || njobs <= 0) return mi355_fail(MI355_ERR_ARG, "wgrad_jobs_sort: bad arguments");
    WgradJob* j = (WgradJob*)host_jobs;
    std::stable_sort(j, j + njobs, [](const WgradJob& x, const WgradJob& y) { return x.a.kt_per_split > y.a.kt_per_split; });
    int first = 0;
    for (int i = 0; i < njobs; ++i) { j[i].first_block = first; first += j[i].nblocks; }
    return first;
}

// jobs: DEVICE copy of njobs records; total_blocks = sum of their block counts.
int mi355_wgrad_grouped(const void* jobs, int njobs, int total_blocks, double flops, void* stream) {
    if (!jobs || njobs <= 0 || total_blocks <= 0) return mi355_fail(MI355_ERR_ARG, "wgrad_grouped: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(FAM_WGRAD, st, flops);
    hipLaunchKernelGGL(wgrad_grouped_kernel, dim3(total_blocks), dim3(256), 0, st, (const WgradJob*)jobs, njobs);
    return mi355_check_launch("wgrad_grouped_kernel");
}

int mi355_wgrad_bf16(const void* dy, const void* x, float* dw, float* dbias, int B, int H, int W, int Cin, int Cout,
                     int ksize, int stride, int ups, int Ho, int Wo, int accumulate, void* stream) {
    if (!dy || !x || !dw) return mi355_fail(MI355_ERR_ARG, "wgrad: null pointer");
    if (ksize != 1 && ksize != 3) return mi355_fail(MI355_ERR_UNSUPPORTED, "wgrad: ksize %d", ksize);
    if ((Cin & 7) || (Cout & 7)) return mi355_fail(MI355_ERR_ARG, "wgrad: channels must be multiples of 8");
    WgradArgs a = {};
    a.dy = (const bf16_t*)dy; a.x = (const bf16_t*)x; a.dw = dw; a.dbias = dbias;
    wgrad_geometry(a, B, H, W, Cin, Cout, ksize, stride, ups, Ho, Wo);
    const int tiles = a.tiles_co * a.tiles_ci * a.taps;
    // The LDS-DMA kernel is the default; operands of 2 GB or more (beyond a buffer descriptor's reach) and
    // MI355_WGRAD_V1=1 take the register-staged kernel.
    static const bool force_v1 = getenv("MI355_WGRAD_V1") != nullptr;
    const bool dma = wgrad_dma_extents(a) && !force_v1;
    const int bk = dma ? WD_BK : WG_BK;
    const int nkt = (a.M + bk - 1) / bk;
    // split choice: waves of 512 co-resident workgroups, ~1 us per 64-pixel k-tile, fp32 atomics at ~1.3 TB/s
    int s = 1;
    {
        const double out_bytes = (double)Cout * a.taps * Cin * 4.0;
        double best_t = 1e30;
        const int smax = nkt / 2 < 64 ? (nkt / 2 < 1 ? 1 : nkt / 2) : 64;
        for (int c = 1; c <= smax; ++c) {
            const int kps = (nkt + c - 1) / c;
            if ((nkt + kps - 1) / kps != c) continue;
            const double waves = (double)((tiles * c + 511) / 512);
            double t = waves * kps * (dma ? 0.5 : 1.0);
            if (c > 1) { const double ta = c * out_bytes / 1.3e6; t = (t > ta ? t : ta) + 0.25 * (t < ta ? t : ta); }
            if (t < best_t) { best_t = t; s = c; }
        }
    }
    a.kt_per_split = (nkt + s - 1) / s;
    a.splits = (nkt + a.kt_per_split - 1) / a.kt_per_split;
    // accumulate: 0 = overwrite (zero first when split), 1 = add to existing values,
    //             2 = dw and dbias were zeroed by the caller (plain stores when unsplit, no memsets here)
    a.use_atomic = (a.splits > 1 || accumulate == 1) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    if (a.splits > 1 && accumulate == 0) mi355_zero_async(dw, (size_t)Cout * a.taps * Cin * sizeof(float), st);
    if (dbias && accumulate == 0) mi355_zero_async(dbias, (size_t)Cout * sizeof(float), st);
    ProfScope prof(FAM_WGRAD, st, 2.0 * a.M * Cout * (double)a.taps * Cin);
    if (dma) {
        const int blocks = wgrad_dma_blocks(a);      // also completes `a` (block count, block order)
        hipLaunchKernelGGL(wgrad_dma_kernel, dim3(blocks), dim3(256), 0, st, a);
    }
    else
        hipLaunchKernelGGL(wgrad_kernel, dim3(tiles * a.splits), dim3(256), 0, st, a);
    return mi355_check_launch("wgrad_kernel");
}

long long mi355_colsum_scratch_floats(int M, int N, int seg) {
    if (seg <= 0 || M % seg) return -1;
    int rows, chunks;
    colsum_geometry(M, N, seg, &rows, &chunks);
    return (long long)(M / seg) * chunks * N;
}

int mi355_colsum_bf16(const void* y, float* out, float* scratch, int M, int N, int seg, void* stream) {
    if (!y || !out || !scratch) return mi355_fail(MI355_ERR_ARG, "colsum: null pointer");
    if ((N & 7) || seg <= 0 || M % seg) return mi355_fail(MI355_ERR_ARG, "colsum: N %% 8 == 0 and M %% seg == 0 required");
    const int nc = N / 8, cpb = nc < 256 ? nc : 256, rphase = 256 / cpb;
    if ((size_t)rphase * N * sizeof(float) > 64 * 1024) return mi355_fail(MI355_ERR_UNSUPPORTED, "colsum: N=%d too large", N);
    hipStream_t st = (hipStream_t)stream;
    const int nseg = M / seg;
    int rows, chunks;
    colsum_geometry(M, N, seg, &rows, &chunks);
    ProfScope prof(FAM_ELEM, st, (double)M * N * 2);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nseg * chunks), dim3(256), (size_t)rphase * N * sizeof(float), st,
                       (const bf16_t*)y, scratch, M, N, seg, rows, chunks);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((N + 63) / 64, nseg), dim3(256), 0, st, scratch, out, chunks, N);
    return mi355_check_launch("colsum");
}

}  // extern "C"
