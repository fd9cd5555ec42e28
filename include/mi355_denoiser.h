/* C ABI of libmi355_denoiser.so - hand-written gfx950 (MI355X) kernels for the text-conditioned
 * denoiser train/sampling step.
 *
 * The reference (hyomin777/image-generator) has no native layer and no FFI: every operation below
 * replaces arithmetic that the reference reaches through third-party Python calls.  Each entry point
 * names the reference call site whose arithmetic it carries:
 *   - generator/image_generator.py:124  `self.unet(scaled_latent, t, encoder_hidden_states=...)` (train)
 *   - generator/image_generator.py:93-95 the same network inside the sampling loop
 *   - generator/image_generator.py:125  `F.mse_loss(noise_pred, noise)`
 *   - generator/image_generator.py:116  `scheduler.add_noise`, :97-99 guidance + `infer_scheduler.step`
 *   - lora.py:23-24                     `original(x) + alpha * lora_B(lora_A(x))`
 *   - loss.py:16-28                     `ContrastiveLoss.forward`
 *   - trainer.py:257-259                grad-norm clip + AdamW step (`scaler.step(optimizer)`)
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch owns all buffers); the library
 *     allocates no persistent device memory and frees nothing;
 *   - activations are bf16, token-major / NHWC: a feature map is [B][H][W][C] == [B*H*W][C];
 *   - `stream` is a hipStream_t; all work is enqueued asynchronously on it, no call synchronises;
 *   - return value 0 = ok, negative = error (MI355_ERR_*); mi355_last_error() gives the message
 *     (thread local).  No C++ exception crosses this boundary;
 *   - "workspace"/"scratch" arguments are caller-provided device buffers of the documented size;
 *   - threads: the compute entry points keep no mutable state between calls (one-time kernel attributes are set under
 *     std::call_once) and may be called concurrently from several host threads, e.g. autograd's backward threads, each
 *     with its own stream and buffers.  Exceptions, each documented at its declaration: mi355_groupnorm_fwd_static (one
 *     scratch array per GPU), the mi355_prof_* measurement calls (one process-wide record list, mutex-guarded),
 *     mi355_tune_set (process-wide switches for A/B measurements; not for use while other threads launch).
 */
#ifndef MI355_DENOISER_H
#define MI355_DENOISER_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_OK 0
#define MI355_ERR_ARG (-1)
#define MI355_ERR_LAUNCH (-2)
#define MI355_ERR_UNSUPPORTED (-3)

const char* mi355_last_error(void);
int mi355_abi_version(void);

/* Launch timing (HIP events on the launch stream around every call of a kernel family).
 * Families: 0 gemm/conv, 1 weight-gradient, 2 attention, 3 normalisation, 4 elementwise. */
int mi355_prof_enable(int on);
int mi355_prof_collect(double* ms, double* work, long long* count);
/* What one timed call carries besides its kernels, in microseconds: the stream's processing of the closing event.
 * Measured as (n launches with an event pair around each - the same n launches bare) / 2n on `stream`; each launch
 * clears `scratch_floats` device floats at `scratch` (size it so that one launch runs about as long as the timed
 * kernels: part of an event's cost hides behind a running kernel).  Synchronises the stream; negative on failure.
 * No counterpart in the reference (measurement aid). */
double mi355_prof_event_overhead_us(void* stream, float* scratch, long long scratch_floats, int n);

/* What the GPU sustains on the two resources the kernels are priced against (measurement aids, no counterpart in the
 * reference; nothing in the product calls them).
 * peak_mfma: `blocks` workgroups of four waves each issue `iters` rounds of eight independent v_mfma_f32_16x16x32_bf16
 * on registers only (iters a multiple of 8): flops = blocks * 4 * iters * 8 * 16384.  pattern 0: zero operands, 1: mixed-sign changing values.
 * peak_copy: 16-byte-per-lane streaming copy, traffic = 2 * bytes (bytes and both pointers multiples of 16). */
int mi355_peak_mfma_bf16(float* sink, int blocks, int iters, int pattern, void* stream);
/* Run-time switch between kernel variants of the contraction launchers, for A/B timing inside one process (measurement
 * aid, no counterpart in the reference; the defaults are what the product runs).  key 0..15, see csrc/gemm.hip. */
int mi355_tune_set(int key, int value);
int mi355_peak_copy(const void* src, void* dst, long long bytes, void* stream);

/* ---- contractions (image_generator.py:124: Linear / Conv2d layers of the denoiser) -------------
 * out[M][N] = alpha * A[M][K] . Bw[N][K]^T + bias[n] + rowbias[m / rowbias_div][n]
 *             + lora_alpha * (lora_t[m][:] . lora_b[n][:]) + residual[m][n]
 * A, Bw, residual bf16; bias, rowbias, lora_t ([M][lora_ldt], first r columns), lora_b fp32; out bf16 or fp32.
 * lora_b is [N][r], or [r][N] when lora_rmajor (the data-gradient GEMM passes lora_A as stored).  lora_seg > 0:
 * the N outputs are row-stacked adapters of lora_seg outputs each and adapter s uses lora_t[m][s*r .. s*r + r).
 * lora_t_bf16: lora_t is bf16 [M][lora_ldt] with lora_ldt 8 or 16 (the output of a skinny GEMM x.A^T); lora_ldt <= 16.
 * Optional pointers may be NULL.  N, K and all leading dimensions must be multiples of 8. */
size_t mi355_gemm_workspace_bytes(int M, int N);
int mi355_gemm_bf16(const void* A, int lda, const void* Bw, int ldb, int M, int N, int K, void* out, int ldo,
                    int out_fp32, const float* bias, const float* rowbias, int rowbias_div, int ld_rowbias,
                    const void* residual, int ldr, float alpha, const float* lora_t, int lora_ldt, const float* lora_b, int lora_r,
                    int lora_seg, int lora_rmajor, int lora_t_bf16, float lora_alpha, void* workspace, size_t workspace_bytes,
                    void* stream);
/* The same product on a column-concatenated input read in place, and with a column-split output - the 1x1 shortcut
 * convolution of the up-path residual blocks, whose input is hidden state | skip connection (the
 * `torch.cat([hidden_states, res_hidden_states], dim=1)` of the third-party up block the reference calls at
 * image_generator.py:93-95), forward and data gradient, without building the concatenation:
 *   mi355_gemm_cat_bf16:   out [M][N] = [A1 | A2] Bw^T + bias; A1 [M][K1], A2 [M][K2] contiguous, Bw [N][K1 + K2] (row
 *                          stride ldb); K1, K2 multiples of 64, N of 160.
 *   mi355_gemm_split_bf16: [out1 | out2] = A Bw^T as two tensors, out1 [M][N1], out2 [M][N2]; Bw [N1 + N2][K]; N1 a
 *                          multiple of 320 and N1 + N2 of 160 (a tile goes to one destination: none may straddle N1);
 *                          other widths return MI355_ERR_UNSUPPORTED. */
/* Data gradient of the feed-forward output projection with the GEGLU backward in its epilogue: the product
 * dy [M][F] = A Bw^T is not written; dpre [M][2F] = (dy * gelu(g) | dy * h * gelu'(g)) is, from pre [M][2F] = (h | g), the
 * pre-activation mi355_gemm_geglu_bf16 kept.  A [M][K] (row stride lda), Bw [F][K] (row stride ldb); F, K multiples of 8. */
int mi355_gemm_geglu_bwd_bf16(const void* A, int lda, const void* Bw, int ldb, int M, int F, int K, const void* pre,
                              void* dpre, void* workspace, size_t workspace_bytes, void* stream);
int mi355_gemm_cat_bf16(const void* A1, int K1, const void* A2, int K2, const void* Bw, int ldb, int M, int N, void* out,
                        int ldo, const float* bias, void* workspace, size_t workspace_bytes, void* stream);
int mi355_gemm_split_bf16(const void* A, int lda, const void* Bw, int ldb, int M, int N1, int N2, int K, void* out1,
                          void* out2, void* workspace, size_t workspace_bytes, void* stream);
/* Data gradients from the weights as they are stored (what autograd derives for the linear and convolution layers of the
 * denoiser the reference trains, trainer.py:233-262): no transposed copy of a weight is kept or rebuilt after an
 * optimizer step.
 *   mi355_gemm_bt_bf16:          out [M][N] = A [M][K] W, W [K][N] with row stride ldw (the [out][in] weight of a linear
 *                                layer, A its output gradient).  Optional: out2 / N1 - columns n >= N1 go to out2
 *                                [M][N - N1] (N1 a multiple of 160); pre / dpre - the GEGLU backward of
 *                                mi355_gemm_geglu_bwd_bf16 in the epilogue (out may then be null).  N % 160 == 0, K % 64 == 0.
 *   mi355_conv2d_dgrad_bt_bf16:  dx [B][Ho][Wo][Cin] from dy [B][H][W][Cout] and w [Cout][k][k][Cin] of a 3x3 / 1x1
 *                                layer with padding k / 2; dil = 1 for a stride-2 layer (dy read through a zero-dilation
 *                                by 2, Ho in {2H - 1, 2H}).  Cin % 160 == 0, Cout % 64 == 0. */
int mi355_gemm_bt_bf16(const void* A, int lda, const void* W, int ldw, int M, int N, int K, void* out, int ldo,
                       void* out2, int N1, const void* pre, void* dpre, void* workspace, size_t workspace_bytes, void* stream);
int mi355_conv2d_dgrad_bt_bf16(const void* dy, int B, int H, int W, int Cout, const void* w, int Cin, int ksize, int dil,
                               int Ho, int Wo, void* dx, void* workspace, size_t workspace_bytes, void* stream);

/* GroupNorm statistics out of the producing epilogue (the residual blocks' and transformer wrappers' GroupNorms of the
 * third-party denoiser called at image_generator.py:93-95 / :124 read what a convolution or a 1x1 projection has just
 * written): the convolution / product below also emits, per row tile and run of `unit` (8 or 10) adjacent output
 * channels, the sum and the sum of squares of the bf16 values it stores - gstat [M / rows][N / unit][2], rows = *rows_out
 * (128 or 256 output rows, all of one sample) - and mi355_groupnorm_fwd_pre normalises from those sums without a
 * statistics pass over its input.  *rows_out = 0: this launch could not emit them (split contraction, a row tile that
 * would span two samples, channel counts outside the 160-column tiles): it ran as the plain product, gstat is untouched,
 * the caller takes mi355_groupnorm_fwd.  gstat must hold (M / 128) * (N / unit) * 2 floats.
 *   mi355_conv2d_gnstats_bf16: as mi355_conv2d_bf16 (bf16 output, alpha 1, no dilation).
 *   mi355_gemm_gnstats_bf16:   out = A Bw^T + bias + residual; rows_per_sample = rows of A per sample (H * W).
 *   mi355_groupnorm_fwd_pre:   x2 / C1 as in mi355_groupnorm_fwd_cat (x2 NULL: one source, C1 ignored); pre1 / tiles1 the
 *                              first (only) source's sums and row tiles per sample, pre2 / tiles2 the second's; every
 *                              group must be whole runs ((C / G) % unit == 0, C1 % unit == 0). */
int mi355_conv2d_gnstats_bf16(const void* x, int B, int H, int W, int C, const void* w, int Cout, int ksize, int stride,
                              int ups, int Ho, int Wo, void* out, const float* bias, const float* rowbias, int ld_rowbias,
                              const void* residual, float* gstat, int unit, int* rows_out, void* workspace,
                              size_t workspace_bytes, void* stream);
int mi355_gemm_gnstats_bf16(const void* A, int lda, const void* Bw, int ldb, int M, int N, int K, void* out, int ldo,
                            const float* bias, const void* residual, int ldr, int rows_per_sample, float* gstat, int unit,
                            int* rows_out, void* workspace, size_t workspace_bytes, void* stream);
int mi355_groupnorm_fwd_pre(const void* x, const void* x2, int C1, const float* gamma, const float* beta, void* y, float* stats,
                            const float* pre1, int tiles1, const float* pre2, int tiles2, int unit, int B, int HW, int C, int G,
                            float eps, int silu, void* stream);
/* 1 when mi355_groupnorm_fwd of this geometry takes two passes (statistics, then normalisation) and
 * mi355_groupnorm_fwd_pre saves the first; 0 for the small maps that one workgroup per (group, sample) holds in registers. */
int mi355_groupnorm_fwd_two_pass(int HW, int C, int G);

/* LayerNorm folded into the two products around it, for the no-gradient denoiser call of the sampling loop
 * (image_generator.py:92-95: the transformer blocks' `norm1/2/3` followed by `attn1.to_q/k/v`, `attn2.to_q`, `ff.net.0.proj`
 * in the third-party block): the normalised activations are never written or re-read.
 *   mi355_gemm_rowstats_bf16: out [M][N] = A Bw^T + bias + residual (as mi355_gemm_bf16) and, from the same epilogue,
 *       rowstat [M][parts][2] = per row and column tile the sum and the sum of squares of the bf16 values stored;
 *       *parts_out (host int) = the number of column tiles taken (N / 160 or N / 320); rowstat holds M * (N / 160) * 2
 *       floats.  N % 160 == 0, K % 64 == 0.
 *   mi355_ln_fold_weight: from a layer's fp32 weight W [N][K] (row stride ldw), its bias (or NULL) and the norm's gamma /
 *       beta [K]: Wp = bf16(scale gamma (.) W) [N][K], colsum[n] = sum_k Wp[n][k], bias_out[n] = scale (sum_k beta[k]
 *       W[n][k] + bias[n]); scale: a factor on the layer's output folded in as well (1, or the softmax scale times log2(e)
 *       for a query projection whose attention runs through mi355_attention_fwd_prescaled).
 *   mi355_gemm_ln_bf16: out = LayerNorm(A) W^T + b computed as rstd * (A Wp^T - mean * colsum) + bias_out, with mean and
 *       rstd of each row of A [M][K] from ln_part [M][ln_nparts][2] (the partial sums above; K values per row, eps as
 *       the norm's).  geglu != 0: N = 2F weight rows (hidden | gate), out [M][F] = h * gelu(g) as mi355_gemm_geglu_bf16. */
int mi355_gemm_rowstats_bf16(const void* A, int lda, const void* Bw, int ldb, int M, int N, int K, void* out, int ldo,
                             const float* bias, const void* residual, int ldr, float* rowstat, int* parts_out,
                             void* workspace, size_t workspace_bytes, void* stream);
int mi355_ln_fold_weight(const float* W, int ldw, const float* gamma, const float* beta, const float* bias, void* Wp,
                         float* colsum, float* bias_out, int N, int K, float scale, void* stream);
int mi355_gemm_ln_bf16(const void* A, int lda, const void* Bw, int ldb, int M, int N, int K, void* out, int ldo,
                       const float* bias, const float* ln_part, int ln_nparts, const float* colsum, float eps, int geglu,
                       void* workspace, size_t workspace_bytes, void* stream);

/* 3x3 stride-2 convolution with one pixel of zero padding on the bottom and right only (the AutoencoderKL
 * down-sampler reached through image_generator.py:107-108 `self.vae.encode`): x [B][H][W][C] -> out [B][H/2][W/2][Cout]. */
int mi355_conv2d_down_asym_bf16(const void* x, int B, int H, int W, int C, const void* w, int Cout, void* out,
                                const float* bias, void* workspace, size_t workspace_bytes, void* stream);

/* out[M][W] (bf16) = alpha * A[M][K] . Bw[W][K]^T for W = 8 or 16 and K a multiple of 32: the rank-padded products
 * x A^T and dy B of the LoRA adapters (lora.py:23-24) in one launch each, no workspace. */
int mi355_gemm_skinny_bf16(const void* A, int lda, const void* Bw, int M, int W, int K, void* out, float alpha, void* stream);

/* GEGLU feed-forward input projection with the gate in the epilogue (the transformer block's `ff.net[0]`, reached
 * through image_generator.py:93-95 and :124): Bw [2F][K] = hidden rows then gate rows, bias [2F] or NULL,
 * out[M][F] = (A.Bw_h^T + b_h) * gelu_erf(A.Bw_g^T + b_g).  F a multiple of 80, K of 64.
 * preact (optional, bf16 [M][2F]): receives the pre-activation (hidden | gate) a backward pass needs; the gate is
 * then applied to those rounded values. */
int mi355_gemm_geglu_bf16(const void* A, int lda, const void* Bw, int ldb, int M, int F, int K, void* out, int ldo,
                          const float* bias, void* preact, void* stream);

/* Implicit-GEMM convolution, NHWC bf16.  x [B][H][W][C]; w [Cout][k][k][C]; out [B][Ho][Wo][Cout].
 * ksize 1|3 (padding k/2), stride 1|2; ups=1 reads x through a nearest 2x upsample; dil=1 reads x
 * through a zero-dilation by 2 (data gradient of a stride-2 convolution, with w = flipped/transposed
 * weights).  rowbias is [B][ld_rowbias] (broadcast over pixels: timestep-embedding projection). */
int mi355_conv2d_bf16(const void* x, int B, int H, int W, int C, const void* w, int Cout, int ksize, int stride,
                      int ups, int dil, int Ho, int Wo, void* out, int out_fp32, const float* bias,
                      const float* rowbias, int ld_rowbias, const void* residual, float alpha, void* workspace,
                      size_t workspace_bytes, void* stream);

/* Weight gradient dw[Cout][k*k][Cin] (fp32) = sum over output pixels of dy[m][co] * im2col(x)[m][tap][ci].
 * Linear layers: ksize=1, H=W=Ho=Wo=1, B = rows.  accumulate: 0 overwrites dw/dbias, 1 adds to them,
 * 2 = the caller zeroed dw/dbias beforehand (no memset nodes are issued; unsplit tiles use plain stores).
 * dbias (optional, [Cout] fp32) receives the column sums of dy (bias gradient) from the same pass. */
int mi355_wgrad_bf16(const void* dy, const void* x, float* dw, float* dbias, int B, int H, int W, int Cin, int Cout,
                     int ksize, int stride, int ups, int Ho, int Wo, int accumulate, void* stream);

/* Deferred, batched weight gradients: fill job records on the host, copy them to the device, launch once.
 * mi355_wgrad_job_fill returns the job's block count (multiple of 8); first_block = running sum of them.
 * dw/dbias of every job must be zero before the launch (split jobs and bias sums accumulate atomically).
 * sumsq_slots (optional, mi355_wgrad_sumsq_slots() device floats, zeroed by the caller): a job that is NOT pixel-split (*splits_out == 1)
 * adds the sum of squares of the dw it stores to these slots, so the global gradient norm of the optimizer
 * (trainer.py:257) needs no second pass over that gradient; split jobs ignore it.
 * dw_bf16 (optional, same [Cout][k*k][Cin] layout in bf16): a job that is not pixel-split stores its result there,
 * rounded to nearest even, INSTEAD of fp32 into dw (the staging buffer of a bf16 gradient all-reduce); split jobs
 * ignore it and accumulate into dw. */
int mi355_wgrad_job_bytes(void);
int mi355_wgrad_sumsq_slots(void);
int mi355_wgrad_job_fill(void* host_job, const void* dy, const void* x, float* dw, float* dbias, int B, int H, int W,
                         int Cin, int Cout, int ksize, int stride, int ups, int Ho, int Wo, int first_block,
                         int* splits_out, float* sumsq_slots, void* dw_bf16);
/* The job of a dense layer whose input is the channel concatenation [x1 | x2] (x1 [M][C1], x2 [M][C2]) read in place:
 * dw [Cout][C1 + C2] as if the concatenation had been built (the 1x1 shortcut of the up-path residual blocks, whose
 * input is hidden state | skip connection - image_generator.py:93-95 through the third-party up block).  C1 must be a
 * multiple of 160, C2 of 8.  Other arguments and the return value as for mi355_wgrad_job_fill. */
int mi355_wgrad_job_fill_cat(void* host_job, const void* dy, const void* x1, int C1, const void* x2, int C2, float* dw,
                             float* dbias, int M, int Cout, int first_block, int* splits_out, float* sumsq_slots, void* dw_bf16);
/* Optional, before the records are copied to the device: puts the jobs whose workgroups run longest (most pixels per
 * workgroup) first and renumbers first_block, so that the grouped launch ends on its shortest workgroups.  Returns the
 * total block count.  No counterpart in the reference (its weight gradients are autograd's, trainer.py:247). */
int mi355_wgrad_jobs_sort(void* host_jobs, int njobs);
int mi355_wgrad_grouped(const void* jobs, int njobs, int total_blocks, double flops, void* stream);

/* out[s][n] = sum of the rows of segment s (seg rows each) of y[M][N] (bias / broadcast gradients). */
long long mi355_colsum_scratch_floats(int M, int N, int seg);
int mi355_colsum_bf16(const void* y, float* out, float* scratch, int M, int N, int seg, void* stream);

/* ---- attention (self- and cross-attention of the transformer blocks) ---------------------------
 * q [B][Nq][H*D] (row stride ldq), k/v [B][Nk][H*D] (ldk, ldv), out [B][Nq][H*D] (ldo); no mask
 * (image_generator.py:124 passes none).  lse [B][H][Nq] receives log2-domain row statistics. */
int mi355_attention_fwd(const void* q, const void* k, const void* v, void* out, float* lse, int B, int H, int Nq,
                        int Nk, int D, int ldq, int ldk, int ldv, int ldo, float scale, void* stream);
/* The same forward on queries that already carry the softmax scale and the change of base (q' = q * scale * log2(e),
 * folded by the caller into the projection that produces q - the no-gradient denoiser call of image_generator.py:92-95
 * folds it into the LayerNorm-folded query weights): scores q' k^T are exponents of 2.  Heads with a spare padding column
 * (D < 32 or 32 < D < 64) run a key loop without a running maximum (a workgroup whose row sums leave 2^-100 .. 2^100
 * repeats its loop with one), with 64 queries per wave from 2 048 queries and 1 024 keys on; other head sizes take the
 * plain kernel.  Results equal mi355_attention_fwd's up to the
 * rounding of q'. */
int mi355_attention_fwd_prescaled(const void* q, const void* k, const void* v, void* out, float* lse, int B, int H, int Nq,
                                  int Nk, int D, int ldq, int ldk, int ldv, int ldo, void* stream);
/* Forward with a key-padding mask (encoder/text_encoder.py:31-34 `src_key_padding_mask=~attention_mask`), given as
 * the number of valid (leading) keys of each sample: kv_len [B] int32 on the device, 1 <= kv_len[b] <= Nk. */
int mi355_attention_fwd_masked(const void* q, const void* k, const void* v, void* out, float* lse, const int* kv_len,
                               int B, int H, int Nq, int Nk, int D, int ldq, int ldk, int ldv, int ldo, float scale,
                               void* stream);
/* Backward: delta [B][H][Nq] is scratch (row sums of dout * o, produced by the dQ launch for the dK / dV launch; with at
 * most 128 keys one launch computes dQ, dK and dV and leaves it untouched); dq uses ldq, dout / o use ldo.  dk / dv are
 * written with row strides lddk / lddv
 * (0: ldk / ldv): keys and values that are column slices of a wider matrix - the text keys / values of all
 * cross-attention layers come out of one product - can have contiguous gradients. */
int mi355_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                        float* delta, void* dq, void* dk, void* dv, int B, int H, int Nq, int Nk, int D, int ldq,
                        int ldk, int ldv, int ldo, int lddk, int lddv, float scale, void* stream);

/* Self-attention of the text encoder in TRAINING mode (encoder/text_encoder.py:16-23,31-34 as run by
 * encoder/train_encoder.py:103-105): the key-padding mask as valid key counts (kv_len, device int32 [B], or NULL) and
 * dropout with probability drop_p on the attention probabilities, forward and backward.  The dropout mask of an entry
 * is a hash of (seed, head, query, key): pass the same seed to the backward call; nothing is stored.  drop_p = 0 runs
 * the plain kernels (any head dim <= 160); drop_p > 0 needs D <= 96.  Gradients of masked keys are written as zeros. */
int mi355_attention_fwd_train(const void* q, const void* k, const void* v, void* out, float* lse, const int* kv_len,
                              float drop_p, unsigned seed, int B, int H, int Nq, int Nk, int D, int ldq, int ldk, int ldv,
                              int ldo, float scale, void* stream);
int mi355_attention_bwd_train(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                              float* delta, void* dq, void* dk, void* dv, const int* kv_len, float drop_p, unsigned seed,
                              int B, int H, int Nq, int Nk, int D, int ldq, int ldk, int ldv, int ldo, float scale,
                              void* stream);

/* ---- normalisation ------------------------------------------------------------------------------
 * GroupNorm over [B][HW][C] with G groups, optional fused SiLU.  stats [B][G][2] = (mean, rstd).
 * scratch (both calls): mi355_groupnorm_scratch_floats(B, HW, C, G) floats.
 * Backward: dgamma and dbeta may both be NULL (frozen affine parameters: only dx is produced).  dres (optional,
 * bf16, shape of x) is added to dx in the same pass: the gradient of a branch that reads x beside the norm. */
long long mi355_groupnorm_scratch_floats(int B, int HW, int C, int G);
int mi355_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, float* scratch,
                        int B, int HW, int C, int G, float eps, int silu, void* stream);
int mi355_groupnorm_bwd(const void* x, const void* dy, const float* stats, const float* gamma, const float* beta,
                        void* dx, float* dgamma, float* dbeta, float* scratch, int B, int HW, int C, int G, int silu,
                        const void* dres, void* stream);
/* mi355_groupnorm_bwd that also returns dx_colsum [B][C] fp32 = per-sample column sums of the dx it stores: when x is the
 * output of a convolution with a per-sample bias (the timestep-embedding projection added by the first convolution of
 * a residual block), that is the bias's gradient, and the separate pass over dx is not needed.  colsum_scratch: a
 * second buffer of mi355_groupnorm_scratch_floats(B, HW, C, G) floats. */
int mi355_groupnorm_bwd_colsum(const void* x, const void* dy, const float* stats, const float* gamma, const float* beta,
                               void* dx, float* dgamma, float* dbeta, float* scratch, int B, int HW, int C, int G, int silu,
                               const void* dres, float* dx_colsum, float* colsum_scratch, void* stream);
/* GroupNorm(+SiLU) forward on the channel concatenation [x1 | x2] of two NHWC tensors (x1 [B][HW][C1], x2 [B][HW][C2])
 * without materialising it: the input of the up-path residual blocks, hidden state | skip connection (the
 * `torch.cat([hidden_states, res_hidden_states], dim=1)` of the third-party up block the reference calls at
 * image_generator.py:93-95).  y [B][HW][C1 + C2]; stats and scratch as for mi355_groupnorm_fwd with C = C1 + C2. */
int mi355_groupnorm_fwd_cat(const void* x1, int C1, const void* x2, int C2, const float* gamma, const float* beta, void* y,
                            float* stats, float* scratch, int B, int HW, int G, float eps, int silu, void* stream);
/* Backward of mi355_groupnorm_fwd_cat: x1 / x2 as in the forward call, dy [B][HW][C1 + C2]; the input gradient is written
 * as two tensors, dx1 [B][HW][C1] and dx2 [B][HW][C2], and dres1 / dres2 (optional, both or neither; shapes of x1 / x2)
 * are added to them in the same pass.  dgamma / dbeta [C1 + C2] or both NULL; scratch as for mi355_groupnorm_bwd with
 * C = C1 + C2 (the deferred parameter-gradient reduction of mi355_reduce_rows_batched applies unchanged). */
int mi355_groupnorm_bwd_cat(const void* x1, int C1, const void* x2, int C2, const void* dy, const float* stats,
                            const float* gamma, const float* beta, void* dx1, void* dx2, float* dgamma, float* dbeta,
                            float* scratch, int B, int HW, int G, int silu, const void* dres1, const void* dres2, void* stream);
/* mi355_groupnorm_fwd without a scratch argument, for callers that cannot allocate device memory (code under HIP graph
 * capture, serving loops and harnesses that forbid hipMalloc).  No counterpart in the reference.  The strips' partial
 * sums live in a 1 MiB device array built into the library (one copy per GPU, held until the process ends), so the
 * call needs B * strips * G * 2 <= 262144 floats (strips <= 32: e.g. B <= 128 at G = 32) and returns
 * MI355_ERR_UNSUPPORTED beyond that; it calls no allocating routine.  Same results as mi355_groupnorm_fwd, bit for bit.
 * Every call on a GPU shares that GPU's array: calls that overlap on one GPU (different streams) race.  The first call
 * on a device looks the array's address up and must therefore be made outside a stream capture. */
int mi355_groupnorm_fwd_static(const void* x, const float* gamma, const float* beta, void* y, float* stats, int B, int HW,
                               int C, int G, float eps, int silu, void* stream);
/* The reductions that the two backward calls skip when given NULL parameter-gradient pointers, for many layers in
 * one launch: n records of 6 int64 in device memory {partial, out, out_hi | 0, rows, N, split}; out[c] = sum over
 * rows of partial[r][c] for c < split (every c when out_hi is 0), out_hi[c - split] otherwise.  GroupNorm: partial =
 * scratch + B*strips*G*2 floats with strips = scratch_floats / (B*(2G + 2C)), rows = B*strips, N = 2C, out = dbeta,
 * out_hi = dgamma, split = C.  LayerNorm: partial = scratch, rows = scratch_floats / 2C, N = 2C, out = dgamma_dbeta. */
int mi355_reduce_rows_batched(const long long* table, int n, int max_cols, void* stream);
/* LayerNorm over rows of [M][C]; stats [M][2] = (mean, rstd). */
int mi355_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, int M, int C,
                        float eps, void* stream);
long long mi355_layernorm_bwd_scratch_floats(int M, int C);
/* dgamma_dbeta: [2][C] floats (dgamma row, then dbeta row), or NULL when only dx is wanted; dres as above. */
int mi355_layernorm_bwd(const void* x, const void* dy, const float* stats, const float* gamma, void* dx,
                        float* dgamma_dbeta, float* scratch, int M, int C, const void* dres, void* stream);

/* ---- elementwise / layout -----------------------------------------------------------------------*/
/* out[m][:] = softmax(scale * in[m][:]) over bf16 rows (single-head attention of the auto-encoder mid block) */
int mi355_softmax_rows_bf16(const void* in, void* out, int M, int N, float scale, void* stream);
int mi355_gelu_bf16(const void* in, void* out, long long n, void* stream);                   /* erf GELU (text_encoder.py:22) */
/* Training mode of the text encoder (encoder/text_encoder.py:16-23: TransformerEncoderLayer(dropout=0.2, activation='gelu')).
 * The dropout mask of element i is a hash of (seed, i): the backward call regenerates it from the same seed, nothing is
 * stored.  n a multiple of 8, bf16.
 *   dropout_add : y = residual + dropout_p(x)   (residual may be NULL; backward of x: the same call on dy without residual)
 *   gelu_dropout: y = dropout_p(gelu_erf(x));  backward: dx = dropout_p(dy) * gelu_erf'(x) */
int mi355_dropout_add_bf16(const void* x, const void* residual, void* y, long long n, float p, unsigned seed, void* stream);
int mi355_gelu_dropout_fwd(const void* x, void* y, long long n, float p, unsigned seed, void* stream);
int mi355_gelu_dropout_bwd(const void* x, const void* dy, void* dx, long long n, float p, unsigned seed, void* stream);
/* Pooled output (encoder/text_encoder.py:37-40 F.normalize): y = x / max(|x|, 1e-12) per fp32 row, norms[b] = the
 * denominator; backward dx = (dy - y (y . dy)) / norms. */
int mi355_l2norm_rows_f32(const float* x, float* y, float* norms, int B, int D, void* stream);
int mi355_l2norm_rows_bwd_f32(const float* y, const float* norms, const float* dy, float* dx, int B, int D, void* stream);
int mi355_geglu_fwd(const void* in, void* out, long long M, int F, void* stream);           /* in [M][2F] */
int mi355_geglu_bwd(const void* in, const void* dout, void* din, long long M, int F, void* stream);
int mi355_silu_f32_to_bf16(const float* x, void* y, long long n, void* stream);
int mi355_silu_bwd_f32(const float* x, const float* dy, float* dx, long long n, void* stream);
int mi355_timestep_embed(const float* t, void* out, int B, int dim, void* stream);          /* cos | sin */
int mi355_cast_f32_to_bf16(const float* x, void* y, long long n, void* stream);
int mi355_cast_bf16_to_f32(const void* x, float* y, long long n, void* stream);
/* dst[i] = bf16(src[i]) over n ranges in ONE launch (the pack pass of the bf16 gradient staging buffer, trainer.py:257-259's
 * gradients on their way to the optimizer / the all-reduce): table (device) holds n records of 3 int64 {fp32 source
 * address (16-byte aligned), bf16 destination address (8-byte aligned), count (multiple of 4)}; blocks_per_range
 * workgroups walk each range. */
int mi355_cast_ranges_f32_to_bf16(const long long* table, int n, int blocks_per_range, void* stream);
int mi355_transpose_bf16(const void* in, void* out, int G, int R, int C, void* stream);    /* [G][R][C]->[G][C][R] */
int mi355_conv_weight_transpose(const void* w, void* wt, int Co, int taps, int Ci, void* stream);
/* Many weight transposes in ONE launch.  jobs (device, 6 int64 each): src ptr, dst ptr, R, C, T, first tile;
 * src [R][T][C] -> dst [C][T reversed][R] (T = 1: plain transpose).  total_tiles = sum of 64x64 tiles x T. */
int mi355_transpose_batched(const long long* jobs, int njobs, int total_tiles, void* stream);
int mi355_nchw_f32_to_nhwc_bf16(const float* x, void* y, int B, int Cs, int Cd, int HW, void* stream);
int mi355_nhwc_f32_to_nchw_f32(const float* x, float* y, int B, int Cs, int Cd, int HW, void* stream);
int mi355_sumpool2x2_bf16(const void* dy, void* dx, int B, int H, int W, int C, void* stream);

/* ---- objectives (image_generator.py:125 and loss.py:16-28) ---------------------------------------
 * loss = mean((pred - target)^2); grad (optional) = 2 (pred - target) / n.  loss is one device float. */
int mi355_mse_loss_fused(const float* pred, const float* target, float* grad, float* loss, long long n, void* stream);
long long mi355_contrastive_scratch_floats(int B, int D);
int mi355_contrastive_loss_fused(const float* text, const float* image, const float* log_temp, float* loss,
                                 float* d_text, float* d_image, float* d_log_temp, float* scratch, int B, int D,
                                 void* stream);

/* ---- diffusion schedule (image_generator.py:116 and :97-99) --------------------------------------*/
int mi355_ddpm_add_noise(const float* x0, const float* noise, const long long* t, const float* alphas_cumprod,
                         float* out, int B, long long per_sample, void* stream);
/* coef (device) = {alpha_bar_t, alpha_bar_prev, guidance}; eps = [uncond | text] when cfg != 0. */
int mi355_ddim_step_cfg(const float* eps, const float* x, float* out, const float* coef, long long n, int cfg,
                        void* stream);

/* ---- optimizer (trainer.py:257-259) ---------------------------------------------------------------
 * hyper (device floats): lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, max_grad_norm,
 * grad pre-scale.  state (two device floats): [0] = sum of squared gradients (input); a non-finite norm skips the
 * update (the reference's NaN guard, trainer.py:247-253, without its host synchronisation) and adds 1 to state[1], which
 * the library never clears: the caller's count of skipped steps. */
int mi355_sumsq_f32(const float* g, float* out, long long n, void* stream);
/* *out += the sum of squares over n ranges in one launch: table (device, int64 pairs) = address of fp32 values, count.
 * A NEGATIVE count marks |count| values that are partial sums already (added as they are): the slots that
 * mi355_wgrad_job_fill's sumsq_slots names. */
int mi355_sumsq_ranges(const long long* table, int n, float* out, void* stream);
int mi355_zero_f32(float* p, long long n, void* stream);
/* Zero n ranges in one launch: table (device, int64 pairs) = address (16-byte aligned), bytes (multiple of 16). */
int mi355_zero_ranges(const long long* table, int n, void* stream);
/* dst <- src over n ranges in one launch: table (device) = n records of 3 int64 {source address, destination address,
 * bytes}; addresses 16-byte aligned, bytes a multiple of 16; blocks_per_range workgroups walk each range.  Packs the
 * to_k / to_v weights of every cross-attention layer (operands of image_generator.py:124's denoiser call that all
 * multiply the same text states) into one operand after an optimizer step, so that they run as ONE product. */
int mi355_copy_ranges(const long long* table, int n, int blocks_per_range, void* stream);
int mi355_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, const float* hyper, float* state,
                     long long n, void* stream);
/* The same two steps on bf16 gradients: a data-parallel step feeds the optimizer the all-reduced bf16 staging
 * buffer directly instead of unpacking it to fp32 first (n a multiple of 8 for the norm, of 4 for the update). */
int mi355_sumsq_bf16(const void* g16, float* out, long long n, void* stream);
/* The same sum in a fixed order of additions (per-block partials into `partials`, at most max_partials of them, then one
 * block adds them by index): identical reduced gradients give bit-identical norms on every data-parallel rank, so the
 * replicas apply the same clip factor and stay bit-identical (the atomic form above differs in the last bit from run
 * to run). */
int mi355_sumsq_bf16_ordered(const void* g16, float* out, float* partials, int max_partials, long long n, void* stream);
int mi355_adamw_step_g16(float* p, const void* g16, float* m, float* v, void* p_bf16, const float* hyper,
                         float* state, long long n, void* stream);

/* ---- LoRA (lora.py:23-24): rank-r side products around the fused GEMM epilogue --------------------*/
/* t = x A^T (fp32 [M][R]; A may be several adapters' factors stacked, R <= 16).  t16 (optional): bf16 copy
 * scale16 * t, [M][8] when R <= 8 else [M][16], zero padded. */
int mi355_lora_down(const void* x, const float* A, float* t, int M, int K, int R, void* t16, float scale16, void* stream);
/* dt[m][s*R + r] = alpha * sum_{n in segment s} dy[m][n] Bm[n][r]: S adapters row-stacked in Bm [N][R] (S = 1: one
 * adapter; S > 1 needs R = 4, R*S <= 16).  dt16 (optional): bf16 copy, [M][8] when R*S <= 8 else [M][16]. */
int mi355_lora_dt(const void* dy, const float* Bm, float* dt, int M, int N, int R, int S, float alpha, void* dt16, void* stream);
/* out[c][r] = scale * sum_m y[m][c] t[m][r].  flags bit 0: out is [R][C]; bit 1: add to `out` instead of overwriting.
 * (A training step computes the same sums on the MFMA weight-gradient queue from the bf16 copies above.) */
int mi355_lora_outer(const void* y, const float* t, float* out, int M, int C, int R, float scale, int flags,
                     void* stream);
/* bf16 operands of the adapters' skinny GEMMs (t = x A^T, dt = dy B), all adapters in one launch.  n records of 6
 * int64 in device memory: {0, src, dst, count, 0, 0}: dst[i] = bf16(src[i]);  {1, src fp32 [C][r], dst, C, r, ld}:
 * dst[q*ld + c] = bf16(src[c*r + q]). */
int mi355_lora_pack(const long long* table, int n, void* stream);
/* n records {src, dst, rows, cols, src_ld, dst_ld} (int64, device memory): dst[r][c] = src[r][c], fp32. */
int mi355_copy2d_f32_batched(const long long* table, int n, void* stream);

/* ---- data plane: read-only LMDB (dataset.py:173-183,213-216; the file format make_lmdb.py:39-75 writes) ------------
 * Host code, no GPU.  liblmdb is not a dependency: the environment's data.mdb is mapped and its B+tree walked in place.
 * open: `path` is the environment directory (with data.mdb) or the data file; NULL + mi355_last_error() on failure.
 * entries: what `txn.stat()['entries']` returns (the data set's length is entries // 2, dataset.py:183).
 * get: 0 = found (*data points into the map, valid until close, *size bytes), 1 = no such key (`txn.get` -> None),
 * negative = damaged or unsupported file.  Safe to call from many threads (the map is read-only). */
void* mi355_lmdb_open(const char* path);
long long mi355_lmdb_entries(void* env);
int mi355_lmdb_depth(void* env);
int mi355_lmdb_page_size(void* env);
int mi355_lmdb_get(void* env, const void* key, int klen, const void** data, long long* size);
int mi355_lmdb_close(void* env);

/* ---- data parallelism over RCCL (replaces DistributedDataParallel's gradient reducer: setup_training.py:39-41,
 * trainer.py:74-77 - one process per device, gradients summed over the ranks while backward runs) ----------------------
 * RCCL is bound when the first of these is called (the librccl.so.1 already in the process, else the system's): a
 * single-GPU user never loads it.  Rank 0 makes a unique id (mi355_dp_unique_id_bytes() bytes of host memory) and hands it
 * to the other ranks by any means it has; every rank then calls comm_create with its device current (collective: returns
 * when all `world` ranks have joined).  allreduce_sum: in place over `count` elements at the device pointer `buf`, dtype
 * 0 = bf16, 1 = fp32, asynchronous on `stream` with no hop through another stream; the ranks must issue their
 * all-reduces in the same order.  comm_world returns the number of ranks RCCL reports.  One communicator per process;
 * destroy it once, at exit, after the stream has drained. */
int mi355_dp_unique_id_bytes(void);
int mi355_dp_unique_id(void* out, int bytes);
int mi355_dp_comm_create(void** comm, const void* unique_id, int bytes, int rank, int world);
int mi355_dp_comm_world(void* comm);
int mi355_dp_allreduce_sum(void* comm, void* buf, long long count, int dtype, void* stream);
int mi355_dp_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif
