/*
 * mfc.h -- C ABI of the MI355X (gfx950) MeanFlow-audio-codec hot path.
 *
 * The reference (gabrieldernbach/meanflow_audio_codec) has no FFI: its hot
 * path sits behind Python strategy objects.  Each entry point below names the
 * reference interface (file:line, relative to the reference root) whose
 * arithmetic it replaces.  The Python host package `meanflow_audio_codec_amd`
 * binds these with ctypes and re-presents them under the reference's names
 * (TokenizationStrategy, LossStrategy, model apply, sample, train_step).
 *
 * Conventions
 *  - every function returns 0 on success, a negative MFC_E* code otherwise;
 *    nothing is launched when an argument check fails;
 *  - all pointers are caller-owned DEVICE pointers unless stated otherwise;
 *    no function allocates, frees or synchronises (graph-capture safe);
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *  - `dtype` selects the storage type of activations/weights: MFC_F32 or
 *    MFC_BF16; accumulation is always fp32;
 *  - matrices are row-major with explicit leading dimensions in ELEMENTS.
 */
#ifndef MFC_H
#define MFC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFC_OK 0
#define MFC_EINVAL (-22) /* bad shape / argument            */
#define MFC_ENOSYS (-38) /* unsupported configuration       */
#define MFC_EFAULT (-14) /* null pointer                    */
#define MFC_EHIP (-5)    /* HIP launch error (hipGetLastError) */

#define MFC_F32 0
#define MFC_BF16 1

/* library / ABI version and a one-line description of the build.  Version 2 (round 2): row_stride / data_size arguments
 * of mfc_sample_tr and mfc_flow_prepare, caller-owned workspaces instead of atomics (mfc_gemm_ws_elems,
 * mfc_cnx_ws_elems, MFC_FLOW_LOSS_WS_PER_ROW), colsum output of mfc_gemm_adamw.  Version 3 (round 3): additions only --
 * mfc_randn_dev (noise draw as a graph node); mfc_cnx_stats_save / mfc_cnx_apply_n1 / mfc_cnx_bwd_stats_n1 /
 * mfc_cnx_bwd_main_n1 (ConvNeXt passes from a kept n1); mfc_colsum_tall / mfc_colsum_ws_elems; mfc_chanmlp_fwd /
 * mfc_chanmlp_bwd / mfc_chanmlp_ws_elems (fused channel MLP of the Mixer). */
#define MFC_ABI_VERSION 3
int mfc_abi_version(void);
const char* mfc_build_info(void);

/* ------------------------------------------------------------------ */
/* MDCT tokenizer  (preprocessing/mdct.py)                             */
/* ------------------------------------------------------------------ */

/* _prepare_mdct frame count, preprocessing/mdct.py:491:
 * 1 if T < N else (T-N)/hop + 1.  Returns <0 on bad arguments. */
int64_t mfc_mdct_num_frames(int64_t T, int N, int hop);
/* output length of the inverse, preprocessing/mdct.py:513:
 * (n_frames-1)*hop + 2N */
int64_t mfc_mdct_out_len(int64_t n_frames, int N, int hop);

/* Forward MDCT, replaces mdct()/_mdct_direct (preprocessing/mdct.py:143-198,
 * 317-327, 347-358, 410-422, 476-495).
 *   x [B, T] fp32 (row stride ldx >= T)  ->  X [B, n_frames, N] fp32 (dense).
 * Implicit right zero padding to (n_frames-1)*hop+2N.  N must be even; a
 * power-of-two N uses the LDS FFT kernel, any other even N the direct-basis
 * kernel (same definition, O(N^2)). */
int mfc_mdct_fwd(const float* x, int64_t B, int64_t T, int64_t ldx, int N, int hop,
                 float* X, void* stream);

/* Inverse MDCT + overlap-add, replaces imdct()/_imdct_direct/_overlap_add
 * (preprocessing/mdct.py:200-256, 330-340, 361-372, 517-540).
 *   X [B, n_frames, N] fp32 -> y [B, out_len] fp32 (row stride ldy >= out_len),
 *   out_len = (n_frames-1)*hop + 2N.  Deterministic (no atomics). */
int mfc_mdct_inv(const float* X, int64_t B, int64_t n_frames, int N, int hop,
                 float* y, int64_t ldy, void* stream);

/* ------------------------------------------------------------------ */
/* Data front end (datasets/audio.py)                                  */
/* ------------------------------------------------------------------ */

/* Rational-ratio polyphase resampler: the device-side step between the
 * reference's 44.1 kHz-only loader (datasets/audio.py:236-262 keeps the file rate)
 * and the 24 kHz clips of the audio configuration.
 *   y[r,n] = sum_j x[r,j] * h[n*down - j*up + (nh-1)/2],  0 <= n < T_out = ceil(T_in*up/down)
 * (scipy.signal.resample_poly, padtype="constant"; h = odd-length low-pass at rate
 * up*fs_in, already multiplied by up).  x [rows, T_in] fp32 (row stride ldx),
 * y [rows, T_out] fp32 (row stride ldy), h [nh] fp32 on the device, nh odd, <= 16384. */
int64_t mfc_resample_out_len(int64_t T_in, int up, int down);
int mfc_resample_poly(const float* x, int64_t rows, int64_t T_in, int64_t ldx, int up, int down,
                      const float* h, int nh, float* y, int64_t ldy, void* stream);

/* ------------------------------------------------------------------ */
/* Dense layers (flax.linen.Dense call sites: models/mlp_flow.py:18-31,  */
/* models/conv_flow.py:142-160,188-202, models/mlp_mixer.py)            */
/* ------------------------------------------------------------------ */

/* flags for mfc_gemm */
#define MFC_GEMM_TRANS_A 1   /* A is stored [K, M] (lda = row stride of that) */
#define MFC_GEMM_TRANS_B 2   /* B is stored [N, K]                            */
#define MFC_GEMM_ACCUM 4     /* C += result (C read in its own dtype)         */
#define MFC_GEMM_GELU 8      /* tanh-GELU on rows < act_rows; rows >= act_rows
                                are tangents: t * gelu'(pre) with pre taken
                                from row (r - act_rows) of the SAME product   */
#define MFC_GEMM_LN16 16     /* rows < bias_rows: LayerNorm (no affine, eps 1e-6) over every aligned
                                group of 16 output columns (one NHWC pixel of the ConvNeXt map);
                                1/sigma of each group goes to ln_rstd[row*(N/16) + group].  Needs
                                N % 16 == 0, 16-byte aligned C, no split-K.                    */
#define MFC_GEMM_LN16T 32    /* with MFC_GEMM_LN16: rows >= bias_rows are tangents of the rows
                                (r - bias_rows) and receive the tangent of that LayerNorm,
                                rho (xd - mean(xd) - n mean(n (xd - mean(xd)))); needs
                                M - bias_rows <= bias_rows.                                     */

/* C[M,N] = alpha * (op(A)[M,K] . op(B)[K,N] + bias[N] on rows < bias_rows)
 *          + beta_res * R[M,N]
 * A,B,C,R in `dtype`; bias fp32 (may be NULL); R may be NULL.
 * Used for every Dense forward, its tangent (row-stacked [x; xdot] so the
 * weight tile is read once), input-gradient and
 * weight-gradient products.  `splitk` > 1 splits K over workgroups: slice z writes its partial
 * product into slab z of the fp32 workspace `ws` (mfc_gemm_ws_elems floats; not initialised by
 * the caller) and the epilogue, a second kernel, sums the slabs in slice order -- DETERMINISTIC
 * (bitwise reproducible run to run; no atomics). */
int64_t mfc_gemm_ws_elems(int flags, int64_t M, int64_t N, int64_t K, int splitk);
int mfc_gemm(int dtype, int flags, int64_t M, int64_t N, int64_t K,
             const void* A, int64_t lda, const void* B, int64_t ldb,
             void* C, int64_t ldc,
             const float* bias, int64_t bias_rows, int64_t act_rows,
             float alpha, const void* R, int64_t ldr, float beta_res,
             int splitk, float* ws, float* ln_rstd, void* stream);

/* Weight-gradient product with the AdamW update of that weight as its epilogue (single-GPU fast path: no gradient
 * exchange between jax.value_and_grad and optax.adamw, trainers/training_steps.py:32-33):
 *   g = grad_scale * op(A)[M,K] . op(B)[K,N]   (rounded to bf16, as the two-kernel path stores it)
 *   m, v, p <- AdamW(g)  (mfc_adamw's arithmetic; fp32 [M,N] dense);  p_bf16[M,N] <- bf16(p)
 * bf16 operands; N % 16 == 0 and 16-byte aligned buffers, else MFC_ENOSYS (use mfc_gemm + mfc_adamw).
 * colsum (may be NULL): fp32 [N], overwritten with colsum_scale * sum_k B[k][n] -- the bias gradient of the same
 * Dense layer when B = dY (value_and_grad gives dW = X^T dY and db = sum_rows dY together); needs a non-transposed B
 * and K > 32, else MFC_ENOSYS.  Fixed-order sums (deterministic). */
int mfc_gemm_adamw(int flags, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                   int64_t ldb, float grad_scale, float* p, float* m, float* v, void* p_bf16, float lr,
                   float b1, float b2, float eps, float wd, int64_t step, float* colsum, float colsum_scale,
                   void* stream);

/* ------------------------------------------------------------------ */
/* ConvNeXt block interior (models/conv_flow.py:65-115,162-186)        */
/* ------------------------------------------------------------------ */
/* The spatial part of ConditionalConvNeXtBlock, between input_proj2 and
 * output_proj1, on NHWC maps h0 [R, s, s, 16] (R = rows of the row-stacked
 * batch; channel count C = min(16, cond/4) = 16 for every shipped config,
 * other C -> MFC_ENOSYS).  The kernels take h1 = LN_C(h0) and, for a tangent, h1dot = the tangent of
 * that LayerNorm (both produced by mfc_gemm's MFC_GEMM_LN16 / MFC_GEMM_LN16T epilogue, or by
 * mfc_ln16_fwd / mfc_ln16_jvp); rho0 = 1/sqrt(var+eps) per pixel [R, s*s] fp32 is only needed by
 * mfc_cnx_bwd_conv.  s <= 8000.
 *   h2 = (1+scale) * LN_C(h0) + shift                 conv_flow.py:181-186
 *   c1 = Conv3x3_SAME(h2) ; n1 = LN_C(c1)             conv_flow.py:74-84
 *   g1 = gelu(Conv1x1_{16->32}(n1))                   conv_flow.py:87-88
 *   y  = GRN(g1)                                      conv_flow.py:22-45
 *   o  = Conv1x1_{32->16}(y) * layer_scale + h2       conv_flow.py:95-115
 * GRN needs sum_{hw} g1^2 per (row, channel) before it can be applied, so the
 * chain runs twice (stats pass, apply pass) and never writes the 32-channel
 * intermediates to HBM.  All "dot" arguments are the forward-mode tangents
 * (SURVEY Appendix C); pass NULL for a primal-only call.  mfc_cnx_params /
 * mfc_cnx_grads are HOST structs holding DEVICE pointers. */
typedef struct {
    const void* conv_w;      /* [3,3,16,16] dtype   conv_block/Conv_0/kernel */
    const float* conv_b;     /* [16]                                        */
    const void* exp_w;       /* [16,32] dtype       conv_block/Conv_1/kernel */
    const float* exp_b;      /* [32]                                        */
    const float* grn_gamma;  /* [32]  GlobalResponseNormalization_0/gamma   */
    const float* grn_beta;   /* [32]                                        */
    const void* con_w;       /* [32,16] dtype       conv_block/Conv_2/kernel */
    const float* con_b;      /* [16]                                        */
    const float* ls;         /* [16]  layer_scale_gamma                     */
} mfc_cnx_params;

/* fp32 gradient accumulators (+=) for the small parameters above */
/* DETERMINISM.  No kernel of this section uses atomics: every workgroup stores its partial sums into its own record of
 * the caller's workspace `ws` (mfc_cnx_ws_elems(R, s) floats, contents irrelevant on entry, may be shared by all
 * calls on one stream) and a follow-up kernel inside the same call adds the records in workgroup order.  Results are
 * bitwise reproducible run to run for a given (R, s) and mfc_cnx_max_blocks setting. */
typedef struct {
    float* conv_w; float* conv_b; float* exp_w; float* exp_b;
    float* grn_gamma; float* grn_beta; float* con_w; float* con_b; float* ls;
} mfc_cnx_grads;

int64_t mfc_cnx_ws_elems(int64_t R, int s);

/* pass 1: S1[r,ch] = sum_hw g1^2 ; S2[r,ch] = sum_hw g1*g1dot (if h1dot).
 * scale/shift (and their tangents) are fp32 [R,16]; S1/S2 fp32 [R,32] are overwritten. */
int mfc_cnx_stats(int dtype, int64_t R, int s, const void* h1, const void* h1dot,
                  const float* scale, const float* shift, const float* scaledot, const float* shiftdot,
                  const mfc_cnx_params* p, float* S1, float* S2, float* ws, void* stream);

/* GRN scalars from the statistics (conv_flow.py:32-37 and Appendix C):
 * G = sqrt(S1); n = mean_ch G; q = G/(n+1e-6); qdot from S2 (NULL: skipped). */
int mfc_grn_finalize(int64_t R, const float* S1, const float* S2, float* G, float* q, float* qdot,
                     void* stream);

/* pass 2: o (and odot) [R,s,s,16] dtype. */
int mfc_cnx_apply(int dtype, int64_t R, int s, const void* h1, const void* h1dot,
                  const float* scale, const float* shift, const float* scaledot, const float* shiftdot,
                  const mfc_cnx_params* p, const float* q, const float* qdot,
                  void* o, void* odot, void* stream);

/* backward pass 1: dq[r,ch] = sum_hw dy*g1   (overwritten) */
int mfc_cnx_bwd_stats(int dtype, int64_t R, int s, const void* h0, const float* scale, const float* shift,
                      const mfc_cnx_params* p, const float* q, const void* dout,
                      float* dq, float* ws, void* stream);

/* kG[r,ch] = (dL/dG)/G from dq (zero where G == 0); dgamma[ch] += sum_r dq */
int mfc_grn_bwd_finalize(int64_t R, const float* G, const float* dq, float* kG, float* dgamma, void* stream);

/* backward pass 2: dc1 [R,s,s,16] dtype (gradient at the 3x3 conv output) and the
 * gradients of exp_w, exp_b, con_w and ls. */
int mfc_cnx_bwd_main(int dtype, int64_t R, int s, const void* h0, const float* scale, const float* shift,
                     const mfc_cnx_params* p, const float* q, const float* kG, const void* dout,
                     void* dc1, const mfc_cnx_grads* g, float* ws, void* stream);

/* ---- the same passes starting from a kept n1 (ABI 3) ----------------------------------------------------------
 * Everything downstream of n1 = LN_C(conv3x3(FiLM(h1))) (conv_flow.py:74-84) is per pixel, so a statistics pass that
 * keeps n1 and its 1/sigma lets the apply pass and the first two reverse passes run as plain streaming kernels (no
 * halo, no repeated conv / LayerNorm).  Results equal the h1-based entry points up to the rounding of n1 to `dtype`
 * (bit-identical forward in fp32 storage; in bf16 the expansion consumed the same bf16 n1 anyway). */

/* mfc_cnx_stats that also writes n1 [R,s,s,16] (dtype) and rho1 [R,s,s] (fp32) of the primal rows and, when h1dot
 * and n1dot_out are given, the tangent of n1 [R,s,s,16] (dtype). */
int mfc_cnx_stats_save(int dtype, int64_t R, int s, const void* h1, const void* h1dot,
                       const float* scale, const float* shift, const float* scaledot, const float* shiftdot,
                       const mfc_cnx_params* p, float* S1, float* S2, float* ws, void* n1_out, float* rho1_out,
                       void* n1dot_out, void* stream);
/* mfc_cnx_apply from n1 (and, n1dot != NULL, the tangent from n1dot: then h1dot, scaledot, shiftdot, qdot, odot are
 * required); h1 / h1dot are read for the residual branch o = ... + FiLM(h1).  Bit-identical to mfc_cnx_apply. */
int mfc_cnx_apply_n1(int dtype, int64_t R, int s, const void* n1, const void* n1dot, const void* h1, const void* h1dot,
                     const float* scale, const float* shift, const float* scaledot, const float* shiftdot,
                     const mfc_cnx_params* p, const float* q, const float* qdot, void* o, void* odot, void* stream);
/* mfc_cnx_bwd_stats / mfc_cnx_bwd_main from n1 (and rho1): same outputs, same workspace contract. */
int mfc_cnx_bwd_stats_n1(int dtype, int64_t R, int s, const void* n1, const mfc_cnx_params* p, const float* q,
                         const void* dout, float* dq, float* ws, void* stream);
int mfc_cnx_bwd_main_n1(int dtype, int64_t R, int s, const void* n1, const float* rho1, const mfc_cnx_params* p,
                        const float* q, const float* kG, const void* dout, void* dc1, const mfc_cnx_grads* g,
                        float* ws, void* stream);

/* backward pass 3: dh0 = LN/FiLM-backward(conv3x3^T(dc1) + dout); gradients of conv_w, conv_b, con_b and
 * grn_beta (the last two are linear in sum_hw dout); dscale/dshift [R,16] fp32 (overwritten). */
int mfc_cnx_bwd_conv(int dtype, int64_t R, int s, const void* h1, const float* rho0, const float* scale,
                     const float* shift, const mfc_cnx_params* p, const void* dc1, const void* dout,
                     void* dh0, const mfc_cnx_grads* g, float* dscale, float* dshift, float* ws, void* stream);

/* Tuning / test hook: the persistent ConvNeXt kernels launch at most this many workgroups, each walking a contiguous
 * range of tiles.  Default 0 = a per-kernel multiple of the resident workgroups (512 .. 3072); n > 0 (or env
 * MFC_CNX_MAX_BLOCKS) forces one value for every kernel, n = 0 restores the defaults; returns the previous value.
 * mfc_cnx_ws_elems follows the current setting: size the workspace after changing it. */
int64_t mfc_cnx_max_blocks(int64_t n);

/* First LayerNorm of the block (conv_flow.py:181, nn.LayerNorm over the 16 channels of each pixel):
 * y = LN_16(x) [n_pixels,16] dtype, rstd[n_pixels] fp32 (may be NULL).  The hot path fuses this into the
 * producing product (mfc_gemm flag MFC_GEMM_LN16); this entry point is the standalone form. */
int mfc_ln16_fwd(int dtype, int64_t n_pixels, const void* x, void* y, float* rstd, void* stream);

/* Tangent of that LayerNorm (what MFC_GEMM_LN16T fuses): n = LN_16(x), rstd from mfc_ln16_fwd, xdot the raw
 * tangent of x; ndot = rstd * (xd - n * mean(n * xd)) with xd = xdot - mean(xdot).  ndot may alias xdot. */
int mfc_ln16_jvp(int dtype, int64_t n_pixels, const void* n, const float* rstd, const void* xdot, void* ndot,
                 void* stream);

/* ------------------------------------------------------------------ */
/* Loss-step element-wise kernels                                      */
/* ------------------------------------------------------------------ */

/* sinusoidal_embedding (meanflow_audio_codec/utils.py:5-13) of the model time
 * input [t, h] (models/conv_flow.py:257-259, mlp_flow.py:181-183):
 *   cond[r,:] = emb(t[r]) + emb(h[r]) (+ add[r,:] if add != NULL), fp32 [R,dim]
 *   cond_dot  = tdot emb'(t) + hdot emb'(h)  (NULL tdot/hdot = 1; cond_dot NULL: skipped) */
int mfc_time_embed(int64_t R, int dim, const float* t, const float* h, const float* tdot,
                   const float* hdot, const float* add, float* cond, float* cond_dot, void* stream);

/* sample_tr / logit_normal (meanflow_audio_codec/utils.py:32-45,
 * trainers/time_sampling.py:39-135) from a Philox stream keyed (seed, step,
 * GLOBAL row): t,r = sigmoid(N(mean,std)), t=max, r=min, global rows
 * < data_size get r = t.  data_size = int(Bglobal*data_proportion) is computed by the CALLER, once, in double
 * precision as utils.py:41 does (the same integer then drives the host's row bookkeeping).  Local row i is global
 * row row0 + i*row_stride (data-parallel shards: contiguous = stride 1, interleaved = stride world_size).
 * pair=0: only t (logit-normal). */
int mfc_sample_tr(uint64_t seed, uint64_t step, int64_t row0, int64_t row_stride, int64_t B, int64_t data_size,
                  float mean, float std, int pair, float* t, float* r, void* stream);

/* LinearNoiseSchedule (trainers/noise_schedules.py:52-88; noise_min=0, noise_max=1 gives the
 * UniformNoiseSchedule :91-115):  z = (1-t) x + (nmin + nmax t) e  [dtype],
 * target = nmax e - x [fp32].  e is drawn N(0,1) from Philox (seed, step, global row row0 + b*row_stride)
 * when e_in == NULL (and written to e_out if given). */
int mfc_flow_prepare(int dtype, int64_t B, int64_t D, const float* x, const float* e_in,
                     const float* t, float noise_min, float noise_max, uint64_t seed, uint64_t step,
                     int64_t row0, int64_t row_stride, void* z, float* target, float* e_out, void* stream);

/* N(0,1) fp32 [B,D] keyed (seed, stream_id, row0+b): sampler start noise
 * (evaluators/sampling.py:50). */
int mfc_randn(uint64_t seed, uint64_t stream_id, int64_t row0, int64_t B, int64_t D, float* out,
              void* stream);
/* The same draw as a FIXED graph node (evaluators/sampling.py:50 inside the captured decoder): the first global row is
 * read from device memory (*row0_dev), the result is written in `dtype` [B,D], and *row0_dev += advance afterwards
 * (a second, one-thread launch on the same stream), so every replay of a captured graph draws fresh noise.
 * Bit-identical to mfc_randn(seed, stream_id, *row0_dev, ...) followed by a cast to `dtype`. */
int mfc_randn_dev(int dtype, uint64_t seed, uint64_t stream_id, int64_t* row0_dev, int64_t advance, int64_t B,
                  int64_t D, void* out, void* stream);

/* tanh-GELU on [M,N]: rows < act_rows primal, rows >= act_rows tangents
 * t*gelu'(pre[row-act_rows]); and its backward din = dout*gelu'(pre). */
int mfc_gelu_fwd(int dtype, int64_t M, int64_t N, int64_t act_rows, const void* pre, void* out,
                 void* stream);
int mfc_gelu_bwd(int dtype, int64_t n, const void* pre, const void* dout, void* din, void* stream);

/* Compound loss + gradient seed (trainers/loss_strategies.py:98-112,184-198,270-277;
 * meanflow_audio_codec/utils.py:16-25).
 *  kind 0: delta = u + (t-r) dudt - target        (iMF; FM with dudt == NULL)
 *  kind 1: delta = u - (target - clip(t-r,0,1) dudt)   (MeanFlow)
 *  only n_tan rows carry a tangent (rows with r == t multiply it by 0): n_tan >= 0: the FIRST n_tan rows, dudt row b;
 *  n_tan < 0 (ABI 3): the LAST -n_tan rows, dudt row b - (B + n_tan).
 *  mode 0: weighted L2  w=sg(1/(pe+c)^p), loss=mean(w pe); mode 1: MSE; mode 2: MeanFlow
 *  adaptive weight on the per-example MEAN square, exponent p = 1-gamma.
 * Means are over Bglobal (data-parallel shards pass their local B rows).
 * Outputs: pe[B], seed[B] (dL/ddelta = seed*delta), loss (scalar, this shard's
 * contribution), du [B,D] dtype (NULL: skipped).  ws: fp32 [B * MFC_FLOW_LOSS_WS_PER_ROW] partial sums (not
 * initialised by the caller); the per-example sums and the batch mean are added in a fixed order (deterministic). */
#define MFC_FLOW_LOSS_WS_PER_ROW 256
int mfc_flow_loss(int dtype, int kind, int mode, int64_t B, int64_t Bglobal, int64_t D, const void* u,
                  const void* dudt, int64_t n_tan, const float* t, const float* r,
                  const float* target, float p, float c, float* pe, float* seed, float* loss,
                  void* du, float* ws, void* stream);

/* out[N] (fp32) = (accumulate ? out : 0) + scale * sum_rows X[M,N]  -- bias gradients */
int mfc_colsum(int dtype, int64_t M, int64_t N, const void* X, int64_t ld, float scale, float* out,
               int accumulate, void* stream);
/* The same for tall matrices (M >= 2048 rows, N a multiple of 16 bytes of `dtype`, 16-byte aligned rows): two fixed-order
 * stages over a caller workspace of mfc_colsum_ws_elems(dtype, M, N) floats (0: shape not taken -> MFC_ENOSYS, use
 * mfc_colsum).  Bias gradients of the Mixer's per-token Dense layers (models/mlp_mixer.py:66-94), M = B * tokens. */
int64_t mfc_colsum_ws_elems(int dtype, int64_t M, int64_t N);
int mfc_colsum_tall(int dtype, int64_t M, int64_t N, const void* X, int64_t ld, float scale, float* out,
                    int accumulate, float* ws, void* stream);

/* out = a x + b y (y may be NULL) -- Heun update, evaluators/sampling.py:84 */
int mfc_axpby(int dtype, int64_t n, float a, const void* x, float b, const void* y, void* out,
              void* stream);
int mfc_cast(int src_dtype, int dst_dtype, int64_t n, const void* x, void* out, void* stream);

/* optax.adamw (trainers/train.py:236): m,v moments fp32, p fp32 master, optional
 * bf16 working copy p_bf16, gradient in grad_dtype scaled by grad_scale.
 * step is the 1-based update count (bias correction). */
int mfc_adamw(int grad_dtype, int64_t n, float* p, void* p_bf16, const void* g, float grad_scale,
              float* m, float* v, float lr, float b1, float b2, float eps, float wd, int64_t step,
              void* stream);

/* The same update for many (small) leaves in one launch per MFC_ADAMW_MULTI_MAX items: `items` is a HOST array of
 * n_items descriptors (copied into the kernel arguments; nothing is read from it after the call returns); element
 * arithmetic and results identical to one mfc_adamw per leaf.  A ConvFlow step has ~170 leaves below a million
 * elements (biases, conv kernels, GRN / layer-scale vectors): this replaces their ~6 us launches. */
typedef struct mfc_adamw_item {
    float* p;          /* fp32 master [n] */
    void* p_bf16;      /* optional bf16 working copy [n] (NULL: none) */
    const void* g;     /* gradient [n], grad_dtype */
    float* m;          /* fp32 moments [n] */
    float* v;
    int64_t n;
    int32_t grad_dtype; /* MFC_F32 / MFC_BF16 */
    int32_t reserved;
} mfc_adamw_item;
#define MFC_ADAMW_MULTI_MAX 48
int mfc_adamw_multi(int n_items, const mfc_adamw_item* items, float grad_scale, float lr, float b1, float b2,
                    float eps, float wd, int64_t step, void* stream);

/* ------------------------------------------------------------------ */
/* AdaLN / gating (MLP and MLP-Mixer velocity nets)                    */
/* ------------------------------------------------------------------ */
/* y = (1 + scale) * LN(x) + shift over the last axis (flax LayerNorm, no affine, eps 1e-6):
 * models/mlp_flow.py:96-110 (ConditionalResidualBlock), models/mlp_mixer.py AdaLN.
 * x,y [rows, W] dtype with leading dims ldx/ldy; modulation element of (row, col) is
 * mod[(row / mod_div) * ldm + col].  Rows >= act_rows are forward-mode tangents of rows
 * [0, rows-act_rows): their x/scale/shift rows hold xdot/scaledot/shiftdot. */
int mfc_adaln_fwd(int dtype, int64_t rows, int64_t act_rows, int64_t W, const void* x, int64_t ldx,
                  const void* scale, const void* shift, int64_t ldm, int64_t mod_div, void* y,
                  int64_t ldy, void* stream);
/* reverse pass: dx, dscale = dy*LN(x), dshift = dy.  mod_div == 1: dscale/dshift dtype [rows, W]
 * (ldd); mod_div > 1: fp32 [rows/mod_div, W], OVERWRITTEN with the sum over the mod_div rows of each group taken in
 * ascending row order (a second kernel, one workgroup per group: bitwise reproducible, no atomics). */
int mfc_adaln_bwd(int dtype, int64_t rows, int64_t W, const void* x, int64_t ldx, const void* scale,
                  int64_t ldm, int64_t mod_div, const void* dy, int64_t ldy, void* dx, void* dscale,
                  void* dshift, int64_t ldd, void* stream);
/* out = o * (1 + s2) * inv_k + res (models/mlp_flow.py:116-117) with tangent rows, and its reverse
 * (do = dy (1+s2) inv_k, ds2 = dy o inv_k; dres = dy). */
int mfc_gate_fwd(int dtype, int64_t rows, int64_t act_rows, int64_t W, const void* o, int64_t ldo,
                 const void* s2, int64_t ldm, const void* res, int64_t ldr, float inv_k, void* y,
                 int64_t ldy, void* stream);
int mfc_gate_bwd(int dtype, int64_t rows, int64_t W, const void* dy, int64_t ldy, const void* o,
                 int64_t ldo, const void* s2, int64_t ldm, float inv_k, void* dout_o, void* ds2,
                 int64_t ldd, void* stream);
/* dst[r,c] = (accumulate ? dst : 0) + alpha * src[r,c] on strided 2-D views (concat / slice plumbing
 * of models/mlp_flow.py:190-194 without intermediate tensors). */
int mfc_copy2d(int dtype, int64_t rows, int64_t W, const void* src, int64_t lds, void* dst, int64_t ldd,
               float alpha, int accumulate, void* stream);

/* dst[b, c, r] = alpha * src[b, r, c] (+ add[b, c, r] if add != NULL): the token/channel transposes of
 * MLPMixerBlock (models/mlp_mixer.py:80-84) with the residual add fused into the way back. */
int mfc_transpose(int dtype, int64_t batch, int rows, int cols, const void* src, void* dst, float alpha,
                  const void* add, void* stream);

/* Fused channel-mixing MLP of MLPMixerBlock on 16-channel tokens (models/mlp_mixer.py:66-94: Dense(channel_mix_dim) ->
 * gelu -> Dense(num_channels), + residual), without the [rows, H] hidden activation ever reaching HBM:
 *   out = gelu(a W1 + b1) W2 + b2 + res              a, res, out [rows, 16] dense; W1 [16, H]; W2 [H, 16]; b1 [H], b2 [16] fp32
 * rows [act_rows, rows) are tangents of rows [0, rows - act_rows):  out = (gelu'(a_i W1 + b1) * (adot W1)) W2 + res.
 * H % 16 == 0; res may be NULL.  a / res / out / b1 / b2 16-byte aligned. */
int mfc_chanmlp_fwd(int dtype, int64_t rows, int64_t act_rows, int64_t H, const void* a, const void* W1, const float* b1,
                    const void* W2, const float* b2, const void* res, void* out, void* stream);
/* Its reverse pass (primal rows only), recomputing the hidden activation from a:
 *   da [rows, 16] = (dy W2^T * gelu'(a W1 + b1)) W1^T;  dW1 [16, H], dW2 [H, 16] in `dtype`, db1 [H] fp32 (all OVERWRITTEN;
 *   db2 = column sums of dy: mfc_colsum).  H in {128, 256, 512} or a multiple of 1024 (MFC_ENOSYS otherwise).
 *   ws: mfc_chanmlp_ws_elems(rows, H) floats (per-workgroup partial sums, reduced in index order: no atomics). */
int64_t mfc_chanmlp_ws_elems(int64_t rows, int64_t H);
int mfc_chanmlp_bwd(int dtype, int64_t rows, int64_t H, const void* a, const void* dy, const void* W1, const float* b1,
                    const void* W2, void* da, void* dW1, float* db1, void* dW2, float* ws, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MFC_H */
