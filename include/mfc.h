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

/* library / ABI version and a one-line description of the build */
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

/* C[M,N] = alpha * (op(A)[M,K] . op(B)[K,N] + bias[N] on rows < bias_rows)
 *          + beta_res * R[M,N]
 * A,B,C,R in `dtype`; bias fp32 (may be NULL); R may be NULL.
 * Used for every Dense forward, its tangent (row-stacked [x; xdot] so the
 * weight tile is read once), input-gradient and
 * weight-gradient products.  `splitk` > 1 splits K over workgroups and
 * accumulates through the fp32 workspace `ws` (>= M*N floats, zeroed by the
 * call); the epilogue then runs as a second kernel. */
int mfc_gemm(int dtype, int flags, int64_t M, int64_t N, int64_t K,
             const void* A, int64_t lda, const void* B, int64_t ldb,
             void* C, int64_t ldc,
             const float* bias, int64_t bias_rows, int64_t act_rows,
             float alpha, const void* R, int64_t ldr, float beta_res,
             int splitk, float* ws, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MFC_H */
