// Dense-layer GEMM for gfx950: C = alpha*(op(A) op(B) + bias) + beta*R.
//
// Every flax.linen.Dense on the hot path (models/conv_flow.py:142-160,188-202,
// models/mlp_flow.py:18-31,83-117, models/mlp_mixer.py) and its tangent /
// input-gradient / weight-gradient products go through this kernel:
//   forward            C[R,N]  = X[R,K] W[K,N]            (NN)
//   tangent            rows stacked [x; xdot] in the same launch (bias only on
//                      the primal rows) so each weight tile is read once
//   input gradient     dX[R,K] = dY[R,N] W[K,N]^T         (NT)
//   weight gradient    dW[K,N] = X[R,K]^T dY[R,N]         (TN)
//
// 128x128x32 workgroup tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA tiles of
// 16x16.  fp32 storage -> v_mfma_f32_16x16x4_f32 (exact fp32 fma chain);
// bf16 storage -> v_mfma_f32_16x16x16_bf16.  Both read 4 contiguous k per
// lane from K-contiguous LDS tiles, so one staging/addressing scheme serves
// both.  fp32 accumulate always.
#include "mfc_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, GT = 256;
constexpr int LDS_PAD = 4;
constexpr int LDK = BK + LDS_PAD;  // elements per LDS row

struct GemmArgs {
    int64_t M, N, K;
    const void* A; int64_t lda;
    const void* B; int64_t ldb;
    void* C; int64_t ldc;
    const float* bias; int64_t bias_rows;
    float alpha;
    const void* R; int64_t ldr; float beta;
    int accum;
    int64_t kchunk;  // K range per blockIdx.z
    float* ws;       // split-K workspace (fp32 [M,N]) or null
    int vecA, vecB;  // 4-element vector loads legal
    int vecC;        // 16-byte row-contiguous C (and R) accesses legal
};

// load 4 consecutive elements p[0..3] (valid = number in range), zero fill
template <typename T>
__device__ inline void load4(const T* p, int64_t valid, bool vec, T out[4]) {
    if (vec && valid >= 4) {
        if constexpr (sizeof(T) == 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p);
            out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
        } else {
            const s16x4 v = *reinterpret_cast<const s16x4*>(p);
            out[0] = (T)v[0]; out[1] = (T)v[1]; out[2] = (T)v[2]; out[3] = (T)v[3];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = (i < valid) ? p[i] : (T)0;
    }
}

// Stage one operand tile (ROWS x BK, ROWS = 128) into registers.
//  KC = true : source is K-contiguous   src[row*ld + k]
//  KC = false: source is row-contiguous src[k*ld + row]
template <typename T, bool KC>
__device__ inline void tile_load(const T* src, int64_t ld, int64_t row0, int64_t nrows,
                                 int64_t k0, int64_t kend, bool vec, T regs[16]) {
    const int t = threadIdx.x;
    if constexpr (KC) {
        // 8 threads per row (4 k each), 32 rows per pass, 4 passes
        const int kk = (t & 7) * 4;
        const int rr = t >> 3;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int64_t row = row0 + rr + 32 * p;
            const int64_t k = k0 + kk;
            if (row < nrows && k < kend) load4<T>(src + row * ld + k, kend - k, vec, &regs[4 * p]);
            else { regs[4 * p] = regs[4 * p + 1] = regs[4 * p + 2] = regs[4 * p + 3] = (T)0; }
        }
    } else {
        // wave w covers k rows [8w, 8w+8); lane: lk = lane&7, lm = lane>>3 -> 4 rows each
        const int lane = t & 63, w = t >> 6;
        const int lk = lane & 7, lm = lane >> 3;
        const int64_t k = k0 + 8 * w + lk;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int64_t row = row0 + 32 * p + 4 * lm;
            if (k < kend && row < nrows) load4<T>(src + k * ld + row, nrows - row, vec, &regs[4 * p]);
            else { regs[4 * p] = regs[4 * p + 1] = regs[4 * p + 2] = regs[4 * p + 3] = (T)0; }
        }
    }
}

template <typename T, bool KC>
__device__ inline void tile_store(T* lds, const T regs[16]) {
    const int t = threadIdx.x;
    if constexpr (KC) {
        const int kk = (t & 7) * 4;
        const int rr = t >> 3;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            T* d = lds + (rr + 32 * p) * LDK + kk;
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<f32x4*>(d) = f32x4{regs[4 * p], regs[4 * p + 1], regs[4 * p + 2], regs[4 * p + 3]};
            } else {
                *reinterpret_cast<s16x4*>(d) = s16x4{(short)regs[4 * p], (short)regs[4 * p + 1],
                                                     (short)regs[4 * p + 2], (short)regs[4 * p + 3]};
            }
        }
    } else {
        const int lane = t & 63, w = t >> 6;
        const int lk = lane & 7, lm = lane >> 3;
        const int k = 8 * w + lk;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) lds[(32 * p + 4 * lm + i) * LDK + k] = regs[4 * p + i];
    }
}

template <typename T, bool TA, bool TB>
__global__ void __launch_bounds__(GT)
gemm_kernel(GemmArgs g) {
    // A tile: [BM][LDK], B tile: [BN][LDK], both K-contiguous in LDS
    __shared__ __attribute__((aligned(16))) T lds[(BM + BN) * LDK];
    T* As = lds;
    T* Bs = lds + BM * LDK;
    typedef typename Frag<T>::type frag_t;

    const T* A = (const T*)g.A;
    const T* B = (const T*)g.B;
    const int64_t m0 = (int64_t)blockIdx.y * BM;
    const int64_t n0 = (int64_t)blockIdx.x * BN;
    const int64_t kbeg = (int64_t)blockIdx.z * g.kchunk;
    const int64_t kend = (kbeg + g.kchunk < g.K) ? kbeg + g.kchunk : g.K;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int q = lane >> 4, r = lane & 15;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    T ra[16], rb[16];
    // A: K-contiguous when not transposed ([M][K]); B: K-contiguous when transposed ([N][K])
    tile_load<T, !TA>(A, g.lda, m0, g.M, kbeg, kend, g.vecA, ra);
    tile_load<T, TB>(B, g.ldb, n0, g.N, kbeg, kend, g.vecB, rb);

    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
        tile_store<T, !TA>(As, ra);
        tile_store<T, TB>(Bs, rb);
        __syncthreads();
        if (k0 + BK < kend) {
            tile_load<T, !TA>(A, g.lda, m0, g.M, k0 + BK, kend, g.vecA, ra);
            tile_load<T, TB>(B, g.ldb, n0, g.N, k0 + BK, kend, g.vecB, rb);
        }
#pragma unroll
        for (int c = 0; c < BK / 16; ++c) {
            frag_t af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i] = *reinterpret_cast<const frag_t*>(As + (wm + 16 * i + r) * LDK + 16 * c + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bf[j] = *reinterpret_cast<const frag_t*>(Bs + (wn + 16 * j + r) * LDK + 16 * c + 4 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16(acc[i][j], af[i], bf[j]);
        }
        __syncthreads();
    }

    // epilogue.  C layout of a 16x16 MFMA tile: col = lane&15, row = 4*(lane>>4)+reg
    T* C = (T*)g.C;
    const T* R = (const T*)g.R;
    if (!g.ws && g.vecC) {
        // Row-contiguous stores: each wave transposes its 64x64 result 16 rows at a time through a
        // private LDS slab so every lane writes 16 consecutive columns (32/64 B) of one row --
        // 16x fewer (and full-line) store instructions than storing the MFMA C layout directly.
        constexpr int CSS = 68;
        float* cs = reinterpret_cast<float*>(lds) + wave * (16 * CSS);
        const int lr = lane >> 2, lc = (lane & 3) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) cs[(4 * q + e) * CSS + 16 * j + r] = acc[i][j][e];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int64_t row = m0 + wm + 16 * i + lr;
            const int64_t col0 = n0 + wn + lc;
            float v[16];
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(cs + lr * CSS + lc + 4 * k4);
                v[4 * k4] = t[0]; v[4 * k4 + 1] = t[1]; v[4 * k4 + 2] = t[2]; v[4 * k4 + 3] = t[3];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (row < g.M && col0 < g.N) {
                const bool hb = g.bias && row < g.bias_rows;
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] = (v[k] + (hb ? g.bias[col0 + k] : 0.f)) * g.alpha;
                T* cp = C + row * g.ldc + col0;
                if constexpr (sizeof(T) == 4) {
                    if (R) {
#pragma unroll
                        for (int k4 = 0; k4 < 4; ++k4) {
                            const f32x4 t = *reinterpret_cast<const f32x4*>(R + row * g.ldr + col0 + 4 * k4);
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[4 * k4 + k] += g.beta * t[k];
                        }
                    }
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) {
                        f32x4 t = f32x4{v[4 * k4], v[4 * k4 + 1], v[4 * k4 + 2], v[4 * k4 + 3]};
                        if (g.accum) t += *reinterpret_cast<const f32x4*>(cp + 4 * k4);
                        *reinterpret_cast<f32x4*>(cp + 4 * k4) = t;
                    }
                } else {
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    if (R) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const s16x8 t = *reinterpret_cast<const s16x8*>(R + row * g.ldr + col0 + 8 * h);
#pragma unroll
                            for (int k = 0; k < 8; ++k) v[8 * h + k] += g.beta * bf16_to_f32((u16)t[k]);
                        }
                    }
                    if (g.accum) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const s16x8 t = *reinterpret_cast<const s16x8*>(cp + 8 * h);
#pragma unroll
                            for (int k = 0; k < 8; ++k) v[8 * h + k] += bf16_to_f32((u16)t[k]);
                        }
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const u32x4 t = {pack_bf16x2(v[8 * h], v[8 * h + 1]), pack_bf16x2(v[8 * h + 2], v[8 * h + 3]),
                                         pack_bf16x2(v[8 * h + 4], v[8 * h + 5]), pack_bf16x2(v[8 * h + 6], v[8 * h + 7])};
                        *reinterpret_cast<u32x4*>(cp + 8 * h) = t;
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t col = n0 + wn + 16 * j + r;
            if (col >= g.N) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t row = m0 + wm + 16 * i + 4 * q + e;
                if (row >= g.M) continue;
                float v = acc[i][j][e];
                if (g.ws) {
                    atomicAdd(g.ws + row * g.N + col, v);
                } else {
                    if (g.bias && row < g.bias_rows) v += g.bias[col];
                    v *= g.alpha;
                    if (R) v += g.beta * St<T>::ld(R + row * g.ldr + col);
                    if (g.accum) v += St<T>::ld(C + row * g.ldc + col);
                    St<T>::st(C + row * g.ldc + col, v);
                }
            }
        }
    }
}

// split-K / activation epilogue over the fp32 workspace
template <typename T>
__global__ void __launch_bounds__(256)
gemm_epilogue_kernel(const float* ws, int64_t M, int64_t N, T* C, int64_t ldc,
                     const float* bias, int64_t bias_rows, int gelu, int64_t act_rows,
                     float alpha, const T* R, int64_t ldr, float beta, int accum) {
    const int64_t total = M * N;
    for (int64_t o = blockIdx.x * 256LL + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t row = o / N, col = o - row * N;
        float v = ws[o];
        if (bias && row < bias_rows) v += bias[col];
        if (gelu) {
            if (row < act_rows) {
                v = gelu_f(v);
            } else {
                // tangent row: t * gelu'(pre) with pre = primal pre-activation
                const int64_t pr = row - act_rows;
                float pre = ws[pr * N + col];
                if (bias && pr < bias_rows) pre += bias[col];
                v = v * gelu_grad_f(pre);
            }
        }
        v *= alpha;
        if (R) v += beta * St<T>::ld(R + row * ldr + col);
        if (accum) v += St<T>::ld(C + row * ldc + col);
        St<T>::st(C + row * ldc + col, v);
    }
}

template <typename T>
int launch(int flags, const GemmArgs& g, int splitk, int gelu, int64_t act_rows, hipStream_t st) {
    dim3 grid((unsigned)ceil_div64(g.N, BN), (unsigned)ceil_div64(g.M, BM), (unsigned)splitk);
    const bool ta = flags & MFC_GEMM_TRANS_A, tb = flags & MFC_GEMM_TRANS_B;
    if (!ta && !tb) hipLaunchKernelGGL((gemm_kernel<T, false, false>), grid, dim3(GT), 0, st, g);
    else if (!ta && tb) hipLaunchKernelGGL((gemm_kernel<T, false, true>), grid, dim3(GT), 0, st, g);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_kernel<T, true, false>), grid, dim3(GT), 0, st, g);
    else hipLaunchKernelGGL((gemm_kernel<T, true, true>), grid, dim3(GT), 0, st, g);
    int rc = mfc_launch_status();
    if (rc) return rc;
    if (g.ws) {
        int64_t blocks = ceil_div64(g.M * g.N, 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL((gemm_epilogue_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st,
                           (const float*)g.ws, g.M, g.N, (T*)g.C, g.ldc, g.bias, g.bias_rows, gelu, act_rows,
                           g.alpha, (const T*)g.R, g.ldr, g.beta, g.accum);
        rc = mfc_launch_status();
    }
    return rc;
}

}  // namespace

extern "C" int mfc_gemm(int dtype, int flags, int64_t M, int64_t N, int64_t K,
                        const void* A, int64_t lda, const void* B, int64_t ldb,
                        void* C, int64_t ldc,
                        const float* bias, int64_t bias_rows, int64_t act_rows,
                        float alpha, const void* R, int64_t ldr, float beta_res,
                        int splitk, float* ws, void* stream) {
    if (!A || !B || !C) return MFC_EFAULT;
    if (M <= 0 || N <= 0 || K <= 0) return MFC_EINVAL;
    if (dtype != MFC_F32 && dtype != MFC_BF16) return MFC_EINVAL;
    const bool ta = flags & MFC_GEMM_TRANS_A, tb = flags & MFC_GEMM_TRANS_B;
    if (lda < (ta ? M : K) || ldb < (tb ? K : N) || ldc < N) return MFC_EINVAL;
    if (R && ldr < N) return MFC_EINVAL;
    const int gelu = (flags & MFC_GEMM_GELU) ? 1 : 0;
    if (splitk < 1) splitk = 1;
    if ((splitk > 1 || gelu) && !ws) return MFC_EINVAL;
    if (gelu && (act_rows <= 0 || act_rows > M)) return MFC_EINVAL;
    if (ceil_div64(M, BM) > 65535 || splitk > 65535) return MFC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const size_t es = dtype == MFC_F32 ? 4 : 2;

    GemmArgs g;
    g.M = M; g.N = N; g.K = K;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.bias = bias; g.bias_rows = bias_rows;
    g.alpha = alpha; g.R = R; g.ldr = ldr; g.beta = beta_res;
    g.accum = (flags & MFC_GEMM_ACCUM) ? 1 : 0;
    int64_t kc = ceil_div64(ceil_div64(K, splitk), BK) * BK;
    g.kchunk = kc;
    splitk = (int)ceil_div64(K, kc);
    const bool use_ws = (splitk > 1) || gelu;
    g.ws = use_ws ? ws : nullptr;
    // 4-element vector loads need the run start 4-element aligned for every row
    g.vecA = ((lda % 4) == 0) && (((uintptr_t)A % (4 * es)) == 0);
    g.vecB = ((ldb % 4) == 0) && (((uintptr_t)B % (4 * es)) == 0);
    g.vecC = (N % 16 == 0) && ((ldc * es) % 16 == 0) && (((uintptr_t)C % 16) == 0) &&
             (!R || (((ldr * es) % 16 == 0) && (((uintptr_t)R % 16) == 0)));
    if (use_ws) {
        if (hipMemsetAsync(ws, 0, (size_t)M * N * sizeof(float), st) != hipSuccess) return MFC_EHIP;
    }
    return dtype == MFC_F32 ? launch<float>(flags, g, splitk, gelu, act_rows, st)
                            : launch<u16>(flags, g, splitk, gelu, act_rows, st);
}
