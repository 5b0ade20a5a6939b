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
// 128x128xBK workgroup tile (BK = 64; 32 for tiny K), 4 waves (2x2), each wave 64x64 = 4x4 MFMA tiles of
// 16x16.  fp32 storage -> v_mfma_f32_16x16x4_f32 (exact fp32 fma chain);
// bf16 storage -> v_mfma_f32_16x16x16_bf16.  Both read 4 contiguous k per
// lane from K-contiguous LDS tiles, so one staging/addressing scheme serves
// both.  fp32 accumulate always.
#include "mfc_common.h"
#include <cmath>
#include <cstdlib>

namespace {

constexpr int BN = 128, GT = 256;   // BM is a template parameter (128, or 64 for remainder rows)
constexpr int ldk_of(int BK, int es) { return BK + 16 / es; }   // K-contiguous tile [128][LDK]: +16 B pad
constexpr int ldr_of(int es) { return es == 4 ? 132 : 144; }     // row-contiguous tile [BK][LDR]
constexpr int ldr_of_rows(int rows, int es) { return rows <= 128 ? ldr_of(es) : rows + 16; }   // [BK][rows + pad]
constexpr int tile_elems(int BK, int es, int rows = 128) {
    return rows * ldk_of(BK, es) > BK * ldr_of_rows(rows, es) ? rows * ldk_of(BK, es) : BK * ldr_of_rows(rows, es);
}

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
    float* ws;       // split-K workspace (fp32 [slices][M,N], one slab per K slice) or null
    int64_t ws_slab; // elements per slab (M * N)
    int vecA, vecB;  // 16-byte vector loads legal
    int vecC;        // 16-byte row-contiguous C (and R) accesses legal
    int m_fast;      // blockIdx.x walks M tiles (else N tiles)
    float* ln_rstd;  // MFC_GEMM_LN16: per-(row, 16-column group) 1/sigma, or null
    int64_t m_base;  // first row handled by this launch (rows below belong to another launch)
    // mfc_gemm_adamw: the product is a weight gradient and the epilogue is the AdamW update of that weight
    float* opt_p; float* opt_m; float* opt_v;   // fp32 master and moments, dense [M, N]; C = the bf16 working copy
    float lr, b1, b2, eps, wd, bc1, bc2;
    // mfc_gemm_adamw, optional: column sums of the (non-transposed, bf16) B operand, colsum[n] = colsum_scale * sum_k B[k][n]
    // -- the bias gradient that belongs to this weight gradient (B = dY): the operand is in LDS anyway, so the separate
    // pass over dY (1.6 GB for the [128, S] kernels) disappears.  Written by the workgroups of the first M tile only.
    float* colsum; float colsum_scale;
};
// fp32 fast staging (gemm_f32_fast_kernel): byte sizes of the A / B operands for their buffer resources; separate kernel
// arguments, so that the argument block of every other kernel stays as it was (the bf16 kernels sit at the SGPR / VGPR edge)
struct GemmFast { uint32_t a_bytes, b_bytes; int on; };

// raw buffer resource (base, num_records bytes): out-of-range lanes load 0 / drop their store in hardware
__device__ inline __amdgpu_buffer_rsrc_t gemm_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

template <typename T> struct Vec;   // 16-byte global vector
template <> struct Vec<float> { typedef f32x4 type; static constexpr int W = 4; };
template <> struct Vec<u16> { typedef s16x8 type; static constexpr int W = 8; };

// load one 16-byte vector of W consecutive elements (valid = number in range), zero fill
template <typename T>
__device__ inline typename Vec<T>::type loadv(const T* p, int64_t valid, bool vec) {
    constexpr int W = Vec<T>::W;
    typename Vec<T>::type v;
    if (vec && valid >= W) {
        v = *reinterpret_cast<const typename Vec<T>::type*>(p);
    } else {
#pragma unroll
        for (int i = 0; i < W; ++i) v[i] = (i < valid) ? p[i] : (T)0;
    }
    return v;
}

// Stage one operand tile (128 rows x BK) into registers / from registers into LDS.  The staging
// registers stay packed 16-byte vectors (bf16: 8 elements in 4 VGPRs).
//  KC = true : source is K-contiguous   src[row*ld + k]   -> LDS tile [128][LDK]
//  KC = false: source is row-contiguous src[k*ld + row]   -> LDS tile [BK][LDR] (no transpose on the way in)
template <typename T, int BK, bool KC, int ROWS = 128> struct Stage {
    typedef typename Vec<T>::type vec_t;
    static constexpr int W = Vec<T>::W;
    static constexpr int LDK = ldk_of(BK, sizeof(T));
    static constexpr int LDR = ldr_of_rows(ROWS, sizeof(T));
    static constexpr int TPR = BK / W, RPP = GT / TPR, NP = (ROWS + RPP - 1) / RPP;    // KC geometry
    static constexpr int TPK = ROWS / W, KPP = GT / TPK, NPT = (BK + KPP - 1) / KPP;  // !KC geometry
    static constexpr int NV = KC ? NP : NPT;                                          // vectors per thread

    __device__ static inline void load(const T* src, int64_t ld, int64_t row0, int64_t nrows, int64_t k0,
                                       int64_t kend, bool vec, vec_t* regs) {
        const int t = threadIdx.x;
        vec_t z;
#pragma unroll
        for (int i = 0; i < W; ++i) z[i] = (T)0;
        if constexpr (KC) {
            const int kk = (t % TPR) * W, rr = t / TPR;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int64_t row = row0 + rr + RPP * p, k = k0 + kk;
                regs[p] = (rr + RPP * p < ROWS && row < nrows && k < kend) ? loadv<T>(src + row * ld + k, kend - k, vec) : z;
            }
        } else {
            const int rr = (t % TPK) * W, kk = t / TPK;
#pragma unroll
            for (int p = 0; p < NPT; ++p) {
                const int64_t k = k0 + kk + KPP * p, row = row0 + rr;
                regs[p] = (kk + KPP * p < BK && k < kend && row < nrows) ? loadv<T>(src + k * ld + row, nrows - row, vec) : z;
            }
        }
    }
    // Branch-free staging through a buffer resource over the whole operand (fp32 compute-bound products whose K range is
    // whole K-steps): every vector is one unconditional 16-byte buffer load -- the thread's part of the address is one
    // loop-invariant VGPR, the step's part a scalar offset.  Rows past the operand's last row load zeros (or, on the
    // contiguous axis of a row-contiguous operand, the neighbouring row's finite values): they only feed rows / columns of C
    // that are never stored.  The guarded form above costs these kernels ~13 % of their time.
    __device__ static inline void load_buf(__amdgpu_buffer_rsrc_t rs, int64_t ld, int64_t row0, int64_t k0, vec_t* regs) {
        const int t = threadIdx.x;
        typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
        if constexpr (KC) {
            const int kk = (t % TPR) * W, rr = t / TPR;
            const uint32_t voff = (uint32_t)((rr * ld + kk) * sizeof(T));
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const uint32_t soff = (uint32_t)(((row0 + RPP * p) * ld + k0) * sizeof(T));
                regs[p] = __builtin_bit_cast(vec_t, (u32x4_)__builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
            }
        } else {
            const int rr = (t % TPK) * W, kk = t / TPK;
            const uint32_t voff = (uint32_t)((kk * ld + rr) * sizeof(T));
#pragma unroll
            for (int p = 0; p < NPT; ++p) {
                const uint32_t soff = (uint32_t)(((k0 + KPP * p) * ld + row0) * sizeof(T));
                regs[p] = __builtin_bit_cast(vec_t, (u32x4_)__builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
            }
        }
    }
    __device__ static inline void store(T* lds, const vec_t* regs) {
        const int t = threadIdx.x;
        if constexpr (KC) {
            const int kk = (t % TPR) * W, rr = t / TPR;
#pragma unroll
            for (int p = 0; p < NP; ++p)
                if (rr + RPP * p < ROWS) *reinterpret_cast<vec_t*>(lds + (rr + RPP * p) * LDK + kk) = regs[p];
        } else {
            const int rr = (t % TPK) * W, kk = t / TPK;
#pragma unroll
            for (int p = 0; p < NPT; ++p)
                if (kk + KPP * p < BK) *reinterpret_cast<vec_t*>(lds + (kk + KPP * p) * LDR + rr) = regs[p];
        }
    }

    // MFMA operand fragment of tile row (rowbase + r), k = 16c + 4q .. +3
    __device__ static inline typename Frag<T>::type fetch(const T* S, int rowbase, int c, int q, int r) {
        typedef typename Frag<T>::type frag_t;
        if constexpr (KC) {
            return *reinterpret_cast<const frag_t*>(S + (rowbase + r) * LDK + 16 * c + 4 * q);
        } else if constexpr (sizeof(T) == 4) {
            const T* p = S + (16 * c + 4 * q) * LDR + rowbase + r;
            return frag_t{p[0], p[LDR], p[2 * LDR], p[3 * LDR]};
        } else {
            // gfx950 LDS transpose read: the 16 lanes of group q fetch the 4(k) x 16(row) block and each
            // receives one row's 4 consecutive k -- lane 4q'+p supplies the address of k-row q', rows 4p..4p+3
            const T* p = S + (16 * c + 4 * q + (r >> 2)) * LDR + rowbase + 4 * (r & 3);
            return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
        }
    }
};

// store 16 consecutive columns of one row: v already holds alpha*(acc + bias); adds beta*R and the
// ACCUM read-modify-write, writes one (fp32: four) 16-byte vector(s)
template <typename T>
__device__ inline void store_row16(const GemmArgs& g, int64_t row, int64_t col0, float v[16]) {
    T* cp = (T*)g.C + row * g.ldc + col0;
    const T* R = (const T*)g.R;
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

// AdamW on 16 consecutive elements of one row (same arithmetic, in the same order, as adamw_vec_kernel with a bf16
// gradient: the accumulator is rounded to bf16 first so the fused and the two-kernel paths give identical bits);
// on return v holds the new parameters (the caller stores them as the bf16 working copy)
// AdamW epilogue of one 16-row x 64-column wave tile: pass k4 handles rows 4*k4 + (lane >> 4), and the 16 lanes of a
// row take 4 consecutive columns each, so every load / store instruction of the wave moves four 256-byte row
// segments of p / m / v (a lane owning 16 consecutive columns would scatter 16-byte pieces 64 B apart, and the weights
// whose rows are megabytes apart -- [128, S] -- want few, long segments).  gv[4*k4 + k]: gradient of row 4*k4 + (lane >> 4),
// column 4 * (lane & 15) + k of the tile.
template <typename T, bool GUARD>
__device__ inline void adamw_tile_16x64(const GemmArgs& g, int64_t row0, int64_t col0, int lane, float gv[16]) {
    const int64_t cb = col0 + 4 * (lane & 15);
    f32x4 p0[4], m0[4], v0[4];
    bool in[4];
    int64_t o[4];
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {      // all twelve loads in flight before the first store (which may alias, for all
        const int64_t row = row0 + 4 * k4 + (lane >> 4);   // the compiler knows, and would serialise a load-update-store loop)
        o[k4] = row * g.N + cb;
        in[k4] = !GUARD || (row < g.M && cb < g.N);
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
        p0[k4] = in[k4] ? *reinterpret_cast<const f32x4*>(g.opt_p + o[k4]) : z;
        m0[k4] = in[k4] ? *reinterpret_cast<const f32x4*>(g.opt_m + o[k4]) : z;
        v0[k4] = in[k4] ? *reinterpret_cast<const f32x4*>(g.opt_v + o[k4]) : z;
    }
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pk = p0[k4][k], mk = m0[k4][k], vk = v0[k4][k];
            adamw_elem(pk, mk, vk, bf16_to_f32(f32_to_bf16(gv[4 * k4 + k])), g.lr, g.b1, g.b2, g.eps, g.wd, g.bc1, g.bc2);
            p0[k4][k] = pk; m0[k4][k] = mk; v0[k4][k] = vk;
        }
    }
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
        if (!in[k4]) continue;
        *reinterpret_cast<f32x4*>(g.opt_m + o[k4]) = m0[k4];
        *reinterpret_cast<f32x4*>(g.opt_v + o[k4]) = v0[k4];
        *reinterpret_cast<f32x4*>(g.opt_p + o[k4]) = p0[k4];
        if constexpr (sizeof(T) == 2) {
            const int64_t row = row0 + 4 * k4 + (lane >> 4);
            *reinterpret_cast<u32x2*>((T*)g.C + row * g.ldc + cb) = u32x2{pack_bf16x2(p0[k4][0], p0[k4][1]), pack_bf16x2(p0[k4][2], p0[k4][3])};
        }
    }
}

// LayerNorm (no affine, eps 1e-6) over the 16 values of one pixel held by one lane
// (packed f32 VALU on register pairs: v_pk_add / v_pk_fma / v_pk_mul -- two values per issue slot.  These epilogues run in
// kernels whose instruction stream, not the memory system, sets the pace: the N-streaming product with the fused LayerNorm
// measured 1.12 vs 0.95 ms without it at M = 192 before this form.)
__device__ inline f32x2 pk2(float a, float b) { return f32x2{a, b}; }
__device__ inline float ln16_lane(float v[16]) {
    f32x2 p[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = pk2(v[2 * k], v[2 * k + 1]);
    const f32x2 s01 = (p[0] + p[1]) + (p[2] + p[3]), s23 = (p[4] + p[5]) + (p[6] + p[7]);
    const f32x2 s2 = s01 + s23;
    f32x2 q2 = p[0] * p[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) q2 = __builtin_elementwise_fma(p[k], p[k], q2);
    const float mean = (s2[0] + s2[1]) * (1.0f / 16.0f);
    const float sq = q2[0] + q2[1];
    const float rho = __builtin_amdgcn_rsqf(fmaxf(0.0f, sq * (1.0f / 16.0f) - mean * mean) + 1e-6f);   // argument >= 1e-6: no denormal path
    const f32x2 r2 = pk2(rho, rho), c2 = pk2(-mean * rho, -mean * rho);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const f32x2 n = __builtin_elementwise_fma(p[k], r2, c2);     // (v - mean) rho
        v[2 * k] = n[0]; v[2 * k + 1] = n[1];
    }
    return rho;
}
// its tangent: nd = rho (xd - mean(xd) - n mean(n (xd - mean(xd))))   (n, rho: the primal's output)
__device__ inline void ln16_tangent_lane(float xd[16], const float n[16], float rho) {
    f32x2 x[8], nn[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { x[k] = pk2(xd[2 * k], xd[2 * k + 1]); nn[k] = pk2(n[2 * k], n[2 * k + 1]); }
    const f32x2 s2 = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
    const float md = (s2[0] + s2[1]) * (1.0f / 16.0f);
    const f32x2 m2 = pk2(md, md);
    f32x2 d2 = pk2(0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k) { x[k] = x[k] - m2; d2 = __builtin_elementwise_fma(nn[k], x[k], d2); }
    const float dot = (d2[0] + d2[1]) * (1.0f / 16.0f);
    const f32x2 r2 = pk2(rho, rho), e2 = pk2(-dot * rho, -dot * rho);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const f32x2 o = __builtin_elementwise_fma(nn[k], e2, x[k] * r2);      // rho (xd_c - n dot)
        xd[2 * k] = o[0]; xd[2 * k + 1] = o[1];
    }
}

template <typename T, int BK, bool TA, bool TB, int BMT>
__global__ void __launch_bounds__(GT)
gemm_kernel(GemmArgs g) {
    constexpr bool FAST = false;
    constexpr uint32_t a_bytes = 0u, b_bytes = 0u;
#include "gemm_tile_body.inc"
}
// the same tile loop with branch-free buffer-resource staging (fp32 compute-bound products, see Stage::load_buf)
template <bool TA, bool TB, int BMT>
__global__ void __launch_bounds__(GT)
gemm_f32_fast_kernel(GemmArgs g, uint32_t a_bytes, uint32_t b_bytes) {
    typedef float T;
    constexpr int BK = 64;
    constexpr bool FAST = true;
#include "gemm_tile_body.inc"
}
// the same for the bf16 products without an optimizer epilogue (the K = S / K = D products of the ConvFlow block and, in the
// data-parallel schedule, the un-fused weight gradients: 3-8 % on those HBM-bound launches, tools/bench_gemm.py)
template <bool TA, bool TB, int BMT>
__global__ void __launch_bounds__(GT)
gemm_bf16_fast_kernel(GemmArgs g, uint32_t a_bytes, uint32_t b_bytes) {
    typedef u16 T;
    constexpr int BK = 64;
    constexpr bool FAST = true;
#include "gemm_tile_body.inc"
}

// ---------------------------------------------------------------------------
// N-streaming variant for the skinny products of the ConvNeXt flow (bf16, NN, K = 128, M <= 256,
// N in the millions): the whole A operand lives in registers as MFMA fragments for the lifetime of a
// persistent workgroup, which walks 64-column tiles of B / C.  Every byte of B is read from HBM
// exactly once (the tiled kernel re-reads the weight tile once per 128-row M tile), the next tile's
// loads are in flight during the MFMAs and the epilogue of the current one, and -- because one wave
// holds a primal 16-row tile and the tangent tile of the same samples -- MFC_GEMM_LN16T can apply
// the tangent of the fused LayerNorm in the same epilogue.
// ---------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ inline void mma32(f32x4& acc, const s16x8& a, const s16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
constexpr int NS_BN = 64, NS_K = 128, NS_KS = NS_K / 16, NS_LDR = 80, NS_MAXT = 4;
#ifndef MFC_NS_CT
#define MFC_NS_CT 1
#endif
#ifndef MFC_NS_SWZ
#define MFC_NS_SWZ 1
#endif
// NN weight tile in LDS.  MFC_NS_SWZ: [128 k][64 n] rows of 128 bytes without padding, the 8-byte piece p of row k kept at
// piece p ^ ((k >> 1) & 3): the 32 lanes ds_read_b64_tr_b16 serves per LDS cycle (8 k-rows x 4 pieces 32 bytes apart) then
// start on 32 different bank pairs.  The padded image (rows of 160 bytes) put them on 8: four cycles per half instead of one
// (SQ_LDS_BANK_CONFLICT: 200 of ~260 LDS cycles per wave and 64-column step).
constexpr int NS_LDB = MFC_NS_SWZ ? NS_BN : NS_LDR;
struct NsPlan {
    // per wave: up to 4 16-row tiles of C, processed in order; kind 0 = plain, 1 = LN16 primal,
    // 2 = LN16 tangent of the entry before it
    signed char tile[4][NS_MAXT];
    signed char kind[4][NS_MAXT];
};

// raw buffer resources (base, num_records bytes): out-of-range lanes load 0 / drop their store in
// hardware, so every VMEM instruction of the loop is unconditional and the compiler can count them
// (s_waitcnt vmcnt(n) for the prefetched B tile instead of draining the epilogue stores too)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
constexpr uint32_t NS_OOB = 0xFFFFFF00u;

// NT = false: B is [K, N] row-major (the forward products: 128-byte row segments of 128 weight rows per tile, read back
// with LDS transpose reads).  NT = true: B is [N, K] row-major (the dX products against a [S, 128] / [D, 128] kernel): a
// tile is 64 consecutive rows of 256 bytes = one contiguous 16 KB burst, kept as [n][k] in LDS and read with plain
// ds_read_b128.
template <int MT, bool NT>
__global__ void __launch_bounds__(GT, MT == 3 ? 3 : 1)      // M = 192 (MT = 3): <= 168 VGPRs = three workgroups per CU
gemm_nstream_kernel(GemmArgs g, NsPlan plan, int64_t ntiles) {
    typedef u16 T;
    typedef Frag<T>::type frag_t;
    constexpr int NS_LDN = NS_K + 8;        // NT: [64 n][128 k] rows padded by 16 bytes
    __shared__ __attribute__((aligned(16))) T Bs[NT ? NS_BN * NS_LDN : NS_K * NS_LDB];
    __shared__ __attribute__((aligned(16))) float bias_s[NS_BN];
#if MFC_NS_CT
    constexpr int NS_LDC = NS_BN + 8;       // wave-private image of a 16 x 64 C tile, rows padded by 16 bytes
    __shared__ __attribute__((aligned(16))) T Cs[4 * 16 * NS_LDC];
#endif
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, r = lane & 15;
    const T* A = (const T*)g.A;
    const T* B = (const T*)g.B;
    const int64_t N = g.N;

    int mt[MT], kind[MT];
    // K = 32 MFMA operands (v_mfma_f32_16x16x32_bf16: half the instructions of the K = 16 form for the same matrix-pipe
    // cycles, i.e. half the time the MFMAs hold the SIMD's issue port).  Which k a (lane, slot) pair carries is free as
    // long as both operands agree: slots 0-3 = k 32c + 4q .. +3, slots 4-7 = k 32c + 16 + 4q .. +3 -- two K = 16
    // fragments side by side, so the weight side stays two ds_read_b64_tr_b16 and nothing is shuffled.
    typedef s16x8 frag8_t;
    frag8_t af[MT][NS_KS / 2];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        mt[i] = plan.tile[wave][i];
        kind[i] = plan.kind[wave][i];
        const int64_t row = 16 * (int64_t)(mt[i] < 0 ? 0 : mt[i]) + r;
#pragma unroll
        for (int c = 0; c < NS_KS / 2; ++c) {
            if constexpr (NT) {      // k of a (lane, slot): 32c + 8q + slot on both operands (one 16-byte piece each)
                af[i][c] = frag8_t{0, 0, 0, 0, 0, 0, 0, 0};
                if (mt[i] >= 0 && row < g.M) af[i][c] = *reinterpret_cast<const frag8_t*>(A + row * g.lda + 32 * c + 8 * q);
            } else {
                frag_t lo = frag_t{0, 0, 0, 0}, hi = frag_t{0, 0, 0, 0};
                if (mt[i] >= 0 && row < g.M) {
                    lo = *reinterpret_cast<const frag_t*>(A + row * g.lda + 32 * c + 4 * q);
                    hi = *reinterpret_cast<const frag_t*>(A + row * g.lda + 32 * c + 16 + 4 * q);
                }
                af[i][c] = frag8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        }
    }

    // land the A fragments here: left pending, their wait would sit at the first MFMA inside the loop and
    // (vmcnt being one in-order counter) drain the previous iteration's stores every time round
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int c = 0; c < NS_KS / 2; ++c) asm volatile("" ::"v"(af[i][c]));

    // B tile staging: 128 rows x 8 chunks of 8 columns, 4 chunks per thread (k rows 32 apart).  One VGPR offset each for
    // the global and the LDS side: the other three chunks are reached through the scalar offset of the buffer load and
    // the immediate offset of the LDS store (six VGPRs less: the M = 192 variant drops from 171 to <= 168 registers,
    // i.e. from two to three workgroups per CU).
    const int ch0 = threadIdx.x;
    const int k0 = NT ? (ch0 >> 4) : (ch0 >> 3), c80 = NT ? (ch0 & 15) * 8 : (ch0 & 7) * 8;   // NT: k0 = the tile row n
    const uint32_t boff0 = (uint32_t)((k0 * g.ldb + c80) * 2);
    const uint32_t bstep = (uint32_t)((NT ? 16 : 32) * g.ldb * 2);          // rows per chunk index p: 256 threads / chunks per row
    constexpr int LDS_PSTEP = NT ? 16 * NS_LDN : 32 * NS_LDB;
    const int lds_off0 = k0 * (NT ? NS_LDN : NS_LDB) + c80;
    constexpr bool SWZ = !NT && MFC_NS_SWZ;
    // swizzled image: the chunk's two 8-byte pieces 2 (ch0 & 7) + h go to pieces (2 (ch0 & 7) + h) ^ g of their row, g = (k0 >> 1) & 3
    // (rows 32 apart share g)
    const int lds_sw = k0 * NS_LDB + 4 * ((2 * (ch0 & 7)) ^ ((k0 >> 1) & 3));
    u32x4 rb[4];
    auto commit_b = [&]() {
        // (the second piece's offset is recomputed here, from a copy the compiler cannot fold: one address register live
        // across the loop instead of two -- the M = 192 instantiation sits at its 168-register edge)
        int lds_sw0 = lds_sw;
        asm volatile("" : "+v"(lds_sw0));
        const int lds_sw1 = lds_sw0 ^ 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if constexpr (SWZ) {
                typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u32x2_t*>(Bs + lds_sw0 + p * LDS_PSTEP) = u32x2_t{rb[p][0], rb[p][1]};
                *reinterpret_cast<u32x2_t*>(Bs + lds_sw1 + p * LDS_PSTEP) = u32x2_t{rb[p][2], rb[p][3]};
            } else {
                *reinterpret_cast<u32x4*>(Bs + lds_off0 + p * LDS_PSTEP) = rb[p];
            }
        }
    };
    auto load_b = [&](int64_t tile) {
        const int64_t n0 = tile * NS_BN;
        // NN: columns >= N of rows < K-1 alias the next row (finite weights, results never stored); the last row is clipped.
        // NT: rows >= N are past the end of the resource (zeros).
        const __amdgpu_buffer_rsrc_t rs = NT ? make_rsrc(B + n0 * g.ldb, (uint32_t)(((N - n0 < NS_BN ? N - n0 : NS_BN) - 1) * g.ldb + NS_K) * 2)
                                             : make_rsrc(B + n0, (uint32_t)(((NS_K - 1) * g.ldb + (N - n0)) * 2));
#pragma unroll
        for (int p = 0; p < 4; ++p) rb[p] = __builtin_amdgcn_raw_buffer_load_b128(rs, boff0, p * bstep, 0);
    };

    // The products are issued TRANSPOSED (C^T tile = B^T A^T: the weight tile is the MFMA A operand, the resident
    // activation fragments the B operand) with the 16 rows of the n-tile j mapped to the columns
    // 16 (rho >> 2) + 4 j + (rho & 3): lane (q, r) then ends up with C[row r][16 q .. 16 q + 15] of the 64-column
    // tile in acc[.][j][e] -- a whole LayerNorm group and one 32-byte store per lane, with no LDS transpose.
    const int lr = r, lc = 16 * q;
    const int rd_sw = (4 * q + (r >> 2)) * NS_LDB + 16 * (r & 3) + 4 * ((2 * q + (r >> 3)) & 3);   // element offset of piece 4 (r & 3) + (0 ^ g)
    const bool has_alpha = g.alpha != 1.0f;
    const __amdgpu_buffer_rsrc_t rs_bias = make_rsrc(g.bias, g.bias ? (uint32_t)(N * 4) : 0u);
    // Loop shape: the B tile of iteration t+1 is requested at the top of iteration t and committed to LDS at its
    // bottom, so the wait sits in the same iteration as the request and only has to skip the VMEM instructions
    // issued in between (the epilogue stores) -- it never drains them.  The bias loads go first: vmcnt is in-order,
    // waiting for anything younger than the B request would wait for the B tile as well.
    int64_t tile = blockIdx.x;
    if (tile < ntiles) {
        load_b(tile);
        commit_b();
    }
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x) {
        const int64_t col0 = tile * NS_BN + lc;
        const bool colok = col0 < N;
        // the tile's 64 bias values: 16 lanes x 16 bytes, parked in LDS after the MFMAs (no registers held across them)
        const u32x4 bias_v = __builtin_amdgcn_raw_buffer_load_b128(
            rs_bias, (threadIdx.x < 16 && tile * NS_BN + 4 * threadIdx.x < N) ? (uint32_t)((tile * NS_BN + 4 * threadIdx.x) * 4) : NS_OOB, 0, 0);
        __builtin_amdgcn_sched_barrier(0);   // keep it OLDER than the B request below (in-order vmcnt)
        // prefetch (past the end: re-request this tile, never committed -- keeps the instruction unconditional)
        {
            const int64_t nt = tile + gridDim.x < ntiles ? tile + gridDim.x : tile;
            load_b(nt);
        }

        f32x4 acc[MT][4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NS_KS / 2; ++c) {
            frag8_t bf[4];
            int rd_c = rd_sw;       // (a copy per k-step the compiler cannot fold: the four piece offsets rd_c ^ 4 j live for one step)
            if constexpr (SWZ) asm volatile("" : "+v"(rd_c));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (NT) {
                    // MFMA row rho = r of n-tile j is tile column 16 (r >> 2) + 4 j + (r & 3) (see below): that row of the
                    // [n][k] image, k = 32c + 8q .. +7
                    bf[j] = *reinterpret_cast<const frag8_t*>(Bs + (16 * (r >> 2) + 4 * j + (r & 3)) * NS_LDN + 32 * c + 8 * q);
                } else {
                    // LDS transpose reads: lane 4q'+p of group q supplies k-row 4q+q', columns 16p + 4j .. +3
                    // (swizzled: piece 4 (r & 3) + j of row k sits at piece (4 (r & 3) + j) ^ g(k); g does not depend on c or on the +16 rows)
                    const T* bp = SWZ ? Bs + (32 * c) * NS_LDB + (rd_c ^ (4 * j))
                                      : Bs + (32 * c + 4 * q + (r >> 2)) * NS_LDB + 16 * (r & 3) + 4 * j;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(bp));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(bp + 16 * NS_LDB));
                    bf[j] = frag8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma32(acc[i][j], bf[j], af[i][c]);
        }
        if (threadIdx.x < 16) *reinterpret_cast<u32x4*>(bias_s + 4 * threadIdx.x) = bias_v;
        __syncthreads();   // B tile consumed: the next iteration may overwrite it while slower waves are in the epilogue

        // epilogue: one lane = 16 consecutive columns of one row, straight from the accumulators
        uint32_t npr[8];       // the normalised primal of a (primal, tangent) tile pair, as stored (bf16 pairs)
        float rho_pr = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) npr[k] = 0u;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * j + e] = acc[i][j][e];
            // wave-uniform geometry of this 16-row tile (a wave without an i-th tile gets an empty resource)
            const int64_t row0 = 16 * (int64_t)(mt[i] < 0 ? 0 : mt[i]);
            const int64_t rows_valid = mt[i] < 0 ? 0 : (g.M - row0 < 16 ? g.M - row0 : 16);
            const int64_t rows_bias = g.bias_rows - row0;      // rows [0, rows_bias) of the tile take the bias
            if (rows_bias > 0) {
                const bool hb = lr < rows_bias;
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_s + lc + 4 * k4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[4 * k4 + k] += hb ? b4[k] : 0.f;
                }
            }
            if (has_alpha) {
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] *= g.alpha;
            }
            if (kind[i] == 1) {
                rho_pr = ln16_lane(v);
#pragma unroll
                for (int k = 0; k < 8; ++k) npr[k] = pack_bf16x2(v[2 * k], v[2 * k + 1]);
            } else if (kind[i] == 2) {
                float n[16];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    n[2 * k] = __builtin_bit_cast(float, (uint32_t)(npr[k] << 16));
                    n[2 * k + 1] = __builtin_bit_cast(float, (uint32_t)(npr[k] & 0xffff0000u));
                }
                ln16_tangent_lane(v, n, rho_pr);
            }
            {   // 1/sigma of the LN16 primal rows (empty resource otherwise)
                const uint32_t bytes = kind[i] == 1 ? (uint32_t)(rows_valid * (N >> 4) * 4) : 0u;
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(g.ln_rstd ? g.ln_rstd + row0 * (N >> 4) : nullptr, g.ln_rstd ? bytes : 0u);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, rho_pr), rs,
                                                      colok ? (uint32_t)((lr * (N >> 4) + (col0 >> 4)) * 4) : NS_OOB, 0, 0);
            }
            const uint32_t cbytes = rows_valid > 0 ? (uint32_t)(((rows_valid - 1) * g.ldc + N) * 2) : 0u;
            const uint32_t coff = colok ? (uint32_t)((lr * g.ldc + col0) * 2) : NS_OOB;
            if (g.R) {
                const uint32_t rbytes = rows_valid > 0 ? (uint32_t)(((rows_valid - 1) * g.ldr + N) * 2) : 0u;
                const __amdgpu_buffer_rsrc_t rs = make_rsrc((const T*)g.R + row0 * g.ldr, rbytes);
                const uint32_t roff = colok ? (uint32_t)((lr * g.ldr + col0) * 2) : NS_OOB;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, roff + 16 * h, 0, 0);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[8 * h + 2 * k] += g.beta * __builtin_bit_cast(float, (uint32_t)(t[k] << 16));
                        v[8 * h + 2 * k + 1] += g.beta * __builtin_bit_cast(float, (uint32_t)(t[k] & 0xffff0000u));
                    }
                }
            }
            const __amdgpu_buffer_rsrc_t rs_c = make_rsrc((T*)g.C + row0 * g.ldc, cbytes);
            if (g.accum) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_c, coff + 16 * h, 0, 0);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[8 * h + 2 * k] += __builtin_bit_cast(float, (uint32_t)(t[k] << 16));
                        v[8 * h + 2 * k + 1] += __builtin_bit_cast(float, (uint32_t)(t[k] & 0xffff0000u));
                    }
                }
            }
#if MFC_NS_CT
            {
                // a lane holds 32 bytes of ONE row (16 rows per 16 lanes): stored like that, every store instruction is 64
                // separate 16-byte pieces for L2 to merge.  Through a wave-private LDS image the same tile leaves as two
                // instructions of eight whole 128-byte lines each (8 lanes per line).
                T* cw = Cs + wave * (16 * NS_LDC);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u32x4 t = {pack_bf16x2(v[8 * h], v[8 * h + 1]), pack_bf16x2(v[8 * h + 2], v[8 * h + 3]),
                                     pack_bf16x2(v[8 * h + 4], v[8 * h + 5]), pack_bf16x2(v[8 * h + 6], v[8 * h + 7])};
                    *reinterpret_cast<u32x4*>(cw + lr * NS_LDC + lc + 8 * h) = t;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#if MFC_NS_CT >= 2
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the image is complete before any lane reads it back
#endif
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int64_t ccol = tile * NS_BN + 8 * (lane & 7);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = (lane >> 3) + 8 * h;
                    const u32x4 t = *reinterpret_cast<const u32x4*>(cw + row * NS_LDC + 8 * (lane & 7));
                    __builtin_amdgcn_raw_buffer_store_b128(t, rs_c, ccol < N ? (uint32_t)((row * g.ldc + ccol) * 2) : NS_OOB, 0, 0);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
#else
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u32x4 t = {pack_bf16x2(v[8 * h], v[8 * h + 1]), pack_bf16x2(v[8 * h + 2], v[8 * h + 3]),
                                 pack_bf16x2(v[8 * h + 4], v[8 * h + 5]), pack_bf16x2(v[8 * h + 6], v[8 * h + 7])};
                __builtin_amdgcn_raw_buffer_store_b128(t, rs_c, coff + 16 * h, 0, 0);
            }
#endif
        }
        // commit the prefetched B tile (every wave left the MFMA loop at the barrier above)
        commit_b();
        __syncthreads();
    }
}

// Build the per-wave tile plan.  Returns MT (1..4) or 0 when the shape does not fit.
inline int ns_make_plan(int64_t M, int64_t bias_rows, bool ln, bool ln_tan, NsPlan& plan) {
    for (int w = 0; w < 4; ++w)
        for (int i = 0; i < NS_MAXT; ++i) { plan.tile[w][i] = -1; plan.kind[w][i] = 0; }
    const int tiles = (int)ceil_div64(M, 16);
    if (tiles > 16) return 0;
    int cnt[4] = {0, 0, 0, 0};
    auto put = [&](int w, int t, int kind) { plan.tile[w][cnt[w]] = (signed char)t; plan.kind[w][cnt[w]] = (signed char)kind; ++cnt[w]; };
    auto emptiest = [&]() { int b = 0; for (int w = 1; w < 4; ++w) if (cnt[w] < cnt[b]) b = w; return b; };
    if (!ln) {
        for (int t = 0; t < tiles; ++t) { const int w = emptiest(); if (cnt[w] >= NS_MAXT) return 0; put(w, t, 0); }
    } else {
        if (bias_rows % 16 != 0 && bias_rows < M) return 0;      // tangent tiles must line up with primal tiles
        const int ptiles = (int)ceil_div64(bias_rows < M ? bias_rows : M, 16);
        const int ttiles = tiles - ptiles;
        if (ttiles > ptiles) return 0;
        for (int t = 0; t < ptiles; ++t) {
            const int w = emptiest();
            const bool pair = t < ttiles;
            if (cnt[w] + (pair ? 2 : 1) > NS_MAXT) return 0;
            put(w, t, 1);
            if (pair) put(w, ptiles + t, ln_tan ? 2 : 0);
        }
    }
    int mt = 0;
    for (int w = 0; w < 4; ++w) mt = cnt[w] > mt ? cnt[w] : mt;
    return mt;
}

// standalone tangent of the fused LayerNorm for shapes the N-streaming kernel does not take:
// rows [bias_rows, M) of C hold the raw tangent, rows [0, M - bias_rows) the normalised primal
template <typename T>
__global__ void __launch_bounds__(256)
ln16_tangent_kernel(int64_t rows, int64_t groups, T* C, int64_t ldc, int64_t bias_rows, const float* rstd) {
    const int64_t total = rows * groups;
    for (int64_t o = blockIdx.x * 256LL + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t row = o / groups, grp = o - row * groups;
        float xd[16], n[16];
        const T* pn = C + row * ldc + grp * 16;
        T* pd = C + (bias_rows + row) * ldc + grp * 16;
#pragma unroll
        for (int k = 0; k < 16; ++k) { xd[k] = St<T>::ld(pd + k); n[k] = St<T>::ld(pn + k); }
        ln16_tangent_lane(xd, n, rstd[row * groups + grp]);
#pragma unroll
        for (int k = 0; k < 16; ++k) St<T>::st(pd + k, xd[k]);
    }
}

// split-K slab reduction, fixed-shape tree: 16 waves of a workgroup each add the slabs congruent to their index mod 16
// (ascending; 64 consecutive outputs per wave instruction), then the 16 partial sums are added in wave order and land in
// slab 0.  Deterministic, and parallel enough for the K = S products (hundreds of slabs over a 128 x 128 output).
__global__ void __launch_bounds__(1024)
gemm_slab_reduce_kernel(float* ws, int nslab, int64_t total) {
    __shared__ float part[16][64];
    const int il = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t o = blockIdx.x * 64LL + il;
    float v = 0.f;
    if (o < total)
        for (int z = w; z < nslab; z += 16) v += ws[(int64_t)z * total + o];
    part[w][il] = v;
    __syncthreads();
    if (w == 0 && o < total) {
        float sum = part[0][il];
#pragma unroll
        for (int k = 1; k < 16; ++k) sum += part[k][il];
        ws[o] = sum;
    }
}

// split-K / activation epilogue over the fp32 workspace
template <typename T>
__global__ void __launch_bounds__(256)
gemm_epilogue_kernel(const float* ws, int nslab, int64_t M, int64_t N, T* C, int64_t ldc,
                     const float* bias, int64_t bias_rows, int gelu, int64_t act_rows,
                     float alpha, const T* R, int64_t ldr, float beta, int accum) {
    const int64_t total = M * N;
    auto slabsum = [&](int64_t o) {       // fixed order: slice 0, 1, 2, ...
        float v = ws[o];
        for (int z = 1; z < nslab; ++z) v += ws[(int64_t)z * total + o];
        return v;
    };
    for (int64_t o = blockIdx.x * 256LL + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t row = o / N, col = o - row * N;
        float v = slabsum(o);
        if (bias && row < bias_rows) v += bias[col];
        if (gelu) {
            if (row < act_rows) {
                v = gelu_f(v);
            } else {
                // tangent row: t * gelu'(pre) with pre = primal pre-activation
                const int64_t pr = row - act_rows;
                float pre = slabsum(pr * N + col);
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

template <typename T, int BK, int BMT>
void launch_bk(bool ta, bool tb, dim3 grid, const GemmArgs& g, const GemmFast& f, hipStream_t st) {
    if constexpr (sizeof(T) == 4 && BK == 64 && BMT <= 128) {
        if (f.on) {      // fp32, whole K-steps, operands below 4 GiB: branch-free staging
            if (!ta && !tb) hipLaunchKernelGGL((gemm_f32_fast_kernel<false, false, BMT>), grid, dim3(GT), 0, st, g, f.a_bytes, f.b_bytes);
            else if (!ta && tb) hipLaunchKernelGGL((gemm_f32_fast_kernel<false, true, BMT>), grid, dim3(GT), 0, st, g, f.a_bytes, f.b_bytes);
            else if (ta && !tb) hipLaunchKernelGGL((gemm_f32_fast_kernel<true, false, BMT>), grid, dim3(GT), 0, st, g, f.a_bytes, f.b_bytes);
            else hipLaunchKernelGGL((gemm_f32_fast_kernel<true, true, BMT>), grid, dim3(GT), 0, st, g, f.a_bytes, f.b_bytes);
            return;
        }
    }
    if constexpr (sizeof(T) == 2 && BK == 64) {
        if (f.on) {
            if (!ta && !tb) hipLaunchKernelGGL((gemm_bf16_fast_kernel<false, false, BMT>), grid, dim3(GT), 0, st, g, f.a_bytes, f.b_bytes);
            else if (!ta && tb) hipLaunchKernelGGL((gemm_bf16_fast_kernel<false, true, BMT>), grid, dim3(GT), 0, st, g, f.a_bytes, f.b_bytes);
            else if (ta && !tb) hipLaunchKernelGGL((gemm_bf16_fast_kernel<true, false, BMT>), grid, dim3(GT), 0, st, g, f.a_bytes, f.b_bytes);
            else hipLaunchKernelGGL((gemm_bf16_fast_kernel<true, true, BMT>), grid, dim3(GT), 0, st, g, f.a_bytes, f.b_bytes);
            return;
        }
    }
    if (!ta && !tb) hipLaunchKernelGGL((gemm_kernel<T, BK, false, false, BMT>), grid, dim3(GT), 0, st, g);
    else if (!ta && tb) hipLaunchKernelGGL((gemm_kernel<T, BK, false, true, BMT>), grid, dim3(GT), 0, st, g);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_kernel<T, BK, true, false, BMT>), grid, dim3(GT), 0, st, g);
    else hipLaunchKernelGGL((gemm_kernel<T, BK, true, true, BMT>), grid, dim3(GT), 0, st, g);
}

// one launch over rows [m_base, m_base + rows) with BMT-row tiles
template <typename T, int BMT>
int launch_rows(int flags, GemmArgs g, const GemmFast& f, int bk, int splitk, int64_t m_base, int64_t rows, hipStream_t st) {
    const int64_t tm = ceil_div64(rows, BMT), tn = ceil_div64(g.N, BN);
    g.m_base = m_base;
    g.M = m_base + rows;          // rows >= M are masked; rows < m_base are never touched by this launch
    g.m_fast = (tn <= 65535) ? 1 : 0;
    if (!g.m_fast && tm > 65535) return MFC_EINVAL;
    dim3 grid((unsigned)(g.m_fast ? tm : tn), (unsigned)(g.m_fast ? tn : tm), (unsigned)splitk);
    const bool ta = flags & MFC_GEMM_TRANS_A, tb = flags & MFC_GEMM_TRANS_B;
    if (bk == 32) launch_bk<T, 32, BMT>(ta, tb, grid, g, f, st);
    else launch_bk<T, 64, BMT>(ta, tb, grid, g, f, st);
    return mfc_launch_status();
}

template <typename T>
int launch(int flags, GemmArgs g, const GemmFast& f, int bk, int splitk, int gelu, int64_t act_rows, hipStream_t st) {
    // 128-row tiles for the bulk; a remainder of <= 64 rows (row-stacked [primal; tangent] batches such as
    // 128 + 64) runs as a second launch with 64-row tiles instead of a half-empty 128-row tile
    const int64_t M = g.M;
    const int64_t tail = M % 128;
    int rc = MFC_OK;
    static const int bm192 = getenv("MFC_GEMM_BM192") ? atoi(getenv("MFC_GEMM_BM192")) : 1;
    if (bm192 && M > 128 && M <= 192 && sizeof(T) == 2) {
        // 128 + <= 64 row-stacked rows in ONE 192-row tile: the streamed operand is read once, not once per launch
        rc = launch_rows<T, 192>(flags, g, f, bk, splitk, 0, M, st);
    } else if (tail > 0 && tail <= 64) {
        if (M > tail) rc = launch_rows<T, 128>(flags, g, f, bk, splitk, 0, M - tail, st);
        if (!rc) rc = launch_rows<T, 64>(flags, g, f, bk, splitk, M - tail, tail, st);
    } else {
        rc = launch_rows<T, 128>(flags, g, f, bk, splitk, 0, M, st);
    }
    g.M = M;
    if (rc) return rc;
    if (g.ws) {
        int64_t blocks = ceil_div64(g.M * g.N, 256);
        if (blocks > 4096) blocks = 4096;
        if (splitk > 8) {     // many slabs: tree reduction into slab 0 first (the loop in the epilogue would be serial)
            hipLaunchKernelGGL(gemm_slab_reduce_kernel, dim3((unsigned)ceil_div64(g.M * g.N, 64)), dim3(1024), 0, st, g.ws,
                               splitk, g.M * g.N);
            splitk = 1;
        }
        hipLaunchKernelGGL((gemm_epilogue_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st,
                           (const float*)g.ws, splitk, g.M, g.N, (T*)g.C, g.ldc, g.bias, g.bias_rows, gelu, act_rows,
                           g.alpha, (const T*)g.R, g.ldr, g.beta, g.accum);
        rc = mfc_launch_status();
    }
    return rc;
}

// debugging / A-B switches (environment, read once): MFC_GEMM_NSTREAM=0 disables the N-streaming
// kernel, MFC_GEMM_NS_BLOCKS caps its persistent grid (default 2 workgroups per CU)
inline bool ns_disabled() {
    static const bool off = [] { const char* e = getenv("MFC_GEMM_NSTREAM"); return e && e[0] == '0'; }();
    return off;
}
inline int64_t ns_max_blocks() {
    static const int64_t n = [] { const char* e = getenv("MFC_GEMM_NS_BLOCKS"); const long v = e ? atol(e) : 0; return (int64_t)(v > 0 ? v : 512); }();
    return n;
}

}  // namespace

namespace {
struct OptArgs { float* p; float* m; float* v; float lr, b1, b2, eps, wd, bc1, bc2; float* colsum; float colsum_scale; };
int gemm_impl(int dtype, int flags, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
              int64_t ldb, void* C, int64_t ldc, const float* bias, int64_t bias_rows, int64_t act_rows, float alpha,
              const void* R, int64_t ldr, float beta_res, int splitk, float* ws, float* ln_rstd, const OptArgs* opt,
              void* stream);
}  // namespace

namespace {
// effective number of K slices for a requested split (the chunk is rounded up to whole K-steps of the LDS tiles)
inline int effective_splitk(int64_t K, int splitk) {
    if (splitk < 1) splitk = 1;
    const int bk = K <= 32 ? 32 : 64;
    const int64_t kc = ceil_div64(ceil_div64(K, splitk), bk) * bk;
    return (int)ceil_div64(K, kc);
}
}  // namespace

extern "C" int64_t mfc_gemm_ws_elems(int flags, int64_t M, int64_t N, int64_t K, int splitk) {
    if (M <= 0 || N <= 0 || K <= 0) return -1;
    const int eff = effective_splitk(K, splitk);
    return (eff > 1 || (flags & MFC_GEMM_GELU)) ? (int64_t)eff * M * N : 0;
}

extern "C" int mfc_gemm(int dtype, int flags, int64_t M, int64_t N, int64_t K,
                        const void* A, int64_t lda, const void* B, int64_t ldb,
                        void* C, int64_t ldc,
                        const float* bias, int64_t bias_rows, int64_t act_rows,
                        float alpha, const void* R, int64_t ldr, float beta_res,
                        int splitk, float* ws, float* ln_rstd, void* stream) {
    return gemm_impl(dtype, flags, M, N, K, A, lda, B, ldb, C, ldc, bias, bias_rows, act_rows, alpha, R, ldr, beta_res,
                     splitk, ws, ln_rstd, nullptr, stream);
}

extern "C" int mfc_gemm_adamw(int flags, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                              int64_t ldb, float grad_scale, float* p, float* m, float* v, void* p_bf16, float lr,
                              float b1, float b2, float eps, float wd, int64_t step, float* colsum, float colsum_scale,
                              void* stream) {
    if (!p || !m || !v || !p_bf16) return MFC_EFAULT;
    if (step < 1 || (flags & ~(MFC_GEMM_TRANS_A | MFC_GEMM_TRANS_B))) return MFC_EINVAL;
    if (colsum && ((flags & MFC_GEMM_TRANS_B) || K <= 32)) return MFC_ENOSYS;     // needs the [k][n] LDS image of B, BK = 64
    OptArgs o{p, m, v, lr, b1, b2, eps, wd, 1.0f - powf(b1, (float)step), 1.0f - powf(b2, (float)step), colsum, colsum_scale};
    return gemm_impl(MFC_BF16, flags, M, N, K, A, lda, B, ldb, p_bf16, N, nullptr, 0, M, grad_scale, nullptr, 0, 0.f, 1,
                     nullptr, nullptr, &o, stream);
}

namespace {
int gemm_impl(int dtype, int flags, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
              int64_t ldb, void* C, int64_t ldc, const float* bias, int64_t bias_rows, int64_t act_rows, float alpha,
              const void* R, int64_t ldr, float beta_res, int splitk, float* ws, float* ln_rstd, const OptArgs* opt,
              void* stream) {
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
    if (splitk > 65535) return MFC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const size_t es = dtype == MFC_F32 ? 4 : 2;

    GemmArgs g;
    g.M = M; g.N = N; g.K = K;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.bias = bias; g.bias_rows = bias_rows;
    g.alpha = alpha; g.R = R; g.ldr = ldr; g.beta = beta_res;
    g.accum = (flags & MFC_GEMM_ACCUM) ? 1 : 0;
    const int bk = K <= 32 ? 32 : 64;    // K-step of the LDS tiles
    int64_t kc = ceil_div64(ceil_div64(K, splitk), bk) * bk;
    g.kchunk = kc;
    splitk = (int)ceil_div64(K, kc);
    const bool use_ws = (splitk > 1) || gelu;
    g.ws = use_ws ? ws : nullptr;
    // 16-byte vector loads need every row start 16-byte aligned
    g.vecA = (((lda * es) % 16) == 0) && (((uintptr_t)A % 16) == 0);
    g.vecB = (((ldb * es) % 16) == 0) && (((uintptr_t)B % 16) == 0);
    GemmFast fast = {0u, 0u, 0};
    {
        // operand extents in bytes ([rows][ld] with the last row only as wide as it is used)
        const int64_t a_rows = ta ? K : M, a_cols = ta ? M : K, b_rows = tb ? N : K, b_cols = tb ? K : N;
        const int64_t ab = ((a_rows - 1) * lda + a_cols) * es, bb = ((b_rows - 1) * ldb + b_cols) * es;
        // 32-bit offsets: the slow axis of an operand is walked in tiles of 64 / 128 / 192 rows (M, N) or in whole K-steps
        const int64_t oa = (ta ? K : ceil_div64(M, 64) * 64) * lda * (int64_t)es, ob = (tb ? ceil_div64(N, 128) * 128 : K) * ldb * (int64_t)es;
        fast.on = !opt && g.vecA && g.vecB && K > 32 && K % 64 == 0 && ab < (1LL << 32) && bb < (1LL << 32) &&
                  oa < (1LL << 32) && ob < (1LL << 32);
        fast.a_bytes = (uint32_t)(fast.on ? ab : 0);
        fast.b_bytes = (uint32_t)(fast.on ? bb : 0);
    }
    g.vecC = (N % 16 == 0) && ((ldc * es) % 16 == 0) && (((uintptr_t)C % 16) == 0) &&
             (!R || (((ldr * es) % 16 == 0) && (((uintptr_t)R % 16) == 0)));
    g.opt_p = g.opt_m = g.opt_v = nullptr;
    g.lr = g.b1 = g.b2 = g.eps = g.wd = g.bc1 = g.bc2 = 0.f;
    g.colsum = nullptr; g.colsum_scale = 0.f;
    if (opt) {
        // the fused update lives in the row-contiguous epilogue of the tiled kernel
        if (!g.vecC || use_ws || (((uintptr_t)opt->p | (uintptr_t)opt->m | (uintptr_t)opt->v) & 15)) return MFC_ENOSYS;
        g.opt_p = opt->p; g.opt_m = opt->m; g.opt_v = opt->v;
        g.lr = opt->lr; g.b1 = opt->b1; g.b2 = opt->b2; g.eps = opt->eps; g.wd = opt->wd; g.bc1 = opt->bc1; g.bc2 = opt->bc2;
        g.colsum = opt->colsum; g.colsum_scale = opt->colsum_scale;
    }
    g.ln_rstd = nullptr;
    if (flags & MFC_GEMM_LN16) {
        if (!ln_rstd) return MFC_EFAULT;
        if (!g.vecC || use_ws) return MFC_ENOSYS;
        g.ln_rstd = ln_rstd;
    }
    const bool ln = g.ln_rstd != nullptr, ln_tan = ln && (flags & MFC_GEMM_LN16T) && bias_rows < M;
    if (ln_tan && M - bias_rows > bias_rows) return MFC_EINVAL;
    g.m_base = 0;
    // skinny NN / NT products (K = 128) with the whole A operand in registers
    if (dtype == MFC_BF16 && !opt && !ta && K == NS_K && !use_ws && g.vecA && g.vecB && g.vecC && !ns_disabled() &&
        (tb ? (ldb == NS_K && !ln && !bias && N * (int64_t)NS_K * 2 < (1LL << 32)) : ldb < (1 << 23)) &&
        ldc < (1 << 26) && (!R || ldr < (1 << 26)) && N < (1LL << 29)) {   // 32-bit buffer offsets
        NsPlan plan;
        const int mt = ns_make_plan(M, bias_rows, ln, ln_tan, plan);
        if (mt > 0) {
            const int64_t ntiles = ceil_div64(N, NS_BN);
            int64_t grid = ntiles < ns_max_blocks() ? ntiles : ns_max_blocks();
#define MFC_NS_LAUNCH(MTV)                                                                                                      \
            if (tb) hipLaunchKernelGGL((gemm_nstream_kernel<MTV, true>), dim3((unsigned)grid), dim3(GT), 0, st, g, plan, ntiles);   \
            else hipLaunchKernelGGL((gemm_nstream_kernel<MTV, false>), dim3((unsigned)grid), dim3(GT), 0, st, g, plan, ntiles)
            switch (mt) {
                case 1: MFC_NS_LAUNCH(1); break;
                case 2: MFC_NS_LAUNCH(2); break;
                case 3: MFC_NS_LAUNCH(3); break;
                default: MFC_NS_LAUNCH(4); break;
            }
#undef MFC_NS_LAUNCH
            return mfc_launch_status();
        }
    }
    g.ws_slab = M * N;     // every (row < M, col < N) of every slab is written by exactly one workgroup: no memset
    int rc = dtype == MFC_F32 ? launch<float>(flags, g, fast, bk, splitk, gelu, act_rows, st)
                              : launch<u16>(flags, g, fast, bk, splitk, gelu, act_rows, st);
    if (!rc && ln_tan) {
        const int64_t rows = M - bias_rows, groups = N >> 4;
        int64_t blocks = ceil_div64(rows * groups, 256);
        if (blocks > 8192) blocks = 8192;
        if (dtype == MFC_F32)
            hipLaunchKernelGGL((ln16_tangent_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, rows, groups,
                               (float*)C, ldc, bias_rows, (const float*)ln_rstd);
        else
            hipLaunchKernelGGL((ln16_tangent_kernel<u16>), dim3((unsigned)blocks), dim3(256), 0, st, rows, groups,
                               (u16*)C, ldc, bias_rows, (const float*)ln_rstd);
        rc = mfc_launch_status();
    }
    return rc;
}
}  // namespace

