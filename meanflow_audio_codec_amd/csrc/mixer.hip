// Fused channel-mixing MLP of an MLP-Mixer block on 16-channel tokens (models/mlp_mixer.py:66-94: the
// `Dense(channel_mix_dim) -> gelu -> Dense(num_channels)` pair applied to every token, + residual).
//
//   out = gelu(a W1 + b1) W2 + b2 + res          a, out, res [rows, 16];  W1 [16, H];  W2 [H, 16]
//
// With 16 channels the two Dense layers are K = 16 / N = 16 products around a [rows, H] hidden activation that is 128 x
// larger than the tokens (config #3: 196 608 rows x 2048 = 1.6 GB in fp32, written and re-read four times per block by the
// GEMM + gelu formulation).  Here the hidden activation never leaves the registers: every product is an MFMA step issued
// TRANSPOSED (hidden x rows), so the accumulator of the first product -- lane (q, r): hidden 4q..4q+3 of row r -- is
// already the B operand of the second one, exactly as in the ConvNeXt kernels (convnext.hip).  fp32 storage runs on
// v_mfma_f32_16x16x4_f32 (the exact fp32 chain the tiled GEMM uses), bf16 storage on v_mfma_f32_16x16x16_bf16.
//   forward : one wave = two 16-row tiles (+ the tangent rows of the same tokens: gelu'(h) * hdot), loop over H / 16 hidden tiles
//   reverse : recomputes h from a; one workgroup = 8 waves that split H (1024 of it per pass) between them and keep d W1 / d W2 /
//             d b1 of their hidden range in registers over all the workgroup's rows, two 16-row tiles per step; d a is summed over the waves through LDS in wave
//             order; per-workgroup records are reduced in index order by a second kernel (no atomics: bitwise reproducible).
// Weight fragments are read straight from W1 / W2 (L2-resident: 256 KB) -- no packing pass, no workspace for them.
#include "mfc_common.h"

namespace {

constexpr int CM_C = 16;         // channels per token
constexpr int CM_RT = 2;         // forward: 16-row tiles per wave
constexpr int CM_BW = 8;         // reverse: waves per workgroup
constexpr int CM_TLD = 20;       // padded row of the 16 x 16 transposes (floats)

template <typename T> struct CmIO;
template <> struct CmIO<float> {
    typedef f32x4 frag;
    __device__ static inline frag zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ static inline frag ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    __device__ static inline frag gather(const float* p, int64_t s) { return f32x4{p[0], p[s], p[2 * s], p[3 * s]}; }
    __device__ static inline f32x4 f32(const frag& f) { return f; }
    __device__ static inline void st(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct CmIO<u16> {
    typedef s16x4 frag;
    __device__ static inline frag zero() { return s16x4{0, 0, 0, 0}; }
    __device__ static inline frag ld(const u16* p) { return *reinterpret_cast<const s16x4*>(p); }
    __device__ static inline frag gather(const u16* p, int64_t s) {
        return s16x4{(short)p[0], (short)p[s], (short)p[2 * s], (short)p[3 * s]};
    }
    __device__ static inline f32x4 f32(const frag& f) {
        return f32x4{bf16_to_f32((u16)f[0]), bf16_to_f32((u16)f[1]), bf16_to_f32((u16)f[2]), bf16_to_f32((u16)f[3])};
    }
    __device__ static inline void st(u16* p, const f32x4& v) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2*>(p) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
};

// Weight fragments through raw buffer resources: the lane-dependent part of an address is ONE loop-invariant VGPR per
// fragment kind and the hidden tile enters through the scalar offset -- with plain pointers the compiler keeps a 64-bit
// address per (fragment, hidden tile) alive across the row loop of the reverse kernel (+11 VGPRs per tile).
typedef uint32_t cm_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t cm_u32x2 __attribute__((ext_vector_type(2)));
__device__ inline __amdgpu_buffer_rsrc_t cm_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ inline f32x4 cm_ld_f32x4(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
template <typename T> struct CmBuf;
template <> struct CmBuf<float> {
    __device__ static inline f32x4 ld4(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) { return cm_ld_f32x4(rs, voff, soff); }
    __device__ static inline f32x4 gather(__amdgpu_buffer_rsrc_t rs, const uint32_t (&voff)[4], uint32_t soff) {
        return f32x4{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[0], soff, 0)),
                     __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[1], soff, 0)),
                     __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[2], soff, 0)),
                     __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[3], soff, 0))};
    }
};
template <> struct CmBuf<u16> {
    __device__ static inline s16x4 ld4(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
        return __builtin_bit_cast(s16x4, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
    }
    __device__ static inline s16x4 gather(__amdgpu_buffer_rsrc_t rs, const uint32_t (&voff)[4], uint32_t soff) {
        return s16x4{(short)__builtin_amdgcn_raw_buffer_load_b16(rs, voff[0], soff, 0), (short)__builtin_amdgcn_raw_buffer_load_b16(rs, voff[1], soff, 0),
                     (short)__builtin_amdgcn_raw_buffer_load_b16(rs, voff[2], soff, 0), (short)__builtin_amdgcn_raw_buffer_load_b16(rs, voff[3], soff, 0)};
    }
};

struct CmFwdArgs {
    const void* a; const void* W1; const void* W2; const void* res; const float* b1; const float* b2; void* out;
    int64_t rows, act_rows, H;
};

// One unit = CM_RT primal row tiles (and, TAN, the tangent rows act_rows + the same row indices).
// Operands of the K16 steps (mfc_common.h: A fragment = row r, k 4q..4q+3; B fragment = column r, the same k):
//   H^T [hidden x row] = W1^T a^T      A: W1[4q+s][16t + r]      B: a[row r][4q+s]
//   out^T [ch x row]   = W2^T G        A: W2[16t + 4q+s][r]      B: G[hidden 4q+s][row r] = the accumulator of the first
template <typename T, bool TAN>
__device__ inline void cm_fwd_unit(const CmFwdArgs& g, int64_t row0, int lane) {
    typedef typename CmIO<T>::frag frag_t;
    const int q = lane >> 4, r = lane & 15;
    const T* a = (const T*)g.a;
    const T* W1 = (const T*)g.W1;
    const T* W2 = (const T*)g.W2;
    const int64_t H = g.H, ntan = g.rows - g.act_rows;
    frag_t ab[CM_RT], adb[CM_RT];
#pragma unroll
    for (int i = 0; i < CM_RT; ++i) {
        const int64_t row = row0 + 16 * i + r;
        ab[i] = row < g.act_rows ? CmIO<T>::ld(a + row * CM_C + 4 * q) : CmIO<T>::zero();
        adb[i] = (TAN && row < ntan) ? CmIO<T>::ld(a + (g.act_rows + row) * CM_C + 4 * q) : CmIO<T>::zero();
    }
    f32x4 x2[CM_RT], xd2[CM_RT];
#pragma unroll
    for (int i = 0; i < CM_RT; ++i) { x2[i] = f32x4{0.f, 0.f, 0.f, 0.f}; xd2[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int64_t nt = H >> 4;
    // fragments of hidden tile 0; the next tile's are requested before this tile's products
    frag_t f1 = CmIO<T>::gather(W1 + (int64_t)(4 * q) * H + r, H);
    frag_t f2 = CmIO<T>::gather(W2 + (int64_t)(4 * q) * CM_C + r, CM_C);
    f32x4 bias = *reinterpret_cast<const f32x4*>(g.b1 + 4 * q);
    for (int64_t t = 0; t < nt; ++t) {
        const int64_t tn = t + 1 < nt ? t + 1 : t;
        const frag_t f1n = CmIO<T>::gather(W1 + (int64_t)(4 * q) * H + 16 * tn + r, H);
        const frag_t f2n = CmIO<T>::gather(W2 + (16 * tn + 4 * q) * CM_C + r, CM_C);
        const f32x4 biasn = *reinterpret_cast<const f32x4*>(g.b1 + 16 * tn + 4 * q);
#pragma unroll
        for (int i = 0; i < CM_RT; ++i) {
            f32x4 h = bias;
            mma16(h, f1, ab[i]);
            frag_t gf;
            if constexpr (TAN) {
                f32x4 hd = f32x4{0.f, 0.f, 0.f, 0.f};
                mma16(hd, f1, adb[i]);
                f32x4 gv, dg;
                gelu_both4(h, gv, dg);
                const f32x4 gd = dg * hd;
                make_frag(gf, gv[0], gv[1], gv[2], gv[3]);
                frag_t gdf;
                make_frag(gdf, gd[0], gd[1], gd[2], gd[3]);
                mma16(x2[i], f2, gf);
                mma16(xd2[i], f2, gdf);
            } else {
                const f32x4 gv = gelu4(h);
                make_frag(gf, gv[0], gv[1], gv[2], gv[3]);
                mma16(x2[i], f2, gf);
            }
        }
        f1 = f1n; f2 = f2n; bias = biasn;
    }
    // lane (q, r): channels 4q..4q+3 of row r
    const T* res = (const T*)g.res;
    T* out = (T*)g.out;
    const f32x4 b2 = *reinterpret_cast<const f32x4*>(g.b2 + 4 * q);
#pragma unroll
    for (int i = 0; i < CM_RT; ++i) {
        const int64_t row = row0 + 16 * i + r;
        if (row < g.act_rows) {
            f32x4 v = x2[i] + b2;
            if (res) v += CmIO<T>::f32(CmIO<T>::ld(res + row * CM_C + 4 * q));
            CmIO<T>::st(out + row * CM_C + 4 * q, v);
        }
        if (TAN && row < ntan) {
            const int64_t tr = g.act_rows + row;
            f32x4 v = xd2[i];
            if (res) v += CmIO<T>::f32(CmIO<T>::ld(res + tr * CM_C + 4 * q));
            CmIO<T>::st(out + tr * CM_C + 4 * q, v);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) chanmlp_fwd_kernel(CmFwdArgs g) {
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t ntan = g.rows - g.act_rows;
    const int64_t units = (g.act_rows + 16 * CM_RT - 1) / (16 * CM_RT);
    // units with tangent rows cost twice the others and come first: wave w takes w, w + nwaves, ... so every wave gets both kinds
    for (int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); u < units; u += nwaves) {
        const int64_t row0 = u * 16 * CM_RT;
        if (row0 < ntan) cm_fwd_unit<T, true>(g, row0, lane);
        else cm_fwd_unit<T, false>(g, row0, lane);
    }
}

struct CmBwdArgs {
    const void* a; const void* dy; const void* W1; const void* W2; const float* b1; void* da; float* ws; float* dascr;
    int64_t rows, H;
    int npass;
};

// Reverse pass.  Wave w of a workgroup owns hidden tiles [w TPW, (w+1) TPW) (H = 128 TPW) for every row tile the workgroup
// walks.  Per (row tile, hidden tile):
//   H^T  = W1^T a^T + b1        A: W1[4q+s][16t + r]        B: a[row r][4q+s]
//   dG^T = W2 dy^T              A: W2[16t + r][4q+s]        B: dy[row r][4q+s]
//   dH   = dG * gelu'(H)
//   da^T += W1 dH^T  (over this wave's hidden)   A: W1[r][16t + 4q+s]   B: dH[hidden 4q+s][row r] = the registers as they stand
//   dW2 [hidden x ch] += G^T dy   A: G[row 4q+s][hidden r]  (16 x 16 transpose through LDS)   B: dy[row 4q+s][ch r]
//   dW1^T [hidden x ch] += dH^T a A: dH[row 4q+s][hidden r] (same)                             B: a[row 4q+s][ch r]
//   db1 [hidden r] += sum_s dH[row 4q+s][hidden r]   (per-lane partial over the lane's 4 rows; the 4 q's are summed at the flush)
// Rows past `rows` load zeros: dy = 0 there, so every contribution vanishes.
template <typename T, int TPW, int RG = 2>
__global__ void __launch_bounds__(64 * CM_BW) chanmlp_bwd_kernel(CmBwdArgs g) {
    typedef typename CmIO<T>::frag frag_t;
    __shared__ __attribute__((aligned(16))) float tr[CM_BW][4 * RG][16 * CM_TLD];      // per wave: G / dH transposes, then a^T / dy^T
    __shared__ __attribute__((aligned(16))) float red[2][CM_BW * RG][256];
    const int lane = threadIdx.x & 63, q = lane >> 4, r = lane & 15;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const T* a = (const T*)g.a;
    const T* dy = (const T*)g.dy;
    const int64_t H = g.H;
    constexpr uint32_t ES = sizeof(T);
    const __amdgpu_buffer_rsrc_t rsW1 = cm_rsrc(g.W1, (uint32_t)(CM_C * H * ES)), rsW2 = cm_rsrc(g.W2, (uint32_t)(CM_C * H * ES));
    const __amdgpu_buffer_rsrc_t rsb1 = cm_rsrc(g.b1, (uint32_t)(H * 4));
    uint32_t vo1[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) vo1[s] = (uint32_t)(((4 * q + s) * H + r) * ES);
    const uint32_t vo3 = (uint32_t)((r * CM_C + 4 * q) * ES), vo4 = (uint32_t)((r * H + 4 * q) * ES), vob = (uint32_t)(4 * q * 4);
    const int64_t ngroups = (g.rows + 16 * RG - 1) / (16 * RG);
    float* rec = g.ws + (int64_t)blockIdx.x * (2 * H * CM_C + H);
    int par = 0;
    // H = npass x (8 waves x TPW tiles x 16): a pass walks all the workgroup's rows for one slice of the hidden range (the
    // accumulators of 16 tiles per wave do not fit 256 VGPRs); d a of the earlier passes waits in fp32 in `dascr`, which
    // only this workgroup touches for its rows
    for (int pass = 0; pass < g.npass; ++pass) {
    f32x4 dW1a[TPW], dW2a[TPW];
    float dba[TPW];
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) { dW1a[tt] = f32x4{0.f, 0.f, 0.f, 0.f}; dW2a[tt] = f32x4{0.f, 0.f, 0.f, 0.f}; dba[tt] = 0.f; }
    const uint32_t hw = 16u * (uint32_t)((pass * CM_BW + w) * TPW);
    // RG row tiles per step: the token rows come from HBM and the wait for them (and, from the second pass on, for the
    // parked d a) is exposed once per step -- measured ~3.7 us per step against ~1 us of products per hidden tile -- so a
    // step carries as many row tiles as the registers allow; the weight fragments of a hidden tile serve all of them
    for (int64_t rg = blockIdx.x; rg < ngroups; rg += gridDim.x) {
        // The token rows as B operands with k = channels (aB1 / dyB1: lane (q, r) = row r, channels 4q..4q+3, one 16-byte
        // load) stay in registers for the whole step.  The contraction over rows needs them with k = rows 4q..4q+3, column =
        // channel r: each wave parks a transposed fp32 copy [channel][row] in its own LDS slab once per step and reads the
        // fragment back per use (one ds_read_b128) -- 16 VGPRs less than holding it, and no strided global gather.
        frag_t aB1[RG], dyB1[RG];
        f32x4 da_acc[RG];
#pragma unroll
        for (int i = 0; i < RG; ++i) {
            const int64_t row0 = (rg * RG + i) * 16;
            const bool ok1 = row0 + r < g.rows;
            aB1[i] = ok1 ? CmIO<T>::ld(a + (row0 + r) * CM_C + 4 * q) : CmIO<T>::zero();
            dyB1[i] = ok1 ? CmIO<T>::ld(dy + (row0 + r) * CM_C + 4 * q) : CmIO<T>::zero();
            const f32x4 av = CmIO<T>::f32(aB1[i]), dv = CmIO<T>::f32(dyB1[i]);
            float* aT = &tr[w][2 * RG + 2 * i][0];
            float* dT = &tr[w][2 * RG + 2 * i + 1][0];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                aT[(4 * q + e) * CM_TLD + r] = av[e];
                dT[(4 * q + e) * CM_TLD + r] = dv[e];
            }
            da_acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // weight fragments one hidden tile ahead; the scheduling barrier at the bottom of a tile keeps the compiler from
        // hoisting all TPW tiles' loads to the top (with the 2 TPW + TPW accumulator quads that would not fit 256 VGPRs)
        frag_t f1 = CmBuf<T>::gather(rsW1, vo1, hw * ES);
        f32x4 bias = cm_ld_f32x4(rsb1, vob, hw * 4);
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            // f1 / bias (needed at once) arrive one tile ahead; f3 / f4 are requested here and used after the first chain
            const uint32_t h0 = hw + 16u * tt, hn = hw + 16u * (tt + 1 < TPW ? tt + 1 : tt);
            const frag_t f3 = CmBuf<T>::ld4(rsW2, vo3, h0 * CM_C * ES);
            const frag_t f4 = CmBuf<T>::ld4(rsW1, vo4, h0 * ES);
            const frag_t f1n = CmBuf<T>::gather(rsW1, vo1, hn * ES);
            const f32x4 biasn = cm_ld_f32x4(rsb1, vob, hn * 4);
#pragma unroll
            for (int i = 0; i < RG; ++i) {
                float* trG = &tr[w][2 * i][0];
                float* trD = &tr[w][2 * i + 1][0];
                f32x4 h = bias;
                mma16(h, f1, aB1[i]);
                f32x4 dg = f32x4{0.f, 0.f, 0.f, 0.f};
                mma16(dg, f3, dyB1[i]);
                f32x4 gv, gp;
                gelu_both4(h, gv, gp);
                const f32x4 dh = dg * gp;
                frag_t dhf;
                make_frag(dhf, dh[0], dh[1], dh[2], dh[3]);
                mma16(da_acc[i], f4, dhf);
                // 16 x 16 transposes (wave-private LDS): written [hidden 4q+e][row r], read back [hidden r][rows 4q..4q+3]
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    trG[(4 * q + e) * CM_TLD + r] = gv[e];
                    trD[(4 * q + e) * CM_TLD + r] = dh[e];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const f32x4 gt = *reinterpret_cast<const f32x4*>(trG + r * CM_TLD + 4 * q);
                const f32x4 dht = *reinterpret_cast<const f32x4*>(trD + r * CM_TLD + 4 * q);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const f32x4 a2v = *reinterpret_cast<const f32x4*>(&tr[w][2 * RG + 2 * i][0] + r * CM_TLD + 4 * q);
                const f32x4 d2v = *reinterpret_cast<const f32x4*>(&tr[w][2 * RG + 2 * i + 1][0] + r * CM_TLD + 4 * q);
                frag_t gtf, dhtf, aB2, dyB2;
                make_frag(gtf, gt[0], gt[1], gt[2], gt[3]);
                make_frag(dhtf, dht[0], dht[1], dht[2], dht[3]);
                make_frag(aB2, a2v[0], a2v[1], a2v[2], a2v[3]);
                make_frag(dyB2, d2v[0], d2v[1], d2v[2], d2v[3]);
                mma16(dW2a[tt], gtf, dyB2);
                mma16(dW1a[tt], dhtf, aB2);
                dba[tt] += (dht[0] + dht[1]) + (dht[2] + dht[3]);
            }
            f1 = f1n; bias = biasn;
            __builtin_amdgcn_sched_barrier(0);
        }
        // d a of these row tiles: the waves' partial sums (each over its own hidden range), added in wave order.  One barrier
        // per step: the partials alternate between two LDS slabs (a slab is rewritten two steps later, after the barrier of
        // the step in between, which every wave reaches only after its reads), and every wave sums its share of the rows.
        float* slab = &red[par][0][0];
#pragma unroll
        for (int i = 0; i < RG; ++i)
            *reinterpret_cast<f32x4*>(slab + (w * RG + i) * 256 + r * CM_C + 4 * q) = da_acc[i];
        __syncthreads();
        if (lane < 8 * RG) {
            const int i = lane >> 3, l8 = lane & 7;
            const int rr = 2 * w + (l8 >> 2), c4 = (l8 & 3) * 4;
            const int64_t row = (rg * RG + i) * 16 + rr;
            f32x4 s = *reinterpret_cast<const f32x4*>(slab + i * 256 + rr * CM_C + c4);
#pragma unroll
            for (int k = 1; k < CM_BW; ++k) s += *reinterpret_cast<const f32x4*>(slab + (k * RG + i) * 256 + rr * CM_C + c4);
            if (row < g.rows) {
                float* scr = g.dascr + row * CM_C + c4;
                if (pass > 0) s += *reinterpret_cast<const f32x4*>(scr);
                if (pass + 1 < g.npass) *reinterpret_cast<f32x4*>(scr) = s;
                else CmIO<T>::st((T*)g.da + row * CM_C + c4, s);
            }
        }
        par ^= 1;
    }
    // flush this pass's part of the workgroup's record: [dW1^T: H x 16][dW2: H x 16][db1: H]
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        const int64_t h0 = hw + 16 * tt;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            rec[(h0 + 4 * q + e) * CM_C + r] = dW1a[tt][e];
            rec[H * CM_C + (h0 + 4 * q + e) * CM_C + r] = dW2a[tt][e];
        }
        float s = dba[tt];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (q == 0) rec[2 * H * CM_C + h0 + r] = s;
    }
    __threadfence();       // dascr of this pass before the next pass reads it back (same workgroup, other waves' lanes)
    __syncthreads();
    }
}

// sum the workgroups' records in index order (four interleaved partial sums, combined in a fixed order) and write
// dW1 [16, H] (transposed back), dW2 [H, 16] in the storage type, db1 [H] in fp32
template <typename T>
__global__ void __launch_bounds__(256) chanmlp_reduce_kernel(const float* ws, int nrec, int64_t H, T* dW1, T* dW2, float* db1) {
    const int64_t rec = 2 * H * CM_C + H;
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= rec) return;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 4 <= nrec; k += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += ws[(int64_t)(k + j) * rec + o];
    }
    for (; k < nrec; ++k) s[0] += ws[(int64_t)k * rec + o];
    const float v = (s[0] + s[1]) + (s[2] + s[3]);
    if (o < H * CM_C) {
        const int64_t h = o / CM_C, c = o - h * CM_C;
        St<T>::st(dW1 + c * H + h, v);
    } else if (o < 2 * H * CM_C) {
        St<T>::st(dW2 + (o - H * CM_C), v);
    } else {
        db1[o - 2 * H * CM_C] = v;
    }
}

constexpr int CM_TPW_MAX = 8;    // hidden tiles per wave and pass (2 x 4 + 1 accumulator registers each)
inline bool cm_h_ok(int64_t H) { return H == 128 || H == 256 || H == 512 || (H > 0 && H % 1024 == 0 && H <= 16384); }
inline int cm_tpw(int64_t H) { const int64_t t = H / (16 * CM_BW); return (int)(t < CM_TPW_MAX ? t : CM_TPW_MAX); }
inline int cm_npass(int64_t H) { return (int)(H / (16 * CM_BW * cm_tpw(H))); }
constexpr int CM_RG = 2;         // reverse: 16-row tiles per step
inline int cm_bwd_blocks(int64_t rows) {
    const int64_t tiles = ceil_div64(rows, 16 * CM_RG);
    return (int)(tiles < 256 ? tiles : 256);       // one 8-wave workgroup per CU
}
inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int mfc_chanmlp_fwd(int dtype, int64_t rows, int64_t act_rows, int64_t H, const void* a, const void* W1,
                               const float* b1, const void* W2, const float* b2, const void* res, void* out, void* stream) {
    if (!a || !W1 || !b1 || !W2 || !b2 || !out) return MFC_EFAULT;
    if (rows <= 0 || act_rows <= 0 || act_rows > rows || rows > 2 * act_rows || H <= 0 || (H & 15) ||
        (dtype != MFC_F32 && dtype != MFC_BF16))
        return MFC_EINVAL;
    if (!aligned16(a) || !aligned16(out) || !aligned16(b1) || !aligned16(b2) || (res && !aligned16(res))) return MFC_EINVAL;
    CmFwdArgs g = {a, W1, W2, res, b1, b2, out, rows, act_rows, H};
    const int64_t units = ceil_div64(act_rows, 16 * CM_RT);
    const int64_t wgs = ceil_div64(units, 4);
    const unsigned grid = (unsigned)(wgs < 1024 ? wgs : 1024);       // 4 workgroups of 4 waves per CU
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MFC_F32) hipLaunchKernelGGL(chanmlp_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, g);
    else hipLaunchKernelGGL(chanmlp_fwd_kernel<u16>, dim3(grid), dim3(256), 0, st, g);
    return mfc_launch_status();
}

extern "C" int64_t mfc_chanmlp_ws_elems(int64_t rows, int64_t H) {
    if (rows <= 0 || H <= 0) return MFC_EINVAL;
    if (!cm_h_ok(H)) return MFC_ENOSYS;
    return (int64_t)cm_bwd_blocks(rows) * (2 * H * CM_C + H) + (cm_npass(H) > 1 ? rows * CM_C : 0);
}

extern "C" int mfc_chanmlp_bwd(int dtype, int64_t rows, int64_t H, const void* a, const void* dy, const void* W1,
                               const float* b1, const void* W2, void* da, void* dW1, float* db1, void* dW2, float* ws,
                               void* stream) {
    if (!a || !dy || !W1 || !b1 || !W2 || !da || !dW1 || !db1 || !dW2 || !ws) return MFC_EFAULT;
    if (rows <= 0 || (dtype != MFC_F32 && dtype != MFC_BF16)) return MFC_EINVAL;
    if (!cm_h_ok(H)) return MFC_ENOSYS;
    if (!aligned16(a) || !aligned16(dy) || !aligned16(da) || !aligned16(b1) || !aligned16(W1) || !aligned16(W2) || !aligned16(ws))
        return MFC_EINVAL;
    const int nb = cm_bwd_blocks(rows);
    CmBwdArgs g = {a, dy, W1, W2, b1, da, ws, ws + (int64_t)nb * (2 * H * CM_C + H), rows, H, cm_npass(H)};
    hipStream_t st = (hipStream_t)stream;
#define MFC_CM_BWD(TPW)                                                                                                   \
    if (dtype == MFC_F32) hipLaunchKernelGGL((chanmlp_bwd_kernel<float, TPW>), dim3(nb), dim3(64 * CM_BW), 0, st, g);     \
    else hipLaunchKernelGGL((chanmlp_bwd_kernel<u16, TPW>), dim3(nb), dim3(64 * CM_BW), 0, st, g)
    switch (cm_tpw(H)) {
        case 1: MFC_CM_BWD(1); break;
        case 2: MFC_CM_BWD(2); break;
        case 4: MFC_CM_BWD(4); break;
        default: MFC_CM_BWD(8); break;
    }
#undef MFC_CM_BWD
    const int64_t rec = 2 * H * CM_C + H;
    const unsigned rgrid = (unsigned)ceil_div64(rec, 256);
    if (dtype == MFC_F32) hipLaunchKernelGGL(chanmlp_reduce_kernel<float>, dim3(rgrid), dim3(256), 0, st, ws, nb, H, (float*)dW1, (float*)dW2, db1);
    else hipLaunchKernelGGL(chanmlp_reduce_kernel<u16>, dim3(rgrid), dim3(256), 0, st, ws, nb, H, (u16*)dW1, (u16*)dW2, db1);
    return mfc_launch_status();
}
