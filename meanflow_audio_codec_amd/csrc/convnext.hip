// ConvNeXt block interior (models/conv_flow.py:65-115,162-186) for gfx950.
//
//   h1 = LN_C(h0); h2 = (1+scale) h1 + shift            conv_flow.py:181-186
//   c1 = Conv3x3_SAME(h2); n1 = LN_C(c1)                conv_flow.py:74-84
//   e1 = Conv1x1(n1) [16->32]; g1 = gelu(e1)            conv_flow.py:87-88
//   y  = GRN(g1) = g1 (gamma + q) + beta                conv_flow.py:22-45
//   o  = Conv1x1(y) [32->16] * layer_scale + h2         conv_flow.py:95-115
//
// The first LayerNorm and its tangent come out of the producing GEMM's epilogue (mfc_gemm MFC_GEMM_LN16 /
// MFC_GEMM_LN16T, or the standalone mfc_ln16_fwd / mfc_ln16_jvp): these kernels read h1 = LN_C(h0), its tangent and
// (reverse pass only) the per-pixel 1/sigma rho0.  The FiLM modulation is folded into the conv weights per row r, so
// a halo tile is a verbatim copy of h1 and arrives by LDS-DMA (see Halo below).
// Data layout: NHWC maps [R, s, s, 16] in the storage dtype T (fp32 or bf16).
// One workgroup = one 16x16 pixel tile (+1 halo) of one row r, 4 waves, each
// wave owns 4 tile rows of 16 pixels = one MFMA M-tile.  Every per-pixel
// contraction is an MFMA "K16 step" (mfc_common.h): the 3x3 conv is 9 steps
// (one per tap, K = 16 input channels read straight out of the LDS halo
// tile), the 1x1 convs 1-2 steps.  Every product is issued TRANSPOSED,
// Y^T = W^T X^T (weights as the A operand, pixels as the B operand), so the
// result has the pixel on the lane and 4 output channels (4q..4q+3) in
// registers -- exactly the B-operand layout of the next product: the whole
// chain conv -> LN -> 1x1 -> GELU -> GRN -> 1x1 runs register-to-register with
// no LDS round trip, channel reductions (LayerNorm) are 2 cross-lane steps
// and global I/O is a contiguous 1 KiB per wave instruction.  Only the weight
// gradients (contractions over pixels) need one batched transpose through a
// wave-private LDS scratch per tile row.  The 32-channel intermediates never
// touch HBM; GRN's global statistic makes the chain run twice (stats pass /
// apply pass), forward and backward.
//
// Workgroups are persistent over a contiguous range of tiles so weight-gradient
// and per-row statistics accumulate in registers.  Every reduction is FIXED-ORDER
// (bitwise reproducible run to run, no atomics): the four waves of a workgroup add
// their partial sums into an LDS scratch one after the other (wave 0, 1, 2, 3), the
// workgroup stores the result into ITS OWN record of the caller's workspace with plain
// stores, and a small follow-up kernel sums the records in workgroup order
// (cnx_reduce_rows_kernel for per-row quantities, cnx_reduce_blocks_kernel for the
// parameter gradients).  (Hot-address fp32 atomics also cost ~1 us per thousand.)
#include "mfc_common.h"
#include <cstdlib>

namespace {

constexpr int TW = 16, TH = 16, HW = TW + 2, HH = TH + 2, NHALO = HW * HH;
constexpr int NT = 256, NWAVES = 4, RPW = TH / NWAVES;
constexpr int CS = 20;  // element stride of a 16-channel pixel row in LDS
constexpr float LN_EPS = 1e-6f;
constexpr float GRN_EPS = 1e-6f;
// Profiling ablations are COMPILE-TIME only (-DMFC_CNX_ABL=bits builds a library that computes wrong results on purpose;
// the shipped build defines nothing and no environment variable can switch any of this on): 1 no GELU arithmetic,
// 2 one of the five conv steps, 4 no LayerNorm ladder, 8 skip the row chain, 16 request only the first tile's DMA,
// 32 zero the gradient records.
#if defined(MFC_CNX_ABL)
constexpr bool ABL_NO_CHAIN = (MFC_CNX_ABL & 8) != 0, ABL_NO_DMA = (MFC_CNX_ABL & 16) != 0, ABL_NO_FLUSH = (MFC_CNX_ABL & 32) != 0;
#else
constexpr bool ABL_NO_CHAIN = false, ABL_NO_DMA = false, ABL_NO_FLUSH = false;
#endif

struct Dev {  // device pointers of mfc_cnx_params, by value
    const void* conv_w; const float* conv_b; const void* exp_w; const float* exp_b;
    const float* gamma; const float* beta; const void* con_w; const float* con_b; const float* ls;
};
struct DevG {
    float* conv_w; float* conv_b; float* exp_w; float* exp_b; float* gamma; float* beta;
    float* con_w; float* con_b; float* ls;
};

// LDS accesses of one wave execute in order, so a wave reading back its own scratch needs no s_waitcnt between
// the writes and the reads -- only the compiler must not reorder them
__device__ inline void lds_fence() { asm volatile("" ::: "memory"); }
// >= 5 wait states between dependent MFMAs of different shapes (see chain_row); s_nop 7 = 8 wait states
__device__ inline void mfma_shape_fence(f32x4& a, f32x4& b) { asm volatile("s_nop 7" : "+v"(a), "+v"(b)); }
// make a just-loaded value land HERE: left pending, the compiler's wait for it would sit at its first use in
// the tile loop's common path and (vmcnt being one in-order counter) drain the halo DMA on every tile
__device__ inline void land(const float& v) { asm volatile("" ::"v"(v)); }
__device__ inline void land(const f32x4& v) { asm volatile("" ::"v"(v)); }
__device__ inline void land(const s16x4& v) { asm volatile("" ::"v"(v)); }

// sum over the 4 lanes sharing (lane & 15): lanes 16 and 32 apart.  gfx950's
// v_permlane16_swap / v_permlane32_swap exchange 16-lane rows / 32-lane halves
// in one VALU instruction (no LDS round trip as ds_bpermute would need).
// (Measured alternative: the same sum as one v_mfma_f32_16x16x4_f32 with A = ones, which contracts over lane >> 4 on
// the mostly idle matrix pipe -- 3-7 % SLOWER on every kernel here: the dependent read of the MFMA result stalls the
// wave longer than the eight VALU instructions of the swap ladder take to issue.)
#if !defined(MFC_CNX_EXP)
#define MFC_CNX_EXP 0
#endif
__device__ inline float red_q(float v) {
    // inline asm: with the builtin and identical operands hipcc (ROCm 7.2) folds the two results
    // into one register.  "s_nop 1" = the 2 wait states a VALU-written operand needs before
    // v_permlane*_swap reads it (cdna_hip_programming.md T21).
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // a: rows 0,0,2,2  b: rows 1,1,3,3
    v = a + b;
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // a: lo,lo  b: hi,hi
    return a + b;
}
// Two independent sums at once: the ladder carries x in the even rows and y in the odd rows, so three swaps (and two
// adds) serve both -- same association ((q0+q1)+(q2+q3)) as red_q, bit-identical results.
__device__ inline void red_q2(float& x, float& y) {
    float a = x, b = y;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // a: x0 y0 x2 y2   b: x1 y1 x3 y3
    float t = a + b;                                                                 // x01 y01 x23 y23 (by row)
    a = t; b = t;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // a: x01 y01 x01 y01   b: x23 y23 x23 y23
    t = a + b;                                                                       // X Y X Y
    a = t; b = t;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // a: X X X X   b: Y Y Y Y
    x = a; y = b;
}
__device__ inline float red_m(float v) {  // sum over the 16 lanes sharing (lane >> 4)
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}

// ---- global <-> register helpers (4 / 16 consecutive channels) -------------
__device__ inline void ld4(const float* p, float v[4]) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
__device__ inline void ld4(const u16* p, float v[4]) {
    const s16x4 t = *reinterpret_cast<const s16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = bf16_to_f32((u16)t[i]);
}
__device__ inline void st4(float* p, const float v[4]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
}
__device__ inline void st4(u16* p, const float v[4]) {
    s16x4 f;
    make_frag(f, v[0], v[1], v[2], v[3]);
    *reinterpret_cast<s16x4*>(p) = f;
}
template <typename T> __device__ inline void ld16(const T* p, float v[16]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ld4(p + 4 * i, v + 4 * i);
}
__device__ inline void frag_raw(f32x4& f, float a, float b, float c, float d) { f = f32x4{a, b, c, d}; }
__device__ inline void frag_raw(s16x4& f, u16 a, u16 b, u16 c, u16 d) {
    f = s16x4{(short)a, (short)b, (short)c, (short)d};
}
template <typename F> __device__ inline void frag_of(F& f, const f32x4& v) { make_frag(f, v[0], v[1], v[2], v[3]); }

// B fragment of a row-major matrix W: element (k, n) at W[k*sk + n*sn];
// lane (q, r): k = k0 + 4q + i, n = n0 + r.
template <typename T>
__device__ inline typename Frag<T>::type load_bfrag(const T* W, int sk, int sn, int k0, int n0, int q, int r) {
    typename Frag<T>::type f;
    const T* p = W + (k0 + 4 * q) * sk + (n0 + r) * sn;
    frag_raw(f, p[0], p[sk], p[2 * sk], p[3 * sk]);
    return f;
}

// LayerNorm over the 16 channels of a pixel held as 4 values on each of the 4
// q-lanes (A layout).  flax LayerNorm: var = max(0, E[x^2]-E[x]^2), eps 1e-6.
__device__ inline void ln_fwd_a(const float v[4], float n[4], float& mean, float& rho) {
    float s = v[0] + v[1] + v[2] + v[3];
    float ss = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    red_q2(s, ss);
    mean = s * (1.0f / 16.0f);
    const float var = fmaxf(0.0f, ss * (1.0f / 16.0f) - mean * mean);
    rho = __builtin_amdgcn_rsqf(var + LN_EPS);   // argument >= 1e-6: none of rsqrtf's denormal rescaling (5 VALU) is needed
#pragma unroll
    for (int i = 0; i < 4; ++i) n[i] = (v[i] - mean) * rho;
}
// The same LayerNorm for an input whose channel mean is already zero (see stash_conv_w: the conv weights and bias are
// centred over the OUTPUT channels when they are stashed, so the MFMA chain delivers c1 - mean_c(c1) directly and the
// mean, its subtraction and half of the reduction ladder disappear from the per-pixel VALU work).
__device__ inline void ln_fwd_centered(const float v[4], float n[4], float& rho) {
    const float ss = red_q(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
    rho = __builtin_amdgcn_rsqf(ss * (1.0f / 16.0f) + LN_EPS);
#pragma unroll
    for (int i = 0; i < 4; ++i) n[i] = v[i] * rho;
}
// f32x4 forms (packed f32 VALU for the normalisation; the sum of squares stays a scalar fma chain)
__device__ inline f32x4 ln_fwd_centered4(f32x4 v, float& rho) {
#if defined(MFC_CNX_ABL) && (MFC_CNX_ABL & 4)
    const float ss = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];     // ablation: no cross-lane ladder
#else
    const float ss = red_q(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
#endif
    rho = __builtin_amdgcn_rsqf(ss * (1.0f / 16.0f) + LN_EPS);
    return v * splat4(rho);
}
__device__ inline f32x4 ln_jvp_centered4(f32x4 vd, f32x4 n, float rho) {
    const float dot = red_q(n[0] * vd[0] + n[1] * vd[1] + n[2] * vd[2] + n[3] * vd[3]) * (1.0f / 16.0f);
    return fma4(n, splat4(-dot), vd) * splat4(rho);
}
// ... and its tangent for a centred tangent vd (the tangent conv uses the same centred weights): mean(vd) = 0
__device__ inline void ln_jvp_centered(const float vd[4], const float n[4], float rho, float nd[4]) {
    const float dot = red_q(n[0] * vd[0] + n[1] * vd[1] + n[2] * vd[2] + n[3] * vd[3]) * (1.0f / 16.0f);
#pragma unroll
    for (int i = 0; i < 4; ++i) nd[i] = rho * (vd[i] - n[i] * dot);
}
// tangent of LayerNorm: nd = rho (vd_c - n mean(n vd_c))
__device__ inline void ln_jvp_a(const float vd[4], const float n[4], float rho, float nd[4]) {
    const float md = red_q(vd[0] + vd[1] + vd[2] + vd[3]) * (1.0f / 16.0f);
    float c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = vd[i] - md;
    const float dot = red_q(n[0] * c[0] + n[1] * c[1] + n[2] * c[2] + n[3] * c[3]) * (1.0f / 16.0f);
#pragma unroll
    for (int i = 0; i < 4; ++i) nd[i] = rho * (c[i] - n[i] * dot);
}
// backward of LayerNorm: dx = rho (dn - mean(dn) - n mean(dn n))
__device__ inline void ln_bwd_a(const float dn[4], const float n[4], float rho, float dx[4]) {
    float m1 = dn[0] + dn[1] + dn[2] + dn[3];
    float m2 = dn[0] * n[0] + dn[1] * n[1] + dn[2] * n[2] + dn[3] * n[3];
    red_q2(m1, m2);
    m1 *= 1.0f / 16.0f;
    m2 *= 1.0f / 16.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) dx[i] = rho * (dn[i] - m1 - n[i] * m2);
}

struct Geo {
    int64_t R; int s; int tilesX, tilesY; int64_t tilesPerImg, total, chunk;
    int kmax;   // most rows r a workgroup's contiguous tile range can touch = records per workgroup
};
inline Geo make_geo(int64_t R, int s, int64_t maxBlocks, int64_t& grid) {
    Geo g;
    g.R = R; g.s = s;
    g.tilesX = (s + TW - 1) / TW; g.tilesY = (s + TH - 1) / TH;
    g.tilesPerImg = (int64_t)g.tilesX * g.tilesY;
    g.total = R * g.tilesPerImg;
    grid = g.total < maxBlocks ? g.total : maxBlocks;
    g.chunk = (g.total + grid - 1) / grid;
    grid = (g.total + g.chunk - 1) / g.chunk;
    g.kmax = (int)((g.chunk + g.tilesPerImg - 2) / g.tilesPerImg) + 1;
    return g;
}

// Workspace records (floats).  Per (workgroup, k-th row of its range): REC_STATS forward statistics [S1 32 | S2 32],
// REC_DQ backward statistic [dq 32], REC_CONV [conv_w 2304 | dscale 16 | dshift 16].  Per workgroup: REC_MAIN
// [con_w 512 | exp_w 512 | ls 16 | exp_b 32], REC_TAIL [con_b 16 | conv_b 16 | grn_beta 32].
constexpr int REC_STATS = 64, REC_DQ = 32, REC_CONV = 9 * 256 + 32, REC_MAIN = 1024 + 48, REC_TAIL = 64;
inline Geo make_pix_geo(int64_t R, int s, int64_t maxBlocks, int64_t& grid);
inline int64_t ws_elems_for(int64_t R, int s, int64_t maxBlocks, bool pix = false) {
    int64_t grid;
    const Geo g = pix ? make_pix_geo(R, s, maxBlocks, grid) : make_geo(R, s, maxBlocks, grid);
    const int64_t a = grid * g.kmax * REC_CONV + grid * REC_TAIL, b = grid * REC_MAIN, c = grid * g.kmax * REC_STATS;
    return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

constexpr int WS_TILES = 6;   // y0, y1, dp1, n1, de0, de1: what the weight-gradient transposes of one tile row hold
__device__ inline void unfrag(const f32x4& f, float v[4]) { v[0] = f[0]; v[1] = f[1]; v[2] = f[2]; v[3] = f[3]; }
__device__ inline void unfrag(const s16x4& f, float v[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = bf16_to_f32((u16)f[i]);
}
struct TileCoord { int64_t r; int y0, x0; };
__device__ inline TileCoord tile_coord(const Geo& g, int64_t t) {
    TileCoord c;
    c.r = t / g.tilesPerImg;
    const int ti = (int)(t - c.r * g.tilesPerImg);
    c.y0 = (ti / g.tilesX) * TH;
    c.x0 = (ti % g.tilesX) * TW;
    return c;
}
// the tile after c in a workgroup's contiguous chunk (the 64-bit divisions of tile_coord cost more
// scalar instructions per tile than a whole tile row of the chain costs vector ones)
__device__ inline TileCoord tile_next(const Geo& g, TileCoord c) {
    c.x0 += TW;
    if (c.x0 >= g.tilesX * TW) {
        c.x0 = 0; c.y0 += TH;
        if (c.y0 >= g.tilesY * TH) { c.y0 = 0; c.r += 1; }
    }
    return c;
}

// ---------------------------------------------------------------------------
// Halo tiles by LDS-DMA.  The FiLM modulation h2 = (1+scale) h1 + shift is folded into the conv weights
// per row r (W' = diag(1+scale) Wc, bias' = bc + sum_taps Wc^T shift; border tiles mask the taps that
// fall outside the image), so a halo tile is a verbatim copy of h1 (and of its tangent) and goes
// global -> LDS with global_load_lds_dwordx4: no staging registers, no VALU, double-buffered so the
// request for tile t+1 is in flight during all of tile t.
// ---------------------------------------------------------------------------
__device__ uint4 g_zero16;   // 16 zero bytes: DMA source of halo pixels outside the image

// MFC_CNX_DRAIN=1 (default): the tile kernels wait for EVERYTHING they have in flight (vmcnt(0)) before the barrier that hands a
// DMA buffer over.  MFC_CNX_DRAIN=0 counts the stores issued after the request instead (vmcnt(S_VMEM)), which by the in-order rule
// of the counter should be equivalent -- but with it 6 of ~330 evaluations of the literal-size loss + reverse pass were not
// bitwise repeatable (one traced to this file's conv-gradient kernel: dh0 differed while every reduction agreed, i.e. the
// youngest DMA, the 1/sigma dwords, had not landed), against 0 of 120 with the full wait; the full wait costs <= 1.5 % of
// these kernels.
#ifndef MFC_CNX_DRAIN
#define MFC_CNX_DRAIN 1
#endif
typedef __attribute__((address_space(1))) void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t BUF_OOB = 0xFFFFFF00u;   // offset no buffer resource covers: load 0 / store dropped

__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
// 4 consecutive channels through a buffer resource (unconditional instructions: see the vmcnt note below)
__device__ inline void buf_st4(__amdgpu_buffer_rsrc_t rs, uint32_t off, const float v[4], const float*) {
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{__builtin_bit_cast(uint32_t, v[0]), __builtin_bit_cast(uint32_t, v[1]),
                                                 __builtin_bit_cast(uint32_t, v[2]), __builtin_bit_cast(uint32_t, v[3])},
                                           rs, off, 0, 0);
}
__device__ inline void buf_st4(__amdgpu_buffer_rsrc_t rs, uint32_t off, const float v[4], const u16*) {
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])}, rs, off, 0, 0);
}
__device__ inline f32x4 buf_ld_frag(__amdgpu_buffer_rsrc_t rs, uint32_t off, const float*) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}
__device__ inline s16x4 buf_ld_frag(__amdgpu_buffer_rsrc_t rs, uint32_t off, const u16*) {
    return __builtin_bit_cast(s16x4, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
}

__device__ inline void buf_st_frag(__amdgpu_buffer_rsrc_t rs, uint32_t off, const f32x4& f) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f), rs, off, 0, 0);
}
__device__ inline void buf_st_frag(__amdgpu_buffer_rsrc_t rs, uint32_t off, const s16x4& f) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f), rs, off, 0, 0);
}

template <typename T> struct Halo {
    static constexpr int EPC = 16 / sizeof(T);          // elements per 16-byte chunk
    static constexpr int CPP = 16 / EPC;                // chunks per pixel
    static constexpr int CHUNKS = NHALO * CPP;
    static constexpr int NI = (CHUNKS + NT - 1) / NT;   // DMA instructions per wave
    static constexpr int ELEMS = NHALO * 16;
    // LDS layout [HH][CPP][HW] of 16-byte chunks (chunk c of the HW pixels of a halo row are contiguous): the
    // chain reads channels 4q..4q+3 of 16 consecutive pixels, which then sit in 16 consecutive chunks -- no bank
    // conflicts.  (Pixel-major [HH][HW][CPP] would put pixels m and m+8 on the same banks.)
    __device__ static inline int off(int hy, int hx, int q) {
        const int c = (4 * q) / EPC, w = (4 * q) % EPC;
        return ((hy * CPP + c) * HW + hx) * EPC + w;
    }
    int hyx[NI];        // (hy << 8) | hx of this lane's chunk in instruction i, -1 past the end of the tile
    uint32_t relb[NI];  // byte offset of that chunk relative to the tile's HALO origin pixel (y0 - 1, x0 - 1): never negative
    __device__ inline void init(int s, int wave, int lane) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int ch = (i * NWAVES + wave) * 64 + lane;     // = LDS chunk slot
            const int hy = ch / (CPP * HW), rem = ch - hy * (CPP * HW);
            const int part = rem / HW, hx = rem - part * HW;
            hyx[i] = ch < CHUNKS ? ((hy << 8) | hx) : -1;
            relb[i] = (uint32_t)(((hy * s + hx) * 16 + part * EPC) * (int)sizeof(T));
        }
    }
    // request the halo of tile (y0, x0) of the [s, s, 16] image `img` into the LDS tile at byte address `dst`.
    // Interior tiles (the whole halo lies inside the image: ~90 % of them) take the scalar-base form of the DMA
    // instruction -- address = SGPR pair + the lane's constant offset -- so a request costs no vector ALU work at all;
    // tiles on the image border test every chunk and fetch the ones outside from 16 zero bytes.  Both forms issue
    // the same number of VMEM instructions (vmcnt note).
    __device__ inline void request(uint32_t dst, const T* img, int s, int y0, int x0, int wave) const {
        // halo origin; on a border tile it may lie before the image (only chunks inside the image are dereferenced)
        const char* base = reinterpret_cast<const char*>(img + ((int64_t)(y0 - 1) * s + (x0 - 1)) * 16);
#if MFC_CNX_EXP & 4
        const bool interior = false;
#else
        const bool interior = y0 > 0 && x0 > 0 && y0 + TH < s && x0 + TW < s;
#endif
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (hyx[i] >= 0) {
                // Inline asm, not __builtin_amdgcn_global_load_lds: the compiler cannot tell the two LDS buffers
                // apart and would drain vmcnt before every LDS read of the tile being computed.  (Hiding a VMEM
                // instruction from its in-order vmcnt model only makes the waits it inserts stricter.)
                const uint32_t lds_addr = dst + (i * NWAVES + wave) * 1024;   // 64 lanes x 16 bytes per instruction
                if (interior) {
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                                 :: "v"(relb[i]), "s"(base), "s"(lds_addr) : "memory", "m0");
                } else {
                    const int gy = y0 + (hyx[i] >> 8) - 1, gx = x0 + (hyx[i] & 255) - 1;
                    const bool in = (unsigned)gy < (unsigned)s && (unsigned)gx < (unsigned)s;
                    const void* gp = in ? static_cast<const void*>(base + relb[i]) : static_cast<const void*>(&g_zero16);
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                                 :: "v"(gp), "s"(lds_addr) : "memory", "m0");
                }
            }
        }
    }
};

// bf16 only: two conv taps per MFMA.  v_mfma_f32_16x16x32_bf16 contracts k = 32: lane (q, r) holds k = 8q .. 8q+7 of
// row / column r, so lanes q < 2 carry the 16 input channels of tap 2p and lanes q >= 2 those of tap 2p+1 (one 16-byte
// chunk of the halo pixel each).  Same MFMA cycles as two K16 steps, but half the instructions and half the time the
// MFMA holds the SIMD's issue port -- which is what these VALU-issue-bound kernels are short of.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ inline void mma32(f32x4& acc, const s16x8& a, const s16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
// The 32 -> 16 contraction of the ConvNeXt block as ONE K = 32 step (bf16): the two K = 16 operand fragments of the
// expanded channels 0-15 / 16-31 side by side -- which k a (lane, slot) pair carries is free as long as both operands
// agree.  Same matrix-pipe cycles as two K = 16 steps, one issue slot less.  (No accumulator chain mixes shapes here:
// the accumulator starts from the bias registers.)  Used in the reverse kernel (-3 %); in the forward apply kernels,
// whose other MFMAs are K = 16, it measured 1-2 % slower.
__device__ inline void mma_pair(f32x4& acc, const s16x4& a0, const s16x4& a1, const s16x4& b0, const s16x4& b1) {
    mma32(acc, s16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]},
          s16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]});
}
__device__ inline void mma_pair(f32x4& acc, const f32x4& a0, const f32x4& a1, const f32x4& b0, const f32x4& b1) {
    mma16(acc, a0, b0);      // fp32 storage: the exact f32 MFMA chain, as before
    mma16(acc, a1, b1);
}
__device__ inline s16x8 pack8(const float v[8]) {
    typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(s16x8, u32x4_{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                            pack_bf16x2(v[6], v[7])});
}

// vmcnt note.  vmcnt is one in-order counter over loads, stores and LDS-DMA.  The DMA for tile t+1 is
// requested at the top of tile t and awaited at its bottom with s_waitcnt vmcnt(S), S = the number of
// VMEM instructions this wave issues in between.  For S to be a compile-time constant those
// instructions are buffer loads / stores whose out-of-image lanes are steered to BUF_OOB instead of
// being branched around, and __builtin_amdgcn_sched_barrier pins them on their side of the request.

template <typename T> struct Lds2 {
    T* tile0;             // NTILE double-buffered halo tiles: tile k, buffer b at tile0 + (2k + b) * Halo<T>::ELEMS
    uint32_t tile0_addr;  // LDS byte address of tile0 (M0 of the DMA)
    T* wc0;               // [9][64 lanes][4]: the unscaled conv weights as A-operand fragments
    T* ws;                // per wave [6][16][CS] weight-gradient transpose scratch (backward main)
    float* fsc;           // [4][16] scale, shift, scaledot, shiftdot of the current row r
    float* rho;           // [2][TH*TW] per-pixel 1/sigma of the tile (backward conv)
    uint32_t rho_addr;
    __device__ inline const T* tile(int k, int b) const { return tile0 + (2 * k + b) * Halo<T>::ELEMS; }
    __device__ inline uint32_t tile_addr(int k, int b) const { return tile0_addr + (2 * k + b) * (uint32_t)(Halo<T>::ELEMS * sizeof(T)); }
};
template <typename T>
__host__ __device__ inline size_t lds2_bytes(int ntile, bool ws) {
    size_t b = (size_t)Halo<T>::ELEMS * sizeof(T) * 2 * ntile + (size_t)9 * 64 * 4 * sizeof(T);
    if (ws) b += (size_t)NWAVES * WS_TILES * 16 * CS * sizeof(T);
    b = (b + 15) & ~(size_t)15;
    return b + 64 * sizeof(float) + 2 * TH * TW * sizeof(float);
}
// cnx_bwd_conv_kernel: two double-buffered halo tiles, one compact 16 x 16 dout tile, two 16 x 16 tiles of 1/sigma
template <typename T>
__host__ __device__ inline size_t lds_bwd_conv_bytes() {
    return (size_t)Halo<T>::ELEMS * sizeof(T) * 4 + (size_t)TH * TW * 16 * sizeof(T) + 2 * TH * TW * sizeof(float);
}
template <typename T>
__device__ inline Lds2<T> carve2(unsigned char* base, int ntile, bool ws, int wave) {
    Lds2<T> l;
    T* p = (T*)base;
    const uint32_t base_addr = (uint32_t)(uintptr_t)(lvoid_t*)base;
    l.tile0 = p; l.tile0_addr = base_addr; p += 2 * ntile * Halo<T>::ELEMS;
    l.wc0 = p; p += 9 * 64 * 4;
    l.ws = p + (size_t)wave * WS_TILES * 16 * CS;
    size_t b = (size_t)Halo<T>::ELEMS * sizeof(T) * 2 * ntile + (size_t)9 * 64 * 4 * sizeof(T);
    if (ws) b += (size_t)NWAVES * WS_TILES * 16 * CS * sizeof(T);
    b = (b + 15) & ~(size_t)15;
    l.fsc = (float*)(base + b);
    l.rho = l.fsc + 64;
    l.rho_addr = base_addr + (uint32_t)b + 64 * (uint32_t)sizeof(float);
    return l;
}

// Per-lane weights / constants of the forward chain that do not depend on the row r.  Lane (q, m):
// m = pixel within the tile row, channel constants are those of channels 4q..4q+3.
template <typename T> struct FwdW {
    typedef typename Frag<T>::type frag_t;
    frag_t we[2], wp[2];
    float bc[4], be[2][4], bp[4], ls[4], gam[2][4], bet[2][4];
    int xo[5];   // bf16: element offset (within a tile row band) of this lane's 16-byte chunk for the tap pairs (2p, 2p+1);
                 // the fifth "pair" is tap 8 alone (its upper 16 k carry zero weights)
    __device__ inline void load(const Dev& d, int q, int m) {
#pragma unroll
        for (int pr = 0; pr < 5; ++pr) {
            const int t = pr < 4 ? 2 * pr + (q >> 1) : 8, dy = t / 3, dx = t - 3 * dy;
            xo[pr] = ((dy * Halo<T>::CPP + (q & 1)) * HW + m + dx) * Halo<T>::EPC;
        }
        const T* ew = (const T*)d.exp_w;  // [16][32]
        we[0] = load_bfrag<T>(ew, 32, 1, 0, 0, q, m);
        we[1] = load_bfrag<T>(ew, 32, 1, 0, 16, q, m);
        const T* pw = (const T*)d.con_w;  // [32][16]
        wp[0] = load_bfrag<T>(pw, 16, 1, 0, 0, q, m);
        wp[1] = load_bfrag<T>(pw, 16, 1, 16, 0, q, m);
        float bmean = 0.f;      // the conv bias is centred over the 16 output channels (see ln_fwd_centered)
#pragma unroll
        for (int c = 0; c < 16; ++c) bmean += d.conv_b[c];
        bmean *= 1.0f / 16.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bc[i] = d.conv_b[4 * q + i] - bmean; bp[i] = d.con_b[4 * q + i]; ls[i] = d.ls[4 * q + i];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                be[j][i] = d.exp_b[16 * j + 4 * q + i];
                gam[j][i] = d.gamma[16 * j + 4 * q + i];
                bet[j][i] = d.beta[16 * j + 4 * q + i];
            }
        }
        // see land(): every one of these is first used inside the tile loop
        land(we[0]); land(we[1]); land(wp[0]); land(wp[1]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            land(bc[i]); land(bp[i]); land(ls[i]);
#pragma unroll
            for (int j = 0; j < 2; ++j) { land(be[j][i]); land(gam[j][i]); land(bet[j][i]); }
        }
    }
};
// conv weights [tap][ic][oc] as A-operand fragments of the transposed product (row = oc, k = ic), one per lane --
// CENTRED over the output channels: W'[tap][ic][oc] = W - mean_oc W.  The LayerNorm that follows the conv is invariant
// to a per-pixel shift of all 16 channels, so LN(conv_W(x) + b) == LN(conv_W'(x) + b') with b' = b - mean(b), and the
// centred form needs no mean in the per-pixel chain (ln_fwd_centered).  Everything derived from the stash (the
// FiLM-folded weights and biases of RowW, the border-tile taps) is centred with it.  In bf16 storage W' is rounded to
// bf16 again: the residual channel mean is of the order of one bf16 ulp of the conv output, like any other rounding
// between the kernels of this path.  The reverse kernels take the gradient w.r.t. the ORIGINAL weights (the LayerNorm
// backward output already sums to zero over the channels, so dL/dW = h2^T dc1 is unchanged).
template <typename T>
__device__ inline void stash_conv_w(T* wc0, const Dev& d, int q, int m, int lane) {
    typedef typename Frag<T>::type frag_t;
    const T* cw = (const T*)d.conv_w;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float wv[4];
        unfrag(load_bfrag<T>(cw + t * 256, 16, 1, 0, 0, q, m), wv);      // lane (q, m): W[tap][ic = 4q + i][oc = m]
#pragma unroll
        for (int i = 0; i < 4; ++i) wv[i] -= red_m(wv[i]) * (1.0f / 16.0f);
        frag_t f;
        make_frag(f, wv[0], wv[1], wv[2], wv[3]);
        *reinterpret_cast<frag_t*>(wc0 + (t * 64 + lane) * 4) = f;
    }
}

// The row-r dependent part: FiLM folded into the conv.
// K32W: the conv taps as K = 32 steps (bf16 only; pays where the MFMA count is highest -- the forward JVP kernels)
template <typename T, bool JVP, bool K32W = false> struct RowW {
    typedef typename Frag<T>::type frag_t;
    static constexpr bool K32 = K32W && sizeof(T) == 2;
    frag_t wc[9];               // diag(1 + scale) Wc   (bf16: only tap 8 is used, the others live in w2)
    frag_t wcd[JVP ? 9 : 1];    // diag(scaledot) Wc
    s16x8 w2[K32 ? 5 : 1], w2d[(K32 && JVP) ? 5 : 1];   // bf16: the tap pairs (2p, 2p+1) and (8, -) as K = 32 A operands
    f32x4 b, bd;                // bc + sum_taps Wc^T shift (interior tiles) and its tangent
    frag_t shf, shdf;           // shift / shiftdot as a B operand (border tiles mask it per tap)
    float sc1[4], sh[4];        // 1 + scale, shift of channels 4q..4q+3 (FiLM of the centre pixel)
    __device__ inline void set(const T* wc0, const float bc[4], const float* sc, const float* shp, const float* scd,
                               const float* shd, int64_t r, int q, int lane) {
        float scd4[4] = {0.f, 0.f, 0.f, 0.f}, shd4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sc1[i] = 1.0f + sc[r * 16 + 4 * q + i];
            sh[i] = shp[r * 16 + 4 * q + i];
            if constexpr (JVP) { scd4[i] = scd[r * 16 + 4 * q + i]; shd4[i] = shd[r * 16 + 4 * q + i]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { land(sc1[i]); land(sh[i]); }
        make_frag(shf, sh[0], sh[1], sh[2], sh[3]);
        make_frag(shdf, shd4[0], shd4[1], shd4[2], shd4[3]);
        b = f32x4{bc[0], bc[1], bc[2], bc[3]};
        bd = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const frag_t w0 = *reinterpret_cast<const frag_t*>(wc0 + (t * 64 + lane) * 4);
            float wv[4];
            unfrag(w0, wv);
            if (!K32) make_frag(wc[t], wv[0] * sc1[0], wv[1] * sc1[1], wv[2] * sc1[2], wv[3] * sc1[3]);
            mma16(b, w0, shf);
            if constexpr (JVP) {
                if (!K32) make_frag(wcd[t], wv[0] * scd4[0], wv[1] * scd4[1], wv[2] * scd4[2], wv[3] * scd4[3]);
                mma16(bd, w0, shdf);
            }
        }
        if constexpr (K32) mfma_shape_fence(b, bd);   // b / bd (K16 results) become SrcC of K = 32 steps in chain_row
        if constexpr (K32) {
            // lane (q, m) of pair p: tap 2p + (q >> 1), input channels 8 (q & 1) .. +7, output channel m -- two of the
            // stashed K16 fragments (lane groups 2 (q & 1) and 2 (q & 1) + 1 of that tap, same m)
            const int m = lane & 15, h = q & 1;
            float s8[8], sd8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                s8[i] = 1.0f + sc[r * 16 + 8 * h + i];
                sd8[i] = JVP ? scd[r * 16 + 8 * h + i] : 0.f;
            }
#pragma unroll
            for (int pr = 0; pr < 5; ++pr) {
                const int t = pr < 4 ? 2 * pr + (q >> 1) : 8;
                const float live = (pr < 4 || q < 2) ? 1.0f : 0.0f;
                float wv[8], a8[8];
                unfrag(*reinterpret_cast<const frag_t*>(wc0 + (t * 64 + (2 * h) * 16 + m) * 4), wv);
                unfrag(*reinterpret_cast<const frag_t*>(wc0 + (t * 64 + (2 * h + 1) * 16 + m) * 4), wv + 4);
#pragma unroll
                for (int i = 0; i < 8; ++i) a8[i] = wv[i] * s8[i] * live;
                w2[pr] = pack8(a8);
                if constexpr (JVP) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) a8[i] = wv[i] * sd8[i] * live;
                    w2d[pr] = pack8(a8);
                }
            }
        }
    }
};

// Result of the chain up to gelu for one tile row (16 pixels), all in the
// pixel-on-lane layout: element i of a vector = channel 4q+i (e/g: 16j+4q+i).
template <typename T, bool JVP> struct RowFwd {
    typedef typename Frag<T>::type frag_t;
    float n1[4];         // LN(c1)
    float rho1;
    frag_t n1f, n1df;
    f32x4 g[2], gp[2], gd[2];   // gelu(e), gelu'(e) (when WG), tangent of g (when JVP)
};

// tile / tiled: dense halo of h1 / its tangent.  border: the tile touches the image edge (uniform).
template <typename T, bool JVP, bool WG, bool K32W = false>
__device__ inline void chain_row(const T* tile, const T* tiled, const T* wc0, const FwdW<T>& w, const RowW<T, JVP, K32W>& rw,
                                 bool border, int gy, int gx, int s, int y, int q, int m, int lane, RowFwd<T, JVP>& o) {
    typedef typename Frag<T>::type frag_t;
    f32x4 acc = rw.b, accd = rw.bd;
    if (border) {
        // zero padding applies to h2, not h1: only the taps inside the image carry the shift
        acc = f32x4{w.bc[0], w.bc[1], w.bc[2], w.bc[3]};
        accd = f32x4{0.f, 0.f, 0.f, 0.f};
        frag_t z;
        make_frag(z, 0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const bool in = (unsigned)(gy + dy - 1) < (unsigned)s && (unsigned)(gx + dx - 1) < (unsigned)s;
                const frag_t w0 = *reinterpret_cast<const frag_t*>(wc0 + ((dy * 3 + dx) * 64 + lane) * 4);
                mma16(acc, w0, in ? rw.shf : z);
                if constexpr (JVP) mma16(accd, w0, in ? rw.shdf : z);
            }
        // The K = 32 steps below continue these accumulators.  TOOLCHAIN HAZARD (root-caused in
        // tools/probe/mfma_mixed_shape.hip, measured on MI355X / ROCm 7.2): a dependent MFMA whose SrcC is the result
        // of an MFMA of a DIFFERENT shape (16x16x16 -> 16x16x32 or the reverse) needs >= 5 wait states in between;
        // hipcc emits none for that pair (it does for nothing else in this file: same-shape accumulate chains are
        // interlocked by the hardware), and the back-to-back pair returns wrong rows.  MFMA_SHAPE_FENCE supplies the
        // wait states as an asm statement that NAMES the accumulators, so the compiler cannot move either MFMA across it.
        // (This is also why tap 8 is a zero-padded K = 32 step: no K32 -> K16 link exists anywhere.)
        if constexpr (K32W && sizeof(T) == 2) mfma_shape_fence(acc, accd);
    }
    if constexpr (K32W && sizeof(T) == 2) {
        const int band = y * (Halo<T>::CPP * HW * Halo<T>::EPC);
#if defined(MFC_CNX_ABL) && (MFC_CNX_ABL & 2)
        constexpr int NPR = 1;                    // ablation: one of the five conv steps (LDS reads + MFMAs)
#else
        constexpr int NPR = 5;
#endif
#pragma unroll
        for (int pr = 0; pr < NPR; ++pr) {          // taps (2p, 2p+1) and (8, -): one K = 32 step each (see mma32)
            const s16x8 a = *reinterpret_cast<const s16x8*>(tile + band + w.xo[pr]);
            mma32(acc, rw.w2[pr], a);
            if constexpr (JVP) {
                const s16x8 ad = *reinterpret_cast<const s16x8*>(tiled + band + w.xo[pr]);
                mma32(accd, rw.w2[pr], ad);
                mma32(accd, rw.w2d[pr], a);
            }
        }
    } else {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int off = Halo<T>::off(y + dy, m + dx, q);
                const frag_t a = *reinterpret_cast<const frag_t*>(tile + off);
                mma16(acc, rw.wc[dy * 3 + dx], a);      // c1^T = W'^T h1^T
                if constexpr (JVP) {
                    const frag_t ad = *reinterpret_cast<const frag_t*>(tiled + off);
                    mma16(accd, rw.wc[dy * 3 + dx], ad);
                    mma16(accd, rw.wcd[dy * 3 + dx], a);
                }
            }
    }
    const f32x4 n1v = ln_fwd_centered4(acc, o.rho1);
#pragma unroll
    for (int i = 0; i < 4; ++i) o.n1[i] = n1v[i];
    make_frag(o.n1f, n1v[0], n1v[1], n1v[2], n1v[3]);
    if constexpr (JVP) {
        const f32x4 nd = ln_jvp_centered4(accd, n1v, o.rho1);
        make_frag(o.n1df, nd[0], nd[1], nd[2], nd[3]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        f32x4 e = ld_f32x4(w.be[j]);
        mma16(e, w.we[j], o.n1f);
#if defined(MFC_CNX_ABL) && (MFC_CNX_ABL & 1)
        o.g[j] = e; o.gp[j] = e;          // ablation: no GELU arithmetic
#else
        if constexpr (WG || JVP) gelu_both4(e, o.g[j], o.gp[j]);
        else o.g[j] = gelu4(e);
#endif
        if constexpr (JVP) {
            f32x4 ed = f32x4{0.f, 0.f, 0.f, 0.f};
            mma16(ed, w.we[j], o.n1df);
            o.gd[j] = ed * o.gp[j];
        }
    }
}

struct FwdArgs {
    Geo geo;
    const void* h0; const void* h0d;
    const float* sc; const float* sh; const float* scd; const float* shd;
    Dev p;
    float* S1; float* S2;          // stats mode
    const float* q; const float* qd;  // apply mode
    void* o; void* od;
    float* ws;   // stats mode: per-(workgroup, row) records of REC_STATS floats
    void* n1_out; float* rho1_out;   // stats mode, optional: n1 = LN(c1) [R, s, s, 16] (storage dtype) and its 1/sigma [R, s, s]
    void* n1d_out;                   // ... and (tangent rows) the tangent of n1
};

template <typename T, bool JVP>
__device__ inline void load_film(float* fsc, const float* sc, const float* sh, const float* scd,
                                 const float* shd, int64_t r) {
    if (threadIdx.x < 16) {
        fsc[threadIdx.x] = sc[r * 16 + threadIdx.x];
        fsc[16 + threadIdx.x] = sh[r * 16 + threadIdx.x];
        if constexpr (JVP) {
            fsc[32 + threadIdx.x] = scd[r * 16 + threadIdx.x];
            fsc[48 + threadIdx.x] = shd[r * 16 + threadIdx.x];
        }
    }
}
__device__ inline int rows_in_image(int s, int gy0) { const int n = s - gy0; return n < 0 ? 0 : (n > RPW ? RPW : n); }
__device__ inline bool tile_on_border(int s, int y0, int x0) { return y0 == 0 || x0 == 0 || y0 + TH >= s || x0 + TW >= s; }

// MODE 0: GRN statistics; MODE 1: apply GRN, contract, layer-scale, residual.
template <typename T, bool JVP, int MODE>
__global__ void __launch_bounds__(NT, sizeof(T) != 2 ? 1 : !JVP ? 4 : MODE == 0 ? 3 : 1)   // bf16: plain kernels <= 128 registers, tangent statistics <= 168
cnx_fwd_kernel(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename Frag<T>::type frag_t;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, m = lane & 15;
    Lds2<T> l = carve2<T>(smem, JVP ? 2 : 1, false, wave);   // tile 0: h1, tile 1: its tangent
    FwdW<T> w;
    w.load(a.p, q, m);
    if (wave == 0) stash_conv_w<T>(l.wc0, a.p, q, m, lane);
    const int s = a.geo.s;
    const int64_t img = (int64_t)s * s * 16;
    const T* h0 = (const T*)a.h0;
    const T* h0d = (const T*)a.h0d;
    Halo<T> hl;
    hl.init(s, wave, lane);
    constexpr bool K32 = JVP || MODE == 0;   // measured: -7 % on the JVP kernels, -3 % on the plain stats pass, 0 on plain apply
    RowW<T, JVP, K32> rw;
    // stores between a DMA request and its wait (statistics mode: the n1 / 1-sigma stores, out of bounds when not asked for)
    constexpr int S_VMEM = MODE == 1 ? RPW * (JVP ? 2 : 1) : RPW * (JVP ? 3 : 2);
    __amdgpu_buffer_rsrc_t rs_n1 = make_rsrc(nullptr, 0), rs_r1 = make_rsrc(nullptr, 0), rs_n1d = make_rsrc(nullptr, 0);

    int64_t t0 = (int64_t)blockIdx.x * a.geo.chunk;
    int64_t t1 = t0 + a.geo.chunk < a.geo.total ? t0 + a.geo.chunk : a.geo.total;
    int64_t rcur = -1;
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 s1[2] = {z4, z4}, s2[2] = {z4, z4};
    f32x4 gq[2] = {z4, z4}, qdv[2] = {z4, z4};       // apply mode: gamma + q (the GRN scale of this row r), qdot
    __amdgpu_buffer_rsrc_t rs_o = make_rsrc(nullptr, 0), rs_od = make_rsrc(nullptr, 0);

    int krow = 0;            // rows flushed so far = index of the next record of this workgroup
    float* red = l.rho;      // [NWAVES][REC_STATS] cross-wave scratch (the 1/sigma tile region is unused by this kernel)
    auto flush_stats = [&](int64_t) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v1 = red_m(s1[j][i]);
                    if (m == 0) red[wave * REC_STATS + 16 * j + 4 * q + i] = v1;
                    if constexpr (JVP) {
                        const float v2 = red_m(s2[j][i]);
                        if (m == 0) red[wave * REC_STATS + 32 + 16 * j + 4 * q + i] = v2;
                    }
                    s1[j][i] = 0.f; s2[j][i] = 0.f;
                }
            __syncthreads();
            if (threadIdx.x < (JVP ? 64 : 32)) {     // fixed order: wave 0 + wave 1 + wave 2 + wave 3
                const int c = threadIdx.x;
                const float v = ((red[c] + red[REC_STATS + c]) + red[2 * REC_STATS + c]) + red[3 * REC_STATS + c];
                a.ws[((int64_t)blockIdx.x * a.geo.kmax + krow) * REC_STATS + c] = v;
            }
            ++krow;
            __syncthreads();
        }
    };

    TileCoord tnext = tile_coord(a.geo, t0 < t1 ? t0 : 0);
    if (t0 < t1) {
        hl.request(l.tile_addr(0, 0), h0 + tnext.r * img, s, tnext.y0, tnext.x0, wave);
        if constexpr (JVP) hl.request(l.tile_addr(1, 0), h0d + tnext.r * img, s, tnext.y0, tnext.x0, wave);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int64_t t = t0; t < t1; ++t) {
        const TileCoord tc = tnext;
        tnext = tile_next(a.geo, tc);
        const int64_t r = tc.r;
        const int y0 = tc.y0, x0 = tc.x0;
        const int cur = (int)((t - t0) & 1);
        __syncthreads();  // every wave's share of tile t has landed (each waited for its own); tile t-1 fully consumed
        if (r != rcur) {
            if (rcur >= 0) flush_stats(rcur);
            rcur = r;
            rw.set(l.wc0, w.bc, a.sc, a.sh, a.scd, a.shd, r, q, lane);
            if constexpr (MODE == 0) {
                // (empty resources when the caller does not keep n1: every store is then dropped by the hardware)
                rs_n1 = make_rsrc(a.n1_out ? (const T*)a.n1_out + r * img : nullptr, a.n1_out ? (uint32_t)(img * sizeof(T)) : 0u);
                rs_r1 = make_rsrc(a.rho1_out ? a.rho1_out + r * (img / 16) : nullptr, a.rho1_out ? (uint32_t)(img / 16 * sizeof(float)) : 0u);
                if constexpr (JVP)
                    rs_n1d = make_rsrc(a.n1d_out ? (const T*)a.n1d_out + r * img : nullptr, a.n1d_out ? (uint32_t)(img * sizeof(T)) : 0u);
            }
            if constexpr (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        gq[j][i] = a.q[r * 32 + 16 * j + 4 * q + i];
                        if constexpr (JVP) qdv[j][i] = a.qd[r * 32 + 16 * j + 4 * q + i];
                    }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    land(gq[j]);
                    if constexpr (JVP) land(qdv[j]);
                    gq[j] = gq[j] + ld_f32x4(w.gam[j]);
                }
                rs_o = make_rsrc((const T*)a.o + r * img, (uint32_t)(img * sizeof(T)));
                if constexpr (JVP) {
                    rs_od = make_rsrc((const T*)a.od + r * img, (uint32_t)(img * sizeof(T)));
                    load_film<T, true>(l.fsc, a.sc, a.sh, a.scd, a.shd, r);
                    __syncthreads();
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < t1 && !ABL_NO_DMA) {
            hl.request(l.tile_addr(0, cur ^ 1), h0 + tnext.r * img, s, tnext.y0, tnext.x0, wave);
            if constexpr (JVP) hl.request(l.tile_addr(1, cur ^ 1), h0d + tnext.r * img, s, tnext.y0, tnext.x0, wave);
        }
        __builtin_amdgcn_sched_barrier(0);
        const T* tile = l.tile(0, cur);
        const T* tiled = l.tile(JVP ? 1 : 0, cur);
        const bool border = tile_on_border(s, y0, x0);
        const int gx = x0 + m;
        // Tile rows below the image (only in the last row of tiles: s % 16 of its 16 rows exist) are skipped -- nothing
        // of them is stored or summed.  nrows is wave-uniform.
        const int nrows = ABL_NO_CHAIN ? 0 : rows_in_image(s, y0 + wave * RPW);
#pragma unroll 1
        for (int ri = 0; ri < nrows; ++ri) {
            const int y = wave * RPW + ri;
            const int gy = y0 + y;
            const bool ok = gy < s && gx < s;
            RowFwd<T, JVP> f;
            chain_row<T, JVP, false, K32>(tile, tiled, l.wc0, w, rw, border, gy, gx, s, y, q, m, lane, f);
            if constexpr (MODE == 0) {
                if (ok) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        s1[j] = fma4(f.g[j], f.g[j], s1[j]);
                        if constexpr (JVP) s2[j] = fma4(f.g[j], f.gd[j], s2[j]);
                    }
                }
                // n1 = LN(conv(FiLM(h1))) exactly as the expansion consumed it, and its 1/sigma: the apply pass and the
                // reverse kernels start from these instead of repeating the conv and the LayerNorm (cnx_*_n1 kernels)
                const uint32_t pix = (uint32_t)(gy * s + gx);
                buf_st_frag(rs_n1, ok ? (uint32_t)((pix * 16 + 4 * q) * sizeof(T)) : BUF_OOB, f.n1f);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, f.rho1), rs_r1,
                                                      (ok && q == 0) ? pix * 4u : BUF_OOB, 0, 0);
                if constexpr (JVP) buf_st_frag(rs_n1d, ok ? (uint32_t)((pix * 16 + 4 * q) * sizeof(T)) : BUF_OOB, f.n1df);
            } else {
                f32x4 p1 = ld_f32x4(w.bp), p1d = z4;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    frag_t yf;
                    const f32x4 yv = fma4(f.g[j], gq[j], ld_f32x4(w.bet[j]));
                    make_frag(yf, yv[0], yv[1], yv[2], yv[3]);
                    mma16(p1, w.wp[j], yf);
                    if constexpr (JVP) {
                        frag_t ydf;
                        const f32x4 ydv = fma4(f.gd[j], gq[j], f.g[j] * qdv[j]);
                        make_frag(ydf, ydv[0], ydv[1], ydv[2], ydv[3]);
                        mma16(p1d, w.wp[j], ydf);
                    }
                }
                // residual: h2 of the centre pixel = FiLM(h1); stores are unconditional (vmcnt note)
                const int hoff = Halo<T>::off(y + 1, m + 1, q);
                const uint32_t goff = ok ? (uint32_t)((((int64_t)gy * s + gx) * 16 + 4 * q) * sizeof(T)) : BUF_OOB;
                float n[4], ov[4];
                ld4(tile + hoff, n);
                const f32x4 nv = ld_f32x4(n), ls4 = ld_f32x4(w.ls), sc14 = ld_f32x4(rw.sc1);
                const f32x4 o4 = fma4(p1, ls4, fma4(sc14, nv, ld_f32x4(rw.sh)));
#pragma unroll
                for (int i = 0; i < 4; ++i) ov[i] = o4[i];
                buf_st4(rs_o, goff, ov, (const T*)nullptr);
                if constexpr (JVP) {
                    float nd[4], scd4[4], shd4[4];
                    ld4(tiled + hoff, nd);
                    ld4(l.fsc + 32 + 4 * q, scd4);
                    ld4(l.fsc + 48 + 4 * q, shd4);
                    const f32x4 od4 = fma4(p1d, ls4, fma4(sc14, ld_f32x4(nd), fma4(ld_f32x4(scd4), nv, ld_f32x4(shd4))));
#pragma unroll
                    for (int i = 0; i < 4; ++i) ov[i] = od4[i];
                    buf_st4(rs_od, goff, ov, (const T*)nullptr);
                }
            }
        }
        {
            // the skipped rows' stores are still issued, steered out of bounds: the wait below counts them (vmcnt note)
            const float zv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int ri = nrows; ri < RPW; ++ri) {
                if constexpr (MODE == 1) {
                    buf_st4(rs_o, BUF_OOB, zv, (const T*)nullptr);
                    if constexpr (JVP) buf_st4(rs_od, BUF_OOB, zv, (const T*)nullptr);
                } else {
                    buf_st4(rs_n1, BUF_OOB, zv, (const T*)nullptr);
                    __builtin_amdgcn_raw_buffer_store_b32(0u, rs_r1, BUF_OOB, 0, 0);
                    if constexpr (JVP) buf_st4(rs_n1d, BUF_OOB, zv, (const T*)nullptr);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // this wave's share of tile t+1 has landed once at most the S_VMEM stores issued after the request are pending
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MFC_CNX_DRAIN ? 0 : S_VMEM) : "memory");
    }
    if (rcur >= 0) flush_stats(rcur);
}

// ---------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------
struct BwdArgs {
    Geo geo;
    const void* h0; const float* rho; const float* sc; const float* sh;
    Dev p; DevG g;
    const float* q; const float* kG;
    const void* dout; const void* dc1_in;
    float* dq; void* dc1; void* dh0; float* dsc; float* dsh;
    float* ws;   // partial-sum records (see REC_*)
};

// read a [16 pixel][CS] scratch tile as an operand that has the PIXEL as k:
// lane (q, r): elements (pixel 4q+i, channel r)
template <typename T>
__device__ inline typename Frag<T>::type pix_k_frag(const T* tile, int q, int r) {
    if constexpr (sizeof(T) == 2) {
        // gfx950 LDS transpose read: one instruction fetches the 4(pixel) x 16(channel) block of lane
        // group q column-major -- lane 4q'+p supplies the address of pixel 4q+q', channels 4p..4p+3
        const T* p = tile + (4 * q + (r >> 2)) * CS + 4 * (r & 3);
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
    } else {
        typename Frag<T>::type f;
        const T* p = tile + (4 * q) * CS + r;
        frag_raw(f, p[0], p[CS], p[2 * CS], p[3 * CS]);
        return f;
    }
}

// MODE 0: dq[r,ch] = sum dy*g1, dbeta += sum dy.
// MODE 1: dc1 + small-parameter gradients (con_w, ls, exp_w, exp_b; con_b and conv_b come from cnx_bwd_conv_kernel).
template <typename T, int MODE>
__global__ void __launch_bounds__(NT, sizeof(T) == 2 ? (MODE == 0 ? 4 : 2) : 1)   // bf16: <= 128 / 256 registers (4 / 2 waves per SIMD)
cnx_bwd_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename Frag<T>::type frag_t;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, m = lane & 15;
    Lds2<T> l = carve2<T>(smem, MODE == 1 ? 2 : 1, MODE == 1, wave);   // tile 0: h1; MODE 1: tile 1 = dout
    FwdW<T> w;
    w.load(a.p, q, m);
    if (wave == 0) stash_conv_w<T>(l.wc0, a.p, q, m, lane);
    // transposed 1x1 weights (A operands of the transposed products)
    const T* pw = (const T*)a.p.con_w;  // [32][16]: dy[e] = sum_c Wp[e][c] dp1[c]
    frag_t wpT[2] = {load_bfrag<T>(pw, 1, 16, 0, 0, q, m), load_bfrag<T>(pw, 1, 16, 0, 16, q, m)};
    const T* ew = (const T*)a.p.exp_w;  // [16][32]: dn1[c] = sum_e We[c][e] de[e]
    frag_t weT[2] = {load_bfrag<T>(ew, 1, 32, 0, 0, q, m), load_bfrag<T>(ew, 1, 32, 16, 0, q, m)};
    land(wpT[0]); land(wpT[1]); land(weT[0]); land(weT[1]);
    const int s = a.geo.s;
    const int64_t img = (int64_t)s * s * 16;
    const T* h0 = (const T*)a.h0;
    Halo<T> hl;
    hl.init(s, wave, lane);
    constexpr bool BK32 = MODE == 1;   // K = 32 conv steps where registers are not what limits occupancy (MODE 0 sits at 122 of 128)
    RowW<T, false, BK32> rw;
    // VMEM instructions between a DMA request and its wait.  MODE 0 prefetches dout through registers (RPW buffer
    // loads; it has no stores, so the compiler's own wait for them drains nothing, and a second DMA tile would cost
    // a workgroup per CU in LDS).  MODE 1 takes dout as a second DMA tile: with a register prefetch the compiler
    // would wait for most of the RPW dc1 stores at the top of every tile.
    constexpr int S_VMEM = RPW;

    int64_t t0 = (int64_t)blockIdx.x * a.geo.chunk;
    int64_t t1 = t0 + a.geo.chunk < a.geo.total ? t0 + a.geo.chunk : a.geo.total;
    int64_t rcur = -1;
    float qv[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, kg[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float dqp[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // (d grn_beta = Wp (ls * sum dout): cnx_bwd_conv_kernel)
    // MODE 1 accumulators
    f32x4 aWp[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}}, aWe[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    float dls[4] = {0.f, 0.f, 0.f, 0.f};
    float dbe[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    __amdgpu_buffer_rsrc_t rs_dc = make_rsrc(nullptr, 0);

    int krow = 0;
    float* red = l.rho;      // [NWAVES][REC_DQ] cross-wave scratch
    auto flush_row = [&](int64_t) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = red_m(dqp[j][i]);
                    if (m == 0) red[wave * REC_DQ + 16 * j + 4 * q + i] = v;
                    dqp[j][i] = 0.f;
                }
            __syncthreads();
            if (threadIdx.x < REC_DQ) {
                const int c = threadIdx.x;
                const float v = ((red[c] + red[REC_DQ + c]) + red[2 * REC_DQ + c]) + red[3 * REC_DQ + c];
                a.ws[((int64_t)blockIdx.x * a.geo.kmax + krow) * REC_DQ + c] = v;
            }
            ++krow;
            __syncthreads();
        }
    };

    const T* doutp = (const T*)a.dout;
    // MODE 0: dout of this wave's 4 tile rows, fetched one tile ahead (unconditional buffer loads: vmcnt note)
    auto load_dout = [&](frag_t d[RPW], const TileCoord& c) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(doutp + c.r * img, (uint32_t)(img * sizeof(T)));
#pragma unroll
        for (int ri = 0; ri < RPW; ++ri) {
            const int gy = c.y0 + wave * RPW + ri, gxx = c.x0 + m;
            const uint32_t off = (gy < s && gxx < s) ? (uint32_t)((((int64_t)gy * s + gxx) * 16 + 4 * q) * sizeof(T)) : BUF_OOB;
            d[ri] = buf_ld_frag(rs, off, (const T*)nullptr);
        }
    };
    frag_t dnext[RPW];
    TileCoord tnext = tile_coord(a.geo, t0 < t1 ? t0 : 0);
    if (t0 < t1) {
        hl.request(l.tile_addr(0, 0), h0 + tnext.r * img, s, tnext.y0, tnext.x0, wave);
        if constexpr (MODE == 1) hl.request(l.tile_addr(1, 0), doutp + tnext.r * img, s, tnext.y0, tnext.x0, wave);
        else load_dout(dnext, tnext);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int64_t t = t0; t < t1; ++t) {
        const TileCoord tc = tnext;
        tnext = tile_next(a.geo, tc);
        const int64_t r = tc.r;
        const int y0 = tc.y0, x0 = tc.x0;
        const int gx = x0 + m;
        const int cur = (int)((t - t0) & 1);
        __syncthreads();
        frag_t dcur[RPW];
        if constexpr (MODE == 0) {
#pragma unroll
            for (int ri = 0; ri < RPW; ++ri) { dcur[ri] = dnext[ri]; land(dcur[ri]); }
        }
        if (r != rcur) {
            if (rcur >= 0) flush_row(rcur);
            rcur = r;
            rw.set(l.wc0, w.bc, a.sc, a.sh, nullptr, nullptr, r, q, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    qv[j][i] = a.q[r * 32 + 16 * j + 4 * q + i];
                    if constexpr (MODE == 1) kg[j][i] = a.kG[r * 32 + 16 * j + 4 * q + i];
                }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) { land(qv[j][i]); if constexpr (MODE == 1) land(kg[j][i]); }
            if constexpr (MODE == 1) rs_dc = make_rsrc((const T*)a.dc1 + r * img, (uint32_t)(img * sizeof(T)));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < t1) {
            hl.request(l.tile_addr(0, cur ^ 1), h0 + tnext.r * img, s, tnext.y0, tnext.x0, wave);
            if constexpr (MODE == 1) hl.request(l.tile_addr(1, cur ^ 1), doutp + tnext.r * img, s, tnext.y0, tnext.x0, wave);
        }
        // (past the last tile this re-reads tile t's dout: the instruction count stays fixed)
        if constexpr (MODE == 0) load_dout(dnext, t + 1 < t1 ? tnext : tc);
        __builtin_amdgcn_sched_barrier(0);
        const T* tile = l.tile(0, cur);
        const T* dotile = l.tile(MODE == 1 ? 1 : 0, cur);
        const bool border = tile_on_border(s, y0, x0);
        const int nrows = rows_in_image(s, y0 + wave * RPW);   // rows below the image: dout = 0, no contribution -- skipped
#pragma unroll 1
        for (int ri = 0; ri < nrows; ++ri) {
            const int y = wave * RPW + ri;
            const int gy = y0 + y;
            const bool ok = gy < s && gx < s;
            float dov[4];
            if constexpr (MODE == 0) {
                unfrag(dcur[0], dov);
#pragma unroll
                for (int k = 0; k + 1 < RPW; ++k) dcur[k] = dcur[k + 1];   // rotate: static register indices
            } else {
                ld4(dotile + Halo<T>::off(y + 1, m + 1, q), dov);          // zero outside the image
            }
            RowFwd<T, false> f;
            chain_row<T, false, MODE == 1, BK32>(tile, nullptr, l.wc0, w, rw, border, gy, gx, s, y, q, m, lane, f);
            float dp1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) dp1[i] = dov[i] * w.ls[i];
            frag_t dp1f;
            make_frag(dp1f, dp1[0], dp1[1], dp1[2], dp1[3]);
            f32x4 dy[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                dy[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                mma16(dy[j], wpT[j], dp1f);      // dy^T = Wp dp1^T
            }
            if constexpr (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) dqp[j][i] += dy[j][i] * f.g[j][i];
            } else {
                // y, p1 (needed for dW_contract and d layer_scale)
                float yv[2][4];
                f32x4 p1 = f32x4{w.bp[0], w.bp[1], w.bp[2], w.bp[3]};
                frag_t yf[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) yv[j][i] = f.g[j][i] * (w.gam[j][i] + qv[j][i]) + w.bet[j][i];
                    make_frag(yf[j], yv[j][0], yv[j][1], yv[j][2], yv[j][3]);
                }
                mma_pair(p1, w.wp[0], w.wp[1], yf[0], yf[1]);
#pragma unroll
                for (int i = 0; i < 4; ++i) dls[i] += dov[i] * p1[i];
                // d gelu / d expand
                frag_t def[2];
                f32x4 dn1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float de[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float dg = dy[j][i] * (w.gam[j][i] + qv[j][i]) + f.g[j][i] * kg[j][i];
                        de[i] = ok ? dg * f.gp[j][i] : 0.f;
                        dbe[j][i] += de[i];
                    }
                    make_frag(def[j], de[0], de[1], de[2], de[3]);
                }
                mma_pair(dn1, weT[0], weT[1], def[0], def[1]);      // dn1^T = We de^T
                float dn[4] = {dn1[0], dn1[1], dn1[2], dn1[3]}, dc[4];
                ln_bwd_a(dn, f.n1, f.rho1, dc);
                buf_st4(rs_dc, ok ? (uint32_t)((((int64_t)gy * s + gx) * 16 + 4 * q) * sizeof(T)) : BUF_OOB, dc, (const T*)nullptr);
                // weight gradients contract over the 16 pixels of the row: one batched transpose
                // (pixel-on-lane -> pixel-as-k) of y0, y1, dp1, n1, de0, de1 through the wave scratch
                *reinterpret_cast<frag_t*>(l.ws + (0 * 16 + m) * CS + 4 * q) = yf[0];
                *reinterpret_cast<frag_t*>(l.ws + (1 * 16 + m) * CS + 4 * q) = yf[1];
                *reinterpret_cast<frag_t*>(l.ws + (2 * 16 + m) * CS + 4 * q) = dp1f;
                *reinterpret_cast<frag_t*>(l.ws + (3 * 16 + m) * CS + 4 * q) = f.n1f;
                *reinterpret_cast<frag_t*>(l.ws + (4 * 16 + m) * CS + 4 * q) = def[0];
                *reinterpret_cast<frag_t*>(l.ws + (5 * 16 + m) * CS + 4 * q) = def[1];
                lds_fence();
                const frag_t tdp = pix_k_frag<T>(l.ws + 2 * 16 * CS, q, m);
                const frag_t tn1 = pix_k_frag<T>(l.ws + 3 * 16 * CS, q, m);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const frag_t ty = pix_k_frag<T>(l.ws + j * 16 * CS, q, m);
                    const frag_t tde = pix_k_frag<T>(l.ws + (4 + j) * 16 * CS, q, m);
                    mma16(aWp[j], ty, tdp);    // [e][c] += y^T dp1
                    mma16(aWe[j], tn1, tde);   // [c][e] += n1^T de
                }
                lds_fence();
            }
        }
        if constexpr (MODE == 1) {       // the skipped rows' dc1 stores, out of bounds: the wait below counts them (vmcnt note)
            const float zv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int ri = nrows; ri < RPW; ++ri) buf_st4(rs_dc, BUF_OOB, zv, (const T*)nullptr);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MFC_CNX_DRAIN ? 0 : S_VMEM) : "memory");   // this wave's share of tile t+1 has landed
    }
    if (rcur >= 0) flush_row(rcur);
    if constexpr (MODE == 1) {
        // the four waves add their partial sums into the LDS scratch one after the other (fixed order), then the
        // workgroup stores its record; cnx_reduce_blocks_kernel sums the records in workgroup order
        float* scratch = (float*)const_cast<T*>(l.tile(0, 0));   // con_w [32][16] | exp_w [16][32] | ls [16] | exp_b [32]
        float vls[4], vbe[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            vls[i] = red_m(dls[i]);
#pragma unroll
            for (int j = 0; j < 2; ++j) vbe[j][i] = red_m(dbe[j][i]);
        }
        __syncthreads();                                          // every wave is past its last tile read
        for (int i = threadIdx.x; i < REC_MAIN; i += NT) scratch[i] = 0.f;
        __syncthreads();
        for (int wv = 0; wv < NWAVES; ++wv) {
            if (wave == wv) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        scratch[(16 * j + 4 * q + e) * 16 + m] += aWp[j][e];
                        scratch[512 + (4 * q + e) * 32 + 16 * j + m] += aWe[j][e];
                    }
                if (m == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        scratch[1024 + 4 * q + i] += vls[i];
#pragma unroll
                        for (int j = 0; j < 2; ++j) scratch[1040 + 16 * j + 4 * q + i] += vbe[j][i];
                    }
                }
            }
            __syncthreads();
        }
        float* rec = a.ws + (int64_t)blockIdx.x * REC_MAIN;
        for (int i = threadIdx.x; i < REC_MAIN; i += NT) rec[i] = ABL_NO_FLUSH ? 0.f : scratch[i];
    }
}

// read 16 consecutive halo pixels (row hy, columns hx0..hx0+15) of a DMA tile as an operand that has the PIXEL as k:
// lane (q, r): elements (pixel hx0+4q+i, channel r)
template <typename T>
__device__ inline typename Frag<T>::type pix_k_tile(const T* tile, int hy, int hx0, int q, int r) {
    if constexpr (sizeof(T) == 2) {
        // LDS transpose read: lane 4q'+p of group q supplies the address of pixel 4q+q', channels 4p..4p+3
        const T* p = tile + Halo<T>::off(hy, hx0 + 4 * q + (r >> 2), r & 3);
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
    } else {
        typename Frag<T>::type f;
        const int c = r >> 2, w = r & 3;
        frag_raw(f, tile[Halo<T>::off(hy, hx0 + 4 * q, c) + w], tile[Halo<T>::off(hy, hx0 + 4 * q + 1, c) + w],
                 tile[Halo<T>::off(hy, hx0 + 4 * q + 2, c) + w], tile[Halo<T>::off(hy, hx0 + 4 * q + 3, c) + w]);
        return f;
    }
}

// backward pass 3: conv3x3 transpose + conv weight gradient + FiLM/LN0 backward.
// Both halo tiles (h1 and dc1) are verbatim copies and arrive by LDS-DMA (see Halo).  With h2 = (1+scale) h1 +
// shift the conv weight gradient of one row r splits into
//     dWc[tap][ic][oc] = (1+scale[ic]) * A[tap][ic][oc] + shift[ic] * B[tap][oc],
//     A = sum_p h1[p+tap][ic] dc1[p][oc],   B = sum_{p: p+tap inside the image} dc1[p][oc]
// A is the 9 pixel-contracting MFMAs on the raw tile; B is ONE more MFMA whose A operand is the 0/1 tap-validity
// mask (row = tap, k = pixel; all ones on interior tiles).  Both are flushed when the workgroup's row r changes.
template <typename T>
__global__ void __launch_bounds__(NT, sizeof(T) == 2 ? 3 : 1)   // bf16: <= 168 registers, three workgroups per CU (LDS allows three)
cnx_bwd_conv_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename Frag<T>::type frag_t;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, m = lane & 15;
    // LDS: h1 and dc1 halo tiles (double-buffered DMA), then dout and 1/sigma of the tile's own 16 x 16 pixels.  dout needs
    // no halo and no second buffer: every wave DMA-writes exactly the four rows it reads itself, so it takes its rows
    // into registers at the top of a tile and requests the next tile's rows into the same place -- 51.7 KB per workgroup
    // instead of 69 KB, i.e. three workgroups per CU instead of two (this kernel is the one close to its HBM bound).
    Lds2<T> l = carve2<T>(smem, 2, false, wave);   // tile 0: h1 halo, tile 1: dc1 halo
    T* const dout_c = l.tile0 + 4 * Halo<T>::ELEMS;                       // [TH][CPP][TW] 16-byte chunks
    const uint32_t dout_addr = l.tile0_addr + 4 * (uint32_t)(Halo<T>::ELEMS * sizeof(T));
    l.rho = reinterpret_cast<float*>(dout_c + TH * TW * 16);
    l.rho_addr = dout_addr + (uint32_t)(TH * TW * 16 * sizeof(T));
    const T* cw = (const T*)a.p.conv_w;  // [tap][ic][oc]
    frag_t wcT[9];  // A[row=ic][k=oc] of the transposed product dh2^T = Wc dc1^T
#pragma unroll
    for (int t = 0; t < 9; ++t) { wcT[t] = load_bfrag<T>(cw + t * 256, 1, 16, 0, 0, q, m); land(wcT[t]); }
    const int s = a.geo.s;
    const int64_t img = (int64_t)s * s * 16;
    const T* h0 = (const T*)a.h0;
    const T* dc1 = (const T*)a.dc1_in;
    Halo<T> hl;
    hl.init(s, wave, lane);
    constexpr int S_VMEM = RPW;   // dh0 stores issued after a DMA request (dout and 1/sigma arrive by DMA as well)

    int64_t t0 = (int64_t)blockIdx.x * a.geo.chunk;
    int64_t t1 = t0 + a.geo.chunk < a.geo.total ? t0 + a.geo.chunk : a.geo.total;
    int64_t rcur = -1;
    f32x4 aWc[9], aB = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) aWc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dscp[4] = {0.f, 0.f, 0.f, 0.f}, dshp[4] = {0.f, 0.f, 0.f, 0.f};
    float sc1[4] = {1.f, 1.f, 1.f, 1.f}, shr[4] = {0.f, 0.f, 0.f, 0.f};         // 1 + scale, shift of channels 4q..4q+3, row rcur
    float dsum[4] = {0.f, 0.f, 0.f, 0.f}, dcsum[4] = {0.f, 0.f, 0.f, 0.f};   // -> d con_b (x ls), d conv_b
    // tap-validity mask rows: lane row m = tap m (dy = m / 3, dx = m % 3), rows 9..15 are zero
    const int tdy = m / 3, tdx = m - 3 * tdy;
    frag_t mk_int;
    { const float one = m < 9 ? 1.0f : 0.0f; make_frag(mk_int, one, one, one, one); }
    __amdgpu_buffer_rsrc_t rs_dh = make_rsrc(nullptr, 0);

    // Gradient flushes: the four waves add their partial sums into the LDS scratch one after the other (fixed order),
    // then the workgroup stores record k of its range (conv_w | dscale | dshift of row r); the follow-up kernels sum
    // the records in workgroup order.  `scratch` is a DMA buffer nothing is using at that point (see the call sites).
    int krow = 0;
    auto flush_row = [&](int64_t, float* scratch) {
        constexpr int NW = 9 * 256;
        float v1[4], v2[4], bt[9];
#pragma unroll
        for (int i = 0; i < 4; ++i) { v1[i] = red_m(dscp[i]); v2[i] = red_m(dshp[i]); dscp[i] = 0.f; dshp[i] = 0.f; }
#pragma unroll
        for (int t = 0; t < 9; ++t) bt[t] = __shfl(aB[t & 3], (t >> 2) * 16 + m);   // B[tap t][oc = m]
        __syncthreads();
        for (int i = threadIdx.x; i < NW + 32; i += NT) scratch[i] = 0.f;
        __syncthreads();
        for (int wv = 0; wv < NWAVES; ++wv) {
            if (wave == wv) {
                if (m == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { scratch[NW + 4 * q + i] += v1[i]; scratch[NW + 16 + 4 * q + i] += v2[i]; }
                }
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        scratch[(t * 16 + 4 * q + e) * 16 + m] += sc1[e] * aWc[t][e] + shr[e] * bt[t];
            }
            __syncthreads();
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) aWc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        aB = f32x4{0.f, 0.f, 0.f, 0.f};
        float* rec = a.ws + ((int64_t)blockIdx.x * a.geo.kmax + krow) * REC_CONV;
        for (int i = threadIdx.x; i < NW + 32; i += NT) rec[i] = ABL_NO_FLUSH ? 0.f : scratch[i];
        ++krow;
        __syncthreads();
    };

    const T* doutp = (const T*)a.dout;
    // 1/sigma of the tile's 16 x 16 pixels: one 4-byte LDS-DMA lane per pixel (pixel p = 64 wave + lane)
    auto request_rho = [&](uint32_t dst, const TileCoord& c) {
        const int pidx = wave * 64 + lane, gy = c.y0 + (pidx >> 4), gxx = c.x0 + (pidx & 15);
        const float* gp = (gy < s && gxx < s) ? a.rho + (c.r * s + gy) * (int64_t)s + gxx : reinterpret_cast<const float*>(&g_zero16);
        const uint32_t lds_addr = dst + wave * 256;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(gp), "s"(lds_addr) : "memory", "m0");
    };
    // dout rows RPW*wave .. +RPW-1 of the tile: CPP DMA instructions per wave, chunk slot = 64 i + lane of the wave's
    // [RPW][CPP][TW] region (the layout the row loop reads: 16 consecutive pixels of one chunk index are contiguous)
    constexpr int DCPP = Halo<T>::CPP, DEPC = Halo<T>::EPC;
    auto request_dout = [&](const TileCoord& c) {
#pragma unroll
        for (int i = 0; i < DCPP; ++i) {
            const int slot = i * 64 + lane, yl = slot / (DCPP * TW), rem = slot - yl * (DCPP * TW);
            const int part = rem / TW, px = rem - part * TW;
            const int gy = c.y0 + wave * RPW + yl, gxx = c.x0 + px;
            const T* gp = (gy < s && gxx < s) ? doutp + c.r * img + ((int64_t)gy * s + gxx) * 16 + part * DEPC
                                              : reinterpret_cast<const T*>(&g_zero16);
            const uint32_t lds_addr = dout_addr + (uint32_t)((wave * DCPP + i) * 1024);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gp), "s"(lds_addr) : "memory", "m0");
        }
    };
    auto request_all = [&](int b, const TileCoord& c) {
        hl.request(l.tile_addr(0, b), h0 + c.r * img, s, c.y0, c.x0, wave);
        hl.request(l.tile_addr(1, b), dc1 + c.r * img, s, c.y0, c.x0, wave);
        request_dout(c);
        request_rho(l.rho_addr + b * (uint32_t)(TH * TW * sizeof(float)), c);
    };
    TileCoord tnext = tile_coord(a.geo, t0 < t1 ? t0 : 0);
    if (t0 < t1) request_all(0, tnext);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int64_t t = t0; t < t1; ++t) {
        const TileCoord tc = tnext;
        tnext = tile_next(a.geo, tc);
        const int64_t r = tc.r;
        const int y0 = tc.y0, x0 = tc.x0;
        const int gx = x0 + m;
        const int cur = (int)((t - t0) & 1);
        __syncthreads();
        if (r != rcur) {
            // (the other DMA buffer of tile 0 is idle here: tile t-1 is consumed, tile t+1 not yet requested)
            if (rcur >= 0) flush_row(rcur, (float*)const_cast<T*>(l.tile(0, cur ^ 1)));
            rcur = r;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sc1[i] = 1.0f + a.sc[r * 16 + 4 * q + i];
                shr[i] = a.sh[r * 16 + 4 * q + i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { land(sc1[i]); land(shr[i]); }
            rs_dh = make_rsrc((const T*)a.dh0 + r * img, (uint32_t)(img * sizeof(T)));
        }
        // this wave's RPW rows of dout (zero outside the image), taken before the next tile's rows replace them
        frag_t dor[RPW];
#pragma unroll
        for (int ri = 0; ri < RPW; ++ri) {
            const int cq = (4 * q) / DEPC, wq = (4 * q) % DEPC;
            dor[ri] = *reinterpret_cast<const frag_t*>(dout_c + (((wave * RPW + ri) * DCPP + cq) * TW + m) * DEPC + wq);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < t1) request_all(cur ^ 1, tnext);
        __builtin_amdgcn_sched_barrier(0);
        const T* tile = l.tile(0, cur);
        const T* dtile = l.tile(1, cur);
        const float* rtile = l.rho + cur * (TH * TW);
        const bool border = tile_on_border(s, y0, x0);
        const int nrows = rows_in_image(s, y0 + wave * RPW);   // rows below the image: dout = dc1 = 0, no contribution -- skipped
#pragma unroll 1
        for (int ri = 0; ri < nrows; ++ri) {
            const int y = wave * RPW + ri;
            const int gy = y0 + y;
            const bool ok = gy < s && gx < s;
            float dov[4];
            static_assert(RPW == 4, "row select below");
            unfrag(ri == 0 ? dor[0] : ri == 1 ? dor[1] : ri == 2 ? dor[2] : dor[3], dov);   // (ri is wave-uniform)
            const float rho = rtile[y * TW + m];
            // dh2 = conv^T(dc1): h2[p] feeds c1[p - (i-1, j-1)] through K[i][j]
            f32x4 dh = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const frag_t ad = *reinterpret_cast<const frag_t*>(dtile + Halo<T>::off(y + 2 - i, m + 2 - j, q));
                    mma16(dh, wcT[i * 3 + j], ad);
                    if (i == 1 && j == 1) {   // centre tap: this lane's own dc1 (zero outside the image)
                        float dcv[4];
                        unfrag(ad, dcv);
#pragma unroll
                        for (int k = 0; k < 4; ++k) dcsum[k] += dcv[k];
                    }
                }
#pragma unroll
            for (int k = 0; k < 4; ++k) dsum[k] += dov[k];
            // A[tap][ic][oc] += sum_pixels h1[p + tap][ic] dc1[p][oc];  B[tap][oc] += sum_{valid} dc1[p][oc]
            {
                const frag_t bd = pix_k_tile<T>(dtile, y + 1, 1, q, m);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) mma16(aWc[i * 3 + j], pix_k_tile<T>(tile, y + i, j, q, m), bd);
                frag_t mk = mk_int;
                if (border) {
                    const bool rowin = (unsigned)(gy + tdy - 1) < (unsigned)s && m < 9;
                    float mv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) mv[i] = (rowin && (unsigned)(x0 + 4 * q + i + tdx - 1) < (unsigned)s) ? 1.0f : 0.0f;
                    make_frag(mk, mv[0], mv[1], mv[2], mv[3]);
                }
                mma16(aB, mk, bd);
            }
            {
                float h1[4], d2[4], dh1[4], dx[4];
                ld4(tile + Halo<T>::off(y + 1, m + 1, q), h1);
#pragma unroll
                for (int i = 0; i < 4; ++i) d2[i] = dh[i] + dov[i];  // residual branch o = ... + h2
                if (ok) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { dscp[i] += d2[i] * h1[i]; dshp[i] += d2[i]; }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) dh1[i] = d2[i] * sc1[i];
                ln_bwd_a(dh1, h1, rho, dx);
                buf_st4(rs_dh, ok ? (uint32_t)((((int64_t)gy * s + gx) * 16 + 4 * q) * sizeof(T)) : BUF_OOB, dx, (const T*)nullptr);
            }
        }
        {                                // the skipped rows' stores, out of bounds: the wait below counts them (vmcnt note)
            const float zv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int ri = nrows; ri < RPW; ++ri) buf_st4(rs_dh, BUF_OOB, zv, (const T*)nullptr);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MFC_CNX_DRAIN ? 0 : S_VMEM) : "memory");   // this wave's share of tile t+1 has landed
    }
    float* scratch = (float*)const_cast<T*>(l.tile(1, 0));   // (flush_row's first barrier: every wave is past its last dc1 read)
    if (rcur >= 0) flush_row(rcur, scratch);
    // records this workgroup's range did not reach: zeros (the reduction walks every record)
    for (int k = krow; k < a.geo.kmax; ++k) {
        float* rec = a.ws + ((int64_t)blockIdx.x * a.geo.kmax + k) * REC_CONV;
        for (int i = threadIdx.x; i < REC_CONV; i += NT) rec[i] = 0.f;
    }
    // con_b (16), conv_b (16), grn_beta (32): d con_b = ls * sum dout, d conv_b = sum dc1,
    // d grn_beta[e] = sum_pixels dy[e] = sum_c Wp[e][c] (ls[c] sum_pixels dout[c])
    float vb[4], vc[4], vbeta[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        vb[i] = red_m(dsum[i]) * a.p.ls[4 * q + i];    // this wave's sum of dp1[c], c = 4q + i (on every lane)
        vc[i] = red_m(dcsum[i]);
    }
    {
        const T* pw = (const T*)a.p.con_w;   // [32][16]
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += St<T>::ld(pw + (16 * j + m) * 16 + 4 * q + i) * vb[i];
            vbeta[j] = red_q(acc);
        }
    }
    __syncthreads();
    if (threadIdx.x < REC_TAIL) scratch[threadIdx.x] = 0.f;
    __syncthreads();
    for (int wv = 0; wv < NWAVES; ++wv) {
        if (wave == wv) {
            if (m == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { scratch[4 * q + i] += vb[i]; scratch[16 + 4 * q + i] += vc[i]; }
            }
            if (q == 0) { scratch[32 + m] += vbeta[0]; scratch[48 + m] += vbeta[1]; }
        }
        __syncthreads();
    }
    if (threadIdx.x < REC_TAIL)
        a.ws[(int64_t)gridDim.x * a.geo.kmax * REC_CONV + (int64_t)blockIdx.x * REC_TAIL + threadIdx.x] =
            ABL_NO_FLUSH ? 0.f : scratch[threadIdx.x];
}

// ---------------------------------------------------------------------------
// "From n1" kernels.  The statistics pass can keep n1 = LN(conv3x3(FiLM(h1))) -- the 16-channel map the expansion
// consumes -- its 1/sigma and (tangent rows) its tangent (mfc_cnx_stats_save).  Everything downstream of n1 is PER
// PIXEL (1x1 convs, GELU, GRN, layer scale, residual), so the apply pass and the two reverse kernels that used to
// repeat the conv and the LayerNorm become plain streaming kernels: no halo, no tiles, no workgroup barrier in the loop.
//
// Work unit: one "step" = 128 consecutive pixels of one image's flattened [s*s, 16] map, 32 per wave = two MFMA
// column groups of 16 pixels.  An image has spi = ceil(s*s / 128) steps (the last one partly out of bounds: those
// pixels load zeros and store nowhere); a workgroup walks a contiguous range of steps; per-row (r) sums are flushed when
// the range crosses into the next image, exactly like the tile kernels (same records, same fixed-order reduce kernels).
//
// HBM access is 16 bytes per lane in both directions (PixIO): a wave's 32-pixel span of a tensor is one contiguous
// 1 KB (bf16) / 2 KB (fp32) piece, fetched by LDS-DMA (global_load_lds_dwordx4, double-buffered, requested one step
// ahead: no staging registers, no VALU on full spans -- the address is a scalar base + the lane's constant offset) into
// a wave-private LDS buffer, from which the MFMA B-operand fragments (channels 4q..4q+3 of pixel m: 8 / 16 bytes) are
// read conflict-free; results go the other way -- fragments into a wave-private LDS image, 16-byte chunks out of it
// into buffer stores.  (The first version loaded the 8-byte bf16 fragments straight from global memory: 512 bytes per
// wave instruction reached 4.7 TB/s in the apply kernel; see DESIGN.md for the measured difference.)
// ---------------------------------------------------------------------------
constexpr int PIX_STEP = 128, PIX_WAVE = PIX_STEP / NWAVES;     // pixels per workgroup / per wave and step
static_assert(PIX_WAVE == 32, "two 16-pixel MFMA column groups per wave");

inline Geo make_pix_geo(int64_t R, int s, int64_t maxBlocks, int64_t& grid) {
    Geo g;
    g.R = R; g.s = s; g.tilesX = 0; g.tilesY = 0;
    g.tilesPerImg = ((int64_t)s * s + PIX_STEP - 1) / PIX_STEP;          // steps per image
    g.total = R * g.tilesPerImg;
    grid = g.total < maxBlocks ? g.total : maxBlocks;
    g.chunk = (g.total + grid - 1) / grid;
    grid = (g.total + g.chunk - 1) / g.chunk;
    g.kmax = (int)((g.chunk + g.tilesPerImg - 2) / g.tilesPerImg) + 1;
    return g;
}

struct PixArgs {
    Geo geo;
    const void* n1; const float* rho1; const void* h1; const void* dout;
    const void* n1d; const void* h1d;       // tangent rows (apply)
    const float* sc; const float* sh; const float* scd; const float* shd;
    Dev p;
    const float* q; const float* qd; const float* kG;
    void* o; void* od; void* dc1;
    float* ws;
};

// Wave-private streaming I/O of NIN input and NOUT output tensors ([R, s*s, 16] maps of T), one 32-pixel span per step.
// NBUF input buffers: a step's spans are requested NBUF - 1 steps ahead (the memory-bound kernels take 3: one step of a
// wave is only 2-4 KB, and with five workgroups per CU a single step in flight per wave covers ~3 us of latency at
// 3 TB/s, not at 6).
template <typename T, int NIN, int NOUT, int NBUF = 2> struct PixIO {
    static constexpr int TB = PIX_WAVE * 16 * (int)sizeof(T);   // bytes of one tensor's span: 1 KB (bf16) / 2 KB (fp32)
    static constexpr int NI = TB / 1024;                         // DMA / store instructions per tensor and step
    // VMEM instructions younger than the request of step t + 1 when step t ends (vmcnt note): that request was issued
    // NBUF - 1 steps ago; since then the stores of NBUF - 1 steps and the requests of NBUF - 2 later steps
    static constexpr int S_VMEM = (NBUF - 1) * NOUT * NI + (NBUF - 2) * NIN * NI;
    static constexpr int WAVE_BYTES = (NBUF * NIN + NOUT) * TB;
    static constexpr int DEPTH = NBUF - 1;
    typedef typename Frag<T>::type frag_t;
    unsigned char* in;       // [NBUF][NIN][TB]
    unsigned char* out;      // [NOUT][TB]
    uint32_t in_addr;        // LDS byte address of `in` (M0 of the DMA)
    int64_t npix, img_bytes;
    uint32_t lane16;         // this lane's byte offset inside a 1 KB DMA piece

    __device__ inline void init(unsigned char* wave_base, int64_t npix_, int lane) {
        in = wave_base; out = wave_base + NBUF * NIN * TB;
        in_addr = (uint32_t)(uintptr_t)(lvoid_t*)wave_base;
        npix = npix_; img_bytes = npix_ * 16 * (int64_t)sizeof(T);
        lane16 = (uint32_t)lane * 16u;
    }
    // request the spans [px0, px0 + 32) of image r of every input tensor into buffer b
    __device__ inline void request(int b, const void* const (&base)[NIN], int64_t r, int64_t px0) const {
        const bool full = px0 + PIX_WAVE <= npix;        // wave-uniform
        const int64_t span = r * img_bytes + px0 * 16 * (int64_t)sizeof(T);
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            const char* sb = reinterpret_cast<const char*>(base[k]) + span;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const uint32_t lds_addr = in_addr + (uint32_t)(((b * NIN + k) * NI + i) * 1024);
                const uint32_t off = lane16 + (uint32_t)(i * 1024);
                if (full) {
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                                 :: "v"(off), "s"(sb), "s"(lds_addr) : "memory", "m0");
                } else {
                    const bool in_img = px0 + (int64_t)(off / (16 * sizeof(T))) < npix;
                    const void* gp = in_img ? static_cast<const void*>(sb + off) : static_cast<const void*>(&g_zero16);
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                                 :: "v"(gp), "s"(lds_addr) : "memory", "m0");
                }
            }
        }
    }
    // fragment (channels 4q .. 4q+3 of pixel 16 sub + m) of input tensor k in buffer b
    __device__ inline frag_t frag(int b, int k, int sub, int q, int m) const {
        return *reinterpret_cast<const frag_t*>(in + (b * NIN + k) * TB + ((sub * 16 + m) * 16 + 4 * q) * (int)sizeof(T));
    }
    __device__ inline void put(int k, int sub, int q, int m, const float v[4]) const {
        frag_t f;
        make_frag(f, v[0], v[1], v[2], v[3]);
        *reinterpret_cast<frag_t*>(out + k * TB + ((sub * 16 + m) * 16 + 4 * q) * (int)sizeof(T)) = f;
    }
    // store output tensor k's span (pixels past the end of the image are steered out of bounds)
    __device__ inline void store(int k, __amdgpu_buffer_rsrc_t rs, int64_t px0) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const uint32_t off = lane16 + (uint32_t)(i * 1024);
            const u32x4 v = *reinterpret_cast<const u32x4*>(out + k * TB + off);
            const bool in_img = px0 + (int64_t)(off / (16 * sizeof(T))) < npix;
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, in_img ? (uint32_t)(px0 * 16 * sizeof(T)) + off : BUF_OOB, 0, 0);
        }
    }
};

struct PixPos { int64_t r, j; };
__device__ inline PixPos pix_next(PixPos p, int64_t spi) {
    if (++p.j == spi) { p.j = 0; ++p.r; }
    return p;
}

// expansion + GELU of one wave row from its n1 fragment (the tail of chain_row)
template <typename T, bool WG>
__device__ inline void expand_gelu(const FwdW<T>& w, const typename Frag<T>::type& n1f, f32x4 g[2], f32x4 gp[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        f32x4 e = ld_f32x4(w.be[j]);
        mma16(e, w.we[j], n1f);
        if constexpr (WG) gelu_both4(e, g[j], gp[j]);
        else g[j] = gelu4(e);
    }
}

// MODE 1 of cnx_fwd_kernel from n1: o = (Wp^T (gelu(We^T n1 + be) (gamma + q) + beta) + bp) * ls + (1 + scale) h1 + shift,
// and (JVP) its tangent from the kept tangent of n1 -- the same expressions, in the same order, as the tile kernel
// evaluates after its LayerNorm: results are bit-identical.
template <typename T, bool JVP>
__global__ void __launch_bounds__(NT, sizeof(T) == 2 ? 4 : 1)
cnx_apply_n1_kernel(PixArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename Frag<T>::type frag_t;
    typedef PixIO<T, JVP ? 4 : 2, JVP ? 2 : 1, JVP ? 2 : 3> IO;     // the primal-only kernel is memory-bound: two steps ahead
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, m = lane & 15;
    FwdW<T> w;
    w.load(a.p, q, m);
    const int s = a.geo.s;
    const int64_t npix = (int64_t)s * s, img = npix * 16, spi = a.geo.tilesPerImg;
    IO io;
    io.init(smem + (size_t)wave * IO::WAVE_BYTES, npix, lane);
    const void* ins[JVP ? 4 : 2];
    ins[0] = a.n1; ins[1] = a.h1;
    if constexpr (JVP) { ins[2] = a.n1d; ins[3] = a.h1d; }
    const int64_t t0 = (int64_t)blockIdx.x * a.geo.chunk;
    const int64_t t1 = t0 + a.geo.chunk < a.geo.total ? t0 + a.geo.chunk : a.geo.total;
    if (t0 >= t1) return;
    int64_t r = t0 / spi;
    int64_t j = t0 - r * spi;
    int64_t rcur = -1;
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 gq[2] = {z4, z4}, qdv[2] = {z4, z4}, sc14 = z4, sh4 = z4, scd4 = z4, shd4 = z4;
    __amdgpu_buffer_rsrc_t rs_o = make_rsrc(nullptr, 0), rs_od = make_rsrc(nullptr, 0);
    // prologue: the first DEPTH steps' spans
    PixPos ahead = {r, j};          // the next step to request
    {
        int64_t tt = t0;
#pragma unroll
        for (int d = 0; d < IO::DEPTH; ++d) {
            if (tt < t1) { io.request(d, ins, ahead.r, ahead.j * PIX_STEP + wave * PIX_WAVE); ahead = pix_next(ahead, spi); ++tt; }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int cur = 0, tgt = IO::DEPTH % (IO::DEPTH + 1);       // buffer of this step / of the step requested in it
    for (int64_t t = t0; t < t1; ++t) {
        const int64_t rt = r, jt = j;
        if (++j == spi) { j = 0; ++r; }
        if (rt != rcur) {
            rcur = rt;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    gq[jj][i] = a.q[rt * 32 + 16 * jj + 4 * q + i];
                    if constexpr (JVP) qdv[jj][i] = a.qd[rt * 32 + 16 * jj + 4 * q + i];
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sc14[i] = 1.0f + a.sc[rt * 16 + 4 * q + i];
                sh4[i] = a.sh[rt * 16 + 4 * q + i];
                if constexpr (JVP) { scd4[i] = a.scd[rt * 16 + 4 * q + i]; shd4[i] = a.shd[rt * 16 + 4 * q + i]; }
            }
            land(gq[0]); land(gq[1]); land(sc14); land(sh4);
            if constexpr (JVP) { land(qdv[0]); land(qdv[1]); land(scd4); land(shd4); }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) gq[jj] = gq[jj] + ld_f32x4(w.gam[jj]);
            rs_o = make_rsrc((const T*)a.o + rt * img, (uint32_t)(img * sizeof(T)));
            if constexpr (JVP) rs_od = make_rsrc((const T*)a.od + rt * img, (uint32_t)(img * sizeof(T)));
        }
        __builtin_amdgcn_sched_barrier(0);
        // the spans DEPTH steps ahead, into the buffer the previous step used (past the end: this step again -- the
        // instruction count stays fixed)
        if (t + IO::DEPTH < t1) { io.request(tgt, ins, ahead.r, ahead.j * PIX_STEP + wave * PIX_WAVE); ahead = pix_next(ahead, spi); }
        else io.request(tgt, ins, rt, jt * PIX_STEP + wave * PIX_WAVE);
        __builtin_amdgcn_sched_barrier(0);
        const int64_t px0 = jt * PIX_STEP + wave * PIX_WAVE;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const frag_t n1f = io.frag(cur, 0, sub, q, m), h1f = io.frag(cur, 1, sub, q, m);
            f32x4 g[2], gp[2], gd[2];
            expand_gelu<T, JVP>(w, n1f, g, gp);
            if constexpr (JVP) {
                const frag_t n1df = io.frag(cur, 2, sub, q, m);
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    f32x4 ed = z4;
                    mma16(ed, w.we[jj], n1df);
                    gd[jj] = ed * gp[jj];
                }
            }
            f32x4 p1 = ld_f32x4(w.bp), p1d = z4;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                frag_t yf;
                const f32x4 yv = fma4(g[jj], gq[jj], ld_f32x4(w.bet[jj]));
                make_frag(yf, yv[0], yv[1], yv[2], yv[3]);
                mma16(p1, w.wp[jj], yf);
                if constexpr (JVP) {
                    frag_t ydf;
                    const f32x4 ydv = fma4(gd[jj], gq[jj], g[jj] * qdv[jj]);
                    make_frag(ydf, ydv[0], ydv[1], ydv[2], ydv[3]);
                    mma16(p1d, w.wp[jj], ydf);
                }
            }
            float hv[4], ov[4];
            unfrag(h1f, hv);
            const f32x4 nv = ld_f32x4(hv), ls4 = ld_f32x4(w.ls);
            const f32x4 o4 = fma4(p1, ls4, fma4(sc14, nv, sh4));
#pragma unroll
            for (int i = 0; i < 4; ++i) ov[i] = o4[i];
            io.put(0, sub, q, m, ov);
            if constexpr (JVP) {
                float hdv[4];
                unfrag(io.frag(cur, 3, sub, q, m), hdv);
                const f32x4 od4 = fma4(p1d, ls4, fma4(sc14, ld_f32x4(hdv), fma4(scd4, nv, shd4)));
#pragma unroll
                for (int i = 0; i < 4; ++i) ov[i] = od4[i];
                io.put(1, sub, q, m, ov);
            }
        }
        io.store(0, rs_o, px0);
        if constexpr (JVP) io.store(1, rs_od, px0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IO::S_VMEM) : "memory");   // this wave's NEXT step's spans have landed
        tgt = cur;
        cur = cur + 1 == IO::DEPTH + 1 ? 0 : cur + 1;
    }
}
template <typename T> inline size_t lds_apply_n1_bytes(bool jvp) {
    return (size_t)NWAVES * (jvp ? PixIO<T, 4, 2, 2>::WAVE_BYTES : PixIO<T, 2, 1, 3>::WAVE_BYTES);
}

// MODE 0 / MODE 1 of cnx_bwd_kernel from n1 (and, MODE 1, its 1/sigma rho1).
// Per-pixel arithmetic is kept in accumulator quads (packed f32 VALU) and the layer scale is folded out of the loop:
//   dy = Wp (ls * dout)      ->  A operand Wp * diag(ls), B operand the dout fragment as loaded (no per-pixel product);
//   d con_w[e][c] = ls[c] * sum_p y[p][e] dout[p][c]   ->  the pixel contraction takes the raw dout tile, ls at the flush;
//   d ls[c] = sum_p dout[p][c] p1[p][c], p1 = bp + Wp^T y   ->  bp[c] * sum_p dout[p][c] + sum_e Wp[e][c] M[e][c] with the
//   same M = sum_p y[p][e] dout[p][c] -- the per-pixel contraction p1 disappears; sum_p dout is one more MFMA on the
//   transposed dout tile (a row of ones as the other operand).
template <typename T, int MODE>
__global__ void __launch_bounds__(NT, sizeof(T) == 2 ? (MODE == 0 ? 4 : 3) : 1)
cnx_bwd_n1_kernel(PixArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename Frag<T>::type frag_t;
    typedef PixIO<T, 2, MODE == 1 ? 1 : 0, MODE == 1 ? 2 : 3> IO;       // in: n1, dout; out: dc1 (MODE 1); MODE 0 is memory-bound
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, m = lane & 15;
    // LDS: the waves' I/O buffers, the cross-wave flush scratch, then (MODE 1) per wave a [6][16][CS] transpose scratch
    constexpr size_t IO_BYTES = (size_t)NWAVES * IO::WAVE_BYTES;
    float* red = reinterpret_cast<float*>(smem + IO_BYTES);
    T* wsT = reinterpret_cast<T*>(smem + IO_BYTES + (size_t)REC_MAIN * sizeof(float)) + (size_t)wave * WS_TILES * 16 * CS;
    FwdW<T> w;
    w.load(a.p, q, m);
    const T* pw = (const T*)a.p.con_w;  // [32][16]: dy[e] = sum_c Wp[e][c] ls[c] dout[c]
    frag_t wpT[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        float wv[4];
        unfrag(load_bfrag<T>(pw, 1, 16, 0, 16 * jj, q, m), wv);       // lane (q, m): Wp[e = 16 jj + m][c = 4q + i]
        make_frag(wpT[jj], wv[0] * w.ls[0], wv[1] * w.ls[1], wv[2] * w.ls[2], wv[3] * w.ls[3]);
    }
    const T* ew = (const T*)a.p.exp_w;  // [16][32]: dn1[c] = sum_e We[c][e] de[e]
    frag_t weT[2] = {load_bfrag<T>(ew, 1, 32, 0, 0, q, m), load_bfrag<T>(ew, 1, 32, 16, 0, q, m)};
    land(wpT[0]); land(wpT[1]); land(weT[0]); land(weT[1]);
    frag_t ones;
    make_frag(ones, 1.0f, 1.0f, 1.0f, 1.0f);
    const int s = a.geo.s;
    const int64_t npix = (int64_t)s * s, img = npix * 16, spi = a.geo.tilesPerImg;
    IO io;
    io.init(smem + (size_t)wave * IO::WAVE_BYTES, npix, lane);
    const void* ins[2] = {a.n1, a.dout};
    const int64_t t0 = (int64_t)blockIdx.x * a.geo.chunk;
    const int64_t t1 = t0 + a.geo.chunk < a.geo.total ? t0 + a.geo.chunk : a.geo.total;
    int64_t r = t0 < t1 ? t0 / spi : 0;
    int64_t j = t0 < t1 ? t0 - r * spi : 0;
    int64_t rcur = -1;
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 gq[2] = {z4, z4}, kg[2] = {z4, z4};
    f32x4 dqp[2] = {z4, z4};
    f32x4 aWp[2] = {z4, z4}, aWe[2] = {z4, z4}, aS = z4;   // aS[e][c] = sum_p dout[p][c] on every row e
    f32x4 dbe[2] = {z4, z4};
    const f32x4 bet4[2] = {ld_f32x4(w.bet[0]), ld_f32x4(w.bet[1])};
    __amdgpu_buffer_rsrc_t rs_dc = make_rsrc(nullptr, 0);
    int krow = 0;
    auto flush_row = [&]() {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = red_m(dqp[jj][i]);
                    if (m == 0) red[wave * REC_DQ + 16 * jj + 4 * q + i] = v;
                    dqp[jj][i] = 0.f;
                }
            __syncthreads();
            if (threadIdx.x < REC_DQ) {       // fixed order: wave 0 + wave 1 + wave 2 + wave 3
                const int c = threadIdx.x;
                const float v = ((red[c] + red[REC_DQ + c]) + red[2 * REC_DQ + c]) + red[3 * REC_DQ + c];
                a.ws[((int64_t)blockIdx.x * a.geo.kmax + krow) * REC_DQ + c] = v;
            }
            ++krow;
            __syncthreads();
        }
    };
    // 1/sigma of this lane's pixel in each of the two column groups (MODE 1): plain loads, one step ahead
    auto fetch_rho = [&](int64_t rr, int64_t jj, float rho[2]) {
        if constexpr (MODE == 1) {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.rho1 + rr * npix, (uint32_t)(npix * sizeof(float)));
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int64_t px = jj * PIX_STEP + wave * PIX_WAVE + sub * 16 + m;
                rho[sub] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, px < npix ? (uint32_t)(px * 4) : BUF_OOB, 0, 0));
            }
        }
    };
    auto wgrad_from = [&](const T* wb) {
        const frag_t tdo = pix_k_frag<T>(wb + 2 * 16 * CS, q, m);
        const frag_t tn1 = pix_k_frag<T>(wb + 3 * 16 * CS, q, m);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const frag_t ty = pix_k_frag<T>(wb + jj * 16 * CS, q, m);
            const frag_t tde = pix_k_frag<T>(wb + (4 + jj) * 16 * CS, q, m);
            mma16(aWp[jj], ty, tdo);    // M[e][c] += y^T dout
            mma16(aWe[jj], tn1, tde);   // [c][e] += n1^T de
        }
        mma16(aS, ones, tdo);           // every row: sum_p dout[p][c]
        lds_fence();
    };
    float rn[2] = {0.f, 0.f};
    PixPos ahead = {r, j};          // the next step to request
    if (t0 < t1) {
        fetch_rho(r, j, rn);
        int64_t tt = t0;
#pragma unroll
        for (int d = 0; d < IO::DEPTH; ++d) {
            if (tt < t1) { io.request(d, ins, ahead.r, ahead.j * PIX_STEP + wave * PIX_WAVE); ahead = pix_next(ahead, spi); ++tt; }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int cur = 0, tgt = IO::DEPTH % (IO::DEPTH + 1);
    for (int64_t t = t0; t < t1; ++t) {
        const float rho1[2] = {rn[0], rn[1]};
        if constexpr (MODE == 1) { land(rho1[0]); land(rho1[1]); }
        const int64_t rt = r, jt = j;
        if (++j == spi) { j = 0; ++r; }
        if (rt != rcur) {
            if (rcur >= 0) flush_row();
            rcur = rt;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    gq[jj][i] = a.q[rt * 32 + 16 * jj + 4 * q + i] + w.gam[jj][i];
                    if constexpr (MODE == 1) kg[jj][i] = a.kG[rt * 32 + 16 * jj + 4 * q + i];
                }
            land(gq[0]); land(gq[1]);
            if constexpr (MODE == 1) { land(kg[0]); land(kg[1]); }
            if constexpr (MODE == 1) rs_dc = make_rsrc((const T*)a.dc1 + rt * img, (uint32_t)(img * sizeof(T)));
        }
        __builtin_amdgcn_sched_barrier(0);
        // next step: the 1/sigma loads first (older than the DMA: the counted wait below then covers them too), then the
        // spans (past the end: this step again into the other buffer -- the instruction count stays fixed)
        if (t + 1 < t1) fetch_rho(r, j, rn); else fetch_rho(rt, jt, rn);
        if (t + IO::DEPTH < t1) { io.request(tgt, ins, ahead.r, ahead.j * PIX_STEP + wave * PIX_WAVE); ahead = pix_next(ahead, spi); }
        else io.request(tgt, ins, rt, jt * PIX_STEP + wave * PIX_WAVE);
        __builtin_amdgcn_sched_barrier(0);
        const int64_t px0 = jt * PIX_STEP + wave * PIX_WAVE;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const frag_t n1f = io.frag(cur, 0, sub, q, m), dof = io.frag(cur, 1, sub, q, m);    // dout: zero outside the image
            f32x4 g[2], gp[2];
            expand_gelu<T, MODE == 1>(w, n1f, g, gp);
            f32x4 dy[2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                dy[jj] = z4;
                mma16(dy[jj], wpT[jj], dof);       // dy^T = (Wp diag(ls)) dout^T
            }
            if constexpr (MODE == 0) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) dqp[jj] = fma4(dy[jj], g[jj], dqp[jj]);
            } else {
                frag_t yf[2], def[2];
                f32x4 de[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const f32x4 yv = fma4(g[jj], gq[jj], bet4[jj]);
                    make_frag(yf[jj], yv[0], yv[1], yv[2], yv[3]);
                    de[jj] = fma4(dy[jj], gq[jj], g[jj] * kg[jj]) * gp[jj];
                    dbe[jj] = dbe[jj] + de[jj];
                    make_frag(def[jj], de[jj][0], de[jj][1], de[jj][2], de[jj][3]);
                }
                // The pixels past the end of the image (last step of an image only: n1 = dout = 0 there, but g(be) != 0)
                // must not reach d exp_b -- everything else they touch is multiplied by a zero or never stored.  Taken
                // back here, in a branch the other steps of the image skip.
                if (jt == spi - 1) [[unlikely]] {
                    if (px0 + sub * 16 + m >= npix) { dbe[0] = dbe[0] - de[0]; dbe[1] = dbe[1] - de[1]; }
                }
                f32x4 dn1 = z4;
                mma_pair(dn1, weT[0], weT[1], def[0], def[1]);      // dn1^T = We de^T
                // LayerNorm backward: dc = rho (dn - mean(dn) - n mean(dn n))
                float n1[4];
                unfrag(n1f, n1);
                const f32x4 n4 = ld_f32x4(n1);
                float m1 = (dn1[0] + dn1[1]) + (dn1[2] + dn1[3]);
                float m2 = dn1[0] * n1[0] + dn1[1] * n1[1] + dn1[2] * n1[2] + dn1[3] * n1[3];
                red_q2(m1, m2);
                const f32x4 dc4 = (fma4(n4, splat4(m2 * (-1.0f / 16.0f)), dn1) - splat4(m1 * (1.0f / 16.0f))) * splat4(rho1[sub]);
                float dc[4] = {dc4[0], dc4[1], dc4[2], dc4[3]};
                io.put(0, sub, q, m, dc);
                // weight gradients contract over the 16 pixels of the column group: one batched transpose (pixel-on-lane
                // -> pixel-as-k) through the wave's LDS scratch (a wave's LDS operations execute in order)
                *reinterpret_cast<frag_t*>(wsT + (0 * 16 + m) * CS + 4 * q) = yf[0];
                *reinterpret_cast<frag_t*>(wsT + (1 * 16 + m) * CS + 4 * q) = yf[1];
                *reinterpret_cast<frag_t*>(wsT + (2 * 16 + m) * CS + 4 * q) = dof;
                *reinterpret_cast<frag_t*>(wsT + (3 * 16 + m) * CS + 4 * q) = n1f;     // (pixels outside the image meet de = 0)
                *reinterpret_cast<frag_t*>(wsT + (4 * 16 + m) * CS + 4 * q) = def[0];
                *reinterpret_cast<frag_t*>(wsT + (5 * 16 + m) * CS + 4 * q) = def[1];
                lds_fence();
                wgrad_from(wsT);
            }
        }
        if constexpr (MODE == 1) io.store(0, rs_dc, px0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IO::S_VMEM) : "memory");   // this wave's NEXT step's spans (and 1/sigma) have landed
        tgt = cur;
        cur = cur + 1 == IO::DEPTH + 1 ? 0 : cur + 1;
    }
    if (rcur >= 0) flush_row();
    if constexpr (MODE == 0) {
        // records this workgroup's range did not reach: zeros (the row reduction only reads the ones it did reach, but
        // a fresh workspace must never leak into a sum if the geometry changes)
        for (int k = krow; k < a.geo.kmax; ++k)
            if (threadIdx.x < REC_DQ) a.ws[((int64_t)blockIdx.x * a.geo.kmax + k) * REC_DQ + threadIdx.x] = 0.f;
    }
    if constexpr (MODE == 1) {
        float* scratch = red;   // con_w [32][16] | exp_w [16][32] | ls [16] | exp_b [32]
        // this lane's share of d ls[c = m]: bp[c] sum_p dout[p][c] (once: the lanes q = 0) + sum over its 8 rows e of
        // Wp[e][c] M[e][c]; then the four q-lanes of a column are added
        const float lsm = a.p.ls[m];
        float dlsv = q == 0 ? a.p.con_b[m] * aS[0] : 0.f;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) dlsv += St<T>::ld(pw + (16 * jj + 4 * q + e) * 16 + m) * aWp[jj][e];
        dlsv = red_q(dlsv);
        float vbe[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) vbe[jj][i] = red_m(dbe[jj][i]);
        __syncthreads();
        for (int i = threadIdx.x; i < REC_MAIN; i += NT) scratch[i] = 0.f;
        __syncthreads();
        for (int wv = 0; wv < NWAVES; ++wv) {
            if (wave == wv) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        scratch[(16 * jj + 4 * q + e) * 16 + m] += aWp[jj][e] * lsm;
                        scratch[512 + (4 * q + e) * 32 + 16 * jj + m] += aWe[jj][e];
                    }
                if (q == 0) scratch[1024 + m] += dlsv;
                if (m == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) scratch[1040 + 16 * jj + 4 * q + i] += vbe[jj][i];
                }
            }
            __syncthreads();
        }
        float* rec = a.ws + (int64_t)blockIdx.x * REC_MAIN;
        for (int i = threadIdx.x; i < REC_MAIN; i += NT) rec[i] = ABL_NO_FLUSH ? 0.f : scratch[i];
    }
}
template <typename T> inline size_t lds_bwd_n1_bytes(int mode) {
    const size_t io = (size_t)NWAVES * (mode == 1 ? PixIO<T, 2, 1, 2>::WAVE_BYTES : PixIO<T, 2, 0, 3>::WAVE_BYTES);
    const size_t scratch = (size_t)REC_MAIN * sizeof(float);      // (MODE 0 uses its first NWAVES * REC_DQ floats)
    return io + scratch + (mode == 1 ? (size_t)NWAVES * WS_TILES * 16 * CS * sizeof(T) : 0);
}

// ---------------------------------------------------------------------------
// fixed-order reductions of the workspace records
// ---------------------------------------------------------------------------
// Per-row quantities: out0[r][c] (c < n0) / out1[r][c - n0] (c >= n0) = sum over the workgroups b whose tile range
// touches row r, in ascending b, of record (b, r - first_row(b)) at [off + c].
__global__ void __launch_bounds__(256)
cnx_reduce_rows_kernel(const float* ws, Geo g, int rec, int off, int n, int n0, float* out0, float* out1) {
    const int64_t idx = blockIdx.x * 256LL + threadIdx.x;
    if (idx >= g.R * n) return;
    const int64_t r = idx / n;
    const int c = (int)(idx - r * n);
    const int64_t b0 = (r * g.tilesPerImg) / g.chunk, b1 = ((r + 1) * g.tilesPerImg - 1) / g.chunk;
    // four independent accumulators (a fixed shape: records b0 + 4i + j go to accumulator j): one chain of ~50 dependent
    // L2 round trips per thread was 20 us per launch, 40 launches per step
    auto rec_of = [&](int64_t b) { return ws[(b * g.kmax + (r - (b * g.chunk) / g.tilesPerImg)) * rec + off + c]; };
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    int64_t b = b0;
    for (; b + 3 <= b1; b += 4) { v0 += rec_of(b); v1 += rec_of(b + 1); v2 += rec_of(b + 2); v3 += rec_of(b + 3); }
    for (; b <= b1; ++b) v0 += rec_of(b);
    const float v = (v0 + v1) + (v2 + v3);
    if (c < n0) out0[r * n0 + c] = v;
    else out1[r * (n - n0) + c - n0] = v;
}

// Parameter gradients: dst[i] += sum over all `nrec` records of rec[off + i], in a tree of fixed shape: a workgroup owns
// 16 consecutive elements; its 64 slots (16 waves x 4 quarter-waves) each sum the records congruent to their index mod
// 64 with four independent accumulators (the loads of one slot are serial otherwise: the first version of this kernel
// spent 72 us per launch on 64 dependent L2 round trips), and the 64 partial sums are added in slot order.
struct RedSeg { float* dst; int begin, end; };     // record elements [begin, end) -> dst[0 .. end - begin)
struct RedSegs { RedSeg s[4]; int n; };
constexpr int RB_ELEMS = 16, RB_SLOTS = 64;
__global__ void __launch_bounds__(1024)
cnx_reduce_blocks_kernel(const float* ws, int64_t nrec, int rec, int total, RedSegs segs) {
    __shared__ float part[RB_SLOTS][RB_ELEMS + 1];
    const int il = threadIdx.x & 15, slot = threadIdx.x >> 4;
    const int i = blockIdx.x * RB_ELEMS + il;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (i < total) {
        const float* p = ws + i;
        int64_t b = slot;
        for (; b + 3 * RB_SLOTS < nrec; b += 4 * RB_SLOTS) {
            v0 += p[b * rec];
            v1 += p[(b + RB_SLOTS) * rec];
            v2 += p[(b + 2 * RB_SLOTS) * rec];
            v3 += p[(b + 3 * RB_SLOTS) * rec];
        }
        for (; b < nrec; b += RB_SLOTS) v0 += p[b * rec];
    }
    part[slot][il] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    if (slot == 0 && i < total) {
        float sum = part[0][il];
#pragma unroll 8
        for (int k = 1; k < RB_SLOTS; ++k) sum += part[k][il];
        for (int k = 0; k < segs.n; ++k)
            if (i >= segs.s[k].begin && i < segs.s[k].end) segs.s[k].dst[i - segs.s[k].begin] += sum;
    }
}

// standalone first LayerNorm (what mfc_gemm's MFC_GEMM_LN16 epilogue fuses): y = LN_16(x), rstd per pixel
template <typename T>
__global__ void __launch_bounds__(256) ln16_kernel(int64_t npix, const T* x, T* y, float* rstd) {
    for (int64_t p = blockIdx.x * 256LL + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
        float v[16];
        ld16<T>(x + p * 16, v);
        float sum = 0.f, sq = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) { sum += v[c]; sq += v[c] * v[c]; }
        const float mean = sum * (1.0f / 16.0f);
        const float rho = rsqrtf(fmaxf(0.0f, sq * (1.0f / 16.0f) - mean * mean) + LN_EPS);
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = (v[c] - mean) * rho;
#pragma unroll
        for (int i = 0; i < 4; ++i) st4(y + p * 16 + 4 * i, v + 4 * i);
        if (rstd) rstd[p] = rho;
    }
}

// tangent of that LayerNorm for a raw tangent map xd: nd = rho (xd_c - n mean(n xd_c)), xd_c = xd - mean(xd)
template <typename T>
__global__ void __launch_bounds__(256) ln16_jvp_kernel(int64_t npix, const T* n, const float* rstd, const T* xd, T* nd) {
    for (int64_t p = blockIdx.x * 256LL + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
        float v[16], d[16];
        ld16<T>(n + p * 16, v);
        ld16<T>(xd + p * 16, d);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) sum += d[c];
        const float md = sum * (1.0f / 16.0f);
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) { d[c] -= md; dot += v[c] * d[c]; }
        dot *= (1.0f / 16.0f);
        const float rho = rstd[p];
#pragma unroll
        for (int c = 0; c < 16; ++c) d[c] = rho * (d[c] - v[c] * dot);
#pragma unroll
        for (int i = 0; i < 4; ++i) st4(nd + p * 16 + 4 * i, d + 4 * i);
    }
}

__global__ void grn_finalize_kernel(int64_t R, const float* S1, const float* S2, float* G, float* qo, float* qd) {
    const int64_t r = blockIdx.x * (int64_t)blockDim.x / 32 + threadIdx.x / 32;
    const int c = threadIdx.x & 31;
    if (r >= R) return;
    const float g = sqrtf(S1[r * 32 + c]);
    float n = g;
    for (int o = 16; o > 0; o >>= 1) n += __shfl_xor(n, o, 32);
    n *= (1.0f / 32.0f);
    const float inv = 1.0f / (n + GRN_EPS);
    G[r * 32 + c] = g;
    qo[r * 32 + c] = g * inv;
    if (S2 && qd) {
        const float gd = g > 0.f ? S2[r * 32 + c] / g : 0.f;
        float nd = gd;
        for (int o = 16; o > 0; o >>= 1) nd += __shfl_xor(nd, o, 32);
        nd *= (1.0f / 32.0f);
        qd[r * 32 + c] = gd * inv - g * nd * inv * inv;
    }
}

__global__ void grn_bwd_finalize_kernel(int64_t R, const float* G, const float* dq, float* kG, float* dgamma) {
    const int64_t r = blockIdx.x * (int64_t)blockDim.x / 32 + threadIdx.x / 32;
    const int c = threadIdx.x & 31;
    if (r >= R) return;
    const float g = G[r * 32 + c];
    const float d = dq[r * 32 + c];
    float n = g, sdg = d * g;
    for (int o = 16; o > 0; o >>= 1) { n += __shfl_xor(n, o, 32); sdg += __shfl_xor(sdg, o, 32); }
    n *= (1.0f / 32.0f);
    const float inv = 1.0f / (n + GRN_EPS);
    const float dG = d * inv - sdg * inv * inv * (1.0f / 32.0f);
    kG[r * 32 + c] = g > 0.f ? dG / g : 0.f;
}
// dgamma[c] += sum_r dq[r][c]: fixed-shape tree (8 row groups r = g, g + 8, ... ascending, then groups in order)
__global__ void __launch_bounds__(256) grn_dgamma_kernel(int64_t R, const float* dq, float* dgamma) {
    __shared__ float part[8][32];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    float v = 0.f;
    for (int64_t r = g; r < R; r += 8) v += dq[r * 32 + c];
    part[g][c] = v;
    __syncthreads();
    if (g == 0) {
        float sum = part[0][c];
#pragma unroll
        for (int k = 1; k < 8; ++k) sum += part[k][c];
        dgamma[c] += sum;
    }
}

inline Dev to_dev(const mfc_cnx_params* p) {
    return Dev{p->conv_w, p->conv_b, p->exp_w, p->exp_b, p->grn_gamma, p->grn_beta, p->con_w, p->con_b, p->ls};
}
inline DevG to_devg(const mfc_cnx_grads* g) {
    if (!g) return DevG{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    return DevG{g->conv_w, g->conv_b, g->exp_w, g->exp_b, g->grn_gamma, g->grn_beta, g->con_w, g->con_b, g->ls};
}
inline bool params_ok(const mfc_cnx_params* p) {
    return p && p->conv_w && p->conv_b && p->exp_w && p->exp_b && p->grn_gamma && p->grn_beta && p->con_w &&
           p->con_b && p->ls;
}

// Persistent grid per kernel: a multiple of what is resident at once (256 CUs x 4 / 3 / 2 workgroups per CU by
// registers or LDS), so no partly filled last round; measured at the literal spatial size (tools/bench_cnx.py sweep,
// 2048 for every kernel before): 4-per-CU kernels 3 rounds (-7..-11 %), the 3-per-CU tangent statistics 3 rounds
// (-8 %), the 2-per-CU kernels exactly one round (-3..-7 %); the conv-gradient kernel, three per CU since its dout
// tile became compact and single-buffered, one round of 768 (-7 % against two per CU).  MFC_CNX_MAX_BLOCKS / mfc_cnx_max_blocks override all.
enum CnxKind { K_STATS = 0, K_APPLY, K_STATS_JVP, K_APPLY_JVP, K_BWD_STATS, K_BWD_MAIN, K_BWD_CONV, K_APPLY_N1, K_BWD_STATS_N1,
               K_BWD_MAIN_N1, K_APPLY_JVP_N1, K_NKIND };
static const int64_t DEFAULT_BLOCKS[K_NKIND] = {3072, 3072, 2304, 512, 3072, 512, 768, 3072, 3072, 2304, 3072};
inline bool kind_is_pix(int k) { return k >= K_APPLY_N1; }
static int64_t MAX_BLOCKS = getenv("MFC_CNX_MAX_BLOCKS") ? atoll(getenv("MFC_CNX_MAX_BLOCKS")) : 0;   // 0: per-kernel defaults
// MFC_CNX_BLOCKS="kind:n,kind:n": per-kernel overrides of the persistent grid (tuning sweeps; kind = CnxKind index)
static int64_t KIND_BLOCKS[K_NKIND];
static const bool KIND_BLOCKS_INIT = [] {
    for (int k = 0; k < K_NKIND; ++k) KIND_BLOCKS[k] = 0;
    const char* e = getenv("MFC_CNX_BLOCKS");
    while (e && *e) {
        char* end = nullptr;
        const long k = strtol(e, &end, 10);
        if (end == e || *end != ':') break;
        const long n = strtol(end + 1, &end, 10);
        if (k >= 0 && k < K_NKIND && n > 0) KIND_BLOCKS[k] = n;
        e = (*end == ',') ? end + 1 : end;
        if (*end != ',') break;
    }
    return true;
}();
inline int64_t max_blocks(CnxKind k) { return MAX_BLOCKS > 0 ? MAX_BLOCKS : (KIND_BLOCKS[k] > 0 ? KIND_BLOCKS[k] : DEFAULT_BLOCKS[k]); }
constexpr int MAX_S = 8000;   // one [s, s, 16] fp32 image must stay below the 4 GiB a buffer resource addresses

template <typename K, typename A>
inline int launch_k(K kern, int64_t grid, size_t lds, hipStream_t st, const A& args) {
#if defined(MFC_CNX_LDS_PAD)
    lds += MFC_CNX_LDS_PAD;   // occupancy probe (compile-time): unused extra LDS per workgroup, so fewer workgroups fit a CU
#endif
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, st, args);
    return mfc_launch_status();
}

inline int reduce_rows(const float* ws, const Geo& g, int rec, int off, int n, int n0, float* out0, float* out1,
                       hipStream_t st) {
    const int64_t blocks = ceil_div64(g.R * n, 256);
    hipLaunchKernelGGL(cnx_reduce_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ws, g, rec, off, n, n0, out0, out1);
    return mfc_launch_status();
}
inline int reduce_blocks(const float* ws, int64_t nrec, int rec, int total, const RedSegs& segs, hipStream_t st) {
    hipLaunchKernelGGL(cnx_reduce_blocks_kernel, dim3((unsigned)((total + RB_ELEMS - 1) / RB_ELEMS)), dim3(1024), 0, st, ws, nrec, rec,
                       total, segs);
    return mfc_launch_status();
}

template <typename T>
int fwd_launch(bool jvp, int mode, const FwdArgs& a, int64_t grid, hipStream_t st) {
    const size_t lds = lds2_bytes<T>(jvp ? 2 : 1, false);
    if (jvp) {
        if (mode == 0) return launch_k(cnx_fwd_kernel<T, true, 0>, grid, lds, st, a);
        return launch_k(cnx_fwd_kernel<T, true, 1>, grid, lds, st, a);
    }
    if (mode == 0) return launch_k(cnx_fwd_kernel<T, false, 0>, grid, lds, st, a);
    return launch_k(cnx_fwd_kernel<T, false, 1>, grid, lds, st, a);
}

int fwd_common(int dtype, int mode, int64_t R, int s, const void* h0, const void* h0dot,
               const float* scale, const float* shift, const float* scaledot, const float* shiftdot,
               const mfc_cnx_params* p, float* S1, float* S2, const float* q, const float* qdot,
               void* o, void* odot, float* ws, void* stream, void* n1_out = nullptr, float* rho1_out = nullptr,
               void* n1dot_out = nullptr) {
    if (!h0 || !scale || !shift || !params_ok(p)) return MFC_EFAULT;
    if (R <= 0 || s <= 0) return MFC_EINVAL;
    if (s > MAX_S) return MFC_ENOSYS;
    if (dtype != MFC_F32 && dtype != MFC_BF16) return MFC_EINVAL;
    const bool jvp = h0dot != nullptr;
    if (jvp && (!scaledot || !shiftdot)) return MFC_EFAULT;
    if (mode == 0 && (!S1 || (jvp && !S2) || !ws)) return MFC_EFAULT;
    if (mode == 1 && (!q || !o || (jvp && (!qdot || !odot)))) return MFC_EFAULT;
    FwdArgs a;
    int64_t grid;
    a.geo = make_geo(R, s, max_blocks(mode == 0 ? (jvp ? K_STATS_JVP : K_STATS) : (jvp ? K_APPLY_JVP : K_APPLY)), grid);
    a.h0 = h0; a.h0d = h0dot; a.sc = scale; a.sh = shift; a.scd = scaledot; a.shd = shiftdot;
    a.p = to_dev(p); a.S1 = S1; a.S2 = S2; a.q = q; a.qd = qdot; a.o = o; a.od = odot; a.ws = ws;
    a.n1_out = n1_out; a.rho1_out = rho1_out; a.n1d_out = n1dot_out;
    hipStream_t st = (hipStream_t)stream;
    int rc = dtype == MFC_F32 ? fwd_launch<float>(jvp, mode, a, grid, st) : fwd_launch<u16>(jvp, mode, a, grid, st);
    if (!rc && mode == 0) rc = reduce_rows(ws, a.geo, REC_STATS, 0, jvp ? 64 : 32, 32, S1, S2, st);
    return rc;
}

}  // namespace

extern "C" int mfc_cnx_stats(int dtype, int64_t R, int s, const void* h1, const void* h1dot,
                             const float* scale, const float* shift, const float* scaledot,
                             const float* shiftdot, const mfc_cnx_params* p, float* S1, float* S2,
                             float* ws, void* stream) {
    return fwd_common(dtype, 0, R, s, h1, h1dot, scale, shift, scaledot, shiftdot, p, S1, S2, nullptr, nullptr,
                      nullptr, nullptr, ws, stream);
}

extern "C" int mfc_cnx_stats_save(int dtype, int64_t R, int s, const void* h1, const void* h1dot,
                                  const float* scale, const float* shift, const float* scaledot,
                                  const float* shiftdot, const mfc_cnx_params* p, float* S1, float* S2,
                                  float* ws, void* n1_out, float* rho1_out, void* n1dot_out, void* stream) {
    if (!n1_out || !rho1_out) return MFC_EFAULT;
    if (n1dot_out && !h1dot) return MFC_EINVAL;
    return fwd_common(dtype, 0, R, s, h1, h1dot, scale, shift, scaledot, shiftdot, p, S1, S2, nullptr, nullptr,
                      nullptr, nullptr, ws, stream, n1_out, rho1_out, n1dot_out);
}

namespace {
inline int pix_common(int dtype, int64_t R, int s) {
    if (R <= 0 || s <= 0 || (dtype != MFC_F32 && dtype != MFC_BF16)) return MFC_EINVAL;
    if (s > MAX_S) return MFC_ENOSYS;
    return MFC_OK;
}
}  // namespace

extern "C" int mfc_cnx_apply_n1(int dtype, int64_t R, int s, const void* n1, const void* n1dot, const void* h1,
                                const void* h1dot, const float* scale, const float* shift, const float* scaledot,
                                const float* shiftdot, const mfc_cnx_params* p, const float* q, const float* qdot, void* o,
                                void* odot, void* stream) {
    if (!n1 || !h1 || !scale || !shift || !params_ok(p) || !q || !o) return MFC_EFAULT;
    const bool jvp = n1dot != nullptr;
    if (jvp && (!h1dot || !scaledot || !shiftdot || !qdot || !odot)) return MFC_EFAULT;
    if (int rc = pix_common(dtype, R, s)) return rc;
    PixArgs a = {};
    int64_t grid;
    a.geo = make_pix_geo(R, s, max_blocks(jvp ? K_APPLY_JVP_N1 : K_APPLY_N1), grid);
    a.n1 = n1; a.n1d = n1dot; a.h1 = h1; a.h1d = h1dot; a.sc = scale; a.sh = shift; a.scd = scaledot; a.shd = shiftdot;
    a.p = to_dev(p); a.q = q; a.qd = qdot; a.o = o; a.od = odot;
    hipStream_t st = (hipStream_t)stream;
    if (jvp)
        return dtype == MFC_F32 ? launch_k(cnx_apply_n1_kernel<float, true>, grid, lds_apply_n1_bytes<float>(true), st, a)
                                : launch_k(cnx_apply_n1_kernel<u16, true>, grid, lds_apply_n1_bytes<u16>(true), st, a);
    return dtype == MFC_F32 ? launch_k(cnx_apply_n1_kernel<float, false>, grid, lds_apply_n1_bytes<float>(false), st, a)
                            : launch_k(cnx_apply_n1_kernel<u16, false>, grid, lds_apply_n1_bytes<u16>(false), st, a);
}

extern "C" int mfc_cnx_bwd_stats_n1(int dtype, int64_t R, int s, const void* n1, const mfc_cnx_params* p, const float* q,
                                    const void* dout, float* dq, float* ws, void* stream) {
    if (!n1 || !params_ok(p) || !q || !dout || !dq || !ws) return MFC_EFAULT;
    if (int rc = pix_common(dtype, R, s)) return rc;
    PixArgs a = {};
    int64_t grid;
    a.geo = make_pix_geo(R, s, max_blocks(K_BWD_STATS_N1), grid);
    a.n1 = n1; a.dout = dout; a.p = to_dev(p); a.q = q; a.ws = ws;
    hipStream_t st = (hipStream_t)stream;
    int rc = dtype == MFC_F32 ? launch_k(cnx_bwd_n1_kernel<float, 0>, grid, lds_bwd_n1_bytes<float>(0), st, a)
                              : launch_k(cnx_bwd_n1_kernel<u16, 0>, grid, lds_bwd_n1_bytes<u16>(0), st, a);
    if (!rc) rc = reduce_rows(ws, a.geo, REC_DQ, 0, 32, 32, dq, nullptr, st);
    return rc;
}

extern "C" int mfc_cnx_bwd_main_n1(int dtype, int64_t R, int s, const void* n1, const float* rho1, const mfc_cnx_params* p,
                                   const float* q, const float* kG, const void* dout, void* dc1, const mfc_cnx_grads* g,
                                   float* ws, void* stream) {
    if (!n1 || !rho1 || !params_ok(p) || !q || !kG || !dout || !dc1 || !g || !ws) return MFC_EFAULT;
    if (!g->con_w || !g->ls || !g->exp_w || !g->exp_b) return MFC_EFAULT;
    if (int rc = pix_common(dtype, R, s)) return rc;
    PixArgs a = {};
    int64_t grid;
    a.geo = make_pix_geo(R, s, max_blocks(K_BWD_MAIN_N1), grid);
    a.n1 = n1; a.rho1 = rho1; a.dout = dout; a.p = to_dev(p); a.q = q; a.kG = kG; a.dc1 = dc1; a.ws = ws;
    hipStream_t st = (hipStream_t)stream;
    int rc = dtype == MFC_F32 ? launch_k(cnx_bwd_n1_kernel<float, 1>, grid, lds_bwd_n1_bytes<float>(1), st, a)
                              : launch_k(cnx_bwd_n1_kernel<u16, 1>, grid, lds_bwd_n1_bytes<u16>(1), st, a);
    if (!rc) {
        RedSegs sg = {{{g->con_w, 0, 512}, {g->exp_w, 512, 1024}, {g->ls, 1024, 1040}, {g->exp_b, 1040, 1072}}, 4};
        rc = reduce_blocks(ws, grid, REC_MAIN, REC_MAIN, sg, st);
    }
    return rc;
}

extern "C" int64_t mfc_cnx_ws_elems(int64_t R, int s) {
    if (R <= 0 || s <= 0 || s > MAX_S) return -1;
    int64_t n = 0;
    for (int k = 0; k < K_NKIND; ++k) {
        const int64_t e = ws_elems_for(R, s, max_blocks((CnxKind)k), kind_is_pix(k));
        n = e > n ? e : n;
    }
    return n;
}

extern "C" int mfc_cnx_apply(int dtype, int64_t R, int s, const void* h1, const void* h1dot,
                             const float* scale, const float* shift, const float* scaledot,
                             const float* shiftdot, const mfc_cnx_params* p, const float* q, const float* qdot,
                             void* o, void* odot, void* stream) {
    return fwd_common(dtype, 1, R, s, h1, h1dot, scale, shift, scaledot, shiftdot, p, nullptr, nullptr, q, qdot,
                      o, odot, nullptr, stream);
}

extern "C" int mfc_grn_finalize(int64_t R, const float* S1, const float* S2, float* G, float* q, float* qdot,
                                void* stream) {
    if (!S1 || !G || !q) return MFC_EFAULT;
    if (R <= 0) return MFC_EINVAL;
    const int64_t blocks = ceil_div64(R * 32, 256);
    hipLaunchKernelGGL(grn_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, R, S1, S2,
                       G, q, qdot);
    return mfc_launch_status();
}

extern "C" int mfc_grn_bwd_finalize(int64_t R, const float* G, const float* dq, float* kG, float* dgamma,
                                    void* stream) {
    if (!G || !dq || !kG || !dgamma) return MFC_EFAULT;
    if (R <= 0) return MFC_EINVAL;
    const int64_t blocks = ceil_div64(R * 32, 256);
    hipLaunchKernelGGL(grn_bwd_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, R, G,
                       dq, kG, dgamma);
    hipLaunchKernelGGL(grn_dgamma_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, R, dq, dgamma);
    return mfc_launch_status();
}

extern "C" int mfc_cnx_bwd_stats(int dtype, int64_t R, int s, const void* h0, const float* scale,
                                 const float* shift, const mfc_cnx_params* p, const float* q, const void* dout,
                                 float* dq, float* ws, void* stream) {
    if (!h0 || !scale || !shift || !params_ok(p) || !q || !dout || !dq || !ws) return MFC_EFAULT;
    if (R <= 0 || s <= 0 || (dtype != MFC_F32 && dtype != MFC_BF16)) return MFC_EINVAL;
    if (s > MAX_S) return MFC_ENOSYS;
    BwdArgs a = {};
    int64_t grid;
    a.geo = make_geo(R, s, max_blocks(K_BWD_STATS), grid);
    a.h0 = h0; a.sc = scale; a.sh = shift; a.p = to_dev(p); a.g = to_devg(nullptr);
    a.q = q; a.dout = dout; a.dq = dq; a.ws = ws;
    hipStream_t st = (hipStream_t)stream;
    int rc = dtype == MFC_F32 ? launch_k(cnx_bwd_kernel<float, 0>, grid, lds2_bytes<float>(1, false), st, a)
                              : launch_k(cnx_bwd_kernel<u16, 0>, grid, lds2_bytes<u16>(1, false), st, a);
    if (!rc) rc = reduce_rows(ws, a.geo, REC_DQ, 0, 32, 32, dq, nullptr, st);
    return rc;
}

extern "C" int mfc_cnx_bwd_main(int dtype, int64_t R, int s, const void* h0, const float* scale,
                                const float* shift, const mfc_cnx_params* p, const float* q, const float* kG,
                                const void* dout, void* dc1, const mfc_cnx_grads* g, float* ws, void* stream) {
    if (!h0 || !scale || !shift || !params_ok(p) || !q || !kG || !dout || !dc1 || !g || !ws) return MFC_EFAULT;
    if (!g->con_w || !g->ls || !g->exp_w || !g->exp_b) return MFC_EFAULT;
    if (R <= 0 || s <= 0 || (dtype != MFC_F32 && dtype != MFC_BF16)) return MFC_EINVAL;
    if (s > MAX_S) return MFC_ENOSYS;
    BwdArgs a = {};
    int64_t grid;
    a.geo = make_geo(R, s, max_blocks(K_BWD_MAIN), grid);
    a.h0 = h0; a.sc = scale; a.sh = shift; a.p = to_dev(p); a.g = to_devg(g);
    a.q = q; a.kG = kG; a.dout = dout; a.dc1 = dc1; a.ws = ws;
    hipStream_t st = (hipStream_t)stream;
    int rc = dtype == MFC_F32 ? launch_k(cnx_bwd_kernel<float, 1>, grid, lds2_bytes<float>(2, true), st, a)
                              : launch_k(cnx_bwd_kernel<u16, 1>, grid, lds2_bytes<u16>(2, true), st, a);
    if (!rc) {
        RedSegs sg = {{{g->con_w, 0, 512}, {g->exp_w, 512, 1024}, {g->ls, 1024, 1040}, {g->exp_b, 1040, 1072}}, 4};
        rc = reduce_blocks(ws, grid, REC_MAIN, REC_MAIN, sg, st);
    }
    return rc;
}

extern "C" int mfc_cnx_bwd_conv(int dtype, int64_t R, int s, const void* h0, const float* rho0,
                                const float* scale, const float* shift, const mfc_cnx_params* p, const void* dc1,
                                const void* dout, void* dh0, const mfc_cnx_grads* g, float* dscale, float* dshift,
                                float* ws, void* stream) {
    if (!h0 || !rho0 || !scale || !shift || !params_ok(p) || !dc1 || !dout || !dh0 || !g || !g->conv_w || !g->conv_b ||
        !g->con_b || !g->grn_beta || !dscale || !dshift || !ws)
        return MFC_EFAULT;
    if (R <= 0 || s <= 0 || (dtype != MFC_F32 && dtype != MFC_BF16)) return MFC_EINVAL;
    if (s > MAX_S) return MFC_ENOSYS;
    BwdArgs a = {};
    int64_t grid;
    a.geo = make_geo(R, s, max_blocks(K_BWD_CONV), grid);
    a.h0 = h0; a.sc = scale; a.sh = shift; a.p = to_dev(p); a.g = to_devg(g);
    a.rho = rho0; a.dc1_in = dc1; a.dout = dout; a.dh0 = dh0; a.dsc = dscale; a.dsh = dshift; a.ws = ws;
    hipStream_t st = (hipStream_t)stream;
    int rc = dtype == MFC_F32 ? launch_k(cnx_bwd_conv_kernel<float>, grid, lds_bwd_conv_bytes<float>(), st, a)
                              : launch_k(cnx_bwd_conv_kernel<u16>, grid, lds_bwd_conv_bytes<u16>(), st, a);
    if (!rc) rc = reduce_rows(ws, a.geo, REC_CONV, 9 * 256, 32, 16, dscale, dshift, st);
    if (!rc) {
        RedSegs sg = {{{g->conv_w, 0, 9 * 256}, {nullptr, 0, 0}, {nullptr, 0, 0}, {nullptr, 0, 0}}, 1};
        rc = reduce_blocks(ws, grid * a.geo.kmax, REC_CONV, 9 * 256, sg, st);
    }
    if (!rc) {
        RedSegs sg = {{{g->con_b, 0, 16}, {g->conv_b, 16, 32}, {g->grn_beta, 32, 64}, {nullptr, 0, 0}}, 3};
        rc = reduce_blocks(ws + grid * a.geo.kmax * REC_CONV, grid, REC_TAIL, REC_TAIL, sg, st);
    }
    return rc;
}

extern "C" int mfc_ln16_fwd(int dtype, int64_t n_pixels, const void* x, void* y, float* rstd, void* stream) {
    if (!x || !y) return MFC_EFAULT;
    if (n_pixels <= 0 || (dtype != MFC_F32 && dtype != MFC_BF16)) return MFC_EINVAL;
    int64_t grid = ceil_div64(n_pixels, 256);
    if (grid > 16384) grid = 16384;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MFC_F32)
        hipLaunchKernelGGL(ln16_kernel<float>, dim3((unsigned)grid), dim3(256), 0, st, n_pixels, (const float*)x,
                           (float*)y, rstd);
    else
        hipLaunchKernelGGL(ln16_kernel<u16>, dim3((unsigned)grid), dim3(256), 0, st, n_pixels, (const u16*)x, (u16*)y,
                           rstd);
    return mfc_launch_status();
}

extern "C" int mfc_ln16_jvp(int dtype, int64_t n_pixels, const void* n, const float* rstd, const void* xdot,
                            void* ndot, void* stream) {
    if (!n || !rstd || !xdot || !ndot) return MFC_EFAULT;
    if (n_pixels <= 0 || (dtype != MFC_F32 && dtype != MFC_BF16)) return MFC_EINVAL;
    int64_t grid = ceil_div64(n_pixels, 256);
    if (grid > 16384) grid = 16384;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MFC_F32)
        hipLaunchKernelGGL(ln16_jvp_kernel<float>, dim3((unsigned)grid), dim3(256), 0, st, n_pixels, (const float*)n,
                           rstd, (const float*)xdot, (float*)ndot);
    else
        hipLaunchKernelGGL(ln16_jvp_kernel<u16>, dim3((unsigned)grid), dim3(256), 0, st, n_pixels, (const u16*)n, rstd,
                           (const u16*)xdot, (u16*)ndot);
    return mfc_launch_status();
}

extern "C" int64_t mfc_cnx_max_blocks(int64_t n) {
    const int64_t old = MAX_BLOCKS;
    if (n >= 0) MAX_BLOCKS = n;      // 0 restores the per-kernel defaults
    return old;
}
