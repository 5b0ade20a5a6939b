// Shared device/host helpers for the gfx950 (CDNA4, wave64) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mfc.h"

#define MFC_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

// ---- bf16 <-> f32 (round-to-nearest-even; NaN kept NaN) ----------------
__device__ __host__ inline float bf16_to_f32(u16 h) {
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)h) << 16;
    return c.f;
}
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __host__ inline u16 f32_to_bf16(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    // gfx950 has v_cvt_pk_bf16_f32 (RNE, NaN kept NaN): one instruction instead of ~7
    return __builtin_bit_cast(u16, (__bf16)f);
#else
    union { uint32_t u; float f; } c;
    c.f = f;
    if ((c.u & 0x7fffffffu) > 0x7f800000u) return (u16)((c.u >> 16) | 0x40);  // quiet NaN
    uint32_t r = c.u + 0x7fffu + ((c.u >> 16) & 1u);
    return (u16)(r >> 16);
#endif
}
// two floats -> packed bf16x2 in one v_cvt_pk_bf16_f32
__device__ inline uint32_t pack_bf16x2(float a, float b) {
    const bf16x2_t v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, v);
}

// storage-type traits: T = float or u16 (bf16 bits)
template <typename T> struct St;
template <> struct St<float> {
    __device__ static inline float ld(const float* p) { return *p; }
    __device__ static inline void st(float* p, float v) { *p = v; }
    __device__ static inline float cvt(float v) { return v; }
};
template <> struct St<u16> {
    __device__ static inline float ld(const u16* p) { return bf16_to_f32(*p); }
    __device__ static inline void st(u16* p, float v) { *p = f32_to_bf16(v); }
    __device__ static inline u16 cvt(float v) { return f32_to_bf16(v); }
};

// tanh-GELU (jax.nn.gelu(approximate=True); models/conv_flow.py:88,175,201,
// models/mlp_flow.py:29) and its derivative.
#define MFC_GELU_K0 0.7978845608028654f /* sqrt(2/pi) */
#define MFC_GELU_K1 0.044715f
// 0.5 (1 + tanh a) = sigmoid(2a): one v_exp_f32 + one v_rcp_f32 instead of tanhf.
#define MFC_LOG2E 1.4426950408889634f
__device__ inline float gelu_sig(float x) {
    // sigmoid(2a) = 1 / (1 + 2^(-2a log2 e)); constants folded, raw v_exp_f32 / v_rcp_f32
    const float c0 = -2.0f * MFC_GELU_K0 * MFC_LOG2E, c1 = c0 * MFC_GELU_K1;
    const float t = x * (c0 + c1 * x * x);
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}
__device__ inline float gelu_f(float x) { return x * gelu_sig(x); }
__device__ inline float gelu_grad_f(float x) {
    const float sg = gelu_sig(x);
    const float da2 = (2.0f * MFC_GELU_K0) * (1.0f + 3.0f * MFC_GELU_K1 * x * x);
    return sg + x * sg * (1.0f - sg) * da2;
}
// value and derivative sharing the sigmoid
__device__ inline void gelu_both(float x, float& g, float& dg) {
    const float sg = gelu_sig(x);
    const float da2 = (2.0f * MFC_GELU_K0) * (1.0f + 3.0f * MFC_GELU_K1 * x * x);
    g = x * sg;
    dg = sg + g * (1.0f - sg) * da2;
}
// The same on four values held as one f32x4 (an MFMA accumulator is one): the plain arithmetic compiles to PACKED f32
// VALU (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two lanes' worth of f32 per issue slot -- the f32 VALU peak of
// MI355X_MICROARCH.md is the packed rate), the transcendentals stay scalar.  Probe tools/probe/gelu_rate.hip: 284 vs 366
// cycles per 8 values.  (Compiler-driven SLP packing of arbitrary scalar code is a different thing and stays off: it
// pays v_mov shuffles to build register pairs; here the operands already are aligned register quads.)
__device__ inline f32x4 splat4(float x) { return f32x4{x, x, x, x}; }
__device__ inline f32x4 fma4(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ inline f32x4 ld_f32x4(const float* p) { return f32x4{p[0], p[1], p[2], p[3]}; }
__device__ inline f32x4 gelu_sig4(f32x4 x) {
    const float c0 = -2.0f * MFC_GELU_K0 * MFC_LOG2E, c1 = c0 * MFC_GELU_K1;
    const f32x4 t = x * fma4(x * splat4(c1), x, splat4(c0));
    f32x4 d;
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = __builtin_amdgcn_exp2f(t[i]);
    d = d + splat4(1.0f);
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = __builtin_amdgcn_rcpf(d[i]);
    return d;
}
__device__ inline f32x4 gelu4(f32x4 x) { return x * gelu_sig4(x); }
__device__ inline void gelu_both4(f32x4 x, f32x4& g, f32x4& dg) {
    const f32x4 sg = gelu_sig4(x);
    const f32x4 da2 = fma4(x * splat4(2.0f * MFC_GELU_K0 * 3.0f * MFC_GELU_K1), x, splat4(2.0f * MFC_GELU_K0));
    g = x * sg;
    dg = fma4(fma4(g, -sg, g), da2, sg);      // sg + g (1 - sg) da2; the negation is a source modifier of the packed fma
}
// second derivative, needed for d/dx of (t * gelu'(x)) in the tangent's backward
// (not on the iMF path: the tangent carries no gradient) -- kept out.

// ---- MFMA 16x16 "K16 step": D += A[16 x 16k] * B[16k x 16] ----------------
// Lane l = 16*q + r.  A fragment: 4 consecutive k (4q..4q+3) of row r;
// B fragment: the same 4 k of column r.  fp32 storage issues four
// v_mfma_f32_16x16x4_f32 (exact fp32 fma chain), bf16 storage one
// v_mfma_f32_16x16x16_bf16.  C/D: col = l&15, row = 4*(l>>4) + reg.
template <typename T> struct Frag;
template <> struct Frag<float> { typedef f32x4 type; };
template <> struct Frag<u16> { typedef s16x4 type; };

__device__ inline void mma16(f32x4& acc, const f32x4& a, const f32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc, 0, 0, 0);
}
__device__ inline void mma16(f32x4& acc, const s16x4& a, const s16x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0);
}
__device__ inline void make_frag(f32x4& f, float a, float b, float c, float d) { f = f32x4{a, b, c, d}; }
__device__ inline void make_frag(s16x4& f, float a, float b, float c, float d) {
    const uint32_t lo = pack_bf16x2(a, b), hi = pack_bf16x2(c, d);
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    f = __builtin_bit_cast(s16x4, (u32x2){lo, hi});
}

static inline int mfc_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MFC_OK : MFC_EHIP;
}

// One AdamW element update (optax.adamw: m, v moments, bias correction bc = 1 - beta^step, decoupled weight decay).
// Shared by mfc_adamw and the fused mfc_gemm_adamw epilogue; contraction is off so both evaluate the same
// rounding sequence and stay bit-identical.
__device__ inline void adamw_elem(float& p, float& m, float& v, float g, float lr, float b1, float b2, float eps,
                                  float wd, float bc1, float bc2) {
#pragma clang fp contract(off)
    const float mm = b1 * m + (1.0f - b1) * g;
    const float vv = b2 * v + (1.0f - b2) * g * g;
    const float upd = (mm / bc1) / (sqrtf(vv / bc2) + eps) + wd * p;
    p = p - lr * upd;
    m = mm;
    v = vv;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
