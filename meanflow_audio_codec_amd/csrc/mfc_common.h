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
__device__ __host__ inline u16 f32_to_bf16(float f) {
    union { uint32_t u; float f; } c;
    c.f = f;
    if ((c.u & 0x7fffffffu) > 0x7f800000u) return (u16)((c.u >> 16) | 0x40);  // quiet NaN
    uint32_t r = c.u + 0x7fffu + ((c.u >> 16) & 1u);
    return (u16)(r >> 16);
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
__device__ inline float gelu_f(float x) {
    float a = MFC_GELU_K0 * (x + MFC_GELU_K1 * x * x * x);
    return 0.5f * x * (1.0f + tanhf(a));
}
__device__ inline float gelu_grad_f(float x) {
    float x2 = x * x;
    float a = MFC_GELU_K0 * (x + MFC_GELU_K1 * x * x2);
    float th = tanhf(a);
    float da = MFC_GELU_K0 * (1.0f + 3.0f * MFC_GELU_K1 * x2);
    return 0.5f * (1.0f + th) + 0.5f * x * (1.0f - th * th) * da;
}
// second derivative, needed for d/dx of (t * gelu'(x)) in the tangent's backward
// (not on the iMF path: the tangent carries no gradient) -- kept out.

static inline int mfc_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MFC_OK : MFC_EHIP;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
