// Element-wise / reduction kernels of the loss step (HBM-bound, vectorised):
// time embedding (+tangent), noise + (t, r) sampling (Philox), interpolation
// and target, GELU (+tangent rows, +backward), the compound iMF / MF / FM
// loss with its gradient seed, column sums for bias gradients, axpby and the
// fused AdamW update.
#include "mfc_common.h"
#include <cstdlib>

namespace {

constexpr int ET = 256;

inline unsigned grid_for(int64_t n, int per_thread = 1) {
    int64_t b = ceil_div64(n, (int64_t)ET * per_thread);
    if (b > 16384) b = 16384;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// ---- Philox4x32-10 ---------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };
__device__ inline U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        U4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}
__device__ inline float u01(uint32_t u) { return ((float)(u >> 8) + 0.5f) * (1.0f / 16777216.0f); }
__device__ inline void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
    const float r = sqrtf(-2.0f * logf(u01(a)));
    float s, c;
    sincospif(2.0f * u01(b), &s, &c);
    n0 = r * c;
    n1 = r * s;
}
// 4 standard normals for (stream, row, quad index)
__device__ inline void normal4(uint64_t seed, uint32_t stream, uint64_t row, uint32_t quad, float out[4]) {
    U4 c{quad, (uint32_t)row, (uint32_t)(row >> 32), stream};
    const U4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    box_muller(r.x, r.y, out[0], out[1]);
    box_muller(r.z, r.w, out[2], out[3]);
}

// ---- sinusoidal time embedding (meanflow_audio_codec/utils.py:5-13) ---------
// cond[r] = emb(t[r]) + emb(h[r]) (+ add[r]); cdot[r] = tdot emb'(t) + hdot emb'(h)
__global__ void time_embed_kernel(int64_t R, int dim, const float* t, const float* h, const float* tdot,
                                  const float* hdot, const float* add, float* cond, float* cdot) {
    const int half = dim / 2;
    const int64_t total = R * dim;
    for (int64_t o = blockIdx.x * (int64_t)ET + threadIdx.x; o < total; o += (int64_t)gridDim.x * ET) {
        const int64_t r = o / dim;
        const int c = (int)(o - r * dim);
        const int j = c < half ? c : c - half;
        const float f = expf(-logf(10000.0f) * (float)j / (float)half);
        const float at = t[r] * f, ah = h[r] * f;
        float v, dv = 0.f;
        if (c < half) {
            v = cosf(at) + cosf(ah);
            if (cdot) dv = -(tdot ? tdot[r] : 1.0f) * f * sinf(at) - (hdot ? hdot[r] : 1.0f) * f * sinf(ah);
        } else {
            v = sinf(at) + sinf(ah);
            if (cdot) dv = (tdot ? tdot[r] : 1.0f) * f * cosf(at) + (hdot ? hdot[r] : 1.0f) * f * cosf(ah);
        }
        if (add) v += add[o];
        cond[o] = v;
        if (cdot) cdot[o] = dv;
    }
}

// ---- (t, r) sampling: meanflow_audio_codec/utils.py:32-45 --------------------
// t,r = sigmoid(N(mean,std)); t=max, r=min; GLOBAL rows < data_size: r = t.  data_size = int(Bglobal * data_proportion)
// is computed ONCE on the host (as utils.sample_tr does) so kernel and host orchestration can never disagree.
// Local row i is global row row0 + i * row_stride (data-parallel shards: contiguous stride 1, interleaved stride = world).
__global__ void sample_tr_kernel(uint64_t seed, uint64_t step, int64_t row0, int64_t row_stride, int64_t B,
                                 int64_t data_size, float mean, float stdv, int pair, float* t, float* r) {
    const int64_t i = blockIdx.x * (int64_t)ET + threadIdx.x;
    if (i >= B) return;
    const int64_t grow = row0 + i * row_stride;
    float n[4];
    normal4(seed, 0x7472u, (uint64_t)grow, (uint32_t)step, n);
    float a = 1.0f / (1.0f + expf(-(n[0] * stdv + mean)));
    if (!pair) { t[i] = a; return; }
    float b = 1.0f / (1.0f + expf(-(n[1] * stdv + mean)));
    const float tt = fmaxf(a, b), rr = fminf(a, b);
    t[i] = tt;
    r[i] = grow < data_size ? tt : rr;
}

// ---- noise + interpolation + target -----------------------------------------
// z = (1-t) x + (nmin + nmax t) e ; target = nmax e - x   (noise_schedules.py:69-88)
// e is drawn from Philox (seed, step, global row) when e_in == nullptr.
template <typename T>
__global__ void flow_prepare_kernel(int64_t B, int64_t D, const float* x, const float* e_in, const float* t,
                                    float nmin, float nmax, uint64_t seed, uint64_t step, int64_t row0,
                                    int64_t row_stride, T* z, float* target, float* e_out) {
    const int64_t quads = (D + 3) / 4;
    const int64_t total = B * quads;
    for (int64_t o = blockIdx.x * (int64_t)ET + threadIdx.x; o < total; o += (int64_t)gridDim.x * ET) {
        const int64_t b = o / quads;
        const int64_t qd = o - b * quads;
        float e4[4];
        if (e_in) {
#pragma unroll
            for (int i = 0; i < 4; ++i) e4[i] = (4 * qd + i < D) ? e_in[b * D + 4 * qd + i] : 0.f;
        } else {
            normal4(seed ^ (step * 0x9E3779B97F4A7C15ull), 0x6e6fu, (uint64_t)(row0 + b * row_stride), (uint32_t)qd, e4);
        }
        const float tt = t[b];
        const float a = 1.0f - tt, c = nmin + nmax * tt;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t d = 4 * qd + i;
            if (d < D) {
                const float xv = x[b * D + d];
                St<T>::st(z + b * D + d, a * xv + c * e4[i]);
                target[b * D + d] = nmax * e4[i] - xv;
                if (e_out) e_out[b * D + d] = e4[i];
            }
        }
    }
}

__global__ void randn_kernel(uint64_t seed, uint64_t stream, int64_t row0, int64_t B, int64_t D, float* out) {
    const int64_t quads = (D + 3) / 4;
    const int64_t total = B * quads;
    for (int64_t o = blockIdx.x * (int64_t)ET + threadIdx.x; o < total; o += (int64_t)gridDim.x * ET) {
        const int64_t b = o / quads, qd = o - b * quads;
        float e4[4];
        normal4(seed, (uint32_t)stream, (uint64_t)(row0 + b), (uint32_t)qd, e4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * qd + i < D) out[b * D + 4 * qd + i] = e4[i];
    }
}

// The same draw with the first global row taken from DEVICE memory and the result written in the storage dtype: the
// launch is then a fixed graph node whose noise still changes from replay to replay (randn_advance_kernel moves the
// counter after every block has read it: stream order).
template <typename T>
__global__ void randn_dev_kernel(uint64_t seed, uint64_t stream, const int64_t* row0_dev, int64_t B, int64_t D, T* out) {
    const int64_t row0 = *row0_dev;
    const int64_t quads = (D + 3) / 4;
    const int64_t total = B * quads;
    for (int64_t o = blockIdx.x * (int64_t)ET + threadIdx.x; o < total; o += (int64_t)gridDim.x * ET) {
        const int64_t b = o / quads, qd = o - b * quads;
        float e4[4];
        normal4(seed, (uint32_t)stream, (uint64_t)(row0 + b), (uint32_t)qd, e4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * qd + i < D) St<T>::st(out + b * D + 4 * qd + i, e4[i]);
    }
}
__global__ void randn_advance_kernel(int64_t* row0_dev, int64_t advance) { *row0_dev += advance; }

// ---- GELU on [M,N] with tangent rows / backward ------------------------------
template <typename T>
__global__ void gelu_fwd_kernel(int64_t M, int64_t N, int64_t act_rows, const T* pre, T* out) {
    const int64_t total = M * N;
    for (int64_t o = blockIdx.x * (int64_t)ET + threadIdx.x; o < total; o += (int64_t)gridDim.x * ET) {
        const int64_t row = o / N;
        const float v = St<T>::ld(pre + o);
        float y;
        if (row < act_rows) y = gelu_f(v);
        else y = v * gelu_grad_f(St<T>::ld(pre + o - act_rows * N));
        St<T>::st(out + o, y);
    }
}
template <typename T>
__global__ void gelu_bwd_kernel(int64_t n, const T* pre, const T* dout, T* din) {
    for (int64_t o = blockIdx.x * (int64_t)ET + threadIdx.x; o < n; o += (int64_t)gridDim.x * ET)
        St<T>::st(din + o, St<T>::ld(dout + o) * gelu_grad_f(St<T>::ld(pre + o)));
}

// ---- loss ---------------------------------------------------------------------
// kind 0 (iMF / FM): delta = u + coef*dudt - target, coef = (t - r)      loss_strategies.py:270
// kind 1 (MF):       delta = u - (target - clip(t-r,0,1) dudt)           loss_strategies.py:184-188
// pe[b] = sum_d delta^2: workgroup (x, b) stores its partial sum in part[b * gridDim.x + x]; loss_finalize_kernel adds
// the partials of an example in ascending x (fixed order, no atomics: the loss weight 1/(pe + c) feeds every gradient)
template <typename T>
__global__ void loss_pe_kernel(int kind, int64_t B, int64_t D, const T* u, const T* dudt, int64_t tan0, int64_t n_tan,
                               const float* t, const float* r, const float* target, float* part) {
    __shared__ float red[ET / 64];
    const int64_t b = blockIdx.y;
    float coef = 0.f;
    if (dudt && b >= tan0 && b < tan0 + n_tan) {
        coef = t[b] - r[b];
        if (kind == 1) coef = fminf(fmaxf(coef, 0.f), 1.f);
    }
    float acc = 0.f;
    for (int64_t d = blockIdx.x * (int64_t)ET + threadIdx.x; d < D; d += (int64_t)gridDim.x * ET) {
        float v = St<T>::ld(u + b * D + d) - target[b * D + d];
        if (coef != 0.f) v += coef * St<T>::ld(dudt + (b - tan0) * D + d);
        acc += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < ET / 64; ++i) s += red[i];
        part[b * gridDim.x + blockIdx.x] = s;
    }
}

// per-example weights and the scalar loss:
//  mode 0: weighted_l2_loss (utils.py:16-25)   w = 1/(pe + c)^p, loss = mean_b(w pe),   seed = 2 w / Bg
//  mode 1: plain MSE                           loss = sum pe / (Bg D),                  seed = 2 / (Bg D)
//  mode 2: MeanFlow adaptive (loss_strategies.py:190-196) dsq = pe/D, w = 1/(dsq+c)^(1-gamma),
//          loss = mean_b(w dsq), seed = 2 w / (Bg D)
__global__ void loss_finalize_kernel(int mode, int64_t B, int64_t Bglobal, int64_t D, const float* part, int nparts,
                                     float* pe, float p, float c, float* seed, float* loss) {
    __shared__ float contrib[256];
    float acc = 0.f;
    for (int64_t b0 = 0; b0 < B; b0 += blockDim.x) {
      const int64_t b = b0 + threadIdx.x;
      float l = 0.f;
      if (b < B) {
        float v = 0.f;
        for (int k = 0; k < nparts; ++k) v += part[b * nparts + k];
        pe[b] = v;
        float w;
        if (mode == 0) { w = 1.0f / powf(v + c, p); l = w * v / (float)Bglobal; seed[b] = 2.0f * w / (float)Bglobal; }
        else if (mode == 1) { l = v / ((float)Bglobal * (float)D); seed[b] = 2.0f / ((float)Bglobal * (float)D); }
        else {
            const float dsq = v / (float)D;
            w = 1.0f / powf(dsq + c, p);
            l = w * dsq / (float)Bglobal;
            seed[b] = 2.0f * w / ((float)Bglobal * (float)D);
        }
      }
      // examples in ascending order (the reference's mean over the batch): one thread adds this round's contributions
      contrib[threadIdx.x] = l;
      __syncthreads();
      if (threadIdx.x == 0) {
          const int n = (int)((B - b0) < (int64_t)blockDim.x ? (B - b0) : (int64_t)blockDim.x);
          for (int i = 0; i < n; ++i) acc += contrib[i];
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) *loss = acc;
}

// du[b,d] = seed[b] * delta[b,d]
template <typename T>
__global__ void loss_grad_kernel(int kind, int64_t B, int64_t D, const T* u, const T* dudt, int64_t tan0, int64_t n_tan,
                                 const float* t, const float* r, const float* target, const float* seed, T* du) {
    const int64_t total = B * D;
    for (int64_t o = blockIdx.x * (int64_t)ET + threadIdx.x; o < total; o += (int64_t)gridDim.x * ET) {
        const int64_t b = o / D;
        float v = St<T>::ld(u + o) - target[o];
        if (dudt && b >= tan0 && b < tan0 + n_tan) {
            float coef = t[b] - r[b];
            if (kind == 1) coef = fminf(fmaxf(coef, 0.f), 1.f);
            v += coef * St<T>::ld(dudt + o - tan0 * D);
        }
        St<T>::st(du + o, seed[b] * v);
    }
}

// ---- column sums (bias gradients) ----------------------------------------------
template <typename T>
__global__ void colsum_kernel(int64_t M, int64_t N, const T* X, int64_t ld, float scale, float* out, int accum) {
    for (int64_t c = blockIdx.x * (int64_t)ET + threadIdx.x; c < N; c += (int64_t)gridDim.x * ET) {
        float acc = 0.f;
        for (int64_t m = 0; m < M; ++m) acc += St<T>::ld(X + m * ld + c);
        acc *= scale;
        out[c] = accum ? out[c] + acc : acc;
    }
}
// 16-byte-vector variant: each thread owns VW consecutive columns, 4 rows in flight
template <typename T, int VW>
__global__ void __launch_bounds__(ET) colsum_vec_kernel(int64_t M, int64_t N, const T* X, int64_t ld, float scale,
                                                        float* out, int accum) {
    typedef T vt __attribute__((ext_vector_type(VW)));
    const int64_t groups = N / VW;
    for (int64_t gidx = blockIdx.x * (int64_t)ET + threadIdx.x; gidx < groups; gidx += (int64_t)gridDim.x * ET) {
        const T* p = X + gidx * VW;
        float acc[VW];
#pragma unroll
        for (int i = 0; i < VW; ++i) acc[i] = 0.f;
        int64_t m = 0;
        for (; m + 4 <= M; m += 4) {
            vt v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const vt*>(p + (m + u) * ld);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < VW; ++i) {
                    T e = v[u][i];
                    acc[i] += St<T>::ld(&e);
                }
        }
        for (; m < M; ++m) {
            const vt v = *reinterpret_cast<const vt*>(p + m * ld);
#pragma unroll
            for (int i = 0; i < VW; ++i) {
                T e = v[i];
                acc[i] += St<T>::ld(&e);
            }
        }
#pragma unroll
        for (int i = 0; i < VW; ++i) {
            const float r = acc[i] * scale;
            out[gidx * VW + i] = accum ? out[gidx * VW + i] + r : r;
        }
    }
}

template <typename T>
__global__ void axpby_kernel(int64_t n, float a, const T* x, float b, const T* y, T* out) {
    for (int64_t o = blockIdx.x * (int64_t)ET + threadIdx.x; o < n; o += (int64_t)gridDim.x * ET) {
        float v = a * St<T>::ld(x + o);
        if (y) v += b * St<T>::ld(y + o);
        St<T>::st(out + o, v);
    }
}

// 16 bytes per thread and tensor (n a multiple of the vector width, 16-byte aligned pointers): same arithmetic per element
template <typename T>
__global__ void axpby_vec_kernel(int64_t nvec, float a, const T* x, float b, const T* y, T* out) {
    constexpr int VW = 16 / (int)sizeof(T);
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    for (int64_t o = blockIdx.x * (int64_t)ET + threadIdx.x; o < nvec; o += (int64_t)gridDim.x * ET) {
        const u32x4_t xv = *reinterpret_cast<const u32x4_t*>(x + o * VW);
        u32x4_t yv = u32x4_t{0u, 0u, 0u, 0u};
        if (y) yv = *reinterpret_cast<const u32x4_t*>(y + o * VW);
        T xs[VW], ys[VW], os[VW];
        __builtin_memcpy(xs, &xv, 16);
        __builtin_memcpy(ys, &yv, 16);
#pragma unroll
        for (int k = 0; k < VW; ++k) {
            float v = a * St<T>::ld(xs + k);
            if (y) v += b * St<T>::ld(ys + k);
            St<T>::st(os + k, v);
        }
        u32x4_t ov;
        __builtin_memcpy(&ov, os, 16);
        *reinterpret_cast<u32x4_t*>(out + o * VW) = ov;
    }
}

template <typename TI, typename TO>
__global__ void cast_kernel(int64_t n, const TI* x, TO* out) {
    for (int64_t o = blockIdx.x * (int64_t)ET + threadIdx.x; o < n; o += (int64_t)gridDim.x * ET)
        St<TO>::st(out + o, St<TI>::ld(x + o));
}

// ---- AdamW (optax.adamw, trainers/train.py:236) ----------------
// m <- b1 m + (1-b1) g; v <- b2 v + (1-b2) g^2;
// p <- p - lr (mhat / (sqrt(vhat) + eps) + wd p); optional bf16 working copy.
template <typename TG>
__global__ void adamw_kernel(int64_t n, float* p, u16* pw, const TG* g, float gscale, float* m, float* v,
                             float lr, float b1, float b2, float eps, float wd, float bc1, float bc2) {
    for (int64_t o = blockIdx.x * (int64_t)ET + threadIdx.x; o < n; o += (int64_t)gridDim.x * ET) {
        float pv = p[o], mv = m[o], vv = v[o];
        adamw_elem(pv, mv, vv, St<TG>::ld(g + o) * gscale, lr, b1, b2, eps, wd, bc1, bc2);
        p[o] = pv; m[o] = mv; v[o] = vv;
        if (pw) pw[o] = f32_to_bf16(pv);
    }
}

// 4 elements per lane: 16-byte loads/stores of p, m, v, 8/16-byte of the gradient, 8-byte of the bf16 copy.
// NT: non-temporal (streaming) accesses -- every byte is touched exactly once per step.
template <typename TG, bool NT>
__global__ void __launch_bounds__(ET)
adamw_vec_kernel(int64_t n4, float* p, u16* pw, const TG* g, float gscale, float* m, float* v,
                 float lr, float b1, float b2, float eps, float wd, float bc1, float bc2) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    for (int64_t o = blockIdx.x * (int64_t)ET + threadIdx.x; o < n4; o += (int64_t)gridDim.x * ET) {
        f4 gv;
        if constexpr (sizeof(TG) == 4) {
            const f4* gp = reinterpret_cast<const f4*>(g) + o;
            gv = NT ? __builtin_nontemporal_load(gp) : *gp;
        } else {
            const u2* gp = reinterpret_cast<const u2*>(g) + o;
            const u2 t = NT ? __builtin_nontemporal_load(gp) : *gp;
            gv = f4{__builtin_bit_cast(float, (uint32_t)(t[0] << 16)), __builtin_bit_cast(float, (uint32_t)(t[0] & 0xffff0000u)),
                    __builtin_bit_cast(float, (uint32_t)(t[1] << 16)), __builtin_bit_cast(float, (uint32_t)(t[1] & 0xffff0000u))};
        }
        f4* mp = reinterpret_cast<f4*>(m) + o;
        f4* vp = reinterpret_cast<f4*>(v) + o;
        f4* pp = reinterpret_cast<f4*>(p) + o;
        const f4 m0 = NT ? __builtin_nontemporal_load(mp) : *mp;
        const f4 v0 = NT ? __builtin_nontemporal_load(vp) : *vp;
        const f4 p0 = NT ? __builtin_nontemporal_load(pp) : *pp;
        f4 mm = m0, vv = v0, np = p0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pk = np[k], mk = mm[k], vk = vv[k];
            adamw_elem(pk, mk, vk, gv[k] * gscale, lr, b1, b2, eps, wd, bc1, bc2);
            np[k] = pk; mm[k] = mk; vv[k] = vk;
        }
        if (NT) {
            __builtin_nontemporal_store(mm, mp);
            __builtin_nontemporal_store(vv, vp);
            __builtin_nontemporal_store(np, pp);
        } else {
            *mp = mm; *vp = vv; *pp = np;
        }
        if (pw) {
            const u2 w = {pack_bf16x2(np[0], np[1]), pack_bf16x2(np[2], np[3])};
            u2* wp = reinterpret_cast<u2*>(pw) + o;
            if (NT) __builtin_nontemporal_store(w, wp); else *wp = w;
        }
    }
}

}  // namespace

#define DT_OK(dt) ((dt) == MFC_F32 || (dt) == MFC_BF16)

extern "C" int mfc_time_embed(int64_t R, int dim, const float* t, const float* h, const float* tdot,
                              const float* hdot, const float* add, float* cond, float* cond_dot, void* stream) {
    if (!t || !h || !cond) return MFC_EFAULT;
    if (R <= 0 || dim <= 0 || (dim & 1)) return MFC_EINVAL;
    hipLaunchKernelGGL(time_embed_kernel, dim3(grid_for(R * dim)), dim3(ET), 0, (hipStream_t)stream, R, dim, t, h,
                       tdot, hdot, add, cond, cond_dot);
    return mfc_launch_status();
}

extern "C" int mfc_sample_tr(uint64_t seed, uint64_t step, int64_t row0, int64_t row_stride, int64_t B,
                             int64_t data_size, float mean, float std, int pair, float* t, float* r, void* stream) {
    if (!t || (pair && !r)) return MFC_EFAULT;
    if (B <= 0 || row0 < 0 || row_stride < 1 || data_size < 0) return MFC_EINVAL;
    hipLaunchKernelGGL(sample_tr_kernel, dim3(grid_for(B)), dim3(ET), 0, (hipStream_t)stream, seed, step, row0,
                       row_stride, B, data_size, mean, std, pair, t, r);
    return mfc_launch_status();
}

extern "C" int mfc_flow_prepare(int dtype, int64_t B, int64_t D, const float* x, const float* e_in,
                                const float* t, float noise_min, float noise_max, uint64_t seed, uint64_t step,
                                int64_t row0, int64_t row_stride, void* z, float* target, float* e_out,
                                void* stream) {
    if (!x || !t || !z || !target) return MFC_EFAULT;
    if (B <= 0 || D <= 0 || !DT_OK(dtype) || row0 < 0 || row_stride < 1) return MFC_EINVAL;
    const unsigned grid = grid_for(B * ((D + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MFC_F32)
        hipLaunchKernelGGL(flow_prepare_kernel<float>, dim3(grid), dim3(ET), 0, st, B, D, x, e_in, t, noise_min,
                           noise_max, seed, step, row0, row_stride, (float*)z, target, e_out);
    else
        hipLaunchKernelGGL(flow_prepare_kernel<u16>, dim3(grid), dim3(ET), 0, st, B, D, x, e_in, t, noise_min,
                           noise_max, seed, step, row0, row_stride, (u16*)z, target, e_out);
    return mfc_launch_status();
}

extern "C" int mfc_randn(uint64_t seed, uint64_t stream_id, int64_t row0, int64_t B, int64_t D, float* out,
                         void* stream) {
    if (!out) return MFC_EFAULT;
    if (B <= 0 || D <= 0) return MFC_EINVAL;
    hipLaunchKernelGGL(randn_kernel, dim3(grid_for(B * ((D + 3) / 4))), dim3(ET), 0, (hipStream_t)stream, seed,
                       stream_id, row0, B, D, out);
    return mfc_launch_status();
}

extern "C" int mfc_randn_dev(int dtype, uint64_t seed, uint64_t stream_id, int64_t* row0_dev, int64_t advance, int64_t B,
                             int64_t D, void* out, void* stream) {
    if (!out || !row0_dev) return MFC_EFAULT;
    if (B <= 0 || D <= 0 || !DT_OK(dtype)) return MFC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MFC_F32)
        hipLaunchKernelGGL(randn_dev_kernel<float>, dim3(grid_for(B * ((D + 3) / 4))), dim3(ET), 0, st, seed, stream_id,
                           (const int64_t*)row0_dev, B, D, (float*)out);
    else
        hipLaunchKernelGGL(randn_dev_kernel<u16>, dim3(grid_for(B * ((D + 3) / 4))), dim3(ET), 0, st, seed, stream_id,
                           (const int64_t*)row0_dev, B, D, (u16*)out);
    if (advance) hipLaunchKernelGGL(randn_advance_kernel, dim3(1), dim3(1), 0, st, row0_dev, advance);
    return mfc_launch_status();
}

extern "C" int mfc_gelu_fwd(int dtype, int64_t M, int64_t N, int64_t act_rows, const void* pre, void* out,
                            void* stream) {
    if (!pre || !out) return MFC_EFAULT;
    if (M <= 0 || N <= 0 || act_rows <= 0 || act_rows > M || M > 2 * act_rows || !DT_OK(dtype)) return MFC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MFC_F32)
        hipLaunchKernelGGL(gelu_fwd_kernel<float>, dim3(grid_for(M * N)), dim3(ET), 0, st, M, N, act_rows,
                           (const float*)pre, (float*)out);
    else
        hipLaunchKernelGGL(gelu_fwd_kernel<u16>, dim3(grid_for(M * N)), dim3(ET), 0, st, M, N, act_rows,
                           (const u16*)pre, (u16*)out);
    return mfc_launch_status();
}

extern "C" int mfc_gelu_bwd(int dtype, int64_t n, const void* pre, const void* dout, void* din, void* stream) {
    if (!pre || !dout || !din) return MFC_EFAULT;
    if (n <= 0 || !DT_OK(dtype)) return MFC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MFC_F32)
        hipLaunchKernelGGL(gelu_bwd_kernel<float>, dim3(grid_for(n)), dim3(ET), 0, st, n, (const float*)pre,
                           (const float*)dout, (float*)din);
    else
        hipLaunchKernelGGL(gelu_bwd_kernel<u16>, dim3(grid_for(n)), dim3(ET), 0, st, n, (const u16*)pre,
                           (const u16*)dout, (u16*)din);
    return mfc_launch_status();
}

extern "C" int mfc_flow_loss(int dtype, int kind, int mode, int64_t B, int64_t Bglobal, int64_t D, const void* u,
                             const void* dudt, int64_t n_tan, const float* t, const float* r,
                             const float* target, float p, float c, float* pe, float* seed, float* loss,
                             void* du, float* ws, void* stream) {
    if (!u || !target || !pe || !seed || !loss || !ws) return MFC_EFAULT;
    if (dudt && (!t || !r)) return MFC_EFAULT;
    if (B <= 0 || D <= 0 || Bglobal < B || n_tan < -B || n_tan > B || !DT_OK(dtype)) return MFC_EINVAL;
    // n_tan >= 0: the FIRST n_tan rows carry a tangent (dudt row b); n_tan < 0: the LAST -n_tan rows do (dudt row b - (B + n_tan))
    const int64_t tan0 = n_tan >= 0 ? 0 : B + n_tan;
    if (n_tan < 0) n_tan = -n_tan;
    if (kind < 0 || kind > 1 || mode < 0 || mode > 2 || B > 65535) return MFC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int64_t gx = ceil_div64(D, ET * 8);
    if (gx > MFC_FLOW_LOSS_WS_PER_ROW) gx = MFC_FLOW_LOSS_WS_PER_ROW;
    dim3 g1((unsigned)gx, (unsigned)B);
    if (dtype == MFC_F32)
        hipLaunchKernelGGL(loss_pe_kernel<float>, g1, dim3(ET), 0, st, kind, B, D, (const float*)u,
                           (const float*)dudt, tan0, n_tan, t, r, target, ws);
    else
        hipLaunchKernelGGL(loss_pe_kernel<u16>, g1, dim3(ET), 0, st, kind, B, D, (const u16*)u, (const u16*)dudt,
                           tan0, n_tan, t, r, target, ws);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, mode, B, Bglobal, D, (const float*)ws, (int)gx,
                       pe, p, c, seed, loss);
    if (du) {
        if (dtype == MFC_F32)
            hipLaunchKernelGGL(loss_grad_kernel<float>, dim3(grid_for(B * D)), dim3(ET), 0, st, kind, B, D,
                               (const float*)u, (const float*)dudt, tan0, n_tan, t, r, target, seed, (float*)du);
        else
            hipLaunchKernelGGL(loss_grad_kernel<u16>, dim3(grid_for(B * D)), dim3(ET), 0, st, kind, B, D,
                               (const u16*)u, (const u16*)dudt, tan0, n_tan, t, r, target, seed, (u16*)du);
    }
    return mfc_launch_status();
}

// Tall matrices (M in the 10^5 of the Mixer's [B * tokens, C] activations): the column-parallel kernels above would
// leave all but a few CUs idle and walk M rows serially.  Two fixed-order stages instead: workgroup (chunk, colgroup)
// sums its rows of its 16-byte column groups -- threads [tr][tc], tr strides the rows, the tr partials are added in
// tr order through LDS -- into partial[chunk][N]; the second kernel adds the chunks in order.  Bitwise reproducible.
template <typename T, int VW>
__global__ void __launch_bounds__(ET) colsum_tall_kernel(int64_t M, int64_t N, const T* X, int64_t ld, int tcols, int64_t mchunk,
                                                         float* partial) {
    typedef T vt __attribute__((ext_vector_type(VW)));
    __shared__ float red[ET * VW];
    const int tc = threadIdx.x % tcols, tr = threadIdx.x / tcols, trows = ET / tcols;
    const int64_t cg = blockIdx.x * (int64_t)tcols + tc;                 // 16-byte column group
    const int64_t m0 = blockIdx.y * mchunk, m1 = m0 + mchunk < M ? m0 + mchunk : M;
    float acc[VW];
#pragma unroll
    for (int i = 0; i < VW; ++i) acc[i] = 0.f;
    if (cg * VW < N) {
        const T* p = X + cg * VW;
        int64_t m = m0 + tr;
        for (; m + 3 * trows < m1; m += 4 * trows) {                      // four rows in flight
            vt v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const vt*>(p + (m + u * trows) * ld);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < VW; ++i) { T e = v[u][i]; acc[i] += St<T>::ld(&e); }
        }
        for (; m < m1; m += trows) {
            const vt v = *reinterpret_cast<const vt*>(p + m * ld);
#pragma unroll
            for (int i = 0; i < VW; ++i) { T e = v[i]; acc[i] += St<T>::ld(&e); }
        }
    }
#pragma unroll
    for (int i = 0; i < VW; ++i) red[(tr * tcols + tc) * VW + i] = acc[i];
    __syncthreads();
    if (tr == 0 && cg * VW < N) {
#pragma unroll
        for (int i = 0; i < VW; ++i) {
            float sum = red[tc * VW + i];
            for (int k = 1; k < trows; ++k) sum += red[(k * tcols + tc) * VW + i];
            partial[blockIdx.y * N + cg * VW + i] = sum;
        }
    }
}
__global__ void __launch_bounds__(ET) colsum_chunks_kernel(int64_t nchunk, int64_t N, const float* partial, float scale, float* out,
                                                           int accum) {
    const int64_t c = blockIdx.x * (int64_t)ET + threadIdx.x;
    if (c >= N) return;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    int64_t k = 0;
    for (; k + 3 < nchunk; k += 4) {
        v0 += partial[k * N + c]; v1 += partial[(k + 1) * N + c]; v2 += partial[(k + 2) * N + c]; v3 += partial[(k + 3) * N + c];
    }
    for (; k < nchunk; ++k) v0 += partial[k * N + c];
    const float acc = ((v0 + v1) + (v2 + v3)) * scale;
    out[c] = accum ? out[c] + acc : acc;
}
namespace {
struct TallPlan { int vw, tcols; int64_t cgroups, nchunk, mchunk; };
inline bool tall_plan(int dtype, int64_t M, int64_t N, TallPlan& t) {
    t.vw = dtype == MFC_F32 ? 4 : 8;
    if (N % t.vw || M < 2048) return false;
    const int64_t groups = N / t.vw;
    t.tcols = groups >= 64 ? 64 : (groups >= 32 ? 32 : (groups >= 16 ? 16 : (groups >= 8 ? 8 : (groups >= 4 ? 4 : (groups >= 2 ? 2 : 1)))));
    t.cgroups = ceil_div64(groups, t.tcols);
    const int trows = ET / t.tcols;
    int64_t nchunk = ceil_div64(2048, t.cgroups);                      // ~8 workgroups per CU in total
    const int64_t most = ceil_div64(M, (int64_t)trows * 8);              // at least 8 row sweeps per workgroup
    if (nchunk > most) nchunk = most;
    if (nchunk < 1) nchunk = 1;
    t.mchunk = ceil_div64(M, nchunk);
    t.nchunk = ceil_div64(M, t.mchunk);
    return true;
}
}  // namespace

extern "C" int64_t mfc_colsum_ws_elems(int dtype, int64_t M, int64_t N) {
    TallPlan t;
    if (M <= 0 || N <= 0 || !DT_OK(dtype)) return -1;
    return tall_plan(dtype, M, N, t) ? t.nchunk * N : 0;
}

extern "C" int mfc_colsum_tall(int dtype, int64_t M, int64_t N, const void* X, int64_t ld, float scale, float* out,
                               int accumulate, float* ws, void* stream) {
    if (!X || !out || !ws) return MFC_EFAULT;
    if (M <= 0 || N <= 0 || ld < N || !DT_OK(dtype)) return MFC_EINVAL;
    TallPlan t;
    const size_t es = dtype == MFC_F32 ? 4 : 2;
    if (!tall_plan(dtype, M, N, t) || (ld * es) % 16 != 0 || ((uintptr_t)X % 16) != 0 || t.nchunk > 65535) return MFC_ENOSYS;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)t.cgroups, (unsigned)t.nchunk);
    if (dtype == MFC_F32)
        hipLaunchKernelGGL((colsum_tall_kernel<float, 4>), grid, dim3(ET), 0, st, M, N, (const float*)X, ld, t.tcols, t.mchunk, ws);
    else
        hipLaunchKernelGGL((colsum_tall_kernel<u16, 8>), grid, dim3(ET), 0, st, M, N, (const u16*)X, ld, t.tcols, t.mchunk, ws);
    hipLaunchKernelGGL(colsum_chunks_kernel, dim3((unsigned)ceil_div64(N, ET)), dim3(ET), 0, st, t.nchunk, N, ws, scale, out, accumulate);
    return mfc_launch_status();
}

extern "C" int mfc_colsum(int dtype, int64_t M, int64_t N, const void* X, int64_t ld, float scale, float* out,
                          int accumulate, void* stream) {
    if (!X || !out) return MFC_EFAULT;
    if (M <= 0 || N <= 0 || ld < N || !DT_OK(dtype)) return MFC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int vw = dtype == MFC_F32 ? 4 : 8;
    const size_t es = dtype == MFC_F32 ? 4 : 2;
    if (N % vw == 0 && (ld * es) % 16 == 0 && ((uintptr_t)X % 16) == 0 && N >= 4096) {
        if (dtype == MFC_F32)
            hipLaunchKernelGGL((colsum_vec_kernel<float, 4>), dim3(grid_for(N / 4)), dim3(ET), 0, st, M, N,
                               (const float*)X, ld, scale, out, accumulate);
        else
            hipLaunchKernelGGL((colsum_vec_kernel<u16, 8>), dim3(grid_for(N / 8)), dim3(ET), 0, st, M, N,
                               (const u16*)X, ld, scale, out, accumulate);
        return mfc_launch_status();
    }
    if (dtype == MFC_F32)
        hipLaunchKernelGGL(colsum_kernel<float>, dim3(grid_for(N)), dim3(ET), 0, st, M, N, (const float*)X, ld,
                           scale, out, accumulate);
    else
        hipLaunchKernelGGL(colsum_kernel<u16>, dim3(grid_for(N)), dim3(ET), 0, st, M, N, (const u16*)X, ld, scale,
                           out, accumulate);
    return mfc_launch_status();
}

extern "C" int mfc_axpby(int dtype, int64_t n, float a, const void* x, float b, const void* y, void* out,
                         void* stream) {
    if (!x || !out) return MFC_EFAULT;
    if (n <= 0 || !DT_OK(dtype)) return MFC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int vw = dtype == MFC_F32 ? 4 : 8;
    if (n % vw == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)out) & 15) == 0) {
        const int64_t nvec = n / vw;
        if (dtype == MFC_F32)
            hipLaunchKernelGGL(axpby_vec_kernel<float>, dim3(grid_for(nvec)), dim3(ET), 0, st, nvec, a, (const float*)x, b,
                               (const float*)y, (float*)out);
        else
            hipLaunchKernelGGL(axpby_vec_kernel<u16>, dim3(grid_for(nvec)), dim3(ET), 0, st, nvec, a, (const u16*)x, b,
                               (const u16*)y, (u16*)out);
        return mfc_launch_status();
    }
    if (dtype == MFC_F32)
        hipLaunchKernelGGL(axpby_kernel<float>, dim3(grid_for(n)), dim3(ET), 0, st, n, a, (const float*)x, b,
                           (const float*)y, (float*)out);
    else
        hipLaunchKernelGGL(axpby_kernel<u16>, dim3(grid_for(n)), dim3(ET), 0, st, n, a, (const u16*)x, b,
                           (const u16*)y, (u16*)out);
    return mfc_launch_status();
}

extern "C" int mfc_cast(int src_dtype, int dst_dtype, int64_t n, const void* x, void* out, void* stream) {
    if (!x || !out) return MFC_EFAULT;
    if (n <= 0 || !DT_OK(src_dtype) || !DT_OK(dst_dtype)) return MFC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const unsigned g = grid_for(n);
    if (src_dtype == MFC_F32 && dst_dtype == MFC_BF16)
        hipLaunchKernelGGL((cast_kernel<float, u16>), dim3(g), dim3(ET), 0, st, n, (const float*)x, (u16*)out);
    else if (src_dtype == MFC_BF16 && dst_dtype == MFC_F32)
        hipLaunchKernelGGL((cast_kernel<u16, float>), dim3(g), dim3(ET), 0, st, n, (const u16*)x, (float*)out);
    else if (src_dtype == MFC_F32)
        hipLaunchKernelGGL((cast_kernel<float, float>), dim3(g), dim3(ET), 0, st, n, (const float*)x, (float*)out);
    else
        hipLaunchKernelGGL((cast_kernel<u16, u16>), dim3(g), dim3(ET), 0, st, n, (const u16*)x, (u16*)out);
    return mfc_launch_status();
}

// ---- multi-tensor AdamW: one launch for up to MFC_ADAMW_MULTI_MAX leaves ----------------------------------------
// The descriptors travel in the kernel arguments; a workgroup owns MT_SPAN consecutive elements of one leaf and finds
// its leaf by a linear scan of the block prefix (<= 48 entries, uniform across the workgroup).
namespace {
constexpr int MT_SPAN = 4 * ET;
struct AdamwMultiArgs {
    mfc_adamw_item it[MFC_ADAMW_MULTI_MAX];
    uint32_t first_block[MFC_ADAMW_MULTI_MAX + 1];
    int n_items;
    float gscale, lr, b1, b2, eps, wd, bc1, bc2;
};
__global__ void __launch_bounds__(ET) adamw_multi_kernel(AdamwMultiArgs a) {
    int k = 0;
    while (k + 1 < a.n_items && blockIdx.x >= a.first_block[k + 1]) ++k;
    const mfc_adamw_item& it = a.it[k];
    const int64_t base = (int64_t)(blockIdx.x - a.first_block[k]) * MT_SPAN;
    u16* pw = (u16*)it.p_bf16;
#pragma unroll
    for (int i = 0; i < MT_SPAN / ET; ++i) {
        const int64_t o = base + i * ET + threadIdx.x;
        if (o >= it.n) break;
        const float g = it.grad_dtype == MFC_F32 ? ((const float*)it.g)[o] : bf16_to_f32(((const u16*)it.g)[o]);
        float pv = it.p[o], mv = it.m[o], vv = it.v[o];
        adamw_elem(pv, mv, vv, g * a.gscale, a.lr, a.b1, a.b2, a.eps, a.wd, a.bc1, a.bc2);
        it.p[o] = pv; it.m[o] = mv; it.v[o] = vv;
        if (pw) pw[o] = f32_to_bf16(pv);
    }
}
}  // namespace

extern "C" int mfc_adamw_multi(int n_items, const mfc_adamw_item* items, float grad_scale, float lr, float b1, float b2,
                               float eps, float wd, int64_t step, void* stream) {
    if (n_items < 0 || step < 1) return MFC_EINVAL;
    if (n_items > 0 && !items) return MFC_EFAULT;
    for (int i = 0; i < n_items; ++i) {
        if (!items[i].p || !items[i].g || !items[i].m || !items[i].v) return MFC_EFAULT;
        if (items[i].n <= 0 || !DT_OK(items[i].grad_dtype) || ceil_div64(items[i].n, MT_SPAN) > (1LL << 30)) return MFC_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    for (int i0 = 0; i0 < n_items; i0 += MFC_ADAMW_MULTI_MAX) {
        AdamwMultiArgs a;
        a.n_items = n_items - i0 < MFC_ADAMW_MULTI_MAX ? n_items - i0 : MFC_ADAMW_MULTI_MAX;
        uint64_t blocks = 0;
        for (int k = 0; k < a.n_items; ++k) {
            a.it[k] = items[i0 + k];
            a.first_block[k] = (uint32_t)blocks;
            blocks += (uint64_t)ceil_div64(items[i0 + k].n, MT_SPAN);
        }
        for (int k = a.n_items; k < MFC_ADAMW_MULTI_MAX; ++k) { a.it[k] = mfc_adamw_item{}; a.first_block[k] = (uint32_t)blocks; }
        a.first_block[MFC_ADAMW_MULTI_MAX] = (uint32_t)blocks;
        if (blocks > 0x7fffffffULL) return MFC_EINVAL;
        a.gscale = grad_scale; a.lr = lr; a.b1 = b1; a.b2 = b2; a.eps = eps; a.wd = wd;
        a.bc1 = 1.0f - powf(b1, (float)step); a.bc2 = 1.0f - powf(b2, (float)step);
        hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)blocks), dim3(ET), 0, st, a);
    }
    return mfc_launch_status();
}

extern "C" int mfc_adamw(int grad_dtype, int64_t n, float* p, void* p_bf16, const void* g, float grad_scale,
                         float* m, float* v, float lr, float b1, float b2, float eps, float wd, int64_t step,
                         void* stream) {
    if (!p || !g || !m || !v) return MFC_EFAULT;
    if (n <= 0 || step < 1 || !DT_OK(grad_dtype)) return MFC_EINVAL;
    const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
    hipStream_t st = (hipStream_t)stream;
    static const int mode = getenv("MFC_ADAMW_MODE") ? atoi(getenv("MFC_ADAMW_MODE")) : 1;
    static const int64_t vblocks = getenv("MFC_ADAMW_BLOCKS") ? atoll(getenv("MFC_ADAMW_BLOCKS")) : 65536;
    const bool aligned = (((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)g | (uintptr_t)p_bf16) & 15) == 0;
    int64_t done = 0;
    if (mode > 0 && aligned && n >= 4) {
        const int64_t n4 = n / 4;
        int64_t b = ceil_div64(n4, ET);
        if (b > vblocks) b = vblocks;
#define MFC_ADAMW_LAUNCH(TG, NTF)                                                                                  \
        hipLaunchKernelGGL((adamw_vec_kernel<TG, NTF>), dim3((unsigned)b), dim3(ET), 0, st, n4, p, (u16*)p_bf16,   \
                           (const TG*)g, grad_scale, m, v, lr, b1, b2, eps, wd, bc1, bc2)
        if (grad_dtype == MFC_F32) { if (mode == 2) MFC_ADAMW_LAUNCH(float, true); else MFC_ADAMW_LAUNCH(float, false); }
        else { if (mode == 2) MFC_ADAMW_LAUNCH(u16, true); else MFC_ADAMW_LAUNCH(u16, false); }
#undef MFC_ADAMW_LAUNCH
        done = n4 * 4;
        if (done == n) return mfc_launch_status();
    }
    // scalar path: everything when unaligned, else the < 4 element tail
    const int64_t rem = n - done;
    u16* pwt = p_bf16 ? (u16*)p_bf16 + done : nullptr;
    if (grad_dtype == MFC_F32)
        hipLaunchKernelGGL(adamw_kernel<float>, dim3(grid_for(rem)), dim3(ET), 0, st, rem, p + done, pwt,
                           (const float*)g + done, grad_scale, m + done, v + done, lr, b1, b2, eps, wd, bc1, bc2);
    else
        hipLaunchKernelGGL(adamw_kernel<u16>, dim3(grid_for(rem)), dim3(ET), 0, st, rem, p + done, pwt,
                           (const u16*)g + done, grad_scale, m + done, v + done, lr, b1, b2, eps, wd, bc1, bc2);
    return mfc_launch_status();
}
