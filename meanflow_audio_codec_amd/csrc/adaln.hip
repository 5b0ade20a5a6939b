// AdaLN / gating kernels of the MLP and MLP-Mixer velocity nets (HBM-bound row kernels).
//
//   adaln   : y = (1 + scale) * LN(x) + shift        models/mlp_flow.py:96-110, models/mlp_mixer.py (AdaLN)
//   gate    : out = o * (1 + s2) * inv_k + res        models/mlp_flow.py:116-117
// each with its forward-mode tangent (row-stacked: rows >= act_rows are the tangents of rows
// [0, n_tan)) and its reverse pass.  LN = flax LayerNorm without affine, eps 1e-6, last axis.
// Modulation tensors are addressed as mod[(row / mod_div) * ldm + col], so per-row modulation
// (MLP flow: mod_div = 1, views into the conditioning MLP's output) and per-sample modulation
// broadcast over tokens (Mixer: mod_div = tokens per sample) share the kernels.
#include "mfc_common.h"
#include <initializer_list>

namespace {

constexpr int AT = 256;

__device__ inline float block_sum(float v, float* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < AT / 64; ++i) s += red[i];
    return s;
}

struct AdaArgs {
    int64_t rows, act_rows, W, ldx, ldm, ldy, mod_div;
    const void* x; const void* scale; const void* shift; void* y;
    const void* dy; void* dx; void* dscale; void* dshift; int64_t ldd;
    float inv_k;
};

// one workgroup per row (primal rows and tangent rows)
template <typename T>
__global__ void __launch_bounds__(AT) adaln_fwd_kernel(AdaArgs a) {
    __shared__ float red[AT / 64];
    const int64_t row = blockIdx.x;
    const bool tan = row >= a.act_rows;
    const int64_t prow = tan ? row - a.act_rows : row;     // the primal row this row belongs to
    const T* xp = (const T*)a.x + prow * a.ldx;
    const T* sc = (const T*)a.scale + (prow / a.mod_div) * a.ldm;
    const T* sh = (const T*)a.shift + (prow / a.mod_div) * a.ldm;
    const int64_t W = a.W;
    float s = 0.f, ss = 0.f;
    for (int64_t c = threadIdx.x; c < W; c += AT) { const float v = St<T>::ld(xp + c); s += v; ss += v * v; }
    const float mean = block_sum(s, red) / (float)W;
    const float var = fmaxf(0.f, block_sum(ss, red) / (float)W - mean * mean);
    const float rho = rsqrtf(var + 1e-6f);
    T* yp = (T*)a.y + row * a.ldy;
    if (!tan) {
        for (int64_t c = threadIdx.x; c < W; c += AT) {
            const float n = (St<T>::ld(xp + c) - mean) * rho;
            St<T>::st(yp + c, (1.0f + St<T>::ld(sc + c)) * n + St<T>::ld(sh + c));
        }
        return;
    }
    // tangent row: ydot = scdot * n + (1 + sc) * ndot + shdot, ndot = rho (xd_c - n mean(n xd_c))
    const T* xd = (const T*)a.x + row * a.ldx;
    const T* scd = (const T*)a.scale + (row / a.mod_div) * a.ldm;
    const T* shd = (const T*)a.shift + (row / a.mod_div) * a.ldm;
    float sd = 0.f;
    for (int64_t c = threadIdx.x; c < W; c += AT) sd += St<T>::ld(xd + c);
    const float md = block_sum(sd, red) / (float)W;
    float dt = 0.f;
    for (int64_t c = threadIdx.x; c < W; c += AT)
        dt += (St<T>::ld(xp + c) - mean) * rho * (St<T>::ld(xd + c) - md);
    const float dot = block_sum(dt, red) / (float)W;
    for (int64_t c = threadIdx.x; c < W; c += AT) {
        const float n = (St<T>::ld(xp + c) - mean) * rho;
        const float nd = rho * (St<T>::ld(xd + c) - md - n * dot);
        St<T>::st(yp + c, St<T>::ld(scd + c) * n + (1.0f + St<T>::ld(sc + c)) * nd + St<T>::ld(shd + c));
    }
}

// reverse: dscale = dy * n ; dshift = dy ; dx = LN-bwd(dy * (1 + scale))
// mod_div == 1: dscale/dshift written in T by this kernel.  mod_div > 1 (modulation shared by mod_div rows): they are
// fp32 [rows/mod_div, W] buffers written by adaln_bwd_mod_kernel below (fixed summation order, no atomics).
template <typename T>
__global__ void __launch_bounds__(AT) adaln_bwd_kernel(AdaArgs a) {
    __shared__ float red[AT / 64];
    const int64_t row = blockIdx.x;
    const T* xp = (const T*)a.x + row * a.ldx;
    const T* sc = (const T*)a.scale + (row / a.mod_div) * a.ldm;
    const T* dyp = (const T*)a.dy + row * a.ldy;
    const int64_t W = a.W;
    float s = 0.f, ss = 0.f;
    for (int64_t c = threadIdx.x; c < W; c += AT) { const float v = St<T>::ld(xp + c); s += v; ss += v * v; }
    const float mean = block_sum(s, red) / (float)W;
    const float var = fmaxf(0.f, block_sum(ss, red) / (float)W - mean * mean);
    const float rho = rsqrtf(var + 1e-6f);
    float m1 = 0.f, m2 = 0.f;
    for (int64_t c = threadIdx.x; c < W; c += AT) {
        const float n = (St<T>::ld(xp + c) - mean) * rho;
        const float dn = St<T>::ld(dyp + c) * (1.0f + St<T>::ld(sc + c));
        m1 += dn; m2 += dn * n;
    }
    m1 = block_sum(m1, red) / (float)W;
    m2 = block_sum(m2, red) / (float)W;
    T* dxp = (T*)a.dx + row * a.ldx;
    for (int64_t c = threadIdx.x; c < W; c += AT) {
        const float n = (St<T>::ld(xp + c) - mean) * rho;
        const float dyv = St<T>::ld(dyp + c);
        const float dn = dyv * (1.0f + St<T>::ld(sc + c));
        St<T>::st(dxp + c, rho * (dn - m1 - n * m2));
        if (a.mod_div == 1) {
            St<T>::st((T*)a.dscale + row * a.ldd + c, dyv * n);
            St<T>::st((T*)a.dshift + row * a.ldd + c, dyv);
        }
    }
}

// dscale / dshift of a modulation shared by mod_div consecutive rows: one workgroup per group.  The rows' LayerNorm
// statistics are recomputed in batches into LDS, then every thread sums its columns over the rows of the batch in
// ascending row order -- bitwise reproducible (the sums used to be fp32 atomics).
constexpr int MOD_BATCH = 1024;
template <typename T>
__global__ void __launch_bounds__(AT) adaln_bwd_mod_kernel(AdaArgs a) {
    __shared__ float st_mean[MOD_BATCH], st_rho[MOD_BATCH];
    const int64_t grp = blockIdx.x, W = a.W;
    const int64_t r0 = grp * a.mod_div, r1 = (r0 + a.mod_div < a.rows) ? r0 + a.mod_div : a.rows;
    float* ds = (float*)a.dscale + grp * a.ldd;
    float* dh = (float*)a.dshift + grp * a.ldd;
    for (int64_t b0 = r0; b0 < r1; b0 += MOD_BATCH) {
        const int nb = (int)((r1 - b0 < MOD_BATCH) ? r1 - b0 : MOD_BATCH);
        // statistics: one WAVE per row (butterfly sums, no workgroup barrier per row -- with a barrier pair per reduction
        // the encoder's 544-token groups spent 0.6 ms here)
        for (int i = threadIdx.x >> 6; i < nb; i += AT / 64) {
            const T* xp = (const T*)a.x + (b0 + i) * a.ldx;
            float s = 0.f, ss = 0.f;
            for (int64_t c = threadIdx.x & 63; c < W; c += 64) { const float v = St<T>::ld(xp + c); s += v; ss += v * v; }
            for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
            const float mean = s / (float)W;
            const float var = fmaxf(0.f, ss / (float)W - mean * mean);
            if ((threadIdx.x & 63) == 0) { st_mean[i] = mean; st_rho[i] = rsqrtf(var + 1e-6f); }
        }
        __syncthreads();
        for (int64_t c = threadIdx.x; c < W; c += AT) {
            float acc_s = (b0 == r0) ? 0.f : ds[c], acc_h = (b0 == r0) ? 0.f : dh[c];
#pragma unroll 8
            for (int i = 0; i < nb; ++i) {
                const float n = (St<T>::ld((const T*)a.x + (b0 + i) * a.ldx + c) - st_mean[i]) * st_rho[i];
                const float dyv = St<T>::ld((const T*)a.dy + (b0 + i) * a.ldy + c);
                acc_s += dyv * n;
                acc_h += dyv;
            }
            ds[c] = acc_s;
            dh[c] = acc_h;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Narrow rows (the Mixer: W = 16 channels per token, 10^5 rows; its encoder: 256 channels): a whole workgroup per 64-byte
// row leaves 255 of 256 lanes idle.  Here a row (up to 1 KB) is held by LPR = W * sizeof(T) / 16 lanes (one 16-byte piece
// each), a wave covers 64 / LPR rows per step (fully coalesced), and the row reductions are LPR-lane butterflies.
// ---------------------------------------------------------------------------------------------------------------
template <typename T> struct Piece;
template <> struct Piece<float> { static constexpr int VW = 4; typedef f32x4 vec; };
template <> struct Piece<u16> { static constexpr int VW = 8; typedef s16x8 vec; };
template <typename T> __device__ inline void ld_piece(const T* p, float v[Piece<T>::VW]) {
    typename Piece<T>::vec t = *reinterpret_cast<const typename Piece<T>::vec*>(p);
#pragma unroll
    for (int i = 0; i < Piece<T>::VW; ++i) { T e = (T)t[i]; v[i] = St<T>::ld(&e); }
}
template <typename T> __device__ inline void st_piece(T* p, const float v[Piece<T>::VW]) {
    typename Piece<T>::vec t;
#pragma unroll
    for (int i = 0; i < Piece<T>::VW; ++i) { T e; St<T>::st(&e, v[i]); t[i] = e; }
    *reinterpret_cast<typename Piece<T>::vec*>(p) = t;
}
__device__ inline float lanes_sum(float v, int lpr) {
    for (int o = 1; o < lpr; o <<= 1) v += __shfl_xor(v, o);
    return v;
}
template <typename T> __device__ inline void ln_stats(const float v[Piece<T>::VW], int lpr, float invW, float& mean, float& rho) {
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < Piece<T>::VW; ++i) { s += v[i]; ss += v[i] * v[i]; }
    s = lanes_sum(s, lpr); ss = lanes_sum(ss, lpr);
    mean = s * invW;
    rho = rsqrtf(fmaxf(0.f, ss * invW - mean * mean) + 1e-6f);
}

template <typename T>
__global__ void __launch_bounds__(AT) adaln_fwd_narrow_kernel(AdaArgs a, int lpr) {
    constexpr int VW = Piece<T>::VW;
    const int part = threadIdx.x % lpr;
    const int64_t slots = (int64_t)gridDim.x * (AT / lpr);
    const float invW = 1.0f / (float)a.W;
    for (int64_t row = blockIdx.x * (int64_t)(AT / lpr) + threadIdx.x / lpr; row < a.rows; row += slots) {
        const bool tan = row >= a.act_rows;
        const int64_t prow = tan ? row - a.act_rows : row;
        float x[VW], sc[VW], sh[VW], y[VW];
        ld_piece<T>((const T*)a.x + prow * a.ldx + part * VW, x);
        ld_piece<T>((const T*)a.scale + (prow / a.mod_div) * a.ldm + part * VW, sc);
        float mean, rho;
        ln_stats<T>(x, lpr, invW, mean, rho);
        if (!tan) {
            ld_piece<T>((const T*)a.shift + (prow / a.mod_div) * a.ldm + part * VW, sh);
#pragma unroll
            for (int i = 0; i < VW; ++i) y[i] = (1.0f + sc[i]) * ((x[i] - mean) * rho) + sh[i];
        } else {
            float xd[VW], scd[VW], shd[VW];
            ld_piece<T>((const T*)a.x + row * a.ldx + part * VW, xd);
            ld_piece<T>((const T*)a.scale + (row / a.mod_div) * a.ldm + part * VW, scd);
            ld_piece<T>((const T*)a.shift + (row / a.mod_div) * a.ldm + part * VW, shd);
            float sd = 0.f;
#pragma unroll
            for (int i = 0; i < VW; ++i) sd += xd[i];
            const float md = lanes_sum(sd, lpr) * invW;
            float dt = 0.f;
#pragma unroll
            for (int i = 0; i < VW; ++i) dt += (x[i] - mean) * rho * (xd[i] - md);
            const float dot = lanes_sum(dt, lpr) * invW;
#pragma unroll
            for (int i = 0; i < VW; ++i) {
                const float n = (x[i] - mean) * rho;
                const float nd = rho * (xd[i] - md - n * dot);
                y[i] = scd[i] * n + (1.0f + sc[i]) * nd + shd[i];
            }
        }
        st_piece<T>((T*)a.y + row * a.ldy + part * VW, y);
    }
}

template <typename T>
__global__ void __launch_bounds__(AT) adaln_bwd_narrow_kernel(AdaArgs a, int lpr) {
    constexpr int VW = Piece<T>::VW;
    const int part = threadIdx.x % lpr;
    const int64_t slots = (int64_t)gridDim.x * (AT / lpr);
    const float invW = 1.0f / (float)a.W;
    for (int64_t row = blockIdx.x * (int64_t)(AT / lpr) + threadIdx.x / lpr; row < a.rows; row += slots) {
        float x[VW], sc[VW], dy[VW], dx[VW];
        ld_piece<T>((const T*)a.x + row * a.ldx + part * VW, x);
        ld_piece<T>((const T*)a.scale + (row / a.mod_div) * a.ldm + part * VW, sc);
        ld_piece<T>((const T*)a.dy + row * a.ldy + part * VW, dy);
        float mean, rho;
        ln_stats<T>(x, lpr, invW, mean, rho);
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int i = 0; i < VW; ++i) {
            const float n = (x[i] - mean) * rho, dn = dy[i] * (1.0f + sc[i]);
            m1 += dn; m2 += dn * n;
        }
        m1 = lanes_sum(m1, lpr) * invW;
        m2 = lanes_sum(m2, lpr) * invW;
        float ds[VW];
#pragma unroll
        for (int i = 0; i < VW; ++i) {
            const float n = (x[i] - mean) * rho, dn = dy[i] * (1.0f + sc[i]);
            dx[i] = rho * (dn - m1 - n * m2);
            ds[i] = dy[i] * n;
        }
        st_piece<T>((T*)a.dx + row * a.ldx + part * VW, dx);
        if (a.mod_div == 1) {
            st_piece<T>((T*)a.dscale + row * a.ldd + part * VW, ds);
            st_piece<T>((T*)a.dshift + row * a.ldd + part * VW, dy);
        }
    }
}

// dscale / dshift of a modulation shared by mod_div rows, narrow rows: one workgroup per group, its AT / LPR row slots
// stride the group's rows (ascending), and the slots' partial sums are added in slot order -- bitwise reproducible.
template <typename T>
__global__ void __launch_bounds__(AT) adaln_bwd_mod_narrow_kernel(AdaArgs a, int lpr) {
    constexpr int VW = Piece<T>::VW;
    __shared__ float red[2][AT * VW];
    const int part = threadIdx.x % lpr, slot = threadIdx.x / lpr, nslot = AT / lpr;
    const int64_t grp = blockIdx.x;
    const int64_t r0 = grp * a.mod_div, r1 = (r0 + a.mod_div < a.rows) ? r0 + a.mod_div : a.rows;
    const float invW = 1.0f / (float)a.W;
    float as[VW], ah[VW];
#pragma unroll
    for (int i = 0; i < VW; ++i) { as[i] = 0.f; ah[i] = 0.f; }
    // four independent rows per step: eight 16-byte loads in flight per lane (one row at a time, a group of 544 rows x 1 KB
    // is a chain of 136 HBM round trips per wave); rows past the group load zeros and add nothing
    constexpr int RU = 4;
    for (int64_t row = r0 + slot; row < r1; row += RU * nslot) {
        float x[RU][VW], dy[RU][VW];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int64_t rr = row + (int64_t)u * nslot;
            if (rr < r1) {
                ld_piece<T>((const T*)a.x + rr * a.ldx + part * VW, x[u]);
                ld_piece<T>((const T*)a.dy + rr * a.ldy + part * VW, dy[u]);
            } else {
#pragma unroll
                for (int i = 0; i < VW; ++i) { x[u][i] = 0.f; dy[u][i] = 0.f; }
            }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            float mean, rho;
            ln_stats<T>(x[u], lpr, invW, mean, rho);
#pragma unroll
            for (int i = 0; i < VW; ++i) { as[i] += dy[u][i] * ((x[u][i] - mean) * rho); ah[i] += dy[u][i]; }
        }
    }
#pragma unroll
    for (int i = 0; i < VW; ++i) { red[0][threadIdx.x * VW + i] = as[i]; red[1][threadIdx.x * VW + i] = ah[i]; }
    __syncthreads();
    for (int o = threadIdx.x; o < 2 * (int)a.W; o += AT) {
        const int which = o >= a.W, c = o - which * (int)a.W;
        const int pc = c / VW, ic = c % VW;
        float sum = 0.f;
        for (int k = 0; k < nslot; ++k) sum += red[which][(k * lpr + pc) * VW + ic];
        ((float*)(which ? a.dshift : a.dscale))[grp * a.ldd + c] = sum;
    }
}

inline bool narrow_ok(int dtype, int64_t W, std::initializer_list<int64_t> lds, std::initializer_list<const void*> ptrs, int& lpr) {
    const int64_t es = dtype == MFC_F32 ? 4 : 2;
    const int64_t bytes = W * es;
    if (bytes % 16 || bytes > 1024) return false;     // a row is held by at most one wave (64 lanes x 16 bytes)
    lpr = (int)(bytes / 16);
    if (lpr & (lpr - 1)) return false;
    for (int64_t ld : lds) if ((ld * es) % 16) return false;
    for (const void* p : ptrs) if ((uintptr_t)p % 16) return false;
    return true;
}

struct GateArgs {
    int64_t rows, act_rows, W, ldo, ldm, ldr, ldy;
    const void* o; const void* s2; const void* res; void* y;
    const void* dy; void* dox; void* ds2; int64_t ldd;
    float inv_k;
};

// out = o (1 + s2) inv_k + res ; tangent rows: (odot (1 + s2) + o s2dot) inv_k + resdot
template <typename T>
__global__ void __launch_bounds__(AT) gate_fwd_kernel(GateArgs a) {
    const int64_t total = a.rows * a.W;
    for (int64_t i = blockIdx.x * (int64_t)AT + threadIdx.x; i < total; i += (int64_t)gridDim.x * AT) {
        const int64_t row = i / a.W, c = i - row * a.W;
        const float o = St<T>::ld((const T*)a.o + row * a.ldo + c);
        const float s2 = St<T>::ld((const T*)a.s2 + row * a.ldm + c);
        const float r = St<T>::ld((const T*)a.res + row * a.ldr + c);
        float v;
        if (row < a.act_rows) v = o * (1.0f + s2) * a.inv_k + r;
        else {
            const int64_t pr = row - a.act_rows;
            const float op = St<T>::ld((const T*)a.o + pr * a.ldo + c);
            const float sp = St<T>::ld((const T*)a.s2 + pr * a.ldm + c);
            v = (o * (1.0f + sp) + op * s2) * a.inv_k + r;
        }
        St<T>::st((T*)a.y + row * a.ldy + c, v);
    }
}

// do = dy (1 + s2) inv_k ; ds2 = dy o inv_k   (dres = dy, taken by the caller)
template <typename T>
__global__ void __launch_bounds__(AT) gate_bwd_kernel(GateArgs a) {
    const int64_t total = a.rows * a.W;
    for (int64_t i = blockIdx.x * (int64_t)AT + threadIdx.x; i < total; i += (int64_t)gridDim.x * AT) {
        const int64_t row = i / a.W, c = i - row * a.W;
        const float dy = St<T>::ld((const T*)a.dy + row * a.ldy + c);
        const float o = St<T>::ld((const T*)a.o + row * a.ldo + c);
        const float s2 = St<T>::ld((const T*)a.s2 + row * a.ldm + c);
        St<T>::st((T*)a.dox + row * a.ldo + c, dy * (1.0f + s2) * a.inv_k);
        St<T>::st((T*)a.ds2 + row * a.ldd + c, dy * o * a.inv_k);
    }
}

// strided 2-D copy / accumulate: dst[r, c] (+)= alpha * src[r, c]
template <typename T>
__global__ void __launch_bounds__(AT) copy2d_kernel(int64_t rows, int64_t W, const T* src, int64_t lds_, T* dst,
                                                    int64_t ldd, float alpha, int accum) {
    const int64_t total = rows * W;
    for (int64_t i = blockIdx.x * (int64_t)AT + threadIdx.x; i < total; i += (int64_t)gridDim.x * AT) {
        const int64_t row = i / W, c = i - row * W;
        float v = alpha * St<T>::ld(src + row * lds_ + c);
        if (accum) v += St<T>::ld(dst + row * ldd + c);
        St<T>::st(dst + row * ldd + c, v);
    }
}

// batched 2-D transpose: src [batch, rows, cols] -> dst [batch, cols, rows] (32x32 LDS tiles)
template <typename T>
__global__ void __launch_bounds__(256) transpose_kernel(int64_t batch, int rows, int cols, const T* src, T* dst,
                                                        float alpha, const T* add) {
    __shared__ float tile[32][33];
    const int tilesR = (rows + 31) / 32, tilesC = (cols + 31) / 32;
    const int64_t per = (int64_t)tilesR * tilesC, total = batch * per;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int64_t t = blockIdx.x; t < total; t += gridDim.x) {
        const int64_t b = t / per;
        const int tt = (int)(t - b * per), tr = tt / tilesC, tc = tt - tr * tilesC;
        const T* sp = src + b * (int64_t)rows * cols;
        T* dp = dst + b * (int64_t)rows * cols;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = tr * 32 + ty + 8 * k, c = tc * 32 + tx;
            tile[ty + 8 * k][tx] = (r < rows && c < cols) ? St<T>::ld(sp + (int64_t)r * cols + c) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = tc * 32 + ty + 8 * k, r = tr * 32 + tx;   // dst[c][r]
            if (r < rows && c < cols) {
                float v = alpha * tile[tx][ty + 8 * k];
                const int64_t off = b * (int64_t)rows * cols + (int64_t)c * rows + r;
                if (add) v += St<T>::ld(add + off);
                St<T>::st(dp + (int64_t)c * rows + r, v);
            }
        }
    }
}

inline unsigned grid1d(int64_t n) {
    int64_t b = ceil_div64(n, AT);
    return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

#define DT_OK(dt) ((dt) == MFC_F32 || (dt) == MFC_BF16)

extern "C" int mfc_adaln_fwd(int dtype, int64_t rows, int64_t act_rows, int64_t W, const void* x, int64_t ldx,
                             const void* scale, const void* shift, int64_t ldm, int64_t mod_div, void* y,
                             int64_t ldy, void* stream) {
    if (!x || !scale || !shift || !y) return MFC_EFAULT;
    if (rows <= 0 || W <= 0 || act_rows <= 0 || act_rows > rows || rows > 2 * act_rows || ldx < W || ldy < W ||
        mod_div < 1 || !DT_OK(dtype))
        return MFC_EINVAL;
    AdaArgs a = {};
    a.rows = rows; a.act_rows = act_rows; a.W = W; a.ldx = ldx; a.ldm = ldm; a.ldy = ldy; a.mod_div = mod_div;
    a.x = x; a.scale = scale; a.shift = shift; a.y = y;
    hipStream_t st = (hipStream_t)stream;
    int lpr = 0;
    if (rows >= 1024 && narrow_ok(dtype, W, {ldx, ldm, ldy}, {x, scale, shift, y}, lpr)) {
        const unsigned grid = grid1d(rows * lpr);
        if (dtype == MFC_F32) hipLaunchKernelGGL(adaln_fwd_narrow_kernel<float>, dim3(grid), dim3(AT), 0, st, a, lpr);
        else hipLaunchKernelGGL(adaln_fwd_narrow_kernel<u16>, dim3(grid), dim3(AT), 0, st, a, lpr);
        return mfc_launch_status();
    }
    if (dtype == MFC_F32) hipLaunchKernelGGL(adaln_fwd_kernel<float>, dim3((unsigned)rows), dim3(AT), 0, st, a);
    else hipLaunchKernelGGL(adaln_fwd_kernel<u16>, dim3((unsigned)rows), dim3(AT), 0, st, a);
    return mfc_launch_status();
}

extern "C" int mfc_adaln_bwd(int dtype, int64_t rows, int64_t W, const void* x, int64_t ldx, const void* scale,
                             int64_t ldm, int64_t mod_div, const void* dy, int64_t ldy, void* dx, void* dscale,
                             void* dshift, int64_t ldd, void* stream) {
    if (!x || !scale || !dy || !dx || !dscale || !dshift) return MFC_EFAULT;
    if (rows <= 0 || W <= 0 || ldx < W || ldy < W || ldd < W || mod_div < 1 || !DT_OK(dtype)) return MFC_EINVAL;
    AdaArgs a = {};
    a.rows = rows; a.W = W; a.ldx = ldx; a.ldm = ldm; a.ldy = ldy; a.mod_div = mod_div; a.ldd = ldd;
    a.x = x; a.scale = scale; a.dy = dy; a.dx = dx; a.dscale = dscale; a.dshift = dshift;
    hipStream_t st = (hipStream_t)stream;
    int lpr = 0;
    // (mod_div == 1: dscale / dshift are [rows, W] in `dtype`; mod_div > 1: fp32 [groups, W] -- 16-byte pieces of fp32 need W % 4)
    if (rows >= 1024 && narrow_ok(dtype, W, {ldx, ldm, ldy, mod_div == 1 ? ldd : 0}, {x, scale, dy, dx, mod_div == 1 ? dscale : nullptr,
                                  mod_div == 1 ? dshift : nullptr}, lpr)) {
        const unsigned grid = grid1d(rows * lpr);
        if (dtype == MFC_F32) hipLaunchKernelGGL(adaln_bwd_narrow_kernel<float>, dim3(grid), dim3(AT), 0, st, a, lpr);
        else hipLaunchKernelGGL(adaln_bwd_narrow_kernel<u16>, dim3(grid), dim3(AT), 0, st, a, lpr);
        if (mod_div > 1) {
            const unsigned groups = (unsigned)((rows + mod_div - 1) / mod_div);
            if (dtype == MFC_F32) hipLaunchKernelGGL(adaln_bwd_mod_narrow_kernel<float>, dim3(groups), dim3(AT), 0, st, a, lpr);
            else hipLaunchKernelGGL(adaln_bwd_mod_narrow_kernel<u16>, dim3(groups), dim3(AT), 0, st, a, lpr);
        }
        return mfc_launch_status();
    }
    if (dtype == MFC_F32) hipLaunchKernelGGL(adaln_bwd_kernel<float>, dim3((unsigned)rows), dim3(AT), 0, st, a);
    else hipLaunchKernelGGL(adaln_bwd_kernel<u16>, dim3((unsigned)rows), dim3(AT), 0, st, a);
    if (mod_div > 1) {
        const unsigned groups = (unsigned)((rows + mod_div - 1) / mod_div);
        if (dtype == MFC_F32) hipLaunchKernelGGL(adaln_bwd_mod_kernel<float>, dim3(groups), dim3(AT), 0, st, a);
        else hipLaunchKernelGGL(adaln_bwd_mod_kernel<u16>, dim3(groups), dim3(AT), 0, st, a);
    }
    return mfc_launch_status();
}

extern "C" int mfc_gate_fwd(int dtype, int64_t rows, int64_t act_rows, int64_t W, const void* o, int64_t ldo,
                            const void* s2, int64_t ldm, const void* res, int64_t ldr, float inv_k, void* y,
                            int64_t ldy, void* stream) {
    if (!o || !s2 || !res || !y) return MFC_EFAULT;
    if (rows <= 0 || W <= 0 || act_rows <= 0 || act_rows > rows || rows > 2 * act_rows || !DT_OK(dtype))
        return MFC_EINVAL;
    GateArgs a = {};
    a.rows = rows; a.act_rows = act_rows; a.W = W; a.ldo = ldo; a.ldm = ldm; a.ldr = ldr; a.ldy = ldy;
    a.o = o; a.s2 = s2; a.res = res; a.y = y; a.inv_k = inv_k;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MFC_F32) hipLaunchKernelGGL(gate_fwd_kernel<float>, dim3(grid1d(rows * W)), dim3(AT), 0, st, a);
    else hipLaunchKernelGGL(gate_fwd_kernel<u16>, dim3(grid1d(rows * W)), dim3(AT), 0, st, a);
    return mfc_launch_status();
}

extern "C" int mfc_gate_bwd(int dtype, int64_t rows, int64_t W, const void* dy, int64_t ldy, const void* o,
                            int64_t ldo, const void* s2, int64_t ldm, float inv_k, void* dout_o, void* ds2,
                            int64_t ldd, void* stream) {
    if (!dy || !o || !s2 || !dout_o || !ds2) return MFC_EFAULT;
    if (rows <= 0 || W <= 0 || !DT_OK(dtype)) return MFC_EINVAL;
    GateArgs a = {};
    a.rows = rows; a.W = W; a.ldo = ldo; a.ldm = ldm; a.ldy = ldy; a.ldd = ldd;
    a.o = o; a.s2 = s2; a.dy = dy; a.dox = dout_o; a.ds2 = ds2; a.inv_k = inv_k;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MFC_F32) hipLaunchKernelGGL(gate_bwd_kernel<float>, dim3(grid1d(rows * W)), dim3(AT), 0, st, a);
    else hipLaunchKernelGGL(gate_bwd_kernel<u16>, dim3(grid1d(rows * W)), dim3(AT), 0, st, a);
    return mfc_launch_status();
}

extern "C" int mfc_copy2d(int dtype, int64_t rows, int64_t W, const void* src, int64_t lds, void* dst, int64_t ldd,
                          float alpha, int accumulate, void* stream) {
    if (!src || !dst) return MFC_EFAULT;
    if (rows <= 0 || W <= 0 || lds < W || ldd < W || !DT_OK(dtype)) return MFC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MFC_F32)
        hipLaunchKernelGGL(copy2d_kernel<float>, dim3(grid1d(rows * W)), dim3(AT), 0, st, rows, W, (const float*)src,
                           lds, (float*)dst, ldd, alpha, accumulate);
    else
        hipLaunchKernelGGL(copy2d_kernel<u16>, dim3(grid1d(rows * W)), dim3(AT), 0, st, rows, W, (const u16*)src, lds,
                           (u16*)dst, ldd, alpha, accumulate);
    return mfc_launch_status();
}

extern "C" int mfc_transpose(int dtype, int64_t batch, int rows, int cols, const void* src, void* dst, float alpha,
                             const void* add, void* stream) {
    if (!src || !dst) return MFC_EFAULT;
    if (batch <= 0 || rows <= 0 || cols <= 0 || !DT_OK(dtype)) return MFC_EINVAL;
    const int64_t tiles = batch * (int64_t)((rows + 31) / 32) * ((cols + 31) / 32);
    const unsigned grid = (unsigned)(tiles > 16384 ? 16384 : tiles);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MFC_F32)
        hipLaunchKernelGGL(transpose_kernel<float>, dim3(grid), dim3(256), 0, st, batch, rows, cols, (const float*)src,
                           (float*)dst, alpha, (const float*)add);
    else
        hipLaunchKernelGGL(transpose_kernel<u16>, dim3(grid), dim3(256), 0, st, batch, rows, cols, (const u16*)src,
                           (u16*)dst, alpha, (const u16*)add);
    return mfc_launch_status();
}
