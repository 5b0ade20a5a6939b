// ConvNeXt block interior (models/conv_flow.py:65-115,162-186) for gfx950.
//
//   h1 = LN_C(h0); h2 = (1+scale) h1 + shift            conv_flow.py:181-186
//   c1 = Conv3x3_SAME(h2); n1 = LN_C(c1)                conv_flow.py:74-84
//   e1 = Conv1x1(n1) [16->32]; g1 = gelu(e1)            conv_flow.py:87-88
//   y  = GRN(g1) = g1 (gamma + q) + beta                conv_flow.py:22-45
//   o  = Conv1x1(y) [32->16] * layer_scale + h2         conv_flow.py:95-115
//
// Data layout: NHWC maps [R, s, s, 16] in the storage dtype T (fp32 or bf16).
// One workgroup = one 16x16 pixel tile (+1 halo) of one row r, 4 waves, each
// wave owns 4 tile rows of 16 pixels = one MFMA M-tile.  Every per-pixel
// contraction is an MFMA "K16 step" (mfc_common.h): the 3x3 conv is 9 steps
// (one per tap, K = 16 input channels read straight out of the LDS halo
// tile), the 1x1 convs 1-2 steps.  MFMA results (C layout: channel on the
// lane, 4 pixels in registers) are transposed through a small wave-private LDS
// scratch into the A layout (pixel on the lane, 4 channels in registers),
// where channel reductions are 2 cross-lane steps and global I/O is a
// contiguous 1 KiB per wave instruction.  Weight gradients contract over
// pixels, for which C-layout registers ARE the MFMA operands (no movement).
// The 32-channel intermediates never touch HBM; GRN's global statistic makes
// the chain run twice (stats pass / apply pass), forward and backward.
//
// Workgroups are persistent over a contiguous range of tiles so weight-gradient
// and per-row statistics accumulate in registers and are flushed with a few
// atomics per workgroup (not per tile).
#include "mfc_common.h"

namespace {

constexpr int TW = 16, TH = 16, HW = TW + 2, HH = TH + 2, NHALO = HW * HH;
constexpr int NT = 256, NWAVES = 4, RPW = TH / NWAVES;
constexpr int CS = 20;  // element stride of a 16-channel pixel row in LDS
constexpr int ES = 36;  // float stride of a 32-channel pixel row in LDS scratch
constexpr float LN_EPS = 1e-6f;
constexpr float GRN_EPS = 1e-6f;

struct Dev {  // device pointers of mfc_cnx_params, by value
    const void* conv_w; const float* conv_b; const void* exp_w; const float* exp_b;
    const float* gamma; const float* beta; const void* con_w; const float* con_b; const float* ls;
};
struct DevG {
    float* conv_w; float* conv_b; float* exp_w; float* exp_b; float* gamma; float* beta;
    float* con_w; float* con_b; float* ls;
};

__device__ inline void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ inline float red_q(float v) {  // sum over the 4 lanes sharing (lane & 15)
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
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
    *reinterpret_cast<s16x4*>(p) = s16x4{(short)f32_to_bf16(v[0]), (short)f32_to_bf16(v[1]),
                                         (short)f32_to_bf16(v[2]), (short)f32_to_bf16(v[3])};
}
template <typename T> __device__ inline void ld16(const T* p, float v[16]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ld4(p + 4 * i, v + 4 * i);
}
template <typename T> __device__ inline void st16_lds(T* p, const float v[16]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) st4(p + 4 * i, v + 4 * i);
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
    s = red_q(s); ss = red_q(ss);
    mean = s * (1.0f / 16.0f);
    const float var = fmaxf(0.0f, ss * (1.0f / 16.0f) - mean * mean);
    rho = rsqrtf(var + LN_EPS);
#pragma unroll
    for (int i = 0; i < 4; ++i) n[i] = (v[i] - mean) * rho;
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
    const float m1 = red_q(dn[0] + dn[1] + dn[2] + dn[3]) * (1.0f / 16.0f);
    const float m2 = red_q(dn[0] * n[0] + dn[1] * n[1] + dn[2] * n[2] + dn[3] * n[3]) * (1.0f / 16.0f);
#pragma unroll
    for (int i = 0; i < 4; ++i) dx[i] = rho * (dn[i] - m1 - n[i] * m2);
}

struct Geo {
    int64_t R; int s; int tilesX, tilesY; int64_t tilesPerImg, total, chunk;
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
    return g;
}

// ---- LDS carve --------------------------------------------------------------
template <typename T> struct Lds {
    T* h2s;      // [NHALO][CS]  h2 halo tile (zero outside the image)
    T* aux;      // [NHALO][CS]  tangent halo (fwd JVP) or dc1 halo (bwd conv)
    float* x16a; // per wave [16][CS]
    float* x16b; // per wave [16][CS]
    float* x32a; // per wave [16][ES]
    float* x32b; // per wave [16][ES]
    float* st;   // per wave [2][16]   mu1, rho1 of the current tile row
    float* fsc;  // [4][16] scale, shift, scaledot, shiftdot of the current row r
};
template <typename T>
__host__ __device__ inline size_t lds_bytes(bool aux) {
    size_t b = (size_t)NHALO * CS * sizeof(T) * (aux ? 2 : 1);
    b = (b + 15) & ~(size_t)15;
    b += NWAVES * (2 * 16 * CS + 2 * 16 * ES + 32) * sizeof(float) + 64 * sizeof(float);
    return b;
}
template <typename T>
__device__ inline Lds<T> carve(unsigned char* base, bool aux, int wave) {
    Lds<T> l;
    l.h2s = (T*)base;
    l.aux = l.h2s + NHALO * CS;
    size_t b = (size_t)NHALO * CS * sizeof(T) * (aux ? 2 : 1);
    b = (b + 15) & ~(size_t)15;
    float* f = (float*)(base + b);
    float* w = f + wave * (2 * 16 * CS + 2 * 16 * ES + 32);
    l.x16a = w; l.x16b = w + 16 * CS; l.x32a = w + 2 * 16 * CS; l.x32b = l.x32a + 16 * ES;
    l.st = l.x32b + 16 * ES;
    l.fsc = f + NWAVES * (2 * 16 * CS + 2 * 16 * ES + 32);
    return l;
}

// Stage the (TH+2)x(TW+2) halo of h2 = FiLM(LN(h0)) (and its tangent) into LDS.
template <typename T, bool JVP>
__device__ inline void stage_h2(const Lds<T>& l, const T* h0, const T* h0d, int64_t r, int s, int y0, int x0) {
    for (int hp = threadIdx.x; hp < NHALO; hp += NT) {
        const int hy = hp / HW, hx = hp - hy * HW;
        const int gy = y0 + hy - 1, gx = x0 + hx - 1;
        float h2[16], h2d[16];
        if (gy >= 0 && gy < s && gx >= 0 && gx < s) {
            const int64_t off = ((r * s + gy) * (int64_t)s + gx) * 16;
            float v[16];
            ld16<T>(h0 + off, v);
            float sum = 0.f, sq = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) { sum += v[c]; sq += v[c] * v[c]; }
            const float mean = sum * (1.0f / 16.0f);
            const float rho = rsqrtf(fmaxf(0.0f, sq * (1.0f / 16.0f) - mean * mean) + LN_EPS);
            float h1[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                h1[c] = (v[c] - mean) * rho;
                h2[c] = (1.0f + l.fsc[c]) * h1[c] + l.fsc[16 + c];
            }
            if constexpr (JVP) {
                float vd[16];
                ld16<T>(h0d + off, vd);
                float sd = 0.f;
#pragma unroll
                for (int c = 0; c < 16; ++c) sd += vd[c];
                const float md = sd * (1.0f / 16.0f);
                float dot = 0.f;
#pragma unroll
                for (int c = 0; c < 16; ++c) { vd[c] -= md; dot += h1[c] * vd[c]; }
                dot *= (1.0f / 16.0f);
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const float h1d = rho * (vd[c] - h1[c] * dot);
                    h2d[c] = l.fsc[32 + c] * h1[c] + (1.0f + l.fsc[c]) * h1d + l.fsc[48 + c];
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < 16; ++c) { h2[c] = 0.f; h2d[c] = 0.f; }
        }
        st16_lds<T>(l.h2s + hp * CS, h2);
        if constexpr (JVP) st16_lds<T>(l.aux + hp * CS, h2d);
    }
}

// Per-lane weights / constants of the forward chain.
template <typename T> struct FwdW {
    typedef typename Frag<T>::type frag_t;
    frag_t wc[9], we[2], wp[2];
    float bc, be[2], bp, lsn, gam[2], bet[2];
    __device__ inline void load(const Dev& d, int q, int n) {
        const T* cw = (const T*)d.conv_w;
#pragma unroll
        for (int t = 0; t < 9; ++t) wc[t] = load_bfrag<T>(cw + t * 256, 16, 1, 0, 0, q, n);
        const T* ew = (const T*)d.exp_w;  // [16][32]
        we[0] = load_bfrag<T>(ew, 32, 1, 0, 0, q, n);
        we[1] = load_bfrag<T>(ew, 32, 1, 0, 16, q, n);
        const T* pw = (const T*)d.con_w;  // [32][16]
        wp[0] = load_bfrag<T>(pw, 16, 1, 0, 0, q, n);
        wp[1] = load_bfrag<T>(pw, 16, 1, 16, 0, q, n);
        bc = d.conv_b[n]; be[0] = d.exp_b[n]; be[1] = d.exp_b[16 + n]; bp = d.con_b[n]; lsn = d.ls[n];
        gam[0] = d.gamma[n]; gam[1] = d.gamma[16 + n]; bet[0] = d.beta[n]; bet[1] = d.beta[16 + n];
    }
};

// Result of the chain up to gelu for one tile row (16 pixels).
template <typename T, bool JVP> struct RowFwd {
    typedef typename Frag<T>::type frag_t;
    f32x4 c1;            // conv output (C layout)
    float n1[4];         // LN(c1), A layout (pixel = lane&15, channels 4q..4q+3)
    float rho1;
    frag_t n1f, n1df;
    f32x4 e[2], ed[2], g[2], gd[2];
};

template <typename T, bool JVP>
__device__ inline void chain_row(const Lds<T>& l, const FwdW<T>& w, int y, int q, int m, RowFwd<T, JVP>& o) {
    typedef typename Frag<T>::type frag_t;
    f32x4 acc = f32x4{w.bc, w.bc, w.bc, w.bc};
    f32x4 accd = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int off = ((y + dy) * HW + (m + dx)) * CS + 4 * q;
            const frag_t a = *reinterpret_cast<const frag_t*>(l.h2s + off);
            mma16(acc, a, w.wc[dy * 3 + dx]);
            if constexpr (JVP) {
                const frag_t ad = *reinterpret_cast<const frag_t*>(l.aux + off);
                mma16(accd, ad, w.wc[dy * 3 + dx]);
            }
        }
    o.c1 = acc;
    // C layout -> scratch [pixel][channel]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        l.x16a[(4 * q + e) * CS + m] = acc[e];
        if constexpr (JVP) l.x16b[(4 * q + e) * CS + m] = accd[e];
    }
    lds_fence();
    float v[4], mean;
    ld4(l.x16a + m * CS + 4 * q, v);
    ln_fwd_a(v, o.n1, mean, o.rho1);
    make_frag(o.n1f, o.n1[0], o.n1[1], o.n1[2], o.n1[3]);
    if (q == 0) { l.st[m] = mean; l.st[16 + m] = o.rho1; }
    if constexpr (JVP) {
        float vd[4], nd[4];
        ld4(l.x16b + m * CS + 4 * q, vd);
        ln_jvp_a(vd, o.n1, o.rho1, nd);
        make_frag(o.n1df, nd[0], nd[1], nd[2], nd[3]);
    }
    lds_fence();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        f32x4 e = f32x4{w.be[j], w.be[j], w.be[j], w.be[j]};
        mma16(e, o.n1f, w.we[j]);
        o.e[j] = e;
#pragma unroll
        for (int i = 0; i < 4; ++i) o.g[j][i] = gelu_f(e[i]);
        if constexpr (JVP) {
            f32x4 ed = f32x4{0.f, 0.f, 0.f, 0.f};
            mma16(ed, o.n1df, w.we[j]);
            o.ed[j] = ed;
#pragma unroll
            for (int i = 0; i < 4; ++i) o.gd[j][i] = ed[i] * gelu_grad_f(e[i]);
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
};

template <typename T, bool JVP>
__device__ inline void load_film(const Lds<T>& l, const FwdArgs& a, int64_t r) {
    if (threadIdx.x < 16) {
        l.fsc[threadIdx.x] = a.sc[r * 16 + threadIdx.x];
        l.fsc[16 + threadIdx.x] = a.sh[r * 16 + threadIdx.x];
        if constexpr (JVP) {
            l.fsc[32 + threadIdx.x] = a.scd[r * 16 + threadIdx.x];
            l.fsc[48 + threadIdx.x] = a.shd[r * 16 + threadIdx.x];
        }
    }
}

// MODE 0: GRN statistics; MODE 1: apply GRN, contract, layer-scale, residual.
template <typename T, bool JVP, int MODE>
__global__ void __launch_bounds__(NT)
cnx_fwd_kernel(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename Frag<T>::type frag_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, m = lane & 15;
    Lds<T> l = carve<T>(smem, JVP, wave);
    FwdW<T> w;
    w.load(a.p, q, m);
    const int s = a.geo.s;
    const T* h0 = (const T*)a.h0;
    const T* h0d = (const T*)a.h0d;

    int64_t t0 = (int64_t)blockIdx.x * a.geo.chunk;
    int64_t t1 = t0 + a.geo.chunk < a.geo.total ? t0 + a.geo.chunk : a.geo.total;
    int64_t rcur = -1;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    float qv[2] = {0.f, 0.f}, qdv[2] = {0.f, 0.f};

    auto flush_stats = [&](int64_t r) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float v1 = red_q(s1[j]);
                if (q == 0) atomicAdd(a.S1 + r * 32 + 16 * j + m, v1);
                if constexpr (JVP) {
                    const float v2 = red_q(s2[j]);
                    if (q == 0) atomicAdd(a.S2 + r * 32 + 16 * j + m, v2);
                }
                s1[j] = 0.f; s2[j] = 0.f;
            }
        }
    };

    for (int64_t t = t0; t < t1; ++t) {
        const int64_t r = t / a.geo.tilesPerImg;
        const int ti = (int)(t - r * a.geo.tilesPerImg);
        const int y0 = (ti / a.geo.tilesX) * TH, x0 = (ti % a.geo.tilesX) * TW;
        __syncthreads();  // previous tile fully consumed
        if (r != rcur) {
            if (rcur >= 0) flush_stats(rcur);
            rcur = r;
            load_film<T, JVP>(l, a, r);
            if constexpr (MODE == 1) {
                qv[0] = a.q[r * 32 + m]; qv[1] = a.q[r * 32 + 16 + m];
                if constexpr (JVP) { qdv[0] = a.qd[r * 32 + m]; qdv[1] = a.qd[r * 32 + 16 + m]; }
            }
            __syncthreads();
        }
        stage_h2<T, JVP>(l, h0, h0d, r, s, y0, x0);
        __syncthreads();
#pragma unroll 1
        for (int ri = 0; ri < RPW; ++ri) {
            const int y = wave * RPW + ri;
            const int gy = y0 + y;
            RowFwd<T, JVP> f;
            chain_row<T, JVP>(l, w, y, q, m, f);
            if constexpr (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool ok = gy < s && (x0 + 4 * q + e) < s;
                    if (ok) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            s1[j] += f.g[j][e] * f.g[j][e];
                            if constexpr (JVP) s2[j] += f.g[j][e] * f.gd[j][e];
                        }
                    }
                }
            } else {
                // GRN apply (C layout, channel constants on the lane)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float yv = f.g[j][e] * (w.gam[j] + qv[j]) + w.bet[j];
                        l.x32a[(4 * q + e) * ES + 16 * j + m] = yv;
                        if constexpr (JVP) {
                            const float yd = f.gd[j][e] * (w.gam[j] + qv[j]) + f.g[j][e] * qdv[j];
                            l.x32b[(4 * q + e) * ES + 16 * j + m] = yd;
                        }
                    }
                lds_fence();
                f32x4 p1 = f32x4{w.bp, w.bp, w.bp, w.bp}, p1d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    float yv[4];
                    ld4(l.x32a + m * ES + 16 * c + 4 * q, yv);
                    frag_t yf;
                    make_frag(yf, yv[0], yv[1], yv[2], yv[3]);
                    mma16(p1, yf, w.wp[c]);
                    if constexpr (JVP) {
                        ld4(l.x32b + m * ES + 16 * c + 4 * q, yv);
                        make_frag(yf, yv[0], yv[1], yv[2], yv[3]);
                        mma16(p1d, yf, w.wp[c]);
                    }
                }
                // layer scale + residual (h2 from the halo tile centre), back to A layout
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int hoff = ((y + 1) * HW + (4 * q + e + 1)) * CS + m;
                    l.x16a[(4 * q + e) * CS + m] = p1[e] * w.lsn + St<T>::ld(l.h2s + hoff);
                    if constexpr (JVP) l.x16b[(4 * q + e) * CS + m] = p1d[e] * w.lsn + St<T>::ld(l.aux + hoff);
                }
                lds_fence();
                const int gx = x0 + m;
                if (gy < s && gx < s) {
                    const int64_t off = ((r * s + gy) * (int64_t)s + gx) * 16 + 4 * q;
                    float ov[4];
                    ld4(l.x16a + m * CS + 4 * q, ov);
                    st4((T*)a.o + off, ov);
                    if constexpr (JVP) {
                        ld4(l.x16b + m * CS + 4 * q, ov);
                        st4((T*)a.od + off, ov);
                    }
                }
                lds_fence();
            }
        }
    }
    if (rcur >= 0) flush_stats(rcur);
}

// ---------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------
struct BwdArgs {
    Geo geo;
    const void* h0; const float* sc; const float* sh;
    Dev p; DevG g;
    const float* q; const float* kG;
    const void* dout; const void* dc1_in;
    float* dq; void* dc1; void* dh0; float* dsc; float* dsh;
};

// MODE 0: dq[r,ch] = sum dy*g1, dbeta += sum dy.
// MODE 1: dc1 + small-parameter gradients (con_w, con_b, ls, exp_w, exp_b, conv_b).
template <typename T, int MODE>
__global__ void __launch_bounds__(NT)
cnx_bwd_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename Frag<T>::type frag_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, m = lane & 15;
    Lds<T> l = carve<T>(smem, false, wave);
    FwdW<T> w;
    w.load(a.p, q, m);
    // transposed 1x1 weights as B operands
    const T* pw = (const T*)a.p.con_w;  // [32][16]: dy[e] = sum_c dp1[c] Wp[e][c]
    frag_t wpT[2] = {load_bfrag<T>(pw, 1, 16, 0, 0, q, m), load_bfrag<T>(pw, 1, 16, 0, 16, q, m)};
    const T* ew = (const T*)a.p.exp_w;  // [16][32]: dn1[c] = sum_e de[e] We[c][e]
    frag_t weT[2] = {load_bfrag<T>(ew, 1, 32, 0, 0, q, m), load_bfrag<T>(ew, 1, 32, 16, 0, q, m)};
    float ls4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ls4[i] = a.p.ls[4 * q + i];
    const int s = a.geo.s;
    const T* h0 = (const T*)a.h0;
    const T* dout = (const T*)a.dout;

    FwdArgs fa;  // only the FiLM pointers are used by load_film
    fa.sc = a.sc; fa.sh = a.sh; fa.scd = nullptr; fa.shd = nullptr;

    int64_t t0 = (int64_t)blockIdx.x * a.geo.chunk;
    int64_t t1 = t0 + a.geo.chunk < a.geo.total ? t0 + a.geo.chunk : a.geo.total;
    int64_t rcur = -1;
    float qv[2] = {0.f, 0.f}, kg[2] = {0.f, 0.f};
    float dqp[2] = {0.f, 0.f}, dbeta[2] = {0.f, 0.f};
    // MODE 1 accumulators
    f32x4 aWp[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}}, aWe[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    float dls = 0.f, dbp = 0.f, dbe[2] = {0.f, 0.f}, dbc[4] = {0.f, 0.f, 0.f, 0.f};

    auto flush_row = [&](int64_t r) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float v = red_q(dqp[j]);
                if (q == 0) atomicAdd(a.dq + r * 32 + 16 * j + m, v);
                dqp[j] = 0.f;
            }
        }
    };

    for (int64_t t = t0; t < t1; ++t) {
        const int64_t r = t / a.geo.tilesPerImg;
        const int ti = (int)(t - r * a.geo.tilesPerImg);
        const int y0 = (ti / a.geo.tilesX) * TH, x0 = (ti % a.geo.tilesX) * TW;
        __syncthreads();
        if (r != rcur) {
            if (rcur >= 0) flush_row(rcur);
            rcur = r;
            load_film<T, false>(l, fa, r);
            qv[0] = a.q[r * 32 + m]; qv[1] = a.q[r * 32 + 16 + m];
            if constexpr (MODE == 1) { kg[0] = a.kG[r * 32 + m]; kg[1] = a.kG[r * 32 + 16 + m]; }
            __syncthreads();
        }
        stage_h2<T, false>(l, h0, nullptr, r, s, y0, x0);
        __syncthreads();
#pragma unroll 1
        for (int ri = 0; ri < RPW; ++ri) {
            const int y = wave * RPW + ri;
            const int gy = y0 + y, gx = x0 + m;
            RowFwd<T, false> f;
            chain_row<T, false>(l, w, y, q, m, f);
            // dout for this tile row, A layout (zero outside the image)
            float dov[4] = {0.f, 0.f, 0.f, 0.f};
            const int64_t goff = ((r * s + gy) * (int64_t)s + gx) * 16 + 4 * q;
            if (gy < s && gx < s) ld4(dout + goff, dov);
            frag_t dp1f;
            make_frag(dp1f, dov[0] * ls4[0], dov[1] * ls4[1], dov[2] * ls4[2], dov[3] * ls4[3]);
            f32x4 dy[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                dy[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                mma16(dy[j], dp1f, wpT[j]);
            }
            if constexpr (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { dqp[j] += dy[j][e] * f.g[j][e]; dbeta[j] += dy[j][e]; }
            } else {
                // ---- recompute y, p1 (needed for dW_contract and d layer_scale)
                f32x4 yv[2];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        yv[j][e] = f.g[j][e] * (w.gam[j] + qv[j]) + w.bet[j];
                        l.x32a[(4 * q + e) * ES + 16 * j + m] = yv[j][e];
                    }
                // dout to scratch for the C-layout view
                st4(l.x16b + m * CS + 4 * q, dov);
                lds_fence();
                f32x4 p1 = f32x4{w.bp, w.bp, w.bp, w.bp};
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    float t4[4];
                    ld4(l.x32a + m * ES + 16 * c + 4 * q, t4);
                    frag_t yf;
                    make_frag(yf, t4[0], t4[1], t4[2], t4[3]);
                    mma16(p1, yf, w.wp[c]);
                }
                f32x4 dp1c;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = l.x16b[(4 * q + e) * CS + m];  // dout[pixel 4q+e][channel m]
                    dls += d * p1[e];
                    dp1c[e] = d * w.lsn;
                    dbp += dp1c[e];
                }
                lds_fence();
                // dW_contract[e][c] += sum_pixels y[p][e] dp1[p][c]   (C-layout operands)
                frag_t bdp;
                frag_of(bdp, dp1c);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    frag_t ay;
                    frag_of(ay, yv[j]);
                    mma16(aWp[j], ay, bdp);
                }
                // d gelu / d expand
                f32x4 de[2];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool ok = gy < s && (x0 + 4 * q + e) < s;
                        const float dg = dy[j][e] * (w.gam[j] + qv[j]) + f.g[j][e] * kg[j];
                        const float d = ok ? dg * gelu_grad_f(f.e[j][e]) : 0.f;
                        de[j][e] = d;
                        dbe[j] += d;
                        l.x32b[(4 * q + e) * ES + 16 * j + m] = d;
                    }
                // dW_expand[c][e] += sum_pixels n1[p][c] de[p][e]; n1 in C layout from c1 + stats
                f32x4 n1c;
#pragma unroll
                for (int e = 0; e < 4; ++e) n1c[e] = (f.c1[e] - l.st[4 * q + e]) * l.st[16 + 4 * q + e];
                frag_t an1;
                frag_of(an1, n1c);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    frag_t bde;
                    frag_of(bde, de[j]);
                    mma16(aWe[j], an1, bde);
                }
                lds_fence();
                // dn1 = de . We^T  (A operand: de in A layout via scratch)
                f32x4 dn1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    float t4[4];
                    ld4(l.x32b + m * ES + 16 * c + 4 * q, t4);
                    frag_t df;
                    make_frag(df, t4[0], t4[1], t4[2], t4[3]);
                    mma16(dn1, df, weT[c]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) l.x16a[(4 * q + e) * CS + m] = dn1[e];
                lds_fence();
                float dn[4], dc[4];
                ld4(l.x16a + m * CS + 4 * q, dn);
                ln_bwd_a(dn, f.n1, f.rho1, dc);
                if (gy < s && gx < s) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) dbc[i] += dc[i];
                    st4((T*)a.dc1 + goff, dc);
                }
                lds_fence();
            }
        }
    }
    if (rcur >= 0) flush_row(rcur);
    if constexpr (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float v = red_q(dbeta[j]);
            if (q == 0) atomicAdd(a.g.beta + 16 * j + m, v);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                atomicAdd(a.g.con_w + (16 * j + 4 * q + e) * 16 + m, aWp[j][e]);
                atomicAdd(a.g.exp_w + (4 * q + e) * 32 + 16 * j + m, aWe[j][e]);
            }
        const float v1 = red_q(dls), v2 = red_q(dbp);
        if (q == 0) { atomicAdd(a.g.ls + m, v1); atomicAdd(a.g.con_b + m, v2); }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float v = red_q(dbe[j]);
            if (q == 0) atomicAdd(a.g.exp_b + 16 * j + m, v);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = red_m(dbc[i]);
            if (m == 0) atomicAdd(a.g.conv_b + 4 * q + i, v);
        }
    }
}

// backward pass 3: conv3x3 transpose + conv weight gradient + FiLM/LN0 backward
template <typename T>
__global__ void __launch_bounds__(NT)
cnx_bwd_conv_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename Frag<T>::type frag_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, m = lane & 15;
    Lds<T> l = carve<T>(smem, true, wave);
    const T* cw = (const T*)a.p.conv_w;  // [tap][ic][oc]
    frag_t wcT[9];  // B[k=oc][col=ic]
#pragma unroll
    for (int t = 0; t < 9; ++t) wcT[t] = load_bfrag<T>(cw + t * 256, 1, 16, 0, 0, q, m);
    const int s = a.geo.s;
    const T* h0 = (const T*)a.h0;
    const T* dout = (const T*)a.dout;
    const T* dc1 = (const T*)a.dc1_in;
    FwdArgs fa;
    fa.sc = a.sc; fa.sh = a.sh; fa.scd = nullptr; fa.shd = nullptr;

    int64_t t0 = (int64_t)blockIdx.x * a.geo.chunk;
    int64_t t1 = t0 + a.geo.chunk < a.geo.total ? t0 + a.geo.chunk : a.geo.total;
    int64_t rcur = -1;
    f32x4 aWc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) aWc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dscp[4] = {0.f, 0.f, 0.f, 0.f}, dshp[4] = {0.f, 0.f, 0.f, 0.f};
    float sc4[4] = {0.f, 0.f, 0.f, 0.f};

    auto flush_row = [&](int64_t r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v1 = red_m(dscp[i]), v2 = red_m(dshp[i]);
            if (m == 0) { atomicAdd(a.dsc + r * 16 + 4 * q + i, v1); atomicAdd(a.dsh + r * 16 + 4 * q + i, v2); }
            dscp[i] = 0.f; dshp[i] = 0.f;
        }
    };

    for (int64_t t = t0; t < t1; ++t) {
        const int64_t r = t / a.geo.tilesPerImg;
        const int ti = (int)(t - r * a.geo.tilesPerImg);
        const int y0 = (ti / a.geo.tilesX) * TH, x0 = (ti % a.geo.tilesX) * TW;
        __syncthreads();
        if (r != rcur) {
            if (rcur >= 0) flush_row(rcur);
            rcur = r;
            load_film<T, false>(l, fa, r);
#pragma unroll
            for (int i = 0; i < 4; ++i) sc4[i] = a.sc[r * 16 + 4 * q + i];
            __syncthreads();
        }
        stage_h2<T, false>(l, h0, nullptr, r, s, y0, x0);
        // dc1 halo (zero outside the image)
        for (int hp = threadIdx.x; hp < NHALO; hp += NT) {
            const int hy = hp / HW, hx = hp - hy * HW;
            const int gy = y0 + hy - 1, gx = x0 + hx - 1;
            float v[16];
            if (gy >= 0 && gy < s && gx >= 0 && gx < s) {
                ld16<T>(dc1 + ((r * s + gy) * (int64_t)s + gx) * 16, v);
            } else {
#pragma unroll
                for (int c = 0; c < 16; ++c) v[c] = 0.f;
            }
            st16_lds<T>(l.aux + hp * CS, v);
        }
        __syncthreads();
#pragma unroll 1
        for (int ri = 0; ri < RPW; ++ri) {
            const int y = wave * RPW + ri;
            const int gy = y0 + y, gx = x0 + m;
            // dh2 = conv^T(dc1): h2[p] feeds c1[p - (i-1, j-1)] through K[i][j]
            f32x4 dh = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const frag_t ad = *reinterpret_cast<const frag_t*>(
                        l.aux + ((y + 2 - i) * HW + (m + 2 - j)) * CS + 4 * q);
                    mma16(dh, ad, wcT[i * 3 + j]);
                }
            // dWc[tap][ic][oc] += sum_pixels h2[p + tap][ic] dc1[p][oc]
            {
                const T* dcp = l.aux + ((y + 1) * HW + (4 * q + 1)) * CS + m;
                frag_t bd;
                frag_raw(bd, dcp[0], dcp[CS], dcp[2 * CS], dcp[3 * CS]);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const T* hp = l.h2s + ((y + i) * HW + (4 * q + j)) * CS + m;
                        frag_t ah;
                        frag_raw(ah, hp[0], hp[CS], hp[2 * CS], hp[3 * CS]);
                        mma16(aWc[i * 3 + j], ah, bd);
                    }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) l.x16a[(4 * q + e) * CS + m] = dh[e];
            lds_fence();
            float d2[4];
            ld4(l.x16a + m * CS + 4 * q, d2);
            lds_fence();
            if (gy < s && gx < s) {
                const int64_t goff = ((r * s + gy) * (int64_t)s + gx) * 16 + 4 * q;
                float dov[4], hv[4];
                ld4(dout + goff, dov);
                ld4(h0 + goff, hv);
#pragma unroll
                for (int i = 0; i < 4; ++i) d2[i] += dov[i];  // residual branch o = ... + h2
                float h1[4], mean, rho;
                ln_fwd_a(hv, h1, mean, rho);
                float dh1[4], dx[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    dscp[i] += d2[i] * h1[i];
                    dshp[i] += d2[i];
                    dh1[i] = d2[i] * (1.0f + sc4[i]);
                }
                ln_bwd_a(dh1, h1, rho, dx);
                st4((T*)a.dh0 + goff, dx);
            } else {
                // keep the cross-lane reductions convergent: every lane of a pixel group
                // takes the same branch only when the whole 4-lane group is valid/invalid
                // (gy, gx depend on m only, shared by the 4 q-lanes) -- nothing to do.
            }
        }
    }
    if (rcur >= 0) flush_row(rcur);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(a.g.conv_w + (t * 16 + 4 * q + e) * 16 + m, aWc[t][e]);
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
    atomicAdd(dgamma + c, d);
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

constexpr int64_t MAX_BLOCKS = 2048;

template <typename K, typename A>
inline int launch_k(K kern, int64_t grid, size_t lds, hipStream_t st, const A& args) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, st, args);
    return mfc_launch_status();
}

template <typename T>
int fwd_launch(bool jvp, int mode, const FwdArgs& a, int64_t grid, hipStream_t st) {
    const size_t lds = lds_bytes<T>(jvp);
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
               void* o, void* odot, void* stream) {
    if (!h0 || !scale || !shift || !params_ok(p)) return MFC_EFAULT;
    if (R <= 0 || s <= 0) return MFC_EINVAL;
    if (dtype != MFC_F32 && dtype != MFC_BF16) return MFC_EINVAL;
    const bool jvp = h0dot != nullptr;
    if (jvp && (!scaledot || !shiftdot)) return MFC_EFAULT;
    if (mode == 0 && (!S1 || (jvp && !S2))) return MFC_EFAULT;
    if (mode == 1 && (!q || !o || (jvp && (!qdot || !odot)))) return MFC_EFAULT;
    FwdArgs a;
    int64_t grid;
    a.geo = make_geo(R, s, MAX_BLOCKS, grid);
    a.h0 = h0; a.h0d = h0dot; a.sc = scale; a.sh = shift; a.scd = scaledot; a.shd = shiftdot;
    a.p = to_dev(p); a.S1 = S1; a.S2 = S2; a.q = q; a.qd = qdot; a.o = o; a.od = odot;
    hipStream_t st = (hipStream_t)stream;
    return dtype == MFC_F32 ? fwd_launch<float>(jvp, mode, a, grid, st) : fwd_launch<u16>(jvp, mode, a, grid, st);
}

}  // namespace

extern "C" int mfc_cnx_stats(int dtype, int64_t R, int s, const void* h0, const void* h0dot,
                             const float* scale, const float* shift, const float* scaledot,
                             const float* shiftdot, const mfc_cnx_params* p, float* S1, float* S2,
                             void* stream) {
    return fwd_common(dtype, 0, R, s, h0, h0dot, scale, shift, scaledot, shiftdot, p, S1, S2, nullptr, nullptr,
                      nullptr, nullptr, stream);
}

extern "C" int mfc_cnx_apply(int dtype, int64_t R, int s, const void* h0, const void* h0dot,
                             const float* scale, const float* shift, const float* scaledot,
                             const float* shiftdot, const mfc_cnx_params* p, const float* q, const float* qdot,
                             void* o, void* odot, void* stream) {
    return fwd_common(dtype, 1, R, s, h0, h0dot, scale, shift, scaledot, shiftdot, p, nullptr, nullptr, q, qdot,
                      o, odot, stream);
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
    return mfc_launch_status();
}

extern "C" int mfc_cnx_bwd_stats(int dtype, int64_t R, int s, const void* h0, const float* scale,
                                 const float* shift, const mfc_cnx_params* p, const float* q, const void* dout,
                                 float* dq, float* dbeta, void* stream) {
    if (!h0 || !scale || !shift || !params_ok(p) || !q || !dout || !dq || !dbeta) return MFC_EFAULT;
    if (R <= 0 || s <= 0 || (dtype != MFC_F32 && dtype != MFC_BF16)) return MFC_EINVAL;
    BwdArgs a = {};
    int64_t grid;
    a.geo = make_geo(R, s, MAX_BLOCKS, grid);
    a.h0 = h0; a.sc = scale; a.sh = shift; a.p = to_dev(p); a.g = to_devg(nullptr);
    a.g.beta = dbeta; a.q = q; a.dout = dout; a.dq = dq;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MFC_F32) return launch_k(cnx_bwd_kernel<float, 0>, grid, lds_bytes<float>(false), st, a);
    return launch_k(cnx_bwd_kernel<u16, 0>, grid, lds_bytes<u16>(false), st, a);
}

extern "C" int mfc_cnx_bwd_main(int dtype, int64_t R, int s, const void* h0, const float* scale,
                                const float* shift, const mfc_cnx_params* p, const float* q, const float* kG,
                                const void* dout, void* dc1, const mfc_cnx_grads* g, void* stream) {
    if (!h0 || !scale || !shift || !params_ok(p) || !q || !kG || !dout || !dc1 || !g) return MFC_EFAULT;
    if (!g->con_w || !g->con_b || !g->ls || !g->exp_w || !g->exp_b || !g->conv_b) return MFC_EFAULT;
    if (R <= 0 || s <= 0 || (dtype != MFC_F32 && dtype != MFC_BF16)) return MFC_EINVAL;
    BwdArgs a = {};
    int64_t grid;
    a.geo = make_geo(R, s, MAX_BLOCKS, grid);
    a.h0 = h0; a.sc = scale; a.sh = shift; a.p = to_dev(p); a.g = to_devg(g);
    a.q = q; a.kG = kG; a.dout = dout; a.dc1 = dc1;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MFC_F32) return launch_k(cnx_bwd_kernel<float, 1>, grid, lds_bytes<float>(false), st, a);
    return launch_k(cnx_bwd_kernel<u16, 1>, grid, lds_bytes<u16>(false), st, a);
}

extern "C" int mfc_cnx_bwd_conv(int dtype, int64_t R, int s, const void* h0, const float* scale,
                                const float* shift, const mfc_cnx_params* p, const void* dc1, const void* dout,
                                void* dh0, const mfc_cnx_grads* g, float* dscale, float* dshift, void* stream) {
    if (!h0 || !scale || !shift || !params_ok(p) || !dc1 || !dout || !dh0 || !g || !g->conv_w || !dscale ||
        !dshift)
        return MFC_EFAULT;
    if (R <= 0 || s <= 0 || (dtype != MFC_F32 && dtype != MFC_BF16)) return MFC_EINVAL;
    BwdArgs a = {};
    int64_t grid;
    a.geo = make_geo(R, s, MAX_BLOCKS, grid);
    a.h0 = h0; a.sc = scale; a.sh = shift; a.p = to_dev(p); a.g = to_devg(g);
    a.dc1_in = dc1; a.dout = dout; a.dh0 = dh0; a.dsc = dscale; a.dsh = dshift;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MFC_F32) return launch_k(cnx_bwd_conv_kernel<float>, grid, lds_bytes<float>(true), st, a);
    return launch_k(cnx_bwd_conv_kernel<u16>, grid, lds_bytes<u16>(true), st, a);
}
