// MDCT / IMDCT for the shipped window N = 512 (configs/*tokenization=mdct.json: window_size 512, hop_size 256) on gfx950.
//
// Same transform as mdct.hip (reference direct path, preprocessing/mdct.py:126-136,317-340,361-372,410-422,476-540;
// fast algorithm of SURVEY Appendix A.2/A.3: TDAC fold 2N -> N, DCT-IV as one M = N/2 = 256 point complex FFT with
// pre/post twiddles), restructured around the register file:
//
//   * 256 = 16 x 16.  Sixteen lanes own one frame; each lane holds 16 complex points in registers and runs a 16-point
//     FFT (two radix-4 stages, compile-time indices, no div/mod, no LDS) -- twice, with ONE exchange through LDS in
//     between:   Z[k1 + 16 k2] = sum_j W16^(j k2) * [ W256^(j k1) * sum_i W16^(i k1) z[j + 16 i] ].
//     Pass 1: lane j, registers i.  Pass 2: lane k1, registers j.  The inter-pass twiddles W256^(j k1) come from a
//     [k1][j] LDS table (16 consecutive lanes read 16 consecutive entries; the wave's four frames read the same ones).
//   * The exchange image is [frame][k1][j] of float2 with every 16-point row padded to 144 bytes, so the pass-2 reads
//     (ds_read_b128, 128 contiguous bytes per lane, lanes 144 bytes apart) and the pass-1 writes (ds_write_b64, 16
//     consecutive lanes contiguous) are bank-conflict free.
//   * A workgroup = 256 threads = 16 frames per iteration, persistent over a contiguous range of frames of one clip.
//     Global traffic is 16-byte vectors: the next iteration's input is fetched into registers while the current one
//     computes (issue early / write late) and the output leaves as whole rows through an LDS staging image.
//   * Inverse: each frame's DCT-IV is computed ONCE; the windowed frames are overlap-added in LDS -- output samples
//     that later frames still touch (the last 2N - hop samples of an iteration) are carried in LDS to the next
//     iteration of the same workgroup.  A sample's frames are added in ascending frame order (deterministic, the same
//     order as the gather kernel of mdct.hip).  Only the first iteration of a workgroup's range recomputes the
//     ceil(2N/hop) - 1 frames before it (3 of 96 at hop = 256).
//
// HBM-bound: algorithmic bytes per clip 4T + 4 n_frames N (forward), 4 n_frames N + 4 out_len (inverse), SURVEY 8(d).
#include "mfc_common.h"

namespace {
namespace m512 {

constexpr int N = 512, M = 256, F = 16, NT = 256;
constexpr int EX_ROW = 18;                 // float2 per 16-point row: 16 + 2 pad = 144 bytes
constexpr int EX_FRAME = 16 * EX_ROW;      // float2 per frame
constexpr int EX_FLOATS = 2 * F * EX_FRAME;   // 9216 floats >= F * N = 8192 (the row images alias the exchange image)

__device__ inline float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ inline float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ inline float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// forward radix-4 butterfly (W4 = -i)
__device__ inline void bfly4(float2 x0, float2 x1, float2 x2, float2 x3, float2& y0, float2& y1, float2& y2, float2& y3) {
    const float2 a0 = cadd(x0, x2), a1 = csub(x0, x2), a2 = cadd(x1, x3);
    const float2 d = csub(x1, x3);
    const float2 a3 = make_float2(d.y, -d.x);        // (x1 - x3) * (-i)
    y0 = cadd(a0, a2); y1 = cadd(a1, a3); y2 = csub(a0, a2); y3 = csub(a1, a3);
}

// in-register 16-point forward DFT, natural order in and out: k = q + 4 r, m = n + 4 p,
// W16^(mk) = W4^(pq) * W16^(nq) * W4^(nr)
__device__ inline void fft16(float2 (&v)[16]) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
    float2 t[4][4];
#pragma unroll
    for (int n = 0; n < 4; ++n) bfly4(v[n], v[n + 4], v[n + 8], v[n + 12], t[n][0], t[n][1], t[n][2], t[n][3]);
    // twiddles W16^(n q), n, q in 1..3: angles 1,2,3 / 2,4,6 / 3,6,9 (x 2 pi / 16)
    t[1][1] = cmul(t[1][1], make_float2(C1, -S1));
    t[1][2] = cmul(t[1][2], make_float2(R2, -R2));
    t[1][3] = cmul(t[1][3], make_float2(S1, -C1));
    t[2][1] = cmul(t[2][1], make_float2(R2, -R2));
    t[2][2] = make_float2(t[2][2].y, -t[2][2].x);                      // W16^4 = -i
    t[2][3] = cmul(t[2][3], make_float2(-R2, -R2));
    t[3][1] = cmul(t[3][1], make_float2(S1, -C1));
    t[3][2] = cmul(t[3][2], make_float2(-R2, -R2));
    t[3][3] = cmul(t[3][3], make_float2(-C1, S1));                     // W16^9
#pragma unroll
    for (int q = 0; q < 4; ++q) bfly4(t[0][q], t[1][q], t[2][q], t[3][q], v[q], v[q + 4], v[q + 8], v[q + 12]);
}

struct Lds {
    float2* tw;     // [256] e^{-i pi m / N}            (pre-twiddle)
    float2* twp;    // [256] e^{-i pi (k + 1/4) / N}    (post-twiddle)
    float* win;     // [512] w[n] = sin(pi (n + 1/2) / 2N), n < N   (w[2N-1-n] = w[n])
    float2* wjt;    // [16][16] W256^(j k1) at [k1][j]   (inter-pass twiddle)
    float* ex;      // [EX_FLOATS] exchange image; aliased by the sample span / row images
    float* extra;   // inverse: two carry buffers of 2N - hop floats
};
__device__ inline Lds carve(float* smem) {
    Lds l;
    l.tw = (float2*)smem;
    l.twp = l.tw + M;
    l.win = (float*)(l.twp + M);
    l.wjt = (float2*)(l.win + N);
    l.ex = (float*)(l.wjt + 256);
    l.extra = l.ex + EX_FLOATS;
    return l;
}
__device__ inline void init_tables(const Lds& l, int tid) {
    float s, c;
    sincospif((float)tid / (float)N, &s, &c);
    l.tw[tid] = make_float2(c, -s);
    sincospif(((float)tid + 0.25f) / (float)N, &s, &c);
    l.twp[tid] = make_float2(c, -s);
    l.win[tid] = sinpif(((float)tid + 0.5f) / (float)(2 * N));
    l.win[tid + 256] = sinpif(((float)(tid + 256) + 0.5f) / (float)(2 * N));
    sincospif((float)(((tid & 15) * (tid >> 4)) & 255) / 128.0f, &s, &c);      // entry [k1 = tid >> 4][j = tid & 15]
    l.wjt[tid] = make_float2(c, -s);
}

// passes 1 and 2 of the 256-point FFT of frame f (lane j of it): z[i] = point j + 16 i on entry, Z[j + 16 k2] on exit
__device__ inline void fft256(float2 (&z)[16], const float2* wjt, float* ex, int f, int j) {
    fft16(z);
    float2* exf = (float2*)ex + f * EX_FRAME;
    int oE = f * EX_FRAME + j, oW = j;
    asm volatile("" : "+v"(oE), "+v"(oW));
    float2* exj = (float2*)ex + oE;
    const float2* wj = wjt + oW;
#pragma unroll
    for (int k = 0; k < 16; ++k) exj[k * EX_ROW] = k == 0 ? z[0] : cmul(z[k], wj[16 * k]);
    __syncthreads();
    const f32x4* row = reinterpret_cast<const f32x4*>(exf + j * EX_ROW);   // lane j now plays k1
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const f32x4 t = row[p];
        z[2 * p] = make_float2(t[0], t[1]);
        z[2 * p + 1] = make_float2(t[2], t[3]);
    }
    fft16(z);
}

// ---------------------------------------------------------------------------------------------------------------
// forward.  Group g: frames [i0, i0 + F) of clip b; groups g, g + grid, ... belong to one persistent workgroup.  The
// 2N - hop samples two adjacent groups share are read by both (neighbouring workgroups run at the same time: L2 /
// Infinity Cache hits; the whole batch of clips is 100 MB).
// ---------------------------------------------------------------------------------------------------------------
// MAXV = float4 per thread of one sample span, ceil(((F-1) * hop + 2N) / 4 / NT): 5 for hop <= 256, 9 for hop <= 512
template <bool VEC, int MAXV>
__device__ inline void fwd_fetch(f32x4 (&pre)[MAXV], const float* xb, int64_t s0, int64_t T, int span4, int tid) {
#pragma unroll
    for (int r = 0; r < MAXV; ++r) {
        const int q = tid + r * NT;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (q < span4) {
            const int64_t p = s0 + 4 * (int64_t)q;
            if (VEC && p + 4 <= T) v = *reinterpret_cast<const f32x4*>(xb + p);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (p + e < T) ? xb[p + e] : 0.f;
            }
        }
        pre[r] = v;
    }
}

template <bool VEC, int MAXV>
__global__ void __launch_bounds__(NT, 3)      // <= 168 VGPRs: three workgroups (= waves per SIMD) per CU
mdct512_fwd_kernel(const float* __restrict__ x, int64_t T, int64_t ldx, int hop, int64_t nf, int64_t gpc,
                   int64_t n_groups, float* __restrict__ X) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, f = tid >> 4, j = tid & 15;
    const Lds l = carve(smem);
    init_tables(l, tid);
    const int span4 = ((F - 1) * hop + 2 * N) / 4;
    f32x4 pre[MAXV];
    int64_t g = blockIdx.x;
    auto coords = [&](int64_t gg, int64_t& b, int64_t& i0) { b = gg / gpc; i0 = (gg - b * gpc) * F; };
    if (g < n_groups) {
        int64_t b, i0;
        coords(g, b, i0);
        fwd_fetch<VEC, MAXV>(pre, x + b * ldx, i0 * hop, T, span4, tid);
    }
    for (; g < n_groups; g += gridDim.x) {
        int64_t b, i0;
        coords(g, b, i0);
        const int nfr = (int)((nf - i0) < F ? (nf - i0) : F);
        __syncthreads();                       // tables ready / previous iteration's row image fully stored
        float* span = l.ex;
#pragma unroll
        for (int r = 0; r < MAXV; ++r) {
            const int q = tid + r * NT;
            if (q < span4) *reinterpret_cast<f32x4*>(span + 4 * q) = pre[r];
        }
        // the next iteration's span: requested now (before the 32 registers of z are live), in flight during the whole
        // iteration, written to LDS at the top of the next one
        if (g + gridDim.x < n_groups) {
            int64_t bn, in0;
            coords(g + gridDim.x, bn, in0);
            fwd_fetch<VEC, MAXV>(pre, x + bn * ldx, in0 * hop, T, span4, tid);
        }
        __syncthreads();
        // window + TDAC fold + pre-twiddle: z[m] = (u[2m] + i u[N-1-2m]) e^{-i pi m / N}, m = j + 16 i
        float2 z[16];
        {
            // every index below is (a per-lane base) + (a compile-time constant): two sample bases, two window bases and
            // one twiddle base per lane, the rest folds into the DS instructions' offset fields
            // (the laundered quantities are INDICES, not pointers: an asm-laundered pointer loses its LDS address space
            // and every access through it becomes a flat_load)
            int oA = f * hop + 2 * j, oB = f * hop - 2 * j, oWA = 2 * j, oWB = -2 * j, oT = j;
            asm volatile("" : "+v"(oA), "+v"(oB), "+v"(oWA), "+v"(oWB), "+v"(oT));
            const float* pA = span + oA;           // pA[c] = fr[c + 2j]
            const float* pB = span + oB;           // pB[c] = fr[c - 2j]   (every c used below keeps c - 2j >= 0)
            const float* wA_ = l.win + oWA;
            const float* wB_ = l.win + oWB;
            const float2* twj = l.tw + oT;
#pragma unroll
            for (int i = 0; i < 16; ++i) {         // m = j + 16 i
                float re, im;
                if (i < 8) {      // m < N/4
                    const float wA = wB_[N / 2 - 1 - 32 * i], wB = wA_[N / 2 + 32 * i];
                    re = -pB[3 * N / 2 - 1 - 32 * i] * wB - pA[3 * N / 2 + 32 * i] * wA;
                    im = pB[N / 2 - 1 - 32 * i] * wA - pA[N / 2 + 32 * i] * wB;
                } else {
                    const float wa = wA_[32 * i - N / 2], wb = wB_[3 * N / 2 - 1 - 32 * i];
                    re = pA[32 * i - N / 2] * wa - pB[3 * N / 2 - 1 - 32 * i] * wb;
                    im = -pA[N / 2 + 32 * i] * wb - pB[5 * N / 2 - 1 - 32 * i] * wa;
                }
                z[i] = cmul(make_float2(re, im), twj[16 * i]);
            }
        }
        __syncthreads();                       // every lane has its samples: the span image may be overwritten
        fft256(z, l.wjt, l.ex, f, j);
        __syncthreads();                       // every lane has read its exchange row: the row image may be written
        // post-twiddle: W = Z[k] e^{-i pi (k + 1/4) / N};  X[2k] = Re W, X[N-1-2k] = -Im W
        {
            int oA = f * N + 2 * j, oB = f * N - 2 * j, oT = j;
            asm volatile("" : "+v"(oA), "+v"(oB), "+v"(oT));
            float* rA = l.ex + oA;                  // rA[c] = rows[c + 2j]
            float* rB = l.ex + oB;                  // rB[c] = rows[c - 2j]
            const float2* tpj = l.twp + oT;
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {       // k = j + 16 k2
                const float2 W = cmul(z[k2], tpj[16 * k2]);
                rA[32 * k2] = W.x;
                rB[N - 1 - 32 * k2] = -W.y;
            }
        }
        __syncthreads();
        float* Xo = X + (b * nf + i0) * (int64_t)N;      // nfr consecutive rows: one contiguous block
        const int tot4 = nfr * (N / 4);
        for (int q = tid; q < tot4; q += NT)
            *reinterpret_cast<f32x4*>(Xo + 4 * q) = *reinterpret_cast<const f32x4*>(l.ex + 4 * q);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// inverse.  Workgroup g owns the output samples of frames [fs, fe) of clip b, i.e. samples [fs * hop, fe * hop) (the
// clip's last segment: up to out_len), and walks frames [max(0, fs - lead), fe) in iterations of F frames.
// ---------------------------------------------------------------------------------------------------------------
constexpr int INV_V = F * N / 4 / NT;   // 8 float4 per thread of one 16-row block

__device__ inline void inv_fetch(f32x4 (&pre)[INV_V], const float* Xi, int nfr, int tid) {
#pragma unroll
    for (int r = 0; r < INV_V; ++r) {
        const int q = tid + r * NT;
        pre[r] = q < nfr * (N / 4) ? *reinterpret_cast<const f32x4*>(Xi + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

template <bool HOP256>      // hop == 256 (every shipped config): the overlap-add has a fixed per-thread structure
__global__ void __launch_bounds__(NT, 3)      // <= 168 VGPRs: three workgroups (= waves per SIMD) per CU
mdct512_inv_kernel(const float* __restrict__ X, int64_t nf, int hop, int hop_shift, int64_t out_len, int seg_frames,
                   int64_t spc, int64_t n_segs, float* __restrict__ y, int64_t ldy, int vec_y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, f = tid >> 4, j = tid & 15;
    const Lds l = carve(smem);
    init_tables(l, tid);
    const int CL = 2 * N - hop;                 // samples of an iteration that later frames still touch
    const int lead = (2 * N + hop - 1) / hop - 1;
    const float scale = 2.0f / (float)N;
    f32x4 pre[INV_V];
    // first 16-row block of segment gg (also called one segment ahead, from the last iteration of the previous one)
    auto fetch_first = [&](int64_t gg) {
        const int64_t bb = gg / spc;
        const int64_t s0 = (gg - bb * spc) * seg_frames;
        const int64_t e0 = (s0 + seg_frames < nf) ? s0 + seg_frames : nf;
        const int64_t cb = s0 > lead ? s0 - lead : 0;
        inv_fetch(pre, X + (bb * nf + cb) * (int64_t)N, (int)((e0 - cb) < F ? (e0 - cb) : F), tid);
    };
    if ((int64_t)blockIdx.x < n_segs) fetch_first(blockIdx.x);

    for (int64_t g = blockIdx.x; g < n_segs; g += gridDim.x) {
        const int64_t b = g / spc;
        const int64_t fs = (g - b * spc) * seg_frames;
        const int64_t fe = (fs + seg_frames < nf) ? fs + seg_frames : nf;
        const bool last_seg = fe == nf;
        const int64_t own_lo = fs * hop;
        const int64_t own_hi = last_seg ? out_len : fe * hop;
        const int64_t c_begin = fs > lead ? fs - lead : 0;
        const float* Xb = X + b * nf * (int64_t)N;
        float* yb = y + b * ldy;
        float* carry_in = l.extra;
        float* carry_out = l.extra + CL;
        __syncthreads();
        for (int q = tid; q < CL; q += NT) carry_in[q] = 0.f;
        for (int64_t c0 = c_begin; c0 < fe; c0 += F) {
            const int nfr = (int)((fe - c0) < F ? (fe - c0) : F);
            __syncthreads();                   // previous iteration's overlap-add has finished reading U
            float* rows = l.ex;
#pragma unroll
            for (int r = 0; r < INV_V; ++r) *reinterpret_cast<f32x4*>(rows + 4 * (tid + r * NT)) = pre[r];
            __syncthreads();
            // z[m] = (X[2m] + i X[N-1-2m]) e^{-i pi m / N}
            float2 z[16];
            {
                int oA = f * N + 2 * j, oB = f * N - 2 * j, oT = j;
                asm volatile("" : "+v"(oA), "+v"(oB), "+v"(oT));
                const float* xA = rows + oA;
                const float* xB = rows + oB;
                const float2* twj = l.tw + oT;
#pragma unroll
                for (int i = 0; i < 16; ++i)      // m = j + 16 i
                    z[i] = cmul(make_float2(xA[32 * i], xB[N - 1 - 32 * i]), twj[16 * i]);
            }
            if (c0 + F < fe) {
                const int nn = (int)((fe - c0 - F) < F ? (fe - c0 - F) : F);
                inv_fetch(pre, Xb + (c0 + F) * N, nn, tid);
            } else if (g + gridDim.x < n_segs) {
                fetch_first(g + gridDim.x);       // the next segment's first block, in flight across the segment change
            }
            __syncthreads();
            fft256(z, l.wjt, l.ex, f, j);
            __syncthreads();
            // u = (2/N) DCT-IV(X):  u[2k] = s Re W, u[N-1-2k] = -s Im W
            {
                int oA = f * N + 2 * j, oB = f * N - 2 * j, oT = j;
                asm volatile("" : "+v"(oA), "+v"(oB), "+v"(oT));
                float* uA = l.ex + oA;
                float* uB = l.ex + oB;
                const float2* tpj = l.twp + oT;
#pragma unroll
                for (int k2 = 0; k2 < 16; ++k2) {     // k = j + 16 k2
                    const float2 W = cmul(z[k2], tpj[16 * k2]);
                    uA[32 * k2] = W.x * scale;
                    uB[N - 1 - 32 * k2] = -W.y * scale;
                }
            }
            __syncthreads();
            // overlap-add over the region [c0 * hop, (c0 + nfr - 1) * hop + 2N): carry from earlier frames first, then
            // this iteration's frames in ascending order.  Samples below (c0 + nfr) * hop are complete.
            const int L = (nfr - 1) * hop + 2 * N;
            const int done = nfr * hop;
            const int64_t base = c0 * hop;
            if constexpr (HOP256) {
                // A thread owns 4 consecutive samples p = 4 tid + e + 1024 R (16-byte LDS reads and global stores).  Their
                // 256-sample block r = 4 R + (tid >> 6) is wave-uniform; with t = 4 (tid & 63) + e the position inside
                // the block, frame i = r - d contributes its sample n = t + 256 d:
                //   d = 0:  +u[256 + t] w[t]         d = 1:  -u[511 - t] w[256 + t]
                //   d = 2:  -u[255 - t] w[511 - t]   d = 3:  -u[t] w[255 - t]          (w[n] = w[2N - 1 - n])
                // added in ascending frame order d = 3, 2, 1, 0 on top of the carry.
                const int tl = 4 * (tid & 63), blk = tid >> 6;
                f32x4 w0, w1, w2, w3;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    w0[e] = l.win[tl + e]; w1[e] = -l.win[256 + tl + e];
                    w2[e] = -l.win[511 - tl - e]; w3[e] = -l.win[255 - tl - e];
                }
                const bool flush_all = last_seg && c0 + nfr == nf;
                const int nr = nfr + 3;                           // 256-sample blocks of the region
                for (int R = 0; 4 * R + blk < nr; ++R) {
                    const int r = 4 * R + blk;
                    f32x4 s = r < 3 ? *reinterpret_cast<const f32x4*>(carry_in + 4 * tid + 1024 * R)      // CL = 768
                                    : f32x4{0.f, 0.f, 0.f, 0.f};
                    if (r >= 3 && r - 3 < nfr) {
                        const f32x4 u = *reinterpret_cast<const f32x4*>(l.ex + (r - 3) * N + tl);
                        s += w3 * u;
                    }
                    if (r >= 2 && r - 2 < nfr) {
                        const f32x4 u = *reinterpret_cast<const f32x4*>(l.ex + (r - 2) * N + 252 - tl);
                        s += w2 * f32x4{u[3], u[2], u[1], u[0]};
                    }
                    if (r >= 1 && r - 1 < nfr) {
                        const f32x4 u = *reinterpret_cast<const f32x4*>(l.ex + (r - 1) * N + 508 - tl);
                        s += w1 * f32x4{u[3], u[2], u[1], u[0]};
                    }
                    if (r < nfr) {
                        const f32x4 u = *reinterpret_cast<const f32x4*>(l.ex + r * N + 256 + tl);
                        s += w0 * u;
                    }
                    const int64_t gp = base + 4 * tid + 1024 * R;
                    if (r < nfr || flush_all) {
                        if (gp >= own_lo && gp < own_hi) {
                            if (vec_y) *reinterpret_cast<f32x4*>(yb + gp) = s;
                            else { yb[gp] = s[0]; yb[gp + 1] = s[1]; yb[gp + 2] = s[2]; yb[gp + 3] = s[3]; }
                        }
                    } else {
                        *reinterpret_cast<f32x4*>(carry_out + 4 * tid + 1024 * R - 256 * nfr) = s;
                    }
                }
            } else
            for (int p = tid; p < L; p += NT) {
                float s = p < CL ? carry_in[p] : 0.f;
                // frames i of this iteration with i * hop <= p < i * hop + 2N  (hop_shift >= 0: hop is a power of two)
                int a = p - 2 * N + 1;
                a = a <= 0 ? 0 : (hop_shift >= 0 ? (a + hop - 1) >> hop_shift : (a + hop - 1) / hop);
                int e = hop_shift >= 0 ? p >> hop_shift : p / hop;
                if (e > nfr - 1) e = nfr - 1;
                for (int i = a; i <= e; ++i) {
                    const int n = p - i * hop;
                    const float* u = l.ex + i * N;
                    float v;
                    if (n < N / 2) v = u[N / 2 + n];
                    else if (n < 3 * N / 2) v = -u[3 * N / 2 - 1 - n];
                    else v = -u[n - 3 * N / 2];
                    s += (n < N ? l.win[n] : l.win[2 * N - 1 - n]) * v;
                }
                const int64_t gp = base + p;
                if (p < done || (last_seg && c0 + nfr == nf)) {
                    if (gp >= own_lo && gp < own_hi) yb[gp] = s;
                } else {
                    carry_out[p - done] = s;
                }
            }
            float* t = carry_in; carry_in = carry_out; carry_out = t;
        }
    }
}

// workgroups that are resident at once (persistent grids: more would run as a second, partly idle round)
inline int64_t resident_blocks(const void* kern, size_t lds) {
    int dev = 0, cus = 256, per_cu = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, NT, lds) != hipSuccess || per_cu < 1) per_cu = 2;
    return (int64_t)cus * per_cu;
}

inline size_t lds_bytes(int extra_floats) {
    return (size_t)(2 * M + 256) * sizeof(float2) + (size_t)(N + EX_FLOATS + extra_floats) * sizeof(float);
}

}  // namespace m512
}  // namespace

// Entry points used by mfc_mdct_fwd / mfc_mdct_inv (mdct.hip) for N == 512.  Return MFC_ENOSYS when the shape is outside
// what these kernels cover (the caller then takes the generic kernels).
int mfc_mdct512_fwd(const float* x, int64_t B, int64_t T, int64_t ldx, int hop, int64_t nf, float* X, hipStream_t st) {
    using namespace m512;
    if (hop < 4 || hop > N || (hop & 3)) return MFC_ENOSYS;
    const bool vec = (((uintptr_t)x & 15) == 0) && ((ldx & 3) == 0);
    const int64_t gpc = ceil_div64(nf, F), n_groups = gpc * B;
    const size_t lds = lds_bytes(0);
    auto go = [&](auto kern) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const int64_t slots = resident_blocks((const void*)kern, lds);
        const int64_t grid = n_groups < slots ? n_groups : slots;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, st, x, T, ldx, hop, nf, gpc, n_groups, X);
    };
    const bool small = ((F - 1) * hop + 2 * N) / 4 <= 5 * NT;
    if (vec) { if (small) go(mdct512_fwd_kernel<true, 5>); else go(mdct512_fwd_kernel<true, 9>); }
    else { if (small) go(mdct512_fwd_kernel<false, 5>); else go(mdct512_fwd_kernel<false, 9>); }
    return mfc_launch_status();
}

int mfc_mdct512_inv(const float* X, int64_t B, int64_t nf, int hop, int64_t out_len, float* y, int64_t ldy,
                    hipStream_t st) {
    using namespace m512;
    if (hop < 4 || hop > N || ((uintptr_t)X & 15)) return MFC_ENOSYS;
    const int lead = (2 * N + hop - 1) / hop - 1;
    const size_t lds = lds_bytes(2 * (2 * N - hop));
    const bool h256 = hop == 256;
    const void* kern = h256 ? (const void*)mdct512_inv_kernel<true> : (const void*)mdct512_inv_kernel<false>;
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int64_t slots = resident_blocks(kern, lds);
    // segment = own frames of one workgroup visit; own + lead-in frames = a whole number of 16-frame iterations.  Three
    // iterations per segment keep the lead-in recompute at lead/48 and give every resident workgroup several segments.
    int iters = 3;
    while (iters * F <= lead) ++iters;
    int seg = iters * F - lead;
    while (iters > 1 && ceil_div64(nf, seg) * B < slots && (iters - 1) * F > lead) { --iters; seg = iters * F - lead; }
    const int64_t spc = ceil_div64(nf, seg), n_segs = spc * B;
    const int64_t grid = n_segs < slots ? n_segs : slots;
    int hop_shift = -1;
    if ((hop & (hop - 1)) == 0) for (hop_shift = 0; (1 << hop_shift) < hop; ++hop_shift) {}
    const int vec_y = (((uintptr_t)y & 15) == 0 && (ldy & 3) == 0) ? 1 : 0;      // 16-byte output stores legal
    if (h256)
        hipLaunchKernelGGL(mdct512_inv_kernel<true>, dim3((unsigned)grid), dim3(NT), lds, st, X, nf, hop, hop_shift, out_len,
                           seg, spc, n_segs, y, ldy, vec_y);
    else
        hipLaunchKernelGGL(mdct512_inv_kernel<false>, dim3((unsigned)grid), dim3(NT), lds, st, X, nf, hop, hop_shift, out_len,
                           seg, spc, n_segs, y, ldy, vec_y);
    return mfc_launch_status();
}
