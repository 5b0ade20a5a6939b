// MDCT / IMDCT tokenizer kernels for gfx950.
//
// Definition (reference direct path, preprocessing/mdct.py:126-136, 317-340,
// 361-372, 410-422, 476-495, 517-540):
//   frame i = samples [i*hop, i*hop+2N) (implicit right zero pad)
//   w[n]   = sin(pi (n+1/2) / 2N)
//   X[i,k] = sum_n w[n] x[i*hop+n] cos(pi/N (n + N/2 + 1/2)(k + 1/2))
//   y_i[n] = (2/N) w[n] sum_k X[i,k] cos(pi/N (n + N/2 + 1/2)(k + 1/2))
//   out    = overlap-add of y_i at i*hop.
//
// Fast algorithm (derived from that definition, NOT from the reference's FFT
// path, which is a different transform -- SURVEY defect 7; identities checked
// in SURVEY Appendix A.2/A.3 and tests/test_oracle_mdct.py):
//   fold 2N -> N (TDAC):  u[n] = -f[3N/2-1-n] - f[3N/2+n]          n <  N/2
//                         u[n] =  f[n-N/2]    - f[3N/2-1-n]        n >= N/2
//   DCT-IV by one M=N/2 point complex FFT held in LDS:
//     z[m] = (u[2m] + i u[N-1-2m]) e^{-i pi m/N};  Z = FFT_M(z)
//     W[k] = Z[k] e^{-i pi (k+1/4)/N};  X[2k] = Re W[k];  X[N-1-2k] = -Im W[k]
//   inverse: u = (2/N) DCT-IV(X);
//     y[n] = u[N/2+n] (n<N/2) | -u[3N/2-1-n] (N/2<=n<3N/2) | -u[n-3N/2] (else)
//   then window and a gather-style overlap-add (each output sample sums its
//   covering frames in ascending frame order: deterministic, no atomics).
//
// HBM-bound: algorithmic bytes/clip = 4T + 4 n_frames N (fwd),
// 4 n_frames N + 4 out_len (inv) -- SURVEY 8(d).
#include "mfc_common.h"
#include <cstdlib>

#define MDCT_THREADS 256

// mdct512.hip: the N = 512 kernels (register-resident 16 x 16 FFT); MFC_ENOSYS = shape not covered, take the generic path
int mfc_mdct512_fwd(const float* x, int64_t B, int64_t T, int64_t ldx, int hop, int64_t nf, float* X, hipStream_t st);
int mfc_mdct512_inv(const float* X, int64_t B, int64_t nf, int hop, int64_t out_len, float* y, int64_t ldy,
                    hipStream_t st);

namespace {

__device__ inline float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

struct MdctLds {
    float2* bufA;   // [F*M]
    float2* bufB;   // [F*M]
    float2* twM;    // [M]   e^{-2 pi i j / M}
    float2* twN;    // [M]   e^{-i pi m / N}
    float* win;     // [N]   w[n], n < N  (w[2N-1-n] = w[n])
    float* span;    // fwd: staged samples; inv: u rows [FC*N]
};

__device__ inline MdctLds carve(float* smem, int F, int M, int N) {
    MdctLds l;
    l.bufA = (float2*)smem;
    l.bufB = l.bufA + (size_t)F * M;
    l.twM = l.bufB + (size_t)F * M;
    l.twN = l.twM + M;
    l.win = (float*)(l.twN + M);
    l.span = l.win + N;
    return l;
}

__device__ inline void init_tables(const MdctLds& l, int M, int N) {
    for (int j = threadIdx.x; j < M; j += MDCT_THREADS) {
        float s, c;
        sincospif(2.0f * (float)j / (float)M, &s, &c);
        l.twM[j] = make_float2(c, -s);
        sincospif((float)j / (float)N, &s, &c);
        l.twN[j] = make_float2(c, -s);
    }
    for (int n = threadIdx.x; n < N; n += MDCT_THREADS)
        l.win[n] = sinpif(((float)n + 0.5f) / (float)(2 * N));
}

// Batched Stockham autosort FFT (radix 4, one radix-2 stage when log2 M is
// odd) over nfr transforms of length M stored contiguously.  Returns the
// buffer that holds the result.
__device__ inline float2* fft_batch(float2* x, float2* y, const float2* twM, int nfr, int M) {
    for (int Ns = 1; Ns < M;) {
        const int R = (Ns * 4 <= M) ? 4 : 2;
        const int q = M / R;           // butterflies per transform
        const int tws = M / (Ns * R);  // twiddle stride in the period-M table
        const int total = nfr * q;
        for (int idx = threadIdx.x; idx < total; idx += MDCT_THREADS) {
            const int fr = idx / q, j = idx - fr * q;
            const int k = j & (Ns - 1);
            const float2* xi = x + (size_t)fr * M;
            float2* yo = y + (size_t)fr * M;
            const int j0 = (j - k) * R + k;
            if (R == 4) {
                float2 v0 = xi[j];
                float2 v1 = cmul(xi[j + q], twM[k * tws]);
                float2 v2 = cmul(xi[j + 2 * q], twM[2 * k * tws]);
                float2 v3 = cmul(xi[j + 3 * q], twM[3 * k * tws]);
                float2 a0 = make_float2(v0.x + v2.x, v0.y + v2.y);
                float2 a1 = make_float2(v0.x - v2.x, v0.y - v2.y);
                float2 a2 = make_float2(v1.x + v3.x, v1.y + v3.y);
                // (v1 - v3) * (-i) = (d.y, -d.x)
                float2 a3 = make_float2(v1.y - v3.y, -(v1.x - v3.x));
                yo[j0] = make_float2(a0.x + a2.x, a0.y + a2.y);
                yo[j0 + Ns] = make_float2(a1.x + a3.x, a1.y + a3.y);
                yo[j0 + 2 * Ns] = make_float2(a0.x - a2.x, a0.y - a2.y);
                yo[j0 + 3 * Ns] = make_float2(a1.x - a3.x, a1.y - a3.y);
            } else {
                float2 v0 = xi[j];
                float2 v1 = cmul(xi[j + q], twM[k * tws]);
                yo[j0] = make_float2(v0.x + v1.x, v0.y + v1.y);
                yo[j0 + Ns] = make_float2(v0.x - v1.x, v0.y - v1.y);
            }
        }
        __syncthreads();
        float2* t = x; x = y; y = t;
        Ns *= R;
    }
    return x;
}

__device__ inline float win_at(const float* win, int n, int N) {
    return n < N ? win[n] : win[2 * N - 1 - n];
}

// ---------------------------------------------------------------------
// forward: one workgroup = F consecutive frames of one clip
// ---------------------------------------------------------------------
__global__ void __launch_bounds__(MDCT_THREADS)
mdct_fwd_fft_kernel(const float* __restrict__ x, int64_t B, int64_t T, int64_t ldx, int N, int hop,
                    int64_t nf, int F, int64_t groups_per_clip, int64_t n_groups,
                    float* __restrict__ X) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int M = N / 2;
    MdctLds l = carve(smem, F, M, N);
    init_tables(l, M, N);
    const float2 c4 = [&] { float s, c; sincospif(0.25f / (float)N, &s, &c); return make_float2(c, -s); }();
    __syncthreads();

    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int64_t b = g / groups_per_clip;
        const int64_t i0 = (g - b * groups_per_clip) * F;
        const int nfr = (int)((nf - i0) < F ? (nf - i0) : F);
        const float* xb = x + b * ldx;
        const int64_t s0 = i0 * hop;
        const int span = (nfr - 1) * hop + 2 * N;
        // stage the sample span once (coalesced), zero beyond T
        for (int s = threadIdx.x; s < span; s += MDCT_THREADS) {
            const int64_t p = s0 + s;
            l.span[s] = p < T ? xb[p] : 0.0f;
        }
        __syncthreads();
        // window + fold + pre-twiddle
        for (int idx = threadIdx.x; idx < nfr * M; idx += MDCT_THREADS) {
            const int fr = idx / M, m = idx - fr * M;
            const float* f = l.span + fr * hop;
            float u[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int n = h == 0 ? 2 * m : N - 1 - 2 * m;
                const int i1 = 3 * N / 2 - 1 - n;
                if (n < N / 2) {
                    const int i2 = 3 * N / 2 + n;
                    u[h] = -f[i1] * win_at(l.win, i1, N) - f[i2] * win_at(l.win, i2, N);
                } else {
                    const int i2 = n - N / 2;
                    u[h] = f[i2] * win_at(l.win, i2, N) - f[i1] * win_at(l.win, i1, N);
                }
            }
            l.bufA[idx] = cmul(make_float2(u[0], u[1]), l.twN[m]);
        }
        __syncthreads();
        float2* Z = fft_batch(l.bufA, l.bufB, l.twM, nfr, M);
        float* rows = (float*)(Z == l.bufA ? l.bufB : l.bufA);  // [nfr][N] real
        for (int idx = threadIdx.x; idx < nfr * M; idx += MDCT_THREADS) {
            const int fr = idx / M, k = idx - fr * M;
            float2 W = cmul(cmul(Z[idx], l.twN[k]), c4);
            rows[fr * N + 2 * k] = W.x;
            rows[fr * N + N - 1 - 2 * k] = -W.y;
        }
        __syncthreads();
        float* Xo = X + (b * nf + i0) * (int64_t)N;
        for (int idx = threadIdx.x; idx < nfr * N; idx += MDCT_THREADS) Xo[idx] = rows[idx];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------
// inverse: one workgroup = SP consecutive OUTPUT samples of one clip
// ---------------------------------------------------------------------
template <int OPT>
__global__ void __launch_bounds__(MDCT_THREADS)
mdct_inv_fft_kernel(const float* __restrict__ X, int64_t B, int64_t nf, int N, int hop,
                    int64_t out_len, int FC, int64_t spans_per_clip, int64_t n_spans,
                    float* __restrict__ y, int64_t ldy) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int M = N / 2;
    const int SP = MDCT_THREADS * OPT;
    MdctLds l = carve(smem, FC, M, N);
    float* U = l.span;  // [FC][N]
    init_tables(l, M, N);
    const float2 c4 = [&] { float s, c; sincospif(0.25f / (float)N, &s, &c); return make_float2(c, -s); }();
    const float scale = 2.0f / (float)N;
    __syncthreads();

    for (int64_t g = blockIdx.x; g < n_spans; g += gridDim.x) {
        const int64_t b = g / spans_per_clip;
        const int64_t p0 = (g - b * spans_per_clip) * SP;
        const int64_t pend = (p0 + SP < out_len ? p0 + SP : out_len);  // exclusive
        // frames i with i*hop <= p < i*hop + 2N for some p in [p0, pend)
        int64_t ilo = p0 - 2 * N + 1;
        ilo = ilo <= 0 ? 0 : (ilo + hop - 1) / hop;
        int64_t ihi = (pend - 1) / hop;
        if (ihi > nf - 1) ihi = nf - 1;
        float acc[OPT];
#pragma unroll
        for (int j = 0; j < OPT; ++j) acc[j] = 0.0f;

        for (int64_t c0 = ilo; c0 <= ihi; c0 += FC) {
            const int nfr = (int)((ihi - c0 + 1) < FC ? (ihi - c0 + 1) : FC);
            const float* Xi = X + (b * nf + c0) * (int64_t)N;
            // load + pre-twiddle: z[m] = (X[2m] + i X[N-1-2m]) e^{-i pi m/N}
            // stage rows through U for coalesced reads
            for (int idx = threadIdx.x; idx < nfr * N; idx += MDCT_THREADS) U[idx] = Xi[idx];
            __syncthreads();
            for (int idx = threadIdx.x; idx < nfr * M; idx += MDCT_THREADS) {
                const int fr = idx / M, m = idx - fr * M;
                l.bufA[idx] = cmul(make_float2(U[fr * N + 2 * m], U[fr * N + N - 1 - 2 * m]), l.twN[m]);
            }
            __syncthreads();
            float2* Z = fft_batch(l.bufA, l.bufB, l.twM, nfr, M);
            for (int idx = threadIdx.x; idx < nfr * M; idx += MDCT_THREADS) {
                const int fr = idx / M, k = idx - fr * M;
                float2 W = cmul(cmul(Z[idx], l.twN[k]), c4);
                U[fr * N + 2 * k] = W.x * scale;
                U[fr * N + N - 1 - 2 * k] = -W.y * scale;
            }
            __syncthreads();
            // gather: ascending frame order per output sample
#pragma unroll
            for (int j = 0; j < OPT; ++j) {
                const int64_t p = p0 + threadIdx.x + (int64_t)j * MDCT_THREADS;
                if (p < pend) {
                    int64_t a = p - 2 * N + 1;
                    a = a <= 0 ? 0 : (a + hop - 1) / hop;
                    int64_t e = p / hop;
                    if (a < c0) a = c0;
                    if (e > c0 + nfr - 1) e = c0 + nfr - 1;
                    float s = acc[j];
                    for (int64_t i = a; i <= e; ++i) {
                        const int n = (int)(p - i * hop);
                        const float* u = U + (i - c0) * N;
                        float v;
                        if (n < N / 2) v = u[N / 2 + n];
                        else if (n < 3 * N / 2) v = -u[3 * N / 2 - 1 - n];
                        else v = -u[n - 3 * N / 2];
                        s += win_at(l.win, n, N) * v;
                    }
                    acc[j] = s;
                }
            }
            __syncthreads();
        }
        float* yb = y + b * ldy;
#pragma unroll
        for (int j = 0; j < OPT; ++j) {
            const int64_t p = p0 + threadIdx.x + (int64_t)j * MDCT_THREADS;
            if (p < pend) yb[p] = acc[j];
        }
    }
}

// ---------------------------------------------------------------------
// direct-basis fallback for window sizes that are not a power of two
// (the reference's default window is 576).  Exact integer angle reduction:
// pi/N (n+N/2+1/2)(k+1/2) = pi p/(4N), p = (2n+N+1)(2k+1) mod 8N.
// ---------------------------------------------------------------------
__device__ inline float basis_cos(int n, int k, int N) {
    const int64_t p = ((int64_t)(2 * n + N + 1) * (int64_t)(2 * k + 1)) % (int64_t)(8 * N);
    return cospif((float)p / (float)(4 * N));
}

__global__ void __launch_bounds__(MDCT_THREADS)
mdct_fwd_direct_kernel(const float* __restrict__ x, int64_t B, int64_t T, int64_t ldx, int N, int hop,
                       int64_t nf, float* __restrict__ X) {
    const int64_t total = B * nf * N;
    for (int64_t o = blockIdx.x * (int64_t)MDCT_THREADS + threadIdx.x; o < total;
         o += (int64_t)gridDim.x * MDCT_THREADS) {
        const int k = (int)(o % N);
        const int64_t i = (o / N) % nf;
        const int64_t b = o / ((int64_t)N * nf);
        const float* xb = x + b * ldx;
        float acc = 0.0f;
        for (int n = 0; n < 2 * N; ++n) {
            const int64_t p = i * hop + n;
            if (p >= T) break;
            const float w = sinpif(((float)n + 0.5f) / (float)(2 * N));
            acc += w * xb[p] * basis_cos(n, k, N);
        }
        X[o] = acc;
    }
}

__global__ void __launch_bounds__(MDCT_THREADS)
mdct_inv_direct_kernel(const float* __restrict__ X, int64_t B, int64_t nf, int N, int hop,
                       int64_t out_len, float* __restrict__ y, int64_t ldy) {
    const int64_t total = B * out_len;
    const float scale = 2.0f / (float)N;
    for (int64_t o = blockIdx.x * (int64_t)MDCT_THREADS + threadIdx.x; o < total;
         o += (int64_t)gridDim.x * MDCT_THREADS) {
        const int64_t p = o % out_len, b = o / out_len;
        int64_t a = p - 2 * N + 1;
        a = a <= 0 ? 0 : (a + hop - 1) / hop;
        int64_t e = p / hop;
        if (e > nf - 1) e = nf - 1;
        float s = 0.0f;
        for (int64_t i = a; i <= e; ++i) {
            const int n = (int)(p - i * hop);
            const float* Xi = X + (b * nf + i) * (int64_t)N;
            float acc = 0.0f;
            for (int k = 0; k < N; ++k) acc += Xi[k] * basis_cos(n, k, N);
            s += scale * sinpif(((float)n + 0.5f) / (float)(2 * N)) * acc;
        }
        y[b * ldy + p] = s;
    }
}

inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
inline bool generic_only() {      // MFC_MDCT_GENERIC=1: skip the N = 512 kernels (A/B and cross-check hook)
    static const bool on = [] { const char* e = getenv("MFC_MDCT_GENERIC"); return e && e[0] == '1'; }();
    return on;
}

const size_t MDCT_LDS_CAP = 150 * 1024;

size_t lds_bytes(int F, int N, size_t span_floats) {
    const size_t M = N / 2;
    return (2 * (size_t)F * M + 2 * M) * sizeof(float2) + ((size_t)N + span_floats) * sizeof(float);
}

}  // namespace

extern "C" int64_t mfc_mdct_num_frames(int64_t T, int N, int hop) {
    if (T < 0 || N <= 0 || hop <= 0) return MFC_EINVAL;
    return T < N ? 1 : (T - N) / hop + 1;
}

extern "C" int64_t mfc_mdct_out_len(int64_t n_frames, int N, int hop) {
    if (n_frames <= 0 || N <= 0 || hop <= 0) return MFC_EINVAL;
    return (n_frames - 1) * hop + 2 * (int64_t)N;
}

extern "C" int mfc_mdct_fwd(const float* x, int64_t B, int64_t T, int64_t ldx, int N, int hop,
                            float* X, void* stream) {
    if (!x || !X) return MFC_EFAULT;
    if (B <= 0 || T <= 0 || N <= 0 || hop <= 0 || ldx < T) return MFC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int64_t nf = mfc_mdct_num_frames(T, N, hop);
    if (N == 512 && !generic_only()) {
        const int rc = mfc_mdct512_fwd(x, B, T, ldx, hop, nf, X, st);
        if (rc != MFC_ENOSYS) return rc;
    }
    if (is_pow2(N) && N >= 8) {
        // frames per workgroup: ~2048 complex points, span must fit LDS
        int F = 4096 / N;
        if (F < 1) F = 1;
        if (F > 16) F = 16;
        if (F > nf) F = (int)nf;
        while (F > 1 && lds_bytes(F, N, (size_t)(F - 1) * hop + 2 * N) > MDCT_LDS_CAP) --F;
        const size_t lds = lds_bytes(F, N, (size_t)(F - 1) * hop + 2 * N);
        if (lds <= MDCT_LDS_CAP) {
            const int64_t gpc = ceil_div64(nf, F);
            const int64_t n_groups = gpc * B;
            const int64_t grid = n_groups < 8192 ? n_groups : 8192;
            (void)hipFuncSetAttribute((const void*)mdct_fwd_fft_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(mdct_fwd_fft_kernel, dim3((unsigned)grid), dim3(MDCT_THREADS), lds, st,
                               x, B, T, ldx, N, hop, nf, F, gpc, n_groups, X);
            return mfc_launch_status();
        }
    }
    const int64_t total = B * nf * N;
    int64_t grid = ceil_div64(total, MDCT_THREADS);
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(mdct_fwd_direct_kernel, dim3((unsigned)grid), dim3(MDCT_THREADS), 0, st,
                       x, B, T, ldx, N, hop, nf, X);
    return mfc_launch_status();
}

extern "C" int mfc_mdct_inv(const float* X, int64_t B, int64_t n_frames, int N, int hop,
                            float* y, int64_t ldy, void* stream) {
    if (!X || !y) return MFC_EFAULT;
    if (B <= 0 || n_frames <= 0 || N <= 0 || hop <= 0) return MFC_EINVAL;
    const int64_t out_len = mfc_mdct_out_len(n_frames, N, hop);
    if (ldy < out_len) return MFC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (N == 512 && !generic_only()) {
        const int rc = mfc_mdct512_inv(X, B, n_frames, hop, out_len, y, ldy, st);
        if (rc != MFC_ENOSYS) return rc;
    }
    if (is_pow2(N) && N >= 8) {
        constexpr int OPT = 4;
        const int SP = MDCT_THREADS * OPT;
        // frames that can touch one span
        int64_t need = (SP + 2 * (int64_t)N - 2) / hop + 1;
        if (need > n_frames) need = n_frames;
        int FC = (int)need;
        int capF = 8192 / N;  // ~4096 complex points per chunk
        if (capF < 1) capF = 1;
        if (FC > capF) FC = capF;
        while (FC > 1 && lds_bytes(FC, N, (size_t)FC * N) > MDCT_LDS_CAP) --FC;
        const size_t lds = lds_bytes(FC, N, (size_t)FC * N);
        if (lds <= MDCT_LDS_CAP) {
            const int64_t spc = ceil_div64(out_len, SP);
            const int64_t n_spans = spc * B;
            const int64_t grid = n_spans < 8192 ? n_spans : 8192;
            (void)hipFuncSetAttribute((const void*)mdct_inv_fft_kernel<OPT>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(mdct_inv_fft_kernel<OPT>, dim3((unsigned)grid), dim3(MDCT_THREADS), lds, st,
                               X, B, n_frames, N, hop, out_len, FC, spc, n_spans, y, ldy);
            return mfc_launch_status();
        }
    }
    const int64_t total = B * out_len;
    int64_t grid = ceil_div64(total, MDCT_THREADS);
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(mdct_inv_direct_kernel, dim3((unsigned)grid), dim3(MDCT_THREADS), 0, st,
                       X, B, n_frames, N, hop, out_len, y, ldy);
    return mfc_launch_status();
}
