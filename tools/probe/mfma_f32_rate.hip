// probe: what the matrix pipe delivers for v_mfma_f32_16x16x4_f32 and v_mfma_f32_32x32x2_f32 from registers alone (no LDS, no
// memory) with NACC independent accumulators per wave and 1 / 2 / 4 waves per SIMD -- the ceiling of any fp32 GEMM here.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_f32_rate.hip -o /tmp/mfma_f32_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-3f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void __launch_bounds__(256) k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static void run(const char* name, K kern, int nacc, double flop_per_mfma, int wgs_per_cu, float* out) {
    const int iters = 2000, grid = 256 * wgs_per_cu;
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 10, 1.0f, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(s);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 1.0f);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    const double flops = (double)grid * 4 /*waves*/ * iters * 4.0 * nacc * flop_per_mfma;
    printf("%s NACC=%d waves/SIMD=%d: %.3f ms  %.1f TFLOP/s (%.3f of 157.3)\n", name, nacc, wgs_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 4}) {
        run("16x16x4", k16<1>, 1, 2048.0, w, out);
        run("16x16x4", k16<4>, 4, 2048.0, w, out);
        run("16x16x4", k16<16>, 16, 2048.0, w, out);
        run("32x32x2", k32<1>, 1, 4096.0, w, out);
        run("32x32x2", k32<4>, 4, 4096.0, w, out);
    }
    hipFree(out);
    return 0;
}
