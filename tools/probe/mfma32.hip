// probe: k-index layout of v_mfma_f32_16x16x32_bf16 on gfx950 (assumed: lane (q, r) holds k = 8q .. 8q+7 of row/col r)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D) {   // A [16][32], B [32][16], D [16][16]
    const int lane = threadIdx.x, q = lane >> 4, r = lane & 15;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)A[r * 32 + 8 * q + i]; b[i] = (__bf16)B[(8 * q + i) * 16 + r]; }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    for (int e = 0; e < 4; ++e) D[(4 * q + e) * 16 + r] = acc[e];
}
int main() {
    std::vector<float> A(512), B(512), D(256), R(256, 0.f);
    for (int i = 0; i < 512; ++i) { A[i] = (float)((i * 7) % 13 - 6); B[i] = (float)((i * 5) % 11 - 5); }
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) for (int kk = 0; kk < 32; ++kk) R[m * 16 + n] += A[m * 32 + kk] * B[kk * 16 + n];
    float *dA, *dB, *dD;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) if (D[i] != R[i]) ++bad;
    printf("bad=%d\n", bad);
    return bad != 0;
}
