// probe / reproducer: dependent accumulator chains that MIX MFMA shapes on gfx950 (ROCm 7.2 hipcc).
//
// Round 1 observed wrong rows when a v_mfma_f32_16x16x32_bf16 result was consumed as SrcC of a v_mfma_f32_16x16x16_bf16
// (csrc/convnext.hip border-tile path) and worked around it (tap 8 zero-padded to K = 32; `s_nop 15; s_nop 15` on the
// K16 -> K32 direction).  This program isolates the pattern: exact small-integer operands (every product and sum is
// exact in bf16 / fp32, so any deviation is a hazard, not rounding), chains of both orders, with and without explicit
// wait states, built with the library's flags (-mllvm -amdgpu-mfma-vgpr-form=1 matters: MFMA results land in VGPRs).
//
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/probe/mfma_mixed_shape.hip -o /tmp/mfma_mixed && /tmp/mfma_mixed
//   (add -save-temps and read the s_nop between the dependent v_mfma pairs of kernel `chain`)
//
// Result of the run on MI355X and the ISA reading are recorded in DESIGN.md ("Toolchain hazard").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline short bf(float v) { return __builtin_bit_cast(short, (__bf16)v); }

// A32 [16][32], B32 [32][16] (K = 32 step), A16 [16][16], B16 [16][16] (K = 16 step); D [16][16]
// MODE 0: K32 then K16 (K32 result is SrcC of the K16);  1: K16 then K32;  2/3: the same with s_nop 15 x2 between, tied to acc
// REP: the pair is repeated REP times on the same accumulator (as the 9-tap border chain + 5 K32 steps do)
// MODE 4: alternating chain with `s_nop NA` after every K32 (before the K16 that consumes it) and `s_nop NB` after every
// K16 (before the K32 that consumes it); NA / NB = -1: no instruction.  Finds the wait states each direction needs.
template <int NA, int NB, int REP>
__global__ void chain_nops(const float* A32, const float* B32, const float* A16, const float* B16, float* D) {
    const int lane = threadIdx.x & 63, q = lane >> 4, r = lane & 15;
    bf16x8 a32, b32;
    s16x4 a16, b16;
    for (int i = 0; i < 8; ++i) { a32[i] = (__bf16)A32[r * 32 + 8 * q + i]; b32[i] = (__bf16)B32[(8 * q + i) * 16 + r]; }
    for (int i = 0; i < 4; ++i) { a16[i] = bf(A16[r * 16 + 4 * q + i]); b16[i] = bf(B16[(4 * q + i) * 16 + r]); }
    f32x4 acc = {1.f, 2.f, 3.f, 4.f};
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc), "+v"(a32), "+v"(b32), "+v"(a16), "+v"(b16));   // operands settled
#pragma unroll
    for (int rep = 0; rep < REP; ++rep) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a32, b32, acc, 0, 0, 0);
        if constexpr (NA >= 0) asm volatile("s_nop %1" : "+v"(acc) : "n"(NA));
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a16, b16, acc, 0, 0, 0);
        if constexpr (NB >= 0) asm volatile("s_nop %1" : "+v"(acc) : "n"(NB));
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc));
    for (int e = 0; e < 4; ++e) D[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 256 + (4 * q + e) * 16 + r] = acc[e];
}
template <int NA, int NB>
long run_nops(const float* dA32, const float* dB32, const float* dA16, const float* dB16, float* dD, const std::vector<float>& ref) {
    const int blocks = 1024, waves = 4;
    hipLaunchKernelGGL((chain_nops<NA, NB, 5>), dim3(blocks), dim3(64 * waves), 0, 0, dA32, dB32, dA16, dB16, dD);
    std::vector<float> D((size_t)blocks * waves * 256);
    (void)hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (size_t i = 0; i < D.size(); ++i) if (D[i] != ref[i & 255]) ++bad;
    return bad;
}

template <int MODE, int REP>
__global__ void chain(const float* A32, const float* B32, const float* A16, const float* B16, float* D) {
    const int lane = threadIdx.x & 63, q = lane >> 4, r = lane & 15;
    bf16x8 a32, b32;
    s16x4 a16, b16;
    for (int i = 0; i < 8; ++i) { a32[i] = (__bf16)A32[r * 32 + 8 * q + i]; b32[i] = (__bf16)B32[(8 * q + i) * 16 + r]; }
    for (int i = 0; i < 4; ++i) { a16[i] = bf(A16[r * 16 + 4 * q + i]); b16[i] = bf(B16[(4 * q + i) * 16 + r]); }
    f32x4 acc = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
    for (int rep = 0; rep < REP; ++rep) {
        if (MODE == 0 || MODE == 2) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a32, b32, acc, 0, 0, 0);
            if (MODE == 2) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc));
            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a16, b16, acc, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a16, b16, acc, 0, 0, 0);
            if (MODE == 3) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a32, b32, acc, 0, 0, 0);
        }
    }
    for (int e = 0; e < 4; ++e) D[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 256 + (4 * q + e) * 16 + r] = acc[e];
}

template <int MODE, int REP>
int run(const char* name, const float* dA32, const float* dB32, const float* dA16, const float* dB16, float* dD,
        const std::vector<float>& ref) {
    const int blocks = 1024, waves = 4;
    hipLaunchKernelGGL((chain<MODE, REP>), dim3(blocks), dim3(64 * waves), 0, 0, dA32, dB32, dA16, dB16, dD);
    std::vector<float> D((size_t)blocks * waves * 256);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (size_t i = 0; i < D.size(); ++i) if (D[i] != ref[i & 255]) ++bad;
    printf("%-34s rep=%d  wrong elements: %ld of %zu\n", name, REP, bad, D.size());
    return bad != 0;
}

int main() {
    std::vector<float> A32(512), B32(512), A16(256), B16(256);
    for (int i = 0; i < 512; ++i) { A32[i] = (float)((i * 7) % 5 - 2); B32[i] = (float)((i * 5) % 3 - 1); }
    for (int i = 0; i < 256; ++i) { A16[i] = (float)((i * 3) % 5 - 2); B16[i] = (float)((i * 11) % 3 - 1); }
    auto reference = [&](int mode, int rep) {
        std::vector<float> R(256);
        for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
            float p32 = 0, p16 = 0;
            for (int k = 0; k < 32; ++k) p32 += A32[m * 32 + k] * B32[k * 16 + n];
            for (int k = 0; k < 16; ++k) p16 += A16[m * 16 + k] * B16[k * 16 + n];
            R[m * 16 + n] = (float)(m % 4 + 1) + rep * (p32 + p16);     // initial acc = reg index + 1 = row % 4 + 1
        }
        return R;
    };
    float *dA32, *dB32, *dA16, *dB16, *dD;
    hipMalloc(&dA32, 2048); hipMalloc(&dB32, 2048); hipMalloc(&dA16, 1024); hipMalloc(&dB16, 1024);
    hipMalloc(&dD, (size_t)1024 * 4 * 256 * 4);
    hipMemcpy(dA32, A32.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB32, B32.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(dA16, A16.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB16, B16.data(), 1024, hipMemcpyHostToDevice);
    int bad = 0;
    bad |= run<0, 1>("K32 -> K16 (SrcC = K32 result)", dA32, dB32, dA16, dB16, dD, reference(0, 1));
    bad |= run<1, 1>("K16 -> K32 (SrcC = K16 result)", dA32, dB32, dA16, dB16, dD, reference(1, 1));
    bad |= run<0, 5>("K32 -> K16 alternating", dA32, dB32, dA16, dB16, dD, reference(0, 5));
    bad |= run<1, 5>("K16 -> K32 alternating", dA32, dB32, dA16, dB16, dD, reference(1, 5));
    bad |= run<2, 5>("K32 -> s_nop -> K16 alternating", dA32, dB32, dA16, dB16, dD, reference(0, 5));
    bad |= run<3, 5>("K16 -> s_nop -> K32 alternating", dA32, dB32, dA16, dB16, dD, reference(1, 5));
    printf(bad ? "MIXED-SHAPE HAZARD REPRODUCED\n" : "all mixed-shape chains exact\n");
    // wait-state sweep: s_nop N = N + 1 wait states.  Row "K32->K16 N": N after each K32, 15 after each K16 (that link safe).
    const std::vector<float> ref5 = reference(0, 5);
    printf("wait-state sweep (wrong elements of %d); s_nop N inserts N+1 wait states\n", 1024 * 4 * 256);
#define SWEEP(N) printf("  s_nop %2d:  after K32 (before K16): %8ld   after K16 (before K32): %8ld   both: %8ld\n", N, \
        run_nops<N, 15>(dA32, dB32, dA16, dB16, dD, ref5), run_nops<15, N>(dA32, dB32, dA16, dB16, dD, ref5), \
        run_nops<N, N>(dA32, dB32, dA16, dB16, dD, ref5));
    printf("  none    :  after K32 (before K16): %8ld   after K16 (before K32): %8ld   both: %8ld\n",
           run_nops<-1, 15>(dA32, dB32, dA16, dB16, dD, ref5), run_nops<15, -1>(dA32, dB32, dA16, dB16, dD, ref5),
           run_nops<-1, -1>(dA32, dB32, dA16, dB16, dD, ref5));
    SWEEP(0) SWEEP(1) SWEEP(2) SWEEP(3) SWEEP(4) SWEEP(5) SWEEP(6) SWEEP(7) SWEEP(8) SWEEP(9) SWEEP(10) SWEEP(11) SWEEP(12) SWEEP(15)
    return bad;
}
