// probe: semantics of global_load_lds_dwordx4 on gfx950 (per-lane global address, LDS dst = M0 base + lane*16)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* src, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned buf[4 * 64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lane L fetches global chunk (63 - L) of its wave's 1 KiB block -> LDS chunk L
    const unsigned* gp = src + wave * 256 + (63 - lane) * 4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                     (__attribute__((address_space(3))) void*)(buf + wave * 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) out[i] = buf[i];
}
int main() {
    std::vector<unsigned> h(1024), o(1024);
    for (int i = 0; i < 1024; ++i) h[i] = i;
    unsigned *d, *e;
    hipMalloc(&d, 4096); hipMalloc(&e, 4096);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, e);
    hipMemcpy(o.data(), e, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 4; ++w) for (int L = 0; L < 64; ++L) for (int j = 0; j < 4; ++j) {
        unsigned want = w * 256 + (63 - L) * 4 + j;
        if (o[w * 256 + L * 4 + j] != want) { if (bad < 8) printf("w%d L%d j%d got %u want %u\n", w, L, j, o[w*256+L*4+j], want); ++bad; }
    }
    printf("bad=%d\n", bad);
    return bad != 0;
}
