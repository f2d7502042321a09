// Semantics probe for buffer_load_dwordx4 ... lds (LDS-DMA) on gfx950: destination = M0 + lane * 16? exec-masked lanes? out-of-range lanes?
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/dma_probe tools/dma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint8_t *src, uint32_t bytes, uint32_t *out, int nlanes, int m0off)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    i32x4 rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(uintptr_t)src);
    rs[1] = __builtin_amdgcn_readfirstlane((int)((uintptr_t)src >> 32));
    rs[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    rs[3] = 0x00020000;
    const int voff = threadIdx.x * 16;
    const uint32_t ldsaddr = (uint32_t)(uintptr_t)lds + (uint32_t)__builtin_amdgcn_readfirstlane(m0off);
    if ((int)threadIdx.x < nlanes)
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(ldsaddr), "v"(voff), "s"(rs) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main()
{
    std::vector<uint32_t> h(512);
    for (int i = 0; i < 512; i++) h[i] = 0x1000 + i;
    uint8_t *d; uint32_t *o; hipMalloc(&d, 2048); hipMalloc(&o, 4096);
    hipMemcpy(d, h.data(), 2048, hipMemcpyHostToDevice);
    struct { int nl, m0, bytes; } cases[] = { { 64, 0, 2048 }, { 20, 0, 2048 }, { 64, 256, 2048 }, { 64, 0, 512 } };
    for (auto &c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, (uint32_t)c.bytes, o, c.nl, c.m0);
        std::vector<uint32_t> r(1024);
        hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
        printf("lanes %d m0 +%d bytes %d:", c.nl, c.m0, c.bytes);
        for (int i = 0; i < 1024; i += 4) {   // print the first dword of every 16-byte slot, compressed
            if (i % 64 == 0) printf("\n  slot %3d:", i / 4);
            printf(" %x", r[i]);
        }
        printf("\n");
    }
    return 0;
}
