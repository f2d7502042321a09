// Probe: does ds_read_b32 at a 2-byte-aligned (not 4-byte-aligned) LDS address return the straddling dword on gfx950, and what does it cost?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/lds_unaligned tools/lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void probe(uint32_t *out, int byte_off, int iters, uint32_t *sink)
{
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)lds + threadIdx.x * 4 + byte_off;   // LDS byte address
    uint32_t v, acc = 0;
    for (int k = 0; k < iters; k++) {
        asm volatile("ds_read_b32 %0, %1 offset:0\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr + ((k & 7) << 8)) : "memory");
        acc += v;
    }
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x + blockIdx.x * blockDim.x] = v;
    if (acc == 0x12345) sink[0] = acc;
}

int main()
{
    uint32_t *out, *sink;
    hipMalloc(&out, 1 << 20); hipMalloc(&sink, 64);
    std::vector<uint32_t> h(64);
    for (int off : { 0, 2 }) {
        probe<<<1, 64>>>(out, off, 1, sink);
        hipMemcpy(h.data(), out, 256, hipMemcpyDeviceToHost);
        printf("byte offset %d: lane0 %08x lane1 %08x lane63 %08x (expect lo=%d hi=%d for lane0)\n", off, h[0], h[1], h[63], off / 2, off / 2 + 1);
    }
    for (int off : { 0, 2 }) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        probe<<<1024, 256>>>(out, off, 4096, sink);
        hipEventRecord(a);
        probe<<<1024, 256>>>(out, off, 4096, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("byte offset %d: %.3f ms for 1024 blocks x 256 threads x 4096 dependent ds_read_b32\n", off, ms);
    }
    return 0;
}
