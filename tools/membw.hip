// HBM bandwidth probe (gfx950): read-only, write-only and copy streams with 16-byte accesses, one-shot and grid-stride forms.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/membw tools/membw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_read(const u32x4 *src, u32x4 *sink, size_t n, int per_thread)
{
    size_t i = ((size_t)blockIdx.x * per_thread) * 256 + threadIdx.x;
    u32x4 acc = { 0, 0, 0, 0 };
    for (int k = 0; k < per_thread; k++, i += 256) if (i < n) { u32x4 v = src[i]; acc += v; }
    if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) sink[0] = acc;   // never true in practice: keeps the loads alive
}
__global__ void __launch_bounds__(256) k_write(u32x4 *dst, size_t n, int per_thread)
{
    size_t i = ((size_t)blockIdx.x * per_thread) * 256 + threadIdx.x;
    const u32x4 v = { (uint32_t)i, 1, 2, 3 };
    for (int k = 0; k < per_thread; k++, i += 256) if (i < n) __builtin_nontemporal_store(v, &dst[i]);
}
__global__ void __launch_bounds__(256) k_copy(const u32x4 *src, u32x4 *dst, size_t n, int per_thread)
{
    size_t i = ((size_t)blockIdx.x * per_thread) * 256 + threadIdx.x;
    for (int k = 0; k < per_thread; k++, i += 256) if (i < n) __builtin_nontemporal_store(src[i], &dst[i]);
}
int main()
{
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    u32x4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pt : { 1, 2, 4, 8, 16, 64 }) {
        const unsigned blocks = (unsigned)((n + (size_t)256 * pt - 1) / ((size_t)256 * pt));
        float ms[3];
        for (int which = 0; which < 3; which++) {
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (which == 0) hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, b, n, pt);
                else if (which == 1) hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, b, n, pt);
                else hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n, pt);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[which], e0, e1);
            }
        }
        printf("16 B x %2d per thread: read %.0f GB/s  write %.0f GB/s  copy %.0f GB/s (read+write bytes)\n", pt,
               bytes / ms[0] / 1e6, bytes / ms[1] / 1e6, 2.0 * bytes / ms[2] / 1e6);
    }
    return 0;
}
