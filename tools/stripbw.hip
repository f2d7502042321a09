// Access-pattern probe for the marching strip kernels (gfx950): every wave walks down a column strip of a 2-D picture, one
// row pair per step, D pairs in flight; variants: register loads vs LDS-DMA, window bytes per row, pieces per row, waves per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/stripbw tools/stripbw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct Geo { int stride, rows, strips, win, band_rows, bands, frames; size_t frame_bytes; };

template <int D, bool DMA>
__global__ void __launch_bounds__(256) k_strip(const uint8_t *src, uint32_t *sink, Geo g)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wid = blockIdx.x * 4 + wib;
    if (wid >= g.strips * g.bands) return;
    const int strip = wid % g.strips, band = wid / g.strips;
    const uint8_t *base = src + (size_t)blockIdx.z * g.frame_bytes;
    const int chunks = (g.win + 15) / 16;
    const int parts = (chunks + 63) / 64;
    const int row_bytes = parts * 1024;
    uint8_t *ring = smem + wib * (D * 2 * row_bytes);
    const uint32_t lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)ring);
    i32x4 rs;
    const uint64_t a = (uint64_t)base;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a); rs[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    rs[2] = __builtin_amdgcn_readfirstlane((int)((uint32_t)g.stride * (uint32_t)g.rows)); rs[3] = 0x00020000;
    const int voff0 = strip * (g.win & ~15) + lane * 16;
    const int y0 = band * g.band_rows, y1 = min(g.rows, y0 + g.band_rows);
    u32x4 acc = { 0, 0, 0, 0 };
    if constexpr (DMA) {
        auto dma = [&](int q) {
            for (int r = 0; r < 2; r++) {
                const int row = min(y0 + 2 * q + r, g.rows - 1);
                const uint32_t dst = lds_base + (uint32_t)(((q & (D - 1)) * 2 + r) * row_bytes);
                for (int pt = 0; pt < parts; pt++) {
                    const int n = min(64, chunks - 64 * pt);
                    const uint32_t mlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n >= 32 ? 0xffffffffu : ((1u << n) - 1u)));
                    const uint32_t mhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n >= 64 ? 0xffffffffu : (n > 32 ? ((1u << (n - 32)) - 1u) : 0u)));
                    uint64_t keep;
                    asm volatile("s_nop 4\n\ts_mov_b64 %0, exec\n\ts_mov_b32 exec_lo, %5\n\ts_mov_b32 exec_hi, %6\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                                 "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b64 exec, %0"
                                 : "=&s"(keep) : "s"((uint32_t)__builtin_amdgcn_readfirstlane((int)(dst + 1024u * pt))), "v"(voff0 + 1024 * pt), "s"(rs), "s"(__builtin_amdgcn_readfirstlane(row * g.stride)), "s"(mlo), "s"(mhi) : "memory");
                }
            }
        };
        const int steps = (y1 - y0 + 1) / 2;
        for (int i = 0; i < D; i++) dma(i);
        for (int q = 0; q < steps; q++) {
            if (parts == 1) { if (D == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else if (D == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); }
            else            { if (D == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else if (D == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); }
            const u32x4 v = *(const u32x4 *)(ring + ((q & (D - 1)) * 2) * row_bytes + lane * 16);
            acc += v;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma(q + D);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        u32x4 pre[D][2][2];
        auto ld = [&](int q, int slot) {
            for (int r = 0; r < 2; r++) {
                const int row = min(y0 + 2 * q + r, g.rows - 1);
                for (int pt = 0; pt < 2; pt++) {
                    const bool on = pt < parts && 64 * pt + lane < chunks;
                    const uint8_t *p = base + (size_t)row * g.stride + voff0 + 1024 * pt;
                    if (on) pre[slot][r][pt] = *(const u32x4 *)p; else pre[slot][r][pt] = u32x4{ 0, 0, 0, 0 };
                }
            }
        };
        const int steps = (y1 - y0 + 1) / 2;
#pragma unroll
        for (int i = 0; i < D; i++) ld(i, i);
        for (int q0 = 0; q0 < steps; q0 += D) {
#pragma unroll
            for (int i = 0; i < D; i++) {
                acc += pre[i][0][0] + pre[i][0][1] + pre[i][1][0] + pre[i][1][1];
                ld(q0 + i + D, i);
            }
        }
    }
    if (acc[0] == 0x12345678u && acc[3] == 0x9abcdef0u) sink[0] = acc[1];
}

template <int D, bool DMA>
static void run(const char *name, const uint8_t *src, uint32_t *sink, Geo g, int waves_target)
{
    g.bands = waves_target / (g.strips * g.frames); if (g.bands < 1) g.bands = 1;
    g.band_rows = (g.rows + g.bands - 1) / g.bands; g.band_rows = (g.band_rows + 1) & ~1;
    g.bands = (g.rows + g.band_rows - 1) / g.band_rows;
    const int chunks = (g.win + 15) / 16, parts = (chunks + 63) / 64;
    const size_t lds = DMA ? (size_t)4 * D * 2 * parts * 1024 : 0;
    const dim3 grid((g.strips * g.bands + 3) / 4, 1, g.frames);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_strip<D, DMA>), grid, dim3(256), lds, 0, src, sink, g);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double bytes = (double)g.frames * g.rows * g.strips * g.win;
    printf("%-28s win %4d B x %2d strips, %5d waves (band %4d rows), LDS %3zu KB/block: %.3f ms  %.0f GB/s\n", name, g.win, g.strips,
           g.strips * g.bands * g.frames, g.band_rows, lds / 1024, best, bytes / best / 1e6);
}

int main()
{
    Geo g; g.stride = 15360; g.rows = 4320; g.frames = 8; g.frame_bytes = (size_t)g.stride * g.rows;
    uint8_t *src; uint32_t *sink; hipMalloc(&src, g.frame_bytes * g.frames + 4096); hipMalloc(&sink, 64); hipMemset(src, 1, g.frame_bytes * g.frames);
    for (int waves : { 4096, 8192 }) {
        for (int win : { 1024, 1056, 512, 2048 }) {
            g.win = win; g.strips = g.stride / (win & ~15);
            run<4, true>("dma D=4", src, sink, g, waves);
            run<2, true>("dma D=2", src, sink, g, waves);
            run<8, true>("dma D=8", src, sink, g, waves);
            run<2, false>("regs D=2", src, sink, g, waves);
            run<4, false>("regs D=4", src, sink, g, waves);
        }
    }
    return 0;
}
