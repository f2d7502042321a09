// Translation unit of the packed-RGB LUT writers behind identity horizontal filters (C2b, C4): hScale8To15_c with one tap +
// packed_vscale + yuv2rgb_{1,2,X}_c_template for 8-bit planar / nv12 sources (output.c:1788-1939, vscale.c:109-171).
// The kernels are fully unrolled per (bytes per pixel, source layout): this file is compiled once per part
// (-DRGBU_KIND=0 march / 1 one-shot wave + per-thread fallback, -DRGBU_BPP=3|4, -DRGBU_NV=0|1) so that the parts build in parallel;
// without the macros it compiles the dispatcher.
#include <algorithm>

#include "devstate.hpp"

namespace swship {
int launch_rgbu_march_b3nv0(const LaunchCtx &L); int launch_rgbu_march_b3nv1(const LaunchCtx &L);
int launch_rgbu_march_b4nv0(const LaunchCtx &L); int launch_rgbu_march_b4nv1(const LaunchCtx &L);
int launch_rgbu_wave_b3nv0(const LaunchCtx &L);  int launch_rgbu_wave_b3nv1(const LaunchCtx &L);
int launch_rgbu_wave_b4nv0(const LaunchCtx &L);  int launch_rgbu_wave_b4nv1(const LaunchCtx &L);
}

#ifndef RGBU_KIND
namespace swship {

int launch_rgb_unity(const LaunchCtx &L)
{
    SwsInternal *c = L.c; DeviceState *d = L.d; const SwsDevParams &p = *L.p;
    const bool nv = p.srcKind == SRCK_NV12, b4 = p.dstKind == DSTK_RGB32;
    const bool wave_ok = L.vec && !c->tune.no_wave && d->all_x_mode && d->chr_window2 <= 8 && p.vChrFs <= 64;
    // (rgb_march_ok covers rows of yuv2rgb_1's chroma blend as well: the X arithmetic, dev_plan*.hip)
    const bool march_ok = L.vec && !c->tune.no_wave && d->rgb_march_ok && d->chr_window2 <= 8 && p.vChrFs <= 64;
    if (march_ok && !c->tune.no_march && frames_desc_ok(L.frames, L.n, p.srcH, p.dstH))
        return b4 ? (nv ? launch_rgbu_march_b4nv1(L) : launch_rgbu_march_b4nv0(L)) : (nv ? launch_rgbu_march_b3nv1(L) : launch_rgbu_march_b3nv0(L));
    return b4 ? (nv ? launch_rgbu_wave_b4nv1(L) : launch_rgbu_wave_b4nv0(L)) : (nv ? launch_rgbu_wave_b3nv1(L) : launch_rgbu_wave_b3nv0(L));
}

} // namespace swship
#else

#if RGBU_KIND == 0
#include "kernels_rgbmarch.hpp"
#else
#include "kernels_fast.hpp"
#include "kernels_wave.hpp"
#endif

#define RGBU_CAT2(a, b, c) a##b##nv##c
#define RGBU_CAT(a, b, c) RGBU_CAT2(a, b, c)

namespace swship {

#if RGBU_KIND == 0
// marching kernel: a wave owns a 1024-pixel column strip and walks down a band of output-row pairs
int RGBU_CAT(launch_rgbu_march_b, RGBU_BPP, RGBU_NV)(const LaunchCtx &L)
{
    SwsInternal *c = L.c; DeviceState *d = L.d; const SwsDevParams &p = *L.p; hipStream_t st = L.st; const SwsFrameSet &fs = L.fs;
    const int n = L.n;
    const dim3 blk(256);
    constexpr int B = RGBU_BPP;
    constexpr bool N = RGBU_NV != 0;
    const int segs = (p.dstW + 1023) >> 10;
    const bool swap = B == 4 ? p.lut.swap_rb32 != 0 : p.lut.rgb_order != 0;
    const bool afirst = B == 4 && p.lut.perm32 == 0x02010003u;
    const int ncr = d->rgb_ncr;                 // ring rows: 5 (+ the rounding constant in the sixth slot), 6 or 8
    // one resident round: 4 waves per SIMD on 1024 SIMDs; bands of at least 8 row pairs
    const int target = c->tune.rgb_march_waves;
    const int groups = d->rgb_groups;
    int bands = std::max(1, std::min(target / std::max(1, segs * n), (groups + 7) / 8));
    int band_groups = (groups + bands - 1) / bands;
    bands = (groups + band_groups - 1) / band_groups;
    const dim3 gm(cdiv((int64_t)segs * bands, 4), 1, n);
    const SwsRgbGroupPlan *plan = (const SwsRgbGroupPlan *)d->d_rgbplan;
#ifdef SWS_HIP_PROFILING
    if constexpr (B == 3 && !N) {
        const int mexp = c->tune.debug;
        if (mexp && !swap && ncr == 5) {     // profiling experiments on the C2b instantiation only (results are wrong)
            if (mexp == 1) hipLaunchKernelGGL((swsk::sws_k_rgb_march<3, false, false, false, 5, 1>), gm, blk, 0, st, fs, p, plan, groups, bands, band_groups);
            if (mexp == 2) hipLaunchKernelGGL((swsk::sws_k_rgb_march<3, false, false, false, 5, 2>), gm, blk, 0, st, fs, p, plan, groups, bands, band_groups);
            if (mexp == 3) hipLaunchKernelGGL((swsk::sws_k_rgb_march<3, false, false, false, 5, 3>), gm, blk, 0, st, fs, p, plan, groups, bands, band_groups);
            if (mexp == 4) hipLaunchKernelGGL((swsk::sws_k_rgb_march<3, false, false, false, 5, 4>), gm, blk, 0, st, fs, p, plan, groups, bands, band_groups);
            if (mexp == 5) hipLaunchKernelGGL((swsk::sws_k_rgb_march<3, false, false, false, 5, 5>), gm, blk, 0, st, fs, p, plan, groups, bands, band_groups);
            return 0;
        }
    }
#endif
#define LAUNCH_MARCH(S, A) do { if (ncr == 5) hipLaunchKernelGGL((swsk::sws_k_rgb_march<B, S, N, A, 5>), gm, blk, 0, st, fs, p, plan, groups, bands, band_groups); \
                                else if (ncr == 6) hipLaunchKernelGGL((swsk::sws_k_rgb_march<B, S, N, A, 6>), gm, blk, 0, st, fs, p, plan, groups, bands, band_groups); \
                                else hipLaunchKernelGGL((swsk::sws_k_rgb_march<B, S, N, A, 8>), gm, blk, 0, st, fs, p, plan, groups, bands, band_groups); } while (0)
    if constexpr (B == 4) {
        if (afirst) { if (swap) LAUNCH_MARCH(true, true); else LAUNCH_MARCH(false, true); }
        else        { if (swap) LAUNCH_MARCH(true, false); else LAUNCH_MARCH(false, false); }
    } else {
        if (swap) LAUNCH_MARCH(true, false); else LAUNCH_MARCH(false, false);
    }
#undef LAUNCH_MARCH
    return 0;
}
#else
// one-shot wave kernel (vertically scaled luma, windows the march plan refuses) and the per-thread fallback (rows in _1 / _2 mode,
// unaligned pictures)
int RGBU_CAT(launch_rgbu_wave_b, RGBU_BPP, RGBU_NV)(const LaunchCtx &L)
{
    SwsInternal *c = L.c; DeviceState *d = L.d; const SwsDevParams &p = *L.p; hipStream_t st = L.st; const SwsFrameSet &fs = L.fs;
    const int n = L.n; const bool vec = L.vec;
    const dim3 blk(256);
    constexpr int B = RGBU_BPP;
    constexpr bool N = RGBU_NV != 0;
    if (vec && !c->tune.no_wave && d->all_x_mode && d->chr_window2 <= 8 && p.vChrFs <= 64) { // 1024 pixels x 2 rows per wave
        constexpr int ROWS = 2;
        const int segs = (p.dstW + 1023) >> 10, rgroups = (p.dstH + ROWS - 1) / ROWS;
        const dim3 gridw((cdiv((int64_t)segs * rgroups, 4) + 7) & ~7u, 1, n); // multiple of 8: XCD-aware order
        const bool swap = B == 4 ? p.lut.swap_rb32 != 0 : p.lut.rgb_order != 0;
        const bool afirst = B == 4 && p.lut.perm32 == 0x02010003u;
        const bool ncr6 = d->chr_window2 <= 6;
#define LAUNCH_WAVE(S, A) do { if (ncr6) hipLaunchKernelGGL((swsk::sws_k_rgb_fused_unity_wave2<B, S, N, A, ROWS, 6>), gridw, blk, 0, st, fs, p); \
                               else hipLaunchKernelGGL((swsk::sws_k_rgb_fused_unity_wave2<B, S, N, A, ROWS, 8>), gridw, blk, 0, st, fs, p); } while (0)
        if constexpr (B == 4) {
            if (afirst) { if (swap) LAUNCH_WAVE(true, true); else LAUNCH_WAVE(false, true); }
            else        { if (swap) LAUNCH_WAVE(true, false); else LAUNCH_WAVE(false, false); }
        } else {
            if (swap) LAUNCH_WAVE(true, false); else LAUNCH_WAVE(false, false);
        }
#undef LAUNCH_WAVE
        return 0;
    }
    const int npairs = (p.dstW + 1) >> 1, bpr = (npairs + 3) >> 2;
    const dim3 grid(cdiv((int64_t)bpr * p.dstH, 256), 1, n);
    if (vec) hipLaunchKernelGGL((swsk::sws_k_rgb_fused_unity<B, N, true>), grid, blk, 0, st, fs, p);
    else hipLaunchKernelGGL((swsk::sws_k_rgb_fused_unity<B, N, false>), grid, blk, 0, st, fs, p);
    return 0;
}
#endif

} // namespace swship
#endif
