// Translation unit of the marching strip kernel for scaled packed-RGB sources (kernels_striprgbsrc.hpp: bgra / rgb24 ... -> planar / semi-planar YUV
// with half-width chroma in one launch, no reader pre-pass).  Compiled once per (bytes per pixel, luma ring depth) part
// (-DRSRC_BPP=0|2|3|4|30 -DRSRC_RL=5|8; 0 = planar G / B / R, 2 = packed 8-bit 4:2:2, 30 = x2rgb10 / x2bgr10) so that the chroma ring depths and horizontal tap counts of a part build in parallel with the other parts;
// without the macros it compiles the launcher.
#include <algorithm>
#include <map>
#include <mutex>

#include "devstate.hpp"

namespace swship {
typedef void (*StripRgbSrcFn)(SwsFrameSet, SwsDevParams, SwsStripGeom, SwsStripGeom, int, int);
StripRgbSrcFn striprgbsrc_fn_b3l5(int nph, int rc, int ng); StripRgbSrcFn striprgbsrc_fn_b3l8(int nph, int rc, int ng);
StripRgbSrcFn striprgbsrc_fn_b4l5(int nph, int rc, int ng); StripRgbSrcFn striprgbsrc_fn_b4l8(int nph, int rc, int ng);
StripRgbSrcFn striprgbsrc_fn_b0l5(int nph, int rc, int ng); StripRgbSrcFn striprgbsrc_fn_b0l8(int nph, int rc, int ng);   // (0: planar 8-bit G, B, R planes)
StripRgbSrcFn striprgbsrc_fn_b2l5(int nph, int rc, int ng); StripRgbSrcFn striprgbsrc_fn_b2l8(int nph, int rc, int ng);   // (2: packed 8-bit 4:2:2 -- yuyv422 / uyvy422 / yvyu422)
StripRgbSrcFn striprgbsrc_fn_b30l5(int nph, int rc, int ng); StripRgbSrcFn striprgbsrc_fn_b30l8(int nph, int rc, int ng);  // (30: x2rgb10le / x2bgr10le)
}

#ifndef RSRC_BPP
namespace swship {

// resident waves per SIMD of an instantiation at its LDS size (asked once per kernel and device)
static int rsrc_waves_per_simd(StripRgbSrcFn fn, int lds, int device)
{
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, int> cache;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair((const void *)fn, device * 4096 + lds / 64);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, (const void *)fn, 256, (size_t)lds) != hipSuccess) { (void)hipGetLastError(); blocks = 2; }
    const int w = std::max(1, std::min(8, blocks));     // a block is 4 waves, one per SIMD
    cache[key] = w;
    return w;
}

// 1 = launched, 0 = not a shape of this form (the caller runs the reader pre-pass and the two strip launches)
int launch_strip_rgbsrc(const LaunchCtx &L)
{
    SwsInternal *c = L.c; DeviceState *d = L.d; const SwsDevParams &p = *L.p; hipStream_t st = L.st; const SwsFrameSet &fs = L.fs;
    const int n = L.n;
    if (!d->striprgbsrc_ok || c->tune.no_strip_rgbsrc) return 0;
    SwsStripGeom gl = d->stripL2, gc = d->stripC2;       // (the kernel's own plans: dev_plan*.hip)
    const int nph_need = std::max(gl.nph, gc.nph);
    const int nph = nph_need <= 3 ? 3 : nph_need <= 4 ? 4 : nph_need <= 5 ? 5 : nph_need <= 6 ? 6 : 8;
    const bool l8 = gl.npv > 5;
    const int rc = gc.npv <= 5 ? 5 : gc.npv <= 8 ? 8 : 12;
    const bool b4 = p.srcKind == SRCK_RGB32;
    const int npx = d->striprgbsrc_npx;
    const int ng = npx <= 512 ? 2 : 4;                  // groups of four pixels per lane the widest strip window needs
    // (a packed 4:2:2 source whose split pass launch_plan_le skipped: the planner's parameters describe the planar working picture)
    const bool packed422 = d->striprgb_direct_now && d->striprgb_direct == 3;
    StripRgbSrcFn fn = packed422 ? (l8 ? striprgbsrc_fn_b2l8(nph, rc, ng) : striprgbsrc_fn_b2l5(nph, rc, ng)) : p.srcKind == SRCK_GBRP ? (l8 ? striprgbsrc_fn_b0l8(nph, rc, ng) : striprgbsrc_fn_b0l5(nph, rc, ng)) :
                       p.srcKind == SRCK_RGB30 ? (l8 ? striprgbsrc_fn_b30l8(nph, rc, ng) : striprgbsrc_fn_b30l5(nph, rc, ng)) :
                       b4 ? (l8 ? striprgbsrc_fn_b4l8(nph, rc, ng) : striprgbsrc_fn_b4l5(nph, rc, ng)) : (l8 ? striprgbsrc_fn_b3l8(nph, rc, ng) : striprgbsrc_fn_b3l5(nph, rc, ng));
    if (!fn) return 0;
    const int wave_dw = 2 * (npx + 16);                 // two Y rows of npx + 16 samples, four chroma rows of half as many (u16)
    const int lds = 4 * wave_dw * 4 + 64;
    // one resident round of waves at the instantiation's occupancy; bands of an even number of rows (4:2:0: a band owns whole chroma rows)
    const int wps = c->tune.strip_short_waves > 0 ? c->tune.strip_short_waves : rsrc_waves_per_simd(fn, lds, d->device);
    const int target = std::max(1, (int)((int64_t)c->tune.strip_waves * wps / 4));
    const int minrows = std::max(2, c->tune.strip_min_rows), H = p.dstH;
    const int bands = std::max(1, std::min(target / std::max(1, gl.strips * n), (H + minrows - 1) / minrows));
    gl.band_rows = (((H + bands - 1) / bands) + 1) & ~1;
    gl.bands = (H + gl.band_rows - 1) / gl.band_rows;
    gl.debug = gc.debug = c->tune.debug;
    const dim3 grid(cdiv((int64_t)gl.strips * gl.bands, 4), 1, n), blk(256);
    hipLaunchKernelGGL(fn, grid, blk, (size_t)lds, st, fs, p, gl, gc, npx, wave_dw);
    return 1;
}

} // namespace swship
#else
#include "kernels_striprgbsrc.hpp"

#define RSRC_CAT2(a, b, c) a##b##l##c
#define RSRC_CAT(a, b, c) RSRC_CAT2(a, b, c)

namespace swship {

template <int RC, int NG>
static StripRgbSrcFn rsrc_nph(int nph)
{
    switch (nph) {
    case 3: return swsk::sws_k_strip_rgbsrc<RSRC_BPP, 3, RSRC_RL, RC, NG>;
    case 4: return swsk::sws_k_strip_rgbsrc<RSRC_BPP, 4, RSRC_RL, RC, NG>;
    case 5: return swsk::sws_k_strip_rgbsrc<RSRC_BPP, 5, RSRC_RL, RC, NG>;
    case 6: return swsk::sws_k_strip_rgbsrc<RSRC_BPP, 6, RSRC_RL, RC, NG>;
    case 8: return swsk::sws_k_strip_rgbsrc<RSRC_BPP, 8, RSRC_RL, RC, NG>;
    default: return nullptr;
    }
}

StripRgbSrcFn RSRC_CAT(striprgbsrc_fn_b, RSRC_BPP, RSRC_RL)(int nph, int rc, int ng)
{
    if (ng <= 2) return rc == 5 ? rsrc_nph<5, 2>(nph) : rc == 8 ? rsrc_nph<8, 2>(nph) : rc == 12 ? rsrc_nph<12, 2>(nph) : nullptr;
    return rc == 5 ? rsrc_nph<5, 4>(nph) : rc == 8 ? rsrc_nph<8, 4>(nph) : rc == 12 ? rsrc_nph<12, 4>(nph) : nullptr;
}

} // namespace swship
#endif
