// Translation unit of the marching strip kernel for scaled packed RGB -> packed RGB (kernels_striprgb2rgb.hpp: one launch instead of reader pre-pass +
// strip launches + full-chroma epilogue).  Compiled once per (source bytes per pixel, destination bytes per pixel) part (-DR2R_S=3|4 -DR2R_D=3|4);
// without the macros it compiles the launcher.
#include <algorithm>
#include <map>
#include <mutex>

#include "devstate.hpp"

namespace swship {
typedef void (*StripR2RFn)(SwsFrameSet, SwsDevParams, SwsStripGeom, SwsStripGeom, int, int);
StripR2RFn striprgb2rgb_fn_s3d3(int nph, int rd, bool half, bool alpha); StripR2RFn striprgb2rgb_fn_s3d4(int nph, int rd, bool half, bool alpha);
StripR2RFn striprgb2rgb_fn_s4d3(int nph, int rd, bool half, bool alpha); StripR2RFn striprgb2rgb_fn_s4d4(int nph, int rd, bool half, bool alpha);
}

#ifndef R2R_S
namespace swship {

static int r2r_waves_per_simd(StripR2RFn fn, int lds, int device)
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

// 1 = launched, 0 = not a shape of this form
int launch_strip_rgb2rgb(const LaunchCtx &L)
{
    SwsInternal *c = L.c; DeviceState *d = L.d; const SwsDevParams &p = *L.p; hipStream_t st = L.st; const SwsFrameSet &fs = L.fs;
    const int n = L.n;
    if (!d->rgb2rgb_ok || c->tune.no_strip_rgb2rgb) return 0;
    SwsStripGeom gl = d->stripL2, gc = d->stripC2;
    const int nph_need = std::max(gl.nph, gc.nph);
    const int nph = nph_need <= 3 ? 3 : nph_need <= 4 ? 4 : nph_need <= 5 ? 5 : nph_need <= 6 ? 6 : 8;
    const int rd = gl.npv <= 5 ? 5 : 8;
    const bool s4 = p.srcKind == SRCK_RGB32, d4 = d->fullchr_kind == DSTK_RGB32, alpha = d->fullchr_on == 2, half = p.chr_half != 0;
    StripR2RFn fn = s4 ? (d4 ? striprgb2rgb_fn_s4d4(nph, rd, half, alpha) : striprgb2rgb_fn_s4d3(nph, rd, half, alpha))
                       : (d4 ? striprgb2rgb_fn_s3d4(nph, rd, half, alpha) : striprgb2rgb_fn_s3d3(nph, rd, half, alpha));
    if (!fn) return 0;
    const int npx = d->rgb2rgb_npx;
    // per wave: two rows of npx + 16 samples (u16) for Y (and A), four for U / V (half as long from the half readers)
    const int rowL = (npx + 16) >> 1, rowC = half ? (npx + 16) >> 2 : (npx + 16) >> 1;
    const int wave_dw = (alpha ? 4 : 2) * rowL + 4 * rowC;
    const int lds = 4 * wave_dw * 4 + 64;
    const int wps = c->tune.strip_short_waves > 0 ? c->tune.strip_short_waves : r2r_waves_per_simd(fn, lds, d->device);
    const int target = std::max(1, (int)((int64_t)c->tune.strip_waves * wps / 4));
    const int minrows = std::max(1, c->tune.strip_min_rows), H = p.dstH;
    const int bands = std::max(1, std::min(target / std::max(1, gl.strips * n), (H + minrows - 1) / minrows));
    gl.band_rows = (H + bands - 1) / bands;
    gl.bands = (H + gl.band_rows - 1) / gl.band_rows;
    gl.debug = gc.debug = c->tune.debug;
    const dim3 grid(cdiv((int64_t)gl.strips * gl.bands, 4), 1, n), blk(256);
    hipLaunchKernelGGL(fn, grid, blk, (size_t)lds, st, fs, p, gl, gc, npx, wave_dw);
    return 1;
}

} // namespace swship
#else
#include "kernels_striprgb2rgb.hpp"

#define R2R_CAT2(a, b, c) a##b##d##c
#define R2R_CAT(a, b, c) R2R_CAT2(a, b, c)

namespace swship {

template <bool HALF, bool ALPHA, int RD>
static StripR2RFn r2r_nph(int nph)
{
    switch (nph) {
    case 3: return swsk::sws_k_strip_rgb2rgb<R2R_S, R2R_D, HALF, ALPHA, 3, RD, 2>;
    case 4: return swsk::sws_k_strip_rgb2rgb<R2R_S, R2R_D, HALF, ALPHA, 4, RD, 2>;
    case 5: return swsk::sws_k_strip_rgb2rgb<R2R_S, R2R_D, HALF, ALPHA, 5, RD, 2>;
    case 6: return swsk::sws_k_strip_rgb2rgb<R2R_S, R2R_D, HALF, ALPHA, 6, RD, 2>;
    case 8: return swsk::sws_k_strip_rgb2rgb<R2R_S, R2R_D, HALF, ALPHA, 8, RD, 2>;
    default: return nullptr;
    }
}
template <bool HALF, bool ALPHA>
static StripR2RFn r2r_rd(int nph, int rd) { return rd == 5 ? r2r_nph<HALF, ALPHA, 5>(nph) : rd == 8 ? r2r_nph<HALF, ALPHA, 8>(nph) : nullptr; }

StripR2RFn R2R_CAT(striprgb2rgb_fn_s, R2R_S, R2R_D)(int nph, int rd, bool half, bool alpha)
{
#if R2R_S == 4 && R2R_D == 4
    if (alpha) return half ? r2r_rd<true, true>(nph, rd) : r2r_rd<false, true>(nph, rd);
#else
    if (alpha) return nullptr;
#endif
    return half ? r2r_rd<true, false>(nph, rd) : r2r_rd<false, false>(nph, rd);
}

} // namespace swship
#endif
