// Translation unit of the 19-bit marching strip kernel (kernels_stripwide.hpp): destinations of 16 bits per component.
#include <algorithm>

#include "devstate.hpp"
#include "kernels_stripwide.hpp"

namespace swship {

// which: 1 = the luma launch, 2 = the chroma launch, 3 = both (k_strip.hip launch_strip_planes hands every context with 19-bit intermediates over)
#ifndef SWIDE_S16
int launch_strip_wide_s8(const LaunchCtx &L, int which);
int launch_strip_wide_s16(const LaunchCtx &L, int which);
int launch_strip_wide(const LaunchCtx &L, int which)
{
    const bool s16 = L.p->srcKind == SRCK_PLANAR16 || L.p->srcKind == SRCK_P010;
    return s16 ? launch_strip_wide_s16(L, which) : launch_strip_wide_s8(L, which);
}
#else
#if SWIDE_S16
int launch_strip_wide_s16(const LaunchCtx &L, int which)
#else
int launch_strip_wide_s8(const LaunchCtx &L, int which)
#endif
{
    SwsInternal *c = L.c; DeviceState *d = L.d; const SwsDevParams &p = *L.p; hipStream_t st = L.st; const SwsFrameSet &fs = L.fs;
    const int n = L.n;
    const dim3 blk(256);
    const bool s16 = p.srcKind == SRCK_PLANAR16 || p.srcKind == SRCK_P010;
    auto launch = [&](SwsStripGeom g, int H, bool chroma) -> int {
        const int cols = g.TW / 64, rd = g.npv <= 2 ? 2 : g.npv <= 4 ? 4 : 8;
        // strips of 64 * cols columns (dev_plan*.hip picks the widest whose window fits one chunk per lane and whose rings fit the registers: two int32 rows per pair)
        const bool cols_ok = chroma ? (cols == 1 || cols == 2) : cols == 2;
        if (!cols_ok || g.nph > 8 || g.npv > 8 || g.NCmax / (s16 ? 8 : 16) > 64) {
            log_msg(c, 0, "internal error: strip plan of %d columns per lane, %d x %d tap pairs, windows of %d samples for the 19-bit strip kernel\n", cols, g.nph, g.npv, g.NCmax);
            return SWS_AVERROR(EINVAL);
        }
        // one resident round of waves; bands of at least `minrows` output rows (k_strip.hip)
        const int target = c->tune.strip_waves, minrows = std::max(1, c->tune.strip_min_rows);
        const int bands = std::max(1, std::min(target / std::max(1, g.strips * n), (H + minrows - 1) / minrows));
        g.debug = c->tune.debug;
        g.band_rows = (H + bands - 1) / bands;
        g.bands = (H + g.band_rows - 1) / g.band_rows;
        const dim3 grid(cdiv((int64_t)g.strips * g.bands, 4), 1, n);
#define SWS_WIDE(S, C, K) do { if (rd == 2) hipLaunchKernelGGL((swsk::sws_k_strip_wide<S, C, K, 2>), grid, blk, g.lds_bytes, st, fs, p, g); \
                               else if (rd == 4) hipLaunchKernelGGL((swsk::sws_k_strip_wide<S, C, K, 4>), grid, blk, g.lds_bytes, st, fs, p, g); \
                               else hipLaunchKernelGGL((swsk::sws_k_strip_wide<S, C, K, 8>), grid, blk, g.lds_bytes, st, fs, p, g); } while (0)
#ifdef SWIDE_S16
        constexpr bool S = SWIDE_S16 != 0;
        if (chroma) {
            if (cols == 1) SWS_WIDE(S, true, 1); else SWS_WIDE(S, true, 2);
        } else SWS_WIDE(S, false, 2);
#endif
#undef SWS_WIDE
        return 0;
    };
    if (which & 1) { int r = launch(d->stripL, p.dstH, false); if (r < 0) return r; }
    if (which & 2) { int r = launch(d->stripC, p.chrDstH, true); if (r < 0) return r; }
    return 0;
}
#endif

} // namespace swship
