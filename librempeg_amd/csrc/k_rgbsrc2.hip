// Translation unit of sws_k_rgbsrc_unity2 (kernels_rgbsrc2.hpp): same-size 8-bit RGB -> 8-bit 4:2:0 / 4:2:2 YUV as a wave march.
#include <algorithm>

#include "devstate.hpp"
#include "kernels_rgbsrc2.hpp"

namespace swship {

// 1 = launched, 0 = not a shape of this form (the caller takes sws_k_rgbsrc_unity)
int launch_rgbsrc2(const LaunchCtx &L)
{
    SwsInternal *c = L.c; DeviceState *d = L.d; const SwsDevParams &p = *L.p; hipStream_t st = L.st; const SwsFrameSet &fs = L.fs;
    const int n = L.n;
    if (!d->rgbsrc2_rows || c->tune.no_rgbsrc2 || (p.dstW & 3) || !frames_desc_ok(L.frames, n, p.srcH, p.dstH)) return 0;
    const int npv = d->rgbsrc2_npv;
    const int rd = npv <= 1 ? 1 : npv <= 3 ? 3 : npv <= 5 ? 5 : 8;
    // two groups of 256 pixels per wave where the ring leaves room for them and the picture is wide enough to fill the machine that way
    // (two groups of 256 pixels per wave were measured too: slower at 1080p and at 4K -- the kernel wants waves, not work per wave)
    int G = 1;
    if (!c->tune.strip_cols_auto && rd <= 5 && !p.range_active) G = c->tune.strip_cols_l == 2 ? 2 : 1;      // (experiments)
    swsk::RgbSrc2Geom g;
    g.strips = (int)cdiv(p.dstW, 256 * G);
    g.npv = npv;                                  // (a row is due when pair pf + npv - 1 has entered the ring; its taps sit in the newest npv of the rd slots)
    g.rows = d->rgbsrc2_rows;
    // bands: one resident round of waves, but never shorter than 32 rows -- every band re-reads the npv - 1 row pairs above it to fill its ring
    const int target = 2 * c->tune.strip_waves, minrows = std::max(8, 8 * c->tune.strip_min_rows);
    int bands = std::max(1, std::min((int)cdiv(target, g.strips * n), (int)cdiv(p.dstH, minrows)));
    g.band_rows = (int)((cdiv(p.dstH, bands) + 1) & ~1u);
    g.bands = (int)cdiv(p.dstH, g.band_rows);
    const dim3 grid(cdiv((int64_t)g.strips * g.bands, 4), 1, n), blk(256);
    const bool nv = p.dstKind == DSTK_NV12;
    const bool rng = p.range_active;     // (dev_prepare_on admits a range conversion only into full range: the RNG instantiations)
#define SWS_R2G(B, N, R) do { if (rng) hipLaunchKernelGGL((swsk::sws_k_rgbsrc_unity2<B, N, R, 1, true>), grid, blk, 0, st, fs, p, g); \
                              else if (G == 2) hipLaunchKernelGGL((swsk::sws_k_rgbsrc_unity2<B, N, R, 2>), grid, blk, 0, st, fs, p, g); \
                              else hipLaunchKernelGGL((swsk::sws_k_rgbsrc_unity2<B, N, R, 1>), grid, blk, 0, st, fs, p, g); } while (0)
#define SWS_R2R(B, N) do { if (rd == 1) SWS_R2G(B, N, 1); else if (rd == 3) SWS_R2G(B, N, 3); else if (rd == 5) SWS_R2G(B, N, 5); \
                           else if (rng) hipLaunchKernelGGL((swsk::sws_k_rgbsrc_unity2<B, N, 8, 1, true>), grid, blk, 0, st, fs, p, g); \
                           else hipLaunchKernelGGL((swsk::sws_k_rgbsrc_unity2<B, N, 8, 1>), grid, blk, 0, st, fs, p, g); } while (0)
    if (p.srcKind == SRCK_GBRP) { if (nv) SWS_R2R(0, true); else SWS_R2R(0, false); }
    else if (p.srcKind == SRCK_RGB24) { if (nv) SWS_R2R(3, true); else SWS_R2R(3, false); }
    else { if (nv) SWS_R2R(4, true); else SWS_R2R(4, false); }
#undef SWS_R2R
#undef SWS_R2G
    return 1;
}

} // namespace swship
