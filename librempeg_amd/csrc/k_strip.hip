// Translation unit of the marching strip kernel (C3b and every wide h+v polyphase conversion to planar / semi-planar YUV).
#include <algorithm>

#include "devstate.hpp"
#include "kernels_strip.hpp"

namespace swship {

// which: 1 = the luma launch, 2 = the chroma launch, 3 = both
int launch_strip_planes(const LaunchCtx &L, int which)
{
    SwsInternal *c = L.c; DeviceState *d = L.d; const SwsDevParams &p = *L.p; hipStream_t st = L.st; const SwsFrameSet &fs = L.fs;
    const SwsFramePtrs *frames = L.frames; const int n = L.n, sliceY = L.sliceY, sliceH = L.sliceH; const bool vec = L.vec;
    const dim3 blk(256);
    (void)c; (void)d; (void)frames; (void)vec; (void)sliceY; (void)sliceH;
            const int target = c->tune.strip_waves;
            const bool s16 = p.srcKind == SRCK_PLANAR16 || p.srcKind == SRCK_P010;
            auto launch = [&](SwsStripGeom g, int H, bool chroma) {
                int bands = std::max(1, std::min(target / std::max(1, g.strips * n), (H + 15) / 16));
                g.debug = c->tune.debug;
                g.band_rows = (H + bands - 1) / bands;
                g.bands = (H + g.band_rows - 1) / g.band_rows;
                const dim3 grid(cdiv((int64_t)g.strips * g.bands, 4), 1, n);
#define SWS_STRIP(S, C, K) hipLaunchKernelGGL((swsk::sws_k_strip_march<S, C, K>), grid, blk, g.lds_bytes, st, fs, p, g)
#define SWS_STRIP_DMA(C, K) hipLaunchKernelGGL((swsk::sws_k_strip_dma<C, K>), grid, blk, g.lds_dma_bytes, st, fs, p, g)
                const int cols = g.TW / 64;
                if (s16 && g.dma_ok && !c->tune.no_strip_dma) {   // 16-bit sources: LDS-DMA ring, 3 row pairs in flight per wave
                    if (chroma) { if (cols == 1) SWS_STRIP_DMA(true, 1); else SWS_STRIP_DMA(true, 2); }
                    else        { if (cols == 2) SWS_STRIP_DMA(false, 2); else SWS_STRIP_DMA(false, 4); }
                    return;
                }
                if (chroma) { if (s16) { if (cols == 1) SWS_STRIP(true, true, 1); else SWS_STRIP(true, true, 2); }
                              else     { if (cols == 1) SWS_STRIP(false, true, 1); else SWS_STRIP(false, true, 2); } }
                else        { if (s16) { if (cols == 2) SWS_STRIP(true, false, 2); else SWS_STRIP(true, false, 4); }
                              else     { if (cols == 2) SWS_STRIP(false, false, 2); else SWS_STRIP(false, false, 4); } }
#undef SWS_STRIP
#undef SWS_STRIP_DMA
            };
            if (which & 1) launch(d->stripL, p.dstH, false);
            if (which & 2) launch(d->stripC, p.chrDstH, true);
    return 0;
}

int launch_strip(const LaunchCtx &L) { return launch_strip_planes(L, 3); }

// Same-size planar YUV -> planar / semi-planar YUV whose luma filters are the identity in both directions and whose chroma is scaled
// (yuv422p -> yuv420p, yuv444p -> yuv420p / nv12, 10-bit -> 8-bit twins ...): the luma plane is a streaming per-sample pass (the scaler's own
// arithmetic with one tap: k_layout.hip launch_layout_plane1), the chroma planes are the strip kernel's chroma launch.
int launch_mixed(const LaunchCtx &L)
{
    int r = launch_layout_plane1(L);
    if (r < 0) return r;
    return launch_strip_planes(L, 2);
}

} // namespace swship
