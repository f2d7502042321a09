// Translation unit of the unscaled yuv420p / yuv422p -> 24/32 bpp packed RGB converter (C2a; yuv2rgb.c:68-559).
#include "devstate.hpp"
#include "kernels_fast.hpp"
#include "kernels_wave.hpp"

namespace swship {

int launch_yuv2rgb(const LaunchCtx &L)
{
    SwsInternal *c = L.c; DeviceState *d = L.d; const SwsDevParams &p = *L.p; hipStream_t st = L.st; const SwsFrameSet &fs = L.fs;
    const SwsFramePtrs *frames = L.frames; const int n = L.n, sliceY = L.sliceY, sliceH = L.sliceH; const bool vec = L.vec;
    const dim3 blk(256);
    (void)c; (void)d; (void)frames; (void)vec; (void)sliceY; (void)sliceH;
    const int dstW = p.dstW;
    const int npairs = ((dstW >> 3) << 2) + ((dstW & 4) ? 2 : 0) + ((dstW & 2) ? 1 : 0); // yuv2rgb.c:198-236
    const int is422 = c->opts.src_format == AV_PIX_FMT_YUV422P;
    const int nrowpairs = (sliceH + 1) >> 1; // "for (y = 0; y < srcSliceH; y += 2)"
    const int bpr = (npairs + 3) >> 2;
    if (!bpr || !nrowpairs) return 0;
    const bool bpp4 = p.dstKind == DSTK_RGB32;
    if (vec && !c->tune.no_wave) { // wave-tiled kernel: 1024 pixels x 2 rows per wave, LDS-transposed 16-byte stores
        const int segs = (2 * npairs + 1023) >> 10;
        const dim3 gridw(cdiv((int64_t)segs * nrowpairs, 4), 1, n);
        const bool swap = bpp4 ? p.lut.swap_rb32 != 0 : p.lut.rgb_order != 0;
#define LAUNCH_K1(B, S) hipLaunchKernelGGL((swsk::sws_k_yuv2rgb_unscaled_wave<B, S>), gridw, blk, 0, st, fs, p, is422, npairs, sliceY, nrowpairs)
        if (bpp4) { if (swap) LAUNCH_K1(4, true); else LAUNCH_K1(4, false); }
        else      { if (swap) LAUNCH_K1(3, true); else LAUNCH_K1(3, false); }
#undef LAUNCH_K1
    } else {
        const dim3 grid(cdiv((int64_t)bpr * nrowpairs, 256), 1, n);
        if (bpp4 && vec) hipLaunchKernelGGL((swsk::sws_k_yuv2rgb_unscaled<4, true>), grid, blk, 0, st, fs, p, is422, npairs, sliceY, nrowpairs);
        else if (bpp4) hipLaunchKernelGGL((swsk::sws_k_yuv2rgb_unscaled<4, false>), grid, blk, 0, st, fs, p, is422, npairs, sliceY, nrowpairs);
        else if (vec) hipLaunchKernelGGL((swsk::sws_k_yuv2rgb_unscaled<3, true>), grid, blk, 0, st, fs, p, is422, npairs, sliceY, nrowpairs);
        else hipLaunchKernelGGL((swsk::sws_k_yuv2rgb_unscaled<3, false>), grid, blk, 0, st, fs, p, is422, npairs, sliceY, nrowpairs);
    }
    // yuva2rgba_c / yuva2argb_c (yuv2rgb.c:524-528): source alpha into the A byte
    if (bpp4 && isALPHA(c->opts.src_format)) launch_alpha_merge(L, 2 * npairs, sliceY, 2 * nrowpairs, pix_desc(c->opts.dst_format)->comp[3].offset);
    return 0;
}

} // namespace swship
