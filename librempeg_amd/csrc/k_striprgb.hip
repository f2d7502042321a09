// Translation unit of the marching strip kernel with the packed-RGB epilogue (scaled planar 8-bit YUV -> rgb24 / bgr24 / rgba / bgra /
// argb / abgr and their 0-alpha twins).  Compiled once per (bytes per pixel, ring form) part (-DSRGB_BPP=3|4 -DSRGB_LONG=0|1) so that
// the eight horizontal tap counts of each part build in parallel with the other parts; without the macros it compiles the dispatcher.
#include <algorithm>

#include "devstate.hpp"

namespace swship {
int launch_striprgb_b3l0(const LaunchCtx &L); int launch_striprgb_b3l1(const LaunchCtx &L);
int launch_striprgb_b4l0(const LaunchCtx &L); int launch_striprgb_b4l1(const LaunchCtx &L);
}

#ifndef SRGB_BPP
namespace swship {

int launch_striprgb(const LaunchCtx &L)
{
    const bool b4 = L.p->dstKind == DSTK_RGB32, lng = L.d->striprgb_long;
    return b4 ? (lng ? launch_striprgb_b4l1(L) : launch_striprgb_b4l0(L)) : (lng ? launch_striprgb_b3l1(L) : launch_striprgb_b3l0(L));
}

} // namespace swship
#else
#include "kernels_striprgb.hpp"

#define SRGB_CAT2(a, b, c) a##b##l##c
#define SRGB_CAT(a, b, c) SRGB_CAT2(a, b, c)

namespace swship {

int SRGB_CAT(launch_striprgb_b, SRGB_BPP, SRGB_LONG)(const LaunchCtx &L)
{
    SwsInternal *c = L.c; DeviceState *d = L.d; const SwsDevParams &p = *L.p; hipStream_t st = L.st; const SwsFrameSet &fs = L.fs;
    const int n = L.n;
    SwsStripGeom gl = d->stripRL, gc = d->stripRC;
    // one resident round of waves; bands of at least 16 rows (the ring fill at the top of a band costs npv row pairs per plane)
    const int target = c->tune.strip_waves, H = p.dstH;
    int bands = std::max(1, std::min(target / std::max(1, gl.strips * n), (H + 15) / 16));
    gl.band_rows = (H + bands - 1) / bands;
    gl.bands = (H + gl.band_rows - 1) / gl.band_rows;
    gl.debug = gc.debug = c->tune.debug;
    const int wave_dw = 2 * ((gl.NCmax + 16) >> 1) + 4 * ((gc.NCmax + 16) >> 1) + 128;   // luma rows, chroma rows, exchange row
    const dim3 grid(cdiv((int64_t)gl.strips * gl.bands, 4), 1, n), blk(256);
    switch (gl.nph) {
#define SWS_SR(N) case N: hipLaunchKernelGGL((swsk::sws_k_strip_rgb<SRGB_BPP, SRGB_LONG != 0, N>), grid, blk, (size_t)4 * wave_dw * 4, st, fs, p, gl, gc, wave_dw); break;
    SWS_SR(1) SWS_SR(2) SWS_SR(3) SWS_SR(4) SWS_SR(5) SWS_SR(6) SWS_SR(7) SWS_SR(8)
#undef SWS_SR
    default: return SWS_AVERROR(EINVAL);
    }
    return 0;
}

} // namespace swship
#endif
