// Translation unit of the marching strip kernel with the packed-RGB epilogue (scaled planar 8-bit YUV -> rgb24 / bgr24 / rgba / bgra /
// argb / abgr and their 0-alpha twins).  Compiled once per (bytes per pixel, luma ring depth) part (-DSRGB_BPP=3|4 -DSRGB_RL=5|8) so that
// the chroma ring depths and horizontal tap counts of each part build in parallel with the other parts; without the macros it compiles the
// dispatcher.
#include <algorithm>

#include "devstate.hpp"

namespace swship {
int launch_striprgb_b3l5(const LaunchCtx &L); int launch_striprgb_b3l8(const LaunchCtx &L);
int launch_striprgb_b4l5(const LaunchCtx &L); int launch_striprgb_b4l8(const LaunchCtx &L);
}

#ifndef SRGB_BPP
namespace swship {

int launch_striprgb(const LaunchCtx &L)
{
    const bool b4 = L.p->dstKind == DSTK_RGB32, lng = L.d->stripRL.npv > 5;
    return b4 ? (lng ? launch_striprgb_b4l8(L) : launch_striprgb_b4l5(L)) : (lng ? launch_striprgb_b3l8(L) : launch_striprgb_b3l5(L));
}

} // namespace swship
#else
#include "kernels_striprgb.hpp"

#define SRGB_CAT2(a, b, c) a##b##l##c
#define SRGB_CAT(a, b, c) SRGB_CAT2(a, b, c)

namespace swship {

int SRGB_CAT(launch_striprgb_b, SRGB_BPP, SRGB_RL)(const LaunchCtx &L)
{
    SwsInternal *c = L.c; DeviceState *d = L.d; hipStream_t st = L.st; const SwsFrameSet &fs = L.fs;
    const int n = L.n;
    // (a semi-planar source read directly: the kernel sees the real chroma order / word shift, the planner's parameters describe the split picture)
    const int direct = d->striprgb_direct_now ? d->striprgb_direct : 0;
    SwsDevParams pd = *L.p;
    if (direct == 1) pd.uv_swap_src = d->striprgb_direct_swap;
    if (direct == 2) pd.src_shift = d->striprgb_direct_shift;
    const SwsDevParams &p = pd;
    SwsStripGeom gl = d->stripRL, gc = d->stripRC;
    // one resident round of waves; bands of at least 16 rows (the ring fill at the top of a band costs npv row pairs per plane)
    const int target = c->tune.strip_waves, H = p.dstH;
    const int minrows = std::max(1, c->tune.strip_min_rows);    // (few frames per call: short bands, see k_strip.hip)
    int bands = std::max(1, std::min(target / std::max(1, gl.strips * n), (H + minrows - 1) / minrows));
    gl.band_rows = (H + bands - 1) / bands;
    gl.bands = (H + gl.band_rows - 1) / gl.band_rows;
    gl.debug = gc.debug = c->tune.debug;
    const int cl = gl.TW / 64;
    const bool s16 = p.srcKind == SRCK_PLANAR16;      // 9 .. 15-bit planar sources: eight samples per 16-byte chunk, staged as they are
    const int spc = s16 ? 8 : 16;
    const int wave_dw = 2 * ((gl.NCmax + spc) >> 1) + 4 * ((gc.NCmax + spc) >> 1) + 32 * cl;   // luma rows, chroma rows, exchange row
    const dim3 grid(cdiv((int64_t)gl.strips * gl.bands, 4), 1, n), blk(256);
    const int rc = gc.npv <= 1 ? 1 : gc.npv <= 3 ? 3 : 8;     // chroma ring depth of the instantiation (dev_plan*.hip laid the taps out for it)
    // LDS-DMA form (sws_k_strip_rgb8): byte rows in rings of 4 row pairs per plane class, on 16-byte aligned frames
    if (!s16 && cl == 4 && gl.dma8_ok && gc.dma8_ok && gl.hT8 && gc.hT8 && L.vec && !c->tune.no_strip_dma8) {
        const int wave8 = SWS_RGB8_DEPTH * 2 * ((gl.NCmax + 16) >> 2) + (direct == 1 ? SWS_RGB8_DEPTH * 2 * ((2 * gc.NCmax + 16) >> 2) : SWS_RGB8_DEPTH * 4 * ((gc.NCmax + 16) >> 2)) + 32 * cl;
        const size_t lds8 = (size_t)4 * wave8 * 4 + (c->tune.strip_lds_pad_kb > 0 ? (size_t)c->tune.strip_lds_pad_kb * 1024 : 0);
        const int nph8 = std::max(gl.nph8, gc.nph8);
        if (lds8 <= 60 * 1024 && nph8 <= 6) {
#define SWS_SR8(RC, N) do { if (direct == 1) hipLaunchKernelGGL((swsk::sws_k_strip_rgb8<SRGB_BPP, SRGB_RL, RC, N, 4, true>), grid, blk, lds8, st, fs, p, gl, gc, wave8); \
                            else hipLaunchKernelGGL((swsk::sws_k_strip_rgb8<SRGB_BPP, SRGB_RL, RC, N, 4>), grid, blk, lds8, st, fs, p, gl, gc, wave8); } while (0)
#define SWS_SR8N(RC) switch (nph8) { case 1: SWS_SR8(RC, 1); break; case 2: SWS_SR8(RC, 2); break; case 3: SWS_SR8(RC, 3); break; case 4: SWS_SR8(RC, 4); break; \
                                     case 5: SWS_SR8(RC, 5); break; default: SWS_SR8(RC, 6); break; }
            if (rc == 1) SWS_SR8N(1) else if (rc == 3) SWS_SR8N(3) else SWS_SR8N(8)
#undef SWS_SR8N
#undef SWS_SR8
            return 0;
        }
    }
#define SWS_SR(RC, N) do { if (s16 && direct == 2) hipLaunchKernelGGL((swsk::sws_k_strip_rgb<SRGB_BPP, SRGB_RL, RC, N, 2, true, true>), grid, blk, (size_t)4 * wave_dw * 4, st, fs, p, gl, gc, wave_dw); \
                           else if (s16) hipLaunchKernelGGL((swsk::sws_k_strip_rgb<SRGB_BPP, SRGB_RL, RC, N, 2, true>), grid, blk, (size_t)4 * wave_dw * 4, st, fs, p, gl, gc, wave_dw); \
                           else if (cl == 4) hipLaunchKernelGGL((swsk::sws_k_strip_rgb<SRGB_BPP, SRGB_RL, RC, N, 4>), grid, blk, (size_t)4 * wave_dw * 4, st, fs, p, gl, gc, wave_dw); \
                           else hipLaunchKernelGGL((swsk::sws_k_strip_rgb<SRGB_BPP, SRGB_RL, RC, N, 2>), grid, blk, (size_t)4 * wave_dw * 4, st, fs, p, gl, gc, wave_dw); } while (0)
    if (direct == 1) { log_msg(c, 0, "internal error: an interleaved chroma plane reached the register-staged strip-RGB kernel\n"); return SWS_AVERROR(EINVAL); }
#define SWS_SRN(RC) switch (gl.nph) { case 1: SWS_SR(RC, 1); break; case 2: SWS_SR(RC, 2); break; case 3: SWS_SR(RC, 3); break; case 4: SWS_SR(RC, 4); break; \
                                      case 5: SWS_SR(RC, 5); break; case 6: SWS_SR(RC, 6); break; case 7: SWS_SR(RC, 7); break; case 8: SWS_SR(RC, 8); break; \
                                      default: return SWS_AVERROR(EINVAL); }
    if (rc == 1) SWS_SRN(1) else if (rc == 3) SWS_SRN(3) else SWS_SRN(8)
#undef SWS_SRN
#undef SWS_SR
    return 0;
}

} // namespace swship
#endif
