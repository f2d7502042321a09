// Translation unit of the streaming converters: planarToP01xWrapper / planar8ToP01xleWrapper (C3a) and the fused planar float
// RGB -> 4:4:4 YUV chain (C5).
#include "generic_kinds.hpp"
#include "kernels_fast.hpp"
#include "kernels_striprgb.hpp"   // (lut_pair, LutTabs: sws_k_lut_rgb)
#include "kernels_stream.hpp"
#include "kernels_rgbsrc.hpp"

namespace swship {

int launch_p01x(const LaunchCtx &L)
{
    SwsInternal *c = L.c; DeviceState *d = L.d; const SwsDevParams &p = *L.p; hipStream_t st = L.st; const SwsFrameSet &fs = L.fs;
    const SwsFramePtrs *frames = L.frames; const int n = L.n, sliceY = L.sliceY, sliceH = L.sliceH; const bool vec = L.vec;
    const dim3 blk(256);
    (void)c; (void)d; (void)frames; (void)vec; (void)sliceY; (void)sliceH;
        const int rows = sliceH + ((sliceH + 1) >> 1);
        const dim3 grid(cdiv(cdiv(p.srcW, 8), 256), rows, n);
        const bool s8 = c->plan == PLAN_UNSC_8_P01X;
        const int p01x_ch = c->tune.p01x_ch;
        if (!s8 && vec && p01x_ch > 0) {   // streaming form: CH x 16 bytes per lane, wave-contiguous
            const int row_chunks = cdiv(2 * p.srcW, 16);
            if (p01x_ch == 4) { const dim3 g4(cdiv(row_chunks, 4 * 256), rows, n);
                                hipLaunchKernelGGL((swsk::sws_k_p01x_stream<4>), g4, blk, 0, st, fs, p, sliceY, sliceH); }
            else if (p01x_ch == 1) { const dim3 g1(cdiv(row_chunks, 256), rows, n);
                                hipLaunchKernelGGL((swsk::sws_k_p01x_stream<1>), g1, blk, 0, st, fs, p, sliceY, sliceH); }
            else { const dim3 g2(cdiv(row_chunks, 2 * 256), rows, n);
                   hipLaunchKernelGGL((swsk::sws_k_p01x_stream<2>), g2, blk, 0, st, fs, p, sliceY, sliceH); }
            return 0;
        }
        if (s8 && vec) hipLaunchKernelGGL((swsk::sws_k_p01x_unscaled<true, true>), grid, blk, 0, st, fs, p, sliceY, sliceH);
        else if (s8) hipLaunchKernelGGL((swsk::sws_k_p01x_unscaled<true, false>), grid, blk, 0, st, fs, p, sliceY, sliceH);
        else if (vec) hipLaunchKernelGGL((swsk::sws_k_p01x_unscaled<false, true>), grid, blk, 0, st, fs, p, sliceY, sliceH);
        else hipLaunchKernelGGL((swsk::sws_k_p01x_unscaled<false, false>), grid, blk, 0, st, fs, p, sliceY, sliceH);
    return 0;
}

int launch_f32rgb(const LaunchCtx &L)
{
    SwsInternal *c = L.c; DeviceState *d = L.d; const SwsDevParams &p = *L.p; hipStream_t st = L.st; const SwsFrameSet &fs = L.fs;
    const SwsFramePtrs *frames = L.frames; const int n = L.n, sliceY = L.sliceY, sliceH = L.sliceH; const bool vec = L.vec;
    const dim3 blk(256);
    (void)c; (void)d; (void)frames; (void)vec; (void)sliceY; (void)sliceH;
            const dim3 g(cdiv((int64_t)((p.srcW + 7) >> 3) * p.srcH, 256), 1, n);
            hipLaunchKernelGGL(swsk::sws_k_f32rgb_to_yuv444_unity, g, blk, 0, st, fs, p);
    return 0;
}

// packed RGB -> planar / semi-planar 8-bit YUV of the same size: lanes of four luma columns, bands of rows sized for about 8192 waves (eight per SIMD)
int launch_rgbsrc(const LaunchCtx &L)
{
    const SwsDevParams &p = *L.p; hipStream_t st = L.st; const SwsFrameSet &fs = L.fs;
    const int n = L.n;
    const dim3 blk(256);
    const int lanes = cdiv(p.dstW, 4), waves_row = cdiv(lanes, 64);
    int bands = std::max(1, std::min((int)cdiv(8192, waves_row * n), (int)cdiv(p.dstH, 8)));
    swsk::RgbSrcGeom g;
    g.band_rows = (cdiv(p.dstH, bands) + 1) & ~1;
    g.bands = cdiv(p.dstH, g.band_rows);
    g.rows = L.d->rgbsrc_rows;
    const dim3 grid(cdiv(lanes, 256), g.bands, n);
    const bool nv = p.dstKind == DSTK_NV12;
    const bool r8 = p.vChrFs <= 8;
#define SWS_RGBSRC(B, N) do { if (r8) hipLaunchKernelGGL((swsk::sws_k_rgbsrc_unity<B, N, 8>), grid, blk, 0, st, fs, p, g); \
                              else    hipLaunchKernelGGL((swsk::sws_k_rgbsrc_unity<B, N, 16>), grid, blk, 0, st, fs, p, g); } while (0)
    if (p.srcKind == SRCK_GBRP) { if (nv) SWS_RGBSRC(0, true); else SWS_RGBSRC(0, false); }
    else if (p.srcKind == SRCK_RGB24) { if (nv) SWS_RGBSRC(3, true); else SWS_RGBSRC(3, false); }
    else                         { if (nv) SWS_RGBSRC(4, true); else SWS_RGBSRC(4, false); }
#undef SWS_RGBSRC
    return 0;
}

// reader pre-pass of a scaled packed 24 / 32 bpp RGB source (dev_prepare_on: rgbread_on): 16-bit Y / U / V planes per frame at `base`
int launch_rgb_read16(const LaunchCtx &L, uint8_t *base, int64_t frame_bytes, int64_t offU, int64_t offV, int strideY, int strideC, int64_t offA, int a_pos)
{
    const SwsDevParams &p = *L.p;
    if (p.srcKind != SRCK_RGB24 && p.srcKind != SRCK_RGB32 && p.srcKind != SRCK_GBRP) {   // the other RGB kinds (dev_prepare_on: rgbread_kindN): the per-kind element-per-thread reader
        const GenericKindFns *ks = generic_kind_fns(p.srcKind);
        if (!ks || !ks->read16) { log_msg(L.c, 0, "internal error: no reader pre-pass kernel for source kind %d\n", p.srcKind); return SWS_AVERROR(EINVAL); }
        Read16Layout l16;
        l16.base = base; l16.frame_bytes = frame_bytes; l16.offU = offU; l16.offV = offV; l16.strideY = strideY; l16.strideC = strideC;
        // the vector forms of k_generic_kinds.hip (16-byte loads from 16-byte aligned rows): planar RGB of 9 - 16 bits and rgb48 / rgba64 eight pixels per thread, gbrpf32 four
        l16.vec = 0;
        if (L.vec && p.chrSrcVSub == 0 && p.vline_mode == 0) {
            if (p.srcKind == SRCK_GBRP16 && !(p.srcW & 7) && p.chrSrcW == p.srcW) l16.vec = 8;
            else if (p.srcKind == SRCK_RGB48 && !(p.srcW & 7) && (p.chrSrcW == p.srcW || (p.chr_half && p.chrSrcW == (p.srcW >> 1)))) l16.vec = 8;
            else if (p.srcKind == SRCK_GBRPF32 && !(p.srcW & 3) && p.chrSrcW == p.srcW) l16.vec = 4;
        }
        const dim3 grid(cdiv(l16.vec ? p.srcW / l16.vec : p.chrSrcW, 256), p.srcH, L.n), blk(256);
        hipLaunchKernelGGL(ks->read16, grid, blk, 0, L.st, L.fs, p, l16);
        return 0;
    }
    swsk::RgbReadLayout lay;
    lay.base = base; lay.frame_bytes = frame_bytes; lay.offU = offU; lay.offV = offV; lay.strideY = strideY; lay.strideC = strideC; lay.offA = offA; lay.a_pos = a_pos;
    const dim3 grid(cdiv(cdiv(p.srcW, 4), 256), cdiv(p.srcH, swsk::RGBREAD_RPW), L.n), blk(256);
    if (p.chr_half) {
        if (p.srcKind == SRCK_GBRP) hipLaunchKernelGGL((swsk::sws_k_rgb_read16<0, true>), grid, blk, 0, L.st, L.fs, p, lay);
        else if (p.srcKind == SRCK_RGB24) hipLaunchKernelGGL((swsk::sws_k_rgb_read16<3, true>), grid, blk, 0, L.st, L.fs, p, lay);
        else hipLaunchKernelGGL((swsk::sws_k_rgb_read16<4, true>), grid, blk, 0, L.st, L.fs, p, lay);
    } else {
        if (p.srcKind == SRCK_GBRP) hipLaunchKernelGGL((swsk::sws_k_rgb_read16<0, false>), grid, blk, 0, L.st, L.fs, p, lay);
        else if (p.srcKind == SRCK_RGB24) hipLaunchKernelGGL((swsk::sws_k_rgb_read16<3, false>), grid, blk, 0, L.st, L.fs, p, lay);
        else hipLaunchKernelGGL((swsk::sws_k_rgb_read16<4, false>), grid, blk, 0, L.st, L.fs, p, lay);
    }
    return 0;
}

// yuv420p / nv12 ... -> yuyv422 / uyvy422 / yvyu422 at the same size: the mixed plan and its interleave in one pass over whole groups of 8 pixels (kernels_stream.hpp);
// L.fs holds {the caller's source planes -> the caller's packed picture}.  0 = not a shape of this kernel (the caller runs the three passes)
int launch_mixed_join422(const LaunchCtx &L, bool uyvy)
{
    const SwsDevParams &p = *L.p;
    SwsInternal *c = L.c;
    if (!mixed_join422_shape(c, L.d, p)) return 0;
    for (int i = 0; i < L.n; i++) {
        const SwsFramePtrs &a = L.frames[i];
        for (int k = 0; k < (p.srcKind == SRCK_NV12 ? 2 : 3); k++) if (!a.src[k] || (((uintptr_t)a.src[k] | (uintptr_t)(int64_t)a.srcStride[k]) & 15)) return 0;
        if (!a.dst[0] || (((uintptr_t)a.dst[0] | (uintptr_t)(int64_t)a.dstStride[0]) & 15)) return 0;
    }
    const int groups = p.dstW / 8;
    const dim3 grid(cdiv(groups, 64), cdiv(p.dstH, 4 * swsk::MJ422_ROWS), L.n), blk(256);
    if (p.srcKind == SRCK_NV12) hipLaunchKernelGGL((swsk::sws_k_mixed_join422<true>), grid, blk, 0, L.st, L.fs, p, uyvy ? 1 : 0, groups);
    else hipLaunchKernelGGL((swsk::sws_k_mixed_join422<false>), grid, blk, 0, L.st, L.fs, p, uyvy ? 1 : 0, groups);
    return 1;
}

// a gray source into 24 / 32 bpp RGB through the full-chroma epilogue: its chroma sums are constants of the row, which sws_k_fullchr_rgb<.., 3> computes itself
// (dev_exec.hip skips the sws_k_gray_chroma launch on the same predicate)
bool fullchr_gray_const(const LaunchCtx &L)
{
    const SwsDevParams &p = *L.p;
    return L.d->fullchr_on == 1 && (L.d->fullchr_kind == DSTK_RGB24 || L.d->fullchr_kind == DSTK_RGB32) && p.no_chroma && isGray(L.c->opts.src_format) && !L.d->fullchr_direct &&
           !L.c->tune.no_wave;
}

// the full-chroma RGB epilogue behind the strip kernels (dev_prepare_on: fullchr_on; L.fs holds {src = the int32 sum planes, dst = the packed picture})
void launch_fullchr_rgb(const LaunchCtx &L)
{
    const SwsDevParams &p = *L.p;
    const dim3 grid(cdiv(cdiv(p.dstW, 4), 256), cdiv(p.dstH, swsk::FULLCHR_RPW), L.n), blk(256);
    const int srcm = L.d->fullchr_direct;     // 0: the strip kernels' sums; 1 / 2: the 8 / 16-bit planes of a same-size 4:4:4 source
    if (L.d->fullchr_on == 3) {   // the LUT writers (no full chroma): chroma sums at half the width
        if (p.lut.pix_step == 4) hipLaunchKernelGGL((swsk::sws_k_lut_rgb<4>), grid, blk, 0, L.st, L.fs, p);
        else hipLaunchKernelGGL((swsk::sws_k_lut_rgb<3>), grid, blk, 0, L.st, L.fs, p);
        return;
    }
#define SWS_FC_SRCM(K, ...) do { if (srcm == 1) hipLaunchKernelGGL((swsk::K<__VA_ARGS__, 1>), grid, blk, 0, L.st, L.fs, p); \
                                 else if (srcm == 2) hipLaunchKernelGGL((swsk::K<__VA_ARGS__, 2>), grid, blk, 0, L.st, L.fs, p); \
                                 else hipLaunchKernelGGL((swsk::K<__VA_ARGS__, 0>), grid, blk, 0, L.st, L.fs, p); } while (0)
    if (L.d->fullchr_kind == DSTK_GBRP16 || L.d->fullchr_kind == DSTK_GBRPF32) {   // (the 19-bit strip kernel's sums: fullchr_on == 4 without the generic writer)
        if (L.d->fullchr_kind == DSTK_GBRPF32) hipLaunchKernelGGL((swsk::sws_k_fullchr_gbrp16<true>), grid, blk, 0, L.st, L.fs, p);
        else hipLaunchKernelGGL((swsk::sws_k_fullchr_gbrp16<false>), grid, blk, 0, L.st, L.fs, p);
        return;
    }
    if (L.d->fullchr_kind == DSTK_GBRP) {
        const bool wide = p.dst_bits > 8, alpha = L.d->fullchr_on == 2;
        if (wide && alpha) SWS_FC_SRCM(sws_k_fullchr_gbrp, true, true);
        else if (wide) SWS_FC_SRCM(sws_k_fullchr_gbrp, true, false);
        else if (alpha) SWS_FC_SRCM(sws_k_fullchr_gbrp, false, true);
        else SWS_FC_SRCM(sws_k_fullchr_gbrp, false, false);
        return;
    }
    if (fullchr_gray_const(L)) {     // a gray source: the chroma sums are constants of the row, computed in the epilogue (no sws_k_gray_chroma launch, no U / V planes)
        if (p.lut.pix_step == 4) hipLaunchKernelGGL((swsk::sws_k_fullchr_rgb<4, false, 3>), grid, blk, 0, L.st, L.fs, p);
        else hipLaunchKernelGGL((swsk::sws_k_fullchr_rgb<3, false, 3>), grid, blk, 0, L.st, L.fs, p);
        return;
    }
    if (p.lut.pix_step == 4 && L.d->fullchr_on == 2) SWS_FC_SRCM(sws_k_fullchr_rgb, 4, true);
    else if (p.lut.pix_step == 4) SWS_FC_SRCM(sws_k_fullchr_rgb, 4, false);
    else SWS_FC_SRCM(sws_k_fullchr_rgb, 3, false);
#undef SWS_FC_SRCM
}

// the chroma planes of a gray source in a YUV destination (dev_prepare_on: a gray source on the strip kernels' luma launch)
void launch_gray_chroma(const LaunchCtx &L)
{
    const SwsDevParams &p = *L.p;
    if (p.chrDstW <= 0 || p.chrDstH <= 0) return;
    // (16-byte aligned chroma planes: 16 bytes per thread, row and plane)
    const bool semi = p.dstKind == DSTK_NV12 || p.dstKind == DSTK_P010 || p.dstKind == DSTK_P016;
    const int up = semi ? 1 : p.u_plane_dst, vp = semi ? 1 : p.v_plane_dst;
    bool al = !L.c->tune.no_wave;
    for (int i = 0; i < L.n && al; i++)
        for (int k : { up, vp }) al = al && L.frames[i].dst[k] && (((uintptr_t)L.frames[i].dst[k] | (uintptr_t)(int64_t)L.frames[i].dstStride[k]) & 15) == 0;
    if (al) {
        const int sbytes = p.dstKind == DSTK_RAW32 ? 4 : (p.dstKind == DSTK_PLANAR8 || p.dstKind == DSTK_NV12) ? 1 : 2, spt = 16 / (sbytes * (semi ? 2 : 1));
        hipLaunchKernelGGL(swsk::sws_k_gray_chroma_vec, dim3(cdiv(cdiv(p.chrDstW, spt), 64), p.chrDstH, L.n), dim3(64), 0, L.st, L.fs, p);
        return;
    }
    hipLaunchKernelGGL(swsk::sws_k_gray_chroma, dim3(cdiv(p.chrDstW, 256), p.chrDstH, L.n), dim3(256), 0, L.st, L.fs, p);
}

// plane copies between unaligned pictures and their aligned working copies (dev_exec.hip launch_plan_le; L.fs holds {src[k] -> dst[k]})
void launch_stage_planes(const LaunchCtx &L, const int row_bytes[4], const int rows[4], bool in)
{
    swsk::StageExtents e;
    int wmax = 0, hmax = 0;
    for (int k = 0; k < 4; k++) { e.row_bytes[k] = row_bytes[k]; e.rows[k] = rows[k]; wmax = std::max(wmax, row_bytes[k]); hmax = std::max(hmax, rows[k]); }
    if (wmax <= 0 || hmax <= 0) return;
    const dim3 grid(cdiv(cdiv(wmax, 4), 256), cdiv(hmax, swsk::STAGE_RPW), L.n), blk(256);
    if (in) hipLaunchKernelGGL((swsk::sws_k_stage_planes<true>), grid, blk, 0, L.st, L.fs, e);
    else hipLaunchKernelGGL((swsk::sws_k_stage_planes<false>), grid, blk, 0, L.st, L.fs, e);
}

// the alpha bytes behind sws_k_strip_rgb (dev_exec.hip: alpha_launch == 2; L.fs holds {src[0] = the int32 sums of the A plane, dst[0] = the packed picture})
void launch_alpha_merge32(const LaunchCtx &L)
{
    const SwsDevParams &p = *L.p;
    const dim3 grid(cdiv(cdiv(p.dstW, 4), 256), cdiv(p.dstH, swsk::FULLCHR_RPW), L.n), blk(256);
    hipLaunchKernelGGL((swsk::sws_k_alpha_merge32), grid, blk, 0, L.st, L.fs, p);
}

// packed / planar 8-bit RGB -> planar 8-bit 4:4:4 YUV of the same size: every filter the identity (dev_prepare_on: rgb444_ok)
int launch_rgb444(const LaunchCtx &L)
{
    const SwsDevParams &p = *L.p; hipStream_t st = L.st; const SwsFrameSet &fs = L.fs;
    const int n = L.n;
    const int lanes = cdiv(p.dstW, 4), waves_row = cdiv(lanes, 64);
    const int bands = std::max(1, std::min((int)cdiv(8192, waves_row * n), (int)cdiv(p.dstH, 4)));
    const int band_rows = cdiv(p.dstH, bands);
    const dim3 grid(cdiv(lanes, 256), cdiv(p.dstH, band_rows), n), blk(256);
    if (p.srcKind == SRCK_GBRP) hipLaunchKernelGGL((swsk::sws_k_rgb_yuv444_unity<0>), grid, blk, 0, st, fs, p, band_rows);
    else if (p.srcKind == SRCK_RGB24) hipLaunchKernelGGL((swsk::sws_k_rgb_yuv444_unity<3>), grid, blk, 0, st, fs, p, band_rows);
    else hipLaunchKernelGGL((swsk::sws_k_rgb_yuv444_unity<4>), grid, blk, 0, st, fs, p, band_rows);
    return 0;
}

} // namespace swship
