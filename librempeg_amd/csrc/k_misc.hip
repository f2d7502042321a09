// Translation unit of the element-per-thread unscaled converters (kernels_misc.hpp, kernels_shuffle.hpp) and of the helper
// passes the planner issues around other kernels (alpha fill / merge, byte swap, XYZ).
#include "devstate.hpp"
#include "kernels_misc.hpp"
#include "kernels_shuffle.hpp"

namespace swship {

int launch_misc(const LaunchCtx &L)
{
    SwsInternal *c = L.c; DeviceState *d = L.d; const SwsDevParams &p = *L.p; hipStream_t st = L.st; const SwsFrameSet &fs = L.fs;
    const SwsFramePtrs *frames = L.frames; const int n = L.n, sliceY = L.sliceY, sliceH = L.sliceH; const bool vec = L.vec;
    const dim3 blk(256);
    (void)c; (void)d; (void)frames; (void)vec; (void)sliceY; (void)sliceH;
    if (launch_layout(L) > 0) return 0;   // the streaming form of the layout converters takes 16-byte aligned pictures (k_layout.hip)
    switch (c->plan) {
    case PLAN_UNSC_PLANAR2NV12:
    case PLAN_UNSC_NV122PLANAR:
    case PLAN_UNSC_PLANAR2NV24:
    case PLAN_UNSC_NV242PLANAR:
    case PLAN_UNSC_NV242YUV420:
    case PLAN_UNSC_YVU9_YV12:
    case PLAN_UNSC_PLANARCOPY: {
        swsk::MiscPlan plan;
        std::memset(&plan, 0, sizeof(plan));
        int maxw = 0, rows = 0;
        if (c->plan == PLAN_UNSC_PLANAR2NV24 || c->plan == PLAN_UNSC_NV242PLANAR) {       // 4:4:4: chroma rows == luma rows
            plan.mode = c->plan == PLAN_UNSC_PLANAR2NV24 ? 0 : 1;
            plan.nplanes = 2;
            plan.pl[0] = { 0, 0, p.srcW, sliceH, sliceY, 1, 0, 0 };
            plan.pl[1] = { 1, 1, c->chrSrcW, sliceH, sliceY, 1, 0, 1 };
        } else if (c->plan == PLAN_UNSC_NV242YUV420) {                                     // nv24_to_yuv420p_chroma (:229-251)
            plan.mode = 3;
            plan.nplanes = 2;
            plan.aux = sliceH;                                                             // odd last row repeats itself
            plan.pl[0] = { 0, 0, p.srcW, sliceH, sliceY, 1, 0, 0 };
            plan.pl[1] = { 1, 1, p.srcW / 2, (sliceH + 1) / 2, sliceY / 2, 1, 0, 1 };
        } else if (c->plan == PLAN_UNSC_YVU9_YV12) {                                       // planar2x_c per slice (:2079-2093)
            plan.mode = 4;
            plan.nplanes = 3;
            plan.aux = sliceH >> 2;                                                        // source chroma rows of the slice
            plan.aux2 = c->chrSrcW;
            // planar2x writes 2*chrSrcW columns, one more than chrDstW when srcW % 4 is 1 or 2; only the visible ones are produced
            const int cwv = std::min(2 * c->chrSrcW, c->chrDstW);
            plan.pl[1] = { 1, 1, cwv, 2 * (sliceH >> 2), sliceY >> 1, 1, 0, 1 };
            plan.pl[0] = { 0, 0, p.srcW, sliceH, sliceY, 1, 0, 0 };
            plan.pl[2] = { 2, 2, cwv, 2 * (sliceH >> 2), sliceY >> 1, 1, 0, 1 };
        } else if (c->plan != PLAN_UNSC_PLANARCOPY) {
            plan.mode = c->plan == PLAN_UNSC_PLANAR2NV12 ? 0 : 1;
            plan.nplanes = 2;
            plan.pl[0] = { 0, 0, p.srcW, sliceH, sliceY, 1, 0, 0 };
            plan.pl[1] = { 1, 1, c->chrSrcW, (sliceH + 1) / 2, sliceY / 2, 1, 0, 1 };
        } else {
            plan.mode = 2;
            const PixDesc *ds = pix_desc(c->opts.src_format), *dd = pix_desc(c->opts.dst_format);
            // equal layouts are row memcpy in the reference, except 9..14-bit formats which always take the
            // depth-conversion branch (swscale_unscaled.c:2246-2248) and re-replicate the top bits
            const bool same = ds->comp[0].depth == dd->comp[0].depth && ds->comp[0].shift == dd->comp[0].shift &&
                              !isNBPS(c->opts.src_format) && !isNBPS(c->opts.dst_format);
            plan.bytecopy = same;
            const int np = pix_nb_planes(dd);
            plan.nplanes = np;
            for (int pl = 0; pl < np; pl++) {
                int len = pl == 0 ? p.srcW : -((-p.srcW) >> c->chrDstHSubSample);
                const int y0 = pl == 0 ? sliceY : -((-sliceY) >> c->chrDstVSubSample);
                const int h = pl == 0 ? sliceH : -((-sliceH) >> c->chrDstVSubSample);
                if (pl == 1 && isSemiPlanarYUV(c->opts.dst_format)) len *= 2;
                if (same) len *= (ds->comp[0].depth + 7) / 8;  // byte copy: width counts bytes
                const int shiftonly = pl == 1 || pl == 2 || (!c->opts.src_range && pl == 0);
                const bool missing = pl > 0 && isGray(c->opts.src_format);   // fillPlane / fillPlane16 (:2239-2247); width in samples
                if (missing && same) len /= (ds->comp[0].depth + 7) / 8;
                if (pl == 3) {   // alpha plane (:2226-2247): full size, converted like luma with shiftonly = 0 when the source has one, all ones otherwise
                    const int bytes = same ? (ds->comp[0].depth + 7) / 8 : 1;
                    plan.pl[pl] = { isALPHA(c->opts.src_format) ? 3 : -2, 3, isALPHA(c->opts.src_format) ? p.srcW * bytes : p.srcW, sliceH, sliceY, 1, 0, 0 };
                    continue;
                }
                plan.pl[pl] = { missing ? -1 : pl, pl, len, h, y0, 1, shiftonly, pl != 0 };
            }
        }
        for (int i = 0; i < plan.nplanes; i++) { maxw = std::max(maxw, plan.pl[i].width); rows += plan.pl[i].rows; }
        if (!maxw || !rows) break;
        const dim3 grid(cdiv(maxw, 256), rows, n);
        hipLaunchKernelGGL(swsk::sws_k_planar_misc, grid, blk, 0, st, fs, p, plan);
        break;
    }
    case PLAN_UNSC_RGB2RGB: {
        const PixDesc *ds = pix_desc(c->opts.src_format), *dd = pix_desc(c->opts.dst_format);
        swsk::ShufflePlan sp;
        std::memset(&sp, 0, sizeof(sp));
        sp.src_step = ds->comp[0].step; sp.dst_step = dd->comp[0].step;
        for (int k = 0; k < 4; k++) {
            sp.spos[k] = k < ds->nb_components ? ds->comp[k].offset : -1;
            sp.dpos[k] = k < dd->nb_components ? dd->comp[k].offset : -1;
        }
        // swscale.c:1106-1124: an rgb0-style source feeding a real alpha channel is made opaque first
        sp.opaque = c->src0Alpha && !c->dst0Alpha && isALPHA(c->opts.dst_format);
        const bool s3 = sp.src_step == 3, d3 = sp.dst_step == 3;
        static const int off3[4] = { 0, 3, 2, 1 };   // where pixel i of a 12-byte group starts inside its dword pair
        for (int i = 0; i < 4; i++) {
            uint32_t sel = 0;
            const int base = s3 ? off3[i] : 0;
            for (int j = 0; j < 4; j++) {
                uint32_t b = 0x0c;                    // unused byte -> 0
                for (int k = 0; k < 4; k++)
                    if (sp.dpos[k] == j) b = (k == 3 && (sp.spos[3] < 0 || sp.opaque)) ? 0x0du : (uint32_t)(base + sp.spos[k]);
                sel |= b << (8 * j);
            }
            sp.sel[i] = sel;
        }
        const dim3 grid(cdiv(cdiv(p.srcW, 4), 256), sliceH, n);
        if (s3 && d3) hipLaunchKernelGGL((swsk::sws_k_rgb_shuffle<true, true>), grid, blk, 0, st, fs, sp, p.srcW, sliceY);
        else if (s3) hipLaunchKernelGGL((swsk::sws_k_rgb_shuffle<true, false>), grid, blk, 0, st, fs, sp, p.srcW, sliceY);
        else if (d3) hipLaunchKernelGGL((swsk::sws_k_rgb_shuffle<false, true>), grid, blk, 0, st, fs, sp, p.srcW, sliceY);
        else hipLaunchKernelGGL((swsk::sws_k_rgb_shuffle<false, false>), grid, blk, 0, st, fs, sp, p.srcW, sliceY);
        break;
    }
    case PLAN_UNSC_YUV2GBRP: {
        const int dstW = p.dstW;
        const int npairs = ((dstW >> 3) << 2) + ((dstW & 4) ? 2 : 0) + ((dstW & 2) ? 1 : 0); // yuv2rgb.c:198-236
        const int nrowpairs = (sliceH + 1) >> 1;
        if (!npairs || !nrowpairs) break;
        const dim3 grid(cdiv(npairs, 256), nrowpairs, n);
        hipLaunchKernelGGL(swsk::sws_k_yuv2gbrp_unscaled, grid, blk, 0, st, fs, p, c->opts.src_format == AV_PIX_FMT_YUV422P ? 1 : 0, npairs, sliceY);
        break;
    }
    case PLAN_UNSC_RGB16SHUFFLE:
    case PLAN_UNSC_PACKED16_GBRP16:
    case PLAN_UNSC_GBRP16_PACKED16: {
        const PixDesc *ds = pix_desc(c->opts.src_format), *dd = pix_desc(c->opts.dst_format);
        swsk::Rgb16Plan rp;
        std::memset(&rp, 0, sizeof(rp));
        rp.mode = c->plan == PLAN_UNSC_RGB16SHUFFLE ? 0 : c->plan == PLAN_UNSC_PACKED16_GBRP16 ? 1 : 2;
        rp.sstep = ds->comp[0].step / 2; rp.dstep = dd->comp[0].step / 2;
        for (int k = 0; k < 3; k++) {
            rp.spos[k] = rp.mode == 2 ? ds->comp[k].plane : ds->comp[k].offset / 2;
            rp.dpos[k] = rp.mode == 1 ? dd->comp[k].plane : dd->comp[k].offset / 2;
        }
        rp.depth = rp.mode == 1 ? dd->comp[0].depth : ds->comp[0].depth;
        rp.src_alpha = isALPHA(c->opts.src_format) ? 1 : 0;
        rp.dst_alpha = (rp.mode == 1 && isALPHA(c->opts.dst_format)) ? 1 : 0;
        const dim3 grid(cdiv(p.srcW, 256), sliceH, n);
        hipLaunchKernelGGL(swsk::sws_k_rgb16_convert, grid, blk, 0, st, fs, rp, p.srcW, sliceY);
        break;
    }
    case PLAN_UNSC_U8_TO_F32:
    case PLAN_UNSC_F32_TO_U8: {
        const dim3 grid(cdiv(p.srcW, 256), sliceH, n);
        hipLaunchKernelGGL(swsk::sws_k_gray_f32, grid, blk, 0, st, fs, p.srcW, sliceY, c->plan == PLAN_UNSC_U8_TO_F32 ? 1 : 0);
        break;
    }
    case PLAN_UNSC_YUV2MONO: {
        const int nbytes = (p.dstW + 7) >> 3, nrowpairs = (sliceH + 1) >> 1;
        if (!nrowpairs) break;
        // g = table_gU[128] + table_gV[128] (yuv2rgb.c:460): the closed form's green index for U = V = 128
        const SwsLutParams &L = p.lut;
        const int gidx = L.base_g + (int)(((int64_t)128 * L.cgu) >> 16) + (int)(((int64_t)128 * L.cgv) >> 16);
        const dim3 grid(cdiv(nbytes, 256), nrowpairs, n);
        hipLaunchKernelGGL(swsk::sws_k_yuv2mono_unscaled, grid, blk, 0, st, fs, p, gidx, sliceY);
        break;
    }
    case PLAN_UNSC_RGB30_TO_16:
    case PLAN_UNSC_RGB30_TO_GBRP:
    case PLAN_UNSC_GBRP_TO_RGB30: {
        const PixDesc *ds = pix_desc(c->opts.src_format), *dd = pix_desc(c->opts.dst_format);
        swsk::Rgb30Plan rp;
        std::memset(&rp, 0, sizeof(rp));
        rp.mode = c->plan == PLAN_UNSC_RGB30_TO_16 ? 0 : c->plan == PLAN_UNSC_RGB30_TO_GBRP ? 1 : 2;
        rp.x2rgb = (rp.mode == 2 ? c->opts.dst_format : c->opts.src_format) == AV_PIX_FMT_X2RGB10LE;
        rp.dstep = dd->comp[0].step / 2;
        for (int k = 0; k < 3; k++) rp.pos[k] = rp.mode == 0 ? dd->comp[k].offset / 2 : rp.mode == 1 ? dd->comp[k].plane : ds->comp[k].plane;
        if (rp.mode == 1) { rp.hi = dd->comp[0].depth - 10; rp.lo = 10 - rp.hi; rp.shift = dd->comp[0].shift;
                            rp.dst_alpha = isALPHA(c->opts.dst_format) ? (int)((((1u << dd->comp[0].depth) - 1) << dd->comp[0].shift) & 0xFFFF) : 0; }
        if (rp.mode == 2) rp.shift = ds->comp[0].depth + ds->comp[0].shift - 10;
        const dim3 grid(cdiv(p.srcW, 256), sliceH, n);
        hipLaunchKernelGGL(swsk::sws_k_rgb30_convert, grid, blk, 0, st, fs, rp, p.srcW, sliceY);
        break;
    }
    case PLAN_UNSC_YUV2RGB48: {
        const int dstW = p.dstW;
        const int npairs = ((dstW >> 3) << 2) + ((dstW & 4) ? 2 : 0) + ((dstW & 2) ? 1 : 0); // yuv2rgb.c:198-236
        const int nrowpairs = (sliceH + 1) >> 1;
        if (!npairs || !nrowpairs) break;
        const dim3 grid(cdiv(npairs, 256), nrowpairs, n);
        hipLaunchKernelGGL(swsk::sws_k_yuv2rgb48_unscaled, grid, blk, 0, st, fs, p, c->opts.src_format == AV_PIX_FMT_YUV422P ? 1 : 0, npairs, sliceY);
        break;
    }
    case PLAN_UNSC_YUV2RGB16: {
        const int dstW = p.dstW;
        const int npairs = ((dstW >> 3) << 2) + ((dstW & 4) ? 2 : 0) + ((dstW & 2) ? 1 : 0); // yuv2rgb.c:198-236
        const int nrowpairs = (sliceH + 1) >> 1;
        if (!npairs || !nrowpairs) break;
        const dim3 grid(cdiv(npairs, 256), nrowpairs, n);
        hipLaunchKernelGGL(swsk::sws_k_yuv2rgb16_unscaled, grid, blk, 0, st, fs, p, c->opts.src_format == AV_PIX_FMT_YUV422P ? 1 : 0, npairs, sliceY);
        break;
    }
    case PLAN_UNSC_BAYER: {
        const int sf = c->opts.src_format, df = c->opts.dst_format;
        const int quad = sf == AV_PIX_FMT_BAYER_BGGR8 || sf == AV_PIX_FMT_BAYER_BGGR16LE || sf == AV_PIX_FMT_BAYER_RGGB8 || sf == AV_PIX_FMT_BAYER_RGGB16LE;
        const int rpos = (sf == AV_PIX_FMT_BAYER_BGGR8 || sf == AV_PIX_FMT_BAYER_BGGR16LE || sf == AV_PIX_FMT_BAYER_GBRG8 || sf == AV_PIX_FMT_BAYER_GBRG16LE) ? 0 : 2;
        const int sz = pix_desc(sf)->comp[0].step, mode = df == AV_PIX_FMT_YUV420P ? 2 : df == AV_PIX_FMT_RGB48LE ? 1 : 0;
        const int sh = (sz == 2 && mode != 1) ? 8 : 0;
        if (sliceH < 2) { log_msg(c, 0, "a bayer slice needs two rows\n"); return SWS_AVERROR(EINVAL); }
        const dim3 grid(cdiv((p.srcW + 1) / 2, 256), (sliceH + 1) / 2, n);
        hipLaunchKernelGGL(swsk::sws_k_bayer, grid, blk, 0, st, fs, p, p.srcW, sliceY, sliceH, quad, rpos, sz, sh, mode);
        break;
    }
    case PLAN_UNSC_PAL2RGB: {
        const int df = c->opts.dst_format;
        const bool planar = df == AV_PIX_FMT_GBRP || df == AV_PIX_FMT_GBRAP;
        const int nbytes = planar ? (df == AV_PIX_FMT_GBRAP ? 4 : 3) : pix_desc(df)->comp[0].step;
        if (sliceH <= 0) break;
        const dim3 grid(cdiv(p.srcW, 256), sliceH, n);
        const int gray = c->opts.src_format != AV_PIX_FMT_GRAY8 ? 0 : (df == AV_PIX_FMT_ARGB || df == AV_PIX_FMT_ABGR) ? 2 : 1;
        hipLaunchKernelGGL(swsk::sws_k_pal2rgb, grid, blk, 0, st, fs, p.srcW, sliceY, nbytes, planar ? 1 : 0, gray);
        break;
    }
    case PLAN_UNSC_YUV2RGB8: {
        const int dstW = p.dstW;
        const int npairs = ((dstW >> 3) << 2) + ((dstW & 4) ? 2 : 0) + ((dstW & 2) ? 1 : 0); // yuv2rgb.c:198-236
        const int nrowpairs = (sliceH + 1) >> 1;
        if (!npairs || !nrowpairs) break;
        const dim3 grid(cdiv(npairs, 256), nrowpairs, n);
        hipLaunchKernelGGL(swsk::sws_k_yuv2rgb8_unscaled, grid, blk, 0, st, fs, p, c->opts.src_format == AV_PIX_FMT_YUV422P ? 1 : 0, npairs, sliceY);
        break;
    }
    case PLAN_UNSC_RGBLOW: {
        const int sf = c->opts.src_format, df = c->opts.dst_format;
        auto rgbint = [](int f) { return f == AV_PIX_FMT_RGB24 || f == AV_PIX_FMT_BGRA || f == AV_PIX_FMT_ABGR || f == AV_PIX_FMT_RGB565LE ||
                                         f == AV_PIX_FMT_RGB555LE || f == AV_PIX_FMT_RGB444LE; };
        swsk::RgbLowPlan rp;
        rp.sid = pix_bits_per_pixel(pix_desc(sf)); rp.did = pix_bits_per_pixel(pix_desc(df));
        rp.same = rgbint(sf) == rgbint(df) ? 1 : 0;
        rp.s_alt = (sf == AV_PIX_FMT_ABGR || sf == AV_PIX_FMT_ARGB) ? 1 : 0;
        rp.d_alt = (df == AV_PIX_FMT_ABGR || df == AV_PIX_FMT_ARGB) ? 1 : 0;
        if (!p.srcW || !sliceH) break;
        const dim3 grid(cdiv(p.srcW, 256), sliceH, n);
        hipLaunchKernelGGL(swsk::sws_k_rgb_low_convert, grid, blk, 0, st, fs, rp, p.srcW, sliceY);
        break;
    }
    case PLAN_UNSC_PLANAR2P422: {
        const int npairs = p.srcW >> 1;
        if (!npairs || !sliceH) break;
        const dim3 grid(cdiv(npairs, 256), sliceH, n);
        hipLaunchKernelGGL(swsk::sws_k_planar_to_p422, grid, blk, 0, st, fs, p, npairs, sliceY, c->opts.src_format == AV_PIX_FMT_YUV422P ? 1 : 2);
        break;
    }
    case PLAN_UNSC_PLANARRGB_PLANARRGB: {   // three plane copies (ff_copyPlane with its byte / pixel mix-up), the alpha plane filled if there is one
        if (!sliceH) break;
        const int bps = p.dst_bits > 8 ? 2 : 1, row_bytes = p.srcW * bps;
        for (int pl = 0; pl < 3; pl++) {
            const dim3 grid(cdiv(row_bytes, 256), sliceH, n);
            hipLaunchKernelGGL(swsk::sws_k_planarrgb_copy, grid, blk, 0, st, fs, pl, p.srcW, row_bytes, sliceY, sliceH);
        }
        if (isALPHA(c->opts.dst_format)) launch_fill_alpha(L, p.srcW, sliceY, sliceH, p.dst_bits > 8 ? p.dst_bits : 0);
        break;
    }
    case PLAN_UNSC_ALPHABLEND: {   // ff_sws_alphablendaway: one launch per plane (planar) or one for the packed picture
        const PixDesc *ds = pix_desc(c->opts.src_format);
        swsk::AlphaBlendPlan ap;
        std::memset(&ap, 0, sizeof(ap));
        ap.planar = (ds->flags & PIXFLAG_PLANAR) ? 1 : 0;
        ap.plane_count = isGray(c->opts.src_format) ? 1 : 3;
        ap.depth = ds->comp[0].depth;
        ap.alpha_pos = ds->comp[ap.plane_count].offset;
        ap.lum_w = p.srcW; ap.lum_h = p.srcH; ap.lw = ds->log2_chroma_w; ap.lh = ds->log2_chroma_h;
        for (int pl = 0; pl < ap.plane_count; pl++) {
            int a = 0, b = 0;
            if (c->opts.alpha_blend == SWS_ALPHA_BLEND_CHECKERBOARD) { a = (1 << (ap.depth - 1)) / 2; b = 3 * (1 << (ap.depth - 1)) / 2; }
            const bool mid = pl && !(ds->flags & PIXFLAG_RGB);
            ap.target[0][pl] = mid ? 1 << (ap.depth - 1) : a;
            ap.target[1][pl] = mid ? 1 << (ap.depth - 1) : b;
        }
        if (!sliceH) break;
        if (ap.planar) {
            for (int pl = 0; pl < ap.plane_count; pl++) {
                const int xs = pl ? ap.lw : 0, ys = pl ? ap.lh : 0;
                // (alphablend.c:52: "w = plane ? c->chrSrcW : c->opts.src_w" -- for a gbrap source that is half the width when init halved the
                //  RGB chroma, utils.c:1369-1390 with SWS_FAST_BILINEAR: the right half of planes 1 and 2 is then left alone, as there)
                const int w = pl ? c->chrSrcW : p.srcW, rows = -((-sliceH) >> ys);
                const dim3 grid(cdiv(w, 256), rows, n);
                hipLaunchKernelGGL(swsk::sws_k_alphablend, grid, blk, 0, st, fs, ap, pl, w, sliceY >> ys);
            }
        } else {
            const dim3 grid(cdiv(p.srcW, 256), sliceH, n);
            hipLaunchKernelGGL(swsk::sws_k_alphablend, grid, blk, 0, st, fs, ap, 0, p.srcW, sliceY);
        }
        break;
    }
    case PLAN_UNSC_P4222PLANAR: {
        const dim3 grid(cdiv((p.srcW + 1) >> 1, 256), sliceH, n);
        hipLaunchKernelGGL(swsk::sws_k_p422_to_planar, grid, blk, 0, st, fs, p, p.srcW, sliceY, c->opts.dst_format != AV_PIX_FMT_YUV422P ? 1 : 0);
        break;
    }
    case PLAN_UNSC_PACKED_GBRP: {
        const PixDesc *ds = pix_desc(c->opts.src_format);
        swsk::ShufflePlan sp;
        std::memset(&sp, 0, sizeof(sp));
        sp.src_step = ds->comp[0].step;
        for (int k = 0; k < 4; k++) sp.spos[k] = k < ds->nb_components ? ds->comp[k].offset : -1;
        if (c->opts.dst_format == AV_PIX_FMT_GBRAP) {   // rgbToPlanarRgbaWrapper; an rgb0-style source was made opaque first (swscale.c:1106-1124)
            const bool opaque = c->src0Alpha && !c->dst0Alpha;
            sp.planar_alpha = (ds->comp[0].step == 4 && !opaque) ? 1 : 2;
            if (ds->comp[0].step == 4) sp.spos[3] = (ds->comp[0].offset == 1 || ds->comp[2].offset == 1) ? 0 : 3;
        }
        const dim3 grid(cdiv(p.srcW, 256), sliceH, n);
        hipLaunchKernelGGL(swsk::sws_k_packed_to_gbrp, grid, blk, 0, st, fs, sp, p.srcW, sliceY);
        break;
    }
    case PLAN_UNSC_GBRP_PACKED: {
        const PixDesc *dd = pix_desc(c->opts.dst_format);
        swsk::ShufflePlan sp;
        std::memset(&sp, 0, sizeof(sp));
        for (int k = 0; k < 4; k++) sp.dpos[k] = k < dd->nb_components ? dd->comp[k].offset : -1;
        if (dd->comp[0].step == 4) sp.dpos[3] = (dd->comp[0].offset == 1 || dd->comp[2].offset == 1) ? 0 : 3;
        sp.planar_alpha = c->opts.src_format == AV_PIX_FMT_GBRAP ? 1 : 0;   // planarRgbaToRgbWrapper
        const dim3 grid(cdiv(cdiv(p.srcW, 4), 256), sliceH, n);
        if (dd->comp[0].step == 3) hipLaunchKernelGGL((swsk::sws_k_gbrp_to_packed<true>), grid, blk, 0, st, fs, sp, p.srcW, sliceY);
        else hipLaunchKernelGGL((swsk::sws_k_gbrp_to_packed<false>), grid, blk, 0, st, fs, sp, p.srcW, sliceY);
        break;
    }
    case PLAN_UNSC_BGR24_YV12: {
        // (16-byte aligned frames: whole groups of 8 chroma columns through the vector form, the columns behind them -- and everything else -- through the scalar one)
        bool fits16 = true;     // (the vector form's v_dot2 operands: every table of sws_setColorspaceDetails() built from sane coefficients does)
        for (int k : { 1, 2, 4, 5, 7, 8 }) fits16 = fits16 && p.rgb2yuv[k] >= -32768 && p.rgb2yuv[k] <= 32767;
        for (int k : { 0, 3, 6 }) fits16 = fits16 && p.rgb2yuv[k] >= -(1 << 22) && p.rgb2yuv[k] < (1 << 22);
        const int cw = p.srcW >> 1, cgroups = (L.vec && fits16 && !c->tune.no_wave) ? cw / 8 : 0, cbase = cgroups * 8;
        if (cgroups) {
            const dim3 grid(cdiv(cgroups, 256), (sliceH + 1) / 2, n);
            hipLaunchKernelGGL(swsk::sws_k_bgr24_to_yv12_vec, grid, blk, 0, st, fs, p, sliceY, sliceH, cgroups);
        }
        if (cw > cbase) {
            const dim3 grid(cdiv(cdiv(cw - cbase, 4), 256), (sliceH + 1) / 2, n);
            hipLaunchKernelGGL(swsk::sws_k_bgr24_to_yv12, grid, blk, 0, st, fs, p, sliceY, sliceH, cbase);
        }
        break;
    }
    case PLAN_UNSC_PACKEDCOPY: {
        const PixDesc *ds = pix_desc(c->opts.src_format);
        // the reference copies as many multiples of src_w bytes as fit into both strides (:2138-2157), i.e. the whole visible row:
        // for the packed 4:2:2 layouts that is a whole number of pixel pairs
        const int row_bytes = p.srcKind == SRCK_MONO ? (p.srcW + 7) >> 3 :
                              ds->log2_chroma_w ? ((p.srcW + 1) >> 1) * 2 * ds->comp[0].step : p.srcW * ds->comp[0].step;
        const bool opaque = c->src0Alpha && !c->dst0Alpha && isALPHA(c->opts.dst_format);
        const dim3 grid(cdiv(cdiv(row_bytes, 16), 256), sliceH, n);
        hipLaunchKernelGGL(swsk::sws_k_packed_copy, grid, blk, 0, st, fs, row_bytes, sliceY, opaque ? ds->comp[3].offset : -1);
        break;
    }
    default:
        return SWS_AVERROR(EINVAL);
    }
    return 0;
}

void launch_fill_alpha(const LaunchCtx &L, int w, int y0, int rows, int bits)
{
    if (w <= 0 || rows <= 0) return;
    const dim3 g(cdiv(w, 256), rows, L.n);
    hipLaunchKernelGGL(swsk::sws_k_fill_alpha_plane, g, dim3(256), 0, L.st, L.fs, w, y0, bits);
}

void launch_alpha_merge(const LaunchCtx &L, int npix, int y0, int rows, int a_pos)
{
    if (npix <= 0 || rows <= 0) return;
    const dim3 g(cdiv(npix, 256), rows, L.n);
    hipLaunchKernelGGL(swsk::sws_k_alpha_merge, g, dim3(256), 0, L.st, L.fs, npix, y0, a_pos);
}

void launch_bswap(hipStream_t st, const uint8_t *src, int64_t sstride, uint8_t *dst, int64_t dstride, int rows, int row_bytes, int unit)
{
    if (rows <= 0 || row_bytes < unit) return;
    const dim3 grid((row_bytes / unit + 255) / 256, rows);
    hipLaunchKernelGGL(swsk::sws_k_bswap, grid, dim3(256), 0, st, src, sstride, dst, dstride, rows, row_bytes, unit);
}

void launch_gamma_rgba64(hipStream_t st, uint8_t *img, int64_t stride, int w, int rows, const uint16_t *table)
{
    if (w <= 0 || rows <= 0) return;
    const dim3 grid(cdiv(w, 256), rows);
    hipLaunchKernelGGL(swsk::sws_k_gamma_rgba64, grid, dim3(256), 0, st, img, stride, w, rows, table);
}
void launch_ed_mono(hipStream_t st, const uint8_t *lum, int64_t lumStride, uint8_t *dst, int64_t dstStride, int n, int h, int *errline, int white)
{
    hipLaunchKernelGGL(swsk::sws_k_ed_mono, dim3(1), dim3(1024), 0, st, lum, lumStride, dst, dstStride, n, h, errline, white);
}
void launch_update_palette(const LaunchCtx &L)
{
    hipLaunchKernelGGL(swsk::sws_k_update_palette, dim3(L.n), dim3(256), 0, L.st, L.fs, *L.p, L.c->opts.src_format, L.c->opts.dst_format);
}
void launch_ed_rgb8(hipStream_t st, const uint8_t *rgb, int64_t rgbStride, uint8_t *dst, int64_t dstStride, int w, int h, int *errline,
                    int bpp8, int r8, int g8, int b8)
{
    hipLaunchKernelGGL(swsk::sws_k_ed_rgb8, dim3(1), dim3(1024), 0, st, rgb, rgbStride, dst, dstStride, w, h, errline, bpp8, r8, g8, b8);
}

void launch_xyz12(hipStream_t st, const uint8_t *src, int64_t sstride, uint8_t *dst, int64_t dstride, int w, int rows,
                  const uint16_t *gamma_in, const uint16_t *gamma_out, int to_rgb)
{
    if (w <= 0 || rows <= 0) return;
    const dim3 grid(cdiv(w, 256), rows);
    hipLaunchKernelGGL(swsk::sws_k_xyz12, grid, dim3(256), 0, st, src, sstride, dst, dstride, w, gamma_in, gamma_out, to_rgb);
}

} // namespace swship
