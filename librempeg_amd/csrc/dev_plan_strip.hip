// Planner stage 3 (PLAN_MAIN): the dot2 tile kernel and the marching-strip family -- which conversions they take (the reference's line functions they restate:
// swscale.c:69-159 hScale*, output.c yuv2planeX / yuv2nv12cX / yuv2p01x / the LUT and full-chroma RGB writers, vscale.c:109-171), strip geometry, plan rows,
// the LDS-DMA forms, and what the helper passes fall back to when no strip plan fits.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <map>
#include <mutex>
#include <thread>
#include <vector>
#include "generic_kinds.hpp"
#include "../../include/hwcontext_hip.h"
#include "dev_plan.hpp"

namespace swship {

int plan_strip(PlanBuild &B)
{
    PLAN_LOCALS(B);
    // ---- dot2 tile kernel (sws_k_tile_dot2): planar 8-bit / <= 15-bit sources, 15-bit intermediates, vfs >= 2 ----
    d->dot2_ok = false;
    {
        const bool vlines_pending = d->vlines_on;
        const bool src_ok = p.srcKind == SRCK_PLANAR8 || (p.srcKind == SRCK_PLANAR16 && p.src_depth <= 15 && p.src_shift == 0);
        // nv12 / nv21, p010 / p012 (and the 4:2:2 / 4:4:4 twins): the strip kernel de-interleaves plane 1 (and shifts the p01x samples down) while
        // staging; the dot2 tile kernel does not
        const bool nv_src = (p.srcKind == SRCK_NV12 && c->srcBpc == 8) || (p.srcKind == SRCK_P010 && p.src_depth <= 15);
        // (round 5: samples of 16 significant bits -- yuv4xxp16, gray16, p016 / p216 / p416 -- through the register-staged strip kernels: top bit flipped while
        //  staging, the difference given back as a per-column addend, strip_hstage_b)
        const bool src_u16 = !c->tune.no_strip_u16 && !c->tune.no_strip && c->srcBpc == 16 && p.src_depth == 16 && p.src_shift == 0 && (p.srcKind == SRCK_PLANAR16 || p.srcKind == SRCK_P010);
        // (19-bit intermediates, round 5: destinations of 16 bits per component -- yuv4xxp16, gray16, p016 -- and the int32 sums of the wide planar RGB route,
        //  from sources whose samples are v_dot2 operands as they are: sws_k_strip_wide, k_stripwide.hip)
        const bool wide_dst = p.wide && c->dstBpc >= 16 && !c->tune.no_strip_wide && !c->tune.no_strip &&
                              ((p.dstKind == DSTK_PLANAR16 && p.dst_shift == 0 && !isALPHA(o.dst_format)) || p.dstKind == DSTK_P016 || (p.dstKind == DSTK_RAW32 && d->fullchr_on == 4));
        const bool dst_ok = ((p.dstKind == DSTK_PLANAR8 || p.dstKind == DSTK_PLANARN || p.dstKind == DSTK_NV12 || p.dstKind == DSTK_P010 || p.dstKind == DSTK_RAW32) && !p.wide) || wide_dst;
        auto fs2 = [](int fs) { return (fs + 2) & ~1; };
        // packed 24 / 32 bpp RGB through the LUT writers (not the full-chroma ones): the strip kernel with the RGB epilogue
        // (9 .. 15-bit planar sources too -- decoded HDR pictures for display: 128-column strips, a window of at most 64 eight-sample chunks)
        const bool rgb_s16 = p.srcKind == SRCK_PLANAR16 && p.src_depth <= 15 && p.src_shift == 0;
        // (a planar YUV source with alpha into a 32 bpp destination with alpha -- yuva420p -> bgra, needAlpha: the kernel stores opaque pixels, the A
        //  samples go through one more luma launch with the raw writer and sws_k_alpha_merge32 puts the bytes in: alpha_launch == 2)
        const bool rgb_alpha = c->needAlpha && p.dstKind == DSTK_RGB32 && isPlanarYUV(o.src_format) && !c->tune.no_mixed;
        const bool rgb_ok = (p.dstKind == DSTK_RGB24 || p.dstKind == DSTK_RGB32) && !p.full_chr && ((p.srcKind == SRCK_PLANAR8 && c->srcBpc == 8) || rgb_s16) &&
                            !p.no_chroma && (!p.need_alpha || rgb_alpha) && !(p.dstW & 1) && !p.range_active && !c->tune.no_strip;
        d->striprgb_ok = false; d->striprgb_direct = 0;
        // (planar writers: a one-tap vertical filter takes the reference's yuv2plane1 form -- (s + d) >> 7, (s + (1 << (14 - bits))) >> (15 - bits),
        //  output.c:327-341, :485-493 -- which is the "X" arithmetic of these kernels with the one tap 4096: (4096 s + (d << 12)) >> 19, sample for
        //  sample; the strip kernel takes such planes (4:2:0 -> 4:2:2 at half the size, horizontal-only scaling), the dot2 tile kernel keeps its
        //  two-tap minimum.  The packed writers are in "X" mode unless both vertical filters are short, which all_x_mode checks row by row)
        // scaled packed 24 / 32 bpp RGB sources: a reader pre-pass writes the 16-bit planes the horizontal scaler
        // reads, the strip kernel takes them like a planar 16-bit source (launch_rgbread_strip); other shapes of these sources keep the tile kernel
        // (round 4: the other RGB sources whose readers deliver the same 15-bit lines to the same hScale16To15_c -- x2rgb10 / x2bgr10, the 16 / 15 / 12 bpp
        //  formats (rgb16_32ToY/UV(_half)_c_template, input.c:264-412), planar RGB of 9 - 14 bits (planar_rgb16_s16_to_y / _uv, :1216-1270) -- through the
        //  per-kind element-per-thread reader (k_generic_kinds.hip sws_k_read16_kind); without an alpha plane)
        //  (... and the packed YUV sources of 10 / 12 bits -- y210 / y212, xv30 / v30x, xv36: read_*_c, y21xle_Y/UV_c, input.c:580-606, :663-729, :811-866 -- whose
        //  lines are those of a planar yuv422p10 / yuv444p10 / ...12 picture, sh = depth - 1: the pre-pass de-interleaves them)
        const bool rgbread_kindN = (p.srcKind == SRCK_RGB30 || p.srcKind == SRCK_RGB16 || (p.srcKind == SRCK_GBRP16 && (p.src_depth < 16 || !c->tune.no_strip_u16)) ||
                                    // (round 5: rgb48 / rgba64, planar RGB of 16 bits and float32 -- lines of 16 significant bits, see src_u16)
                                    ((p.srcKind == SRCK_RGB48 || p.srcKind == SRCK_GBRPF32) && !c->tune.no_strip_u16) ||
                                    (p.srcKind == SRCK_PACKEDHI && p.src_depth >= 9 && p.src_depth <= 15 && c->srcBpc == p.src_depth) ||
                                    // (the 8-bit packed 4:4:4 formats -- ayuv / vuya / vuyx / uyva / vyu444: bytes, hScale8To15_c's sh = 7 -- as 16-bit words with 8 significant bits)
                                    (p.srcKind == SRCK_PACKED444 && p.src_depth == 8 && c->srcBpc == 8)) && !p.need_alpha && !c->needAlpha &&   // (an alpha component nobody reads is skipped)
                                   !c->tune.no_rgbread_kinds;
        bool rgbread = (!p.wide || !c->tune.no_strip_wide) && (p.srcKind == SRCK_RGB24 || p.srcKind == SRCK_RGB32 || p.srcKind == SRCK_GBRP || rgbread_kindN) && p.chrSrcHSub <= 1 && p.chrSrcW == (p.srcW >> p.chrSrcHSub) && !(p.srcW & 1) && p.chrSrcVSub == 0 && (!p.need_alpha || d->fullchr_on == 2 || alpha_planar) &&
                       (!p.dst_alpha_fill || d->fullchr_on) && (!p.no_chroma || (isGray(o.dst_format) && !isGray(o.src_format) && !c->tune.no_strip_range)) && !vlines_pending && dst_ok && !c->tune.no_strip && !c->tune.no_rgbsrc && p.dstW >= strip_min_w_eff;
        for (int k = 0; k < 9 && rgbread; k++) rgbread = p.rgb2yuv[k] > -32768 && p.rgb2yuv[k] < 32768;   // (v_dot2_i32_i16 operands)
        d->rgbread_on = false;
        // vertical chroma filters of 17 .. 24 taps (a 4:1 chroma step: packed RGB or 4:2:2 sources into a 4:2:0 picture of half the size): the strip
        // kernel's chroma instantiations with a ring of 12 row pairs; the tile kernel and the RGB epilogue stop at 16
        const bool vchr_long = fs2(vChrB.size) > 16 && fs2(vChrB.size) <= 24 && dst_ok && !rgb_ok && !c->tune.no_strip;
        // gray -> gray (8 .. 14 bit): one plane, the strip kernel's luma launch alone
        // (round 5: ... and planar / semi-planar YUV -> gray: the destination has no chroma planes, so the conversion is the luma launch as well -- thumbnails
        //  for analysis; it needed the range conversion in the strip kernels, gray8 being full range, handle_jpeg utils.c:773-809)
        // (round 5: ... and gray sources into planar / semi-planar YUV: the luma launch, then sws_k_gray_chroma writes what the reference's chroma writers make of
        //  their constant lines -- launch_plan_le_batch)
        const bool gray_src = isGray(o.src_format) && !isGray(o.dst_format) && !c->needAlpha && (!isALPHA(o.dst_format) || (p.dstKind == DSTK_RAW32 && (d->fullchr_on == 3 || d->fullchr_on == 1))) && (src_ok || src_u16) && !c->tune.no_strip_range &&
                              (((p.dstKind == DSTK_PLANAR8 || p.dstKind == DSTK_PLANARN || p.dstKind == DSTK_NV12 || p.dstKind == DSTK_P010) && !p.wide) ||
                               ((p.dstKind == DSTK_PLANAR16 || p.dstKind == DSTK_P016) && wide_dst) || (p.dstKind == DSTK_RAW32 && (d->fullchr_on == 3 || d->fullchr_on == 1) && d->fullchr_kind != DSTK_GBRP)) && !d->join422 &&
                              (!d->fullchr_on || d->fullchr_on == 3 || d->fullchr_on == 1) && !c->tune.no_strip;
        const bool gray_both = gray_src || (isGray(o.dst_format) && (isGray(o.src_format) || ((src_ok || nv_src || src_u16 || rgbread) && !c->tune.no_strip_range)) && !c->needAlpha && (src_ok || nv_src || src_u16 || rgbread) &&
                               (((p.dstKind == DSTK_PLANAR8 || p.dstKind == DSTK_PLANARN) && !p.wide) || (p.dstKind == DSTK_PLANAR16 && wide_dst)) && !c->tune.no_strip);
        // identity luma filters + scaled chroma (yuv422p -> yuv420p, yuv444p -> yuv420p, the 10-bit -> 8-bit twins ...): the luma plane streams
        // (one tap: a per-sample pass), only the chroma planes need the strip kernel
        const bool mixedM = !vlines_pending && !d->fullchr_on && bank_is_identity(hLumB, 1 << 14) && bank_is_identity(c->vLum, 1 << 12) && !(d->unity_h && d->unity_v) &&
                            !p.fast_bilinear && !gray_any && (src_ok || nv_src) && dst_ok && !p.wide && (!p.range_active || (c->srcBpc == 8 && p.dst_bits == 8 && !c->tune.no_strip_range)) && !p.dst_alpha_fill &&
                            fs2(hChrB.size) <= 16 && (fs2(vChrB.size) <= 16 || vchr_long) && !c->tune.no_strip && !c->tune.no_mixed && p.dstW >= c->tune.strip_min_w;
        // (identity horizontal filters: 8-bit sources have kernels of their own -- sws_k_rgb_march, sws_k_rgbsrc_unity, the mixed plan -- but a 10-bit
        //  picture into packed RGB (decoded HDR for display) or packed RGB into a 10-bit 4:2:0 picture at the same size had only the generic
        //  kernels: the strip kernels take them with their one-tap horizontal banks)
        // (... and so had every planar / semi-planar YUV -> YUV conversion whose horizontal filters are the identity and which the mixed plan does not
        //  take: all four filters the identity -- p010le -> yuv420p10le, nv12 -> yuv420p10le, yuv420p10le -> nv12, nv12 <-> nv21, bgra -> yuv444p10le: pure
        //  per-sample conversions the reference has no special converter for -- or vertical-only scaling)
        const bool unity_yuv = d->unity_h && !d->rgbsrc_ok && !d->rgb444_ok && dst_ok && (src_ok || nv_src || rgbread || src_u16) && !c->tune.no_mixed;
        const bool unity_ok = !d->unity_h || (rgb_ok && rgb_s16) || (rgbread && !d->rgbsrc_ok && !d->rgb444_ok) || unity_yuv;   // (sws_k_rgbsrc_unity's row table lives in the same device block as the strip plan)
        // filters of 17 .. 32 taps (ratios of 4:1 and more -- the lower rungs of an ABR ladder, thumbnails: bicubic at 4:1 has 17 taps, at 6:1 25; Lanczos at
        // 3:1 19): the strip kernel's long form (sws_k_strip_long: 16 tap pairs each way, strips of 128 / 64 columns); the RGB epilogue stops at 16
        const bool fs_ok16 = fs2(hLumB.size) <= 16 && (fs2(hChrB.size) <= 16 || gray_both) && fs2(c->vLum.size) <= 16 && (fs2(vChrB.size) <= 16 || vchr_long || gray_both);
        const bool fs_ok32 = fs2(hLumB.size) <= 32 && fs2(hChrB.size) <= 32 && fs2(c->vLum.size) <= 32 && fs2(vChrB.size) <= 48 && dst_ok && !rgb_ok && !gray_both &&
                             !c->tune.no_strip && !c->tune.no_mixed;
        const bool fs_ok64 = fs2(hLumB.size) <= 64 && fs2(hChrB.size) <= 64 && fs2(c->vLum.size) <= 64 && fs2(vChrB.size) <= 64 && dst_ok && !rgb_ok && !gray_both &&
                             !c->tune.no_strip && !c->tune.no_mixed;      // (33 .. 62 taps: the extra-long form, 32 pairs each way on strips of 64 columns)
        const bool wide_ok = wide_dst && (src_ok || nv_src || src_u16 || rgbread) && !(p.range_active && c->tune.no_strip_range) && !c->needAlpha && !p.need_alpha && !p.fast_bilinear && !vlines_pending && fs_ok16 &&
                             !(p.srcKind == SRCK_PLANAR8 && c->srcBpc != 8);
        const bool fullA = !mixedM && unity_ok && !(d->unity_h && d->unity_v && !rgb_ok && !unity_yuv) && !p.fast_bilinear && (!gray_any || gray_both || d->fullchr_on == 2 || alpha_planar || (rgb_ok && rgb_alpha)) && (src_ok || (nv_src && dst_ok) || rgbread || (src_u16 && dst_ok)) &&
                           (dst_ok || rgb_ok) && (!p.wide || wide_ok) && (fs_ok16 || fs_ok32 || fs_ok64) && !c->tune.no_dot2;
        const int long_form = !fullA || fs_ok16 ? 0 : fs_ok32 ? 1 : 2;
        d->mixed_ok = false; d->stripLs_ok = d->stripCs_ok = false; d->striprgbsrc_ok = false; d->rgb2rgb_ok = false;
        // (a gray source into planar / semi-planar YUV at the same size -- a monochrome camera into an encoder: the luma plane is the mixed plan's streaming pass, the
        //  chroma planes are sws_k_gray_chroma's constants; no strip plan at all.  launch_mixed tells the two by the source format)
        const bool gray_mixed = gray_src && !vlines_pending && !d->fullchr_on && bank_is_identity(hLumB, 1 << 14) && bank_is_identity(c->vLum, 1 << 12) && !p.fast_bilinear && src_ok && dst_ok &&
                                p.dstKind != DSTK_RAW32 && !p.wide && (!p.range_active || (c->srcBpc == 8 && p.dst_bits == 8)) && !p.dst_alpha_fill && !c->tune.no_mixed;
        if (gray_mixed) d->mixed_ok = true;
        else if (fullA || mixedM) {
            const int SPC = (p.srcKind == SRCK_PLANAR16 || p.srcKind == SRCK_P010 || rgbread) ? 8 : 16;
            std::vector<uint8_t> blob;
            auto put = [&](const void *ptr, size_t n) { size_t o = (blob.size() + 15) & ~(size_t)15; blob.resize(o + n); std::memcpy(blob.data() + o, ptr, n); return o; };
            auto padded = [&](const FilterBank &b, int f2_long = 0) {
                const int f2 = f2_long ? f2_long : fs2(b.size);
                std::vector<int16_t> t((size_t)b.count * f2, 0);
                for (int i = 0; i < b.count; i++)
                    for (int j = 0; j < b.size; j++) t[(size_t)i * f2 + (b.pos[i] & 1) + j] = b.taps[(size_t)i * b.size + j];
                return t;
            };
            struct Off { size_t rs, rc, cs, cc, ht, vt; };
            auto plan2 = [&](const FilterBank &hb, const FilterBank &vb, int W, int H, int ncomp, SwsTileGeom &g, Off &o) -> bool {
                const int TW = 128, hf2 = fs2(hb.size), vf2 = fs2(vb.size);
                for (int TH : { 64, 32, 16, 8, 4, 2 }) {
                    const int tX = (W + TW - 1) / TW, tY = (H + TH - 1) / TH;
                    std::vector<int32_t> rs(tY), rc(tY), cs(tX), cc(tX);
                    int nrmax = 0, ncmax = 0;
                    for (int t = 0; t < tY; t++) {
                        int lo = INT32_MAX, hi = -1;
                        for (int y = t * TH; y < std::min(H, (t + 1) * TH); y++) { lo = std::min(lo, vb.pos[y] & ~1); hi = std::max(hi, (vb.pos[y] & ~1) + vf2); }
                        rs[t] = lo; rc[t] = (hi - lo + 1) & ~1; nrmax = std::max(nrmax, rc[t]);
                    }
                    for (int t = 0; t < tX; t++) {
                        int lo = INT32_MAX, hi = -1;
                        for (int x = t * TW; x < std::min(W, (t + 1) * TW); x++) { lo = std::min(lo, hb.pos[x] & ~1); hi = std::max(hi, (hb.pos[x] & ~1) + hf2); }
                        lo = lo / SPC * SPC;
                        cs[t] = lo; cc[t] = (hi - lo + SPC - 1) / SPC * SPC; ncmax = std::max(ncmax, cc[t]);
                    }
                    const size_t lds = (size_t)nrmax * ncmax * 2 + (size_t)ncomp * (nrmax / 2) * TW * 4;
                    const int lds_budget = c->tune.tile_lds_kb;
                    if (lds > (size_t)lds_budget * 1024 && TH > 2) continue;
                    if (lds > 64 * 1024) return false;
                    g.TW = TW; g.TH = TH; g.tilesX = tX; g.tilesY = tY; g.NRmax = nrmax; g.NCmax = ncmax; g.lds_bytes = (int32_t)lds;
                    g.hfs2 = hf2; g.vfs2 = vf2;
                    g.debug = c->tune.debug;
                    o.rs = put(rs.data(), rs.size() * 4); o.rc = put(rc.data(), rc.size() * 4);
                    o.cs = put(cs.data(), cs.size() * 4); o.cc = put(cc.data(), cc.size() * 4);
                    const std::vector<int16_t> ht = padded(hb), vt = padded(vb);
                    o.ht = put(ht.data(), ht.size() * 2); o.vt = put(vt.data(), vt.size() * 2);
                    return true;
                }
                return false;
            };
            // marching strip kernel: strips of 64 * cols output columns; window of a strip <= 128 chunks of 16 bytes
            struct SOff { size_t cs, cc, rows; };
            // ring_of(npv) != 0: the kernel multiplies a whole ring of that depth per output sample (sws_k_strip_rgb): the row's tap pairs
            // are laid out against the newest npv slots, the older slots get zero taps
            // plane1_form: the plane's writer has the reference's one-tap form (yuv2plane1_*, yuv2p01xl1_c: (s + d) >> 7 and its N-bit twins), which
            // never looks at the coefficient -- initFilter's error-diffused normalisation leaves 4095 in some one-tap rows -- so a one-tap bank
            // enters the X arithmetic as 4096; the semi-planar chroma writers (yuv2nv12cX_c, yuv2p01xcX_c) have no such form and take the bank's value
            const std::vector<int32_t> *plan_rnd = nullptr;      // per output row: SwsStripRow::rnd_off of the plans made while it is set (the packed writers' short forms below)
            auto plan3 = [&](const FilterBank &hb, const FilterBank &vb, int W, int cols, int ncomp, SwsStripGeom &g, SOff &o, int (*ring_of)(int) = nullptr, bool plane1_form = false, int longf = 0, int tw_over = 0) -> bool {   // longf: 1 the long form (16 / 24 pairs), 2 the extra-long one (32 / 32); tw_over: strips narrower than 64 * cols columns (the lockstep kernels: a window that fits one reader turn less)
                const int TW = tw_over ? tw_over : 64 * cols, hf2 = fs2(hb.size), vf2 = fs2(vb.size);
                const int strips = (W + TW - 1) / TW;
                std::vector<int32_t> cs(strips), cc(strips);
                int ncmax = 0, nph = 1, npv = 1;
                for (int t = 0; t < strips; t++) {
                    int lo = INT32_MAX, hi = -1;
                    for (int x = t * TW; x < std::min(W, (t + 1) * TW); x++) { lo = std::min(lo, hb.pos[x] & ~1); hi = std::max(hi, (hb.pos[x] & ~1) + hf2); }
                    if (lo < 0) return false;
                    lo = lo / SPC * SPC;
                    cs[t] = lo; cc[t] = (hi - lo + SPC - 1) / SPC * SPC; ncmax = std::max(ncmax, cc[t]);
                }
                if (ncmax / SPC > (ncomp == 2 ? 64 : 128)) return false;   // one (chroma) or two (luma) 16-byte chunks per lane and row
                for (int x = 0; x < hb.count; x++) nph = std::max(nph, ((hb.pos[x] & 1) + hb.size + 1) / 2);
                for (int y = 0; y < vb.count; y++) { if (vb.pos[y] < 0) return false; npv = std::max(npv, ((vb.pos[y] & 1) + vb.size + 1) / 2); }
                if (npv > (longf == 2 ? 32 : longf ? (ncomp == 2 ? 24 : 16) : ncomp == 2 && !ring_of ? 12 : 8)) return false;   // ring depth of the instantiations: 8 row pairs, 12 for the planar chroma planes, 16 in the long form
                if (longf == 2) { nph = std::max(20, (nph + 3) & ~3); if (nph > 32) return false; }   // (20 / 24 / 28 / 32 pairs)
                else if (longf) { nph = std::max(10, (nph + 1) & ~1); if (nph > 16) return false; }   // (the long form's instantiations: 10 / 12 / 14 / 16 horizontal tap pairs, zero-padded rows)
                else if (nph > 8) return false;
                for (int y = 1; y < vb.count; y++) if (vb.pos[y] < vb.pos[y - 1]) return false;   // the ring only moves forward
                g.TW = TW; g.strips = strips; g.NCmax = ncmax; g.nph = nph; g.npv = npv; g.hfs2 = longf ? 2 * nph : hf2; g.vfs2 = vf2;
                g.lds_bytes = 4 * ncomp * 2 * ((ncmax + SPC) / 2) * 4;
                g.dma8_ok = 0; g.nph8 = 0; g.lds_dma8_bytes = 0; g.hT8 = nullptr;     // (the byte-row LDS-DMA form: plan3_alt)
                // LDS-DMA form: a ring of 4 row pairs per wave, rows of ncmax 16-bit samples; every pair between the first and the last
                // one a band needs is requested, so the windows of consecutive rows must touch (no skipped pair)
                g.lds_dma_bytes = 4 * 4 * ncomp * 2 * (ncmax / 2) * 4;
                g.dma_ok = !longf && (p.srcKind == SRCK_PLANAR16 || rgbread) && p.src_depth < 16 && g.lds_dma_bytes <= 40 * 1024;   // (LDS-DMA copies rows as they are: planar 16-bit sources, and the reader planes of a packed RGB source)
                for (int y = 1; y < vb.count && g.dma_ok; y++)
                    if (((vb.pos[y] & ~1) >> 1) > ((vb.pos[y - 1] & ~1) >> 1) + npv) g.dma_ok = 0;
                o.cs = put(cs.data(), cs.size() * 4); o.cc = put(cc.data(), cc.size() * 4);
                const int epr = longf == 2 ? 4 : longf ? 2 : 1;   // 64-byte entries per row: the long form's 16 tap pairs run on into a second one
                std::vector<SwsStripRow> rows((size_t)vb.count * epr);
                std::memset(rows.data(), 0, rows.size() * sizeof(SwsStripRow));
                for (int y = 0; y < vb.count; y++) {
                    SwsStripRow &e = rows[(size_t)y * epr];
                    e.pf = (vb.pos[y] & ~1) >> 1;
                    if (plan_rnd && (size_t)y < plan_rnd->size()) e.rnd_off = (*plan_rnd)[(size_t)y];
                    const int lead = ring_of ? 2 * (ring_of(npv) - npv) : 0;
                    if (lead < 0) return false;
                    for (int j = 0; j < vb.size; j++) {
                        const int k = (vb.pos[y] & 1) + j + lead;
                        // (pairs 8 .. 11 of a long chroma filter land in the four spare dwords behind vt[8]: load_strip_row_n)
                        const int16_t tap = (vb.size == 1 && plane1_form) ? (int16_t)4096 : vb.taps[(size_t)y * vb.size + j];
                        reinterpret_cast<uint32_t *>(rows.data())[(size_t)y * epr * 16 + 4 + (k >> 1)] |= (uint32_t)(uint16_t)tap << (16 * (k & 1));
                    }
                }
                o.rows = put(rows.data(), rows.size() * sizeof(SwsStripRow));
                return true;
            };
            // The same plan on strips of another width, for the short-filter instantiations (k_strip2.hip: 8-bit sources, at most 6 tap pairs each way,
            // windows of at most 64 chunks): 3 .. 5 luma / 1 .. 3 chroma columns per lane, whichever leaves the fewest idle lane-columns in the last
            // strip (640 columns: 2 strips of 320 instead of 2.5 of 256); only colStart / colCount differ from the base plan
            auto plan3_alt = [&](const FilterBank &hb, const FilterBank &vb, int W, int ncomp, const SwsStripGeom &base, SwsStripGeom &alt, SOff &o) -> bool {
                if (c->tune.no_strip_short || SPC != 16 || (p.srcKind != SRCK_PLANAR8 && p.srcKind != SRCK_NV12) || base.npv > 8 || base.nph > 7) return false;
                const int hf2 = fs2(hb.size);
                int best = 0; int64_t best_cost = INT64_MAX;
                std::vector<int32_t> bcs, bcc; int bnc = 0;
                const int forced = ncomp == 2 ? c->tune.strip_cols_c : c->tune.strip_cols_l;
                // (the LDS-DMA form -- planar sources, no skipped row pair -- needs fewer registers per column and takes wider strips)
                const bool nvc = ncomp == 2 && p.srcKind == SRCK_NV12;      // interleaved chroma bytes: two bytes per sample in the DMA'd rows
                bool dma8 = !c->tune.no_strip_dma8 && (hb.size + 1) / 2 <= 6;
                for (int y = 1; y < vb.count && dma8; y++)
                    if (((vb.pos[y] & ~1) >> 1) > ((vb.pos[y - 1] & ~1) >> 1) + base.npv) dma8 = false;
                const std::vector<int> cand = dma8 ? (nvc ? std::vector<int>{ 3, 2, 1 } : ncomp == 2 ? std::vector<int>{ 5, 4, 3, 2, 1 } : std::vector<int>{ 7, 6, 5, 4, 3 })
                                                   : (ncomp == 2 ? std::vector<int>{ 3, 2, 1 } : std::vector<int>{ 5, 4, 3 });
                for (int cols : cand) {
                    if (!c->tune.strip_cols_auto && cols != forced) continue;
                    const int TW = 64 * cols, strips = (W + TW - 1) / TW;
                    std::vector<int32_t> cs(strips), cc(strips);
                    int ncmax = 0; bool ok = true;
                    for (int t = 0; t < strips && ok; t++) {
                        int lo = INT32_MAX, hi = -1;
                        for (int x = t * TW; x < std::min(W, (t + 1) * TW); x++) { lo = std::min(lo, hb.pos[x] & ~1); hi = std::max(hi, (hb.pos[x] & ~1) + hf2); }
                        if (lo < 0) { ok = false; break; }
                        lo = lo / SPC * SPC;
                        cs[t] = lo; cc[t] = (hi - lo + SPC - 1) / SPC * SPC; ncmax = std::max(ncmax, cc[t]);
                    }
                    if (!ok || ncmax / SPC > 64 || (nvc && dma8 && 2 * ncmax / SPC > 64)) continue;
                    // what a launch pays per row: every strip its columns plus a fixed share (staging / requests, plan entry, waits, stores):
                    // about two columns' worth (measured on C1: chroma strips of 64 / 128 / 192 columns, 5 / 3 / 2 per row)
                    const int64_t cost = (int64_t)strips * (2 * cols + 3);
                    if (cost < best_cost) { best_cost = cost; best = cols; bcs = cs; bcc = cc; bnc = ncmax; }
                }
                if (!best) return false;
                alt = base;
                alt.TW = 64 * best; alt.strips = (W + alt.TW - 1) / alt.TW; alt.NCmax = bnc;
                alt.lds_bytes = 4 * ncomp * 2 * ((bnc + SPC) / 2) * 4; alt.dma_ok = 0;
                o.cs = put(bcs.data(), bcs.size() * 4); o.cc = put(bcc.data(), bcc.size() * 4); o.rows = 0;
                // LDS-DMA form (kernels_strip8.hpp): raw byte rows, a ring of 4 row pairs per wave; planar sources only (semi-planar chroma bytes are
                // interleaved); every pair between the first and the last one a band needs is requested, so no pair may be skipped.  Its tap rows
                // start at the filter's own first tap (no even-position padding): o.rows carries their offset in the blob
                alt.nph8 = (hb.size + 1) / 2;
                // (4 waves per block x ring depth x rows of a pair x dwords of a row; the chroma rings are 2 pairs deep: kernels_strip8.hpp)
                const int depth8 = ncomp == 2 ? 2 : 4;
                alt.lds_dma8_bytes = nvc ? 4 * depth8 * 2 * ((2 * bnc + 16) / 4) * 4 : 4 * depth8 * ncomp * 2 * ((bnc + 16) / 4) * 4;
                alt.dma8_ok = dma8 && alt.lds_dma8_bytes <= 48 * 1024;
                if (alt.dma8_ok) {
                    const int f8 = 2 * alt.nph8;
                    std::vector<int16_t> t8((size_t)hb.count * f8, 0);
                    for (int i = 0; i < hb.count; i++)
                        for (int j = 0; j < hb.size; j++) t8[(size_t)i * f8 + j] = hb.taps[(size_t)i * hb.size + j];
                    o.rows = put(t8.data(), t8.size() * 2);
                }
                return true;
            };
            d->stripLs_ok = d->stripCs_ok = false;
            SOff sLs, sCs;
            SOff sL, sC;
            // (raw sums: the packed X form's own taps -- except when both vertical filters have one tap: the packed writers then take their "_1" forms
            //  (yuv2rgb_full_1, yuv2rgb_1: vscale.c:136-141), which ignore the coefficients like yuv2plane1 does and equal the X arithmetic with the tap
            //  4096: Y = buf << 2 == (buf << 12 + (1 << 9)) >> 10, (buf + 64) >> 7 == (buf << 12 + (1 << 18)) >> 19; planar RGB has no such form, vscale.c:173-212)
            //  The packed YUV formats of 10 / 12 bits (DSTK_PACKEDHI) have X writers only, which multiply by the bank's value even when it is the only tap
            //  (4095 after initFilter's normalisation, 0 in the zero-vector rows of a source of fewer than four rows with shifted chroma): their own taps.
            const bool raw_one_one = p.dstKind == DSTK_RAW32 && c->vLum.size == 1 && vChrB.size == 1 && d->fullchr_kind != DSTK_GBRP && d->fullchr_kind != DSTK_PACKEDHI && d->fullchr_kind != DSTK_GBRP16 && d->fullchr_kind != DSTK_GBRPF32;
            const bool chr_plane1 = (p.dstKind != DSTK_NV12 && p.dstKind != DSTK_P010 && p.dstKind != DSTK_P016 && p.dstKind != DSTK_RAW32) || raw_one_one, lum_plane1 = p.dstKind != DSTK_RAW32 || raw_one_one;
            // (experiment "exp4", round 6, unmeasured: 192-column luma strips for planar sources of 9 .. 15 bits -- at 2:1 their window is 396 samples = 50 chunks, ONE
            //  LDS-DMA request per row where a 256-column strip's 524 samples need a second request of 32 bytes; k_strip.hip has the two instantiations it can reach)
            const int strip_cols_l = c->tune.strip_cols_l == 2 ? 2 : (c->tune.strip_cols_l == 3 && c->tune.exp[4] && p.srcKind == SRCK_PLANAR16 && p.src_depth < 16 && !p.wide) ? 3 : 4;
            const int strip_cols_c = c->tune.strip_cols_c == 1 ? 1 : 2;
            // (narrow pictures leave most of a 256-column strip idle and pay the per-band ring fill: the tile kernel keeps them)
            const int strip_min_w = c->tune.strip_min_w;
            // the 19-bit kernel's strips: 128 luma columns, 128 chroma columns where the window fits one 16-byte chunk per lane, else 64
            auto wide_cols = [&](const FilterBank &hb, const FilterBank &vb, int W, bool chroma) {
                int npvw = 1;
                for (int y = 0; y < vb.count; y++) npvw = std::max(npvw, ((vb.pos[y] & 1) + vb.size + 1) / 2);
                const int hf2 = fs2(hb.size);
                for (int cols : { 2, 1 }) {      // (measured: 256-column strips spill at 128 registers -- the rings hold two int32 rows per pair: bench w1 0.22 -> 0.08; 128-column chroma strips
                    //  with a ring of 4 pairs: w1 0.21 -> 0.26; with a ring of 8 pairs they spill too: yuv420p 4K -> yuv420p16le 1080p 0.0109 -> 0.0256 ms / frame)
                    if (chroma ? (cols == 2 && npvw > 4) : cols == 1) continue;
                    if (c->tune.strip_cols_auto == 0 && cols != (chroma ? (c->tune.strip_cols_c == 1 ? 1 : 2) : (c->tune.strip_cols_l == 2 ? 2 : 4))) continue;   // (experiments / tests: forced widths)
                    const int TW = 64 * cols; int ncmax = 0;
                    for (int t = 0; t * TW < W; t++) {
                        int lo = INT32_MAX, hi = -1;
                        for (int x = t * TW; x < std::min(W, (t + 1) * TW); x++) { lo = std::min(lo, hb.pos[x] & ~1); hi = std::max(hi, (hb.pos[x] & ~1) + hf2); }
                        lo = std::max(lo, 0) / SPC * SPC;
                        ncmax = std::max(ncmax, (hi - lo + SPC - 1) / SPC * SPC);
                    }
                    if (ncmax / SPC <= 64) return cols;
                }
                return chroma ? 1 : 2;
            };
            const int wcl = p.wide ? wide_cols(hLumB, c->vLum, p.dstW, false) : 0, wcc = (p.wide && !gray_both) ? wide_cols(hChrB, vChrB, p.chrDstW, true) : 0;
            const bool strip_plan = fullA && dst_ok && !(p.range_active && c->tune.no_strip_range) && !c->tune.no_strip && p.dstW >= (long_form ? strip_min_w_eff : strip_min_w) &&   // (the long form's strips are 128 columns, and what it replaces is the element-per-thread tile kernel)
                                    plan3(hLumB, c->vLum, p.dstW, p.wide ? wcl : long_form == 2 ? 1 : long_form ? 2 : strip_cols_l, 1, d->stripL, sL,
                                          p.wide ? +[](int n) { return n <= 2 ? 2 : n <= 4 ? 4 : 8; } : long_form == 2 ? +[](int) { return 32; } : nullptr, lum_plane1, long_form) &&
                                    (gray_both || plan3(hChrB, vChrB, p.chrDstW, p.wide ? wcc : long_form ? 1 : strip_cols_c, 2, d->stripC, sC,
                                                        p.wide ? +[](int n) { return n <= 2 ? 2 : n <= 4 ? 4 : 8; } : long_form == 2 ? +[](int) { return 32; } : long_form ? +[](int) { return 24; } : nullptr, chr_plane1, long_form)) &&
                                    (!p.wide || (d->stripL.NCmax / SPC <= 64 && (gray_both || d->stripC.NCmax / SPC <= 64)));      // (the wide kernel stages one chunk per lane and row)
            d->strip_ok = false;
            log_msg(c, 3, "strip plan: %d (windows %d/%d chunks of %d, taps %d/%d x %d/%d)\n", strip_plan, d->stripL.NCmax / SPC, d->stripC.NCmax / SPC, SPC,
                    d->stripL.nph, d->stripC.nph, d->stripL.npv, d->stripC.npv);
            Off oL, oC;
            if (mixedM) {
                SOff sM;
                if (plan3(hChrB, vChrB, p.chrDstW, strip_cols_c, 2, d->stripC, sM, nullptr, chr_plane1)) {
                    const std::vector<int16_t> htc = padded(hChrB);
                    const size_t ohc = put(htc.data(), htc.size() * 2);
                    const bool altC = plan3_alt(hChrB, vChrB, p.chrDstW, 2, d->stripC, d->stripCs, sCs);
                    { int r_ = table_alloc(c, d, &d->d_dot2, &d->dot2_bytes, blob.size()); if (r_ < 0) return r_; }
                    { int r_ = table_put(c, d, d->d_dot2, blob.data(), blob.size()); if (r_ < 0) return r_; }
                    const uint8_t *b = (const uint8_t *)d->d_dot2;
                    d->stripC.colStart = (const int32_t *)(b + sM.cs); d->stripC.colCount = (const int32_t *)(b + sM.cc);
                    d->stripC.rows = (const SwsStripRow *)(b + sM.rows);
                    d->stripC.hT2 = (const int16_t *)(b + ohc); d->stripC.vT2 = nullptr;
                    if (altC) { d->stripCs.colStart = (const int32_t *)(b + sCs.cs); d->stripCs.colCount = (const int32_t *)(b + sCs.cc);
                                d->stripCs.rows = d->stripC.rows; d->stripCs.hT2 = d->stripC.hT2; d->stripCs.vT2 = nullptr; d->stripCs.hT8 = (const int16_t *)(b + sCs.rows); d->stripCs_ok = true; }
                    d->mixed_ok = true;
                }
            } else
            if (rgb_ok) {
                // RGB epilogue: 256 luma columns + the 128 chroma columns under them per wave, one 16-byte chunk per lane and row for
                // both (windows of up to 1024 source samples), rings of 5 / 3 row pairs (8 / 8 in the long form)
                SOff rL, rC;
                SwsStripGeom &gl = d->stripRL, &gc = d->stripRC;
                auto ringL = [](int npv) { return npv <= 5 ? 5 : npv <= 8 ? 8 : -1; };
                auto ringC = [](int npv) { return npv <= 1 ? 1 : npv <= 3 ? 3 : npv <= 8 ? 8 : -1; };
                const int rcl = (c->tune.strip_rgb_cols == 2 || rgb_s16) ? 2 : 4;
                // (one vertical tap each: yuv2rgb_1_c_template's (buf + 64) >> 7 -- the coefficient is never looked at -- is the X arithmetic with the tap 4096)
                const bool rgb_one_one = c->vLum.size == 1 && vChrB.size == 1;
                // The writers' short forms (packed_vscale, vscale.c:135-157, picks per output row): one luma tap with two chroma taps that sum to 4096 is yuv2rgb_1_c_template
                // with a chroma blend -- (u0 (4096 - a) + u1 a + (128 << 11)) >> 19, output.c:1913-1937: the X arithmetic on the bank's own taps, the luma tap
                // taken as 4096; two taps each that sum to 4096 (bilinear up-scaling: a player's 720p -> 1080p into bgra) is yuv2rgb_2_c_template, the X
                // arithmetic without the rounding constant (SwsStripRow::rnd_off).  Round 5; not with an alpha plane (its own formulas there).
                FilterBank vLumS, vChrS;
                std::vector<int32_t> rnd_rows;
                bool short_rows = false;
                if (!rgb_one_one && !c->needAlpha && !c->tune.no_short_forms && p.chrDstH == p.dstH && (c->vLum.size == 1 || c->vLum.size == 2) && vChrB.size == 2) {
                    vLumS = c->vLum; vChrS = vChrB;
                    rnd_rows.assign((size_t)o.dst_h, 0);
                    for (int y = 0; y < o.dst_h; y++) {
                        int16_t *lf = &vLumS.taps[(size_t)y * vLumS.size], *cf = &vChrS.taps[(size_t)y * 2];
                        const bool csum = (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U;
                        if (vLumS.size == 1 && csum) { lf[0] = 4096; short_rows = true; }      // (the X arithmetic as it is: the blend rounds with 128 << 11 == 1 << 18)
                        else if (vLumS.size == 2 && csum && (uint16_t)lf[1] + (uint16_t)lf[0] == 4096 && (uint16_t)lf[1] <= 4096U) { rnd_rows[(size_t)y] = 1 << 18; short_rows = true; }
                    }
                }
                const FilterBank &vLumR = short_rows ? vLumS : c->vLum, &vChrR = short_rows ? vChrS : vChrB;
                striprgb_short = short_rows;
                plan_rnd = short_rows ? &rnd_rows : nullptr;
                const bool pl = plan3(hLumB, vLumR, p.dstW, rcl, 1, gl, rL, ringL, rgb_one_one || vLumR.size == 1), pc = pl && plan3(hChrB, vChrR, p.chrDstW, rcl / 2, 2, gc, rC, ringC, rgb_one_one);
                plan_rnd = nullptr;
                log_msg(c, 3, "strip_rgb plan: luma %d chroma %d strips %d/%d window %d/%d nph %d/%d npv %d/%d chrDstH %d dstH %d\n", pl, pc, gl.strips, gc.strips,
                        gl.NCmax, gc.NCmax, gl.nph, gc.nph, gl.npv, gc.npv, p.chrDstH, p.dstH);
                SOff sA;
                const bool wantA = p.need_alpha != 0;     // (rgb_ok: then rgb_alpha holds)
                // (not the one-tap form: yuv2rgb_1_c_template's alpha is (a * 255 + 16384) >> 15, output.c:1904, not the X arithmetic)
                const bool pa = !wantA || (!rgb_one_one && plan3(hLumB, c->vLum, p.dstW, strip_cols_l, 1, d->stripL, sA, nullptr, false));   // the plain luma launch, the X form's own taps
                if (pl && pc && pa && gl.strips == gc.strips &&
                    gl.NCmax / SPC <= 64 && gc.NCmax / SPC <= 64 && std::max(gl.nph, gc.nph) <= 8 && gl.npv <= 8 && gc.npv <= 8 && p.chrDstH == p.dstH) {
                    const std::vector<int16_t> htl = padded(hLumB), htc = padded(hChrB);
                    const size_t ohl = put(htl.data(), htl.size() * 2), ohc = put(htc.data(), htc.size() * 2);
                    // LDS-DMA form (kernels_striprgb.hpp sws_k_strip_rgb8): 8-bit planar sources, no source row pair skipped in either plane class;
                    // its tap rows start at the filter's own first tap (no even-position padding)
                    auto dma8_plan = [&](const FilterBank &hb, const FilterBank &vb, SwsStripGeom &g, size_t &off) {
                        g.nph8 = (hb.size + 1) / 2;
                        g.dma8_ok = !c->tune.no_strip_dma8 && p.srcKind == SRCK_PLANAR8 && !rgb_s16 && g.nph8 <= 6;
                        for (int y = 1; y < vb.count && g.dma8_ok; y++)
                            if (((vb.pos[y] & ~1) >> 1) > ((vb.pos[y - 1] & ~1) >> 1) + g.npv) g.dma8_ok = 0;
                        if (!g.dma8_ok) return;
                        const int f8 = 2 * g.nph8;
                        std::vector<int16_t> t8((size_t)hb.count * f8, 0);
                        for (int i = 0; i < hb.count; i++)
                            for (int j = 0; j < hb.size; j++) t8[(size_t)i * f8 + j] = hb.taps[(size_t)i * hb.size + j];
                        off = put(t8.data(), t8.size() * 2);
                    };
                    size_t o8l = 0, o8c = 0;
                    dma8_plan(hLumB, vLumR, gl, o8l); dma8_plan(hChrB, vChrR, gc, o8c);
                    if (!gl.dma8_ok || !gc.dma8_ok) gl.dma8_ok = gc.dma8_ok = 0;
                    { int r_ = table_alloc(c, d, &d->d_dot2, &d->dot2_bytes, blob.size()); if (r_ < 0) return r_; }
                    { int r_ = table_put(c, d, d->d_dot2, blob.data(), blob.size()); if (r_ < 0) return r_; }
                    const uint8_t *b = (const uint8_t *)d->d_dot2;
                    gl.colStart = (const int32_t *)(b + rL.cs); gl.colCount = (const int32_t *)(b + rL.cc); gl.rows = (const SwsStripRow *)(b + rL.rows);
                    gc.colStart = (const int32_t *)(b + rC.cs); gc.colCount = (const int32_t *)(b + rC.cc); gc.rows = (const SwsStripRow *)(b + rC.rows);
                    gl.hT2 = (const int16_t *)(b + ohl); gc.hT2 = (const int16_t *)(b + ohc); gl.vT2 = gc.vT2 = nullptr;
                    if (gl.dma8_ok) { gl.hT8 = (const int16_t *)(b + o8l); gc.hT8 = (const int16_t *)(b + o8c); }
                    gl.nph = gc.nph = std::max(gl.nph, gc.nph);      // one instantiation: the shorter tap rows are zero-extended in the kernel
                    if (wantA) {
                        d->stripL.colStart = (const int32_t *)(b + sA.cs); d->stripL.colCount = (const int32_t *)(b + sA.cc); d->stripL.rows = (const SwsStripRow *)(b + sA.rows);
                        d->stripL.hT2 = gl.hT2; d->stripL.vT2 = nullptr;
                        d->alpha_launch = 2;
                    }
                    d->striprgb_long = gl.npv > 5;
                    d->striprgb_ok = true;                           // (and every row in the "X" writer mode: checked below)
                    // the semi-planar source itself, without the split pass, where the kernel can read it: nv12-like through the LDS-DMA form's selectors
                    // (chroma windows of two bytes per sample: <= 64 chunks), p010-like through the 16-bit instantiation's staging (128-column strips)
                    d->striprgb_direct = 0;
                    if (!c->tune.no_striprgb_direct && !wantA) {
                        const size_t lds_nv = (size_t)16 * (2 * 2 * ((gl.NCmax + 16) >> 2) + 2 * 2 * ((2 * gc.NCmax + 16) >> 2) + 32 * 4);     // (k_striprgb.hip: ring depth 2)
                        if ((d->split_mode & 8) && gl.dma8_ok && gc.dma8_ok && rcl == 4 && 2 * gc.NCmax / 16 <= 64 && lds_nv <= 60 * 1024 && std::max(gl.nph8, gc.nph8) <= 6) {
                            d->striprgb_direct = 1; d->striprgb_direct_swap = (d->split_mode & 16) ? 1 : 0;
                        } else if ((d->split_mode & 32) && rgb_s16 && rcl == 2) {
                            d->striprgb_direct = 2; d->striprgb_direct_shift = d->split_shift;
                        }
                    }
                }
            } else
            {
              const bool tiles = !gray_both && !long_form && plan2(hLumB, c->vLum, p.dstW, p.dstH, 1, d->dotL, oL) && plan2(hChrB, vChrB, p.chrDstW, p.chrDstH, 2, d->dotC, oC);
              size_t ohl = 0, ohc = 0;
              const bool altL = strip_plan && !long_form && !p.wide && plan3_alt(hLumB, c->vLum, p.dstW, 1, d->stripL, d->stripLs, sLs);
              const bool altC = strip_plan && !long_form && !p.wide && !gray_both && plan3_alt(hChrB, vChrB, p.chrDstW, 2, d->stripC, d->stripCs, sCs);
              // scaled packed RGB -> packed RGB in one launch (k_striprgb2rgb.hip): the same filters planned once more on strips of 128 columns for both plane
              // classes (a lane owns the same destination columns of Y, U, V and A).  Luma and chroma share the vertical bank there (same source and destination
              // heights), which the kernel's lockstep march relies on: checked tap position by tap position
              SOff s2l, s2c;
              d->rgb2rgb_ok = false;
              bool r2r = strip_plan && rgbread && !(p.srcW & 3) && !gray_both && !long_form && !c->tune.no_strip_rgb2rgb && (p.srcKind == SRCK_RGB24 || p.srcKind == SRCK_RGB32) && !alpha_planar &&
                         (d->fullchr_on == 1 || d->fullchr_on == 2) && (d->fullchr_kind == DSTK_RGB24 || d->fullchr_kind == DSTK_RGB32) &&
                         (d->fullchr_on != 2 || (p.srcKind == SRCK_RGB32 && d->fullchr_kind == DSTK_RGB32)) && p.chrDstW == p.dstW && p.chrDstH == p.dstH && p.chrSrcVSub == 0 &&
                         c->vLum.size == vChrB.size && c->vLum.pos == vChrB.pos;
              // (strips a little narrower than 128 columns where that brings the widest pixel window down to one reader turn -- 256 pixels: 120 columns at 2:1)
              int r2r_tw = 128;
              if (r2r && c->tune.strip_cols_auto) {
                  const int hm = p.chr_half ? 2 : 1, hfl = fs2(hLumB.size), hfc = fs2(hChrB.size);
                  auto widest = [&](int tw) {
                      int npx = 0;
                      for (int x0 = 0; x0 < p.dstW; x0 += tw) {
                          int lo = INT32_MAX, hi = 0;
                          for (int x = x0; x < std::min(p.dstW, x0 + tw); x++) {
                              lo = std::min(lo, std::min(hLumB.pos[x] & ~1, hm * (hChrB.pos[x] & ~1)));
                              hi = std::max(hi, std::max((hLumB.pos[x] & ~1) + hfl, hm * ((hChrB.pos[x] & ~1) + hfc)));
                          }
                          npx = std::max(npx, ((hi + 7) & ~7) - (lo & ~15));
                      }
                      return npx;
                  };
                  if (widest(128) > 256) for (int tw : { 124, 120, 116, 112, 104, 96 }) if (widest(tw) <= 256) { r2r_tw = tw; break; }
              }
              // (the full-chroma writers' short forms, per output row: SwsStripRow::rnd_off = 1 << 9 where the reference leaves the rounding out -- kernels_stream.hpp
              //  fullchr_row_rnd has the rule and the citations; the kernel subtracts it from its 1 << 9)
              std::vector<int32_t> r2r_rnd;
              if (r2r && !c->tune.no_short_forms && (c->vLum.size == 1 || c->vLum.size == 2) && vChrB.size == 2 && p.chrDstH == p.dstH) {
                  r2r_rnd.assign((size_t)o.dst_h, 0);
                  for (int y = 0; y < o.dst_h; y++) {
                      const int16_t *lf = &c->vLum.taps[(size_t)y * c->vLum.size], *cf = &vChrB.taps[(size_t)y * 2];
                      const bool csum = (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U;
                      if (csum && (c->vLum.size == 1 || ((uint16_t)lf[1] + (uint16_t)lf[0] == 4096 && (uint16_t)lf[1] <= 4096U))) r2r_rnd[(size_t)y] = 1 << 9;
                  }
                  plan_rnd = &r2r_rnd;
              }
              rgb2rgb_short = plan_rnd != nullptr;
              r2r = r2r && plan3(hLumB, c->vLum, p.dstW, 2, 1, d->stripL2, s2l, nullptr, lum_plane1, 0, r2r_tw);
              plan_rnd = nullptr;
              r2r = r2r && plan3(hChrB, vChrB, p.chrDstW, 2, 2, d->stripC2, s2c, nullptr, chr_plane1, 0, r2r_tw) &&
                    d->stripL2.strips == d->stripC2.strips && d->stripL2.npv == d->stripC2.npv && std::max(d->stripL2.nph, d->stripC2.nph) <= 8 && d->stripL2.npv <= 8;
              // the lockstep strip kernel of packed sources into half-width-chroma YUV (k_striprgbsrc.hip) likewise plans for itself: luma strips of up to 256
              // columns over chroma strips of half as many, a few columns narrower where that brings the widest pixel window down by a reader turn of 256
              // pixels (248 columns at 2:1: 504 pixels, two turns instead of three).  (YUV destinations: never together with the RGB -> RGB plans above)
              const bool packed422_src = (d->split_mode & 3) && !(d->split_mode & 40) && p.srcKind == SRCK_PLANAR8 && p.chrSrcW == (p.srcW >> 1) && p.chrSrcVSub == 0 && !vlines_pending;
              SOff s3l, s3c;
              bool rsrc = strip_plan && !r2r && !p.wide && ((rgbread && !(p.srcW & 3) && (p.srcKind == SRCK_RGB24 || p.srcKind == SRCK_RGB32 || p.srcKind == SRCK_GBRP || p.srcKind == SRCK_RGB30) && p.chr_half) || packed422_src) && !gray_both && !long_form &&
                          !c->tune.no_strip_rgbsrc && !alpha_planar && !p.need_alpha && !d->fullchr_on &&
                          p.chrDstW == ((p.dstW + 1) >> 1) && (p.chrDstVSub == 0 ? p.chrDstH == p.dstH : (p.chrDstVSub == 1 && p.chrDstH == ((p.dstH + 1) >> 1)));
              if (rsrc) {
                  const int hfl = fs2(hLumB.size), hfc = fs2(hChrB.size);
                  auto widest = [&](int tw) {
                      int npx = 0;
                      for (int x0 = 0, s = 0; x0 < p.dstW; x0 += tw, s++) {
                          int lo = INT32_MAX, hi = 0;
                          for (int x = x0; x < std::min(p.dstW, x0 + tw); x++) { lo = std::min(lo, hLumB.pos[x] & ~1); hi = std::max(hi, (hLumB.pos[x] & ~1) + hfl); }
                          for (int x = s * (tw / 2); x < std::min(p.chrDstW, (s + 1) * (tw / 2)); x++) { lo = std::min(lo, 2 * (hChrB.pos[x] & ~1)); hi = std::max(hi, 2 * ((hChrB.pos[x] & ~1) + hfc)); }
                          npx = std::max(npx, ((hi + 15) & ~15) - (lo & ~15));
                      }
                      return npx;
                  };
                  int tw3 = 256;
                  if (c->tune.strip_cols_auto) {
                      const int turns = (widest(256) + 255) / 256;
                      if (turns > 1) for (int tw : { 248, 240, 232 }) if ((widest(tw) + 255) / 256 < turns) { tw3 = tw; break; }
                  }
                  rsrc = plan3(hLumB, c->vLum, p.dstW, 4, 1, d->stripL2, s3l, nullptr, lum_plane1, 0, tw3) && plan3(hChrB, vChrB, p.chrDstW, 2, 2, d->stripC2, s3c, nullptr, chr_plane1, 0, tw3 / 2) &&
                         d->stripL2.strips == d->stripC2.strips && std::max(d->stripL2.nph, d->stripC2.nph) <= 8 && d->stripL2.npv <= 8 && d->stripC2.npv <= 12;
              }
              if (!tiles && strip_plan) {   // (the strip kernel shares the tile kernel's padded horizontal taps; without a tile plan it gets its own copy)
                  const std::vector<int16_t> htl = padded(hLumB, long_form ? d->stripL.hfs2 : 0), htc = gray_both ? std::vector<int16_t>(2, 0) : padded(hChrB, long_form ? d->stripC.hfs2 : 0);
                  ohl = put(htl.data(), htl.size() * 2); ohc = put(htc.data(), htc.size() * 2);
              }
              if (tiles || strip_plan) {
                { int r_ = table_alloc(c, d, &d->d_dot2, &d->dot2_bytes, blob.size()); if (r_ < 0) return r_; }
                { int r_ = table_put(c, d, d->d_dot2, blob.data(), blob.size()); if (r_ < 0) return r_; }
                auto bind = [&](SwsTileGeom &g, const Off &o) {
                    const uint8_t *b = (const uint8_t *)d->d_dot2;
                    g.rowStart = (const int32_t *)(b + o.rs); g.rowCount = (const int32_t *)(b + o.rc);
                    g.colStart = (const int32_t *)(b + o.cs); g.colCount = (const int32_t *)(b + o.cc);
                    g.hT2 = (const int16_t *)(b + o.ht); g.vT2 = (const int16_t *)(b + o.vt);
                };
                if (tiles) { bind(d->dotL, oL); bind(d->dotC, oC); }
                d->dot2_ok = tiles && !p.wide && src_ok && fs2(vChrB.size) <= 16 && c->vLum.size >= 2 && vChrB.size >= 2 && !d->fullchr_on && !alpha_planar && !d->unity_h;
                if (strip_plan) {
                    const uint8_t *b = (const uint8_t *)d->d_dot2;
                    d->stripL.colStart = (const int32_t *)(b + sL.cs); d->stripL.colCount = (const int32_t *)(b + sL.cc);
                    if (!gray_both) { d->stripC.colStart = (const int32_t *)(b + sC.cs); d->stripC.colCount = (const int32_t *)(b + sC.cc); }
                    if (tiles) { d->stripL.hT2 = d->dotL.hT2; d->stripL.vT2 = d->dotL.vT2; d->stripC.hT2 = d->dotC.hT2; d->stripC.vT2 = d->dotC.vT2; }
                    else { d->stripL.hT2 = (const int16_t *)(b + ohl); d->stripC.hT2 = (const int16_t *)(b + ohc); d->stripL.vT2 = d->stripC.vT2 = nullptr; }
                    d->stripL.rows = (const SwsStripRow *)(b + sL.rows); if (!gray_both) d->stripC.rows = (const SwsStripRow *)(b + sC.rows);
                    if (altL) { d->stripLs.colStart = (const int32_t *)(b + sLs.cs); d->stripLs.colCount = (const int32_t *)(b + sLs.cc);
                                d->stripLs.rows = d->stripL.rows; d->stripLs.hT2 = d->stripL.hT2; d->stripLs.vT2 = d->stripL.vT2; d->stripLs.hT8 = (const int16_t *)(b + sLs.rows); d->stripLs_ok = true; }
                    if (altC) { d->stripCs.colStart = (const int32_t *)(b + sCs.cs); d->stripCs.colCount = (const int32_t *)(b + sCs.cc);
                                d->stripCs.rows = d->stripC.rows; d->stripCs.hT2 = d->stripC.hT2; d->stripCs.vT2 = d->stripC.vT2; d->stripCs.hT8 = (const int16_t *)(b + sCs.rows); d->stripCs_ok = true; }
                    if (r2r) {
                        const int32_t *csl = (const int32_t *)(blob.data() + s2l.cs), *ccl = (const int32_t *)(blob.data() + s2l.cc);
                        const int32_t *csc = (const int32_t *)(blob.data() + s2c.cs), *ccc = (const int32_t *)(blob.data() + s2c.cc);
                        const int hm = p.chr_half ? 2 : 1;
                        int npx = 0;
                        for (int s = 0; s < d->stripL2.strips; s++) {
                            const int w0 = std::min(csl[s], hm * csc[s]) & ~15, e = std::max(csl[s] + ccl[s], hm * (csc[s] + ccc[s]));
                            npx = std::max(npx, (e - w0 + 15) & ~15);
                        }
                        d->stripL2.colStart = (const int32_t *)(b + s2l.cs); d->stripL2.colCount = (const int32_t *)(b + s2l.cc); d->stripL2.rows = (const SwsStripRow *)(b + s2l.rows);
                        d->stripC2.colStart = (const int32_t *)(b + s2c.cs); d->stripC2.colCount = (const int32_t *)(b + s2c.cc); d->stripC2.rows = (const SwsStripRow *)(b + s2c.rows);
                        d->stripL2.hT2 = d->stripL.hT2; d->stripC2.hT2 = d->stripC.hT2; d->stripL2.vT2 = d->stripC2.vT2 = nullptr;
                        d->rgb2rgb_ok = npx <= 512; d->rgb2rgb_npx = npx;     // (two turns of 64 groups of four pixels)
                    }
                    d->strip_ok = true;
                    d->rgbread_on = rgbread;
                    d->alpha_launch = alpha_planar ? 1 : 0;
                    // scaled packed RGB into half-width-chroma YUV: one launch that reads the RGB rows itself (k_striprgbsrc.hip) on the same plan tables --
                    // luma strips of 256 columns over chroma strips of 128, every strip's pixel window (luma window and twice the chroma window, from a
                    // multiple of 16 pixels on) at most 64 lanes x 16 pixels
                    // (packed 8-bit 4:2:2 sources -- yuyv422 / uyvy422 / yvyu422 -- have the shape of the half readers: chroma samples under pixel pairs on every source
                    //  row; the kernel's byte-selector reader takes them from the caller's frame, without the split pass: striprgb_direct = 3)
                    if (rsrc) {
                        const int32_t *csl = (const int32_t *)(blob.data() + s3l.cs), *ccl = (const int32_t *)(blob.data() + s3l.cc);
                        const int32_t *csc = (const int32_t *)(blob.data() + s3c.cs), *ccc = (const int32_t *)(blob.data() + s3c.cc);
                        int npx = 0;
                        for (int s = 0; s < d->stripL2.strips; s++) {
                            const int w0 = std::min(csl[s], 2 * csc[s]) & ~15, e = std::max(csl[s] + ccl[s], 2 * (csc[s] + ccc[s]));
                            npx = std::max(npx, (e - w0 + 15) & ~15);
                        }
                        d->stripL2.colStart = (const int32_t *)(b + s3l.cs); d->stripL2.colCount = (const int32_t *)(b + s3l.cc); d->stripL2.rows = (const SwsStripRow *)(b + s3l.rows);
                        d->stripC2.colStart = (const int32_t *)(b + s3c.cs); d->stripC2.colCount = (const int32_t *)(b + s3c.cc); d->stripC2.rows = (const SwsStripRow *)(b + s3c.rows);
                        d->stripL2.hT2 = d->stripL.hT2; d->stripC2.hT2 = d->stripC.hT2; d->stripL2.vT2 = d->stripC2.vT2 = nullptr;
                        d->striprgbsrc_ok = npx <= 1024; d->striprgbsrc_npx = npx;
                        if (packed422_src) d->striprgb_direct = d->striprgbsrc_ok ? 3 : 0;
                    }
                }
              }
            }
        }
    }
    return 0;
}

} // namespace swship
