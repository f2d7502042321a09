// HIP side of libswscale_hip, part 2 of 4 -- the PLANNER: dev_prepare_on() turns an initialised context (filter banks, colour tables, the reference's plan
// choice: context.cpp) into kernel parameters, device tables and the launch state of ONE GPU, and names the path.  No kernel is launched here; with the
// option dry_plan nothing of HIP is called at all (tests/test_planner_table.py pins the planner's answers on the CPU box).
// The reference rules restated: libswscale/utils.c:1137-1835 (sws_init_context: formats, filters, the special converters), vscale.c:109-171 (writer forms).
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
// ---- the reference's line schedule (ff_swscale, swscale.c:388-535) for a frame that arrives in one slice ----
// The reference pulls destination rows: for every dstY it makes sure the horizontal ring holds the source lines the row needs -- running
// the line converters and the horizontal scaler over BATCHES of lines, ahead of need as far as the ring allows -- and then runs the
// vertical scaler for that one row.  Two stages make the result depend on that order:
//   mode 1  gamma_convert (gamma.c:31-58), first luma descriptor of the gamma cascade's scaling step (slice.c:325-328), rewrites the lines
//           of a batch in place; a line converted ahead of need and pulled again after a hole (the ring is re-based when the vertical
//           position jumps past lastInLumBuf + 1, :404-417) is converted again;
//   mode 2  chr_convert (hscale.c:211-225) computes plane 0's line index once per batch ("sp0") and steps it by one LUMA line per chroma
//           line: with SWS_SRC_V_CHR_DROP on a planar RGB source the G row of a chroma line depends on where its batch started.
// The walk below replays the cursor arithmetic and writes, for every destination row and tap, which picture line the tap reads and with
// which side term (mode 1: table passes the line had seen when it was h-scaled; mode 2: the plane-0 row).  The two-pass path then h-scales
// those "virtual lines" (one per row and tap) instead of the picture's lines.  `uniform`: every tap of mode 1 saw exactly one pass.
void build_vlines(const SwsInternal *c, int mode, VLines &out)
{
    const int srcH = c->opts.src_h, dstH = c->opts.dst_h, vsub = c->chrSrcVSubSample, chrSrcH = c->chrSrcH;
    const FilterBank &vl = c->vLum, &vc = c->vChr;
    const int chrSliceEnd = -((-srcH) >> vsub);
    // get_min_buffer_size (slice.c:217-243) and the floor of :266-267 (MAX_LINES_AHEAD = 4)
    int lumAvail = vl.size, chrAvail = vc.size;
    for (int y = 0; y < dstH; y++) {
        const int cy = (int)((int64_t)y * c->chrDstH / dstH);
        int next = std::max(vl.pos[y] + vl.size - 1, (vc.pos[cy] + vc.size - 1) << vsub);
        next >>= vsub; next <<= vsub;
        lumAvail = std::max(lumAvail, next - vl.pos[y]);
        chrAvail = std::max(chrAvail, (next >> vsub) - vc.pos[cy]);
    }
    lumAvail = std::max(lumAvail, vl.size + 4); chrAvail = std::max(chrAvail, vc.size + 4);
    std::vector<int32_t> passes((size_t)srcH, 0);       // gamma passes a picture line has seen so far
    std::vector<int32_t> inL((size_t)srcH, 0);          // ... when the ring last took it (luma)
    std::vector<int32_t> inC((size_t)chrSrcH, 0);       // mode 1: passes of the line a chroma line was made from; mode 2: its plane-0 row
    int lastInLum = -1, lastInChr = -1, lumHoles = 1, chrHoles = 1, lumY0 = 0, lumN = 0, chrY0 = 0, chrN = 0;
    out.lum.clear(); out.chr.clear(); out.lumPos.assign((size_t)dstH, 0); out.chrPos.assign((size_t)c->chrDstH, 0); out.uniform = true;
    std::vector<char> chrDone((size_t)c->chrDstH, 0);
    for (int y = 0; y < dstH; y++) {
        const int cy = y >> c->chrDstVSubSample;
        const int firstL = std::max(1 - vl.size, vl.pos[y]), firstC = std::max(1 - vc.size, vc.pos[cy]);
        const int lastL = std::min(srcH, firstL + vl.size) - 1, lastC = std::min(chrSrcH, firstC + vc.size) - 1;
        if (firstL > lastInLum) { lumHoles = lastInLum != firstL - 1; if (lumHoles) { lumY0 = firstL; lumN = 0; } lastInLum = firstL - 1; }
        if (firstC > lastInChr) { chrHoles = lastInChr != firstC - 1; if (chrHoles) { chrY0 = firstC; chrN = 0; } lastInChr = firstC - 1; }
        const int posY = lumY0 + lumN, cPosY = chrY0 + chrN;
        int fp, lp, fcp, lcp;
        if (posY <= lastL && !lumHoles) { fp = std::max(firstL, posY); lp = std::min(firstL + lumAvail - 1, srcH - 1); } else { fp = posY; lp = lastL; }
        if (cPosY <= lastC && !chrHoles) { fcp = std::max(firstC, cPosY); lcp = std::min(firstC + chrAvail - 1, chrSliceEnd - 1); } else { fcp = cPosY; lcp = lastC; }
        if (posY < lastL + 1) {
            for (int k = std::max(fp, 0); k <= lp && k < srcH; k++) { passes[(size_t)k]++; inL[(size_t)k] = passes[(size_t)k]; }
            lumN += lp - fp + 1;
        }
        lastInLum = lastL;
        if (cPosY < lastC + 1) {
            for (int k = std::max(fcp, 0); k <= lcp && k < chrSrcH; k++)
                inC[(size_t)k] = mode == 1 ? passes[(size_t)std::min(k << vsub, srcH - 1)] : (fcp << vsub) + (k - fcp);
            chrN += lcp - fcp + 1;
        }
        lastInChr = lastC;
        // the vertical stage of row y: its taps read the ring as it is now
        out.lumPos[(size_t)y] = (int32_t)(out.lum.size() / 2);
        for (int j = 0; j < vl.size; j++) {
            const int k = std::min(std::max(firstL + j, 0), srcH - 1);
            out.lum.push_back(k); out.lum.push_back(mode == 1 ? inL[(size_t)k] : -1);
            if (mode == 1 && firstL + j <= lastL && inL[(size_t)k] != 1) out.uniform = false;
        }
        if (!chrDone[(size_t)cy]) {   // (a vertically sub-sampled chroma row is written with the first luma row over it, vscale.c:74-107)
            chrDone[(size_t)cy] = 1;
            out.chrPos[(size_t)cy] = (int32_t)(out.chr.size() / 2);
            for (int j = 0; j < vc.size; j++) {
                const int k = std::min(std::max(firstC + j, 0), chrSrcH - 1);
                out.chr.push_back(k); out.chr.push_back(inC[(size_t)k]);
                if (mode == 1 && firstC + j <= lastC && inC[(size_t)k] != 1) out.uniform = false;
            }
        }
    }
}

bool bank_is_identity(const FilterBank &b, int one)
{
    if (b.size != 1) return false;
    for (int i = 0; i < b.count; i++)
        if (b.pos[i] != i || b.taps[i] != one) return false;
    return true;
}

// every tap of a one-tap bank equals `one` (initFilter's error-diffused normalisation leaves one - 1 in some rows)
bool bank_taps_all(const FilterBank &b, int one)
{
    if (b.size != 1) return false;
    for (int i = 0; i < b.count; i++)
        if (b.taps[i] != one) return false;
    return true;
}

int src_kind_of(int f)
{
    const PixDesc *d = pix_desc(f);
    if (!d) return -1;
    if (d->flags & PIXFLAG_BAYER) return SRCK_BAYER;
    if (f == AV_PIX_FMT_PAL8 || f == AV_PIX_FMT_RGB8 || f == AV_PIX_FMT_BGR8 || f == AV_PIX_FMT_RGB4_BYTE || f == AV_PIX_FMT_BGR4_BYTE) return SRCK_PAL;
    if (f == AV_PIX_FMT_UYYVYY411) return SRCK_PACKED411;
    if ((d->flags & PIXFLAG_FLOAT) && f != AV_PIX_FMT_GRAYF32LE && f != AV_PIX_FMT_GBRPF32LE && f != AV_PIX_FMT_GBRAPF32LE) return SRCK_FLOATX;
    if (isPlanarRGB(f)) return (d->flags & PIXFLAG_FLOAT) ? SRCK_GBRPF32 : d->comp[0].depth > 8 ? SRCK_GBRP16 : SRCK_GBRP;
    if (isAnyRGB(f) && d->comp[0].depth == 16) return SRCK_RGB48;
    if (f == AV_PIX_FMT_YA8 || f == AV_PIX_FMT_YA16LE) return SRCK_YA;
    if (f == AV_PIX_FMT_GRAYF32LE) return SRCK_GRAYF32;
    if (f == AV_PIX_FMT_MONOWHITE || f == AV_PIX_FMT_MONOBLACK) return SRCK_MONO;
    if (isAnyRGB(f) && d->comp[0].depth == 10 && d->comp[0].step == 4) return SRCK_RGB30;
    if (isAnyRGB(f) && d->comp[0].step == 2) return SRCK_RGB16;
    if (isAnyRGB(f)) return d->comp[0].step == 3 ? SRCK_RGB24 : SRCK_RGB32;
    if (isYUV(f) && isPackedFmt(f) && d->comp[0].depth > 8) return SRCK_PACKEDHI;
    if (isYUV(f) && isPackedFmt(f)) return d->log2_chroma_w ? SRCK_PACKED422 : SRCK_PACKED444;
    if (isSemiPlanarYUV(f)) return d->comp[0].depth == 8 ? SRCK_NV12 : SRCK_P010;
    if (isPlanarYUV(f) || isGray(f)) return d->comp[0].depth == 8 ? SRCK_PLANAR8 : SRCK_PLANAR16;
    return -1;
}
int dst_kind_of(int f)
{
    const PixDesc *d = pix_desc(f);
    if (!d) return -1;
    if (isPlanarRGB(f)) return (d->flags & PIXFLAG_FLOAT) ? DSTK_GBRPF32 : d->comp[0].depth == 16 ? DSTK_GBRP16 : DSTK_GBRP;
    if (isAnyRGB(f) && d->comp[0].depth == 16) return DSTK_RGB48;
    if (f == AV_PIX_FMT_YA8 || f == AV_PIX_FMT_YA16LE) return DSTK_YA;
    if (f == AV_PIX_FMT_GRAYF32LE) return DSTK_PLANARF32;
    if (f == AV_PIX_FMT_MONOWHITE || f == AV_PIX_FMT_MONOBLACK) return DSTK_MONO;
    if (isAnyRGB(f) && d->comp[0].depth == 10 && d->comp[0].step == 4) return DSTK_RGB30;
    if (isAnyRGB(f) && d->comp[0].step == 2) return DSTK_RGB16;
    if (f == AV_PIX_FMT_RGB4 || f == AV_PIX_FMT_BGR4) return DSTK_RGB4;
    if (isAnyRGB(f) && d->comp[0].step == 1) return DSTK_RGB8;
    if (isAnyRGB(f)) return d->comp[0].step == 3 ? DSTK_RGB24 : DSTK_RGB32;
    if (isYUV(f) && isPackedFmt(f) && d->comp[0].depth > 8) return DSTK_PACKEDHI;
    if (isYUV(f) && isPackedFmt(f)) return d->log2_chroma_w ? DSTK_PACKED422 : DSTK_PACKED444;
    const int depth = d->comp[0].depth;
    if (isSemiPlanarYUV(f)) return depth == 8 ? DSTK_NV12 : depth == 16 ? DSTK_P016 : DSTK_P010;
    if (isPlanarYUV(f) || isGray(f)) return depth == 8 ? DSTK_PLANAR8 : depth == 16 ? DSTK_PLANAR16 : DSTK_PLANARN;
    return -1;
}

// upload filter banks (one blob) and fill SwsDevParams
// Everything a planning run decides starts from what a FRESH state holds (zeros): a context that is planned again -- sws_hip_set_option() or
// sws_setColorspaceDetails() after its first conversion -- must not keep a flag or a geometry of the plan before, which the new plan may not touch (a stage it skips)
// while the tables that geometry points into are rewritten.  (Found in round 6 by tools/replan_hunt.py on the CPU box: 350 of 9 000 re-planned contexts kept e.g.
// `strip_ok` under no_strip.  Buffers, streams, the table blocks and their records are not plan state and stay.)
static void reset_plan_state(DeviceState *d)
{
    d->plan_serial++;
    d->unity_h = d->unity_v = d->all_x_mode = false; d->chr_window2 = 0;
    d->tile_ok = d->rgb_march_ok = d->dot2_ok = d->rgb444_ok = d->rgbsrc_ok = d->mixed_ok = d->strip_ok = d->stripLs_ok = d->stripCs_ok = d->striprgb_ok = d->striprgb_long = false;
    d->rgb_groups = 0; d->rgb_ncr = 6; d->rgbsrc_rows = nullptr; d->rgbsrc2_rows = nullptr; d->rgbsrc2_npv = 0;
    d->striprgb_direct = d->striprgb_direct_swap = d->striprgb_direct_shift = 0; d->striprgb_direct_now = false;
    d->rgb2rgb_ok = d->rgb2rgb_now = false; d->rgb2rgb_npx = 0; d->striprgbsrc_ok = false; d->striprgbsrc_npx = 0;
    d->rgbread_on = false; d->fullchr_on = d->fullchr_kind = d->fullchr_direct = d->alpha_launch = d->join422 = d->split_mode = d->split_shift = 0; d->vlines_on = false;
    for (SwsTileGeom *g : { &d->tileL, &d->tileC, &d->dotL, &d->dotC }) std::memset(g, 0, sizeof(*g));
    for (SwsStripGeom *g : { &d->stripL, &d->stripC, &d->stripLs, &d->stripCs, &d->stripRL, &d->stripRC, &d->stripL2, &d->stripC2 }) std::memset(g, 0, sizeof(*g));
}

// ---- stage 1: the parameters every path reads, and the helper passes around the scaler ----
int plan_common(PlanBuild &B)
{
    PLAN_HANDLES(B);
    reset_plan_state(d);
    std::memset(&p, 0, sizeof(p));
    // SWS_FAST_BILINEAR on 8-bit lines (ff_hyscale_fast_c / ff_hcscale_fast_c, hscale_fast_bilinear.c:23-55: dst = a (128 - xalpha) + b xalpha for luma and alpha,
    // a (127 - xalpha) + b xalpha for chroma, with xalpha = the top 7 bits of the 16-bit position fraction; the columns at and behind the last source sample: 128 x that
    // sample) is hScale8To15_c over a two-tap bank with the taps {(128 - xalpha) << 7, xalpha << 7} (chroma: {(xalpha ^ 127) << 7, xalpha << 7}) -- the sum's low 7 bits
    // are zero, so the >> 7 is exact and the 15-bit clip never acts.  Round 5: every plan below is made on these banks (hLumB / hChrB), which puts the strip kernels
    // behind the flag that players and capture tools pass most often (4K -> 1080p yuv420p: 0.090 ms per frame on the two-pass kernels, 0.012 with SWS_BILINEAR);
    // same widths give one-tap banks (chroma: 127 << 7, the reference's own quirk).  The context's banks stay what build_filter_bank() made (sws_hip_get_filter, the blob).
    FilterBank &fastL = B.fastL, &fastC = B.fastC;
    const bool fast_banks = c->plan == PLAN_MAIN && (o.flags & SWS_FAST_BILINEAR) && c->srcBpc == 8 && c->dstBpc <= 14 && !c->tune.no_fast_banks && o.src_w >= 2 && c->chrSrcW >= 2;
    if (fast_banks) {
        auto make = [](FilterBank &fb, int dstW, int srcW, int xInc, bool chroma) {
            std::vector<int32_t> pos((size_t)dstW); std::vector<int16_t> t0((size_t)dstW), t1((size_t)dstW);
            bool two = false;
            for (int x = 0; x < dstW; x++) {
                const uint32_t xpos = (uint32_t)x * (uint32_t)xInc;
                const int xx = (int)(xpos >> 16), xalpha = (int)((xpos & 0xFFFF) >> 9);
                if (xx >= srcW - 1) { pos[(size_t)x] = srcW - 1; t0[(size_t)x] = 1 << 14; t1[(size_t)x] = 0; }
                else { pos[(size_t)x] = xx; t0[(size_t)x] = (int16_t)((chroma ? (xalpha ^ 127) : 128 - xalpha) << 7); t1[(size_t)x] = (int16_t)(xalpha << 7); two = two || xalpha; }
            }
            fb.size = two ? 2 : 1; fb.count = dstW;
            fb.pos.assign((size_t)dstW + 3, 0); fb.taps.assign(((size_t)dstW + 3) * (size_t)fb.size, 0);
            for (int x = 0; x < dstW + 3; x++) {        // (+ 3 replicated rows like build_filter_bank's)
                const size_t q = (size_t)std::min(x, dstW - 1);
                int ps = pos[q]; int16_t a = t0[q], b = t1[q];
                if (two && ps >= srcW - 1) { ps = srcW - 2; b = a; a = 0; }       // (the window of two taps ends inside the row: the last sample is its second tap)
                fb.pos[(size_t)x] = ps; fb.taps[(size_t)x * (size_t)fb.size] = a;
                if (two) fb.taps[(size_t)x * 2 + 1] = b;
            }
        };
        make(fastL, o.dst_w, o.src_w, c->lumXInc, false);
        make(fastC, c->chrDstW, c->chrSrcW, c->chrXInc, true);
    }
    const FilterBank &hLumB = fast_banks ? fastL : c->hLum, &hChrB = fast_banks ? fastC : c->hChr;
    FilterBank &vChrJ = B.vChrJ; bool join_short = false;      // (the chroma bank of a packed 4:2:2 destination whose rows take yuv2422_1_c_template's blend: below)
    bool striprgb_short = false, rgb2rgb_short = false;     // (the strip-RGB / one-launch RGB -> RGB plans carry the packed writers' short forms in their rounding offsets: below)
    // (the flag with the fast functions still in the kernels -- the element-per-thread readers; sources whose lines are not 8-bit -- RGB, 9 .. 16-bit YUV -- get
    //  bilinear banks from the flag and nothing else, swscale.c:676-681)
    const bool fast_flag = (o.flags & SWS_FAST_BILINEAR) && c->srcBpc == 8 && c->dstBpc <= 14 && !fast_banks;
    p.srcW = o.src_w; p.srcH = o.src_h; p.dstW = o.dst_w; p.dstH = o.dst_h;
    p.chrSrcW = c->chrSrcW; p.chrSrcH = c->chrSrcH; p.chrDstW = c->chrDstW; p.chrDstH = c->chrDstH;
    p.chrSrcHSub = c->chrSrcHSubSample; p.chrSrcVSub = c->chrSrcVSubSample;
    p.chrDstHSub = c->chrDstHSubSample; p.chrDstVSub = c->chrDstVSubSample;
    p.srcKind = src_kind_of(o.src_format); p.dstKind = dst_kind_of(o.dst_format);
    p.srcBpc = c->srcBpc; p.dstBpc = c->dstBpc;
    p.src_depth = ds->comp[0].depth;
    p.src_shift = ds->comp[0].shift;
    p.wide = c->dstBpc > 14;
    p.hclip = p.wide ? (1 << 19) - 1 : (1 << 15) - 1;
    if (c->srcBpc == 8) p.hshift = p.wide ? 3 : 7;                       // hScale8To15_c / hScale8To19_c
    else if (p.wide) {                                                    // hScale16To19_c, swscale.c:69-97
        p.hshift = ds->comp[0].depth - 1 - 4;
        if ((isAnyRGB(o.src_format) || o.src_format == AV_PIX_FMT_PAL8) && ds->comp[0].depth < 16) p.hshift = 9;
        else if (ds->flags & PIXFLAG_FLOAT) p.hshift = 16 - 1 - 4;
    } else {                                                              // hScale16To15_c, swscale.c:99-125
        p.hshift = ds->comp[0].depth - 1;
        if (p.hshift < 15) p.hshift = (isAnyRGB(o.src_format) || o.src_format == AV_PIX_FMT_PAL8) ? 13 : ds->comp[0].depth - 1;
        else if (ds->flags & PIXFLAG_FLOAT) p.hshift = 16 - 1;
    }
    p.dst_bits = dd->comp[0].depth; p.dst_shift = dd->comp[0].shift;
    p.uv_swap_src = isSwappedChroma(o.src_format); p.uv_swap_dst = isSwappedChroma(o.dst_format);
    p.u_plane_src = ds->comp[1].plane; p.v_plane_src = ds->comp[2].plane;
    p.u_plane_dst = dd->comp[1].plane; p.v_plane_dst = dd->comp[2].plane;
    const bool gray_any = isGray(o.src_format) || isGray(o.dst_format) || c->needAlpha || p.srcKind == SRCK_MONO;   // paths the fused kernels do not cover
    p.should_dither = isNBPS(o.src_format) || is16BPS(o.src_format);     // swscale.c:292-293
    // ---- packed 4:2:2 destinations (yuyv422 / uyvy422 / yvyu422) through the planar writers and an interleaving pass ----
    // With an 8-bit source (no dither pattern: the constant 64) the packed writer's general form is the planar writers' arithmetic sample for
    // sample: yuv2422_X_c_template (output.c:843-881) sums from 1 << 18, >> 19, clip; yuv2planeX_8_c / yuv2plane1_8_c (output.c:438-493) sum from
    // 64 << 12, >> 19, and one tap of 4096 is (s + 64) >> 7 in both; yuv2422_1 with one chroma tap is the same.  The other short forms are not
    // (yuv2422_1 with two chroma taps takes the nearer row or the plain mean, yuv2422_2 sums without a rounding term; vscale.c:136-158): those
    // filter shapes keep the packed writer of the generic kernels.  The planner below then sees a planar 8-bit 4:2:2 destination (in a working
    // picture per frame), so the conversion gets the strip / mixed / tile kernel of its shape; launch_plan_le interleaves afterwards
    // (yuvPlanartoyuy2_c / yuvPlanartouyvy_c are plain byte interleaves: the streaming join of kernels_layout.hpp).
    // ---- packed 4:2:2 SOURCES (yuyv422 / uyvy422 / yvyu422: cameras, capture cards) through the planar kernels: yuy2ToY_c / yuy2ToUV_c / uyvyToY_c /
    //      uyvyToUV_c / yvy2ToUV_c (input.c:550-578, :890-907) copy bytes, so a streaming de-interleave into a planar 4:2:2 working picture per frame
    //      (yuyvtoyuv422_c / uyvytoyuv422_c of the layout kernel) followed by the kernels of a planar 8-bit source is the same arithmetic ----
    d->split_mode = 0;
    if (c->plan == PLAN_MAIN && p.srcKind == SRCK_PACKED422 && ds->comp[0].depth == 8 && !(o.src_w & 1) && !gray_any && !(o.flags & SWS_SRC_V_CHR_DROP_MASK) &&
        !c->tune.no_mixed && !c->tune.no_layout_stream) {
        p.srcKind = SRCK_PLANAR8;
        p.u_plane_src = 1; p.v_plane_src = 2;
        d->split_mode = (ds->comp[0].offset == 1 ? 2 : 1) | (ds->comp[2].offset < ds->comp[1].offset ? 4 : 0);   // 1 yuyv-like, 2 uyvy; 4: V before U (yvyu422)
    }
    // ---- semi-planar 8-bit sources (nv12 / nv21 / nv16 / nv24 / nv42: what the hardware decoders deliver) scaled into the packed-RGB LUT writers:
    //      nvXXtoUV_c (input.c:926-948) de-interleaves bytes, so the chroma plane is split into planar working planes first and the conversion takes the
    //      strip kernel with the RGB epilogue like a planar source (planar / semi-planar destinations: the strip kernel de-interleaves while staging) ----
    if (c->plan == PLAN_MAIN && p.srcKind == SRCK_NV12 && c->srcBpc == 8 && (p.dstKind == DSTK_RGB24 || p.dstKind == DSTK_RGB32) && !((o.flags & SWS_FULL_CHR_H_INT)) &&
        !(bank_is_identity(hLumB, 1 << 14) && bank_is_identity(hChrB, 1 << 14)) && !gray_any && !(o.flags & SWS_SRC_V_CHR_DROP_MASK) && !fast_flag &&
        !c->tune.no_mixed && !c->tune.no_layout_stream && !c->tune.no_strip) {
        d->split_mode = 8 | (p.uv_swap_src ? 16 : 0);
        p.srcKind = SRCK_PLANAR8;
        p.u_plane_src = 1; p.v_plane_src = 2; p.uv_swap_src = 0;
    }
    // (the 10 / 12-bit twins -- p010, p012, p210, p410 ...: words with the samples in the high bits -- become a planar working picture with the samples
    //  shifted down, luma included, and take the 16-bit instantiation of that kernel)
    //  (same-size pictures too -- a hardware decoder's p010 into RGB for display: the 16-bit instantiation takes identity horizontal filters as one-tap banks)
    if (c->plan == PLAN_MAIN && p.srcKind == SRCK_P010 && p.src_depth <= 15 && (p.dstKind == DSTK_RGB24 || p.dstKind == DSTK_RGB32) && !((o.flags & SWS_FULL_CHR_H_INT)) &&
        !gray_any && !(o.flags & SWS_SRC_V_CHR_DROP_MASK) && !fast_flag &&
        !c->tune.no_mixed && !c->tune.no_layout_stream && !c->tune.no_strip) {
        d->split_mode = 32; d->split_shift = p.src_shift;
        p.srcKind = SRCK_PLANAR16; p.src_shift = 0;
        p.u_plane_src = 1; p.v_plane_src = 2; p.uv_swap_src = 0;
    }
    d->join422 = 0;
    if (c->plan == PLAN_MAIN && p.dstKind == DSTK_PACKED422 && dd->comp[0].depth == 8 && !(o.dst_w & 1) && !c->needAlpha && !gray_any &&
        !(c->vLum.size == 2 && c->vChr.size == 2) && !(c->vLum.size == 1 && c->vChr.size == 2 && c->tune.no_short_forms) && !c->tune.no_mixed && !c->tune.no_layout_stream &&
        // (one tap on ONE side only: the packed X form multiplies by the bank's value, which initFilter's normalisation leaves at 4095 in some
        //  rows, where the planar one-tap form ignores it; one tap on both sides is yuv2422_1, which ignores both)
        ((c->vLum.size == 1) == (c->vChr.size == 1) || bank_taps_all(c->vLum.size == 1 ? c->vLum : c->vChr, 1 << 12))) {
        const bool uyvy = dd->comp[0].offset == 1, vfirst = dd->comp[2].offset < dd->comp[1].offset;   // (yvyu422: V before U)
        p.dstKind = DSTK_PLANAR8;
        p.u_plane_dst = vfirst ? 2 : 1; p.v_plane_dst = vfirst ? 1 : 2;
        d->join422 = uyvy ? 2 : 1;
        p.should_dither = 0;     // (9 .. 16-bit sources: the ordered dither belongs to the planar 8-bit writers, swscale.c:292-300; the packed ones round with 1 << 18 == the undithered 64 << 12)
        // One luma tap with two chroma taps (4:2:0 -> packed 4:2:2 at the same height with SWS_BILINEAR): the rows whose chroma taps sum to 4096 go to yuv2422_1_c_template
        // with a chroma blend -- the first chroma row alone while the second tap is below 2048, the mean of the two rows from there on, (u0 + u1 + 128) >> 8
        // (output.c:959-990; vscale.c:139-145) -- which is the X arithmetic over the taps {4096, 0} / {2048, 2048}: the chroma bank every plan and kernel below sees (vChrB).  Round 5.
        if (c->vLum.size == 1 && c->vChr.size == 2) {
            vChrJ = c->vChr;
            for (size_t y = 0; y + 1 < vChrJ.taps.size(); y += 2) {
                int16_t *cf = &vChrJ.taps[y];
                if ((uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U) { if (cf[1] < 2048) { cf[0] = 4096; cf[1] = 0; } else cf[0] = cf[1] = 2048; join_short = true; }
            }
        }
    }
    const FilterBank &vChrB = join_short ? vChrJ : c->vChr;
    p.full_chr = ((o.flags & SWS_FULL_CHR_H_INT) && isAnyRGB(o.dst_format)) ? 1 : 0;
    // ---- full-chroma 24 / 32 bpp RGB destinations (RGB -> RGB scaling, 4:4:4 sources, odd widths, the user's full_chroma_int) through the strip kernels:
    //      Y, U and V are all scaled to the destination size; the kernels store their vertical sums as int32 planes (DSTK_RAW32) into a working picture
    //      per frame and sws_k_fullchr_rgb finishes yuv2rgb_full_X_c_template.  Tentative: undone below when no strip plan fits or a row takes one of
    //      the writer's short forms ----
    // (a source alpha plane scaled into a 32 bpp destination -- bgra -> bgra, yuva420p -> rgba: needAlpha -- goes through the luma filters as a fourth sum
    //  plane: the A bytes of a packed 32 bpp source come from the reader pre-pass, a planar source has them in plane 3)
    // (planar RGB destinations of 8 .. 14 bits -- gbrp, gbrap, gbrp10le ...: yuv2gbrp_full_X_c is the same matrix with its own shifts, sws_k_fullchr_gbrp)
    const bool fc_alpha = c->needAlpha && (p.dstKind == DSTK_RGB32 || p.dstKind == DSTK_GBRP) &&
                          ((p.srcKind == SRCK_RGB32 && !(o.src_w & 1)) || ((p.srcKind == SRCK_PLANAR8 || p.srcKind == SRCK_PLANAR16) && isPlanarYUV(o.src_format)));
    // (the same for a planar YUV destination with an alpha plane -- bgra -> yuva420p, yuva444p10le -> yuva420p: the A samples through the luma filters
    //  and the luma plane's writer into dst[3], swscale.c:478-486 / vscale.c:66-70; decided with the strip plan below)
    const bool alpha_planar = c->plan == PLAN_MAIN && c->needAlpha && (p.dstKind == DSTK_PLANAR8 || p.dstKind == DSTK_PLANARN) && isPlanarYUV(o.dst_format) &&
                              !isGray(o.src_format) && !fast_flag && !c->tune.no_strip && !c->tune.no_mixed &&
                              ((p.srcKind == SRCK_RGB32 && !(o.src_w & 1)) || ((p.srcKind == SRCK_PLANAR8 || p.srcKind == SRCK_PLANAR16) && isPlanarYUV(o.src_format)));
    d->alpha_launch = 0;
    // (filters of more than 16 taps -- ratios of 4:1 and more -- have the strip kernel's long form with 128-column strips on one side and the element-per-thread
    //  kernels on the other: the planner's width threshold for them is 64 columns)
    const bool long_taps = c->plan == PLAN_MAIN && (hLumB.size >= 16 || hChrB.size >= 16 || c->vLum.size >= 16 || vChrB.size >= 24);   // (padded to an even start: 16 taps already take 9 pairs)
    const int strip_min_w_eff = long_taps ? std::min(c->tune.strip_min_w, 64) : c->tune.strip_min_w;   // (one strip of the long forms: thumbnails of 160 x 90 from 1080p are 0.05 ms on the two-pass kernels)
    const bool fc_plain = !isGray(o.src_format) && !isGray(o.dst_format) && p.srcKind != SRCK_MONO;
    d->fullchr_on = 0;
    // (round 5: gray sources of up to 16 bits into 24 / 32 bpp RGB take these routes too -- the luma launch's sums, the chroma sums written by sws_k_gray_chroma from the
    //  reference's constant chroma lines; a gray source counts as 4:4:4, so it is the full-chroma route unless the caller's flags say otherwise)
    const bool lut_gray = isGray(o.src_format) && !isALPHA(o.src_format) && !c->needAlpha && (p.srcKind == SRCK_PLANAR8 || p.srcKind == SRCK_PLANAR16) && !c->tune.no_strip_range && !p.wide &&
                          (p.dstKind == DSTK_RGB24 || p.dstKind == DSTK_RGB32);
    if (c->plan == PLAN_MAIN && (p.dstKind == DSTK_RGB24 || p.dstKind == DSTK_RGB32 || p.dstKind == DSTK_GBRP) && p.full_chr && (!c->needAlpha || fc_alpha) && (fc_plain || lut_gray) && !fast_flag &&
        o.dst_w >= strip_min_w_eff && !c->tune.no_strip && !c->tune.no_mixed) {
        d->fullchr_on = c->needAlpha ? 2 : 1; d->fullchr_kind = p.dstKind;
        p.dstKind = DSTK_RAW32; p.u_plane_dst = 1; p.v_plane_dst = 2;
    }
    // (the LUT writers with filters too long for sws_k_strip_rgb -- ratios of 4:1 and more: the same route with the chroma sums at half the width and
    //  sws_k_lut_rgb as the epilogue: fullchr_on == 3)
    // (round 5: ... and for planar / semi-planar sources with samples of 16 significant bits -- yuv4xxp16, p016: sws_k_strip_rgb's own staging takes samples of up to
    //  15 bits, the planar strip kernels take these, strip_hstage_b)
    const bool lut_u16 = !c->tune.no_strip_u16 && c->srcBpc == 16 && p.src_depth == 16 && p.src_shift == 0 && (p.srcKind == SRCK_PLANAR16 || p.srcKind == SRCK_P010);
    // (... and for the packed YUV sources the per-kind reader pre-pass serves -- y210 / y212 / xv30 / v30x / xv36, vyu444 / vuyx: sws_k_strip_rgb reads planar sources only)
    const bool lut_kind = !c->tune.no_rgbread_kinds && !isALPHA(o.src_format) &&
                          ((p.srcKind == SRCK_PACKEDHI && p.src_depth >= 9 && p.src_depth <= 15 && c->srcBpc == p.src_depth) || (p.srcKind == SRCK_PACKED444 && p.src_depth == 8 && c->srcBpc == 8));
    // (... and for RGB sources whose destination is not forced to full chroma -- RGB -> RGB with SWS_FAST_BILINEAR or an ordered dither, utils.c:1277-1285: the reader
    //  pre-pass in front, the LUT epilogue behind; round 5, without an alpha plane; rgb565 / x2rgb10 / 9 .. 16-bit RGB sources likewise through their per-kind readers)
    const bool lut_rgbsrc = !c->tune.no_short_forms && isAnyRGB(o.src_format) && !(o.src_w & 1) &&
                            (p.srcKind == SRCK_RGB24 || p.srcKind == SRCK_RGB32 || p.srcKind == SRCK_GBRP || p.srcKind == SRCK_RGB30 || p.srcKind == SRCK_RGB16 || p.srcKind == SRCK_GBRP16 || p.srcKind == SRCK_RGB48);
    if (!d->fullchr_on && c->plan == PLAN_MAIN && (p.dstKind == DSTK_RGB24 || p.dstKind == DSTK_RGB32) && !p.full_chr && (long_taps || lut_u16 || lut_kind || lut_gray || lut_rgbsrc) && !c->needAlpha && (fc_plain || lut_gray) &&
        !fast_flag && !(o.dst_w & 1) && o.dst_w >= strip_min_w_eff && !c->tune.no_strip && !c->tune.no_mixed) {
        d->fullchr_on = 3; d->fullchr_kind = p.dstKind;
        p.dstKind = DSTK_RAW32; p.u_plane_dst = 1; p.v_plane_dst = 2;
    }
    if (p.srcKind == SRCK_PACKEDHI)
        for (int k = 0; k < ds->nb_components; k++) {
            p.shi_step[k] = ds->comp[k].step; p.shi_off[k] = ds->comp[k].offset; p.shi_shift[k] = ds->comp[k].shift;
            p.shi_mask[k] = (1 << ds->comp[k].depth) - 1;
        }
    if (p.dstKind == DSTK_PACKEDHI) {
        const int df = o.dst_format;
        p.dhi_sub = dd->log2_chroma_w; p.dhi_bits = dd->comp[0].depth;
        p.dhi_unit_bytes = p.dhi_sub ? 8 : dd->comp[0].step;
        p.full_chr = p.dhi_sub ? 0 : 1;       // work unit: pixel pair (4:2:2) or pixel
        for (int k = 0; k < 3; k++) p.dhi_bitpos[k] = 8 * dd->comp[k].offset + dd->comp[k].shift;
        p.dhi_bitpos[3] = p.dhi_bitpos[0] + 32;                                  // second luma sample of a 4:2:2 unit
        p.dhi_alpha = df == AV_PIX_FMT_AYUV64LE; p.dhi_bitpos[4] = 0;
        uint64_t fill = 0;
        if (df == AV_PIX_FMT_XV30LE) fill = 3ull << 30;                          // yuv2v30_X_c_template: A = 3
        else if (df == AV_PIX_FMT_V30XLE) fill = 3ull;
        else if (df == AV_PIX_FMT_XV36LE) fill = 0xFFF0ull << 48;                // av_clip_uintp2(65535, 12) << 4
        else if (df == AV_PIX_FMT_XV48LE) fill = 0xFFFFull << 48;
        p.dhi_fill_lo = (uint32_t)fill; p.dhi_fill_hi = (uint32_t)(fill >> 32);
    }
    if (p.dstKind == DSTK_PACKED444) {   // one work unit per pixel, like the full-chroma RGB writers
        p.full_chr = 1;
        p.d444_step = dd->comp[0].step; p.d444_y = dd->comp[0].offset; p.d444_u = dd->comp[1].offset; p.d444_v = dd->comp[2].offset; p.d444_a = dd->comp[3].offset;
    }
    if (p.srcKind == SRCK_PACKED444) {
        p.s444_step = ds->comp[0].step; p.s444_y = ds->comp[0].offset; p.s444_u = ds->comp[1].offset; p.s444_v = ds->comp[2].offset; p.s444_a = ds->comp[3].offset;
    }
    if (isAnyRGB(o.src_format) && !isPlanarRGB(o.src_format)) {
        p.src_pix_step = ds->comp[0].step;
        p.src_r_pos = ds->comp[0].offset; p.src_g_pos = ds->comp[1].offset; p.src_b_pos = ds->comp[2].offset;
    }
    if (p.srcKind == SRCK_FLOATX) {
        p.sf_half = ds->comp[0].depth == 16;
        p.sf_layout = isPlanarRGB(o.src_format) ? 2 : isAnyRGB(o.src_format) ? 0 : 1;
        p.sf_step = ds->comp[0].step;
        p.sf_a_off = isALPHA(o.src_format) ? ds->comp[ds->nb_components - 1].offset : 0;
    }
    if (p.srcKind == SRCK_MONO) p.s16_is565 = o.src_format == AV_PIX_FMT_MONOWHITE;   // (reused: bits are stored inverted)
    p.dst_mono_white = o.dst_format == AV_PIX_FMT_MONOWHITE;
    p.mono_y16 = c->mono_y16 ? 1 : 0;
    if (p.srcKind == SRCK_RGB30) p.s16_is565 = o.src_format == AV_PIX_FMT_X2RGB10LE;   // (reused as the field-order flag of the 30 bpp reader)
    if (p.srcKind == SRCK_RGB16) {   // RGB16_32FUNCS rows of input.c:396-401
        switch (o.src_format) {
        case AV_PIX_FMT_BGR565LE: p.s16_maskr = 0x001F; p.s16_maskg = 0x07E0; p.s16_maskb = 0xF800; p.s16_rsh = 11; p.s16_gsh = 5; p.s16_bsh = 0; p.s16_S = 15 + 8; break;
        case AV_PIX_FMT_BGR555LE: p.s16_maskr = 0x001F; p.s16_maskg = 0x03E0; p.s16_maskb = 0x7C00; p.s16_rsh = 10; p.s16_gsh = 5; p.s16_bsh = 0; p.s16_S = 15 + 7; break;
        case AV_PIX_FMT_BGR444LE: p.s16_maskr = 0x000F; p.s16_maskg = 0x00F0; p.s16_maskb = 0x0F00; p.s16_rsh = 8; p.s16_gsh = 4; p.s16_bsh = 0; p.s16_S = 15 + 4; break;
        case AV_PIX_FMT_RGB565LE: p.s16_maskr = 0xF800; p.s16_maskg = 0x07E0; p.s16_maskb = 0x001F; p.s16_rsh = 0; p.s16_gsh = 5; p.s16_bsh = 11; p.s16_S = 15 + 8; break;
        case AV_PIX_FMT_RGB555LE: p.s16_maskr = 0x7C00; p.s16_maskg = 0x03E0; p.s16_maskb = 0x001F; p.s16_rsh = 0; p.s16_gsh = 5; p.s16_bsh = 10; p.s16_S = 15 + 7; break;
        default:                  p.s16_maskr = 0x0F00; p.s16_maskg = 0x00F0; p.s16_maskb = 0x000F; p.s16_rsh = 0; p.s16_gsh = 4; p.s16_bsh = 8; p.s16_S = 15 + 4; break;
        }
        p.s16_is565 = o.src_format == AV_PIX_FMT_RGB565LE || o.src_format == AV_PIX_FMT_BGR565LE;
    }
    p.chr_half = isAnyRGB(o.src_format) && c->chrSrcHSubSample;
    std::memcpy(p.rgb2yuv, c->rgb2yuv, sizeof(p.rgb2yuv));
    p.src_range = o.src_range;

    if (isAnyRGB(o.dst_format) && c->lut.valid) {
        const Yuv2RgbLut &l = c->lut;
        SwsLutParams &L = p.lut;
        auto fits = [](int64_t v) { return v >= INT32_MIN && v <= INT32_MAX; };
        auto fits24 = [](int64_t v) { return v > -(1 << 23) && v < (1 << 23); };
        if (!fits(l.yb0 + 0x8000 + 2048 * l.cy) || !fits(l.yb0 + 0x8000 - 2048 * l.cy) || !fits(255 * l.crv) || !fits(255 * l.cbu) ||
            !fits(255 * l.cgu) || !fits(255 * l.cgv) || !fits24(l.cy) || !fits24(l.crv) || !fits24(l.cbu) || !fits24(l.cgu) || !fits24(l.cgv)) {
            log_msg(c, 0, "brightness/contrast/saturation out of the range the HIP LUT closed form supports\n");
            return SWS_AVERROR(ENOTSUP);
        }
        L.cy = (int32_t)l.cy; L.yb0r = (int32_t)(l.yb0 + 0x8000);
        L.crv = (int32_t)l.crv; L.cbu = (int32_t)l.cbu; L.cgu = (int32_t)l.cgu; L.cgv = (int32_t)l.cgv;
        L.base_r = l.yoffs - (int32_t)(l.crv >> 9);
        L.base_b = l.yoffs - (int32_t)(l.cbu >> 9);
        L.base_g = l.yoffs - (int32_t)(l.cgu >> 9) - (int32_t)(l.cgv >> 9);
        const int df = o.dst_format;
        // yuv2rgb.c:941-961: AV_PIX_FMT_RGB32 == BGRA, RGB32_1 == ABGR, BGR32 == RGBA, BGR32_1 == ARGB (little endian)
        const bool isRgb = df == AV_PIX_FMT_BGRA || df == AV_PIX_FMT_ABGR || df == AV_PIX_FMT_BGR24;
        const int base = (df == AV_PIX_FMT_ABGR || df == AV_PIX_FMT_ARGB) ? 8 : 0;
        L.rshift = base + (isRgb ? 16 : 0); L.gshift = base + 8; L.bshift = base + (isRgb ? 0 : 16);
        L.alpha_or = isALPHA(o.src_format) ? 0u : (255u << ((base + 24) & 31));
        L.a_shift = (base + 24) & 31;
        L.rgb_order = df == AV_PIX_FMT_BGR24 ? 1 : 0;
        if (p.dstKind == DSTK_RGB30) {   // yuv2rgb.c:915-941: "255u << 30" keeps the two X bits set unless the source has alpha
            const bool x2rgb = df == AV_PIX_FMT_X2RGB10LE;
            L.bpp30 = 1;
            L.rshift = x2rgb ? 20 : 0; L.gshift = 10; L.bshift = x2rgb ? 0 : 20;
            L.alpha_or = isALPHA(o.src_format) ? 0u : 0xC0000000u;
        }
        if (p.dstKind == DSTK_RGB16) {   // yuv2rgb.c:853-897 (isRgb: the RGB565 / RGB555 / RGB444 orders, R in the high bits)
            const int bpp = pix_bits_per_pixel(dd);
            const bool rgb16 = df == AV_PIX_FMT_RGB565LE || df == AV_PIX_FMT_RGB555LE || df == AV_PIX_FMT_RGB444LE;
            L.bpp16 = bpp;
            L.r16 = bpp == 12 ? (rgb16 ? 8 : 0) : (rgb16 ? bpp - 5 : 0);
            L.g16 = bpp == 12 ? 4 : 5;
            L.b16 = bpp == 12 ? (rgb16 ? 0 : 8) : (rgb16 ? 0 : bpp - 5);
        }
        if (p.dstKind == DSTK_RGB8 || p.dstKind == DSTK_RGB4) {   // yuv2rgb.c:817-856 (isRgb: rgb8 / rgb4 / rgb4_byte, R in the high bits)
            const bool rgbo = df == AV_PIX_FMT_RGB8 || df == AV_PIX_FMT_RGB4 || df == AV_PIX_FMT_RGB4_BYTE;
            L.bpp8 = pix_bits_per_pixel(dd);
            if (L.bpp8 == 8) { L.r8 = rgbo ? 5 : 0; L.g8 = rgbo ? 2 : 3; L.b8 = rgbo ? 0 : 6; }
            else { L.r8 = rgbo ? 3 : 0; L.g8 = 1; L.b8 = rgbo ? 0 : 3; }
            L.dither8 = o.dither;
        }
        {   // 32 bpp wave kernels pack bytes as {c0, g, c2, 255} with c0 = R (or B when swap_rb32) and then permute:
            // rgba: R,G,B,A  bgra: B,G,R,A (swap)  argb: A,R,G,B  abgr: A,B,G,R (swap).  v_perm_b32(px, px, sel):
            // result byte i = source byte sel[i] (0..3 select from the second operand = px).
            L.swap_rb32 = (df == AV_PIX_FMT_BGRA || df == AV_PIX_FMT_ABGR) ? 1 : 0;
            L.perm32 = (df == AV_PIX_FMT_ARGB || df == AV_PIX_FMT_ABGR) ? 0x02010003u : 0x03020100u;
        }
        L.y_offset = l.y_offset; L.y_coeff = l.y_coeff; L.v2r = l.v2r; L.v2g = l.v2g; L.u2g = l.u2g; L.u2b = l.u2b;
        L.pix_step = dd->comp[0].step;
        L.r_pos = dd->comp[0].offset; L.g_pos = dd->comp[1].offset; L.b_pos = dd->comp[2].offset;
        L.a_pos = dd->nb_components > 3 ? dd->comp[3].offset : 0;
    }
    p.range_active = c->range.active; p.range_to_jpeg = !o.src_range;
    p.lumCoeff = c->range.lumCoeff; p.chrCoeff = c->range.chrCoeff;
    p.lumOffset = c->range.lumOffset; p.chrOffset = c->range.chrOffset;
    if (c->plan == PLAN_UNSC_P01X || c->plan == PLAN_UNSC_8_P01X) {       // swscale_unscaled.c:285-293
        p.shiftY = dd->comp[0].depth + dd->comp[0].shift - ds->comp[0].depth - ds->comp[0].shift;
        p.shiftU = dd->comp[1].depth + dd->comp[1].shift - ds->comp[1].depth - ds->comp[1].shift;
        p.shiftV = dd->comp[2].depth + dd->comp[2].shift - ds->comp[2].depth - ds->comp[2].shift;
    }
    p.s16_step = ds->comp[0].step / 2; p.s16_r = ds->comp[0].offset / 2; p.s16_g = ds->comp[1].offset / 2; p.s16_b = ds->comp[2].offset / 2;
    p.d16_step = dd->comp[0].step / 2; p.d16_r = dd->comp[0].offset / 2; p.d16_g = dd->comp[1].offset / 2; p.d16_b = dd->comp[2].offset / 2;
    p.s422_y = ds->comp[0].offset; p.s422_u = ds->comp[1].offset; p.s422_v = ds->comp[2].offset;
    p.d422_y = dd->comp[0].offset; p.d422_u = dd->comp[1].offset; p.d422_v = dd->comp[2].offset;
    p.need_alpha = c->needAlpha;                                                             // utils.c:1746
    p.src_a_pos = (isALPHA(o.src_format) && !isPlanarFmt(o.src_format)) ? ds->comp[3].offset : 0;
    p.src_alpha_opaque = c->src0Alpha && !c->dst0Alpha && isALPHA(o.dst_format);
    p.dst_alpha_fill = isALPHA(o.dst_format) && isPlanarFmt(o.dst_format) && !c->needAlpha;
    p.no_chroma = isGray(o.src_format) || isGray(o.dst_format) || p.srcKind == SRCK_MONO;                              // swscale.c:692-694
    p.fast_bilinear = (o.flags & SWS_FAST_BILINEAR) && c->srcBpc == 8 && c->dstBpc <= 14 && !fast_banks;   // swscale.c:676-681
    p.lumXInc = c->lumXInc; p.chrXInc = c->chrXInc;
    p.copy_depth_src = ds->comp[0].depth; p.copy_depth_dst = dd->comp[0].depth;
    p.copy_shift_src = ds->comp[0].shift; p.copy_shift_dst = dd->comp[0].shift;
    p.copy_shiftonly_luma = !o.src_range;
    p.dither_mode = o.dither;

    // ---- the other packed destinations behind the strip kernels (fullchr_on == 4): rgb565 / 555 / 444, x2rgb10 / x2bgr10, the 8-bit packed 4:4:4 formats
    //      (ayuv / vuya / vuyx / uyva / vyu444) and the packed YUV formats of 10 / 12 bits (y210 / y212, xv30 / v30x, xv36).  Same route as fullchr_on 1 / 3:
    //      the strip kernels store the vertical sums of Y, U and V as int32 planes (chroma at the writer's own chroma width), and the epilogue is the generic
    //      writer itself in its X form over those sums (k_generic_dst.hip sws_k_sum_writer).  Tentative like the others: undone below when no strip plan
    //      fits or a row takes one of the writer's short forms (the 10 / 12-bit packed YUV formats have X writers only) ----
    // (round 5: planar RGB of 16 bits and float32 -- gbrp16le, gbrpf32le: yuv2gbrp16_full_X_c / yuv2gbrpf32_full_X_c, output.c:2424-2610, always the X form -- over the
    //  sums of the 19-bit strip kernel, sws_k_strip_wide: decoded video into planar float RGB for inference)
    // (... and rgb48 / bgr48 / rgba64 / bgra64 without a scaled alpha plane: yuv2rgba64_X_c_template / yuv2rgba64_full_X_c_template, output.c:1115-1652, the generic
    //  writer's X form over the sums; rows in the writer's _1 / _2 forms keep the old kernels like the other packed kinds)
    const bool wide_gbrp = (((p.dstKind == DSTK_GBRP16 || p.dstKind == DSTK_GBRPF32) && !isALPHA(o.dst_format)) || p.dstKind == DSTK_RGB48) && p.wide && c->dstBpc >= 16 && !c->tune.no_strip_wide;
    if (!d->fullchr_on && c->plan == PLAN_MAIN && (((p.dstKind == DSTK_RGB16 || p.dstKind == DSTK_RGB30 || p.dstKind == DSTK_PACKED444 || p.dstKind == DSTK_PACKEDHI) && !p.wide &&
        c->dstBpc <= 14) || wide_gbrp) && !c->needAlpha && fc_plain && !fast_flag && !(o.dst_w & 1) && o.dst_w >= strip_min_w_eff && c->chrDstVSubSample == 0 &&
        !(bank_is_identity(hLumB, 1 << 14) && bank_is_identity(hChrB, 1 << 14)) &&   // (identity horizontal filters: the single-pass per-kind kernels are as fast or faster -- no sum planes)
        !c->tune.no_strip && !c->tune.no_mixed && !(c->tune.no_rgbread_kinds & 2)) {
        d->fullchr_on = 4; d->fullchr_kind = p.dstKind;
        p.dstKind = DSTK_RAW32; p.u_plane_dst = 1; p.v_plane_dst = 2;
    }
    // ---- what the later stages read (dev_plan.hpp PLAN_LOCALS gives them their names back) ----
    B.hLumB = &hLumB; B.hChrB = &hChrB; B.vChrB = &vChrB;
    B.fast_banks = fast_banks; B.join_short = join_short; B.striprgb_short = striprgb_short; B.rgb2rgb_short = rgb2rgb_short; B.fast_flag = fast_flag;
    B.gray_any = gray_any; B.fc_alpha = fc_alpha; B.alpha_planar = alpha_planar; B.long_taps = long_taps; B.strip_min_w_eff = strip_min_w_eff;
    B.fc_plain = fc_plain; B.lut_gray = lut_gray; B.lut_u16 = lut_u16; B.lut_kind = lut_kind; B.lut_rgbsrc = lut_rgbsrc; B.wide_gbrp = wide_gbrp;
    return 0;
}

// ---- stage 5: name the path (for SWS_PRINT_INFO, tests and rocprof matching) ----
void plan_name(PlanBuild &B)
{
    PLAN_LOCALS(B);
    switch (c->plan) {
    case PLAN_UNSC_YUV2RGB: c->path_name = "unscaled:yuv2rgb"; c->kernel_name = "sws_k_yuv2rgb_unscaled_wave"; break;
    case PLAN_UNSC_P01X: c->path_name = "unscaled:planarToP01x"; c->kernel_name = "sws_k_p01x_stream"; break;
    case PLAN_UNSC_8_P01X: c->path_name = "unscaled:planar8ToP01xle"; c->kernel_name = "sws_k_p01x_unscaled"; break;
    case PLAN_UNSC_PLANAR2NV12: c->path_name = "unscaled:planarToNv12"; c->kernel_name = "sws_k_planar_misc"; break;
    case PLAN_UNSC_NV122PLANAR: c->path_name = "unscaled:nv12ToPlanar"; c->kernel_name = "sws_k_planar_misc"; break;
    case PLAN_UNSC_PLANARCOPY: c->path_name = "unscaled:planarCopy"; c->kernel_name = "sws_k_planar_misc"; break;
    case PLAN_UNSC_RGB2RGB: c->path_name = "unscaled:rgbToRgb"; c->kernel_name = "sws_k_rgb_shuffle"; break;
    case PLAN_UNSC_PACKEDCOPY: c->path_name = "unscaled:packedCopy"; c->kernel_name = "sws_k_packed_copy"; break;
    case PLAN_UNSC_BGR24_YV12: c->path_name = "unscaled:bgr24ToYv12"; c->kernel_name = "sws_k_bgr24_to_yv12"; break;
    case PLAN_UNSC_GBRP_PACKED: c->path_name = "unscaled:planarRgbToRgb"; c->kernel_name = "sws_k_gbrp_to_packed"; break;
    case PLAN_UNSC_PLANAR2NV24: c->path_name = "unscaled:planarToNv24"; c->kernel_name = "sws_k_planar_misc"; break;
    case PLAN_UNSC_NV242PLANAR: c->path_name = "unscaled:nv24ToPlanar"; c->kernel_name = "sws_k_planar_misc"; break;
    case PLAN_UNSC_NV242YUV420: c->path_name = "unscaled:nv24ToYuv420"; c->kernel_name = "sws_k_planar_misc"; break;
    case PLAN_UNSC_YVU9_YV12: c->path_name = "unscaled:yvu9ToYv12"; c->kernel_name = "sws_k_planar_misc"; break;
    case PLAN_UNSC_YUV2GBRP: c->path_name = "unscaled:yuv2rgb"; c->kernel_name = "sws_k_yuv2gbrp_unscaled"; break;
    case PLAN_UNSC_PACKED_GBRP: c->path_name = "unscaled:rgbToPlanarRgb"; c->kernel_name = "sws_k_packed_to_gbrp"; break;
    case PLAN_UNSC_RGB16SHUFFLE: c->path_name = "unscaled:rgb16Shuffle"; c->kernel_name = "sws_k_rgb16_convert"; break;
    case PLAN_UNSC_PACKED16_GBRP16: c->path_name = "unscaled:Rgb16ToPlanarRgb16"; c->kernel_name = "sws_k_rgb16_convert"; break;
    case PLAN_UNSC_GBRP16_PACKED16: c->path_name = "unscaled:planarRgb16ToRgb16"; c->kernel_name = "sws_k_rgb16_convert"; break;
    case PLAN_UNSC_U8_TO_F32: c->path_name = "unscaled:uint_y_to_float_y"; c->kernel_name = "sws_k_gray_f32"; break;
    case PLAN_UNSC_F32_TO_U8: c->path_name = "unscaled:float_y_to_uint_y"; c->kernel_name = "sws_k_gray_f32"; break;
    case PLAN_UNSC_YUV2MONO: c->path_name = "unscaled:yuv2rgb"; c->kernel_name = "sws_k_yuv2mono_unscaled"; break;
    case PLAN_UNSC_RGB30_TO_16: c->path_name = "unscaled:rgbToRgb"; c->kernel_name = "sws_k_rgb30_convert"; break;
    case PLAN_UNSC_RGB30_TO_GBRP: c->path_name = "unscaled:Rgb16ToPlanarRgb16"; c->kernel_name = "sws_k_rgb30_convert"; break;
    case PLAN_UNSC_GBRP_TO_RGB30: c->path_name = "unscaled:planarRgb16ToRgb16"; c->kernel_name = "sws_k_rgb30_convert"; break;
    case PLAN_UNSC_YUV2RGB48: c->path_name = "unscaled:yuv2rgb"; c->kernel_name = "sws_k_yuv2rgb48_unscaled"; break;
    case PLAN_UNSC_YUV2RGB16: c->path_name = "unscaled:yuv2rgb"; c->kernel_name = "sws_k_yuv2rgb16_unscaled"; break;
    case PLAN_UNSC_YUV2RGB8: c->path_name = "unscaled:yuv2rgb"; c->kernel_name = "sws_k_yuv2rgb8_unscaled"; break;
    case PLAN_UNSC_PAL2RGB: c->path_name = "unscaled:palToRgb"; c->kernel_name = "sws_k_pal2rgb"; break;
    case PLAN_UNSC_BAYER: c->path_name = "unscaled:bayer"; c->kernel_name = "sws_k_bayer"; break;
    case PLAN_UNSC_RGBLOW: c->path_name = "unscaled:rgbToRgb"; c->kernel_name = "sws_k_rgb_low_convert"; break;
    case PLAN_UNSC_PLANAR2P422: c->path_name = "unscaled:planarToYuy2"; c->kernel_name = "sws_k_planar_to_p422"; break;
    case PLAN_UNSC_P4222PLANAR: c->path_name = "unscaled:yuyvToPlanar"; c->kernel_name = "sws_k_p422_to_planar"; break;
    case PLAN_UNSC_ALPHABLEND: c->path_name = "unscaled:alphablendaway"; c->kernel_name = "sws_k_alphablend"; break;
    case PLAN_UNSC_PLANARRGB_PLANARRGB: c->path_name = "unscaled:planarRgbToplanarRgb"; c->kernel_name = "sws_k_planarrgb_copy"; break;
    case PLAN_CASCADE: c->path_name = "cascade"; c->kernel_name = ""; break;
    case PLAN_MAIN: {
        const bool rgb_lut = (p.dstKind == DSTK_RGB24 || p.dstKind == DSTK_RGB32) && !p.full_chr;
        if (d->vlines_on) {
            c->path_name = "main:two_pass"; c->kernel_name = "sws_k_hscale";
        } else if (d->unity_h && rgb_lut && !p.no_chroma && !p.need_alpha && c->srcBpc == 8 && (p.srcKind == SRCK_PLANAR8 || p.srcKind == SRCK_NV12)) {
            c->path_name = "main:fused_rgb_unity";
            c->kernel_name = d->rgb_march_ok ? "sws_k_rgb_march" : (d->all_x_mode && d->chr_window2 <= 8 ? "sws_k_rgb_fused_unity_wave2" : "sws_k_rgb_fused_unity");
        } else if (d->unity_h && d->unity_v && !p.no_chroma && !p.need_alpha && p.srcKind == SRCK_GBRPF32 && p.chrDstHSub == 0 && p.chrDstVSub == 0 && p.dst_shift == 0 &&
                   (p.dstKind == DSTK_PLANAR8 || p.dstKind == DSTK_PLANARN || p.dstKind == DSTK_PLANAR16)) {
            c->path_name = "main:fused_f32rgb_yuv444"; c->kernel_name = "sws_k_f32rgb_to_yuv444_unity";
        } else if (d->rgbsrc_ok) {
            c->path_name = "main:rgbsrc_unity"; c->kernel_name = (d->rgbsrc2_rows && !c->tune.no_rgbsrc2 && !(p.dstW & 3)) ? "sws_k_rgbsrc_unity2" : "sws_k_rgbsrc_unity";
        } else if (d->rgb444_ok) {
            c->path_name = "main:rgb_yuv444_unity"; c->kernel_name = "sws_k_rgb_yuv444_unity";
        } else if (d->striprgb_ok) {
            c->path_name = "main:strip_rgb"; c->kernel_name = (d->stripRL.dma8_ok && !c->tune.no_strip_dma8) ? "sws_k_strip_rgb8" : "sws_k_strip_rgb";
        } else if (d->mixed_ok) {
            c->path_name = "main:plane1+strip_chroma"; c->kernel_name = "sws_k_strip_march";
            if (isGray(c->opts.src_format)) { c->path_name = "main:plane1+gray_chroma"; c->kernel_name = "sws_k_layout_stream"; }
        } else if (d->strip_ok) {
            c->path_name = d->rgbread_on ? ((d->striprgbsrc_ok && !c->tune.no_strip_rgbsrc) ? "main:strip_rgbsrc" : "main:rgbread+strip_march") : "main:strip_march";
            c->kernel_name = p.wide ? "sws_k_strip_wide" : ((p.srcKind == SRCK_PLANAR16 || d->rgbread_on) && d->stripL.dma_ok && !c->tune.no_strip_dma) ? "sws_k_strip_dma" : "sws_k_strip_march";
            if (d->rgbread_on && d->striprgbsrc_ok && !c->tune.no_strip_rgbsrc) c->kernel_name = "sws_k_strip_rgbsrc";
            else if (d->stripL.nph > 8 || d->stripL.npv > 8) c->kernel_name = d->stripL.nph > 16 ? "sws_k_strip_xlong" : "sws_k_strip_long";   // (filters of 17 .. 32 / 33 .. 62 taps)
            else if (!p.wide && !c->tune.no_strip_short && (p.srcKind == SRCK_PLANAR8 || p.srcKind == SRCK_NV12)) {   // the short family (k_strip2.hip launch_strip_short decides per launch: this is its choice for 16-byte aligned frames)
                const SwsStripGeom &gs = d->stripLs_ok ? d->stripLs : d->stripL;
                const bool d8 = gs.dma8_ok && !c->tune.no_strip_dma8;
                if (gs.NCmax / 16 <= 64 && (d8 ? gs.npv <= 8 && gs.nph8 <= 6 : gs.npv <= 6 && gs.nph <= 6)) c->kernel_name = d8 ? "sws_k_strip_dma8" : "sws_k_strip_short";
            }
        } else if (d->dot2_ok) {
            c->path_name = "main:fused_tile_dot2"; c->kernel_name = "sws_k_tile_dot2";
        } else if (d->tile_ok) {
            c->path_name = "main:fused_tile"; c->kernel_name = "sws_k_tile_planar";
        } else if (d->unity_h) {
            c->path_name = "main:fused_generic_unity";
            c->kernel_name = (p.dstKind == DSTK_RGB24 || p.dstKind == DSTK_RGB32) ? "sws_k_vscale_rgb" : "sws_k_vscale_planar";
        } else {
            c->path_name = "main:two_pass";
            c->kernel_name = "sws_k_hscale";
        }
        break;
    }
    default: c->path_name = "none"; c->kernel_name = ""; break;
    }
    // (nvdirect: the strip-RGB kernel reads the semi-planar source itself on 16-byte aligned frames -- launch_plan_le falls back to the split pass otherwise)
    if (c->plan == PLAN_MAIN && d->split_mode) c->path_name = ((d->split_mode & 40) ? ((d->striprgb_ok && d->striprgb_direct && !c->tune.no_striprgb_direct) ? "main:nvdirect+" : "main:splitnv+") : "main:split422+") +
                                                             c->path_name.substr(c->path_name.find(':') + 1);
    if (c->plan == PLAN_MAIN && d->split_mode && d->striprgb_direct == 3 && d->strip_ok && d->striprgbsrc_ok && !d->mixed_ok && !d->striprgb_ok && !d->fullchr_on && !d->alpha_launch && !d->rgbread_on &&
        !c->tune.no_strip_rgbsrc) { c->path_name = "main:strip_packed422"; c->kernel_name = "sws_k_strip_rgbsrc"; }     // (aligned frames; launch_plan_le falls back to split422 + strip_march otherwise)
    if (c->plan == PLAN_MAIN && d->join422) c->path_name += "+join422";
    // (aligned frames: k_stream.hip launch_mixed_join422 takes the mixed plan and its interleave as one pass)
    if (c->plan == PLAN_MAIN && d->join422 && d->mixed_ok && !d->fullchr_on && !isGray(c->opts.src_format) && mixed_join422_shape(c, d, p)) { c->path_name = "main:mixed_join422"; c->kernel_name = "sws_k_mixed_join422"; }
    if (c->plan == PLAN_MAIN && d->fullchr_on) c->path_name += d->fullchr_on == 3 ? "+lut_rgb" : d->fullchr_on == 4 ? (((d->fullchr_kind == DSTK_GBRP16 || d->fullchr_kind == DSTK_GBRPF32) && c->tune.no_wide_epilogue != 1) ? ((d->rgbread_on || c->tune.no_wide_epilogue == 2) ? "+fullchr_gbrp16" : "+fused_gbrp16") : "+sum_writer") : "+fullchr_rgb";
    if (c->plan == PLAN_MAIN && d->rgb2rgb_ok && !c->tune.no_strip_rgb2rgb && (d->fullchr_on == 1 || d->fullchr_on == 2) && !d->fullchr_direct && d->rgbread_on && d->strip_ok) {
        c->path_name = "main:strip_rgb2rgb"; c->kernel_name = "sws_k_strip_rgb2rgb";      // (aligned frames; launch_plan_le falls back to rgbread + strip_march + fullchr_rgb otherwise)
    }
    if (c->plan == PLAN_MAIN && d->fullchr_on && d->fullchr_direct) { c->path_name = "main:fullchr_rgb_direct"; c->kernel_name = d->fullchr_kind == DSTK_GBRP ? "sws_k_fullchr_gbrp" : "sws_k_fullchr_rgb"; }
    if (c->plan == PLAN_MAIN && ((d->alpha_launch == 1 && d->strip_ok) || (d->alpha_launch == 2 && d->striprgb_ok))) c->path_name += "+alpha";
    // (16-byte aligned pictures of the layout converters take the streaming kernel, k_layout.hip; the names above are the fallback's)
    if (!c->tune.no_layout_stream && (c->plan == PLAN_UNSC_PLANAR2NV12 || c->plan == PLAN_UNSC_NV122PLANAR || c->plan == PLAN_UNSC_PLANARCOPY || c->plan == PLAN_UNSC_PLANAR2NV24 ||
                                      c->plan == PLAN_UNSC_NV242PLANAR || c->plan == PLAN_UNSC_P4222PLANAR || c->plan == PLAN_UNSC_PLANAR2P422))
        c->kernel_name = "sws_k_layout_stream";
    log_msg(c, 2, "HIP path: %s (dominant kernel %s)\n", c->path_name.c_str(), c->kernel_name.c_str());
}

int dev_prepare_on(SwsInternal *c, DeviceState *d)
{
    if (d->epoch == c->tables_epoch && d->stream) return 0;
    if (d->dry) d->stream = (hipStream_t)(uintptr_t)1;      // (never handed to HIP: a planner-only state is never launched on)
    else {
    HIPCHK(hipSetDevice(d->device));
    if (!d->stream) { HIPCHK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking)); d->own_stream = true; }
    // a rebuild rewrites plan tables that launches still in flight on the (non-blocking) stream may be reading
    else HIPCHK(hipStreamSynchronize(d->stream));
    }
    PlanBuild B;
    B.c = c; B.d = d; B.ds = pix_desc(c->opts.src_format); B.dd = pix_desc(c->opts.dst_format);
    { int r_ = plan_common(B); if (r_ < 0) return r_; }
    // ---- filter tables -> one device blob; the plans of the PLAN_MAIN kernels ----
    d->unity_h = false;
    if (c->plan == PLAN_MAIN) {
        { int r_ = plan_tables(B); if (r_ < 0) return r_; }
        { int r_ = plan_strip(B); if (r_ < 0) return r_; }
        { int r_ = plan_tile(B); if (r_ < 0) return r_; }
    }
    plan_name(B);
    d->params_hash = fnv1a64(&d->params, sizeof(d->params));
    d->epoch = c->tables_epoch;
    return 0;
}

int dev_prepare(SwsInternal *c)
{
    int ret = ensure_dev(c);
    if (ret < 0) return ret;
    DeviceGuard guard;
    return dev_prepare_on(c, c->dev);
}

// The plan of a context as two numbers: a digest of every table block the planner uploaded (sizes, order of the blocks in the state, contents) and a digest of the
// kernel parameters (SwsDevParams: geometry, constants and the pointers into the table blocks).  The first is the same with and without a GPU; the second is
// reproducible for dry_plan contexts only (fake table addresses).  tests/test_planner_table.py pins (path, kernel, digests) per conversion on the CPU box.
// the host-side half of a plan: which kernels run and on what geometry (DeviceState's flags and strip / tile geometries, whose pointers point into the table blocks)
static uint64_t state_digest(const DeviceState *d)
{
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t n) { for (size_t i = 0; i < n; i++) { h ^= ((const uint8_t *)p)[i]; h *= 1099511628211ull; } };
    const int flags[] = { d->unity_h, d->unity_v, d->all_x_mode, d->chr_window2, d->tile_ok, d->rgb_march_ok, d->rgb_groups, d->rgb_ncr, d->dot2_ok, d->rgbsrc2_npv, d->rgb444_ok,
                          d->rgbsrc_ok, d->mixed_ok, d->strip_ok, d->stripLs_ok, d->stripCs_ok, d->striprgb_ok, d->striprgb_long, d->striprgb_direct, d->striprgb_direct_swap,
                          d->striprgb_direct_shift, d->rgb2rgb_ok, d->rgb2rgb_npx, d->striprgbsrc_ok, d->striprgbsrc_npx, d->rgbread_on, d->fullchr_on, d->fullchr_kind,
                          d->fullchr_direct, d->alpha_launch, d->join422, d->split_mode, d->split_shift, d->vlines_on };
    mix(flags, sizeof(flags));
    if (d->tile_ok) { mix(&d->tileL, sizeof(d->tileL)); mix(&d->tileC, sizeof(d->tileC)); }
    if (d->dot2_ok) { mix(&d->dotL, sizeof(d->dotL)); mix(&d->dotC, sizeof(d->dotC)); }
    if (d->strip_ok) { mix(&d->stripL, sizeof(d->stripL)); mix(&d->stripC, sizeof(d->stripC)); }
    if (d->stripLs_ok) mix(&d->stripLs, sizeof(d->stripLs));
    if (d->stripCs_ok) mix(&d->stripCs, sizeof(d->stripCs));
    if (d->striprgb_ok) { mix(&d->stripRL, sizeof(d->stripRL)); mix(&d->stripRC, sizeof(d->stripRC)); }
    if (d->rgb2rgb_ok) { mix(&d->stripL2, sizeof(d->stripL2)); mix(&d->stripC2, sizeof(d->stripC2)); }
    const void *ptrs[] = { d->rgbsrc_rows, d->rgbsrc2_rows };
    mix(ptrs, sizeof(ptrs));
    return h;
}

int dev_plan_digest(SwsInternal *c, uint64_t out[3])
{
    int r = dev_prepare(c);
    if (r < 0) return r;
    DeviceState *d = c->dev;
    std::vector<TableRecord> recs;
    for (const TableRecord &r : d->tab_recs) if ((r.serial & ~TAB_VERIFIED) == d->plan_serial) recs.push_back(r);      // what THIS plan wrote (a re-planned context may still hold blocks of an earlier plan)
    // (by size and contents, not by address: the digest of a context that ran on a GPU is then comparable with a fresh context's and with the dry planner's)
    std::sort(recs.begin(), recs.end(), [](const TableRecord &a, const TableRecord &b) { return a.bytes != b.bytes ? a.bytes < b.bytes : a.hash < b.hash; });
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { for (int i = 0; i < 8; i++) { h ^= (v >> (8 * i)) & 0xff; h *= 1099511628211ull; } };
    for (size_t i = 0; i < recs.size(); i++) { mix(i); mix(recs[i].bytes); mix(recs[i].hash); }
    out[0] = h;
    out[1] = d->params_hash;
    out[2] = state_digest(d);
    return 0;
}

} // namespace swship

