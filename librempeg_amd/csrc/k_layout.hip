// Translation unit of the streaming layout / depth converters (kernels_layout.hpp): the plan of row classes for
// planarCopyWrapper (+DITHER_COPY), planarToNv12 / Nv24, nv12 / nv24ToPlanar, yuyv / uyvy <-> planar.
#include <algorithm>

#include "devstate.hpp"
#include "kernels_layout.hpp"

namespace swship {

// 1: launched; 0: not a shape of this family (the caller falls back to the element-per-thread kernels)
int launch_layout(const LaunchCtx &L)
{
    SwsInternal *c = L.c; const SwsDevParams &p = *L.p; hipStream_t st = L.st; const SwsFrameSet &fs = L.fs;
    const int n = L.n, sliceY = L.sliceY, sliceH = L.sliceH;
    if (!L.vec || c->tune.no_layout_stream || sliceH <= 0) return 0;
    using namespace swsk;
    LayoutPlan plan;
    std::memset(&plan, 0, sizeof(plan));
    auto add = [&](int op, int rows, int ys, int yd, int sa, int sb, int da, int db, int nbytes, int a0 = 0, int a1 = 0, int a2 = 0, int a3 = 0, int a4 = 0) {
        if (rows <= 0 || nbytes <= 0) return;
        LayoutJob &j = plan.job[plan.njobs++];
        j.op = op; j.rows = rows; j.ys = ys; j.yd = yd; j.sa = sa; j.sb = sb; j.da = da; j.db = db; j.n = nbytes;
        j.a0 = a0; j.a1 = a1; j.a2 = a2; j.a3 = a3; j.a4 = a4;
    };
    const int sf = c->opts.src_format, df = c->opts.dst_format;
    switch (c->plan) {
    case PLAN_UNSC_PLANAR2NV12: case PLAN_UNSC_PLANAR2NV24: {   // planarToNv12Wrapper / planarToNv24Wrapper (swscale_unscaled.c:147-227)
        const bool nv24 = c->plan == PLAN_UNSC_PLANAR2NV24;
        const int a = p.uv_swap_dst ? 2 : 1;
        add(LOP_COPY, sliceH, sliceY, sliceY, 0, 0, 0, 0, p.srcW);
        add(LOP_IL, nv24 ? sliceH : (sliceH + 1) / 2, nv24 ? sliceY : sliceY / 2, nv24 ? sliceY : sliceY / 2, a, 3 - a, 1, 1, c->chrSrcW);
        break;
    }
    case PLAN_UNSC_NV122PLANAR: case PLAN_UNSC_NV242PLANAR: {
        const bool nv24 = c->plan == PLAN_UNSC_NV242PLANAR;
        const int a = p.uv_swap_src ? 2 : 1;
        add(LOP_COPY, sliceH, sliceY, sliceY, 0, 0, 0, 0, p.srcW);
        add(LOP_DIL, nv24 ? sliceH : (sliceH + 1) / 2, nv24 ? sliceY : sliceY / 2, nv24 ? sliceY : sliceY / 2, 1, 1, a, 3 - a, 2 * c->chrSrcW);
        break;
    }
    case PLAN_UNSC_PLANARCOPY: {   // planarCopyWrapper (:2220-2384), plane by plane like launch_misc builds its MiscPlan
        const PixDesc *ds = pix_desc(sf), *dd = pix_desc(df);
        const int sd = ds->comp[0].depth, ddp = dd->comp[0].depth, ss = ds->comp[0].shift, dsh = dd->comp[0].shift;
        if ((sd != 8 && sd < 9) || (ddp != 8 && ddp < 9) || sd > 16 || ddp > 16) return 0;   // (float gray and the like keep the old path)
        const bool same = sd == ddp && ss == dsh && !isNBPS(sf) && !isNBPS(df);
        const int np = pix_nb_planes(dd);
        const int sbytes = (sd + 7) / 8, dbytes = (ddp + 7) / 8;
        for (int pl = 0; pl < np; pl++) {
            int len = pl == 0 || pl == 3 ? p.srcW : -((-p.srcW) >> c->chrDstHSubSample);      // samples
            const int y0 = pl == 0 || pl == 3 ? sliceY : -((-sliceY) >> c->chrDstVSubSample);
            const int h = pl == 0 || pl == 3 ? sliceH : -((-sliceH) >> c->chrDstVSubSample);
            if (pl == 1 && isSemiPlanarYUV(df)) len *= 2;
            const int shiftonly = pl == 1 || pl == 2 || (!c->opts.src_range && pl == 0);   // (plane 3: converted like luma with shiftonly = 0)
            const bool missing = (pl == 1 || pl == 2) && isGray(sf);
            if (pl == 3 && !isALPHA(sf)) { const uint32_t v = ddp > 8 ? (0xFFFFu >> (16 - ddp)) * 0x10001u : 0xFFFFFFFFu; add(LOP_FILL, h, y0, y0, 0, 0, 3, 3, len * dbytes, (int)v); continue; }
            if (missing) { const uint32_t v = ddp > 8 ? (1u << (ddp - 1)) * 0x10001u : 0x80808080u; add(LOP_FILL, h, y0, y0, 0, 0, pl, pl, len * dbytes, (int)v); continue; }
            const int so = pl == 3 ? 0 : shiftonly;
            if (same) add(LOP_COPY, h, y0, y0, pl, pl, pl, pl, len * sbytes);
            else if (ddp == 8 || sd > ddp) {   // DITHER_COPY (:2159-2218)
                const int shift = sd - ddp, body_end = len - 7 > 0 ? ((len - 7 + 7) / 8) * 8 : 0;
                const int mode = c->opts.dither == SWS_DITHER_NONE ? 0 : so ? 1 : 2;
                add(ddp == 8 ? LOP_16TO8 : LOP_16TO16, h, y0, y0, pl, pl, pl, pl, 2 * len, mode | (sd <= 15 ? 16 : 0), shift, ss, dsh, ddp | (body_end << 8));
            } else if (sd == 8) add(LOP_8TO16, h, y0, y0, pl, pl, pl, pl, len, ddp - 8, so ? 32 : 16 - ddp, dsh);
            else add(LOP_16TO16, h, y0, y0, pl, pl, pl, pl, 2 * len, (so ? 3 : 4) | 16, ddp - sd, ss, dsh, 2 * sd - ddp);   // (widening: nothing leaves 16 bits)
        }
        break;
    }
    case PLAN_UNSC_P4222PLANAR: {   // yuyv / uyvy ToYuv420 / 422Wrapper (:424-484)
        const int w = p.srcW, cw = (w + 1) >> 1;
        const int uyvy = p.s422_y == 1, swap = p.s422_u > p.s422_v;
        if (df != AV_PIX_FMT_YUV422P) add(LOP_P422_SPLIT420, (sliceH + 1) / 2, sliceY, sliceY, 0, 0, 0, 1, 4 * cw, uyvy, swap, w, sliceH);
        else add(LOP_P422_SPLIT, sliceH, sliceY, sliceY, 0, 0, 0, 1, 4 * cw, uyvy, swap, w);
        break;
    }
    case PLAN_UNSC_PLANAR2P422: {   // yuv422pToYuy2 / UyvyWrapper, planarToYuy2 / UyvyWrapper (:376-422)
        const int vlpc = sf == AV_PIX_FMT_YUV422P ? 1 : 2;
        add(LOP_P422_JOIN, sliceH, sliceY, sliceY, 0, 0, 0, 0, 2 * (p.srcW >> 1), p.d422_y == 1, vlpc, vlpc == 2 ? (sliceY >> 1) : sliceY);
        break;
    }
    default: return 0;
    }
    if (!plan.njobs) return 0;
    int groups = 0, chunks = 0;
    for (int i = 0; i < plan.njobs; i++) {
        const LayoutJob &j = plan.job[i];
        groups += cdiv(j.rows, LAYOUT_RPW);
        const int unit = (j.op == LOP_IL || j.op == LOP_8TO16 || j.op == LOP_P422_JOIN) ? 8 : 16;
        chunks = std::max(chunks, (int)cdiv(j.n, unit));
    }
    const dim3 blk(256), grid(cdiv(chunks, 256), groups, n);
    if (c->tune.layout_ch == 2) hipLaunchKernelGGL((sws_k_layout_stream<2>), grid, blk, 0, st, fs, p, plan);
    else hipLaunchKernelGGL((sws_k_layout_stream<4>), grid, blk, 0, st, fs, p, plan);
    return 1;
}

// The de-interleaving pass ahead of the kernels for a packed 4:2:2 SOURCE of the scaler (dev_exec.hip; L.fs holds {src = the packed picture, dst = the
// planes of the working picture}): yuyvtoyuv422_c / uyvytoyuv422_c over the whole source picture.
void launch_layout_split422(const LaunchCtx &L, bool uyvy, bool vfirst)
{
    const SwsDevParams &p = *L.p;
    using namespace swsk;
    LayoutPlan plan;
    std::memset(&plan, 0, sizeof(plan));
    LayoutJob &j = plan.job[0];
    plan.njobs = 1;
    const int w = p.srcW, cw = (w + 1) >> 1;
    j.op = LOP_P422_SPLIT; j.rows = p.srcH; j.ys = 0; j.yd = 0; j.sa = j.sb = 0; j.da = 0; j.db = 1;
    j.n = 4 * cw; j.a0 = uyvy ? 1 : 0; j.a1 = vfirst ? 1 : 0; j.a2 = w;
    hipLaunchKernelGGL((sws_k_layout_stream<4>), dim3(cdiv(cdiv(j.n, 16), 256), cdiv(j.rows, LAYOUT_RPW), L.n), dim3(256), 0, L.st, L.fs, p, plan);
}

// The chroma plane of a semi-planar 8-bit source of the scaler, split into planar working planes (dev_exec.hip; L.fs holds {src[1] = the interleaved
// plane, dst[1] / dst[2] = the U / V planes}): nvXXtoUV_c over the whole plane (vfirst: nv21 / nv42).
void launch_layout_splitnv(const LaunchCtx &L, bool vfirst)
{
    const SwsDevParams &p = *L.p;
    using namespace swsk;
    LayoutPlan plan;
    std::memset(&plan, 0, sizeof(plan));
    LayoutJob &j = plan.job[0];
    plan.njobs = 1;
    const int a = vfirst ? 2 : 1;
    j.op = LOP_DIL; j.rows = p.chrSrcH; j.ys = 0; j.yd = 0; j.sa = j.sb = 1; j.da = a; j.db = 3 - a;
    j.n = 2 * p.chrSrcW;
    hipLaunchKernelGGL((sws_k_layout_stream<4>), dim3(cdiv(cdiv(j.n, 16), 256), cdiv(j.rows, LAYOUT_RPW), L.n), dim3(256), 0, L.st, L.fs, p, plan);
}

// A p010 / p012 / p210 / p410-style source of the scaler (16-bit words, samples in the high bits, chroma interleaved) as a planar working picture
// with the samples in the low bits (dev_exec.hip; L.fs holds {src[0] / src[1] = the two planes, dst[0..2] = Y / U / V planes}): p010LEToY_c /
// p010LEToUV_c (input.c:950-1008) are `word >> shift` and a de-interleave.
void launch_layout_splitp01x(const LaunchCtx &L, int shift)
{
    const SwsDevParams &p = *L.p;
    using namespace swsk;
    LayoutPlan plan;
    std::memset(&plan, 0, sizeof(plan));
    plan.njobs = 2;
    LayoutJob &y = plan.job[0], &cjob = plan.job[1];
    y.op = LOP_16TO16; y.rows = p.srcH; y.sa = y.sb = y.da = y.db = 0; y.n = 2 * p.srcW;
    y.a0 = 3 | 16; y.a1 = 0; y.a2 = shift; y.a3 = 0; y.a4 = 0;        // mode 3 (shift only), packed: (word >> shift) << 0
    cjob.op = LOP_DIL16; cjob.rows = p.chrSrcH; cjob.sa = cjob.sb = 1; cjob.da = 1; cjob.db = 2; cjob.n = 4 * p.chrSrcW; cjob.a0 = shift;
    int groups = 0, chunks = 0;
    for (int i = 0; i < plan.njobs; i++) { groups += cdiv(plan.job[i].rows, LAYOUT_RPW); chunks = std::max(chunks, (int)cdiv(plan.job[i].n, 16)); }
    hipLaunchKernelGGL((sws_k_layout_stream<4>), dim3(cdiv(chunks, 256), groups, L.n), dim3(256), 0, L.st, L.fs, p, plan);
}

// The interleaving pass behind a packed 4:2:2 destination of the scaler (dev_exec.hip: the planar writers filled a yuv422p working picture per
// frame; L.fs holds {src = its planes, dst = the packed picture}): yuvPlanartoyuy2_c / yuvPlanartouyvy_c with one chroma row per luma row.
void launch_layout_join422(const LaunchCtx &L, bool uyvy)
{
    const SwsDevParams &p = *L.p;
    using namespace swsk;
    LayoutPlan plan;
    std::memset(&plan, 0, sizeof(plan));
    LayoutJob &j = plan.job[0];
    plan.njobs = 1;
    j.op = LOP_P422_JOIN; j.rows = p.dstH; j.ys = 0; j.yd = 0; j.sa = j.sb = j.da = j.db = 0;
    j.n = 2 * (p.dstW >> 1); j.a0 = uyvy ? 1 : 0; j.a1 = 1; j.a2 = 0;
    hipLaunchKernelGGL((sws_k_layout_stream<4>), dim3(cdiv(cdiv(j.n, 8), 256), cdiv(j.rows, LAYOUT_RPW), L.n), dim3(256), 0, L.st, L.fs, p, plan);
}

// The luma plane of a PLAN_MAIN context whose horizontal and vertical luma filters are the identity (dev_prepare_on: mixed_ok): one tap of
// 1 << 14 through hScale8To15_c / hScale16To15_c, one tap through yuv2plane1_* -- per sample, so a streaming pass.  8 -> 8 bit is the copy
// ((s << 7) + 64) >> 7 == s), 8 -> N bit the plain left shift (the rounding term never carries).
int launch_layout_plane1(const LaunchCtx &L)
{
    SwsInternal *c = L.c; const SwsDevParams &p = *L.p; hipStream_t st = L.st; const SwsFrameSet &fs = L.fs;
    using namespace swsk;
    LayoutPlan plan;
    std::memset(&plan, 0, sizeof(plan));
    LayoutJob &j = plan.job[0];
    plan.njobs = 1;
    j.rows = p.dstH; j.ys = 0; j.yd = 0; j.sa = j.sb = j.da = j.db = 0;
    const bool s8 = c->srcBpc == 8, d8 = p.dst_bits == 8;
    if (s8 && d8 && p.range_active) {   // (dev_prepare_on admits a range conversion into the mixed plan for 8-bit planes only)
        j.op = LOP_P1_8TO8R; j.n = p.srcW; j.a0 = (int)(128u * (uint32_t)(uint16_t)p.lumCoeff); j.a1 = (int32_t)p.lumOffset; j.a2 = p.range_to_jpeg ? 32767 : 0x7fffffff;
    }
    else if (s8 && d8) { j.op = LOP_COPY; j.n = p.srcW; }
    else if (s8) { j.op = LOP_8TO16; j.n = p.srcW; j.a0 = p.dst_bits - 8; j.a1 = 32; j.a2 = p.dst_shift; }
    else { j.op = d8 ? LOP_P1_16TO8 : LOP_P1_16TO16; j.n = 2 * p.srcW; j.a0 = p.src_shift; j.a1 = p.hshift; j.a2 = p.dst_bits; j.a3 = p.dst_shift; j.a4 = p.should_dither ? 0 : 1; }
    const int unit = j.op == LOP_8TO16 ? 8 : 16;
    hipLaunchKernelGGL((sws_k_layout_stream<4>), dim3(cdiv(cdiv(j.n, unit), 256), cdiv(j.rows, LAYOUT_RPW), L.n), dim3(256), 0, st, fs, p, plan);
    return 0;
}

} // namespace swship
