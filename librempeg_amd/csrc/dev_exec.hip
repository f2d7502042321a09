// HIP side of libswscale_hip, part 3 of 4 -- EXECUTION: the launch set of a plan over device-resident frames (helper passes, sub-batches, byte order, XYZ),
// host-frame staging, cascades, the sharding of sws_scale_frames() over the GPUs, slices, sws_scale().
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


#include "dev_internal.hpp"
#include "generic_kinds.hpp"
#include "../../include/hwcontext_hip.h"

namespace swship {
void launch_layout_split422(const LaunchCtx &L, bool uyvy, bool vfirst);   // k_layout.hip: yuyv422 / uyvy422 / yvyu422 -> planar 4:2:2 working picture
void launch_layout_splitnv(const LaunchCtx &L, bool vfirst);   // k_layout.hip: plane 1 of a semi-planar 8-bit picture -> planar U / V working planes
void launch_layout_splitp01x(const LaunchCtx &L, int shift);   // k_layout.hip: p010-style planes -> planar working picture, words >> shift
void launch_alpha_merge32(const LaunchCtx &L);                 // k_stream.hip: the alpha bytes behind sws_k_strip_rgb (alpha_launch == 2)
void launch_layout_join422(const LaunchCtx &L, bool uyvy);   // k_layout.hip: planar 4:2:2 working picture -> yuyv422 / uyvy422 (yvyu422: planes swapped by the planner)
// ------------------------------------------------------------------------------------------
// frame layout helpers (hwcontext: frames_get_buffer analogue of libavutil/hwcontext_cuda.c:132-197)
// ------------------------------------------------------------------------------------------
int plane_geometry(int format, int w, int h, int plane, int *row_bytes, int *rows)
{
    const PixDesc *d = pix_desc(format);
    if (!d) return -1;
    const int np = pix_nb_planes(d);
    if (plane >= np) { *row_bytes = 0; *rows = 0; return 0; }
    if ((d->flags & PIXFLAG_PAL) && plane == 1) { *row_bytes = 1024; *rows = 1; return 0; }   // the palette: 256 words, one "row"
    // bytes per row = max over components in this plane of (step * samples); libavutil/imgutils.c av_image_get_linesize
    int step = 0; bool chroma = false;
    for (int c = 0; c < d->nb_components; c++)
        if (d->comp[c].plane == plane) { step = d->comp[c].step; chroma = (c == 1 || c == 2); }
    const bool sub = chroma && !(d->flags & PIXFLAG_RGB);
    const int sw = sub ? -((-w) >> d->log2_chroma_w) : w, sh = sub ? -((-h) >> d->log2_chroma_h) : h;
    *row_bytes = (format == AV_PIX_FMT_MONOWHITE || format == AV_PIX_FMT_MONOBLACK) ? (w + 7) >> 3 :
                 (format == AV_PIX_FMT_RGB4 || format == AV_PIX_FMT_BGR4) ? (4 * w + 7) >> 3 :
                 format == AV_PIX_FMT_UYYVYY411 ? 6 * ((w + 3) >> 2) : sw * step;   // bit streams: av_image_get_linesize, imgutils.c
    *rows = sh;
    return 0;
}

int rows_of_slice(int format, int plane, int sliceY, int sliceH, int *y0, int *rows)
{
    const PixDesc *d = pix_desc(format);
    bool chroma = false;
    if ((d->flags & PIXFLAG_PAL) && plane == 1) { *y0 = 0; *rows = 1; return 0; }   // every slice comes with the whole palette
    for (int c = 0; c < d->nb_components; c++) if (d->comp[c].plane == plane) chroma = (c == 1 || c == 2);
    const bool sub = chroma && !(d->flags & PIXFLAG_RGB);
    if (sub) { *y0 = sliceY >> d->log2_chroma_h; *rows = -((-sliceH) >> d->log2_chroma_h); }
    else { *y0 = sliceY; *rows = sliceH; }
    return 0;
}

void plane_extent(const PixDesc *d, int w, int h, int k, int *rows, int *row_bytes, int *vsub);

static bool frames_vec_ok(const SwsFramePtrs *fr, int n)
{
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 4; k++) {
            if (fr[i].src[k] && (((uintptr_t)fr[i].src[k] | (uintptr_t)(uint32_t)fr[i].srcStride[k]) & 15)) return false;
            if (fr[i].dst[k] && (((uintptr_t)fr[i].dst[k] | (uintptr_t)(uint32_t)fr[i].dstStride[k]) & 15)) return false;
        }
    return true;
}

// launch the kernels of one (non-cascaded) context over `n` device-resident frames (one sub-batch of launch_plan_le: rec0 / rec1 say whether
// this sub-batch starts / ends the timed region of the call)
static int launch_plan_le_batch(SwsInternal *c, DeviceState *d, const SwsFramePtrs *frames, int n, int sliceY, int sliceH, bool rec0, bool rec1)
{
    const SwsDevParams &p = d->params;
    hipStream_t st = d->stream;
    { int r_ = table_batch_end(c, d, st); if (r_ < 0) return r_; }   // (a launch set an error path left open)
    // SWS_SRC_V_CHR_DROP (swscale.c:333-334): "srcStride2[1] *= 1 << c->vChrDrop; srcStride2[2] *= 1 << c->vChrDrop" -- the scaler (not the
    // special converters) reads every 2^vChrDrop-th row of the chroma planes; packed sources reach the same rows through
    // `row << chrSrcVSub` in their readers
    std::vector<SwsFramePtrs> dropped;
    const int drop = (c->opts.flags & SWS_SRC_V_CHR_DROP_MASK) >> SWS_SRC_V_CHR_DROP_SHIFT;
    if (drop && c->plan == PLAN_MAIN) {
        dropped.assign(frames, frames + n);
        for (auto &f : dropped) { f.srcStride[1] *= 1 << drop; f.srcStride[2] *= 1 << drop; }
        frames = dropped.data();
    }
    // Palette-expanded sources (usePal, swscale_internal.h:937-950): scale_internal runs ff_update_palette before every conversion
    // (swscale.c:1088-1089).  Here every frame of the batch gets its own pair of 256-word tables in HBM - pal_yuv, then pal_rgb in the
    // destination's byte order - filled by sws_k_update_palette ahead of the converter on the same stream; the kernels find them through
    // src[1] of the frame, the caller's palette (pal8 only) moves to src[2].
    std::vector<SwsFramePtrs> palfr;
    if (p.srcKind == SRCK_PAL) {
        int r = grow(c, &d->d_pal, &d->d_pal_bytes, (size_t)n * 512 * sizeof(uint32_t));
        if (r < 0) return r;
        palfr.assign(frames, frames + n);
        for (int i = 0; i < n; i++) {
            palfr[i].src[2] = c->opts.src_format == AV_PIX_FMT_PAL8 ? palfr[i].src[1] : nullptr;
            palfr[i].src[1] = (const uint8_t *)((uint32_t *)d->d_pal + (size_t)i * 512);
            if (c->opts.src_format == AV_PIX_FMT_PAL8 && !palfr[i].src[2]) { log_msg(c, 0, "pal8 picture without a palette in data[1]\n"); return SWS_AVERROR(EINVAL); }
        }
        frames = palfr.data();
    }
    // frame tables of the helper passes around a packed 4:2:2 side (slot 0: the interleave behind the kernels, slot 1: the de-interleave ahead of
    // them; slot 2: the alpha launch of a full-chroma RGB destination; slots 3 / 4: staging copies in / out): spans of the frame-table ring (table_upload), cached per slot
    auto aux_table = [&](int slot, const std::vector<SwsFramePtrs> &v, const SwsFramePtrs **out) -> int {
        *out = table_upload(c, d, st, TAB_AUX0 + slot, v.data(), n);
        return *out ? 0 : d->ring.last_err;
    };
    bool timing_started = !rec0;
    // pictures whose planes are not 16-byte aligned (a cropped view, a tightly packed rgb24 row) or bottom-up (negative line sizes) under the helper passes, which read and write 16-byte
    // granules and have no per-byte twins: such planes are copied into aligned working planes first, and the written ones back afterwards (the visible
    // bytes of every row only).  Contexts without helper passes fall back to their per-sample kernels instead
    std::vector<SwsFramePtrs> stfr, st_in, st_out;
    int st_rb[2][4] = { { 0 } }, st_rows[2][4] = { { 0 } };
    bool stage_out = false;
    if (c->plan == PLAN_MAIN && (d->split_mode || d->join422 || d->fullchr_on || d->alpha_launch) && (!frames_vec_ok(frames, n) || !frames_desc_ok(frames, n, p.srcH, p.dstH))) {
        auto a256 = [](int64_t v) { return (v + 255) & ~(int64_t)255; };
        const PixDesc *ds = pix_desc(c->opts.src_format), *dd = pix_desc(c->opts.dst_format);
        int64_t off[2][4], fbytes = 0; int strd[2][4];
        for (int side = 0; side < 2; side++)
            for (int k = 0; k < 4; k++) {
                int vs = 0;
                plane_extent(side ? dd : ds, side ? c->opts.dst_w : c->opts.src_w, side ? c->opts.dst_h : c->opts.src_h, k, &st_rows[side][k], &st_rb[side][k], &vs);
                strd[side][k] = (int)a256(st_rb[side][k]); off[side][k] = fbytes; fbytes += (int64_t)strd[side][k] * st_rows[side][k];
            }
        int r = grow(c, &d->stage_img, &d->stage_bytes, (size_t)fbytes * (size_t)n);
        if (r < 0) return r;
        stfr.assign(frames, frames + n); st_in.resize((size_t)n); st_out.resize((size_t)n);
        bool stage_in = false;
        for (int i = 0; i < n; i++) {
            uint8_t *base = (uint8_t *)d->stage_img + (size_t)i * (size_t)fbytes;
            SwsFramePtrs &a = stfr[(size_t)i], &in = st_in[(size_t)i], &out = st_out[(size_t)i];
            std::memset(&in, 0, sizeof(in)); std::memset(&out, 0, sizeof(out));
            for (int k = 0; k < 4; k++) {
                if (a.src[k] && st_rows[0][k] && ((((uintptr_t)a.src[k] | (uintptr_t)(uint32_t)a.srcStride[k]) & 15) || a.srcStride[k] <= 0)) {
                    in.src[k] = a.src[k]; in.srcStride[k] = a.srcStride[k]; in.dst[k] = base + off[0][k]; in.dstStride[k] = strd[0][k];
                    a.src[k] = base + off[0][k]; a.srcStride[k] = strd[0][k]; stage_in = true;
                }
                if (a.dst[k] && st_rows[1][k] && ((((uintptr_t)a.dst[k] | (uintptr_t)(uint32_t)a.dstStride[k]) & 15) || a.dstStride[k] <= 0)) {
                    out.dst[k] = a.dst[k]; out.dstStride[k] = a.dstStride[k]; out.src[k] = base + off[1][k]; out.srcStride[k] = strd[1][k];
                    a.dst[k] = base + off[1][k]; a.dstStride[k] = strd[1][k]; stage_out = true;
                }
            }
        }
        if (!frames_vec_ok(stfr.data(), n)) { log_msg(c, 0, "internal error: staged pictures still unaligned\n"); return SWS_AVERROR(EINVAL); }
        // (what staging cannot cure: a plane of 2 GiB or more -- the strip kernels address a plane as base + 32-bit offset and the helper passes have no other kernels)
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 4; k++)
                if ((stfr[(size_t)i].src[k] && (int64_t)stfr[(size_t)i].srcStride[k] * p.srcH >= (int64_t)1 << 31) ||
                    (stfr[(size_t)i].dst[k] && (int64_t)stfr[(size_t)i].dstStride[k] * p.dstH >= (int64_t)1 << 31)) {
                    log_msg(c, 0, "planes of 2 GiB or more are not supported on this path\n"); return SWS_AVERROR(ENOTSUP);
                }
        if (stage_in) {
            LaunchCtx S;
            std::memset(&S.fs, 0, sizeof(S.fs));
            S.c = c; S.d = d; S.p = &p; S.st = st; S.frames = st_in.data(); S.n = n; S.sliceY = sliceY; S.sliceH = sliceH; S.vec = false;
            S.fs.count = n;
            if (n == 1) { S.fs.table = nullptr; S.fs.one = st_in[0]; }
            else { const SwsFramePtrs *t = nullptr; r = aux_table(3, st_in, &t); if (r < 0) return r; S.fs.table = t; }
            if (d->timing && !timing_started) { HIPCHK(hipEventRecord(d->ev0, st)); timing_started = true; }
            launch_stage_planes(S, st_rb[0], st_rows[0], true);
        }
        frames = stfr.data();
    }
    // packed 4:2:2 source through the planar kernels (dev_prepare_on): de-interleave into a planar 4:2:2 working picture per frame first
    std::vector<SwsFramePtrs> s422fr, s422split;
    // (a semi-planar source the strip-RGB kernel reads itself: no split pass on aligned frames)
    // (... or a packed 4:2:2 source the lockstep strip kernel reads itself: striprgb_direct == 3, k_striprgbsrc.hip)
    const bool direct422 = d->striprgb_direct == 3 && d->strip_ok && d->striprgbsrc_ok && !d->mixed_ok && !d->striprgb_ok && !d->fullchr_on && !d->alpha_launch && !d->rgbread_on && !c->tune.no_strip_rgbsrc;
    d->striprgb_direct_now = c->plan == PLAN_MAIN && d->split_mode && d->striprgb_direct && ((d->striprgb_ok && d->striprgb_direct != 3 && !c->tune.no_striprgb_direct) || direct422) &&
                             frames_vec_ok(frames, n) && frames_desc_ok(frames, n, p.srcH, p.dstH);
    if (c->plan == PLAN_MAIN && d->split_mode && !d->striprgb_direct_now) {
        auto a256 = [](int64_t v) { return (v + 255) & ~(int64_t)255; };
        const bool nv = (d->split_mode & 8) != 0;    // semi-planar 8-bit source: only the chroma plane is split, the luma plane stays where it is
        const bool p01x = (d->split_mode & 32) != 0; // semi-planar 10 / 12-bit source: both planes (every word is shifted down)
        const int sY = nv ? 0 : (int)a256(p01x ? 2 * p.srcW : p.srcW), sC = (int)a256(p01x ? 2 * p.chrSrcW : nv ? p.chrSrcW : p.srcW >> 1), crows = (nv || p01x) ? p.chrSrcH : p.srcH;
        const int64_t offU = (int64_t)sY * p.srcH, offV = offU + (int64_t)sC * crows, fbytes = a256(offV + (int64_t)sC * crows);
        int r = grow(c, &d->split_img, &d->split_bytes, (size_t)fbytes * (size_t)n);
        if (r < 0) return r;
        s422fr.assign(frames, frames + n);
        s422split.resize((size_t)n);
        for (int i = 0; i < n; i++) {
            uint8_t *base = (uint8_t *)d->split_img + (size_t)i * (size_t)fbytes;
            SwsFramePtrs &a = s422fr[(size_t)i], &j = s422split[(size_t)i];
            std::memset(&j, 0, sizeof(j));
            if (nv || p01x) { j.src[1] = a.src[1]; j.srcStride[1] = a.srcStride[1]; }
            if (!nv) { j.src[0] = a.src[0]; j.srcStride[0] = a.srcStride[0]; j.dst[0] = base; j.dstStride[0] = sY; a.src[0] = base; a.srcStride[0] = sY; }
            j.dst[1] = base + offU; j.dst[2] = base + offV; j.dstStride[1] = j.dstStride[2] = sC;
            a.src[1] = base + offU; a.src[2] = base + offV; a.src[3] = nullptr;
            a.srcStride[1] = a.srcStride[2] = sC; a.srcStride[3] = 0;
        }
        LaunchCtx S;
        std::memset(&S.fs, 0, sizeof(S.fs));
        S.c = c; S.d = d; S.p = &p; S.st = st; S.frames = s422split.data(); S.n = n; S.sliceY = sliceY; S.sliceH = sliceH; S.vec = true;
        S.fs.count = n;
        if (n == 1) { S.fs.table = nullptr; S.fs.one = s422split[0]; }
        else { const SwsFramePtrs *t = nullptr; r = aux_table(1, s422split, &t); if (r < 0) return r; S.fs.table = t; }
        if (d->timing && !timing_started) { HIPCHK(hipEventRecord(d->ev0, st)); timing_started = true; }
        if (p01x) launch_layout_splitp01x(S, d->split_shift);
        else if (nv) launch_layout_splitnv(S, (d->split_mode & 16) != 0);
        else launch_layout_split422(S, (d->split_mode & 3) == 2, (d->split_mode & 4) != 0);
        frames = s422fr.data();
    }
    // (scaled packed RGB -> packed RGB on aligned frames: one launch, no working pictures at all -- k_striprgb2rgb.hip)
    d->rgb2rgb_now = c->plan == PLAN_MAIN && d->rgb2rgb_ok && !c->tune.no_strip_rgb2rgb && (d->fullchr_on == 1 || d->fullchr_on == 2) && !d->fullchr_direct && d->rgbread_on && d->strip_ok &&
                     !d->mixed_ok && !d->striprgb_ok && !d->rgbsrc_ok && !d->rgb444_ok && frames_vec_ok(frames, n) && frames_desc_ok(frames, n, p.srcH, p.dstH);
    // full-chroma RGB destination (dev_prepare_on): the strip kernels write three int32 sum planes per frame, sws_k_fullchr_rgb follows
    std::vector<SwsFramePtrs> p422fr, p422join;
    uint8_t *sum_tab = nullptr;   // fullchr_on == 4: the epilogue's one-tap bank (behind the sum planes of the call)
    if (c->plan == PLAN_MAIN && d->fullchr_on && d->fullchr_direct) {   // the epilogue alone, on the caller's planes (Y, U, V, A order)
        const bool u1 = p.u_plane_src == 1;
        p422join.resize((size_t)n);
        for (int i = 0; i < n; i++) {
            const SwsFramePtrs &a = frames[i]; SwsFramePtrs &j = p422join[(size_t)i];
            std::memset(&j, 0, sizeof(j));
            for (int k = 0; k < 4; k++) { j.dst[k] = a.dst[k]; j.dstStride[k] = a.dstStride[k]; }
            j.src[0] = a.src[0]; j.srcStride[0] = a.srcStride[0];
            j.src[1] = a.src[u1 ? 1 : 2]; j.srcStride[1] = a.srcStride[u1 ? 1 : 2];
            j.src[2] = a.src[u1 ? 2 : 1]; j.srcStride[2] = a.srcStride[u1 ? 2 : 1];
            j.src[3] = a.src[3]; j.srcStride[3] = a.srcStride[3];
        }
    } else
    if (c->plan == PLAN_MAIN && d->fullchr_on && !d->rgb2rgb_now) {
        auto a256 = [](int64_t v) { return (v + 255) & ~(int64_t)255; };
        const int sP = (int)a256(4 * (int64_t)p.dstW);
        const int nraw = d->fullchr_on == 2 ? 4 : 3;     // (2: the alpha sums as a fourth plane)
        const int64_t plane = (int64_t)sP * p.dstH, fbytes = a256(nraw * plane);
        int r = grow(c, &d->join_img, &d->join_bytes, (size_t)fbytes * (size_t)n + (d->fullchr_on == 4 ? sum_writer_table_bytes(p.dstH) : 0));   // (4: the epilogue's one-tap bank behind the frames)
        if (r < 0) return r;
        sum_tab = (uint8_t *)d->join_img + (size_t)fbytes * (size_t)n;
        p422fr.assign(frames, frames + n);
        p422join.resize((size_t)n);
        for (int i = 0; i < n; i++) {
            uint8_t *base = (uint8_t *)d->join_img + (size_t)i * (size_t)fbytes;
            SwsFramePtrs &a = p422fr[(size_t)i], &j = p422join[(size_t)i];
            std::memset(&j, 0, sizeof(j));
            for (int k = 0; k < 4; k++) { j.dst[k] = a.dst[k]; j.dstStride[k] = a.dstStride[k]; }
            for (int k = 0; k < nraw; k++) { j.src[k] = base + k * plane; j.srcStride[k] = sP; }
            for (int k = 0; k < 3; k++) { a.dst[k] = base + k * plane; a.dstStride[k] = sP; }
            if (!p.dst_alpha_fill) { a.dst[3] = nullptr; a.dstStride[3] = 0; }   // (gbrap without source alpha: launch_fill_alpha writes the caller's plane 3)
        }
        frames = p422fr.data();
    }
    // packed 4:2:2 destination through planar writers (dev_prepare_on): the kernels write a planar 4:2:2 working picture per frame
    if (c->plan == PLAN_MAIN && d->join422) {
        auto a256 = [](int64_t v) { return (v + 255) & ~(int64_t)255; };
        const int sY = (int)a256(p.dstW), sC = (int)a256(p.dstW >> 1);
        const int64_t offU = (int64_t)sY * p.dstH, offV = offU + (int64_t)sC * p.dstH, fbytes = a256(offV + (int64_t)sC * p.dstH);
        int r = grow(c, &d->join_img, &d->join_bytes, (size_t)fbytes * (size_t)n);
        if (r < 0) return r;
        p422fr.assign(frames, frames + n);
        p422join.resize((size_t)n);
        for (int i = 0; i < n; i++) {
            uint8_t *base = (uint8_t *)d->join_img + (size_t)i * (size_t)fbytes;
            SwsFramePtrs &a = p422fr[(size_t)i], &j = p422join[(size_t)i];
            std::memset(&j, 0, sizeof(j));
            j.dst[0] = a.dst[0]; j.dstStride[0] = a.dstStride[0];
            j.src[0] = base; j.src[1] = base + offU; j.src[2] = base + offV; j.srcStride[0] = sY; j.srcStride[1] = j.srcStride[2] = sC;
            a.dst[0] = base; a.dst[1] = base + offU; a.dst[2] = base + offV; a.dst[3] = nullptr;
            a.dstStride[0] = sY; a.dstStride[1] = a.dstStride[2] = sC; a.dstStride[3] = 0;
        }
        frames = p422fr.data();
    }
    bool wide_fused = false;     // (the chroma launch of sws_k_strip_wide wrote the planar RGB destination itself: no epilogue)
    LaunchCtx L;
    std::memset(&L.fs, 0, sizeof(L.fs));
    SwsFrameSet &fs = L.fs;
    fs.count = n;
    if (n == 1) { fs.table = nullptr; fs.one = frames[0]; }
    else {
        fs.table = table_upload(c, d, st, TAB_MAIN, frames, n);
        if (!fs.table) return d->ring.last_err;
    }
    const bool vec = frames_vec_ok(frames, n);
    L.c = c; L.d = d; L.p = &p; L.st = st; L.frames = frames; L.n = n; L.sliceY = sliceY; L.sliceH = sliceH; L.vec = vec;
    if (d->timing && !timing_started) { HIPCHK(hipEventRecord(d->ev0, st)); }

    // bgr24ToYv12Wrapper (:2062-2077), yvu9ToYv12Wrapper (:2079-2093), yuyv/uyvyToYuv420Wrapper (:423-470) with a yuva420p
    // destination: fillPlane(dst[3], ..., src_w, srcSliceH, srcSliceY, 255)
    if (c->opts.dst_format == AV_PIX_FMT_YUVA420P && sliceH > 0 &&
        (c->plan == PLAN_UNSC_BGR24_YV12 || c->plan == PLAN_UNSC_YVU9_YV12 || c->plan == PLAN_UNSC_P4222PLANAR))
        launch_fill_alpha(L, p.srcW, sliceY, sliceH, 0);
    if (p.srcKind == SRCK_PAL) launch_update_palette(L);
    int ret = 0;
    switch (c->plan) {
    case PLAN_UNSC_YUV2RGB: ret = launch_yuv2rgb(L); break;
    case PLAN_UNSC_P01X:
    case PLAN_UNSC_8_P01X: ret = launch_p01x(L); break;
    case PLAN_MAIN: {
        if (p.dst_alpha_fill) launch_fill_alpha(L, p.dstW, 0, p.dstH, p.dstKind == DSTK_GBRPF32 ? 32 : p.dst_bits > 8 ? p.dst_bits : 0);   // swscale.c:536-552
        if (d->fullchr_on && d->fullchr_direct) break;     // (the epilogue below is the whole conversion)
        const bool rgb_lut = (p.dstKind == DSTK_RGB24 || p.dstKind == DSTK_RGB32) && !p.full_chr;
        if (d->vlines_on) ret = launch_generic(L);   // (virtual source lines: pass 1 of the two-pass path materialises them)
        else if (d->unity_h && rgb_lut && !p.no_chroma && !p.need_alpha && c->srcBpc == 8 && (p.srcKind == SRCK_PLANAR8 || p.srcKind == SRCK_NV12))
            ret = launch_rgb_unity(L);
        else if (vec && d->unity_h && d->unity_v && !p.no_chroma && !p.need_alpha && p.srcKind == SRCK_GBRPF32 && p.chrDstHSub == 0 && p.chrDstVSub == 0 && p.dst_shift == 0 &&
                 (p.dstKind == DSTK_PLANAR8 || p.dstKind == DSTK_PLANARN || p.dstKind == DSTK_PLANAR16))
            ret = launch_f32rgb(L);
        else if (d->rgbsrc_ok && vec && (!p.range_active || (d->rgbsrc2_rows && !c->tune.no_rgbsrc2 && !(p.dstW & 3) && frames_desc_ok(frames, n, p.srcH, p.dstH)))) { if (!launch_rgbsrc2(L)) ret = launch_rgbsrc(L); }   // (range conversion: the wave-march form only)                                             // packed RGB source, same size
        else if (d->rgb444_ok && vec) ret = launch_rgb444(L);                                             // 8-bit RGB -> planar 4:4:4, same size
        else if (d->striprgb_ok && vec && frames_desc_ok(frames, n, p.srcH, p.dstH)) ret = launch_striprgb(L);   // marching strip kernel, RGB epilogue
        else if (d->mixed_ok && vec && frames_desc_ok(frames, n, p.srcH, p.dstH)) {                        // identity luma: streaming pass + strip kernel on chroma
            bool fusedj = false;
            if (d->join422 && !p422join.empty() && !d->fullchr_on && !isGray(c->opts.src_format)) {
                // ... into packed 4:2:2 with identity horizontal filters (yuv420p / nv12 -> yuyv422 / uyvy422 at the same size): one pass from the caller's planes into the
                // caller's picture instead of plane pass + chroma strip launch + interleave (k_stream.hip; it refuses what it does not take)
                std::vector<SwsFramePtrs> fj(frames, frames + n);
                for (int i = 0; i < n; i++) {
                    SwsFramePtrs &a = fj[(size_t)i];
                    a.dst[0] = p422join[(size_t)i].dst[0]; a.dstStride[0] = p422join[(size_t)i].dstStride[0];
                    a.dst[1] = a.dst[2] = a.dst[3] = nullptr; a.dstStride[1] = a.dstStride[2] = a.dstStride[3] = 0;
                }
                LaunchCtx F = L;
                std::memset(&F.fs, 0, sizeof(F.fs));
                F.fs.count = n; F.frames = fj.data();
                if (n == 1) { F.fs.table = nullptr; F.fs.one = fj[0]; }
                else { const SwsFramePtrs *t = nullptr; int r = aux_table(0, fj, &t); if (r < 0) return r; F.fs.table = t; }
                if (launch_mixed_join422(F, d->join422 == 2)) { fusedj = true; wide_fused = true; }
            }
            if (!fusedj) ret = launch_mixed(L);
        }
        else if (d->strip_ok && vec && frames_desc_ok(frames, n, p.srcH, p.dstH)) {   // marching strip kernel
            if (d->rgb2rgb_now) {
                if (!launch_strip_rgb2rgb(L)) { log_msg(c, 0, "internal error: no one-launch RGB -> RGB strip kernel for a plan that counted on it\n"); return SWS_AVERROR(EINVAL); }
            } else if (d->rgbread_on) ret = launch_rgbread_strip(L);
            else if (d->striprgb_direct_now && d->striprgb_direct == 3) {
                if (!launch_strip_rgbsrc(L)) { log_msg(c, 0, "internal error: no lockstep strip kernel for a packed 4:2:2 source whose split pass was skipped\n"); return SWS_AVERROR(EINVAL); }
            } else if (d->fullchr_on == 4 && (d->fullchr_kind == DSTK_GBRP16 || d->fullchr_kind == DSTK_GBRPF32) && !c->tune.no_wide_epilogue && !p422join.empty() && !p.no_chroma &&
                       frames_desc_ok(p422join.data(), n, 1, p.dstH)) {
                // planar RGB of 16 bits / float32 behind the 19-bit strip kernel, fused form: the luma launch leaves its sums in the working plane, the chroma launch reads
                // them next to its own U / V sums and writes the destination planes (kernels_stripwide.hpp: the U / V sums and the epilogue's pass over all three are gone)
                ret = launch_strip_wide(L, 1);
                if (ret < 0) return ret;
                std::vector<SwsFramePtrs> ffr(frames, frames + n);
                for (int i = 0; i < n; i++) {
                    SwsFramePtrs &a = ffr[(size_t)i];
                    a.src[3] = frames[i].dst[0]; a.srcStride[3] = frames[i].dstStride[0];
                    for (int k = 0; k < 3; k++) { a.dst[k] = p422join[(size_t)i].dst[k]; a.dstStride[k] = p422join[(size_t)i].dstStride[k]; }
                }
                SwsDevParams pF = p;
                pF.dstKind = d->fullchr_kind;
                LaunchCtx F = L;
                std::memset(&F.fs, 0, sizeof(F.fs));
                F.fs.count = n; F.frames = ffr.data(); F.p = &pF;
                if (n == 1) { F.fs.table = nullptr; F.fs.one = ffr[0]; }
                else { const SwsFramePtrs *t = nullptr; int r = aux_table(2, ffr, &t); if (r < 0) return r; F.fs.table = t; }
                ret = launch_strip_wide(F, 2);
                if (ret < 0) return ret;
                wide_fused = true;
            } else ret = launch_strip(L);
            if (ret >= 0 && p.no_chroma && isGray(c->opts.src_format) && !isGray(c->opts.dst_format) && (p.dstKind != DSTK_RAW32 || d->fullchr_on == 3 || d->fullchr_on == 1) && !fullchr_gray_const(L)) launch_gray_chroma(L);   // (a gray source: the luma launch alone ran)
        }
        else if (d->dot2_ok && vec) ret = launch_tile_dot2(L);                                              // dot2 LDS-tile kernel
        else if (d->tile_ok) ret = launch_tile(L);                                                          // fused h+v LDS-tile kernel
        else ret = launch_generic(L);                                                                       // optional pass 1 into scratch, then writers
        break;
    }
    case PLAN_NONE:
    case PLAN_CASCADE:
        log_msg(c, 0, "internal error: no execution plan\n");
        return SWS_AVERROR(EINVAL);
    default: ret = launch_misc(L); break;
    }
    if (ret < 0) return ret;
    // full-chroma RGB destination with a scaled alpha plane: the A samples -- the reader pre-pass's fourth plane for a packed 32 bpp source
    // (rgbaToA_c, input.c: a << 6 | a >> 2), plane 3 of a planar source -- go through the luma filters into the fourth sum plane
    // (swscale.c:440-470 scales alpPixBuf with the luma banks; yuv2rgb_full_X: (sum + (1 << 18)) >> 19, output.c:2027-2038)
    std::vector<SwsFramePtrs> alfr;
    SwsDevParams pA;
    const bool alpha_run = c->plan == PLAN_MAIN && d->alpha_launch == 1 && d->strip_ok && !d->fullchr_on && vec && frames_desc_ok(frames, n, p.srcH, p.dstH);   // (the strip launch above ran)
    const bool alpha_rgb = c->plan == PLAN_MAIN && d->alpha_launch == 2 && d->striprgb_ok && !d->fullchr_on && vec && frames_desc_ok(frames, n, p.srcH, p.dstH);
    if (alpha_rgb) {   // the LUT writers' strip kernel stored opaque pixels: the A sums into a working plane, then the alpha bytes into the picture
        auto a256 = [](int64_t v) { return (v + 255) & ~(int64_t)255; };
        const int sP = (int)a256(4 * (int64_t)p.dstW);
        const int64_t fbytes = a256((int64_t)sP * p.dstH);
        int r = grow(c, &d->join_img, &d->join_bytes, (size_t)fbytes * (size_t)n);
        if (r < 0) return r;
        std::vector<SwsFramePtrs> afr(frames, frames + n), mfr((size_t)n);
        SwsDevParams pR = p;
        pR.dstKind = DSTK_RAW32; pR.range_active = 0;   // (the alpha line is h-scaled by the luma function but never range converted: hscale.c:61-63 vs :66-79)
        for (int i = 0; i < n; i++) {
            uint8_t *base = (uint8_t *)d->join_img + (size_t)i * (size_t)fbytes;
            SwsFramePtrs &a = afr[(size_t)i], &m = mfr[(size_t)i];
            std::memset(&m, 0, sizeof(m));
            a.src[0] = frames[i].src[3]; a.srcStride[0] = frames[i].srcStride[3];
            a.src[1] = a.src[2] = a.src[3] = nullptr; a.srcStride[1] = a.srcStride[2] = a.srcStride[3] = 0;
            a.dst[0] = base; a.dstStride[0] = sP;
            a.dst[1] = a.dst[2] = a.dst[3] = nullptr; a.dstStride[1] = a.dstStride[2] = a.dstStride[3] = 0;
            m.src[0] = base; m.srcStride[0] = sP; m.dst[0] = frames[i].dst[0]; m.dstStride[0] = frames[i].dstStride[0];
        }
        LaunchCtx A = L;
        std::memset(&A.fs, 0, sizeof(A.fs));
        A.fs.count = n; A.frames = afr.data(); A.p = &pR;
        if (n == 1) { A.fs.table = nullptr; A.fs.one = afr[0]; }
        else { const SwsFramePtrs *t = nullptr; r = aux_table(2, afr, &t); if (r < 0) return r; A.fs.table = t; }
        ret = launch_strip_luma(A);
        if (ret < 0) return ret;
        LaunchCtx M = L;
        std::memset(&M.fs, 0, sizeof(M.fs));
        M.fs.count = n; M.frames = mfr.data();
        if (n == 1) { M.fs.table = nullptr; M.fs.one = mfr[0]; }
        else { const SwsFramePtrs *t = nullptr; r = aux_table(0, mfr, &t); if (r < 0) return r; M.fs.table = t; }
        launch_alpha_merge32(M);
    }
    if ((c->plan == PLAN_MAIN && d->fullchr_on == 2 && !d->fullchr_direct && !d->rgb2rgb_now) || alpha_run) {
        if ((!alpha_run && p422join.empty()) || !d->strip_ok) { log_msg(c, 0, "internal error: alpha launch without the strip plan\n"); return SWS_AVERROR(EINVAL); }
        alfr.assign(frames, frames + n);
        pA = p; pA.range_active = 0;   // (no range conversion on the alpha line)
        const bool rd = d->rgbread_on;
        if (rd) { if (d->rgbread_offA < 0) { log_msg(c, 0, "internal error: reader pre-pass without an alpha plane\n"); return SWS_AVERROR(EINVAL); }
                  pA.srcKind = SRCK_PLANAR16; pA.src_shift = 0; }
        for (int i = 0; i < n; i++) {
            SwsFramePtrs &a = alfr[(size_t)i];
            if (rd) { a.src[0] = (const uint8_t *)d->rgbread_img + (size_t)i * (size_t)d->rgbread_frame_bytes + d->rgbread_offA; a.srcStride[0] = d->rgbread_strideY; }
            else { a.src[0] = frames[i].src[3]; a.srcStride[0] = frames[i].srcStride[3]; }
            a.src[1] = a.src[2] = a.src[3] = nullptr; a.srcStride[1] = a.srcStride[2] = a.srcStride[3] = 0;
            if (alpha_run) { a.dst[0] = frames[i].dst[3]; a.dstStride[0] = frames[i].dstStride[3]; }
            else { a.dst[0] = const_cast<uint8_t *>(p422join[(size_t)i].src[3]); a.dstStride[0] = p422join[(size_t)i].srcStride[3]; }
            a.dst[1] = a.dst[2] = a.dst[3] = nullptr; a.dstStride[1] = a.dstStride[2] = a.dstStride[3] = 0;
        }
        LaunchCtx A = L;
        std::memset(&A.fs, 0, sizeof(A.fs));
        A.fs.count = n; A.frames = alfr.data(); A.p = &pA;
        if (n == 1) { A.fs.table = nullptr; A.fs.one = alfr[0]; }
        else { const SwsFramePtrs *t = nullptr; int r = aux_table(2, alfr, &t); if (r < 0) return r; A.fs.table = t; }
        ret = launch_strip_luma(A);
        if (ret < 0) return ret;
    }
    if (!p422join.empty() && !wide_fused) {   // interleave the planar 4:2:2 working pictures into the packed destinations
        LaunchCtx J = L;
        std::memset(&J.fs, 0, sizeof(J.fs));
        J.fs.count = n;
        J.frames = p422join.data();
        if (n == 1) { J.fs.table = nullptr; J.fs.one = p422join[0]; }
        else { const SwsFramePtrs *t = nullptr; int r = aux_table(0, p422join, &t); if (r < 0) return r; J.fs.table = t; }
        if (d->fullchr_on == 4 && (d->fullchr_kind == DSTK_GBRP16 || d->fullchr_kind == DSTK_GBRPF32) && c->tune.no_wide_epilogue != 1) launch_fullchr_rgb(J);   // (sws_k_fullchr_gbrp16: the vector form of the writer)
        else if (d->fullchr_on == 4) { int r = launch_sum_writer(J, d->fullchr_kind, sum_tab); if (r < 0) return r; }
        else if (d->fullchr_on) launch_fullchr_rgb(J);
        else launch_layout_join422(J, d->join422 == 2);
    }
    if (stage_out) {   // the staged destination planes back into the caller's picture
        LaunchCtx S = L;
        std::memset(&S.fs, 0, sizeof(S.fs));
        S.fs.count = n; S.frames = st_out.data(); S.vec = false;
        if (n == 1) { S.fs.table = nullptr; S.fs.one = st_out[0]; }
        else { const SwsFramePtrs *t = nullptr; int r = aux_table(4, st_out, &t); if (r < 0) return r; S.fs.table = t; }
        launch_stage_planes(S, st_rb[1], st_rows[1], false);
    }
    HIPCHK(hipGetLastError());
    if (d->timing && rec1) { HIPCHK(hipEventRecord(d->ev1, st)); d->timed = true; }
    { int r_ = table_batch_end(c, d, st); if (r_ < 0) return r_; }
    return guards_check(c, st);
}

// The helper passes keep per-FRAME working pictures (the reader pre-pass's 16-bit planes, the split / join pictures, the int32 sum planes of the
// full-chroma routes, staging copies of unaligned frames): bytes per frame x the frames of the call.  A large sws_scale_frames() batch is
// therefore cut into sub-batches whose working pictures fit a budget (Tuning::work_mb, 2 GiB by default: bgra 4K -> rgb24 1080p needs ~83 MB per
// frame, i.e. 24 frames per sub-batch -- far more than it takes to fill the GPU); the buffers are reused from sub-batch to sub-batch (same stream:
// ordered).  Contexts without helper passes have no per-frame working memory and always go out as one launch set.
static size_t helper_bytes_per_frame(const SwsInternal *c, const DeviceState *d, const SwsFramePtrs *frames, int n)
{
    if (c->plan != PLAN_MAIN) return 0;
    const SwsDevParams &p = d->params;
    // the one-launch forms of round 4 read the caller's aligned frames themselves and keep no working picture (the conditions of launch_plan_le_batch)
    if (frames_vec_ok(frames, n) && frames_desc_ok(frames, n, p.srcH, p.dstH)) {
        if (d->rgb2rgb_ok && !c->tune.no_strip_rgb2rgb && (d->fullchr_on == 1 || d->fullchr_on == 2) && !d->fullchr_direct && d->rgbread_on && d->strip_ok &&
            !d->mixed_ok && !d->striprgb_ok && !d->rgbsrc_ok && !d->rgb444_ok) return 0;
        if (d->rgbread_on && d->strip_ok && d->striprgbsrc_ok && !c->tune.no_strip_rgbsrc && !d->fullchr_on && !d->alpha_launch && !d->split_mode && !d->join422 &&
            !d->mixed_ok && !d->striprgb_ok && !d->rgbsrc_ok && !d->rgb444_ok) return 0;
        if (d->split_mode && d->striprgb_direct == 3 && d->strip_ok && d->striprgbsrc_ok && !d->mixed_ok && !d->striprgb_ok && !d->fullchr_on && !d->alpha_launch && !d->rgbread_on &&
            !d->join422 && !c->tune.no_strip_rgbsrc) return 0;
        if (d->split_mode && (d->striprgb_direct == 1 || d->striprgb_direct == 2) && d->striprgb_ok && !c->tune.no_striprgb_direct && !d->fullchr_on && !d->alpha_launch && !d->join422) return 0;
    }
    auto a256 = [](int64_t v) { return (v + 255) & ~(int64_t)255; };
    int64_t b = 0;
    const bool helpers = d->split_mode || d->join422 || d->fullchr_on || d->alpha_launch || d->rgbread_on;
    if (!helpers) return 0;
    // staging copies of unaligned / bottom-up planes (launch_plan_le_batch stages under the same condition; worst case: every plane of both pictures)
    if ((d->split_mode || d->join422 || d->fullchr_on || d->alpha_launch) && (!frames_vec_ok(frames, n) || !frames_desc_ok(frames, n, p.srcH, p.dstH))) {
        const PixDesc *ds = pix_desc(c->opts.src_format), *dd = pix_desc(c->opts.dst_format);
        for (int side = 0; side < 2; side++)
            for (int k = 0; k < 4; k++) {
                int rows = 0, rb = 0, vs = 0;
                plane_extent(side ? dd : ds, side ? c->opts.dst_w : c->opts.src_w, side ? c->opts.dst_h : c->opts.src_h, k, &rows, &rb, &vs);
                b += a256(rb) * rows;
            }
    }
    if (d->split_mode) b += a256(2 * (int64_t)p.srcW) * p.srcH + 2 * a256(2 * (int64_t)std::max(p.chrSrcW, p.srcW >> 1)) * p.srcH + 256;
    if (d->fullchr_on && !d->fullchr_direct) b += 4 * a256(4 * (int64_t)p.dstW) * p.dstH + 256;
    if (d->join422) b += (a256(p.dstW) + 2 * a256(p.dstW >> 1)) * (int64_t)p.dstH + 256;
    if (d->alpha_launch == 2) b += a256(4 * (int64_t)p.dstW) * p.dstH + 256;
    if (d->rgbread_on && !(d->striprgbsrc_ok && !c->tune.no_strip_rgbsrc)) b += 2 * a256(2 * (int64_t)p.srcW) * p.srcH + 2 * a256(2 * (int64_t)p.chrSrcW) * p.srcH + 512;
    return (size_t)b;
}

static int launch_plan_le(SwsInternal *c, DeviceState *d, const SwsFramePtrs *frames, int n, int sliceY, int sliceH)
{
    const size_t per = n > 1 ? helper_bytes_per_frame(c, d, frames, n) : 0;
    const size_t budget = (size_t)std::max(1, c->tune.work_mb) << 20;
    if (!per || per * (size_t)n <= budget) return launch_plan_le_batch(c, d, frames, n, sliceY, sliceH, true, true);
    const int chunk = (int)std::max<size_t>(1, budget / per);
    for (int i = 0; i < n; i += chunk) {
        const int m = std::min(chunk, n - i);
        int r = launch_plan_le_batch(c, d, frames + i, m, sliceY, sliceH, i == 0, i + m >= n);
        if (r < 0) return r;
    }
    return 0;
}

int image_layout(int format, int w, int h, int align, int linesize[4], size_t offset[4], size_t *total);

// rows and visible bytes per row of plane k of a picture
void plane_extent(const PixDesc *d, int w, int h, int k, int *rows, int *row_bytes, int *vsub)
{
    int maxb = 0; bool chroma = false, used = false;
    for (int i = 0; i < d->nb_components; i++) {
        if (d->comp[i].plane != k) continue;
        used = true;
        if ((i == 1 || i == 2) && !(d->flags & PIXFLAG_RGB)) chroma = true;
    }
    *rows = 0; *row_bytes = 0; *vsub = 0;
    if (!used) return;
    const int pw = chroma ? -((-w) >> d->log2_chroma_w) : w;
    for (int i = 0; i < d->nb_components; i++)
        if (d->comp[i].plane == k) maxb = std::max(maxb, d->comp[i].step * pw);
    *vsub = chroma ? d->log2_chroma_h : 0;
    *rows = chroma ? -((-h) >> d->log2_chroma_h) : h;
    *row_bytes = maxb;
}

// sws_scale's XYZ stages (swscale.c:1126-1139, :1194-1210): an xyz12 source slice is converted into an rgb48 scratch picture first
// (xyz12Torgb48_c :745-802), the written rows of an xyz12 destination are converted in place afterwards (rgb48Toxyz12_c :804-861);
// both are skipped for xyz12 -> xyz12 at equal sizes.  Gamma LUTs as init_xyz_tables (utils.c:709-733) builds them, from libm pow().
static int launch_plan_xyz(SwsInternal *c, DeviceState *d, const SwsFramePtrs *frames, int n, int sliceY, int sliceH)
{
    const SwsContext &o = c->opts;
    if ((!c->srcXYZ && !c->dstXYZ) || (c->srcXYZ && c->dstXYZ && o.src_w == o.dst_w && o.src_h == o.dst_h))
        return launch_plan_le(c, d, frames, n, sliceY, sliceH);
    hipStream_t st = d->stream;
    if (!d->d_xyz_tab) {
        static std::vector<uint16_t> tab;   // xyzgamma[4096], rgbgammainv[4096], rgbgamma[65536], xyzgammainv[65536]
        static std::once_flag once;
        std::call_once(once, [] {
            tab.resize(2 * 4096 + 2 * 65536);
            for (int i = 0; i < 4096; i++) {
                tab[i] = (uint16_t)lrint(pow(i / 4095.0, 2.6) * 65535.0);
                tab[4096 + i] = (uint16_t)lrint(pow(i / 4095.0, 2.2) * 65535.0);
            }
            for (int i = 0; i < 65536; i++) {
                tab[8192 + i] = (uint16_t)lrint(pow(i / 65535.0, 1.0 / 2.2) * 4095.0);
                tab[8192 + 65536 + i] = (uint16_t)lrint(pow(i / 65535.0, 1.0 / 2.6) * 4095.0);
            }
        });
        HIPCHK(hipMalloc(&d->d_xyz_tab, tab.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpyAsync(d->d_xyz_tab, tab.data(), tab.size() * sizeof(uint16_t), hipMemcpyHostToDevice, st));   // (static storage: stays valid)
    }
    const uint16_t *T = (const uint16_t *)d->d_xyz_tab;
    std::vector<SwsFramePtrs> fr(frames, frames + n);
    const dim3 blk(256);
    if (c->srcXYZ && sliceH > 0) {
        const int ls = (o.src_w * 6 + 255) & ~255;
        const size_t total = (size_t)ls * o.src_h;
        if ((size_t)n * total > d->xyz_bytes) {
            HIPCHK(hipStreamSynchronize(st));
            if (d->d_xyz) HIPCHK(hipFree(d->d_xyz));
            d->d_xyz = nullptr;
            HIPCHK(hipMalloc(&d->d_xyz, (size_t)n * total));
            { int pr = poison(c, d->d_xyz, (size_t)n * total); if (pr < 0) return pr; }
            d->xyz_bytes = (size_t)n * total;
        }
        for (int i = 0; i < n; i++) {
            uint8_t *scr = (uint8_t *)d->d_xyz + (size_t)i * total;
            const int y1 = std::min(o.src_h, sliceY + sliceH);
            if (y1 > sliceY)
                launch_xyz12(st, frames[i].src[0] + (int64_t)sliceY * frames[i].srcStride[0], (int64_t)frames[i].srcStride[0],
                             scr + (int64_t)sliceY * ls, (int64_t)ls, o.src_w, y1 - sliceY, T, T + 8192, 1);
            fr[i].src[0] = scr; fr[i].srcStride[0] = ls;
        }
    }
    int ret = launch_plan_le(c, d, fr.data(), n, sliceY, sliceH);
    if (ret >= 0 && c->dstXYZ) {
        const bool whole = c->plan == PLAN_MAIN || c->plan == PLAN_CASCADE || (sliceY == 0 && sliceH == o.src_h);
        const int y0 = whole ? 0 : sliceY, y1 = whole ? o.dst_h : std::min(o.dst_h, sliceY + sliceH);
        for (int i = 0; i < n && y1 > y0; i++) {
            uint8_t *p0 = frames[i].dst[0] + (int64_t)y0 * frames[i].dstStride[0];
            launch_xyz12(st, p0, (int64_t)frames[i].dstStride[0], p0, (int64_t)frames[i].dstStride[0], o.dst_w, y1 - y0, T + 4096, T + 8192 + 65536, 0);
        }
    }
    return ret;
}

int launch_plan(SwsInternal *c, DeviceState *d, const SwsFramePtrs *frames, int n, int sliceY, int sliceH)
{
    { int v_ = verify_tables_once(c, d); if (v_ < 0) return v_; }       // (first launch of a plan: its table blocks are read back once, dev_state.hip)
    if (!c->srcBE && !c->dstBE) return launch_plan_xyz(c, d, frames, n, sliceY, sliceH);
    hipStream_t st = d->stream;
    const SwsContext &o = c->opts;
    std::vector<SwsFramePtrs> fr(frames, frames + n);
    const dim3 blk(256);
    if (c->srcBE) {
        const PixDesc *ds = pix_desc(o.src_format);
        const int unit = ds->comp[0].depth == 32 ? 4 : 2;
        int ls[4]; size_t offs[4], total = 0;
        int r = image_layout(o.src_format, o.src_w, o.src_h, 256, ls, offs, &total);
        if (r < 0) return r;
        if ((size_t)n * total > d->be_bytes) {
            HIPCHK(hipStreamSynchronize(st));
            if (d->d_be) HIPCHK(hipFree(d->d_be));
            d->d_be = nullptr;
            HIPCHK(hipMalloc(&d->d_be, (size_t)n * total));
            { int pr = poison(c, d->d_be, (size_t)n * total); if (pr < 0) return pr; }
            d->be_bytes = (size_t)n * total;
        }
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 4; k++) {
                int rows, rb, vs;
                plane_extent(ds, o.src_w, o.src_h, k, &rows, &rb, &vs);
                if (!rows || !frames[i].src[k]) continue;
                const int y0 = sliceY >> vs, y1 = std::min(rows, -((-(sliceY + sliceH)) >> vs));
                uint8_t *scr = (uint8_t *)d->d_be + (size_t)i * total + offs[k];
                if (y1 > y0)
                    launch_bswap(st, frames[i].src[k] + (int64_t)y0 * frames[i].srcStride[k], (int64_t)frames[i].srcStride[k],
                                 scr + (int64_t)y0 * ls[k], (int64_t)ls[k], y1 - y0, rb, unit);
                fr[i].src[k] = scr; fr[i].srcStride[k] = ls[k];
            }
    }
    int ret = launch_plan_xyz(c, d, fr.data(), n, sliceY, sliceH);
    if (ret >= 0 && c->dstBE) {
        const PixDesc *dd = pix_desc(o.dst_format);
        const int unit = dd->comp[0].depth == 32 ? 4 : 2;
        const bool whole = c->plan == PLAN_MAIN || c->plan == PLAN_CASCADE || (sliceY == 0 && sliceH == o.src_h);
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 4; k++) {
                int rows, rb, vs;
                plane_extent(dd, o.dst_w, o.dst_h, k, &rows, &rb, &vs);
                if (!rows || !frames[i].dst[k]) continue;
                const int y0 = whole ? 0 : sliceY >> vs, y1 = whole ? rows : std::min(rows, -((-(sliceY + sliceH)) >> vs));
                if (y1 <= y0) continue;
                uint8_t *p0 = frames[i].dst[k] + (int64_t)y0 * frames[i].dstStride[k];
                launch_bswap(st, p0, (int64_t)frames[i].dstStride[k], p0, (int64_t)frames[i].dstStride[k], y1 - y0, rb, unit);
            }
    }
    return ret;
}


// build device-side frame descriptors for host or device user pointers; stages host memory
struct Staging {
    bool src_host = false, dst_host = false;
};

int image_layout(int format, int w, int h, int align, int linesize[4], size_t offset[4], size_t *total)
{
    const PixDesc *d = pix_desc(format);
    if (!d) return SWS_AVERROR(EINVAL);
    size_t off = 0;
    for (int pl = 0; pl < 4; pl++) {
        int rb = 0, rows = 0;
        plane_geometry(format, w, h, pl, &rb, &rows);
        linesize[pl] = rb ? (rb + align - 1) / align * align : 0;
        offset[pl] = off;
        off += (size_t)linesize[pl] * rows;
        off = (off + 255) & ~(size_t)255;
    }
    // 2 KiB of readable tail: the marching kernels fetch a row in 16-byte chunks through a whole-plane buffer descriptor with the row offset in the SGPR operand,
    // which the descriptor's range check does not cover -- in the last rows of a plane the chunks of lanes beyond the picture (never stored) reach behind its end:
    // by up to one 1 024-byte row segment in sws_k_rgb_march (768 bytes seen on the x86 emulation of the kernels, profiles/r06_emulation.md), 16 bytes in the strip
    // kernels.  INTEGRATION.md 3 states the contract for callers' own frames
    *total = off + 2048;
    return 0;
}

// casc_flip0: a bottom-up slice sequence through scale_cascaded: only the first context sees the flipped picture (it writes the intermediate one
// upside down, i.e. upright again), the second one runs top-down on it
// (run_single: declared in dev_internal.hpp)

} // namespace swship


// The partition rule of sws_scale_frames(), as a pure function so that it can be tested without GPUs:
// a frame that lives in HBM is converted on the GPU that holds it (the two sides of a frame must not live on different GPUs: -1);
// frames in host memory are dealt round-robin over the first `nb_devices` GPUs starting at the context's home GPU.
extern "C" int sws_hip_plan_shards(int nb_frames, const int *src_device, const int *dst_device, int nb_devices, int home, int *out_device)
{
    if (nb_frames < 0 || nb_devices <= 0 || !out_device) return SWS_AVERROR(EINVAL);
    if (home < 0 || home >= nb_devices) home = 0;
    int rr = 0;
    for (int i = 0; i < nb_frames; i++) {
        const int sd = src_device ? src_device[i] : -1, dd = dst_device ? dst_device[i] : -1;
        if (sd >= 0 && dd >= 0 && sd != dd) return SWS_AVERROR(EINVAL);
        if (sd >= 0 || dd >= 0) out_device[i] = sd >= 0 ? sd : dd;
        else out_device[i] = (home + rr++) % nb_devices;
    }
    return 0;
}

namespace swship {

int dev_run(SwsInternal *c, const uint8_t *const src[4], const int srcStride[4], int srcSliceY, int srcSliceH,
            uint8_t *const dst[4], const int dstStride[4], int nb_frames,
            const SwsFrameView *const *srcFrames, SwsFrameView *const *dstFrames)
{
    int ret = ensure_dev(c);
    if (ret < 0) return ret;
    if (c->dev->dry) { log_msg(c, 0, "a dry_plan context only plans: it cannot convert\n"); return SWS_AVERROR(ENOSYS); }
    DeviceGuard guard;
    if (nb_frames <= 0) {   // sws_scale(): the home GPU, or the GPU the caller's device buffers live on
        DeviceState *d = c->dev;
        const int sd = ptr_device(src[0]), dd = ptr_device(dst[0]);
        if (sd >= 0 && dd >= 0 && sd != dd) { log_msg(c, 0, "source and destination live on different GPUs\n"); return SWS_AVERROR(EINVAL); }
        const int dev = sd >= 0 ? sd : dd;
        if (dev >= 0 && dev != d->device) d = dev_state_for(c, dev);
        if (!d) return AVERROR_EXTERNAL_;
        ret = dev_prepare_on(c, d);
        if (ret < 0) return ret;
        HIPCHK(hipSetDevice(d->device));
        return run_single(c, d, src, srcStride, srcSliceY, srcSliceH, dst, dstStride);
    }

    // ---- sws_scale_frames(): shard the independent frames over the visible GPUs ----
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return AVERROR_EXTERNAL_; }
    if (c->tune.max_devices > 0) ndev = std::max(std::min(ndev, c->tune.max_devices), c->dev->device + 1);
    std::vector<int> sdev(nb_frames), ddev(nb_frames), owner(nb_frames);
    for (int i = 0; i < nb_frames; i++) { sdev[i] = ptr_device(srcFrames[i]->data[0]); ddev[i] = ptr_device(dstFrames[i]->data[0]); }
    ret = sws_hip_plan_shards(nb_frames, sdev.data(), ddev.data(), ndev, c->dev->device, owner.data());
    // an error-diffusion context carries its error line from frame to frame, per GPU: host frames all go to the home GPU, in order
    if (ret >= 0 && c->cascade_ed) for (int i = 0; i < nb_frames; i++) if (sdev[i] < 0 && ddev[i] < 0) owner[i] = c->dev->device;
    if (ret < 0) { log_msg(c, 0, "sws_scale_frames(): a frame's source and destination live on different GPUs\n"); return ret; }

    const int nps = pix_nb_planes(pix_desc(c->opts.src_format)), npd = pix_nb_planes(pix_desc(c->opts.dst_format));
    // per GPU: the HBM-resident frames go out as ONE launch set on that GPU's stream; launches are issued on every GPU before
    // anything is waited for.  Frames with a host side are staged frame by frame, one host thread per GPU.
    std::vector<std::vector<int>> resident(ndev), staged(ndev);
    for (int i = 0; i < nb_frames; i++) {
        if (owner[i] >= ndev) { log_msg(c, 0, "sws_scale_frames(): frame %d lives on GPU %d, beyond the %d GPUs in use\n", i, owner[i], ndev); return SWS_AVERROR(EINVAL); }
        const bool res = sdev[i] >= 0 && ddev[i] >= 0 && c->plan != PLAN_CASCADE;
        (res ? resident : staged)[(size_t)owner[i]].push_back(i);
    }
    std::vector<DeviceState *> st(ndev, nullptr);
    // first use on a GPU: its own copy of the tables.  The home GPU's go up by a host -> device copy; the peers' likewise, or -- option rccl_tables -- by
    // ONE ncclBroadcast per table block from the home GPU over xGMI (dev_rccl.hip: the peers plan with their uploads held back, then the blocks are delivered)
    const bool use_rccl = rccl_tables_wanted(c) && c->plan != PLAN_CASCADE;
    std::vector<DeviceState *> fed;
    if (use_rccl) { ret = dev_prepare_on(c, c->dev); if (ret < 0) return ret; }
    for (int g = 0; g < ndev; g++) {
        if (resident[g].empty() && staged[g].empty()) continue;
        st[g] = dev_state_for(c, g);
        if (!st[g]) return AVERROR_EXTERNAL_;
        const bool peer = st[g] != c->dev;
        if (use_rccl && peer && !(st[g]->epoch == c->tables_epoch && st[g]->stream)) { st[g]->defer_uploads = true; st[g]->deferred.clear(); fed.push_back(st[g]); }
        ret = dev_prepare_on(c, st[g]);
        if (ret < 0) { for (DeviceState *d : fed) { d->defer_uploads = false; d->deferred.clear(); d->epoch = 0; } return ret; }
    }
    if (!fed.empty()) {
        ret = rccl_deliver_tables(c, c->dev, fed);
        if (ret < 0) { for (DeviceState *d : fed) { d->defer_uploads = false; d->deferred.clear(); d->epoch = 0; } return ret; }
    }
    for (int g = 0; g < ndev; g++) {
        if (resident[g].empty()) continue;
        HIPCHK(hipSetDevice(g));
        std::vector<SwsFramePtrs> fr(resident[g].size());
        for (size_t j = 0; j < fr.size(); j++) {
            const int i = resident[g][j];
            std::memset(&fr[j], 0, sizeof(SwsFramePtrs));
            for (int k = 0; k < nps; k++) { fr[j].src[k] = srcFrames[i]->data[k]; fr[j].srcStride[k] = srcFrames[i]->linesize[k]; }
            for (int k = 0; k < npd; k++) { fr[j].dst[k] = dstFrames[i]->data[k]; fr[j].dstStride[k] = dstFrames[i]->linesize[k]; }
        }
        ret = launch_plan(c, st[g], fr.data(), (int)fr.size(), 0, c->opts.src_h);
        if (ret < 0) return ret;
    }
    int nthreads = 0;
    for (int g = 0; g < ndev; g++) nthreads += !staged[g].empty();
    auto run_staged = [&](int g) -> int {
        if (hipSetDevice(g) != hipSuccess) { (void)hipGetLastError(); return AVERROR_EXTERNAL_; }
        for (int i : staged[g]) {
            int r = run_single(c, st[g], srcFrames[i]->data, srcFrames[i]->linesize, 0, c->opts.src_h, dstFrames[i]->data, dstFrames[i]->linesize);
            if (r < 0) return r;
        }
        return 0;
    };
    // (a cascade's children -- their per-GPU device state, plan names and tables -- hang off the one shared context and are prepared inside
    //  run_single(): those frames are staged GPU after GPU on this thread instead of one thread per GPU)
    if (nthreads <= 1 || c->plan == PLAN_CASCADE) {
        for (int g = 0; g < ndev; g++) if (!staged[g].empty()) { ret = run_staged(g); if (ret < 0) return ret; }
    } else {
        std::vector<std::thread> th;
        std::vector<int> rc(ndev, 0);
        for (int g = 0; g < ndev; g++) if (!staged[g].empty()) th.emplace_back([&, g] { rc[g] = run_staged(g); });
        for (auto &t : th) t.join();
        for (int g = 0; g < ndev; g++) if (rc[g] < 0) return rc[g];
    }
    return nb_frames;
}

int run_single(SwsInternal *c, DeviceState *d, const uint8_t *const src[4], const int srcStride[4], int sliceY, int sliceH,
                      uint8_t *const dst[4], const int dstStride[4], bool casc_flip0)
{
    const SwsContext &o = c->opts;

    if (c->plan == PLAN_CASCADE) { // scale_cascaded, swscale.c:992-1018; scale_gamma, :959-990 (whole frames)
        SwsInternal *c0 = c->cascade[0], *c1 = c->cascade[1], *c2 = c->cascade[2];
        DeviceState *cd[3] = { nullptr, nullptr, nullptr };
        for (int k = 0; k < 3; k++) {
            SwsInternal *cc = c->cascade[k];
            if (!cc) continue;
            // children run on the parent's GPU and stream
            cd[k] = dev_state_for(cc, d->device);
            if (!cd[k]) return AVERROR_EXTERNAL_;
            if (cd[k]->stream != d->stream) {
                if (cd[k]->own_stream && cd[k]->stream) { (void)hipStreamSynchronize(cd[k]->stream); (void)hipStreamDestroy(cd[k]->stream); }
                cd[k]->stream = d->stream; cd[k]->own_stream = false;
            }
            int r = dev_prepare_on(cc, cd[k]);
            if (r < 0) return r;
        }
        int ls[4]; size_t offs[4], total;
        image_layout(c->cascade_fmt, c->cascade_w, c->cascade_h, 256, ls, offs, &total);
        const size_t had = d->casc_bytes;
        int r = grow(c, &d->casc_img, &d->casc_bytes, total);
        if (r < 0) return r;
        // av_image_alloc() leaves the intermediate picture uninitialised and the pair-wise yuv2rgb converters never write the last
        // pixel of an odd width: start from zeros (as the oracle does) so that the result does not depend on stale memory
        if (d->casc_bytes != had) { HIPCHK(hipMemsetAsync(d->casc_img, 0, d->casc_bytes, d->stream)); }
        uint8_t *tmp[4] = { nullptr, nullptr, nullptr, nullptr };   // bgr24 / bgra / bgr48 / bgra64 (matrix cascade) or yuv420p / yuva420p (extreme ratios)
        int tls[4] = { 0, 0, 0, 0 };
        for (int k = 0; k < pix_nb_planes(pix_desc(c->cascade_fmt)); k++) { tmp[k] = (uint8_t *)d->casc_img + offs[k]; tls[k] = ls[k]; }
        if (casc_flip0) {
            uint8_t *ft[4] = { nullptr, nullptr, nullptr, nullptr };
            int fls[4] = { 0, 0, 0, 0 };
            for (int k = 0; k < pix_nb_planes(pix_desc(c->cascade_fmt)); k++) {
                int rb, prow; plane_geometry(c->cascade_fmt, c->cascade_w, c->cascade_h, k, &rb, &prow);
                ft[k] = tmp[k] + (int64_t)(prow - 1) * tls[k]; fls[k] = -tls[k];
            }
            r = run_single(c0, cd[0], src, srcStride, sliceY, sliceH, ft, fls);
        } else
        r = run_single(c0, cd[0], src, srcStride, sliceY, sliceH, tmp, tls);
        if (r < 0) return r;
        if (c->cascade_ed && c0->mono_y16) {   // error diffusion of the luma words into a 1 bpp destination (context.cpp; sws_k_ed_mono)
            const int n = (o.dst_w + 1) & ~1, H = o.dst_h, nbytes = (o.dst_w + 7) >> 3;
            if (!d->d_ed_err) {
                HIPCHK(hipMalloc(&d->d_ed_err, sizeof(int) * (size_t)(n + 4)));
                HIPCHK(hipMemsetAsync(d->d_ed_err, 0, sizeof(int) * (size_t)(n + 4), d->stream));
            } else if (o.flags & SWS_BITEXACT) {   // swscale.c:1084-1086
                HIPCHK(hipMemsetAsync(d->d_ed_err, 0, sizeof(int) * (size_t)(n + 4), d->stream));
            }
            const int white = o.dst_format == AV_PIX_FMT_MONOWHITE;
            if (is_device_ptr(dst[0])) {
                launch_ed_mono(d->stream, tmp[0], tls[0], dst[0], dstStride[0], n, H, (int *)d->d_ed_err, white);
                HIPCHK(hipGetLastError());
                return r;
            }
            // a host destination: the 2 / 1 forms leave a trailing partial byte alone, so the staging picture starts from the caller's bytes
            const int ls2 = (nbytes + 255) & ~255;
            r = grow(c, &d->casc_img2, &d->casc_bytes2, (size_t)ls2 * H);
            if (r < 0) return r;
            HIPCHK(hipMemcpy2DAsync(d->casc_img2, (size_t)ls2, dst[0], (size_t)dstStride[0], (size_t)nbytes, (size_t)H, hipMemcpyHostToDevice, d->stream));
            launch_ed_mono(d->stream, tmp[0], tls[0], (uint8_t *)d->casc_img2, ls2, n, H, (int *)d->d_ed_err, white);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpy2DAsync(dst[0], (size_t)dstStride[0], d->casc_img2, (size_t)ls2, (size_t)nbytes, (size_t)H, hipMemcpyDeviceToHost, d->stream));
            HIPCHK(hipStreamSynchronize(d->stream));
            return c0->opts.dst_h;
        }
        if (c->cascade_ed) {   // error diffusion of the rgb24 picture into the 8 / 4 bpp destination (context.cpp; sws_k_ed_rgb8)
            const int df = o.dst_format, W = o.dst_w, H = o.dst_h;
            const bool rgbo = df == AV_PIX_FMT_RGB8 || df == AV_PIX_FMT_RGB4_BYTE, b8pp = df == AV_PIX_FMT_RGB8 || df == AV_PIX_FMT_BGR8;
            const int r8 = b8pp ? (rgbo ? 5 : 0) : (rgbo ? 3 : 0), g8 = b8pp ? (rgbo ? 2 : 3) : 1, b8 = b8pp ? (rgbo ? 0 : 6) : (rgbo ? 0 : 3);
            if (!d->d_ed_err) {   // FF_ALLOCZ_TYPED_ARRAY(c->dither_error[i], dst_w + 3), utils.c:1744-1747
                HIPCHK(hipMalloc(&d->d_ed_err, sizeof(int) * 3 * (size_t)(W + 3)));
                HIPCHK(hipMemsetAsync(d->d_ed_err, 0, sizeof(int) * 3 * (size_t)(W + 3), d->stream));
            } else if (o.flags & SWS_BITEXACT) {   // scale_internal (swscale.c:1084-1086): a bit-exact context starts every frame from a clean line
                HIPCHK(hipMemsetAsync(d->d_ed_err, 0, sizeof(int) * 3 * (size_t)(W + 3), d->stream));
            }
            if (is_device_ptr(dst[0])) {
                launch_ed_rgb8(d->stream, tmp[0], tls[0], dst[0], dstStride[0], W, H, (int *)d->d_ed_err, b8pp ? 8 : 4, r8, g8, b8);
                HIPCHK(hipGetLastError());
                return r;
            }
            const int ls2 = (W + 255) & ~255;
            r = grow(c, &d->casc_img2, &d->casc_bytes2, (size_t)ls2 * H);
            if (r < 0) return r;
            launch_ed_rgb8(d->stream, tmp[0], tls[0], (uint8_t *)d->casc_img2, ls2, W, H, (int *)d->d_ed_err, b8pp ? 8 : 4, r8, g8, b8);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpy2DAsync(dst[0], (size_t)dstStride[0], d->casc_img2, (size_t)ls2, (size_t)W, (size_t)H, hipMemcpyDeviceToHost, d->stream));
            HIPCHK(hipStreamSynchronize(d->stream));
            return c0->opts.dst_h;
        }
        if (!c->cascade_gamma) return run_single(c1, cd[1], tmp, tls, 0, c0->opts.dst_h, dst, dstStride);
        // gamma cascade: table pass over the RGBA64 source of the scaling step (in place, like gamma_convert on the cascade's own
        // intermediate), scale, table pass over its output, then the conversion to the destination format
        if (!d->d_gamma_tab) {   // alloc_gamma_tbl (utils.c:1046-1058): tbl[i] = pow(i / 65535.0, e) * 65535.0 stored to uint16_t; [0] = 2.2, [1] = 1 / 2.2
            static std::vector<uint16_t> tab;
            static std::once_flag once;
            std::call_once(once, [] {
                tab.resize(2 * 65536);
                for (int i = 0; i < 65536; i++) {
                    tab[(size_t)i] = (uint16_t)(std::pow(i / 65535.0, 2.2) * 65535.0);
                    tab[(size_t)65536 + i] = (uint16_t)(std::pow(i / 65535.0, 1.f / 2.2) * 65535.0);
                }
            });
            HIPCHK(hipMalloc(&d->d_gamma_tab, tab.size() * sizeof(uint16_t)));
            HIPCHK(hipMemcpyAsync(d->d_gamma_tab, tab.data(), tab.size() * sizeof(uint16_t), hipMemcpyHostToDevice, d->stream));   // (static storage: stays valid)
        }
        const uint16_t *gt = (const uint16_t *)d->d_gamma_tab;
        // the inverse table on the scaling step's source: one pass over the picture when the step's line schedule converts every line it reads
        // exactly once; otherwise (holes: FAST_BILINEAR / POINT down-scaling) pass 1 of the step applies it per virtual line (build_vlines)
        if (c1->gamma_in_reader) cd[1]->params.gamma_tab = gt + 65536;
        else launch_gamma_rgba64(d->stream, tmp[0], tls[0], o.src_w, o.src_h, gt + 65536);
        uint8_t *out1[4] = { dst[0], dst[1], dst[2], dst[3] };
        int os1[4] = { dstStride[0], dstStride[1], dstStride[2], dstStride[3] };
        bool out1_dev = true;
        if (c2) {
            int ls2[4]; size_t offs2[4], total2;
            image_layout(AV_PIX_FMT_RGBA64LE, o.dst_w, o.dst_h, 256, ls2, offs2, &total2);
            r = grow(c, &d->casc_img2, &d->casc_bytes2, total2);
            if (r < 0) return r;
            out1[0] = (uint8_t *)d->casc_img2; out1[1] = out1[2] = out1[3] = nullptr; os1[0] = ls2[0]; os1[1] = os1[2] = os1[3] = 0;
        } else out1_dev = is_device_ptr(dst[0]);
        if (!out1_dev) {
            // the scaling step writes straight into a host picture: its output has to pass the table before it leaves the device, so the
            // step runs into a device picture first
            int ls2[4]; size_t offs2[4], total2;
            image_layout(AV_PIX_FMT_RGBA64LE, o.dst_w, o.dst_h, 256, ls2, offs2, &total2);
            r = grow(c, &d->casc_img2, &d->casc_bytes2, total2);
            if (r < 0) return r;
            uint8_t *t1[4] = { (uint8_t *)d->casc_img2, nullptr, nullptr, nullptr };
            int s1[4] = { ls2[0], 0, 0, 0 };
            r = run_single(c1, cd[1], tmp, tls, 0, o.src_h, t1, s1);
            if (r < 0) return r;
            launch_gamma_rgba64(d->stream, t1[0], s1[0], o.dst_w, o.dst_h, gt);
            HIPCHK(hipMemcpy2DAsync(dst[0], (size_t)dstStride[0], t1[0], (size_t)s1[0], (size_t)o.dst_w * 8, (size_t)o.dst_h, hipMemcpyDeviceToHost, d->stream));
            HIPCHK(hipStreamSynchronize(d->stream));
            return r;
        }
        r = run_single(c1, cd[1], tmp, tls, 0, o.src_h, out1, os1);
        if (r < 0) return r;
        launch_gamma_rgba64(d->stream, out1[0], os1[0], o.dst_w, o.dst_h, gt);
        if (c2) r = run_single(c2, cd[2], out1, os1, 0, o.dst_h, dst, dstStride);
        return r;
    }

    const int nps = pix_nb_planes(pix_desc(o.src_format)), npd = pix_nb_planes(pix_desc(o.dst_format));
    const bool src_dev = is_device_ptr(src[0]), dst_dev = is_device_ptr(dst[0]);
    const bool unscaled = c->plan != PLAN_MAIN;
    // destination rows produced by this call
    const int outY = unscaled ? sliceY : 0, outH = unscaled ? sliceH : o.dst_h;

    SwsFramePtrs fr;
    std::memset(&fr, 0, sizeof(fr));
    hipStream_t st = d->stream;

    if (src_dev) {
        for (int k = 0; k < nps; k++) {
            int y0, rows; rows_of_slice(o.src_format, k, sliceY, sliceH, &y0, &rows);
            // unscaled converters take slice-relative source pointers (swscale.c:1163-1188): rebase to absolute rows
            fr.src[k] = src[k] - (int64_t)(unscaled ? y0 : 0) * srcStride[k];
            fr.srcStride[k] = srcStride[k];
        }
    } else {
        int ls[4]; size_t offs[4], total;
        image_layout(o.src_format, o.src_w, o.src_h, 256, ls, offs, &total);
        int r = grow(c, &d->stage_src, &d->stage_src_bytes, total);
        if (r < 0) return r;
        for (int k = 0; k < nps; k++) {
            int rb, prow; plane_geometry(o.src_format, o.src_w, o.src_h, k, &rb, &prow);
            int y0, rows; rows_of_slice(o.src_format, k, sliceY, sliceH, &y0, &rows);
            if (!unscaled) { y0 = 0; rows = prow; }
            rows = std::min(rows, prow - y0);
            uint8_t *dbase = (uint8_t *)d->stage_src + offs[k];
            const uint8_t *s = src[k]; // slice-relative for unscaled, whole plane otherwise
            if (rows == 1) {   // (also the palette of a pal8 picture, whose linesize means nothing)
                HIPCHK(hipMemcpyAsync(dbase + (size_t)y0 * ls[k], s, rb, hipMemcpyHostToDevice, st));
            } else if (srcStride[k] >= 0) {
                HIPCHK(hipMemcpy2DAsync(dbase + (size_t)y0 * ls[k], ls[k], s, srcStride[k], rb, rows, hipMemcpyHostToDevice, st));
            } else { // bottom-up image: copy row by row in reverse so the staged plane is top-down
                for (int y = 0; y < rows; y++)
                    HIPCHK(hipMemcpyAsync(dbase + (size_t)(y0 + y) * ls[k], s + (int64_t)y * srcStride[k], rb, hipMemcpyHostToDevice, st));
            }
            fr.src[k] = dbase; fr.srcStride[k] = ls[k];
        }
    }
    int dls[4]; size_t doffs[4], dtotal = 0;
    if (dst_dev) {
        for (int k = 0; k < npd; k++) { fr.dst[k] = dst[k]; fr.dstStride[k] = dstStride[k]; }
    } else {
        image_layout(o.dst_format, o.dst_w, o.dst_h, 256, dls, doffs, &dtotal);
        int r = grow(c, &d->stage_dst, &d->stage_dst_bytes, dtotal);
        if (r < 0) return r;
        for (int k = 0; k < npd; k++) { fr.dst[k] = (uint8_t *)d->stage_dst + doffs[k]; fr.dstStride[k] = dls[k]; }
        // the special converters leave the last pixel (pair) of an odd width untouched (yuv2rgb.c pair loops, planarToP01x's
        // "src_w / 2" chroma loop, nv24_to_yuv420p_chroma, planarToYuy2 ...): the staging picture starts from the caller's data
        // (planarRgbToplanarRgbWrapper on 16-bit formats leaves the second half of the slice's last row -- or of every row -- untouched)
        // (ff_sws_alphablendaway covers chrSrcW columns of planes 1 and 2: half of a gbrap picture when init halved the RGB chroma width)
        if (unscaled && ((o.dst_w & 1) || (o.dst_h & 1) || c->plan == PLAN_UNSC_PLANARRGB_PLANARRGB || c->plan == PLAN_UNSC_ALPHABLEND)) {   // (odd heights: yuyvtoyuv420 writes chroma on odd rows only)
            for (int k = 0; k < npd; k++) {
                int rb, prow; plane_geometry(o.dst_format, o.dst_w, o.dst_h, k, &rb, &prow);
                int y0, rows; rows_of_slice(o.dst_format, k, outY, outH, &y0, &rows);
                rows = std::min(rows, prow - y0);
                if (dstStride[k] >= 0) {
                    HIPCHK(hipMemcpy2DAsync(fr.dst[k] + (size_t)y0 * dls[k], dls[k], dst[k] + (int64_t)y0 * dstStride[k], dstStride[k], rb, rows, hipMemcpyHostToDevice, st));
                } else {
                    for (int y = 0; y < rows; y++)
                        HIPCHK(hipMemcpyAsync(fr.dst[k] + (size_t)(y0 + y) * dls[k], dst[k] + (int64_t)(y0 + y) * dstStride[k], rb, hipMemcpyHostToDevice, st));
                }
            }
        }
    }

    int ret = launch_plan(c, d, &fr, 1, sliceY, sliceH);
    if (ret < 0) return ret;

    if (!dst_dev) {
        for (int k = 0; k < npd; k++) {
            int rb, prow; plane_geometry(o.dst_format, o.dst_w, o.dst_h, k, &rb, &prow);
            int y0, rows; rows_of_slice(o.dst_format, k, outY, outH, &y0, &rows);
            rows = std::min(rows, prow - y0);
            const uint8_t *sbase = (const uint8_t *)d->stage_dst + doffs[k] + (size_t)y0 * dls[k];
            if (dstStride[k] >= 0) {
                HIPCHK(hipMemcpy2DAsync(dst[k] + (int64_t)y0 * dstStride[k], dstStride[k], sbase, dls[k], rb, rows, hipMemcpyDeviceToHost, st));
            } else {
                for (int y = 0; y < rows; y++)
                    HIPCHK(hipMemcpyAsync(dst[k] + (int64_t)(y0 + y) * dstStride[k], sbase + (size_t)y * dls[k], rb, hipMemcpyDeviceToHost, st));
            }
        }
    }
    if (!dst_dev || !src_dev) HIPCHK(hipStreamSynchronize(st)); // host buffers: synchronous like the reference
    if (c->plan == PLAN_UNSC_ALPHABLEND) return 0;   // ff_sws_alphablendaway returns 0 (alphablend.c:176) and sws_scale() passes it on
    return unscaled ? sliceH : o.dst_h;
}

} // namespace swship

using namespace swship;

// ------------------------------------------------------------------------------------------
// public entry points
// ------------------------------------------------------------------------------------------
bool swship::check_image_pointers(const uint8_t *const data[4], int fmt, const int linesizes[4]) // swscale.c:729-743
{
    const PixDesc *d = pix_desc(fmt);
    for (int i = 0; i < d->nb_components; i++) {
        const int plane = d->comp[i].plane;
        if (!data[plane] || !linesizes[plane]) return false;
    }
    return true;
}

// Slices on the scaled path (scale_internal swscale.c:1076-1104, ff_swscale :372-381, :404-470, :566).
// The reference pulls destination rows as soon as the ring buffer holds the source rows they need and returns how many
// it produced.  Here the slices are assembled into a context-owned device copy of the source picture, the return
// value of every call is the count the reference's cursor logic gives, and the picture is converted in one go when
// the last slice arrives (rows become valid then).  Bottom-up sequences are the reference's flipped image
// (negative strides on both sides), so the result is flip(scale(flip(src))) exactly as there.
static int scale_slice(SwsInternal *c, const uint8_t *const src[], const int srcStride[], int srcSliceY, int srcSliceH,
                       uint8_t *const dst[], const int dstStride[])
{
    const SwsContext &o = c->opts;
    if (c->sliceDir == 0 && srcSliceY != 0 && srcSliceY + srcSliceH != o.src_h) {
        log_msg(c, 0, "Slices start in the middle!\n");                       // swscale.c:1096-1099
        return SWS_AVERROR(EINVAL);
    }
    if (c->sliceDir == 0) c->sliceDir = srcSliceY == 0 ? 1 : -1;
    const int yint = c->sliceDir == 1 ? srcSliceY : o.src_h - srcSliceY - srcSliceH;   // srcSliceY_internal (:1158)
    int ret = dev_prepare(c);
    if (ret < 0) return ret;
    DeviceGuard guard;
    DeviceState *d = c->dev;
    HIPCHK(hipSetDevice(d->device));
    hipStream_t st = d->stream;
    int ls[4]; size_t offs[4], total;
    image_layout(o.src_format, o.src_w, o.src_h, 256, ls, offs, &total);
    if (total > d->slice_bytes) {
        if (d->slice_img) HIPCHK(hipFree(d->slice_img));
        d->slice_img = nullptr; d->slice_bytes = 0;
        HIPCHK(hipMalloc(&d->slice_img, total));
        { int pr = poison(c, d->slice_img, total); if (pr < 0) return pr; }
        d->slice_bytes = total;
    }
    const int nps = pix_nb_planes(pix_desc(o.src_format));
    const bool src_dev = is_device_ptr(src[0]);
    for (int k = 0; k < nps; k++) {
        int rb, prow; plane_geometry(o.src_format, o.src_w, o.src_h, k, &rb, &prow);
        int y0, rows; rows_of_slice(o.src_format, k, srcSliceY, srcSliceH, &y0, &rows);
        uint8_t *dp = (uint8_t *)d->slice_img + offs[k] + (size_t)y0 * ls[k];
        const hipMemcpyKind kind = src_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        if (srcStride[k] >= rb) HIPCHK(hipMemcpy2DAsync(dp, ls[k], src[k], srcStride[k], rb, rows, kind, st));
        else for (int y = 0; y < rows; y++) HIPCHK(hipMemcpyAsync(dp + (size_t)y * ls[k], src[k] + (int64_t)y * srcStride[k], rb, kind, st));
    }
    if (!src_dev) HIPCHK(hipStreamSynchronize(st));                            // the caller may reuse its slice buffer
    // ---- the reference's cursor: how many destination rows this slice completes ----
    // Cascades: scale_cascaded (swscale.c:993-1020) lets its first context assemble the intermediate picture slice by slice and answers 0 until
    // that context's sliceDir has reset, then runs the second one once and returns its row count.  scale_gamma (:959-990) hands the slice
    // coordinates to its scaling context, whose cursor is the one that answers; the error-diffusion cascade of this library stands for ONE
    // main-path context of the reference, the inner context's geometry.  (In the gamma cascade the reference's second context reads the rows of
    // the slice from the intermediate picture whether or not the first one has produced them yet; here every row is there when it is read.)
    // (force_scaler marks the inner context of a cascade that stands for ONE main-path context of the reference: error diffusion, packed 4:2:2)
    const bool casc = c->plan == PLAN_CASCADE, casc_inner = casc && c->cascade[0] && c->cascade[0]->force_scaler;
    const bool casc_plain = casc && !c->cascade_gamma && !casc_inner;
    const SwsInternal *cur = !casc ? c : casc_inner ? c->cascade[0] : c->cascade[1];
    if (yint == 0) c->slice_dstY = 0;
    const int last = c->slice_dstY;
    int dstY = last;
    const int cvs = cur->chrDstVSubSample;
    const bool cur_main = cur && cur->plan == PLAN_MAIN && cur->opts.src_h == o.src_h;
    for (; cur_main && !casc_plain && dstY < cur->opts.dst_h; dstY++) {
        const int chrDstY = dstY >> cvs;
        const int firstLum2 = std::max(1 - cur->vLum.size, cur->vLum.pos[std::min(dstY | ((1 << cvs) - 1), cur->opts.dst_h - 1)]);
        const int firstChr = std::max(1 - cur->vChr.size, cur->vChr.pos[chrDstY]);
        const int lastLum2 = std::min(o.src_h, firstLum2 + cur->vLum.size) - 1;
        const int lastChr = std::min(cur->chrSrcH, firstChr + cur->vChr.size) - 1;
        const bool enough = lastLum2 < yint + srcSliceH && lastChr < -((-(yint + srcSliceH)) >> cur->chrSrcVSubSample);
        if (!enough) break;
    }
    if (casc && !casc_plain && !cur_main && yint + srcSliceH == o.src_h) dstY = o.dst_h;   // (an answering context without a row cursor: everything at the end)
    c->slice_dstY = dstY;
    if (yint + srcSliceH == o.src_h) {                                         // sequence complete (:1189-1190): convert
        const bool flip = c->sliceDir == -1;
        c->sliceDir = 0;
        const uint8_t *s4[4] = { nullptr, nullptr, nullptr, nullptr };
        uint8_t *d4[4] = { nullptr, nullptr, nullptr, nullptr };
        int ss4[4] = { 0, 0, 0, 0 }, ds4[4] = { 0, 0, 0, 0 };
        const int npd = pix_nb_planes(pix_desc(o.dst_format));
        for (int k = 0; k < nps; k++) {
            int rb, prow; plane_geometry(o.src_format, o.src_w, o.src_h, k, &rb, &prow);
            s4[k] = (const uint8_t *)d->slice_img + offs[k] + (flip ? (size_t)(prow - 1) * ls[k] : 0);
            ss4[k] = flip ? -ls[k] : ls[k];
        }
        const bool flip_dst = flip && !casc_plain;   // (scale_cascaded: the second context runs once, top-down, whatever the slice order was)
        for (int k = 0; k < npd; k++) {
            int rb, prow; plane_geometry(o.dst_format, o.dst_w, o.dst_h, k, &rb, &prow);
            d4[k] = dst[k] + (flip_dst ? (int64_t)(prow - 1) * dstStride[k] : 0);
            ds4[k] = flip_dst ? -dstStride[k] : dstStride[k];
        }
        ret = run_single(c, d, s4, ss4, 0, o.src_h, d4, ds4, flip && casc_plain);
        if (ret < 0) return ret;
        if (casc_plain) return ret;
    }
    return dstY - last;
}

extern "C" {

int sws_scale(SwsContext *sws, const uint8_t *const srcSlice[], const int srcStride[], int srcSliceY,
              int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    if (!sws) return SWS_AVERROR(EINVAL);
    SwsInternal *c = internal(sws);
    if (!c->legacy_init) {                      // swscale.c:1633-1634
        log_msg(c, 0, "sws_scale() called on a context that was not initialised with sws_init_context()\n");
        return SWS_AVERROR(EINVAL);
    }
    // scale_internal(), swscale.c:1022-1070
    if (!srcStride || !dstStride || !dst || !srcSlice) {
        log_msg(c, 0, "One of the input parameters to sws_scale() is NULL, please check the calling code\n");
        return SWS_AVERROR(EINVAL);
    }
    const int mh_src = 1 << c->chrSrcVSubSample;
    if ((srcSliceY & (mh_src - 1)) || ((srcSliceH & (mh_src - 1)) && srcSliceY + srcSliceH != sws->src_h) ||
        srcSliceY + srcSliceH > sws->src_h || srcSliceY < 0 || srcSliceH < 0) {
        log_msg(c, 0, "Slice parameters %d, %d are invalid\n", srcSliceY, srcSliceH);
        return SWS_AVERROR(EINVAL);
    }
    if (!check_image_pointers(srcSlice, sws->src_format, srcStride)) {
        log_msg(c, 0, "bad src image pointers\n");
        return SWS_AVERROR(EINVAL);
    }
    if (!check_image_pointers((const uint8_t *const *)dst, sws->dst_format, dstStride)) {
        log_msg(c, 0, "bad dst image pointers\n");
        return SWS_AVERROR(EINVAL);
    }
    if (srcSliceH == 0) return 0;               // :1072-1074
    const bool whole = srcSliceY == 0 && srcSliceH == sws->src_h;
    if ((c->plan == PLAN_MAIN || c->plan == PLAN_CASCADE) && (!whole || c->sliceDir != 0)) return scale_slice(c, srcSlice, srcStride, srcSliceY, srcSliceH, dst, dstStride);
    const uint8_t *s4[4] = { srcSlice[0], nullptr, nullptr, nullptr };
    uint8_t *d4[4] = { dst[0], nullptr, nullptr, nullptr };
    int ss4[4] = { srcStride[0], 0, 0, 0 }, ds4[4] = { dstStride[0], 0, 0, 0 };
    const int nps = pix_nb_planes(pix_desc(sws->src_format)), npd = pix_nb_planes(pix_desc(sws->dst_format));
    for (int k = 1; k < nps; k++) { s4[k] = srcSlice[k]; ss4[k] = srcStride[k]; }
    for (int k = 1; k < npd; k++) { d4[k] = dst[k]; ds4[k] = dstStride[k]; }
    return dev_run(c, s4, ss4, srcSliceY, srcSliceH, d4, ds4, 0, nullptr, nullptr);
}

} // extern "C"
