// The AVFrame side of the API: sws_scale_frame() / sws_scale_frames() / sws_frame_setup() / sws_is_noop() / sws_test_frame()
// and the slice API on frames.
//
// Two modes, as in the reference (libswscale/swscale.c:1404-1480):
//  * a context initialised with sws_init_context() / sws_getContext() ("legacy"): the frames must match the context; frame metadata
//    is ignored, the conversion is the one the context was built for (sws_frame_start + sws_send_slice + sws_receive_slice).
//  * a context that was only sws_alloc_context()ed ("dynamic"): sws_frame_setup() derives one conversion per field from the frames'
//    own properties (format, size, color_range, colorspace, chroma_location, interlacing; libswscale/format.c:344-478
//    ff_fmt_from_frame) and keeps it until the properties or the context's options change (graph.c ff_sws_graph_reinit).  The
//    conversion itself is the legacy scaler configured the way libswscale/graph.c:558-661 add_legacy_sws_pass() configures it
//    (the only stable back-end: swscale.h:113 SWS_BACKEND_STABLE = SWS_BACKEND_LEGACY).  Colour mapping between different
//    primaries / transfer functions (graph.c:760-794, the 3-D LUT pre-pass) is not part of the hot path: refused with ENOTSUP.
//
// Frames of format AV_PIX_FMT_HIP (include/hwcontext_hip.h) carry HBM pointers and a hw_frames_ctx naming the layout; they are
// accepted in both modes.  This is where the reference says "Only Vulkan devices are supported" (swscale.c:1530-1533).
#include "swsint.hpp"
#include "../../include/hwcontext_hip.h"
#include <cstring>
#include <vector>

namespace swship {

enum { COL_SPC_RGB = 0, COL_SPC_UNSPECIFIED = 2, COL_PRI_BT709 = 1, COL_PRI_UNSPECIFIED = 2, COL_TRC_BT709 = 1, COL_TRC_UNSPECIFIED = 2,
       COL_TRC_SMPTE2084 = 16, COL_TRC_SMPTE428 = 17, COL_TRC_ARIB_STD_B67 = 18,
       FRAME_DATA_MASTERING_DISPLAY_METADATA = 11, FRAME_DATA_DYNAMIC_HDR_PLUS = 17 };   // libavutil/pixfmt.h, frame.h:49-206

static uint64_t fnv1a(uint64_t h, const void *p, size_t n)
{
    const uint8_t *b = (const uint8_t *)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}

// sanitize_fmt(), format.c:305-342
static void sanitize_fmt(SwsFmt *fmt, const PixDesc *desc)
{
    if (desc->flags & (PIXFLAG_RGB | PIXFLAG_PAL | PIXFLAG_BAYER)) {   // RGB-like family
        fmt->csp = COL_SPC_RGB;
        fmt->range = SWS_COL_RANGE_JPEG;
    } else if (pix_is_xyz(fmt->format)) {
        fmt->csp = COL_SPC_UNSPECIFIED;
        fmt->prim = COL_PRI_BT709;
        fmt->trc = COL_TRC_SMPTE428;
    } else if (desc->nb_components < 3) {       // grayscale
        fmt->prim = COL_PRI_UNSPECIFIED;
        fmt->csp = COL_SPC_UNSPECIFIED;
        fmt->range = (desc->flags & PIXFLAG_FLOAT) ? SWS_COL_RANGE_UNSPECIFIED : SWS_COL_RANGE_JPEG;
    }
    switch (fmt->format) {
    case AV_PIX_FMT_YUVJ420P: case AV_PIX_FMT_YUVJ411P: case AV_PIX_FMT_YUVJ422P: case AV_PIX_FMT_YUVJ444P: case AV_PIX_FMT_YUVJ440P:
        fmt->range = SWS_COL_RANGE_JPEG;
        break;
    }
    if (!desc->log2_chroma_w && !desc->log2_chroma_h) fmt->loc = SWS_CHROMA_LOC_UNSPECIFIED;
}

// ff_fmt_from_frame(), format.c:344-478.  Returns 0, or a negative AVERROR for a hardware frame this library cannot read.
static int fmt_from_frame(const SwsFrameView *f, int field, SwsFmt *out)
{
    SwsFmt fmt;
    std::memset(&fmt, 0, sizeof(fmt));
    fmt.format = f->format;
    fmt.hw_format = AV_PIX_FMT_NONE;
    fmt.hip_device = -1;
    if (f->hw_frames_ctx) {
        const SwsHWFramesContext *fc = (const SwsHWFramesContext *)f->hw_frames_ctx->data;
        if (!fc || !fc->device_ctx) return SWS_AVERROR(EINVAL);
        fmt.hw_format = f->format;
        fmt.format = fc->sw_format;
        fmt.device_ref = fc->device_ctx;
        if (fc->device_ctx->type == AV_HWDEVICE_TYPE_HIP && fc->device_ctx->hwctx) {
            const AVHIPDeviceContext *hc = (const AVHIPDeviceContext *)fc->device_ctx->hwctx;
            fmt.hip_device = hc->device;
            fmt.hip_stream = hc->stream;
        }
    } else if (f->format == AV_PIX_FMT_HIP) return SWS_AVERROR(EINVAL);   // a hardware format without its frames context
    fmt.width = f->width; fmt.height = f->height;
    fmt.range = f->color_range; fmt.csp = f->colorspace; fmt.loc = f->chroma_location;
    fmt.prim = f->color_primaries; fmt.trc = f->color_trc;
    const PixDesc *desc = pix_desc(fmt.format);
    if (!desc) { *out = fmt; return 0; }          // unknown format: ff_test_fmt() refuses it
    sanitize_fmt(&fmt, desc);
    if (f->flags & SWS_FRAME_FLAG_INTERLACED) {
        fmt.height = (fmt.height + (field == 0)) >> 1;
        fmt.interlaced = 1;
        fmt.field = field;
    }
    // Mastering-display and HDR10+ side data change the gamut / luminance range the colour mapping works with (format.c:404-470).
    // This library does no colour mapping: it only needs to know whether the two frames carry the same information.
    fmt.side_hash = 0xcbf29ce484222325ull;
    for (int i = 0; i < f->nb_side_data && f->side_data; i++) {
        const SwsFrameSideData *sd = f->side_data[i];
        if (!sd || (sd->type != FRAME_DATA_MASTERING_DISPLAY_METADATA && sd->type != FRAME_DATA_DYNAMIC_HDR_PLUS)) continue;
        fmt.side_hash = fnv1a(fmt.side_hash, &sd->type, sizeof(sd->type));
        if (sd->data && sd->size) fmt.side_hash = fnv1a(fmt.side_hash, sd->data, sd->size);
    }
    *out = fmt;
    return 0;
}

static bool color_equal(const SwsFmt &a, const SwsFmt &b) { return a.prim == b.prim && a.trc == b.trc && a.side_hash == b.side_hash; }

// ff_fmt_equal(), format.h:125-136
static bool fmt_equal(const SwsFmt &a, const SwsFmt &b)
{
    return a.width == b.width && a.height == b.height && a.interlaced == b.interlaced && a.field == b.field && a.format == b.format &&
           a.range == b.range && a.csp == b.csp && a.loc == b.loc && color_equal(a, b);
}
// what a cached conversion depends on beyond ff_fmt_equal: where the pixels live -- the hardware format and the GPU ordinal.  The stream and the
// device context travel with the FRAMES, not with the graph: the stream is borrowed per call (dev_borrow_stream) and nothing cached depends on it,
// so frames of another frames context on the same GPU, or frames that carry another stream, reuse the cached graph (graph_reinit refreshes the two
// fields from the frames of the call; a rebuild would re-initialise the context, copy the tables synchronously and reallocate the working buffers)
static bool fmt_same(const SwsFmt &a, const SwsFmt &b)
{
    return fmt_equal(a, b) && a.hw_format == b.hw_format && a.hip_device == b.hip_device;
}

// ff_test_fmt(), format.c:683-693 with SWS_BACKEND_LEGACY
static bool test_fmt(const SwsFmt &f, int output)
{
    return f.width > 0 && f.height > 0 && pix_desc(f.format) &&
           (output ? sws_isSupportedOutput((enum AVPixelFormat)f.format) : sws_isSupportedInput((enum AVPixelFormat)f.format)) &&
           sws_test_colorspace(f.csp, output) && sws_test_primaries(f.prim, output) && sws_test_transfer(f.trc, output) &&
           sws_test_hw_format((enum AVPixelFormat)f.hw_format) && (unsigned)f.range < 3u && (unsigned)f.loc < 7u;
}

// ff_sws_chroma_pos(), format.c:554-592
static void chroma_pos(const SwsFmt &fmt, const PixDesc *desc, bool *incomplete, int *out_x, int *out_y)
{
    int loc = fmt.loc;
    const int sub_x = desc->log2_chroma_w, sub_y = desc->log2_chroma_h;
    if (loc == SWS_CHROMA_LOC_UNSPECIFIED) {    // center siting, for compatibility with sws_getContext()
        loc = SWS_CHROMA_LOC_CENTER;
        *incomplete |= sub_x || sub_y;
    }
    const int pos = loc - 1;                    // av_chroma_location_enum_to_pos(), libavutil/pixdesc.c:3902-3912
    int x_pos = (pos & 1) * 128, y_pos = ((pos >> 1) ^ (pos < 4)) * 128;
    x_pos *= (1 << sub_x) - 1;
    y_pos *= (1 << sub_y) - 1;
    if (sub_y && fmt.interlaced) {              // chroma samples sit next to the even rows of the frame
        if (fmt.field == 1) y_pos += (256 << sub_y) - 256;
        y_pos >>= 1;                            // the luma row distance of a field is twice the frame's
    }
    *out_x = x_pos; *out_y = y_pos;
}

// infer_prim_ref / infer_trc_ref / ff_infer_colors (format.c:487-552) followed by ff_sws_color_map_noop (cms.c:34-57): does the
// reference insert the 3-D LUT pre-pass for this pair?
static bool needs_color_mapping(SwsFmt src, SwsFmt dst, bool *incomplete)
{
    if (isGray(dst.format)) { dst.prim = src.prim; dst.trc = src.trc; dst.side_hash = src.side_hash; }       // graph.c:770-774
    else if (isGray(src.format)) { src.prim = dst.prim; src.trc = dst.trc; src.side_hash = dst.side_hash; }
    auto infer_prim = [&](SwsFmt &c, const SwsFmt &ref) {
        if (c.prim != COL_PRI_UNSPECIFIED) return;
        switch (ref.prim) { case 1: case 4: case 5: case 6: case 7: c.prim = ref.prim; break;   // BT709, BT470M, BT470BG, SMPTE170M, SMPTE240M
                            default: c.prim = COL_PRI_BT709; break; }
        *incomplete = true;
    };
    auto infer_trc = [&](SwsFmt &c, const SwsFmt &ref) {
        if (c.trc != COL_TRC_UNSPECIFIED) return;
        switch (ref.trc) { case COL_TRC_UNSPECIFIED: case COL_TRC_SMPTE2084: case COL_TRC_ARIB_STD_B67: c.trc = COL_TRC_BT709; break;
                           default: c.trc = ref.trc; break; }
        *incomplete = true;
    };
    infer_prim(dst, src); infer_prim(src, dst);
    infer_trc(dst, src); infer_trc(src, dst);
    // same encoding and the same (or no) mastering / HDR10+ information on both sides: same gamut, same luminance range, a no-op
    // for every intent.  Anything else goes through the LUT in the reference (or might: side data that differs is not examined).
    return !(src.prim == dst.prim && src.trc == dst.trc && src.side_hash == dst.side_hash);
}

static void graph_free(FrameGraph *g)
{
    if (g->legacy) sws_freeContext(g->legacy);
    *g = FrameGraph();
}

void frames_release(SwsInternal *c)
{
    graph_free(&c->graph[0]);
    graph_free(&c->graph[1]);
}

// validate_params(), swscale.c:1482-1501
static int validate_params(const SwsContext *o)
{
    if (o->threads < 0 || o->threads > 8192) return SWS_AVERROR(EINVAL);                  // SWS_MAX_THREADS
    if ((int)o->dither < 0 || (int)o->dither > SWS_DITHER_NB - 1) return SWS_AVERROR(EINVAL);
    if ((int)o->alpha_blend < 0 || (int)o->alpha_blend > SWS_ALPHA_BLEND_NB - 1) return SWS_AVERROR(EINVAL);
    if (o->intent < 0 || o->intent > 3) return SWS_AVERROR(EINVAL);                       // SWS_INTENT_NB - 1
    if ((int)o->scaler < 0 || (int)o->scaler > SWS_SCALE_NB - 1) return SWS_AVERROR(EINVAL);
    if ((int)o->scaler_sub < 0 || (int)o->scaler_sub > SWS_SCALE_NB - 1) return SWS_AVERROR(EINVAL);
    return 0;
}

// add_legacy_sws_pass(), graph.c:558-661: the legacy scaler for one field, configured from the two formats
static int build_legacy(SwsInternal *c, const SwsFmt &src, const SwsFmt &dst, FrameGraph *g)
{
    const SwsContext &ctx = c->opts;
    const PixDesc *sd = pix_desc(src.format), *dd = pix_desc(dst.format);
    SwsContext *sws = sws_alloc_context();
    if (!sws) return SWS_AVERROR(ENOMEM);
    sws->flags = ctx.flags; sws->dither = ctx.dither; sws->alpha_blend = ctx.alpha_blend; sws->gamma_flag = ctx.gamma_flag;
    sws->scaler = ctx.scaler; sws->scaler_sub = ctx.scaler_sub;
    sws->src_w = src.width; sws->src_h = src.height; sws->src_format = src.format; sws->src_range = src.range == SWS_COL_RANGE_JPEG;
    sws->dst_w = dst.width; sws->dst_h = dst.height; sws->dst_format = dst.format; sws->dst_range = dst.range == SWS_COL_RANGE_JPEG;
    bool inc = false;
    chroma_pos(src, sd, &inc, &sws->src_h_chr_pos, &sws->src_v_chr_pos);
    chroma_pos(dst, dd, &inc, &sws->dst_h_chr_pos, &sws->dst_v_chr_pos);
    inc |= src.range == SWS_COL_RANGE_UNSPECIFIED || dst.range == SWS_COL_RANGE_UNSPECIFIED;
    // the deprecated way of setting the chroma position still wins (legacy_chr_pos, graph.c:430-444)
    auto legacy_pos = [](int *pos, int over) { if (over != -513 && over != *pos) *pos = over; };
    legacy_pos(&sws->src_h_chr_pos, ctx.src_h_chr_pos); legacy_pos(&sws->src_v_chr_pos, ctx.src_v_chr_pos);
    legacy_pos(&sws->dst_h_chr_pos, ctx.dst_h_chr_pos); legacy_pos(&sws->dst_v_chr_pos, ctx.dst_v_chr_pos);
    // no offsets without subsampling: they would interfere with SWS_FULL_CHR_H_INP (graph.c:617-626)
    if (!sd->log2_chroma_w) sws->src_h_chr_pos = -513;
    if (!sd->log2_chroma_h) sws->src_v_chr_pos = -513;
    if (!dd->log2_chroma_w) sws->dst_h_chr_pos = -513;
    if (!dd->log2_chroma_h) sws->dst_v_chr_pos = -513;
    for (int i = 0; i < SWS_NUM_SCALER_PARAMS; i++) sws->scaler_params[i] = ctx.scaler_params[i];
    internal(sws)->tune = c->tune;
    int ret = sws_init_context(sws, nullptr, nullptr);
    if (ret < 0) { sws_freeContext(sws); return ret; }
    {   // the colour matrices of the two frames (graph.c:638-657)
        int in_full, out_full, brightness, contrast, saturation;
        int *inv_table, *table;
        if (sws_getColorspaceDetails(sws, &inv_table, &in_full, &table, &out_full, &brightness, &contrast, &saturation) >= 0) {
            inc |= src.csp != dst.csp && (src.csp == COL_SPC_UNSPECIFIED || dst.csp == COL_SPC_UNSPECIFIED);
            (void)sws_setColorspaceDetails(sws, sws_getCoefficients(src.csp), in_full, sws_getCoefficients(dst.csp), out_full,
                                           brightness, contrast, saturation);
        }
    }
    g->legacy = sws;
    g->incomplete |= inc;
    return 0;
}

// ff_sws_graph_reinit() + init_passes() (graph.c:800-840, :935-960) for one field
static int graph_reinit(SwsInternal *c, int field, const SwsFmt &src, const SwsFmt &dst)
{
    FrameGraph *g = &c->graph[field];
    if (g->valid && fmt_same(g->src, src) && fmt_same(g->dst, dst) && !std::memcmp(&g->opts_copy, &c->opts, sizeof(SwsContext))) {
        g->src.hip_stream = src.hip_stream; g->src.device_ref = src.device_ref;     // (this call's frames: frames_stream() reads them)
        g->dst.hip_stream = dst.hip_stream; g->dst.device_ref = dst.device_ref;
        return 0;
    }
    graph_free(g);
    g->src = src; g->dst = dst; g->opts_copy = c->opts;
    bool inc = false;
    if (needs_color_mapping(src, dst, &inc)) {
        log_msg(c, 0, "conversion between different primaries / transfer characteristics / mastering metadata needs the reference's "
                      "3-D LUT pre-pass, which libswscale_hip does not implement\n");
        return SWS_AVERROR(ENOTSUP);
    }
    g->incomplete = inc;
    if (fmt_equal(src, dst)) g->noop = true;     // "No passes were added, so no operations were necessary": a plane copy
    else {
        int ret = build_legacy(c, src, dst, g);
        if (ret < 0) { graph_free(g); return ret; }
    }
    g->valid = true;
    return 0;
}

// the checks of sws_frame_setup() on hardware frames (swscale.c:1515-1538), with HIP in Vulkan's place
static int check_hw_frames(const SwsFrameView *dst, const SwsFrameView *src)
{
    if (!!src->hw_frames_ctx != !!dst->hw_frames_ctx) return SWS_AVERROR(ENOTSUP);   // "if a single frame has a context, then both need a context"
    if (!src->hw_frames_ctx) return 0;
    if (!src->data[0] || !dst->data[0]) return SWS_AVERROR(EINVAL);                    // both hardware frames must already be allocated
    const SwsHWFramesContext *sf = (const SwsHWFramesContext *)src->hw_frames_ctx->data, *df = (const SwsHWFramesContext *)dst->hw_frames_ctx->data;
    if (!sf || !df || !sf->device_ctx || !df->device_ctx) return SWS_AVERROR(EINVAL);
    if (sf->device_ctx != df->device_ctx) return SWS_AVERROR(EINVAL);                  // both frames must live on the same device
    if (sf->device_ctx->type != AV_HWDEVICE_TYPE_HIP) return SWS_AVERROR(ENOTSUP);     // only HIP devices
    return 0;
}

static int dynamic_setup(SwsInternal *c, const SwsFrameView *dst, const SwsFrameView *src)
{
    int ret = validate_params(&c->opts);
    if (ret < 0) return ret;
    if ((ret = check_hw_frames(dst, src)) < 0) return ret;
    for (int field = 0; field < 2; field++) {
        SwsFmt sf, df;
        if ((ret = fmt_from_frame(src, field, &sf)) < 0 || (ret = fmt_from_frame(dst, field, &df)) < 0) { frames_release(c); return ret; }
        if ((src->flags ^ dst->flags) & SWS_FRAME_FLAG_INTERLACED) {
            log_msg(c, 0, "Cannot convert interlaced to progressive frames or vice versa.\n");
            frames_release(c);
            return SWS_AVERROR(EINVAL);
        }
        const bool src_ok = test_fmt(sf, 0), dst_ok = test_fmt(df, 1);
        if ((!src_ok || !dst_ok) && !(pix_desc(sf.format) && pix_desc(df.format) && fmt_equal(sf, df))) {
            log_msg(c, 0, "%s\n", src_ok ? "Unsupported output" : "Unsupported input");
            frames_release(c);
            return SWS_AVERROR(ENOTSUP);
        }
        ret = graph_reinit(c, field, sf, df);
        if (ret < 0) { frames_release(c); return ret; }
        if (c->graph[field].incomplete && (c->opts.flags & SWS_STRICT)) {
            log_msg(c, 0, "Incomplete scaling graph\n");
            frames_release(c);
            return SWS_AVERROR(EINVAL);
        }
        if (!sf.interlaced) { graph_free(&c->graph[1]); break; }
    }
    return 0;
}

// the software format a frame is matched against a legacy context with
static int frame_sw_format(const SwsFrameView *f)
{
    if (f->hw_frames_ctx && f->hw_frames_ctx->data) return ((const SwsHWFramesContext *)f->hw_frames_ctx->data)->sw_format;
    return f->format;
}

static bool frame_matches(const SwsInternal *c, const SwsFrameView *f, bool is_src)
{
    const int raw = frame_sw_format(f);
    const int fmt = canonical_pix_fmt(raw);     // the context stores canonicalised formats (yuvj420p -> yuv420p, bgr0 -> bgra)
    const SwsContext &o = c->opts;
    if (!(is_src ? src_tags_match(c, raw) : dst_tags_match(c, raw))) return false;
    return is_src ? (fmt == o.src_format && f->width == o.src_w && f->height == o.src_h)
                  : (fmt == o.dst_format && f->width == o.dst_w && f->height == o.dst_h);
}

// get_field(), graph.c:998-1024: the view of one field of an interlaced frame
static void field_view(SwsFrameView *v, const SwsFrameView *f, const SwsFmt &fmt)
{
    *v = *f;
    v->format = fmt.format;
    v->hw_frames_ctx = nullptr;                 // below this point the pointers speak for themselves
    if (!(f->flags & SWS_FRAME_FLAG_INTERLACED)) return;
    if (fmt.field == 1) for (int i = 0; i < 4; i++) if (v->data[i]) v->data[i] += v->linesize[i];   // odd rows
    for (int i = 0; i < 4; i++) v->linesize[i] <<= 1;
    v->height = (v->height + (fmt.field == 0)) >> 1;
}

// stream of the frames' HIP device context, if they have one: conversions are ordered after the uploads queued on it
static void *frames_stream(const FrameGraph &g, bool *have)
{
    *have = g.src.hip_device >= 0 && g.src.hip_stream;
    return *have ? g.src.hip_stream : nullptr;
}

static int run_graphs(SwsInternal *c, SwsFrameView *const dst[], const SwsFrameView *const src[], int n)
{
    const int nfields = c->graph[1].valid ? 2 : 1;
    std::vector<SwsFrameView> sv((size_t)n), dv((size_t)n);
    std::vector<const SwsFrameView *> sp((size_t)n);
    std::vector<SwsFrameView *> dp((size_t)n);
    for (int field = 0; field < nfields; field++) {
        FrameGraph &g = c->graph[field];
        for (int i = 0; i < n; i++) {
            field_view(&sv[(size_t)i], src[i], g.src); field_view(&dv[(size_t)i], dst[i], g.dst);
            sp[(size_t)i] = &sv[(size_t)i]; dp[(size_t)i] = &dv[(size_t)i];
        }
        bool have; void *st = frames_stream(g, &have);
        int ret;
        if (g.noop) {
            for (int i = 0; i < n; i++) if ((ret = dev_copy_frame(c, dp[(size_t)i], sp[(size_t)i], have, st)) < 0) return ret;
            continue;
        }
        SwsInternal *lc = internal(g.legacy);
        if ((ret = dev_inherit(lc, c, have, st, have ? g.src.hip_device : -1)) < 0) return ret;   // HIP frames: on THEIR GPU, ordered on their device context's stream
        for (int i = 0; i < n; i++)
            if (!check_image_pointers(sp[(size_t)i]->data, lc->opts.src_format, sp[(size_t)i]->linesize) ||
                !check_image_pointers(dp[(size_t)i]->data, lc->opts.dst_format, dp[(size_t)i]->linesize)) return SWS_AVERROR(EINVAL);
        StreamLoan loan;
        if (have && (ret = dev_borrow_stream(lc, st, &loan)) < 0) return ret;
        ret = dev_run(lc, nullptr, nullptr, 0, lc->opts.src_h, nullptr, nullptr, n, sp.data(), dp.data());
        dev_return_stream(lc, loan);
        if (ret < 0) return ret;
    }
    return 0;
}

} // namespace swship

using namespace swship;

extern "C" {

int sws_test_frame(const SwsFrameView *frame, int output)   // format.c:695-706
{
    if (!frame) return 0;
    for (int field = 0; field < 2; field++) {
        SwsFmt fmt;
        if (fmt_from_frame(frame, field, &fmt) < 0 || !test_fmt(fmt, output)) return 0;
        if (!fmt.interlaced) break;
    }
    return 1;
}

int sws_is_noop(const SwsFrameView *dst, const SwsFrameView *src)   // format.c:708-721
{
    if (!dst || !src) return 0;
    for (int field = 0; field < 2; field++) {
        SwsFmt d, s;
        if (fmt_from_frame(dst, field, &d) < 0 || fmt_from_frame(src, field, &s) < 0 || !fmt_equal(d, s)) return 0;
        if (!d.interlaced) break;
    }
    return 1;
}

int sws_frame_setup(SwsContext *sws, const SwsFrameView *dst, const SwsFrameView *src)   // swscale.c:1503-1619
{
    if (!sws || !dst || !src) return SWS_AVERROR(EINVAL);
    SwsInternal *c = internal(sws);
    if (!c->legacy_init) return dynamic_setup(c, dst, src);
    // a context built for one conversion: the frames have to be that conversion's (the reference does not look, its behaviour
    // on a mismatch is undefined)
    int r = check_hw_frames(dst, src);
    if (r < 0) return r;
    return frame_matches(c, src, true) && frame_matches(c, dst, false) ? 0 : SWS_AVERROR(EINVAL);
}

int sws_scale_frame(SwsContext *sws, SwsFrameView *dstf, const SwsFrameView *srcf)   // swscale.c:1404-1480
{
    if (!sws || !dstf || !srcf) return SWS_AVERROR(EINVAL);
    SwsInternal *c = internal(sws);
    if (c->legacy_init) {
        // "Context has been initialized with explicit values, fall back to legacy API behavior": sws_frame_start() +
        // sws_send_slice(0, src->height) + sws_receive_slice(0, dst->height), the return value is the one of the scaler
        int r = sws_frame_setup(sws, dstf, srcf);
        if (r < 0) return r;
        if (!dstf->data[0]) return SWS_AVERROR(ENOMEM);   // no libavutil here to allocate AVFrame buffers with: bring your own
        SwsFrameView sv = *srcf, dv = *dstf;
        const SwsFrameView *s1[1] = { &sv };
        SwsFrameView *d1[1] = { &dv };
        StreamLoan loan;
        if (srcf->hw_frames_ctx) {
            SwsFmt f; (void)fmt_from_frame(srcf, 0, &f);
            // the frames' AVHIPDeviceContext: the conversion runs on that GPU (it becomes the context's home) and on that stream, after the
            // uploads queued there and before whatever the caller queues next
            if (f.hip_stream && f.hip_device >= 0 && (r = sws_hip_set_device(sws, f.hip_device)) < 0) return r;
            if (f.hip_stream && (r = dev_borrow_stream(c, f.hip_stream, &loan)) < 0) return r;
        }
        r = dev_run(c, nullptr, nullptr, 0, sws->src_h, nullptr, nullptr, 1, s1, d1);
        dev_return_stream(c, loan);
        return r < 0 ? r : sws->dst_h;
    }
    int ret = sws_frame_setup(sws, dstf, srcf);
    if (ret < 0) return ret;
    if (!srcf->data[0]) return 0;
    if (!dstf->data[0]) return SWS_AVERROR(ENOMEM);       // (the reference allocates from its frame pool here)
    const SwsFrameView *s1[1] = { srcf };
    SwsFrameView *d1[1] = { dstf };
    return run_graphs(c, d1, s1, 1);
}

int sws_scale_frames(SwsContext *sws, SwsFrameView *const dst[], const SwsFrameView *const src[], int nb_frames)
{
    if (!sws || !dst || !src || nb_frames < 0) return SWS_AVERROR(EINVAL);
    if (!nb_frames) return 0;
    SwsInternal *c = internal(sws);
    for (int i = 0; i < nb_frames; i++) if (!src[i] || !dst[i]) return SWS_AVERROR(EINVAL);
    if (c->legacy_init) {
        for (int i = 0; i < nb_frames; i++) {
            int r = check_hw_frames(dst[i], src[i]);
            if (r < 0) return r;
            if (!frame_matches(c, src[i], true) || !frame_matches(c, dst[i], false)) return SWS_AVERROR(EINVAL);
            if (!check_image_pointers(src[i]->data, sws->src_format, src[i]->linesize) ||
                !check_image_pointers(dst[i]->data, sws->dst_format, dst[i]->linesize)) return SWS_AVERROR(EINVAL);
        }
        StreamLoan loan;
        if (src[0]->hw_frames_ctx) {
            SwsFmt f; (void)fmt_from_frame(src[0], 0, &f);
            int r;
            if (f.hip_stream && f.hip_device >= 0 && (r = sws_hip_set_device(sws, f.hip_device)) < 0) return r;
            if (f.hip_stream && (r = dev_borrow_stream(c, f.hip_stream, &loan)) < 0) return r;
        }
        const int rr = dev_run(c, nullptr, nullptr, 0, sws->src_h, nullptr, nullptr, nb_frames, src, dst);
        dev_return_stream(c, loan);
        return rr;
    }
    // dynamic: one conversion for the batch, configured from the first pair; every other pair must have the same properties
    int ret = sws_frame_setup(sws, dst[0], src[0]);
    if (ret < 0) return ret;
    const int nfields = c->graph[1].valid ? 2 : 1;
    for (int i = 1; i < nb_frames; i++) {
        if ((ret = check_hw_frames(dst[i], src[i])) < 0) return ret;
        for (int field = 0; field < nfields; field++) {
            SwsFmt sf, df;
            if ((ret = fmt_from_frame(src[i], field, &sf)) < 0 || (ret = fmt_from_frame(dst[i], field, &df)) < 0) return ret;
            if (!fmt_equal(sf, c->graph[field].src) || !fmt_equal(df, c->graph[field].dst) ||
                sf.hw_format != c->graph[field].src.hw_format || df.hw_format != c->graph[field].dst.hw_format) return SWS_AVERROR(EINVAL);
        }
    }
    for (int i = 0; i < nb_frames; i++) if (!src[i]->data[0] || !dst[i]->data[0]) return SWS_AVERROR(EINVAL);
    ret = run_graphs(c, dst, src, nb_frames);
    return ret < 0 ? ret : nb_frames;
}

// ---- slice API (swscale.c:1305-1402): legacy-initialised contexts only ----
int sws_frame_start(SwsContext *sws, SwsFrameView *dst, const SwsFrameView *src)
{
    if (!sws || !dst || !src) return SWS_AVERROR(EINVAL);
    SwsInternal *c = internal(sws);
    if (!c->legacy_init) return SWS_AVERROR(EINVAL);
    int r = sws_frame_setup(sws, dst, src);
    if (r < 0) return r;
    if (!dst->data[0]) return SWS_AVERROR(ENOMEM);   // this library cannot allocate AVFrame buffers (no libavutil): bring your own
    c->frame_src = src; c->frame_dst = dst; c->frame_ranges.clear(); c->frame_done = false;
    return 0;
}

void sws_frame_end(SwsContext *sws)   // swscale.c:1236-1244
{
    if (!sws) return;
    SwsInternal *c = internal(sws);
    c->frame_src = nullptr; c->frame_dst = nullptr; c->frame_ranges.clear(); c->frame_done = false;
}

// sws_send_slice (swscale.c:1337-1351) only records which source rows are there: ff_range_add (utils.c:2384-2440) keeps a sorted list of
// disjoint ranges, refuses a slice that overlaps one it has and merges neighbours; the conversion happens in sws_receive_slice() once the
// list is the single range [0, src_h).
int sws_send_slice(SwsContext *sws, unsigned int slice_start, unsigned int slice_height)
{
    if (!sws) return SWS_AVERROR(EINVAL);
    SwsInternal *c = internal(sws);
    if (!c->legacy_init || !c->frame_src || !c->frame_dst) return SWS_AVERROR(EINVAL);
    auto &rl = c->frame_ranges;   // pairs {start, len}
    size_t idx = 0;
    while (idx < rl.size() && rl[idx].first <= slice_start) idx++;
    if (idx > 0 && (uint64_t)rl[idx - 1].first + rl[idx - 1].second > slice_start) return SWS_AVERROR(EINVAL);
    if (idx < rl.size() && (uint64_t)slice_start + slice_height > rl[idx].first) return SWS_AVERROR(EINVAL);
    rl.insert(rl.begin() + (ptrdiff_t)idx, std::make_pair(slice_start, slice_height));
    if (idx > 0 && rl[idx - 1].first + rl[idx - 1].second == rl[idx].first) { rl[idx - 1].second += rl[idx].second; rl.erase(rl.begin() + (ptrdiff_t)idx); idx--; }
    if (idx + 1 < rl.size() && rl[idx].first + rl[idx].second == rl[idx + 1].first) { rl[idx].second += rl[idx + 1].second; rl.erase(rl.begin() + (ptrdiff_t)idx + 1); }
    return 0;
}

unsigned int sws_receive_slice_alignment(const SwsContext *sws)
{
    if (!sws) return 1;
    const SwsInternal *c = (const SwsInternal *)sws;
    return c->dst_slice_align > 0 ? (unsigned)c->dst_slice_align : 1u;
}

int sws_receive_slice(SwsContext *sws, unsigned int slice_start, unsigned int slice_height)   // swscale.c:1362-1402
{
    if (!sws) return SWS_AVERROR(EINVAL);
    SwsInternal *c = internal(sws);
    if (!c->legacy_init || !c->frame_src || !c->frame_dst) return SWS_AVERROR(EINVAL);
    // "wait until complete input has been received"
    if (!(c->frame_ranges.size() == 1 && c->frame_ranges[0].first == 0 && c->frame_ranges[0].second == (unsigned)sws->src_h)) return SWS_AVERROR(EAGAIN);
    const unsigned align = sws_receive_slice_alignment(sws);
    if ((slice_start > 0 || slice_height < (unsigned)sws->dst_h) && (slice_start % align || slice_height % align)) return SWS_AVERROR(EINVAL);
    if ((uint64_t)slice_start + slice_height > (unsigned)sws->dst_h) return SWS_AVERROR(EINVAL);   // (scale_internal's destination slice check, :1053-1058)
    // The reference scales the whole source into the requested destination rows on every call (scale_internal with a destination slice).
    // Here the frame is converted once, on the first call after the input is complete; later calls find their rows already there.
    if (!c->frame_done) {
        const SwsFrameView *s = c->frame_src;
        int r = sws_scale(sws, (const uint8_t *const *)s->data, s->linesize, 0, sws->src_h, (uint8_t *const *)c->frame_dst->data, c->frame_dst->linesize);
        if (r < 0) return r;
        c->frame_done = true;
    }
    return c->plan == PLAN_UNSC_ALPHABLEND ? 0 : (int)slice_height;   // what scale_internal returns for the slice (ff_sws_alphablendaway: 0)
}

} // extern "C"
