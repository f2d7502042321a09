/*
 * vf_scale_hip.c -- "scale_hip": scaling and pixel format conversion of AV_PIX_FMT_HIP frames, HBM in, HBM out.
 *
 * Drop-in for libavfilter/ (the place libavfilter/vf_scale_cuda.c has for CUDA).  Where scale_cuda brings its own
 * resize kernels, this filter is a thin caller of the swscale API: one sws_alloc_context()ed context, one
 * sws_scale_frame() per frame (libswscale/swscale.h:439), with libswscale_hip.so behind it.  Everything the
 * conversion depends on travels in the AVFrame properties (color_range, colorspace, chroma_location, interlacing),
 * exactly as for the software "scale" filter's sws_scale_frame() path (libavfilter/vf_scale.c).
 *
 * The file has two parts:
 *   1. the core (ScaleHIPCore): output pool + per-frame conversion.  Plain libavutil + swscale calls; compiled and
 *      tested inside this repository against integration/shim/avutil_min.h (-DHWCONTEXT_HIP_STANDALONE).
 *   2. the AVFilter wrapper (options, link configuration, filter_frame), only built inside the libavfilter tree.
 */
#ifdef HWCONTEXT_HIP_STANDALONE
#include "shim/avutil_min.h"
#include "hwcontext_hip.h"
#else
#include "libavutil/common.h"
#include "libavutil/hwcontext.h"
#include "libavutil/hwcontext_hip.h"
#include "libavutil/internal.h"
#include "libavutil/opt.h"
#include "libavutil/pixdesc.h"
#include "libswscale/swscale.h"
#include "avfilter.h"
#include "filters.h"
#include "scale_eval.h"
#include "video.h"
#endif

/* ------------------------------------------------------------------------------------------------------------
 * 1. core
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct ScaleHIPCore {
    SwsContext  *sws;            /* sws_alloc_context()ed: configured by the frames themselves */
    AVBufferRef *frames_ctx;     /* output pool on the input's device */
    int w, h;
    enum AVPixelFormat out_fmt;  /* software format of the output frames */
    /* output properties; -1 = same as the input frame */
    int out_range, out_color_matrix, out_chroma_loc;
    int passthrough;             /* hand the input frame on when the conversion would change nothing */
} ScaleHIPCore;

void scale_hip_core_uninit(ScaleHIPCore *s)
{
    sws_free_context(&s->sws);
    av_buffer_unref(&s->frames_ctx);
}

/* in_frames_ref: the hw_frames_ctx of the input link.  out_fmt == AV_PIX_FMT_NONE keeps the input's software format. */
int scale_hip_core_init(ScaleHIPCore *s, AVBufferRef *in_frames_ref, int out_w, int out_h,
                        enum AVPixelFormat out_fmt, unsigned sws_flags)
{
    AVHWFramesContext *in_ctx, *out_ctx;
    AVBufferRef *out_ref;
    int ret;

    if (!in_frames_ref)
        return AVERROR(EINVAL);
    in_ctx = (AVHWFramesContext *)in_frames_ref->data;
    if (in_ctx->format != AV_PIX_FMT_HIP)
        return AVERROR(EINVAL);
    if (out_fmt == AV_PIX_FMT_NONE)
        out_fmt = in_ctx->sw_format;
    if (!sws_test_format(in_ctx->sw_format, 0) || !sws_test_format(out_fmt, 1))
        return AVERROR(ENOSYS);

    out_ref = av_hwframe_ctx_alloc(in_ctx->device_ref);
    if (!out_ref)
        return AVERROR(ENOMEM);
    out_ctx = (AVHWFramesContext *)out_ref->data;
    out_ctx->format    = AV_PIX_FMT_HIP;
    out_ctx->sw_format = out_fmt;
    out_ctx->width     = out_w;
    out_ctx->height    = out_h;
    ret = av_hwframe_ctx_init(out_ref);
    if (ret < 0) {
        av_buffer_unref(&out_ref);
        return ret;
    }

    scale_hip_core_uninit(s);
    s->frames_ctx = out_ref;
    s->w = out_w;
    s->h = out_h;
    s->out_fmt = out_fmt;
    s->sws = sws_alloc_context();
    if (!s->sws) {
        scale_hip_core_uninit(s);
        return AVERROR(ENOMEM);
    }
    s->sws->flags = sws_flags;
    return 0;
}

/* 1 if `in` can be passed on untouched, 0 if `out` now holds the converted picture, negative AVERROR on failure.
 * `out` must be a clean frame (av_frame_alloc() / av_frame_unref()). */
int scale_hip_core_frame(ScaleHIPCore *s, AVFrame *out, const AVFrame *in)
{
    int ret;

    ret = av_hwframe_get_buffer(s->frames_ctx, out, 0);
    if (ret < 0)
        return ret;
    ret = av_frame_copy_props(out, in);
    if (ret < 0)
        goto fail;
    out->width  = s->w;
    out->height = s->h;
    if (s->out_range >= 0)
        out->color_range = s->out_range;
    if (s->out_color_matrix >= 0)
        out->colorspace = s->out_color_matrix;
    if (s->out_chroma_loc >= 0)
        out->chroma_location = s->out_chroma_loc;

    if (s->passthrough && sws_is_noop(out, in)) {
        av_frame_unref(out);
        return 1;
    }
    /* queued on the stream of the frames' AVHIPDeviceContext: ordered after the upload / decode that produced `in`,
     * before whatever the next filter queues on the same stream */
    ret = sws_scale_frame(s->sws, out, in);
    if (ret < 0)
        goto fail;
    return 0;
fail:
    av_frame_unref(out);
    return ret;
}

#ifndef HWCONTEXT_HIP_STANDALONE
/* ------------------------------------------------------------------------------------------------------------
 * 2. AVFilter wrapper
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct ScaleHIPContext {
    const AVClass *class;
    ScaleHIPCore core;
    char *w_expr, *h_expr;
    enum AVPixelFormat format;
    unsigned sws_flags;
    int out_range, out_color_matrix, out_chroma_loc;
    int passthrough;
    int force_original_aspect_ratio, force_divisible_by, reset_sar;
} ScaleHIPContext;

static av_cold void scale_hip_uninit(AVFilterContext *ctx)
{
    ScaleHIPContext *s = ctx->priv;
    scale_hip_core_uninit(&s->core);
}

static av_cold int scale_hip_config_props(AVFilterLink *outlink)
{
    AVFilterContext *ctx = outlink->src;
    AVFilterLink *inlink = ctx->inputs[0];
    FilterLink *inl = ff_filter_link(inlink), *outl = ff_filter_link(outlink);
    ScaleHIPContext *s = ctx->priv;
    double w_adj = 1.0;
    int w, h, ret;

    if (!inl->hw_frames_ctx) {
        av_log(ctx, AV_LOG_ERROR, "No hw context provided on input\n");
        return AVERROR(EINVAL);
    }
    if ((ret = ff_scale_eval_dimensions(s, s->w_expr, s->h_expr, inlink, outlink, &w, &h)) < 0)
        return ret;
    if (s->reset_sar)
        w_adj = inlink->sample_aspect_ratio.num ?
                (double)inlink->sample_aspect_ratio.num / inlink->sample_aspect_ratio.den : 1;
    if ((ret = ff_scale_adjust_dimensions(inlink, &w, &h, s->force_original_aspect_ratio, s->force_divisible_by, w_adj)) < 0)
        return ret;
    outlink->w = w;
    outlink->h = h;

    s->core.out_range = s->out_range;
    s->core.out_color_matrix = s->out_color_matrix;
    s->core.out_chroma_loc = s->out_chroma_loc;
    s->core.passthrough = s->passthrough;
    ret = scale_hip_core_init(&s->core, inl->hw_frames_ctx, w, h, s->format, s->sws_flags);
    if (ret < 0)
        return ret;
    outl->hw_frames_ctx = av_buffer_ref(s->core.frames_ctx);
    if (!outl->hw_frames_ctx)
        return AVERROR(ENOMEM);

    if (s->reset_sar)
        outlink->sample_aspect_ratio = (AVRational){ 1, 1 };
    else if (inlink->sample_aspect_ratio.num)
        outlink->sample_aspect_ratio = av_mul_q((AVRational){ outlink->h * inlink->w, outlink->w * inlink->h },
                                                inlink->sample_aspect_ratio);
    else
        outlink->sample_aspect_ratio = inlink->sample_aspect_ratio;
    return 0;
}

static int scale_hip_filter_frame(AVFilterLink *link, AVFrame *in)
{
    AVFilterContext *ctx = link->dst;
    AVFilterLink *outlink = ctx->outputs[0];
    ScaleHIPContext *s = ctx->priv;
    AVFrame *out = av_frame_alloc();
    int ret;

    if (!out) {
        av_frame_free(&in);
        return AVERROR(ENOMEM);
    }
    ret = scale_hip_core_frame(&s->core, out, in);
    if (ret == 1) {                      /* nothing to do: pass the input on */
        av_frame_free(&out);
        return ff_filter_frame(outlink, in);
    }
    if (ret < 0) {
        av_frame_free(&in);
        av_frame_free(&out);
        return ret;
    }
    if (s->reset_sar)
        out->sample_aspect_ratio = (AVRational){ 1, 1 };
    else
        av_reduce(&out->sample_aspect_ratio.num, &out->sample_aspect_ratio.den,
                  (int64_t)in->sample_aspect_ratio.num * outlink->h * link->w,
                  (int64_t)in->sample_aspect_ratio.den * outlink->w * link->h, INT_MAX);
    if (out->width != in->width || out->height != in->height)
        av_frame_side_data_remove_by_props(&out->side_data, &out->nb_side_data, AV_SIDE_DATA_PROP_SIZE_DEPENDENT);
    av_frame_free(&in);
    return ff_filter_frame(outlink, out);
}

#define OFFSET(x) offsetof(ScaleHIPContext, x)
#define FLAGS (AV_OPT_FLAG_FILTERING_PARAM | AV_OPT_FLAG_VIDEO_PARAM)
static const AVOption scale_hip_options[] = {
    { "w", "Output video width",  OFFSET(w_expr), AV_OPT_TYPE_STRING, { .str = "iw" }, .flags = FLAGS },
    { "h", "Output video height", OFFSET(h_expr), AV_OPT_TYPE_STRING, { .str = "ih" }, .flags = FLAGS },
    { "format", "Output video pixel format", OFFSET(format), AV_OPT_TYPE_PIXEL_FMT, { .i64 = AV_PIX_FMT_NONE }, INT_MIN, INT_MAX, FLAGS },
    { "flags", "swscale flags (scaler, accurate_rnd, bitexact, ...)", OFFSET(sws_flags), AV_OPT_TYPE_FLAGS, { .i64 = SWS_BICUBIC }, 0, UINT_MAX, FLAGS },
    { "out_range", "Output color range (-1 = input's)", OFFSET(out_range), AV_OPT_TYPE_INT, { .i64 = -1 }, -1, AVCOL_RANGE_NB - 1, FLAGS },
    { "out_color_matrix", "Output YCbCr matrix (-1 = input's)", OFFSET(out_color_matrix), AV_OPT_TYPE_INT, { .i64 = -1 }, -1, AVCOL_SPC_NB - 1, FLAGS },
    { "out_chroma_loc", "Output chroma sample location (-1 = input's)", OFFSET(out_chroma_loc), AV_OPT_TYPE_INT, { .i64 = -1 }, -1, AVCHROMA_LOC_NB - 1, FLAGS },
    { "passthrough", "Do not process frames at all if parameters match", OFFSET(passthrough), AV_OPT_TYPE_BOOL, { .i64 = 1 }, 0, 1, FLAGS },
    { "force_original_aspect_ratio", "decrease or increase w/h if necessary to keep the original AR", OFFSET(force_original_aspect_ratio), AV_OPT_TYPE_INT, { .i64 = 0 }, 0, SCALE_FORCE_OAR_NB - 1, FLAGS },
    { "force_divisible_by", "enforce that the output resolution is divisible by a defined integer", OFFSET(force_divisible_by), AV_OPT_TYPE_INT, { .i64 = 1 }, 1, 256, FLAGS },
    { "reset_sar", "reset SAR to 1 and scale to square pixels if scaling proportionally", OFFSET(reset_sar), AV_OPT_TYPE_BOOL, { .i64 = 0 }, 0, 1, FLAGS },
    { NULL },
};

static const AVClass scale_hip_class = {
    .class_name = "scale_hip",
    .item_name  = av_default_item_name,
    .option     = scale_hip_options,
    .version    = LIBAVUTIL_VERSION_INT,
};

static const AVFilterPad scale_hip_inputs[] = {
    { .name = "default", .type = AVMEDIA_TYPE_VIDEO, .filter_frame = scale_hip_filter_frame },
};

static const AVFilterPad scale_hip_outputs[] = {
    { .name = "default", .type = AVMEDIA_TYPE_VIDEO, .config_props = scale_hip_config_props },
};

const FFFilter ff_vf_scale_hip = {
    .p.name         = "scale_hip",
    .p.description  = NULL_IF_CONFIG_SMALL("Scale and convert AMD GPU (HIP) frames with libswscale_hip."),
    .p.priv_class   = &scale_hip_class,
    .uninit         = scale_hip_uninit,
    .priv_size      = sizeof(ScaleHIPContext),
    FILTER_INPUTS(scale_hip_inputs),
    FILTER_OUTPUTS(scale_hip_outputs),
    FILTER_SINGLE_PIXFMT(AV_PIX_FMT_HIP),
    .flags_internal = FF_FILTER_FLAG_HWFRAME_AWARE,
};
#endif /* !HWCONTEXT_HIP_STANDALONE */
