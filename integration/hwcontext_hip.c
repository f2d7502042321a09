/*
 * hwcontext_hip.c -- AV_HWDEVICE_TYPE_HIP: AMD GPU memory as a libavutil hardware frames pool.
 *
 * Drop-in for libavutil/ (next to hwcontext_cuda.c, whose role it plays for HIP): fills the
 * HWContextType vtable of libavutil/hwcontext_internal.h:29-91.  Every device operation goes through
 * the C ABI of libswscale_hip.so (include/hwcontext_hip.h): this file needs no HIP header and no
 * HIP compiler, only -lswscale_hip at link time.
 *
 * Frames of this type: format AV_PIX_FMT_HIP, one hipMalloc()ed block per frame with the planes of
 * AVHWFramesContext.sw_format at 256-byte aligned offsets, linesize[] aligned to 256 bytes (the layout
 * libswscale_hip's kernels issue whole 16-byte vector accesses on).  Transfers are hipMemcpy2DAsync on
 * AVHIPDeviceContext.stream; like hwcontext_cuda.c:642-646 only a download to a software frame waits.
 *
 * In-repo build (tests): -DHWCONTEXT_HIP_STANDALONE compiles against integration/shim/avutil_min.h, a
 * declaration-only stand-in for the libavutil headers named below.
 */
#ifdef HWCONTEXT_HIP_STANDALONE
#include "shim/avutil_min.h"
#else
#include "buffer.h"
#include "common.h"
#include "hwcontext.h"
#include "hwcontext_internal.h"
#include "mem.h"
#include "pixdesc.h"
#include "pixfmt.h"
#endif
#include "hwcontext_hip.h"

#include <stdlib.h>
#include <string.h>

#define HIP_FRAME_ALIGN 256

typedef struct HIPDeviceContext {
    AVHIPDeviceContext p;       /* the public struct comes first (hwcontext_cuda.c:38-41 idiom) */
    int stream_owned;           /* device_create made the stream: device_uninit destroys it */
} HIPDeviceContext;

typedef struct HIPFramesContext {
    int    linesize[4];
    size_t offset[4];
    size_t size;
    int    nb_planes;
} HIPFramesContext;

static int hip_frames_get_constraints(AVHWDeviceContext *ctx, const void *hwconfig,
                                      AVHWFramesConstraints *constraints)
{
    int n = 0;

    constraints->valid_hw_formats = av_malloc_array(2, sizeof(*constraints->valid_hw_formats));
    if (!constraints->valid_hw_formats)
        return AVERROR(ENOMEM);
    constraints->valid_hw_formats[0] = AV_PIX_FMT_HIP;
    constraints->valid_hw_formats[1] = AV_PIX_FMT_NONE;

    /* every layout the converter behind these frames reads and writes */
    constraints->valid_sw_formats = av_malloc_array(AV_PIX_FMT_HIP + 1, sizeof(*constraints->valid_sw_formats));
    if (!constraints->valid_sw_formats)
        return AVERROR(ENOMEM);
    for (int i = 0; i < AV_PIX_FMT_HIP; i++)
        if (sws_hip_frames_format_supported(i))
            constraints->valid_sw_formats[n++] = i;
    constraints->valid_sw_formats[n] = AV_PIX_FMT_NONE;
    return 0;
}

static void hip_buffer_free(void *opaque, uint8_t *data)
{
    AVHWFramesContext *ctx = opaque;
    AVHIPDeviceContext *hwctx = ctx->device_ctx->hwctx;

    /* frames handed out earlier may still be read by work queued on the device's stream */
    sws_hip_stream_sync(hwctx->device, hwctx->stream);
    sws_hip_mem_free(hwctx->device, data);
}

static AVBufferRef *hip_pool_alloc(void *opaque, size_t size)
{
    AVHWFramesContext *ctx = opaque;
    AVHIPDeviceContext *hwctx = ctx->device_ctx->hwctx;
    AVBufferRef *ref;
    void *data = NULL;

    if (sws_hip_mem_alloc(hwctx->device, size, &data) < 0)
        return NULL;
    ref = av_buffer_create(data, size, hip_buffer_free, ctx, 0);
    if (!ref)
        sws_hip_mem_free(hwctx->device, data);
    return ref;
}

static int hip_frames_init(AVHWFramesContext *ctx)
{
    HIPFramesContext *priv = ctx->hwctx;
    int ret;

    if (!sws_hip_frames_format_supported(ctx->sw_format)) {
        av_log(ctx, AV_LOG_ERROR, "Pixel format %d is not supported by the HIP converter\n", ctx->sw_format);
        return AVERROR(ENOSYS);
    }
    ret = sws_hip_image_layout(ctx->sw_format, ctx->width, ctx->height, HIP_FRAME_ALIGN,
                               priv->linesize, priv->offset, &priv->size);
    if (ret < 0)
        return ret;
    priv->nb_planes = 0;
    for (int i = 0; i < 4; i++)
        if (priv->linesize[i])
            priv->nb_planes = i + 1;

    if (!ctx->pool) {
        ffhwframesctx(ctx)->pool_internal = av_buffer_pool_init2(priv->size, ctx, hip_pool_alloc, NULL);
        if (!ffhwframesctx(ctx)->pool_internal)
            return AVERROR(ENOMEM);
    }
    return 0;
}

static int hip_get_buffer(AVHWFramesContext *ctx, AVFrame *frame)
{
    HIPFramesContext *priv = ctx->hwctx;

    frame->buf[0] = av_buffer_pool_get(ctx->pool);
    if (!frame->buf[0])
        return AVERROR(ENOMEM);
    for (int i = 0; i < priv->nb_planes; i++) {
        frame->data[i]     = frame->buf[0]->data + priv->offset[i];
        frame->linesize[i] = priv->linesize[i];
    }
    frame->format = AV_PIX_FMT_HIP;
    frame->width  = ctx->width;
    frame->height = ctx->height;
    return 0;
}

static int hip_transfer_get_formats(AVHWFramesContext *ctx, enum AVHWFrameTransferDirection dir,
                                    enum AVPixelFormat **formats)
{
    enum AVPixelFormat *fmts = av_malloc_array(2, sizeof(*fmts));

    if (!fmts)
        return AVERROR(ENOMEM);
    fmts[0] = ctx->sw_format;
    fmts[1] = AV_PIX_FMT_NONE;
    *formats = fmts;
    return 0;
}

/* transfer_data_to and transfer_data_from: the copy direction follows from where the pointers live */
static int hip_transfer_data(AVHWFramesContext *ctx, AVFrame *dst, const AVFrame *src)
{
    HIPFramesContext *priv = ctx->hwctx;
    AVHIPDeviceContext *hwctx = ctx->device_ctx->hwctx;
    int ret = 0, queued = 0;

    if (src->width > ctx->width || src->height > ctx->height || dst->width != src->width || dst->height != src->height)
        return AVERROR(EINVAL);
    for (int i = 0; i < priv->nb_planes && src->data[i] && dst->data[i]; i++) {
        int bytewidth, rows;

        ret = sws_hip_plane_geometry(ctx->sw_format, src->width, src->height, i, &bytewidth, &rows);
        if (ret < 0)
            break;
        ret = sws_hip_copy_plane(hwctx->device, hwctx->stream, dst->data[i], dst->linesize[i],
                                 src->data[i], src->linesize[i], bytewidth, rows);
        if (ret < 0)
            break;
        queued = 1;
    }
    /* a software destination must be complete on return; so must a failed transfer's source */
    if (queued && (!dst->hw_frames_ctx || ret < 0)) {
        int err = sws_hip_stream_sync(hwctx->device, hwctx->stream);
        if (ret >= 0)
            ret = err;
    }
    return ret;
}

static void hip_device_uninit(AVHWDeviceContext *device_ctx)
{
    HIPDeviceContext *hwctx = device_ctx->hwctx;

    if (hwctx->stream_owned && hwctx->p.stream)
        sws_hip_stream_destroy(hwctx->p.device, hwctx->p.stream);
    hwctx->p.stream = NULL;
    hwctx->stream_owned = 0;
}

static int hip_device_init(AVHWDeviceContext *device_ctx)
{
    HIPDeviceContext *hwctx = device_ctx->hwctx;

    /* a caller that filled AVHIPDeviceContext itself (av_hwdevice_ctx_alloc + init) may bring its own stream */
    if (hwctx->p.device < 0 || hwctx->p.device >= sws_hip_device_count()) {
        av_log(device_ctx, AV_LOG_ERROR, "No HIP device %d\n", hwctx->p.device);
        return AVERROR(ENODEV);
    }
    if (!hwctx->p.stream) {
        int ret = sws_hip_stream_create(hwctx->p.device, &hwctx->p.stream);
        if (ret < 0)
            return ret;
        hwctx->stream_owned = 1;
    }
    return 0;
}

static int hip_device_create(AVHWDeviceContext *device_ctx, const char *device, AVDictionary *opts, int flags)
{
    HIPDeviceContext *hwctx = device_ctx->hwctx;

    hwctx->p.device = device ? (int)strtol(device, NULL, 0) : 0;
    hwctx->p.stream = NULL;
    hwctx->stream_owned = 0;
    return 0;       /* device_init validates the ordinal and makes the stream */
}

const HWContextType ff_hwcontext_type_hip = {
    .type                   = AV_HWDEVICE_TYPE_HIP,
    .name                   = "HIP",

    .device_hwctx_size      = sizeof(HIPDeviceContext),
    .frames_hwctx_size      = sizeof(HIPFramesContext),

    .device_create          = hip_device_create,
    .device_init            = hip_device_init,
    .device_uninit          = hip_device_uninit,
    .frames_get_constraints = hip_frames_get_constraints,
    .frames_init            = hip_frames_init,
    .frames_get_buffer      = hip_get_buffer,
    .transfer_get_formats   = hip_transfer_get_formats,
    .transfer_data_to       = hip_transfer_data,
    .transfer_data_from     = hip_transfer_data,

    .pix_fmts               = (const enum AVPixelFormat[]){ AV_PIX_FMT_HIP, AV_PIX_FMT_NONE },
};
