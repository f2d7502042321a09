/*
 * hwcontext_hip.h -- the public side of the AV_HWDEVICE_TYPE_HIP slot.
 *
 * Plays the role libavutil/hwcontext_cuda.h:42-62 plays for CUDA: the struct a caller finds in
 * AVHWDeviceContext.hwctx of a HIP device, plus the device-level helpers of libswscale_hip.so that
 * integration/hwcontext_hip.c (the HWContextType implementation, libavutil/hwcontext_internal.h:29-91)
 * is written against.  No HIP header is needed to include this file.
 *
 * Frames of this device type have format == AV_PIX_FMT_HIP, their AVHWFramesContext.sw_format names the
 * pixel layout, data[] are HBM pointers on AVHIPDeviceContext.device and linesize[] are byte strides
 * (like AV_PIX_FMT_CUDA, hwcontext_cuda.c:132-197).
 */
#ifndef HWCONTEXT_HIP_H
#define HWCONTEXT_HIP_H

#include <stddef.h>
#include <stdint.h>
#ifdef SWS_HIP_PREFIXED
#include "swscale_hip_prefix.h"
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* AVHWDeviceContext.hwctx of an AV_HWDEVICE_TYPE_HIP device (hwcontext_cuda.h:42-62 analogue) */
typedef struct AVHIPDeviceContext {
    int   device;    /* HIP device ordinal */
    void *stream;    /* hipStream_t all work on this device's frames is ordered on (transfers, conversions) */
} AVHIPDeviceContext;

/* AVHWFramesContext.hwctx is unused (hwcontext_cuda.h:64-66 "currently unused") */

#ifndef AVUTIL_HWCONTEXT_H
/* Mirrors of libavutil/hwcontext.h:63-106 AVHWDeviceContext and :118-221 AVHWFramesContext (same field
 * order and types) for code built without libavutil: libswscale_hip.so reads sw_format, width, height,
 * device_ctx->type and device_ctx->hwctx from the hw_frames_ctx of an AV_PIX_FMT_HIP frame. */
typedef struct SwsHWDeviceContext {
    const void *av_class;
    int   type;                     /* enum AVHWDeviceType: AV_HWDEVICE_TYPE_HIP */
    void *hwctx;                    /* AVHIPDeviceContext * */
    void (*free)(struct SwsHWDeviceContext *ctx);
    void *user_opaque;
} SwsHWDeviceContext;

typedef struct SwsHWFramesContext {
    const void *av_class;
    void *device_ref;               /* AVBufferRef *, ->data is the SwsHWDeviceContext */
    SwsHWDeviceContext *device_ctx;
    void *hwctx;
    void (*free)(struct SwsHWFramesContext *ctx);
    void *user_opaque;
    void *pool;                     /* AVBufferPool * */
    int   initial_pool_size;
    int   format;                   /* AV_PIX_FMT_HIP */
    int   sw_format;
    int   width, height;
} SwsHWFramesContext;
#endif

#pragma GCC visibility push(default)

/* ---- device-level helpers of libswscale_hip.so: everything hwcontext_hip.c needs from the HIP runtime ----
 * All return 0 or a negative AVERROR (AVERROR_EXTERNAL for HIP failures, EINVAL for bad arguments). */
int   sws_hip_device_count(void);
/* hipMalloc / hipFree on `device`, with the caller's current device restored (cuda_pool_alloc, hwcontext_cuda.c:104-130) */
int   sws_hip_mem_alloc(int device, size_t size, void **ptr);
void  sws_hip_mem_free(int device, void *ptr);
/* a non-blocking stream on `device` (device_create: hwcontext_cuda.c:780-830 creates the context + stream) */
int   sws_hip_stream_create(int device, void **stream);
void  sws_hip_stream_destroy(int device, void *stream);
int   sws_hip_stream_sync(int device, void *stream);
/* one plane of transfer_data_to / transfer_data_from (cuda_transfer_data, hwcontext_cuda.c:523-655):
 * hipMemcpy2DAsync on `stream`; the pointers may be host or device memory */
int   sws_hip_copy_plane(int device, void *stream, void *dst, int dst_linesize, const void *src, int src_linesize,
                         int bytewidth, int height);
/* HIP device that holds `ptr`, or -1 for host memory */
int   sws_hip_pointer_device(const void *ptr);
/* plane layout of one linear allocation: linesize[] aligned to `align`, planes at `align`ed offsets
 * (av_image_fill_linesizes / av_image_fill_plane_sizes / av_image_fill_pointers, hwcontext_cuda.c:160-185) */
int   sws_hip_image_layout(int format, int width, int height, int align, int linesize[4], size_t offset[4], size_t *total);
/* visible bytes per row and rows of plane `plane` of a width x height picture (the 2-D extent a transfer copies) */
int   sws_hip_plane_geometry(int format, int width, int height, int plane, int *bytewidth, int *rows);
/* the sw_formats frames_init accepts == the formats libswscale_hip converts (cuda's supported_formats[], hwcontext_cuda.c:44-80) */
int   sws_hip_frames_format_supported(int sw_format);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* HWCONTEXT_HIP_H */
