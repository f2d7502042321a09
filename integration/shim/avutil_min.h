/*
 * avutil_min.h -- declaration-only stand-in for the libavutil headers the sources under integration/ include, so that those
 * files can be compiled and exercised inside this repository (which has no libavutil).  TEST SCAFFOLDING: not
 * part of the product, not shipped, never linked into libswscale_hip.so.
 *
 * Types have the reference's layouts (libavutil/buffer.h:82-95, frame.h:472-828 via include/swscale_hip.h,
 * hwcontext.h:63-221, :444-471, hwcontext_internal.h:29-117); functions have the reference's prototypes and
 * are implemented in avutil_min.c with just enough behaviour for the tests (reference counting, a trivial pool).
 */
#ifndef AVUTIL_MIN_H
#define AVUTIL_MIN_H

#include <errno.h>
#include <stddef.h>
#include <stdint.h>
#include "swscale_hip.h"          /* SwsFrameView (the AVFrame mirror), SwsBufferRef, enum AVPixelFormat, AV_HWDEVICE_TYPE_HIP */

#define AVUTIL_HWCONTEXT_H        /* hwcontext_hip.h: the real struct names are declared here */

#define AVERROR(e) (-(e))
#define AVERROR_EXTERNAL (-(int)(('E') | (('X') << 8) | (('T') << 16) | ((unsigned)(' ') << 24)))
#define AV_LOG_ERROR 16
#define AV_LOG_VERBOSE 40
#define av_log(ctx, level, ...) ((void)(ctx))
#define FFMIN(a, b) ((a) > (b) ? (b) : (a))
#define FFMAX(a, b) ((a) > (b) ? (a) : (b))
#define FFABS(a) ((a) >= 0 ? (a) : (-(a)))
#define av_cold

typedef SwsBufferRef AVBufferRef;
typedef SwsFrameView AVFrame;
typedef struct AVBufferPool AVBufferPool;
typedef struct AVDictionary AVDictionary;
typedef struct AVClass AVClass;
typedef SwsRational AVRational;
enum AVHWDeviceType { AV_HWDEVICE_TYPE_NONE_ = 0 };
enum AVHWFrameTransferDirection { AV_HWFRAME_TRANSFER_DIRECTION_FROM, AV_HWFRAME_TRANSFER_DIRECTION_TO };

typedef struct AVHWDeviceContext {
    const AVClass *av_class;
    enum AVHWDeviceType type;
    void *hwctx;
    void (*free)(struct AVHWDeviceContext *ctx);
    void *user_opaque;
} AVHWDeviceContext;

typedef struct AVHWFramesContext {
    const AVClass *av_class;
    AVBufferRef *device_ref;
    AVHWDeviceContext *device_ctx;
    void *hwctx;
    void (*free)(struct AVHWFramesContext *ctx);
    void *user_opaque;
    AVBufferPool *pool;
    int initial_pool_size;
    enum AVPixelFormat format;
    enum AVPixelFormat sw_format;
    int width, height;
} AVHWFramesContext;

typedef struct AVHWFramesConstraints {
    enum AVPixelFormat *valid_hw_formats;
    enum AVPixelFormat *valid_sw_formats;
    int min_width, min_height;
    int max_width, max_height;
} AVHWFramesConstraints;

typedef struct HWContextType {
    enum AVHWDeviceType type;
    const char *name;
    const enum AVPixelFormat *pix_fmts;
    size_t device_hwctx_size;
    size_t device_hwconfig_size;
    size_t frames_hwctx_size;
    int  (*device_create)(AVHWDeviceContext *ctx, const char *device, AVDictionary *opts, int flags);
    int  (*device_derive)(AVHWDeviceContext *dst_ctx, AVHWDeviceContext *src_ctx, AVDictionary *opts, int flags);
    int  (*device_init)(AVHWDeviceContext *ctx);
    void (*device_uninit)(AVHWDeviceContext *ctx);
    int  (*frames_get_constraints)(AVHWDeviceContext *ctx, const void *hwconfig, AVHWFramesConstraints *constraints);
    int  (*frames_init)(AVHWFramesContext *ctx);
    void (*frames_uninit)(AVHWFramesContext *ctx);
    int  (*frames_get_buffer)(AVHWFramesContext *ctx, AVFrame *frame);
    int  (*transfer_get_formats)(AVHWFramesContext *ctx, enum AVHWFrameTransferDirection dir, enum AVPixelFormat **formats);
    int  (*transfer_data_to)(AVHWFramesContext *ctx, AVFrame *dst, const AVFrame *src);
    int  (*transfer_data_from)(AVHWFramesContext *ctx, AVFrame *dst, const AVFrame *src);
    int  (*map_to)(AVHWFramesContext *ctx, AVFrame *dst, const AVFrame *src, int flags);
    int  (*map_from)(AVHWFramesContext *ctx, AVFrame *dst, const AVFrame *src, int flags);
    int  (*frames_derive_to)(AVHWFramesContext *dst_ctx, AVHWFramesContext *src_ctx, int flags);
    int  (*frames_derive_from)(AVHWFramesContext *dst_ctx, AVHWFramesContext *src_ctx, int flags);
} HWContextType;

typedef struct FFHWFramesContext {
    AVHWFramesContext p;
    const HWContextType *hw_type;
    AVBufferPool *pool_internal;
    AVBufferRef *source_frames;
    int source_allocation_map_flags;
} FFHWFramesContext;
static inline FFHWFramesContext *ffhwframesctx(AVHWFramesContext *ctx) { return (FFHWFramesContext *)ctx; }

/* libavutil/mem.h, buffer.h: the reference's prototypes */
void *av_malloc_array(size_t nmemb, size_t size);
void *av_mallocz(size_t size);
void  av_free(void *ptr);
void  av_freep(void *ptr);
AVBufferRef *av_buffer_create(uint8_t *data, size_t size, void (*free)(void *opaque, uint8_t *data), void *opaque, int flags);
AVBufferRef *av_buffer_ref(const AVBufferRef *buf);
void  av_buffer_unref(AVBufferRef **buf);
AVBufferPool *av_buffer_pool_init2(size_t size, void *opaque, AVBufferRef *(*alloc)(void *opaque, size_t size), void (*pool_free)(void *opaque));
AVBufferRef *av_buffer_pool_get(AVBufferPool *pool);
void  av_buffer_pool_uninit(AVBufferPool **pool);

/* libavutil/hwcontext.h, frame.h: the reference's prototypes (the subset the integration sources use) */
AVBufferRef *av_hwframe_ctx_alloc(AVBufferRef *device_ctx);
int   av_hwframe_ctx_init(AVBufferRef *ref);
int   av_hwframe_get_buffer(AVBufferRef *hwframe_ctx, AVFrame *frame, int flags);
int   av_hwframe_transfer_data(AVFrame *dst, const AVFrame *src, int flags);
AVFrame *av_frame_alloc(void);
void  av_frame_free(AVFrame **frame);
void  av_frame_unref(AVFrame *frame);
int   av_frame_copy_props(AVFrame *dst, const AVFrame *src);

/* test entry: av_hwdevice_ctx_create() for a given vtable (the reference looks the type up in its hw_table[]) */
int   shim_hwdevice_ctx_create(AVBufferRef **device_ctx, const HWContextType *type, const char *device);
int   shim_live_buffers(void);

#endif /* AVUTIL_MIN_H */
