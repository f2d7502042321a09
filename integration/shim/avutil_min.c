/* avutil_min.c -- the few libavutil functions the sources under integration/ call, implemented just far enough for the in-repo tests
 * (see avutil_min.h: TEST SCAFFOLDING, not product).  Behaviour follows libavutil/buffer.c (reference counting, a pool that
 * recycles released buffers) and libavutil/hwcontext.c:140-420 (device / frames context life cycle). */
#include "avutil_min.h"
#include "hwcontext_hip.h"
#include <stdlib.h>
#include <string.h>

static int live_buffers;
int shim_live_buffers(void) { return live_buffers; }

void *av_malloc_array(size_t nmemb, size_t size) { size_t n = nmemb * size; return malloc(n ? n : 1); }
void *av_mallocz(size_t size) { return calloc(1, size ? size : 1); }
void  av_free(void *ptr) { free(ptr); }
void  av_freep(void *arg) { void **p = arg; free(*p); *p = NULL; }

struct AVBufferPool;
typedef struct Buffer {          /* AVBuffer */
    uint8_t *data; size_t size; int refcount;
    void (*free)(void *opaque, uint8_t *data); void *opaque;
    struct AVBufferPool *pool;   /* owner to return to, or NULL */
    struct Buffer *next;
} Buffer;
struct AVBufferPool {
    size_t size; void *opaque; int refcount;
    AVBufferRef *(*alloc)(void *opaque, size_t size);
    void (*pool_free)(void *opaque);
    Buffer *spare;
};

static void pool_release(AVBufferPool *pool);

static AVBufferRef *ref_of(Buffer *b)
{
    AVBufferRef *r = av_mallocz(sizeof(*r));
    if (!r) return NULL;
    r->buffer = b; r->data = b->data; r->size = b->size;
    return r;
}

AVBufferRef *av_buffer_create(uint8_t *data, size_t size, void (*free_cb)(void *, uint8_t *), void *opaque, int flags)
{
    Buffer *b = av_mallocz(sizeof(*b));
    if (!b) return NULL;
    b->data = data; b->size = size; b->refcount = 1; b->free = free_cb; b->opaque = opaque;
    live_buffers++;
    AVBufferRef *r = ref_of(b);
    if (!r) { free(b); live_buffers--; }
    return r;
}

AVBufferRef *av_buffer_ref(const AVBufferRef *buf)
{
    Buffer *b = buf->buffer;
    AVBufferRef *r = ref_of(b);
    if (r) b->refcount++;
    return r;
}

void av_buffer_unref(AVBufferRef **buf)
{
    if (!buf || !*buf) return;
    Buffer *b = (*buf)->buffer;
    free(*buf);
    *buf = NULL;
    if (--b->refcount) return;
    if (b->pool && b->pool->refcount > 0) {   /* back to the pool */
        AVBufferPool *pool = b->pool;
        b->next = pool->spare; pool->spare = b;
        pool_release(pool);
        return;
    }
    if (b->free) b->free(b->opaque, b->data);
    live_buffers--;
    free(b);
}

AVBufferPool *av_buffer_pool_init2(size_t size, void *opaque, AVBufferRef *(*alloc)(void *, size_t), void (*pool_free)(void *))
{
    AVBufferPool *p = av_mallocz(sizeof(*p));
    if (!p) return NULL;
    p->size = size; p->opaque = opaque; p->alloc = alloc; p->pool_free = pool_free; p->refcount = 1;
    return p;
}

static void pool_destroy(AVBufferPool *pool)
{
    while (pool->spare) {
        Buffer *b = pool->spare;
        pool->spare = b->next;
        if (b->free) b->free(b->opaque, b->data);
        live_buffers--;
        free(b);
    }
    if (pool->pool_free) pool->pool_free(pool->opaque);
    free(pool);
}

static void pool_release(AVBufferPool *pool) { if (--pool->refcount == 0) pool_destroy(pool); }

AVBufferRef *av_buffer_pool_get(AVBufferPool *pool)
{
    AVBufferRef *r;
    if (pool->spare) {
        Buffer *b = pool->spare;
        pool->spare = b->next;
        b->refcount = 1;
        r = ref_of(b);
        if (!r) { b->next = pool->spare; pool->spare = b; return NULL; }
    } else {
        r = pool->alloc(pool->opaque, pool->size);
        if (!r) return NULL;
        ((Buffer *)r->buffer)->pool = pool;
    }
    pool->refcount++;
    return r;
}

void av_buffer_pool_uninit(AVBufferPool **ppool)
{
    if (!ppool || !*ppool) return;
    AVBufferPool *pool = *ppool;
    *ppool = NULL;
    /* buffers still out return here and are freed when the last one comes back */
    while (pool->spare) {
        Buffer *b = pool->spare;
        pool->spare = b->next;
        if (b->free) b->free(b->opaque, b->data);
        live_buffers--;
        free(b);
    }
    pool_release(pool);
}

/* ---- hwcontext.c life cycle ---- */
typedef struct FFHWDeviceContext { AVHWDeviceContext p; const HWContextType *hw_type; } FFHWDeviceContext;

static void hwdevice_free(void *opaque, uint8_t *data)
{
    FFHWDeviceContext *c = (FFHWDeviceContext *)data;
    if (c->hw_type->device_uninit) c->hw_type->device_uninit(&c->p);
    free(c->p.hwctx);
    free(c);
}

int shim_hwdevice_ctx_create(AVBufferRef **pref, const HWContextType *type, const char *device)
{
    FFHWDeviceContext *c = av_mallocz(sizeof(*c));
    int ret;
    if (!c) return AVERROR(ENOMEM);
    c->hw_type = type;
    c->p.type = type->type;
    c->p.hwctx = av_mallocz(type->device_hwctx_size);
    if ((ret = type->device_create(&c->p, device, NULL, 0)) < 0 || (type->device_init && (ret = type->device_init(&c->p)) < 0)) {
        if (type->device_uninit) type->device_uninit(&c->p);
        free(c->p.hwctx); free(c);
        return ret;
    }
    *pref = av_buffer_create((uint8_t *)c, sizeof(*c), hwdevice_free, NULL, 0);
    return *pref ? 0 : AVERROR(ENOMEM);
}

static void hwframes_free(void *opaque, uint8_t *data)
{
    FFHWFramesContext *c = (FFHWFramesContext *)data;
    if (c->pool_internal) av_buffer_pool_uninit(&c->pool_internal);
    if (c->hw_type->frames_uninit) c->hw_type->frames_uninit(&c->p);
    av_buffer_unref(&c->p.device_ref);
    free(c->p.hwctx);
    free(c);
}

AVBufferRef *av_hwframe_ctx_alloc(AVBufferRef *device_ref)
{
    FFHWDeviceContext *dev = (FFHWDeviceContext *)device_ref->data;
    FFHWFramesContext *c = av_mallocz(sizeof(*c));
    if (!c) return NULL;
    c->hw_type = dev->hw_type;
    c->p.hwctx = av_mallocz(dev->hw_type->frames_hwctx_size);
    c->p.device_ref = av_buffer_ref(device_ref);
    c->p.device_ctx = &dev->p;
    c->p.format = c->p.sw_format = AV_PIX_FMT_NONE;
    return av_buffer_create((uint8_t *)c, sizeof(*c), hwframes_free, NULL, 0);
}

int av_hwframe_ctx_init(AVBufferRef *ref)
{
    FFHWFramesContext *c = (FFHWFramesContext *)ref->data;
    int ok = 0;
    for (const enum AVPixelFormat *p = c->hw_type->pix_fmts; *p != AV_PIX_FMT_NONE; p++) ok |= *p == c->p.format;
    if (!ok || c->p.width <= 0 || c->p.height <= 0) return AVERROR(EINVAL);
    int ret = c->hw_type->frames_init ? c->hw_type->frames_init(&c->p) : 0;
    if (ret < 0) return ret;
    if (c->pool_internal && !c->p.pool) c->p.pool = c->pool_internal;
    return 0;
}

int av_hwframe_get_buffer(AVBufferRef *hwframe_ref, AVFrame *frame, int flags)
{
    FFHWFramesContext *c = (FFHWFramesContext *)hwframe_ref->data;
    frame->hw_frames_ctx = av_buffer_ref(hwframe_ref);
    if (!frame->hw_frames_ctx) return AVERROR(ENOMEM);
    int ret = c->hw_type->frames_get_buffer(&c->p, frame);
    if (ret < 0) { av_buffer_unref(&frame->hw_frames_ctx); return ret; }
    frame->extended_data = frame->data;
    return 0;
}

int av_hwframe_transfer_data(AVFrame *dst, const AVFrame *src, int flags)
{
    if (src->hw_frames_ctx) {
        FFHWFramesContext *c = (FFHWFramesContext *)src->hw_frames_ctx->data;
        return c->hw_type->transfer_data_from(&c->p, dst, src);
    } else if (dst->hw_frames_ctx) {
        FFHWFramesContext *c = (FFHWFramesContext *)dst->hw_frames_ctx->data;
        return c->hw_type->transfer_data_to(&c->p, dst, src);
    }
    return AVERROR(ENOSYS);
}

AVFrame *av_frame_alloc(void)
{
    AVFrame *f = av_mallocz(sizeof(*f));
    if (!f) return NULL;
    f->format = -1;
    f->extended_data = f->data;
    f->color_primaries = f->color_trc = f->colorspace = 2;      /* get_frame_defaults(), libavutil/frame.c */
    f->sample_aspect_ratio.den = 1;
    return f;
}

void av_frame_unref(AVFrame *f)
{
    if (!f) return;
    for (int i = 0; i < 8; i++) av_buffer_unref(&f->buf[i]);
    av_buffer_unref(&f->hw_frames_ctx);
    memset(f, 0, sizeof(*f));
    f->format = -1;
    f->extended_data = f->data;
    f->color_primaries = f->color_trc = f->colorspace = 2;
    f->sample_aspect_ratio.den = 1;
}

void av_frame_free(AVFrame **f)
{
    if (!f || !*f) return;
    av_frame_unref(*f);
    free(*f);
    *f = NULL;
}

int av_frame_copy_props(AVFrame *dst, const AVFrame *src)   /* the fields libavutil/frame.c frame_copy_props() copies, minus side data */
{
    dst->pict_type = src->pict_type; dst->sample_aspect_ratio = src->sample_aspect_ratio;
    dst->pts = src->pts; dst->pkt_dts = src->pkt_dts; dst->time_base = src->time_base; dst->duration = src->duration;
    dst->quality = src->quality; dst->repeat_pict = src->repeat_pict; dst->flags = src->flags;
    dst->color_range = src->color_range; dst->color_primaries = src->color_primaries; dst->color_trc = src->color_trc;
    dst->colorspace = src->colorspace; dst->chroma_location = src->chroma_location;
    dst->best_effort_timestamp = src->best_effort_timestamp; dst->alpha_mode = src->alpha_mode;
    return 0;
}
