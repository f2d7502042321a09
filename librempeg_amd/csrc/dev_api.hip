// HIP side of libswscale_hip, part 4 of 4 -- PLUMBING: stream loans and inheritance for the frame API, frame copies, the hwcontext-shaped device helpers
// (include/hwcontext_hip.h) and the sws_hip_* introspection / tuning entry points.
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
// A dynamic context's per-field child runs where the parent would: same home GPU, same launch heuristics, and on the stream of
// the frames' AVHIPDeviceContext if they have one, else on the stream the caller gave the parent, else on its own.
int dev_inherit(SwsInternal *child, SwsInternal *parent, bool have_stream, void *stream, int device)
{
    int r = ensure_dev(parent);
    if (r < 0) return r;
    if ((r = ensure_dev(child)) < 0) return r;
    // (HIP frames name their GPU: a stream of that GPU's device context must not end up on the state of another one)
    const int want_dev = device >= 0 ? device : parent->dev->device;
    if (child->dev->device != want_dev && (r = sws_hip_set_device(&child->opts, want_dev)) < 0) return r;
    if (std::memcmp(&child->tune, &parent->tune, sizeof(Tuning))) {
        child->tune = parent->tune;
        mark_tables_dirty(child);
        for (SwsInternal *cc : child->cascade) if (cc) { cc->tune = parent->tune; mark_tables_dirty(cc); }
    }
    if (child->dev->timing != parent->dev->timing && (r = sws_hip_set_timing(&child->opts, parent->dev->timing)) < 0) return r;
    // (a frames' hwdevice stream is not installed here: run_graphs() borrows it around the call, dev_borrow_stream)
    (void)have_stream; (void)stream;
    void *want = parent->dev->stream && !parent->dev->own_stream ? (void *)parent->dev->stream : nullptr;
    if (want) return dev_use_stream(child, want);
    if (child->dev->stream && !child->dev->own_stream) return sws_hip_set_stream(&child->opts, nullptr);
    return 0;
}

// A frames' hwdevice stream is BORROWED for one call: the context's own stream comes back afterwards, so that nothing of the caller's is held
// (or synchronised, or found destroyed) later.  Work of the context on either stream stays ordered through two events: the borrowed stream
// first waits for what the context still has in flight on its own stream, and the own stream then waits for the call's work.
int dev_borrow_stream(SwsInternal *c, void *stream, StreamLoan *loan)
{
    loan->active = false;
    int r = ensure_dev(c);
    if (r < 0 || !stream) return r;
    DeviceState *d = c->dev;
    if ((void *)d->stream == stream) return 0;
    DeviceGuard guard;
    HIPCHK(hipSetDevice(d->device));
    if (!d->stream) { HIPCHK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking)); d->own_stream = true; }
    if (!d->ev_loan) HIPCHK(hipEventCreateWithFlags(&d->ev_loan, hipEventDisableTiming));
    HIPCHK(hipEventRecord(d->ev_loan, d->stream));
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, d->ev_loan, 0));
    loan->prev = (void *)d->stream; loan->prev_own = d->own_stream; loan->active = true;
    d->stream = (hipStream_t)stream; d->own_stream = false;
    return 0;
}
void dev_return_stream(SwsInternal *c, const StreamLoan &loan)
{
    if (!loan.active || !c->dev) return;
    DeviceState *d = c->dev;
    DeviceGuard guard;
    (void)hipSetDevice(d->device);
    if (d->ev_loan && hipEventRecord(d->ev_loan, d->stream) == hipSuccess) (void)hipStreamWaitEvent((hipStream_t)loan.prev, d->ev_loan, 0);
    else (void)hipGetLastError();
    // a cascade's children copied the borrowed handle at run time (run_single: cd[k]->stream = d->stream): they go back to the parent's own
    // stream too, so that nothing of the caller's is held once the loan is over
    const hipStream_t lent = d->stream;
    for (SwsInternal *cc : c->cascade) {
        if (!cc) continue;
        for (DeviceState *cd : { cc->dev, (size_t)d->device < cc->peers.size() ? cc->peers[(size_t)d->device] : nullptr })
            if (cd && cd->device == d->device && cd->stream == lent && !cd->own_stream) cd->stream = (hipStream_t)loan.prev;
        for (SwsInternal *gc : cc->cascade) {     // (a cascade step that is itself a cascade: error-diffusion contexts)
            if (!gc) continue;
            for (DeviceState *gd : { gc->dev, (size_t)d->device < gc->peers.size() ? gc->peers[(size_t)d->device] : nullptr })
                if (gd && gd->device == d->device && gd->stream == lent && !gd->own_stream) gd->stream = (hipStream_t)loan.prev;
        }
    }
    d->stream = (hipStream_t)loan.prev; d->own_stream = loan.prev_own;
}

int dev_use_stream(SwsInternal *c, void *stream)
{
    int r = ensure_dev(c);
    if (r < 0) return r;
    if ((void *)c->dev->stream == stream && !c->dev->own_stream) return 0;
    return sws_hip_set_stream(&c->opts, stream);
}

// a conversion that changes nothing (ff_fmt_equal): the reference's threaded plane copy (graph.c:817-830)
int dev_copy_frame(SwsInternal *c, SwsFrameView *dstf, const SwsFrameView *srcf, bool have_stream, void *stream)
{
    int r = ensure_dev(c);
    if (r < 0) return r;
    DeviceGuard guard;
    const int sd = ptr_device(srcf->data[0]), dd = ptr_device(dstf->data[0]);
    const int dev = sd >= 0 ? sd : dd >= 0 ? dd : c->dev->device;
    HIPCHK(hipSetDevice(dev));
    hipStream_t st = (hipStream_t)stream;
    if (!have_stream) {
        DeviceState *d = dev_state_for(c, dev);
        if (!d) return AVERROR_EXTERNAL_;
        if (!d->stream) { HIPCHK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking)); d->own_stream = true; }
        st = d->stream;
    }
    const int np = pix_nb_planes(pix_desc(srcf->format));
    for (int k = 0; k < np; k++) {
        int rb, rows; plane_geometry(srcf->format, srcf->width, srcf->height, k, &rb, &rows);
        HIPCHK(hipMemcpy2DAsync(dstf->data[k], (size_t)dstf->linesize[k], srcf->data[k], (size_t)srcf->linesize[k], (size_t)rb, (size_t)rows, hipMemcpyDefault, st));
    }
    if (dd < 0) HIPCHK(hipStreamSynchronize(st));   // a host destination is complete on return
    return 0;
}

} // namespace swship

using namespace swship;

extern "C" {
// ---- device-level helpers (include/hwcontext_hip.h): what integration/hwcontext_hip.c needs from the HIP runtime ----
int sws_hip_mem_alloc(int device, size_t size, void **ptr)
{
    if (!ptr || device < 0) return SWS_AVERROR(EINVAL);
    *ptr = nullptr;
    DeviceGuard guard;
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return AVERROR_EXTERNAL_; }
    if (hipMalloc(ptr, size ? size : 256) != hipSuccess) { (void)hipGetLastError(); *ptr = nullptr; return SWS_AVERROR(ENOMEM); }
    return 0;
}
void sws_hip_mem_free(int device, void *ptr)
{
    if (!ptr) return;
    DeviceGuard guard;
    if (device >= 0) (void)hipSetDevice(device);
    (void)hipFree(ptr);
}
int sws_hip_stream_create(int device, void **stream)
{
    if (!stream || device < 0) return SWS_AVERROR(EINVAL);
    DeviceGuard guard;
    hipStream_t st = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return AVERROR_EXTERNAL_; }
    *stream = (void *)st;
    return 0;
}
void sws_hip_stream_destroy(int device, void *stream)
{
    if (!stream) return;
    DeviceGuard guard;
    if (device >= 0) (void)hipSetDevice(device);
    (void)hipStreamSynchronize((hipStream_t)stream);
    (void)hipStreamDestroy((hipStream_t)stream);
}
int sws_hip_stream_sync(int device, void *stream)
{
    DeviceGuard guard;
    if (device >= 0 && hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return AVERROR_EXTERNAL_; }
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) { (void)hipGetLastError(); return AVERROR_EXTERNAL_; }
    return 0;
}
int sws_hip_copy_plane(int device, void *stream, void *dst, int dst_linesize, const void *src, int src_linesize, int bytewidth, int height)
{
    if (!dst || !src || bytewidth < 0 || height < 0) return SWS_AVERROR(EINVAL);
    if (!bytewidth || !height) return 0;
    DeviceGuard guard;
    if (device >= 0 && hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return AVERROR_EXTERNAL_; }
    if (hipMemcpy2DAsync(dst, (size_t)dst_linesize, src, (size_t)src_linesize, (size_t)bytewidth, (size_t)height, hipMemcpyDefault,
                         (hipStream_t)stream) != hipSuccess) { (void)hipGetLastError(); return AVERROR_EXTERNAL_; }
    return 0;
}
int sws_hip_pointer_device(const void *ptr) { return ptr_device(ptr); }
int sws_hip_plane_geometry(int format, int width, int height, int plane, int *bytewidth, int *rows)
{
    const PixDesc *d = pix_desc(format);
    if (!d || width <= 0 || height <= 0 || plane < 0 || plane >= pix_nb_planes(d) || !bytewidth || !rows) return SWS_AVERROR(EINVAL);
    return plane_geometry(format, width, height, plane, bytewidth, rows);
}
int sws_hip_frames_format_supported(int sw_format)
{
    return pix_desc(sw_format) && sws_isSupportedInput((enum AVPixelFormat)sw_format) && sws_isSupportedOutput((enum AVPixelFormat)sw_format);
}

// ---- HIP device plumbing ----
int sws_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int sws_hip_set_device(SwsContext *sws, int device)
{
    if (!sws) return SWS_AVERROR(EINVAL);
    SwsInternal *c = internal(sws);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { (void)hipGetLastError(); return AVERROR_EXTERNAL_; }
    if (c->dev && c->dev->device == device) return 0;
    // a new home GPU: everything the context (and the children of a cascade) holds on any GPU is released and rebuilt on first use
    dev_release(c);
    for (SwsInternal *cc : c->cascade) if (cc) dev_release(cc);
    frames_release(c);      // the per-field conversions of a dynamic context are rebuilt on the new GPU
    int r = ensure_dev(c);
    if (r < 0) return r;
    c->dev->device = device;
    mark_tables_dirty(c);
    return 0;
}

int sws_hip_get_device(SwsContext *sws)
{
    if (!sws) return SWS_AVERROR(EINVAL);
    SwsInternal *c = internal(sws);
    int r = ensure_dev(c);
    return r < 0 ? r : c->dev->device;
}

int sws_hip_set_stream(SwsContext *sws, void *stream)
{
    if (!sws) return SWS_AVERROR(EINVAL);
    SwsInternal *c = internal(sws);
    int r = ensure_dev(c);
    if (r < 0) return r;
    DeviceGuard guard;
    DeviceState *d = c->dev;
    if (hipSetDevice(d->device) != hipSuccess) { (void)hipGetLastError(); return AVERROR_EXTERNAL_; }
    if (d->stream) (void)hipStreamSynchronize(d->stream);
    if (d->stream && d->own_stream) (void)hipStreamDestroy(d->stream);
    d->stream = (hipStream_t)stream;
    d->own_stream = false;
    if (!stream) { // back to a context-owned stream
        if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return AVERROR_EXTERNAL_; }
        d->own_stream = true;
    }
    return 0;
}

void *sws_hip_get_stream(SwsContext *sws)
{
    if (!sws) return nullptr;
    SwsInternal *c = internal(sws);
    if (dev_prepare(c) < 0) return nullptr;
    return (void *)c->dev->stream;
}

int sws_hip_plan(SwsContext *sws, uint64_t digest[3])
{
    if (!sws) return SWS_AVERROR(EINVAL);
    SwsInternal *c = internal(sws);
    if (!c->legacy_init) return SWS_AVERROR(EINVAL);     // (a dynamic context plans per frame)
    uint64_t dg[3] = { 0, 0, 0 };
    int r = dev_plan_digest(c, dg);
    if (r < 0) return r;
    if (digest) { digest[0] = dg[0]; digest[1] = dg[1]; digest[2] = dg[2]; }
    return 0;
}

int sws_hip_debug_check(SwsContext *sws, char *buf, int cap)
{
    if (!sws) return SWS_AVERROR(EINVAL);
    SwsInternal *c = internal(sws);
    DeviceGuard guard;
    std::string out;
    int bad = 0;
    { int r = dev_check_state(c, c->dev, out); if (r < 0) return r; bad += r; }
    for (DeviceState *d : c->peers) { int r = dev_check_state(c, d, out); if (r < 0) return r; bad += r; }
    for (const FrameGraph &g : c->graph) if (g.legacy) { int r = sws_hip_debug_check(g.legacy, nullptr, 0); if (r > 0) { bad += r; out += "(a child context of the frame graph differs); "; } }
    if (buf && cap > 0) { std::snprintf(buf, (size_t)cap, "%s", out.c_str()); }
    return bad;
}

int sws_hip_sync(SwsContext *sws)   // waits for the context's work on every GPU it has used
{
    if (!sws) return SWS_AVERROR(EINVAL);
    SwsInternal *c = internal(sws);
    int ret = 0;
    if (c->dev && c->dev->stream && hipStreamSynchronize(c->dev->stream) != hipSuccess) { (void)hipGetLastError(); ret = AVERROR_EXTERNAL_; }
    for (DeviceState *d : c->peers)
        if (d && d->stream && hipStreamSynchronize(d->stream) != hipSuccess) { (void)hipGetLastError(); ret = AVERROR_EXTERNAL_; }
    for (const FrameGraph &g : c->graph) if (g.legacy) { int r = sws_hip_sync(g.legacy); if (r < 0) ret = r; }
    return ret;
}

int sws_hip_set_timing(SwsContext *sws, int enable)
{
    if (!sws) return SWS_AVERROR(EINVAL);
    SwsInternal *c = internal(sws);
    int r = ensure_dev(c);
    if (r < 0) return r;
    DeviceGuard guard;
    DeviceState *d = c->dev;
    if (hipSetDevice(d->device) != hipSuccess) { (void)hipGetLastError(); return AVERROR_EXTERNAL_; }
    if (enable && !d->ev0) {
        if (hipEventCreate(&d->ev0) != hipSuccess || hipEventCreate(&d->ev1) != hipSuccess) return AVERROR_EXTERNAL_;
    }
    d->timing = enable != 0;
    d->timed = false;
    return 0;
}

double sws_hip_last_kernel_ms(SwsContext *sws)
{
    if (!sws) return -1.0;
    SwsInternal *c = internal(sws);
    if (!c->legacy_init && c->graph[0].legacy) return sws_hip_last_kernel_ms(c->graph[0].legacy);   // dynamic context: its top-field conversion
    if (!c->dev || !c->dev->timed) return -1.0;
    float ms = 0.f;
    if (hipEventSynchronize(c->dev->ev1) != hipSuccess) return -1.0;
    if (hipEventElapsedTime(&ms, c->dev->ev0, c->dev->ev1) != hipSuccess) return -1.0;
    return ms;
}

// launch heuristics (swsint.hpp: Tuning); returns 0, or AVERROR(EINVAL) for an unknown name
int sws_hip_set_option(SwsContext *sws, const char *name, int value)
{
    if (!sws || !name) return SWS_AVERROR(EINVAL);
    SwsInternal *c = internal(sws);
    struct { const char *n; int *v; } tab[] = {
        { "strip_min_w", &c->tune.strip_min_w }, { "strip_cols_l", &c->tune.strip_cols_l }, { "strip_cols_c", &c->tune.strip_cols_c },
        { "strip_waves", &c->tune.strip_waves }, { "strip_rgb_cols", &c->tune.strip_rgb_cols }, { "rgb_march_waves", &c->tune.rgb_march_waves }, { "tile_lds_kb", &c->tune.tile_lds_kb },
        { "tile_threads", &c->tune.tile_threads }, { "p01x_ch", &c->tune.p01x_ch }, { "layout_ch", &c->tune.layout_ch }, { "no_mixed", &c->tune.no_mixed }, { "no_layout_stream", &c->tune.no_layout_stream }, { "no_wave", &c->tune.no_wave }, { "no_march", &c->tune.no_march },
        { "no_rgbsrc", &c->tune.no_rgbsrc }, { "no_strip", &c->tune.no_strip }, { "no_strip_dma", &c->tune.no_strip_dma }, { "no_dot2", &c->tune.no_dot2 }, { "no_tile", &c->tune.no_tile }, { "max_devices", &c->tune.max_devices },
        { "strip_min_rows", &c->tune.strip_min_rows }, { "no_strip_fuse", &c->tune.no_strip_fuse }, { "work_mb", &c->tune.work_mb }, { "no_strip_range", &c->tune.no_strip_range }, { "no_strip_wide", &c->tune.no_strip_wide }, { "no_wide_epilogue", &c->tune.no_wide_epilogue }, { "no_strip_u16", &c->tune.no_strip_u16 },
        { "no_strip_dma8", &c->tune.no_strip_dma8 }, { "strip_lds_pad_kb", &c->tune.strip_lds_pad_kb }, { "no_striprgb_direct", &c->tune.no_striprgb_direct }, { "no_rgbsrc2", &c->tune.no_rgbsrc2 }, { "no_strip_rgbsrc", &c->tune.no_strip_rgbsrc }, { "no_strip_rgb2rgb", &c->tune.no_strip_rgb2rgb },
        { "no_strip_short", &c->tune.no_strip_short }, { "no_generic_kinds", &c->tune.no_generic_kinds }, { "no_rgbread_kinds", &c->tune.no_rgbread_kinds }, { "strip_cols_auto", &c->tune.strip_cols_auto }, { "no_fast_banks", &c->tune.no_fast_banks }, { "no_short_forms", &c->tune.no_short_forms }, { "strip_short_waves", &c->tune.strip_short_waves }, { "rccl_tables", &c->tune.rccl_tables }, { "dry_plan", &c->tune.dry_plan }, { "exp0", &c->tune.exp[0] }, { "exp1", &c->tune.exp[1] }, { "exp2", &c->tune.exp[2] }, { "exp3", &c->tune.exp[3] }, { "exp4", &c->tune.exp[4] }, { "exp5", &c->tune.exp[5] }, { "exp6", &c->tune.exp[6] }, { "exp7", &c->tune.exp[7] },
        { "debug", &c->tune.debug },
    };
    for (auto &e : tab)
        if (!std::strcmp(e.n, name)) {
            *e.v = value;
            mark_tables_dirty(c);
            for (SwsInternal *cc : c->cascade) if (cc) { cc->tune = c->tune; mark_tables_dirty(cc); }
            return 0;
        }
    return SWS_AVERROR(EINVAL);
}

int sws_hip_image_layout(int format, int width, int height, int align, int linesize[4], size_t offset[4], size_t *total)
{
    if (align <= 0) align = 256;
    return image_layout(format, width, height, align, linesize, offset, total);
}

int sws_hip_frame_alloc(SwsFrameView *f, int format, int width, int height, int device)
{
    if (!f) return SWS_AVERROR(EINVAL);
    std::memset(f, 0, sizeof(*f));
    int ls[4]; size_t offs[4], total;
    int r = image_layout(format, width, height, 256, ls, offs, &total);
    if (r < 0) return r;
    DeviceGuard guard;      // the caller's current device is left as it was
    if (device >= 0 && hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return AVERROR_EXTERNAL_; }
    void *base = nullptr;
    if (hipMalloc(&base, total ? total : 256) != hipSuccess) { (void)hipGetLastError(); return SWS_AVERROR(ENOMEM); }
    const int np = pix_nb_planes(pix_desc(format));
    for (int k = 0; k < np; k++) { f->data[k] = (uint8_t *)base + offs[k]; f->linesize[k] = ls[k]; }
    f->extended_data = f->data;
    f->width = width; f->height = height; f->format = format;
    f->color_primaries = f->color_trc = f->colorspace = 2;   // *_UNSPECIFIED: get_frame_defaults(), libavutil/frame.c
    f->sample_aspect_ratio.den = 1;
    return 0;
}

void sws_hip_frame_free(SwsFrameView *f)
{
    if (!f || !f->data[0]) return;
    (void)hipFree(f->data[0]);
    std::memset(f, 0, sizeof(*f));
}

static int frame_copy(SwsContext *sws, SwsFrameView *dstf, const SwsFrameView *srcf, hipMemcpyKind kind)
{
    if (!dstf || !srcf || dstf->format != srcf->format || dstf->width != srcf->width || dstf->height != srcf->height)
        return SWS_AVERROR(EINVAL);
    hipStream_t st = nullptr;
    SwsInternal *c = sws ? internal(sws) : nullptr;
    DeviceGuard guard;
    const int fdev = ptr_device(kind == hipMemcpyHostToDevice ? (const void *)dstf->data[0] : (const void *)srcf->data[0]);
    if (fdev >= 0 && hipSetDevice(fdev) != hipSuccess) { (void)hipGetLastError(); return AVERROR_EXTERNAL_; }
    if (c) {   // the copy is ordered on the context's stream of the GPU that holds the frame
        DeviceState *d = fdev >= 0 ? dev_state_for(c, fdev) : (ensure_dev(c) < 0 ? nullptr : c->dev);
        if (!d) return AVERROR_EXTERNAL_;
        if (!d->stream) { if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) return AVERROR_EXTERNAL_; d->own_stream = true; }
        st = d->stream;
    }
    const int np = pix_nb_planes(pix_desc(srcf->format));
    for (int k = 0; k < np; k++) {
        int rb, rows; plane_geometry(srcf->format, srcf->width, srcf->height, k, &rb, &rows);
        if (hipMemcpy2DAsync(dstf->data[k], dstf->linesize[k], srcf->data[k], srcf->linesize[k], rb, rows, kind, st) != hipSuccess) {
            (void)hipGetLastError(); return AVERROR_EXTERNAL_;
        }
    }
    // sync only when the destination is a software frame (hwcontext_cuda.c:642-646)
    if (kind == hipMemcpyDeviceToHost) { if (hipStreamSynchronize(st) != hipSuccess) return AVERROR_EXTERNAL_; }
    return 0;
}

int sws_hip_frame_upload(SwsContext *sws, SwsFrameView *dev, const SwsFrameView *host)
{
    int r = frame_copy(sws, dev, host, hipMemcpyHostToDevice);
    if (r == 0 && sws) (void)sws_hip_sync(sws); // pageable host memory: make the source reusable on return
    return r;
}
int sws_hip_frame_download(SwsContext *sws, SwsFrameView *host, const SwsFrameView *dev)
{
    return frame_copy(sws, host, dev, hipMemcpyDeviceToHost);
}

} // extern "C"

