// HIP side of libswscale_hip: device state, table upload, launch planning, the sws_scale() / sws_scale_frame() /
// sws_scale_frames() entry points and the hwcontext-shaped helpers.  The kernels live in the k_*.hip translation units.
// gfx950 only; no CPU fallback: if HIP is unavailable every call fails with AVERROR_EXTERNAL.
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

#include "devstate.hpp"
#include "generic_kinds.hpp"
#include "../../include/hwcontext_hip.h"

namespace swship {
void launch_layout_split422(const LaunchCtx &L, bool uyvy, bool vfirst);   // k_layout.hip: yuyv422 / uyvy422 / yvyu422 -> planar 4:2:2 working picture
void launch_layout_splitnv(const LaunchCtx &L, bool vfirst);   // k_layout.hip: plane 1 of a semi-planar 8-bit picture -> planar U / V working planes
void launch_layout_splitp01x(const LaunchCtx &L, int shift);   // k_layout.hip: p010-style planes -> planar working picture, words >> shift
void launch_alpha_merge32(const LaunchCtx &L);                 // k_stream.hip: the alpha bytes behind sws_k_strip_rgb (alpha_launch == 2)
void launch_layout_join422(const LaunchCtx &L, bool uyvy);   // k_layout.hip: planar 4:2:2 working picture -> yuyv422 / uyvy422 (yvyu422: planes swapped by the planner)

// the plan geometries of a fresh state start out as zeros (`new DeviceState()` runs the default member initialisers and leaves members without one as the
// heap had them: a field a planner path does not set -- band counts, the byte-row form's flags -- would otherwise differ from process to process)
static int ensure_dev(SwsInternal *c)
{
    if (c->dev) return 0;
    if (c->tune.dry_plan) {      // planner-only context: no device is asked for
        DeviceState *d = new DeviceState();
        d->dry = true;
        c->dev = d;
        return 0;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        log_msg(c, 0, "no HIP device available: libswscale_hip has no CPU fallback\n");
        return AVERROR_EXTERNAL_;
    }
    DeviceState *d = new DeviceState();          // (value-initialised: every member without an initialiser starts as zero)
    if (hipGetDevice(&d->device) != hipSuccess) d->device = 0;
    c->dev = d;
    return 0;
}

// the context's state on HIP device `device`: the home state, or a peer state created on first use (sws_scale_frames() sharding)
static DeviceState *dev_state_for(SwsInternal *c, int device)
{
    if (ensure_dev(c) < 0) return nullptr;
    if (c->dev->device == device) return c->dev;
    if (device < 0) return nullptr;
    if ((size_t)device >= c->peers.size()) c->peers.resize((size_t)device + 1, nullptr);
    if (!c->peers[(size_t)device]) {
        DeviceState *d = new DeviceState();      // (value-initialised: every member without an initialiser starts as zero)
        d->device = device;
        d->timing = false;
        c->peers[(size_t)device] = d;
    }
    return c->peers[(size_t)device];
}

static void guard_forget(void *p);      // (SWS_HIP_DEBUG & 64: below)
static bool guards_enabled();
static void dev_state_free(DeviceState *d)
{
    if (!d) return;
    if (d->dry) { delete d; return; }      // (its table "blocks" are fake addresses; it holds no stream, event or allocation)
    (void)hipSetDevice(d->device);
    // (a stream the context does not own -- the caller's, or a frames' hwdevice stream on loan -- may be gone by now: it is never touched here)
    // Wait for what THIS state queued before its blocks, pinned tables and events go (no reliance on hipFree's implicit wait, which an asynchronous
    // allocator need not give) -- and for nothing else: freeing one scaler must not stall the other contexts' pipelines on the GPU (advisor r5).
    //  * its own stream: everything it launched there, and what it launched on a borrowed stream (dev_return_stream made the own stream wait for it);
    //  * the frame-table ring's launch sets (events recorded behind them on whatever stream they ran);
    //  * a loan that was never returned: its event;
    //  * a state that last ran on a stream it does not own (the caller's, sws_hip_set_stream; a cascade child on its parent's): that handle may be gone,
    //    so it cannot be waited for -- the device is, as before.
    const bool foreign = d->stream && !d->own_stream;
    if (d->stream && d->own_stream) { (void)hipStreamSynchronize(d->stream); (void)hipStreamDestroy(d->stream); }
    for (auto &b : d->ring.inflight) if (b.ev) (void)hipEventSynchronize(b.ev);
    if (d->ev_loan) (void)hipEventSynchronize(d->ev_loan);
    if (foreign || guards_enabled()) (void)hipDeviceSynchronize();
    (void)hipGetLastError();
    for (void *p : { d->d_tables, d->scratch, d->stage_src, d->stage_dst, (void *)d->ring.dev, d->casc_img, d->slice_img, d->d_tilegeom,
                     d->d_rgbplan, d->d_be, d->d_xyz, d->d_xyz_tab, d->d_dot2, d->casc_img2, d->d_gamma_tab, d->d_ed_err, d->d_pal, d->d_vlines, d->rgbread_img })
        if (p) { guard_forget(p); (void)hipFree(p); }
    if (d->ring.host) (void)hipHostFree(d->ring.host);
    for (auto &r : d->ring.retired) { if (r.dev) (void)hipFree(r.dev); if (r.host) (void)hipHostFree(r.host); }
    for (auto &b : d->ring.inflight) if (b.ev) (void)hipEventDestroy(b.ev);
    for (hipEvent_t e : d->ring.pool) (void)hipEventDestroy(e);
    for (void *p : { d->join_img, d->split_img, d->stage_img }) if (p) { guard_forget(p); (void)hipFree(p); }
    if (d->ev0) (void)hipEventDestroy(d->ev0);
    if (d->ev1) (void)hipEventDestroy(d->ev1);
    if (d->ev_loan) (void)hipEventDestroy(d->ev_loan);
    delete d;
}

void dev_release(SwsInternal *c)
{
    if (!c->dev && c->peers.empty()) return;
    DeviceGuard guard;
    dev_state_free(c->dev);
    c->dev = nullptr;
    for (DeviceState *d : c->peers) dev_state_free(d);
    c->peers.clear();
}

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
struct VLines { std::vector<int32_t> lum, chr, lumPos, chrPos; bool uniform = true; };
static void build_vlines(const SwsInternal *c, int mode, VLines &out)
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

static bool bank_is_identity(const FilterBank &b, int one)
{
    if (b.size != 1) return false;
    for (int i = 0; i < b.count; i++)
        if (b.pos[i] != i || b.taps[i] != one) return false;
    return true;
}

// every tap of a one-tap bank equals `one` (initFilter's error-diffused normalisation leaves one - 1 in some rows)
static bool bank_taps_all(const FilterBank &b, int one)
{
    if (b.size != 1) return false;
    for (int i = 0; i < b.count; i++)
        if (b.taps[i] != one) return false;
    return true;
}

static int src_kind_of(int f)
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
static int dst_kind_of(int f)
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
static bool poison_enabled();
static int poison(SwsInternal *c, void *buf, size_t bytes);
// SWS_HIP_DEBUG & 64 (round 5, DESIGN.md 8 iv): every working buffer and table block of the library carries GUARD_BYTES of a known pattern behind its last byte, and every
// conversion ends by waiting for its stream and comparing the guards of ALL live blocks of the process: a kernel that writes behind a working picture or a table fails the call
// that did it -- loudly, with the block and the offset -- instead of damaging whatever the allocator placed there (another context's tables, another picture).
static const size_t GUARD_BYTES = 64 * 1024;
static bool guards_enabled()
{
    static const bool on = std::getenv("SWS_HIP_DEBUG") && (std::atoi(std::getenv("SWS_HIP_DEBUG")) & 64);
    return on;
}
// (leaked on purpose: contexts freed during process teardown -- python finalisers, atexit -- still find the registry alive)
struct GuardRegistry { std::mutex mu; std::map<void *, size_t> blocks; };      // block -> bytes in front of its guard
static GuardRegistry &guard_registry() { static GuardRegistry *r = new GuardRegistry(); return *r; }
#define g_guard_mu (guard_registry().mu)
#define g_guarded (guard_registry().blocks)
static void guard_forget(void *p) { if (p && guards_enabled()) { std::lock_guard<std::mutex> lk(g_guard_mu); g_guarded.erase(p); } }
static int guard_arm(SwsInternal *c, void *p, size_t bytes)
{
    if (!guards_enabled() || !p) return 0;
    HIPCHK(hipMemset((uint8_t *)p + bytes, 0xA7, GUARD_BYTES));
    HIPCHK(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_guard_mu);
    g_guarded[p] = bytes;
    return 0;
}
static int guards_check(SwsInternal *c, hipStream_t st)
{
    if (!guards_enabled()) return 0;
    // the guards of ALL live blocks of the process are compared: every stream has to be quiet, or another context's conversion still in flight could be blamed on this one
    (void)st;
    HIPCHK(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_guard_mu);
    std::vector<uint8_t> h(GUARD_BYTES);
    for (const auto &e : g_guarded) {
        HIPCHK(hipMemcpy(h.data(), (const uint8_t *)e.first + e.second, GUARD_BYTES, hipMemcpyDeviceToHost));
        size_t bad = 0, first = 0;
        for (size_t i = 0; i < GUARD_BYTES; i++) if (h[i] != 0xA7) { if (!bad) first = i; bad++; }
        if (bad) {
            log_msg(c, 0, "GUARD VIOLATION: %zu bytes written behind the block %p (%zu bytes), first at +%zu (value %u) -- %s -> %s %dx%d -> %dx%d flags 0x%x path %s\n", bad, e.first, e.second, first,
                    (unsigned)h[first], pix_desc(c->opts.src_format)->name, pix_desc(c->opts.dst_format)->name, c->opts.src_w, c->opts.src_h, c->opts.dst_w, c->opts.dst_h, (unsigned)c->opts.flags, c->path_name.c_str());
            return AVERROR_EXTERNAL_;
        }
    }
    return 0;
}
// Device tables (filter banks, plan blobs, geometry): grown like grow().  Under SWS_HIP_DEBUG & 16 every table block carries 4 KiB of slack and is
// refilled with 0xCD before each upload: a kernel that reads past the end of its table (a vector load over the last tap row, a row entry fetched ahead)
// then meets garbage on every run, as it would in a block recycled from another context, instead of the zeros of a fresh allocation.
static uint64_t fnv1a64(const void *p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= ((const uint8_t *)p)[i]; h *= 1099511628211ull; }
    return h;
}
static void table_records_drop(DeviceState *d, const void *lo, size_t bytes)
{
    auto &v = d->tab_recs;
    for (size_t i = 0; i < v.size();)
        if ((const uint8_t *)v[i].dst < (const uint8_t *)lo + bytes && (const uint8_t *)lo < (const uint8_t *)v[i].dst + v[i].bytes) { v[i] = v.back(); v.pop_back(); }
        else i++;
}
static int table_alloc(SwsInternal *c, DeviceState *d, void **buf, size_t *cap, size_t need)
{
    if (d->dry) {     // a fixed fake address per table member of the state: the plan (pointers into the blocks included) is the same on every run
        table_records_drop(d, *buf, *cap);
        *buf = (void *)(uintptr_t)(0x100000000000ull + (uint64_t)((const char *)buf - (const char *)d) * 0x100000000ull);
        *cap = need;
        return 0;
    }
    const size_t slack = guards_enabled() ? GUARD_BYTES : poison_enabled() ? 4096 : 0;
    if (need + slack > *cap) {
        if (*buf) table_records_drop(d, *buf, *cap);
        guard_forget(*buf);
        if (*buf) HIPCHK(hipFree(*buf));
        *buf = nullptr; *cap = 0;
        HIPCHK(hipMalloc(buf, need + slack));
        *cap = need + slack;
    }
    { int r_ = poison(c, *buf, *cap); if (r_ < 0) return r_; }
    return guard_arm(c, *buf, need);
}

// Uploads of the per-context device tables go out on the CONTEXT'S OWN STREAM and are waited for there (round 5): the kernels that read them are launched on that
// stream, so table and reader share one queue whatever the other queues of the process -- or of the other processes on the GPU -- are doing.  (Rounds 1 - 4 used
// the blocking hipMemcpy of the null stream for most of them.  Two of the rare parity events of section 8 were contexts whose results were wrong for their whole
// lifetime while a fresh context was right, on paths that read such tables; nothing proves the copy was at fault, this just removes the cross-queue step.)
// SWS_HIP_DEBUG & 32: every upload is read back and compared -- a mismatch fails the call loudly instead of giving wrong pixels.
static bool verify_uploads()
{
    static const bool on = std::getenv("SWS_HIP_DEBUG") && (std::atoi(std::getenv("SWS_HIP_DEBUG")) & 32);
    return on;
}
static int table_put(SwsInternal *c, DeviceState *d, void *dst, const void *src, size_t bytes)
{
    if (!bytes) return 0;
    if (d->dry) {
        table_records_drop(d, dst, bytes);
        d->tab_recs.push_back({ dst, bytes, fnv1a64(src, bytes) });
        return 0;
    }
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    table_records_drop(d, dst, bytes);
    d->tab_recs.push_back({ dst, bytes, fnv1a64(src, bytes) });
    if (verify_uploads()) {
        std::vector<uint8_t> back(bytes);
        HIPCHK(hipMemcpyAsync(back.data(), dst, bytes, hipMemcpyDeviceToHost, d->stream));
        HIPCHK(hipStreamSynchronize(d->stream));
        if (std::memcmp(back.data(), src, bytes)) {
            size_t bad = 0, first = bytes;
            for (size_t i = 0; i < bytes; i++) if (back[i] != ((const uint8_t *)src)[i]) { if (first == bytes) first = i; bad++; }
            log_msg(c, 0, "table upload verification FAILED: %zu of %zu bytes differ at %p (first at +%zu)\n", bad, bytes, dst, first);
            return AVERROR_EXTERNAL_;
        }
    }
    return 0;
}

static int dev_prepare_on(SwsInternal *c, DeviceState *d)
{
    if (d->epoch == c->tables_epoch && d->stream) return 0;
    if (d->dry) d->stream = (hipStream_t)(uintptr_t)1;      // (never handed to HIP: a planner-only state is never launched on)
    else {
    HIPCHK(hipSetDevice(d->device));
    if (!d->stream) { HIPCHK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking)); d->own_stream = true; }
    // a rebuild rewrites plan tables that launches still in flight on the (non-blocking) stream may be reading
    else HIPCHK(hipStreamSynchronize(d->stream));
    }

    SwsDevParams &p = d->params;
    std::memset(&p, 0, sizeof(p));
    const SwsContext &o = c->opts;
    const PixDesc *ds = pix_desc(o.src_format), *dd = pix_desc(o.dst_format);
    // SWS_FAST_BILINEAR on 8-bit lines (ff_hyscale_fast_c / ff_hcscale_fast_c, hscale_fast_bilinear.c:23-55: dst = a (128 - xalpha) + b xalpha for luma and alpha,
    // a (127 - xalpha) + b xalpha for chroma, with xalpha = the top 7 bits of the 16-bit position fraction; the columns at and behind the last source sample: 128 x that
    // sample) is hScale8To15_c over a two-tap bank with the taps {(128 - xalpha) << 7, xalpha << 7} (chroma: {(xalpha ^ 127) << 7, xalpha << 7}) -- the sum's low 7 bits
    // are zero, so the >> 7 is exact and the 15-bit clip never acts.  Round 5: every plan below is made on these banks (hLumB / hChrB), which puts the strip kernels
    // behind the flag that players and capture tools pass most often (4K -> 1080p yuv420p: 0.090 ms per frame on the two-pass kernels, 0.012 with SWS_BILINEAR);
    // same widths give one-tap banks (chroma: 127 << 7, the reference's own quirk).  The context's banks stay what build_filter_bank() made (sws_hip_get_filter, the blob).
    FilterBank fastL, fastC;
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
    FilterBank vChrJ; bool join_short = false;      // (the chroma bank of a packed 4:2:2 destination whose rows take yuv2422_1_c_template's blend: below)
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
    // ---- filter tables -> one device blob ----
    d->unity_h = false;
    if (c->plan == PLAN_MAIN) {
        const FilterBank *banks[4] = { &hLumB, &hChrB, &c->vLum, &vChrB };
        size_t off = 0, offs_t[4], offs_p[4];
        auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
        for (int i = 0; i < 4; i++) {
            offs_t[i] = off; off = align(off + banks[i]->taps.size() * sizeof(int16_t));
            offs_p[i] = off; off = align(off + banks[i]->pos.size() * sizeof(int32_t));
        }
        { int r_ = table_alloc(c, d, &d->d_tables, &d->tables_bytes, off); if (r_ < 0) return r_; }
        std::vector<uint8_t> host(off, 0);
        for (int i = 0; i < 4; i++) {
            std::memcpy(host.data() + offs_t[i], banks[i]->taps.data(), banks[i]->taps.size() * sizeof(int16_t));
            std::memcpy(host.data() + offs_p[i], banks[i]->pos.data(), banks[i]->pos.size() * sizeof(int32_t));
        }
        { int r_ = table_put(c, d, d->d_tables, host.data(), off); if (r_ < 0) return r_; }
        uint8_t *b = (uint8_t *)d->d_tables;
        p.hLumF = (const int16_t *)(b + offs_t[0]); p.hLumPos = (const int32_t *)(b + offs_p[0]); p.hLumFs = hLumB.size;
        p.hChrF = (const int16_t *)(b + offs_t[1]); p.hChrPos = (const int32_t *)(b + offs_p[1]); p.hChrFs = hChrB.size;
        p.vLumF = (const int16_t *)(b + offs_t[2]); p.vLumPos = (const int32_t *)(b + offs_p[2]); p.vLumFs = c->vLum.size;
        p.vChrF = (const int16_t *)(b + offs_t[3]); p.vChrPos = (const int32_t *)(b + offs_p[3]); p.vChrFs = vChrB.size;
        // the fast-bilinear chroma function weighs with (xalpha ^ 127): not the identity even at equal widths
        d->unity_h = bank_is_identity(hLumB, 1 << 14) && bank_is_identity(hChrB, 1 << 14) && !p.fast_bilinear;
        d->unity_v = bank_is_identity(c->vLum, 1 << 12) && bank_is_identity(vChrB, 1 << 12);
        // ---- contexts whose result depends on the reference's line schedule (build_vlines): the two-pass path over virtual lines ----
        d->vlines_on = false; c->gamma_in_reader = false; d->mixed_ok = false;
        {
            const int drop = (o.flags & SWS_SRC_V_CHR_DROP_MASK) >> SWS_SRC_V_CHR_DROP_SHIFT;
            const int mode = c->internal_gamma ? 1 : (drop && isPlanarRGB(o.src_format) && !p.no_chroma) ? 2 : 0;
            VLines vlx;
            if (mode) build_vlines(c, mode, vlx);
            if (mode == 2 || (mode == 1 && !vlx.uniform)) {
                const size_t nL = vlx.lum.size() / 2, nC = vlx.chr.size() / 2;
                std::vector<int32_t> hostv;
                hostv.insert(hostv.end(), vlx.lum.begin(), vlx.lum.end());
                hostv.insert(hostv.end(), vlx.chr.begin(), vlx.chr.end());
                const size_t o_lp = hostv.size(); hostv.insert(hostv.end(), vlx.lumPos.begin(), vlx.lumPos.end());
                const size_t o_cp = hostv.size(); hostv.insert(hostv.end(), vlx.chrPos.begin(), vlx.chrPos.end());
                const size_t bytes = hostv.size() * sizeof(int32_t);
                { int r_ = table_alloc(c, d, &d->d_vlines, &d->vlines_bytes, bytes); if (r_ < 0) return r_; }
                { int r_ = table_put(c, d, d->d_vlines, hostv.data(), bytes); if (r_ < 0) return r_; }
                const int32_t *bv = (const int32_t *)d->d_vlines;
                p.vlines = bv; p.nVL = (int32_t)nL; p.vline_mode = mode;
                p.vLumPos = bv + o_lp; p.vChrPos = bv + o_cp;
                p.srcH = (int32_t)nL; p.chrSrcH = (int32_t)nC;   // what the two passes see: one scratch row per (destination row, tap)
                d->vlines_on = true; d->unity_h = false; d->unity_v = false;
                c->gamma_in_reader = mode == 1;
                log_msg(c, 2, "line-schedule dependent context (mode %d): %zu + %zu virtual lines\n", mode, nL, nC);
            }
        }
        // ---- packed 24 / 32 bpp RGB into 8-bit 4:2:0 / 4:2:2 YUV of the same size (sws_k_rgbsrc_unity): identity horizontal filters and luma
        //      vertical filter, chroma of the "half" readers through a vertical filter of up to 16 taps whose positions only move forward ----
        d->rgbsrc_ok = false; d->rgbsrc2_rows = nullptr;
        if (d->unity_h && !d->vlines_on && (p.srcKind == SRCK_RGB24 || p.srcKind == SRCK_RGB32 || p.srcKind == SRCK_GBRP) && p.chr_half && (!p.range_active || (!(p.dstW & 3) && p.range_to_jpeg && !c->tune.no_strip_range && !c->tune.no_rgbsrc2)) && !p.need_alpha && !p.no_chroma &&
            !p.wide && !p.dst_alpha_fill && (p.dstKind == DSTK_PLANAR8 || p.dstKind == DSTK_NV12) && p.dst_bits == 8 && p.chrDstHSub == 1 && p.chrDstVSub <= 1 &&
            p.chrSrcVSub == 0 && p.srcH == p.dstH && bank_is_identity(c->vLum, 1 << 12) && vChrB.size <= 16 && p.chrDstW == ((p.dstW + 1) >> 1) &&
            !p.should_dither && !c->tune.no_rgbsrc) {
            bool fwd = true;
            for (int k = 0; k < 9; k++) fwd = fwd && p.rgb2yuv[k] > -32768 && p.rgb2yuv[k] < 32768;   // (v_dot2_i32_i16 operands)
            for (int y = 0; y < vChrB.count && fwd; y++) fwd = vChrB.pos[y] >= 0 && (y == 0 || vChrB.pos[y] >= vChrB.pos[y - 1]);
            if (fwd) {
                const bool x_form = p.dstKind == DSTK_NV12 || vChrB.size > 1;   // yuv2nv12cX_c has no one-tap form
                std::vector<SwsRgbSrcRow> rows((size_t)vChrB.count);
                for (int y = 0; y < vChrB.count; y++) {
                    SwsRgbSrcRow &e = rows[(size_t)y];
                    std::memset(&e, 0, sizeof(e));
                    e.first = std::max(1 - vChrB.size, vChrB.pos[y]);
                    e.last = e.first + vChrB.size - 1;
                    for (int j = 0; j < vChrB.size; j++)
                        e.vt[j >> 1] |= (uint32_t)(uint16_t)(x_form ? vChrB.taps[(size_t)y * vChrB.size + j] : 1) << (16 * (j & 1));
                }
                // the wave-march form (sws_k_rgbsrc_unity2): per chroma row the first source-row PAIR and the tap pairs aligned to even source rows, laid out
                // against the newest slots of its register ring (1 / 3 / 5 / 8 pairs); the planar one-tap form enters as the tap 4096
                std::vector<SwsStripRow> rows2;
                int npv2 = 1;
                for (int y = 0; y < vChrB.count; y++) npv2 = std::max(npv2, ((vChrB.pos[y] & 1) + vChrB.size + 1) / 2);
                const int rd2 = npv2 <= 1 ? 1 : npv2 <= 3 ? 3 : npv2 <= 5 ? 5 : 8;
                if (npv2 <= 8) {
                    rows2.resize((size_t)vChrB.count);
                    std::memset(rows2.data(), 0, rows2.size() * sizeof(SwsStripRow));
                    for (int y = 0; y < vChrB.count; y++) {
                        SwsStripRow &e2 = rows2[(size_t)y];
                        e2.pf = (vChrB.pos[y] & ~1) >> 1;
                        for (int j = 0; j < vChrB.size; j++) {
                            const int k = (vChrB.pos[y] & 1) + j + 2 * (rd2 - npv2);
                            const int16_t tap = x_form ? vChrB.taps[(size_t)y * vChrB.size + j] : (int16_t)4096;
                            e2.vt[k >> 1] |= (uint32_t)(uint16_t)tap << (16 * (k & 1));
                        }
                    }
                }
                const size_t bytes1 = (rows.size() * sizeof(SwsRgbSrcRow) + 63) & ~(size_t)63;
                const size_t bytes = bytes1 + rows2.size() * sizeof(SwsStripRow);
                { int r_ = table_alloc(c, d, &d->d_dot2, &d->dot2_bytes, bytes); if (r_ < 0) return r_; }
                { int r_ = table_put(c, d, d->d_dot2, rows.data(), rows.size() * sizeof(SwsRgbSrcRow)); if (r_ < 0) return r_; }
                if (!rows2.empty()) { int r_ = table_put(c, d, (uint8_t *)d->d_dot2 + bytes1, rows2.data(), rows2.size() * sizeof(SwsStripRow)); if (r_ < 0) return r_; }
                d->rgbsrc_rows = (const SwsRgbSrcRow *)d->d_dot2;
                d->rgbsrc2_rows = rows2.empty() ? nullptr : (const SwsStripRow *)((const uint8_t *)d->d_dot2 + bytes1);
                d->rgbsrc2_npv = npv2;
                d->rgbsrc_ok = true;
            }
        }
        // ---- 8-bit RGB (packed 24 / 32 bpp, planar) into planar 8-bit 4:4:4 YUV of the same size: all four banks the identity, full chroma readers ----
        d->rgb444_ok = false;
        if (d->unity_h && d->unity_v && !d->vlines_on && (p.srcKind == SRCK_RGB24 || p.srcKind == SRCK_RGB32 || p.srcKind == SRCK_GBRP) && !p.chr_half && !p.range_active &&
            !p.need_alpha && !p.no_chroma && !p.wide && !p.dst_alpha_fill && p.dstKind == DSTK_PLANAR8 && p.dst_bits == 8 && p.chrDstHSub == 0 && p.chrDstVSub == 0 &&
            p.chrSrcVSub == 0 && !p.should_dither && !c->tune.no_rgbsrc) {
            bool fits = true;
            for (int k = 0; k < 9; k++) fits = fits && p.rgb2yuv[k] > -32768 && p.rgb2yuv[k] < 32768;   // (v_dot2_i32_i16 operands)
            d->rgb444_ok = fits;
        }
        // ---- dot2 tile kernel (sws_k_tile_dot2): planar 8-bit / <= 15-bit sources, 15-bit intermediates, vfs >= 2 ----
        d->dot2_ok = false;
        {
            const bool vlines_pending = d->vlines_on;
            const bool src_ok = p.srcKind == SRCK_PLANAR8 || (p.srcKind == SRCK_PLANAR16 && p.src_depth <= 15 && p.src_shift == 0);
            // nv12 / nv21, p010 / p012 (and the 4:2:2 / 4:4:4 twins): the strip kernel de-interleaves plane 1 (and shifts the p01x samples down) while
            // staging; the dot2 tile kernel does not
            const bool nv_src = (p.srcKind == SRCK_NV12 && c->srcBpc == 8) || (p.srcKind == SRCK_P010 && p.src_depth <= 15);
            // (round 5: samples of 16 significant bits -- yuv4xxp16, gray16, p016 / p216 / p416 -- through the register-staged strip kernels: top bit flipped while
            //  staging, the difference given back as a per-column addend, strip_hstage_b)
            const bool src_u16 = !c->tune.no_strip_u16 && !c->tune.no_strip && c->srcBpc == 16 && p.src_depth == 16 && p.src_shift == 0 && (p.srcKind == SRCK_PLANAR16 || p.srcKind == SRCK_P010);
            // (19-bit intermediates, round 5: destinations of 16 bits per component -- yuv4xxp16, gray16, p016 -- and the int32 sums of the wide planar RGB route,
            //  from sources whose samples are v_dot2 operands as they are: sws_k_strip_wide, k_stripwide.hip)
            const bool wide_dst = p.wide && c->dstBpc >= 16 && !c->tune.no_strip_wide && !c->tune.no_strip &&
                                  ((p.dstKind == DSTK_PLANAR16 && p.dst_shift == 0 && !isALPHA(o.dst_format)) || p.dstKind == DSTK_P016 || (p.dstKind == DSTK_RAW32 && d->fullchr_on == 4));
            const bool dst_ok = ((p.dstKind == DSTK_PLANAR8 || p.dstKind == DSTK_PLANARN || p.dstKind == DSTK_NV12 || p.dstKind == DSTK_P010 || p.dstKind == DSTK_RAW32) && !p.wide) || wide_dst;
            auto fs2 = [](int fs) { return (fs + 2) & ~1; };
            // packed 24 / 32 bpp RGB through the LUT writers (not the full-chroma ones): the strip kernel with the RGB epilogue
            // (9 .. 15-bit planar sources too -- decoded HDR pictures for display: 128-column strips, a window of at most 64 eight-sample chunks)
            const bool rgb_s16 = p.srcKind == SRCK_PLANAR16 && p.src_depth <= 15 && p.src_shift == 0;
            // (a planar YUV source with alpha into a 32 bpp destination with alpha -- yuva420p -> bgra, needAlpha: the kernel stores opaque pixels, the A
            //  samples go through one more luma launch with the raw writer and sws_k_alpha_merge32 puts the bytes in: alpha_launch == 2)
            const bool rgb_alpha = c->needAlpha && p.dstKind == DSTK_RGB32 && isPlanarYUV(o.src_format) && !c->tune.no_mixed;
            const bool rgb_ok = (p.dstKind == DSTK_RGB24 || p.dstKind == DSTK_RGB32) && !p.full_chr && ((p.srcKind == SRCK_PLANAR8 && c->srcBpc == 8) || rgb_s16) &&
                                !p.no_chroma && (!p.need_alpha || rgb_alpha) && !(p.dstW & 1) && !p.range_active && !c->tune.no_strip;
            d->striprgb_ok = false; d->striprgb_direct = 0;
            // (planar writers: a one-tap vertical filter takes the reference's yuv2plane1 form -- (s + d) >> 7, (s + (1 << (14 - bits))) >> (15 - bits),
            //  output.c:327-341, :485-493 -- which is the "X" arithmetic of these kernels with the one tap 4096: (4096 s + (d << 12)) >> 19, sample for
            //  sample; the strip kernel takes such planes (4:2:0 -> 4:2:2 at half the size, horizontal-only scaling), the dot2 tile kernel keeps its
            //  two-tap minimum.  The packed writers are in "X" mode unless both vertical filters are short, which all_x_mode checks row by row)
            // scaled packed 24 / 32 bpp RGB sources: a reader pre-pass writes the 16-bit planes the horizontal scaler
            // reads, the strip kernel takes them like a planar 16-bit source (launch_rgbread_strip); other shapes of these sources keep the tile kernel
            // (round 4: the other RGB sources whose readers deliver the same 15-bit lines to the same hScale16To15_c -- x2rgb10 / x2bgr10, the 16 / 15 / 12 bpp
            //  formats (rgb16_32ToY/UV(_half)_c_template, input.c:264-412), planar RGB of 9 - 14 bits (planar_rgb16_s16_to_y / _uv, :1216-1270) -- through the
            //  per-kind element-per-thread reader (k_generic_kinds.hip sws_k_read16_kind); without an alpha plane)
            //  (... and the packed YUV sources of 10 / 12 bits -- y210 / y212, xv30 / v30x, xv36: read_*_c, y21xle_Y/UV_c, input.c:580-606, :663-729, :811-866 -- whose
            //  lines are those of a planar yuv422p10 / yuv444p10 / ...12 picture, sh = depth - 1: the pre-pass de-interleaves them)
            const bool rgbread_kindN = (p.srcKind == SRCK_RGB30 || p.srcKind == SRCK_RGB16 || (p.srcKind == SRCK_GBRP16 && (p.src_depth < 16 || !c->tune.no_strip_u16)) ||
                                        // (round 5: rgb48 / rgba64, planar RGB of 16 bits and float32 -- lines of 16 significant bits, see src_u16)
                                        ((p.srcKind == SRCK_RGB48 || p.srcKind == SRCK_GBRPF32) && !c->tune.no_strip_u16) ||
                                        (p.srcKind == SRCK_PACKEDHI && p.src_depth >= 9 && p.src_depth <= 15 && c->srcBpc == p.src_depth) ||
                                        // (the 8-bit packed 4:4:4 formats -- ayuv / vuya / vuyx / uyva / vyu444: bytes, hScale8To15_c's sh = 7 -- as 16-bit words with 8 significant bits)
                                        (p.srcKind == SRCK_PACKED444 && p.src_depth == 8 && c->srcBpc == 8)) && !p.need_alpha && !c->needAlpha &&   // (an alpha component nobody reads is skipped)
                                       !c->tune.no_rgbread_kinds;
            bool rgbread = (!p.wide || !c->tune.no_strip_wide) && (p.srcKind == SRCK_RGB24 || p.srcKind == SRCK_RGB32 || p.srcKind == SRCK_GBRP || rgbread_kindN) && p.chrSrcHSub <= 1 && p.chrSrcW == (p.srcW >> p.chrSrcHSub) && !(p.srcW & 1) && p.chrSrcVSub == 0 && (!p.need_alpha || d->fullchr_on == 2 || alpha_planar) &&
                           (!p.dst_alpha_fill || d->fullchr_on) && (!p.no_chroma || (isGray(o.dst_format) && !isGray(o.src_format) && !c->tune.no_strip_range)) && !vlines_pending && dst_ok && !c->tune.no_strip && !c->tune.no_rgbsrc && p.dstW >= strip_min_w_eff;
            for (int k = 0; k < 9 && rgbread; k++) rgbread = p.rgb2yuv[k] > -32768 && p.rgb2yuv[k] < 32768;   // (v_dot2_i32_i16 operands)
            d->rgbread_on = false;
            // vertical chroma filters of 17 .. 24 taps (a 4:1 chroma step: packed RGB or 4:2:2 sources into a 4:2:0 picture of half the size): the strip
            // kernel's chroma instantiations with a ring of 12 row pairs; the tile kernel and the RGB epilogue stop at 16
            const bool vchr_long = fs2(vChrB.size) > 16 && fs2(vChrB.size) <= 24 && dst_ok && !rgb_ok && !c->tune.no_strip;
            // gray -> gray (8 .. 14 bit): one plane, the strip kernel's luma launch alone
            // (round 5: ... and planar / semi-planar YUV -> gray: the destination has no chroma planes, so the conversion is the luma launch as well -- thumbnails
            //  for analysis; it needed the range conversion in the strip kernels, gray8 being full range, handle_jpeg utils.c:773-809)
            // (round 5: ... and gray sources into planar / semi-planar YUV: the luma launch, then sws_k_gray_chroma writes what the reference's chroma writers make of
            //  their constant lines -- launch_plan_le_batch)
            const bool gray_src = isGray(o.src_format) && !isGray(o.dst_format) && !c->needAlpha && (!isALPHA(o.dst_format) || (p.dstKind == DSTK_RAW32 && (d->fullchr_on == 3 || d->fullchr_on == 1))) && (src_ok || src_u16) && !c->tune.no_strip_range &&
                                  (((p.dstKind == DSTK_PLANAR8 || p.dstKind == DSTK_PLANARN || p.dstKind == DSTK_NV12 || p.dstKind == DSTK_P010) && !p.wide) ||
                                   ((p.dstKind == DSTK_PLANAR16 || p.dstKind == DSTK_P016) && wide_dst) || (p.dstKind == DSTK_RAW32 && (d->fullchr_on == 3 || d->fullchr_on == 1) && d->fullchr_kind != DSTK_GBRP)) && !d->join422 &&
                                  (!d->fullchr_on || d->fullchr_on == 3 || d->fullchr_on == 1) && !c->tune.no_strip;
            const bool gray_both = gray_src || (isGray(o.dst_format) && (isGray(o.src_format) || ((src_ok || nv_src || src_u16 || rgbread) && !c->tune.no_strip_range)) && !c->needAlpha && (src_ok || nv_src || src_u16 || rgbread) &&
                                   (((p.dstKind == DSTK_PLANAR8 || p.dstKind == DSTK_PLANARN) && !p.wide) || (p.dstKind == DSTK_PLANAR16 && wide_dst)) && !c->tune.no_strip);
            // identity luma filters + scaled chroma (yuv422p -> yuv420p, yuv444p -> yuv420p, the 10-bit -> 8-bit twins ...): the luma plane streams
            // (one tap: a per-sample pass), only the chroma planes need the strip kernel
            const bool mixedM = !vlines_pending && !d->fullchr_on && bank_is_identity(hLumB, 1 << 14) && bank_is_identity(c->vLum, 1 << 12) && !(d->unity_h && d->unity_v) &&
                                !p.fast_bilinear && !gray_any && (src_ok || nv_src) && dst_ok && !p.wide && (!p.range_active || (c->srcBpc == 8 && p.dst_bits == 8 && !c->tune.no_strip_range)) && !p.dst_alpha_fill &&
                                fs2(hChrB.size) <= 16 && (fs2(vChrB.size) <= 16 || vchr_long) && !c->tune.no_strip && !c->tune.no_mixed && p.dstW >= c->tune.strip_min_w;
            // (identity horizontal filters: 8-bit sources have kernels of their own -- sws_k_rgb_march, sws_k_rgbsrc_unity, the mixed plan -- but a 10-bit
            //  picture into packed RGB (decoded HDR for display) or packed RGB into a 10-bit 4:2:0 picture at the same size had only the generic
            //  kernels: the strip kernels take them with their one-tap horizontal banks)
            // (... and so had every planar / semi-planar YUV -> YUV conversion whose horizontal filters are the identity and which the mixed plan does not
            //  take: all four filters the identity -- p010le -> yuv420p10le, nv12 -> yuv420p10le, yuv420p10le -> nv12, nv12 <-> nv21, bgra -> yuv444p10le: pure
            //  per-sample conversions the reference has no special converter for -- or vertical-only scaling)
            const bool unity_yuv = d->unity_h && !d->rgbsrc_ok && !d->rgb444_ok && dst_ok && (src_ok || nv_src || rgbread || src_u16) && !c->tune.no_mixed;
            const bool unity_ok = !d->unity_h || (rgb_ok && rgb_s16) || (rgbread && !d->rgbsrc_ok && !d->rgb444_ok) || unity_yuv;   // (sws_k_rgbsrc_unity's row table lives in the same device block as the strip plan)
            // filters of 17 .. 32 taps (ratios of 4:1 and more -- the lower rungs of an ABR ladder, thumbnails: bicubic at 4:1 has 17 taps, at 6:1 25; Lanczos at
            // 3:1 19): the strip kernel's long form (sws_k_strip_long: 16 tap pairs each way, strips of 128 / 64 columns); the RGB epilogue stops at 16
            const bool fs_ok16 = fs2(hLumB.size) <= 16 && (fs2(hChrB.size) <= 16 || gray_both) && fs2(c->vLum.size) <= 16 && (fs2(vChrB.size) <= 16 || vchr_long || gray_both);
            const bool fs_ok32 = fs2(hLumB.size) <= 32 && fs2(hChrB.size) <= 32 && fs2(c->vLum.size) <= 32 && fs2(vChrB.size) <= 48 && dst_ok && !rgb_ok && !gray_both &&
                                 !c->tune.no_strip && !c->tune.no_mixed;
            const bool fs_ok64 = fs2(hLumB.size) <= 64 && fs2(hChrB.size) <= 64 && fs2(c->vLum.size) <= 64 && fs2(vChrB.size) <= 64 && dst_ok && !rgb_ok && !gray_both &&
                                 !c->tune.no_strip && !c->tune.no_mixed;      // (33 .. 62 taps: the extra-long form, 32 pairs each way on strips of 64 columns)
            const bool wide_ok = wide_dst && (src_ok || nv_src || src_u16 || rgbread) && !(p.range_active && c->tune.no_strip_range) && !c->needAlpha && !p.need_alpha && !p.fast_bilinear && !vlines_pending && fs_ok16 &&
                                 !(p.srcKind == SRCK_PLANAR8 && c->srcBpc != 8);
            const bool fullA = !mixedM && unity_ok && !(d->unity_h && d->unity_v && !rgb_ok && !unity_yuv) && !p.fast_bilinear && (!gray_any || gray_both || d->fullchr_on == 2 || alpha_planar || (rgb_ok && rgb_alpha)) && (src_ok || (nv_src && dst_ok) || rgbread || (src_u16 && dst_ok)) &&
                               (dst_ok || rgb_ok) && (!p.wide || wide_ok) && (fs_ok16 || fs_ok32 || fs_ok64) && !c->tune.no_dot2;
            const int long_form = !fullA || fs_ok16 ? 0 : fs_ok32 ? 1 : 2;
            d->mixed_ok = false; d->stripLs_ok = d->stripCs_ok = false; d->striprgbsrc_ok = false; d->rgb2rgb_ok = false;
            // (a gray source into planar / semi-planar YUV at the same size -- a monochrome camera into an encoder: the luma plane is the mixed plan's streaming pass, the
            //  chroma planes are sws_k_gray_chroma's constants; no strip plan at all.  launch_mixed tells the two by the source format)
            const bool gray_mixed = gray_src && !vlines_pending && !d->fullchr_on && bank_is_identity(hLumB, 1 << 14) && bank_is_identity(c->vLum, 1 << 12) && !p.fast_bilinear && src_ok && dst_ok &&
                                    p.dstKind != DSTK_RAW32 && !p.wide && (!p.range_active || (c->srcBpc == 8 && p.dst_bits == 8)) && !p.dst_alpha_fill && !c->tune.no_mixed;
            if (gray_mixed) d->mixed_ok = true;
            else if (fullA || mixedM) {
                const int SPC = (p.srcKind == SRCK_PLANAR16 || p.srcKind == SRCK_P010 || rgbread) ? 8 : 16;
                std::vector<uint8_t> blob;
                auto put = [&](const void *ptr, size_t n) { size_t o = (blob.size() + 15) & ~(size_t)15; blob.resize(o + n); std::memcpy(blob.data() + o, ptr, n); return o; };
                auto padded = [&](const FilterBank &b, int f2_long = 0) {
                    const int f2 = f2_long ? f2_long : fs2(b.size);
                    std::vector<int16_t> t((size_t)b.count * f2, 0);
                    for (int i = 0; i < b.count; i++)
                        for (int j = 0; j < b.size; j++) t[(size_t)i * f2 + (b.pos[i] & 1) + j] = b.taps[(size_t)i * b.size + j];
                    return t;
                };
                struct Off { size_t rs, rc, cs, cc, ht, vt; };
                auto plan2 = [&](const FilterBank &hb, const FilterBank &vb, int W, int H, int ncomp, SwsTileGeom &g, Off &o) -> bool {
                    const int TW = 128, hf2 = fs2(hb.size), vf2 = fs2(vb.size);
                    for (int TH : { 64, 32, 16, 8, 4, 2 }) {
                        const int tX = (W + TW - 1) / TW, tY = (H + TH - 1) / TH;
                        std::vector<int32_t> rs(tY), rc(tY), cs(tX), cc(tX);
                        int nrmax = 0, ncmax = 0;
                        for (int t = 0; t < tY; t++) {
                            int lo = INT32_MAX, hi = -1;
                            for (int y = t * TH; y < std::min(H, (t + 1) * TH); y++) { lo = std::min(lo, vb.pos[y] & ~1); hi = std::max(hi, (vb.pos[y] & ~1) + vf2); }
                            rs[t] = lo; rc[t] = (hi - lo + 1) & ~1; nrmax = std::max(nrmax, rc[t]);
                        }
                        for (int t = 0; t < tX; t++) {
                            int lo = INT32_MAX, hi = -1;
                            for (int x = t * TW; x < std::min(W, (t + 1) * TW); x++) { lo = std::min(lo, hb.pos[x] & ~1); hi = std::max(hi, (hb.pos[x] & ~1) + hf2); }
                            lo = lo / SPC * SPC;
                            cs[t] = lo; cc[t] = (hi - lo + SPC - 1) / SPC * SPC; ncmax = std::max(ncmax, cc[t]);
                        }
                        const size_t lds = (size_t)nrmax * ncmax * 2 + (size_t)ncomp * (nrmax / 2) * TW * 4;
                        const int lds_budget = c->tune.tile_lds_kb;
                        if (lds > (size_t)lds_budget * 1024 && TH > 2) continue;
                        if (lds > 64 * 1024) return false;
                        g.TW = TW; g.TH = TH; g.tilesX = tX; g.tilesY = tY; g.NRmax = nrmax; g.NCmax = ncmax; g.lds_bytes = (int32_t)lds;
                        g.hfs2 = hf2; g.vfs2 = vf2;
                        g.debug = c->tune.debug;
                        o.rs = put(rs.data(), rs.size() * 4); o.rc = put(rc.data(), rc.size() * 4);
                        o.cs = put(cs.data(), cs.size() * 4); o.cc = put(cc.data(), cc.size() * 4);
                        const std::vector<int16_t> ht = padded(hb), vt = padded(vb);
                        o.ht = put(ht.data(), ht.size() * 2); o.vt = put(vt.data(), vt.size() * 2);
                        return true;
                    }
                    return false;
                };
                // marching strip kernel: strips of 64 * cols output columns; window of a strip <= 128 chunks of 16 bytes
                struct SOff { size_t cs, cc, rows; };
                // ring_of(npv) != 0: the kernel multiplies a whole ring of that depth per output sample (sws_k_strip_rgb): the row's tap pairs
                // are laid out against the newest npv slots, the older slots get zero taps
                // plane1_form: the plane's writer has the reference's one-tap form (yuv2plane1_*, yuv2p01xl1_c: (s + d) >> 7 and its N-bit twins), which
                // never looks at the coefficient -- initFilter's error-diffused normalisation leaves 4095 in some one-tap rows -- so a one-tap bank
                // enters the X arithmetic as 4096; the semi-planar chroma writers (yuv2nv12cX_c, yuv2p01xcX_c) have no such form and take the bank's value
                const std::vector<int32_t> *plan_rnd = nullptr;      // per output row: SwsStripRow::rnd_off of the plans made while it is set (the packed writers' short forms below)
                auto plan3 = [&](const FilterBank &hb, const FilterBank &vb, int W, int cols, int ncomp, SwsStripGeom &g, SOff &o, int (*ring_of)(int) = nullptr, bool plane1_form = false, int longf = 0, int tw_over = 0) -> bool {   // longf: 1 the long form (16 / 24 pairs), 2 the extra-long one (32 / 32); tw_over: strips narrower than 64 * cols columns (the lockstep kernels: a window that fits one reader turn less)
                    const int TW = tw_over ? tw_over : 64 * cols, hf2 = fs2(hb.size), vf2 = fs2(vb.size);
                    const int strips = (W + TW - 1) / TW;
                    std::vector<int32_t> cs(strips), cc(strips);
                    int ncmax = 0, nph = 1, npv = 1;
                    for (int t = 0; t < strips; t++) {
                        int lo = INT32_MAX, hi = -1;
                        for (int x = t * TW; x < std::min(W, (t + 1) * TW); x++) { lo = std::min(lo, hb.pos[x] & ~1); hi = std::max(hi, (hb.pos[x] & ~1) + hf2); }
                        if (lo < 0) return false;
                        lo = lo / SPC * SPC;
                        cs[t] = lo; cc[t] = (hi - lo + SPC - 1) / SPC * SPC; ncmax = std::max(ncmax, cc[t]);
                    }
                    if (ncmax / SPC > (ncomp == 2 ? 64 : 128)) return false;   // one (chroma) or two (luma) 16-byte chunks per lane and row
                    for (int x = 0; x < hb.count; x++) nph = std::max(nph, ((hb.pos[x] & 1) + hb.size + 1) / 2);
                    for (int y = 0; y < vb.count; y++) { if (vb.pos[y] < 0) return false; npv = std::max(npv, ((vb.pos[y] & 1) + vb.size + 1) / 2); }
                    if (npv > (longf == 2 ? 32 : longf ? (ncomp == 2 ? 24 : 16) : ncomp == 2 && !ring_of ? 12 : 8)) return false;   // ring depth of the instantiations: 8 row pairs, 12 for the planar chroma planes, 16 in the long form
                    if (longf == 2) { nph = std::max(20, (nph + 3) & ~3); if (nph > 32) return false; }   // (20 / 24 / 28 / 32 pairs)
                    else if (longf) { nph = std::max(10, (nph + 1) & ~1); if (nph > 16) return false; }   // (the long form's instantiations: 10 / 12 / 14 / 16 horizontal tap pairs, zero-padded rows)
                    else if (nph > 8) return false;
                    for (int y = 1; y < vb.count; y++) if (vb.pos[y] < vb.pos[y - 1]) return false;   // the ring only moves forward
                    g.TW = TW; g.strips = strips; g.NCmax = ncmax; g.nph = nph; g.npv = npv; g.hfs2 = longf ? 2 * nph : hf2; g.vfs2 = vf2;
                    g.lds_bytes = 4 * ncomp * 2 * ((ncmax + SPC) / 2) * 4;
                    g.dma8_ok = 0; g.nph8 = 0; g.lds_dma8_bytes = 0; g.hT8 = nullptr;     // (the byte-row LDS-DMA form: plan3_alt)
                    // LDS-DMA form: a ring of 4 row pairs per wave, rows of ncmax 16-bit samples; every pair between the first and the last
                    // one a band needs is requested, so the windows of consecutive rows must touch (no skipped pair)
                    g.lds_dma_bytes = 4 * 4 * ncomp * 2 * (ncmax / 2) * 4;
                    g.dma_ok = !longf && (p.srcKind == SRCK_PLANAR16 || rgbread) && p.src_depth < 16 && g.lds_dma_bytes <= 40 * 1024;   // (LDS-DMA copies rows as they are: planar 16-bit sources, and the reader planes of a packed RGB source)
                    for (int y = 1; y < vb.count && g.dma_ok; y++)
                        if (((vb.pos[y] & ~1) >> 1) > ((vb.pos[y - 1] & ~1) >> 1) + npv) g.dma_ok = 0;
                    o.cs = put(cs.data(), cs.size() * 4); o.cc = put(cc.data(), cc.size() * 4);
                    const int epr = longf == 2 ? 4 : longf ? 2 : 1;   // 64-byte entries per row: the long form's 16 tap pairs run on into a second one
                    std::vector<SwsStripRow> rows((size_t)vb.count * epr);
                    std::memset(rows.data(), 0, rows.size() * sizeof(SwsStripRow));
                    for (int y = 0; y < vb.count; y++) {
                        SwsStripRow &e = rows[(size_t)y * epr];
                        e.pf = (vb.pos[y] & ~1) >> 1;
                        if (plan_rnd && (size_t)y < plan_rnd->size()) e.rnd_off = (*plan_rnd)[(size_t)y];
                        const int lead = ring_of ? 2 * (ring_of(npv) - npv) : 0;
                        if (lead < 0) return false;
                        for (int j = 0; j < vb.size; j++) {
                            const int k = (vb.pos[y] & 1) + j + lead;
                            // (pairs 8 .. 11 of a long chroma filter land in the four spare dwords behind vt[8]: load_strip_row_n)
                            const int16_t tap = (vb.size == 1 && plane1_form) ? (int16_t)4096 : vb.taps[(size_t)y * vb.size + j];
                            reinterpret_cast<uint32_t *>(rows.data())[(size_t)y * epr * 16 + 4 + (k >> 1)] |= (uint32_t)(uint16_t)tap << (16 * (k & 1));
                        }
                    }
                    o.rows = put(rows.data(), rows.size() * sizeof(SwsStripRow));
                    return true;
                };
                // The same plan on strips of another width, for the short-filter instantiations (k_strip2.hip: 8-bit sources, at most 6 tap pairs each way,
                // windows of at most 64 chunks): 3 .. 5 luma / 1 .. 3 chroma columns per lane, whichever leaves the fewest idle lane-columns in the last
                // strip (640 columns: 2 strips of 320 instead of 2.5 of 256); only colStart / colCount differ from the base plan
                auto plan3_alt = [&](const FilterBank &hb, const FilterBank &vb, int W, int ncomp, const SwsStripGeom &base, SwsStripGeom &alt, SOff &o) -> bool {
                    if (c->tune.no_strip_short || SPC != 16 || (p.srcKind != SRCK_PLANAR8 && p.srcKind != SRCK_NV12) || base.npv > 8 || base.nph > 7) return false;
                    const int hf2 = fs2(hb.size);
                    int best = 0; int64_t best_cost = INT64_MAX;
                    std::vector<int32_t> bcs, bcc; int bnc = 0;
                    const int forced = ncomp == 2 ? c->tune.strip_cols_c : c->tune.strip_cols_l;
                    // (the LDS-DMA form -- planar sources, no skipped row pair -- needs fewer registers per column and takes wider strips)
                    const bool nvc = ncomp == 2 && p.srcKind == SRCK_NV12;      // interleaved chroma bytes: two bytes per sample in the DMA'd rows
                    bool dma8 = !c->tune.no_strip_dma8 && (hb.size + 1) / 2 <= 6;
                    for (int y = 1; y < vb.count && dma8; y++)
                        if (((vb.pos[y] & ~1) >> 1) > ((vb.pos[y - 1] & ~1) >> 1) + base.npv) dma8 = false;
                    const std::vector<int> cand = dma8 ? (nvc ? std::vector<int>{ 3, 2, 1 } : ncomp == 2 ? std::vector<int>{ 5, 4, 3, 2, 1 } : std::vector<int>{ 7, 6, 5, 4, 3 })
                                                       : (ncomp == 2 ? std::vector<int>{ 3, 2, 1 } : std::vector<int>{ 5, 4, 3 });
                    for (int cols : cand) {
                        if (!c->tune.strip_cols_auto && cols != forced) continue;
                        const int TW = 64 * cols, strips = (W + TW - 1) / TW;
                        std::vector<int32_t> cs(strips), cc(strips);
                        int ncmax = 0; bool ok = true;
                        for (int t = 0; t < strips && ok; t++) {
                            int lo = INT32_MAX, hi = -1;
                            for (int x = t * TW; x < std::min(W, (t + 1) * TW); x++) { lo = std::min(lo, hb.pos[x] & ~1); hi = std::max(hi, (hb.pos[x] & ~1) + hf2); }
                            if (lo < 0) { ok = false; break; }
                            lo = lo / SPC * SPC;
                            cs[t] = lo; cc[t] = (hi - lo + SPC - 1) / SPC * SPC; ncmax = std::max(ncmax, cc[t]);
                        }
                        if (!ok || ncmax / SPC > 64 || (nvc && dma8 && 2 * ncmax / SPC > 64)) continue;
                        // what a launch pays per row: every strip its columns plus a fixed share (staging / requests, plan entry, waits, stores):
                        // about two columns' worth (measured on C1: chroma strips of 64 / 128 / 192 columns, 5 / 3 / 2 per row)
                        const int64_t cost = (int64_t)strips * (2 * cols + 3);
                        if (cost < best_cost) { best_cost = cost; best = cols; bcs = cs; bcc = cc; bnc = ncmax; }
                    }
                    if (!best) return false;
                    alt = base;
                    alt.TW = 64 * best; alt.strips = (W + alt.TW - 1) / alt.TW; alt.NCmax = bnc;
                    alt.lds_bytes = 4 * ncomp * 2 * ((bnc + SPC) / 2) * 4; alt.dma_ok = 0;
                    o.cs = put(bcs.data(), bcs.size() * 4); o.cc = put(bcc.data(), bcc.size() * 4); o.rows = 0;
                    // LDS-DMA form (kernels_strip8.hpp): raw byte rows, a ring of 4 row pairs per wave; planar sources only (semi-planar chroma bytes are
                    // interleaved); every pair between the first and the last one a band needs is requested, so no pair may be skipped.  Its tap rows
                    // start at the filter's own first tap (no even-position padding): o.rows carries their offset in the blob
                    alt.nph8 = (hb.size + 1) / 2;
                    // (4 waves per block x ring depth x rows of a pair x dwords of a row; the chroma rings are 2 pairs deep: kernels_strip8.hpp)
                    const int depth8 = ncomp == 2 ? 2 : 4;
                    alt.lds_dma8_bytes = nvc ? 4 * depth8 * 2 * ((2 * bnc + 16) / 4) * 4 : 4 * depth8 * ncomp * 2 * ((bnc + 16) / 4) * 4;
                    alt.dma8_ok = dma8 && alt.lds_dma8_bytes <= 48 * 1024;
                    if (alt.dma8_ok) {
                        const int f8 = 2 * alt.nph8;
                        std::vector<int16_t> t8((size_t)hb.count * f8, 0);
                        for (int i = 0; i < hb.count; i++)
                            for (int j = 0; j < hb.size; j++) t8[(size_t)i * f8 + j] = hb.taps[(size_t)i * hb.size + j];
                        o.rows = put(t8.data(), t8.size() * 2);
                    }
                    return true;
                };
                d->stripLs_ok = d->stripCs_ok = false;
                SOff sLs, sCs;
                SOff sL, sC;
                // (raw sums: the packed X form's own taps -- except when both vertical filters have one tap: the packed writers then take their "_1" forms
                //  (yuv2rgb_full_1, yuv2rgb_1: vscale.c:136-141), which ignore the coefficients like yuv2plane1 does and equal the X arithmetic with the tap
                //  4096: Y = buf << 2 == (buf << 12 + (1 << 9)) >> 10, (buf + 64) >> 7 == (buf << 12 + (1 << 18)) >> 19; planar RGB has no such form, vscale.c:173-212)
                //  The packed YUV formats of 10 / 12 bits (DSTK_PACKEDHI) have X writers only, which multiply by the bank's value even when it is the only tap
                //  (4095 after initFilter's normalisation, 0 in the zero-vector rows of a source of fewer than four rows with shifted chroma): their own taps.
                const bool raw_one_one = p.dstKind == DSTK_RAW32 && c->vLum.size == 1 && vChrB.size == 1 && d->fullchr_kind != DSTK_GBRP && d->fullchr_kind != DSTK_PACKEDHI && d->fullchr_kind != DSTK_GBRP16 && d->fullchr_kind != DSTK_GBRPF32;
                const bool chr_plane1 = (p.dstKind != DSTK_NV12 && p.dstKind != DSTK_P010 && p.dstKind != DSTK_P016 && p.dstKind != DSTK_RAW32) || raw_one_one, lum_plane1 = p.dstKind != DSTK_RAW32 || raw_one_one;
                const int strip_cols_l = c->tune.strip_cols_l == 2 ? 2 : 4;
                const int strip_cols_c = c->tune.strip_cols_c == 1 ? 1 : 2;
                // (narrow pictures leave most of a 256-column strip idle and pay the per-band ring fill: the tile kernel keeps them)
                const int strip_min_w = c->tune.strip_min_w;
                // the 19-bit kernel's strips: 128 luma columns, 128 chroma columns where the window fits one 16-byte chunk per lane, else 64
                auto wide_cols = [&](const FilterBank &hb, const FilterBank &vb, int W, bool chroma) {
                    int npvw = 1;
                    for (int y = 0; y < vb.count; y++) npvw = std::max(npvw, ((vb.pos[y] & 1) + vb.size + 1) / 2);
                    const int hf2 = fs2(hb.size);
                    for (int cols : { 2, 1 }) {      // (measured: 256-column strips spill at 128 registers -- the rings hold two int32 rows per pair: bench w1 0.22 -> 0.08; 128-column chroma strips
                        //  with a ring of 4 pairs: w1 0.21 -> 0.26; with a ring of 8 pairs they spill too: yuv420p 4K -> yuv420p16le 1080p 0.0109 -> 0.0256 ms / frame)
                        if (chroma ? (cols == 2 && npvw > 4) : cols == 1) continue;
                        if (c->tune.strip_cols_auto == 0 && cols != (chroma ? (c->tune.strip_cols_c == 1 ? 1 : 2) : (c->tune.strip_cols_l == 2 ? 2 : 4))) continue;   // (experiments / tests: forced widths)
                        const int TW = 64 * cols; int ncmax = 0;
                        for (int t = 0; t * TW < W; t++) {
                            int lo = INT32_MAX, hi = -1;
                            for (int x = t * TW; x < std::min(W, (t + 1) * TW); x++) { lo = std::min(lo, hb.pos[x] & ~1); hi = std::max(hi, (hb.pos[x] & ~1) + hf2); }
                            lo = std::max(lo, 0) / SPC * SPC;
                            ncmax = std::max(ncmax, (hi - lo + SPC - 1) / SPC * SPC);
                        }
                        if (ncmax / SPC <= 64) return cols;
                    }
                    return chroma ? 1 : 2;
                };
                const int wcl = p.wide ? wide_cols(hLumB, c->vLum, p.dstW, false) : 0, wcc = (p.wide && !gray_both) ? wide_cols(hChrB, vChrB, p.chrDstW, true) : 0;
                const bool strip_plan = fullA && dst_ok && !(p.range_active && c->tune.no_strip_range) && !c->tune.no_strip && p.dstW >= (long_form ? strip_min_w_eff : strip_min_w) &&   // (the long form's strips are 128 columns, and what it replaces is the element-per-thread tile kernel)
                                        plan3(hLumB, c->vLum, p.dstW, p.wide ? wcl : long_form == 2 ? 1 : long_form ? 2 : strip_cols_l, 1, d->stripL, sL,
                                              p.wide ? +[](int n) { return n <= 2 ? 2 : n <= 4 ? 4 : 8; } : long_form == 2 ? +[](int) { return 32; } : nullptr, lum_plane1, long_form) &&
                                        (gray_both || plan3(hChrB, vChrB, p.chrDstW, p.wide ? wcc : long_form ? 1 : strip_cols_c, 2, d->stripC, sC,
                                                            p.wide ? +[](int n) { return n <= 2 ? 2 : n <= 4 ? 4 : 8; } : long_form == 2 ? +[](int) { return 32; } : long_form ? +[](int) { return 24; } : nullptr, chr_plane1, long_form)) &&
                                        (!p.wide || (d->stripL.NCmax / SPC <= 64 && (gray_both || d->stripC.NCmax / SPC <= 64)));      // (the wide kernel stages one chunk per lane and row)
                d->strip_ok = false;
                log_msg(c, 3, "strip plan: %d (windows %d/%d chunks of %d, taps %d/%d x %d/%d)\n", strip_plan, d->stripL.NCmax / SPC, d->stripC.NCmax / SPC, SPC,
                        d->stripL.nph, d->stripC.nph, d->stripL.npv, d->stripC.npv);
                Off oL, oC;
                if (mixedM) {
                    SOff sM;
                    if (plan3(hChrB, vChrB, p.chrDstW, strip_cols_c, 2, d->stripC, sM, nullptr, chr_plane1)) {
                        const std::vector<int16_t> htc = padded(hChrB);
                        const size_t ohc = put(htc.data(), htc.size() * 2);
                        const bool altC = plan3_alt(hChrB, vChrB, p.chrDstW, 2, d->stripC, d->stripCs, sCs);
                        { int r_ = table_alloc(c, d, &d->d_dot2, &d->dot2_bytes, blob.size()); if (r_ < 0) return r_; }
                        { int r_ = table_put(c, d, d->d_dot2, blob.data(), blob.size()); if (r_ < 0) return r_; }
                        const uint8_t *b = (const uint8_t *)d->d_dot2;
                        d->stripC.colStart = (const int32_t *)(b + sM.cs); d->stripC.colCount = (const int32_t *)(b + sM.cc);
                        d->stripC.rows = (const SwsStripRow *)(b + sM.rows);
                        d->stripC.hT2 = (const int16_t *)(b + ohc); d->stripC.vT2 = nullptr;
                        if (altC) { d->stripCs.colStart = (const int32_t *)(b + sCs.cs); d->stripCs.colCount = (const int32_t *)(b + sCs.cc);
                                    d->stripCs.rows = d->stripC.rows; d->stripCs.hT2 = d->stripC.hT2; d->stripCs.vT2 = nullptr; d->stripCs.hT8 = (const int16_t *)(b + sCs.rows); d->stripCs_ok = true; }
                        d->mixed_ok = true;
                    }
                } else
                if (rgb_ok) {
                    // RGB epilogue: 256 luma columns + the 128 chroma columns under them per wave, one 16-byte chunk per lane and row for
                    // both (windows of up to 1024 source samples), rings of 5 / 3 row pairs (8 / 8 in the long form)
                    SOff rL, rC;
                    SwsStripGeom &gl = d->stripRL, &gc = d->stripRC;
                    auto ringL = [](int npv) { return npv <= 5 ? 5 : npv <= 8 ? 8 : -1; };
                    auto ringC = [](int npv) { return npv <= 1 ? 1 : npv <= 3 ? 3 : npv <= 8 ? 8 : -1; };
                    const int rcl = (c->tune.strip_rgb_cols == 2 || rgb_s16) ? 2 : 4;
                    // (one vertical tap each: yuv2rgb_1_c_template's (buf + 64) >> 7 -- the coefficient is never looked at -- is the X arithmetic with the tap 4096)
                    const bool rgb_one_one = c->vLum.size == 1 && vChrB.size == 1;
                    // The writers' short forms (packed_vscale, vscale.c:135-157, picks per output row): one luma tap with two chroma taps that sum to 4096 is yuv2rgb_1_c_template
                    // with a chroma blend -- (u0 (4096 - a) + u1 a + (128 << 11)) >> 19, output.c:1913-1937: the X arithmetic on the bank's own taps, the luma tap
                    // taken as 4096; two taps each that sum to 4096 (bilinear up-scaling: a player's 720p -> 1080p into bgra) is yuv2rgb_2_c_template, the X
                    // arithmetic without the rounding constant (SwsStripRow::rnd_off).  Round 5; not with an alpha plane (its own formulas there).
                    FilterBank vLumS, vChrS;
                    std::vector<int32_t> rnd_rows;
                    bool short_rows = false;
                    if (!rgb_one_one && !c->needAlpha && !c->tune.no_short_forms && p.chrDstH == p.dstH && (c->vLum.size == 1 || c->vLum.size == 2) && vChrB.size == 2) {
                        vLumS = c->vLum; vChrS = vChrB;
                        rnd_rows.assign((size_t)o.dst_h, 0);
                        for (int y = 0; y < o.dst_h; y++) {
                            int16_t *lf = &vLumS.taps[(size_t)y * vLumS.size], *cf = &vChrS.taps[(size_t)y * 2];
                            const bool csum = (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U;
                            if (vLumS.size == 1 && csum) { lf[0] = 4096; short_rows = true; }      // (the X arithmetic as it is: the blend rounds with 128 << 11 == 1 << 18)
                            else if (vLumS.size == 2 && csum && (uint16_t)lf[1] + (uint16_t)lf[0] == 4096 && (uint16_t)lf[1] <= 4096U) { rnd_rows[(size_t)y] = 1 << 18; short_rows = true; }
                        }
                    }
                    const FilterBank &vLumR = short_rows ? vLumS : c->vLum, &vChrR = short_rows ? vChrS : vChrB;
                    striprgb_short = short_rows;
                    plan_rnd = short_rows ? &rnd_rows : nullptr;
                    const bool pl = plan3(hLumB, vLumR, p.dstW, rcl, 1, gl, rL, ringL, rgb_one_one || vLumR.size == 1), pc = pl && plan3(hChrB, vChrR, p.chrDstW, rcl / 2, 2, gc, rC, ringC, rgb_one_one);
                    plan_rnd = nullptr;
                    log_msg(c, 3, "strip_rgb plan: luma %d chroma %d strips %d/%d window %d/%d nph %d/%d npv %d/%d chrDstH %d dstH %d\n", pl, pc, gl.strips, gc.strips,
                            gl.NCmax, gc.NCmax, gl.nph, gc.nph, gl.npv, gc.npv, p.chrDstH, p.dstH);
                    SOff sA;
                    const bool wantA = p.need_alpha != 0;     // (rgb_ok: then rgb_alpha holds)
                    // (not the one-tap form: yuv2rgb_1_c_template's alpha is (a * 255 + 16384) >> 15, output.c:1904, not the X arithmetic)
                    const bool pa = !wantA || (!rgb_one_one && plan3(hLumB, c->vLum, p.dstW, strip_cols_l, 1, d->stripL, sA, nullptr, false));   // the plain luma launch, the X form's own taps
                    if (pl && pc && pa && gl.strips == gc.strips &&
                        gl.NCmax / SPC <= 64 && gc.NCmax / SPC <= 64 && std::max(gl.nph, gc.nph) <= 8 && gl.npv <= 8 && gc.npv <= 8 && p.chrDstH == p.dstH) {
                        const std::vector<int16_t> htl = padded(hLumB), htc = padded(hChrB);
                        const size_t ohl = put(htl.data(), htl.size() * 2), ohc = put(htc.data(), htc.size() * 2);
                        // LDS-DMA form (kernels_striprgb.hpp sws_k_strip_rgb8): 8-bit planar sources, no source row pair skipped in either plane class;
                        // its tap rows start at the filter's own first tap (no even-position padding)
                        auto dma8_plan = [&](const FilterBank &hb, const FilterBank &vb, SwsStripGeom &g, size_t &off) {
                            g.nph8 = (hb.size + 1) / 2;
                            g.dma8_ok = !c->tune.no_strip_dma8 && p.srcKind == SRCK_PLANAR8 && !rgb_s16 && g.nph8 <= 6;
                            for (int y = 1; y < vb.count && g.dma8_ok; y++)
                                if (((vb.pos[y] & ~1) >> 1) > ((vb.pos[y - 1] & ~1) >> 1) + g.npv) g.dma8_ok = 0;
                            if (!g.dma8_ok) return;
                            const int f8 = 2 * g.nph8;
                            std::vector<int16_t> t8((size_t)hb.count * f8, 0);
                            for (int i = 0; i < hb.count; i++)
                                for (int j = 0; j < hb.size; j++) t8[(size_t)i * f8 + j] = hb.taps[(size_t)i * hb.size + j];
                            off = put(t8.data(), t8.size() * 2);
                        };
                        size_t o8l = 0, o8c = 0;
                        dma8_plan(hLumB, vLumR, gl, o8l); dma8_plan(hChrB, vChrR, gc, o8c);
                        if (!gl.dma8_ok || !gc.dma8_ok) gl.dma8_ok = gc.dma8_ok = 0;
                        { int r_ = table_alloc(c, d, &d->d_dot2, &d->dot2_bytes, blob.size()); if (r_ < 0) return r_; }
                        { int r_ = table_put(c, d, d->d_dot2, blob.data(), blob.size()); if (r_ < 0) return r_; }
                        const uint8_t *b = (const uint8_t *)d->d_dot2;
                        gl.colStart = (const int32_t *)(b + rL.cs); gl.colCount = (const int32_t *)(b + rL.cc); gl.rows = (const SwsStripRow *)(b + rL.rows);
                        gc.colStart = (const int32_t *)(b + rC.cs); gc.colCount = (const int32_t *)(b + rC.cc); gc.rows = (const SwsStripRow *)(b + rC.rows);
                        gl.hT2 = (const int16_t *)(b + ohl); gc.hT2 = (const int16_t *)(b + ohc); gl.vT2 = gc.vT2 = nullptr;
                        if (gl.dma8_ok) { gl.hT8 = (const int16_t *)(b + o8l); gc.hT8 = (const int16_t *)(b + o8c); }
                        gl.nph = gc.nph = std::max(gl.nph, gc.nph);      // one instantiation: the shorter tap rows are zero-extended in the kernel
                        if (wantA) {
                            d->stripL.colStart = (const int32_t *)(b + sA.cs); d->stripL.colCount = (const int32_t *)(b + sA.cc); d->stripL.rows = (const SwsStripRow *)(b + sA.rows);
                            d->stripL.hT2 = gl.hT2; d->stripL.vT2 = nullptr;
                            d->alpha_launch = 2;
                        }
                        d->striprgb_long = gl.npv > 5;
                        d->striprgb_ok = true;                           // (and every row in the "X" writer mode: checked below)
                        // the semi-planar source itself, without the split pass, where the kernel can read it: nv12-like through the LDS-DMA form's selectors
                        // (chroma windows of two bytes per sample: <= 64 chunks), p010-like through the 16-bit instantiation's staging (128-column strips)
                        d->striprgb_direct = 0;
                        if (!c->tune.no_striprgb_direct && !wantA) {
                            const size_t lds_nv = (size_t)16 * (2 * 2 * ((gl.NCmax + 16) >> 2) + 2 * 2 * ((2 * gc.NCmax + 16) >> 2) + 32 * 4);     // (k_striprgb.hip: ring depth 2)
                            if ((d->split_mode & 8) && gl.dma8_ok && gc.dma8_ok && rcl == 4 && 2 * gc.NCmax / 16 <= 64 && lds_nv <= 60 * 1024 && std::max(gl.nph8, gc.nph8) <= 6) {
                                d->striprgb_direct = 1; d->striprgb_direct_swap = (d->split_mode & 16) ? 1 : 0;
                            } else if ((d->split_mode & 32) && rgb_s16 && rcl == 2) {
                                d->striprgb_direct = 2; d->striprgb_direct_shift = d->split_shift;
                            }
                        }
                    }
                } else
                {
                  const bool tiles = !gray_both && !long_form && plan2(hLumB, c->vLum, p.dstW, p.dstH, 1, d->dotL, oL) && plan2(hChrB, vChrB, p.chrDstW, p.chrDstH, 2, d->dotC, oC);
                  size_t ohl = 0, ohc = 0;
                  const bool altL = strip_plan && !long_form && !p.wide && plan3_alt(hLumB, c->vLum, p.dstW, 1, d->stripL, d->stripLs, sLs);
                  const bool altC = strip_plan && !long_form && !p.wide && !gray_both && plan3_alt(hChrB, vChrB, p.chrDstW, 2, d->stripC, d->stripCs, sCs);
                  // scaled packed RGB -> packed RGB in one launch (k_striprgb2rgb.hip): the same filters planned once more on strips of 128 columns for both plane
                  // classes (a lane owns the same destination columns of Y, U, V and A).  Luma and chroma share the vertical bank there (same source and destination
                  // heights), which the kernel's lockstep march relies on: checked tap position by tap position
                  SOff s2l, s2c;
                  d->rgb2rgb_ok = false;
                  bool r2r = strip_plan && rgbread && !(p.srcW & 3) && !gray_both && !long_form && !c->tune.no_strip_rgb2rgb && (p.srcKind == SRCK_RGB24 || p.srcKind == SRCK_RGB32) && !alpha_planar &&
                             (d->fullchr_on == 1 || d->fullchr_on == 2) && (d->fullchr_kind == DSTK_RGB24 || d->fullchr_kind == DSTK_RGB32) &&
                             (d->fullchr_on != 2 || (p.srcKind == SRCK_RGB32 && d->fullchr_kind == DSTK_RGB32)) && p.chrDstW == p.dstW && p.chrDstH == p.dstH && p.chrSrcVSub == 0 &&
                             c->vLum.size == vChrB.size && c->vLum.pos == vChrB.pos;
                  // (strips a little narrower than 128 columns where that brings the widest pixel window down to one reader turn -- 256 pixels: 120 columns at 2:1)
                  int r2r_tw = 128;
                  if (r2r && c->tune.strip_cols_auto) {
                      const int hm = p.chr_half ? 2 : 1, hfl = fs2(hLumB.size), hfc = fs2(hChrB.size);
                      auto widest = [&](int tw) {
                          int npx = 0;
                          for (int x0 = 0; x0 < p.dstW; x0 += tw) {
                              int lo = INT32_MAX, hi = 0;
                              for (int x = x0; x < std::min(p.dstW, x0 + tw); x++) {
                                  lo = std::min(lo, std::min(hLumB.pos[x] & ~1, hm * (hChrB.pos[x] & ~1)));
                                  hi = std::max(hi, std::max((hLumB.pos[x] & ~1) + hfl, hm * ((hChrB.pos[x] & ~1) + hfc)));
                              }
                              npx = std::max(npx, ((hi + 7) & ~7) - (lo & ~15));
                          }
                          return npx;
                      };
                      if (widest(128) > 256) for (int tw : { 124, 120, 116, 112, 104, 96 }) if (widest(tw) <= 256) { r2r_tw = tw; break; }
                  }
                  // (the full-chroma writers' short forms, per output row: SwsStripRow::rnd_off = 1 << 9 where the reference leaves the rounding out -- kernels_stream.hpp
                  //  fullchr_row_rnd has the rule and the citations; the kernel subtracts it from its 1 << 9)
                  std::vector<int32_t> r2r_rnd;
                  if (r2r && !c->tune.no_short_forms && (c->vLum.size == 1 || c->vLum.size == 2) && vChrB.size == 2 && p.chrDstH == p.dstH) {
                      r2r_rnd.assign((size_t)o.dst_h, 0);
                      for (int y = 0; y < o.dst_h; y++) {
                          const int16_t *lf = &c->vLum.taps[(size_t)y * c->vLum.size], *cf = &vChrB.taps[(size_t)y * 2];
                          const bool csum = (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U;
                          if (csum && (c->vLum.size == 1 || ((uint16_t)lf[1] + (uint16_t)lf[0] == 4096 && (uint16_t)lf[1] <= 4096U))) r2r_rnd[(size_t)y] = 1 << 9;
                      }
                      plan_rnd = &r2r_rnd;
                  }
                  rgb2rgb_short = plan_rnd != nullptr;
                  r2r = r2r && plan3(hLumB, c->vLum, p.dstW, 2, 1, d->stripL2, s2l, nullptr, lum_plane1, 0, r2r_tw);
                  plan_rnd = nullptr;
                  r2r = r2r && plan3(hChrB, vChrB, p.chrDstW, 2, 2, d->stripC2, s2c, nullptr, chr_plane1, 0, r2r_tw) &&
                        d->stripL2.strips == d->stripC2.strips && d->stripL2.npv == d->stripC2.npv && std::max(d->stripL2.nph, d->stripC2.nph) <= 8 && d->stripL2.npv <= 8;
                  // the lockstep strip kernel of packed sources into half-width-chroma YUV (k_striprgbsrc.hip) likewise plans for itself: luma strips of up to 256
                  // columns over chroma strips of half as many, a few columns narrower where that brings the widest pixel window down by a reader turn of 256
                  // pixels (248 columns at 2:1: 504 pixels, two turns instead of three).  (YUV destinations: never together with the RGB -> RGB plans above)
                  const bool packed422_src = (d->split_mode & 3) && !(d->split_mode & 40) && p.srcKind == SRCK_PLANAR8 && p.chrSrcW == (p.srcW >> 1) && p.chrSrcVSub == 0 && !vlines_pending;
                  SOff s3l, s3c;
                  bool rsrc = strip_plan && !r2r && !p.wide && ((rgbread && !(p.srcW & 3) && (p.srcKind == SRCK_RGB24 || p.srcKind == SRCK_RGB32 || p.srcKind == SRCK_GBRP || p.srcKind == SRCK_RGB30) && p.chr_half) || packed422_src) && !gray_both && !long_form &&
                              !c->tune.no_strip_rgbsrc && !alpha_planar && !p.need_alpha && !d->fullchr_on &&
                              p.chrDstW == ((p.dstW + 1) >> 1) && (p.chrDstVSub == 0 ? p.chrDstH == p.dstH : (p.chrDstVSub == 1 && p.chrDstH == ((p.dstH + 1) >> 1)));
                  if (rsrc) {
                      const int hfl = fs2(hLumB.size), hfc = fs2(hChrB.size);
                      auto widest = [&](int tw) {
                          int npx = 0;
                          for (int x0 = 0, s = 0; x0 < p.dstW; x0 += tw, s++) {
                              int lo = INT32_MAX, hi = 0;
                              for (int x = x0; x < std::min(p.dstW, x0 + tw); x++) { lo = std::min(lo, hLumB.pos[x] & ~1); hi = std::max(hi, (hLumB.pos[x] & ~1) + hfl); }
                              for (int x = s * (tw / 2); x < std::min(p.chrDstW, (s + 1) * (tw / 2)); x++) { lo = std::min(lo, 2 * (hChrB.pos[x] & ~1)); hi = std::max(hi, 2 * ((hChrB.pos[x] & ~1) + hfc)); }
                              npx = std::max(npx, ((hi + 15) & ~15) - (lo & ~15));
                          }
                          return npx;
                      };
                      int tw3 = 256;
                      if (c->tune.strip_cols_auto) {
                          const int turns = (widest(256) + 255) / 256;
                          if (turns > 1) for (int tw : { 248, 240, 232 }) if ((widest(tw) + 255) / 256 < turns) { tw3 = tw; break; }
                      }
                      rsrc = plan3(hLumB, c->vLum, p.dstW, 4, 1, d->stripL2, s3l, nullptr, lum_plane1, 0, tw3) && plan3(hChrB, vChrB, p.chrDstW, 2, 2, d->stripC2, s3c, nullptr, chr_plane1, 0, tw3 / 2) &&
                             d->stripL2.strips == d->stripC2.strips && std::max(d->stripL2.nph, d->stripC2.nph) <= 8 && d->stripL2.npv <= 8 && d->stripC2.npv <= 12;
                  }
                  if (!tiles && strip_plan) {   // (the strip kernel shares the tile kernel's padded horizontal taps; without a tile plan it gets its own copy)
                      const std::vector<int16_t> htl = padded(hLumB, long_form ? d->stripL.hfs2 : 0), htc = gray_both ? std::vector<int16_t>(2, 0) : padded(hChrB, long_form ? d->stripC.hfs2 : 0);
                      ohl = put(htl.data(), htl.size() * 2); ohc = put(htc.data(), htc.size() * 2);
                  }
                  if (tiles || strip_plan) {
                    { int r_ = table_alloc(c, d, &d->d_dot2, &d->dot2_bytes, blob.size()); if (r_ < 0) return r_; }
                    { int r_ = table_put(c, d, d->d_dot2, blob.data(), blob.size()); if (r_ < 0) return r_; }
                    auto bind = [&](SwsTileGeom &g, const Off &o) {
                        const uint8_t *b = (const uint8_t *)d->d_dot2;
                        g.rowStart = (const int32_t *)(b + o.rs); g.rowCount = (const int32_t *)(b + o.rc);
                        g.colStart = (const int32_t *)(b + o.cs); g.colCount = (const int32_t *)(b + o.cc);
                        g.hT2 = (const int16_t *)(b + o.ht); g.vT2 = (const int16_t *)(b + o.vt);
                    };
                    if (tiles) { bind(d->dotL, oL); bind(d->dotC, oC); }
                    d->dot2_ok = tiles && !p.wide && src_ok && fs2(vChrB.size) <= 16 && c->vLum.size >= 2 && vChrB.size >= 2 && !d->fullchr_on && !alpha_planar && !d->unity_h;
                    if (strip_plan) {
                        const uint8_t *b = (const uint8_t *)d->d_dot2;
                        d->stripL.colStart = (const int32_t *)(b + sL.cs); d->stripL.colCount = (const int32_t *)(b + sL.cc);
                        if (!gray_both) { d->stripC.colStart = (const int32_t *)(b + sC.cs); d->stripC.colCount = (const int32_t *)(b + sC.cc); }
                        if (tiles) { d->stripL.hT2 = d->dotL.hT2; d->stripL.vT2 = d->dotL.vT2; d->stripC.hT2 = d->dotC.hT2; d->stripC.vT2 = d->dotC.vT2; }
                        else { d->stripL.hT2 = (const int16_t *)(b + ohl); d->stripC.hT2 = (const int16_t *)(b + ohc); d->stripL.vT2 = d->stripC.vT2 = nullptr; }
                        d->stripL.rows = (const SwsStripRow *)(b + sL.rows); if (!gray_both) d->stripC.rows = (const SwsStripRow *)(b + sC.rows);
                        if (altL) { d->stripLs.colStart = (const int32_t *)(b + sLs.cs); d->stripLs.colCount = (const int32_t *)(b + sLs.cc);
                                    d->stripLs.rows = d->stripL.rows; d->stripLs.hT2 = d->stripL.hT2; d->stripLs.vT2 = d->stripL.vT2; d->stripLs.hT8 = (const int16_t *)(b + sLs.rows); d->stripLs_ok = true; }
                        if (altC) { d->stripCs.colStart = (const int32_t *)(b + sCs.cs); d->stripCs.colCount = (const int32_t *)(b + sCs.cc);
                                    d->stripCs.rows = d->stripC.rows; d->stripCs.hT2 = d->stripC.hT2; d->stripCs.vT2 = d->stripC.vT2; d->stripCs.hT8 = (const int16_t *)(b + sCs.rows); d->stripCs_ok = true; }
                        if (r2r) {
                            const int32_t *csl = (const int32_t *)(blob.data() + s2l.cs), *ccl = (const int32_t *)(blob.data() + s2l.cc);
                            const int32_t *csc = (const int32_t *)(blob.data() + s2c.cs), *ccc = (const int32_t *)(blob.data() + s2c.cc);
                            const int hm = p.chr_half ? 2 : 1;
                            int npx = 0;
                            for (int s = 0; s < d->stripL2.strips; s++) {
                                const int w0 = std::min(csl[s], hm * csc[s]) & ~15, e = std::max(csl[s] + ccl[s], hm * (csc[s] + ccc[s]));
                                npx = std::max(npx, (e - w0 + 15) & ~15);
                            }
                            d->stripL2.colStart = (const int32_t *)(b + s2l.cs); d->stripL2.colCount = (const int32_t *)(b + s2l.cc); d->stripL2.rows = (const SwsStripRow *)(b + s2l.rows);
                            d->stripC2.colStart = (const int32_t *)(b + s2c.cs); d->stripC2.colCount = (const int32_t *)(b + s2c.cc); d->stripC2.rows = (const SwsStripRow *)(b + s2c.rows);
                            d->stripL2.hT2 = d->stripL.hT2; d->stripC2.hT2 = d->stripC.hT2; d->stripL2.vT2 = d->stripC2.vT2 = nullptr;
                            d->rgb2rgb_ok = npx <= 512; d->rgb2rgb_npx = npx;     // (two turns of 64 groups of four pixels)
                        }
                        d->strip_ok = true;
                        d->rgbread_on = rgbread;
                        d->alpha_launch = alpha_planar ? 1 : 0;
                        // scaled packed RGB into half-width-chroma YUV: one launch that reads the RGB rows itself (k_striprgbsrc.hip) on the same plan tables --
                        // luma strips of 256 columns over chroma strips of 128, every strip's pixel window (luma window and twice the chroma window, from a
                        // multiple of 16 pixels on) at most 64 lanes x 16 pixels
                        // (packed 8-bit 4:2:2 sources -- yuyv422 / uyvy422 / yvyu422 -- have the shape of the half readers: chroma samples under pixel pairs on every source
                        //  row; the kernel's byte-selector reader takes them from the caller's frame, without the split pass: striprgb_direct = 3)
                        if (rsrc) {
                            const int32_t *csl = (const int32_t *)(blob.data() + s3l.cs), *ccl = (const int32_t *)(blob.data() + s3l.cc);
                            const int32_t *csc = (const int32_t *)(blob.data() + s3c.cs), *ccc = (const int32_t *)(blob.data() + s3c.cc);
                            int npx = 0;
                            for (int s = 0; s < d->stripL2.strips; s++) {
                                const int w0 = std::min(csl[s], 2 * csc[s]) & ~15, e = std::max(csl[s] + ccl[s], 2 * (csc[s] + ccc[s]));
                                npx = std::max(npx, (e - w0 + 15) & ~15);
                            }
                            d->stripL2.colStart = (const int32_t *)(b + s3l.cs); d->stripL2.colCount = (const int32_t *)(b + s3l.cc); d->stripL2.rows = (const SwsStripRow *)(b + s3l.rows);
                            d->stripC2.colStart = (const int32_t *)(b + s3c.cs); d->stripC2.colCount = (const int32_t *)(b + s3c.cc); d->stripC2.rows = (const SwsStripRow *)(b + s3c.rows);
                            d->stripL2.hT2 = d->stripL.hT2; d->stripC2.hT2 = d->stripC.hT2; d->stripL2.vT2 = d->stripC2.vT2 = nullptr;
                            d->striprgbsrc_ok = npx <= 1024; d->striprgbsrc_npx = npx;
                            if (packed422_src) d->striprgb_direct = d->striprgbsrc_ok ? 3 : 0;
                        }
                    }
                  }
                }
            }
        }
        // ---- fused h+v tile kernel geometry (planar / semi-planar YUV outputs, non-identity horizontal filters) ----
        d->tile_ok = false;
        if (!d->unity_h && !p.fast_bilinear && !gray_any && (p.dstKind == DSTK_PLANAR8 || p.dstKind == DSTK_PLANARN || p.dstKind == DSTK_PLANAR16 ||
                            p.dstKind == DSTK_NV12 || p.dstKind == DSTK_P010) && !c->tune.no_tile) {
            const size_t hsz = p.wide ? 4 : 2;
            auto plan = [&](const FilterBank &hb, const FilterBank &vb, int W, int H, int sW, int sH, int ncomp,
                            SwsTileGeom &g, std::vector<int32_t> &arr) -> bool {
                const int TW = 128;
                for (int TH : { 32, 16, 8, 4, 2, 1 }) {
                    const int tX = (W + TW - 1) / TW, tY = (H + TH - 1) / TH;
                    std::vector<int32_t> rs(tY), rc(tY), cs(tX), cc(tX);
                    int nrmax = 0, ncmax = 0;
                    for (int t = 0; t < tY; t++) {
                        int lo = INT32_MAX, hi = -1;
                        for (int y = t * TH; y < std::min(H, (t + 1) * TH); y++) {
                            lo = std::min(lo, vb.pos[y]); hi = std::max(hi, std::min(vb.pos[y] + vb.size - 1, sH - 1));
                        }
                        rs[t] = lo; rc[t] = hi - lo + 1; nrmax = std::max(nrmax, rc[t]);
                    }
                    for (int t = 0; t < tX; t++) {
                        int lo = INT32_MAX, hi = -1;
                        for (int x = t * TW; x < std::min(W, (t + 1) * TW); x++) {
                            lo = std::min(lo, hb.pos[x]); hi = std::max(hi, hb.pos[x] + hb.size - 1);
                        }
                        cs[t] = lo; cc[t] = hi - lo + 1; ncmax = std::max(ncmax, cc[t]);
                    }
                    const size_t lds = (((size_t)nrmax * ncmax * 2 + 15) & ~(size_t)15) + (size_t)ncomp * nrmax * TW * hsz;
                    if (lds > 64 * 1024) continue;
                    g.TW = TW; g.TH = TH; g.tilesX = tX; g.tilesY = tY; g.NRmax = nrmax; g.NCmax = ncmax; g.lds_bytes = (int32_t)lds;
                    arr.clear();
                    arr.insert(arr.end(), rs.begin(), rs.end()); arr.insert(arr.end(), rc.begin(), rc.end());
                    arr.insert(arr.end(), cs.begin(), cs.end()); arr.insert(arr.end(), cc.begin(), cc.end());
                    return true;
                }
                return false;
            };
            std::vector<int32_t> aL, aC;
            if (plan(hLumB, c->vLum, p.dstW, p.dstH, p.srcW, p.srcH, 1, d->tileL, aL) &&
                plan(hChrB, vChrB, p.chrDstW, p.chrDstH, p.chrSrcW, p.chrSrcH, 2, d->tileC, aC)) {
                const size_t bytes = (aL.size() + aC.size()) * sizeof(int32_t);
                { int r_ = table_alloc(c, d, &d->d_tilegeom, &d->tilegeom_bytes, bytes); if (r_ < 0) return r_; }
                std::vector<int32_t> all(aL); all.insert(all.end(), aC.begin(), aC.end());
                { int r_ = table_put(c, d, d->d_tilegeom, all.data(), bytes); if (r_ < 0) return r_; }
                const int32_t *bL = (const int32_t *)d->d_tilegeom, *bC = bL + aL.size();
                auto bind = [](SwsTileGeom &g, const int32_t *b) {
                    g.rowStart = b; g.rowCount = b + g.tilesY; g.colStart = b + 2 * g.tilesY; g.colCount = b + 2 * g.tilesY + g.tilesX;
                };
                bind(d->tileL, bL); bind(d->tileC, bC);
                d->tile_ok = true;
            }
        }
        {   // packed_vscale picks yuv2packed1 / yuv2packed2 per row from (lfs, cfs, taps): vscale.c:135-157
            const int lfs = c->vLum.size, cfs = vChrB.size;
            bool all_x = !(lfs == 1 && cfs == 1);
            for (int y = 0; y < o.dst_h && all_x; y++) {
                const int cy = y >> c->chrDstVSubSample;
                const int16_t *lf = &c->vLum.taps[(size_t)y * lfs], *cf = &vChrB.taps[(size_t)cy * cfs];
                if (lfs == 1 && cfs == 2 && (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U) all_x = false;
                if (lfs == 2 && cfs == 2 && (uint16_t)lf[1] + (uint16_t)lf[0] == 4096 && (uint16_t)lf[1] <= 4096U &&
                    (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U) all_x = false;
            }
            d->all_x_mode = all_x;
            d->striprgb_ok = d->striprgb_ok && (all_x || (lfs == 1 && cfs == 1) || striprgb_short);     // (striprgb_short: the plan's taps and rounding offsets carry the short forms)
            if (d->alpha_launch == 2 && !d->striprgb_ok) d->alpha_launch = 0;
            const bool kind_x = d->fullchr_kind == DSTK_GBRP || d->fullchr_kind == DSTK_PACKEDHI || d->fullchr_kind == DSTK_GBRP16 || d->fullchr_kind == DSTK_GBRPF32;   // (writers with the X form only)
            // (round 5: sws_k_fullchr_rgb tells the rows of the short forms by their taps and leaves the rounding constant out there -- yuv2rgb_full_2_c_template, the chroma
            //  blend of yuv2rgb_full_1_c_template; the one-launch RGB -> RGB kernel reads the same decision from its row entries)
            const bool short_full = !all_x && !c->tune.no_short_forms && (d->fullchr_on == 1 || d->fullchr_on == 2 || d->fullchr_on == 3) && (d->fullchr_kind == DSTK_RGB24 || d->fullchr_kind == DSTK_RGB32) &&
                                    (lfs == 1 || lfs == 2) && cfs == 2;      // (3: sws_k_lut_rgb likewise, for yuv2rgb_2 rows)
            if (!all_x && !(lfs == 1 && cfs == 1) && !(short_full && rgb2rgb_short)) d->rgb2rgb_ok = false;
            if (d->fullchr_on && ((!all_x && !(lfs == 1 && cfs == 1) && !kind_x && !short_full) ||
                                  (d->fullchr_on == 4 && lfs == 1 && cfs == 1 && d->fullchr_kind != DSTK_PACKEDHI && d->fullchr_kind != DSTK_GBRP16 && d->fullchr_kind != DSTK_GBRPF32) || !d->strip_ok)) {   // (planar RGB: any_vscale, always the X form)   // no strip plan, or a row in one of the short writer forms: the generic full-chroma writer keeps it
                d->fullchr_on = 0; d->strip_ok = false; d->rgbread_on = false; d->striprgbsrc_ok = false; d->rgb2rgb_ok = false;
                p.dstKind = d->fullchr_kind; p.u_plane_dst = dd->comp[1].plane; p.v_plane_dst = dd->comp[2].plane;
            }
            // a 4:4:4 planar source at the same size into a full-chroma destination: four identity filters, so the epilogue reads the source planes itself
            // (sws_k_fullchr_rgb / sws_k_fullchr_gbrp with SRCM 1 / 2) -- no strip launch, no working picture
            d->fullchr_direct = 0;
            if ((d->fullchr_on == 1 || d->fullchr_on == 2) && d->strip_ok && d->unity_h && d->unity_v && !d->rgbread_on && !d->split_mode && isPlanarYUV(o.src_format) &&
                ((p.srcKind == SRCK_PLANAR8 && c->srcBpc == 8) || (p.srcKind == SRCK_PLANAR16 && p.src_depth <= 15 && p.src_shift == 0)) &&
                p.chrSrcW == p.srcW && p.chrSrcH == p.srcH && c->vLum.size == 1 && vChrB.size == 1 && !c->tune.no_mixed)
                d->fullchr_direct = p.srcKind == SRCK_PLANAR8 ? 1 : 2;
            int win = 0;
            for (int y = 0; y < o.dst_h; y += 2) {
                const int c0 = y >> c->chrDstVSubSample, c1 = std::min(y + 1, o.dst_h - 1) >> c->chrDstVSubSample;
                const int lo = std::min(vChrB.pos[c0], vChrB.pos[c1]), hi = std::max(vChrB.pos[c0], vChrB.pos[c1]) + cfs - 1;
                win = std::max(win, hi - lo + 1);
            }
            d->chr_window2 = win;
            // plan of the marching packed-RGB kernel: identity vertical luma filter, chroma window of a row pair <= 8 rows,
            // ring advance of at most 1 row per step
            d->rgb_march_ok = false;
            bool lum_unity = lfs == 1;
            for (int y = 0; y < o.dst_h && lum_unity; y++) lum_unity = c->vLum.taps[y] == 4096;
            // (round 5: one luma tap with two chroma taps -- SWS_BILINEAR at the same size from 4:2:0: yuv2rgb_1_c_template with its chroma blend, which rounds with 128 << 11 and IS the
            //  X arithmetic on the bank's taps, output.c:1913-1937; its one-row form (u0 + 64) >> 7 is the X arithmetic over {4096, 0}.  Not with an alpha plane: other formulas there)
            const bool lut_rows_x = all_x || (lfs == 1 && cfs == 2 && !c->needAlpha && !c->tune.no_short_forms);
            if (lut_rows_x && lum_unity && win <= 8 && cfs <= 8) {
                const int groups = (o.dst_h + 1) / 2;
                std::vector<SwsRgbGroupPlan> plan((size_t)groups);
                bool ok = true;
                int prev = INT32_MIN, maxspan = 0;
                for (int g = 0; g < groups && ok; g++) {
                    SwsRgbGroupPlan &e = plan[(size_t)g];
                    std::memset(&e, 0, sizeof(e));
                    int first[2], yy[2];
                    for (int r = 0; r < 2; r++) {
                        yy[r] = std::min(2 * g + r, o.dst_h - 1);
                        first[r] = std::max(1 - cfs, vChrB.pos[yy[r] >> c->chrDstVSubSample]);
                    }
                    e.cbase = std::min(first[0], first[1]);
                    if (prev != INT32_MIN && (e.cbase < prev || e.cbase - prev > 1)) ok = false;
                    prev = e.cbase;
                    e.ylum0 = std::min(std::max(c->vLum.pos[yy[0]], 0), o.src_h - 1);
                    e.ylum1 = std::min(std::max(c->vLum.pos[yy[1]], 0), o.src_h - 1);
                    for (int r = 0; r < 2; r++) {
                        const int16_t *cf = &vChrB.taps[(size_t)(yy[r] >> c->chrDstVSubSample) * cfs];
                        if (first[r] + cfs - 1 - e.cbase >= 8) ok = false;
                        maxspan = std::max(maxspan, first[r] + cfs - 1 - e.cbase);
                        for (int ip = 0; ip < 4; ip++) {
                            const int j0 = e.cbase + 2 * ip - first[r], j1 = j0 + 1;
                            const uint32_t lo = (j0 >= 0 && j0 < cfs) ? (uint16_t)cf[j0] : 0u, hi = (j1 >= 0 && j1 < cfs) ? (uint16_t)cf[j1] : 0u;
                            e.wp[r][ip] = lo | (hi << 16);
                        }
                    }
                }
                // ring rows of the kernel instantiation: 5 (2x chroma up-sampling with 4 taps, the common case), 6 or 8.  With 5 rows the
                // sixth slot of the three v_dot2 pairs is free: it carries the rounding constant (sample 1 x tap 2048)
                d->rgb_ncr = maxspan <= 4 ? 5 : maxspan <= 5 ? 6 : 8;
                if (ok && d->rgb_ncr == 5)
                    for (auto &e : plan)
                        for (int r = 0; r < 2; r++) e.wp[r][2] = (e.wp[r][2] & 0xFFFFu) | (2048u << 16);
                if (ok) {
                    const size_t bytes = plan.size() * sizeof(SwsRgbGroupPlan);
                    { int r_ = table_alloc(c, d, &d->d_rgbplan, &d->rgbplan_bytes, bytes); if (r_ < 0) return r_; }
                    { int r_ = table_put(c, d, d->d_rgbplan, plan.data(), bytes); if (r_ < 0) return r_; }
                    d->rgb_groups = groups;
                    d->rgb_march_ok = true;
                }
            }
        }
    }

    // ---- name the path (for SWS_PRINT_INFO, tests and rocprof matching) ----
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
    d->params_hash = fnv1a64(&d->params, sizeof(d->params));
    d->epoch = c->tables_epoch;
    return 0;
}

// sws_hip_debug_check(): every table block of the context's device states is read back and compared (by hash) with what was uploaded, and the
// host-side kernel parameters with what dev_prepare_on() left.  Returns the number of anomalies (0 = intact), < 0 on a HIP error; a short text
// per anomaly goes to `buf`.  A debugging aid for the test harness: a parity failure records whether the context's tables were still the
// host's (a wild writer's victim) or not.
int dev_check_state(SwsInternal *c, DeviceState *d, std::string &out)
{
    if (!d || d->dry) return 0;
    int bad = 0;
    HIPCHK(hipSetDevice(d->device));
    if (d->stream) HIPCHK(hipStreamSynchronize(d->stream));
    std::vector<uint8_t> back;
    for (const TableRecord &r : d->tab_recs) {
        back.resize(r.bytes);
        HIPCHK(hipMemcpy(back.data(), r.dst, r.bytes, hipMemcpyDeviceToHost));
        if (fnv1a64(back.data(), r.bytes) != r.hash) {
            bad++;
            char line[160];
            std::snprintf(line, sizeof(line), "gpu %d: table block %p (%zu bytes) differs from its upload; ", d->device, r.dst, r.bytes);
            out += line;
        }
    }
    if (d->epoch && d->params_hash != fnv1a64(&d->params, sizeof(d->params))) {
        bad++;
        out += "host-side SwsDevParams changed since dev_prepare_on(); ";
    }
    char line[96];
    std::snprintf(line, sizeof(line), "gpu %d: %zu table blocks checked; ", d->device, d->tab_recs.size());
    out += line;
    return bad;
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
int dev_plan_digest(SwsInternal *c, uint64_t out[2])
{
    int r = dev_prepare(c);
    if (r < 0) return r;
    DeviceState *d = c->dev;
    std::vector<TableRecord> recs = d->tab_recs;
    std::sort(recs.begin(), recs.end(), [](const TableRecord &a, const TableRecord &b) { return (uintptr_t)a.dst < (uintptr_t)b.dst; });
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { for (int i = 0; i < 8; i++) { h ^= (v >> (8 * i)) & 0xff; h *= 1099511628211ull; } };
    for (size_t i = 0; i < recs.size(); i++) { mix(i); mix(recs[i].bytes); mix(recs[i].hash); }
    out[0] = h;
    out[1] = d->params_hash;
    return 0;
}

// ------------------------------------------------------------------------------------------
// frame layout helpers (hwcontext: frames_get_buffer analogue of libavutil/hwcontext_cuda.c:132-197)
// ------------------------------------------------------------------------------------------
static int plane_geometry(int format, int w, int h, int plane, int *row_bytes, int *rows)
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

static int rows_of_slice(int format, int plane, int sliceY, int sliceH, int *y0, int *rows)
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

// HIP device that owns a pointer, -1 for host memory
static int ptr_device(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged || a.type == hipMemoryTypeArray) return a.device;
    return -1;
}
static bool is_device_ptr(const void *p) { return ptr_device(p) >= 0; }

// SWS_HIP_DEBUG & 16: every working buffer the library allocates (scratch planes, staging copies, byte-swapped / XYZ copies, slice
// assembly) starts out filled with 0xCD instead of whatever the allocator hands back.  A kernel that reads working memory nobody
// wrote in the same call then fails its parity test every time instead of once in 310 000 runs (the audit behind the
// one unreproduced failure of round 1).
static bool poison_enabled()
{
    static const bool on = std::getenv("SWS_HIP_DEBUG") && (std::atoi(std::getenv("SWS_HIP_DEBUG")) & 16);
    return on;
}
static int poison(SwsInternal *c, void *buf, size_t bytes)
{
    if (!poison_enabled() || !buf || !bytes) return 0;
    HIPCHK(hipMemset(buf, 0xCD, bytes));
    HIPCHK(hipDeviceSynchronize());
    return 0;
}

int grow(SwsInternal *c, void **buf, size_t *cap, size_t need)
{
    if (need <= *cap) return 0;
    guard_forget(*buf);
    if (*buf) HIPCHK(hipFree(*buf));   // (hipFree waits for the device: nothing in flight can still be using the old block)
    *buf = nullptr; *cap = 0;
    HIPCHK(hipMalloc(buf, need + (guards_enabled() ? GUARD_BYTES : 0)));
    *cap = need;
    { int r_ = poison(c, *buf, need); if (r_ < 0) return r_; }
    return guard_arm(c, *buf, need);
}

static void plane_extent(const PixDesc *d, int w, int h, int k, int *rows, int *row_bytes, int *vsub);

// ---- frame tables of the batched launches (TableRing, devstate.hpp) ----
static void ring_release(TableRing &R, size_t i)
{
    if (R.inflight[i].ev) R.pool.push_back(R.inflight[i].ev);
    R.inflight.erase(R.inflight.begin() + (long)i);
}

static int ring_regrow(SwsInternal *c, DeviceState *d, hipStream_t st, int need)
{
    TableRing &R = d->ring;
    // every launch set that reads the old blocks has to be over: the closed ones have events, the open one is on `st`
    for (auto &b : R.inflight) if (b.ev) { HIPCHK(hipEventSynchronize(b.ev)); R.pool.push_back(b.ev); b.ev = nullptr; }
    R.inflight.clear();
    const bool open_set = !R.cur.empty();
    if (open_set) HIPCHK(hipStreamSynchronize(st));
    // blocks retired by an earlier regrow: the set that held pointers into them has been launched and waited for by now
    for (auto &r : R.retired) { if (r.dev) (void)hipFree(r.dev); if (r.host) (void)hipHostFree(r.host); }
    R.retired.clear();
    R.cur.clear();
    for (auto &cc : R.cache) { cc.off = -1; cc.n = 0; }
    // A launch set under construction may hold table pointers it has NOT launched with yet (the main table is taken before the helper passes'
    // tables): the old blocks outlive this regrow and are freed by the next one or with the state (advisor r5: a freed block under such a pointer
    // would hand a kernel another context's frame addresses)
    if (open_set) R.retired.push_back({ R.dev, R.host });
    else {
        if (R.dev) HIPCHK(hipFree(R.dev));
        if (R.host) HIPCHK(hipHostFree(R.host));
    }
    R.dev = nullptr; R.host = nullptr; R.cap = 0; R.head = 0;
    int cap = 512;
    while (cap < need) cap *= 2;
    HIPCHK(hipMalloc((void **)&R.dev, sizeof(SwsFramePtrs) * (size_t)cap));
    HIPCHK(hipHostMalloc((void **)&R.host, sizeof(SwsFramePtrs) * (size_t)cap, hipHostMallocDefault));
    R.cap = cap;
    return 0;
}

static const SwsFramePtrs *table_upload_(SwsInternal *c, DeviceState *d, hipStream_t st, int slot, const SwsFramePtrs *v, int n, int *err)
{
    TableRing &R = d->ring;
    auto fail = [&](int e) -> const SwsFramePtrs * { *err = e; return nullptr; };
    TableRing::Cached &cc = R.cache[slot];
    if (cc.off >= 0 && cc.n == n && !std::memcmp(R.host + cc.off, v, sizeof(SwsFramePtrs) * (size_t)n)) {   // the table of the previous call (a caller looping over the same frames)
        R.cur.push_back({ cc.off, n });
        return R.dev + cc.off;
    }
    auto overlaps = [](const TableRing::Span &s, int off, int m) { return s.off < off + m && off < s.off + s.n; };
    for (int attempt = 0; ; attempt++) {
        if (8 * (int64_t)n > R.cap) { int r = ring_regrow(c, d, st, 8 * n); if (r < 0) return fail(r); }
        if (R.head + n > R.cap) R.head = 0;
        const int off = R.head;
        bool open_hit = false;
        for (const auto &sp : R.cur) open_hit = open_hit || overlaps(sp, off, n);
        // (callers launch with a table before they ask for the next one -- launch_plan_le_batch, launch_rgbread_strip -- so a regrow below, which frees the old device block
        //  behind a stream synchronisation, never strands a pointer that has not been launched with yet; a ring holds eight tables of the largest batch seen, a call uses at most six)
        if (open_hit) {   // the launch set being built already fills the ring: twice the size (one synchronisation, once)
            if (attempt) { log_msg(c, 0, "internal error: frame-table ring\n"); return fail(SWS_AVERROR(EINVAL)); }
            int r = ring_regrow(c, d, st, std::max(2 * R.cap, 8 * n)); if (r < 0) return fail(r);
            continue;
        }
        for (size_t i = 0; i < R.inflight.size(); ) {
            bool hit = false;
            for (const auto &sp : R.inflight[i].spans) hit = hit || overlaps(sp, off, n);
            if (!hit) { i++; continue; }
            if (R.inflight[i].ev && hipEventSynchronize(R.inflight[i].ev) != hipSuccess) { (void)hipGetLastError(); return fail(AVERROR_EXTERNAL_); }
            ring_release(R, i);
        }
        for (auto &o : R.cache) if (o.off >= 0 && overlaps({ o.off, o.n }, off, n)) { o.off = -1; o.n = 0; }
        std::memcpy(R.host + off, v, sizeof(SwsFramePtrs) * (size_t)n);
        if (hipMemcpyAsync(R.dev + off, R.host + off, sizeof(SwsFramePtrs) * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess) {
            (void)hipGetLastError(); log_msg(c, 0, "HIP error uploading a frame table\n"); return fail(AVERROR_EXTERNAL_);
        }
        cc.off = off; cc.n = n;
        R.cur.push_back({ off, n });
        R.head = off + n;
        return R.dev + off;
    }
}

const SwsFramePtrs *table_upload(SwsInternal *c, DeviceState *d, hipStream_t st, int slot, const SwsFramePtrs *v, int n)
{
    int err = 0;
    const SwsFramePtrs *t = table_upload_(c, d, st, slot, v, n, &err);
    d->ring.last_err = t ? 0 : (err ? err : AVERROR_EXTERNAL_);
    return t;
}

int table_batch_end(SwsInternal *c, DeviceState *d, hipStream_t st)
{
    TableRing &R = d->ring;
    // forget the launch sets that are over (oldest first; an event that is still pending ends the sweep)
    while (!R.inflight.empty() && (!R.inflight[0].ev || hipEventQuery(R.inflight[0].ev) == hipSuccess)) ring_release(R, 0);
    (void)hipGetLastError();   // (hipErrorNotReady from the query is not an error)
    if (R.cur.empty()) return 0;
    hipEvent_t ev = nullptr;
    if (!R.pool.empty()) { ev = R.pool.back(); R.pool.pop_back(); }
    else HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    if (hipEventRecord(ev, st) != hipSuccess) { (void)hipGetLastError(); R.pool.push_back(ev); HIPCHK(hipStreamSynchronize(st)); R.cur.clear(); return 0; }
    TableRing::Batch b;
    b.spans.swap(R.cur);
    b.ev = ev;
    R.inflight.push_back(std::move(b));
    return 0;
}

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

static int image_layout(int format, int w, int h, int align, int linesize[4], size_t offset[4], size_t *total);

// rows and visible bytes per row of plane k of a picture
static void plane_extent(const PixDesc *d, int w, int h, int k, int *rows, int *row_bytes, int *vsub)
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

static int launch_plan(SwsInternal *c, DeviceState *d, const SwsFramePtrs *frames, int n, int sliceY, int sliceH)
{
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

static int image_layout(int format, int w, int h, int align, int linesize[4], size_t offset[4], size_t *total)
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
    *total = off;
    return 0;
}

// casc_flip0: a bottom-up slice sequence through scale_cascaded: only the first context sees the flipped picture (it writes the intermediate one
// upside down, i.e. upright again), the second one runs top-down on it
static int run_single(SwsInternal *c, DeviceState *d, const uint8_t *const src[4], const int srcStride[4], int sliceY, int sliceH,
                      uint8_t *const dst[4], const int dstStride[4], bool casc_flip0 = false);

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
    for (int g = 0; g < ndev; g++) {
        if (resident[g].empty() && staged[g].empty()) continue;
        st[g] = dev_state_for(c, g);
        if (!st[g]) return AVERROR_EXTERNAL_;
        ret = dev_prepare_on(c, st[g]);     // first use on a GPU: its own copy of the tables (one H2D copy of the blob per GPU)
        if (ret < 0) return ret;
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

static int run_single(SwsInternal *c, DeviceState *d, const uint8_t *const src[4], const int srcStride[4], int sliceY, int sliceH,
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

int sws_hip_plan(SwsContext *sws, uint64_t digest[2])
{
    if (!sws) return SWS_AVERROR(EINVAL);
    SwsInternal *c = internal(sws);
    if (!c->legacy_init) return SWS_AVERROR(EINVAL);     // (a dynamic context plans per frame)
    uint64_t dg[2] = { 0, 0 };
    int r = dev_plan_digest(c, dg);
    if (r < 0) return r;
    if (digest) { digest[0] = dg[0]; digest[1] = dg[1]; }
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
        { "no_strip_short", &c->tune.no_strip_short }, { "no_generic_kinds", &c->tune.no_generic_kinds }, { "no_rgbread_kinds", &c->tune.no_rgbread_kinds }, { "strip_cols_auto", &c->tune.strip_cols_auto }, { "no_fast_banks", &c->tune.no_fast_banks }, { "no_short_forms", &c->tune.no_short_forms }, { "strip_short_waves", &c->tune.strip_short_waves }, { "dry_plan", &c->tune.dry_plan }, { "exp0", &c->tune.exp[0] }, { "exp1", &c->tune.exp[1] }, { "exp2", &c->tune.exp[2] }, { "exp3", &c->tune.exp[3] }, { "exp4", &c->tune.exp[4] }, { "exp5", &c->tune.exp[5] }, { "exp6", &c->tune.exp[6] }, { "exp7", &c->tune.exp[7] },
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
