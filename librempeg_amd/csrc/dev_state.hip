// HIP side of libswscale_hip, part 1 of 4 -- the per-(context, GPU) DEVICE STATE: creation and teardown, the debug switches (SWS_HIP_DEBUG: poisoned working
// buffers, verified uploads, guard bands), table blocks and their upload records, the frame-table ring of the batched launches, sws_hip_debug_check()'s read-back.
// (dev_plan.hip: the planner; dev_exec.hip: launch plans, staging, sharding, sws_scale(); dev_api.hip: streams, hwcontext helpers, sws_hip_*.)
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


#include "dev_internal.hpp"
#include "generic_kinds.hpp"
#include "../../include/hwcontext_hip.h"

namespace swship {

// the plan geometries of a fresh state start out as zeros (`new DeviceState()` runs the default member initialisers and leaves members without one as the
// heap had them: a field a planner path does not set -- band counts, the byte-row form's flags -- would otherwise differ from process to process)
int ensure_dev(SwsInternal *c)
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
DeviceState *dev_state_for(SwsInternal *c, int device)
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

void guard_forget(void *p);      // (SWS_HIP_DEBUG & 64: below)
bool guards_enabled();
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

bool poison_enabled();
int poison(SwsInternal *c, void *buf, size_t bytes);
// SWS_HIP_DEBUG & 64 (round 5, DESIGN.md 8 iv): every working buffer and table block of the library carries GUARD_BYTES of a known pattern behind its last byte, and every
// conversion ends by waiting for its stream and comparing the guards of ALL live blocks of the process: a kernel that writes behind a working picture or a table fails the call
// that did it -- loudly, with the block and the offset -- instead of damaging whatever the allocator placed there (another context's tables, another picture).
static const size_t GUARD_BYTES = 64 * 1024;
bool guards_enabled()
{
    static const bool on = std::getenv("SWS_HIP_DEBUG") && (std::atoi(std::getenv("SWS_HIP_DEBUG")) & 64);
    return on;
}
// (leaked on purpose: contexts freed during process teardown -- python finalisers, atexit -- still find the registry alive)
struct GuardRegistry { std::mutex mu; std::map<void *, size_t> blocks; };      // block -> bytes in front of its guard
static GuardRegistry &guard_registry() { static GuardRegistry *r = new GuardRegistry(); return *r; }
#define g_guard_mu (guard_registry().mu)
#define g_guarded (guard_registry().blocks)
void guard_forget(void *p) { if (p && guards_enabled()) { std::lock_guard<std::mutex> lk(g_guard_mu); g_guarded.erase(p); } }
int guard_arm(SwsInternal *c, void *p, size_t bytes)
{
    if (!guards_enabled() || !p) return 0;
    HIPCHK(hipMemset((uint8_t *)p + bytes, 0xA7, GUARD_BYTES));
    HIPCHK(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_guard_mu);
    g_guarded[p] = bytes;
    return 0;
}
int guards_check(SwsInternal *c, hipStream_t st)
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
uint64_t fnv1a64(const void *p, size_t n)
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
int table_alloc(SwsInternal *c, DeviceState *d, void **buf, size_t *cap, size_t need)
{
    if (d->dry) {     // a fixed fake address per table member of the state: the plan (pointers into the blocks included) is the same on every run
        table_records_drop(d, *buf, *cap);
        void **const slots[] = { &d->d_tables, &d->d_tilegeom, &d->d_rgbplan, &d->d_dot2, &d->d_vlines };      // (by role, not by offset: a new member of DeviceState must not move the fake addresses)
        uint64_t slot = 15;
        for (size_t i = 0; i < sizeof(slots) / sizeof(slots[0]); i++) if (buf == slots[i]) slot = i;
        *buf = (void *)(uintptr_t)(0x100000000000ull + slot * 0x1000000000ull);
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
int table_put(SwsInternal *c, DeviceState *d, void *dst, const void *src, size_t bytes)
{
    if (!bytes) return 0;
    if (d->dry) {
        table_records_drop(d, dst, bytes);
        d->tab_recs.push_back({ dst, bytes, fnv1a64(src, bytes), d->plan_serial });
        return 0;
    }
    if (d->defer_uploads) {     // a peer GPU of an RCCL-fed call: the block arrives by broadcast from the home GPU (dev_rccl.hip), or by this very copy if that fails
        table_records_drop(d, dst, bytes);
        const uint64_t h = fnv1a64(src, bytes);
        d->tab_recs.push_back({ dst, bytes, h, d->plan_serial });
        for (size_t i = 0; i < d->deferred.size();)      // (a block rewritten within one planning run: the last contents count)
            if ((const uint8_t *)d->deferred[i].dst < (const uint8_t *)dst + bytes && (const uint8_t *)dst < (const uint8_t *)d->deferred[i].dst + d->deferred[i].bytes) d->deferred.erase(d->deferred.begin() + (long)i);
            else i++;
        d->deferred.push_back({ dst, bytes, h, std::vector<uint8_t>((const uint8_t *)src, (const uint8_t *)src + bytes) });
        return 0;
    }
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    table_records_drop(d, dst, bytes);
    d->tab_recs.push_back({ dst, bytes, fnv1a64(src, bytes), d->plan_serial });
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

// The always-on integrity check of the advisor's round-5 review: at the FIRST launch of a plan, every table block that plan wrote is read back once and compared with the
// hash of its upload -- a block damaged between upload and use (another context's stray store, a failed copy) fails that conversion loudly and names the context, instead
// of giving wrong pictures for the context's lifetime.  One device -> host copy of the tables (tens to a few hundred KB) and one wait per plan;
// SWS_HIP_NO_TABLE_VERIFY=1 switches it off.  Exercised on the CPU box over tests/hipstub (every conversion of tools/hipstub_hunt.py passes through it).
int verify_tables_once(SwsInternal *c, DeviceState *d)
{
    static const bool off = std::getenv("SWS_HIP_NO_TABLE_VERIFY") && std::atoi(std::getenv("SWS_HIP_NO_TABLE_VERIFY")) > 0;
    if (off || !d || d->dry || !d->stream) return 0;
    std::vector<uint8_t> back;
    for (TableRecord &r : d->tab_recs) {
        if ((r.serial & ~TAB_VERIFIED) != d->plan_serial || (r.serial & TAB_VERIFIED)) continue;
        back.resize(r.bytes);
        HIPCHK(hipMemcpyAsync(back.data(), r.dst, r.bytes, hipMemcpyDeviceToHost, d->stream));
        HIPCHK(hipStreamSynchronize(d->stream));
        if (fnv1a64(back.data(), r.bytes) != r.hash) {
            log_msg(c, 0, "gpu %d: table block %p (%zu bytes) no longer holds what was uploaded to it: refusing to convert with it (the context re-plans on the next call)\n", d->device, r.dst, r.bytes);
            d->epoch = 0;
            return AVERROR_EXTERNAL_;
        }
        r.serial |= TAB_VERIFIED;
    }
    return 0;
}

// HIP device that owns a pointer, -1 for host memory
int ptr_device(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged || a.type == hipMemoryTypeArray) return a.device;
    return -1;
}
bool is_device_ptr(const void *p) { return ptr_device(p) >= 0; }

// SWS_HIP_DEBUG & 16: every working buffer the library allocates (scratch planes, staging copies, byte-swapped / XYZ copies, slice
// assembly) starts out filled with 0xCD instead of whatever the allocator hands back.  A kernel that reads working memory nobody
// wrote in the same call then fails its parity test every time instead of once in 310 000 runs (the audit behind the
// one unreproduced failure of round 1).
bool poison_enabled()
{
    static const bool on = std::getenv("SWS_HIP_DEBUG") && (std::atoi(std::getenv("SWS_HIP_DEBUG")) & 16);
    return on;
}
int poison(SwsInternal *c, void *buf, size_t bytes)
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
    // (+ 2 KiB: the readable tail the marching kernels' chunked row loads may reach behind a picture that ends the block, dev_exec.hip image_layout)
    HIPCHK(hipMalloc(buf, need + (guards_enabled() ? GUARD_BYTES : 0) + 2048));
    *cap = need;
    { int r_ = poison(c, *buf, need); if (r_ < 0) return r_; }
    return guard_arm(c, *buf, need);
}

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

} // namespace swship
