// TEST DOUBLE of the HIP runtime entry points libswscale_hip.so imports -- test infrastructure for the CPU box, never shipped, never linked by the product.
//
// Preloaded (LD_PRELOAD) in front of the real libamdhip64 by tests/test_host_hipstub.py and tools/hipstub_hunt.py so that the product's HOST side -- device states, table
// blocks and their uploads, the frame-table ring, staging copies, shard planning over several GPUs, teardown -- runs on a box without a GPU, and runs under ASan.
// It COMPUTES NOTHING: a kernel launch is validated (registered function, non-empty grid, block and LDS within gfx950's limits, its arguments: below), logged and
// dropped; destination pictures keep whatever they held.  No parity claim can come out of it and none is made: parity is the -m gpu suite's business.
//
//   "device" memory   = host heap blocks of a registry that remembers the owning fake ordinal (HIPSTUB_DEVICES of them, default 2); every copy / memset that
//                       touches a registered block is bounds-checked against it and a violation aborts the process with a message (so it shows without ASan
//                       too).  That is the part that hunts: an upload larger than the table block it goes into would be a silent wild write on a real GPU.
//                       hipFree() only marks a block dead (its bytes are overwritten with 0xDD and kept, HIPSTUB_QUARANTINE_MB of them, default 768): a later copy
//                       into it, or a kernel argument that still points into it, is a use-after-free the stub reports instead of a silent read of whatever
//                       owns the memory next.
//   kernel arguments  = checked at every launch, by the kernel's demangled parameter list and the product's own argument structs (csrc/devparams.h): every
//                       pointer -- plane pointers of every frame of the set (read from the "device" frame table), filter banks, strip / tile geometry, raw
//                       pointer parameters -- must point into a LIVE device block, the frame table must hold `count` entries.  A context that kept a pointer
//                       into a table block it had regrown or freed (the ring hazard of ADVICE r5) shows here; on a GPU it reads someone else's tables.
//   streams / events  = HIPSTUB_DEFER=0 (default): handles; everything completes at once.
//                       HIPSTUB_DEFER=1: the LAZIEST GPU the API allows.  Work queued on a stream (copies, memsets, launches, event records and waits) runs only
//                       when the host forces it -- hipStreamSynchronize, hipEventSynchronize, hipDeviceSynchronize, a free (which synchronises its device), a
//                       device -> pageable-host copy (host-synchronous), stream destruction -- in stream order, event waits honoured; hipEventQuery says "not
//                       ready" until then.  A copy from pageable host memory takes its bytes at the call (the runtime stages them); a copy from PINNED host
//                       memory reads them when it runs, as DMA does, and the stub compares their hash with the one taken at the call: the host rewriting a
//                       pinned span the GPU has not consumed yet (the frame-table ring reusing a span too early) is reported.  Kernel arguments are captured
//                       at the call and checked when the launch runs: a block freed or a frame table overwritten in between shows.
//   HIPSTUB_LOG=path  = one line per launch / copy / allocation, with the ordinal that was current
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <cxxabi.h>

#include "devparams.h"   // the product's kernel-argument structs (librempeg_amd/csrc), for the launch-time pointer checks

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

namespace {

struct Block { size_t bytes; int device; bool host; bool dead; };
struct StubStream;
struct StubEvent { int device; unsigned magic; StubStream *stream; unsigned long long recorded, done; };
struct Arg { std::string type; std::vector<uint8_t> bytes; };
struct Op {
    enum Kind { COPY, MEMSET, LAUNCH, RECORD, WAIT } kind;
    void *dst = nullptr; const void *src = nullptr; size_t n = 0; hipMemcpyKind ck = hipMemcpyDefault; int value = 0;
    std::vector<uint8_t> staged;                 // COPY from pageable host memory: the bytes, taken at the call
    bool is_staged = false;
    bool pinned_src = false; unsigned long long src_hash = 0;
    std::string name; std::vector<Arg> args;     // LAUNCH
    StubEvent *ev = nullptr; unsigned long long gen = 0;
    int device = 0;
};
struct StubStream { int device; unsigned magic; std::deque<Op> q; bool draining; };

struct State {
    std::map<uintptr_t, Block> blocks;           // base -> block
    std::map<const void *, std::string> kernels; // host stub function -> device name
    std::map<const void *, std::vector<std::string>> params;   // host stub function -> demangled parameter types
    std::set<StubStream *> streams;
    int ndev = 2;
    bool defer = false;
    FILE *log = nullptr;
    unsigned long launches = 0, copies = 0, checked_ptrs = 0, unchecked_args = 0, deferred_ops = 0, pinned_checked = 0;
    std::deque<uintptr_t> graveyard;             // dead blocks, oldest first
    size_t graveyard_bytes = 0, graveyard_cap = 768u << 20;
};
std::recursive_mutex G;     // every entry point holds it: the stub is a model, not a fast path
State &S()
{
    static State *s = [] {
        State *p = new State();   // leaked on purpose: contexts freed by atexit handlers still find it
        if (const char *e = std::getenv("HIPSTUB_DEVICES")) p->ndev = std::atoi(e) > 0 ? std::atoi(e) : 0;
        if (const char *e = std::getenv("HIPSTUB_LOG")) p->log = std::fopen(e, "a");
        if (const char *e = std::getenv("HIPSTUB_QUARANTINE_MB")) p->graveyard_cap = (size_t)std::atol(e) << 20;
        if (const char *e = std::getenv("HIPSTUB_DEFER")) p->defer = std::atoi(e) != 0;
        return p;
    }();
    return *s;
}
thread_local int t_dev = 0;
thread_local struct { dim3 grid, block; size_t shmem; hipStream_t stream; } t_cfg;

[[noreturn]] void die(const char *what, const void *p, size_t n)
{
    std::fprintf(stderr, "hipstub: %s (pointer %p, %zu bytes, device %d)\n", what, p, n, t_dev);
    std::fflush(stderr);
    std::abort();
}

unsigned long long fnv(const void *p, size_t n)
{
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) h = (h ^ ((const uint8_t *)p)[i]) * 1099511628211ull;
    return h;
}

// the registered block (live or dead) that holds p, or nullptr
const Block *find(const void *p, uintptr_t *base)
{
    State &s = S();
    auto it = s.blocks.upper_bound((uintptr_t)p);
    if (it == s.blocks.begin()) return nullptr;
    --it;
    if ((uintptr_t)p >= it->first + it->second.bytes) return nullptr;
    *base = it->first;
    return &it->second;
}

// [p, p + n) must lie inside one live registered block when p points into one; `must` = the copy kind says this side is device memory
void check_range(const void *p, size_t n, bool must, const char *what)
{
    if (!n) return;
    uintptr_t base;
    const Block *b = find(p, &base);
    if (!b) {
        if (must) die(what, p, n);
        return;
    }
    if (b->dead) die("use after free: a copy / memset touches a device block that was freed", p, n);
    if ((uintptr_t)p + n > base + b->bytes) die(what, p, n);
}

void logf(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void logf(const char *fmt, ...)
{
    State &s = S();
    if (!s.log) return;
    va_list ap;
    va_start(ap, fmt);
    std::vfprintf(s.log, fmt, ap);
    std::fflush(s.log);
    va_end(ap);
}

void copy_checked(void *dst, const void *src, size_t n, hipMemcpyKind kind, bool src_staged)
{
    const bool dd = kind == hipMemcpyHostToDevice || kind == hipMemcpyDeviceToDevice;
    const bool sd = kind == hipMemcpyDeviceToHost || kind == hipMemcpyDeviceToDevice;
    check_range(dst, n, dd, "copy writes outside a device allocation");
    if (!src_staged) check_range(src, n, sd, "copy reads outside a device allocation");
}

StubStream *stream_of(hipStream_t st)
{
    StubStream *s = reinterpret_cast<StubStream *>(st);
    if (s && (!S().streams.count(s) || s->magic != 0x57AEA3u)) die("use of a stream that is not live", st, 0);
    return s;
}
StubEvent *event_of(hipEvent_t ev)
{
    StubEvent *e = reinterpret_cast<StubEvent *>(ev);
    if (!e || e->magic != 0xE7E27u) die("use of an event that is not live", ev, 0);
    return e;
}

// ---- launch-time checks of the kernel arguments
std::vector<std::string> split_params(const char *mangled)
{
    std::vector<std::string> out;
    int status = 0;
    char *dm = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
    if (!dm) return out;
    std::string s(dm);
    std::free(dm);
    const size_t close = s.rfind(')');
    if (close == std::string::npos) return out;
    int depth = 0;
    size_t open = std::string::npos;
    for (size_t i = close; i-- > 0;) {          // the '(' that matches the last ')'
        if (s[i] == ')' || s[i] == '>') depth++;
        else if (s[i] == '<') depth--;
        else if (s[i] == '(') { if (!depth) { open = i; break; } depth--; }
    }
    if (open == std::string::npos) return out;
    std::string cur;
    depth = 0;
    for (size_t i = open + 1; i < close; i++) {
        const char ch = s[i];
        if (ch == '<' || ch == '(') depth++;
        if (ch == '>' || ch == ')') depth--;
        if (ch == ',' && !depth) { out.push_back(cur); cur.clear(); i++; continue; }
        cur += ch;
    }
    if (!cur.empty()) out.push_back(cur);
    return out;
}

size_t arg_size(const std::string &t)       // 0 = a type the stub does not know (the plan structs of kernels_*.hpp) or does not look at
{
    if (t == "SwsFrameSet") return sizeof(SwsFrameSet);
    if (t == "SwsDevParams") return sizeof(SwsDevParams);
    if (t == "SwsStripGeom") return sizeof(SwsStripGeom);
    if (t == "SwsTileGeom") return sizeof(SwsTileGeom);
    if (!t.empty() && t.back() == '*') return sizeof(void *);
    return 0;
}

struct LaunchCheck {
    const std::string &kernel;
    int argno;
    void ptr(const void *p, size_t need, const char *what)
    {
        if (!p) return;
        State &s = S();
        uintptr_t base;
        const Block *b = find(p, &base);
        s.checked_ptrs++;
        if (b && !b->dead && !b->host && (uintptr_t)p + need <= base + b->bytes) return;
        std::fprintf(stderr, "hipstub: launch of %s: argument %d, %s = %p (%zu bytes needed) %s\n", kernel.c_str(), argno, what, p, need,
                     !b ? "points into no device allocation" : b->dead ? "points into a device block that was FREED" : b->host ? "is pinned host memory" : "runs past the end of its device block");
        std::fflush(stderr);
        std::abort();
    }
    void frames(const SwsFrameSet &fs)
    {
        if (fs.count < 0 || fs.count > 1 << 20) { std::fprintf(stderr, "hipstub: launch of %s: frame count %d\n", kernel.c_str(), fs.count); std::abort(); }
        const SwsFramePtrs *tab = fs.table;
        int n = fs.count;
        if (tab) ptr(tab, (size_t)fs.count * sizeof(SwsFramePtrs), "SwsFrameSet.table");
        else { tab = &fs.one; n = 1; }
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 4; k++) {
                ptr(tab[i].src[k], 1, "a source plane of the frame table");
                ptr(tab[i].dst[k], 1, "a destination plane of the frame table");
            }
    }
    void params(const SwsDevParams &p)
    {
        ptr(p.hLumF, 2, "SwsDevParams.hLumF"); ptr(p.hChrF, 2, "SwsDevParams.hChrF"); ptr(p.vLumF, 2, "SwsDevParams.vLumF"); ptr(p.vChrF, 2, "SwsDevParams.vChrF");
        ptr(p.hLumPos, 4, "SwsDevParams.hLumPos"); ptr(p.hChrPos, 4, "SwsDevParams.hChrPos"); ptr(p.vLumPos, 4, "SwsDevParams.vLumPos"); ptr(p.vChrPos, 4, "SwsDevParams.vChrPos");
        ptr(p.vlines, 4, "SwsDevParams.vlines"); ptr(p.gamma_tab, 2, "SwsDevParams.gamma_tab");
    }
    void strip(const SwsStripGeom &g)
    {
        const size_t n = g.strips > 0 ? (size_t)g.strips * 4 : 4;
        ptr(g.colStart, n, "SwsStripGeom.colStart"); ptr(g.colCount, n, "SwsStripGeom.colCount");
        ptr(g.hT2, 2, "SwsStripGeom.hT2"); ptr(g.vT2, 2, "SwsStripGeom.vT2"); ptr(g.rows, sizeof(SwsStripRow), "SwsStripGeom.rows"); ptr(g.hT8, 2, "SwsStripGeom.hT8");
    }
    void tile(const SwsTileGeom &g)
    {
        const size_t ny = g.tilesY > 0 ? (size_t)g.tilesY * 4 : 4, nx = g.tilesX > 0 ? (size_t)g.tilesX * 4 : 4;
        ptr(g.rowStart, ny, "SwsTileGeom.rowStart"); ptr(g.rowCount, ny, "SwsTileGeom.rowCount"); ptr(g.colStart, nx, "SwsTileGeom.colStart"); ptr(g.colCount, nx, "SwsTileGeom.colCount");
        ptr(g.hT2, 2, "SwsTileGeom.hT2"); ptr(g.vT2, 2, "SwsTileGeom.vT2");
    }
};

void check_launch_args(const std::string &name, const std::vector<Arg> &args)
{
    LaunchCheck c{ name, 0 };
    for (size_t i = 0; i < args.size(); i++) {
        const std::string &t = args[i].type;
        const void *v = args[i].bytes.data();
        c.argno = (int)i;
        if (args[i].bytes.empty()) {
            if (t != "int" && t != "long" && t != "unsigned int" && t != "bool") S().unchecked_args++;
        } else if (t == "SwsFrameSet") c.frames(*(const SwsFrameSet *)v);
        else if (t == "SwsDevParams") c.params(*(const SwsDevParams *)v);
        else if (t == "SwsStripGeom") c.strip(*(const SwsStripGeom *)v);
        else if (t == "SwsTileGeom") c.tile(*(const SwsTileGeom *)v);
        else c.ptr(*(const void *const *)v, 1, t.c_str());
    }
}

// ---- execution
void drain_event(StubEvent *e, unsigned long long gen);

void run(Op &op)
{
    State &s = S();
    const int keep = t_dev;
    t_dev = op.device;
    switch (op.kind) {
    case Op::COPY: {
        const void *src = op.is_staged ? op.staged.data() : op.src;
        copy_checked(op.dst, op.src, op.n, op.ck, op.is_staged);
        if (op.pinned_src) {
            s.pinned_checked++;
            if (fnv(op.src, op.n) != op.src_hash)
                die("a copy from PINNED host memory runs after the host rewrote its source: the bytes the GPU gets are not the ones queued (a ring span reused before its event?)", op.src, op.n);
        }
        if (op.n) std::memmove(op.dst, src, op.n);
        s.copies++;
        break;
    }
    case Op::MEMSET:
        check_range(op.dst, op.n, true, "memset outside a device allocation");
        std::memset(op.dst, op.value, op.n);
        break;
    case Op::LAUNCH:
        check_launch_args(op.name, op.args);
        s.launches++;
        break;
    case Op::RECORD:
        if (op.ev->magic == 0xE7E27u && op.ev->done < op.gen) op.ev->done = op.gen;
        break;
    case Op::WAIT:
        if (op.ev->magic == 0xE7E27u) drain_event(op.ev, op.gen);
        break;
    }
    t_dev = keep;
}

void drain(StubStream *st, StubEvent *until = nullptr, unsigned long long gen = 0)
{
    if (!st || st->draining) return;       // (a wait on an event of the stream being drained: already ordered)
    st->draining = true;
    while (!st->q.empty()) {
        if (until && until->done >= gen) break;
        Op op = std::move(st->q.front());
        st->q.pop_front();
        run(op);
    }
    st->draining = false;
}

void drain_event(StubEvent *e, unsigned long long gen)
{
    if (e->done >= gen) return;
    if (e->stream && S().streams.count(e->stream)) drain(e->stream, e, gen);
    if (e->done < gen) e->done = gen;       // recorded on the null stream, or its stream is gone: complete
}

void drain_device(int device)
{
    std::vector<StubStream *> all(S().streams.begin(), S().streams.end());
    for (StubStream *st : all)
        if (S().streams.count(st) && (device < 0 || st->device == device)) drain(st);
}

// queue on a stream when deferring, run at once otherwise (and always on the null stream)
void submit(hipStream_t stream, Op &&op)
{
    StubStream *st = stream_of(stream);
    op.device = t_dev;
    if (S().defer && st) { S().deferred_ops++; st->q.push_back(std::move(op)); }
    else run(op);
}

hipError_t do_alloc(void **out, size_t n, bool host)
{
    if (!out) return hipErrorInvalidValue;
    void *p = nullptr;
    if (posix_memalign(&p, 256, n ? n : 1)) return hipErrorOutOfMemory;
    std::memset(p, 0xA7, n);   // device memory is not zeroed by the allocator
    S().blocks[(uintptr_t)p] = Block{ n ? n : 1, t_dev, host, false };
    logf("%s dev=%d bytes=%zu\n", host ? "hostalloc" : "malloc", t_dev, n);
    *out = p;
    return hipSuccess;
}

hipError_t do_free(void *p, bool host)
{
    if (!p) return hipSuccess;
    State &s = S();
    auto it = s.blocks.find((uintptr_t)p);
    if (it == s.blocks.end() || it->second.host != host || it->second.dead) die("free of a pointer that is not a live allocation of this kind", p, 0);
    drain_device(it->second.device);            // hipFree / hipHostFree wait for the device
    it = s.blocks.find((uintptr_t)p);
    it->second.dead = true;
    std::memset(p, 0xDD, it->second.bytes);
    s.graveyard.push_back((uintptr_t)p);
    s.graveyard_bytes += it->second.bytes;
    while (s.graveyard_bytes > s.graveyard_cap && s.graveyard.size() > 1) {
        auto old = s.blocks.find(s.graveyard.front());
        s.graveyard.pop_front();
        s.graveyard_bytes -= old->second.bytes;
        std::free((void *)old->first);
        s.blocks.erase(old);
    }
    logf("%s dev=%d\n", host ? "hostfree" : "free", t_dev);
    return hipSuccess;
}

hipError_t copy_async(void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t stream)
{
    uintptr_t base;
    const Block *sb = find(src, &base), *db = find(dst, &base);
    Op op;
    op.kind = Op::COPY; op.dst = dst; op.src = src; op.n = n; op.ck = kind;
    if (S().defer && stream_of(stream)) {
        if (!db) {                              // into pageable host memory: host-synchronous -- everything queued before it on the stream has run when it returns
            drain(stream_of(stream));
            op.device = t_dev;
            run(op);
            return hipSuccess;
        }
        if (!sb) { op.is_staged = true; op.staged.assign((const uint8_t *)src, (const uint8_t *)src + n); }            // from pageable host memory: staged at the call
        else if (sb->host) { op.pinned_src = true; op.src_hash = fnv(src, n); }                                          // from pinned host memory: read when it runs
    }
    submit(stream, std::move(op));
    return hipSuccess;
}

} // namespace

extern "C" {

#define LOCK std::lock_guard<std::recursive_mutex> lock_(G)

hipError_t hipGetDeviceCount(int *n) { LOCK; if (!n) return hipErrorInvalidValue; *n = S().ndev; return S().ndev ? hipSuccess : hipErrorNoDevice; }
hipError_t hipGetDevice(int *d) { if (!d) return hipErrorInvalidValue; *d = t_dev; return hipSuccess; }
hipError_t hipSetDevice(int d) { LOCK; if (d < 0 || d >= S().ndev) return hipErrorInvalidDevice; t_dev = d; return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { LOCK; drain_device(t_dev); return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorName(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipstubError"; }

hipError_t hipMalloc(void **p, size_t n) { LOCK; return do_alloc(p, n, false); }
hipError_t hipFree(void *p) { LOCK; return do_free(p, false); }
hipError_t hipHostMalloc(void **p, size_t n, unsigned int) { LOCK; return do_alloc(p, n, true); }
hipError_t hipHostFree(void *p) { LOCK; return do_free(p, true); }

hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind kind)
{
    LOCK;       // (the null stream does not wait for the non-blocking streams the product works on)
    Op op;
    op.kind = Op::COPY; op.dst = dst; op.src = src; op.n = n; op.ck = kind; op.device = t_dev;
    run(op);
    logf("copy dev=%d kind=%d bytes=%zu\n", t_dev, (int)kind, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t stream)
{
    LOCK;
    logf("copy dev=%d kind=%d bytes=%zu\n", t_dev, (int)kind, n);
    return copy_async(dst, src, n, kind, stream);
}
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t stream)
{
    LOCK;
    if (width > dpitch || width > spitch) return hipErrorInvalidPitchValue;
    logf("copy2d dev=%d kind=%d width=%zu height=%zu\n", t_dev, (int)kind, width, height);
    const unsigned long before = S().copies;
    for (size_t y = 0; y < height; y++) (void)copy_async((uint8_t *)dst + y * dpitch, (const uint8_t *)src + y * spitch, width, kind, stream);
    if (S().copies > before) S().copies = before + 1;      // counted as one copy when it ran at once
    return hipSuccess;
}
hipError_t hipMemset(void *dst, int v, size_t n)
{
    LOCK;
    Op op;
    op.kind = Op::MEMSET; op.dst = dst; op.n = n; op.value = v; op.device = t_dev;
    run(op);
    logf("memset dev=%d bytes=%zu\n", t_dev, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t stream)
{
    LOCK;
    Op op;
    op.kind = Op::MEMSET; op.dst = dst; op.n = n; op.value = v;
    logf("memset dev=%d bytes=%zu\n", t_dev, n);
    submit(stream, std::move(op));
    return hipSuccess;
}

hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p)
{
    LOCK;
    if (!a) return hipErrorInvalidValue;
    uintptr_t base;
    const Block *b = find(p, &base);
    std::memset(a, 0, sizeof(*a));
    if (!b) { a->type = hipMemoryTypeUnregistered; a->device = -1; return hipErrorInvalidValue; }   // what ROCm answers for plain host memory
    a->type = b->host ? hipMemoryTypeHost : hipMemoryTypeDevice;
    a->device = b->device;
    a->devicePointer = const_cast<void *>(p);
    a->hostPointer = b->host ? const_cast<void *>(p) : nullptr;
    return hipSuccess;
}

hipError_t hipStreamCreateWithFlags(hipStream_t *st, unsigned int)
{
    LOCK;
    if (!st) return hipErrorInvalidValue;
    StubStream *s = new StubStream{ t_dev, 0x57AEA3u, {}, false };
    S().streams.insert(s);
    *st = reinterpret_cast<hipStream_t>(s);
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t st)
{
    LOCK;
    StubStream *s = stream_of(st);
    if (!s) return hipErrorInvalidHandle;
    drain(s);                                   // (the runtime lets queued work finish)
    S().streams.erase(s);
    s->magic = 0;
    delete s;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t st) { LOCK; drain(stream_of(st)); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *ev, unsigned)
{
    LOCK;
    if (!ev) return hipErrorInvalidValue;
    *ev = reinterpret_cast<hipEvent_t>(new StubEvent{ t_dev, 0xE7E27u, nullptr, 0, 0 });
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *ev) { return hipEventCreateWithFlags(ev, 0); }
hipError_t hipEventDestroy(hipEvent_t ev)
{
    LOCK;
    StubEvent *e = event_of(ev);
    drain_event(e, e->recorded);                // (queued records / waits of it must not outlive it here: the runtime keeps it alive for them)
    for (StubStream *st : S().streams) {
        if (std::any_of(st->q.begin(), st->q.end(), [e](const Op &op) { return op.ev == e; })) drain(st);
    }
    e->magic = 0;
    delete e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t ev, hipStream_t st)
{
    LOCK;
    StubEvent *e = event_of(ev);
    StubStream *s = stream_of(st);
    e->recorded++;
    e->stream = s;
    Op op;
    op.kind = Op::RECORD; op.ev = e; op.gen = e->recorded;
    submit(st, std::move(op));
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t ev) { LOCK; StubEvent *e = event_of(ev); drain_event(e, e->recorded); return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t ev) { LOCK; StubEvent *e = event_of(ev); return e->done >= e->recorded ? hipSuccess : hipErrorNotReady; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    LOCK;
    StubEvent *x = event_of(a), *y = event_of(b);
    if (x->done < x->recorded || y->done < y->recorded) return hipErrorNotReady;
    if (ms) *ms = 1.0f;
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t st, hipEvent_t ev, unsigned int)
{
    LOCK;
    StubEvent *e = event_of(ev);
    Op op;
    op.kind = Op::WAIT; op.ev = e; op.gen = e->recorded;
    submit(st, std::move(op));
    return hipSuccess;
}

hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, const void *f, int, size_t)
{
    if (!n || !f) return hipErrorInvalidValue;
    *n = 8;
    return hipSuccess;
}

void **__hipRegisterFatBinary(const void *) { static void *handle[1]; return handle; }
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *hostFunction, char *, const char *deviceName, unsigned int, void *, void *, void *, void *, int *)
{
    LOCK;
    S().kernels[hostFunction] = deviceName ? deviceName : "?";
    if (deviceName) S().params[hostFunction] = split_params(deviceName);
}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream)
{
    t_cfg.grid = grid; t_cfg.block = block; t_cfg.shmem = shmem; t_cfg.stream = stream;
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *stream)
{
    *grid = t_cfg.grid; *block = t_cfg.block; *shmem = t_cfg.shmem; *stream = t_cfg.stream;
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void *f, dim3 grid, dim3 block, void **args, size_t shmem, hipStream_t stream)
{
    LOCK;
    auto it = S().kernels.find(f);
    if (it == S().kernels.end()) die("launch of a function that was never registered", f, 0);
    const std::string &name = it->second;
    StubStream *st = stream_of(stream);
    const unsigned long long threads = (unsigned long long)block.x * block.y * block.z;
    if (!grid.x || !grid.y || !grid.z || !threads || threads > 1024 || shmem > 160 * 1024 || !args) {   // hipErrorInvalidConfiguration on the real runtime
        std::fprintf(stderr, "hipstub: invalid launch of %s: grid %u,%u,%u block %u,%u,%u lds %zu\n", name.c_str(), grid.x, grid.y, grid.z, block.x, block.y, block.z, shmem);
        std::abort();
    }
    if (st && st->device != t_dev) die("launch on a stream of another GPU than the current one", stream, 0);
    Op op;
    op.kind = Op::LAUNCH; op.name = name;
    const std::vector<std::string> &types = S().params[f];
    for (size_t i = 0; i < types.size(); i++) {     // the kernel arguments are taken at the call, as the runtime copies them into the launch packet
        Arg a{ types[i], {} };
        if (const size_t n = arg_size(types[i])) a.bytes.assign((const uint8_t *)args[i], (const uint8_t *)args[i] + n);
        op.args.push_back(std::move(a));
    }
    logf("launch dev=%d stream=%d name=%s grid=%u,%u,%u block=%u,%u,%u lds=%zu\n", t_dev, st ? st->device : -1, name.c_str(), grid.x, grid.y, grid.z, block.x, block.y, block.z,
         shmem);
    submit(stream, std::move(op));
    return hipSuccess;
}

// for the tests
// the live device block that holds p: 1 and its bounds, or 0 (tests/hipemu checks the kernels' buffer accesses against it)
int hipstub_find_block(const void *p, unsigned long long *base, unsigned long long *bytes)
{
    LOCK;
    uintptr_t b;
    const Block *blk = find(p, &b);
    if (!blk || blk->dead) return 0;
    *base = b; *bytes = blk->bytes;
    return 1;
}
// flips one byte in the middle of the first live device block of exactly `bytes` bytes (the tests damage a table block with it); 1 if there was one
int hipstub_scribble(unsigned long long bytes)
{
    LOCK;
    for (auto &b : S().blocks)
        if (!b.second.dead && !b.second.host && b.second.bytes == bytes) { ((unsigned char *)b.first)[bytes / 2] ^= 0x40; return 1; }
    return 0;
}
unsigned long hipstub_launches(void) { LOCK; return S().launches; }
unsigned long hipstub_copies(void) { LOCK; return S().copies; }
unsigned long hipstub_live_blocks(void)
{
    LOCK;
    unsigned long n = 0;
    for (auto &b : S().blocks) n += !b.second.dead;
    return n;
}
unsigned long hipstub_checked_pointers(void) { LOCK; return S().checked_ptrs; }
unsigned long hipstub_unchecked_args(void) { LOCK; return S().unchecked_args; }
unsigned long hipstub_deferred_ops(void) { LOCK; return S().deferred_ops; }
unsigned long hipstub_pinned_checked(void) { LOCK; return S().pinned_checked; }
unsigned long hipstub_pending_ops(void) { LOCK; unsigned long n = 0; for (StubStream *st : S().streams) n += st->q.size(); return n; }

} // extern "C"
