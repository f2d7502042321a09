// tests/hipemu: executes a kernel launch of the x86 build thread by thread (see include/hip/hip_runtime.h and README.md).  One block at a time; the threads of a block
// are fibers (ucontext) scheduled round-robin, which switch only at synchronisation points: __syncthreads (all live threads of the block) and the wave-level points
// the kernels mark -- wave barriers, fences, s_waitcnt (the hand-written waits around LDS-DMA and LDS exchange) -- at which all live lanes of a 64-lane wave meet, so
// that what one lane wrote to LDS before the point is there for the others after it.  A wave-synchronous exchange that the source does not mark would be missed: the
// emulation is a check of the marked algorithm, not of the hardware.
#include <hip/hip_runtime.h>

#include <sys/mman.h>
#include <ucontext.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

// AddressSanitizer must be told about every stack switch (make asan)
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#include <sanitizer/common_interface_defs.h>
#define HIPEMU_ASAN 1
#endif
#endif
#ifdef HIPEMU_ASAN
#define FIBER_START_SWITCH(save, bottom, size) __sanitizer_start_switch_fiber((save), (bottom), (size))
#define FIBER_FINISH_SWITCH(save, bottom, size) __sanitizer_finish_switch_fiber((save), (bottom), (size))
#else
#define FIBER_START_SWITCH(save, bottom, size) ((void)0)
#define FIBER_FINISH_SWITCH(save, bottom, size) ((void)0)
#endif

namespace swsk { alignas(4096) unsigned char smem[160 * 1024 + 4096]; }      // the dynamic LDS of `extern __shared__ uint8_t smem[]`
namespace swship { alignas(4096) unsigned char smem[160 * 1024 + 4096]; }

namespace hipemu {

Ctx *cur = nullptr;

namespace {
constexpr size_t STACK = 4096 * 1024;     // (-O0 frames of the generic kernels pass 800-byte argument structs down many levels)
constexpr int MAXT = 1024, MAXW = MAXT / 64;
enum { READY, WAIT_WAVE, WAIT_BLOCK, DONE };
struct Fiber { ucontext_t ctx; Ctx c; int state; unsigned gen; };
Fiber fib[MAXT];
unsigned char *stacks = nullptr;
ucontext_t sched;
int cur_i = -1, nthreads = 0;
int live_block, arrived_block; unsigned gen_block;
int live_wave[MAXW], arrived_wave[MAXW]; unsigned gen_wave[MAXW];
unsigned long long slots[MAXW][64];
void (*g_thunk)(void *); void *g_closure; const char *g_name = "?";
unsigned long g_launches = 0, g_threads = 0;
const bool g_reverse = std::getenv("HIPEMU_REVERSE") && std::atoi(std::getenv("HIPEMU_REVERSE"));

const void *sched_bottom = nullptr; size_t sched_size = 0;
// from a fiber back to the scheduler, to be resumed later
void yield(Fiber &f)
{
    void *fake = nullptr;
    FIBER_START_SWITCH(&fake, sched_bottom, sched_size);
    swapcontext(&f.ctx, &sched);
    FIBER_FINISH_SWITCH(fake, nullptr, nullptr);
}

void release_wave(int w) { arrived_wave[w] = 0; gen_wave[w]++; }
void release_block() { arrived_block = 0; gen_block++; }

void entry()
{
    Fiber &f = fib[cur_i];
    FIBER_FINISH_SWITCH(nullptr, &sched_bottom, &sched_size);
    g_thunk(g_closure);
    f.state = DONE;
    const int w = f.c.wave;
    live_block--; live_wave[w]--;
    if (arrived_wave[w] && arrived_wave[w] == live_wave[w]) release_wave(w);       // the others were waiting for this lane only
    if (arrived_block && arrived_block == live_block) release_block();
    FIBER_START_SWITCH(nullptr, sched_bottom, sched_size);      // (this fiber's stack is done with)
    swapcontext(&f.ctx, &sched);
}
} // namespace

void sync_wave()
{
    Fiber &f = fib[cur_i];
    const int w = f.c.wave;
    if (++arrived_wave[w] == live_wave[w]) { release_wave(w); return; }
    f.state = WAIT_WAVE; f.gen = gen_wave[w];
    while (gen_wave[w] == f.gen) yield(f);
    f.state = READY;
    cur = &f.c;
}

void sync_block()
{
    Fiber &f = fib[cur_i];
    if (++arrived_block == live_block) { release_block(); return; }
    f.state = WAIT_BLOCK; f.gen = gen_block;
    while (gen_block == f.gen) yield(f);
    f.state = READY;
    cur = &f.c;
}

unsigned long long wave_read(unsigned long long v, int src_lane)
{
    Fiber &f = fib[cur_i];
    slots[f.c.wave][f.c.lane] = v;
    sync_wave();
    const unsigned long long r = slots[f.c.wave][src_lane & 63];
    sync_wave();
    return r;
}

bool check_buffers = std::getenv("HIPEMU_CHECK_BUFFERS") && std::atoi(std::getenv("HIPEMU_CHECK_BUFFERS"));
namespace {
struct Oob { unsigned long count; unsigned long long max_past; };
std::map<std::string, Oob> g_oob;
typedef int (*find_block_t)(const void *, unsigned long long *, unsigned long long *);
}
bool in_block(const void *p, unsigned bytes, bool store)
{
    static find_block_t find = (find_block_t)dlsym(RTLD_DEFAULT, "hipstub_find_block");
    if (!find) return true;
    unsigned long long base = 0, size = 0;
    if (find(p, &base, &size) && (unsigned long long)(uintptr_t)p + bytes <= base + size) return true;
    if (store) {
        std::fprintf(stderr, "hipemu: %s: a buffer STORE of %u bytes at %p leaves the device block it starts in (or is in none)\n", g_name, bytes, p);
        std::abort();
    }
    // how far past the end of the nearest block below
    unsigned long long past = 0;
    for (unsigned back = 1; back <= 4096 && !past; back++)
        if (find((const unsigned char *)p - back, &base, &size)) past = (unsigned long long)(uintptr_t)p + bytes - (base + size);
    Oob &o = g_oob[g_name];
    o.count++;
    if (past > o.max_past) o.max_past = past;
    return false;
}

unsigned char *lds_ptr(unsigned lds_addr)
{
    const uintptr_t hi = (uintptr_t)swsk::smem & ~(uintptr_t)0xFFFFFFFFull;
    return (unsigned char *)(hi | lds_addr);
}

void run_grid(const char *name, dim3 grid, dim3 block, size_t shmem, void (*thunk)(void *), void *closure)
{
    const unsigned long long n = (unsigned long long)block.x * block.y * block.z;
    if (!grid.x || !grid.y || !grid.z || !n || n > MAXT || shmem > 160 * 1024) { std::fprintf(stderr, "hipemu: invalid launch of %s\n", name); std::abort(); }
    if (((uintptr_t)swsk::smem >> 32) != (((uintptr_t)swsk::smem + sizeof(swsk::smem)) >> 32)) { std::fprintf(stderr, "hipemu: the LDS arena crosses a 4 GiB line\n"); std::abort(); }
    static std::mutex one_launch;                 // (the library launches from one host thread per GPU when a batch spans several: the emulator runs them in turn)
    std::lock_guard<std::mutex> only(one_launch);
    if (cur_i >= 0) { std::fprintf(stderr, "hipemu: nested launch\n"); std::abort(); }
    if (!stacks) {
        stacks = (unsigned char *)mmap(nullptr, STACK * MAXT, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (stacks == MAP_FAILED) { std::perror("hipemu: mmap"); std::abort(); }
    }
    g_thunk = thunk; g_closure = closure; g_name = name; nthreads = (int)n;
    g_launches++;
    const int nwaves = (nthreads + 63) / 64;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                live_block = nthreads; arrived_block = 0;
                for (int w = 0; w < nwaves; w++) { live_wave[w] = (w + 1) * 64 <= nthreads ? 64 : nthreads - w * 64; arrived_wave[w] = 0; }
                for (int t = 0; t < nthreads; t++) {
                    Fiber &f = fib[t];
                    f.c.tid = { (unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y) };
                    f.c.bid = { bx, by, bz };
                    f.c.bdim = { block.x, block.y, block.z };
                    f.c.gdim = { grid.x, grid.y, grid.z };
                    f.c.lane = t & 63; f.c.wave = t >> 6;
                    f.state = READY;
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = stacks + (size_t)t * STACK;
                    f.ctx.uc_stack.ss_size = STACK;
                    f.ctx.uc_link = nullptr;
                    makecontext(&f.ctx, entry, 0);
                }
                g_threads += nthreads;
                int done = 0;
                while (done < nthreads) {
                    bool progress = false;
                    for (int tt = 0; tt < nthreads; tt++) {
                        const int t = g_reverse ? nthreads - 1 - tt : tt;      // (HIPEMU_REVERSE=1: the other schedule -- a result that depends on it is an unmarked exchange)
                        Fiber &f = fib[t];
                        if (f.state == DONE) continue;
                        if (f.state == WAIT_WAVE && gen_wave[f.c.wave] == f.gen) continue;
                        if (f.state == WAIT_BLOCK && gen_block == f.gen) continue;
                        cur_i = t; cur = &f.c;
                        {
                            void *fake = nullptr;
                            FIBER_START_SWITCH(&fake, stacks + (size_t)t * STACK, STACK);
                            swapcontext(&sched, &f.ctx);
                            FIBER_FINISH_SWITCH(fake, nullptr, nullptr);
                        }
                        progress = true;
                        if (f.state == DONE) done++;
                    }
                    if (!progress) {
                        std::fprintf(stderr, "hipemu: %s, block %u,%u,%u: every thread that is left waits (divergent barrier): %d of %d done\n", name, bx, by, bz, done, nthreads);
                        std::abort();
                    }
                }
            }
    cur_i = -1; cur = nullptr;
}

} // namespace hipemu

// one line per kernel whose buffer loads left their block: "name count max_bytes_past_the_end"
extern "C" int hipemu_oob_report(char *buf, int cap)
{
    std::string s;
    for (auto &e : hipemu::g_oob) { char line[512]; std::snprintf(line, sizeof(line), "%s %lu %llu\n", e.first.c_str(), e.second.count, e.second.max_past); s += line; }
    if (buf && cap > 0) { std::snprintf(buf, cap, "%s", s.c_str()); }
    return (int)hipemu::g_oob.size();
}
extern "C" unsigned long hipemu_launches(void) { return hipemu::g_launches; }
extern "C" unsigned long hipemu_threads(void) { return hipemu::g_threads; }
