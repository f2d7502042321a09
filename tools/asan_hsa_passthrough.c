/* LD_PRELOAD shim for the HOST sanitizer run (tools/asan_env.sh): ROCm's clang AddressSanitizer runtime interposes the HSA memory
 * entry points (hsa_amd_memory_pool_allocate ...) for DEVICE ASan and fails every device allocation of an uninstrumented stack ("out of
 * memory: allocator is trying to allocate") -- this image has no ASan build of ROCr / HIP under /opt/rocm/lib/asan, and torch brings its own
 * runtime.  The host-only run wants the stock behaviour: each function below hands the call straight to the real libhsa-runtime64 that is
 * already in the process.  Preloaded BEFORE the sanitizer runtime (ASAN_OPTIONS=verify_asan_link_order=0), so it wins symbol lookup.
 *   gcc -O2 -shared -fPIC -I/opt/rocm/include -o tools/bin/libasan_hsa_passthrough.so tools/asan_hsa_passthrough.c -ldl            */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <link.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

static void *g_hsa;
static int find_cb(struct dl_phdr_info *info, size_t size, void *data)
{
    (void)size;
    if (info->dlpi_name && strstr(info->dlpi_name, "libhsa-runtime64")) { strncpy((char *)data, info->dlpi_name, 4095); return 1; }
    return 0;
}
static void *real(const char *name)
{
    if (!g_hsa) {
        char path[4096] = "";
        dl_iterate_phdr(find_cb, path);
        g_hsa = dlopen(path[0] ? path : "libhsa-runtime64.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!g_hsa) g_hsa = dlopen(path[0] ? path : "libhsa-runtime64.so.1", RTLD_NOW);
        if (!g_hsa) { fprintf(stderr, "asan_hsa_passthrough: no libhsa-runtime64 in the process (%s)\n", dlerror()); abort(); }
    }
    void *f = dlsym(g_hsa, name);
    if (!f) { fprintf(stderr, "asan_hsa_passthrough: %s not found\n", name); abort(); }
    return f;
}
#define FWD(name, params, args) \
    hsa_status_t name params { static __typeof__(&name) f; if (!f) f = (__typeof__(&name))real(#name); return f args; }

FWD(hsa_amd_memory_pool_allocate, (hsa_amd_memory_pool_t pool, size_t size, uint32_t flags, void **ptr), (pool, size, flags, ptr))
FWD(hsa_amd_memory_pool_free, (void *ptr), (ptr))
FWD(hsa_amd_agents_allow_access, (uint32_t n, const hsa_agent_t *agents, const uint32_t *flags, const void *ptr), (n, agents, flags, ptr))
FWD(hsa_memory_copy, (void *dst, const void *src, size_t size), (dst, src, size))
FWD(hsa_amd_memory_async_copy, (void *dst, hsa_agent_t da, const void *src, hsa_agent_t sa, size_t size, uint32_t nd, const hsa_signal_t *deps, hsa_signal_t done),
    (dst, da, src, sa, size, nd, deps, done))
FWD(hsa_amd_memory_async_copy_on_engine, (void *dst, hsa_agent_t da, const void *src, hsa_agent_t sa, size_t size, uint32_t nd, const hsa_signal_t *deps,
                                          hsa_signal_t done, hsa_amd_sdma_engine_id_t eng, bool force), (dst, da, src, sa, size, nd, deps, done, eng, force))
FWD(hsa_amd_ipc_memory_create, (void *ptr, size_t len, hsa_amd_ipc_memory_t *h), (ptr, len, h))
FWD(hsa_amd_ipc_memory_attach, (const hsa_amd_ipc_memory_t *h, size_t len, uint32_t n, const hsa_agent_t *agents, void **mapped), (h, len, n, agents, mapped))
FWD(hsa_amd_ipc_memory_detach, (void *mapped), (mapped))
FWD(hsa_amd_vmem_address_reserve_align, (void **va, size_t size, uint64_t address, uint64_t alignment, uint64_t flags), (va, size, address, alignment, flags))
FWD(hsa_amd_vmem_address_free, (void *va, size_t size), (va, size))
