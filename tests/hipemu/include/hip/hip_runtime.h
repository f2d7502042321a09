// tests/hipemu: the HIP kernel LANGUAGE for an x86 build of the product's own sources -- test infrastructure for the box without a GPU, never shipped, never loaded by
// librempeg_amd.  See tests/hipemu/README.md: what it is (the kernels' C++ executed thread by thread on the host, against the HIP runtime test double of
// tests/hipstub), what it is good for (the host restructuring of a round still gives the oracle's pictures; AddressSanitizer over the kernels' addressing) and
// what it is NOT (it is not the gfx950 code object, says nothing about the compiler, the ISA, timing or the memory model: GPU parity is the -m gpu suite's business).
#pragma once
#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>

namespace hipemu {
struct u3 { unsigned x, y, z; };
struct Ctx { u3 tid, bid, bdim, gdim; int lane, wave; };
extern Ctx *cur;
void sync_block();
void sync_wave();
unsigned long long wave_read(unsigned long long v, int src_lane);     // every live lane of the wave calls it; returns lane src_lane's v
void run_grid(const char *name, dim3 grid, dim3 block, size_t shmem, void (*thunk)(void *), void *closure);
unsigned char *lds_ptr(unsigned lds_addr);                            // the LDS arena address whose low 32 bits are lds_addr

template <class K, class... A> inline void launch(const char *name, K k, dim3 grid, dim3 block, size_t shmem, hipStream_t, A... a)
{
    auto body = [&]() { k(a...); };
    run_grid(name, grid, block, shmem, [](void *c) { (*static_cast<decltype(body) *>(c))(); }, &body);
}
} // namespace hipemu

#define threadIdx (hipemu::cur->tid)
#define blockIdx (hipemu::cur->bid)
#define blockDim (hipemu::cur->bdim)
#define gridDim (hipemu::cur->gdim)
#define warpSize 64
#undef __global__
#undef __device__
#undef __host__
#undef __constant__
#undef __forceinline__
#undef __noinline__
#define __global__ inline              /* (kernels and device functions live in headers that several translation units include) */
#define __device__ inline
#define __host__
#define __constant__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __syncthreads() hipemu::sync_block()
#define __threadfence_block() hipemu::sync_wave()
#define hipLaunchKernelGGL(k, g, b, s, st, ...) hipemu::launch(#k, (k), dim3(g), dim3(b), (size_t)(s), (st), ##__VA_ARGS__)

// ---- arithmetic helpers of the HIP device library
template <class A, class B> inline typename std::common_type<A, B>::type min(A a, B b) { typedef typename std::common_type<A, B>::type T; return (T)a < (T)b ? (T)a : (T)b; }
template <class A, class B> inline typename std::common_type<A, B>::type max(A a, B b) { typedef typename std::common_type<A, B>::type T; return (T)a < (T)b ? (T)b : (T)a; }
inline int __mul24(int a, int b) { return (int)((unsigned)((a << 8) >> 8) * (unsigned)((b << 8) >> 8)); }
inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
inline int __float2int_rn(float f) { return (int)std::nearbyintf(f); }

// ---- gfx950 builtins the kernels use
typedef short hipemu_s16x2 __attribute__((ext_vector_type(2)));
typedef int hipemu_i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned hipemu_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned hipemu_u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned hipemu_u32x4 __attribute__((ext_vector_type(4)));
struct __amdgpu_buffer_rsrc_t { unsigned long long base_stride; unsigned num_records; unsigned flags; };     // the 128-bit V# (base 48 bits, stride 14 bits)
static_assert(sizeof(__amdgpu_buffer_rsrc_t) == 16, "V# is four dwords");

namespace hipemu {
inline unsigned perm(unsigned s0, unsigned s1, unsigned sel)      // v_perm_b32: bytes 0-3 of the pool are s1's, 4-7 s0's
{
    const unsigned long long pool = ((unsigned long long)s0 << 32) | s1;
    unsigned r = 0;
    for (int i = 0; i < 4; i++) {
        const unsigned c = (sel >> (8 * i)) & 0xff;
        unsigned b;
        if (c <= 7) b = (unsigned)(pool >> (8 * c)) & 0xff;
        else if (c == 8) b = (s1 & 0x8000u) ? 0xff : 0;
        else if (c == 9) b = (s1 & 0x80000000u) ? 0xff : 0;
        else if (c == 10) b = (s0 & 0x8000u) ? 0xff : 0;
        else if (c == 11) b = (s0 & 0x80000000u) ? 0xff : 0;
        else if (c == 12) b = 0;
        else b = 0xff;
        r |= b << (8 * i);
    }
    return r;
}
inline int sat16(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }
inline unsigned sat_u8(int v) { return v < 0 ? 0u : v > 255 ? 255u : (unsigned)v; }
// two v_ashr_pk_u8_i32: bytes 0..3 = sat_u8(t0 >> n) .. sat_u8(t3 >> n)
inline unsigned pack4_u8_shr(int t0, int t1, int t2, int t3, int n) { return sat_u8(t0 >> n) | (sat_u8(t1 >> n) << 8) | (sat_u8(t2 >> n) << 16) | (sat_u8(t3 >> n) << 24); }
inline int dot2_i16(unsigned a, unsigned b, int c) { return (int)((unsigned)((short)a * (int)(short)b) + (unsigned)((short)(a >> 16) * (int)(short)(b >> 16)) + (unsigned)c); }

// HIPEMU_CHECK_BUFFERS=1: every dword a buffer instruction moves is looked up in the test double's device blocks first.  A LOAD that leaves its block is counted per
// kernel and answers 0 (on the GPU it reads whatever is mapped there -- the descriptor's range check does not cover the SGPR offset the kernels put the row into);
// a STORE that leaves its block aborts.
extern bool check_buffers;
bool in_block(const void *p, unsigned bytes, bool store);
struct Range { const unsigned char *p; unsigned long long room; };
inline Range buf_at(const __amdgpu_buffer_rsrc_t &r, int voff, int soff)
{
    const unsigned char *base = (const unsigned char *)(uintptr_t)(r.base_stride & 0xFFFFFFFFFFFFull);
    const unsigned off = (unsigned)voff;                       // raw buffer, offen: the range check is on the VGPR offset (+ the instruction offset, always 0 here)
    Range g;
    g.p = base + (long long)off + (long long)soff;
    g.room = off < r.num_records ? (unsigned long long)r.num_records - off : 0;
    return g;
}
template <int NDW> inline void buf_load(const __amdgpu_buffer_rsrc_t &r, int voff, int soff, unsigned *out)
{
    const Range g = buf_at(r, voff, soff);
    for (int i = 0; i < NDW; i++) { if (g.room >= 4ull * i + 4 && (!check_buffers || in_block(g.p + 4 * i, 4, false))) std::memcpy(&out[i], g.p + 4 * i, 4); else out[i] = 0; }
}
template <int NDW> inline void buf_store(const __amdgpu_buffer_rsrc_t &r, int voff, int soff, const unsigned *in)
{
    const Range g = buf_at(r, voff, soff);
    for (int i = 0; i < NDW; i++) if (g.room >= 4ull * i + 4 && (!check_buffers || in_block(g.p + 4 * i, 4, true))) std::memcpy(const_cast<unsigned char *>(g.p) + 4 * i, &in[i], 4);
}
inline void buf_store_small(const __amdgpu_buffer_rsrc_t &r, int voff, int soff, unsigned v, int bytes)
{
    const Range g = buf_at(r, voff, soff);
    if (g.room >= (unsigned)bytes && (!check_buffers || in_block(g.p, (unsigned)bytes, true))) std::memcpy(const_cast<unsigned char *>(g.p), &v, bytes);
}
// buffer_load_dwordx4 ... offen lds under an EXEC mask: lane L's 16 bytes go to LDS address M0 + 16 L
inline void dma16(unsigned lds_dst, int voff, const hipemu_i32x4 &rsrc, int soff, unsigned mask_lo, unsigned mask_hi)
{
    const int lane = cur->lane;
    if (!(((((unsigned long long)mask_hi) << 32 | mask_lo) >> lane) & 1)) return;
    __amdgpu_buffer_rsrc_t r;
    std::memcpy(&r, &rsrc, 16);
    unsigned w[4];
    buf_load<4>(r, voff, soff, w);
    std::memcpy(lds_ptr(lds_dst) + 16 * lane, w, 16);
}
} // namespace hipemu

inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void *p, short stride, int num_records, int flags)
{
    __amdgpu_buffer_rsrc_t r;
    r.base_stride = ((unsigned long long)(uintptr_t)p & 0xFFFFFFFFFFFFull) | ((unsigned long long)(unsigned short)stride << 48);
    r.num_records = (unsigned)num_records; r.flags = (unsigned)flags;
    return r;
}
#define __builtin_amdgcn_perm(a, b, s) hipemu::perm((unsigned)(a), (unsigned)(b), (unsigned)(s))
#define __builtin_amdgcn_readfirstlane(v) (v)                     /* the kernels pass wave-uniform values (kernels_common.hpp U()) */
#define __builtin_amdgcn_readlane(v, l) ((int)hipemu::wave_read((unsigned long long)(unsigned)(v), (l)))
#define __builtin_amdgcn_fence(...) hipemu::sync_wave()
#define __builtin_amdgcn_wave_barrier() hipemu::sync_wave()
#define __builtin_amdgcn_s_waitcnt(x) hipemu::sync_wave()
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
inline unsigned __builtin_amdgcn_ubfe(unsigned s, unsigned off, unsigned w) { off &= 31; w &= 31; return w ? (s >> off) & ((1u << w) - 1) : 0u; }
inline hipemu_s16x2 __builtin_amdgcn_cvt_pk_i16(int a, int b) { hipemu_s16x2 r = { (short)hipemu::sat16(a), (short)hipemu::sat16(b) }; return r; }
inline int __builtin_amdgcn_sdot2(hipemu_s16x2 a, hipemu_s16x2 b, int c, bool clamp)
{
    const long long v = (long long)a.x * b.x + (long long)a.y * b.y + c;
    if (clamp) return v > 2147483647ll ? 2147483647 : v < -2147483648ll ? (int)-2147483648ll : (int)v;
    return (int)(unsigned)(unsigned long long)v;
}
inline unsigned __builtin_amdgcn_raw_buffer_load_b32(const __amdgpu_buffer_rsrc_t &r, int voff, int soff, int) { unsigned w; hipemu::buf_load<1>(r, voff, soff, &w); return w; }
inline hipemu_u32x2 __builtin_amdgcn_raw_buffer_load_b64(const __amdgpu_buffer_rsrc_t &r, int voff, int soff, int) { unsigned w[2]; hipemu::buf_load<2>(r, voff, soff, w); hipemu_u32x2 v = { w[0], w[1] }; return v; }
inline hipemu_u32x3 __builtin_amdgcn_raw_buffer_load_b96(const __amdgpu_buffer_rsrc_t &r, int voff, int soff, int) { unsigned w[3]; hipemu::buf_load<3>(r, voff, soff, w); hipemu_u32x3 v = { w[0], w[1], w[2] }; return v; }
inline hipemu_u32x4 __builtin_amdgcn_raw_buffer_load_b128(const __amdgpu_buffer_rsrc_t &r, int voff, int soff, int) { unsigned w[4]; hipemu::buf_load<4>(r, voff, soff, w); hipemu_u32x4 v = { w[0], w[1], w[2], w[3] }; return v; }
template <class T> inline void __builtin_amdgcn_raw_buffer_store_b8(T v, const __amdgpu_buffer_rsrc_t &r, int voff, int soff, int) { hipemu::buf_store_small(r, voff, soff, (unsigned)v, 1); }
template <class T> inline void __builtin_amdgcn_raw_buffer_store_b16(T v, const __amdgpu_buffer_rsrc_t &r, int voff, int soff, int) { hipemu::buf_store_small(r, voff, soff, (unsigned)v, 2); }
template <class T> inline void __builtin_amdgcn_raw_buffer_store_b32(T v, const __amdgpu_buffer_rsrc_t &r, int voff, int soff, int) { unsigned w; static_assert(sizeof(T) == 4, ""); std::memcpy(&w, &v, 4); hipemu::buf_store<1>(r, voff, soff, &w); }
template <class T> inline void __builtin_amdgcn_raw_buffer_store_b64(T v, const __amdgpu_buffer_rsrc_t &r, int voff, int soff, int) { unsigned w[2]; static_assert(sizeof(T) == 8, ""); std::memcpy(w, &v, 8); hipemu::buf_store<2>(r, voff, soff, w); }
template <class T> inline void __builtin_amdgcn_raw_buffer_store_b128(T v, const __amdgpu_buffer_rsrc_t &r, int voff, int soff, int) { unsigned w[4]; static_assert(sizeof(T) == 16, ""); std::memcpy(w, &v, 16); hipemu::buf_store<4>(r, voff, soff, w); }
