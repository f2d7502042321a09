// Wave-level helpers shared by the wave-tiled, marching and streaming kernels (gfx950, wave64):
// 16-byte global / buffer-descriptor accesses, byte unpacking, v_ashr_pk_u8_i32 packing, v_dot2_i32_i16.
#pragma once
#include "kernels_common.hpp"

namespace swsk {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define SWS_GLOBAL __attribute__((address_space(1)))

__device__ __forceinline__ u32x4 gload16(const uint8_t *p) { return *(const SWS_GLOBAL u32x4 *)p; }
__device__ __forceinline__ u32x2 gload8(const uint8_t *p) { return *(const SWS_GLOBAL u32x2 *)p; }
__device__ __forceinline__ void gstore16_nt(uint8_t *p, u32x4 v) { __builtin_nontemporal_store(v, (SWS_GLOBAL u32x4 *)p); }

// row tails: n (< 16 / < 8) valid bytes, the rest reads as 0.  Out of line: rare, keeps the hot loop small.
__device__ __noinline__ u32x4 gload16_partial(const uint8_t *s, int n)
{
    uint32_t w[4] = { 0, 0, 0, 0 };
    for (int k = 0; k < n; k++) w[k >> 2] |= (uint32_t)s[k] << (8 * (k & 3));
    u32x4 v = { w[0], w[1], w[2], w[3] };
    return v;
}
__device__ __noinline__ void gstore_partial(uint8_t *d, u32x4 v, int n)
{
    uint32_t w[4] = { v[0], v[1], v[2], v[3] };
    for (int b = 0; b < n; b++) d[b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
}
__device__ __forceinline__ u32x4 load16_or_tail(const uint8_t *s, int nvalid)
{
    return nvalid >= 16 ? gload16(s) : gload16_partial(s, max(nvalid, 0));
}
__device__ __forceinline__ u32x2 load8_or_tail(const uint8_t *s, int nvalid)
{
    if (nvalid >= 8) return gload8(s);
    const u32x4 t = gload16_partial(s, max(nvalid, 0));
    u32x2 r = { t[0], t[1] };
    return r;
}

__device__ __forceinline__ void unpack16(u32x4 v, int (&o)[16])
{
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int k = 0; k < 4; k++) o[4 * q + k] = (v[q] >> (8 * k)) & 0xFF;
}
__device__ __forceinline__ void unpack8(u32x2 v, int (&o)[8])
{
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int k = 0; k < 4; k++) o[4 * q + k] = (v[q] >> (8 * k)) & 0xFF;
}

// Four values t0..t3 -> one dword of bytes clip_u8(t >> 16).  v_ashr_pk_u8_i32 (new on gfx950) shifts two int32, saturates
// them to u8 and writes ONE HALF of the destination (low half, or high half with op_sel[3]); the other half is preserved
// (semantics verified on hardware with tools/isa_probe.hip).  Two of them clamp, shift and pack 4 channel values.
__device__ __forceinline__ uint32_t pack4_u8_shr16(int t0, int t1, int t2, int t3)
{
    uint32_t d;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 16\n\tv_ashr_pk_u8_i32 %0, %3, %4, 16 op_sel:[0,0,0,1]"
        : "=&v"(d) : "v"(t0), "v"(t1), "v"(t2), "v"(t3));
    return d;
}

__device__ __forceinline__ uint32_t pack4_u8_shr12(int t0, int t1, int t2, int t3)
{
    uint32_t d;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 12\n\tv_ashr_pk_u8_i32 %0, %3, %4, 12 op_sel:[0,0,0,1]"
        : "=&v"(d) : "v"(t0), "v"(t1), "v"(t2), "v"(t3));
    return d;
}
__device__ __forceinline__ int sdot2(uint32_t a, uint32_t b, int c)
{
    typedef short s16x2w __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2w, a), __builtin_bit_cast(s16x2w, b), c, false);
}
// first tap pair of a chain: VOP3P form with an inline 0 addend (the compiler would emit v_mov 0 + v_dot2c)
__device__ __forceinline__ int sdot2_first(uint32_t a, uint32_t b)
{
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ int sdot2_first_s(uint32_t a, uint32_t b_uniform)   // tap pair in an SGPR
{
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "s"(b_uniform));
    return d;
}
// 16 bytes of row A and row B -> 16 dwords (A_k | B_k << 16)
__device__ __forceinline__ void interleave_rows(const u32x4 &A, const u32x4 &B, uint32_t (&P)[16])
{
#pragma unroll
    for (int q = 0; q < 4; q++) {
        P[4 * q + 0] = __builtin_amdgcn_perm(B[q], A[q], 0x0c040c00u);
        P[4 * q + 1] = __builtin_amdgcn_perm(B[q], A[q], 0x0c050c01u);
        P[4 * q + 2] = __builtin_amdgcn_perm(B[q], A[q], 0x0c060c02u);
        P[4 * q + 3] = __builtin_amdgcn_perm(B[q], A[q], 0x0c070c03u);
    }
}

//  * memory goes through buffer descriptors: the per-lane offset is a constant VGPR, the row offset an SGPR (no 64-bit vector
//    address arithmetic), reads past the end of a row stay inside the plane's descriptor (out-of-range dwords read 0 and feed
//    pixels that are never stored), stores past the end of a row are dropped by the per-row destination descriptor.
typedef __amdgpu_buffer_rsrc_t sws_rsrc_t;
__device__ __forceinline__ sws_rsrc_t make_rsrc(const void *base, uint32_t bytes)
{
    // descriptor inputs go through readfirstlane so that their uniformity is provable (else every buffer op gets a waterfall loop)
    void *b = (void *)uniform_u64((uint64_t)base);
    return __builtin_amdgcn_make_buffer_rsrc(b, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}
__device__ __forceinline__ u32x4 bload16(sws_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ u32x2 bload8(sws_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}

} // namespace swsk
