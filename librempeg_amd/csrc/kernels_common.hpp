// Device-side helpers shared by all HIP kernels of libswscale_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "devparams.h"

namespace swsk {

__device__ __forceinline__ int clip_u8(int a) { return min(max(a, 0), 255); }
// clip_u8(x >> sh), written as clamp-then-shift.  hipcc 7.2 pattern-matches sat_u8(x >> sh) pairs into the new
// gfx950 instruction v_ashr_pk_u8_i32 and then ORs further bytes into the upper half of its result, but the
// instruction leaves stale data in bits 31:16 -> corrupted 3rd/4th bytes (caught by the parity tests against
// the oracle).  Clamping first is arithmetically identical and does not match that pattern.
__device__ __forceinline__ int clip_u8_shr(int x, int sh) { return min(max(x, 0), (256 << sh) - 1) >> sh; }
// a*b + c with |a|,|b| < 2^23: v_mad_i32_i24 (full rate; v_mul_lo_u32 is quarter rate)
__device__ __forceinline__ int mad24(int a, int b, int c) { return __mul24(a, b) + c; }
__device__ __forceinline__ int clip_u16(int a) { return min(max(a, 0), 65535); }
__device__ __forceinline__ int clip_i16(int a) { return min(max(a, -32768), 32767); }
__device__ __forceinline__ int clip_uintp2(int a, int p) { return min(max(a, 0), (1 << p) - 1); }

__device__ __forceinline__ const SwsFramePtrs &frame_of(const SwsFrameSet &fs, int idx)
{
    return fs.table ? fs.table[idx] : fs.one;
}

// Frame descriptor in scalar registers: the frame index is wave-uniform (blockIdx.z), so every plane pointer
// and stride is read once and pinned to SGPRs; later address arithmetic is scalar-base + vector-offset.
struct FrameRegs {
    const uint8_t *src[4]; uint8_t *dst[4]; int srcStride[4]; int dstStride[4];
};
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ FrameRegs load_frame(const SwsFrameSet &fs, int idx)
{
    const SwsFramePtrs &f = frame_of(fs, idx);
    FrameRegs r;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        r.src[k] = (const uint8_t *)uniform_u64((uint64_t)f.src[k]);
        r.dst[k] = (uint8_t *)uniform_u64((uint64_t)f.dst[k]);
        r.srcStride[k] = __builtin_amdgcn_readfirstlane(f.srcStride[k]);
        r.dstStride[k] = __builtin_amdgcn_readfirstlane(f.dstStride[k]);
    }
    return r;
}

// ---- closed form of the yuv2rgb LUTs (yuv2rgb.c:680-703, :901-961; colorspace.cpp) ----
struct ChromaIdx { int r, g, b; };
__device__ __forceinline__ ChromaIdx lut_chroma(const SwsLutParams &L, int U, int V)
{
    const int cu = clip_u8(U), cv = clip_u8(V);
    ChromaIdx k;
    // operands are < 2^23 in magnitude (checked on the host): full-rate 24-bit multiplies
    k.r = L.base_r + (__mul24(cv, L.crv) >> 16);
    k.g = L.base_g + (__mul24(cu, L.cgu) >> 16) + (__mul24(cv, L.cgv) >> 16);
    k.b = L.base_b + (__mul24(cu, L.cbu) >> 16);
    return k;
}
__device__ __forceinline__ int lut_luma(const SwsLutParams &L, int k) // y_table[k]
{
    return clip_u8_shr(L.yb0r + __mul24(k, L.cy), 16);
}
__device__ __forceinline__ uint32_t lut_rgb32(const SwsLutParams &L, const ChromaIdx &k, int Y)
{
    return ((uint32_t)lut_luma(L, k.r + Y) << L.rshift) | ((uint32_t)lut_luma(L, k.g + Y) << L.gshift) |
           ((uint32_t)lut_luma(L, k.b + Y) << L.bshift) | L.alpha_or;
}

// ordered-dither rows (swscale.c:42-52 ff_dither_8x8_128, :54 sws_pb_64)
__device__ __constant__ const uint8_t k_dither_8x8_128[8][8] = {
    {  36, 68,  60, 92,  34, 66,  58, 90 }, { 100,  4, 124, 28,  98,  2, 122, 26 },
    {  52, 84,  44, 76,  50, 82,  42, 74 }, { 116, 20, 108, 12, 114, 18, 106, 10 },
    {  32, 64,  56, 88,  38, 70,  62, 94 }, {  96,  0, 120, 24, 102,  6, 126, 30 },
    {  48, 80,  40, 72,  54, 86,  46, 78 }, { 112, 16, 104,  8, 118, 22, 110, 14 },
};
__device__ __forceinline__ int dither8(bool should_dither, int row, int col)
{
    return should_dither ? k_dither_8x8_128[row & 7][col & 7] : 64;
}

// lrintf(av_clipf(65535.0f * x, 0, 65535)) (input.c:1300): round-to-nearest-even like the host default mode
__device__ __forceinline__ int f32_to_u16(float x)
{
    float v = 65535.0f * x;
    v = fmaxf(v, 0.0f);       // av_clipf_c = FFMIN(FFMAX(a, amin), amax); NaN inputs are unspecified in the reference
    v = fminf(v, 65535.0f);
    return __float2int_rn(v);
}

} // namespace swsk
