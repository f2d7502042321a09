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
// the frame descriptor by value (a reference that may point at the by-value argument would force that argument into scratch memory)
__device__ __forceinline__ SwsFramePtrs frame_copy(const SwsFrameSet &fs, int idx)
{
    SwsFramePtrs f;
    if (fs.table) f = fs.table[idx]; else f = fs.one;
    return f;
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

// Run-time selection among fields of the by-value argument structs.  A run-time index (p.arr[comp]), and also a plain select between two
// fields ("c ? p.a : p.b", or a chain of them), is turned by LLVM into a load through a computed ADDRESS, which forces the whole struct
// (SwsDevParams is about 800 bytes) into scratch memory for every lane: the generic kernels ran at 2.7 Gpix/s because of it.  Every value
// that takes part in such a selection therefore goes through U(): readfirstlane of a wave-uniform value is free, and a select over
// intrinsic results stays a select over values.
__device__ __forceinline__ int U(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t U(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
template <typename T> __device__ __forceinline__ const T *U(const T *q) { return (const T *)uniform_u64((uint64_t)q); }
template <typename T> __device__ __forceinline__ T *U(T *q) { return (T *)uniform_u64((uint64_t)q); }
__device__ __forceinline__ int64_t U(int64_t v) { return (int64_t)uniform_u64((uint64_t)v); }
template <typename T> __device__ __forceinline__ T pick4(const T (&a)[4], int i)
{
    const T a0 = U(a[0]), a1 = U(a[1]), a2 = U(a[2]), a3 = U(a[3]);
    return i == 0 ? a0 : i == 1 ? a1 : i == 2 ? a2 : a3;
}
template <typename T> __device__ __forceinline__ T pick5(const T (&a)[5], int i)
{
    const T a0 = U(a[0]), a1 = U(a[1]), a2 = U(a[2]), a3 = U(a[3]), a4 = U(a[4]);
    return i == 0 ? a0 : i == 1 ? a1 : i == 2 ? a2 : i == 3 ? a3 : a4;
}
struct Rgb2YuvRow { int32_t r, g, b; };   // one row of the rgb2yuv table: o = 0 (Y), 3 (U), 6 (V)
__device__ __forceinline__ Rgb2YuvRow rgb2yuv_row(const int32_t (&t)[9], int o)
{
    const int t0 = U(t[0]), t1 = U(t[1]), t2 = U(t[2]), t3 = U(t[3]), t4 = U(t[4]), t5 = U(t[5]), t6 = U(t[6]), t7 = U(t[7]), t8 = U(t[8]);
    Rgb2YuvRow w;
    w.r = o == 0 ? t0 : o == 3 ? t3 : t6; w.g = o == 0 ? t1 : o == 3 ? t4 : t7; w.b = o == 0 ? t2 : o == 3 ? t5 : t8;
    return w;
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

// 30 bpp pixel (yuv2rgb.c:915-941 y_table32 planes with 10-bit ramps, summed by yuv2rgb_write output.c:1748-1754)
__device__ __forceinline__ uint32_t lut_luma10(const SwsLutParams &L, int k)
{
    const int v = (L.yb0r + __mul24(k, L.cy)) >> 14;
    return (uint32_t)(v < 0 ? 0 : v > 1023 ? 1023 : v);
}
__device__ __forceinline__ uint32_t lut_rgb30(const SwsLutParams &L, const ChromaIdx &k, int Y)
{
    return (lut_luma10(L, k.r + Y) << L.rshift) + (lut_luma10(L, k.g + Y) << L.gshift) + (lut_luma10(L, k.b + Y) << L.bshift) + L.alpha_or;
}

// 12/15/16 bpp pixel from the three luma-table indices (yuv2rgb.c:853-897: y_table16 planes, summed by yuv2rgb_write)
__device__ __forceinline__ uint32_t lut_rgb16(const SwsLutParams &L, int ir, int ig, int ib)
{
    const int r = lut_luma(L, ir), g = lut_luma(L, ig), b = lut_luma(L, ib);
    if (L.bpp16 == 12) return (uint32_t)(((r >> 4) << L.r16) | ((g >> 4) << L.g16) | ((b >> 4) << L.b16));
    return (uint32_t)(((r >> 3) << L.r16) | ((g >> (18 - L.bpp16)) << L.g16) | ((b >> 3) << L.b16));
}
// ordered dither of the 12/15/16 bpp writers (output.c:40-58 ff_dither_2x2_4 / 2x2_8 / 4x4_16), row r, column c
__device__ __forceinline__ int dither_2x2_4(int r, int c) { return (r & 1) ? ((c & 1) ? 0 : 2) : ((c & 1) ? 3 : 1); }
__device__ __forceinline__ int dither_2x2_8(int r, int c) { return (r & 1) ? ((c & 1) ? 4 : 0) : ((c & 1) ? 2 : 6); }
__device__ __forceinline__ int dither_4x4_16(int r, int c)
{
    const uint32_t rows[4] = { 0x070b0408u, 0x0d010e02u, 0x0509060au, 0x0f030c00u };   // { 8,4,11,7 }, { 2,14,1,13 }, { 10,6,9,5 }, { 0,12,3,15 }
    return (int)((rows[r & 3] >> (8 * (c & 3))) & 0xff);
}
// yuv2rgb_write (output.c:1721-1745): dither of channel ch (0 r, 1 g, 2 b) for the pixel of parity e in destination row y
__device__ __forceinline__ int dither_rgb16_main(int bpp, int y, int e, int ch)
{
    if (bpp == 16) return ch == 0 ? dither_2x2_8(y, e) : ch == 1 ? dither_2x2_4(y, e) : dither_2x2_8(y ^ 1, e);
    if (bpp == 15) return ch == 0 ? dither_2x2_8(y, e) : ch == 1 ? dither_2x2_8(y, e ^ 1) : dither_2x2_8(y ^ 1, e);
    return ch == 0 ? dither_4x4_16(y, e) : ch == 1 ? dither_4x4_16(y, e ^ 1) : dither_4x4_16(y ^ 3, e);
}

// ---- 8 / 4 bpp destinations (rgb8, bgr8, rgb4, bgr4, rgb4_byte, bgr4_byte) ----
// ff_dither_8x8_32 / ff_dither_8x8_73 / ff_dither_8x8_220 (output.c:60-95); the reference's ninth row repeats the first
__device__ __constant__ const uint8_t k_dither_8x8_32[8][8] = {
    { 17, 9, 23, 15, 16, 8, 22, 14 }, { 5, 29, 3, 27, 4, 28, 2, 26 }, { 21, 13, 19, 11, 20, 12, 18, 10 }, { 0, 24, 6, 30, 1, 25, 7, 31 },
    { 16, 8, 22, 14, 17, 9, 23, 15 }, { 4, 28, 2, 26, 5, 29, 3, 27 }, { 20, 12, 18, 10, 21, 13, 19, 11 }, { 1, 25, 7, 31, 0, 24, 6, 30 },
};
__device__ __constant__ const uint8_t k_dither_8x8_73[8][8] = {
    { 0, 55, 14, 68, 3, 58, 17, 72 }, { 37, 18, 50, 32, 40, 22, 54, 35 }, { 9, 64, 5, 59, 13, 67, 8, 63 }, { 46, 27, 41, 23, 49, 31, 44, 26 },
    { 2, 57, 16, 71, 1, 56, 15, 70 }, { 39, 21, 52, 34, 38, 19, 51, 33 }, { 11, 66, 7, 62, 10, 65, 6, 60 }, { 48, 30, 43, 25, 47, 29, 42, 24 },
};
__device__ __constant__ const uint8_t k_dither_8x8_220[8][8] = {
    { 117, 62, 158, 103, 113, 58, 155, 100 }, { 34, 199, 21, 186, 31, 196, 17, 182 }, { 144, 89, 131, 76, 141, 86, 127, 72 },
    { 0, 165, 41, 206, 10, 175, 52, 217 }, { 110, 55, 151, 96, 120, 65, 162, 107 }, { 28, 193, 14, 179, 38, 203, 24, 189 },
    { 138, 83, 124, 69, 148, 93, 134, 79 }, { 7, 172, 48, 213, 3, 168, 45, 210 },
};
// one field of the byte tables of yuv2rgb.c:817-856.  A plane holds a ramp of n elements that starts `off` elements in, so that
// index + dither is a centred threshold; what lies around the ramp is unwritten in the reference (never reached) and reads 0 here,
// like the oracle.  kind 0: yval >> 7, 1: (yval + 43) / 85, 2: (yval + 18) / 36
__device__ __forceinline__ int lut8_field(const SwsLutParams &L, int k, int off, int n, int kind)
{
    const int j = k - off;
    if ((unsigned)j >= (unsigned)n) return 0;
    const int yval = lut_luma(L, j);
    return kind == 0 ? yval >> 7 : kind == 1 ? (yval + 43) / 85 : (yval + 18) / 36;
}
// pixel of destination row y, column x from the chroma indices and the luma value (yuv2rgb_write output.c:1755-1784: the dither of
// the column is added to the table index)
__device__ __forceinline__ uint32_t lut_rgb8(const SwsLutParams &L, const ChromaIdx &k, int Y, int y, int x)
{
    const int TPS = 2048;   // table_plane_size = 1024 + 2 * YUVRGB_TABLE_LUMA_HEADROOM
    if (L.bpp8 == 8) {
        const int d32 = k_dither_8x8_32[y & 7][x & 7], d73 = k_dither_8x8_73[y & 7][x & 7];
        return (uint32_t)((lut8_field(L, k.r + Y + d32, 16, TPS - 38, 2) << L.r8) + (lut8_field(L, k.g + Y + d32, 16, TPS - 38, 2) << L.g8) +
                          (lut8_field(L, k.b + Y + d73, 37, TPS - 38, 1) << L.b8));
    }
    const int d220 = k_dither_8x8_220[y & 7][x & 7], d73 = k_dither_8x8_73[y & 7][x & 7];
    return (uint32_t)((lut8_field(L, k.r + Y + d220, 110, TPS - 110, 0) << L.r8) + (lut8_field(L, k.g + Y + d73, 37, TPS - 110, 1) << L.g8) +
                      (lut8_field(L, k.b + Y + d220, 110, TPS - 110, 0) << L.b8));
}
// yuv2rgb_write_full's 8 / 4 bpp arm without error diffusion (output.c:2064-2158): R, G, B are the clipped 30-bit sums
__device__ __forceinline__ uint32_t full_rgb8(const SwsLutParams &L, int R, int G, int B, int i, int y)
{
    const bool rgb8 = L.bpp8 == 8;
    int r, g, b;
    if (L.dither8 == 0) {   // SWS_DITHER_NONE
        if (rgb8) { r = clip_uintp2(R >> 27, 3); g = clip_uintp2(G >> 27, 3); b = clip_uintp2(B >> 28, 2); }
        else { r = clip_uintp2(R >> 29, 1); g = clip_uintp2(G >> 28, 2); b = clip_uintp2(B >> 29, 1); }
    } else {
        int dr, dg, db;
        if (L.dither8 == 4) {   // A_DITHER(u, v) = ((u + v * 236) * 119) & 0xff
            dr = ((i + y * 236) * 119) & 0xff; dg = ((i + 17 + y * 236) * 119) & 0xff; db = ((i + 34 + y * 236) * 119) & 0xff;
        } else {                // X_DITHER(u, v) = (((u ^ (v * 237)) * 181) & 0x1ff) / 2
            dr = (((i ^ (y * 237)) * 181) & 0x1ff) / 2; dg = ((((i + 17) ^ (y * 237)) * 181) & 0x1ff) / 2; db = ((((i + 34) ^ (y * 237)) * 181) & 0x1ff) / 2;
        }
        if (rgb8) {
            r = clip_uintp2(((R >> 19) + dr - 96) >> 8, 3); g = clip_uintp2(((G >> 19) + dg - 96) >> 8, 3); b = clip_uintp2(((B >> 20) + db - 96) >> 8, 2);
        } else {
            r = clip_uintp2(((R >> 21) + dr - 256) >> 8, 1); g = clip_uintp2(((G >> 19) + dg - 256) >> 8, 2); b = clip_uintp2(((B >> 21) + db - 256) >> 8, 1);
        }
    }
    return (uint32_t)((r << L.r8) + (g << L.g8) + (b << L.b8));
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
