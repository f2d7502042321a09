// Element-per-thread helpers of the unscaled converters and of the passes around the main path: planarToNv12 / nv12ToPlanar /
// nv24 / yvu9 / planarCopy (+DITHER_COPY), alpha-plane fill and merge, byte swapping.  Non-template kernels: this header belongs to
// exactly one translation unit (k_misc.hip).
#pragma once
#include "kernels_common.hpp"

namespace swsk {

// ------------------------------------------------------------------------------------------
// remaining unscaled planar converters, one element per thread:
//   mode 0 planarToNv12Wrapper / 1 nv12ToPlanarWrapper (swscale_unscaled.c:147-186)
//   mode 2 planarCopyWrapper (:2220-2384; little-endian, incl. depth change with ordered dither)
// grid.x over bytes/elements of a row, grid.y over rows of all planes stacked, grid.z = frame
// ------------------------------------------------------------------------------------------
__device__ __constant__ const uint8_t k_copy_dithers[8][8][8] = { // swscale_unscaled.c:39-112
{ {0,1,0,1,0,1,0,1},{1,0,1,0,1,0,1,0},{0,1,0,1,0,1,0,1},{1,0,1,0,1,0,1,0},{0,1,0,1,0,1,0,1},{1,0,1,0,1,0,1,0},{0,1,0,1,0,1,0,1},{1,0,1,0,1,0,1,0} },
{ {1,2,1,2,1,2,1,2},{3,0,3,0,3,0,3,0},{1,2,1,2,1,2,1,2},{3,0,3,0,3,0,3,0},{1,2,1,2,1,2,1,2},{3,0,3,0,3,0,3,0},{1,2,1,2,1,2,1,2},{3,0,3,0,3,0,3,0} },
{ {2,4,3,5,2,4,3,5},{6,0,7,1,6,0,7,1},{3,5,2,4,3,5,2,4},{7,1,6,0,7,1,6,0},{2,4,3,5,2,4,3,5},{6,0,7,1,6,0,7,1},{3,5,2,4,3,5,2,4},{7,1,6,0,7,1,6,0} },
{ {4,8,7,11,4,8,7,11},{12,0,15,3,12,0,15,3},{6,10,5,9,6,10,5,9},{14,2,13,1,14,2,13,1},{4,8,7,11,4,8,7,11},{12,0,15,3,12,0,15,3},{6,10,5,9,6,10,5,9},{14,2,13,1,14,2,13,1} },
{ {9,17,15,23,8,16,14,22},{25,1,31,7,24,0,30,6},{13,21,11,19,12,20,10,18},{29,5,27,3,28,4,26,2},{8,16,14,22,9,17,15,23},{24,0,30,6,25,1,31,7},{12,20,10,18,13,21,11,19},{28,4,26,2,29,5,27,3} },
{ {18,34,30,46,17,33,29,45},{50,2,62,14,49,1,61,13},{26,42,22,38,25,41,21,37},{58,10,54,6,57,9,53,5},{16,32,28,44,19,35,31,47},{48,0,60,12,51,3,63,15},{24,40,20,36,27,43,23,39},{56,8,52,4,59,11,55,7} },
{ {18,34,30,46,17,33,29,45},{50,2,62,14,49,1,61,13},{26,42,22,38,25,41,21,37},{58,10,54,6,57,9,53,5},{16,32,28,44,19,35,31,47},{48,0,60,12,51,3,63,15},{24,40,20,36,27,43,23,39},{56,8,52,4,59,11,55,7} },
{ {36,68,60,92,34,66,58,90},{100,4,124,28,98,2,122,26},{52,84,44,76,50,82,42,74},{116,20,108,12,114,18,106,10},{32,64,56,88,38,70,62,94},{96,0,120,24,102,6,126,30},{48,80,40,72,54,86,46,78},{112,16,104,8,118,22,110,14} },
};

struct MiscPlane { int srcPlane, dstPlane, width /*elements*/, rows, y0, elem /*bytes per element*/, shiftonly, chroma; };
struct MiscPlan { int mode; int nplanes; int bytecopy; int aux, aux2; MiscPlane pl[4]; };

__global__ void __launch_bounds__(256) sws_k_planar_misc(SwsFrameSet fs, SwsDevParams p, MiscPlan plan)
{
    const SwsFramePtrs &f = frame_of(fs, blockIdx.z);
    int r = blockIdx.y, pi = 0;
    while (pi < plan.nplanes && r >= plan.pl[pi].rows) { r -= plan.pl[pi].rows; pi++; }
    if (pi >= plan.nplanes) return;
    const MiscPlane &P = plan.pl[pi];
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= P.width) return;
    const int ys = P.y0 + r, yd = P.y0 + r;  // absolute plane rows (the host rebases slice pointers)
    if (plan.mode == 0) {             // planar -> nv12/nv21
        if (pi == 0) { f.dst[0][(int64_t)yd * f.dstStride[0] + x] = f.src[0][(int64_t)ys * f.srcStride[0] + x]; return; }
        const int a = p.uv_swap_dst ? 2 : 1, b = 3 - a;
        uint8_t *d = f.dst[1] + (int64_t)yd * f.dstStride[1] + 2 * x;
        d[0] = f.src[a][(int64_t)ys * f.srcStride[a] + x];
        d[1] = f.src[b][(int64_t)ys * f.srcStride[b] + x];
    } else if (plan.mode == 1) {      // nv12/nv21 -> planar
        if (pi == 0) { f.dst[0][(int64_t)yd * f.dstStride[0] + x] = f.src[0][(int64_t)ys * f.srcStride[0] + x]; return; }
        const int a = p.uv_swap_src ? 2 : 1, b = 3 - a;
        const uint8_t *s = f.src[1] + (int64_t)ys * f.srcStride[1] + 2 * x;
        f.dst[a][(int64_t)yd * f.dstStride[a] + x] = s[0];
        f.dst[b][(int64_t)yd * f.dstStride[b] + x] = s[1];
    } else if (plan.mode == 3) {      // nv24ToYuv420Wrapper: luma copy + truncating 2x2 chroma mean (swscale_unscaled.c:229-271)
        if (pi == 0) { f.dst[0][(int64_t)yd * f.dstStride[0] + x] = f.src[0][(int64_t)ys * f.srcStride[0] + x]; return; }
        const int a = p.uv_swap_src ? 2 : 1, b = 3 - a;
        // chroma source rows are luma rows: r counts chroma rows of the slice, the slice starts at luma row 2*P.y0
        const int sr = 2 * r, sr2 = (sr + 1 == plan.aux) ? sr : sr + 1;
        const uint8_t *s1 = f.src[1] + (int64_t)(2 * P.y0 + sr) * f.srcStride[1] + 4 * x;
        const uint8_t *s2 = f.src[1] + (int64_t)(2 * P.y0 + sr2) * f.srcStride[1] + 4 * x;
        f.dst[a][(int64_t)yd * f.dstStride[a] + x] = (uint8_t)((s1[0] + s1[2] + s2[0] + s2[2]) >> 2);
        f.dst[b][(int64_t)yd * f.dstStride[b] + x] = (uint8_t)((s1[1] + s1[3] + s2[1] + s2[3]) >> 2);
    } else if (plan.mode == 4) {      // yvu9ToYv12Wrapper: luma copy + planar2x_c chroma (rgb2rgb_template.c:531-574), per slice
        if (pi == 0) { f.dst[0][(int64_t)yd * f.dstStride[0] + x] = f.src[0][(int64_t)ys * f.srcStride[0] + x]; return; }
        const int W = plan.aux2, H = plan.aux;                 // source chroma size of this slice
        const int s0row = P.y0 >> 1;                           // first source chroma row of the slice (absolute)
        const uint8_t *src = f.src[P.srcPlane] + (int64_t)s0row * f.srcStride[P.srcPlane];
        const int st = f.srcStride[P.srcPlane];
        const int Y = r, X = x;
        int v;
        // column taps: output X=0 and X=2W-1 take one source column; odd X = 2k+1 mixes k (near) and k+1, even X = 2k+2 mixes k+1 (near) and k
        const int k = (X - 1) >> 1;
        const bool edgeL = X == 0, edgeR = X == 2 * W - 1;
        const int cn = edgeL ? 0 : edgeR ? W - 1 : (X & 1) ? k : k + 1;     // "near" column (weight 3 horizontally on border lines)
        const int cf = edgeL ? 0 : edgeR ? W - 1 : (X & 1) ? k + 1 : k;     // "far" column
        if (Y == 0 || Y == 2 * H - 1) {                        // first / last line: horizontal taps only
            const uint8_t *s = src + (int64_t)(Y ? H - 1 : 0) * st;
            v = (edgeL || edgeR) ? s[cn] : (3 * s[cn] + s[cf]) >> 2;
        } else {                                               // lines 2y-1 (3A+B) and 2y (A+3B): A = row y-1, B = row y, diagonal taps
            const int y = (Y + 1) >> 1;
            const uint8_t *A = src + (int64_t)(y - 1) * st, *B = A + st;
            if (Y & 1) v = (3 * A[cn] + B[cf]) >> 2;           // dst[2x+1] = 3A[x]+B[x+1]; dst[2x+2] = 3A[x+1]+B[x]
            else       v = (A[cf] + 3 * B[cn]) >> 2;           // dst[2x+2] = A[x]+3B[x+1]; dst[2x+1] = A[x+1]+3B[x]
        }
        f.dst[P.dstPlane][(int64_t)yd * f.dstStride[P.dstPlane] + x] = (uint8_t)v;
    } else if (P.srcPlane < 0) {      // planarCopyWrapper, plane missing in a gray source: fillPlane / fillPlane16 (:2239-2247)
        uint8_t *drow = f.dst[P.dstPlane] + (int64_t)yd * f.dstStride[P.dstPlane];
        if (P.srcPlane == -2) {                                   // alpha plane of a destination the source cannot feed: 255 / all ones
            if (p.copy_depth_dst > 8) ((uint16_t *)drow)[x] = (uint16_t)(0xFFFF >> (16 - p.copy_depth_dst)); else drow[x] = 255;
        }
        else if (p.copy_depth_dst > 8) ((uint16_t *)drow)[x] = (uint16_t)(1 << (p.copy_depth_dst - 1));
        else drow[x] = 128;
    } else {                          // planarCopyWrapper
        const uint8_t *srow = f.src[P.srcPlane] + (int64_t)ys * f.srcStride[P.srcPlane];
        uint8_t *drow = f.dst[P.dstPlane] + (int64_t)yd * f.dstStride[P.dstPlane];
        const int sd = p.copy_depth_src, dd = p.copy_depth_dst, ss = p.copy_shift_src, dsh = p.copy_shift_dst;
        if (plan.bytecopy) {           // same layout: byte copy (P.width counts bytes)
            drow[x] = srow[x];
        } else if (dd == 8 || sd > dd) { // DITHER_COPY (:2159-2218); the macro's scalar tail skips src_shift / dst_shift
            const unsigned shift = sd - dd, bias = 1u << (shift - 1);
            const int body_end = P.width - 7 > 0 ? ((P.width - 7 + 7) / 8) * 8 : 0;
            const bool body = x < body_end;
            const unsigned sv = ((const uint16_t *)srow)[x];
            const unsigned dth = k_copy_dithers[shift - 1][r & 7][x & 7];
            unsigned tmp, v;
            if (p.dither_mode == 0) {
                if (body) { tmp = ((sv >> ss) + bias) >> shift; v = (tmp - (tmp >> dd)) << dsh; }
                else { tmp = (sv + bias) >> shift; v = (tmp - (tmp >> dd)) << dsh; }
            } else if (P.shiftonly) {
                if (body) { tmp = ((sv >> ss) + dth) >> shift; v = (tmp - (tmp >> dd)) << dsh; }
                else { tmp = (sv + dth) >> shift; v = (tmp - (tmp >> dd)) << dsh; }
            } else {
                if (body) { tmp = sv >> ss; v = ((tmp - (tmp >> dd) + dth) >> shift) << dsh; }
                else { tmp = sv; v = (tmp - (tmp >> dd) + dth) >> shift; }
            }
            if (dd == 8) drow[x] = (uint8_t)v; else ((uint16_t *)drow)[x] = (uint16_t)v;
        } else if (sd == 8) {           // 8 -> N (:2266-2284)
            const unsigned s8 = srow[x];
            ((uint16_t *)drow)[x] = P.shiftonly ? (uint16_t)((s8 << (dd - 8)) << dsh)
                                                : (uint16_t)(((s8 << (dd - 8)) | (s8 >> (2 * 8 - dd))) << dsh);
        } else {                        // N -> M up (:2285-2331)
            const unsigned shift = dd - sd, v = ((const uint16_t *)srow)[x] >> ss;
            ((uint16_t *)drow)[x] = P.shiftonly ? (uint16_t)((v << shift) << dsh)
                                                : (uint16_t)(((v << shift) | (v >> (2 * sd - dd))) << dsh);
        }
    }
}

// destination alpha plane the source cannot feed: fillPlane(dst[3], ..., 255) (swscale.c:536-552)
// bits: 0 = fillPlane(…, 255); 9..16 = fillPlane16(…, alpha = 1, bits) = 0xFFFF >> (16 - bits) in 16-bit samples
__global__ void __launch_bounds__(256) sws_k_fill_alpha_plane(SwsFrameSet fs, int w, int y0, int bits)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= w) return;
    const SwsFramePtrs &f = frame_of(fs, blockIdx.z);
    uint8_t *row = f.dst[3] + (int64_t)(y0 + blockIdx.y) * f.dstStride[3];
    if (bits == 32) ((uint32_t *)row)[x] = 0x3f800000u;   // fillPlane32, float: 1.0f (swscale_internal.h:1082-1100)
    else if (bits) ((uint16_t *)row)[x] = (uint16_t)(0xFFFF >> (16 - bits));
    else row[x] = 255;
}

// yuva2rgba_c / yuva2argb_c (PUTRGBA, yuv2rgb.c:101-105): the 32 bpp LUT converter's pixels (written with A = 0 because
// the table was built for a source with alpha) get the source alpha byte.  npix = pixels per row the converter covered.
__global__ void __launch_bounds__(256) sws_k_alpha_merge(SwsFrameSet fs, int npix, int y0, int a_pos)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= npix) return;
    const SwsFramePtrs &f = frame_of(fs, blockIdx.z);
    const int y = y0 + blockIdx.y;   // absolute row (the host rebases slice pointers)
    f.dst[0][(int64_t)y * f.dstStride[0] + 4 * x + a_pos] = f.src[3][(int64_t)y * f.srcStride[3] + x];
}

// big-endian pictures: byte-swap pass around the little-endian conversion (context.cpp: be_alias); unit = 2 or 4 bytes
__global__ void __launch_bounds__(256) sws_k_bswap(const uint8_t *src, int64_t sstride, uint8_t *dst, int64_t dstride, int rows, int row_bytes, int unit)
{
    const int i = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (y >= rows || (i + 1) * unit > row_bytes) return;
    const uint8_t *s = src + y * sstride + (int64_t)i * unit;
    uint8_t *d = dst + y * dstride + (int64_t)i * unit;
    if (unit == 2) { const uint16_t v = *(const uint16_t *)s; *(uint16_t *)d = (uint16_t)((v >> 8) | (v << 8)); }
    else { const uint32_t v = *(const uint32_t *)s; *(uint32_t *)d = __builtin_bswap32(v); }
}

// planarRgbToplanarRgbWrapper (swscale_unscaled.c:1380-1402) with ff_copyPlane (:126-145) as it is: the width is handed over in pixels and
// used as a byte count, so a 16-bit row is copied in full only when the two strides are equal (one memcpy over the slice: whole rows
// but for the last one).  Thread = byte of a row of plane `plane`.
__global__ void __launch_bounds__(256) sws_k_planarrgb_copy(SwsFrameSet fs, int plane, int w, int row_bytes, int y0, int rows)
{
    const int x = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    const SwsFramePtrs &f = frame_of(fs, blockIdx.z);
    const bool whole = f.srcStride[plane] == f.dstStride[plane] && f.srcStride[plane] > 0 && r < rows - 1;
    if (x >= (whole ? row_bytes : w)) return;
    f.dst[plane][(int64_t)(y0 + r) * f.dstStride[plane] + x] = f.src[plane][(int64_t)(y0 + r) * f.srcStride[plane] + x];
}

// gamma_convert (gamma.c:31-58): table look-up on the R, G, B words of an RGBA64LE picture, in place; the alpha word stays
__global__ void __launch_bounds__(256) sws_k_gamma_rgba64(uint8_t *img, int64_t stride, int w, int rows, const uint16_t *__restrict__ table)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= rows) return;
    uint16_t *px = (uint16_t *)(img + y * stride) + 4 * x;
    px[0] = table[px[0]]; px[1] = table[px[1]]; px[2] = table[px[2]];
}

// Error diffusion of yuv2rgb_write_full (output.c:2084-2108) for rgb8 / bgr8 / rgb4_byte / bgr4_byte, over the rgb24 picture the
// inner context wrote (R >> 22, G >> 22, B >> 22).  Pixel (y, i) needs the error of (y, i - 1) and those of (y - 1, i - 1 .. i + 1):
// a wavefront.  One workgroup; thread r owns row y0 + r of the current group of up to 1024 rows and runs two pixels behind thread
// r - 1, so that step t handles pixel t - 2r of every row.  Rows hand their errors down through a 4-slot LDS ring per row; the row
// above the group and the row that outlives the picture live in `errline` (3 x (W + 3) ints, the reference's c->dither_error:
// element k holds the error of pixel k - 1, elements 0, W + 1 and W + 2 stay 0).  The line is kept between frames like the
// reference's, which never resets it.
__global__ void __launch_bounds__(1024) sws_k_ed_rgb8(const uint8_t *__restrict__ rgb, int64_t rgbStride, uint8_t *__restrict__ dst, int64_t dstStride,
                                                      int W, int H, int *__restrict__ errline, int bpp8, int r8, int g8, int b8)
{
    __shared__ int ring[3][1024][4];
    const int r = threadIdx.x;
    const bool rgb8 = bpp8 == 8;
    const int shr = rgb8 ? 5 : 7, shg = rgb8 ? 5 : 6, shb = rgb8 ? 6 : 7;
    const int maxr = rgb8 ? 7 : 1, maxg = rgb8 ? 7 : 3, maxb = rgb8 ? 3 : 1;
    const int qr = rgb8 ? 36 : 255, qg = rgb8 ? 36 : 85, qb = rgb8 ? 85 : 255;
    int *el[3] = { errline, errline + (W + 3), errline + 2 * (W + 3) };
    for (int y0 = 0; y0 < H; y0 += 1024) {
        const int NR = min(1024, H - y0);
        const int y = y0 + r;
        const bool live = r < NR, last = r == NR - 1;
        const uint8_t *srow = rgb + (int64_t)y * rgbStride;
        uint8_t *drow = dst + (int64_t)y * dstStride;
        const bool dw_ok = (((uintptr_t)drow) & 3) == 0;      // four output bytes go out as one dword where the row allows it
        int err[3] = { 0, 0, 0 };
        const int steps = W + 2 * (NR - 1);
        // Nothing that comes from HBM is on the recurrence.  The input pixels (three dwords per four pixels) and, for the first row of the
        // group, the error line above it (six ints per channel and four pixels) are fetched one four-pixel group ahead of their use; the
        // per-step barrier waits for LDS only (s_waitcnt lgkmcnt(0); s_barrier), so those loads and the byte / error-line stores stay in
        // flight across steps.  (A dword of the input is read only where it lies inside the row's linesize.)
        uint32_t cur[3] = { 0, 0, 0 }, nxt[3] = { 0, 0, 0 };
        int elc[3][6], eln[3][6];
        auto fetch = [&](int g, uint32_t (&q)[3]) {   // pixels 4g .. 4g + 3
#pragma unroll
            for (int k = 0; k < 3; k++) { const int64_t off = 12 * (int64_t)g + 4 * k; q[k] = (live && off + 4 <= rgbStride) ? *(const uint32_t *)(srow + off) : 0u; }
        };
        auto fetch_el = [&](int g, int (&q)[3][6]) {  // el[c][4g .. 4g + 5]: the errors of pixels 4g - 1 .. 4g + 4 of the line above
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int k = 0; k < 6; k++) { const int idx = 4 * g + k; q[c][k] = (r == 0 && idx < W + 3) ? el[c][idx] : 0; }
        };
        fetch(0, nxt);
        fetch_el(0, eln);
        uint32_t outw = 0;
        for (int t = 0; t < steps; t++) {
            const int i = t - 2 * r;
            if (live && i >= 0 && i < W) {
                if (!(i & 3)) {
                    cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2]; fetch((i >> 2) + 1, nxt);
                    if (r == 0) {
#pragma unroll
                        for (int c = 0; c < 3; c++)
#pragma unroll
                            for (int k = 0; k < 6; k++) elc[c][k] = eln[c][k];
                        fetch_el((i >> 2) + 1, eln);
                    }
                }
                const int b0 = 3 * (i & 3);   // byte position of the pixel inside the 12-byte group
                int v[3];
#pragma unroll
                for (int c = 0; c < 3; c++) { const int b = b0 + c; v[c] = (int)((cur[b >> 2] >> (8 * (b & 3))) & 0xff); }
                int out[3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    int em, e0, ep;
                    if (r == 0) {
                        const int k = i & 3;
                        em = k == 0 ? elc[c][0] : k == 1 ? elc[c][1] : k == 2 ? elc[c][2] : elc[c][3];
                        e0 = k == 0 ? elc[c][1] : k == 1 ? elc[c][2] : k == 2 ? elc[c][3] : elc[c][4];
                        ep = k == 0 ? elc[c][2] : k == 1 ? elc[c][3] : k == 2 ? elc[c][4] : elc[c][5];
                    } else {
                        em = i > 0 ? ring[c][r - 1][(i - 1) & 3] : 0;
                        e0 = ring[c][r - 1][i & 3];
                        ep = i + 1 < W ? ring[c][r - 1][(i + 1) & 3] : 0;
                    }
                    const int V = v[c] + ((7 * err[c] + em + 5 * e0 + 3 * ep) >> 4);
                    if (last) el[c][i] = err[c];                       // c->dither_error[c][i] = err[c]: the error of pixel i - 1
                    const int sh = c == 0 ? shr : c == 1 ? shg : shb, mx = c == 0 ? maxr : c == 1 ? maxg : maxb, q = c == 0 ? qr : c == 1 ? qg : qb;
                    const int lv = min(max(V >> sh, 0), mx);
                    err[c] = V - lv * q;
                    out[c] = lv;
                    ring[c][r][i & 3] = err[c];
                    if (last && i == W - 1) el[c][W] = err[c];         // the row's closing store (output.c:2204-2206)
                }
                const uint32_t px = (uint32_t)((out[0] << r8) + (out[1] << g8) + (out[2] << b8)) & 0xffu;
                if (dw_ok) {
                    outw = (i & 3) ? (outw | (px << (8 * (i & 3)))) : px;
                    if ((i & 3) == 3) *(uint32_t *)(drow + (i & ~3)) = outw;
                    else if (i == W - 1) { for (int k = 0; k <= (i & 3); k++) drow[(i & ~3) + k] = (uint8_t)(outw >> (8 * k)); }
                } else drow[i] = (uint8_t)px;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        __threadfence_block();
        __syncthreads();
    }
}

// Error diffusion of yuv2mono_{X,2,1}_c_template (output.c:690-700, :734-753, :792-811) over the luma words the inner context stored
// (n = the even-rounded width: the reference's pair loop runs a phantom pixel through the recurrence for an odd width).  The same
// wavefront as sws_k_ed_rgb8 with one channel: V = Y + ((7 * err + e(-1) + 5 * e(0) + 3 * e(+1) + 8 - 256) >> 4), bit = V >= 128,
// err = V - 220 * bit.  Bits go MSB-first through one running accumulator per row; a byte is stored after every eighth pixel and -
// by the X form only - the low eight bits of the accumulator once more at the end of a row that is not a multiple of 8.
__global__ void __launch_bounds__(1024) sws_k_ed_mono(const uint8_t *__restrict__ lum, int64_t lumStride, uint8_t *__restrict__ dst, int64_t dstStride,
                                                      int n, int H, int *__restrict__ errline, int white)
{
    __shared__ int ring[1024][4];
    const int r = threadIdx.x;
    for (int y0 = 0; y0 < H; y0 += 1024) {
        const int NR = min(1024, H - y0);
        const int y = y0 + r;
        const bool live = r < NR, last = r == NR - 1;
        const int16_t *srow = (const int16_t *)(lum + (int64_t)y * lumStride);
        uint8_t *drow = dst + (int64_t)y * dstStride;
        int err = 0;
        unsigned acc = 0;
        const int steps = n + 2 * (NR - 1);
        uint32_t cur[2] = { 0, 0 }, nxt[2] = { 0, 0 };   // four luma words per group, fetched one group ahead (see sws_k_ed_rgb8)
        auto fetch = [&](int g, uint32_t (&q)[2]) {
#pragma unroll
            for (int k = 0; k < 2; k++) { const int64_t off = 8 * (int64_t)g + 4 * k; q[k] = (live && off + 4 <= lumStride) ? *(const uint32_t *)((const uint8_t *)srow + off) : 0u; }
        };
        int elc[6], eln[6];                               // the error line above the group's first row, fetched the same way
        auto fetch_el = [&](int g, int (&q)[6]) {
#pragma unroll
            for (int k = 0; k < 6; k++) { const int idx = 4 * g + k; q[k] = (r == 0 && idx < n + 3) ? errline[idx] : 0; }
        };
        fetch(0, nxt);
        fetch_el(0, eln);
        for (int t = 0; t < steps; t++) {
            const int i = t - 2 * r;
            if (live && i >= 0 && i < n) {
                if (!(i & 3)) {
                    cur[0] = nxt[0]; cur[1] = nxt[1]; fetch((i >> 2) + 1, nxt);
                    if (r == 0) {
#pragma unroll
                        for (int k = 0; k < 6; k++) elc[k] = eln[k];
                        fetch_el((i >> 2) + 1, eln);
                    }
                }
                const int Yi = (int)(int16_t)(cur[(i & 3) >> 1] >> (16 * (i & 1)));
                int em, e0, ep;
                if (r == 0) {
                    const int k = i & 3;
                    em = k == 0 ? elc[0] : k == 1 ? elc[1] : k == 2 ? elc[2] : elc[3];
                    e0 = k == 0 ? elc[1] : k == 1 ? elc[2] : k == 2 ? elc[3] : elc[4];
                    ep = k == 0 ? elc[2] : k == 1 ? elc[3] : k == 2 ? elc[4] : elc[5];
                } else {
                    em = i > 0 ? ring[r - 1][(i - 1) & 3] : 0;
                    e0 = ring[r - 1][i & 3];
                    ep = i + 1 < n ? ring[r - 1][(i + 1) & 3] : 0;
                }
                const int V = Yi + ((7 * err + em + 5 * e0 + 3 * ep + 8 - 256) >> 4);
                if (last) errline[i] = err;
                const int bit = V >= 128;
                acc = 2 * acc + (unsigned)bit;
                err = V - 220 * bit;
                ring[r][i & 3] = err;
                if ((i & 7) == 7) drow[i >> 3] = (uint8_t)(white ? ~acc : acc);
                if (i == n - 1) {
                    if (last) errline[n] = err;
                    if (srow[n] == 0 && (n & 6)) drow[n >> 3] = (uint8_t)(white ? ~acc : acc);   // word n of the row: the form (X = 0, 2, 1) the writer took for it
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS only: the HBM traffic stays in flight (see sws_k_ed_rgb8)
        }
        __threadfence_block();
        __syncthreads();
    }
}

} // namespace swsk
