// Packed RGB byte shuffles and packed copies: rgbToRgbWrapper / findRgbConvFn (swscale_unscaled.c:1843-2060, with the
// C converters of rgb2rgb.c / rgb2rgb_template.c) and packedCopyWrapper (:2138-2157) for the 8-bit 24/32 bpp formats.
// Pure HBM streaming: one thread moves 4 pixels (12 or 16 bytes in, 12 or 16 bytes out) with v_perm_b32.
#pragma once
#include "kernels_common.hpp"
#include "wave_util.hpp"

namespace swsk {

struct ShufflePlan {
    uint32_t sel[4];     // v_perm_b32 selector of pixel i of a 4-pixel group: destination byte j <- byte sel[j] of the
                         // source dword pair that holds the pixel (0x0d.. = constant 0xff, 0x0c = 0)
    int32_t src_step, dst_step;         // 3 or 4 bytes per pixel
    int32_t spos[4], dpos[4];           // byte offsets of R,G,B,A in a source / destination pixel (A: -1 = none)
    int32_t opaque;                     // write 255 to destination alpha even though the source has an alpha byte
    int32_t planar_alpha;               // gbrap side: sws_k_gbrp_to_packed reads the alpha plane (1); sws_k_packed_to_gbrp writes one: 1 = from spos[3], 2 = 255
};

struct Tri { uint32_t a, b, c; };

__device__ __forceinline__ void shuffle_px_bytes(const ShufflePlan &sp, const uint8_t *s, uint8_t *d)
{
#pragma unroll
    for (int k = 0; k < 3; k++) d[sp.dpos[k]] = s[sp.spos[k]];
    if (sp.dpos[3] >= 0) d[sp.dpos[3]] = (sp.spos[3] >= 0 && !sp.opaque) ? s[sp.spos[3]] : 255;
}

template <bool S3, bool D3>
__global__ void __launch_bounds__(256) sws_k_rgb_shuffle(SwsFrameSet fs, ShufflePlan sp, int w, int sliceY)
{
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x0 >= w) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int y = blockIdx.y;
    const uint8_t *srow = f.src[0] + (int64_t)(sliceY + y) * f.srcStride[0];   // the host rebases slice pointers to absolute rows
    uint8_t *drow = f.dst[0] + (int64_t)(sliceY + y) * f.dstStride[0];
    constexpr int SS = S3 ? 3 : 4, DS = D3 ? 3 : 4;
    const uint8_t *s = srow + x0 * SS;
    uint8_t *d = drow + x0 * DS;
    const bool aligned = ((((uintptr_t)srow) & (S3 ? 3 : 15)) | (((uintptr_t)drow) & (D3 ? 3 : 15))) == 0; // wave-uniform
    if (!aligned || x0 + 4 > w) {
        const int n = min(4, w - x0);
        for (int i = 0; i < n; i++) shuffle_px_bytes(sp, s + i * SS, d + i * DS);
        return;
    }
    uint32_t px[4];
    if (S3) {
        const Tri t = *reinterpret_cast<const Tri *>(s);
        px[0] = __builtin_amdgcn_perm(t.a, t.a, sp.sel[0]);
        px[1] = __builtin_amdgcn_perm(t.b, t.a, sp.sel[1]);
        px[2] = __builtin_amdgcn_perm(t.c, t.b, sp.sel[2]);
        px[3] = __builtin_amdgcn_perm(t.c, t.c, sp.sel[3]);
    } else {
        const uint4 q = *reinterpret_cast<const uint4 *>(s);
        px[0] = __builtin_amdgcn_perm(q.x, q.x, sp.sel[0]);
        px[1] = __builtin_amdgcn_perm(q.y, q.y, sp.sel[1]);
        px[2] = __builtin_amdgcn_perm(q.z, q.z, sp.sel[2]);
        px[3] = __builtin_amdgcn_perm(q.w, q.w, sp.sel[3]);
    }
    if (D3) {
        Tri o;
        o.a = __builtin_amdgcn_perm(px[1], px[0], 0x04020100u);
        o.b = __builtin_amdgcn_perm(px[2], px[1], 0x05040201u);
        o.c = __builtin_amdgcn_perm(px[3], px[2], 0x06050402u);
        *reinterpret_cast<Tri *>(d) = o;
    } else {
        *reinterpret_cast<uint4 *>(d) = make_uint4(px[0], px[1], px[2], px[3]);
    }
}

// packedCopyWrapper: visible bytes of every row; `alpha_pos` >= 0 forces that byte of every 4-byte pixel to 255
// (the rgb0 -> rgba "scratch copy" of swscale.c:1106-1124 folded into the copy)
__global__ void __launch_bounds__(256) sws_k_packed_copy(SwsFrameSet fs, int row_bytes, int sliceY, int alpha_pos)
{
    const int b0 = (blockIdx.x * 256 + threadIdx.x) * 16;
    if (b0 >= row_bytes) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int y = blockIdx.y;
    const uint8_t *srow = f.src[0] + (int64_t)(sliceY + y) * f.srcStride[0];
    uint8_t *drow = f.dst[0] + (int64_t)(sliceY + y) * f.dstStride[0];
    const bool aligned = ((((uintptr_t)srow) | ((uintptr_t)drow)) & 15) == 0;
    if (aligned && b0 + 16 <= row_bytes) {
        uint4 q = *reinterpret_cast<const uint4 *>(srow + b0);
        if (alpha_pos >= 0) {
            const uint32_t m = 0xffu << (8 * alpha_pos);
            q.x |= m; q.y |= m; q.z |= m; q.w |= m;
        }
        *reinterpret_cast<uint4 *>(drow + b0) = q;
        return;
    }
    const int n = min(16, row_bytes - b0);
    for (int i = 0; i < n; i++) {
        const int b = b0 + i;
        drow[b] = (alpha_pos >= 0 && (b & 3) == alpha_pos) ? 255 : srow[b];
    }
}

// planarRgbToRgbWrapper (swscale_unscaled.c:1322-1378, gbr24ptopacked24/32 :1188-1233): gbrp -> packed 24/32 bpp,
// A = 255.  pl[k] = source plane of destination colour k (R,G,B); dpos[] = byte positions in the destination pixel.
template <bool D3>
__global__ void __launch_bounds__(256) sws_k_gbrp_to_packed(SwsFrameSet fs, ShufflePlan sp, int w, int sliceY)
{
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x0 >= w) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int y = blockIdx.y;
    const uint8_t *pr = f.src[2] + (int64_t)(sliceY + y) * f.srcStride[2] + x0;   // gbrp: plane 0 = G, 1 = B, 2 = R
    const uint8_t *pg = f.src[0] + (int64_t)(sliceY + y) * f.srcStride[0] + x0;
    const uint8_t *pb = f.src[1] + (int64_t)(sliceY + y) * f.srcStride[1] + x0;
    constexpr int DS = D3 ? 3 : 4;
    uint8_t *d = f.dst[0] + (int64_t)(sliceY + y) * f.dstStride[0] + x0 * DS;
    const int n = min(4, w - x0);
    for (int i = 0; i < n; i++) {
        uint8_t *q = d + i * DS;
        q[sp.dpos[0]] = pr[i]; q[sp.dpos[1]] = pg[i]; q[sp.dpos[2]] = pb[i];
        if (!D3) q[sp.dpos[3]] = sp.planar_alpha ? f.src[3][(int64_t)(sliceY + y) * f.srcStride[3] + x0 + i] : 255;   // gbraptopacked32 (:1235-1262) / gbr24ptopacked32
    }
}

// rgbToPlanarRgbWrapper (swscale_unscaled.c:1436-1490, packedtogbr24p :1404-1434): packed 24/32 bpp -> gbrp, alpha dropped
__global__ void __launch_bounds__(256) sws_k_packed_to_gbrp(SwsFrameSet fs, ShufflePlan sp, int w, int sliceY)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= w) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int y = blockIdx.y;
    const uint8_t *s = f.src[0] + (int64_t)(sliceY + y) * f.srcStride[0] + x * sp.src_step;
    f.dst[2][(int64_t)(sliceY + y) * f.dstStride[2] + x] = s[sp.spos[0]];   // R
    f.dst[0][(int64_t)(sliceY + y) * f.dstStride[0] + x] = s[sp.spos[1]];   // G
    f.dst[1][(int64_t)(sliceY + y) * f.dstStride[1] + x] = s[sp.spos[2]];   // B
    // rgbToPlanarRgbaWrapper (swscale_unscaled.c:1543-1590): packed32togbrap copies the fourth byte, packed24togbrap writes 255
    if (sp.planar_alpha) f.dst[3][(int64_t)(sliceY + y) * f.dstStride[3] + x] = sp.planar_alpha == 1 ? s[sp.spos[3]] : 255;
}

// yuv420p_gbrp_c / yuv422p_gbrp_c (yuv2rgb.c:127-135 PUTGBRP, :532, :553): the 24 bpp LUT values written to the
// G, B and R planes.  One thread = one chroma sample = 2 pixels x 2 rows.
__global__ void __launch_bounds__(256) sws_k_yuv2gbrp_unscaled(SwsFrameSet fs, SwsDevParams p, int is422, int npairs, int sliceY)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npairs) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const SwsLutParams &L = p.lut;
    const int yrow = 2 * blockIdx.y;                       // slice-relative luma row of the pair
#pragma unroll
    for (int l = 0; l < 2; l++) {
        const int yy = yrow + l;
        const int cr = is422 ? sliceY + yy : ((sliceY + yrow) >> 1);   // absolute rows: the host rebases slice pointers
        const int U = f.src[1][(int64_t)cr * f.srcStride[1] + i], V = f.src[2][(int64_t)cr * f.srcStride[2] + i];
        const ChromaIdx k = lut_chroma(L, U, V);
        const uint8_t *py = f.src[0] + (int64_t)(sliceY + yy) * f.srcStride[0] + 2 * i;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int Y = py[h];
            f.dst[0][(int64_t)(sliceY + yy) * f.dstStride[0] + 2 * i + h] = (uint8_t)lut_luma(L, k.g + Y);
            f.dst[1][(int64_t)(sliceY + yy) * f.dstStride[1] + 2 * i + h] = (uint8_t)lut_luma(L, k.b + Y);
            f.dst[2][(int64_t)(sliceY + yy) * f.dstStride[2] + 2 * i + h] = (uint8_t)lut_luma(L, k.r + Y);
        }
    }
}

// yuv422pToYuy2/UyvyWrapper, planarToYuy2/UyvyWrapper (swscale_unscaled.c:376-422) -> yuvPlanartoyuy2_c / yuvPlanartouyvy_c
// (rgb2rgb_template.c:379-470): thread = pixel pair; vlpc = luma rows per chroma row (1 or 2), counted from the slice start.
__global__ void __launch_bounds__(256) sws_k_planar_to_p422(SwsFrameSet fs, SwsDevParams p, int npairs, int sliceY, int vlpc)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npairs) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int y = blockIdx.y;                                   // slice-relative row
    const int cr = (vlpc == 2 ? (sliceY >> 1) : sliceY) + y / vlpc;   // absolute chroma row (host rebases slice pointers)
    const uint8_t *ys = f.src[0] + (int64_t)(sliceY + y) * f.srcStride[0] + 2 * i;
    uint8_t *d = f.dst[0] + (int64_t)(sliceY + y) * f.dstStride[0] + 4 * i;
    d[p.d422_y] = ys[0]; d[p.d422_y + 2] = ys[1];
    d[p.d422_u] = f.src[1][(int64_t)cr * f.srcStride[1] + i];
    d[p.d422_v] = f.src[2][(int64_t)cr * f.srcStride[2] + i];
}

// yuyv/uyvy ToYuv420/422Wrapper (swscale_unscaled.c:424-484) -> yuyvtoyuv420_c .. uyvytoyuv422_c (rgb2rgb_template.c:751-825).
// grid.y = luma rows of the slice; thread = pixel pair: two luma samples, and the chroma sample of the row (4:2:2) or the
// truncating mean of rows (y-1, y) on odd slice rows (4:2:0).
__global__ void __launch_bounds__(256) sws_k_p422_to_planar(SwsFrameSet fs, SwsDevParams p, int w, int sliceY, int to420)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int cw = (w + 1) >> 1;
    if (i >= cw) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int y = blockIdx.y;
    const uint8_t *s = f.src[0] + (int64_t)(sliceY + y) * f.srcStride[0] + 4 * i;
    uint8_t *yd = f.dst[0] + (int64_t)(sliceY + y) * f.dstStride[0] + 2 * i;
    yd[0] = s[p.s422_y];
    if (2 * i + 1 < w) yd[1] = s[p.s422_y + 2];
    if (!to420) {
        f.dst[1][(int64_t)(sliceY + y) * f.dstStride[1] + i] = s[p.s422_u];
        f.dst[2][(int64_t)(sliceY + y) * f.dstStride[2] + i] = s[p.s422_v];
    } else if (y & 1) {
        const uint8_t *s0 = s - f.srcStride[0];
        const int cr = (sliceY >> 1) + (y >> 1);
        f.dst[1][(int64_t)cr * f.dstStride[1] + i] = (uint8_t)((s0[p.s422_u] + s[p.s422_u]) >> 1);
        f.dst[2][(int64_t)cr * f.dstStride[2] + i] = (uint8_t)((s0[p.s422_v] + s[p.s422_v]) >> 1);
    }
}

// ff_sws_alphablendaway (alphablend.c:23-175): the source's alpha channel blends the picture over a uniform (black) or 32-pixel
// checkerboard background into the alpha-less twin of the source format.  Thread = one sample of plane `plane` (planar: chroma planes take the
// mean of the 2 or 2x2 alpha samples over them) or one pixel (packed).  Rows are absolute picture rows.
struct AlphaBlendPlan { int planar, plane_count, depth, alpha_pos, lum_w, lum_h, lw, lh, target[2][3]; };
__global__ void __launch_bounds__(256) sws_k_alphablend(SwsFrameSet fs, AlphaBlendPlan ap, int plane, int w, int y0)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= w) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int y = y0 + blockIdx.y;
    const int depth = ap.depth;
    const bool wide = depth >= 9;
    const unsigned off = 1u << (depth - 1), shift = depth, maxv = (1u << shift) - 1;
    if (ap.planar) {
        const int xs = plane ? ap.lw : 0, ys = plane ? ap.lh : 0, pc = ap.plane_count;
        const bool subsample_row = ys && (y << ys) + 1 < ap.lum_h;
        const uint8_t *sp = pick4(f.src, plane) + (int64_t)pick4(f.srcStride, plane) * y;
        uint8_t *dp = pick4(f.dst, plane) + (int64_t)pick4(f.dstStride, plane) * y;
        const uint8_t *ap0 = pick4(f.src, pc) + (int64_t)pick4(f.srcStride, pc) * (y << ys), *ap1 = ap0 + pick4(f.srcStride, pc);
        const unsigned t = (unsigned)ap.target[((x ^ y) >> 5) & 1][plane];
        unsigned alpha;
        if (xs || subsample_row) {
            const int xn = min(2 * x + 1, ap.lum_w - 1);
            if (wide) {
                const uint16_t *a = (const uint16_t *)ap0, *a2 = (const uint16_t *)ap1;
                alpha = subsample_row ? (unsigned)(a[2 * x] + a[xn] + 2 + a2[2 * x] + a2[xn]) >> 2 : (unsigned)(a[2 * x] + a[xn]) >> 1;
            } else
                alpha = subsample_row ? (unsigned)(ap0[2 * x] + ap0[xn] + 2 + ap1[2 * x] + ap1[xn]) >> 2 : (unsigned)(ap0[2 * x] + ap0[xn]) >> 1;
        } else alpha = wide ? ((const uint16_t *)ap0)[x] : ap0[x];
        if (wide) {
            unsigned u = ((const uint16_t *)sp)[x] * alpha + t * (maxv - alpha) + off;
            u = (u + (u >> shift)) >> shift;
            ((uint16_t *)dp)[x] = (uint16_t)min(u, maxv);
        } else {
            const unsigned u = sp[x] * alpha + t * (255u - alpha) + 128u;
            dp[x] = (uint8_t)((257u * u) >> 16);
        }
    } else {
        const int pc = ap.plane_count, xi = (pc + 1) * x;
        const uint8_t *row = f.src[0] + (int64_t)f.srcStride[0] * y;
        uint8_t *drow = f.dst[0] + (int64_t)f.dstStride[0] * y;
        if (wide) {
            const uint16_t *s = (const uint16_t *)(row + 2 * !ap.alpha_pos), *a = (const uint16_t *)(row + ap.alpha_pos);
            uint16_t *d = (uint16_t *)drow;
            const unsigned alpha = a[xi];
            for (int pl = 0; pl < pc; pl++) {
                unsigned u = s[xi + pl] * alpha + (unsigned)ap.target[((x ^ y) >> 5) & 1][pl] * (maxv - alpha) + off;
                u = (u + (u >> shift)) >> shift;
                d[pc * x + pl] = (uint16_t)min(u, maxv);
            }
        } else {
            const uint8_t *s = row + !ap.alpha_pos, *a = row + ap.alpha_pos;
            const unsigned alpha = a[xi];
            for (int pl = 0; pl < pc; pl++) {
                const unsigned u = s[xi + pl] * alpha + (unsigned)ap.target[((x ^ y) >> 5) & 1][pl] * (255u - alpha) + 128u;
                drow[pc * x + pl] = (uint8_t)((257u * u) >> 16);
            }
        }
    }
}

// 16-bit packed RGB converters; thread = pixel.
//   mode 0: rgb48tobgr48 / rgb48to64 / rgb48tobgr64 / rgb64to48 / rgb64tobgr48 (rgb2rgb.c:322-413): word moves, A = 0xFFFF
//   mode 1: Rgb16ToPlanarRgb16Wrapper / packed16togbra16 (swscale_unscaled.c:685-962): plane[x] = word >> (16 - depth)
//   mode 2: planarRgb16ToRgb16Wrapper / gbr16ptopacked16 (:964-1186): word = c << (16 - bpp) | c >> ((bpp - 8) * 2), A = 0xFFFF
struct Rgb16Plan { int mode, sstep, dstep, spos[3], dpos[3], depth, src_alpha, dst_alpha; };   // src_alpha / dst_alpha: the planar side has / gets an alpha plane, the packed side a fourth word   // positions in 16-bit words (packed) or plane index (planar)
__global__ void __launch_bounds__(256) sws_k_rgb16_convert(SwsFrameSet fs, Rgb16Plan rp, int w, int sliceY)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= w) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int y = sliceY + blockIdx.y;                          // absolute row (the host rebases slice pointers)
    uint16_t v[3];
    if (rp.mode == 2) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int pl = rp.spos[k];
            const uint8_t *sp = pl == 0 ? f.src[0] : pl == 1 ? f.src[1] : f.src[2];
            const int ss = pl == 0 ? f.srcStride[0] : pl == 1 ? f.srcStride[1] : f.srcStride[2];
            const uint16_t c = ((const uint16_t *)(sp + (int64_t)y * ss))[x];
            v[k] = (uint16_t)(c << (16 - rp.depth) | c >> ((rp.depth - 8) * 2));
        }
    } else {
        const uint16_t *s = (const uint16_t *)(f.src[0] + (int64_t)y * f.srcStride[0]) + rp.sstep * x;
#pragma unroll
        for (int k = 0; k < 3; k++) v[k] = s[rp.spos[k]];
    }
    if (rp.mode == 1) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int pl = rp.dpos[k];
            uint8_t *dp = pl == 0 ? f.dst[0] : pl == 1 ? f.dst[1] : f.dst[2];
            const int dst = pl == 0 ? f.dstStride[0] : pl == 1 ? f.dstStride[1] : f.dstStride[2];
            ((uint16_t *)(dp + (int64_t)y * dst))[x] = (uint16_t)(v[k] >> (16 - rp.depth));
        }
        if (rp.dst_alpha) {   // packed16togbra16: the source's alpha word, or 0xFFFF, >> shift
            const unsigned a = rp.src_alpha ? ((const uint16_t *)(f.src[0] + (int64_t)y * f.srcStride[0]))[rp.sstep * x + 3] : 0xFFFFu;
            ((uint16_t *)(f.dst[3] + (int64_t)y * f.dstStride[3]))[x] = (uint16_t)(a >> (16 - rp.depth));
        }
    } else {
        uint16_t *d = (uint16_t *)(f.dst[0] + (int64_t)y * f.dstStride[0]) + rp.dstep * x;
#pragma unroll
        for (int k = 0; k < 3; k++) d[rp.dpos[k]] = v[k];
        if (rp.dstep == 4) {
            if (rp.mode == 2 && rp.src_alpha) {   // gbr16ptopacked16: the alpha plane scaled like the colours
                const uint16_t c = ((const uint16_t *)(f.src[3] + (int64_t)y * f.srcStride[3]))[x];
                d[3] = (uint16_t)(c << (16 - rp.depth) | c >> ((rp.depth - 8) * 2));
            } else d[3] = 0xFFFF;
        }
    }
}

// uint_y_to_float_y_wrapper (swscale_unscaled.c:2095-2113; uint2float_lut[i] = (float)i * (1 / 255), utils.c:1552-1556) when to_float,
// float_y_to_uint_y_wrapper (:2115-2135) otherwise
__global__ void __launch_bounds__(256) sws_k_gray_f32(SwsFrameSet fs, int w, int sliceY, int to_float)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= w) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int y = sliceY + blockIdx.y;
    if (to_float) ((float *)(f.dst[0] + (int64_t)y * f.dstStride[0]))[x] = (float)f.src[0][(int64_t)y * f.srcStride[0] + x] * (1.0f / 255.0f);
    else {
        const int v = (int)lrintf(255.0f * ((const float *)(f.src[0] + (int64_t)y * f.srcStride[0]))[x]);
        f.dst[0][(int64_t)y * f.dstStride[0] + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

// yuv2rgb_c_1_ordered_dither (yuv2rgb.c:457-517): chroma ignored (g = the table index of U = V = 128); the 1-bit table is
// (y_table[k] >> 7) with the ramp starting at element 110 (yuv2rgb.c:806-816).  One thread = one output byte of a row pair.  The tail
// (dst_w & 7) counts PIXEL PAIRS across both rows in the macro's order and shifts the rest.
__global__ void __launch_bounds__(256) sws_k_yuv2mono_unscaled(SwsFrameSet fs, SwsDevParams p, int gidx, int sliceY)
{
    const int bx = blockIdx.x * 256 + threadIdx.x;
    const int nfull = p.dstW >> 3;
    if (bx > nfull || (bx == nfull && !(p.dstW & 7))) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int y = 2 * blockIdx.y, yd = y + sliceY;                    // slice-relative pair, absolute rows (the host rebases slice pointers)
    const uint8_t *py0 = f.src[0] + (int64_t)yd * f.srcStride[0] + 8 * bx, *py1 = py0 + f.srcStride[0];
    uint8_t *o0 = f.dst[0] + (int64_t)yd * f.dstStride[0] + bx, *o1 = o0 + f.dstStride[0];
    const uint32_t drow_lo[9] = { 0x679e3e75u, 0xba15c722u, 0x4c835990u, 0xce29a500u, 0x6097376eu, 0xb30ec11cu, 0x457c538au, 0xd530ac07u, 0x679e3e75u };
    const uint32_t drow_hi[9] = { 0x649b3a71u, 0xb611c41fu, 0x487f568du, 0xd934af0au, 0x6ba24178u, 0xbd18cb26u, 0x4f865d94u, 0xd22da803u, 0x649b3a71u };
    auto dth = [&](int l, int k) { const uint32_t w = (k & 4) ? drow_hi[(yd & 7) + l] : drow_lo[(yd & 7) + l]; return (int)(w >> (8 * (k & 3))) & 0xff; };
    auto bit = [&](int Y, int d) { const int idx = gidx + Y + d - 110; return idx < 0 ? 0u : (unsigned)(lut_luma(p.lut, idx) >> 7); };
    unsigned a0 = 0, a1 = 0;
    if (bx < nfull) {
        for (int k = 0; k < 8; k++) { a0 = a0 + a0 + bit(py0[k], dth(0, k)); a1 = a1 + a1 + bit(py1[k], dth(1, k)); }
    } else {
        int left = p.dstW & 7;
        const int order[8] = { 0x00, 0x10, 0x11, 0x01, 0x02, 0x12, 0x13, 0x03 };   // (row << 4) | pair
        for (int s = 0; s < 8; s++) {
            const int l = order[s] >> 4, i = order[s] & 15;
            unsigned &a = l ? a1 : a0;
            const uint8_t *py = l ? py1 : py0;
            if (left) { a = a + a + bit(py[2 * i], dth(l, 2 * i)); a = a + a + bit(py[2 * i + 1], dth(l, 2 * i + 1)); left--; }
            else a <<= 2;
        }
    }
    *o0 = (uint8_t)a0; *o1 = (uint8_t)a1;
}

// xyz12Torgb48_c (swscale.c:745-802) when to_rgb, rgb48Toxyz12_c (:804-861) otherwise: gamma LUT in, Q12 matrix, clip, gamma LUT out,
// 12-bit result in the high bits; little-endian words, in place allowed.  One thread = one pixel.
__global__ void __launch_bounds__(256) sws_k_xyz12(const uint8_t *src, int64_t sstride, uint8_t *dst, int64_t dstride, int w,
                                                   const uint16_t *gin, const uint16_t *gout, int to_rgb)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= w) return;
    const uint16_t *s = (const uint16_t *)(src + (int64_t)blockIdx.y * sstride) + 3 * x;
    uint16_t *d = (uint16_t *)(dst + (int64_t)blockIdx.y * dstride) + 3 * x;
    const int a = gin[s[0] >> 4], b = gin[s[1] >> 4], e = gin[s[2] >> 4];
    int o0, o1, o2;
    if (to_rgb) {
        o0 = (13270 * a - 6295 * b - 2041 * e) >> 12; o1 = (-3969 * a + 7682 * b + 170 * e) >> 12; o2 = (228 * a - 835 * b + 4329 * e) >> 12;
    } else {
        o0 = (1689 * a + 1464 * b + 739 * e) >> 12; o1 = (871 * a + 2929 * b + 296 * e) >> 12; o2 = (79 * a + 488 * b + 3891 * e) >> 12;
    }
    o0 = min(max(o0, 0), 65535); o1 = min(max(o1, 0), 65535); o2 = min(max(o2, 0), 65535);
    d[0] = (uint16_t)(gout[o0] << 4); d[1] = (uint16_t)(gout[o1] << 4); d[2] = (uint16_t)(gout[o2] << 4);
}

// x2rgb10le / x2bgr10le special converters: mode 0 = x2rgb10to48 / to64 / tobgr48 / tobgr64 (rgb2rgb.c:415-471), mode 1 =
// packed30togbra10 (swscale_unscaled.c:819-889), mode 2 = gbr16ptopacked30 (:1076-1103).  pos[] = R, G, B word offsets (mode 0)
// or plane indices (modes 1, 2); hi / lo = bit replication shifts, shift = the planar format's sample shift
struct Rgb30Plan { int mode, x2rgb, dstep, pos[3], hi, lo, shift, dst_alpha; };   // dst_alpha (mode 1): the alpha plane's value, 0 = no plane
__global__ void __launch_bounds__(256) sws_k_rgb30_convert(SwsFrameSet fs, Rgb30Plan rp, int w, int sliceY)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= w) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int y = sliceY + blockIdx.y;
    if (rp.mode == 2) {
        uint32_t v[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int pl = rp.pos[k];
            const uint8_t *sp = pl == 0 ? f.src[0] : pl == 1 ? f.src[1] : f.src[2];
            const int ss = pl == 0 ? f.srcStride[0] : pl == 1 ? f.srcStride[1] : f.srcStride[2];
            v[k] = (uint32_t)((const uint16_t *)(sp + (int64_t)y * ss))[x] >> rp.shift;
        }
        ((uint32_t *)(f.dst[0] + (int64_t)y * f.dstStride[0]))[x] =
            rp.x2rgb ? (3u << 30) + (v[0] << 20) + (v[1] << 10) + v[2] : (3u << 30) + (v[2] << 20) + (v[1] << 10) + v[0];
        return;
    }
    const uint32_t px = ((const uint32_t *)(f.src[0] + (int64_t)y * f.srcStride[0]))[x];
    uint32_t v[3];
    v[1] = (px >> 10) & 0x3FF;
    v[0] = rp.x2rgb ? (px >> 20) & 0x3FF : px & 0x3FF;
    v[2] = rp.x2rgb ? px & 0x3FF : (px >> 20) & 0x3FF;
    if (rp.mode == 0) {
        uint16_t *d = (uint16_t *)(f.dst[0] + (int64_t)y * f.dstStride[0]) + rp.dstep * x;
#pragma unroll
        for (int k = 0; k < 3; k++) d[rp.pos[k]] = (uint16_t)(v[k] << 6 | v[k] >> 4);
        if (rp.dstep == 4) d[3] = 0xFFFF;
    } else {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int pl = rp.pos[k];
            uint8_t *dp = pl == 0 ? f.dst[0] : pl == 1 ? f.dst[1] : f.dst[2];
            const int dst = pl == 0 ? f.dstStride[0] : pl == 1 ? f.dstStride[1] : f.dstStride[2];
            ((uint16_t *)(dp + (int64_t)y * dst))[x] = (uint16_t)((v[k] << rp.hi | v[k] >> rp.lo) << rp.shift);
        }
        if (rp.dst_alpha) ((uint16_t *)(f.dst[3] + (int64_t)y * f.dstStride[3]))[x] = (uint16_t)rp.dst_alpha;   // alpha_val = (1 << bpc) - 1, << shift
    }
}

// yuv2rgb_c_48 / yuv2rgb_c_bgr48 (PUTRGB48 / PUTBGR48, yuv2rgb.c:107-125): the 8-bit LUT value fills both bytes of the
// 16-bit component.  One thread = one chroma sample = 2 pixels x 2 rows.
__global__ void __launch_bounds__(256) sws_k_yuv2rgb48_unscaled(SwsFrameSet fs, SwsDevParams p, int is422, int npairs, int sliceY)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npairs) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const SwsLutParams &L = p.lut;
    const int yrow = 2 * blockIdx.y;
#pragma unroll
    for (int l = 0; l < 2; l++) {
        const int yy = yrow + l;
        const int cr = is422 ? sliceY + yy : ((sliceY + yrow) >> 1);
        const int U = f.src[1][(int64_t)cr * f.srcStride[1] + i], V = f.src[2][(int64_t)cr * f.srcStride[2] + i];
        const ChromaIdx k = lut_chroma(L, U, V);
        const uint8_t *py = f.src[0] + (int64_t)(sliceY + yy) * f.srcStride[0] + 2 * i;
        uint16_t *d = (uint16_t *)(f.dst[0] + (int64_t)(sliceY + yy) * f.dstStride[0]) + 6 * i;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int Y = py[h];
            d[3 * h + p.d16_r] = (uint16_t)(lut_luma(L, k.r + Y) * 257);
            d[3 * h + p.d16_g] = (uint16_t)(lut_luma(L, k.g + Y) * 257);
            d[3 * h + p.d16_b] = (uint16_t)(lut_luma(L, k.b + Y) * 257);
        }
    }
}

// yuv2rgb_c_16/15/12_ordered_dither and yuv422p_bgr16/15/12 (YUV420FUNC_DITHER / YUV422FUNC_DITHER + PUTRGB16/15/12,
// yuv2rgb.c:283-330, :371-411).  One thread = one chroma sample = 2 pixels x 2 rows.  The dither row is selected by the loop's
// slice-relative even row (always table row 0 / 1 for the 2x2 tables, row (y & 3) (+1 for the second line) of the 4x4 table).
__global__ void __launch_bounds__(256) sws_k_yuv2rgb16_unscaled(SwsFrameSet fs, SwsDevParams p, int is422, int npairs, int sliceY)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npairs) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const SwsLutParams &L = p.lut;
    const int yrow = 2 * blockIdx.y, bpp = L.bpp16, o = 2 * (i & 3);
#pragma unroll
    for (int l = 0; l < 2; l++) {
        const int yy = yrow + l;
        const int cr = is422 ? sliceY + yy : ((sliceY + yrow) >> 1);
        const int U = f.src[1][(int64_t)cr * f.srcStride[1] + i], V = f.src[2][(int64_t)cr * f.srcStride[2] + i];
        const ChromaIdx k = lut_chroma(L, U, V);
        const uint8_t *py = f.src[0] + (int64_t)(sliceY + yy) * f.srcStride[0] + 2 * i;
        uint16_t *d = (uint16_t *)(f.dst[0] + (int64_t)(sliceY + yy) * f.dstStride[0]) + 2 * i;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int Y = py[e];
            int dr, dg, db;
            if (bpp == 16) { dr = dither_2x2_8(l, o + e); dg = dither_2x2_4(l, o + e); db = dither_2x2_8(1 + l, o + e); }
            else if (bpp == 15) { dr = dither_2x2_8(l, o + e); dg = dither_2x2_8(l, o + (e ^ 1)); db = dither_2x2_8(1 + l, o + e); }
            else { dr = dg = db = dither_4x4_16((yrow & 3) + l, o + e); }
            d[e] = (uint16_t)lut_rgb16(L, k.r + Y + dr, k.g + Y + dg, k.b + Y + db);
        }
    }
}

// bayer_to_rgb24_wrapper / bayer_to_rgb48_wrapper / bayer_to_yv12_wrapper (swscale_unscaled.c:1652-1806) over bayer_template.c.  One thread =
// one 2x2 block.  Block rows: the first one and the last one are "copied" (nearest samples of the block itself), the others "interpolated"
// from the 4x4 neighbourhood except for their first and last block; an odd height ends with a copy that runs upwards from the last row
// (negative strides in the reference) and owns the row above it too, which the block row before it therefore leaves alone.  The template's
// R() / B() names are byte positions (rpos = 0 for bggr / gbrg, 2 for rggb / grbg).  sh = BAYER_SHIFT (8 for 16-bit samples into the 8-bit
// destinations).  mode: 0 rgb24, 1 rgb48, 2 yuv420p (rgb24toyv12_2x2: ff_rgb24toyv12_c reads byte 0 as B and the wrapper swaps the chroma
// pointers).  A column beyond the picture (odd widths) is read from the row's padding like the reference does, where the stride holds it.
__global__ void __launch_bounds__(256) sws_k_bayer(SwsFrameSet fs, SwsDevParams p, int W, int sliceY, int H, int quad, int rpos, int sz, int sh, int mode)
{
    const int bx = blockIdx.x * 256 + threadIdx.x, x0 = 2 * bx, r = blockIdx.y;
    if (x0 >= W) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const bool flipped = (H & 1) && 2 * r == H - 1;
    const int row0 = 2 * r, dir = flipped ? -1 : 1;
    const bool interp = !flipped && row0 >= 2 && row0 < H - 2 && x0 >= 2 && x0 < W - 2;
    const bool skip_second = !flipped && (H & 1) && row0 + 1 == H - 2;   // that row belongs to the upward copy
    const int64_t ss = f.srcStride[0];
    const int avail = (int)(ss < 0 ? -ss : ss);
    const uint8_t *base = f.src[0] + (int64_t)(sliceY + row0) * ss;
    auto T = [&](int y, int x) -> unsigned {
        const int col = x0 + x;
        if ((col + 1) * sz > avail) return 0u;
        const uint8_t *q = base + (int64_t)(dir * y) * ss + (int64_t)sz * col;
        return sz == 1 ? (unsigned)q[0] : (unsigned)*(const uint16_t *)q;
    };
    unsigned v[2][2][3];
    if (quad) {
        if (!interp) {
            v[0][0][0] = v[0][1][0] = v[1][1][0] = v[1][0][0] = T(1, 1) >> sh;
            v[0][1][1] = T(0, 1) >> sh; v[0][0][1] = v[1][1][1] = (T(0, 1) + T(1, 0)) >> (1 + sh); v[1][0][1] = T(1, 0) >> sh;
            v[1][1][2] = v[0][0][2] = v[0][1][2] = v[1][0][2] = T(0, 0) >> sh;
        } else {
            v[0][0][0] = (T(-1, -1) + T(-1, 1) + T(1, -1) + T(1, 1)) >> (2 + sh); v[0][0][1] = (T(-1, 0) + T(0, -1) + T(0, 1) + T(1, 0)) >> (2 + sh); v[0][0][2] = T(0, 0) >> sh;
            v[0][1][0] = (T(-1, 1) + T(1, 1)) >> (1 + sh); v[0][1][1] = T(0, 1) >> sh; v[0][1][2] = (T(0, 0) + T(0, 2)) >> (1 + sh);
            v[1][0][0] = (T(1, -1) + T(1, 1)) >> (1 + sh); v[1][0][1] = T(1, 0) >> sh; v[1][0][2] = (T(0, 0) + T(2, 0)) >> (1 + sh);
            v[1][1][0] = T(1, 1) >> sh; v[1][1][1] = (T(0, 1) + T(1, 0) + T(1, 2) + T(2, 1)) >> (2 + sh); v[1][1][2] = (T(0, 0) + T(0, 2) + T(2, 0) + T(2, 2)) >> (2 + sh);
        }
    } else {
        if (!interp) {
            v[0][0][0] = v[0][1][0] = v[1][1][0] = v[1][0][0] = T(1, 0) >> sh;
            v[0][0][1] = T(0, 0) >> sh; v[1][1][1] = T(1, 1) >> sh; v[0][1][1] = v[1][0][1] = (T(0, 0) + T(1, 1)) >> (1 + sh);
            v[1][1][2] = v[0][0][2] = v[0][1][2] = v[1][0][2] = T(0, 1) >> sh;
        } else {
            v[0][0][0] = (T(-1, 0) + T(1, 0)) >> (1 + sh); v[0][0][1] = T(0, 0) >> sh; v[0][0][2] = (T(0, -1) + T(0, 1)) >> (1 + sh);
            v[0][1][0] = (T(-1, 0) + T(-1, 2) + T(1, 0) + T(1, 2)) >> (2 + sh); v[0][1][1] = (T(-1, 1) + T(0, 0) + T(0, 2) + T(1, 1)) >> (2 + sh); v[0][1][2] = T(0, 1) >> sh;
            v[1][0][0] = T(1, 0) >> sh; v[1][0][1] = (T(0, 0) + T(1, -1) + T(1, 1) + T(2, 0)) >> (2 + sh); v[1][0][2] = (T(0, -1) + T(0, 1) + T(2, -1) + T(2, 1)) >> (2 + sh);
            v[1][1][0] = (T(1, 0) + T(1, 2)) >> (1 + sh); v[1][1][1] = T(1, 1) >> sh; v[1][1][2] = (T(0, 1) + T(2, 1)) >> (1 + sh);
        }
    }
    if (mode == 2) {
        const int32_t *t = p.rgb2yuv;
        unsigned sb = 0, sg = 0, sr = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned *px = v[k >> 1][k & 1];
            const unsigned byte0 = (rpos == 0 ? px[0] : px[2]) & 0xff, byte1 = px[1] & 0xff, byte2 = (rpos == 0 ? px[2] : px[0]) & 0xff;   // block bytes in memory order
            sb += byte0; sg += byte1; sr += byte2;                                                                                      // ff_rgb24toyv12_c: b = byte 0, r = byte 2
            if (x0 + (k & 1) < W && !((k >> 1) && skip_second))
                f.dst[0][(int64_t)(sliceY + row0 + dir * (k >> 1)) * f.dstStride[0] + x0 + (k & 1)] =
                    (uint8_t)((((unsigned)t[0] * byte2 + (unsigned)t[1] * byte1 + (unsigned)t[2] * byte0) >> 15) + 16);
        }
        const unsigned bxm = sb >> 2, gxm = sg >> 2, rxm = sr >> 2;
        const int64_t crow = (sliceY + row0) >> 1;
        f.dst[2][crow * f.dstStride[2] + bx] = (uint8_t)((((unsigned)t[3] * rxm + (unsigned)t[4] * gxm + (unsigned)t[5] * bxm) >> 15) + 128);
        f.dst[1][crow * f.dstStride[1] + bx] = (uint8_t)((((unsigned)t[6] * rxm + (unsigned)t[7] * gxm + (unsigned)t[8] * bxm) >> 15) + 128);
        return;
    }
#pragma unroll
    for (int py = 0; py < 2; py++) {
        if (py && skip_second) continue;
        uint8_t *o = f.dst[0] + (int64_t)(sliceY + row0 + dir * py) * f.dstStride[0];
#pragma unroll
        for (int px = 0; px < 2; px++) {
            if (x0 + px >= W) continue;
            if (mode == 1) {
                uint16_t *o16 = (uint16_t *)o + 3 * (x0 + px);
                o16[rpos] = (uint16_t)v[py][px][0]; o16[1] = (uint16_t)v[py][px][1]; o16[2 - rpos] = (uint16_t)v[py][px][2];
            } else {
                uint8_t *o8 = o + 3 * (x0 + px);
                o8[rpos] = (uint8_t)v[py][px][0]; o8[1] = (uint8_t)v[py][px][1]; o8[2 - rpos] = (uint8_t)v[py][px][2];
            }
        }
    }
}

// ff_update_palette (swscale.c:873-951): one workgroup per frame, one thread per palette entry.  pal8 reads the caller's 0xAARRGGBB
// words (src[2]), the 8 / 4 bpp RGB formats expand their bit fields (an index beyond a 4-bit format's 16 values spills a channel
// into its neighbours, like the reference's plain sums).  Writes pal_yuv[256] then pal_rgb[256] (the word whose bytes are the
// destination's, little-endian rows of :916-949) at src[1].
__global__ void __launch_bounds__(256) sws_k_update_palette(SwsFrameSet fs, SwsDevParams p, int srcFormat, int dstFormat)
{
    const int i = threadIdx.x;
    const FrameRegs f = load_frame(fs, blockIdx.x);
    uint32_t *tab = (uint32_t *)f.src[1];
    int r, g, b, a = 0xff;
    if (srcFormat == AV_PIX_FMT_PAL8) { const uint32_t e = ((const uint32_t *)f.src[2])[i]; a = (e >> 24) & 0xFF; r = (e >> 16) & 0xFF; g = (e >> 8) & 0xFF; b = e & 0xFF; }
    else if (srcFormat == AV_PIX_FMT_RGB8) { r = (i >> 5) * 36; g = ((i >> 2) & 7) * 36; b = (i & 3) * 85; }
    else if (srcFormat == AV_PIX_FMT_BGR8) { b = (i >> 6) * 85; g = ((i >> 3) & 7) * 36; r = (i & 7) * 36; }
    else if (srcFormat == AV_PIX_FMT_RGB4_BYTE) { r = (i >> 3) * 255; g = ((i >> 1) & 3) * 85; b = (i & 1) * 255; }
    else { b = (i >> 3) * 255; g = ((i >> 1) & 3) * 85; r = (i & 1) * 255; }
    const int32_t *t = p.rgb2yuv;
    // (clip_u8_shr, not clip_u8(x >> 15): packing three of those into one word is the pattern hipcc 7.2 turns into v_ashr_pk_u8_i32 with
    // stale upper bits, see kernels_common.hpp)
    const int y = clip_u8_shr(t[0] * r + t[1] * g + t[2] * b + (33 << 14), 15);
    const int u = clip_u8_shr(t[3] * r + t[4] * g + t[5] * b + (257 << 14), 15);
    const int v = clip_u8_shr(t[6] * r + t[7] * g + t[8] * b + (257 << 14), 15);
    tab[i] = (uint32_t)y + ((uint32_t)u << 8) + ((uint32_t)v << 16) + ((uint32_t)a << 24);
    uint32_t w;
    switch (dstFormat) {
    case AV_PIX_FMT_RGBA: case AV_PIX_FMT_RGB24: w = (uint32_t)(r + (g << 8) + (b << 16)) + ((uint32_t)a << 24); break;   // BGR32, RGB24
    case AV_PIX_FMT_ARGB: w = (uint32_t)(a + (r << 8) + (g << 16)) + ((uint32_t)b << 24); break;                          // BGR32_1
    case AV_PIX_FMT_ABGR: w = (uint32_t)(a + (b << 8) + (g << 16)) + ((uint32_t)r << 24); break;                          // RGB32_1
    case AV_PIX_FMT_GBRP: case AV_PIX_FMT_GBRAP: w = (uint32_t)(g + (b << 8) + (r << 16)) + ((uint32_t)a << 24); break;
    default: w = (uint32_t)(b + (g << 8) + (r << 16)) + ((uint32_t)a << 24); break;                                       // RGB32, BGR24, ...
    }
    tab[256 + i] = w;
}

// palToRgbWrapper with sws_convertPalette8ToPacked32 / 24 (swscale_unscaled.c:600-644, :2707-2730) and palToGbrpWrapper with pal8ToPlanar8
// (:531-545, :646-683): the bytes of the pal_rgb word are the destination's bytes.  nbytes = 3 / 4 (packed) or the number of planes.
// gray: a gray8 source (usePal names it too): its palette is the grey ramp r = g = b = i, a = 255 (swscale.c:901-902), so the word is built
// from the sample itself; 1 = alpha in the last byte, 2 = in the first (argb / abgr)
__global__ void __launch_bounds__(256) sws_k_pal2rgb(SwsFrameSet fs, int w, int sliceY, int nbytes, int planar, int gray)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= w) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int y = sliceY + blockIdx.y;
    const uint32_t idx = f.src[0][(int64_t)y * f.srcStride[0] + x];
    const uint32_t e = gray == 2 ? (idx * 0x01010100u | 0xFFu) : gray ? (idx * 0x00010101u | 0xFF000000u) : ((const uint32_t *)f.src[1])[256 + idx];
    if (planar) {   // (static plane indices: a run-time index into the frame descriptor would put it into scratch memory)
        f.dst[0][(int64_t)y * f.dstStride[0] + x] = (uint8_t)e;
        f.dst[1][(int64_t)y * f.dstStride[1] + x] = (uint8_t)(e >> 8);
        f.dst[2][(int64_t)y * f.dstStride[2] + x] = (uint8_t)(e >> 16);
        if (nbytes == 4) f.dst[3][(int64_t)y * f.dstStride[3] + x] = (uint8_t)(e >> 24);
    } else if (nbytes == 4) {
        ((uint32_t *)(f.dst[0] + (int64_t)y * f.dstStride[0]))[x] = e;
    } else {
        uint8_t *d = f.dst[0] + (int64_t)y * f.dstStride[0] + 3 * x;
        d[0] = (uint8_t)e; d[1] = (uint8_t)(e >> 8); d[2] = (uint8_t)(e >> 16);
    }
}

// yuv2rgb_c_8 / 4 / 4b_ordered_dither and yuv422p_bgr8 / bgr4 / bgr4_byte (YUV420FUNC_DITHER / YUV422FUNC_DITHER + PUTRGB8 / PUTRGB4D /
// PUTRGB4DB, yuv2rgb.c:283-369, :413-455).  One thread = one chroma sample = 2 pixels x 2 rows.  LOADDITHER8 / 4D / 4DB select the rows
// of the 8x8 tables by the ABSOLUTE even row (yd & 7), the second line of the pair reads the following row; the column is the pixel's.
__global__ void __launch_bounds__(256) sws_k_yuv2rgb8_unscaled(SwsFrameSet fs, SwsDevParams p, int is422, int npairs, int sliceY)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npairs) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const SwsLutParams &L = p.lut;
    const int yrow = 2 * blockIdx.y;
#pragma unroll
    for (int l = 0; l < 2; l++) {
        const int yy = sliceY + yrow + l;
        const int cr = is422 ? yy : ((sliceY + yrow) >> 1);
        const int U = f.src[1][(int64_t)cr * f.srcStride[1] + i], V = f.src[2][(int64_t)cr * f.srcStride[2] + i];
        const ChromaIdx k = lut_chroma(L, U, V);
        const uint8_t *py = f.src[0] + (int64_t)yy * f.srcStride[0] + 2 * i;
        uint8_t *drow = f.dst[0] + (int64_t)yy * f.dstStride[0];
        // (the 2-pixel tail behind a 4-pixel tail -- dstW & 6 == 6 -- starts its dither row over at column 0: every section of YUV420FUNC_DITHER begins with
        //  PUTFUNC(1, 0, 0), yuv2rgb.c:283-318; round 6, found against the real reference on the CPU box)
        const int xd = ((p.dstW & 6) == 6 && i == npairs - 1) ? 0 : 2 * i;
        const uint32_t v0 = lut_rgb8(L, k, py[0], yy, xd), v1 = lut_rgb8(L, k, py[1], yy, xd + 1);
        if (p.dstKind == DSTK_RGB4) drow[i] = (uint8_t)(v0 | (v1 << 4));
        else { drow[2 * i] = (uint8_t)v0; drow[2 * i + 1] = (uint8_t)v1; }
    }
}

// rgbToRgbWrapper with the 12/15/16 bpp converters of rgb2rgb.c:179-320 and rgb2rgb_template.c:85-316: bit-field shuffles in "int"
// order (a 16-bit pixel has a high, a middle and a low field; a 24/32 bpp pixel three bytes in memory order, after the alpha byte
// for the _1 layouts).  same = both formats have the same channel order "in int" (findRgbConvFn's first switch), else the second.
struct RgbLowPlan { int32_t sid, did, same, s_alt, d_alt; };
__global__ void __launch_bounds__(256) sws_k_rgb_low_convert(SwsFrameSet fs, RgbLowPlan rp, int w, int sliceY)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= w) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int y = sliceY + blockIdx.y;
    const uint8_t *sp = f.src[0] + (int64_t)y * f.srcStride[0] + (rp.s_alt ? 1 : 0);
    uint8_t *dp = f.dst[0] + (int64_t)y * f.dstStride[0];
    const int sid = rp.sid, did = rp.did;
    const bool same = rp.same != 0;
    unsigned px = 0, b0 = 0, b1 = 0, b2 = 0;
    if (sid <= 16) px = ((const uint16_t *)sp)[x];
    else { const uint8_t *q = sp + (sid == 24 ? 3 : 4) * x; b0 = q[0]; b1 = q[1]; b2 = q[2]; }
    if (did <= 16) {
        unsigned out;
        if (sid == 12 && did == 15) {                                   // rgb12to15
            unsigned r = px & 0xF00, g = px & 0x0F0, b = px & 0x00F;
            r = (r << 3) | ((r & 0x800) >> 1); g = (g << 2) | ((g & 0x080) >> 2); b = (b << 1) | (b >> 3);
            out = r | g | b;
        } else if (sid == 12) out = (px << 8 | (px & 0xF0) | px >> 8) & 0xFFF;                                              // rgb12tobgr12
        else if (sid == 15 && did == 16) out = same ? (px & 0x7FFF) + (px & 0x7FE0) : ((px & 0x7C00) >> 10) | ((px & 0x3E0) << 1) | (px << 11);   // rgb15to16 / rgb15tobgr16
        else if (sid == 16 && did == 15) out = same ? ((px >> 1) & 0x7FE0) | (px & 0x001F) : (px >> 11) | ((px & 0x7C0) >> 1) | ((px & 0x1F) << 10); // rgb16to15 / rgb16tobgr15
        else if (sid == 16 && did == 16) out = (px >> 11) | (px & 0x7E0) | (px << 11);                                      // rgb16tobgr16
        else if (sid == 15 && did == 15) { const unsigned br = px & 0x7C1F; out = (br >> 10) | (px & 0x3E0) | (br << 10); }  // rgb15tobgr15
        else if (sid == 24) {                                           // rgb24to16/15 (first byte -> high field), rgb24tobgr16/15 (first byte -> low field)
            const unsigned hi = same ? b0 : b2, lo = same ? b2 : b0;
            out = did == 16 ? (lo >> 3) | ((b1 & 0xFC) << 3) | ((hi & 0xF8) << 8) : (lo >> 3) | ((b1 & 0xF8) << 2) | ((hi & 0xF8) << 7);
        } else {                                                        // rgb32to16/15 (byte 0 -> low field), rgb32tobgr16/15 (byte 0 -> high field)
            const unsigned lo = same ? b0 : b2, hi = same ? b2 : b0;
            out = did == 16 ? (lo >> 3) + ((b1 & 0xFC) << 3) + ((hi & 0xF8) << 8) : (lo >> 3) + ((b1 & 0xF8) << 2) + ((hi & 0xF8) << 7);
        }
        ((uint16_t *)dp)[x] = (uint16_t)out;
    } else {
        unsigned hi8, mid8, lo8;
        if (sid == 16) { hi8 = ((px & 0xF800) >> 8) | ((px & 0xF800) >> 13); mid8 = ((px & 0x07E0) >> 3) | ((px & 0x07E0) >> 9); lo8 = ((px & 0x001F) << 3) | ((px & 0x001F) >> 2); }
        else           { hi8 = ((px & 0x7C00) >> 7) | ((px & 0x7C00) >> 12); mid8 = ((px & 0x03E0) >> 2) | ((px & 0x03E0) >> 7); lo8 = ((px & 0x001F) << 3) | ((px & 0x001F) >> 2); }
        if (did == 24) {                                                // rgb16to24 / rgb15to24: high field first; ...tobgr24: low field first
            uint8_t *q = dp + 3 * x;
            q[0] = (uint8_t)(same ? hi8 : lo8); q[1] = (uint8_t)mid8; q[2] = (uint8_t)(same ? lo8 : hi8);
        } else {                                                        // rgb16to32 / rgb15to32: low field first; ...tobgr32: high field first; opaque alpha
            uint8_t *q = dp + 4 * x;
            const unsigned o0 = same ? lo8 : hi8, o2 = same ? hi8 : lo8;
            if (rp.d_alt) { q[0] = 255; q[1] = (uint8_t)o0; q[2] = (uint8_t)mid8; q[3] = (uint8_t)o2; }
            else          { q[0] = (uint8_t)o0; q[1] = (uint8_t)mid8; q[2] = (uint8_t)o2; q[3] = 255; }
        }
    }
}

// bgr24ToYv12Wrapper (swscale_unscaled.c:2062-2078) -> ff_rgb24toyv12_c (rgb2rgb_template.c:580-641).
// One thread = 4 chroma samples = 8 pixels x 2 rows: 2 x 24 bytes in, 2 x 8 luma + 4 U + 4 V bytes out.
// All arithmetic is unsigned and the results are stored modulo 256 exactly like the reference's uint8_t stores.
__device__ __forceinline__ uint32_t rgb24toyv12_dot(int32_t cr, int32_t cg, int32_t cb, uint32_t r, uint32_t g, uint32_t b, uint32_t bias)
{
    return ((((uint32_t)cr * r + (uint32_t)cg * g + (uint32_t)cb * b) >> 15) + bias) & 0xffu;
}

__global__ void __launch_bounds__(256) sws_k_bgr24_to_yv12(SwsFrameSet fs, SwsDevParams p, int sliceY, int sliceH, int cbase)
{
    const int cw = p.srcW >> 1;
    const int c0 = cbase + (blockIdx.x * 256 + threadIdx.x) * 4;     // (cbase: the columns behind those of the vector form)
    if (c0 >= cw) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int y = blockIdx.y * 2, y2 = (y + 1 == sliceH) ? y : y + 1;
    const uint8_t *s1 = f.src[0] + (int64_t)(sliceY + y) * f.srcStride[0] + c0 * 6;   // absolute rows (host rebases slice pointers)
    const uint8_t *s2 = f.src[0] + (int64_t)(sliceY + y2) * f.srcStride[0] + c0 * 6;
    uint8_t *d1 = f.dst[0] + (int64_t)(sliceY + y) * f.dstStride[0] + c0 * 2;
    uint8_t *d2 = f.dst[0] + (int64_t)(sliceY + y2) * f.dstStride[0] + c0 * 2;
    uint8_t *du = f.dst[1] + (int64_t)((sliceY >> 1) + (y >> 1)) * f.dstStride[1] + c0;
    uint8_t *dv = f.dst[2] + (int64_t)((sliceY >> 1) + (y >> 1)) * f.dstStride[2] + c0;
    const int n = min(4, cw - c0);
    const int32_t ry = p.rgb2yuv[0], gy = p.rgb2yuv[1], by = p.rgb2yuv[2], ru = p.rgb2yuv[3], gu = p.rgb2yuv[4], bu = p.rgb2yuv[5],
                  rv = p.rgb2yuv[6], gv = p.rgb2yuv[7], bv = p.rgb2yuv[8];
    uint8_t row1[24], row2[24];
    const bool fast = n == 4 && ((((uintptr_t)s1) | ((uintptr_t)s2)) & 3) == 0;
    if (fast) {
#pragma unroll
        for (int k = 0; k < 6; k++) {
            reinterpret_cast<uint32_t *>(row1)[k] = reinterpret_cast<const uint32_t *>(s1)[k];
            reinterpret_cast<uint32_t *>(row2)[k] = reinterpret_cast<const uint32_t *>(s2)[k];
        }
    } else {
        for (int k = 0; k < 6 * n; k++) { row1[k] = s1[k]; row2[k] = s2[k]; }
    }
    uint8_t Y1[8], Y2[8], U[4], V[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t b[4], g[4], r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint8_t *px = (k < 2 ? row1 : row2) + 6 * i + 3 * (k & 1);
            b[k] = px[0]; g[k] = px[1]; r[k] = px[2];
        }
        Y1[2 * i] = rgb24toyv12_dot(ry, gy, by, r[0], g[0], b[0], 16); Y1[2 * i + 1] = rgb24toyv12_dot(ry, gy, by, r[1], g[1], b[1], 16);
        Y2[2 * i] = rgb24toyv12_dot(ry, gy, by, r[2], g[2], b[2], 16); Y2[2 * i + 1] = rgb24toyv12_dot(ry, gy, by, r[3], g[3], b[3], 16);
        const uint32_t bx = (b[0] + b[1] + b[2] + b[3]) >> 2, gx = (g[0] + g[1] + g[2] + g[3]) >> 2, rx = (r[0] + r[1] + r[2] + r[3]) >> 2;
        U[i] = rgb24toyv12_dot(ru, gu, bu, rx, gx, bx, 128);
        V[i] = rgb24toyv12_dot(rv, gv, bv, rx, gx, bx, 128);
    }
    for (int i = 0; i < n; i++) {
        d1[2 * i] = Y1[2 * i]; d1[2 * i + 1] = Y1[2 * i + 1];
        if (y2 != y) { d2[2 * i] = Y2[2 * i]; d2[2 * i + 1] = Y2[2 * i + 1]; }
        else { d1[2 * i] = Y2[2 * i]; d1[2 * i + 1] = Y2[2 * i + 1]; }   // odd last row: ydst2 == ydst1, the later stores win
        du[i] = U[i]; dv[i] = V[i];
    }
}

// The same conversion for 16-byte aligned frames (round 5: OpenCV-style bgr24 pictures into an encoder's yuv420p at the same size, 0.038 ms per 4K frame on the
// kernel above -- 24 one-byte stores per thread): one thread = 8 chroma samples = 16 pixels x 2 rows, three 16-byte loads per row (a wave reads 3 KiB of a row
// contiguously), one 16-byte luma store per row and one 8-byte store per chroma plane.  Same arithmetic: 32-bit wrap-around sums (v_dot2_i32_i16 + v_mad_i32_i24:
// coefficients of 16 bits against bytes), logical shift, the bias, the low byte.  cgroups = whole groups of 8 chroma columns; the columns behind them take the kernel above.
__global__ void __launch_bounds__(256) sws_k_bgr24_to_yv12_vec(SwsFrameSet fs, SwsDevParams p, int sliceY, int sliceH, int cgroups)
{
    const int gidx = blockIdx.x * 256 + threadIdx.x;
    if (gidx >= cgroups) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int y = blockIdx.y * 2, y2 = (y + 1 == sliceH) ? y : y + 1;
    const u32x4 *s1 = reinterpret_cast<const u32x4 *>(f.src[0] + (int64_t)(sliceY + y) * f.srcStride[0]) + 3 * gidx;
    const u32x4 *s2 = reinterpret_cast<const u32x4 *>(f.src[0] + (int64_t)(sliceY + y2) * f.srcStride[0]) + 3 * gidx;
    uint32_t rw[2][12];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const u32x4 a = __builtin_nontemporal_load(s1 + k), b = __builtin_nontemporal_load(s2 + k);
#pragma unroll
        for (int j = 0; j < 4; j++) { rw[0][4 * k + j] = a[j]; rw[1][4 * k + j] = b[j]; }
    }
    // {byte 0, byte 1} of a pixel as a pair of 16-bit operands of v_dot2_i32_i16 against {by, gy}; byte 2 through v_mad_i32_i24; the bias rides in the accumulator
    // (((S >> 15) + bias) & 255 == ((S + (bias << 15)) >> 15) & 255: one v_bfe_u32).  The host checks that the six dot2 coefficients fit 16 bits.
    const uint32_t kY = ((uint32_t)p.rgb2yuv[2] & 0xFFFFu) | (uint32_t)p.rgb2yuv[1] << 16, kU = ((uint32_t)p.rgb2yuv[5] & 0xFFFFu) | (uint32_t)p.rgb2yuv[4] << 16,
                   kV = ((uint32_t)p.rgb2yuv[8] & 0xFFFFu) | (uint32_t)p.rgb2yuv[7] << 16;
    const int ry = p.rgb2yuv[0], ru = p.rgb2yuv[3], rv = p.rgb2yuv[6];
    auto bg_of = [&](int row, int n) -> uint32_t {          // bytes n, n + 1 of the row -> {b, g} halves
        const int w = n >> 2, o = n & 3;
        const uint32_t lo = rw[row][w], hi = rw[row][o == 3 ? w + 1 : w];
        return __builtin_amdgcn_perm(hi, lo, o == 3 ? 0x0C040C03u : (0x0C000C00u | (uint32_t)o | (uint32_t)(o + 1) << 16));
    };
    auto r_of = [&](int row, int n) -> uint32_t { return (rw[row][n >> 2] >> (8 * (n & 3))) & 0xFFu; };
    uint32_t Y[2][4] = {}, U[2] = {}, V[2] = {};
#pragma unroll
    for (int i = 0; i < 8; i++) {               // chroma sample i: pixels 2 i, 2 i + 1 of both rows
        uint32_t sbg = 0, sr = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int row = k >> 1, px = 2 * i + (k & 1);
            const uint32_t bg = bg_of(row, 3 * px), r = r_of(row, 3 * px + 2);
            sbg += bg; sr += r;                 // (sums of four bytes per half: no carry between the halves)
            const int S = sdot2(bg, kY, mad24((int)r, ry, 16 << 15));
            Y[row][px >> 2] |= (((uint32_t)S >> 15) & 0xFFu) << (8 * (px & 3));
        }
        sbg = (sbg >> 2) & 0x00FF00FFu; sr >>= 2;
        const int Su = sdot2(sbg, kU, mad24((int)sr, ru, 128 << 15)), Sv = sdot2(sbg, kV, mad24((int)sr, rv, 128 << 15));
        U[i >> 2] |= (((uint32_t)Su >> 15) & 0xFFu) << (8 * (i & 3));
        V[i >> 2] |= (((uint32_t)Sv >> 15) & 0xFFu) << (8 * (i & 3));
    }
    u32x4 *d1 = reinterpret_cast<u32x4 *>(f.dst[0] + (int64_t)(sliceY + y) * f.dstStride[0]) + gidx;
    u32x4 *d2 = reinterpret_cast<u32x4 *>(f.dst[0] + (int64_t)(sliceY + y2) * f.dstStride[0]) + gidx;
    u32x2 *du = reinterpret_cast<u32x2 *>(f.dst[1] + (int64_t)((sliceY >> 1) + (y >> 1)) * f.dstStride[1]) + gidx;
    u32x2 *dv = reinterpret_cast<u32x2 *>(f.dst[2] + (int64_t)((sliceY >> 1) + (y >> 1)) * f.dstStride[2]) + gidx;
    const u32x4 o1 = { Y[0][0], Y[0][1], Y[0][2], Y[0][3] }, o2 = { Y[1][0], Y[1][1], Y[1][2], Y[1][3] };
    if (y2 != y) { *d1 = o1; *d2 = o2; }
    else *d1 = o2;                              // odd last row: ydst2 == ydst1, the later stores win (both rows are this row: the same bytes)
    const u32x2 ou = { U[0], U[1] }, ov = { V[0], V[1] };
    *du = ou; *dv = ov;
}

} // namespace swsk
