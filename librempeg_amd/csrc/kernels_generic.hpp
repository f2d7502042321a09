// Generic (any supported format / any tap count) kernels: input readers, horizontal stage,
// vertical stage + output writers.  One thread per output element; used directly for uncommon
// shapes and as the structural template of the specialised kernels in kernels_fast.hpp.
//
// Arithmetic contract: every shift, rounding constant and clip sits at the
// same point as in the reference's C functions cited next to each routine.
#pragma once
#include <type_traits>
#include "kernels_common.hpp"

namespace swsk {

// ------------------------------------------------------------------------------------------
// input readers (libswscale/input.c): value of the "formatConv" line for component comp
// (0 = Y, 1 = U, 2 = V) at source row `row` (luma or chroma row), column x.
// ------------------------------------------------------------------------------------------
template <typename P>
__device__ __forceinline__ int read_sample(const P &p, const SwsFramePtrs &f, int comp, int row, int x, int aux = -1)
{
    if (comp == 3 && p.srcKind == SRCK_RGB48)   // rgba64leToA_c: the 16-bit A word as is
        return ((const uint16_t *)(f.src[0] + (int64_t)row * f.srcStride[0]))[4 * x + 3];
    // single-plane sources: chroma row r of the scaler is picture row r << chrSrcVSub (non-zero only with SWS_SRC_V_CHR_DROP: ff_swscale
    // multiplies the chroma strides, which for a packed source are the strides of plane 0, swscale.c:318-334)
    const int prow = (comp == 1 || comp == 2) ? (row << p.chrSrcVSub) : row;
    // planar RGB sources: every line needs all three planes.  The reference's slices index planes 1 and 2 by chroma row and planes 0 and 3
    // by luma row whatever the line is for (slice.c ff_init_slice_from_src, hscale.c:lum_convert / chr_convert): a luma line y reads B and R
    // at chroma row y >> chrSrcVSub, a chroma line reads G at luma row y << chrSrcVSub (both the same row unless SWS_SRC_V_CHR_DROP is set)
    // (chr_convert derives the plane-0 row from the start of the BATCH of chroma lines it is called for, "sp0 + i": when the line schedule matters
    //  -- SWS_SRC_V_CHR_DROP on a planar RGB source -- pass 1 gets that row as the line's side term, see SwsDevParams::vlines)
    const int grow = ((comp == 1 || comp == 2) && aux >= 0 && p.vline_mode == 2) ? aux : prow, brow = (comp == 1 || comp == 2) ? row : (row >> p.chrSrcVSub);
    if (p.srcKind == SRCK_PACKEDHI) {   // the descriptor's field of component comp
        const uint8_t *s = f.src[0] + (int64_t)prow * f.srcStride[0] + pick4(p.shi_step, comp) * x + pick4(p.shi_off, comp);
        return (*(const uint16_t *)s >> pick4(p.shi_shift, comp)) & pick4(p.shi_mask, comp);
    }
    if (p.srcKind == SRCK_FLOATX) {
        // float / half-float sources: every element becomes lrintf(av_clipf(65535.0f * x, 0.0f, 65535.0f)) first, then the 16-bit RGB
        // arithmetic of the rgb48 readers.  rgbf32_to_y_c / _uv_c / _uv_half_c (input.c:1336-1397), rgbf16To*_endian / rgbaf16To*_endian
        // (:1629-1740), grayf16ToY16_c / read_yaf16_* / read_yaf32_* (:1411-1431, :1601-1627), planar_rgbf16_to_y / _uv / _a (:1561-1599).
        // half2float (libavutil/half2float.h) is the exact widening, like the hardware conversion.
        auto ld = [&](const uint8_t *q) { return p.sf_half ? f32_to_u16((float)*(const _Float16 *)q) : f32_to_u16(*(const float *)q); };
        const int esz = p.sf_half ? 2 : 4;
        if (p.sf_layout == 1)   // gray, gray + alpha: comp 0 = gray, 3 = alpha
            return ld(f.src[0] + (int64_t)row * f.srcStride[0] + p.sf_step * x + (comp == 3 ? p.sf_a_off : 0));
        int r, g, b;
        if (p.sf_layout == 2) {
            if (comp == 3) return ld(f.src[3] + (int64_t)row * f.srcStride[3] + esz * x);
            g = ld(f.src[0] + (int64_t)grow * f.srcStride[0] + esz * x);
            b = ld(f.src[1] + (int64_t)brow * f.srcStride[1] + esz * x);
            r = ld(f.src[2] + (int64_t)brow * f.srcStride[2] + esz * x);
        } else {
            if (comp == 3) return ld(f.src[0] + (int64_t)row * f.srcStride[0] + p.sf_step * x + p.sf_a_off);
            const uint8_t *s = f.src[0] + (int64_t)prow * f.srcStride[0];
            if (comp != 0 && p.chr_half) {   // the half forms average the two converted pixels with a plain >> 1 (no rounding term)
                const uint8_t *q = s + 2 * p.sf_step * x;
                r = (ld(q) + ld(q + p.sf_step)) >> 1; g = (ld(q + esz) + ld(q + p.sf_step + esz)) >> 1; b = (ld(q + 2 * esz) + ld(q + p.sf_step + 2 * esz)) >> 1;
            } else { const uint8_t *q = s + p.sf_step * x; r = ld(q); g = ld(q + esz); b = ld(q + 2 * esz); }
        }
        const int32_t *t = p.rgb2yuv;
        const int o = comp == 0 ? 0 : comp == 1 ? 3 : 6;
        const Rgb2YuvRow tr = rgb2yuv_row(p.rgb2yuv, o);
        return (uint16_t)((int)((unsigned)tr.r * r + (unsigned)tr.g * g + (unsigned)tr.b * b + ((comp == 0 ? 0x2001u : 0x10001u) << 14)) >> 15);
    }
    if (p.srcKind == SRCK_PAL) {   // palToY_c / palToUV_c / palToA_c (input.c:474-512) on the frame's pal_yuv table (sws_k_update_palette)
        const uint32_t e = ((const uint32_t *)f.src[1])[f.src[0][(int64_t)prow * f.srcStride[0] + x]];
        if (comp == 3) return (int)((e >> 24) << 6 | e >> 26);
        return (int)((e >> (8 * comp)) & 0xFF) << 6;
    }
    if (p.srcKind == SRCK_PACKED411) {   // uyyvyyToY_c / uyyvyyToUV_c (input.c:909-925): U Y Y V Y Y groups
        const uint8_t *s = f.src[0] + (int64_t)prow * f.srcStride[0];
        return comp == 0 ? s[3 * (x >> 1) + 1 + (x & 1)] : s[6 * x + (comp == 1 ? 0 : 3)];
    }
    if (p.srcKind == SRCK_YA) {   // ya8: yuy2ToY_c / uyvyToY_c on the two bytes; ya16le: read_ya16le_gray_c / _alpha_c (comp 0 = gray, 3 = alpha)
        const uint8_t *s = f.src[0] + (int64_t)row * f.srcStride[0];
        if (p.src_depth == 8) return s[2 * x + (comp == 3 ? 1 : 0)];
        return ((const uint16_t *)s)[2 * x + (comp == 3 ? 1 : 0)];
    }
    if (comp == 3 && p.srcKind == SRCK_PACKED444)   // read_vuya_A_c / read_ayuv_A_c
        return f.src[0][(int64_t)row * f.srcStride[0] + 4 * x + p.s444_a];
    if (comp == 3) {   // alpha line: plane 3 of yuva (8 bit), or rgbaToA_c / abgrToA_c (input.c:454-472) for 32 bpp RGB
        if (p.srcKind == SRCK_RGB32) {
            const int a = p.src_alpha_opaque ? 255 : f.src[0][(int64_t)row * f.srcStride[0] + 4 * x + p.src_a_pos];
            return a << 6 | a >> 2;
        }
        if (p.srcKind == SRCK_GBRP) return f.src[3][(int64_t)row * f.srcStride[3] + x] << 6;                                   // planar_rgb_to_a input.c:1188-1194
        if (p.srcKind == SRCK_GBRP16) {                                                                                            // planar_rgb16_s16_to_a :1235-1247
            const int bpc = p.src_depth;
            return (uint16_t)(*(const uint16_t *)(f.src[3] + (int64_t)row * f.srcStride[3] + 2 * x) << (14 - (bpc < 16 ? bpc : 14)));
        }
        if (p.srcKind == SRCK_GBRPF32) return f32_to_u16(*(const float *)(f.src[3] + (int64_t)row * f.srcStride[3] + 4 * x));    // planar_rgbf32_to_a :1289-1298
        if (p.srcKind == SRCK_PLANAR16) return *(const uint16_t *)(f.src[3] + (int64_t)row * f.srcStride[3] + 2 * x) >> p.src_shift;   // yuva4xxp9..16: the plane as is
        return f.src[3][(int64_t)row * f.srcStride[3] + x];
    }
    switch (p.srcKind) {
    case SRCK_PLANAR8: {
        const int pl = comp == 0 ? 0 : comp == 1 ? U(p.u_plane_src) : U(p.v_plane_src);
        return pick4(f.src, pl)[(int64_t)row * pick4(f.srcStride, pl) + x];
    }
    case SRCK_PLANAR16: {
        const int pl = comp == 0 ? 0 : comp == 1 ? U(p.u_plane_src) : U(p.v_plane_src);
        return *(const uint16_t *)(pick4(f.src, pl) + (int64_t)row * pick4(f.srcStride, pl) + 2 * x) >> p.src_shift;   // shf16_NNLEToY/UV_c for the msb formats
    }
    case SRCK_NV12: // nv12ToUV_c / nv21ToUV_c, input.c:926-948
        if (comp == 0) return f.src[0][(int64_t)row * f.srcStride[0] + x];
        return f.src[1][(int64_t)row * f.srcStride[1] + 2 * x + ((comp == 1) ^ p.uv_swap_src ? 0 : 1)];
    case SRCK_P010: // p010LEToY_c / p010LEToUV_c, input.c:950-1008
        if (comp == 0) return *(const uint16_t *)(f.src[0] + (int64_t)row * f.srcStride[0] + 2 * x) >> p.src_shift;
        return *(const uint16_t *)(f.src[1] + (int64_t)row * f.srcStride[1] + 4 * x + (comp == 1 ? 0 : 2)) >> p.src_shift;
    case SRCK_RGB24: { // rgb24ToY_c, rgb24ToUV_c, rgb24ToUV_half_c (and bgr24*), input.c:1068-1172
        const int srow = comp == 0 ? row : (row << p.chrSrcVSub);
        const uint8_t *s = f.src[0] + (int64_t)srow * f.srcStride[0];
        const int32_t *t = p.rgb2yuv;
        if (comp == 0) {
            const int r = s[3 * x + p.src_r_pos], g = s[3 * x + 1], b = s[3 * x + p.src_b_pos];
            return (uint16_t)((t[0] * r + t[1] * g + t[2] * b + (32 << 14) + (1 << 8)) >> 9); // stored int16, read back as u16 by hscale
        }
        const int o = comp == 1 ? 3 : 6;
        const Rgb2YuvRow tr = rgb2yuv_row(p.rgb2yuv, o);
        if (p.chr_half) {
            const int r = s[6 * x + p.src_r_pos] + s[6 * x + 3 + p.src_r_pos], g = s[6 * x + 1] + s[6 * x + 4];
            const int b = s[6 * x + p.src_b_pos] + s[6 * x + 3 + p.src_b_pos];
            return (uint16_t)((tr.r * r + tr.g * g + tr.b * b + (256 << 15) + (1 << 9)) >> 10);
        }
        const int r = s[3 * x + p.src_r_pos], g = s[3 * x + 1], b = s[3 * x + p.src_b_pos];
        return (uint16_t)((tr.r * r + tr.g * g + tr.b * b + (256 << 14) + (1 << 8)) >> 9);
    }
    case SRCK_RGB32: { // rgb16_32ToY/UV/UV_half_c_template with the 32-bit parameter rows, input.c:264-393
        const int srow = comp == 0 ? row : (row << p.chrSrcVSub);
        const uint8_t *s = f.src[0] + (int64_t)srow * f.srcStride[0];
        const int32_t *t = p.rgb2yuv;
        const int S = 15 + 8;
        // (the pixel as ONE dword -- the reference reads it with AV_RN32A -- and the bytes out of the register: a third of the load instructions, which
        //  is what this per-sample reader is bound by)
        const int rsh = 8 * p.src_r_pos, gsh = 8 * p.src_g_pos, bsh = 8 * p.src_b_pos;
        if (comp == 0) {
            const uint32_t px = *(const uint32_t *)(s + 4 * x);
            const int r = (px >> rsh) & 0xFF, g = ((px >> gsh) & 0xFF) << 8, b = (px >> bsh) & 0xFF;
            const unsigned rnd = (32u << (S - 1)) + (1u << (S - 7));
            return (uint16_t)((unsigned)((t[0] << 8) * r + t[1] * g + (t[2] << 8) * b + rnd) >> (S - 6));
        }
        const int o = comp == 1 ? 3 : 6;
        const Rgb2YuvRow tr = rgb2yuv_row(p.rgb2yuv, o);
        const int cr = tr.r * (1 << 8), cg = tr.g, cb = tr.b * (1 << 8);
        if (p.chr_half) {
            const uint32_t p0 = *(const uint32_t *)(s + 8 * x), p1 = *(const uint32_t *)(s + 8 * x + 4);
            const int r = (int)((p0 >> rsh) & 0xFF) + (int)((p1 >> rsh) & 0xFF);
            const int g = (int)(((p0 >> gsh) & 0xFF) + ((p1 >> gsh) & 0xFF)) << 8;
            const int b = (int)((p0 >> bsh) & 0xFF) + (int)((p1 >> bsh) & 0xFF);
            const unsigned rnd = (256U << S) + (1 << (S - 6));
            return (uint16_t)((unsigned)(cr * r + cg * g + cb * b + rnd) >> (S - 6 + 1)); // unsigned expr, logical shift
        }
        const uint32_t px = *(const uint32_t *)(s + 4 * x);
        const int r = (px >> rsh) & 0xFF, g = ((px >> gsh) & 0xFF) << 8, b = (px >> bsh) & 0xFF;
        const unsigned rnd = (256u << (S - 1)) + (1 << (S - 7));
        return (uint16_t)((unsigned)(cr * r + cg * g + cb * b + rnd) >> (S - 6));
    }
    case SRCK_GRAYF32:   // grayf32ToY16_c input.c:1399-1409
        return f32_to_u16(*(const float *)(f.src[0] + (int64_t)row * f.srcStride[0] + 4 * x));
    case SRCK_MONO: { // monowhite2Y_c / monoblack2Y_c (input.c:514-548); s16_is565 = monowhite.  Chroma is never read (swscale.c:692-694)
        const int v = f.src[0][(int64_t)row * f.srcStride[0] + (x >> 3)];
        return ((((p.s16_is565 ? ~v : v) >> (7 - (x & 7))) & 1) * 16383);
    }
    case SRCK_RGB30: { // rgb16_32ToY/UV/UV_half_c_template with the rgb30le / bgr30le rows (input.c:264-372, :411-412); s16_is565 = x2rgb10le
        const int srow = comp == 0 ? row : (row << p.chrSrcVSub);
        const uint32_t *s = (const uint32_t *)(f.src[0] + (int64_t)srow * f.srcStride[0]);
        const int32_t *t = p.rgb2yuv;
        const int S = 15 + 6, o = comp == 0 ? 0 : comp == 1 ? 3 : 6;
        const Rgb2YuvRow tr = rgb2yuv_row(p.rgb2yuv, o);
        const int x2rgb = p.s16_is565, shr = x2rgb ? 16 : 0, shb = x2rgb ? 0 : 16;
        const int maskr = x2rgb ? 0x3FF00000 : 0x3FF, maskg = 0xFFC00, maskb = x2rgb ? 0x3FF : 0x3FF00000;
        const int cr = tr.r * (x2rgb ? 1 : 16), cg = tr.g, cb = tr.b * (x2rgb ? 16 : 1);
        if (comp != 0 && p.chr_half) {
            const unsigned maskgx = ~(unsigned)(maskr | maskb);
            const unsigned px0 = s[2 * x], px1 = s[2 * x + 1];
            int g = (int)((px0 & maskgx) + (px1 & maskgx));
            const int rb = (int)(px0 + px1 - (unsigned)g);
            const int b = (rb & (maskb | (maskb << 1))) >> shb, r = (rb & (maskr | (maskr << 1))) >> shr;
            g = (g & (maskg | (maskg << 1))) >> 6;
            const unsigned rnd = (256U << S) + (1u << (S - 6));
            return (uint16_t)((unsigned)(cr * r + cg * g + cb * b + rnd) >> (S - 6 + 1));
        }
        const int px = (int)s[x], b = (px & maskb) >> shb, g = (px & maskg) >> 6, r = (px & maskr) >> shr;
        const unsigned rnd = ((comp == 0 ? 32u : 256u) << (S - 1)) + (1u << (S - 7));
        return (uint16_t)((unsigned)(cr * r + cg * g + cb * b + rnd) >> (S - 6));
    }
    case SRCK_RGB16: { // rgb16_32ToY/UV/UV_half_c_template with the 16 bpp rows (input.c:264-372, :396-401): masks on the unshifted pixel
        const int srow = comp == 0 ? row : (row << p.chrSrcVSub);
        const uint16_t *s = (const uint16_t *)(f.src[0] + (int64_t)srow * f.srcStride[0]);
        const int32_t *t = p.rgb2yuv;
        const int S = p.s16_S, o = comp == 0 ? 0 : comp == 1 ? 3 : 6;
        const Rgb2YuvRow tr = rgb2yuv_row(p.rgb2yuv, o);
        const int cr = tr.r * (1 << p.s16_rsh), cg = tr.g * (1 << p.s16_gsh), cb = tr.b * (1 << p.s16_bsh);
        if (comp != 0 && p.chr_half) {
            const unsigned maskgx = ~(unsigned)(p.s16_maskr | p.s16_maskb);
            const unsigned px0 = s[2 * x], px1 = s[2 * x + 1];
            int g = (int)((px0 & maskgx) + (px1 & maskgx));
            const int rb = (int)(px0 + px1) - g;
            const int b = rb & (p.s16_maskb | (p.s16_maskb << 1)), r = rb & (p.s16_maskr | (p.s16_maskr << 1));
            if (!p.s16_is565) g &= p.s16_maskg | (p.s16_maskg << 1);
            const unsigned rnd = (256U << S) + (1u << (S - 6));
            return (uint16_t)((unsigned)(cr * r + cg * g + cb * b + rnd) >> (S - 6 + 1));
        }
        const int px = s[x], b = px & p.s16_maskb, g = px & p.s16_maskg, r = px & p.s16_maskr;
        const unsigned rnd = ((comp == 0 ? 32u : 256u) << (S - 1)) + (1u << (S - 7));
        return (uint16_t)((unsigned)(cr * r + cg * g + cb * b + rnd) >> (S - 6));
    }
    case SRCK_GBRP: { // planar_rgb_to_y / planar_rgb_to_uv input.c:1174-1211; gbr24pToUV_half_c :414-432
        const uint8_t *G = f.src[0] + (int64_t)grow * f.srcStride[0], *B = f.src[1] + (int64_t)brow * f.srcStride[1],
                      *R = f.src[2] + (int64_t)brow * f.srcStride[2];
        const int32_t *t = p.rgb2yuv;
        if (comp == 0)
            return (uint16_t)((int)((unsigned)t[0] * R[x] + (unsigned)t[1] * G[x] + (unsigned)t[2] * B[x] + (0x801 << 8)) >> 9);
        const int o = comp == 1 ? 3 : 6;
        const Rgb2YuvRow tr = rgb2yuv_row(p.rgb2yuv, o);
        if (p.chr_half) {
            const unsigned g = G[2 * x] + G[2 * x + 1], b = B[2 * x] + B[2 * x + 1], r = R[2 * x] + R[2 * x + 1];
            return (uint16_t)((tr.r * r + tr.g * g + tr.b * b + (0x4001u << 9)) >> 10);
        }
        return (uint16_t)((int)((unsigned)tr.r * R[x] + (unsigned)tr.g * G[x] + (unsigned)tr.b * B[x] + (0x4001 << 8)) >> 9);
    }
    case SRCK_RGB48: { // rgb48ToY/UV(_half)_c_template, rgb64ToY/UV(_half)_c_template (input.c:45-203)
        const int srow = comp == 0 ? row : (row << p.chrSrcVSub);
        const uint16_t *s = (const uint16_t *)(f.src[0] + (int64_t)srow * f.srcStride[0]);
        const int32_t *t = p.rgb2yuv;
        const int st = p.s16_step;
        unsigned r, g, b;
        // the scaling step of the gamma cascade: gamma_convert (gamma.c:31-58) has rewritten the line in place `aux` times by the time the
        // ring takes it (a line pulled again after a hole is converted again, swscale.c:404-451): the table is applied that often here
        const int npass = (p.vline_mode == 1 && aux > 0) ? aux : 0;
        auto gw = [&](unsigned v) { for (int k = 0; k < npass; k++) v = p.gamma_tab[v]; return v; };
        if (comp != 0 && p.chr_half) {
            const uint16_t *q = s + 2 * st * x;
            r = (gw(q[p.s16_r]) + gw(q[st + p.s16_r]) + 1u) >> 1; g = (gw(q[p.s16_g]) + gw(q[st + p.s16_g]) + 1u) >> 1; b = (gw(q[p.s16_b]) + gw(q[st + p.s16_b]) + 1u) >> 1;
        } else { const uint16_t *q = s + st * x; r = gw(q[p.s16_r]); g = gw(q[p.s16_g]); b = gw(q[p.s16_b]); }
        const int o = comp == 0 ? 0 : comp == 1 ? 3 : 6;
        const Rgb2YuvRow tr = rgb2yuv_row(p.rgb2yuv, o);
        return (uint16_t)(((unsigned)tr.r * r + (unsigned)tr.g * g + (unsigned)tr.b * b + ((comp == 0 ? 0x2001u : 0x10001u) << 14)) >> 15);
    }
    case SRCK_PACKED444: {
        const uint8_t *s = f.src[0] + (int64_t)prow * f.srcStride[0] + p.s444_step * x;
        return s[comp == 0 ? U(p.s444_y) : comp == 1 ? U(p.s444_u) : U(p.s444_v)];
    }
    case SRCK_PACKED422: { // yuy2ToY_c / yuy2ToUV_c / yvy2ToUV_c (input.c:550-578), uyvyToY_c / uyvyToUV_c (:890-907)
        const uint8_t *s = f.src[0] + (int64_t)prow * f.srcStride[0];
        return comp == 0 ? s[2 * x + p.s422_y] : s[4 * x + (comp == 1 ? U(p.s422_u) : U(p.s422_v))];
    }
    case SRCK_GBRP16: { // planar_rgb16_s16_to_y / _to_uv, input.c:1216-1270
        // (gbrp10msb / gbrp12msb: planar_rgb16_s10 / s12 shift the samples down first, input.c:1462-1474)
        const int g = *(const uint16_t *)(f.src[0] + (int64_t)grow * f.srcStride[0] + 2 * x) >> p.src_shift;
        const int b = *(const uint16_t *)(f.src[1] + (int64_t)brow * f.srcStride[1] + 2 * x) >> p.src_shift;
        const int r = *(const uint16_t *)(f.src[2] + (int64_t)brow * f.srcStride[2] + 2 * x) >> p.src_shift;
        const int32_t *t = p.rgb2yuv;
        const int bpc = p.src_depth, shift = bpc < 16 ? bpc : 14;
        const int o = comp == 0 ? 0 : comp == 1 ? 3 : 6;
        const Rgb2YuvRow tr = rgb2yuv_row(p.rgb2yuv, o);
        const unsigned bias = ((comp == 0 ? 16u : 128u) << (15 + bpc - 8)) + (1u << shift);
        return (uint16_t)((int)((unsigned)tr.r * r + (unsigned)tr.g * g + (unsigned)tr.b * b + bias) >> (shift + 1));
    }
    case SRCK_GBRPF32: { // planar_rgbf32_to_y / _to_uv, input.c:1300-1334
        const int g = f32_to_u16(*(const float *)(f.src[0] + (int64_t)grow * f.srcStride[0] + 4 * x));
        const int b = f32_to_u16(*(const float *)(f.src[1] + (int64_t)brow * f.srcStride[1] + 4 * x));
        const int r = f32_to_u16(*(const float *)(f.src[2] + (int64_t)brow * f.srcStride[2] + 4 * x));
        const int32_t *t = p.rgb2yuv;
        if (comp == 0)
            return (uint16_t)((int)((unsigned)t[0] * r + (unsigned)t[1] * g + (unsigned)t[2] * b + (0x2001u << 14)) >> 15);
        const int o = comp == 1 ? 3 : 6;
        const Rgb2YuvRow tr = rgb2yuv_row(p.rgb2yuv, o);
        return (uint16_t)((int)((unsigned)tr.r * r + (unsigned)tr.g * g + (unsigned)tr.b * b + (0x10001u << 14)) >> 15);
    }
    default: break;   // (the kinds answered above the switch)
    }
    return 0;
}

// range conversion of one intermediate sample (lum/chrRange{To,From}Jpeg(16)_c, swscale.c:163-255)
template <typename P>
__device__ __forceinline__ int range_sample(const P &p, int v, int chroma)
{
    if (!p.range_active) return v;
    if (!p.wide) {
        const uint16_t coeff = (uint16_t)(chroma ? U(p.chrCoeff) : U(p.lumCoeff));
        const int32_t offset = (int32_t)(chroma ? U(p.chrOffset) : U(p.lumOffset));
        int r = (v * coeff + offset) >> 14;
        if (p.range_to_jpeg) r = min(r, (1 << 15) - 1);
        return (int16_t)r;
    }
    const uint32_t coeff = chroma ? U(p.chrCoeff) : U(p.lumCoeff);
    const int64_t offset = chroma ? U(p.chrOffset) : U(p.lumOffset);
    int r = (int)(((int64_t)v * coeff + offset) >> 18);
    if (p.range_to_jpeg) r = min(r, (1 << 19) - 1);
    return r;
}

// the parameter block with the chr_half field (half-width chroma readers: rgb24ToUV_half_c and friends) behind a constant, like KindView below: a COPY of the
// block in a derived type whose static member of the same name hides the field, so `p.chr_half` names the constant in the routines that see the copy.  (The copy is
// a real object of its type -- rounds 4 / 5 cast the kernel argument to a type it was not created as -- and costs nothing: the block is a read-only kernel argument,
// the compiler forwards every field read to it; tools/kregs.sh and the ISA comparison of round 6 show identical code.)
template <typename P, int H> struct ChrHalfView : P {
    static constexpr int32_t chr_half = H;
    __device__ __forceinline__ explicit ChrHalfView(const P &b) : P(b) {}
};
template <int H, typename P>
__device__ __forceinline__ ChrHalfView<P, H> chr_half_view(const P &p) { return ChrHalfView<P, H>(p); }

// sum of fs taps of one output sample of component COMP.  Four taps at a time, their loads issued together: a thread that waits for every sample
// before it asks for the next one spends the pass on memory latency (4K bgra -> 1080p, 8 taps: 0.49 ms per frame for 66 M samples in the rolled
// loop).  The sum is the reference's 32-bit sum in any order.
template <int COMP, typename P>
__device__ __forceinline__ int tap_sum(const P &p, const SwsFramePtrs &f, int row, int sp, const int16_t *taps, int fs, int aux)
{
    int val = 0, j = 0;
    for (; j + 4 <= fs; j += 4) {
        int s[4], t[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { s[u] = read_sample(p, f, COMP, row, sp + j + u, aux); t[u] = taps[j + u]; }
#pragma unroll
        for (int u = 0; u < 4; u++) val += s[u] * t[u];
    }
    for (; j < fs; j++) val += read_sample(p, f, COMP, row, sp + j, aux) * taps[j];
    return val;
}

// the same for the U and the V sample of one chroma position: one pass over the taps, so the loads both components share (a packed or semi-planar
// source's pixel / chroma pair) are issued once
template <typename P>
__device__ __forceinline__ void tap_sum_uv(const P &p, const SwsFramePtrs &f, int row, int sp, const int16_t *taps, int fs, int aux, int &su, int &sv)
{
    int u = 0, v = 0, j = 0;
    for (; j + 4 <= fs; j += 4) {
        int a[4], b[4], t[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { a[k] = read_sample(p, f, 1, row, sp + j + k, aux); b[k] = read_sample(p, f, 2, row, sp + j + k, aux); t[k] = taps[j + k]; }
#pragma unroll
        for (int k = 0; k < 4; k++) { u += a[k] * t[k]; v += b[k] * t[k]; }
    }
    for (; j < fs; j++) { const int t = taps[j]; u += read_sample(p, f, 1, row, sp + j, aux) * t; v += read_sample(p, f, 2, row, sp + j, aux) * t; }
    su = u; sv = v;
}

// horizontal stage for one output sample (hScale8To15_c / 8To19 / 16To15 / 16To19, swscale.c:69-159)
template <typename P>
__device__ __forceinline__ int hscale_sample(const P &p, const SwsFramePtrs &f, int comp, int row, int x, int aux = -1)
{
    if (p.no_chroma && comp != 0 && comp != 3) return p.wide ? 1 << 18 : 1 << 14;   // ff_init_desc_no_chr: fill_ones() value, never range converted
    const bool lumlike = comp == 0 || comp == 3;   // the alpha plane goes through the luma functions (hscale.c:39-131)
    if (p.fast_bilinear) {   // ff_hyscale_fast_c / ff_hcscale_fast_c, hscale_fast_bilinear.c:23-55
        const int sW = lumlike ? U(p.srcW) : U(p.chrSrcW);
        const uint32_t xpos = (uint32_t)x * (uint32_t)(lumlike ? U(p.lumXInc) : U(p.chrXInc));
        const int xx = (int)(xpos >> 16), xalpha = (int)((xpos & 0xFFFF) >> 9);
        int r;
        if (xx >= sW - 1) r = read_sample(p, f, comp, row, sW - 1, aux) * 128;        // the tail loop of the reference
        else {
            const int a = read_sample(p, f, comp, row, xx, aux), b = read_sample(p, f, comp, row, xx + 1, aux);
            r = lumlike ? (a << 7) + (b - a) * xalpha : a * (xalpha ^ 127) + b * xalpha;
        }
        return comp == 3 ? (int16_t)r : range_sample(p, (int16_t)r, comp != 0);
    }
    const int16_t *filter = lumlike ? U(p.hLumF) : U(p.hChrF);
    const int32_t *pos = lumlike ? U(p.hLumPos) : U(p.hChrPos);
    const int fs = lumlike ? U(p.hLumFs) : U(p.hChrFs);
    const int sp = pos[x];
    // (the component and the half-width chroma readers are the same for a whole block: decided here, outside the tap loop, so that the loop body is
    //  the straight-line code of one reader form)
    const int16_t *taps = filter + (int64_t)fs * x;
    int val;
    if (p.chr_half) {
        const auto &q = chr_half_view<1>(p);
        val = comp == 0 ? tap_sum<0>(q, f, row, sp, taps, fs, aux) : comp == 1 ? tap_sum<1>(q, f, row, sp, taps, fs, aux) :
              comp == 2 ? tap_sum<2>(q, f, row, sp, taps, fs, aux) : tap_sum<3>(q, f, row, sp, taps, fs, aux);
    } else {
        const auto &q = chr_half_view<0>(p);
        val = comp == 0 ? tap_sum<0>(q, f, row, sp, taps, fs, aux) : comp == 1 ? tap_sum<1>(q, f, row, sp, taps, fs, aux) :
              comp == 2 ? tap_sum<2>(q, f, row, sp, taps, fs, aux) : tap_sum<3>(q, f, row, sp, taps, fs, aux);
    }
    int r = min(val >> p.hshift, p.hclip);
    if (!p.wide) r = (int16_t)r;
    return comp == 3 ? r : range_sample(p, r, comp != 0);
}

// ... and for the U and V samples of chroma column x together (callers: contexts with chroma and without the fast-bilinear functions)
template <typename P>
__device__ __forceinline__ void hscale_sample_uv(const P &p, const SwsFramePtrs &f, int row, int x, int aux, int &ru, int &rv)
{
    const int fs = U(p.hChrFs);
    const int sp = U(p.hChrPos)[x];
    const int16_t *taps = U(p.hChrF) + (int64_t)fs * x;
    int u, v;
    if (p.chr_half) tap_sum_uv(chr_half_view<1>(p), f, row, sp, taps, fs, aux, u, v); else tap_sum_uv(chr_half_view<0>(p), f, row, sp, taps, fs, aux, u, v);
    u = min(u >> p.hshift, p.hclip); v = min(v >> p.hshift, p.hclip);
    if (!p.wide) { u = (int16_t)u; v = (int16_t)v; }
    ru = range_sample(p, u, 1); rv = range_sample(p, v, 1);
}

// ---- samplers: where the vertical stage gets h-scaled samples from ----
template <typename T> struct ScratchSampler { // pass-1 output in HBM
    const T *lum, *u, *v; int lumW, chrW; const T *a;
    __device__ __forceinline__ int get(int comp, int row, int x) const
    {
        return comp == 0 ? lum[(int64_t)row * lumW + x] : comp == 1 ? u[(int64_t)row * chrW + x] :
               comp == 2 ? v[(int64_t)row * chrW + x] : a[(int64_t)row * lumW + x];
    }
};
template <typename P = SwsDevParams>
struct DirectSampler { // horizontal filters are 1-tap identity: compute the sample on the fly
    const P *p; const SwsFramePtrs *f;
    __device__ __forceinline__ int get(int comp, int row, int x) const
    {
        if (p->no_chroma && comp != 0 && comp != 3) return p->wide ? 1 << 18 : 1 << 14;
        int r = min((read_sample(*p, *f, comp, row, x) * 16384) >> p->hshift, p->hclip);
        if (!p->wide) r = (int16_t)r;
        return comp == 3 ? r : range_sample(*p, r, comp != 0);
    }
};

// SK / DK >= 0: an instantiation for ONE source / destination kind -- the same routines with the kind switches folded at compile time.  (The all-kinds
// form of the single-pass kernels carries every reader at every tap of every writer: 24 000 instructions, 256 + 256 registers and spills, one wave
// per SIMD with every load of a tap waiting for the one before.)  The routines take the parameter block as a template type; the views below hide the
// kind fields behind constants of the same names, so `p.srcKind == SRCK_...` is decided by the compiler and everything else reads the kernel argument.
template <int SK, int DK> struct KindView : SwsDevParams { static constexpr int32_t srcKind = SK, dstKind = DK; __device__ __forceinline__ explicit KindView(const SwsDevParams &b) : SwsDevParams(b) {} };
template <int SK> struct SrcKindView : SwsDevParams { static constexpr int32_t srcKind = SK; __device__ __forceinline__ explicit SrcKindView(const SwsDevParams &b) : SwsDevParams(b) {} };
template <int DK> struct DstKindView : SwsDevParams { static constexpr int32_t dstKind = DK; __device__ __forceinline__ explicit DstKindView(const SwsDevParams &b) : SwsDevParams(b) {} };
static_assert(std::is_standard_layout<SwsDevParams>::value && std::is_trivially_copyable<SwsDevParams>::value, "the parameter block is plain data");
// (a copy of the kernel argument in the view type, see ChrHalfView; SK = DK = -1: the argument itself)
template <int SK, int DK>
__device__ __forceinline__ decltype(auto) kind_view(const SwsDevParams &p)
{
    if constexpr (SK >= 0 && DK >= 0) return KindView<SK, DK>(p);
    else if constexpr (SK >= 0) return SrcKindView<SK>(p);
    else if constexpr (DK >= 0) return DstKindView<DK>(p);
    else return (p);
}

// ------------------------------------------------------------------------------------------
// pass 1: reader + hscale + range -> scratch planes
// grid: x over output columns, y over source rows, z = frame * 3 + comp
// ------------------------------------------------------------------------------------------
template <typename T, int SK = -1>
__global__ void __launch_bounds__(256) sws_k_hscale(SwsFrameSet fs, SwsDevParams pa, T *scratch, int64_t frame_elems)
{
    const auto &p = kind_view<SK, -1>(pa);
    const int ncomp = p.need_alpha ? 4 : 3;
    const int comp = blockIdx.z % ncomp, fi = blockIdx.z / ncomp;
    const int W = (comp == 0 || comp == 3) ? U(p.dstW) : U(p.chrDstW), H = (comp == 0 || comp == 3) ? U(p.srcH) : U(p.chrSrcH);
    const int x = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
    // the blocks of component 1 produce the U and the V line (one pass over the taps, shared loads issued once); the blocks of component 2 have nothing to do
    const bool uv_together = !p.no_chroma && !p.fast_bilinear;
    if (x >= W || row >= H || (uv_together && comp == 2)) return;
    const SwsFramePtrs f = frame_copy(fs, fi);
    T *base = scratch + fi * frame_elems;
    const int64_t lumElems = (int64_t)p.srcH * p.dstW, chrElems = (int64_t)p.chrSrcH * p.chrDstW;
    T *plane = comp == 0 ? base : comp == 1 ? base + lumElems : comp == 2 ? base + lumElems + chrElems : base + lumElems + 2 * chrElems;
    int srow = row, aux = -1;
    if (p.vlines) {   // (the row of pass 1 is a virtual line: which picture line it is, and its side term)
        const int32_t *e = p.vlines + 2 * ((comp == 0 || comp == 3) ? row : U(p.nVL) + row);
        srow = e[0]; aux = e[1];
    }
    if (uv_together && comp == 1) {
        int u, v;
        hscale_sample_uv(p, f, srow, x, aux, u, v);
        plane[(int64_t)row * W + x] = (T)u; plane[chrElems + (int64_t)row * W + x] = (T)v;
        return;
    }
    plane[(int64_t)row * W + x] = (T)hscale_sample(p, f, comp, srow, x, aux);
}

// sum of fs vertical taps of one output sample, from `init`: four taps at a time with their loads in flight together (like tap_sum); the
// reference's 32-bit sum in any order (its signed and its unsigned-cast forms are the same bits)
template <typename S>
__device__ __forceinline__ int vtap_sum(const S &smp, int comp, int first, int last, int x, const int16_t *vf, int fs, int init)
{
    unsigned val = (unsigned)init;
    int j = 0;
    for (; j + 4 <= fs; j += 4) {
        int s[4], t[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { s[u] = smp.get(comp, min(first + j + u, last), x); t[u] = vf[j + u]; }
#pragma unroll
        for (int u = 0; u < 4; u++) val += (unsigned)s[u] * (unsigned)t[u];
    }
    for (; j < fs; j++) val += (unsigned)smp.get(comp, min(first + j, last), x) * (unsigned)(int)vf[j];
    return (int)val;
}

// ------------------------------------------------------------------------------------------
// vertical stage + planar writers (lum_planar_vscale / chr_planar_vscale vscale.c:41-107,
// writers output.c:149-187, :327-357, :468-493, :538-569)
// comp 0 -> luma plane, 1/2 -> separate chroma planes (planar YUV only)
// ------------------------------------------------------------------------------------------
template <typename S, typename P>
__device__ __forceinline__ void planar_write_one(const P &p, const S &smp, const SwsFramePtrs &f, int comp, int x, int y)
{
    const bool lumlike = comp == 0 || comp == 3;   // alpha: vscale.c:59-71
    const int fs = lumlike ? U(p.vLumFs) : U(p.vChrFs);
    const int16_t *vf = (lumlike ? U(p.vLumF) : U(p.vChrF)) + y * fs;
    const int first = max(1 - fs, (lumlike ? U(p.vLumPos) : U(p.vChrPos))[y]);
    const int srcRows = lumlike ? U(p.srcH) : U(p.chrSrcH);
    const int plane = lumlike ? comp : comp == 1 ? U(p.u_plane_dst) : U(p.v_plane_dst);
    uint8_t *drow = pick4(f.dst, plane) + (int64_t)y * pick4(f.dstStride, plane);
    const int bits = p.dst_bits;
    if (p.dstKind == DSTK_P010) { // luma of P010: yuv2p01xl1_c / yuv2p01xlX_c
        uint16_t *d = (uint16_t *)drow;
        if (fs == 1) {
            const int shift = 15 - bits;
            d[x] = (uint16_t)(clip_uintp2((smp.get(comp, min(first, srcRows - 1), x) + (1 << (shift - 1))) >> shift, bits) << p.dst_shift);
        } else {
            const int shift = 11 + 16 - bits;
            int val = 1 << (shift - 1);
            val = vtap_sum(smp, comp, first, srcRows - 1, x, vf, fs, val);
            d[x] = (uint16_t)(clip_uintp2(val >> shift, bits) << p.dst_shift);
        }
    } else if (p.dstKind == DSTK_PLANARF32) {   // yuv2plane1_float / yuv2planeX_float_c_template (output.c:219-260): the 16-bit value times 1 / 65535
        float *d = (float *)drow;
        const float float_mult = 1.0f / 65535.0f;
        if (fs == 1) {
            d[x] = float_mult * (float)clip_u16((smp.get(comp, min(first, srcRows - 1), x) + 4) >> 3);
        } else {
            int val = (1 << 14) - 0x40000000;
            val = vtap_sum(smp, comp, first, srcRows - 1, x, vf, fs, val);
            d[x] = float_mult * (float)(0x8000 + clip_i16(val >> 15));
        }
    } else if (p.dstKind == DSTK_PLANAR16 || p.dstKind == DSTK_P016) {
        uint16_t *d = (uint16_t *)drow;
        if (fs == 1) {
            d[x] = (uint16_t)clip_u16((smp.get(comp, min(first, srcRows - 1), x) + 4) >> 3);
        } else {
            int val = (1 << 14) - 0x40000000;
            val = vtap_sum(smp, comp, first, srcRows - 1, x, vf, fs, val);
            d[x] = (uint16_t)(0x8000 + clip_i16(val >> 15));
        }
    } else if (p.dstKind == DSTK_PLANARN) {
        uint16_t *d = (uint16_t *)drow;
        if (fs == 1) {
            const int shift = 15 - bits;
            d[x] = (uint16_t)(clip_uintp2((smp.get(comp, min(first, srcRows - 1), x) + (1 << (shift - 1))) >> shift, bits) << p.dst_shift);
        } else {
            const int shift = 11 + 16 - bits;
            int val = 1 << (shift - 1);
            val = vtap_sum(smp, comp, first, srcRows - 1, x, vf, fs, val);
            d[x] = (uint16_t)(clip_uintp2(val >> shift, bits) << p.dst_shift);
        }
    } else { // 8 bit (also the luma plane of NV12)
        const int off = comp == 2 ? 3 : 0; // V plane uses dither offset 3 (vscale.c:99-102)
        const int dv = dither8(p.should_dither, y, x + off);
        if (fs == 1) {
            drow[x] = (uint8_t)clip_u8_shr(smp.get(comp, min(first, srcRows - 1), x) + dv, 7);
        } else {
            int val = dv << 12;
            val = vtap_sum(smp, comp, first, srcRows - 1, x, vf, fs, val);
            drow[x] = (uint8_t)clip_u8_shr(val, 19);
        }
    }
}

// interleaved chroma writers: yuv2nv12cX_c (output.c:495-528), yuv2p01xcX_c (:571-589)
template <typename S, typename P>
__device__ __forceinline__ void nv_chroma_write_one(const P &p, const S &smp, const SwsFramePtrs &f, int x, int cy)
{
    const int fs = p.vChrFs;
    const int16_t *vf = p.vChrF + cy * fs;
    const int first = max(1 - fs, p.vChrPos[cy]);
    uint8_t *drow = f.dst[1] + (int64_t)cy * f.dstStride[1];
    if (p.dstKind == DSTK_P016) {   // yuv2nv12cX_16_c_template, output.c:189-217
        int u = (1 << 14) - 0x40000000, v = (1 << 14) - 0x40000000;
        u = vtap_sum(smp, 1, first, p.chrSrcH - 1, x, vf, fs, u);
        v = vtap_sum(smp, 2, first, p.chrSrcH - 1, x, vf, fs, v);
        uint16_t *d = (uint16_t *)drow;
        d[2 * x] = (uint16_t)(0x8000 + clip_i16(u >> 15));
        d[2 * x + 1] = (uint16_t)(0x8000 + clip_i16(v >> 15));
    } else if (p.dstKind == DSTK_P010) {
        const int bits = p.dst_bits, shift = 11 + 16 - bits;
        int u = 1 << (shift - 1), v = 1 << (shift - 1);
        u = vtap_sum(smp, 1, first, p.chrSrcH - 1, x, vf, fs, u);
        v = vtap_sum(smp, 2, first, p.chrSrcH - 1, x, vf, fs, v);
        uint16_t *d = (uint16_t *)drow;
        d[2 * x] = (uint16_t)(clip_uintp2(u >> shift, bits) << p.dst_shift);
        d[2 * x + 1] = (uint16_t)(clip_uintp2(v >> shift, bits) << p.dst_shift);
    } else {
        int u = dither8(p.should_dither, cy, x) << 12, v = dither8(p.should_dither, cy, x + 3) << 12;
        u = vtap_sum(smp, 1, first, p.chrSrcH - 1, x, vf, fs, u);
        v = vtap_sum(smp, 2, first, p.chrSrcH - 1, x, vf, fs, v);
        drow[2 * x + p.uv_swap_dst] = (uint8_t)clip_u8_shr(u, 19);
        drow[2 * x + 1 - p.uv_swap_dst] = (uint8_t)clip_u8_shr(v, 19);
    }
}

// packed RGB: packed_vscale (vscale.c:109-171) choosing yuv2rgb_{1,2,X}_c_template (output.c:1788-1939)
// or yuv2rgb_full_{1,2,X}_c_template (output.c:2163-2312); unit = pixel pair (LUT) or pixel (full chroma)
template <typename S, typename P>
__device__ __forceinline__ void rgb_write_unit(const P &p, const S &smp, const SwsFramePtrs &f, int i, int y)
{
    const int cy = y >> p.chrDstVSub;
    const int lfs = p.vLumFs, cfs = p.vChrFs;
    const int16_t *lf = p.vLumF + y * lfs, *cf = p.vChrF + cy * cfs;
    const int firstL = max(1 - lfs, p.vLumPos[y]), firstC = max(1 - cfs, p.vChrPos[cy]);
    const int lH = p.srcH - 1, cH = p.chrSrcH - 1;
    uint8_t *drow = f.dst[0] + (int64_t)y * f.dstStride[0];
    const SwsLutParams &L = p.lut;
// a luma column beyond the line (second pixel of the last pair of an odd-width 4:2:2 picture) holds the line buffers' initial value
// (fill_ones, slice.c:190-208): 1 << 14, or 1 << 18 for the 19-bit lines
#define LUM(j, xx) ((xx) < p.dstW ? smp.get(0, min(firstL + (j), lH), (xx)) : (p.wide ? 1 << 18 : 1 << 14))
#define CHU(j, xx) smp.get(1, min(firstC + (j), cH), (xx))
#define CHV(j, xx) smp.get(2, min(firstC + (j), cH), (xx))
#define ALP(j, xx) smp.get(3, min(firstL + (j), lH), (xx))
    if (p.dstKind == DSTK_GBRP || p.dstKind == DSTK_GBRP16 || p.dstKind == DSTK_GBRPF32) {
        // any_vscale (vscale.c:173-212): planar RGB always takes the X form; planes are G, B, R
        uint8_t *dg = drow, *db = f.dst[1] + (int64_t)y * f.dstStride[1], *dr = f.dst[2] + (int64_t)y * f.dstStride[2];
        int Y, U, V, R, G, B;
        if (p.dstKind == DSTK_GBRP) {          // yuv2gbrp_full_X_c, output.c:2342-2421 (15-bit intermediates)
            const int SH = 22 + 8 - p.dst_bits;
            Y = 1 << 9; U = (1 << 9) - (128 << 19); V = (1 << 9) - (128 << 19);
            for (int j = 0; j < lfs; j++) Y += (int)((unsigned)LUM(j, i) * (unsigned)(int)lf[j]);
            for (int j = 0; j < cfs; j++) {
                U += (int)((unsigned)CHU(j, i) * (unsigned)(int)cf[j]);
                V += (int)((unsigned)CHV(j, i) * (unsigned)(int)cf[j]);
            }
            Y >>= 10; U >>= 10; V >>= 10;
            Y -= L.y_offset;
            Y = (int)((unsigned)Y * (unsigned)L.y_coeff);
            Y = (int)((unsigned)Y + (1u << (SH - 1)));
            R = (int)((unsigned)Y + (unsigned)V * (unsigned)L.v2r);
            G = (int)((unsigned)Y + (unsigned)V * (unsigned)L.v2g + (unsigned)U * (unsigned)L.u2g);
            B = (int)((unsigned)Y + (unsigned)U * (unsigned)L.u2b);
            if ((R | G | B) & 0xC0000000) { R = clip_uintp2(R, 30); G = clip_uintp2(G, 30); B = clip_uintp2(B, 30); }
            if (SH != 22) {
                // (yuv2gbrpmsb_full_X_c output.c:2424-2462: the same samples << (16 - depth))
                ((uint16_t *)dg)[i] = (uint16_t)((G >> SH) << p.dst_shift); ((uint16_t *)db)[i] = (uint16_t)((B >> SH) << p.dst_shift);
                ((uint16_t *)dr)[i] = (uint16_t)((R >> SH) << p.dst_shift);
            } else {
                dg[i] = (uint8_t)(G >> 22); db[i] = (uint8_t)(B >> 22); dr[i] = (uint8_t)(R >> 22);
            }
        } else {                                // yuv2gbrp16_full_X_c :2467-2530 / yuv2gbrpf32_full_X_c :2533-2605 (19-bit)
            Y = -0x40000000; U = -(128 << 23); V = -(128 << 23);
            for (int j = 0; j < lfs; j++) Y += (int)((unsigned)LUM(j, i) * (unsigned)(int)lf[j]);
            for (int j = 0; j < cfs; j++) {
                U += (int)((unsigned)CHU(j, i) * (unsigned)(int)cf[j]);
                V += (int)((unsigned)CHV(j, i) * (unsigned)(int)cf[j]);
            }
            Y >>= 14; Y += 0x10000; U >>= 14; V >>= 14;
            Y -= L.y_offset;
            Y = (int)((unsigned)Y * (unsigned)L.y_coeff);
            Y = (int)((unsigned)Y + (unsigned)((1 << 13) - (1 << 29)));
            R = (int)((unsigned)V * (unsigned)L.v2r);
            G = (int)((unsigned)V * (unsigned)L.v2g + (unsigned)U * (unsigned)L.u2g);
            B = (int)((unsigned)U * (unsigned)L.u2b);
            if (p.dstKind == DSTK_GBRP16) {     // 64-bit sums
                ((uint16_t *)dr)[i] = (uint16_t)clip_uintp2((int)(((int64_t)Y + R) >> 14) + (1 << 15), 16);
                ((uint16_t *)dg)[i] = (uint16_t)clip_uintp2((int)(((int64_t)Y + G) >> 14) + (1 << 15), 16);
                ((uint16_t *)db)[i] = (uint16_t)clip_uintp2((int)(((int64_t)Y + B) >> 14) + (1 << 15), 16);
            } else {                            // the float writer adds in 32 bits, then float_mult * (float)v
                const float float_mult = 1.0f / 65535.0f;
                R = clip_uintp2(((int)((unsigned)Y + (unsigned)R) >> 14) + (1 << 15), 16);
                G = clip_uintp2(((int)((unsigned)Y + (unsigned)G) >> 14) + (1 << 15), 16);
                B = clip_uintp2(((int)((unsigned)Y + (unsigned)B) >> 14) + (1 << 15), 16);
                ((float *)dg)[i] = __fmul_rn(float_mult, (float)G); ((float *)db)[i] = __fmul_rn(float_mult, (float)B);
                ((float *)dr)[i] = __fmul_rn(float_mult, (float)R);
            }
        }
        if (p.need_alpha) {   // hasAlpha: the A plane through the luma filter (gbrap*); an alpha plane without source alpha is filled by launch_fill_alpha
            uint8_t *da = f.dst[3] + (int64_t)y * f.dstStride[3];
            int A;
            if (p.dstKind == DSTK_GBRP) {
                const int SH = 22 + 8 - p.dst_bits;
                A = 1 << 18;
                for (int j = 0; j < lfs; j++) A += (int)((unsigned)ALP(j, i) * (unsigned)(int)lf[j]);
                if (A & 0xF8000000) A = clip_uintp2(A, 27);
                if (SH != 22) ((uint16_t *)da)[i] = (uint16_t)(A >> (SH - 3)); else da[i] = (uint8_t)(A >> 19);
            } else {
                A = -0x40000000;
                for (int j = 0; j < lfs; j++) A += (int)((unsigned)ALP(j, i) * (unsigned)(int)lf[j]);
                A >>= 1; A += 0x20002000;
                if (p.dstKind == DSTK_GBRP16) ((uint16_t *)da)[i] = (uint16_t)(clip_uintp2(A, 30) >> 14);
                else ((float *)da)[i] = __fmul_rn(1.0f / 65535.0f, (float)(clip_uintp2(A, 30) >> 14));
            }
        }
        return;
    }
    int mode = 0, ua = 0, ya = 0; // 1: packed1, 2: packed2, 0: X
    if (lfs == 1 && cfs == 1) { mode = 1; }
    else if (lfs == 1 && cfs == 2 && (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U) { mode = 1; ua = (uint16_t)cf[1]; }
    else if (lfs == 2 && cfs == 2 && (uint16_t)lf[1] + (uint16_t)lf[0] == 4096 && (uint16_t)lf[1] <= 4096U &&
             (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U) { mode = 2; ya = (uint16_t)lf[1]; ua = (uint16_t)cf[1]; }
    if (p.dstKind == DSTK_YA) {   // yuv2ya8_{1,2,X}_c (output.c:2613-2705), yuv2ya16_{X,2,1}_c_template (:1016-1113); unit i = pixel i
        int Y, A = 0;
        if (!p.wide) {
            if (mode == 1) {
                Y = clip_u8((LUM(0, i) + 64) >> 7);
                if (p.need_alpha) { A = (ALP(0, i) + 64) >> 7; if (A & 0x100) A = clip_u8(A); }
            } else if (mode == 2) {
                Y = clip_u8((LUM(0, i) * (4096 - ya) + LUM(1, i) * ya) >> 19);
                if (p.need_alpha) A = clip_u8((ALP(0, i) * (4096 - ya) + ALP(1, i) * ya) >> 19);
            } else {
                Y = 1 << 18; A = 1 << 18;
                for (int j = 0; j < lfs; j++) Y += (int)((unsigned)LUM(j, i) * (unsigned)(int)lf[j]);
                Y >>= 19; if (Y & 0x100) Y = clip_u8(Y);
                if (p.need_alpha) { for (int j = 0; j < lfs; j++) A += (int)((unsigned)ALP(j, i) * (unsigned)(int)lf[j]); A >>= 19; if (A & 0x100) A = clip_u8(A); }
            }
            drow[2 * i] = (uint8_t)Y; drow[2 * i + 1] = p.need_alpha ? (uint8_t)A : 255;
        } else {
            if (mode == 1) {
                Y = clip_u16(LUM(0, i) >> 3);
                if (p.need_alpha) { A = ALP(0, i) >> 3; if (A & 0x100) A = clip_u16(A); }   // (sic: the 8-bit test, output.c:1106-1107)
                else A = 65535;
            } else if (mode == 2) {
                const unsigned ya1 = 4096 - ya;
                Y = clip_u16((int)((unsigned)LUM(0, i) * ya1 + (unsigned)LUM(1, i) * (unsigned)ya) >> 15);
                A = p.need_alpha ? clip_u16((int)((unsigned)ALP(0, i) * ya1 + (unsigned)ALP(1, i) * (unsigned)ya) >> 15) : 65535;
            } else {
                Y = -0x40000000; A = 0xffff;
                for (int j = 0; j < lfs; j++) Y += (int)((unsigned)LUM(j, i) * (unsigned)(int)lf[j]);
                Y >>= 15; Y += (1 << 3) + 0x8000; Y = clip_u16(Y);
                if (p.need_alpha) {
                    A = -0x40000000 + (1 << 14);
                    for (int j = 0; j < lfs; j++) A += (int)((unsigned)ALP(j, i) * (unsigned)(int)lf[j]);
                    A >>= 15; A += 0x8000; A = clip_u16(A);
                }
            }
            ((uint16_t *)drow)[2 * i] = (uint16_t)Y; ((uint16_t *)drow)[2 * i + 1] = (uint16_t)A;
        }
        return;
    }
    if (p.dstKind == DSTK_MONO) {   // yuv2mono_{X,2,1}_c_template (output.c:654-860), ordered dither (ff_dither_8x8_220 :84-95); unit i = byte i
        const uint32_t drow_lo[8] = { 0x679e3e75u, 0xba15c722u, 0x4c835990u, 0xce29a500u, 0x6097376eu, 0xb30ec11cu, 0x457c538au, 0xd530ac07u };
        const uint32_t drow_hi[8] = { 0x649b3a71u, 0xb611c41fu, 0x487f568du, 0xd934af0au, 0x6ba24178u, 0xbd18cb26u, 0x4f865d94u, 0xd22da803u };
        const uint32_t dlo = drow_lo[y & 7], dhi = drow_hi[y & 7];
        // the X form shifts every pixel through one running accumulator: a trailing partial byte holds the LAST 8 bits seen, counted
        // over the even-rounded width; the 2 / 1 forms build whole bytes, entries past dstW being the line buffers' fill value
        const int n = (p.dstW + 1) & ~1;
        const bool tail = mode == 0 && 8 * i + 8 > n && !p.mono_y16;
        const int x0 = tail ? n - 8 : 8 * i;
        unsigned acc = 0;
        for (int k = 0; k < 8; k++) {
            const int x = x0 + k;
            int Y;
            if (x < 0) { acc <<= 1; continue; }
            if (p.mono_y16 && x >= n) break;
            if (mode == 0) {
                Y = 1 << 18;
                for (int j = 0; j < lfs; j++) Y += (int)((unsigned)LUM(j, x) * (unsigned)(int)lf[j]);
                Y >>= 19;
                if (Y & 0x100) Y = clip_u8(Y);   // (the pair test "(Y1 | Y2) & 0x100" clips both: clipping an in-range value is the identity)
            } else if (mode == 2) Y = (LUM(0, x) * (4096 - ya) + LUM(1, x) * ya) >> 19;
            else Y = (LUM(0, x) + 64) >> 7;
            if (p.mono_y16) { ((int16_t *)drow)[x] = (int16_t)Y; continue; }   // error diffusion: sws_k_ed_mono takes it from here
            const int dth = (int)(((x & 4) ? dhi : dlo) >> (8 * (x & 3))) & 0xff;
            acc = (acc << 1) | (unsigned)(Y + dth >= 234);
        }
        if (p.mono_y16) { if (i == 0) ((int16_t *)drow)[n] = (int16_t)mode; return; }   // (the diffusion pass stores a trailing partial byte for the X form only)
        drow[i] = (uint8_t)(p.dst_mono_white ? ~acc : acc);
        return;
    }
    if (p.dstKind == DSTK_RGB48) {
        // yuv2rgba64_{X,2,1}_c_template and yuv2rgba64_full_{X,2,1}_c_template (output.c:1115-1560): 19-bit lines, 32-bit
        // wrap-around arithmetic with the reference's signedness of every shift (the full_1 blend shifts LOGICALLY, :1538-1539)
        const int npx = p.full_chr ? 1 : 2;
        uint16_t *drow16 = (uint16_t *)drow;
        for (int h = 0; h < npx; h++) {
            const int x = p.full_chr ? i : 2 * i + h;
            if (x >= p.dstW) break;
            unsigned Y, U, V;
            int A = 0xffff << 14;
            if (mode == 0) {
                Y = (unsigned)-0x40000000; U = (unsigned)-(128 << 23); V = (unsigned)-(128 << 23);
                for (int j = 0; j < lfs; j++) Y += (unsigned)LUM(j, x) * (unsigned)(int)lf[j];
                for (int j = 0; j < cfs; j++) { U += (unsigned)CHU(j, i) * (unsigned)(int)cf[j]; V += (unsigned)CHV(j, i) * (unsigned)(int)cf[j]; }
                if (p.need_alpha) {
                    unsigned a = (unsigned)-0x40000000;
                    for (int j = 0; j < lfs; j++) a += (unsigned)ALP(j, x) * (unsigned)(int)lf[j];
                    A = ((int)a >> 1) + 0x20002000;
                }
                Y = (unsigned)((int)Y >> 14) + 0x10000u;
                U = (unsigned)((int)U >> 14); V = (unsigned)((int)V >> 14);
            } else if (mode == 2) {
                const unsigned ya1 = 4096 - ya, ua1 = 4096 - ua;
                Y = (unsigned)((int)((unsigned)LUM(0, x) * ya1 + (unsigned)LUM(1, x) * (unsigned)ya) >> 14);
                U = (unsigned)((int)((unsigned)CHU(0, i) * ua1 + (unsigned)CHU(1, i) * (unsigned)ua - (128u << 23)) >> 14);
                V = (unsigned)((int)((unsigned)CHV(0, i) * ua1 + (unsigned)CHV(1, i) * (unsigned)ua - (128u << 23)) >> 14);
                if (p.need_alpha) A = ((int)((unsigned)ALP(0, x) * ya1 + (unsigned)ALP(1, x) * (unsigned)ya) >> 1) + (1 << 13);
            } else {
                Y = (unsigned)(LUM(0, x) >> 2);
                if (ua == 0) { U = (unsigned)((CHU(0, i) - (128 << 11)) >> 2); V = (unsigned)((CHV(0, i) - (128 << 11)) >> 2); }
                else {
                    const unsigned ua1 = 4096 - ua;
                    const unsigned tu = (unsigned)CHU(0, i) * ua1 + (unsigned)CHU(1, i) * (unsigned)ua - (128u << 23);
                    const unsigned tv = (unsigned)CHV(0, i) * ua1 + (unsigned)CHV(1, i) * (unsigned)ua - (128u << 23);
                    if (p.full_chr) { U = tu >> 14; V = tv >> 14; } else { U = (unsigned)((int)tu >> 14); V = (unsigned)((int)tv >> 14); }
                }
                if (p.need_alpha) A = (int)((unsigned)ALP(0, x) * (1u << 11)) + (1 << 13);
            }
            Y -= (unsigned)L.y_offset;
            Y *= (unsigned)L.y_coeff;
            Y += (unsigned)((1 << 13) - (1 << 29));
            const unsigned R = V * (unsigned)L.v2r, G = V * (unsigned)L.v2g + U * (unsigned)L.u2g, B = U * (unsigned)L.u2b;
            uint16_t *d = drow16 + p.d16_step * x;
            d[p.d16_r] = (uint16_t)clip_uintp2(((int)(R + Y) >> 14) + (1 << 15), 16);
            d[p.d16_g] = (uint16_t)clip_uintp2(((int)(G + Y) >> 14) + (1 << 15), 16);
            d[p.d16_b] = (uint16_t)clip_uintp2(((int)(B + Y) >> 14) + (1 << 15), 16);
            if (p.d16_step == 4) d[3] = (uint16_t)(clip_uintp2(A, 30) >> 14);
        }
        return;
    }
    if (p.dstKind == DSTK_PACKEDHI) {   // X writers only (no packed1 / packed2 functions exist for these formats)
        const int bits = p.dhi_bits, n = p.dhi_sub ? 4 : 3;
        uint64_t px = (uint64_t)p.dhi_fill_lo | ((uint64_t)p.dhi_fill_hi << 32);
        for (int k = 0; k < n; k++) {
            const bool chroma = k == 1 || k == 2;
            const int fs = chroma ? cfs : lfs;
            const int16_t *fl = chroma ? cf : lf;
            const int x = chroma ? i : (p.dhi_sub ? 2 * i + (k == 3) : i);
            int v;
            if (bits == 16) {
                int acc = (1 << 14) - 0x40000000;
                for (int j = 0; j < fs; j++) acc += (int)((unsigned)(k == 1 ? CHU(j, x) : k == 2 ? CHV(j, x) : LUM(j, x)) * (unsigned)(int)fl[j]);
                v = 0x8000 + min(max(acc >> 15, -32768), 32767);
            } else {
                const int shift = 11 + 16 - bits;
                int acc = 1 << (shift - 1);
                for (int j = 0; j < fs; j++) acc += (int)((unsigned)(k == 1 ? CHU(j, x) : k == 2 ? CHV(j, x) : LUM(j, x)) * (unsigned)(int)fl[j]);
                v = clip_uintp2(acc >> shift, bits);
            }
            px |= (uint64_t)(uint32_t)v << pick5(p.dhi_bitpos, k);
        }
        if (p.dhi_alpha) {
            int v = 65535;
            if (p.need_alpha) {
                int acc = (1 << 14) - 0x40000000;
                for (int j = 0; j < lfs; j++) acc += (int)((unsigned)ALP(j, i) * (unsigned)(int)lf[j]);
                v = 0x8000 + min(max(acc >> 15, -32768), 32767);
            }
            px |= (uint64_t)(uint32_t)v << p.dhi_bitpos[4];
        }
        uint8_t *d = drow + (int64_t)p.dhi_unit_bytes * i;
        if (p.dhi_unit_bytes == 4) *(uint32_t *)d = (uint32_t)px;
        else { ((uint32_t *)d)[0] = (uint32_t)px; ((uint32_t *)d)[1] = (uint32_t)(px >> 32); }
        return;
    }
    if (p.dstKind == DSTK_PACKED444) {   // yuv2ayuv_{X,2,1}_c_template (output.c:2903-3060), yuv2vyu444_{X,2,1}_c (:3171-3290); unit = pixel
        int Y, U, V, A = 255;
        if (mode == 0) {
            Y = U = V = 1 << 18;
            for (int j = 0; j < lfs; j++) Y += (int)((unsigned)LUM(j, i) * (unsigned)(int)lf[j]);
            for (int j = 0; j < cfs; j++) { U += (int)((unsigned)CHU(j, i) * (unsigned)(int)cf[j]); V += (int)((unsigned)CHV(j, i) * (unsigned)(int)cf[j]); }
            Y >>= 19; U >>= 19; V >>= 19;
            if (p.need_alpha) {
                A = 1 << 18;
                for (int j = 0; j < lfs; j++) A += (int)((unsigned)ALP(j, i) * (unsigned)(int)lf[j]);
                A >>= 19;
                if (A & 0x100) A = clip_u8(A);
            }
        } else if (mode == 2) {
            Y = (LUM(0, i) * (4096 - ya) + LUM(1, i) * ya) >> 19;
            U = (CHU(0, i) * (4096 - ua) + CHU(1, i) * ua) >> 19;
            V = (CHV(0, i) * (4096 - ua) + CHV(1, i) * ua) >> 19;
            if (p.need_alpha) A = clip_u8((ALP(0, i) * (4096 - ya) + ALP(1, i) * ya) >> 19);
        } else {
            Y = (LUM(0, i) + 64) >> 7;
            if (ua < 2048) { U = (CHU(0, i) + 64) >> 7; V = (CHV(0, i) + 64) >> 7; }
            else { U = (CHU(0, i) + CHU(1, i) + 128) >> 8; V = (CHV(0, i) + CHV(1, i) + 128) >> 8; }
            if (p.need_alpha) { A = (ALP(0, i) + 64) >> 7; if (A & 0x100) A = clip_u8(A); }
        }
        if (Y & 0x100) Y = clip_u8(Y);
        if (U & 0x100) U = clip_u8(U);
        if (V & 0x100) V = clip_u8(V);
        uint8_t *d = drow + p.d444_step * i;
        d[p.d444_y] = (uint8_t)Y; d[p.d444_u] = (uint8_t)U; d[p.d444_v] = (uint8_t)V;
        if (p.d444_step == 4) d[p.d444_a] = (uint8_t)A;
        return;
    }
    if (p.dstKind == DSTK_PACKED422) {   // yuv2422_{X,2,1}_c_template, output.c:883-1000
        int Y1, Y2, U, V;
        if (mode == 0) {
            Y1 = Y2 = U = V = 1 << 18;
            for (int j = 0; j < lfs; j++) {
                Y1 += (int)((unsigned)LUM(j, 2 * i) * (unsigned)(int)lf[j]);
                Y2 += (int)((unsigned)LUM(j, 2 * i + 1) * (unsigned)(int)lf[j]);
            }
            for (int j = 0; j < cfs; j++) {
                U += (int)((unsigned)CHU(j, i) * (unsigned)(int)cf[j]);
                V += (int)((unsigned)CHV(j, i) * (unsigned)(int)cf[j]);
            }
            Y1 >>= 19; Y2 >>= 19; U >>= 19; V >>= 19;
        } else if (mode == 2) {
            Y1 = (LUM(0, 2 * i) * (4096 - ya) + LUM(1, 2 * i) * ya) >> 19;
            Y2 = (LUM(0, 2 * i + 1) * (4096 - ya) + LUM(1, 2 * i + 1) * ya) >> 19;
            U = (CHU(0, i) * (4096 - ua) + CHU(1, i) * ua) >> 19;
            V = (CHV(0, i) * (4096 - ua) + CHV(1, i) * ua) >> 19;
        } else {
            Y1 = (LUM(0, 2 * i) + 64) >> 7; Y2 = (LUM(0, 2 * i + 1) + 64) >> 7;
            if (ua < 2048) { U = (CHU(0, i) + 64) >> 7; V = (CHV(0, i) + 64) >> 7; }
            else { U = (CHU(0, i) + CHU(1, i) + 128) >> 8; V = (CHV(0, i) + CHV(1, i) + 128) >> 8; }
        }
        if ((Y1 | Y2 | U | V) & 0x100) { Y1 = clip_u8(Y1); Y2 = clip_u8(Y2); U = clip_u8(U); V = clip_u8(V); }
        uint8_t *d = drow + 4 * i;
        d[p.d422_y] = (uint8_t)Y1; d[p.d422_y + 2] = (uint8_t)Y2; d[p.d422_u] = (uint8_t)U; d[p.d422_v] = (uint8_t)V;
        return;
    }
    if (!p.full_chr) {
        int Y1, Y2, U, V;
        if (mode == 0) {
            Y1 = Y2 = U = V = 1 << 18;
            for (int j = 0; j < lfs; j++) {
                Y1 += (int)((unsigned)LUM(j, 2 * i) * (unsigned)(int)lf[j]);
                Y2 += (int)((unsigned)LUM(j, 2 * i + 1) * (unsigned)(int)lf[j]);
            }
            for (int j = 0; j < cfs; j++) {
                U += (int)((unsigned)CHU(j, i) * (unsigned)(int)cf[j]);
                V += (int)((unsigned)CHV(j, i) * (unsigned)(int)cf[j]);
            }
            Y1 >>= 19; Y2 >>= 19; U >>= 19; V >>= 19;
        } else if (mode == 2) {
            const int ya1 = 4096 - ya, ua1 = 4096 - ua;
            Y1 = (LUM(0, 2 * i) * ya1 + LUM(1, 2 * i) * ya) >> 19;
            Y2 = (LUM(0, 2 * i + 1) * ya1 + LUM(1, 2 * i + 1) * ya) >> 19;
            U = (CHU(0, i) * ua1 + CHU(1, i) * ua) >> 19;
            V = (CHV(0, i) * ua1 + CHV(1, i) * ua) >> 19;
        } else {
            Y1 = (LUM(0, 2 * i) + 64) >> 7;
            Y2 = (LUM(0, 2 * i + 1) + 64) >> 7;
            if (ua == 0) { U = (CHU(0, i) + 64) >> 7; V = (CHV(0, i) + 64) >> 7; }
            else {
                const int ua1 = 4096 - ua;
                U = (CHU(0, i) * ua1 + CHU(1, i) * ua + (128 << 11)) >> 19;
                V = (CHV(0, i) * ua1 + CHV(1, i) * ua + (128 << 11)) >> 19;
            }
        }
        const ChromaIdx k = lut_chroma(L, U, V);
        if (p.dstKind == DSTK_RGB32) {
            uint32_t *d = (uint32_t *)drow;
            uint32_t a1 = 0, a2 = 0;
            if (p.need_alpha) {   // output.c:1818-1830 (X), :1870-1875 (2), :1904-1908 / :1929-1933 (1); yuv2rgb_write :1680-1687
                int A1, A2;
                if (mode == 0) {
                    A1 = A2 = 1 << 18;
                    for (int j = 0; j < lfs; j++) {
                        A1 += (int)((unsigned)ALP(j, 2 * i) * (unsigned)(int)lf[j]);
                        A2 += (int)((unsigned)ALP(j, 2 * i + 1) * (unsigned)(int)lf[j]);
                    }
                    A1 >>= 19; A2 >>= 19;
                    if ((A1 | A2) & 0x100) { A1 = clip_u8(A1); A2 = clip_u8(A2); }
                } else if (mode == 2) {
                    A1 = clip_u8((ALP(0, 2 * i) * (4096 - ya) + ALP(1, 2 * i) * ya) >> 19);
                    A2 = clip_u8((ALP(0, 2 * i + 1) * (4096 - ya) + ALP(1, 2 * i + 1) * ya) >> 19);
                } else if (ua == 0) {
                    A1 = clip_u8((ALP(0, 2 * i) * 255 + 16384) >> 15);
                    A2 = clip_u8((ALP(0, 2 * i + 1) * 255 + 16384) >> 15);
                } else {
                    A1 = clip_u8((ALP(0, 2 * i) + 64) >> 7);
                    A2 = clip_u8((ALP(0, 2 * i + 1) + 64) >> 7);
                }
                const int ash = L.a_shift;
                a1 = (uint32_t)A1 << ash; a2 = (uint32_t)A2 << ash;
            }
            d[2 * i] = lut_rgb32(L, k, Y1) + a1;
            d[2 * i + 1] = lut_rgb32(L, k, Y2) + a2;
        } else if (p.dstKind == DSTK_RGB30) {   // yuv2rgb_write, x2rgb10 / x2bgr10 (output.c:1748-1754)
            uint32_t *d = (uint32_t *)drow;
            d[2 * i] = lut_rgb30(L, k, Y1);
            if (2 * i + 1 < p.dstW) d[2 * i + 1] = lut_rgb30(L, k, Y2);
        } else if (p.dstKind == DSTK_RGB8) {    // yuv2rgb_write, "8/4 bits" (output.c:1755-1784): 8x8 ordered dither on the table index
            drow[2 * i] = (uint8_t)lut_rgb8(L, k, Y1, y, 2 * i);
            if (2 * i + 1 < p.dstW) drow[2 * i + 1] = (uint8_t)lut_rgb8(L, k, Y2, y, 2 * i + 1);
        } else if (p.dstKind == DSTK_RGB4) {    // two pixels per byte, the first in the low nibble (output.c:1778-1780); the byte of the last pair of an
                                                // odd width is stored whole, its high nibble from the line buffers' fill value like the reference
            drow[i] = (uint8_t)(lut_rgb8(L, k, Y1, y, 2 * i) + (lut_rgb8(L, k, Y2, y, 2 * i + 1) << 4));
        } else if (p.dstKind == DSTK_RGB16) {   // yuv2rgb_write, 12/15/16 bpp: ordered dither on the luma index (output.c:1714-1748)
            uint16_t *d = (uint16_t *)drow;
            const int bpp = L.bpp16;
            d[2 * i] = (uint16_t)lut_rgb16(L, k.r + Y1 + dither_rgb16_main(bpp, y, 0, 0), k.g + Y1 + dither_rgb16_main(bpp, y, 0, 1),
                                           k.b + Y1 + dither_rgb16_main(bpp, y, 0, 2));
            // (odd widths reach this pair writer: the second pixel of the last pair lies beyond the picture and is not stored)
            if (2 * i + 1 < p.dstW)
                d[2 * i + 1] = (uint16_t)lut_rgb16(L, k.r + Y2 + dither_rgb16_main(bpp, y, 1, 0), k.g + Y2 + dither_rgb16_main(bpp, y, 1, 1),
                                                   k.b + Y2 + dither_rgb16_main(bpp, y, 1, 2));
        } else {
            uint8_t *d = drow + 6 * i;
            const int k0 = L.rgb_order ? k.b : k.r, k2 = L.rgb_order ? k.r : k.b;
            d[0] = (uint8_t)lut_luma(L, k0 + Y1); d[1] = (uint8_t)lut_luma(L, k.g + Y1); d[2] = (uint8_t)lut_luma(L, k2 + Y1);
            d[3] = (uint8_t)lut_luma(L, k0 + Y2); d[4] = (uint8_t)lut_luma(L, k.g + Y2); d[5] = (uint8_t)lut_luma(L, k2 + Y2);
        }
    } else {
        int Y, U, V;
        if (mode == 0) {
            Y = 1 << 9; U = (1 << 9) - (128 << 19); V = (1 << 9) - (128 << 19);
            for (int j = 0; j < lfs; j++) Y += (int)((unsigned)LUM(j, i) * (unsigned)(int)lf[j]);
            for (int j = 0; j < cfs; j++) {
                U += (int)((unsigned)CHU(j, i) * (unsigned)(int)cf[j]);
                V += (int)((unsigned)CHV(j, i) * (unsigned)(int)cf[j]);
            }
            Y >>= 10; U >>= 10; V >>= 10;
        } else if (mode == 2) {
            const int ya1 = 4096 - ya, ua1 = 4096 - ua;
            Y = (LUM(0, i) * ya1 + LUM(1, i) * ya) >> 10;
            U = (CHU(0, i) * ua1 + CHU(1, i) * ua - (128 << 19)) >> 10;
            V = (CHV(0, i) * ua1 + CHV(1, i) * ua - (128 << 19)) >> 10;
        } else {
            Y = LUM(0, i) * 4;
            if (ua == 0) { U = (CHU(0, i) - (128 << 7)) * 4; V = (CHV(0, i) - (128 << 7)) * 4; }
            else {
                const int ua1 = 4096 - ua;
                U = (CHU(0, i) * ua1 + CHU(1, i) * ua - (128 << 19)) >> 10;
                V = (CHV(0, i) * ua1 + CHV(1, i) * ua - (128 << 19)) >> 10;
            }
        }
        // yuv2rgb_write_full, output.c:2005-2070 (8-bit-per-channel targets, no dithering)
        Y -= L.y_offset;
        Y = (int)((unsigned)Y * (unsigned)L.y_coeff);
        Y = (int)((unsigned)Y + (1u << 21));
        int R = (int)((unsigned)Y + (unsigned)V * (unsigned)L.v2r);
        int G = (int)((unsigned)Y + (unsigned)V * (unsigned)L.v2g + (unsigned)U * (unsigned)L.u2g);
        int B = (int)((unsigned)Y + (unsigned)U * (unsigned)L.u2b);
        if ((R | G | B) & 0xC0000000) { R = clip_uintp2(R, 30); G = clip_uintp2(G, 30); B = clip_uintp2(B, 30); }
        if (p.dstKind == DSTK_RGB30) {   // output.c:2052-2063
            const uint32_t r = (uint32_t)(R >> 20), g = (uint32_t)(G >> 20), b = (uint32_t)(B >> 20);
            ((uint32_t *)drow)[i] = (3u << 30) + (r << L.rshift) + (g << 10) + (b << L.bshift);
            return;
        }
        if (p.dstKind == DSTK_RGB8) {    // output.c:2064-2158 (dither none / a_dither / x_dither; error diffusion is a pass of its own)
            drow[i] = (uint8_t)full_rgb8(L, R, G, B, i, y);
            return;
        }
        uint8_t *d = drow + L.pix_step * i;
        d[L.r_pos] = (uint8_t)(R >> 22); d[L.g_pos] = (uint8_t)(G >> 22); d[L.b_pos] = (uint8_t)(B >> 22);
        if (L.pix_step == 4) {
            int A = 255;
            if (p.need_alpha) {   // output.c:2193-2201 (X), :2241-2245 (2), :2278-2283 / :2298-2303 (1)
                if (mode == 0) {
                    A = 1 << 18;
                    for (int j = 0; j < lfs; j++) A += (int)((unsigned)ALP(j, i) * (unsigned)(int)lf[j]);
                    A >>= 19;
                } else if (mode == 2) A = (ALP(0, i) * (4096 - ya) + ALP(1, i) * ya + (1 << 18)) >> 19;
                else A = (ALP(0, i) + 64) >> 7;
                if (A & 0x100) A = clip_u8(A);
            }
            d[L.a_pos] = (uint8_t)A;
        }
    }
#undef LUM
#undef CHU
#undef CHV
#undef ALP
}

// ---- pass-2 / fused kernels over the generic per-element routines ----
// DIRECT = true : horizontal filters are identity, samples are computed on the fly (single pass)
// DIRECT = false: samples come from the pass-1 scratch planes
template <bool DIRECT, typename T>
struct SamplerFor {
    template <typename P>
    static __device__ __forceinline__ auto make(const P &p, const SwsFramePtrs &f, const T *scratch, int64_t frame_elems, int fi)
    {
        if constexpr (DIRECT) {
            return DirectSampler<P>{&p, &f};
        } else {
            const T *base = scratch + fi * frame_elems;
            const int64_t lumElems = (int64_t)p.srcH * p.dstW, chrElems = (int64_t)p.chrSrcH * p.chrDstW;
            return ScratchSampler<T>{base, base + lumElems, base + lumElems + chrElems, p.dstW, p.chrDstW, base + lumElems + 2 * chrElems};
        }
    }
};

// planar: grid x over columns, y over output rows of that plane, z = frame * ncomp + comp
template <bool DIRECT, typename T, int SK = -1, int DK = -1>
__global__ void __launch_bounds__(256) sws_k_vscale_planar(SwsFrameSet fs, SwsDevParams pa, const T *scratch, int64_t frame_elems, int ncomp)
{
    const auto &p = kind_view<SK, DK>(pa);
    const int comp = blockIdx.z % ncomp, fi = blockIdx.z / ncomp;
    const int W = (comp == 0 || comp == 3) ? U(p.dstW) : U(p.chrDstW), H = (comp == 0 || comp == 3) ? U(p.dstH) : U(p.chrDstH);
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    const SwsFramePtrs f = frame_copy(fs, fi);
    if constexpr (DIRECT && SK >= 0) {   // (the component and the reader form are the same for a whole block: one straight-line reader per branch)
        auto body = [&](const auto &q) {
            const auto smp = SamplerFor<DIRECT, T>::make(q, f, scratch, frame_elems, fi);
            if (comp == 0) planar_write_one(q, smp, f, 0, x, y); else if (comp == 1) planar_write_one(q, smp, f, 1, x, y);
            else if (comp == 2) planar_write_one(q, smp, f, 2, x, y); else planar_write_one(q, smp, f, 3, x, y);
        };
        if (p.chr_half) body(chr_half_view<1>(p)); else body(chr_half_view<0>(p));
    } else {
        const auto smp = SamplerFor<DIRECT, T>::make(p, f, scratch, frame_elems, fi);
        planar_write_one(p, smp, f, comp, x, y);
    }
}

// semi-planar chroma: grid x over chroma columns, y over chroma rows, z = frame
template <bool DIRECT, typename T, int SK = -1, int DK = -1>
__global__ void __launch_bounds__(256) sws_k_vscale_nvchroma(SwsFrameSet fs, SwsDevParams pa, const T *scratch, int64_t frame_elems)
{
    const auto &p = kind_view<SK, DK>(pa);
    const int fi = blockIdx.z;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, cy = blockIdx.y;
    if (x >= p.chrDstW || cy >= p.chrDstH) return;
    const SwsFramePtrs f = frame_copy(fs, fi);
    if constexpr (DIRECT && SK >= 0) {
        auto body = [&](const auto &q) {
            const auto smp = SamplerFor<DIRECT, T>::make(q, f, scratch, frame_elems, fi);
            nv_chroma_write_one(q, smp, f, x, cy);
        };
        if (p.chr_half) body(chr_half_view<1>(p)); else body(chr_half_view<0>(p));
    } else {
        const auto smp = SamplerFor<DIRECT, T>::make(p, f, scratch, frame_elems, fi);
        nv_chroma_write_one(p, smp, f, x, cy);
    }
}

// packed RGB: grid x over units (pixel pairs, or pixels with full chroma), y over rows, z = frame
template <bool DIRECT, typename T, int SK = -1, int DK = -1>
__global__ void __launch_bounds__(256) sws_k_vscale_rgb(SwsFrameSet fs, SwsDevParams pa, const T *scratch, int64_t frame_elems)
{
    const auto &p = kind_view<SK, DK>(pa);
    const int fi = blockIdx.z;
    const int i = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    const int units = p.dstKind == DSTK_MONO ? (p.dstW + 7) >> 3 : (p.full_chr || p.dstKind == DSTK_YA) ? p.dstW : (p.dstW + 1) >> 1;
    if (i >= units || y >= p.dstH) return;
    const SwsFramePtrs f = frame_copy(fs, fi);
    if constexpr (DIRECT && SK >= 0) {
        auto body = [&](const auto &q) {
            const auto smp = SamplerFor<DIRECT, T>::make(q, f, scratch, frame_elems, fi);
            rgb_write_unit(q, smp, f, i, y);
        };
        if (p.chr_half) body(chr_half_view<1>(p)); else body(chr_half_view<0>(p));
    } else {
        const auto smp = SamplerFor<DIRECT, T>::make(p, f, scratch, frame_elems, fi);
        rgb_write_unit(p, smp, f, i, y);
    }
}

} // namespace swsk
