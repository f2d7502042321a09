// Parameter blocks shared by host code and HIP kernels (plain C structs, passed by value).
#pragma once
#include <stdint.h>

// how a component line is fetched from the source image ("input reader", libswscale/input.c)
enum SrcKind {
    SRCK_PLANAR8 = 0,   // u8 planes read directly (yuv420p/422p/444p)
    SRCK_PLANAR16,      // u16 LE planes read directly (yuv4xxpNNle)
    SRCK_NV12,          // Y direct, UV interleaved bytes (nv12ToUV_c input.c:926-948); swap for nv21
    SRCK_P010,          // u16 >> 6, UV interleaved (input.c:950-1008)
    SRCK_RGB24,         // packed 3 bytes; rgb24ToY/UV(_half) input.c:1068-1172
    SRCK_RGB32,         // packed 4 bytes; rgb16_32To*_c_template input.c:264-372
    SRCK_GBRP,          // planar 8-bit RGB; planar_rgb_to_y/uv input.c:1174-1211, gbr24pToUV_half_c :414
    SRCK_GBRPF32,       // planar float RGB; planar_rgbf32_to_y/uv input.c:1287-1334
    SRCK_RGB48,         // rgb48le / bgr48le / rgba64le / bgra64le; rgb48/64ToY/UV(_half)_c_template input.c:45-203
    SRCK_PACKED422,     // yuyv422 / uyvy422 / yvyu422: yuy2ToY/UV, yvy2ToUV, uyvyToY/UV input.c:550-578, :890-907
    SRCK_GBRP16,        // planar 9..16-bit RGB; planar_rgb16_s16_to_y/uv input.c:1216-1270
    SRCK_PACKEDHI,      // y210 / y212 / y216, xv30 / v30x, xv36, xv48, ayuv64: (16-bit word at the descriptor offset) >> shift, masked (input.c:580-606, :663-729, :811-866)
    SRCK_PACKED444,     // ayuv / vuya / vuyx / uyva / vyu444: read_*_Y/UV/A_c, vyuToY/UV_c input.c:731-809 (bytes at the descriptor offsets)
    SRCK_YA,            // ya8 / ya16le: gray and alpha samples interleaved (input.c:631-645, :2400-2403, :2773-2775)
    SRCK_GRAYF32,       // grayf32: grayf32ToY16_c input.c:1399-1409
    SRCK_MONO,          // monowhite / monoblack: monowhite2Y_c / monoblack2Y_c input.c:514-548
    SRCK_RGB30,         // x2rgb10le / x2bgr10le: rgb16_32To*_c_template with the rgb30le / bgr30le rows of input.c:411-412
    SRCK_FLOATX,        // rgbf32 / rgbf16 / rgbaf16 (input.c:1336-1397, :1629-1740), grayf16 / yaf32 / yaf16 (:1411-1431, :1601-1627), gbrpf16 / gbrapf16 (:1561-1599)
    SRCK_PAL,           // pal8 and rgb8 / bgr8 / rgb4_byte / bgr4_byte through the palette of ff_update_palette (swscale.c:873-951): palToY_c / palToUV_c / palToA_c input.c:474-512
    SRCK_BAYER,         // bayer mosaics: no scaler reader, only sws_k_bayer (bayer_template.c)
    SRCK_PACKED411,     // uyyvyy411: uyyvyyToY_c / uyyvyyToUV_c input.c:909-925
    SRCK_RGB16,         // rgb565 / rgb555 / rgb444 and the bgr orders: rgb16_32To*_c_template with the 16 bpp rows of input.c:396-401
};

// how an output line is produced ("output writer", libswscale/output.c)
enum DstKind {
    DSTK_PLANAR8 = 0,   // yuv2plane1_8_c / yuv2planeX_8_c output.c:468-493
    DSTK_PLANARN,       // 9..14 bit LE, yuv2planeX_10_c_template output.c:327-357
    DSTK_PLANAR16,      // yuv2planeX_16_c_template output.c:149-187
    DSTK_NV12,          // luma DSTK_PLANAR8 + yuv2nv12cX_c output.c:495-528
    DSTK_P010,          // yuv2p01xl1_c/lX_c/cX_c output.c:538-589
    DSTK_RGB24,         // yuv2rgb_write 24 bpp (rgb24 / bgr24 by rgb_order)
    DSTK_RGB32,         // yuv2rgb_write 32 bpp (rgba/bgra/argb/abgr by shifts)
    DSTK_GBRP,          // planar RGB 8..14 bit: yuv2gbrp_full_X_c output.c:2342-2421
    DSTK_GBRP16,        // planar RGB 16 bit: yuv2gbrp16_full_X_c output.c:2467-2530
    DSTK_GBRPF32,       // planar RGB float: yuv2gbrpf32_full_X_c output.c:2533-2605
    DSTK_RGB48,         // rgb48le / bgr48le / rgba64le / bgra64le: yuv2rgba64_{X,2,1}_c_template + _full_ variants output.c:1115-1560
    DSTK_PACKED422,     // yuyv422 / yvyu422 / uyvy422: yuv2422_{1,2,X}_c_template output.c:883-1000
    DSTK_P016,          // 16-bit semi-planar: luma yuv2planeX_16_c, chroma yuv2nv12cX_16_c_template output.c:189-217
    DSTK_PACKEDHI,      // yuv2y2xxle_X_c, yuv2y216le_X_c, yuv2xv30le / v30xle_X_c, yuv2xv36le_X_c, yuv2ayuv64le / xv48le_X_c (output.c:2712-2866, :3088-3169)
    DSTK_PACKED444,     // ayuv / vuya / vuyx / uyva: yuv2ayuv_{1,2,X}_c_template output.c:2903-3060; vyu444: yuv2vyu444_{1,2,X}_c :3171-3290
    DSTK_YA,            // ya8 / ya16le: yuv2ya8_{1,2,X}_c output.c:2613-2705, yuv2ya16_{X,2,1}_c_template :1016-1113
    DSTK_PLANARF32,     // grayf32: yuv2plane1_float / yuv2planeX_float_c_template output.c:219-260
    DSTK_MONO,          // monowhite / monoblack: yuv2mono_{X,2,1}_c_template output.c:654-860 (ordered dither)
    DSTK_RGB30,         // x2rgb10le / x2bgr10le: yuv2rgb_write 30 bpp (output.c:1748-1754), yuv2rgb_write_full (:2052-2063)
    DSTK_RGB8,          // rgb8 / bgr8 / rgb4_byte / bgr4_byte (one byte per pixel): yuv2rgb_write "8/4 bits" output.c:1755-1784, yuv2rgb_write_full :2064-2158
    DSTK_RGB4,          // rgb4 / bgr4 (two pixels per byte): yuv2rgb_write output.c:1778-1780
    DSTK_RGB16,         // rgb565 / rgb555 / rgb444 (+ bgr): yuv2rgb_write 16/15/12 bpp with ordered dither output.c:1714-1748
    DSTK_RAW32,         // not a pixel format: the vertical sums of the strip kernels as int32 planes (Y, U, V at the destination size), the input of the
                        // full-chroma RGB epilogue (sws_k_fullchr_rgb: yuv2rgb_full_X_c_template + yuv2rgb_write_full, output.c:2005-2070, :2163-2207)
};

struct SwsFramePtrs {       // one frame: plane base pointers (device addresses) and byte strides
    const uint8_t *src[4];
    uint8_t *dst[4];
    int32_t srcStride[4];
    int32_t dstStride[4];
};

struct SwsFrameSet {        // batch of frames for one launch
    const SwsFramePtrs *table;  // device array, or NULL -> use `one`
    SwsFramePtrs one;
    int32_t count;
};

struct SwsLutParams {       // closed form of the yuv2rgb LUTs (yuv2rgb.c:680-703, :901-961)
    int32_t cy;             // y ramp increment
    int32_t yb0r;           // yb0 + 0x8000
    int32_t crv, cbu, cgu, cgv; // chroma increments (scaled by cy)
    int32_t base_r, base_b, base_g; // yoffs - (inc>>9) [g: yoffs - (cgu>>9) - (cgv>>9)]
    int32_t rshift, gshift, bshift; // 32 bpp channel positions
    uint32_t alpha_or;      // 255 << abase (32 bpp, no source alpha) or 0
    int32_t a_shift;        // 32 bpp: bit position of the alpha byte (abase)
    int32_t rgb_order;      // 24 bpp: 0 = R first (rgb24), 1 = B first (bgr24)
    uint32_t perm32;        // 32 bpp: v_perm_b32 selector moving canonical bytes {first,g,third,alpha} to the format's order
    int32_t swap_rb32;      // 32 bpp: canonical 'first' channel is B (bgra / abgr)
    int32_t bpp30;          // 30 bpp tables (yuv2rgb.c:915-941): 10-bit ramps, rshift / gshift / bshift = 20 / 10 / 0 (or swapped), alpha_or = the X bits
    int32_t bpp8, r8, g8, b8;       // 8 / 4 bpp byte tables (yuv2rgb.c:817-856): 8 or 4 (0 = not such a format) and the bit position of each field
    int32_t dither8;                // full-chroma 8 / 4 bpp writers: SwsDither (NONE, A_DITHER or X_DITHER; error diffusion runs as its own pass)
    int32_t bpp16, r16, g16, b16;   // 12/15/16 bpp tables (yuv2rgb.c:853-897): bits per pixel (0 = not such a format) and the bit position of each field
    // 13-bit coefficients for the full-chroma writers (output.c:2005-2020)
    int32_t y_offset, y_coeff, v2r, v2g, u2g, u2b;
    // byte positions inside the pixel for the full-chroma writers
    int32_t r_pos, g_pos, b_pos, a_pos, pix_step;
};

struct SwsTileGeom {        // fused h+v tile kernel: per component group, from the filter banks (host)
    int32_t TW, TH, tilesX, tilesY, NRmax, NCmax;
    const int32_t *rowStart, *rowCount;   // [tilesY] first source row / number of source rows a tile row needs
    const int32_t *colStart, *colCount;   // [tilesX] first source column / number of source columns
    int32_t lds_bytes;
    // dot2 variant: tap rows padded to start on an even sample/row and to an even length (host)
    const int16_t *hT2, *vT2; int32_t hfs2, vfs2;
    int32_t debug;            // profiling experiments only (SWS_HIP_TILE_DEBUG): 1 = skip phase 2, 2 = skip phase 3, 4 = skip phase 1
};

struct SwsMarchGeom {       // wave-marching fused kernel (kernels_march.hpp), per component group
    int32_t chroma;           // 0 = luma group, 1 = chroma group
    int32_t strips, bands, BAND;
    int32_t NCmax;            // max source-window width (samples, multiple of the 16-byte chunk) over strips
    const int32_t *colStart, *colCount;   // [strips] window origin (chunk aligned) / width per strip
    const int16_t *hT4; int32_t hfs4;     // horizontal taps padded to start on a multiple-of-4 sample, length multiple of 4
    const int16_t *vT2; int32_t vfs2;     // vertical taps padded to start on an even row, even length
    int32_t lds_bytes;
};

struct SwsStripRow {         // marching strip kernel: the scalars of one output row (64 bytes, fetched with one scalar load)
    int32_t pf;               // first source-row PAIR of the row's vertical window ((vpos & ~1) >> 1)
    int32_t rnd_off;          // what the packed writers' rounding constant of this row lacks: 0 (the X forms, yuv2packed1), 1 << 18 for a row that takes yuv2packed2 (no rounding: vscale.c:146-157, output.c:1853-1895)
    int32_t pad0[2];
    uint32_t vt[8];           // vertical taps as pairs aligned to even source rows, zero beyond the filter
    int32_t pad1[4];
};

struct SwsRgbSrcRow {        // sws_k_rgbsrc_unity: the scalars of one chroma output row (64 bytes, through the scalar data cache)
    int32_t first, last;      // chroma source rows of the vertical window: max(1 - vfs, vpos) and first + vfs - 1, both before clamping to the plane
    int32_t pad0[2];
    uint32_t vt[8];           // taps j, j + 1 as a pair of int16 (1 for the one-tap yuv2plane1 form), zero beyond the filter
    int32_t pad1[4];
};

struct SwsStripGeom {        // marching strip kernel (kernels_strip.hpp), per plane class
    int32_t TW, strips, NCmax;            // output columns per strip (64 per lane column), strips per row, max window width (samples)
    int32_t nph, npv;                     // tap pairs that can be non-zero (<= hfs2 / 2, vfs2 / 2)
    int32_t hfs2, vfs2;                   // padded tap-row lengths of hT2 / vT2
    const int32_t *colStart, *colCount;   // [strips] source-window origin (chunk aligned) / width (samples)
    const int16_t *hT2, *vT2;
    const SwsStripRow *rows;              // [plane height]
    int32_t bands, band_rows;             // filled per launch
    int32_t lds_bytes;                    // per block of 4 waves
    int32_t dma_ok, lds_dma_bytes;        // LDS-DMA form (16-bit sources): no source row pair is skipped inside a band; its LDS per block
    int32_t debug;                        // profiling builds only: 1 no h-stage, 2 no v-stage, 4 no stores
    // LDS-DMA form for 8-bit planar sources (kernels_strip8.hpp): raw byte rows in LDS, the h-stage unpacks byte pairs with v_perm, so windows start at
    // any byte and the tap rows carry no alignment padding
    int32_t dma8_ok, nph8, lds_dma8_bytes;   // usable; tap pairs per output column ((taps + 1) / 2); LDS per block of 4 waves
    const int16_t *hT8;                   // [plane width][2 * nph8] horizontal taps from the filter's own first tap on
};

struct SwsRgbGroupPlan {    // marching packed-RGB kernel: everything one pair of output rows needs, as scalars (64 bytes)
    int32_t cbase;            // first chroma source row of the register ring for this group (may be negative: loads clamp)
    int32_t ylum0, ylum1;     // luma source rows of the two output rows (identity vertical luma filter)
    int32_t pad0;
    uint32_t wp[2][4];        // per output row: vertical chroma taps of ring rows (2i, 2i+1) packed (lo & 0xffff) | hi << 16
    int32_t pad1[4];
};

struct SwsDevParams {
    int32_t srcW, srcH, dstW, dstH;
    int32_t chrSrcW, chrSrcH, chrDstW, chrDstH;
    int32_t chrSrcHSub, chrSrcVSub, chrDstHSub, chrDstVSub;
    int32_t srcKind, dstKind;
    int32_t dst_mono_white;   // DSTK_MONO: bytes are stored inverted (monowhite)
    int32_t mono_y16;         // DSTK_MONO: store the vertically scaled luma values as 16-bit words (input of sws_k_ed_mono) instead of bits
    int32_t srcBpc, dstBpc;
    int32_t src_depth;        // bits per source component
    int32_t src_shift;        // right shift of p010/p012-style sources (input.c:950-1008)
    int32_t hshift;           // hscale output shift (swscale.c:69-159)
    int32_t hclip;            // (1<<15)-1 or (1<<19)-1
    int32_t wide;             // 1 -> 19-bit int32 intermediates, 0 -> 15-bit int16
    int32_t dst_bits, dst_shift; // writer depth / left shift (p010: 10, 6)
    int32_t uv_swap_src, uv_swap_dst; // nv21-style chroma order
    int32_t u_plane_src, v_plane_src, u_plane_dst, v_plane_dst;
    int32_t should_dither;    // source is >8 bit (swscale.c:292-293)
    int32_t full_chr;         // SWS_FULL_CHR_H_INT writers
    int32_t src_pix_step;     // packed RGB: 3 or 4
    int32_t src_r_pos, src_g_pos, src_b_pos; // byte offsets of R,G,B in a packed source pixel
    int32_t chr_half;         // RGB source: chroma from averaged pixel pairs (chrSrcHSubSample)
    // filters (device pointers)
    const int16_t *hLumF, *hChrF, *vLumF, *vChrF;
    const int32_t *hLumPos, *hChrPos, *vLumPos, *vChrPos;
    int32_t hLumFs, hChrFs, vLumFs, vChrFs;
    // colour
    int32_t rgb2yuv[9];
    SwsLutParams lut;
    // range conversion (swscale.c:163-255)
    int32_t range_active, range_to_jpeg;
    uint32_t lumCoeff, chrCoeff;
    int64_t lumOffset, chrOffset;
    // unscaled helpers
    int32_t shiftY, shiftU, shiftV;   // planarToP01x net shifts
    int32_t copy_depth_src, copy_depth_dst, copy_shift_src, copy_shift_dst, copy_shiftonly_luma;
    int32_t dither_mode;              // SwsDither for planarCopy depth reduction
    int32_t src_range;
    // SWS_FAST_BILINEAR with 8-bit sources and <= 14-bit intermediates: ff_hyscale_fast_c / ff_hcscale_fast_c
    // (hscale_fast_bilinear.c:23-55) replace the polyphase horizontal stage
    int32_t fast_bilinear, lumXInc, chrXInc;
    // gray on either side: chroma is never h-scaled (needs_hcscale == 0, swscale.c:692-694); the vertical stage sees the
    // ring buffer's initial value (fill_ones, slice.c:190-208): 1 << 14 (15-bit lines) or 1 << 18 (19-bit lines)
    int32_t no_chroma;
    // alpha: both formats carry one (needAlpha, utils.c:1746): plane 3 / the A byte is h-scaled with the LUMA filter and
    // written by the planar or packed writers; dst_alpha_fill: the destination has an alpha plane the source cannot feed
    int32_t need_alpha, src_a_pos, dst_alpha_fill;
    int32_t s16_step, s16_r, s16_g, s16_b, d16_step, d16_r, d16_g, d16_b;   // 16-bit packed RGB: words per pixel and word offsets of R, G, B
    int32_t shi_step[4], shi_off[4], shi_shift[4], shi_mask[4];   // SRCK_PACKEDHI: per component (Y, U, V, A) byte step / offset, right shift, mask
    int32_t dhi_unit_bytes, dhi_bits, dhi_sub, dhi_alpha, dhi_bitpos[5];   // DSTK_PACKEDHI: bytes per unit (pixel, or pixel pair when dhi_sub), sample depth, bit position of Y, U, V, Y2, A inside the unit
    uint32_t dhi_fill_lo, dhi_fill_hi;   // constant bits of a unit (the X fields)
    int32_t sf_half, sf_layout, sf_step, sf_a_off;   // SRCK_FLOATX: half-float elements; 0 packed RGB(A), 1 gray (+ alpha), 2 planar G B R (A); bytes per pixel; byte offset of the alpha element
    int32_t s444_step, s444_y, s444_u, s444_v, s444_a, d444_step, d444_y, d444_u, d444_v, d444_a;   // packed 4:4:4: pixel step and byte offsets
    int32_t s422_y, s422_u, s422_v, d422_y, d422_u, d422_v;   // packed 4:2:2: byte offsets of Y0, U, V inside a 4-byte pixel pair
    int32_t s16_maskr, s16_maskg, s16_maskb, s16_rsh, s16_gsh, s16_bsh, s16_S, s16_is565;   // SRCK_RGB16 reader rows (input.c:396-401)
    int32_t src_alpha_opaque;   // rgb0-style source feeding a real alpha channel: its X byte counts as 255 (swscale.c:1106-1124)
    // "virtual source lines" of the two-pass path (dev_plan*.hip build_vlines): row r of pass 1 is picture line vlines[2r] read with the
    // side term vlines[2r + 1]; luma / alpha rows first (srcH of them), chroma rows from entry nVL on.  vline_mode 1: the side term is the
    // number of gamma table passes the line has seen (gamma_tab, gamma.c:31-58); 2: the plane-0 row of a planar RGB chroma line
    // (hscale.c:211-225).  Null for every context whose result does not depend on the reference's line schedule.
    const int32_t *vlines;
    const uint16_t *gamma_tab;
    int32_t nVL, vline_mode;
};
