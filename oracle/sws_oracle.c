/*
 * sws_oracle.c -- TEST INFRASTRUCTURE ONLY (see sws_oracle.h).
 *
 * Scalar CPU restatement of the librempeg libswscale legacy path
 * (sws_getContext / sws_setColorspaceDetails / sws_scale) for the pixel
 * formats on the hot path.  Written from the reference's arithmetic, not
 * copied: whole-frame intermediates replace the reference's line ring buffer
 * (libswscale/slice.c), one generic routine per stage replaces the macro
 * families.  Each routine cites the reference lines whose results it must
 * reproduce bit for bit.
 */
#include "sws_oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* small helpers (libavutil/common.h, libavutil/macros.h)              */
/* ------------------------------------------------------------------ */
#define ORMIN(a, b) ((a) < (b) ? (a) : (b))
#define ORMAX(a, b) ((a) > (b) ? (a) : (b))
#define ORABS(a) ((a) >= 0 ? (a) : -(a))
#define CEIL_RSHIFT(a, b) (-((-(a)) >> (b)))

static int64_t rounded_div(int64_t a, int64_t b) /* ROUNDED_DIV, libavutil/common.h:58 */
{
    return (a >= 0 ? a + (b >> 1) : a - (b >> 1)) / b;
}
static int ilog2(unsigned v) /* av_log2: floor(log2(v|1)) */
{
    int n = 0;
    v |= 1;
    while (v >>= 1) n++;
    return n;
}
static int clip_u8(int a) { return a < 0 ? 0 : a > 255 ? 255 : a; }
static int clip_u16(int a) { return a < 0 ? 0 : a > 65535 ? 65535 : a; }
static int clip_uintp2(int a, int p)
{
    if (a & ~((1 << p) - 1)) return (~a) >> 31 & ((1 << p) - 1);
    return a;
}

static int clip_i16(int a) { return a < -32768 ? -32768 : a > 32767 ? 32767 : a; }

/* ------------------------------------------------------------------ */
/* pixel format descriptors (libavutil/pixdesc.c rows, subset)         */
/* ------------------------------------------------------------------ */
#define PF_BE     (1 << 0)
#define PF_PLANAR (1 << 4)
#define PF_RGB    (1 << 5)
#define PF_ALPHA  (1 << 7)
#define PF_FLOAT  (1 << 9)
#define PF_PAL    (1 << 1)
#define PF_BAYER  (1 << 8)

typedef struct { int plane, step, offset, shift, depth; } Comp;
typedef struct {
    int fmt; const char *name; int nb; int lw, lh; Comp c[4]; unsigned flags;
} Desc;

static const Desc descs[] = {
    { ORF_YUV420P, "yuv420p", 3, 1, 1, {{0,1,0,0,8},{1,1,0,0,8},{2,1,0,0,8}}, PF_PLANAR },
    { ORF_YUVJ420P,"yuvj420p",3, 1, 1, {{0,1,0,0,8},{1,1,0,0,8},{2,1,0,0,8}}, PF_PLANAR },
    { ORF_YUV422P, "yuv422p", 3, 1, 0, {{0,1,0,0,8},{1,1,0,0,8},{2,1,0,0,8}}, PF_PLANAR },
    { ORF_YUV444P, "yuv444p", 3, 0, 0, {{0,1,0,0,8},{1,1,0,0,8},{2,1,0,0,8}}, PF_PLANAR },
    { ORF_NV12,    "nv12",    3, 1, 1, {{0,1,0,0,8},{1,2,0,0,8},{1,2,1,0,8}}, PF_PLANAR },
    { ORF_NV21,    "nv21",    3, 1, 1, {{0,1,0,0,8},{1,2,1,0,8},{1,2,0,0,8}}, PF_PLANAR },
    { ORF_YUV420P10LE, "yuv420p10le", 3, 1, 1, {{0,2,0,0,10},{1,2,0,0,10},{2,2,0,0,10}}, PF_PLANAR },
    { ORF_YUV444P10LE, "yuv444p10le", 3, 0, 0, {{0,2,0,0,10},{1,2,0,0,10},{2,2,0,0,10}}, PF_PLANAR },
    { ORF_YUV420P16LE, "yuv420p16le", 3, 1, 1, {{0,2,0,0,16},{1,2,0,0,16},{2,2,0,0,16}}, PF_PLANAR },
    { ORF_YUV444P16LE, "yuv444p16le", 3, 0, 0, {{0,2,0,0,16},{1,2,0,0,16},{2,2,0,0,16}}, PF_PLANAR },
    { ORF_P010LE,  "p010le",  3, 1, 1, {{0,2,0,6,10},{1,4,0,6,10},{1,4,2,6,10}}, PF_PLANAR },
#define PL8(F, N, LW, LH)      { F, N, 3, LW, LH, {{0,1,0,0,8},{1,1,0,0,8},{2,1,0,0,8}}, PF_PLANAR }
#define PLN(F, N, LW, LH, D)   { F, N, 3, LW, LH, {{0,2,0,0,D},{1,2,0,0,D},{2,2,0,0,D}}, PF_PLANAR }
#define SP8(F, N, LW, LH, UO)  { F, N, 3, LW, LH, {{0,1,0,0,8},{1,2,UO,0,8},{1,2,1-(UO),0,8}}, PF_PLANAR }
#define SPN(F, N, LW, LH, D)   { F, N, 3, LW, LH, {{0,2,0,16-(D),D},{1,4,0,16-(D),D},{1,4,2,16-(D),D}}, PF_PLANAR }
    { ORF_RGB48LE,  "rgb48le",  3, 0, 0, {{0,6,0,0,16},{0,6,2,0,16},{0,6,4,0,16}}, PF_RGB },
    { ORF_BGR48LE,  "bgr48le",  3, 0, 0, {{0,6,4,0,16},{0,6,2,0,16},{0,6,0,0,16}}, PF_RGB },
    { ORF_RGBA64LE, "rgba64le", 4, 0, 0, {{0,8,0,0,16},{0,8,2,0,16},{0,8,4,0,16},{0,8,6,0,16}}, PF_RGB | PF_ALPHA },
    { ORF_BGRA64LE, "bgra64le", 4, 0, 0, {{0,8,4,0,16},{0,8,2,0,16},{0,8,0,0,16},{0,8,6,0,16}}, PF_RGB | PF_ALPHA },
    { ORF_YUYV422, "yuyv422", 3, 1, 0, {{0,2,0,0,8},{0,4,1,0,8},{0,4,3,0,8}}, 0 },
    { ORF_UYVY422, "uyvy422", 3, 1, 0, {{0,2,1,0,8},{0,4,0,0,8},{0,4,2,0,8}}, 0 },
    { ORF_YVYU422, "yvyu422", 3, 1, 0, {{0,2,0,0,8},{0,4,3,0,8},{0,4,1,0,8}}, 0 },
#define PLA(F, N, LW, LH)      { F, N, 4, LW, LH, {{0,1,0,0,8},{1,1,0,0,8},{2,1,0,0,8},{3,1,0,0,8}}, PF_PLANAR | PF_ALPHA }
#define PLAN_(F, N, LW, LH, D) { F, N, 4, LW, LH, {{0,2,0,0,D},{1,2,0,0,D},{2,2,0,0,D},{3,2,0,0,D}}, PF_PLANAR | PF_ALPHA }
    PLAN_(ORF_YUVA420P9LE, "yuva420p9le", 1, 1, 9),
    PLAN_(ORF_YUVA420P10LE, "yuva420p10le", 1, 1, 10),
    PLAN_(ORF_YUVA420P16LE, "yuva420p16le", 1, 1, 16),
    PLAN_(ORF_YUVA422P9LE, "yuva422p9le", 1, 0, 9),
    PLAN_(ORF_YUVA422P10LE, "yuva422p10le", 1, 0, 10),
    PLAN_(ORF_YUVA422P12LE, "yuva422p12le", 1, 0, 12),
    PLAN_(ORF_YUVA422P16LE, "yuva422p16le", 1, 0, 16),
    PLAN_(ORF_YUVA444P9LE, "yuva444p9le", 0, 0, 9),
    PLAN_(ORF_YUVA444P10LE, "yuva444p10le", 0, 0, 10),
    PLAN_(ORF_YUVA444P12LE, "yuva444p12le", 0, 0, 12),
    PLAN_(ORF_YUVA444P16LE, "yuva444p16le", 0, 0, 16),
    PLA(ORF_YUVA420P, "yuva420p", 1, 1), PLA(ORF_YUVA422P, "yuva422p", 1, 0), PLA(ORF_YUVA444P, "yuva444p", 0, 0),
    PL8(ORF_YUV410P, "yuv410p", 2, 2), PL8(ORF_YUV411P, "yuv411p", 2, 0), PL8(ORF_YUV440P, "yuv440p", 0, 1),
    PL8(ORF_YUVJ422P, "yuvj422p", 1, 0), PL8(ORF_YUVJ444P, "yuvj444p", 0, 0), PL8(ORF_YUVJ440P, "yuvj440p", 0, 1),
    PLN(ORF_YUV420P9LE, "yuv420p9le", 1, 1, 9), PLN(ORF_YUV422P9LE, "yuv422p9le", 1, 0, 9), PLN(ORF_YUV444P9LE, "yuv444p9le", 0, 0, 9),
    PLN(ORF_YUV422P10LE, "yuv422p10le", 1, 0, 10), PLN(ORF_YUV440P10LE, "yuv440p10le", 0, 1, 10),
    PLN(ORF_YUV420P12LE, "yuv420p12le", 1, 1, 12), PLN(ORF_YUV422P12LE, "yuv422p12le", 1, 0, 12),
    PLN(ORF_YUV444P12LE, "yuv444p12le", 0, 0, 12), PLN(ORF_YUV440P12LE, "yuv440p12le", 0, 1, 12),
    PLN(ORF_YUV420P14LE, "yuv420p14le", 1, 1, 14), PLN(ORF_YUV422P14LE, "yuv422p14le", 1, 0, 14), PLN(ORF_YUV444P14LE, "yuv444p14le", 0, 0, 14),
    PLN(ORF_YUV422P16LE, "yuv422p16le", 1, 0, 16),
    SP8(ORF_NV16, "nv16", 1, 0, 0), SP8(ORF_NV24, "nv24", 0, 0, 0), SP8(ORF_NV42, "nv42", 0, 0, 1),
    SPN(ORF_P210LE, "p210le", 1, 0, 10), SPN(ORF_P410LE, "p410le", 0, 0, 10),
    SPN(ORF_P012LE, "p012le", 1, 1, 12), SPN(ORF_P212LE, "p212le", 1, 0, 12), SPN(ORF_P412LE, "p412le", 0, 0, 12),
    SPN(ORF_P016LE, "p016le", 1, 1, 16), SPN(ORF_P216LE, "p216le", 1, 0, 16), SPN(ORF_P416LE, "p416le", 0, 0, 16),
    { ORF_RGB24,   "rgb24",   3, 0, 0, {{0,3,0,0,8},{0,3,1,0,8},{0,3,2,0,8}}, PF_RGB },
    { ORF_BGR24,   "bgr24",   3, 0, 0, {{0,3,2,0,8},{0,3,1,0,8},{0,3,0,0,8}}, PF_RGB },
    { ORF_ARGB,    "argb",    4, 0, 0, {{0,4,1,0,8},{0,4,2,0,8},{0,4,3,0,8},{0,4,0,0,8}}, PF_RGB | PF_ALPHA },
    { ORF_RGBA,    "rgba",    4, 0, 0, {{0,4,0,0,8},{0,4,1,0,8},{0,4,2,0,8},{0,4,3,0,8}}, PF_RGB | PF_ALPHA },
    { ORF_ABGR,    "abgr",    4, 0, 0, {{0,4,3,0,8},{0,4,2,0,8},{0,4,1,0,8},{0,4,0,0,8}}, PF_RGB | PF_ALPHA },
    { ORF_BGRA,    "bgra",    4, 0, 0, {{0,4,2,0,8},{0,4,1,0,8},{0,4,0,0,8},{0,4,3,0,8}}, PF_RGB | PF_ALPHA },
    { ORF_0RGB,    "0rgb",    3, 0, 0, {{0,4,1,0,8},{0,4,2,0,8},{0,4,3,0,8}}, PF_RGB },
    { ORF_RGB0,    "rgb0",    3, 0, 0, {{0,4,0,0,8},{0,4,1,0,8},{0,4,2,0,8}}, PF_RGB },
    { ORF_0BGR,    "0bgr",    3, 0, 0, {{0,4,3,0,8},{0,4,2,0,8},{0,4,1,0,8}}, PF_RGB },
    { ORF_BGR0,    "bgr0",    3, 0, 0, {{0,4,2,0,8},{0,4,1,0,8},{0,4,0,0,8}}, PF_RGB },
    { ORF_GBRP,    "gbrp",    3, 0, 0, {{2,1,0,0,8},{0,1,0,0,8},{1,1,0,0,8}}, PF_PLANAR | PF_RGB },
    { ORF_GRAY8, "gray", 1, 0, 0, {{0,1,0,0,8}}, 0 },
#define GRAYN(F, N, D) { F, N, 1, 0, 0, {{0,2,0,0,D}}, 0 }
    GRAYN(ORF_GRAY9LE, "gray9le", 9), GRAYN(ORF_GRAY10LE, "gray10le", 10), GRAYN(ORF_GRAY12LE, "gray12le", 12),
    GRAYN(ORF_GRAY14LE, "gray14le", 14), GRAYN(ORF_GRAY16LE, "gray16le", 16),
#define GBRN(F, N, D) { F, N, 3, 0, 0, {{2,2,0,0,D},{0,2,0,0,D},{1,2,0,0,D}}, PF_PLANAR | PF_RGB }
    GBRN(ORF_GBRP9LE, "gbrp9le", 9), GBRN(ORF_GBRP10LE, "gbrp10le", 10), GBRN(ORF_GBRP12LE, "gbrp12le", 12),
    GBRN(ORF_GBRP14LE, "gbrp14le", 14), GBRN(ORF_GBRP16LE, "gbrp16le", 16),
    { ORF_GBRPF32LE, "gbrpf32le", 3, 0, 0, {{2,4,0,0,32},{0,4,0,0,32},{1,4,0,0,32}}, PF_PLANAR | PF_RGB | PF_FLOAT },
    /* planar RGB + alpha plane (pixdesc.c: gbrap, gbrap10/12/14/16, gbrapf32) */
    { ORF_GBRAP, "gbrap", 4, 0, 0, {{2,1,0,0,8},{0,1,0,0,8},{1,1,0,0,8},{3,1,0,0,8}}, PF_PLANAR | PF_RGB | PF_ALPHA },
#define GBRAN(F, N, D) { F, N, 4, 0, 0, {{2,2,0,0,D},{0,2,0,0,D},{1,2,0,0,D},{3,2,0,0,D}}, PF_PLANAR | PF_RGB | PF_ALPHA }
    GBRAN(ORF_GBRAP10LE, "gbrap10le", 10), GBRAN(ORF_GBRAP12LE, "gbrap12le", 12), GBRAN(ORF_GBRAP14LE, "gbrap14le", 14), GBRAN(ORF_GBRAP16LE, "gbrap16le", 16),
    { ORF_GBRAPF32LE, "gbrapf32le", 4, 0, 0, {{2,4,0,0,32},{0,4,0,0,32},{1,4,0,0,32},{3,4,0,0,32}}, PF_PLANAR | PF_RGB | PF_FLOAT | PF_ALPHA },
    /* packed YUV with 10..16-bit samples (pixdesc.c:239-262, :2327-2350, :2973-3090, :3248-3270); the X fields are not components */
    { ORF_Y210LE, "y210le", 3, 1, 0, {{0,4,0,6,10},{0,8,2,6,10},{0,8,6,6,10}}, 0 },
    { ORF_Y212LE, "y212le", 3, 1, 0, {{0,4,0,4,12},{0,8,2,4,12},{0,8,6,4,12}}, 0 },
    { ORF_Y216LE, "y216le", 3, 1, 0, {{0,4,0,0,16},{0,8,2,0,16},{0,8,6,0,16}}, 0 },
    { ORF_XV30LE, "xv30le", 3, 0, 0, {{0,4,1,2,10},{0,4,0,0,10},{0,4,2,4,10}}, 0 },
    { ORF_V30XLE, "v30xle", 3, 0, 0, {{0,4,1,4,10},{0,4,0,2,10},{0,4,2,6,10}}, 0 },
    PL8(ORF_YUVJ411P, "yuvj411p", 2, 0),
    { ORF_NV20LE, "nv20le", 3, 1, 0, {{0,2,0,0,10},{1,4,0,0,10},{1,4,2,0,10}}, PF_PLANAR },
    { ORF_GBRP10MSBLE, "gbrp10msble", 3, 0, 0, {{2,2,0,6,10},{0,2,0,6,10},{1,2,0,6,10}}, PF_PLANAR | PF_RGB },
    { ORF_GBRP12MSBLE, "gbrp12msble", 3, 0, 0, {{2,2,0,4,12},{0,2,0,4,12},{1,2,0,4,12}}, PF_PLANAR | PF_RGB },
    { ORF_YA8, "ya8", 2, 0, 0, {{0,2,0,0,8},{0,2,1,0,8}}, PF_ALPHA },
    { ORF_YA16LE, "ya16le", 2, 0, 0, {{0,4,0,0,16},{0,4,2,0,16}}, PF_ALPHA },
    { ORF_GRAYF32LE, "grayf32le", 1, 0, 0, {{0,4,0,0,32}}, PF_FLOAT },
    { ORF_MONOWHITE, "monow", 1, 0, 0, {{0,1,0,0,1}}, PF_RGB },   /* 1 bit per pixel, MSB first; isAnyRGB() counts them in (swscale_internal.h:876-882) */
    { ORF_MONOBLACK, "monob", 1, 0, 0, {{0,1,0,7,1}}, PF_RGB },
    /* bayer mosaics (pixdesc.c: one plane, "RGB" components of depth 2 / 4 / 2 per sample for the 8-bit ones, 4 / 8 / 4 for the 16-bit ones) */
#define BAYER8(F, N)  { F, N, 3, 0, 0, {{0,1,0,0,2},{0,1,0,0,4},{0,1,0,0,2}}, PF_RGB | PF_BAYER }
#define BAYER16(F, N) { F, N, 3, 0, 0, {{0,2,0,0,4},{0,2,0,0,8},{0,2,0,0,4}}, PF_RGB | PF_BAYER }
    BAYER8(ORF_BAYER_BGGR8, "bayer_bggr8"), BAYER8(ORF_BAYER_RGGB8, "bayer_rggb8"), BAYER8(ORF_BAYER_GBRG8, "bayer_gbrg8"), BAYER8(ORF_BAYER_GRBG8, "bayer_grbg8"),
    BAYER16(ORF_BAYER_BGGR16LE, "bayer_bggr16le"), BAYER16(ORF_BAYER_RGGB16LE, "bayer_rggb16le"), BAYER16(ORF_BAYER_GBRG16LE, "bayer_gbrg16le"), BAYER16(ORF_BAYER_GRBG16LE, "bayer_grbg16le"),
    { ORF_PAL8, "pal8", 1, 0, 0, {{0,1,0,0,8}}, PF_PAL | PF_ALPHA },   /* pixdesc.c: one index plane + the palette in data[1] */
    /* float and half-float sources (pixdesc.c:2583-2717, :2932-2971, :3108-3119), the packed 4:1:1 source (:484-494): inputs only */
    { ORF_RGBF32LE, "rgbf32le", 3, 0, 0, {{0,12,0,0,32},{0,12,4,0,32},{0,12,8,0,32}}, PF_RGB | PF_FLOAT },
    { ORF_RGBF16LE, "rgbf16le", 3, 0, 0, {{0,6,0,0,16},{0,6,2,0,16},{0,6,4,0,16}}, PF_RGB | PF_FLOAT },
    { ORF_RGBAF16LE, "rgbaf16le", 4, 0, 0, {{0,8,0,0,16},{0,8,2,0,16},{0,8,4,0,16},{0,8,6,0,16}}, PF_RGB | PF_FLOAT | PF_ALPHA },
    { ORF_GRAYF16LE, "grayf16le", 1, 0, 0, {{0,2,0,0,16}}, PF_FLOAT },
    { ORF_YAF32LE, "yaf32le", 2, 0, 0, {{0,8,0,0,32},{0,8,4,0,32}}, PF_FLOAT | PF_ALPHA },
    { ORF_YAF16LE, "yaf16le", 2, 0, 0, {{0,4,0,0,16},{0,4,2,0,16}}, PF_FLOAT | PF_ALPHA },
    { ORF_GBRPF16LE, "gbrpf16le", 3, 0, 0, {{2,2,0,0,16},{0,2,0,0,16},{1,2,0,0,16}}, PF_PLANAR | PF_RGB | PF_FLOAT },
    { ORF_GBRAPF16LE, "gbrapf16le", 4, 0, 0, {{2,2,0,0,16},{0,2,0,0,16},{1,2,0,0,16},{3,2,0,0,16}}, PF_PLANAR | PF_RGB | PF_FLOAT | PF_ALPHA },
    { ORF_UYYVYY411, "uyyvyy411", 3, 2, 0, {{0,4,1,0,8},{0,6,0,0,8},{0,6,3,0,8}}, 0 },
    /* 8 / 4 bpp RGB (pixdesc.c:495-566); rgb4 / bgr4 are bit streams of two pixels per byte, first pixel in the low nibble as the
     * writers store them (output.c:1778-1780) */
    { ORF_BGR8, "bgr8", 3, 0, 0, {{0,1,0,0,3},{0,1,0,3,3},{0,1,0,6,2}}, PF_RGB },
    { ORF_BGR4, "bgr4", 3, 0, 0, {{0,4,3,0,1},{0,4,1,0,2},{0,4,0,0,1}}, PF_RGB },
    { ORF_BGR4_BYTE, "bgr4_byte", 3, 0, 0, {{0,1,0,0,1},{0,1,0,1,2},{0,1,0,3,1}}, PF_RGB },
    { ORF_RGB8, "rgb8", 3, 0, 0, {{0,1,0,5,3},{0,1,0,2,3},{0,1,0,0,2}}, PF_RGB },
    { ORF_RGB4, "rgb4", 3, 0, 0, {{0,4,0,0,1},{0,4,1,0,2},{0,4,3,0,1}}, PF_RGB },
    { ORF_RGB4_BYTE, "rgb4_byte", 3, 0, 0, {{0,1,0,3,1},{0,1,0,1,2},{0,1,0,0,1}}, PF_RGB },
    { ORF_XYZ12LE, "xyz12le", 3, 0, 0, {{0,6,0,4,12},{0,6,2,4,12},{0,6,4,4,12}}, 0 },   /* only ever seen before handle_xyz() */
    { ORF_X2RGB10LE, "x2rgb10le", 3, 0, 0, {{0,4,2,4,10},{0,4,1,2,10},{0,4,0,0,10}}, PF_RGB },
    { ORF_X2BGR10LE, "x2bgr10le", 3, 0, 0, {{0,4,0,0,10},{0,4,1,2,10},{0,4,2,4,10}}, PF_RGB },
    { ORF_XV36LE, "xv36le", 3, 0, 0, {{0,8,2,4,12},{0,8,0,4,12},{0,8,4,4,12}}, 0 },
    { ORF_XV48LE, "xv48le", 3, 0, 0, {{0,8,2,0,16},{0,8,0,0,16},{0,8,4,0,16}}, 0 },
    { ORF_AYUV64LE, "ayuv64le", 4, 0, 0, {{0,8,2,0,16},{0,8,4,0,16},{0,8,6,0,16},{0,8,0,0,16}}, PF_ALPHA },
    /* packed 4:4:4, 8 bit (pixdesc.c:2290-2324, :2895-2917) */
    { ORF_VYU444, "vyu444", 3, 0, 0, {{0,3,1,0,8},{0,3,2,0,8},{0,3,0,0,8}}, 0 },
    { ORF_UYVA, "uyva", 4, 0, 0, {{0,4,1,0,8},{0,4,0,0,8},{0,4,2,0,8},{0,4,3,0,8}}, PF_ALPHA },
    { ORF_AYUV, "ayuv", 4, 0, 0, {{0,4,1,0,8},{0,4,2,0,8},{0,4,3,0,8},{0,4,0,0,8}}, PF_ALPHA },
    { ORF_VUYA, "vuya", 4, 0, 0, {{0,4,2,0,8},{0,4,1,0,8},{0,4,0,0,8},{0,4,3,0,8}}, PF_ALPHA },
    { ORF_VUYX, "vuyx", 4, 0, 0, {{0,4,2,0,8},{0,4,1,0,8},{0,4,0,0,8},{0,4,3,0,8}}, 0 },
    /* planar 4:4:4 with the samples in the high bits (pixdesc.c yuv444p10msb / yuv444p12msb) */
    { ORF_YUV444P10MSBLE, "yuv444p10msble", 3, 0, 0, {{0,2,0,6,10},{1,2,0,6,10},{2,2,0,6,10}}, PF_PLANAR },
    { ORF_YUV444P12MSBLE, "yuv444p12msble", 3, 0, 0, {{0,2,0,4,12},{1,2,0,4,12},{2,2,0,4,12}}, PF_PLANAR },
    /* 16 bits per pixel packed RGB (libavutil/pixdesc.c:1229-1420) */
    { ORF_RGB565LE, "rgb565le", 3, 0, 0, {{0,2,1,3,5},{0,2,0,5,6},{0,2,0,0,5}}, PF_RGB },
    { ORF_RGB555LE, "rgb555le", 3, 0, 0, {{0,2,1,2,5},{0,2,0,5,5},{0,2,0,0,5}}, PF_RGB },
    { ORF_RGB444LE, "rgb444le", 3, 0, 0, {{0,2,1,0,4},{0,2,0,4,4},{0,2,0,0,4}}, PF_RGB },
    { ORF_BGR565LE, "bgr565le", 3, 0, 0, {{0,2,0,0,5},{0,2,0,5,6},{0,2,1,3,5}}, PF_RGB },
    { ORF_BGR555LE, "bgr555le", 3, 0, 0, {{0,2,0,0,5},{0,2,0,5,5},{0,2,1,2,5}}, PF_RGB },
    { ORF_BGR444LE, "bgr444le", 3, 0, 0, {{0,2,0,0,4},{0,2,0,4,4},{0,2,1,0,4}}, PF_RGB },
};
static int isPackedHi(int f) { return f == ORF_Y210LE || f == ORF_Y212LE || f == ORF_Y216LE || f == ORF_XV30LE || f == ORF_V30XLE || f == ORF_XV36LE || f == ORF_XV48LE || f == ORF_AYUV64LE; }
static int isPacked444(int f) { return f == ORF_VYU444 || f == ORF_UYVA || f == ORF_AYUV || f == ORF_VUYA || f == ORF_VUYX; }
static int isRGB30(int f) { return f == ORF_X2RGB10LE || f == ORF_X2BGR10LE; }
static int isRGB8class(int f) { return f == ORF_RGB8 || f == ORF_BGR8 || f == ORF_RGB4_BYTE || f == ORF_BGR4_BYTE; }   /* one byte per pixel */
static int isRGB4bits(int f) { return f == ORF_RGB4 || f == ORF_BGR4; }                                                   /* two pixels per byte */
static int isRGB16(int f) { return f == ORF_RGB565LE || f == ORF_RGB555LE || f == ORF_RGB444LE || f == ORF_BGR565LE || f == ORF_BGR555LE || f == ORF_BGR444LE; }

static const Desc *desc_get(int fmt)
{
    for (size_t i = 0; i < sizeof(descs) / sizeof(descs[0]); i++)
        if (descs[i].fmt == fmt) return &descs[i];
    return NULL;
}

/* big-endian formats: {BE, LE} values of libavutil/pixfmt.h.  The oracle converts through the little-endian twin: a BE source is
 * byte-swapped into a scratch copy first, a BE destination is byte-swapped in place afterwards (the reference's BE readers and
 * writers are the LE ones behind AV_RB16 / AV_WB16: input.c:608-629, output.c output_pixel macros); converter selection follows
 * the reference's rules, which only name a byte order for planarToP01xWrapper / planar8ToP01xleWrapper (native-endian only). */
static const int be_pairs[][2] = {
    { ORF_GBRAP10BE, ORF_GBRAP10LE }, { ORF_GBRAP12BE, ORF_GBRAP12LE }, { ORF_GBRAP14BE, ORF_GBRAP14LE }, { ORF_GBRAP16BE, ORF_GBRAP16LE }, { ORF_GBRAPF32BE, ORF_GBRAPF32LE },
    { ORF_XV36BE, ORF_XV36LE }, { ORF_XV48BE, ORF_XV48LE }, { ORF_AYUV64BE, ORF_AYUV64LE },
    { ORF_YUVA420P9BE, ORF_YUVA420P9LE }, { ORF_YUVA420P10BE, ORF_YUVA420P10LE }, { ORF_YUVA420P16BE, ORF_YUVA420P16LE }, { ORF_YUVA422P9BE, ORF_YUVA422P9LE }, { ORF_YUVA422P10BE, ORF_YUVA422P10LE }, { ORF_YUVA422P12BE, ORF_YUVA422P12LE }, { ORF_YUVA422P16BE, ORF_YUVA422P16LE }, { ORF_YUVA444P9BE, ORF_YUVA444P9LE }, { ORF_YUVA444P10BE, ORF_YUVA444P10LE }, { ORF_YUVA444P12BE, ORF_YUVA444P12LE }, { ORF_YUVA444P16BE, ORF_YUVA444P16LE },
    { ORF_BAYER_BGGR16BE, ORF_BAYER_BGGR16LE }, { ORF_BAYER_RGGB16BE, ORF_BAYER_RGGB16LE }, { ORF_BAYER_GBRG16BE, ORF_BAYER_GBRG16LE }, { ORF_BAYER_GRBG16BE, ORF_BAYER_GRBG16LE },
    { ORF_RGBF32BE, ORF_RGBF32LE }, { ORF_RGBF16BE, ORF_RGBF16LE }, { ORF_RGBAF16BE, ORF_RGBAF16LE }, { ORF_GRAYF16BE, ORF_GRAYF16LE }, { ORF_YAF32BE, ORF_YAF32LE },
    { ORF_YAF16BE, ORF_YAF16LE }, { ORF_GBRPF16BE, ORF_GBRPF16LE }, { ORF_GBRAPF16BE, ORF_GBRAPF16LE },
    { ORF_YA16BE, ORF_YA16LE }, { ORF_GRAYF32BE, ORF_GRAYF32LE }, { ORF_XYZ12BE, ORF_XYZ12LE }, { ORF_NV20BE, ORF_NV20LE }, { ORF_GBRP10MSBBE, ORF_GBRP10MSBLE }, { ORF_GBRP12MSBBE, ORF_GBRP12MSBLE },
    { ORF_YUV444P10MSBBE, ORF_YUV444P10MSBLE }, { ORF_YUV444P12MSBBE, ORF_YUV444P12MSBLE },
    { ORF_RGB565BE, ORF_RGB565LE }, { ORF_RGB555BE, ORF_RGB555LE }, { ORF_RGB444BE, ORF_RGB444LE },
    { ORF_BGR565BE, ORF_BGR565LE }, { ORF_BGR555BE, ORF_BGR555LE }, { ORF_BGR444BE, ORF_BGR444LE },
    { 59, 60 } /* yuv420p9 */,
    { 61, 62 } /* yuv420p10 */,
    { 122, 123 } /* yuv420p12 */,
    { 124, 125 } /* yuv420p14 */,
    { 46, 45 } /* yuv420p16 */,
    { 69, 70 } /* yuv422p9 */,
    { 63, 64 } /* yuv422p10 */,
    { 126, 127 } /* yuv422p12 */,
    { 128, 129 } /* yuv422p14 */,
    { 48, 47 } /* yuv422p16 */,
    { 65, 66 } /* yuv444p9 */,
    { 67, 68 } /* yuv444p10 */,
    { 130, 131 } /* yuv444p12 */,
    { 132, 133 } /* yuv444p14 */,
    { 50, 49 } /* yuv444p16 */,
    { 152, 151 } /* yuv440p10 */,
    { 154, 153 } /* yuv440p12 */,
    { 172, 173 } /* gray9 */,
    { 167, 168 } /* gray10 */,
    { 165, 166 } /* gray12 */,
    { 180, 181 } /* gray14 */,
    { 29, 30 } /* gray16 */,
    { 72, 73 } /* gbrp9 */,
    { 74, 75 } /* gbrp10 */,
    { 134, 135 } /* gbrp12 */,
    { 136, 137 } /* gbrp14 */,
    { 76, 77 } /* gbrp16 */,
    { 174, 175 } /* gbrpf32 */,
    { 159, 158 } /* p010 */,
    { 210, 209 } /* p012 */,
    { 170, 169 } /* p016 */,
    { 197, 198 } /* p210 */,
    { 221, 222 } /* p212 */,
    { 201, 202 } /* p216 */,
    { 199, 200 } /* p410 */,
    { 223, 224 } /* p412 */,
    { 203, 204 } /* p416 */,
    { 34, 35 } /* rgb48 */,
    { 57, 58 } /* bgr48 */,
    { 104, 105 } /* rgba64 */,
    { 106, 107 } /* bgra64 */
};
/* the formats IS_DIFFERENT_ENDIANESS() is asked about for bswap_16bpc (swscale_unscaled.c:2560-2612), as LE twins */
static int bswap16_listed(int f)
{
    const Desc *d = desc_get(f);
    const char *n = d ? d->name : "";
    static const char *const names[] = { "bayer_bggr16le", "bayer_rggb16le", "bayer_gbrg16le", "bayer_grbg16le", "bgr444le", "bgr48le", "bgr555le", "bgr565le", "bgra64le",
        "gray9le", "gray10le", "gray12le", "gray14le", "gray16le", "ya16le", "ayuv64le", "gbrp9le", "gbrp10le", "gbrp12le", "gbrp14le", "gbrp16le", "gbrp10msble", "gbrp12msble",
        "gbrap10le", "gbrap12le", "gbrap14le", "gbrap16le", "rgb444le", "rgb48le", "rgb555le", "rgb565le", "rgba64le", "xv36le", "xv48le", "xyz12le",
        "yuv420p9le", "yuv420p10le", "yuv420p12le", "yuv420p14le", "yuv420p16le", "yuv422p9le", "yuv422p10le", "yuv422p12le", "yuv422p14le", "yuv422p16le",
        "yuv440p10le", "yuv440p12le", "yuv444p9le", "yuv444p10le", "yuv444p12le", "yuv444p14le", "yuv444p16le", "yuv444p10msble", "yuv444p12msble" };
    for (size_t i = 0; i < sizeof(names) / sizeof(names[0]); i++) if (!strcmp(n, names[i])) return 1;
    return 0;
}

static int be_twin(int *fmt)
{
    for (size_t i = 0; i < sizeof(be_pairs) / sizeof(be_pairs[0]); i++)
        if (be_pairs[i][0] == *fmt) { *fmt = be_pairs[i][1]; return 1; }
    return 0;
}
/* libswscale/swscale_internal.h:746-988 */
static int is16BPS(int f) { return desc_get(f)->c[0].depth == 16; }
static int isNBPS(int f) { int d = desc_get(f)->c[0].depth; return d >= 9 && d <= 14; }
static int isYUV(int f) { const Desc *d = desc_get(f); return !(d->flags & PF_RGB) && d->nb >= 2; }
static int isPlanarYUV(int f) { return (desc_get(f)->flags & PF_PLANAR) && isYUV(f); }
static int isSemiPlanarYUV(int f) { const Desc *d = desc_get(f); return isPlanarYUV(f) && d->c[1].plane == d->c[2].plane; }
static int isAnyRGB(int f) { return !!(desc_get(f)->flags & PF_RGB); }
static int isYA(int f) { return f == ORF_YA8 || f == ORF_YA16LE; }
static int isMono(int f) { return f == ORF_MONOWHITE || f == ORF_MONOBLACK; }
static int isGray(int f) { return desc_get(f)->nb <= 2 && !isMono(f) && !(desc_get(f)->flags & PF_PAL); }   /* swscale_internal.h:805-815 */
static int isFloat(int f) { return !!(desc_get(f)->flags & PF_FLOAT); }
static int isFloat16(int f) { return isFloat(f) && desc_get(f)->c[0].depth == 16; }   /* swscale_internal.h:890-895 */
/* the formats the reference's table (format.c legacy_format_entries) lists as inputs only */
/* usePal (swscale_internal.h:937-950) without gray8, whose grey palette only feeds palToRgbWrapper / palToGbrpWrapper: the scaler chain gives
 * the same bytes for it (tests/test_oracle_properties_extra.py) */
static int isPalSrc(int f) { return f == ORF_PAL8 || f == ORF_RGB8 || f == ORF_BGR8 || f == ORF_RGB4_BYTE || f == ORF_BGR4_BYTE; }
static int usePal(int f) { return isPalSrc(f) || f == ORF_GRAY8; }   /* swscale_internal.h:937-950: gray8 goes through the palette wrappers too (its readers are the plain ones) */
static int isBayer(int f) { return !!(desc_get(f)->flags & PF_BAYER); }
static int isInputOnly(int f) { return isBayer(f) || f == ORF_PAL8 || f == ORF_UYYVYY411 || f == ORF_RGBF32LE || f == ORF_RGBF16LE || f == ORF_RGBAF16LE || f == ORF_GRAYF16LE || f == ORF_YAF32LE || f == ORF_YAF16LE || f == ORF_GBRPF16LE || f == ORF_GBRAPF16LE; }
static int isALPHA(int f) { return !!(desc_get(f)->flags & PF_ALPHA); }
static int isPlanarRGB(int f) { return (desc_get(f)->flags & (PF_PLANAR | PF_RGB)) == (PF_PLANAR | PF_RGB); }
static int isPacked(int f) { const Desc *d = desc_get(f); return (d->nb >= 2 && !(d->flags & PF_PLANAR)) || isMono(f) || f == ORF_PAL8; }   /* swscale_internal.h:906-914 */
static int isSwappedChroma(int f)
{
    const Desc *d = desc_get(f);
    if (!isYUV(f) || d->nb < 3) return 0;
    if (!isPlanarYUV(f) || isSemiPlanarYUV(f)) return d->c[1].offset > d->c[2].offset;
    return d->c[1].plane > d->c[2].plane;
}
static int isDataInHighBits(int f)
{
    const Desc *d = desc_get(f);
    for (int i = 0; i < d->nb; i++) {
        if (!d->c[i].shift) return 0;
        if ((d->c[i].shift + d->c[i].depth) & 7) return 0;
    }
    return 1;
}
static int bits_per_pixel(const Desc *d) /* av_get_bits_per_pixel, libavutil/pixdesc.c */
{
    int bits = 0, log2_pixels = d->lw + d->lh;
    for (int c = 0; c < d->nb; c++) {
        int s = c == 1 || c == 2 ? 0 : log2_pixels;
        bits += d->c[c].depth << s;
    }
    return bits >> log2_pixels;
}

/* ------------------------------------------------------------------ */
/* context                                                            */
/* ------------------------------------------------------------------ */
enum { RY, GY, BY, RU, GU, BU, RV, GV, BV };

#define HEADROOM 512   /* YUVRGB_TABLE_HEADROOM / _LUMA_HEADROOM, swscale_internal.h */
#define TABLE_PLANE 2048

enum { UNSC_NONE = 0, UNSC_YUV2RGB, UNSC_P01X, UNSC_8_P01X, UNSC_PLANAR2NV12,
       UNSC_NV122PLANAR, UNSC_PLANARCOPY, UNSC_RGB2RGB, UNSC_RGBLOW, UNSC_PACKEDCOPY, UNSC_BGR24_YV12, UNSC_GBRP2PACKED,
       UNSC_PLANAR2NV24, UNSC_NV242PLANAR, UNSC_NV242YUV420, UNSC_YVU9_YV12, UNSC_PACKED2GBRP, UNSC_RGB30_TO_16, UNSC_RGB30_TO_GBRP, UNSC_GBRP_TO_RGB30, UNSC_YUV2MONO, UNSC_U8_TO_F32, UNSC_F32_TO_U8,
       UNSC_PLANAR2P422, UNSC_P4222PLANAR,
       UNSC_RGB16SHUFFLE, UNSC_PACKED16_TO_GBRP16, UNSC_GBRP16_TO_PACKED16, UNSC_ALPHABLEND, UNSC_PLANARRGB_PLANARRGB, UNSC_PAL2RGB, UNSC_BAYER, UNSC_BSWAP16,
       UNSC_REFUSE = -1 /* a special converter of the reference that is not restated */ };

struct OrSws {
    OrSwsOpts o;
    int src0Alpha, dst0Alpha;
    int src_xyz, dst_xyz; /* handle_xyz (utils.c:822-842): the formats were xyz12, o.src_format / o.dst_format hold rgb48le */
    int casc_child_dst_be; /* the alpha-blend cascade's second step writes the caller's (big-endian) byte order itself */
    int bswap16;          /* the conversion is bswap_16bpc (swscale_unscaled.c:545-570): or_sws_scale() runs it on the caller's planes */
    int src_be, dst_be;   /* the caller's formats were big-endian: o.src_format / o.dst_format hold the LE twins */
    int brightness, contrast, saturation;
    int srcColorspaceTable[4], dstColorspaceTable[4];
    int dstFormatBpp, srcFormatBpp;
    int chrSrcHSub, chrSrcVSub, chrDstHSub, chrDstVSub;
    int chrSrcW, chrSrcH, chrDstW, chrDstH;
    int srcBpc, dstBpc;
    int lumXInc, lumYInc, chrXInc, chrYInc;
    int unscaled_kind;
    int16_t *hLumFilter, *hChrFilter, *vLumFilter, *vChrFilter;
    int32_t *hLumFilterPos, *hChrFilterPos, *vLumFilterPos, *vChrFilterPos;
    int hLumFilterSize, hChrFilterSize, vLumFilterSize, vChrFilterSize;
    int32_t rgb2yuv[9];
    int yuv2rgb_y_offset, yuv2rgb_y_coeff, yuv2rgb_v2r, yuv2rgb_v2g, yuv2rgb_u2g, yuv2rgb_u2b;
    /* LUTs (yuv2rgb.c:717-973): yuvTable + offsets (in elements) into it */
    uint8_t *yuvTable; int lut_elem; /* 1 or 4 bytes */
    int table_rV[256 + 2 * HEADROOM], table_gU[256 + 2 * HEADROOM], table_bU[256 + 2 * HEADROOM];
    int table_gV[256 + 2 * HEADROOM];
    int has_lut;
    int range_active; uint32_t lumCoeff, chrCoeff; int64_t lumOffset, chrOffset;
    int needAlpha;
    OrSws *cascade[3]; uint8_t *casc_tmp[4]; int casc_stride[4];
    /* gamma cascade (utils.c:1461-1522): cascade[1] scales RGBA64 -> RGBA64 between two in-place table passes, cascade[2] converts to the
     * destination format from a second intermediate */
    int casc_mainindex;   /* the child sws_setColorspaceDetails() is forwarded to (utils.c:909-910): 1 for the alpha-blend cascade */
    uint32_t pal_yuv[256], pal_rgb[256];   /* ff_update_palette (swscale.c:873-951) */
    int *dither_error[3];   /* utils.c:1744-1747: dst_w + 3 zeroed ints per channel; never reset between frames or slices */
    const uint16_t *internal_gamma_tab;   /* is_internal_gamma (utils.c:1493-1497): the inverse-gamma table of the cascade's scaling step, applied by main_path() */
    int casc_gamma; uint8_t *casc_tmp2; int casc_stride2; uint16_t *gamma_tab, *inv_gamma_tab;
    int initialized;
};

static const int32_t yuv2rgb_coeffs[11][4] = { /* yuv2rgb.c:47-59 */
    { 104597, 132201, 25675, 53279 }, { 117489, 138438, 13975, 34925 },
    { 104597, 132201, 25675, 53279 }, { 104597, 132201, 25675, 53279 },
    { 104448, 132798, 24759, 53109 }, { 104597, 132201, 25675, 53279 },
    { 104597, 132201, 25675, 53279 }, { 117579, 136230, 16907, 35559 },
    { 0 }, { 110013, 140363, 12277, 42626 }, { 110013, 140363, 12277, 42626 },
};
const int *or_sws_get_coefficients(int cs)
{
    if (cs > 10 || cs < 0 || cs == 8) cs = 5;
    return yuv2rgb_coeffs[cs];
}

static const uint8_t dither_8x8_128[9][8] = { /* swscale.c:42-52 */
    {  36, 68,  60, 92,  34, 66,  58, 90 }, { 100,  4, 124, 28,  98,  2, 122, 26 },
    {  52, 84,  44, 76,  50, 82,  42, 74 }, { 116, 20, 108, 12, 114, 18, 106, 10 },
    {  32, 64,  56, 88,  38, 70,  62, 94 }, {  96,  0, 120, 24, 102,  6, 126, 30 },
    {  48, 80,  40, 72,  54, 86,  46, 78 }, { 112, 16, 104,  8, 118, 22, 110, 14 },
    {  36, 68,  60, 92,  34, 66,  58, 90 },
};
static const uint8_t pb_64[8] = { 64, 64, 64, 64, 64, 64, 64, 64 };

/* ------------------------------------------------------------------ */
/* initFilter  (libswscale/utils.c:197-612)                            */
/* ------------------------------------------------------------------ */
#define RET_CASCADE -12345

static double spline_coeff(double a, double b, double c, double d, double dist) /* utils.c:152-166 */
{
    if (dist <= 1.0) return ((d * dist + c) * dist + b) * dist + a;
    return spline_coeff(0.0, b + 2.0 * c + 3.0 * d, c + 3.0 * d, -b - 3.0 * c - 6.0 * d, dist - 1.0);
}

static int init_filter(int16_t **outFilter, int32_t **filterPos, int *outFilterSize,
                       int xInc, int srcW, int dstW, int filterAlign, int one,
                       int scaler, int flags, const double param[2], int srcPos, int dstPos,
                       const double *srcVec, int srcVecLen, int dstVecLen)
{
    int filterSize, filter2Size, minFilterSize, i, j, ret = -1;
    int64_t *filter = NULL, *filter2 = NULL;
    const int64_t fone = 1LL << (54 - ORMIN(ilog2(srcW / dstW), 8));

    *filterPos = malloc((dstW + 3) * sizeof(int32_t));

    if (ORABS(xInc - 0x10000) < 10 && srcPos == dstPos) { /* :219 unscaled */
        filterSize = 1;
        filter = calloc(dstW, sizeof(*filter));
        for (i = 0; i < dstW; i++) { filter[i] = fone; (*filterPos)[i] = i; }
    } else if (scaler == OR_SWS_POINT) { /* :229 */
        int64_t xDstInSrc;
        filterSize = 1;
        filter = malloc(dstW * sizeof(*filter));
        xDstInSrc = ((dstPos * (int64_t)xInc) >> 8) - ((srcPos * 0x8000LL) >> 7);
        for (i = 0; i < dstW; i++) {
            int xx = (int)((xDstInSrc - ((int64_t)(filterSize - 1) << 15) + (1 << 15)) >> 16);
            (*filterPos)[i] = xx; filter[i] = fone; xDstInSrc += xInc;
        }
    } else if ((xInc <= (1 << 16) && scaler == OR_SWS_AREA) || scaler == OR_SWS_FAST_BILINEAR) { /* :244 */
        int64_t xDstInSrc;
        filterSize = 2;
        filter = malloc(dstW * filterSize * sizeof(*filter));
        xDstInSrc = ((dstPos * (int64_t)xInc) >> 8) - ((srcPos * 0x8000LL) >> 7);
        for (i = 0; i < dstW; i++) {
            int xx = (int)((xDstInSrc - ((int64_t)(filterSize - 1) << 15) + (1 << 15)) >> 16);
            (*filterPos)[i] = xx;
            for (j = 0; j < filterSize; j++) {
                int64_t coeff = fone - ORABS((int64_t)xx * (1 << 16) - xDstInSrc) * (fone >> 16);
                if (coeff < 0) coeff = 0;
                filter[i * filterSize + j] = coeff;
                xx++;
            }
            xDstInSrc += xInc;
        }
    } else { /* :268 general */
        int64_t xDstInSrc;
        int sizeFactor = -1;
        switch (scaler) {
        case OR_SWS_AREA: sizeFactor = 1; break;
        case OR_SWS_BICUBIC: sizeFactor = 4; break;
        case OR_SWS_BILINEAR: sizeFactor = 2; break;
        case OR_SWS_GAUSS: sizeFactor = 8; break;
        case OR_SWS_SINC: sizeFactor = 20; break;
        case OR_SWS_SPLINE: sizeFactor = 20; break;
        case OR_SWS_X: sizeFactor = 8; break;
        case OR_SWS_LANCZOS:
            sizeFactor = param[0] != OR_SWS_PARAM_DEFAULT ? (int)ceil(2 * param[0]) : 6; break;
        }
        if (sizeFactor <= 0 || sizeFactor > 50) goto fail;

        if (xInc <= 1 << 16) filterSize = 1 + sizeFactor;
        else filterSize = 1 + (sizeFactor * srcW + dstW - 1) / dstW;
        filterSize = ORMIN(filterSize, srcW - 2);
        filterSize = ORMAX(filterSize, 1);

        filter = malloc((size_t)dstW * filterSize * sizeof(*filter));
        xDstInSrc = ((dstPos * (int64_t)xInc) >> 7) - ((srcPos * 0x10000LL) >> 7);
        for (i = 0; i < dstW; i++) {
            int xx = (int)((xDstInSrc - (filterSize - 2) * (1LL << 16)) / (1 << 17));
            (*filterPos)[i] = xx;
            for (j = 0; j < filterSize; j++) {
                int64_t d = ORABS(((int64_t)xx * (1 << 17)) - xDstInSrc) << 13;
                double floatd;
                int64_t coeff;

                if (xInc > 1 << 16) d = d * dstW / srcW;
                floatd = d * (1.0 / (1 << 30));

                if (scaler == OR_SWS_BICUBIC) { /* :312 */
                    int64_t B = (int64_t)((param[0] != OR_SWS_PARAM_DEFAULT ? param[0] : 0) * (1 << 24));
                    int64_t C = (int64_t)((param[1] != OR_SWS_PARAM_DEFAULT ? param[1] : 0.6) * (1 << 24));
                    if (d >= 1LL << 31) {
                        coeff = 0;
                    } else {
                        int64_t dd = (d * d) >> 30;
                        int64_t ddd = (dd * d) >> 30;
                        if (d < 1LL << 30)
                            coeff = (12 * (1 << 24) - 9 * B - 6 * C) * ddd +
                                    (-18 * (1 << 24) + 12 * B + 6 * C) * dd +
                                    (6 * (1 << 24) - 2 * B) * (1 << 30);
                        else
                            coeff = (-B - 6 * C) * ddd + (6 * B + 30 * C) * dd +
                                    (-12 * B - 48 * C) * d + (8 * B + 24 * C) * (1 << 30);
                    }
                    coeff /= (1LL << 54) / fone;
                } else if (scaler == OR_SWS_X) {
                    double A = param[0] != OR_SWS_PARAM_DEFAULT ? param[0] : 1.0;
                    double c = floatd < 1.0 ? cos(floatd * M_PI) : -1.0;
                    if (c < 0.0) c = -pow(-c, A); else c = pow(c, A);
                    coeff = (int64_t)((c * 0.5 + 0.5) * fone);
                } else if (scaler == OR_SWS_AREA) {
                    int64_t d2 = d - (1 << 29);
                    if (d2 * xInc < -(1LL << (29 + 16))) coeff = 1LL << (30 + 16);
                    else if (d2 * xInc < (1LL << (29 + 16))) coeff = -d2 * xInc + (1LL << (29 + 16));
                    else coeff = 0;
                    coeff *= fone >> (30 + 16);
                } else if (scaler == OR_SWS_GAUSS) {
                    double p = param[0] != OR_SWS_PARAM_DEFAULT ? param[0] : 3.0;
                    coeff = (int64_t)(exp2(-p * floatd * floatd) * fone);
                } else if (scaler == OR_SWS_SINC) {
                    coeff = (int64_t)((d ? sin(floatd * M_PI) / (floatd * M_PI) : 1.0) * fone);
                } else if (scaler == OR_SWS_LANCZOS) { /* :360 */
                    double p = param[0] != OR_SWS_PARAM_DEFAULT ? param[0] : 3.0;
                    coeff = (int64_t)((d ? sin(floatd * M_PI) * sin(floatd * M_PI / p) /
                                           (floatd * floatd * M_PI * M_PI / p) : 1.0) * fone);
                    if (floatd > p) coeff = 0;
                } else if (scaler == OR_SWS_BILINEAR) { /* :366 */
                    coeff = (1 << 30) - d;
                    if (coeff < 0) coeff = 0;
                    coeff *= fone >> 30;
                } else if (scaler == OR_SWS_SPLINE) {
                    double p = -2.196152422706632;
                    coeff = (int64_t)(spline_coeff(1.0, 0.0, p, -p - 1.0, floatd) * fone);
                } else {
                    goto fail;
                }
                filter[i * filterSize + j] = coeff;
                xx++;
            }
            xDstInSrc += 2LL * xInc;
        }
    }

    /* apply src & dst SwsVector to filter -> filter2 (:385-415): the source vector is convolved in (double products
     * accumulated into the int64 taps), the destination vector only widens the row ("FIXME dstFilter") */
    filter2Size = filterSize;
    if (srcVec) filter2Size += srcVecLen - 1;
    if (dstVecLen) filter2Size += dstVecLen - 1;
    filter2 = calloc((size_t)dstW * filter2Size, sizeof(*filter2));
    for (i = 0; i < dstW; i++) {
        if (srcVec) {
            for (int k = 0; k < srcVecLen; k++)
                for (j = 0; j < filterSize; j++)
                    filter2[(size_t)i * filter2Size + k + j] += srcVec[k] * filter[(size_t)i * filterSize + j];
        } else {
            for (j = 0; j < filterSize; j++) filter2[(size_t)i * filter2Size + j] = filter[(size_t)i * filterSize + j];
        }
        (*filterPos)[i] += (filterSize - 1) / 2 - (filter2Size - 1) / 2;
    }
    free(filter); filter = NULL;

    /* reduce, step 1 (:417-457) */
    minFilterSize = 0;
    for (i = dstW - 1; i >= 0; i--) {
        int min = filter2Size;
        int64_t cutOff = 0;
        for (j = 0; j < filter2Size; j++) {
            int k;
            cutOff += ORABS(filter2[i * filter2Size]);
            if (cutOff > 0.002 * fone) break;
            if (i < dstW - 1 && (*filterPos)[i] >= (*filterPos)[i + 1]) break;
            for (k = 1; k < filter2Size; k++)
                filter2[i * filter2Size + k - 1] = filter2[i * filter2Size + k];
            filter2[i * filter2Size + k - 1] = 0;
            (*filterPos)[i]++;
        }
        cutOff = 0;
        for (j = filter2Size - 1; j > 0; j--) {
            cutOff += ORABS(filter2[i * filter2Size + j]);
            if (cutOff > 0.002 * fone) break;
            min--;
        }
        if (min > minFilterSize) minFilterSize = min;
    }

    filterSize = (minFilterSize + (filterAlign - 1)) & (~(filterAlign - 1));
    filter = malloc((size_t)dstW * filterSize * sizeof(*filter));
    if (filterSize >= 256 * 16 / ((flags & OR_SWS_ACCURATE_RND) ? 16 : 16)) { /* :492; APCK_SIZE == 16 in the
                                                                                * C-only (ARCH_X86_64 0) build, swscale_internal.h:64-73 */
        ret = RET_CASCADE;
        goto fail;
    }
    *outFilterSize = filterSize;

    for (i = 0; i < dstW; i++) /* reduce, step 2 (:503-515) */
        for (j = 0; j < filterSize; j++) {
            filter[i * filterSize + j] = j >= filter2Size ? 0 : filter2[i * filter2Size + j];
            if ((flags & OR_SWS_BITEXACT) && j >= minFilterSize)
                filter[i * filterSize + j] = 0;
        }

    for (i = 0; i < dstW; i++) { /* fix borders (:519-560) */
        if ((*filterPos)[i] < 0) {
            for (j = 1; j < filterSize; j++) {
                int left = ORMAX(j + (*filterPos)[i], 0);
                filter[i * filterSize + left] += filter[i * filterSize + j];
                filter[i * filterSize + j] = 0;
            }
            (*filterPos)[i] = 0;
        }
        if ((*filterPos)[i] + filterSize > srcW) {
            int shift = (*filterPos)[i] + ORMIN(filterSize - srcW, 0);
            int64_t acc = 0;
            for (j = filterSize - 1; j >= 0; j--)
                if ((*filterPos)[i] + j >= srcW) {
                    acc += filter[i * filterSize + j];
                    filter[i * filterSize + j] = 0;
                }
            for (j = filterSize - 1; j >= 0; j--) {
                if (j < shift) filter[i * filterSize + j] = 0;
                else filter[i * filterSize + j] = filter[i * filterSize + j - shift];
            }
            (*filterPos)[i] -= shift;
            filter[i * filterSize + srcW - 1 - (*filterPos)[i]] += acc;
        }
    }

    *outFilter = calloc((size_t)(dstW + 3) * filterSize, sizeof(int16_t));
    for (i = 0; i < dstW; i++) { /* normalise (:568-588) */
        int64_t error = 0, sum = 0;
        for (j = 0; j < filterSize; j++) sum += filter[i * filterSize + j];
        sum = (sum + one / 2) / one;
        if (!sum) sum = 1;
        for (j = 0; j < filterSize; j++) {
            int64_t v = filter[i * filterSize + j] + error;
            int intV = (int)rounded_div(v, sum);
            (*outFilter)[i * filterSize + j] = (int16_t)intV;
            error = v - intV * sum;
        }
    }
    (*filterPos)[dstW + 0] = (*filterPos)[dstW + 1] = (*filterPos)[dstW + 2] = (*filterPos)[dstW - 1];
    for (i = 0; i < filterSize; i++) {
        int k = (dstW - 1) * filterSize + i;
        (*outFilter)[k + 1 * filterSize] = (*outFilter)[k + 2 * filterSize] =
        (*outFilter)[k + 3 * filterSize] = (*outFilter)[k];
    }
    ret = 0;
fail:
    free(filter); free(filter2);
    return ret;
}

/* ------------------------------------------------------------------ */
/* colour tables                                                       */
/* ------------------------------------------------------------------ */
static int round_to_int16(int64_t f) /* yuv2rgb.c:705-715; returns value as int16 */
{
    int r = (int)((f + (1 << 15)) >> 16);
    if (r < -0x7FFF) return (int16_t)0x8000;
    if (r > 0x7FFF) return 0x7FFF;
    return r;
}

static void fill_table(int *table, int64_t inc, int yoffs) /* yuv2rgb.c:680-692 (offsets, in elements) */
{
    int base = yoffs - (int)(inc >> 9);
    for (int i = 0; i < 256 + 2 * HEADROOM; i++) {
        int64_t cb = clip_u8(i - HEADROOM) * inc;
        table[i] = base + (int)(cb >> 16);
    }
}
static void fill_gv_table(int *table, int64_t inc) /* yuv2rgb.c:694-703 */
{
    int off = -(int)(inc >> 9);
    for (int i = 0; i < 256 + 2 * HEADROOM; i++) {
        int64_t cb = clip_u8(i - HEADROOM) * inc;
        table[i] = off + (int)(cb >> 16);
    }
}

/* ff_yuv2rgb_c_init_tables, yuv2rgb.c:717-973 (24 and 32 bpp cases) */
static int yuv2rgb_init_tables(OrSws *c, const int inv_table[4], int fullRange,
                               int brightness, int contrast, int saturation)
{
    const int df = c->o.dst_format;
    /* AV_PIX_FMT_RGB32 = BGRA, RGB32_1 = ABGR, BGR32 = RGBA, BGR32_1 = ARGB on little endian */
    const int isRgb = df == ORF_BGRA || df == ORF_ABGR || df == ORF_BGR24 || df == ORF_RGB565LE || df == ORF_RGB555LE || df == ORF_RGB444LE || df == ORF_X2RGB10LE ||
                      df == ORF_RGB8 || df == ORF_RGB4 || df == ORF_RGB4_BYTE;
    const int bpp = c->dstFormatBpp;
    const int yoffs = (fullRange ? 384 : 326) + HEADROOM;
    int64_t crv = inv_table[0], cbu = inv_table[1], cgu = -inv_table[2], cgv = -inv_table[3];
    int64_t cy = 1 << 16, oy = 0, yb;
    int i;

    if (!fullRange) {
        cy = (cy * 255) / 219;
        oy = 16 << 16;
    } else {
        crv = (crv * 224) / 255; cbu = (cbu * 224) / 255;
        cgu = (cgu * 224) / 255; cgv = (cgv * 224) / 255;
    }
    cy  = (cy * contrast) >> 16;
    crv = (crv * contrast * saturation) >> 32;
    cbu = (cbu * contrast * saturation) >> 32;
    cgu = (cgu * contrast * saturation) >> 32;
    cgv = (cgv * contrast * saturation) >> 32;
    oy -= 256LL * brightness;

    c->yuv2rgb_y_coeff  = (int16_t)round_to_int16(cy * (1 << 13));
    c->yuv2rgb_y_offset = (int16_t)round_to_int16(oy * (1 << 9));
    c->yuv2rgb_v2r = (int16_t)round_to_int16(crv * (1 << 13));
    c->yuv2rgb_v2g = (int16_t)round_to_int16(cgv * (1 << 13));
    c->yuv2rgb_u2g = (int16_t)round_to_int16(cgu * (1 << 13));
    c->yuv2rgb_u2b = (int16_t)round_to_int16(cbu * (1 << 13));

    crv = ((crv * (1 << 16)) + 0x8000) / ORMAX(cy, 1);
    cbu = ((cbu * (1 << 16)) + 0x8000) / ORMAX(cy, 1);
    cgu = ((cgu * (1 << 16)) + 0x8000) / ORMAX(cy, 1);
    cgv = ((cgv * (1 << 16)) + 0x8000) / ORMAX(cy, 1);

    free(c->yuvTable); c->yuvTable = NULL; c->has_lut = 0;
    yb = -(384 << 16) - HEADROOM * cy - oy;
    switch (bpp) {
    case 24:
    case 48:
        c->yuvTable = malloc(TABLE_PLANE);
        c->lut_elem = 1;
        for (i = 0; i < TABLE_PLANE; i++) {
            c->yuvTable[i] = (uint8_t)clip_u8((int)((yb + 0x8000) >> 16));
            yb += cy;
        }
        fill_table(c->table_rV, crv, yoffs);
        fill_table(c->table_gU, cgu, yoffs);
        fill_table(c->table_bU, cbu, yoffs);
        fill_gv_table(c->table_gV, cgv);
        c->has_lut = 1;
        break;
    case 32:
    case 64: {
        int base = (df == ORF_ABGR || df == ORF_ARGB) ? 8 : 0;
        int rbase = base + (isRgb ? 16 : 0), gbase = base + 8, bbase = base + (isRgb ? 0 : 16);
        int needAlpha = isALPHA(c->o.src_format);
        int abase = (base + 24) & 31;
        uint32_t *t = malloc(TABLE_PLANE * 3 * 4);
        c->yuvTable = (uint8_t *)t;
        c->lut_elem = 4;
        for (i = 0; i < TABLE_PLANE; i++) {
            unsigned yval = clip_u8((int)((yb + 0x8000) >> 16));
            t[i] = (yval << rbase) + (needAlpha ? 0 : (255u << abase));
            t[i + TABLE_PLANE] = yval << gbase;
            t[i + 2 * TABLE_PLANE] = yval << bbase;
            yb += cy;
        }
        fill_table(c->table_rV, crv, yoffs);
        fill_table(c->table_gU, cgu, yoffs + TABLE_PLANE);
        fill_table(c->table_bU, cbu, yoffs + 2 * TABLE_PLANE);
        fill_gv_table(c->table_gV, cgv);
        c->has_lut = 1;
        break;
    }
    case 1: { /* yuv2rgb.c:806-816: one plane of 0 / 1, the ramp starts at element 110 (the first 110 are never reached: zero here) */
        c->yuvTable = calloc(TABLE_PLANE, 1);
        c->lut_elem = 1;
        for (i = 0; i < TABLE_PLANE - 110; i++) {
            c->yuvTable[i + 110] = (uint8_t)(clip_u8((int)((yb + 0x8000) >> 16)) >> 7);
            yb += cy;
        }
        fill_table(c->table_gU, cgu, yoffs);
        fill_gv_table(c->table_gV, cgv);
        c->has_lut = 1;
        break;
    }
    case 4: { /* yuv2rgb.c:817-836 (rgb4 / bgr4 / rgb4_byte / bgr4_byte): three byte planes whose ramps start at element 110 / 37 / 110, so
               * that index + dither (0..220 / 0..72) is a centred threshold.  The elements before a ramp and the 73 after the green one are
               * left unwritten by the reference (av_malloc) and never reached by sane coefficients: zero here */
        const int rbase = isRgb ? 3 : 0, gbase = 1, bbase = isRgb ? 0 : 3;
        c->yuvTable = calloc(TABLE_PLANE * 3, 1);
        c->lut_elem = 1;
        for (i = 0; i < TABLE_PLANE - 110; i++) {
            const int yval = clip_u8((int)((yb + 0x8000) >> 16));
            c->yuvTable[i + 110] = (uint8_t)((yval >> 7) << rbase);
            c->yuvTable[i + 37 + TABLE_PLANE] = (uint8_t)(((yval + 43) / 85) << gbase);
            c->yuvTable[i + 110 + 2 * TABLE_PLANE] = (uint8_t)((yval >> 7) << bbase);
            yb += cy;
        }
        fill_table(c->table_rV, crv, yoffs);
        fill_table(c->table_gU, cgu, yoffs + TABLE_PLANE);
        fill_table(c->table_bU, cbu, yoffs + 2 * TABLE_PLANE);
        fill_gv_table(c->table_gV, cgv);
        c->has_lut = 1;
        break;
    }
    case 8: { /* yuv2rgb.c:837-856 (rgb8 / bgr8): ramps start at element 16 / 16 / 37 (dither 0..31 / 0..31 / 0..72); unwritten elements zero */
        const int rbase = isRgb ? 5 : 0, gbase = isRgb ? 2 : 3, bbase = isRgb ? 0 : 6;
        c->yuvTable = calloc(TABLE_PLANE * 3, 1);
        c->lut_elem = 1;
        for (i = 0; i < TABLE_PLANE - 38; i++) {
            const int yval = clip_u8((int)((yb + 0x8000) >> 16));
            c->yuvTable[i + 16] = (uint8_t)(((yval + 18) / 36) << rbase);
            c->yuvTable[i + 16 + TABLE_PLANE] = (uint8_t)(((yval + 18) / 36) << gbase);
            c->yuvTable[i + 37 + 2 * TABLE_PLANE] = (uint8_t)(((yval + 43) / 85) << bbase);
            yb += cy;
        }
        fill_table(c->table_rV, crv, yoffs);
        fill_table(c->table_gU, cgu, yoffs + TABLE_PLANE);
        fill_table(c->table_bU, cbu, yoffs + 2 * TABLE_PLANE);
        fill_gv_table(c->table_gV, cgv);
        c->has_lut = 1;
        break;
    }
    case 30: { /* yuv2rgb.c:915-941: three planes of 10-bit ramps at bit 20 / 10 / 0; "255u << 30" keeps the two X bits set
                * unless the source has alpha */
        const int rbase = isRgb ? 20 : 0, gbase = 10, bbase = isRgb ? 0 : 20;
        const int needAlpha = isALPHA(c->o.src_format);
        uint32_t *t = malloc(TABLE_PLANE * 3 * 4);
        c->yuvTable = (uint8_t *)t;
        c->lut_elem = 4;
        for (i = 0; i < TABLE_PLANE; i++) {
            const unsigned yval = (unsigned)clip_uintp2((int)((yb + 0x8000) >> 14), 10);
            t[i] = (yval << rbase) + (needAlpha ? 0 : (255u << 30));
            t[i + TABLE_PLANE] = yval << gbase;
            t[i + 2 * TABLE_PLANE] = yval << bbase;
            yb += cy;
        }
        fill_table(c->table_rV, crv, yoffs);
        fill_table(c->table_gU, cgu, yoffs + TABLE_PLANE);
        fill_table(c->table_bU, cbu, yoffs + 2 * TABLE_PLANE);
        fill_gv_table(c->table_gV, cgv);
        c->has_lut = 1;
        break;
    }
    case 12: case 15: case 16: { /* yuv2rgb.c:853-897; the byte swap of the non-native-endian tables is the oracle's BE pass */
        const int rbase = bpp == 12 ? (isRgb ? 8 : 0) : (isRgb ? bpp - 5 : 0);
        const int gbase = bpp == 12 ? 4 : 5;
        const int bbase = bpp == 12 ? (isRgb ? 0 : 8) : (isRgb ? 0 : bpp - 5);
        uint16_t *t = malloc(TABLE_PLANE * 3 * 2);
        c->yuvTable = (uint8_t *)t;
        c->lut_elem = 2;
        for (i = 0; i < TABLE_PLANE; i++) {
            const unsigned yval = (uint8_t)clip_u8((int)((yb + 0x8000) >> 16));
            if (bpp == 12) {
                t[i] = (uint16_t)((yval >> 4) << rbase); t[i + TABLE_PLANE] = (uint16_t)((yval >> 4) << gbase); t[i + 2 * TABLE_PLANE] = (uint16_t)((yval >> 4) << bbase);
            } else {
                t[i] = (uint16_t)((yval >> 3) << rbase); t[i + TABLE_PLANE] = (uint16_t)((yval >> (18 - bpp)) << gbase); t[i + 2 * TABLE_PLANE] = (uint16_t)((yval >> 3) << bbase);
            }
            yb += cy;
        }
        fill_table(c->table_rV, crv, yoffs);
        fill_table(c->table_gU, cgu, yoffs + TABLE_PLANE);
        fill_table(c->table_bU, cbu, yoffs + 2 * TABLE_PLANE);
        fill_gv_table(c->table_gV, cgv);
        c->has_lut = 1;
        break;
    }
    default:
        return -1; /* other bpp: not restated (reference returns EINVAL for planar >24) */
    }
    return 0;
}

/* fill_rgb2yuv_table, utils.c:614-706 */
static void fill_rgb2yuv_table(OrSws *c, const int table[4])
{
    int64_t W, V, Z, Cy, Cu, Cv;
    int64_t vr = table[0], ub = table[1], ug = -table[2], vg = -table[3];
    const int64_t ONE = 65536;
    int64_t cy = ONE;

    cy = cy * 255 / 219; /* dstRange forced 0, utils.c:663 */
    W = rounded_div(ONE * ONE * ug, ub);
    V = rounded_div(ONE * ONE * vg, vr);
    Z = ONE * ONE - W - V;
    Cy = rounded_div(cy * Z, ONE);
    Cu = rounded_div(ub * Z, ONE);
    Cv = rounded_div(vr * Z, ONE);

    c->rgb2yuv[RY] = (int32_t)-rounded_div((1 << 15) * V, Cy);
    c->rgb2yuv[GY] = (int32_t) rounded_div((1 << 15) * ONE * ONE, Cy);
    c->rgb2yuv[BY] = (int32_t)-rounded_div((1 << 15) * W, Cy);
    c->rgb2yuv[RU] = (int32_t) rounded_div((1 << 15) * V, Cu);
    c->rgb2yuv[GU] = (int32_t)-rounded_div((1 << 15) * ONE * ONE, Cu);
    c->rgb2yuv[BU] = (int32_t) rounded_div((1 << 15) * (Z + W), Cu);
    c->rgb2yuv[RV] = (int32_t) rounded_div((1 << 15) * (V + Z), Cv);
    c->rgb2yuv[GV] = (int32_t)-rounded_div((1 << 15) * ONE * ONE, Cv);
    c->rgb2yuv[BV] = (int32_t) rounded_div((1 << 15) * W, Cv);

    if (!memcmp(table, yuv2rgb_coeffs[5], sizeof(int) * 4)) { /* :693-703 */
        c->rgb2yuv[BY] =  ((int)(0.114 * 219 / 255 * (1 << 15) + 0.5));
        c->rgb2yuv[BV] = (-(int)(0.081 * 224 / 255 * (1 << 15) + 0.5));
        c->rgb2yuv[BU] =  ((int)(0.500 * 224 / 255 * (1 << 15) + 0.5));
        c->rgb2yuv[GY] =  ((int)(0.587 * 219 / 255 * (1 << 15) + 0.5));
        c->rgb2yuv[GV] = (-(int)(0.419 * 224 / 255 * (1 << 15) + 0.5));
        c->rgb2yuv[GU] = (-(int)(0.331 * 224 / 255 * (1 << 15) + 0.5));
        c->rgb2yuv[RY] =  ((int)(0.299 * 219 / 255 * (1 << 15) + 0.5));
        c->rgb2yuv[RV] =  ((int)(0.500 * 224 / 255 * (1 << 15) + 0.5));
        c->rgb2yuv[RU] = (-(int)(0.169 * 224 / 255 * (1 << 15) + 0.5));
    }
}

/* swscale.c:577-660 */
static void solve_range_convert(uint16_t src_min, uint16_t src_max, uint16_t dst_min, uint16_t dst_max,
                                int src_shift, int mult_shift, uint32_t *coeff, int64_t *offset)
{
    uint16_t src_range = src_max - src_min, dst_range = dst_max - dst_min;
    int total_shift = mult_shift + src_shift;
    *coeff = (uint32_t)CEIL_RSHIFT((int64_t)(((uint64_t)dst_range << total_shift) / src_range), src_shift);
    *offset = ((int64_t)dst_max << total_shift) - ((int64_t)src_max << src_shift) * *coeff +
              (1U << (mult_shift - 1));
}
static void init_range_convert(OrSws *c)
{
    c->range_active = 0;
    if (c->o.src_range != c->o.dst_range && !isAnyRGB(c->o.dst_format) && c->dstBpc < 32) {
        const int bit_depth = c->dstBpc ? ORMIN(c->dstBpc, 16) : 8;
        const int src_bits = bit_depth <= 14 ? 15 : 19;
        const int src_shift = src_bits - bit_depth;
        const int mult_shift = bit_depth <= 14 ? 14 : 18;
        const uint16_t mpeg_min = 16U << (bit_depth - 8);
        const uint16_t mpeg_max_lum = 235U << (bit_depth - 8);
        const uint16_t mpeg_max_chr = 240U << (bit_depth - 8);
        const uint16_t jpeg_max = (1U << bit_depth) - 1;
        if (c->o.src_range) {
            solve_range_convert(0, jpeg_max, mpeg_min, mpeg_max_lum, src_shift, mult_shift, &c->lumCoeff, &c->lumOffset);
            solve_range_convert(0, jpeg_max, mpeg_min, mpeg_max_chr, src_shift, mult_shift, &c->chrCoeff, &c->chrOffset);
        } else {
            solve_range_convert(mpeg_min, mpeg_max_lum, 0, jpeg_max, src_shift, mult_shift, &c->lumCoeff, &c->lumOffset);
            solve_range_convert(mpeg_min, mpeg_max_chr, 0, jpeg_max, src_shift, mult_shift, &c->chrCoeff, &c->chrOffset);
        }
        c->range_active = 1;
    }
}

/* ------------------------------------------------------------------ */
/* init                                                                */
/* ------------------------------------------------------------------ */
static int handle_0alpha(int *format) /* utils.c:811-820 */
{
    switch (*format) {
    case ORF_0BGR: *format = ORF_ABGR; return 1;
    case ORF_BGR0: *format = ORF_BGRA; return 4;
    case ORF_0RGB: *format = ORF_ARGB; return 1;
    case ORF_RGB0: *format = ORF_RGBA; return 4;
    default: return 0;
    }
}
static void handle_formats(OrSws *c)
{
    c->src0Alpha |= handle_0alpha(&c->o.src_format);
    c->dst0Alpha |= handle_0alpha(&c->o.dst_format);
    if (c->o.src_format == ORF_XYZ12LE) { c->o.src_format = ORF_RGB48LE; c->src_xyz = 1; }   /* handle_xyz utils.c:822-829 */
    if (c->o.dst_format == ORF_XYZ12LE) { c->o.dst_format = ORF_RGB48LE; c->dst_xyz = 1; }
}
static int get_local_pos(int chr_subsample, int pos) /* utils.c:168-175 */
{
    if (pos == -1 || pos <= -513) pos = (128 << chr_subsample) - 128;
    pos += 128;
    return pos >> chr_subsample;
}

static int or_init(OrSws *c);

void or_sws_default_opts(OrSwsOpts *o)
{
    memset(o, 0, sizeof(*o));
    o->flags = OR_SWS_BICUBIC; /* options.c:35 */
    o->scaler_params[0] = o->scaler_params[1] = OR_SWS_PARAM_DEFAULT;
    o->dither = 1; /* AUTO */
    o->src_v_chr_pos = o->src_h_chr_pos = o->dst_v_chr_pos = o->dst_h_chr_pos = -513;
    o->src_format = o->dst_format = ORF_NONE;
}

static OrSws *alloc_set_opts(int srcW, int srcH, int srcFmt, int dstW, int dstH, int dstFmt,
                             int flags, const double *param) /* utils.c:75-95 */
{
    OrSws *c = calloc(1, sizeof(*c));
    or_sws_default_opts(&c->o);
    c->o.flags = flags;
    c->o.src_w = srcW; c->o.src_h = srcH; c->o.dst_w = dstW; c->o.dst_h = dstH;
    c->o.src_format = srcFmt; c->o.dst_format = dstFmt;
    if (param) { c->o.scaler_params[0] = param[0]; c->o.scaler_params[1] = param[1]; }
    return c;
}

static int handle_jpeg(int *format) /* utils.c:773 */
{
    if (*format == ORF_YUVJ420P) { *format = ORF_YUV420P; return 1; }
    if (*format == ORF_YUVJ422P) { *format = ORF_YUV422P; return 1; }
    if (*format == ORF_YUVJ444P) { *format = ORF_YUV444P; return 1; }
    if (*format == ORF_YUVJ440P) { *format = ORF_YUV440P; return 1; }
    if (*format == ORF_YUVJ411P) { *format = ORF_YUV411P; return 1; }
    /* gray8, ya8, gray9 .. gray16, ya16 (LE and BE): always full range (utils.c:791-805).  The float gray formats are NOT in the list: a
     * grayf32 / grayf16 / yaf32 / yaf16 picture keeps the range it was given (0 by default, i.e. "limited") */
    if (isGray(*format) && !isFloat(*format)) return 1;
    return 0;
}

static int init_context(OrSws *c) /* sws_init_context, utils.c:1884 */
{
    c->o.src_range |= handle_jpeg(&c->o.src_format);
    c->o.dst_range |= handle_jpeg(&c->o.dst_format);
    return or_init(c);
}

OrSws *or_sws_create(const OrSwsOpts *o)
{
    OrSws *c = calloc(1, sizeof(*c));
    c->o = *o;
    c->src_be = be_twin(&c->o.src_format);
    c->dst_be = be_twin(&c->o.dst_format);
    if (!desc_get(c->o.src_format) || !desc_get(c->o.dst_format) || init_context(c) < 0) {
        or_sws_free(c);
        return NULL;
    }
    return c;
}

OrSws *or_sws_get_context(int srcW, int srcH, int srcFmt, int dstW, int dstH, int dstFmt,
                          int flags, const double *param)
{
    OrSws *c;
    const int sbe = be_twin(&srcFmt), dbe = be_twin(&dstFmt);
    if (!desc_get(srcFmt) || !desc_get(dstFmt)) return NULL;
    c = alloc_set_opts(srcW, srcH, srcFmt, dstW, dstH, dstFmt, flags, param);
    c->src_be = sbe; c->dst_be = dbe;
    if (init_context(c) < 0) { or_sws_free(c); return NULL; }
    return c;
}

void or_sws_free(OrSws *c)
{
    if (!c) return;
    free(c->hLumFilter); free(c->hChrFilter); free(c->vLumFilter); free(c->vChrFilter);
    free(c->hLumFilterPos); free(c->hChrFilterPos); free(c->vLumFilterPos); free(c->vChrFilterPos);
    free(c->yuvTable);
    free(c->dither_error[0]); free(c->dither_error[1]); free(c->dither_error[2]);
    or_sws_free(c->cascade[0]); or_sws_free(c->cascade[1]); or_sws_free(c->cascade[2]);
    free(c->casc_tmp2); free(c->gamma_tab); free(c->inv_gamma_tab);
    free(c->casc_tmp[0]); free(c->casc_tmp[1]); free(c->casc_tmp[2]); free(c->casc_tmp[3]);
    free(c);
}

static int alphaless_fmt(int f) /* utils.c:1060-1118 (the big-endian rows are the same formats here: byte order is handled outside) */
{
    switch (f) {
    case ORF_ARGB: case ORF_RGBA: return ORF_RGB24;
    case ORF_ABGR: case ORF_BGRA: return ORF_BGR24;
    case ORF_YA8: return ORF_GRAY8;
    case ORF_YUVA420P: return ORF_YUV420P; case ORF_YUVA422P: return ORF_YUV422P; case ORF_YUVA444P: return ORF_YUV444P;
    case ORF_RGBA64LE: return ORF_RGB48LE; case ORF_BGRA64LE: return ORF_BGR48LE;
    case ORF_YA16LE: return ORF_GRAY16LE;
    case ORF_YUVA420P9LE: return ORF_YUV420P9LE; case ORF_YUVA422P9LE: return ORF_YUV422P9LE; case ORF_YUVA444P9LE: return ORF_YUV444P9LE;
    case ORF_YUVA420P10LE: return ORF_YUV420P10LE; case ORF_YUVA422P10LE: return ORF_YUV422P10LE; case ORF_YUVA444P10LE: return ORF_YUV444P10LE;
    case ORF_YUVA420P16LE: return ORF_YUV420P16LE; case ORF_YUVA422P16LE: return ORF_YUV422P16LE; case ORF_YUVA444P16LE: return ORF_YUV444P16LE;
    case ORF_GBRAP: return ORF_GBRP;   /* utils.c:1073-1085 */
    case ORF_GBRAP10LE: return ORF_GBRP10LE; case ORF_GBRAP12LE: return ORF_GBRP12LE;
    case ORF_GBRAP14LE: return ORF_GBRP14LE; case ORF_GBRAP16LE: return ORF_GBRP16LE;
    }
    return ORF_NONE;
}

/* ff_sws_alphablendaway, alphablend.c:23-175 (little-endian words) */
static int unscaled_alphablend(const OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY, int srcSliceH,
                               uint8_t *const dst[], const int dstStride[])
{
    const Desc *desc = desc_get(c->o.src_format);
    const int lum_w = c->o.src_w, lum_h = c->o.src_h;
    const int plane_count = isGray(c->o.src_format) ? 1 : 3;
    const int depth = desc->c[0].depth, sixteen_bits = depth >= 9;
    const unsigned off = 1u << (depth - 1), shift = depth, max = (1u << shift) - 1;
    int target_table[2][3], plane, x, ysrc;
    for (plane = 0; plane < plane_count; plane++) {
        int a = 0, b = 0;
        if (c->o.alpha_blend == 2) { a = (1 << (depth - 1)) / 2; b = 3 * (1 << (depth - 1)) / 2; }
        target_table[0][plane] = plane && !(desc->flags & PF_RGB) ? 1 << (depth - 1) : a;
        target_table[1][plane] = plane && !(desc->flags & PF_RGB) ? 1 << (depth - 1) : b;
    }
    if (desc->flags & PF_PLANAR) {
        for (plane = 0; plane < plane_count; plane++) {
            const int w = plane ? c->chrSrcW : c->o.src_w;
            const int x_subsample = plane ? desc->lw : 0, y_subsample = plane ? desc->lh : 0;
            for (ysrc = 0; ysrc < CEIL_RSHIFT(srcSliceH, y_subsample); ysrc++) {
                const int y = ysrc + (srcSliceY >> y_subsample);
                const int subsample_row = y_subsample && (y << y_subsample) + 1 < lum_h;
                const uint8_t *sp = src[plane] + (ptrdiff_t)srcStride[plane] * ysrc;
                uint8_t *dp = dst[plane] + (ptrdiff_t)dstStride[plane] * y;
                if (x_subsample || subsample_row) {
                    const uint8_t *ap = src[plane_count] + (((ptrdiff_t)srcStride[plane_count] * ysrc) << y_subsample);
                    for (x = 0; x < w; x++) {
                        const int xnext = 2 * x + 1 < lum_w - 1 ? 2 * x + 1 : lum_w - 1;
                        int alpha;
                        if (sixteen_bits) {
                            const uint16_t *a = (const uint16_t *)ap, *a2 = (const uint16_t *)(ap + srcStride[plane_count]);
                            unsigned u;
                            alpha = subsample_row ? (a[2 * x] + a[xnext] + 2 + a2[2 * x] + a2[xnext]) >> 2 : (a[2 * x] + a[xnext]) >> 1;
                            u = ((const uint16_t *)sp)[x] * (unsigned)alpha + target_table[((x ^ y) >> 5) & 1][plane] * (max - alpha) + off;
                            u = (u + (u >> shift)) >> shift;
                            ((uint16_t *)dp)[x] = (uint16_t)(u > max ? max : u);
                        } else {
                            const uint8_t *a = ap, *a2 = ap + srcStride[plane_count];
                            unsigned u;
                            alpha = subsample_row ? (a[2 * x] + a[xnext] + 2 + a2[2 * x] + a2[xnext]) >> 2 : (a[2 * x] + a[xnext]) >> 1;
                            u = sp[x] * alpha + target_table[((x ^ y) >> 5) & 1][plane] * (255 - alpha) + 128;
                            dp[x] = (uint8_t)((257 * u) >> 16);
                        }
                    }
                } else {
                    const uint8_t *ap = src[plane_count] + (ptrdiff_t)srcStride[plane_count] * ysrc;
                    for (x = 0; x < w; x++) {
                        if (sixteen_bits) {
                            const unsigned a = ((const uint16_t *)ap)[x];
                            unsigned u = ((const uint16_t *)sp)[x] * a + target_table[((x ^ y) >> 5) & 1][plane] * (max - a) + off;
                            u = (u + (u >> shift)) >> shift;
                            ((uint16_t *)dp)[x] = (uint16_t)(u > max ? max : u);
                        } else {
                            const unsigned u = sp[x] * ap[x] + target_table[((x ^ y) >> 5) & 1][plane] * (255 - ap[x]) + 128;
                            dp[x] = (uint8_t)((257 * u) >> 16);
                        }
                    }
                }
            }
        }
    } else {
        const int alpha_pos = desc->c[plane_count].offset, w = c->o.src_w;
        for (ysrc = 0; ysrc < srcSliceH; ysrc++) {
            const int y = ysrc + srcSliceY;
            if (sixteen_bits) {
                const uint16_t *s = (const uint16_t *)(src[0] + (ptrdiff_t)srcStride[0] * ysrc + 2 * !alpha_pos);
                const uint16_t *a = (const uint16_t *)(src[0] + (ptrdiff_t)srcStride[0] * ysrc + alpha_pos);
                uint16_t *d = (uint16_t *)(dst[0] + (ptrdiff_t)dstStride[0] * y);
                for (x = 0; x < w; x++)
                    for (plane = 0; plane < plane_count; plane++) {
                        const int x_index = (plane_count + 1) * x;
                        unsigned u = s[x_index + plane] * (unsigned)a[x_index] + target_table[((x ^ y) >> 5) & 1][plane] * (max - a[x_index]) + off;
                        u = (u + (u >> shift)) >> shift;
                        d[plane_count * x + plane] = (uint16_t)(u > max ? max : u);
                    }
            } else {
                const uint8_t *s = src[0] + (ptrdiff_t)srcStride[0] * ysrc + !alpha_pos;
                const uint8_t *a = src[0] + (ptrdiff_t)srcStride[0] * ysrc + alpha_pos;
                uint8_t *d = dst[0] + (ptrdiff_t)dstStride[0] * y;
                for (x = 0; x < w; x++)
                    for (plane = 0; plane < plane_count; plane++) {
                        const int x_index = (plane_count + 1) * x;
                        const unsigned u = s[x_index + plane] * a[x_index] + target_table[((x ^ y) >> 5) & 1][plane] * (255 - a[x_index]) + 128;
                        d[plane_count * x + plane] = (uint8_t)((257 * u) >> 16);
                    }
            }
        }
    }
    return 0;   /* alphablend.c:176: sws_scale() reports no rows for this converter */
}

/* planarRgbToplanarRgbWrapper (swscale_unscaled.c:1380-1402) with ff_copyPlane (:126-145) as it is: `width` is passed in pixels
 * and used as a byte count, so a 16-bit row is copied in full only when the two strides are equal (one memcpy over the whole slice) */
static int unscaled_planarrgb_planarrgb(const OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY, int srcSliceH,
                                        uint8_t *const dst[], const int dstStride[])
{
    const Desc *dd = desc_get(c->o.dst_format);
    const int w = c->o.src_w;
    if (!srcSliceH) return 0;
    for (int k = 0; k < 3; k++) {
        uint8_t *d = dst[k] + (ptrdiff_t)dstStride[k] * srcSliceY;
        if (dstStride[k] == srcStride[k] && srcStride[k] > 0) memcpy(d, src[k], (size_t)(srcSliceH - 1) * dstStride[k] + w);
        else for (int i = 0; i < srcSliceH; i++) memcpy(d + (ptrdiff_t)i * dstStride[k], src[k] + (ptrdiff_t)i * srcStride[k], w);
    }
    if (dst[3] && isALPHA(c->o.dst_format))
        for (int i = 0; i < srcSliceH; i++) {
            uint8_t *row = dst[3] + (ptrdiff_t)(srcSliceY + i) * dstStride[3];
            if (dd->c[0].depth > 8) { uint16_t *r16 = (uint16_t *)row; for (int j = 0; j < w; j++) r16[j] = (uint16_t)(0xFFFF >> (16 - dd->c[3].depth)); }
            else memset(row, 255, w);
        }
    return srcSliceH;
}

static void get_unscaled(OrSws *c) /* ff_get_unscaled_swscale, swscale_unscaled.c:2392-2706 (subset) */
{
    const int s = c->o.src_format, d = c->o.dst_format, flags = c->o.flags;
    c->unscaled_kind = UNSC_NONE;
    if ((s == ORF_YUV420P || s == ORF_YUVA420P) && (d == ORF_NV12 || d == ORF_NV21)) c->unscaled_kind = UNSC_PLANAR2NV12;
    if (d == ORF_YUV420P && (s == ORF_NV12 || s == ORF_NV21)) c->unscaled_kind = UNSC_NV122PLANAR;
    if ((s == ORF_YUV420P || s == ORF_YUV422P || s == ORF_YUVA420P) && isAnyRGB(d) && !(flags & OR_SWS_ACCURATE_RND) &&
        (c->o.dither == 2 || c->o.dither == 1) && !(c->o.dst_h & 1)) { /* :2425-2431 */
        /* ff_yuv2rgb_get_func_ptr, yuv2rgb.c:561-678: 24/32 bpp C converters */
        /* yuv2rgb_c_24_rgb/_bgr, yuv2rgb_c_32, yuv420p_gbrp_c / yuv422p_gbrp_c; NULL (-> scaler chain) for gbrp9..16/f32 */
        if (d == ORF_RGB24 || d == ORF_BGR24 || d == ORF_RGBA || d == ORF_BGRA || d == ORF_ARGB || d == ORF_ABGR || d == ORF_GBRP ||
            d == ORF_RGB48LE || d == ORF_BGR48LE ||   /* yuv2rgb_c_48 / yuv2rgb_c_bgr48 (yuv2rgb.c:107-125, :505-508) */
            isRGB16(d) ||                              /* yuv2rgb_c_16/15/12_ordered_dither, yuv422p_bgr16/15/12 (:533-535, :554-556, :612-640) */
            isRGB8class(d) || isRGB4bits(d))           /* yuv2rgb_c_8/4/4b_ordered_dither, yuv422p_bgr8/4/4_byte (:536-538, :557-559, :615-623) */
            c->unscaled_kind = UNSC_YUV2RGB;
        else if (d == ORF_MONOBLACK) c->unscaled_kind = UNSC_YUV2MONO;   /* yuv2rgb_c_1_ordered_dither (yuv2rgb.c:457-517, :624, :671) */
        else if (d == ORF_RGBA64LE || d == ORF_BGRA64LE) c->unscaled_kind = UNSC_NONE; /* no C converter: ff_yuv2rgb_get_func_ptr returns NULL */
    }
    if (s == ORF_YUV444P && (d == ORF_NV24 || d == ORF_NV42)) c->unscaled_kind = UNSC_PLANAR2NV24;   /* :2410-2413 */
    if (d == ORF_YUV444P && (s == ORF_NV24 || s == ORF_NV42)) c->unscaled_kind = UNSC_NV242PLANAR;   /* :2420-2423 */
    if ((s == ORF_YUV420P10LE || s == ORF_YUVA420P10LE || s == ORF_YUV420P12LE || s == ORF_YUV420P14LE || s == ORF_YUV420P16LE || s == ORF_YUVA420P16LE) &&   /* (the two yuva420p formats of the reference's list: found missing by tools/ref/ref_crosscheck.py, round 6) */
        (d == ORF_P010LE || d == ORF_P016LE) && !c->src_be && !c->dst_be) c->unscaled_kind = UNSC_P01X;                           /* :2432-2439 */
    if ((s == ORF_YUV420P || s == ORF_YUVA420P) && (d == ORF_P010LE || d == ORF_P016LE) && !c->dst_be) c->unscaled_kind = UNSC_8_P01X; /* :2440-2444 */
    if (s == ORF_YUV410P && !(c->o.dst_h & 3) && (d == ORF_YUV420P || d == ORF_YUVA420P) && !(flags & OR_SWS_BITEXACT))
        c->unscaled_kind = UNSC_YVU9_YV12;                                                            /* :2446-2451 */
    /* bgr24toYV12 (:2452-2456) */
    if (s == ORF_BGR24 && (d == ORF_YUV420P || d == ORF_YUVA420P) && !(flags & OR_SWS_ACCURATE_RND) && !(c->o.dst_w & 1))
        c->unscaled_kind = UNSC_BGR24_YV12;
    /* rgbToRgbWrapper (:2459-2463) when findRgbConvFn (:1843-1998) has a converter; 8-bit 24/32 bpp formats on a
     * little-endian host.  needsDither is 0 for >= 24 bpp destinations.  ":1991-1994 Maintain symmetry between
     * endianness": with BITEXACT a 24 bpp source is not shuffled into RGB32/BGR32 (= bgra/rgba bytes on LE). */
    if (isAnyRGB(s) && isAnyRGB(d) && isPacked(s) && isPacked(d) && s != d &&
        desc_get(s)->c[0].depth == 8 && desc_get(d)->c[0].depth == 8) {
        const int s32 = desc_get(s)->c[0].step == 4;
        if (!(!s32 && (d == ORF_BGRA || d == ORF_RGBA) && (flags & OR_SWS_BITEXACT)))
            c->unscaled_kind = UNSC_RGB2RGB;
    }
    /* AYUV / VUYA / UYVA -> AYUV / VUYA / VUYX / UYVA byte shuffles (:1938-1949, :2459-2461); vuyx is never a source, uyva never a destination but of ayuv / vuya */
    if (s != d && ((s == ORF_AYUV && (d == ORF_VUYA || d == ORF_VUYX || d == ORF_UYVA)) || (s == ORF_VUYA && (d == ORF_AYUV || d == ORF_UYVA)) ||
                   (s == ORF_UYVA && (d == ORF_AYUV || d == ORF_VUYA || d == ORF_VUYX))))
        c->unscaled_kind = UNSC_RGB2RGB;
    if (isAnyRGB(s) && isAnyRGB(d) && isPacked(s) && isPacked(d) && (isRGB16(s) || isRGB16(d))) {
        /* findRgbConvFn's two switch tables (:1941-1979) on (srcFormatBpp, dstFormatBpp) for formats of the same / of opposite
         * "in int" channel order; rgbToRgbWrapper only without dither need, or with FAST_BILINEAR / POINT (:2459-2463) */
        const int sid = c->srcFormatBpp, did = c->dstFormatBpp;
        const int s_rgbint = s == ORF_RGB24 || s == ORF_BGRA || s == ORF_ABGR || s == ORF_RGB565LE || s == ORF_RGB555LE || s == ORF_RGB444LE;
        const int d_rgbint = d == ORF_RGB24 || d == ORF_BGRA || d == ORF_ABGR || d == ORF_RGB565LE || d == ORF_RGB555LE || d == ORF_RGB444LE;
        const int needsDither = did < 24 && did < sid;   /* (:2400-2403), both sides are RGB here */
        int have;
        if (s_rgbint == d_rgbint)
            have = (did == 15 && (sid == 12 || sid == 16 || sid == 24 || sid == 32)) || (did == 16 && (sid == 15 || sid == 24 || sid == 32)) ||
                   (did == 24 && (sid == 15 || sid == 16)) || (did == 32 && (sid == 15 || sid == 16));
        else
            have = (did == 12 && sid == 12) || (did == 15 && (sid == 15 || sid == 16 || sid == 24 || sid == 32)) ||
                   (did == 16 && (sid == 15 || sid == 16 || sid == 24 || sid == 32)) || (did == 24 && (sid == 15 || sid == 16)) ||
                   (did == 32 && (sid == 15 || sid == 16));
        if ((d == ORF_BGRA || d == ORF_RGBA) && (flags & OR_SWS_BITEXACT)) have = 0;   /* :1991-1994 */
        if (have && (!needsDither || (flags & (OR_SWS_FAST_BILINEAR | OR_SWS_POINT)))) c->unscaled_kind = UNSC_RGBLOW;
    }
    {   /* 16-bit packed RGB: findRgbConvFn rows for rgb48 <-> bgr48, rgb48 -> rgba64, rgba64 -> rgb48 (:1869-1911);
         * Rgb16ToPlanarRgb16Wrapper (:2488-2507) and planarRgb16ToRgb16Wrapper (:2514-2533) */
        const int s48 = s == ORF_RGB48LE || s == ORF_BGR48LE, s64 = s == ORF_RGBA64LE || s == ORF_BGRA64LE;
        const int d48 = d == ORF_RGB48LE || d == ORF_BGR48LE, d64 = d == ORF_RGBA64LE || d == ORF_BGRA64LE;
        const int sp16 = isPlanarRGB(s) && !isFloat(s) && desc_get(s)->c[0].depth > 8, dp16 = isPlanarRGB(d) && !isFloat(d) && desc_get(d)->c[0].depth > 8;
        if (s != d && ((s48 && d48) || (s48 && d64) || (s64 && d48))) c->unscaled_kind = UNSC_RGB16SHUFFLE;
        if (isRGB30(s) && (d48 || d64)) c->unscaled_kind = UNSC_RGB30_TO_16;   /* x2rgb10to48 / x2rgb10tobgr48 / ..64 (:1912-1937, :2459-2463) */
        if ((s48 || s64) && dp16 && !desc_get(d)->c[0].shift) c->unscaled_kind = UNSC_PACKED16_TO_GBRP16;   /* (the rule lists gbrp9..16 and gbrap10..16: not the msb formats) */
        if (isRGB30(s) && isPlanarRGB(d) && !isFloat(d) && desc_get(d)->c[0].depth >= 10) c->unscaled_kind = UNSC_RGB30_TO_GBRP;   /* :2509-2512 */
        if (sp16 && !desc_get(s)->c[0].shift && (d48 || d64)) c->unscaled_kind = UNSC_GBRP16_TO_PACKED16;
        if (isRGB30(d) && isPlanarRGB(s) && !isFloat(s) && desc_get(s)->c[0].depth >= 10) c->unscaled_kind = UNSC_GBRP_TO_RGB30;   /* :2535-2538 */
    }
    /* rgbToPlanarRgbWrapper (:2542-2544): 8-bit packed RGB -> gbrp */
    if (isAnyRGB(s) && isPacked(s) && desc_get(s)->c[0].depth == 8 && d == ORF_GBRP) c->unscaled_kind = UNSC_PACKED2GBRP;
    /* planarRgbToRgbWrapper (:2480-2481): gbrp -> byte RGB */
    if (s == ORF_GBRP && isAnyRGB(d) && isPacked(d) && desc_get(d)->c[0].depth == 8) c->unscaled_kind = UNSC_GBRP2PACKED;
    /* planarRgbToplanarRgbWrapper (:2469-2479): gbrp <-> gbrap at the same depth (native byte order only) */
    if (!c->src_be && !c->dst_be && isPlanarRGB(s) && isPlanarRGB(d) && !isFloat(s) && !isFloat(d) && isALPHA(s) != isALPHA(d) &&
        desc_get(s)->c[0].depth == desc_get(d)->c[0].depth && desc_get(s)->c[0].depth != 9 && !desc_get(s)->c[0].shift && !desc_get(d)->c[0].shift)
        c->unscaled_kind = UNSC_PLANARRGB_PLANARRGB;
    /* planarRgbaToRgbWrapper (:2492-2493): gbrap -> byte RGB; rgbToPlanarRgbaWrapper (:2546-2548): 8-bit packed RGB -> gbrap */
    if (s == ORF_GBRAP && (d == ORF_RGB24 || d == ORF_BGR24 || d == ORF_RGBA || d == ORF_BGRA || d == ORF_ARGB || d == ORF_ABGR)) c->unscaled_kind = UNSC_GBRP2PACKED;
    if (d == ORF_GBRAP && (s == ORF_RGB24 || s == ORF_BGR24 || s == ORF_RGBA || s == ORF_BGRA || s == ORF_ARGB || s == ORF_ABGR)) c->unscaled_kind = UNSC_PACKED2GBRP;
    if (isBayer(s) && (d == ORF_RGB24 || d == ORF_RGB48LE || d == ORF_YUV420P)) c->unscaled_kind = UNSC_BAYER;   /* bayer_to_rgb24 / rgb48 / yv12_wrapper (:2543-2555) */
    /* palToRgbWrapper / palToGbrpWrapper (:2619-2630) for the palette-expanded sources, gray8 with its grey ramp among them: the index is
     * replicated whatever sws_setColorspaceDetails() was given */
    if (usePal(s) && (d == ORF_GBRP || d == ORF_GBRAP || d == ORF_RGB24 || d == ORF_BGR24 || d == ORF_RGBA || d == ORF_BGRA || d == ORF_ARGB || d == ORF_ABGR))
        c->unscaled_kind = UNSC_PAL2RGB;
    /* bswap_16bpc (:545-570, rule :2560-2612): the same format in the other byte order.  Where the simple-copy rule below applies too it comes later and wins (same bytes);
     * it does NOT apply to planar YUV with SWS_SRC_V_CHR_DROP (chrSrcVSubSample != chrDstVSubSample), and then bswap_16bpc's own row count shows: EVERY plane, luma
     * included, gets srcSliceH >> chrDstVSubSample rows.  Formats of the list only (YUVA planes and the semi-planar families are not on it: they go through the scaler).
     * (c->o formats are the LE twins: "s == d" alone does not say the caller's formats are equal.)  Round 6, found by tools/ref/ref_crosscheck.py */
    c->bswap16 = 0;
    if (s == d && c->src_be != c->dst_be && bswap16_listed(s)) {
        /* everywhere but the row-dropping case the words bswap_16bpc writes are the copy wrappers' between this oracle's byte-order wrappers (which also carry the
         * xyz12 conversions: xyz12 is rgb48 here as in the reference, utils.c:822-823), so only that case runs the function itself */
        if (isPlanarYUV(s) && c->chrDstVSub > 0 && c->chrDstVSub != c->chrSrcVSub) { c->unscaled_kind = UNSC_BSWAP16; c->bswap16 = 1; }
        else c->unscaled_kind = isPacked(s) ? UNSC_PACKEDCOPY : UNSC_PLANARCOPY;
    }
    /* bswap_32bpc (:572-597, rule :2614-2617): gbrpf32 / gbrapf32 in the other byte order -- every row of every plane (no chroma planes): the plane copy's bytes */
    if (s == d && c->src_be != c->dst_be && (s == ORF_GBRPF32LE || s == ORF_GBRAPF32LE)) c->unscaled_kind = UNSC_PLANARCOPY;
    /* simple copy (:2647-2668) */
    if ((s == d && c->src_be == c->dst_be) || (s == ORF_YUVA420P && d == ORF_YUV420P) || (s == ORF_YUV420P && d == ORF_YUVA420P) ||
        (isFloat(s) == isFloat(d) && isFloat16(s) == isFloat16(d) &&
         ((isPlanarYUV(s) && isGray(d) && !isALPHA(d)) || (isPlanarYUV(d) && isGray(s) && !isALPHA(s)) ||
          (isGray(d) && !isALPHA(d) && isGray(s) && !isALPHA(s)))) ||   /* isPlanarGray(x) = isGray(x) && !isALPHA(x) (:2673) */
        (isFloat(s) == isFloat(d) && isFloat16(s) == isFloat16(d) &&
         (isPlanarYUV(s) && isPlanarYUV(d) && c->chrDstHSub == c->chrSrcHSub && c->chrDstVSub == c->chrSrcVSub &&
          isSemiPlanarYUV(s) == isSemiPlanarYUV(d) && isSwappedChroma(s) == isSwappedChroma(d)))) {
        if (!isPacked(s)) c->unscaled_kind = UNSC_PLANARCOPY;
        else c->unscaled_kind = UNSC_PACKEDCOPY; /* packedCopyWrapper (:2138-2157) */
        c->bswap16 = 0;
    }
    /* uint_y_to_float_y_wrapper / float_y_to_uint_y_wrapper (:2639-2647); rules name the native-endian format */
    if (s == ORF_GRAY8 && d == ORF_GRAYF32LE && !c->dst_be) c->unscaled_kind = UNSC_U8_TO_F32;
    if (s == ORF_GRAYF32LE && d == ORF_GRAY8 && !c->src_be) c->unscaled_kind = UNSC_F32_TO_U8;
    if (s == ORF_YUV422P && (d == ORF_YUYV422 || d == ORF_UYVY422)) c->unscaled_kind = UNSC_PLANAR2P422;          /* :2667-2672 */
    if ((flags & (OR_SWS_FAST_BILINEAR | OR_SWS_POINT)) && (s == ORF_YUV420P || s == ORF_YUVA420P) &&
        (d == ORF_YUYV422 || d == ORF_UYVY422)) c->unscaled_kind = UNSC_PLANAR2P422;                               /* :2684-2692 */
    if ((s == ORF_YUYV422 || s == ORF_UYVY422) && (d == ORF_YUV420P || d == ORF_YUVA420P || d == ORF_YUV422P)) c->unscaled_kind = UNSC_P4222PLANAR; /* :2693-2702 */
    if (d == ORF_YUV420P && (s == ORF_NV24 || s == ORF_NV42)) c->unscaled_kind = UNSC_NV242YUV420;    /* :2703-2705 */
}

static int or_init(OrSws *c) /* ff_sws_init_single_context, utils.c:1137-1835 */
{
    int srcW = c->o.src_w, srcH = c->o.src_h, dstW = c->o.dst_w, dstH = c->o.dst_h;
    int flags, unscaled, i, srcFormat, dstFormat, ret;
    const Desc *ds, *dd;
    int64_t lumXInc, lumYInc, chrXInc, chrYInc;
    int lum_scaler, chr_scaler;

    flags = c->o.flags;
    unscaled = srcW == dstW && srcH == dstH;

    if (!c->contrast && !c->saturation && !c->dstFormatBpp)
        or_sws_set_colorspace(c, yuv2rgb_coeffs[5], c->o.src_range, yuv2rgb_coeffs[5],
                              c->o.dst_range, 0, 1 << 16, 1 << 16);
    handle_formats(c);
    srcFormat = c->o.src_format; dstFormat = c->o.dst_format;
    ds = desc_get(srcFormat); dd = desc_get(dstFormat);
    if (!ds || !dd) return -1;
    if (isRGB4bits(srcFormat)) return -1;   /* rgb4 / bgr4: outputs only (format.c legacy_format_entries) */
    if (isInputOnly(dstFormat)) return -1;   /* "... is not supported as output pixel format" (utils.c:1198-1208) */   /* palette-expanded inputs (usePal, swscale_internal.h:936-953): not restated */

    i = flags & (OR_SWS_POINT | OR_SWS_AREA | OR_SWS_BILINEAR | OR_SWS_FAST_BILINEAR | OR_SWS_BICUBIC |
                 OR_SWS_X | OR_SWS_GAUSS | OR_SWS_LANCZOS | OR_SWS_SINC | OR_SWS_SPLINE | OR_SWS_BICUBLIN);
    if (!i) { i = OR_SWS_BICUBIC; flags |= i; c->o.flags = flags; }
    else if (i & (i - 1)) return -1;
    if (i == OR_SWS_FAST_BILINEAR) {
        if (srcW < 8 || dstW <= 8) { i = OR_SWS_BILINEAR; flags ^= OR_SWS_FAST_BILINEAR | i; c->o.flags = flags; }
    }
    lum_scaler = i == OR_SWS_BICUBLIN ? OR_SWS_BICUBIC : i;
    chr_scaler = i == OR_SWS_BICUBLIN ? OR_SWS_BILINEAR : i;

    if (srcW < 1 || srcH < 1 || dstW < 1 || dstH < 1) return -1;

    lumXInc = (((int64_t)srcW << 16) + (dstW >> 1)) / dstW;
    lumYInc = (((int64_t)srcH << 16) + (dstH >> 1)) / dstH;
    c->dstFormatBpp = bits_per_pixel(dd);
    c->srcFormatBpp = bits_per_pixel(ds);

    c->chrSrcHSub = ds->lw; c->chrSrcVSub = ds->lh;
    c->chrDstHSub = dd->lw; c->chrDstVSub = dd->lh;

    if (isAnyRGB(dstFormat) && !(flags & OR_SWS_FULL_CHR_H_INT)) { /* :1270-1286 */
        if (dstW & 1) { flags |= OR_SWS_FULL_CHR_H_INT; c->o.flags = flags; }
        if (c->chrSrcHSub == 0 && c->chrSrcVSub == 0 && c->o.dither != 2 && !(c->o.flags & OR_SWS_FAST_BILINEAR)) {
            flags |= OR_SWS_FULL_CHR_H_INT; c->o.flags = flags;
        }
    }
    if (c->o.dither == 1 && (flags & OR_SWS_ERROR_DIFFUSION)) c->o.dither = 3;   /* :1288-1291 */
    if (isRGB8class(dstFormat)) {   /* :1293-1316: ordered dither only with chroma pairs, everything else only with full chroma */
        if (c->o.dither == 1) c->o.dither = (flags & OR_SWS_FULL_CHR_H_INT) ? 3 : 2;
        if (!(flags & OR_SWS_FULL_CHR_H_INT) && (c->o.dither == 3 || c->o.dither == 4 || c->o.dither == 5 || c->o.dither == 0)) {
            flags |= OR_SWS_FULL_CHR_H_INT; c->o.flags = flags;
        }
        if ((flags & OR_SWS_FULL_CHR_H_INT) && c->o.dither == 2) c->o.dither = 3;
    }
    if (isPlanarRGB(dstFormat) && !(flags & OR_SWS_FULL_CHR_H_INT)) { flags |= OR_SWS_FULL_CHR_H_INT; c->o.flags = flags; }
    if ((flags & OR_SWS_FULL_CHR_H_INT) && (isRGB16(dstFormat) || isMono(dstFormat) || isRGB4bits(dstFormat))) { /* "full chroma interpolation ... not yet implemented" (:1325-1358) */
        flags &= ~OR_SWS_FULL_CHR_H_INT; c->o.flags = flags;
    }
    if (isAnyRGB(dstFormat) && !(flags & OR_SWS_FULL_CHR_H_INT)) c->chrDstHSub = 1; /* :1359 */

    c->chrSrcVSub += (flags & 0x30000) >> 16; /* vChrDrop: "drop some chroma lines if the user wants it" (utils.c:1362-1365) */

    if (isAnyRGB(srcFormat) && !(srcW & 1) && !(flags & OR_SWS_FULL_CHR_H_INP) && !isRGB8class(srcFormat) &&
        !(isPlanarRGB(srcFormat) && ds->c[0].depth > 8) && /* gbrp9..16, gbrpf32: no _half readers (:1369-1388) */
        ((dstW >> c->chrDstHSub) <= (srcW >> 1) || (flags & OR_SWS_FAST_BILINEAR))) /* :1369-1390 */
        c->chrSrcHSub = 1;

    c->chrSrcW = CEIL_RSHIFT(srcW, c->chrSrcHSub);
    c->chrSrcH = CEIL_RSHIFT(srcH, c->chrSrcVSub);
    c->chrDstW = CEIL_RSHIFT(dstW, c->chrDstHSub);
    c->chrDstH = CEIL_RSHIFT(dstH, c->chrDstVSub);

    c->srcBpc = ds->c[0].depth; if (c->srcBpc < 8) c->srcBpc = 8;
    c->dstBpc = dd->c[0].depth; if (c->dstBpc < 8) c->dstBpc = 8;
    if (isAnyRGB(srcFormat) || srcFormat == ORF_PAL8) c->srcBpc = 16;
    if (isFloat(srcFormat) && !isAnyRGB(srcFormat)) c->srcBpc = 16;   /* "float will be converted to uint16_t" (utils.c:1558-1563; the unscaled exceptions never reach the scaler) */

    chrXInc = (((int64_t)c->chrSrcW << 16) + (c->chrDstW >> 1)) / c->chrDstW;
    chrYInc = (((int64_t)c->chrSrcH << 16) + (c->chrDstH >> 1)) / c->chrDstH;
    if (chrXInc < 10 || chrXInc > 0x7fffffff || chrYInc < 10 || chrYInc > 0x7fffffff ||
        lumXInc < 10 || lumXInc > 0x7fffffff || lumYInc < 10 || lumYInc > 0x7fffffff)
        return -1;
    c->lumXInc = (int)lumXInc; c->lumYInc = (int)lumYInc; c->chrXInc = (int)chrXInc; c->chrYInc = (int)chrYInc;

    if (!unscaled && c->o.gamma_flag && (c->o.src_format != ORF_RGBA64LE || c->o.dst_format != ORF_RGBA64LE || c->src_be || c->dst_be)) {
        /* utils.c:1461-1522: source -> RGBA64LE (same size), RGBA64LE scaled between pow(x, 1/2.2) and pow(x, 2.2) table passes, RGBA64LE ->
         * destination (same size).  Children are plain sws_getContext() contexts; the filters go to the scaling step. */
        int k;
        c->casc_gamma = 1;
        c->casc_stride[0] = (((srcW + 7) & ~7) * 8 + 63) & ~63;
        c->casc_tmp[0] = calloc((size_t)c->casc_stride[0] * srcH + 64, 1);
        c->cascade[0] = alloc_set_opts(srcW, srcH, c->o.src_format, srcW, srcH, ORF_RGBA64LE, c->o.flags, c->o.scaler_params);
        c->cascade[1] = alloc_set_opts(srcW, srcH, ORF_RGBA64LE, dstW, dstH, ORF_RGBA64LE, c->o.flags, c->o.scaler_params);
        for (k = 0; k < 4; k++) {
            c->cascade[1]->o.src_vec[k] = c->o.src_vec[k]; c->cascade[1]->o.src_vec_len[k] = c->o.src_vec_len[k];
            c->cascade[1]->o.dst_vec_len[k] = c->o.dst_vec_len[k];
        }
        if (c->o.dst_format != ORF_RGBA64LE || c->dst_be) {
            c->casc_stride2 = (((dstW + 7) & ~7) * 8 + 63) & ~63;
            c->casc_tmp2 = calloc((size_t)c->casc_stride2 * dstH + 64, 1);
            c->cascade[2] = alloc_set_opts(dstW, dstH, ORF_RGBA64LE, dstW, dstH, c->o.dst_format, c->o.flags, c->o.scaler_params);
        }
        if (init_context(c->cascade[0]) < 0 || init_context(c->cascade[1]) < 0 || (c->cascade[2] && init_context(c->cascade[2]) < 0)) {
            or_sws_free(c->cascade[0]); or_sws_free(c->cascade[1]); or_sws_free(c->cascade[2]);
            free(c->casc_tmp[0]); free(c->casc_tmp2);
            c->cascade[0] = c->cascade[1] = c->cascade[2] = NULL; c->casc_tmp[0] = c->casc_tmp2 = NULL; c->casc_gamma = 0;
            return -1;
        }
        /* alloc_gamma_tbl (utils.c:1046-1058): tbl[i] = pow(i / 65535.0, e) * 65535.0, converted to uint16_t by the assignment */
        c->gamma_tab = malloc(65536 * sizeof(uint16_t)); c->inv_gamma_tab = malloc(65536 * sizeof(uint16_t));
        for (k = 0; k < 65536; k++) {
            c->gamma_tab[k] = (uint16_t)(pow(k / 65535.0, 2.2) * 65535.0);
            c->inv_gamma_tab[k] = (uint16_t)(pow(k / 65535.0, 1.f / 2.2) * 65535.0);
        }
        c->initialized = 1;
        return 0;
    }

    if (isBayer(srcFormat)) {   /* utils.c:1524-1550: anything but the three direct conversions goes through rgb24 / rgb48 at the source size */
        if (srcH < 2) return -1;   /* (bayer_to_*_wrapper: av_assert0(srcSliceH > 1)) */
        if (!unscaled || c->dst_be || (dstFormat != ORF_RGB24 && dstFormat != ORF_YUV420P && dstFormat != ORF_RGB48LE)) {
            const int tmpFormat = ds->c[1].depth == 8 ? ORF_RGB48LE : ORF_RGB24;   /* isBayer16BPS */
            int k;
            c->casc_stride[0] = (srcW * desc_get(tmpFormat)->c[0].step + 63) & ~63;
            c->casc_tmp[0] = calloc((size_t)c->casc_stride[0] * srcH + 64, 1);
            c->cascade[0] = alloc_set_opts(srcW, srcH, srcFormat, srcW, srcH, tmpFormat, flags, c->o.scaler_params);
            c->cascade[1] = alloc_set_opts(srcW, srcH, tmpFormat, dstW, dstH, dstFormat, flags, c->o.scaler_params);   /* (byte order is this context's business) */
            for (k = 0; k < 4; k++) {   /* srcFilter to the first step, dstFilter to the second */
                c->cascade[0]->o.src_vec[k] = c->o.src_vec[k]; c->cascade[0]->o.src_vec_len[k] = c->o.src_vec_len[k];
                c->cascade[1]->o.dst_vec_len[k] = c->o.dst_vec_len[k];
            }
            if (init_context(c->cascade[0]) < 0 || init_context(c->cascade[1]) < 0) {
                or_sws_free(c->cascade[0]); or_sws_free(c->cascade[1]); free(c->casc_tmp[0]);
                c->cascade[0] = c->cascade[1] = NULL; c->casc_tmp[0] = NULL;
                return -1;
            }
            c->initialized = 1;
            return 0;
        }
    }
    /* alpha: src alpha dropped -> reference cascades through alpha blend only if alpha_blend != NONE (default NONE) */
    c->needAlpha = isALPHA(srcFormat) && isALPHA(dstFormat);
    if (isRGB8class(dstFormat) || isMono(dstFormat))   /* utils.c:1744-1747 (allocated for every context there; only these writers use them) */
        for (i = 0; i < 3; i++) if (!c->dither_error[i]) c->dither_error[i] = calloc((size_t)dstW + 3, sizeof(int));

    const int usesHFilter = (c->o.src_vec[0] && c->o.src_vec_len[0] > 1) || (c->o.src_vec[2] && c->o.src_vec_len[2] > 1) ||
                            c->o.dst_vec_len[0] > 1 || c->o.dst_vec_len[2] > 1;   /* utils.c:1256-1263 */
    const int usesVFilter = (c->o.src_vec[1] && c->o.src_vec_len[1] > 1) || (c->o.src_vec[3] && c->o.src_vec_len[3] > 1) ||
                            c->o.dst_vec_len[1] > 1 || c->o.dst_vec_len[3] > 1;
    if (isALPHA(srcFormat) && !isALPHA(dstFormat)) {   /* utils.c:1565-1601 */
        const int tmpFormat = alphaless_fmt(srcFormat);
        if (tmpFormat != ORF_NONE && c->o.alpha_blend != 0) {
            if (!unscaled || dstFormat != tmpFormat || c->dst_be || usesHFilter || usesVFilter || c->o.src_range != c->o.dst_range) {
                /* blend at the source size into the alpha-less twin of the source format, then convert / scale that */
                const Desc *td = desc_get(tmpFormat);
                const int np = (td->flags & PF_PLANAR) ? td->nb : 1, aw = (srcW + 7) & ~7;
                int k;
                c->casc_mainindex = 1;
                for (k = 0; k < np; k++) {
                    const int chroma = np > 1 && (k == 1 || k == 2) && !(td->flags & PF_RGB);
                    const int pw = chroma ? CEIL_RSHIFT(aw, td->lw) : aw, ph = chroma ? CEIL_RSHIFT(srcH, td->lh) : srcH;
                    c->casc_stride[k] = (pw * (np > 1 ? (td->c[0].depth > 8 ? 2 : 1) : td->c[0].step) + 63) & ~63;
                    c->casc_tmp[k] = calloc((size_t)c->casc_stride[k] * ph + 64, 1);
                }
                c->cascade[0] = alloc_set_opts(srcW, srcH, srcFormat, srcW, srcH, tmpFormat, flags, c->o.scaler_params);
                c->cascade[0]->o.alpha_blend = c->o.alpha_blend;
                c->cascade[1] = alloc_set_opts(srcW, srcH, tmpFormat, dstW, dstH, dstFormat, flags, c->o.scaler_params);
                c->cascade[1]->o.src_range = c->o.src_range; c->cascade[1]->o.dst_range = c->o.dst_range;
                /* the second step is given the caller's destination format in the reference, byte order included (its temporary is native): its own rules see
                 * "native -> the other byte order" (bswap_16bpc with SWS_SRC_V_CHR_DROP among them), so it writes the caller's byte order itself */
                c->cascade[1]->dst_be = c->dst_be; c->casc_child_dst_be = c->dst_be;
                for (k = 0; k < 4; k++) {
                    c->cascade[1]->o.src_vec[k] = c->o.src_vec[k]; c->cascade[1]->o.src_vec_len[k] = c->o.src_vec_len[k];
                    c->cascade[1]->o.dst_vec_len[k] = c->o.dst_vec_len[k];
                }
                if (init_context(c->cascade[0]) < 0 || init_context(c->cascade[1]) < 0) {
                    or_sws_free(c->cascade[0]); or_sws_free(c->cascade[1]);
                    for (k = 0; k < 4; k++) { free(c->casc_tmp[k]); c->casc_tmp[k] = NULL; }
                    c->cascade[0] = c->cascade[1] = NULL; c->casc_mainindex = 0;
                    return -1;
                }
                c->initialized = 1;
                return 0;
            }
        }
    }
    /* "alpha blend special case, note this has been split via cascaded contexts if its scaled" (utils.c:1603-1616) */
    if (unscaled && !usesHFilter && !usesVFilter && c->o.alpha_blend != 0 && isALPHA(srcFormat) &&
        (c->o.src_range == c->o.dst_range || isAnyRGB(dstFormat)) && alphaless_fmt(srcFormat) == dstFormat && !c->dst_be) {
        c->unscaled_kind = UNSC_ALPHABLEND;
        c->initialized = 1;
        return 0;
    }
    if (unscaled && !usesHFilter && !usesVFilter &&
        (c->o.src_range == c->o.dst_range || isAnyRGB(dstFormat) || isFloat(srcFormat) || isFloat(dstFormat) || isBayer(srcFormat))) {
        get_unscaled(c);
        if (c->unscaled_kind == UNSC_REFUSE) return -1;
        if (c->unscaled_kind) { c->initialized = 1; return 0; }
    }
    if (isBayer(srcFormat)) return -1;   /* (a source filter on one of the three direct conversions: there is no scaler reader to fall back on) */

    /* filters (:1675-1735), filterAlign == 1 in the C-only build */
    {
        int r[4];
        r[0] = init_filter(&c->hLumFilter, &c->hLumFilterPos, &c->hLumFilterSize, c->lumXInc, srcW, dstW, 1, 1 << 14,
                           lum_scaler, flags, c->o.scaler_params, get_local_pos(0, 0), get_local_pos(0, 0),
                           c->o.src_vec[0], c->o.src_vec_len[0], c->o.dst_vec_len[0]);
        r[1] = r[0] < 0 ? r[0] : init_filter(&c->hChrFilter, &c->hChrFilterPos, &c->hChrFilterSize, c->chrXInc, c->chrSrcW, c->chrDstW, 1, 1 << 14,
                           chr_scaler, flags, c->o.scaler_params,
                           get_local_pos(c->chrSrcHSub, c->o.src_h_chr_pos), get_local_pos(c->chrDstHSub, c->o.dst_h_chr_pos),
                           c->o.src_vec[2], c->o.src_vec_len[2], c->o.dst_vec_len[2]);
        /* vertical: a cascade request of the luma filter is remembered while the chroma filter is still built (:1719-1735) */
        r[2] = r[1] < 0 ? r[1] : init_filter(&c->vLumFilter, &c->vLumFilterPos, &c->vLumFilterSize, c->lumYInc, srcH, dstH, 1, 1 << 12,
                           lum_scaler, flags, c->o.scaler_params, get_local_pos(0, 0), get_local_pos(0, 0),
                           c->o.src_vec[1], c->o.src_vec_len[1], c->o.dst_vec_len[1]);
        r[3] = (r[2] < 0 && r[2] != RET_CASCADE) ? r[2] : init_filter(&c->vChrFilter, &c->vChrFilterPos, &c->vChrFilterSize, c->chrYInc, c->chrSrcH, c->chrDstH, 1, 1 << 12,
                           chr_scaler, flags, c->o.scaler_params,
                           get_local_pos(c->chrSrcVSub, c->o.src_v_chr_pos), get_local_pos(c->chrDstVSub, c->o.dst_v_chr_pos),
                           c->o.src_vec[3], c->o.src_vec_len[3], c->o.dst_vec_len[3]);
        ret = r[3] < 0 ? r[3] : r[2];
    }
    if (ret == RET_CASCADE) {
        /* utils.c:1803-1833: two steps through a yuv420p / yuva420p picture of the geometric-mean size.  The children are plain
         * sws_getContext() contexts (flags and scaler parameters only); srcFilter goes to the first, dstFilter to the second. */
        const int tmpW = (int)sqrt((double)(srcW * (int64_t)dstW)), tmpH = (int)sqrt((double)(srcH * (int64_t)dstH));
        const int tmpFormat = isALPHA(srcFormat) ? ORF_YUVA420P : ORF_YUV420P;
        const int aw = (tmpW + 7) & ~7;                      /* av_image_alloc(..., 64): linesizes of the width rounded up to 8, aligned to 64 */
        const int np = tmpFormat == ORF_YUVA420P ? 4 : 3;
        int k;
        if (srcW * (int64_t)srcH <= 4LL * dstW * dstH) return -1;
        for (k = 0; k < np; k++) {
            const int chroma = k == 1 || k == 2;
            const int rows = chroma ? (tmpH + 1) >> 1 : tmpH;
            c->casc_stride[k] = ((chroma ? (aw + 1) >> 1 : aw) + 63) & ~63;
            c->casc_tmp[k] = calloc((size_t)c->casc_stride[k] * rows + 64, 1);
        }
        c->cascade[0] = alloc_set_opts(srcW, srcH, c->o.src_format, tmpW, tmpH, tmpFormat, c->o.flags, c->o.scaler_params);
        for (k = 0; k < 4; k++) { c->cascade[0]->o.src_vec[k] = c->o.src_vec[k]; c->cascade[0]->o.src_vec_len[k] = c->o.src_vec_len[k]; }
        c->cascade[1] = alloc_set_opts(tmpW, tmpH, tmpFormat, dstW, dstH, c->o.dst_format, c->o.flags, c->o.scaler_params);
        for (k = 0; k < 4; k++) c->cascade[1]->o.dst_vec_len[k] = c->o.dst_vec_len[k];
        if (init_context(c->cascade[0]) < 0 || init_context(c->cascade[1]) < 0) {
            or_sws_free(c->cascade[0]); or_sws_free(c->cascade[1]);
            for (k = 0; k < 4; k++) { free(c->casc_tmp[k]); c->casc_tmp[k] = NULL; }
            c->cascade[0] = c->cascade[1] = NULL;
            return -1;
        }
        c->initialized = 1;
        return 0;
    }
    if (ret < 0) return -1;

    init_range_convert(c); /* ff_sws_init_scale -> sws_init_swscale, swscale.c:662-695 */
    c->initialized = 1;
    return 0;
}

static int range_override_needed(int f) { return !isYUV(f) && !isGray(f); }

int or_sws_set_colorspace(OrSws *c, const int inv_table[4], int srcRange, const int table[4],
                          int dstRange, int brightness, int contrast, int saturation) /* utils.c:849-1005 */
{
    int need_reinit = 0;
    const Desc *dd, *ds;

    handle_formats(c);
    dd = desc_get(c->o.dst_format); ds = desc_get(c->o.src_format);
    if (range_override_needed(c->o.dst_format)) dstRange = 0;
    if (range_override_needed(c->o.src_format)) srcRange = 0;

    if (c->o.src_range != srcRange || c->o.dst_range != dstRange || c->brightness != brightness ||
        c->contrast != contrast || c->saturation != saturation ||
        memcmp(c->srcColorspaceTable, inv_table, sizeof(int) * 4) ||
        memcmp(c->dstColorspaceTable, table, sizeof(int) * 4))
        need_reinit = 1;

    memmove(c->srcColorspaceTable, inv_table, sizeof(int) * 4);
    memmove(c->dstColorspaceTable, table, sizeof(int) * 4);
    c->brightness = brightness; c->contrast = contrast; c->saturation = saturation;
    c->o.src_range = srcRange; c->o.dst_range = dstRange;

    if (need_reinit) init_range_convert(c);

    c->dstFormatBpp = bits_per_pixel(dd);
    c->srcFormatBpp = bits_per_pixel(ds);

    if (c->cascade[c->casc_mainindex])
        return or_sws_set_colorspace(c->cascade[c->casc_mainindex], inv_table, srcRange, table, dstRange, brightness, contrast, saturation);
    if (!need_reinit) return 0;

    if ((isYUV(c->o.dst_format) || isGray(c->o.dst_format)) && (isYUV(c->o.src_format) || isGray(c->o.src_format))) {
        if (!c->cascade[0] && memcmp(c->dstColorspaceTable, c->srcColorspaceTable, sizeof(int) * 4) &&
            c->o.src_w && c->o.src_h && c->o.dst_w && c->o.dst_h) { /* :915-984 */
            int tmp_format, tmp_w, tmp_h, srcW = c->o.src_w, srcH = c->o.src_h, dstW = c->o.dst_w, dstH = c->o.dst_h;
            const int both_alpha = isALPHA(c->o.src_format) && isALPHA(c->o.dst_format);
            if (isNBPS(c->o.dst_format) || is16BPS(c->o.dst_format)) tmp_format = both_alpha ? ORF_BGRA64LE : ORF_BGR48LE; /* :927-933 */
            else tmp_format = both_alpha ? ORF_BGRA : ORF_BGR24;                                                       /* :934-940 */
            if (srcW * srcH > dstW * dstH) { tmp_w = dstW; tmp_h = dstH; } else { tmp_w = srcW; tmp_h = srcH; }
            c->casc_stride[0] = (tmp_w * desc_get(tmp_format)->c[0].step + 63) & ~63;   /* av_image_alloc(..., 64) */
            c->casc_tmp[0] = calloc((size_t)c->casc_stride[0] * tmp_h + 64, 1);   /* av_image_alloc leaves it uninitialised; the pair-wise yuv2rgb
                                                                                   * converters never write the last pixel of an odd width: zero here and in the product */

            c->cascade[0] = alloc_set_opts(srcW, srcH, c->o.src_format, tmp_w, tmp_h, tmp_format, c->o.flags, c->o.scaler_params);
            if (init_context(c->cascade[0]) < 0) goto casc_fail;
            or_sws_set_colorspace(c->cascade[0], inv_table, srcRange, table, dstRange, brightness, contrast, saturation);

            c->cascade[1] = alloc_set_opts(tmp_w, tmp_h, tmp_format, dstW, dstH, c->o.dst_format, c->o.flags, c->o.scaler_params);
            c->cascade[1]->o.src_range = srcRange;
            c->cascade[1]->o.dst_range = dstRange;
            if (init_context(c->cascade[1]) < 0) {
            casc_fail:  /* the reference returns the error with the half-built children in place; the oracle drops them so that a
                         * later or_sws_scale() takes the plain path instead of a broken cascade */
                or_sws_free(c->cascade[0]); or_sws_free(c->cascade[1]); free(c->casc_tmp[0]);
                c->cascade[0] = c->cascade[1] = NULL; c->casc_tmp[0] = NULL;
                return -1;
            }
            or_sws_set_colorspace(c->cascade[1], inv_table, srcRange, table, dstRange, 0, 1 << 16, 1 << 16);
            return 0;
        }
        if (c->cascade[0] && memcmp(c->dstColorspaceTable, c->srcColorspaceTable, sizeof(int) * 4)) return -1;
        return 0;
    }
    if (!isYUV(c->o.dst_format) && !isGray(c->o.dst_format))
        yuv2rgb_init_tables(c, inv_table, srcRange, brightness, contrast, saturation);
    fill_rgb2yuv_table(c, table);
    return 0;
}

/* ------------------------------------------------------------------ */
/* unscaled converters                                                 */
/* ------------------------------------------------------------------ */
static inline uint32_t lut_at(const OrSws *c, int idx)
{
    if (c->lut_elem == 1) return c->yuvTable[idx];
    if (c->lut_elem == 2) return ((const uint16_t *)c->yuvTable)[idx];
    return ((const uint32_t *)c->yuvTable)[idx];
}

/* ff_dither_8x8_32 / ff_dither_8x8_73 (output.c:60-82), nine rows like the reference's tables (row 8 = row 0) */
static const uint8_t dither_8x8_32[9][8] = {
    { 17, 9, 23, 15, 16, 8, 22, 14 }, { 5, 29, 3, 27, 4, 28, 2, 26 }, { 21, 13, 19, 11, 20, 12, 18, 10 }, { 0, 24, 6, 30, 1, 25, 7, 31 },
    { 16, 8, 22, 14, 17, 9, 23, 15 }, { 4, 28, 2, 26, 5, 29, 3, 27 }, { 20, 12, 18, 10, 21, 13, 19, 11 }, { 1, 25, 7, 31, 0, 24, 6, 30 },
    { 17, 9, 23, 15, 16, 8, 22, 14 },
};
static const uint8_t dither_8x8_73[9][8] = {
    { 0, 55, 14, 68, 3, 58, 17, 72 }, { 37, 18, 50, 32, 40, 22, 54, 35 }, { 9, 64, 5, 59, 13, 67, 8, 63 }, { 46, 27, 41, 23, 49, 31, 44, 26 },
    { 2, 57, 16, 71, 1, 56, 15, 70 }, { 39, 21, 52, 34, 38, 19, 51, 33 }, { 11, 66, 7, 62, 10, 65, 6, 60 }, { 48, 30, 43, 25, 47, 29, 42, 24 },
    { 0, 55, 14, 68, 3, 58, 17, 72 },
};
static const uint8_t dither_8x8_220_w[9][8] = {   /* ff_dither_8x8_220 (output.c:84-95, the `#if 1` variant) */
    { 117, 62, 158, 103, 113, 58, 155, 100 }, { 34, 199, 21, 186, 31, 196, 17, 182 }, { 144, 89, 131, 76, 141, 86, 127, 72 },
    { 0, 165, 41, 206, 10, 175, 52, 217 }, { 110, 55, 151, 96, 120, 65, 162, 107 }, { 28, 193, 14, 179, 38, 203, 24, 189 },
    { 138, 83, 124, 69, 148, 93, 134, 79 }, { 7, 172, 48, 213, 3, 168, 45, 210 }, { 117, 62, 158, 103, 113, 58, 155, 100 },
};

/* the three ordered-dither offsets of pixel column x in row y for the 8 / 4 bpp tables (yuv2rgb_write output.c:1762-1776; the same
 * rows LOADDITHER8 / LOADDITHER4D / LOADDITHER4DB pick in yuv2rgb.c:415-455) */
static void dither_rgb8_rows(int fmt, int row, int col, int *dr, int *dg, int *db)   /* row 0..8, col 0..7 */
{
    if (fmt == ORF_RGB8 || fmt == ORF_BGR8) { *dr = *dg = dither_8x8_32[row][col]; *db = dither_8x8_73[row][col]; }
    else { *dr = *db = dither_8x8_220_w[row][col]; *dg = dither_8x8_73[row][col]; }
}
static void dither_rgb8(int fmt, int y, int x, int *dr, int *dg, int *db) { dither_rgb8_rows(fmt, y & 7, x & 7, dr, dg, db); }

/* ordered-dither rows of output.c:40-58 (ff_dither_2x2_4, ff_dither_2x2_8, ff_dither_4x4_16) */
static const uint8_t dither_2x2_4[3][8] = { { 1, 3, 1, 3, 1, 3, 1, 3 }, { 2, 0, 2, 0, 2, 0, 2, 0 }, { 1, 3, 1, 3, 1, 3, 1, 3 } };
static const uint8_t dither_2x2_8[3][8] = { { 6, 2, 6, 2, 6, 2, 6, 2 }, { 0, 4, 0, 4, 0, 4, 0, 4 }, { 6, 2, 6, 2, 6, 2, 6, 2 } };
static const uint8_t dither_4x4_16[5][8] = { { 8, 4, 11, 7, 8, 4, 11, 7 }, { 2, 14, 1, 13, 2, 14, 1, 13 }, { 10, 6, 9, 5, 10, 6, 9, 5 },
                                             { 0, 12, 3, 15, 0, 12, 3, 15 }, { 8, 4, 11, 7, 8, 4, 11, 7 } };

/* YUV420FUNC/YUV422FUNC + PUTRGB24/PUTBGR24/PUTRGB, yuv2rgb.c:68-559.
 * The 8/4/2-pixel block structure of the macros reduces to: pixel pair i on
 * rows (2k, 2k+1) uses chroma sample i of chroma row k (420) / row y (422);
 * widths that are not a multiple of 2 leave the last column untouched. */
static int unscaled_yuv2rgb(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                            int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    const int is422 = c->o.src_format == ORF_YUV422P;
    const int d = c->o.dst_format;
    const int npairs = ((c->o.dst_w >> 3) << 2) + ((c->o.dst_w & 4) ? 2 : 0) + ((c->o.dst_w & 2) ? 1 : 0);
    if ((d == ORF_RGB24 || d == ORF_BGR24) && c->lut_elem == 1 && !is422 && !(srcSliceH & 1)) {
        /* yuv2rgb_c_24_rgb / yuv2rgb_c_24_bgr (yuv2rgb.c:237-281) in the reference's own loop shape -- LOADCHROMA once per 2x2 block (three
         * table pointers), PUTRGB24 for both lines -- so that the timed CPU baseline (bench.py cpu_baseline, BASELINE config C2a) runs the
         * algorithm at the speed the reference's C path does.  Same tables, same results as the general loop below. */
        const uint8_t *tab = c->yuvTable;
        const int ro = d == ORF_RGB24 ? 0 : 2, bo = 2 - ro;
        for (int y = 0; y < srcSliceH; y += 2) {
            const uint8_t *py1 = src[0] + (ptrdiff_t)y * srcStride[0], *py2 = py1 + srcStride[0];
            const uint8_t *pu = src[1] + (ptrdiff_t)(y >> 1) * srcStride[1], *pv = src[2] + (ptrdiff_t)(y >> 1) * srcStride[2];
            uint8_t *d1 = dst[0] + (ptrdiff_t)(y + srcSliceY) * dstStride[0], *d2 = d1 + dstStride[0];
            for (int i = 0; i < npairs; i++) {
                const int U = pu[i], V = pv[i];
                const uint8_t *r = tab + c->table_rV[V + HEADROOM], *g = tab + c->table_gU[U + HEADROOM] + c->table_gV[V + HEADROOM], *b = tab + c->table_bU[U + HEADROOM];
                int Y;
                Y = py1[2 * i];     d1[6 * i + ro] = r[Y]; d1[6 * i + 1] = g[Y]; d1[6 * i + bo] = b[Y];
                Y = py1[2 * i + 1]; d1[6 * i + 3 + ro] = r[Y]; d1[6 * i + 4] = g[Y]; d1[6 * i + 3 + bo] = b[Y];
                Y = py2[2 * i];     d2[6 * i + ro] = r[Y]; d2[6 * i + 1] = g[Y]; d2[6 * i + bo] = b[Y];
                Y = py2[2 * i + 1]; d2[6 * i + 3 + ro] = r[Y]; d2[6 * i + 4] = g[Y]; d2[6 * i + 3 + bo] = b[Y];
            }
        }
        return srcSliceH;
    }
    for (int y = 0; y < srcSliceH; y += 2) {
        for (int l = 0; l < 2; l++) {
            int yy = y + l;
            const uint8_t *py = src[0] + yy * srcStride[0];
            const uint8_t *pu = src[1] + (is422 ? yy : (y >> 1)) * srcStride[1];
            const uint8_t *pv = src[2] + (is422 ? yy : (y >> 1)) * srcStride[2];
            uint8_t *out = dst[0] + (yy + srcSliceY) * dstStride[0];
            for (int i = 0; i < npairs; i++) {
                int U = pu[i], V = pv[i];
                int r = c->table_rV[V + HEADROOM];
                int g = c->table_gU[U + HEADROOM] + c->table_gV[V + HEADROOM];
                int b = c->table_bU[U + HEADROOM];
                for (int k = 0; k < 2; k++) {
                    int Y = py[2 * i + k];
                    if (isRGB16(d)) { /* YUV420FUNC_DITHER / YUV422FUNC_DITHER + PUTRGB16/15/12 (yuv2rgb.c:283-330, :371-411): the dither row is
                                        * chosen by the loop's slice-relative even y, the second line of a pair reads the following table row */
                        const int e = k, o = 2 * (i & 3);
                        int dr, dg, db;
                        uint16_t v;
                        if (d == ORF_RGB565LE || d == ORF_BGR565LE) { dr = dither_2x2_8[l][o + e]; dg = dither_2x2_4[l][o + e]; db = dither_2x2_8[1 + l][o + e]; }
                        else if (d == ORF_RGB555LE || d == ORF_BGR555LE) { dr = dither_2x2_8[l][o + e]; dg = dither_2x2_8[l][o + (e ^ 1)]; db = dither_2x2_8[1 + l][o + e]; }
                        else { dr = dg = db = dither_4x4_16[(y & 3) + l][o + e]; }
                        v = (uint16_t)(lut_at(c, r + Y + dr) + lut_at(c, g + Y + dg) + lut_at(c, b + Y + db));
                        memcpy(out + 4 * i + 2 * k, &v, 2);
                    } else if (isRGB8class(d) || isRGB4bits(d)) { /* PUTRGB8 / PUTRGB4D / PUTRGB4DB with LOADDITHER8 / 4D / 4DB (yuv2rgb.c:413-455): rows
                                                                   * of the 8x8 tables by the ABSOLUTE even row, the second line reads the following row */
                        int dr, dg, db;
                        uint8_t v;
                        /* (the 2-pixel tail behind a 4-pixel tail -- dst_w & 6 == 6 -- starts its dither row over at column 0: every section of YUV420FUNC_DITHER begins
                         *  with PUTFUNC(1, 0, 0), yuv2rgb.c:283-318; found against the real reference by tools/ref/ref_crosscheck.py, round 6) */
                        const int tail2 = (c->o.dst_w & 6) == 6 && i == npairs - 1;
                        dither_rgb8_rows(d, ((y + srcSliceY) & 7) + l, (tail2 ? 0 : 2 * (i & 3)) + k, &dr, &dg, &db);
                        v = (uint8_t)(lut_at(c, r + Y + dr) + lut_at(c, g + Y + dg) + lut_at(c, b + Y + db));
                        if (isRGB4bits(d)) { if (!k) out[i] = v; else out[i] = (uint8_t)(out[i] | (v << 4)); }
                        else out[2 * i + k] = v;
                    } else if (d == ORF_RGB48LE || d == ORF_BGR48LE) { /* PUTRGB48 / PUTBGR48 yuv2rgb.c:107-125: each 8-bit LUT value fills both bytes */
                        uint8_t R = (uint8_t)lut_at(c, r + Y), G = (uint8_t)lut_at(c, g + Y), B = (uint8_t)lut_at(c, b + Y);
                        uint8_t *p = out + 12 * i + 6 * k;
                        uint8_t first = d == ORF_RGB48LE ? R : B, third = d == ORF_RGB48LE ? B : R;
                        p[0] = p[1] = first; p[2] = p[3] = G; p[4] = p[5] = third;
                    } else if (d == ORF_GBRP) { /* PUTGBRP yuv2rgb.c:127-135 */
                        dst[0][(ptrdiff_t)(yy + srcSliceY) * dstStride[0] + 2 * i + k] = (uint8_t)lut_at(c, g + Y);
                        dst[1][(ptrdiff_t)(yy + srcSliceY) * dstStride[1] + 2 * i + k] = (uint8_t)lut_at(c, b + Y);
                        dst[2][(ptrdiff_t)(yy + srcSliceY) * dstStride[2] + 2 * i + k] = (uint8_t)lut_at(c, r + Y);
                    } else if (d == ORF_RGB24 || d == ORF_BGR24) {
                        uint8_t R = (uint8_t)lut_at(c, r + Y), G = (uint8_t)lut_at(c, g + Y), B = (uint8_t)lut_at(c, b + Y);
                        uint8_t *p = out + 6 * i + 3 * k;
                        if (d == ORF_RGB24) { p[0] = R; p[1] = G; p[2] = B; }
                        else { p[0] = B; p[1] = G; p[2] = R; }
                    } else {
                        uint32_t v = lut_at(c, r + Y) + lut_at(c, g + Y) + lut_at(c, b + Y);
                        if (isALPHA(c->o.src_format)) { /* yuva2rgba_c / yuva2argb_c: PUTRGBA yuv2rgb.c:101-105, :524-528 */
                            const int abase = (d == ORF_ARGB || d == ORF_ABGR) ? 0 : 24;
                            v += (uint32_t)src[3][(ptrdiff_t)yy * srcStride[3] + 2 * i + k] << abase;
                        }
                        memcpy(out + 8 * i + 4 * k, &v, 4);
                    }
                }
            }
        }
    }
    return srcSliceH;
}

static const uint8_t dither_8x8_220_u[9][8] = {   /* ff_dither_8x8_220 (output.c:84-95), nine rows: the second row of a pair reads row + 1 */
    { 117,  62, 158, 103, 113,  58, 155, 100 }, {  34, 199,  21, 186,  31, 196,  17, 182 }, { 144,  89, 131,  76, 141,  86, 127,  72 },
    {   0, 165,  41, 206,  10, 175,  52, 217 }, { 110,  55, 151,  96, 120,  65, 162, 107 }, {  28, 193,  14, 179,  38, 203,  24, 189 },
    { 138,  83, 124,  69, 148,  93, 134,  79 }, {   7, 172,  48, 213,   3, 168,  45, 210 }, { 117,  62, 158, 103, 113,  58, 155, 100 },
};
/* yuv2rgb_c_1_ordered_dither (yuv2rgb.c:457-517): chroma is ignored (g is the table pointer of U = V = 128); each 8-pixel group of a row
 * pair becomes one byte per row, the second row with the following dither row.  The tail (dst_w & 7) counts PIXEL PAIRS across both
 * rows in the macro's order - row 1 pair 0, row 2 pair 0, row 2 pair 1, row 1 pair 1, row 1 pair 2, ... - and shifts the rest */
static int unscaled_yuv2mono(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                             int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    const int g = c->table_gU[128 + HEADROOM] + c->table_gV[128 + HEADROOM];
    static const int order[8][2] = { { 0, 0 }, { 1, 0 }, { 1, 1 }, { 0, 1 }, { 0, 2 }, { 1, 2 }, { 1, 3 }, { 0, 3 } };   /* { row, pair } */
    for (int y = 0; y < srcSliceH; y += 2) {
        const int yd = y + srcSliceY;
        const uint8_t *py[2] = { src[0] + (ptrdiff_t)y * srcStride[0], src[0] + (ptrdiff_t)(y + 1) * srcStride[0] };
        uint8_t *out[2] = { dst[0] + (ptrdiff_t)yd * dstStride[0], dst[0] + (ptrdiff_t)(yd + 1) * dstStride[0] };
        const uint8_t *d128 = dither_8x8_220_u[yd & 7];
        int x;
        for (x = 0; x + 8 <= c->o.dst_w; x += 8)
            for (int l = 0; l < 2; l++) {
                unsigned acc = 0;
                for (int k = 0; k < 8; k++) acc = acc + acc + lut_at(c, g + py[l][x + k] + d128[k + 8 * l]);
                out[l][x >> 3] = (uint8_t)acc;
            }
        if (c->o.dst_w & 7) {
            int pixels_left = c->o.dst_w & 7;
            unsigned acc[2] = { 0, 0 };
            for (int s = 0; s < 8; s++) {
                const int l = order[s][0], i = order[s][1];
                if (pixels_left) {
                    acc[l] = acc[l] + acc[l] + lut_at(c, g + py[l][x + 2 * i] + d128[2 * i + 8 * l]);
                    acc[l] = acc[l] + acc[l] + lut_at(c, g + py[l][x + 2 * i + 1] + d128[2 * i + 1 + 8 * l]);
                    pixels_left--;
                } else acc[l] <<= 2;
            }
            out[0][x >> 3] = (uint8_t)acc[0]; out[1][x >> 3] = (uint8_t)acc[1];
        }
    }
    return srcSliceH;
}

/* planarToP01xWrapper, swscale_unscaled.c:273-322 */
static int unscaled_p01x(OrSws *c, const uint8_t *const src8[], const int srcStride[], int srcSliceY,
                         int srcSliceH, uint8_t *const dst8[], const int dstStride[])
{
    const Desc *sf = desc_get(c->o.src_format), *df = desc_get(c->o.dst_format);
    int shift[3];
    for (int k = 0; k < 3; k++)
        shift[k] = df->c[k].depth + df->c[k].shift - sf->c[k].depth - sf->c[k].shift;
    for (int y = 0; y < srcSliceH; y++) {
        const uint16_t *s0 = (const uint16_t *)(src8[0] + y * srcStride[0]);
        uint16_t *dY = (uint16_t *)(dst8[0] + (srcSliceY + y) * dstStride[0]);
        for (int x = 0; x < c->o.src_w; x++) dY[x] = (uint16_t)(s0[x] << shift[0]);
        if (!(y & 1)) {
            const uint16_t *s1 = (const uint16_t *)(src8[1] + (y >> 1) * srcStride[1]);
            const uint16_t *s2 = (const uint16_t *)(src8[2] + (y >> 1) * srcStride[2]);
            uint16_t *dUV = (uint16_t *)(dst8[1] + ((srcSliceY + y) >> 1) * dstStride[1]);
            for (int x = 0; x < c->o.src_w / 2; x++) {
                dUV[2 * x] = (uint16_t)(s1[x] << shift[1]);
                dUV[2 * x + 1] = (uint16_t)(s2[x] << shift[2]);
            }
        }
    }
    return srcSliceH;
}

/* planar8ToP01xleWrapper, swscale_unscaled.c:333-372 */
static int unscaled_8_p01x(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                           int srcSliceH, uint8_t *const dst8[], const int dstStride[])
{
    for (int y = 0; y < srcSliceH; y++) {
        const uint8_t *s0 = src[0] + y * srcStride[0];
        uint16_t *dY = (uint16_t *)(dst8[0] + (srcSliceY + y) * dstStride[0]);
        for (int x = 0; x < c->o.src_w; x++) dY[x] = (uint16_t)(s0[x] << 8);
        if (!(y & 1)) {
            const uint8_t *s1 = src[1] + (y >> 1) * srcStride[1], *s2 = src[2] + (y >> 1) * srcStride[2];
            uint16_t *dUV = (uint16_t *)(dst8[1] + ((srcSliceY + y) >> 1) * dstStride[1]);
            for (int x = 0; x < c->o.src_w / 2; x++) {
                dUV[2 * x] = (uint16_t)(s1[x] << 8);
                dUV[2 * x + 1] = (uint16_t)(s2[x] << 8);
            }
        }
    }
    return srcSliceH;
}

static void copy_plane(const uint8_t *src, int srcStride, int y0, int h, int w, uint8_t *dst, int dstStride)
{
    dst += dstStride * y0;
    for (int i = 0; i < h; i++) { memcpy(dst, src, w); src += srcStride; dst += dstStride; }
}

/* planarToNv12Wrapper / nv12ToPlanarWrapper, swscale_unscaled.c:147-186 */
static int unscaled_planar2nv12(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                                int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    uint8_t *d = dst[1] + dstStride[1] * srcSliceY / 2;
    int a = c->o.dst_format == ORF_NV12 ? 1 : 2, b = 3 - a;
    copy_plane(src[0], srcStride[0], srcSliceY, srcSliceH, c->o.src_w, dst[0], dstStride[0]);
    for (int y = 0; y < (srcSliceH + 1) / 2; y++) {
        const uint8_t *s1 = src[a] + y * srcStride[a], *s2 = src[b] + y * srcStride[b];
        for (int x = 0; x < c->chrSrcW; x++) { d[2 * x] = s1[x]; d[2 * x + 1] = s2[x]; }
        d += dstStride[1];
    }
    return srcSliceH;
}
static int unscaled_nv122planar(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                                int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    int a = c->o.src_format == ORF_NV12 ? 1 : 2, b = 3 - a;
    copy_plane(src[0], srcStride[0], srcSliceY, srcSliceH, c->o.src_w, dst[0], dstStride[0]);
    for (int y = 0; y < (srcSliceH + 1) / 2; y++) {
        const uint8_t *s = src[1] + y * srcStride[1];
        uint8_t *d1 = dst[a] + dstStride[a] * (srcSliceY / 2 + y), *d2 = dst[b] + dstStride[b] * (srcSliceY / 2 + y);
        for (int x = 0; x < c->chrSrcW; x++) { d1[x] = s[2 * x]; d2[x] = s[2 * x + 1]; }
    }
    return srcSliceH;
}

/* planarToNv24Wrapper / nv24ToPlanarWrapper / nv24ToYuv420Wrapper, swscale_unscaled.c:188-271 */
static int unscaled_planar2nv24(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                                int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    int a = c->o.dst_format == ORF_NV24 ? 1 : 2, b = 3 - a;
    copy_plane(src[0], srcStride[0], srcSliceY, srcSliceH, c->o.src_w, dst[0], dstStride[0]);
    for (int y = 0; y < srcSliceH; y++) {
        const uint8_t *s1 = src[a] + y * srcStride[a], *s2 = src[b] + y * srcStride[b];
        uint8_t *d = dst[1] + dstStride[1] * (srcSliceY + y);
        for (int x = 0; x < c->chrSrcW; x++) { d[2 * x] = s1[x]; d[2 * x + 1] = s2[x]; }
    }
    return srcSliceH;
}
static int unscaled_nv242planar(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                                int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    int a = c->o.src_format == ORF_NV24 ? 1 : 2, b = 3 - a;
    copy_plane(src[0], srcStride[0], srcSliceY, srcSliceH, c->o.src_w, dst[0], dstStride[0]);
    for (int y = 0; y < srcSliceH; y++) {
        const uint8_t *s = src[1] + y * srcStride[1];
        uint8_t *d1 = dst[a] + dstStride[a] * (srcSliceY + y), *d2 = dst[b] + dstStride[b] * (srcSliceY + y);
        for (int x = 0; x < c->chrSrcW; x++) { d1[x] = s[2 * x]; d2[x] = s[2 * x + 1]; }
    }
    return srcSliceH;
}
static int unscaled_nv242yuv420(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                                int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    int a = c->o.src_format == ORF_NV24 ? 1 : 2, b = 3 - a;
    const int w = c->o.src_w / 2;
    copy_plane(src[0], srcStride[0], srcSliceY, srcSliceH, c->o.src_w, dst[0], dstStride[0]);
    for (int y = 0; y < srcSliceH; y += 2) { /* nv24_to_yuv420p_chroma :229-251: truncating 2x2 mean */
        const uint8_t *s1 = src[1] + y * srcStride[1], *s2 = y + 1 == srcSliceH ? s1 : s1 + srcStride[1];
        uint8_t *d1 = dst[a] + dstStride[a] * (srcSliceY / 2 + y / 2), *d2 = dst[b] + dstStride[b] * (srcSliceY / 2 + y / 2);
        for (int x = 0; x < w; x++) {
            d1[x] = (uint8_t)((s1[4 * x + 0] + s1[4 * x + 2] + s2[4 * x + 0] + s2[4 * x + 2]) >> 2);
            d2[x] = (uint8_t)((s1[4 * x + 1] + s1[4 * x + 3] + s2[4 * x + 1] + s2[4 * x + 3]) >> 2);
        }
    }
    return srcSliceH;
}

/* yvu9ToYv12Wrapper (swscale_unscaled.c:2079-2093) -> planar2x_c (rgb2rgb_template.c:531-574): 2x chroma up-sampling
 * with (3,1)/4 weights, diagonal neighbours on the interior lines.  Written per output sample. */
static void planar2x(const uint8_t *src, uint8_t *dst, int W, int H, int srcStride, int dstStride)
{
    for (int Y = 0; Y < 2 * H; Y++) {
        uint8_t *d = dst + (ptrdiff_t)Y * dstStride;
        if (Y == 0 || Y == 2 * H - 1) { /* first / last line: horizontal only */
            const uint8_t *s = src + (ptrdiff_t)(Y ? H - 1 : 0) * srcStride;
            d[0] = s[0];
            for (int x = 0; x < W - 1; x++) { d[2 * x + 1] = (uint8_t)((3 * s[x] + s[x + 1]) >> 2); d[2 * x + 2] = (uint8_t)((s[x] + 3 * s[x + 1]) >> 2); }
            d[2 * W - 1] = s[W - 1];
        } else {
            const int y = (Y + 1) >> 1;                        /* lines 2y-1 and 2y come from source rows y-1 (A) and y (B) */
            const uint8_t *A = src + (ptrdiff_t)(y - 1) * srcStride, *B = A + srcStride;
            if (Y & 1) {                                         /* line 2y-1: 3*A + B */
                d[0] = (uint8_t)((3 * A[0] + B[0]) >> 2);
                for (int x = 0; x < W - 1; x++) { d[2 * x + 1] = (uint8_t)((3 * A[x] + B[x + 1]) >> 2); d[2 * x + 2] = (uint8_t)((3 * A[x + 1] + B[x]) >> 2); }
                d[2 * W - 1] = (uint8_t)((3 * A[W - 1] + B[W - 1]) >> 2);
            } else {                                             /* line 2y: A + 3*B */
                d[0] = (uint8_t)((A[0] + 3 * B[0]) >> 2);
                for (int x = 0; x < W - 1; x++) { d[2 * x + 2] = (uint8_t)((A[x] + 3 * B[x + 1]) >> 2); d[2 * x + 1] = (uint8_t)((A[x + 1] + 3 * B[x]) >> 2); }
                d[2 * W - 1] = (uint8_t)((A[W - 1] + 3 * B[W - 1]) >> 2);
            }
        }
    }
}
static int unscaled_yvu9_yv12(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                              int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    copy_plane(src[0], srcStride[0], srcSliceY, srcSliceH, c->o.src_w, dst[0], dstStride[0]);
    planar2x(src[1], dst[1] + dstStride[1] * (srcSliceY >> 1), c->chrSrcW, srcSliceH >> 2, srcStride[1], dstStride[1]);
    planar2x(src[2], dst[2] + dstStride[2] * (srcSliceY >> 1), c->chrSrcW, srcSliceH >> 2, srcStride[2], dstStride[2]);
    return srcSliceH;
}

/* rgbToRgbWrapper (swscale_unscaled.c:2001-2060) with the little-endian C converters of rgb2rgb.c / rgb2rgb_template.c
 * (shuffle_bytes_*, rgb24to32, rgb32to24, rgb24tobgr24, rgb24tobgr32, rgb32tobgr24): every one of them moves the R, G
 * and B bytes of the source pixel to the R, G and B positions of the destination pixel, copies A when both have one
 * and writes 255 when only the destination has one.  The ALT32_CORR offset tricks for argb/abgr (:2021-2032) give
 * the same visible bytes (the row-start 255 plus each pixel's trailing 255 land on the A positions); the one byte
 * they spill past the end of a row is outside the picture and not restated. */
static int unscaled_rgb2rgb(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                            int srcSliceH, uint8_t *const dst[], const int dstStride[], int force_opaque)
{
    const Desc *ds = desc_get(c->o.src_format), *dd = desc_get(c->o.dst_format);
    const int sa = isALPHA(c->o.src_format) && !force_opaque, da = isALPHA(c->o.dst_format);
    (void)srcSliceY;
    for (int y = 0; y < srcSliceH; y++) {
        const uint8_t *s = src[0] + (ptrdiff_t)y * srcStride[0];
        uint8_t *d = dst[0] + (ptrdiff_t)y * dstStride[0];
        for (int x = 0; x < c->o.src_w; x++, s += ds->c[0].step, d += dd->c[0].step) {
            for (int k = 0; k < 3; k++) d[dd->c[k].offset] = s[ds->c[k].offset];
            if (da) d[dd->c[3].offset] = sa ? s[ds->c[3].offset] : 255;
            else if (dd->nb == 4 && ds->nb == 4) d[dd->c[3].offset] = s[ds->c[3].offset];   /* vuyx: the X byte is the shuffled source alpha */
        }
    }
    return srcSliceH;
}


/* rgbToRgbWrapper (swscale_unscaled.c:2000-2060) with the 12/15/16 bpp converters of rgb2rgb.c:179-320 and
 * rgb2rgb_template.c:85-316.  They are bit-field shuffles in "int" order: a 16-bit pixel has a high, a middle and a low field,
 * a 24/32 bpp pixel three bytes in memory order (32_1 formats: after the alpha byte, ALT32_CORR). */
static int unscaled_rgblow(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                           int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    const int s = c->o.src_format, d = c->o.dst_format, sid = c->srcFormatBpp, did = c->dstFormatBpp;
    const int s_rgbint = s == ORF_RGB24 || s == ORF_BGRA || s == ORF_ABGR || s == ORF_RGB565LE || s == ORF_RGB555LE || s == ORF_RGB444LE;
    const int d_rgbint = d == ORF_RGB24 || d == ORF_BGRA || d == ORF_ABGR || d == ORF_RGB565LE || d == ORF_RGB555LE || d == ORF_RGB444LE;
    const int same = s_rgbint == d_rgbint;
    const int s_alt = s == ORF_ABGR || s == ORF_ARGB, d_alt = d == ORF_ABGR || d == ORF_ARGB;   /* RGB32_1 / BGR32_1 */
    (void)srcSliceY;
    for (int y = 0; y < srcSliceH; y++) {
        const uint8_t *sp = src[0] + (ptrdiff_t)y * srcStride[0] + (s_alt ? 1 : 0);
        uint8_t *dp = dst[0] + (ptrdiff_t)y * dstStride[0];
        for (int x = 0; x < c->o.src_w; x++) {
            unsigned px = 0, b0 = 0, b1 = 0, b2 = 0;       /* 16-bit source pixel, or the three bytes of a 24/32 bpp one */
            unsigned out16 = 0, o0 = 0, o1 = 0, o2 = 0;
            if (sid <= 16) { uint16_t t; memcpy(&t, sp + 2 * x, 2); px = t; }
            else { const uint8_t *q = sp + (sid == 24 ? 3 : 4) * x; b0 = q[0]; b1 = q[1]; b2 = q[2]; }
            if (did <= 16) {
                if (sid == 12 && did == 15) {                 /* rgb12to15 */
                    unsigned r = px & 0xF00, g = px & 0x0F0, b = px & 0x00F;
                    r = (r << 3) | ((r & 0x800) >> 1); g = (g << 2) | ((g & 0x080) >> 2); b = (b << 1) | (b >> 3);
                    out16 = (r | g | b) & 0xFFFF;
                } else if (sid == 12 && did == 12) out16 = (px << 8 | (px & 0xF0) | px >> 8) & 0xFFF;                          /* rgb12tobgr12 */
                else if (sid == 15 && did == 16) out16 = same ? ((px & 0x7FFF) + (px & 0x7FE0)) & 0xFFFF                     /* rgb15to16 */
                                                              : (((px & 0x7C00) >> 10) | ((px & 0x3E0) << 1) | (px << 11)) & 0xFFFF; /* rgb15tobgr16 */
                else if (sid == 16 && did == 15) out16 = same ? (((px >> 1) & 0x7FE0) | (px & 0x001F))                      /* rgb16to15 */
                                                              : ((px >> 11) | ((px & 0x7C0) >> 1) | ((px & 0x1F) << 10)) & 0xFFFF; /* rgb16tobgr15 */
                else if (sid == 16 && did == 16) out16 = ((px >> 11) | (px & 0x7E0) | (px << 11)) & 0xFFFF;                   /* rgb16tobgr16 */
                else if (sid == 15 && did == 15) { const unsigned br = px & 0x7C1F; out16 = ((br >> 10) | (px & 0x3E0) | (br << 10)) & 0xFFFF; } /* rgb15tobgr15 */
                else if (sid == 24) {
                    /* rgb24to16/15: first byte -> high field; rgb24tobgr16/15: first byte -> low field (rgb2rgb_template.c:185-241) */
                    const unsigned hi = same ? b0 : b2, lo = same ? b2 : b0;
                    out16 = did == 16 ? ((lo >> 3) | ((b1 & 0xFC) << 3) | ((hi & 0xF8) << 8)) : ((lo >> 3) | ((b1 & 0xF8) << 2) | ((hi & 0xF8) << 7));
                } else {
                    /* rgb32to16/15: byte 0 -> low field; rgb32tobgr16/15: byte 0 -> high field (:123-183) */
                    const unsigned lo = same ? b0 : b2, hi = same ? b2 : b0;
                    out16 = did == 16 ? ((lo >> 3) + ((b1 & 0xFC) << 3) + ((hi & 0xF8) << 8)) : ((lo >> 3) + ((b1 & 0xF8) << 2) + ((hi & 0xF8) << 7));
                }
                { const uint16_t t = (uint16_t)out16; memcpy(dp + 2 * x, &t, 2); }
            } else {
                /* 15/16 -> 24/32: fields widened by bit replication */
                unsigned hi8, mid8, lo8;
                if (sid == 16) { hi8 = ((px & 0xF800) >> 8) | ((px & 0xF800) >> 13); mid8 = ((px & 0x07E0) >> 3) | ((px & 0x07E0) >> 9); lo8 = ((px & 0x001F) << 3) | ((px & 0x001F) >> 2); }
                else           { hi8 = ((px & 0x7C00) >> 7) | ((px & 0x7C00) >> 12); mid8 = ((px & 0x03E0) >> 2) | ((px & 0x03E0) >> 7); lo8 = ((px & 0x001F) << 3) | ((px & 0x001F) >> 2); }
                if (did == 24) {           /* rgb16to24 / rgb15to24: high field first; rgb16tobgr24 / rgb15tobgr24: low field first */
                    o0 = same ? hi8 : lo8; o1 = mid8; o2 = same ? lo8 : hi8;
                    dp[3 * x] = (uint8_t)o0; dp[3 * x + 1] = (uint8_t)o1; dp[3 * x + 2] = (uint8_t)o2;
                } else {                   /* rgb16to32 / rgb15to32: low field first; ...tobgr32: high field first; 255 last (first for the _1 layouts) */
                    uint8_t *q = dp + 4 * x;
                    o0 = same ? lo8 : hi8; o1 = mid8; o2 = same ? hi8 : lo8;
                    if (d_alt) { q[0] = 255; q[1] = (uint8_t)o0; q[2] = (uint8_t)o1; q[3] = (uint8_t)o2; }
                    else       { q[0] = (uint8_t)o0; q[1] = (uint8_t)o1; q[2] = (uint8_t)o2; q[3] = 255; }
                }
            }
        }
    }
    return srcSliceH;
}

/* bgr24ToYv12Wrapper (swscale_unscaled.c:2062-2078) -> ff_rgb24toyv12_c (rgb2rgb_template.c:580-641): truncating
 * Y per pixel, U/V from the truncated 2x2 mean of each channel; everything unsigned, stored modulo 256. */
static int unscaled_bgr24_yv12(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                               int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    const int32_t *t = c->rgb2yuv;
    const int cw = c->o.src_w >> 1;
    (void)srcSliceY;
    for (int y = 0; y < srcSliceH; y += 2) {
        const int y2 = y + 1 == srcSliceH ? y : y + 1;
        const uint8_t *s1 = src[0] + (ptrdiff_t)y * srcStride[0], *s2 = src[0] + (ptrdiff_t)y2 * srcStride[0];
        uint8_t *d1 = dst[0] + (ptrdiff_t)y * dstStride[0], *d2 = dst[0] + (ptrdiff_t)y2 * dstStride[0];
        uint8_t *du = dst[1] + (ptrdiff_t)(y >> 1) * dstStride[1], *dv = dst[2] + (ptrdiff_t)(y >> 1) * dstStride[2];
        for (int i = 0; i < cw; i++) {
            unsigned b[4], g[4], r[4], Y[4], bx, gx, rx;
            for (int k = 0; k < 4; k++) {
                const uint8_t *p = (k < 2 ? s1 : s2) + 6 * i + 3 * (k & 1);
                b[k] = p[0]; g[k] = p[1]; r[k] = p[2];
                Y[k] = (((unsigned)t[RY] * r[k] + (unsigned)t[GY] * g[k] + (unsigned)t[BY] * b[k]) >> 15) + 16;
            }
            bx = (b[0] + b[1] + b[2] + b[3]) >> 2; gx = (g[0] + g[1] + g[2] + g[3]) >> 2; rx = (r[0] + r[1] + r[2] + r[3]) >> 2;
            /* order of the stores as in the reference: for an odd last row d2 == d1 and the second pair wins */
            d1[2 * i] = (uint8_t)Y[0]; d1[2 * i + 1] = (uint8_t)Y[1];
            d2[2 * i] = (uint8_t)Y[2]; d2[2 * i + 1] = (uint8_t)Y[3];
            du[i] = (uint8_t)((((unsigned)t[RU] * rx + (unsigned)t[GU] * gx + (unsigned)t[BU] * bx) >> 15) + 128);
            dv[i] = (uint8_t)((((unsigned)t[RV] * rx + (unsigned)t[GV] * gx + (unsigned)t[BV] * bx) >> 15) + 128);
        }
    }
    return srcSliceH;
}

/* planarRgbToRgbWrapper (swscale_unscaled.c:1322-1378) with gbr24ptopacked24/32 (:1188-1233): interleave, A = 255 */
static int unscaled_gbrp2packed(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                                int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    const Desc *ds = desc_get(c->o.src_format), *dd = desc_get(c->o.dst_format);
    (void)srcSliceY;
    for (int y = 0; y < srcSliceH; y++) {
        uint8_t *d = dst[0] + (ptrdiff_t)y * dstStride[0];
        for (int x = 0; x < c->o.src_w; x++, d += dd->c[0].step) {
            for (int k = 0; k < 3; k++)
                d[dd->c[k].offset] = src[ds->c[k].plane][(ptrdiff_t)y * srcStride[ds->c[k].plane] + x];
            if (dd->c[0].step == 4)   /* gbraptopacked32 (:1235-1262) takes the alpha plane, gbr24ptopacked32 (:1213-1233) writes 255 */
                d[isALPHA(c->o.dst_format) ? dd->c[3].offset : 0] = c->o.src_format == ORF_GBRAP ? src[3][(ptrdiff_t)y * srcStride[3] + x] : 0xff;
        }
    }
    return srcSliceH;
}

/* yuv422pToYuy2Wrapper / yuv422pToUyvyWrapper / planarToYuy2Wrapper / planarToUyvyWrapper (swscale_unscaled.c:376-422)
 * -> yuvPlanartoyuy2_c / yuvPlanartouyvy_c (rgb2rgb_template.c:379-470): width >> 1 pixel pairs per row, the chroma row
 * advances every vertLumPerChroma (1 or 2) luma rows counted from the slice start. */
static int unscaled_planar2p422(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                                int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    const int vlpc = c->o.src_format == ORF_YUV422P ? 1 : 2;
    const Desc *dd = desc_get(c->o.dst_format);
    (void)srcSliceY;
    for (int y = 0; y < srcSliceH; y++) {
        const uint8_t *ys = src[0] + (ptrdiff_t)y * srcStride[0];
        const uint8_t *us = src[1] + (ptrdiff_t)(y / vlpc) * srcStride[1], *vs = src[2] + (ptrdiff_t)(y / vlpc) * srcStride[2];
        uint8_t *d = dst[0] + (ptrdiff_t)y * dstStride[0];
        for (int i = 0; i < (c->o.src_w >> 1); i++) {
            d[4 * i + dd->c[0].offset] = ys[2 * i]; d[4 * i + dd->c[0].offset + 2] = ys[2 * i + 1];
            d[4 * i + dd->c[1].offset] = us[i]; d[4 * i + dd->c[2].offset] = vs[i];
        }
    }
    return srcSliceH;
}

/* yuyvToYuv420/422Wrapper, uyvyToYuv420/422Wrapper (swscale_unscaled.c:424-484) -> yuyvtoyuv420_c .. uyvytoyuv422_c
 * (rgb2rgb_template.c:751-825): luma extracted, 4:2:2 chroma extracted, 4:2:0 chroma = truncating mean of the two rows of
 * each row pair counted from the slice start (an odd last row produces no chroma row). */
static int unscaled_p4222planar(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                                int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    const Desc *ds = desc_get(c->o.src_format);
    const int w = c->o.src_w, cw = CEIL_RSHIFT(w, 1), to420 = c->o.dst_format != ORF_YUV422P;
    const int yo = ds->c[0].offset, uo = ds->c[1].offset, vo = ds->c[2].offset;
    (void)srcSliceY;
    for (int y = 0; y < srcSliceH; y++) {
        const uint8_t *s = src[0] + (ptrdiff_t)y * srcStride[0];
        uint8_t *yd = dst[0] + (ptrdiff_t)y * dstStride[0];
        for (int i = 0; i < w; i++) yd[i] = s[2 * i + yo];
        if (!to420) {
            uint8_t *ud = dst[1] + (ptrdiff_t)y * dstStride[1], *vd = dst[2] + (ptrdiff_t)y * dstStride[2];
            for (int i = 0; i < cw; i++) { ud[i] = s[4 * i + uo]; vd[i] = s[4 * i + vo]; }
        } else if (y & 1) {
            const uint8_t *s0 = s - srcStride[0];
            uint8_t *ud = dst[1] + (ptrdiff_t)(y >> 1) * dstStride[1], *vd = dst[2] + (ptrdiff_t)(y >> 1) * dstStride[2];
            for (int i = 0; i < cw; i++) { ud[i] = (uint8_t)((s0[4 * i + uo] + s[4 * i + uo]) >> 1); vd[i] = (uint8_t)((s0[4 * i + vo] + s[4 * i + vo]) >> 1); }
        }
    }
    return srcSliceH;
}

/* rgb48tobgr48 / rgb48to64 / rgb48tobgr64 / rgb64to48 / rgb64tobgr48 (rgb2rgb.c:322-413): 16-bit channel moves, A = 0xFFFF */
static int unscaled_rgb16shuffle(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                                 int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    const Desc *ds = desc_get(c->o.src_format), *dd = desc_get(c->o.dst_format);
    (void)srcSliceY;
    for (int y = 0; y < srcSliceH; y++) {
        const uint16_t *s = (const uint16_t *)(src[0] + (ptrdiff_t)y * srcStride[0]);
        uint16_t *d = (uint16_t *)(dst[0] + (ptrdiff_t)y * dstStride[0]);
        for (int x = 0; x < c->o.src_w; x++, s += ds->c[0].step / 2, d += dd->c[0].step / 2) {
            for (int k = 0; k < 3; k++) d[dd->c[k].offset / 2] = s[ds->c[k].offset / 2];
            if (dd->nb == 4) d[3] = 0xFFFF;
        }
    }
    return srcSliceH;
}
/* Rgb16ToPlanarRgb16Wrapper + packed16togbra16 (swscale_unscaled.c:685-889, :891-962): sample >> (16 - depth) */
static int unscaled_packed16_gbrp16(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                                    int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    const Desc *ds = desc_get(c->o.src_format), *dd = desc_get(c->o.dst_format);
    const int shift = 16 - dd->c[0].depth;
    (void)srcSliceY;
    for (int y = 0; y < srcSliceH; y++) {
        const uint16_t *s = (const uint16_t *)(src[0] + (ptrdiff_t)y * srcStride[0]);
        for (int x = 0; x < c->o.src_w; x++, s += ds->c[0].step / 2) {
            for (int k = 0; k < 3; k++)
                ((uint16_t *)(dst[dd->c[k].plane] + (ptrdiff_t)y * dstStride[dd->c[k].plane]))[x] = (uint16_t)(s[ds->c[k].offset / 2] >> shift);
            if (isALPHA(c->o.dst_format) && dst[3])   /* packed16togbra16 (:685-817): the source's alpha word, or 0xFFFF, >> shift */
                ((uint16_t *)(dst[3] + (ptrdiff_t)y * dstStride[3]))[x] = (uint16_t)((isALPHA(c->o.src_format) ? s[3] : 0xFFFF) >> shift);
        }
    }
    return srcSliceH;
}
/* planarRgb16ToRgb16Wrapper + gbr16ptopacked16 (swscale_unscaled.c:964-1120, :1122-1186): c << (16-bpp) | c >> ((bpp-8)*2) */
static int unscaled_gbrp16_packed16(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                                    int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    const Desc *ds = desc_get(c->o.src_format), *dd = desc_get(c->o.dst_format);
    const int bpp = ds->c[0].depth, hi = 16 - bpp, lo = (bpp - 8) * 2;
    (void)srcSliceY;
    for (int y = 0; y < srcSliceH; y++) {
        uint16_t *d = (uint16_t *)(dst[0] + (ptrdiff_t)y * dstStride[0]);
        for (int x = 0; x < c->o.src_w; x++, d += dd->c[0].step / 2) {
            for (int k = 0; k < 3; k++) {
                const uint16_t v = ((const uint16_t *)(src[ds->c[k].plane] + (ptrdiff_t)y * srcStride[ds->c[k].plane]))[x];
                d[dd->c[k].offset / 2] = (uint16_t)(v << hi | v >> lo);
            }
            if (dd->nb == 4) {   /* gbr16ptopacked16 (:964-1081): the alpha plane scaled like the colours, or 0xffff */
                if (isALPHA(c->o.src_format) && src[3]) { const uint16_t v = ((const uint16_t *)(src[3] + (ptrdiff_t)y * srcStride[3]))[x]; d[3] = (uint16_t)(v << hi | v >> lo); }
                else d[3] = 0xffff;
            }
        }
    }
    return srcSliceH;
}

/* the 10-bit fields of an x2rgb10le / x2bgr10le pixel in R, G, B order */
static void rgb30_fields(int f, uint32_t p, unsigned v[3])
{
    v[1] = (p >> 10) & 0x3FF;
    if (f == ORF_X2RGB10LE) { v[0] = (p >> 20) & 0x3FF; v[2] = p & 0x3FF; } else { v[0] = p & 0x3FF; v[2] = (p >> 20) & 0x3FF; }
}
/* x2rgb10to48 / x2rgb10to64 / x2rgb10tobgr48 / x2rgb10tobgr64 (rgb2rgb.c:415-471): component << 6 | component >> 4, A = 0xffff */
static int unscaled_rgb30_to_16(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                                int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    const Desc *dd = desc_get(c->o.dst_format);
    (void)srcSliceY;
    for (int y = 0; y < srcSliceH; y++) {
        const uint8_t *s = src[0] + (ptrdiff_t)y * srcStride[0];
        uint16_t *d = (uint16_t *)(dst[0] + (ptrdiff_t)y * dstStride[0]);
        for (int x = 0; x < c->o.src_w; x++, d += dd->c[0].step / 2) {
            uint32_t p; unsigned v[3];
            memcpy(&p, s + 4 * x, 4);
            rgb30_fields(c->o.src_format, p, v);
            for (int k = 0; k < 3; k++) d[dd->c[k].offset / 2] = (uint16_t)(v[k] << 6 | v[k] >> 4);
            if (dd->nb == 4) d[3] = 0xffff;
        }
    }
    return srcSliceH;
}
/* Rgb16ToPlanarRgb16Wrapper + packed30togbra10 (swscale_unscaled.c:819-889, :933-953): (c << (bpc-10) | c >> (20-bpc)) << shift */
static int unscaled_rgb30_to_gbrp(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                                  int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    const Desc *dd = desc_get(c->o.dst_format);
    const int bpc = dd->c[0].depth, shift = dd->c[0].shift, hi = bpc - 10, lo = 10 - hi;
    (void)srcSliceY;
    for (int y = 0; y < srcSliceH; y++) {
        const uint8_t *s = src[0] + (ptrdiff_t)y * srcStride[0];
        for (int x = 0; x < c->o.src_w; x++) {
            uint32_t p; unsigned v[3];
            memcpy(&p, s + 4 * x, 4);
            rgb30_fields(c->o.src_format, p, v);
            for (int k = 0; k < 3; k++)
                ((uint16_t *)(dst[dd->c[k].plane] + (ptrdiff_t)y * dstStride[dd->c[k].plane]))[x] = (uint16_t)((v[k] << hi | v[k] >> lo) << shift);
            if (isALPHA(c->o.dst_format) && dst[3])   /* packed30togbra10 (:819-889): alpha_val = (1 << bpc) - 1 */
                ((uint16_t *)(dst[3] + (ptrdiff_t)y * dstStride[3]))[x] = (uint16_t)(((1u << bpc) - 1) << shift);
        }
    }
    return srcSliceH;
}
/* planarRgb16ToRgb16Wrapper + gbr16ptopacked30 (swscale_unscaled.c:1076-1103, :1169-1178): sample >> (depth + shift - 10), fields added */
static int unscaled_gbrp_to_rgb30(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                                  int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    const Desc *ds = desc_get(c->o.src_format);
    const int shift = ds->c[0].depth + ds->c[0].shift - 10, x2rgb = c->o.dst_format == ORF_X2RGB10LE;
    (void)srcSliceY;
    for (int y = 0; y < srcSliceH; y++) {
        uint8_t *d = dst[0] + (ptrdiff_t)y * dstStride[0];
        for (int x = 0; x < c->o.src_w; x++) {
            unsigned v[3]; uint32_t p;
            for (int k = 0; k < 3; k++) v[k] = (unsigned)((const uint16_t *)(src[ds->c[k].plane] + (ptrdiff_t)y * srcStride[ds->c[k].plane]))[x] >> shift;
            p = x2rgb ? (3U << 30) + (v[0] << 20) + (v[1] << 10) + v[2] : (3U << 30) + (v[2] << 20) + (v[1] << 10) + v[0];
            memcpy(d + 4 * x, &p, 4);
        }
    }
    return srcSliceH;
}

/* rgbToPlanarRgbWrapper (swscale_unscaled.c:1436-1490) with packedtogbr24p (:1404-1434): de-interleave, alpha dropped */
static int unscaled_packed2gbrp(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                                int srcSliceH, uint8_t *const dst[], const int dstStride[], int force_opaque)
{
    const Desc *ds = desc_get(c->o.src_format), *dd = desc_get(c->o.dst_format);
    /* rgbToPlanarRgbaWrapper: packed24togbrap writes 255 (:1480-1505), packed32togbrap copies the fourth byte (:1507-1541), which the
     * caller made 255 for an rgb0-style source (swscale.c:1106-1124) */
    const int want_a = c->o.dst_format == ORF_GBRAP && dst[3], have_a = ds->c[0].step == 4 && !force_opaque;
    const int aoff = ds->c[0].step == 4 ? (ds->c[0].offset == 1 || ds->c[2].offset == 1 ? 0 : 3) : 0;
    (void)srcSliceY;
    for (int y = 0; y < srcSliceH; y++) {
        const uint8_t *s = src[0] + (ptrdiff_t)y * srcStride[0];
        for (int x = 0; x < c->o.src_w; x++, s += ds->c[0].step) {
            for (int k = 0; k < 3; k++)
                dst[dd->c[k].plane][(ptrdiff_t)y * dstStride[dd->c[k].plane] + x] = s[ds->c[k].offset];
            if (want_a) dst[3][(ptrdiff_t)y * dstStride[3] + x] = have_a ? s[aoff] : 0xff;
        }
    }
    return srcSliceH;
}

/* packedCopyWrapper (swscale_unscaled.c:2138-2157); only the visible bytes of each row are restated.
 * rgb0-style sources going to a real-alpha destination were made opaque by the caller (swscale.c:1106-1124). */
static int unscaled_packedcopy(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                               int srcSliceH, uint8_t *const dst[], const int dstStride[], int force_opaque)
{
    const Desc *ds = desc_get(c->o.src_format);
    const int step = ds->c[0].step;
    /* the reference copies as many multiples of src_w bytes as fit into both strides (:2138-2157), i.e. the whole visible row:
     * for the packed 4:2:2 layouts that is a whole number of pixel pairs */
    const size_t row_bytes = isMono(c->o.src_format) ? (size_t)((c->o.src_w + 7) >> 3) :
                             ds->lw ? (size_t)((c->o.src_w + 1) >> 1) * 2 * step : (size_t)c->o.src_w * step;
    (void)srcSliceY;
    for (int y = 0; y < srcSliceH; y++) {
        const uint8_t *s = src[0] + (ptrdiff_t)y * srcStride[0];
        uint8_t *d = dst[0] + (ptrdiff_t)y * dstStride[0];
        memcpy(d, s, row_bytes);
        if (force_opaque)
            for (int x = 0; x < c->o.src_w; x++) d[x * step + ds->c[3].offset] = 255;
    }
    return srcSliceH;
}

static const uint8_t dithers[8][8][8] = { /* swscale_unscaled.c:39-112 */
{ {0,1,0,1,0,1,0,1},{1,0,1,0,1,0,1,0},{0,1,0,1,0,1,0,1},{1,0,1,0,1,0,1,0},{0,1,0,1,0,1,0,1},{1,0,1,0,1,0,1,0},{0,1,0,1,0,1,0,1},{1,0,1,0,1,0,1,0} },
{ {1,2,1,2,1,2,1,2},{3,0,3,0,3,0,3,0},{1,2,1,2,1,2,1,2},{3,0,3,0,3,0,3,0},{1,2,1,2,1,2,1,2},{3,0,3,0,3,0,3,0},{1,2,1,2,1,2,1,2},{3,0,3,0,3,0,3,0} },
{ {2,4,3,5,2,4,3,5},{6,0,7,1,6,0,7,1},{3,5,2,4,3,5,2,4},{7,1,6,0,7,1,6,0},{2,4,3,5,2,4,3,5},{6,0,7,1,6,0,7,1},{3,5,2,4,3,5,2,4},{7,1,6,0,7,1,6,0} },
{ {4,8,7,11,4,8,7,11},{12,0,15,3,12,0,15,3},{6,10,5,9,6,10,5,9},{14,2,13,1,14,2,13,1},{4,8,7,11,4,8,7,11},{12,0,15,3,12,0,15,3},{6,10,5,9,6,10,5,9},{14,2,13,1,14,2,13,1} },
{ {9,17,15,23,8,16,14,22},{25,1,31,7,24,0,30,6},{13,21,11,19,12,20,10,18},{29,5,27,3,28,4,26,2},{8,16,14,22,9,17,15,23},{24,0,30,6,25,1,31,7},{12,20,10,18,13,21,11,19},{28,4,26,2,29,5,27,3} },
{ {18,34,30,46,17,33,29,45},{50,2,62,14,49,1,61,13},{26,42,22,38,25,41,21,37},{58,10,54,6,57,9,53,5},{16,32,28,44,19,35,31,47},{48,0,60,12,51,3,63,15},{24,40,20,36,27,43,23,39},{56,8,52,4,59,11,55,7} },
{ {18,34,30,46,17,33,29,45},{50,2,62,14,49,1,61,13},{26,42,22,38,25,41,21,37},{58,10,54,6,57,9,53,5},{16,32,28,44,19,35,31,47},{48,0,60,12,51,3,63,15},{24,40,20,36,27,43,23,39},{56,8,52,4,59,11,55,7} },
{ {36,68,60,92,34,66,58,90},{100,4,124,28,98,2,122,26},{52,84,44,76,50,82,42,74},{116,20,108,12,114,18,106,10},{32,64,56,88,38,70,62,94},{96,0,120,24,102,6,126,30},{48,80,40,72,54,86,46,78},{112,16,104,8,118,22,110,14} },
};

/* planarCopyWrapper, swscale_unscaled.c:2220-2384 (little-endian formats only) */
static int unscaled_planarcopy(OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY,
                               int srcSliceH, uint8_t *const dst[], const int dstStride[])
{
    const Desc *ds = desc_get(c->o.src_format), *dd = desc_get(c->o.dst_format);
    const int sf = c->o.src_format, df = c->o.dst_format;
    int nplanes = isGray(df) ? 1 : isSemiPlanarYUV(df) ? 2 : 3;
    const int with_alpha = isALPHA(df) && (isPlanarYUV(df) || isPlanarRGB(df));
    for (int plane = 0; plane < 4; plane++) {
        if (plane >= nplanes && !(plane == 3 && with_alpha)) continue;
        if (plane == 3 && !isALPHA(sf)) { /* plane 3 the source cannot feed (:2239-2247): fillPlane 255 / fillPlane16 all ones */
            for (int i = 0; i < srcSliceH; i++) {
                uint8_t *row = dst[3] + (ptrdiff_t)(srcSliceY + i) * dstStride[3];
                if (is16BPS(df) || isNBPS(df)) { uint16_t *r16 = (uint16_t *)row; for (int j = 0; j < c->o.src_w; j++) r16[j] = (uint16_t)(0xFFFF >> (16 - dd->c[3].depth)); }
                else memset(row, 255, c->o.src_w);
            }
            continue;
        }
        if (plane > 0 && plane < 3 && isGray(sf)) { /* gray source: chroma planes are filled with mid-grey (fillPlane / fillPlane16 :2239-2247) */
            int flen = CEIL_RSHIFT(c->o.src_w, c->chrDstHSub) * (isSemiPlanarYUV(df) ? 2 : 1);
            int fy = CEIL_RSHIFT(srcSliceY, c->chrDstVSub), fh = CEIL_RSHIFT(srcSliceH, c->chrDstVSub);
            for (int i = 0; i < fh; i++) {
                uint8_t *row = dst[plane] + (ptrdiff_t)(fy + i) * dstStride[plane];
                if (is16BPS(df) || isNBPS(df)) { uint16_t *r16 = (uint16_t *)row; for (int j = 0; j < flen; j++) r16[j] = (uint16_t)(1 << (dd->c[plane].depth - 1)); }
                else memset(row, 128, flen);
            }
            continue;
        }
        int length = (plane == 0 || plane == 3) ? c->o.src_w : CEIL_RSHIFT(c->o.src_w, c->chrDstHSub);
        int y = (plane == 0 || plane == 3) ? srcSliceY : CEIL_RSHIFT(srcSliceY, c->chrDstVSub);
        int height = (plane == 0 || plane == 3) ? srcSliceH : CEIL_RSHIFT(srcSliceH, c->chrDstVSub);
        const uint8_t *srcPtr = src[plane];
        uint8_t *dstPtr = dst[plane] + dstStride[plane] * y;
        int shiftonly = plane == 1 || plane == 2 || (!c->o.src_range && plane == 0);
        int i, j;
        if (plane == 1 && isSemiPlanarYUV(df)) length *= 2;
        if (isNBPS(sf) || isNBPS(df) || (is16BPS(sf) != is16BPS(df))) {
            const int src_depth = ds->c[plane].depth, dst_depth = dd->c[plane].depth;
            const int src_shift = ds->c[plane].shift, dst_shift = dd->c[plane].shift;
            const uint16_t *srcPtr2 = (const uint16_t *)srcPtr;
            uint16_t *dstPtr2 = (uint16_t *)dstPtr;
            if (dst_depth == 8 || src_depth > dst_depth) { /* DITHER_COPY :2159-2218 */
                unsigned shift = src_depth - dst_depth, tmp, bias = 1u << (shift - 1);
                int to8 = dst_depth == 8;
                int sstride = srcStride[plane] / 2, dstride = to8 ? dstStride[plane] : dstStride[plane] / 2;
                uint8_t *d8 = dstPtr; uint16_t *d16 = dstPtr2;
                /* the macro's vector body (j < length-7) applies src_shift/dst_shift, its scalar
                 * tail does not; restated exactly */
                for (i = 0; i < height; i++) {
                    const uint8_t *dither = dithers[shift - 1][i & 7];
                    for (j = 0; j < length; j++) {
                        int body = j < ((length - 7 > 0) ? ((length - 7 + 7) / 8) * 8 : 0);
                        unsigned v;
                        if (c->o.dither == 0) {
                            if (body) { tmp = ((srcPtr2[j] >> src_shift) + bias) >> shift; v = (tmp - (tmp >> dst_depth)) << dst_shift; }
                            else { tmp = (srcPtr2[j] + bias) >> shift; v = (tmp - (tmp >> dst_depth)) << dst_shift; /* '-' binds tighter than '<<' */ }
                        } else if (shiftonly) {
                            if (body) { tmp = ((srcPtr2[j] >> src_shift) + dither[j & 7]) >> shift; v = (tmp - (tmp >> dst_depth)) << dst_shift; }
                            else { tmp = (srcPtr2[j] + dither[j & 7]) >> shift; v = (tmp - (tmp >> dst_depth)) << dst_shift; /* '-' binds tighter than '<<' */ }
                        } else {
                            if (body) { tmp = srcPtr2[j] >> src_shift; v = ((tmp - (tmp >> dst_depth) + dither[j & 7]) >> shift) << dst_shift; }
                            else { tmp = srcPtr2[j]; v = (tmp - (tmp >> dst_depth) + dither[j & 7]) >> shift; }
                        }
                        if (to8) d8[j] = (uint8_t)v; else d16[j] = (uint16_t)v;
                    }
                    if (to8) d8 += dstride; else d16 += dstride;
                    srcPtr2 += sstride;
                }
            } else if (src_depth == 8) { /* :2266-2284 */
                for (i = 0; i < height; i++) {
                    for (j = 0; j < length; j++)
                        dstPtr2[j] = shiftonly ? (uint16_t)((srcPtr[j] << (dst_depth - 8)) << dst_shift)
                                               : (uint16_t)(((srcPtr[j] << (dst_depth - 8)) | (srcPtr[j] >> (2 * 8 - dst_depth))) << dst_shift);
                    dstPtr2 += dstStride[plane] / 2; srcPtr += srcStride[plane];
                }
            } else { /* src_depth <= dst_depth :2285-2331 */
                unsigned shift = dst_depth - src_depth;
                for (i = 0; i < height; i++) {
                    for (j = 0; j < length; j++) {
                        unsigned v = srcPtr2[j] >> src_shift;
                        dstPtr2[j] = shiftonly ? (uint16_t)((v << shift) << dst_shift)
                                               : (uint16_t)(((v << shift) | (v >> (2 * src_depth - dst_depth))) << dst_shift);
                    }
                    dstPtr2 += dstStride[plane] / 2; srcPtr2 += srcStride[plane] / 2;
                }
            }
        } else {
            if (is16BPS(sf) && is16BPS(df)) length *= 2;
            else if (isFloat(sf) && isFloat(df)) length *= 4;
            for (i = 0; i < height; i++) { memcpy(dstPtr, srcPtr, length); srcPtr += srcStride[plane]; dstPtr += dstStride[plane]; }
        }
    }
    return srcSliceH;
}

/* ------------------------------------------------------------------ */
/* main path: readers                                                  */
/* ------------------------------------------------------------------ */
static int f2u16(float x) /* lrintf(av_clipf(65535.0f * x, 0, 65535)), input.c:1300 */
{
    float v = 65535.0f * x;
    v = v > 0.0f ? v : 0.0f; /* av_clipf_c: FFMIN(FFMAX(a, amin), amax) with FFMAX(a, b) = a > b ? a : b: a NaN becomes amin, like maxss does */
    v = v > 65535.0f ? 65535.0f : v;
    return (int)lrintf(v);
}

/* half2float (libavutil/half2float.h:43-60, half2float.c:22-66): the exact binary16 -> binary32 widening, NaNs quieted */
static float half_to_float(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000) << 16, e = (h >> 10) & 31, m = h & 0x3ff;
    uint32_t bits;
    float f;
    if (e == 0) {
        if (!m) bits = sign;
        else { /* subnormal: normalise */
            uint32_t mm = m << 13; int ee = 0;
            while (!(mm & 0x00800000)) { ee--; mm <<= 1; }
            bits = sign | ((uint32_t)(ee + 113) << 23) | (mm & 0x007fffff);
        }
    } else if (e == 31) bits = sign | 0x7f800000u | (m ? (m << 13) | 0x400000u : 0);
    else bits = sign | ((e + 112) << 23) | (m << 13);
    memcpy(&f, &bits, 4);
    return f;
}
/* one float / half-float element as the readers see it: lrintf(av_clipf(65535.0f * x, 0.0f, 65535.0f)) (input.c:1289-1431, :1558-1740) */
static int rdf16(const uint8_t *p, int half)
{
    if (half) { uint16_t h; memcpy(&h, p, 2); return f2u16(half_to_float(h)); }
    else { float v; memcpy(&v, p, 4); return f2u16(v); }
}
static int isPackedFloatRGB(int f) { return f == ORF_RGBF32LE || f == ORF_RGBF16LE || f == ORF_RGBAF16LE; }
static int isFloatGrayX(int f) { return f == ORF_GRAYF16LE || f == ORF_YAF32LE || f == ORF_YAF16LE; }   /* (grayf32 has its own case) */

/* ff_update_palette (swscale.c:873-951), run by scale_internal before every conversion of a usePal() source (:1088-1089): pal8 reads
 * data[1] (256 native-endian 0xAARRGGBB words), the 8 / 4 bpp RGB formats expand their bit fields */
static void update_palette(OrSws *c, const uint8_t *pal)
{
    const int32_t *t = c->rgb2yuv;
    const int f = c->o.src_format;
    for (int i = 0; i < 256; i++) {
        int r, g, b, y, u, v, a = 0xff;
        if (f == ORF_PAL8) { uint32_t p; memcpy(&p, pal + 4 * i, 4); a = (p >> 24) & 0xFF; r = (p >> 16) & 0xFF; g = (p >> 8) & 0xFF; b = p & 0xFF; }
        else if (f == ORF_RGB8) { r = (i >> 5) * 36; g = ((i >> 2) & 7) * 36; b = (i & 3) * 85; }
        else if (f == ORF_BGR8) { b = (i >> 6) * 85; g = ((i >> 3) & 7) * 36; r = (i & 7) * 36; }
        else if (f == ORF_RGB4_BYTE) { r = (i >> 3) * 255; g = ((i >> 1) & 3) * 85; b = (i & 1) * 255; }
        else if (f == ORF_GRAY8) { r = g = b = i; }
        else { b = (i >> 3) * 255; g = ((i >> 1) & 3) * 85; r = (i & 1) * 255; }   /* bgr4_byte */
        y = clip_u8((t[RY] * r + t[GY] * g + t[BY] * b + (33 << (15 - 1))) >> 15);
        u = clip_u8((t[RU] * r + t[GU] * g + t[BU] * b + (257 << (15 - 1))) >> 15);
        v = clip_u8((t[RV] * r + t[GV] * g + t[BV] * b + (257 << (15 - 1))) >> 15);
        c->pal_yuv[i] = (uint32_t)y + ((uint32_t)u << 8) + ((uint32_t)v << 16) + ((uint32_t)a << 24);
        /* the word whose memory bytes are the destination's (little-endian rows of swscale.c:916-949); plain sums, so an index beyond a
         * 4-bit format's 16 values spills a channel into its neighbours exactly like there */
        switch (c->o.dst_format) {
        case ORF_RGBA: case ORF_RGB24: c->pal_rgb[i] = (uint32_t)(r + (g << 8) + (b << 16)) + ((uint32_t)a << 24); break;   /* BGR32, RGB24 */
        case ORF_ARGB: c->pal_rgb[i] = (uint32_t)(a + (r << 8) + (g << 16)) + ((uint32_t)b << 24); break;                  /* BGR32_1 */
        case ORF_ABGR: c->pal_rgb[i] = (uint32_t)(a + (b << 8) + (g << 16)) + ((uint32_t)r << 24); break;                  /* RGB32_1 */
        case ORF_GBRP: case ORF_GBRAP: c->pal_rgb[i] = (uint32_t)(g + (b << 8) + (r << 16)) + ((uint32_t)a << 24); break;
        default: c->pal_rgb[i] = (uint32_t)(b + (g << 8) + (r << 16)) + ((uint32_t)a << 24); break;                        /* RGB32, BGR24, everything else */
        }
    }
}

/* palToRgbWrapper with sws_convertPalette8ToPacked32 / 24 (swscale_unscaled.c:600-644, :2707-2730) and palToGbrpWrapper with pal8ToPlanar8
 * (:531-545, :646-683): the bytes of the pal_rgb word are the destination's bytes */
static int unscaled_pal2rgb(const OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY, int srcSliceH,
                            uint8_t *const dst[], const int dstStride[])
{
    const int d = c->o.dst_format, w = c->o.src_w;
    const Desc *dd = desc_get(d);
    for (int y = 0; y < srcSliceH; y++) {
        const uint8_t *s = src[0] + (ptrdiff_t)y * srcStride[0];
        for (int x = 0; x < w; x++) {
            const uint32_t p = c->pal_rgb[s[x]];
            if (d == ORF_GBRP || d == ORF_GBRAP) {
                for (int k = 0; k < (d == ORF_GBRAP && dst[3] ? 4 : 3); k++) dst[k][(ptrdiff_t)(y + srcSliceY) * dstStride[k] + x] = (uint8_t)(p >> (8 * k));
            } else {
                uint8_t *o = dst[0] + (ptrdiff_t)(y + srcSliceY) * dstStride[0] + dd->c[0].step * x;
                for (int k = 0; k < dd->c[0].step; k++) o[k] = (uint8_t)(p >> (8 * k));
            }
        }
    }
    return srcSliceH;
}

/* bayer_to_rgb24_wrapper / bayer_to_rgb48_wrapper / bayer_to_yv12_wrapper (swscale_unscaled.c:1652-1806) over bayer_template.c.
 * A picture is a grid of 2x2 blocks.  The first and the last block row, and the first and last block of every other row, are "copied"
 * (nearest samples of the block itself), the rest is "interpolated" from the 4x4 neighbourhood.  An odd height ends with a copy that runs
 * upwards from the last row (negative strides) and so rewrites the row above it.  The R() / B() names of the template are byte positions:
 * BAYER_R = 0 for bggr / gbrg and 2 for rggb / grbg, so that the bytes always come out as R, G, B.  16-bit samples are reduced by
 * BAYER_SHIFT = 8 for the 8-bit destinations; the rgb48 destination takes the samples as they are (an 8-bit mosaic too).
 * A column beyond the picture (odd widths) is read from the row's padding like the reference does, where the stride holds it. */
typedef struct { const uint8_t *p; ptrdiff_t stride; int sz, w, avail; } BayerSrc;
static unsigned bayer_s(const BayerSrc *b, int y, int x)   /* S(y, x) relative to the block, x counted from the row start */
{
    const uint8_t *q = b->p + y * b->stride + (ptrdiff_t)b->sz * x;
    if ((x + 1) * b->sz > b->avail) return 0;
    if (b->sz == 1) return q[0];
    return (unsigned)q[0] | ((unsigned)q[1] << 8);
}
/* one block at column x0: v[py][px][0..2] = the template's R, G, B names */
static void bayer_block(const BayerSrc *b, int x0, int quad, int interp, int sh, unsigned v[2][2][3])
{
#define T(y, x) bayer_s(b, (y), x0 + (x))
    if (quad) {   /* BAYER_BGGR / BAYER_RGGB */
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
    } else {      /* BAYER_GBRG / BAYER_GRBG */
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
#undef T
}
static int unscaled_bayer(const OrSws *c, const uint8_t *const src[], const int srcStride[], int srcSliceY, int srcSliceH,
                          uint8_t *const dst[], const int dstStride[])
{
    const int sf = c->o.src_format, df = c->o.dst_format, W = c->o.src_w, H = srcSliceH;
    const int quad = sf == ORF_BAYER_BGGR8 || sf == ORF_BAYER_BGGR16LE || sf == ORF_BAYER_RGGB8 || sf == ORF_BAYER_RGGB16LE;
    const int rpos = (sf == ORF_BAYER_BGGR8 || sf == ORF_BAYER_BGGR16LE || sf == ORF_BAYER_GBRG8 || sf == ORF_BAYER_GBRG16LE) ? 0 : 2;
    const int sz = desc_get(sf)->c[0].step;
    const int sh = (sz == 2 && df != ORF_RGB48LE) ? 8 : 0;
    const int32_t *t = c->rgb2yuv;
    const int avail = srcStride[0] < 0 ? -srcStride[0] : srcStride[0];
    if (H < 2) return -22;
    /* block rows in the reference's order: (row0, direction, interpolated) */
    for (int pass = 0, i = 0; ; pass++) {
        int row0, dir = 1, interp_row;
        if (pass == 0) { row0 = 0; interp_row = 0; i = 2; }
        else if (i < H - 2) { row0 = i; interp_row = 1; i += 2; }
        else if (i + 1 == H) { row0 = i; dir = -1; interp_row = 0; i = H + 2; }
        else if (i < H) { row0 = i; interp_row = 0; i = H + 2; }
        else break;
        {
            BayerSrc b = { src[0] + (ptrdiff_t)row0 * srcStride[0], (ptrdiff_t)dir * srcStride[0], sz, W, avail };
            for (int x0 = 0; x0 < W; x0 += 2) {
                const int interp = interp_row && x0 >= 2 && x0 < W - 2;
                unsigned v[2][2][3];
                bayer_block(&b, x0, quad, interp, sh, v);
                if (df == ORF_YUV420P) {   /* rgb24toyv12_2x2: ff_rgb24toyv12(block, dstY, dstV, dstU, ...) reads byte 0 as B and swaps the chroma pointers */
                    unsigned bb[4], gg[4], rr[4], bx, gx, rx;
                    for (int k = 0; k < 4; k++) {
                        const unsigned *px = v[k >> 1][k & 1];
                        unsigned byte[3];
                        byte[rpos] = px[0] & 0xff; byte[1] = px[1] & 0xff; byte[2 - rpos] = px[2] & 0xff;
                        bb[k] = byte[0]; gg[k] = byte[1]; rr[k] = byte[2];
                        if (x0 + (k & 1) < W)
                            dst[0][(ptrdiff_t)(srcSliceY + row0 + dir * (k >> 1)) * dstStride[0] + x0 + (k & 1)] =
                                (uint8_t)((((unsigned)t[RY] * rr[k] + (unsigned)t[GY] * gg[k] + (unsigned)t[BY] * bb[k]) >> 15) + 16);
                    }
                    bx = (bb[0] + bb[1] + bb[2] + bb[3]) >> 2; gx = (gg[0] + gg[1] + gg[2] + gg[3]) >> 2; rx = (rr[0] + rr[1] + rr[2] + rr[3]) >> 2;
                    dst[2][(ptrdiff_t)((srcSliceY + row0) >> 1) * dstStride[2] + (x0 >> 1)] = (uint8_t)((((unsigned)t[RU] * rx + (unsigned)t[GU] * gx + (unsigned)t[BU] * bx) >> 15) + 128);
                    dst[1][(ptrdiff_t)((srcSliceY + row0) >> 1) * dstStride[1] + (x0 >> 1)] = (uint8_t)((((unsigned)t[RV] * rx + (unsigned)t[GV] * gx + (unsigned)t[BV] * bx) >> 15) + 128);
                } else {
                    for (int py = 0; py < 2; py++) for (int px = 0; px < 2 && x0 + px < W; px++) {
                        uint8_t *o = dst[0] + (ptrdiff_t)(srcSliceY + row0 + dir * py) * dstStride[0];
                        if (df == ORF_RGB48LE) {
                            uint16_t *o16 = (uint16_t *)o + 3 * (x0 + px);
                            o16[rpos] = (uint16_t)v[py][px][0]; o16[1] = (uint16_t)v[py][px][1]; o16[2 - rpos] = (uint16_t)v[py][px][2];
                        } else {
                            o += 3 * (x0 + px);
                            o[rpos] = (uint8_t)v[py][px][0]; o[1] = (uint8_t)v[py][px][1]; o[2 - rpos] = (uint8_t)v[py][px][2];
                        }
                    }
                }
            }
        }
    }
    return srcSliceH;
}

/* Produce the 8/16-bit "formatConv" luma line for source row y (NULL if the plane is read directly).
 * returns pointer to the line to feed to hscale. */
/* RGB16_32FUNCS rows for the 16 bits-per-pixel formats (input.c:396-401): masks on the unshifted pixel, coefficient shifts, S */
static void rgb16_params(int f, int *maskr, int *maskg, int *maskb, int *rsh, int *gsh, int *bsh, int *S)
{
    switch (f) {
    case ORF_BGR565LE: *maskr = 0x001F; *maskg = 0x07E0; *maskb = 0xF800; *rsh = 11; *gsh = 5; *bsh = 0; *S = 15 + 8; break;
    case ORF_BGR555LE: *maskr = 0x001F; *maskg = 0x03E0; *maskb = 0x7C00; *rsh = 10; *gsh = 5; *bsh = 0; *S = 15 + 7; break;
    case ORF_BGR444LE: *maskr = 0x000F; *maskg = 0x00F0; *maskb = 0x0F00; *rsh = 8; *gsh = 4; *bsh = 0; *S = 15 + 4; break;
    case ORF_RGB565LE: *maskr = 0xF800; *maskg = 0x07E0; *maskb = 0x001F; *rsh = 0; *gsh = 5; *bsh = 11; *S = 15 + 8; break;
    case ORF_RGB555LE: *maskr = 0x7C00; *maskg = 0x03E0; *maskb = 0x001F; *rsh = 0; *gsh = 5; *bsh = 10; *S = 15 + 7; break;
    default:           *maskr = 0x0F00; *maskg = 0x00F0; *maskb = 0x000F; *rsh = 0; *gsh = 4; *bsh = 8; *S = 15 + 4; break;   /* rgb444le */
    }
}

static const uint8_t *read_lum_line(const OrSws *c, const uint8_t *const src[], const int stride[], int y, uint8_t *tmp)
{
    const int f = c->o.src_format, w = c->o.src_w;
    const int32_t *t = c->rgb2yuv;
    /* planar RGB: planes 1 and 2 of a slice are indexed by chroma row, also when a luma line is made (slice.c ff_init_slice_from_src,
     * hscale.c lum_convert): the same row unless SWS_SRC_V_CHR_DROP gave the RGB source a vertical chroma sub-sampling */
    const ptrdiff_t yc = y >> c->chrSrcVSub;
    int i;
    if (isSemiPlanarYUV(f) && desc_get(f)->c[0].depth > 8) { /* p010/p012 LEToY_c input.c:979-1007; p016 reads the plane directly */
        const int sh = desc_get(f)->c[0].shift;
        const uint16_t *s = (const uint16_t *)(src[0] + y * stride[0]); uint16_t *d = (uint16_t *)tmp;
        if (!sh) return src[0] + y * stride[0];
        for (i = 0; i < w; i++) d[i] = s[i] >> sh;
        return tmp;
    }
    if (isPackedHi(f)) { /* y210/y212/y216 le_Y_c (input.c:580-606), read_ayuv64le_Y_c (:663), read_xv30le/v30xle/xv36le_Y_c (:811-857): the
                          * descriptor's field, (16-bit word at offset) >> shift, masked to the depth */
        const Desc *ds = desc_get(f);
        const uint8_t *s = src[0] + y * stride[0] + ds->c[0].offset;
        uint16_t *d = (uint16_t *)tmp;
        for (i = 0; i < w; i++) { uint16_t v; memcpy(&v, s + ds->c[0].step * i, 2); d[i] = (uint16_t)((v >> ds->c[0].shift) & ((1u << ds->c[0].depth) - 1)); }
        return tmp;
    }
    if (isPacked444(f)) { /* read_vuyx_Y_c / read_ayuv_Y_c / vyuToY_c input.c:741-799: the byte at the descriptor's Y offset */
        const Desc *ds = desc_get(f);
        const uint8_t *s = src[0] + y * stride[0] + ds->c[0].offset;
        for (i = 0; i < w; i++) tmp[i] = s[ds->c[0].step * i];
        return tmp;
    }
    if (f == ORF_YUYV422 || f == ORF_UYVY422 || f == ORF_YVYU422) { /* yuy2ToY_c input.c:550-556, uyvyToY_c :890-896 */
        const uint8_t *s = src[0] + y * stride[0] + desc_get(f)->c[0].offset;
        for (i = 0; i < w; i++) tmp[i] = s[2 * i];
        return tmp;
    }
    if (f == ORF_RGB48LE || f == ORF_BGR48LE || f == ORF_RGBA64LE || f == ORF_BGRA64LE) { /* rgb48ToY_c_template / rgb64ToY_c_template input.c:45-57, :136-150 */
        const Desc *ds = desc_get(f);
        const int st = ds->c[0].step / 2, ro = ds->c[0].offset / 2, go = ds->c[1].offset / 2, bo = ds->c[2].offset / 2;
        const uint16_t *s = (const uint16_t *)(src[0] + y * stride[0]);
        uint16_t *d = (uint16_t *)tmp;
        for (i = 0; i < w; i++)
            d[i] = (uint16_t)(((unsigned)t[RY] * s[st * i + ro] + (unsigned)t[GY] * s[st * i + go] + (unsigned)t[BY] * s[st * i + bo] + (0x2001u << 14)) >> 15);
        return tmp;
    }
    if (isMono(f)) { /* monowhite2Y_c / monoblack2Y_c input.c:514-548: MSB-first bits -> 0 or 16383 */
        const uint8_t *s = src[0] + y * stride[0]; int16_t *d = (int16_t *)tmp;
        for (i = 0; i < w; i++) {
            const int v = f == ORF_MONOWHITE ? ~s[i >> 3] : s[i >> 3];
            d[i] = (int16_t)(((v >> (7 - (i & 7))) & 1) * 16383);
        }
        return tmp;
    }
    if (isRGB30(f)) { /* rgb16_32ToY_c_template input.c:264-293 with the rgb30le / bgr30le rows of :411-412 */
        const uint8_t *s = src[0] + y * stride[0]; int16_t *d = (int16_t *)tmp;
        const int x2rgb = f == ORF_X2RGB10LE, shr = x2rgb ? 16 : 0, shg = 6, shb = x2rgb ? 0 : 16, S = 15 + 6;
        const int maskr = x2rgb ? 0x3FF00000 : 0x3FF, maskg = 0xFFC00, maskb = x2rgb ? 0x3FF : 0x3FF00000;
        const int ry = t[RY] << (x2rgb ? 0 : 4), gy = t[GY], by = t[BY] << (x2rgb ? 4 : 0);
        const unsigned rnd = (32u << (S - 1)) + (1u << (S - 7));
        for (i = 0; i < w; i++) {
            uint32_t pv; int px, r, g, b;
            memcpy(&pv, s + 4 * i, 4);
            px = (int)pv; b = (px & maskb) >> shb; g = (px & maskg) >> shg; r = (px & maskr) >> shr;
            d[i] = (int16_t)((unsigned)(ry * r + gy * g + by * b + rnd) >> (S - 6));
        }
        return tmp;
    }
    if (isRGB16(f)) { /* rgb16_32ToY_c_template input.c:264-293 with the 12/15/16 bpp rows of :396-401 */
        const uint16_t *s = (const uint16_t *)(src[0] + y * stride[0]); int16_t *d = (int16_t *)tmp;
        int maskr, maskg, maskb, rsh, gsh, bsh, S;
        rgb16_params(f, &maskr, &maskg, &maskb, &rsh, &gsh, &bsh, &S);
        {
            const int ry = t[RY] << rsh, gy = t[GY] << gsh, by = t[BY] << bsh;
            const unsigned rnd = (32u << (S - 1)) + (1u << (S - 7));
            for (i = 0; i < w; i++) {
                const int px = s[i], b = px & maskb, g = px & maskg, r = px & maskr;
                d[i] = (int16_t)((unsigned)(ry * r + gy * g + by * b + rnd) >> (S - 6));
            }
        }
        return tmp;
    }
    if (isPalSrc(f)) { /* palToY_c input.c:486-496 */
        const uint8_t *s = src[0] + y * stride[0]; int16_t *d = (int16_t *)tmp;
        for (i = 0; i < w; i++) d[i] = (int16_t)((c->pal_yuv[s[i]] & 0xFF) << 6);
        return tmp;
    }
    if (f == ORF_UYYVYY411) { /* uyyvyyToY_c input.c:909-914 */
        const uint8_t *s = src[0] + y * stride[0];
        for (i = 0; i < w; i++) tmp[i] = s[3 * (i >> 1) + 1 + (i & 1)];
        return tmp;
    }
    if (isPackedFloatRGB(f)) { /* rgbf32_to_y_c (input.c:1380-1397), rgbf16ToY_endian (:1726-1740), rgbaf16ToY_endian (:1666-1678) */
        const Desc *ds = desc_get(f);
        const int half = ds->c[0].depth == 16, esz = half ? 2 : 4, st = ds->c[0].step;
        const uint8_t *s = src[0] + y * stride[0];
        uint16_t *d = (uint16_t *)tmp;
        for (i = 0; i < w; i++) {
            const int r = rdf16(s + st * i, half), g = rdf16(s + st * i + esz, half), b = rdf16(s + st * i + 2 * esz, half);
            d[i] = (uint16_t)((int)((unsigned)t[RY] * r + (unsigned)t[GY] * g + (unsigned)t[BY] * b + (0x2001u << (15 - 1))) >> 15);
        }
        return tmp;
    }
    if (isFloatGrayX(f)) { /* grayf16ToY16_c (input.c:1601-1609), read_yaf32_gray_c (:1411-1420), read_yaf16_gray_c (:1611-1618) */
        const Desc *ds = desc_get(f);
        const int half = ds->c[0].depth == 16, st = ds->c[0].step;
        const uint8_t *s = src[0] + y * stride[0];
        uint16_t *d = (uint16_t *)tmp;
        for (i = 0; i < w; i++) d[i] = (uint16_t)rdf16(s + st * i, half);
        return tmp;
    }
    if (f == ORF_GBRPF16LE || f == ORF_GBRAPF16LE) { /* planar_rgbf16_to_y input.c:1586-1599 */
        const uint8_t *G = src[0] + y * stride[0], *B = src[1] + yc * stride[1], *R = src[2] + yc * stride[2];
        uint16_t *d = (uint16_t *)tmp;
        for (i = 0; i < w; i++) {
            const int g = rdf16(G + 2 * i, 1), b = rdf16(B + 2 * i, 1), r = rdf16(R + 2 * i, 1);
            d[i] = (uint16_t)((int)((unsigned)t[RY] * r + (unsigned)t[GY] * g + (unsigned)t[BY] * b + (0x2001u << (15 - 1))) >> 15);
        }
        return tmp;
    }
    switch (f) {
    case ORF_RGB24: case ORF_BGR24: { /* rgb24ToY_c / bgr24ToY_c input.c:1068-1124 */
        const uint8_t *s = src[0] + y * stride[0]; int16_t *d = (int16_t *)tmp;
        int ro = f == ORF_RGB24 ? 0 : 2, bo = 2 - ro;
        for (i = 0; i < w; i++) {
            int r = s[3 * i + ro], g = s[3 * i + 1], b = s[3 * i + bo];
            d[i] = (int16_t)((t[RY] * r + t[GY] * g + t[BY] * b + (32 << (15 - 1)) + (1 << (15 - 7))) >> (15 - 6));
        }
        return tmp; }
    case ORF_RGBA: case ORF_BGRA: case ORF_ARGB: case ORF_ABGR: { /* rgb16_32ToY_c_template input.c:264-293 */
        const uint8_t *s = src[0] + y * stride[0]; int16_t *d = (int16_t *)tmp;
        /* byte positions of r,g,b inside the pixel */
        const Desc *ds = desc_get(f);
        int ro = ds->c[0].offset, go = ds->c[1].offset, bo = ds->c[2].offset;
        const int S = 15 + 8;
        const int ry = t[RY] << 8, gy = t[GY] << 0, by = t[BY] << 8;
        const unsigned rnd = (32u << (S - 1)) + (1u << (S - 7));
        for (i = 0; i < w; i++) {
            /* the template extracts g as (px & 0xFF00) (i.e. g<<8) and r,b as 0..255 with coeffs <<8 */
            int r = s[4 * i + ro], g = s[4 * i + go] << 8, b = s[4 * i + bo];
            d[i] = (int16_t)((unsigned)(ry * r + gy * g + by * b + rnd) >> (S - 6)); /* unsigned expression, logical shift */
        }
        return tmp; }
    case ORF_GBRAP:
    case ORF_GBRP: { /* planar_rgb_to_y input.c:1174-1186 */
        const uint8_t *G = src[0] + y * stride[0], *B = src[1] + yc * stride[1], *R = src[2] + yc * stride[2];
        uint16_t *d = (uint16_t *)tmp;
        for (i = 0; i < w; i++)
            d[i] = (uint16_t)((int)((unsigned)t[RY] * R[i] + (unsigned)t[GY] * G[i] + (unsigned)t[BY] * B[i] + (0x801 << (15 - 7))) >> (15 - 6));
        return tmp; }
    case ORF_GBRP10MSBLE: case ORF_GBRP12MSBLE:   /* planar_rgb16_s10 / s12_to_y: samples >> (16 - bits) (input.c:1216-1232, :1462-1474) */
    case ORF_GBRAP10LE: case ORF_GBRAP12LE: case ORF_GBRAP14LE: case ORF_GBRAP16LE:
    case ORF_GBRP9LE: case ORF_GBRP10LE: case ORF_GBRP12LE: case ORF_GBRP14LE: case ORF_GBRP16LE: { /* planar_rgb16_s16_to_y input.c:1216-1232 */
        const uint16_t *G = (const uint16_t *)(src[0] + y * stride[0]), *B = (const uint16_t *)(src[1] + yc * stride[1]),
                       *R = (const uint16_t *)(src[2] + yc * stride[2]);
        const int bpc = desc_get(f)->c[0].depth, shift = bpc < 16 ? bpc : 14, ms = desc_get(f)->c[0].shift;
        uint16_t *d = (uint16_t *)tmp;
        for (i = 0; i < w; i++)
            d[i] = (uint16_t)((int)((unsigned)t[RY] * (R[i] >> ms) + (unsigned)t[GY] * (G[i] >> ms) + (unsigned)t[BY] * (B[i] >> ms) +
                                    (16u << (15 + bpc - 8)) + (1u << (15 + shift - 15))) >> (15 + shift - 14));
        return tmp; }
    case ORF_YA8: /* yuy2ToY_c on the gray byte (input.c:2400-2403) */
        for (i = 0; i < w; i++) tmp[i] = src[0][(ptrdiff_t)y * stride[0] + 2 * i];
        return tmp;
    case ORF_YA16LE: { /* read_ya16le_gray_c input.c:631-637 */
        uint16_t *d = (uint16_t *)tmp;
        for (i = 0; i < w; i++) memcpy(&d[i], src[0] + (ptrdiff_t)y * stride[0] + 4 * i, 2);
        return tmp; }
    case ORF_GRAYF32LE: { /* grayf32ToY16_c input.c:1399-1409 */
        const float *s = (const float *)(src[0] + y * stride[0]);
        uint16_t *d = (uint16_t *)tmp;
        for (i = 0; i < w; i++) d[i] = (uint16_t)f2u16(s[i]);
        return tmp; }
    case ORF_GBRAPF32LE:
    case ORF_GBRPF32LE: { /* planar_rgbf32_to_y input.c:1319-1334 */
        const float *G = (const float *)(src[0] + y * stride[0]), *B = (const float *)(src[1] + yc * stride[1]),
                    *R = (const float *)(src[2] + yc * stride[2]);
        uint16_t *d = (uint16_t *)tmp;
        for (i = 0; i < w; i++) {
            int g = f2u16(G[i]), b = f2u16(B[i]), r = f2u16(R[i]);
            d[i] = (uint16_t)((int)((unsigned)t[RY] * r + (unsigned)t[GY] * g + (unsigned)t[BY] * b + (0x2001u << (15 - 1))) >> 15);
        }
        return tmp; }
    default:
        if ((desc_get(f)->flags & PF_PLANAR) && desc_get(f)->c[0].shift && desc_get(f)->c[0].depth > 8) { /* shf16_10LEToY_c / shf16_12LEToY_c (yuv444pNNmsb) */
            const int sh = desc_get(f)->c[0].shift;
            const uint16_t *s = (const uint16_t *)(src[0] + y * stride[0]); uint16_t *d = (uint16_t *)tmp;
            for (i = 0; i < w; i++) d[i] = s[i] >> sh;
            return tmp;
        }
        return src[0] + y * stride[0];
    }
}

/* chroma line for chroma source row y -> (u,v) lines of chrSrcW samples */
static void read_chr_line(const OrSws *c, const uint8_t *const src[], const int stride[], int y, int lum_row,
                          uint8_t *tu, uint8_t *tv, const uint8_t **pu, const uint8_t **pv)
{
    const int f = c->o.src_format, w = c->chrSrcW;
    const int32_t *t = c->rgb2yuv;
    /* planar RGB: plane 0 is indexed by luma row.  chr_convert (hscale.c:211-225) computes that row ONCE per batch of chroma lines,
     * "sp0 = (sliceY - (plane[0].sliceY >> v_chr_sub_sample)) << v_chr_sub_sample", and then steps it by one luma line per chroma line
     * ("line[sp0 + i]"): the caller (main_path) hands in batch_start << chrSrcVSub + (y - batch_start) */
    const ptrdiff_t yl = lum_row;
    int i;
    *pu = tu; *pv = tv;
    if (isPalSrc(f)) { /* palToUV_c input.c:498-512 */
        const uint8_t *s = src[0] + (ptrdiff_t)(y << c->chrSrcVSub) * stride[0];
        uint16_t *du = (uint16_t *)tu; int16_t *dv = (int16_t *)tv;
        for (i = 0; i < w; i++) { const uint32_t p = c->pal_yuv[s[i]]; du[i] = (uint16_t)((uint8_t)(p >> 8) << 6); dv[i] = (int16_t)((uint8_t)(p >> 16) << 6); }
        return;
    }
    if (f == ORF_UYYVYY411) { /* uyyvyyToUV_c input.c:916-925 */
        const uint8_t *s = src[0] + (ptrdiff_t)(y << c->chrSrcVSub) * stride[0];
        for (i = 0; i < w; i++) { tu[i] = s[6 * i]; tv[i] = s[6 * i + 3]; }
        return;
    }
    if (isPackedFloatRGB(f)) { /* rgbf32_to_uv_c / _uv_half_c (input.c:1336-1378), rgbf16ToUV(_half)_endian (:1689-1724), rgbaf16ToUV(_half)_endian (:1629-1664):
                                * the half forms average the two converted pixels with a plain >> 1 */
        const Desc *ds = desc_get(f);
        const int half = ds->c[0].depth == 16, esz = half ? 2 : 4, st = ds->c[0].step;
        const uint8_t *s = src[0] + (ptrdiff_t)(y << c->chrSrcVSub) * stride[0];
        uint16_t *du = (uint16_t *)tu, *dv = (uint16_t *)tv;
        for (i = 0; i < w; i++) {
            int r, g, b;
            if (c->chrSrcHSub) {
                r = (rdf16(s + 2 * st * i, half) + rdf16(s + 2 * st * i + st, half)) >> 1;
                g = (rdf16(s + 2 * st * i + esz, half) + rdf16(s + 2 * st * i + st + esz, half)) >> 1;
                b = (rdf16(s + 2 * st * i + 2 * esz, half) + rdf16(s + 2 * st * i + st + 2 * esz, half)) >> 1;
            } else { r = rdf16(s + st * i, half); g = rdf16(s + st * i + esz, half); b = rdf16(s + st * i + 2 * esz, half); }
            du[i] = (uint16_t)((int)((unsigned)t[RU] * r + (unsigned)t[GU] * g + (unsigned)t[BU] * b + (0x10001u << (15 - 1))) >> 15);
            dv[i] = (uint16_t)((int)((unsigned)t[RV] * r + (unsigned)t[GV] * g + (unsigned)t[BV] * b + (0x10001u << (15 - 1))) >> 15);
        }
        return;
    }
    if (f == ORF_GBRPF16LE || f == ORF_GBRAPF16LE) { /* planar_rgbf16_to_uv input.c:1570-1584 */
        const uint8_t *G = src[0] + yl * stride[0], *B = src[1] + (ptrdiff_t)y * stride[1], *R = src[2] + (ptrdiff_t)y * stride[2];
        uint16_t *du = (uint16_t *)tu, *dv = (uint16_t *)tv;
        for (i = 0; i < w; i++) {
            const int g = rdf16(G + 2 * i, 1), b = rdf16(B + 2 * i, 1), r = rdf16(R + 2 * i, 1);
            du[i] = (uint16_t)((int)((unsigned)t[RU] * r + (unsigned)t[GU] * g + (unsigned)t[BU] * b + (0x10001u << (15 - 1))) >> 15);
            dv[i] = (uint16_t)((int)((unsigned)t[RV] * r + (unsigned)t[GV] * g + (unsigned)t[BV] * b + (0x10001u << (15 - 1))) >> 15);
        }
        return;
    }
    if (isRGB30(f)) { /* rgb16_32ToUV_c_template / rgb16_32ToUV_half_c_template input.c:295-372 with the rows of :411-412 */
        const uint8_t *s = src[0] + (ptrdiff_t)(y << c->chrSrcVSub) * stride[0];
        int16_t *du = (int16_t *)tu, *dv = (int16_t *)tv;
        const int x2rgb = f == ORF_X2RGB10LE, shr = x2rgb ? 16 : 0, shg = 6, shb = x2rgb ? 0 : 16, S = 15 + 6;
        int maskr = x2rgb ? 0x3FF00000 : 0x3FF, maskg = 0xFFC00, maskb = x2rgb ? 0x3FF : 0x3FF00000;
        const int rsh = x2rgb ? 0 : 4, bsh = x2rgb ? 4 : 0;
        const int ru = t[RU] * (1 << rsh), gu = t[GU], bu = t[BU] * (1 << bsh), rv = t[RV] * (1 << rsh), gv = t[GV], bv = t[BV] * (1 << bsh);
        if (c->chrSrcHSub) {
            const unsigned maskgx = ~(unsigned)(maskr | maskb);
            const unsigned rnd = (256U << S) + (1u << (S - 6));
            maskr |= maskr << 1; maskb |= maskb << 1; maskg |= maskg << 1;
            for (i = 0; i < w; i++) {
                uint32_t px0, px1; int b, r, g, rb;
                memcpy(&px0, s + 8 * i, 4); memcpy(&px1, s + 8 * i + 4, 4);
                g = (int)((px0 & maskgx) + (px1 & maskgx));
                rb = (int)(px0 + px1 - (unsigned)g);
                b = (rb & maskb) >> shb; g = (g & maskg) >> shg; r = (rb & maskr) >> shr;
                du[i] = (int16_t)((unsigned)(ru * r + gu * g + bu * b + rnd) >> (S - 6 + 1));
                dv[i] = (int16_t)((unsigned)(rv * r + gv * g + bv * b + rnd) >> (S - 6 + 1));
            }
        } else {
            const unsigned rnd = (256u << (S - 1)) + (1u << (S - 7));
            for (i = 0; i < w; i++) {
                uint32_t pv; int px, r, g, b;
                memcpy(&pv, s + 4 * i, 4);
                px = (int)pv; b = (px & maskb) >> shb; g = (px & maskg) >> shg; r = (px & maskr) >> shr;
                du[i] = (int16_t)((unsigned)(ru * r + gu * g + bu * b + rnd) >> (S - 6));
                dv[i] = (int16_t)((unsigned)(rv * r + gv * g + bv * b + rnd) >> (S - 6));
            }
        }
        return;
    }
    if (isRGB16(f)) { /* rgb16_32ToUV_c_template / rgb16_32ToUV_half_c_template input.c:295-372 */
        const uint16_t *s = (const uint16_t *)(src[0] + (ptrdiff_t)(y << c->chrSrcVSub) * stride[0]);
        int16_t *du = (int16_t *)tu, *dv = (int16_t *)tv;
        int maskr, maskg, maskb, rsh, gsh, bsh, S;
        rgb16_params(f, &maskr, &maskg, &maskb, &rsh, &gsh, &bsh, &S);
        {
            const int ru = t[RU] * (1 << rsh), gu = t[GU] * (1 << gsh), bu = t[BU] * (1 << bsh);
            const int rv = t[RV] * (1 << rsh), gv = t[GV] * (1 << gsh), bv = t[BV] * (1 << bsh);
            if (c->chrSrcHSub) {
                const unsigned maskgx = ~(unsigned)(maskr | maskb);
                const unsigned rnd = (256U << S) + (1u << (S - 6));
                const int is565 = f == ORF_RGB565LE || f == ORF_BGR565LE;
                const int mr = maskr | (maskr << 1), mb = maskb | (maskb << 1), mg = maskg | (maskg << 1);
                for (i = 0; i < w; i++) {
                    const unsigned px0 = s[2 * i], px1 = s[2 * i + 1];
                    int g = (int)((px0 & maskgx) + (px1 & maskgx));
                    const int rb = (int)(px0 + px1) - g;
                    const int b = rb & mb, r = rb & mr;
                    if (!is565) g = g & mg;            /* shp == 0: only the 565 formats keep the unmasked green sum (:344-351) */
                    du[i] = (int16_t)((unsigned)(ru * r + gu * g + bu * b + rnd) >> (S - 6 + 1));
                    dv[i] = (int16_t)((unsigned)(rv * r + gv * g + bv * b + rnd) >> (S - 6 + 1));
                }
            } else {
                const unsigned rnd = (256u << (S - 1)) + (1u << (S - 7));
                for (i = 0; i < w; i++) {
                    const int px = s[i], b = px & maskb, g = px & maskg, r = px & maskr;
                    du[i] = (int16_t)((unsigned)(ru * r + gu * g + bu * b + rnd) >> (S - 6));
                    dv[i] = (int16_t)((unsigned)(rv * r + gv * g + bv * b + rnd) >> (S - 6));
                }
            }
        }
        return;
    }
    if (f == ORF_RGB48LE || f == ORF_BGR48LE || f == ORF_RGBA64LE || f == ORF_BGRA64LE) { /* rgb48/64ToUV(_half)_c_template input.c:58-96, :151-203 */
        const Desc *ds = desc_get(f);
        const int st = ds->c[0].step / 2, ro = ds->c[0].offset / 2, go = ds->c[1].offset / 2, bo = ds->c[2].offset / 2;
        const uint16_t *s = (const uint16_t *)(src[0] + (ptrdiff_t)(y << c->chrSrcVSub) * stride[0]);
        uint16_t *du = (uint16_t *)tu, *dv = (uint16_t *)tv;
        for (i = 0; i < w; i++) {
            unsigned r, g, b;
            if (c->chrSrcHSub) {
                r = (s[2 * st * i + ro] + s[2 * st * i + st + ro] + 1u) >> 1;
                g = (s[2 * st * i + go] + s[2 * st * i + st + go] + 1u) >> 1;
                b = (s[2 * st * i + bo] + s[2 * st * i + st + bo] + 1u) >> 1;
            } else { r = s[st * i + ro]; g = s[st * i + go]; b = s[st * i + bo]; }
            du[i] = (uint16_t)(((unsigned)t[RU] * r + (unsigned)t[GU] * g + (unsigned)t[BU] * b + (0x10001u << 14)) >> 15);
            dv[i] = (uint16_t)(((unsigned)t[RV] * r + (unsigned)t[GV] * g + (unsigned)t[BV] * b + (0x10001u << 14)) >> 15);
        }
        return;
    }
    if (isPackedHi(f)) { /* y2xxle_UV_c, read_ayuv64le/xv48le_UV_c, read_xv30le/v30xle/xv36le_UV_c */
        const Desc *ds = desc_get(f);
        const uint8_t *s = src[0] + (ptrdiff_t)(y << c->chrSrcVSub) * stride[0];   /* (non-zero only with SWS_SRC_V_CHR_DROP) */
        uint16_t *a = (uint16_t *)tu, *b = (uint16_t *)tv;
        for (i = 0; i < w; i++) {
            uint16_t u, v;
            memcpy(&u, s + ds->c[1].step * i + ds->c[1].offset, 2); memcpy(&v, s + ds->c[2].step * i + ds->c[2].offset, 2);
            a[i] = (uint16_t)((u >> ds->c[1].shift) & ((1u << ds->c[1].depth) - 1)); b[i] = (uint16_t)((v >> ds->c[2].shift) & ((1u << ds->c[2].depth) - 1));
        }
        return;
    }
    if (isPacked444(f)) { /* read_vuyx_UV_c / read_ayuv_UV_c / read_uyva_UV_c / vyuToUV_c input.c:731-809 */
        const Desc *ds = desc_get(f);
        const uint8_t *s = src[0] + (ptrdiff_t)(y << c->chrSrcVSub) * stride[0];
        for (i = 0; i < w; i++) { tu[i] = s[ds->c[0].step * i + ds->c[1].offset]; tv[i] = s[ds->c[0].step * i + ds->c[2].offset]; }
        return;
    }
    if (f == ORF_YUYV422 || f == ORF_UYVY422 || f == ORF_YVYU422) { /* yuy2ToUV_c / yvy2ToUV_c input.c:558-578, uyvyToUV_c :898-907 */
        const Desc *ds = desc_get(f);
        const uint8_t *s = src[0] + (ptrdiff_t)(y << c->chrSrcVSub) * stride[0];
        for (i = 0; i < w; i++) { tu[i] = s[4 * i + ds->c[1].offset]; tv[i] = s[4 * i + ds->c[2].offset]; }
        return;
    }
    if (isSemiPlanarYUV(f) && desc_get(f)->c[0].depth == 8) { /* nv12ToUV_c / nv21ToUV_c input.c:926-948 (nv12/16/24, nv21/42) */
        const uint8_t *s = src[1] + y * stride[1];
        const int swapped = isSwappedChroma(f);
        uint8_t *a = swapped ? tv : tu, *b = swapped ? tu : tv;
        for (i = 0; i < w; i++) { a[i] = s[2 * i]; b[i] = s[2 * i + 1]; }
        return;
    }
    if (isSemiPlanarYUV(f)) { /* p010/p012/p016 LEToUV_c input.c:950-1008 (and the 4:2:2 / 4:4:4 siblings) */
        const int sh = desc_get(f)->c[1].shift;
        const uint16_t *s = (const uint16_t *)(src[1] + y * stride[1]);
        uint16_t *a = (uint16_t *)tu, *b = (uint16_t *)tv;
        for (i = 0; i < w; i++) { a[i] = s[2 * i] >> sh; b[i] = s[2 * i + 1] >> sh; }
        return;
    }
    switch (f) {
    case ORF_RGB24: case ORF_BGR24: { /* rgb24ToUV(_half)_c, bgr24ToUV(_half)_c input.c:1082-1172 */
        /* chroma row y reads source row y << chrSrcVSub (hscale.c:212) */
        const uint8_t *s = src[0] + (y << c->chrSrcVSub) * stride[0];
        int16_t *du = (int16_t *)tu, *dv = (int16_t *)tv;
        int ro = f == ORF_RGB24 ? 0 : 2, bo = 2 - ro;
        if (c->chrSrcHSub) {
            for (i = 0; i < w; i++) {
                int r = s[6 * i + ro] + s[6 * i + 3 + ro], g = s[6 * i + 1] + s[6 * i + 4], b = s[6 * i + bo] + s[6 * i + 3 + bo];
                du[i] = (int16_t)((t[RU] * r + t[GU] * g + t[BU] * b + (256 << 15) + (1 << (15 - 6))) >> (15 - 5));
                dv[i] = (int16_t)((t[RV] * r + t[GV] * g + t[BV] * b + (256 << 15) + (1 << (15 - 6))) >> (15 - 5));
            }
        } else {
            for (i = 0; i < w; i++) {
                int r = s[3 * i + ro], g = s[3 * i + 1], b = s[3 * i + bo];
                du[i] = (int16_t)((t[RU] * r + t[GU] * g + t[BU] * b + (256 << (15 - 1)) + (1 << (15 - 7))) >> (15 - 6));
                dv[i] = (int16_t)((t[RV] * r + t[GV] * g + t[BV] * b + (256 << (15 - 1)) + (1 << (15 - 7))) >> (15 - 6));
            }
        }
        return; }
    case ORF_RGBA: case ORF_BGRA: case ORF_ARGB: case ORF_ABGR: { /* rgb16_32ToUV(_half)_c_template input.c:295-372 */
        const uint8_t *s = src[0] + (y << c->chrSrcVSub) * stride[0];
        int16_t *du = (int16_t *)tu, *dv = (int16_t *)tv;
        const Desc *ds = desc_get(f);
        int ro = ds->c[0].offset, go = ds->c[1].offset, bo = ds->c[2].offset;
        const int S = 15 + 8;
        const int ru = t[RU] * (1 << 8), gu = t[GU], bu = t[BU] * (1 << 8);
        const int rv = t[RV] * (1 << 8), gv = t[GV], bv = t[BV] * (1 << 8);
        if (c->chrSrcHSub) {
            const unsigned rnd = (256U << S) + (1 << (S - 6));
            for (i = 0; i < w; i++) {
                /* sums of two pixels; g keeps its <<8 position (maskg|maskg<<1 applied to the sum) */
                int r = s[8 * i + ro] + s[8 * i + 4 + ro];
                int g = (s[8 * i + go] + s[8 * i + 4 + go]) << 8;
                int b = s[8 * i + bo] + s[8 * i + 4 + bo];
                /* rnd is unsigned (256U << S == 2^31): the whole expression is unsigned, the shift is logical */
                du[i] = (int16_t)((unsigned)(ru * r + gu * g + bu * b + rnd) >> (S - 6 + 1));
                dv[i] = (int16_t)((unsigned)(rv * r + gv * g + bv * b + rnd) >> (S - 6 + 1));
            }
        } else {
            const unsigned rnd = (256u << (S - 1)) + (1 << (S - 7));
            for (i = 0; i < w; i++) {
                int r = s[4 * i + ro], g = s[4 * i + go] << 8, b = s[4 * i + bo];
                du[i] = (int16_t)((unsigned)(ru * r + gu * g + bu * b + rnd) >> (S - 6));
                dv[i] = (int16_t)((unsigned)(rv * r + gv * g + bv * b + rnd) >> (S - 6));
            }
        }
        return; }
    case ORF_GBRAP:
    case ORF_GBRP: {
        const uint8_t *G = src[0] + yl * stride[0], *B = src[1] + y * stride[1], *R = src[2] + y * stride[2];
        uint16_t *du = (uint16_t *)tu, *dv = (uint16_t *)tv;
        if (c->chrSrcHSub) { /* gbr24pToUV_half_c input.c:412-432 */
            for (i = 0; i < w; i++) {
                unsigned g = G[2 * i] + G[2 * i + 1], b = B[2 * i] + B[2 * i + 1], r = R[2 * i] + R[2 * i + 1];
                du[i] = (uint16_t)((int)(t[RU] * r + t[GU] * g + t[BU] * b + (0x4001 << (15 - 6))) >> (15 - 6 + 1));
                dv[i] = (uint16_t)((int)(t[RV] * r + t[GV] * g + t[BV] * b + (0x4001 << (15 - 6))) >> (15 - 6 + 1));
            }
        } else { /* planar_rgb_to_uv input.c:1196-1211 */
            for (i = 0; i < w; i++) {
                int g = G[i], b = B[i], r = R[i];
                du[i] = (uint16_t)((int)((unsigned)t[RU] * r + (unsigned)t[GU] * g + (unsigned)t[BU] * b + (0x4001 << (15 - 7))) >> (15 - 6));
                dv[i] = (uint16_t)((int)((unsigned)t[RV] * r + (unsigned)t[GV] * g + (unsigned)t[BV] * b + (0x4001 << (15 - 7))) >> (15 - 6));
            }
        }
        return; }
    case ORF_GBRP10MSBLE: case ORF_GBRP12MSBLE:
    case ORF_GBRAP10LE: case ORF_GBRAP12LE: case ORF_GBRAP14LE: case ORF_GBRAP16LE:
    case ORF_GBRP9LE: case ORF_GBRP10LE: case ORF_GBRP12LE: case ORF_GBRP14LE: case ORF_GBRP16LE: { /* planar_rgb16_s16_to_uv input.c:1248-1270 */
        const uint16_t *G = (const uint16_t *)(src[0] + yl * stride[0]), *B = (const uint16_t *)(src[1] + y * stride[1]),
                       *R = (const uint16_t *)(src[2] + y * stride[2]);
        const int bpc = desc_get(f)->c[0].depth, shift = bpc < 16 ? bpc : 14, ms = desc_get(f)->c[0].shift;
        uint16_t *du = (uint16_t *)tu, *dv = (uint16_t *)tv;
        for (i = 0; i < w; i++) {
            du[i] = (uint16_t)((int)((unsigned)t[RU] * (R[i] >> ms) + (unsigned)t[GU] * (G[i] >> ms) + (unsigned)t[BU] * (B[i] >> ms) +
                                     (128u << (15 + bpc - 8)) + (1u << (15 + shift - 15))) >> (15 + shift - 14));
            dv[i] = (uint16_t)((int)((unsigned)t[RV] * (R[i] >> ms) + (unsigned)t[GV] * (G[i] >> ms) + (unsigned)t[BV] * (B[i] >> ms) +
                                     (128u << (15 + bpc - 8)) + (1u << (15 + shift - 15))) >> (15 + shift - 14));
        }
        return; }
    case ORF_GBRAPF32LE:
    case ORF_GBRPF32LE: { /* planar_rgbf32_to_uv input.c:1300-1317 */
        const float *G = (const float *)(src[0] + yl * stride[0]), *B = (const float *)(src[1] + y * stride[1]),
                    *R = (const float *)(src[2] + y * stride[2]);
        uint16_t *du = (uint16_t *)tu, *dv = (uint16_t *)tv;
        for (i = 0; i < w; i++) {
            int g = f2u16(G[i]), b = f2u16(B[i]), r = f2u16(R[i]);
            du[i] = (uint16_t)((int)((unsigned)t[RU] * r + (unsigned)t[GU] * g + (unsigned)t[BU] * b + (0x10001u << (15 - 1))) >> 15);
            dv[i] = (uint16_t)((int)((unsigned)t[RV] * r + (unsigned)t[GV] * g + (unsigned)t[BV] * b + (0x10001u << (15 - 1))) >> 15);
        }
        return; }
    default: { /* planar YUV: direct */
        const Desc *ds = desc_get(f);
        if (ds->c[1].shift && ds->c[1].depth > 8) { /* shf16_10LEToUV_c / shf16_12LEToUV_c */
            const uint16_t *su = (const uint16_t *)(src[ds->c[1].plane] + y * stride[ds->c[1].plane]);
            const uint16_t *sv = (const uint16_t *)(src[ds->c[2].plane] + y * stride[ds->c[2].plane]);
            uint16_t *a = (uint16_t *)tu, *b = (uint16_t *)tv;
            for (i = 0; i < w; i++) { a[i] = su[i] >> ds->c[1].shift; b[i] = sv[i] >> ds->c[2].shift; }
            return;
        }
        *pu = src[ds->c[1].plane] + y * stride[ds->c[1].plane];
        *pv = src[ds->c[2].plane] + y * stride[ds->c[2].plane];
        return; }
    }
}

/* ------------------------------------------------------------------ */
/* main path: horizontal stage (swscale.c:69-159)                      */
/* ------------------------------------------------------------------ */
static void hscale_line(const OrSws *c, int32_t *dst, int dstW, const uint8_t *src,
                        const int16_t *filter, const int32_t *filterPos, int fs)
{
    const Desc *ds = desc_get(c->o.src_format);
    const int depth = ds->c[0].depth;
    const int rgbish = isAnyRGB(c->o.src_format) || c->o.src_format == ORF_PAL8;
    int i, j;
    if (c->srcBpc == 8 && c->dstBpc <= 14 && (c->o.flags & OR_SWS_FAST_BILINEAR)) {
        /* ff_hyscale_fast_c / ff_hcscale_fast_c (hscale_fast_bilinear.c:23-55), selected in sws_init_swscale (swscale.c:676-681) */
        const int chroma = filter == c->hChrFilter && filterPos == c->hChrFilterPos;
        const int srcW = chroma ? c->chrSrcW : c->o.src_w;
        const unsigned xInc = (unsigned)(chroma ? c->chrXInc : c->lumXInc);
        unsigned xpos = 0;
        for (i = 0; i < dstW; i++) {
            const unsigned xx = xpos >> 16, xalpha = (xpos & 0xFFFF) >> 9;
            if ((int)xx < srcW - 1)   /* the tail loop below overwrites every other position */
                dst[i] = chroma ? (int16_t)(src[xx] * (xalpha ^ 127) + src[xx + 1] * xalpha)
                                : (int16_t)((src[xx] << 7) + (src[xx + 1] - src[xx]) * xalpha);
            xpos += xInc;
        }
        for (i = dstW - 1; i >= 0 && ((i * (int64_t)xInc) >> 16) >= srcW - 1; i--) dst[i] = src[srcW - 1] * 128;
        return;
    }
    if (c->srcBpc == 8) {
        for (i = 0; i < dstW; i++) {
            int val = 0, sp = filterPos[i];
            for (j = 0; j < fs; j++) val += ((int)src[sp + j]) * filter[fs * i + j];
            if (c->dstBpc <= 14) dst[i] = (int16_t)ORMIN(val >> 7, (1 << 15) - 1);   /* hScale8To15_c */
            else dst[i] = ORMIN(val >> 3, (1 << 19) - 1);                             /* hScale8To19_c */
        }
    } else {
        const uint16_t *s = (const uint16_t *)src;
        int sh;
        if (c->dstBpc > 14) { /* hScale16To19_c :69-97 */
            sh = depth - 1 - 4;
            if (rgbish && depth < 16) sh = 9;
            else if (ds->flags & PF_FLOAT) sh = 16 - 1 - 4;
        } else { /* hScale16To15_c :99-125 */
            sh = depth - 1;
            if (sh < 15) sh = rgbish ? 13 : depth - 1;
            else if (ds->flags & PF_FLOAT) sh = 16 - 1;
        }
        for (i = 0; i < dstW; i++) {
            int val = 0, sp = filterPos[i];
            for (j = 0; j < fs; j++) val += s[sp + j] * filter[fs * i + j];
            if (c->dstBpc > 14) dst[i] = ORMIN(val >> sh, (1 << 19) - 1);
            else dst[i] = (int16_t)ORMIN(val >> sh, (1 << 15) - 1);
        }
    }
}

/* range conversion on the intermediate (swscale.c:163-255) */
static void range_line(const OrSws *c, int32_t *d, int w, int chroma)
{
    int i;
    if (c->dstBpc <= 14) {
        uint16_t coeff = (uint16_t)(chroma ? c->chrCoeff : c->lumCoeff);
        int32_t offset = (int32_t)(chroma ? c->chrOffset : c->lumOffset);
        for (i = 0; i < w; i++) {
            int v = (d[i] * coeff + offset) >> 14;
            if (!c->o.src_range) v = ORMIN(v, (1 << 15) - 1); /* ToJpeg */
            d[i] = (int16_t)v;
        }
    } else {
        uint32_t coeff = chroma ? c->chrCoeff : c->lumCoeff;
        int64_t offset = chroma ? c->chrOffset : c->lumOffset;
        for (i = 0; i < w; i++) {
            int v = (int)(((int64_t)d[i] * coeff + offset) >> 18);
            if (!c->o.src_range) v = ORMIN(v, (1 << 19) - 1);
            d[i] = v;
        }
    }
}

/* ------------------------------------------------------------------ */
/* main path: vertical stage + writers (vscale.c, output.c)            */
/* ------------------------------------------------------------------ */
typedef struct { const int32_t *rows[64]; } RowSet; /* only used for small fs; general path indexes planes */

/* planar writers: one output line of width w from fs rows. rows(j) = plane + (first+j clipped)*w */
/* slot of line r in a ring of mask + 1 lines (Planes below) */
#define RING(mask, r) ((size_t)((r) & (mask)))
static void write_planar_line(const OrSws *c, uint8_t *dest, int w, const int32_t *plane, int mask, int planeW, int planeH,
                              int first, const int16_t *filter, int fs, const uint8_t *dither, int offset,
                              int is_luma_of_p01x)
{
    const Desc *dd = desc_get(c->o.dst_format);
    const int bits = dd->c[0].depth;
    int i, j;
    const int32_t *rows_[fs > 0 ? fs : 1];       /* the lines of the row's vertical window, looked up once */
    for (j = 0; j < fs; j++) rows_[j] = plane + RING(mask, ORMIN(first + j, planeH - 1)) * planeW;
#define ROW(j) rows_[j]
    (void)is_luma_of_p01x;
    if (isDataInHighBits(c->o.dst_format) && bits < 16) { /* yuv2p01xl1_c / lX_c output.c:538-569; yuv2msbplane1/X_10_c_template :396-426 (same arithmetic) */
        uint16_t *d = (uint16_t *)dest;
        int oshift = 16 - bits;
        if (fs == 1) {
            int shift = 15 - bits; const int32_t *s = ROW(0);
            for (i = 0; i < w; i++) d[i] = (uint16_t)(clip_uintp2((s[i] + (1 << (shift - 1))) >> shift, bits) << oshift);
        } else {
            int shift = 11 + 16 - bits;
            for (i = 0; i < w; i++) {
                int val = 1 << (shift - 1);
                for (j = 0; j < fs; j++) val += ROW(j)[i] * filter[j];
                d[i] = (uint16_t)(clip_uintp2(val >> shift, bits) << oshift);
            }
        }
    } else if (bits == 32) { /* yuv2plane1_float / yuv2planeX_float_c_template (output.c:219-260): the 16-bit value times 1 / 65535 */
        static const float float_mult = 1.0f / 65535.0f;
        float *d = (float *)dest;
        if (fs == 1) {
            const int32_t *s = ROW(0);
            for (i = 0; i < w; i++) d[i] = float_mult * (float)clip_u16((s[i] + (1 << 2)) >> 3);
        } else {
            for (i = 0; i < w; i++) {
                int val = (1 << 14) - 0x40000000;
                for (j = 0; j < fs; j++) val = (int)((unsigned)val + ROW(j)[i] * (unsigned)filter[j]);
                d[i] = float_mult * (float)(0x8000 + clip_i16(val >> 15));
            }
        }
    } else if (bits == 16) { /* yuv2plane1_16 / planeX_16 output.c:149-187 */
        uint16_t *d = (uint16_t *)dest;
        if (fs == 1) {
            const int32_t *s = ROW(0);
            for (i = 0; i < w; i++) d[i] = (uint16_t)clip_u16((s[i] + (1 << 2)) >> 3);
        } else {
            for (i = 0; i < w; i++) {
                int val = (1 << 14) - 0x40000000;
                for (j = 0; j < fs; j++) val = (int)((unsigned)val + ROW(j)[i] * (unsigned)filter[j]);
                d[i] = (uint16_t)(0x8000 + clip_i16(val >> 15));
            }
        }
    } else if (bits >= 9 && bits <= 14) { /* yuv2plane1_10 / planeX_10 output.c:327-357 */
        uint16_t *d = (uint16_t *)dest;
        if (fs == 1) {
            int shift = 15 - bits; const int32_t *s = ROW(0);
            for (i = 0; i < w; i++) d[i] = (uint16_t)clip_uintp2((s[i] + (1 << (shift - 1))) >> shift, bits);
        } else {
            int shift = 11 + 16 - bits;
            for (i = 0; i < w; i++) {
                int val = 1 << (shift - 1);
                for (j = 0; j < fs; j++) val += ROW(j)[i] * filter[j];
                d[i] = (uint16_t)clip_uintp2(val >> shift, bits);
            }
        }
    } else { /* 8 bit: yuv2plane1_8_c / yuv2planeX_8_c output.c:468-493 */
        if (fs == 1) {
            const int32_t *s = ROW(0);
            for (i = 0; i < w; i++) dest[i] = (uint8_t)clip_u8((s[i] + dither[(i + offset) & 7]) >> 7);
        } else {
            for (i = 0; i < w; i++) {
                int val = dither[(i + offset) & 7] << 12;
                for (j = 0; j < fs; j++) val += (int)(unsigned)(ROW(j)[i] * filter[j]);
                dest[i] = (uint8_t)clip_u8(val >> 19);
            }
        }
    }
#undef ROW
}

/* interleaved chroma writers: yuv2nv12cX_c, yuv2p01xcX_c (output.c:495-528, 571-589) */
static void write_nv_chroma_line(const OrSws *c, uint8_t *dest, int w, const int32_t *up, const int32_t *vp, int mask,
                                 int planeW, int planeH, int first, const int16_t *filter, int fs, const uint8_t *dither)
{
    const Desc *dd = desc_get(c->o.dst_format);
    const int bits = dd->c[0].depth;
    int i, j;
    const int32_t *urows_[fs > 0 ? fs : 1], *vrows_[fs > 0 ? fs : 1];       /* the lines of the row's vertical window, looked up once */
    for (j = 0; j < fs; j++) { urows_[j] = up + RING(mask, ORMIN(first + j, planeH - 1)) * planeW; vrows_[j] = vp + RING(mask, ORMIN(first + j, planeH - 1)) * planeW; }
#define ROWU(j) urows_[j]
#define ROWV(j) vrows_[j]
    if (bits == 16) { /* yuv2nv12cX_16_c_template output.c:189-217 */
        uint16_t *d = (uint16_t *)dest;
        for (i = 0; i < w; i++) {
            int u = (1 << 14) - 0x40000000, v = (1 << 14) - 0x40000000;
            for (j = 0; j < fs; j++) {
                u = (int)((unsigned)u + ROWU(j)[i] * (unsigned)filter[j]);
                v = (int)((unsigned)v + ROWV(j)[i] * (unsigned)filter[j]);
            }
            d[2 * i] = (uint16_t)(0x8000 + clip_i16(u >> 15));
            d[2 * i + 1] = (uint16_t)(0x8000 + clip_i16(v >> 15));
        }
    } else if (isDataInHighBits(c->o.dst_format) || bits > 8) {   /* yuv2p010cX / yuv2p012cX / yuv2nv20cX (output.c:571-593, :650-652) */
        uint16_t *d = (uint16_t *)dest;
        int shift = 11 + 16 - bits, oshift = desc_get(c->o.dst_format)->c[0].shift;
        for (i = 0; i < w; i++) {
            int u = 1 << (shift - 1), v = 1 << (shift - 1);
            for (j = 0; j < fs; j++) {
                u = (int)((unsigned)u + ROWU(j)[i] * (unsigned)filter[j]);
                v = (int)((unsigned)v + ROWV(j)[i] * (unsigned)filter[j]);
            }
            d[2 * i] = (uint16_t)(clip_uintp2(u >> shift, bits) << oshift);
            d[2 * i + 1] = (uint16_t)(clip_uintp2(v >> shift, bits) << oshift);
        }
    } else {
        int swap = isSwappedChroma(c->o.dst_format);
        for (i = 0; i < w; i++) {
            int u = dither[i & 7] << 12, v = dither[(i + 3) & 7] << 12;
            for (j = 0; j < fs; j++) {
                u = (int)((unsigned)u + ROWU(j)[i] * (unsigned)filter[j]);
                v = (int)((unsigned)v + ROWV(j)[i] * (unsigned)filter[j]);
            }
            dest[2 * i + swap] = (uint8_t)clip_u8(u >> 19);
            dest[2 * i + 1 - swap] = (uint8_t)clip_u8(v >> 19);
        }
    }
#undef ROWU
#undef ROWV
}

/* LUT rgb pixel-pair write (yuv2rgb_write, output.c:1662-1785; 24/32 bpp) */
static void rgb_write2(const OrSws *c, uint8_t *dest, int i, int y, int Y1, int Y2, int U, int V, int hasAlpha, unsigned A1, unsigned A2, int second)
{
    const int d = c->o.dst_format;
    int r = c->table_rV[V + HEADROOM];
    int g = c->table_gU[U + HEADROOM] + c->table_gV[V + HEADROOM];
    int b = c->table_bU[U + HEADROOM];
    if (c->lut_elem == 2) { /* yuv2rgb_write output.c:1714-1748: ordered dither added to the luma index */
        int dr1, dg1, db1, dr2, dg2, db2;
        uint16_t v1, v2;
        if (d == ORF_RGB565LE || d == ORF_BGR565LE) {
            dr1 = dither_2x2_8[y & 1][0]; dg1 = dither_2x2_4[y & 1][0]; db1 = dither_2x2_8[(y & 1) ^ 1][0];
            dr2 = dither_2x2_8[y & 1][1]; dg2 = dither_2x2_4[y & 1][1]; db2 = dither_2x2_8[(y & 1) ^ 1][1];
        } else if (d == ORF_RGB555LE || d == ORF_BGR555LE) {
            dr1 = dither_2x2_8[y & 1][0]; dg1 = dither_2x2_8[y & 1][1]; db1 = dither_2x2_8[(y & 1) ^ 1][0];
            dr2 = dither_2x2_8[y & 1][1]; dg2 = dither_2x2_8[y & 1][0]; db2 = dither_2x2_8[(y & 1) ^ 1][1];
        } else {
            dr1 = dither_4x4_16[y & 3][0]; dg1 = dither_4x4_16[y & 3][1]; db1 = dither_4x4_16[(y & 3) ^ 3][0];
            dr2 = dither_4x4_16[y & 3][1]; dg2 = dither_4x4_16[y & 3][0]; db2 = dither_4x4_16[(y & 3) ^ 3][1];
        }
        v1 = (uint16_t)(lut_at(c, r + Y1 + dr1) + lut_at(c, g + Y1 + dg1) + lut_at(c, b + Y1 + db1));
        v2 = (uint16_t)(lut_at(c, r + Y2 + dr2) + lut_at(c, g + Y2 + dg2) + lut_at(c, b + Y2 + db2));
        memcpy(dest + 4 * i, &v1, 2); if (second) memcpy(dest + 4 * i + 2, &v2, 2);
    } else if (c->dstFormatBpp == 8 || c->dstFormatBpp == 4) { /* "8/4 bits", output.c:1755-1784: byte sums of the three planes */
        int dr1, dg1, db1, dr2, dg2, db2;
        uint8_t v1, v2;
        dither_rgb8(d, y, 2 * i, &dr1, &dg1, &db1);
        dither_rgb8(d, y, 2 * i + 1, &dr2, &dg2, &db2);
        v1 = (uint8_t)(lut_at(c, r + Y1 + dr1) + lut_at(c, g + Y1 + dg1) + lut_at(c, b + Y1 + db1));
        v2 = (uint8_t)(lut_at(c, r + Y2 + dr2) + lut_at(c, g + Y2 + dg2) + lut_at(c, b + Y2 + db2));
        if (isRGB4bits(d)) dest[i] = (uint8_t)(v1 + (v2 << 4));   /* the pair's byte is stored whole, also for the last pair of an odd width */
        else { dest[2 * i] = v1; if (second) dest[2 * i + 1] = v2; }
    } else if (c->lut_elem == 4) {
        uint32_t v1 = lut_at(c, r + Y1) + lut_at(c, g + Y1) + lut_at(c, b + Y1);
        uint32_t v2 = lut_at(c, r + Y2) + lut_at(c, g + Y2) + lut_at(c, b + Y2);
        if (hasAlpha) { /* yuv2rgb_write output.c:1680-1687 */
            const int sh = (d == ORF_ABGR || d == ORF_ARGB) ? 0 : 24;
            v1 += A1 << sh; v2 += A2 << sh;
        }
        memcpy(dest + 8 * i, &v1, 4); if (second) memcpy(dest + 8 * i + 4, &v2, 4);
    } else {
        uint8_t *p = dest + 6 * i;
        int rb = d == ORF_RGB24 ? r : b, br = d == ORF_RGB24 ? b : r;
        p[0] = (uint8_t)lut_at(c, rb + Y1); p[1] = (uint8_t)lut_at(c, g + Y1); p[2] = (uint8_t)lut_at(c, br + Y1);
        if (second) { p[3] = (uint8_t)lut_at(c, rb + Y2); p[4] = (uint8_t)lut_at(c, g + Y2); p[5] = (uint8_t)lut_at(c, br + Y2); }
    }
}

/* full-chroma pixel write (yuv2rgb_write_full, output.c:2005-2070; 8-bit per channel targets) */
static void rgb_write_full(OrSws *c, uint8_t *dest, int i, int y, int Y, int U, int V, int hasAlpha, int A, int err[4])
{
    const int d = c->o.dst_format;
    int R, G, B;
    Y -= c->yuv2rgb_y_offset;
    Y = (int)((unsigned)Y * (unsigned)c->yuv2rgb_y_coeff);
    Y = (int)((unsigned)Y + (1U << 21));
    R = (int)((unsigned)Y + (unsigned)V * (unsigned)c->yuv2rgb_v2r);
    G = (int)((unsigned)Y + (unsigned)V * (unsigned)c->yuv2rgb_v2g + (unsigned)U * (unsigned)c->yuv2rgb_u2g);
    B = (int)((unsigned)Y + (unsigned)U * (unsigned)c->yuv2rgb_u2b);
    if ((R | G | B) & 0xC0000000) {
        R = clip_uintp2(R, 30); G = clip_uintp2(G, 30); B = clip_uintp2(B, 30);
    }
    if (d == ORF_X2RGB10LE || d == ORF_X2BGR10LE) {   /* output.c:2052-2063 */
        uint32_t v;
        R >>= 20; G >>= 20; B >>= 20;
        v = d == ORF_X2RGB10LE ? (3U << 30) + ((unsigned)R << 20) + ((unsigned)G << 10) + (unsigned)B
                               : (3U << 30) + ((unsigned)B << 20) + ((unsigned)G << 10) + (unsigned)R;
        memcpy(dest, &v, 4);
        return;
    }
    if (isRGB8class(d)) {   /* output.c:2064-2158 */
        const int isrgb8 = d == ORF_BGR8 || d == ORF_RGB8;
        int r, g, b;
#define A_DITHER(u, v) (((((u) + ((v) * 236)) * 119) & 0xff))
#define X_DITHER(u, v) (((((u) ^ ((v) * 237)) * 181) & 0x1ff) / 2)
        switch (c->o.dither) {
        case 0: /* SWS_DITHER_NONE */
            if (isrgb8) { r = clip_uintp2(R >> 27, 3); g = clip_uintp2(G >> 27, 3); b = clip_uintp2(B >> 28, 2); }
            else { r = clip_uintp2(R >> 29, 1); g = clip_uintp2(G >> 28, 2); b = clip_uintp2(B >> 29, 1); }
            break;
        default: /* SWS_DITHER_AUTO / SWS_DITHER_ED: error diffusion along the row with the previous row's errors (c->dither_error) */
            R >>= 22; G >>= 22; B >>= 22;
            R += (7 * err[0] + 1 * c->dither_error[0][i] + 5 * c->dither_error[0][i + 1] + 3 * c->dither_error[0][i + 2]) >> 4;
            G += (7 * err[1] + 1 * c->dither_error[1][i] + 5 * c->dither_error[1][i + 1] + 3 * c->dither_error[1][i + 2]) >> 4;
            B += (7 * err[2] + 1 * c->dither_error[2][i] + 5 * c->dither_error[2][i + 1] + 3 * c->dither_error[2][i + 2]) >> 4;
            c->dither_error[0][i] = err[0]; c->dither_error[1][i] = err[1]; c->dither_error[2][i] = err[2];
            r = R >> (isrgb8 ? 5 : 7); g = G >> (isrgb8 ? 5 : 6); b = B >> (isrgb8 ? 6 : 7);
            r = ORMAX(0, ORMIN(r, isrgb8 ? 7 : 1)); g = ORMAX(0, ORMIN(g, isrgb8 ? 7 : 3)); b = ORMAX(0, ORMIN(b, isrgb8 ? 3 : 1));
            err[0] = R - r * (isrgb8 ? 36 : 255); err[1] = G - g * (isrgb8 ? 36 : 85); err[2] = B - b * (isrgb8 ? 85 : 255);
            break;
        case 4: /* SWS_DITHER_A_DITHER */
            if (isrgb8) {
                r = ((R >> 19) + A_DITHER(i, y) - 96) >> 8; g = ((G >> 19) + A_DITHER(i + 17, y) - 96) >> 8; b = ((B >> 20) + A_DITHER(i + 17 * 2, y) - 96) >> 8;
                r = clip_uintp2(r, 3); g = clip_uintp2(g, 3); b = clip_uintp2(b, 2);
            } else {
                r = ((R >> 21) + A_DITHER(i, y) - 256) >> 8; g = ((G >> 19) + A_DITHER(i + 17, y) - 256) >> 8; b = ((B >> 21) + A_DITHER(i + 17 * 2, y) - 256) >> 8;
                r = clip_uintp2(r, 1); g = clip_uintp2(g, 2); b = clip_uintp2(b, 1);
            }
            break;
        case 5: /* SWS_DITHER_X_DITHER */
            if (isrgb8) {
                r = ((R >> 19) + X_DITHER(i, y) - 96) >> 8; g = ((G >> 19) + X_DITHER(i + 17, y) - 96) >> 8; b = ((B >> 20) + X_DITHER(i + 17 * 2, y) - 96) >> 8;
                r = clip_uintp2(r, 3); g = clip_uintp2(g, 3); b = clip_uintp2(b, 2);
            } else {
                r = ((R >> 21) + X_DITHER(i, y) - 256) >> 8; g = ((G >> 19) + X_DITHER(i + 17, y) - 256) >> 8; b = ((B >> 21) + X_DITHER(i + 17 * 2, y) - 256) >> 8;
                r = clip_uintp2(r, 1); g = clip_uintp2(g, 2); b = clip_uintp2(b, 1);
            }
            break;
        }
#undef A_DITHER
#undef X_DITHER
        if (d == ORF_BGR4_BYTE) dest[0] = (uint8_t)(r + 2 * g + 8 * b);
        else if (d == ORF_RGB4_BYTE) dest[0] = (uint8_t)(b + 2 * g + 8 * r);
        else if (d == ORF_BGR8) dest[0] = (uint8_t)(r + 8 * g + 64 * b);
        else dest[0] = (uint8_t)(b + 4 * g + 32 * r);
        return;
    }
    R >>= 22; G >>= 22; B >>= 22;
    switch (d) {
    case ORF_ARGB: dest[0] = hasAlpha ? (uint8_t)A : 255; dest[1] = (uint8_t)R; dest[2] = (uint8_t)G; dest[3] = (uint8_t)B; break;
    case ORF_RGB24: dest[0] = (uint8_t)R; dest[1] = (uint8_t)G; dest[2] = (uint8_t)B; break;
    case ORF_RGBA: dest[0] = (uint8_t)R; dest[1] = (uint8_t)G; dest[2] = (uint8_t)B; dest[3] = hasAlpha ? (uint8_t)A : 255; break;
    case ORF_ABGR: dest[0] = hasAlpha ? (uint8_t)A : 255; dest[1] = (uint8_t)B; dest[2] = (uint8_t)G; dest[3] = (uint8_t)R; break;
    case ORF_BGR24: dest[0] = (uint8_t)B; dest[1] = (uint8_t)G; dest[2] = (uint8_t)R; break;
    case ORF_BGRA: dest[0] = (uint8_t)B; dest[1] = (uint8_t)G; dest[2] = (uint8_t)R; dest[3] = hasAlpha ? (uint8_t)A : 255; break;
    }
}

typedef struct {
    int32_t *lum, *chrU, *chrV; /* h-scaled lines, [ring][dstW] and [ring][chrDstW]: line r of a plane lives in slot r & mask -- like the reference's own line rings
                                 * (slice.c), only addressed by the absolute line number; the ring is at least twice what ring_sizes() says is ever alive at once,
                                 * so the working set of a 4K frame stays in the cache instead of streaming through 33 MB planes */
    int32_t *alp;               /* h-scaled alpha lines when needAlpha (hscale.c:137, vscale.c:59-71), the luma ring's size */
    int lmask, cmask;           /* ring sizes - 1 (powers of two) of the luma / alpha and the chroma lines */
} Planes;

/* packed_vscale (vscale.c:109-171) + yuv2rgb_{X,2,1}_c_template (output.c:1788-1939)
 * + yuv2rgb_full_{X,2,1}_c_template (output.c:2163-2312) */
static void write_packed_rgb_line(OrSws *c, const Planes *P, uint8_t *dest, int y)
{
    const int dstW = c->o.dst_w, lw = dstW, cw = c->chrDstW;
    const int srcH = c->o.src_h, chrSrcH = c->chrSrcH;
    const int chrY = y >> c->chrDstVSub;
    const int lfs = c->vLumFilterSize, cfs = c->vChrFilterSize;
    const int16_t *lf = c->vLumFilter + y * lfs, *cf = c->vChrFilter + chrY * cfs;
    const int firstLum = ORMAX(1 - lfs, c->vLumFilterPos[y]);
    const int firstChr = ORMAX(1 - cfs, c->vChrFilterPos[chrY]);
    const int full = !!(c->o.flags & OR_SWS_FULL_CHR_H_INT);
    const int step = isRGB8class(c->o.dst_format) ? 1 : c->lut_elem == 4 || c->dstFormatBpp == 32 ? 4 : 3;
    int err[4] = { 0, 0, 0, 0 };   /* the running error of the row (yuv2rgb_full_{X,2,1}_c_template: "int err[4] = {0}") */
    int i, j;
    /* the lines of the row's two vertical windows, looked up once (the pixel loops below index them by tap) */
    const int32_t *lrow[lfs > 0 ? lfs : 1], *urow[cfs > 0 ? cfs : 1], *vrow[cfs > 0 ? cfs : 1], *arow[lfs > 0 ? lfs : 1];
    for (j = 0; j < lfs; j++) {
        lrow[j] = P->lum + RING(P->lmask, ORMIN(firstLum + j, srcH - 1)) * lw;
        arow[j] = P->alp ? P->alp + RING(P->lmask, ORMIN(firstLum + j, srcH - 1)) * lw : NULL;
    }
    for (j = 0; j < cfs; j++) {
        urow[j] = P->chrU + RING(P->cmask, ORMIN(firstChr + j, chrSrcH - 1)) * cw;
        vrow[j] = P->chrV + RING(P->cmask, ORMIN(firstChr + j, chrSrcH - 1)) * cw;
    }
#define L(j) lrow[j]
#define CU(j) urow[j]
#define CV(j) vrow[j]
#define AL(j) arow[j]
    const int hasAlpha = c->needAlpha;
    int mode; /* 1: packed1 (uvalpha in ua), 2: packed2, 0: X */
    int ua = 0, ya = 0;
    if (lfs == 1 && cfs == 1) { mode = 1; ua = 0; }
    else if (lfs == 1 && cfs == 2 && (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U) { mode = 1; ua = (uint16_t)cf[1]; }
    else if (lfs == 2 && cfs == 2 && (uint16_t)lf[1] + (uint16_t)lf[0] == 4096 && (uint16_t)lf[1] <= 4096U &&
             (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U) { mode = 2; ya = (uint16_t)lf[1]; ua = (uint16_t)cf[1]; }
    else mode = 0;

    if (!full && mode == 0 && !hasAlpha && !(dstW & 1) && ((c->lut_elem == 1 && c->dstFormatBpp == 24) || (c->lut_elem == 4 && c->dstFormatBpp == 32))) {
        /* the common case on its own (yuv2rgb24_X_c / yuv2rgbx32_X_c: the X form, no alpha, whole pixel pairs): the arithmetic of the general loop below
         * with the per-pixel mode / format tests taken out of the loop -- what the reference gets from instantiating its template per format */
        const int d = c->o.dst_format;
        const int swap = !(d == ORF_RGB24);       /* 24 bpp: byte order R G B for rgb24, B G R for bgr24 */
        for (i = 0; i < (dstW >> 1); i++) {
            int Y1 = 1 << 18, Y2 = 1 << 18, U = 1 << 18, V = 1 << 18;
            for (j = 0; j < lfs; j++) { Y1 = (int)((unsigned)Y1 + lrow[j][2 * i] * (unsigned)lf[j]); Y2 = (int)((unsigned)Y2 + lrow[j][2 * i + 1] * (unsigned)lf[j]); }
            for (j = 0; j < cfs; j++) { U = (int)((unsigned)U + urow[j][i] * (unsigned)cf[j]); V = (int)((unsigned)V + vrow[j][i] * (unsigned)cf[j]); }
            Y1 >>= 19; Y2 >>= 19; U >>= 19; V >>= 19;
            const int r = c->table_rV[V + HEADROOM], g = c->table_gU[U + HEADROOM] + c->table_gV[V + HEADROOM], bb = c->table_bU[U + HEADROOM];
            if (c->lut_elem == 4) {
                const uint32_t *T = (const uint32_t *)c->yuvTable;
                const uint32_t v1 = T[r + Y1] + T[g + Y1] + T[bb + Y1], v2 = T[r + Y2] + T[g + Y2] + T[bb + Y2];
                memcpy(dest + 8 * i, &v1, 4); memcpy(dest + 8 * i + 4, &v2, 4);
            } else {
                const uint8_t *T = c->yuvTable;
                uint8_t *q = dest + 6 * i;
                const int rb = swap ? bb : r, br = swap ? r : bb;
                q[0] = T[rb + Y1]; q[1] = T[g + Y1]; q[2] = T[br + Y1];
                q[3] = T[rb + Y2]; q[4] = T[g + Y2]; q[5] = T[br + Y2];
            }
        }
    } else
    if (!full) {
        /* odd widths reach the pair writers only for the 16 bpp formats (no full-chroma writer): the second pixel of the last pair lies
         * beyond the picture; the reference computes it from the line buffers' fill value and stores it into the row padding, the oracle
         * computes it the same way and does not store it */
#define LB(j) (2 * i + 1 < dstW ? L(j)[2 * i + 1] : (1 << 14))
        for (i = 0; i < ((dstW + 1) >> 1); i++) {
            int Y1, Y2, U, V;
            if (mode == 0) {
                Y1 = Y2 = U = V = 1 << 18;
                for (j = 0; j < lfs; j++) { Y1 = (int)((unsigned)Y1 + L(j)[2 * i] * (unsigned)lf[j]); Y2 = (int)((unsigned)Y2 + LB(j) * (unsigned)lf[j]); }
                for (j = 0; j < cfs; j++) { U = (int)((unsigned)U + CU(j)[i] * (unsigned)cf[j]); V = (int)((unsigned)V + CV(j)[i] * (unsigned)cf[j]); }
                Y1 >>= 19; Y2 >>= 19; U >>= 19; V >>= 19;
            } else if (mode == 2) {
                int ya1 = 4096 - ya, ua1 = 4096 - ua;
                Y1 = (L(0)[2 * i] * ya1 + L(1)[2 * i] * ya) >> 19;
                Y2 = (LB(0) * ya1 + LB(1) * ya) >> 19;
                U = (CU(0)[i] * ua1 + CU(1)[i] * ua) >> 19;
                V = (CV(0)[i] * ua1 + CV(1)[i] * ua) >> 19;
            } else {
                Y1 = (L(0)[2 * i] + 64) >> 7;
                Y2 = (LB(0) + 64) >> 7;
                if (ua == 0) { U = (CU(0)[i] + 64) >> 7; V = (CV(0)[i] + 64) >> 7; }
                else {
                    int ua1 = 4096 - ua;
                    U = (CU(0)[i] * ua1 + CU(1)[i] * ua + (128 << 11)) >> 19;
                    V = (CV(0)[i] * ua1 + CV(1)[i] * ua + (128 << 11)) >> 19;
                }
            }
            {   /* alpha of the pixel pair: output.c:1818-1830 (X), :1870-1875 (2), :1904-1908 / :1929-1933 (1) */
                int A1 = 0, A2 = 0;
                if (hasAlpha) {
                    if (mode == 0) {
                        A1 = A2 = 1 << 18;
                        for (j = 0; j < lfs; j++) { A1 = (int)((unsigned)A1 + AL(j)[2 * i] * (unsigned)lf[j]); A2 = (int)((unsigned)A2 + AL(j)[2 * i + 1] * (unsigned)lf[j]); }
                        A1 >>= 19; A2 >>= 19;
                        if ((A1 | A2) & 0x100) { A1 = clip_u8(A1); A2 = clip_u8(A2); }
                    } else if (mode == 2) {
                        A1 = clip_u8((AL(0)[2 * i] * (4096 - ya) + AL(1)[2 * i] * ya) >> 19);
                        A2 = clip_u8((AL(0)[2 * i + 1] * (4096 - ya) + AL(1)[2 * i + 1] * ya) >> 19);
                    } else if (ua == 0) {
                        A1 = clip_u8((AL(0)[2 * i] * 255 + 16384) >> 15);
                        A2 = clip_u8((AL(0)[2 * i + 1] * 255 + 16384) >> 15);
                    } else {
                        A1 = clip_u8((AL(0)[2 * i] + 64) >> 7);
                        A2 = clip_u8((AL(0)[2 * i + 1] + 64) >> 7);
                    }
                }
                rgb_write2(c, dest, i, y, Y1, Y2, U, V, hasAlpha, (unsigned)A1, (unsigned)A2, 2 * i + 1 < dstW);
            }
        }
    } else {
        for (i = 0; i < dstW; i++) {
            int Y, U, V;
            if (mode == 0) { /* :2163-2212 */
                Y = 1 << 9; U = (1 << 9) - (128 << 19); V = (1 << 9) - (128 << 19);
                for (j = 0; j < lfs; j++) Y = (int)((unsigned)Y + L(j)[i] * (unsigned)lf[j]);
                for (j = 0; j < cfs; j++) { U = (int)((unsigned)U + CU(j)[i] * (unsigned)cf[j]); V = (int)((unsigned)V + CV(j)[i] * (unsigned)cf[j]); }
                Y >>= 10; U >>= 10; V >>= 10;
            } else if (mode == 2) { /* :2214-2260 */
                int ya1 = 4096 - ya, ua1 = 4096 - ua;
                Y = (L(0)[i] * ya1 + L(1)[i] * ya) >> 10;
                U = (CU(0)[i] * ua1 + CU(1)[i] * ua - (128 << 19)) >> 10;
                V = (CV(0)[i] * ua1 + CV(1)[i] * ua - (128 << 19)) >> 10;
            } else { /* :2262-2312 */
                Y = L(0)[i] * 4;
                if (ua == 0) { U = (CU(0)[i] - (128 << 7)) * 4; V = (CV(0)[i] - (128 << 7)) * 4; }
                else {
                    int ua1 = 4096 - ua;
                    U = (CU(0)[i] * ua1 + CU(1)[i] * ua - (128 << 19)) >> 10;
                    V = (CV(0)[i] * ua1 + CV(1)[i] * ua - (128 << 19)) >> 10;
                }
            }
            {   /* output.c:2193-2201 (X), :2241-2245 (2), :2278-2283 / :2298-2303 (1) */
                int A = 0;
                if (hasAlpha) {
                    if (mode == 0) {
                        A = 1 << 18;
                        for (j = 0; j < lfs; j++) A = (int)((unsigned)A + AL(j)[i] * (unsigned)lf[j]);
                        A >>= 19;
                    } else if (mode == 2) A = (AL(0)[i] * (4096 - ya) + AL(1)[i] * ya + (1 << 18)) >> 19;
                    else A = (AL(0)[i] + 64) >> 7;
                    if (A & 0x100) A = clip_u8(A);
                }
                rgb_write_full(c, dest + step * i, i, y, Y, U, V, hasAlpha, A, err);
            }
        }
        if (c->dither_error[0]) {   /* output.c:2204-2206 / :2249-2251 / :2306-2308: the last pixel's error closes the line */
            c->dither_error[0][i] = err[0]; c->dither_error[1][i] = err[1]; c->dither_error[2][i] = err[2];
        }
    }
#undef L
#undef CU
#undef CV
#undef AL
}

/* packed_vscale (vscale.c:109-171) + yuv2rgba64_{X,2,1}_c_template and yuv2rgba64_full_{X,2,1}_c_template (output.c:1115-1560)
 * for rgb48le / bgr48le / rgba64le / bgra64le: 19-bit intermediates, 32-bit wrap-around arithmetic with the reference's
 * exact signedness of every shift (one of the full_1 shifts is logical, see the SUINT lines at :1538-1539). */
static void write_packed_rgb16_line(const OrSws *c, const Planes *P, uint8_t *dest8, int y)
{
    const int dstW = c->o.dst_w, lw = dstW, cw = c->chrDstW;
    const int srcH = c->o.src_h, chrSrcH = c->chrSrcH;
    const int chrY = y >> c->chrDstVSub;
    const int lfs = c->vLumFilterSize, cfs = c->vChrFilterSize;
    const int16_t *lf = c->vLumFilter + y * lfs, *cf = c->vChrFilter + chrY * cfs;
    const int firstLum = ORMAX(1 - lfs, c->vLumFilterPos[y]);
    const int firstChr = ORMAX(1 - cfs, c->vChrFilterPos[chrY]);
    const int full = !!(c->o.flags & OR_SWS_FULL_CHR_H_INT);
    const Desc *dd = desc_get(c->o.dst_format);
    const int st = dd->c[0].step / 2, ro = dd->c[0].offset / 2, go = dd->c[1].offset / 2, bo = dd->c[2].offset / 2;
    const int hasAlpha = c->needAlpha;
    uint16_t *dest = (uint16_t *)dest8;
    int j, mode;
    unsigned ua = 0, ya = 0;
#define L(j) (P->lum + RING(P->lmask, ORMIN(firstLum + (j), srcH - 1)) * lw)
#define CU(j) (P->chrU + RING(P->cmask, ORMIN(firstChr + (j), chrSrcH - 1)) * cw)
#define CV(j) (P->chrV + RING(P->cmask, ORMIN(firstChr + (j), chrSrcH - 1)) * cw)
#define AL(j) (P->alp + RING(P->lmask, ORMIN(firstLum + (j), srcH - 1)) * lw)
    if (lfs == 1 && cfs == 1) mode = 1;
    else if (lfs == 1 && cfs == 2 && (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U) { mode = 1; ua = (uint16_t)cf[1]; }
    else if (lfs == 2 && cfs == 2 && (uint16_t)lf[1] + (uint16_t)lf[0] == 4096 && (uint16_t)lf[1] <= 4096U &&
             (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U) { mode = 2; ya = (uint16_t)lf[1]; ua = (uint16_t)cf[1]; }
    else mode = 0;
    const int npx = full ? dstW : 2 * ((dstW + 1) >> 1);      /* the pair writers also emit the pixel after an odd width */
    for (int x = 0; x < npx; x++) {
        const int ci = full ? x : x >> 1;                       /* chroma sample of this pixel */
        unsigned Y, U, V;
        int A = 0xffff << 14;
        if (mode == 0) {
            Y = (unsigned)-0x40000000; U = (unsigned)-(128 << 23); V = (unsigned)-(128 << 23);
            for (j = 0; j < lfs; j++) Y += L(j)[x] * (unsigned)lf[j];
            for (j = 0; j < cfs; j++) { U += CU(j)[ci] * (unsigned)cf[j]; V += CV(j)[ci] * (unsigned)cf[j]; }
            if (hasAlpha) {
                A = -0x40000000;
                for (j = 0; j < lfs; j++) A = (int)((unsigned)A + AL(j)[x] * (unsigned)lf[j]);
                A >>= 1; A += 0x20002000;
            }
            Y = (unsigned)((int)Y >> 14); Y += 0x10000;
            U = (unsigned)((int)U >> 14); V = (unsigned)((int)V >> 14);
        } else if (mode == 2) {
            const unsigned ya1 = 4096 - ya, ua1 = 4096 - ua;
            Y = (unsigned)((int)(L(0)[x] * ya1 + L(1)[x] * ya) >> 14);
            U = (unsigned)((int)(CU(0)[ci] * ua1 + CU(1)[ci] * ua - (128 << 23)) >> 14);
            V = (unsigned)((int)(CV(0)[ci] * ua1 + CV(1)[ci] * ua - (128 << 23)) >> 14);
            if (hasAlpha) { A = (int)(AL(0)[x] * ya1 + AL(1)[x] * ya) >> 1; A += 1 << 13; }
        } else {
            Y = (unsigned)(L(0)[x] >> 2);
            if (ua == 0) { U = (unsigned)((CU(0)[ci] - (128 << 11)) >> 2); V = (unsigned)((CV(0)[ci] - (128 << 11)) >> 2); }
            else {
                const unsigned ua1 = 4096 - ua;
                const unsigned tu = CU(0)[ci] * ua1 + CU(1)[ci] * ua - (128 << 23), tv = CV(0)[ci] * ua1 + CV(1)[ci] * ua - (128 << 23);
                if (full) { U = tu >> 14; V = tv >> 14; }           /* :1538-1539: SUINT expression, LOGICAL shift */
                else { U = (unsigned)((int)tu >> 14); V = (unsigned)((int)tv >> 14); }   /* :1318-1319: (int) cast, arithmetic */
            }
            if (hasAlpha) { A = (int)((unsigned)AL(0)[x] * (1u << 11)); A += 1 << 13; }
        }
        Y -= (unsigned)c->yuv2rgb_y_offset;
        Y *= (unsigned)c->yuv2rgb_y_coeff;
        Y += (unsigned)((1 << 13) - (1 << 29));
        {
            const unsigned R = V * (unsigned)c->yuv2rgb_v2r, G = V * (unsigned)c->yuv2rgb_v2g + U * (unsigned)c->yuv2rgb_u2g,
                           B = U * (unsigned)c->yuv2rgb_u2b;
            uint16_t *d = dest + st * x;
            if (x >= dstW) continue;      /* the extra pixel of an odd width lands in the row padding: not restated */
            d[ro] = (uint16_t)clip_uintp2(((int)(R + Y) >> 14) + (1 << 15), 16);
            d[go] = (uint16_t)clip_uintp2(((int)(G + Y) >> 14) + (1 << 15), 16);
            d[bo] = (uint16_t)clip_uintp2(((int)(B + Y) >> 14) + (1 << 15), 16);
            if (st == 4) d[3] = (uint16_t)(clip_uintp2(A, 30) >> 14);
        }
    }
#undef L
#undef CU
#undef CV
#undef AL
}

/* packed_vscale (vscale.c:109-171) + yuv2422_{X,2,1}_c_template (output.c:883-1000) for yuyv422 / yvyu422 / uyvy422 */
/* packed_vscale + yuv2ya8_{1,2,X}_c (output.c:2613-2705), yuv2ya16_{X,2,1}_c_template (:1016-1113): gray + alpha pairs */
static void write_ya_line(const OrSws *c, const Planes *P, uint8_t *dest, int y)
{
    const int dstW = c->o.dst_w, lw = dstW, srcH = c->o.src_h;
    const int chrY = y >> c->chrDstVSub;
    const int lfs = c->vLumFilterSize, cfs = c->vChrFilterSize;
    const int16_t *lf = c->vLumFilter + y * lfs, *cf = c->vChrFilter + chrY * cfs;
    const int firstLum = ORMAX(1 - lfs, c->vLumFilterPos[y]);
    const int hasAlpha = c->needAlpha, wide = c->o.dst_format == ORF_YA16LE;
    int i, j, mode, ya = 0;
#define L(j) (P->lum + RING(P->lmask, ORMIN(firstLum + (j), srcH - 1)) * lw)
#define AL(j) (P->alp + RING(P->lmask, ORMIN(firstLum + (j), srcH - 1)) * lw)
    if (lfs == 1 && cfs == 1) mode = 1;
    else if (lfs == 1 && cfs == 2 && (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U) mode = 1;
    else if (lfs == 2 && cfs == 2 && (uint16_t)lf[1] + (uint16_t)lf[0] == 4096 && (uint16_t)lf[1] <= 4096U &&
             (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U) { mode = 2; ya = (uint16_t)lf[1]; }
    else mode = 0;
    for (i = 0; i < dstW; i++) {
        int Y, A = 0;
        if (!wide) {
            if (mode == 1) {
                Y = clip_u8((L(0)[i] + 64) >> 7);
                if (hasAlpha) { A = (AL(0)[i] + 64) >> 7; if (A & 0x100) A = clip_u8(A); }
            } else if (mode == 2) {
                Y = clip_u8((L(0)[i] * (4096 - ya) + L(1)[i] * ya) >> 19);
                if (hasAlpha) A = clip_u8((AL(0)[i] * (4096 - ya) + AL(1)[i] * ya) >> 19);
            } else {
                Y = 1 << 18; A = 1 << 18;
                for (j = 0; j < lfs; j++) Y = (int)((unsigned)Y + L(j)[i] * (unsigned)lf[j]);
                Y >>= 19; if (Y & 0x100) Y = clip_u8(Y);
                if (hasAlpha) { for (j = 0; j < lfs; j++) A = (int)((unsigned)A + AL(j)[i] * (unsigned)lf[j]); A >>= 19; if (A & 0x100) A = clip_u8(A); }
            }
            dest[2 * i] = (uint8_t)Y; dest[2 * i + 1] = hasAlpha ? (uint8_t)A : 255;
        } else {
            uint16_t v[2];
            if (mode == 1) {
                Y = clip_u16(L(0)[i] >> 3);
                if (hasAlpha) { A = AL(0)[i] >> 3; if (A & 0x100) A = clip_u16(A); }   /* (sic: the 8-bit test, :1106-1107) */
                else A = 65535;
            } else if (mode == 2) {
                const unsigned ya1 = 4096 - ya;
                Y = clip_u16((int)((unsigned)L(0)[i] * ya1 + (unsigned)L(1)[i] * (unsigned)ya) >> 15);
                A = hasAlpha ? clip_u16((int)((unsigned)AL(0)[i] * ya1 + (unsigned)AL(1)[i] * (unsigned)ya) >> 15) : 65535;
            } else {
                Y = -0x40000000; A = 0xffff;
                for (j = 0; j < lfs; j++) Y = (int)((unsigned)Y + L(j)[i] * (unsigned)lf[j]);
                Y >>= 15; Y += (1 << 3) + 0x8000; Y = clip_u16(Y);
                if (hasAlpha) {
                    A = -0x40000000 + (1 << 14);
                    for (j = 0; j < lfs; j++) A = (int)((unsigned)A + AL(j)[i] * (unsigned)lf[j]);
                    A >>= 15; A += 0x8000; A = clip_u16(A);
                }
            }
            v[0] = (uint16_t)Y; v[1] = (uint16_t)A;
            memcpy(dest + 4 * i, v, 4);
        }
    }
#undef L
#undef AL
}

/* ff_dither_8x8_220 (output.c:84-95, the `#if 1` variant), nine rows */
static const uint8_t dither_8x8_220[9][8] = {
    { 117,  62, 158, 103, 113,  58, 155, 100 }, {  34, 199,  21, 186,  31, 196,  17, 182 }, { 144,  89, 131,  76, 141,  86, 127,  72 },
    {   0, 165,  41, 206,  10, 175,  52, 217 }, { 110,  55, 151,  96, 120,  65, 162, 107 }, {  28, 193,  14, 179,  38, 203,  24, 189 },
    { 138,  83, 124,  69, 148,  93, 134,  79 }, {   7, 172,  48, 213,   3, 168,  45, 210 }, { 117,  62, 158, 103, 113,  58, 155, 100 },
};
/* packed_vscale + yuv2mono_{X,2,1}_c_template (output.c:654-860), ordered dither only.  The X form shifts bits through one running
 * accumulator (a trailing partial byte holds the last 8 bits seen); the 2 and 1 forms build whole bytes from 8 luma entries, the ones
 * past dstW being the line buffers' fill_ones() value (slice.c:190-208) */
static void write_mono_line(OrSws *c, const Planes *P, uint8_t *dest, int y)
{
    const int dstW = c->o.dst_w, lw = dstW, srcH = c->o.src_h;
    const int chrY = y >> c->chrDstVSub;
    const int lfs = c->vLumFilterSize, cfs = c->vChrFilterSize;
    const int16_t *lf = c->vLumFilter + y * lfs, *cf = c->vChrFilter + chrY * cfs;
    const int firstLum = ORMAX(1 - lfs, c->vLumFilterPos[y]);
    const uint8_t *d128 = dither_8x8_220[y & 7];
    const int white = c->o.dst_format == ORF_MONOWHITE;
    int i, j, mode, ya = 0;
#define L(j) (P->lum + RING(P->lmask, ORMIN(firstLum + (j), srcH - 1)) * lw)
#define LBM(j, x) ((x) < dstW ? L(j)[x] : (1 << 14))
    if (lfs == 1 && cfs == 1) mode = 1;
    else if (lfs == 1 && cfs == 2 && (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U) mode = 1;
    else if (lfs == 2 && cfs == 2 && (uint16_t)lf[1] + (uint16_t)lf[0] == 4096 && (uint16_t)lf[1] <= 4096U &&
             (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U) { mode = 2; ya = (uint16_t)lf[1]; }
    else mode = 0;
    if (c->o.dither == 3) {   /* SWS_DITHER_ED (output.c:690-700, :734-753, :792-811): Floyd-Steinberg over the pixel pairs with threshold 128 and step 220;
                               * c->dither_error[0][k] holds the error of pixel k - 1 of the row above.  The 2 / 1 forms never store a trailing partial byte */
        unsigned acc = 0;
        int err = 0;
        int *de = c->dither_error[0];
        for (i = 0; i < dstW; i += 2) {
            int Y1, Y2;
            if (mode == 0) {
                Y1 = Y2 = 1 << 18;
                for (j = 0; j < lfs; j++) { Y1 = (int)((unsigned)Y1 + L(j)[i] * (unsigned)lf[j]); Y2 = (int)((unsigned)Y2 + LBM(j, i + 1) * (unsigned)lf[j]); }
                Y1 >>= 19; Y2 >>= 19;
                if ((Y1 | Y2) & 0x100) { Y1 = clip_u8(Y1); Y2 = clip_u8(Y2); }
            } else if (mode == 2) {
                Y1 = (LBM(0, i) * (4096 - ya) + LBM(1, i) * ya) >> 19; Y2 = (LBM(0, i + 1) * (4096 - ya) + LBM(1, i + 1) * ya) >> 19;
            } else { Y1 = (LBM(0, i) + 64) >> 7; Y2 = (LBM(0, i + 1) + 64) >> 7; }
            Y1 += (7 * err + 1 * de[i] + 5 * de[i + 1] + 3 * de[i + 2] + 8 - 256) >> 4;
            de[i] = err;
            acc = 2 * acc + (Y1 >= 128);
            Y1 -= 220 * (int)(acc & 1);
            err = Y2 + ((7 * Y1 + 1 * de[i + 1] + 5 * de[i + 2] + 3 * de[i + 3] + 8 - 256) >> 4);
            de[i + 1] = Y1;
            acc = 2 * acc + (err >= 128);
            err -= 220 * (int)(acc & 1);
            if ((i & 7) == 6) *dest++ = (uint8_t)(white ? ~acc : acc);
        }
        de[i] = err;
        if (mode == 0 && (i & 6)) *dest = (uint8_t)(white ? ~acc : acc);
    } else if (mode == 0) {
        unsigned acc = 0;
        for (i = 0; i < dstW; i += 2) {
            int Y1 = 1 << 18, Y2 = 1 << 18;
            for (j = 0; j < lfs; j++) { Y1 = (int)((unsigned)Y1 + L(j)[i] * (unsigned)lf[j]); Y2 = (int)((unsigned)Y2 + LBM(j, i + 1) * (unsigned)lf[j]); }
            Y1 >>= 19; Y2 >>= 19;
            if ((Y1 | Y2) & 0x100) { Y1 = clip_u8(Y1); Y2 = clip_u8(Y2); }
            acc = (acc << 1) | (Y1 + d128[i & 7] >= 234);
            acc = (acc << 1) | (Y2 + d128[(i + 1) & 7] >= 234);
            if ((i & 7) == 6) *dest++ = (uint8_t)(white ? ~acc : acc);
        }
        if (i & 6) *dest = (uint8_t)(white ? ~acc : acc);
    } else {
        for (i = 0; i < dstW; i += 8) {
            unsigned acc = 0;
            for (int k = 0; k < 8; k++) {
                const int Y = mode == 2 ? (LBM(0, i + k) * (4096 - ya) + LBM(1, i + k) * ya) >> 19 : (LBM(0, i + k) + 64) >> 7;
                acc = (acc << 1) | (Y + d128[k] >= 234);
            }
            *dest++ = (uint8_t)(white ? ~acc : acc);
        }
    }
#undef L
#undef LBM
}

static void write_packed422_line(const OrSws *c, const Planes *P, uint8_t *dest, int y)
{
    const int dstW = c->o.dst_w, lw = dstW, cw = c->chrDstW;
    const int srcH = c->o.src_h, chrSrcH = c->chrSrcH;
    const int chrY = y >> c->chrDstVSub;
    const int lfs = c->vLumFilterSize, cfs = c->vChrFilterSize;
    const int16_t *lf = c->vLumFilter + y * lfs, *cf = c->vChrFilter + chrY * cfs;
    const int firstLum = ORMAX(1 - lfs, c->vLumFilterPos[y]);
    const int firstChr = ORMAX(1 - cfs, c->vChrFilterPos[chrY]);
    const Desc *dd = desc_get(c->o.dst_format);
    int i, j, mode, ua = 0, ya = 0;
#define L(j) (P->lum + RING(P->lmask, ORMIN(firstLum + (j), srcH - 1)) * lw)
#define CU(j) (P->chrU + RING(P->cmask, ORMIN(firstChr + (j), chrSrcH - 1)) * cw)
#define CV(j) (P->chrV + RING(P->cmask, ORMIN(firstChr + (j), chrSrcH - 1)) * cw)
    /* the second pixel of the last pair of an odd-width picture: the line buffers are pre-filled with 1 << 14 (fill_ones, slice.c:190-208)
     * and the horizontal scaler writes dstW entries only */
#define L2(j) (2 * i + 1 < dstW ? L(j)[2 * i + 1] : (1 << 14))
    if (lfs == 1 && cfs == 1) mode = 1;
    else if (lfs == 1 && cfs == 2 && (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U) { mode = 1; ua = (uint16_t)cf[1]; }
    else if (lfs == 2 && cfs == 2 && (uint16_t)lf[1] + (uint16_t)lf[0] == 4096 && (uint16_t)lf[1] <= 4096U &&
             (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U) { mode = 2; ya = (uint16_t)lf[1]; ua = (uint16_t)cf[1]; }
    else mode = 0;
    for (i = 0; i < ((dstW + 1) >> 1); i++) {
        int Y1, Y2, U, V;
        if (mode == 0) {
            Y1 = Y2 = U = V = 1 << 18;
            for (j = 0; j < lfs; j++) { Y1 = (int)((unsigned)Y1 + L(j)[2 * i] * (unsigned)lf[j]); Y2 = (int)((unsigned)Y2 + L2(j) * (unsigned)lf[j]); }
            for (j = 0; j < cfs; j++) { U = (int)((unsigned)U + CU(j)[i] * (unsigned)cf[j]); V = (int)((unsigned)V + CV(j)[i] * (unsigned)cf[j]); }
            Y1 >>= 19; Y2 >>= 19; U >>= 19; V >>= 19;
        } else if (mode == 2) {
            Y1 = (L(0)[2 * i] * (4096 - ya) + L(1)[2 * i] * ya) >> 19;
            Y2 = (L2(0) * (4096 - ya) + L2(1) * ya) >> 19;
            U = (CU(0)[i] * (4096 - ua) + CU(1)[i] * ua) >> 19;
            V = (CV(0)[i] * (4096 - ua) + CV(1)[i] * ua) >> 19;
        } else {
            Y1 = (L(0)[2 * i] + 64) >> 7; Y2 = (L2(0) + 64) >> 7;
            if (ua < 2048) { U = (CU(0)[i] + 64) >> 7; V = (CV(0)[i] + 64) >> 7; }
            else { U = (CU(0)[i] + CU(1)[i] + 128) >> 8; V = (CV(0)[i] + CV(1)[i] + 128) >> 8; }
        }
        if ((Y1 | Y2 | U | V) & 0x100) { Y1 = clip_u8(Y1); Y2 = clip_u8(Y2); U = clip_u8(U); V = clip_u8(V); }
        dest[4 * i + dd->c[0].offset] = (uint8_t)Y1; dest[4 * i + dd->c[0].offset + 2] = (uint8_t)Y2;   /* output_pixels :864-881 */
        dest[4 * i + dd->c[1].offset] = (uint8_t)U; dest[4 * i + dd->c[2].offset] = (uint8_t)V;
    }
#undef L
#undef CU
#undef CV
}



/* packed_vscale, X writers only: yuv2y210le/y212le_X_c (output.c:3088-3127), yuv2y216le_X_c (:3129-3169), yuv2xv30le/v30xle_X_c (:2784-2837),
 * yuv2xv36le_X_c (:2839-2866), yuv2ayuv64le / yuv2xv48le_X_c (:2712-2776).  15-bit lines for <= 14-bit targets, 19-bit lines for 16-bit ones. */
static void write_packedhi_line(const OrSws *c, const Planes *P, uint8_t *dest, int y)
{
    const int df = c->o.dst_format;
    const int dstW = c->o.dst_w, lw = dstW, cw = c->chrDstW;
    const int srcH = c->o.src_h, chrSrcH = c->chrSrcH;
    const int chrY = y >> c->chrDstVSub;
    const int lfs = c->vLumFilterSize, cfs = c->vChrFilterSize;
    const int16_t *lf = c->vLumFilter + y * lfs, *cf = c->vChrFilter + chrY * cfs;
    const int firstLum = ORMAX(1 - lfs, c->vLumFilterPos[y]);
    const int firstChr = ORMAX(1 - cfs, c->vChrFilterPos[chrY]);
    const Desc *dd = desc_get(df);
    const int bits = dd->c[0].depth, sub = dd->lw;          /* sub: 4:2:2 (y21x) */
    const int units = sub ? (dstW + 1) >> 1 : dstW;
    int i, j;
#define L(j) (P->lum + RING(P->lmask, ORMIN(firstLum + (j), srcH - 1)) * lw)
#define CU(j) (P->chrU + RING(P->cmask, ORMIN(firstChr + (j), chrSrcH - 1)) * cw)
#define CV(j) (P->chrV + RING(P->cmask, ORMIN(firstChr + (j), chrSrcH - 1)) * cw)
#define AL(j) (P->alp + RING(P->lmask, ORMIN(firstLum + (j), srcH - 1)) * lw)
    for (i = 0; i < units; i++) {
        int v[5];        /* Y (Y1), U, V, Y2, A as final sample values */
        int n = sub ? 4 : 3, k;
        for (k = 0; k < n; k++) {
            const int chroma = k == 1 || k == 2;
            const int fs = chroma ? cfs : lfs;
            const int16_t *f = chroma ? cf : lf;
            const int x = chroma ? i : (sub ? 2 * i + (k == 3) : i);
            int acc;
            /* a luma column beyond the line (second pixel of the last pair, odd widths) holds fill_ones()' value: 1 << 14 / 1 << 18 */
#define SMP(j) (chroma ? (k == 1 ? CU(j) : CV(j))[x] : (x < dstW ? L(j)[x] : (bits == 16 ? 1 << 18 : 1 << 14)))
            if (bits == 16) { /* 0x40000000 bias trick of yuv2planeX_16_c_template */
                acc = (1 << 14) - 0x40000000;
                for (j = 0; j < fs; j++) acc = (int)((unsigned)acc + SMP(j) * (unsigned)f[j]);
                v[k] = 0x8000 + clip_i16(acc >> 15);
            } else {
                const int shift = 11 + 16 - bits;
                acc = 1 << (shift - 1);
                for (j = 0; j < fs; j++) acc = (int)((unsigned)acc + SMP(j) * (unsigned)f[j]);
                v[k] = clip_uintp2(acc >> shift, bits);
            }
        }
        if (df == ORF_AYUV64LE) {
            if (c->needAlpha) {
                int acc = (1 << 14) - 0x40000000;
                for (j = 0; j < lfs; j++) acc = (int)((unsigned)acc + AL(j)[i] * (unsigned)lf[j]);
                v[4] = 0x8000 + clip_i16(acc >> 15);
            } else v[4] = 65535;
        }
        if (df == ORF_XV30LE || df == ORF_V30XLE) { /* one little-endian dword */
            const int sh = df == ORF_XV30LE ? 0 : 2;
            const uint32_t px = (uint32_t)v[1] << (sh + 0) | (uint32_t)v[0] << (sh + 10) | (uint32_t)v[2] << (sh + 20) | 3u << (sh ? 0 : 30);
            memcpy(dest + 4 * i, &px, 4);
        } else {
            const int step = sub ? 8 : dd->c[0].step;      /* bytes per unit */
            uint16_t w[4] = { 0, 0, 0, 0 };
            w[dd->c[0].offset / 2] = (uint16_t)(v[0] << dd->c[0].shift);
            w[dd->c[1].offset / 2] = (uint16_t)(v[1] << dd->c[1].shift);
            w[dd->c[2].offset / 2] = (uint16_t)(v[2] << dd->c[2].shift);
            if (sub) w[dd->c[0].offset / 2 + 2] = (uint16_t)(v[3] << dd->c[0].shift);
            else if (df == ORF_AYUV64LE) w[0] = (uint16_t)v[4];
            else if (df == ORF_XV36LE) w[3] = 0xFFF0;      /* av_clip_uintp2(65535, 12) << 4 */
            else w[3] = 65535;                              /* xv48 */
            memcpy(dest + (size_t)step * i, w, 8);
        }
    }
#undef L
#undef CU
#undef CV
#undef AL
}

/* packed_vscale + yuv2ayuv_{1,2,X}_c_template (output.c:2903-3060: ayuv / vuya / vuyx / uyva) and yuv2vyu444_{1,2,X}_c (:3171-3290) */
static void write_packed444_line(const OrSws *c, const Planes *P, uint8_t *dest, int y)
{
    const int dstW = c->o.dst_w, lw = dstW, cw = c->chrDstW;
    const int srcH = c->o.src_h, chrSrcH = c->chrSrcH;
    const int chrY = y >> c->chrDstVSub;
    const int lfs = c->vLumFilterSize, cfs = c->vChrFilterSize;
    const int16_t *lf = c->vLumFilter + y * lfs, *cf = c->vChrFilter + chrY * cfs;
    const int firstLum = ORMAX(1 - lfs, c->vLumFilterPos[y]);
    const int firstChr = ORMAX(1 - cfs, c->vChrFilterPos[chrY]);
    const Desc *dd = desc_get(c->o.dst_format);
    const int step = dd->c[0].step, hasAlpha = c->needAlpha;
    int i, j, mode, ua = 0, ya = 0;
#define L(j) (P->lum + RING(P->lmask, ORMIN(firstLum + (j), srcH - 1)) * lw)
#define CU(j) (P->chrU + RING(P->cmask, ORMIN(firstChr + (j), chrSrcH - 1)) * cw)
#define CV(j) (P->chrV + RING(P->cmask, ORMIN(firstChr + (j), chrSrcH - 1)) * cw)
#define AL(j) (P->alp + RING(P->lmask, ORMIN(firstLum + (j), srcH - 1)) * lw)
    if (lfs == 1 && cfs == 1) mode = 1;
    else if (lfs == 1 && cfs == 2 && (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U) { mode = 1; ua = (uint16_t)cf[1]; }
    else if (lfs == 2 && cfs == 2 && (uint16_t)lf[1] + (uint16_t)lf[0] == 4096 && (uint16_t)lf[1] <= 4096U &&
             (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U) { mode = 2; ya = (uint16_t)lf[1]; ua = (uint16_t)cf[1]; }
    else mode = 0;
    for (i = 0; i < dstW; i++) {
        int Y, U, V, A = 255;
        if (mode == 0) {
            Y = U = V = 1 << 18;
            for (j = 0; j < lfs; j++) Y = (int)((unsigned)Y + L(j)[i] * (unsigned)lf[j]);
            for (j = 0; j < cfs; j++) { U = (int)((unsigned)U + CU(j)[i] * (unsigned)cf[j]); V = (int)((unsigned)V + CV(j)[i] * (unsigned)cf[j]); }
            Y >>= 19; U >>= 19; V >>= 19;
            if (hasAlpha) {
                A = 1 << 18;
                for (j = 0; j < lfs; j++) A = (int)((unsigned)A + AL(j)[i] * (unsigned)lf[j]);
                A >>= 19;
                if (A & 0x100) A = clip_u8(A);
            }
        } else if (mode == 2) {
            Y = (L(0)[i] * (4096 - ya) + L(1)[i] * ya) >> 19;
            U = (CU(0)[i] * (4096 - ua) + CU(1)[i] * ua) >> 19;
            V = (CV(0)[i] * (4096 - ua) + CV(1)[i] * ua) >> 19;
            if (hasAlpha) A = clip_u8((AL(0)[i] * (4096 - ya) + AL(1)[i] * ya) >> 19);
        } else {
            Y = (L(0)[i] + 64) >> 7;
            if (ua < 2048) { U = (CU(0)[i] + 64) >> 7; V = (CV(0)[i] + 64) >> 7; }
            else { U = (CU(0)[i] + CU(1)[i] + 128) >> 8; V = (CV(0)[i] + CV(1)[i] + 128) >> 8; }
            if (hasAlpha) { A = (AL(0)[i] + 64) >> 7; if (A & 0x100) A = clip_u8(A); }
        }
        if (Y & 0x100) Y = clip_u8(Y);
        if (U & 0x100) U = clip_u8(U);
        if (V & 0x100) V = clip_u8(V);
        dest[step * i + dd->c[0].offset] = (uint8_t)Y; dest[step * i + dd->c[1].offset] = (uint8_t)U; dest[step * i + dd->c[2].offset] = (uint8_t)V;
        if (step == 4) dest[4 * i + dd->c[3].offset] = (uint8_t)A;
    }
#undef L
#undef CU
#undef CV
#undef AL
}

/* any_vscale (vscale.c:173-212) + yuv2gbrp_full_X_c / yuv2gbrp16_full_X_c / yuv2gbrpf32_full_X_c (output.c:2342-2580):
 * planar RGB destinations always use the X form.  dst planes are G, B, R. */
static void write_planar_rgb_line(const OrSws *c, const Planes *P, uint8_t *const dst[], const int dstStride[], int y)
{
    const int dstW = c->o.dst_w, lw = dstW, cw = c->chrDstW;
    const int srcH = c->o.src_h, chrSrcH = c->chrSrcH;
    const int chrY = y >> c->chrDstVSub;
    const int lfs = c->vLumFilterSize, cfs = c->vChrFilterSize;
    const int16_t *lf = c->vLumFilter + y * lfs, *cf = c->vChrFilter + chrY * cfs;
    const int firstLum = ORMAX(1 - lfs, c->vLumFilterPos[y]);
    const int firstChr = ORMAX(1 - cfs, c->vChrFilterPos[chrY]);
    const Desc *dd = desc_get(c->o.dst_format);
    const int depth = dd->c[0].depth, isf = !!(dd->flags & PF_FLOAT);
    uint8_t *dg = dst[0] + (size_t)y * dstStride[0], *db = dst[1] + (size_t)y * dstStride[1], *dr = dst[2] + (size_t)y * dstStride[2];
    int i, j;
#define L(j) (P->lum + RING(P->lmask, ORMIN(firstLum + (j), srcH - 1)) * lw)
#define CU(j) (P->chrU + RING(P->cmask, ORMIN(firstChr + (j), chrSrcH - 1)) * cw)
#define CV(j) (P->chrV + RING(P->cmask, ORMIN(firstChr + (j), chrSrcH - 1)) * cw)
    for (i = 0; i < dstW; i++) {
        int Y, U, V, R, G, B;
        if (depth <= 14) { /* yuv2gbrp_full_X_c :2342-2421, 15-bit intermediates */
            const int SH = 22 + 8 - depth;
            Y = 1 << 9; U = (1 << 9) - (128 << 19); V = (1 << 9) - (128 << 19);
            for (j = 0; j < lfs; j++) Y = (int)((unsigned)Y + L(j)[i] * (unsigned)lf[j]);
            for (j = 0; j < cfs; j++) { U = (int)((unsigned)U + CU(j)[i] * (unsigned)cf[j]); V = (int)((unsigned)V + CV(j)[i] * (unsigned)cf[j]); }
            Y >>= 10; U >>= 10; V >>= 10;
            Y -= c->yuv2rgb_y_offset;
            Y = (int)((unsigned)Y * (unsigned)c->yuv2rgb_y_coeff);
            Y = (int)((unsigned)Y + (1u << (SH - 1)));
            R = (int)((unsigned)Y + (unsigned)V * (unsigned)c->yuv2rgb_v2r);
            G = (int)((unsigned)Y + (unsigned)V * (unsigned)c->yuv2rgb_v2g + (unsigned)U * (unsigned)c->yuv2rgb_u2g);
            B = (int)((unsigned)Y + (unsigned)U * (unsigned)c->yuv2rgb_u2b);
            if ((R | G | B) & 0xC0000000) { R = clip_uintp2(R, 30); G = clip_uintp2(G, 30); B = clip_uintp2(B, 30); }
            if (SH != 22) {   /* (yuv2gbrpmsb_full_X_c output.c:2424-2462: the same samples << (16 - depth)) */
                const int ms = desc_get(c->o.dst_format)->c[0].shift;
                ((uint16_t *)dg)[i] = (uint16_t)((G >> SH) << ms); ((uint16_t *)db)[i] = (uint16_t)((B >> SH) << ms); ((uint16_t *)dr)[i] = (uint16_t)((R >> SH) << ms);
            } else {
                dg[i] = (uint8_t)(G >> 22); db[i] = (uint8_t)(B >> 22); dr[i] = (uint8_t)(R >> 22);
            }
        } else { /* yuv2gbrp16_full_X_c :2467-2530 / yuv2gbrpf32_full_X_c :2533-2605, 19-bit intermediates */
            Y = -0x40000000; U = -(128 << 23); V = -(128 << 23);
            for (j = 0; j < lfs; j++) Y = (int)((unsigned)Y + L(j)[i] * (unsigned)lf[j]);
            for (j = 0; j < cfs; j++) { U = (int)((unsigned)U + CU(j)[i] * (unsigned)cf[j]); V = (int)((unsigned)V + CV(j)[i] * (unsigned)cf[j]); }
            Y >>= 14; Y += 0x10000; U >>= 14; V >>= 14;
            Y -= c->yuv2rgb_y_offset;
            Y = (int)((unsigned)Y * (unsigned)c->yuv2rgb_y_coeff);
            Y = (int)((unsigned)Y + (unsigned)((1 << 13) - (1 << 29)));
            R = (int)((unsigned)V * (unsigned)c->yuv2rgb_v2r);
            G = (int)((unsigned)V * (unsigned)c->yuv2rgb_v2g + (unsigned)U * (unsigned)c->yuv2rgb_u2g);
            B = (int)((unsigned)U * (unsigned)c->yuv2rgb_u2b);
            if (!isf) { /* 64-bit sums */
                ((uint16_t *)dr)[i] = (uint16_t)clip_uintp2((int)(((int64_t)Y + R) >> 14) + (1 << 15), 16);
                ((uint16_t *)dg)[i] = (uint16_t)clip_uintp2((int)(((int64_t)Y + G) >> 14) + (1 << 15), 16);
                ((uint16_t *)db)[i] = (uint16_t)clip_uintp2((int)(((int64_t)Y + B) >> 14) + (1 << 15), 16);
            } else {   /* 32-bit sums (the float writer adds in int), then float_mult * (float)v */
                static const float float_mult = 1.0f / 65535.0f;
                R = clip_uintp2(((int)((unsigned)Y + (unsigned)R) >> 14) + (1 << 15), 16);
                G = clip_uintp2(((int)((unsigned)Y + (unsigned)G) >> 14) + (1 << 15), 16);
                B = clip_uintp2(((int)((unsigned)Y + (unsigned)B) >> 14) + (1 << 15), 16);
                ((float *)dg)[i] = float_mult * (float)G; ((float *)db)[i] = float_mult * (float)B; ((float *)dr)[i] = float_mult * (float)R;
            }
        }
    }
    if (isALPHA(c->o.dst_format)) {   /* the A plane: from the alpha lines through the luma filter (hasAlpha), or filled (swscale.c:536-552) */
        uint8_t *da = dst[3] + (size_t)y * dstStride[3];
        const int abits = dd->c[3].depth;
        for (i = 0; i < dstW; i++) {
            if (P->alp) {
                int A;
                const int32_t *al;
                if (depth <= 14) {
                    const int SH = 22 + 8 - depth;
                    A = 1 << 18;
                    for (j = 0; j < lfs; j++) { al = P->alp + RING(P->lmask, ORMIN(firstLum + j, srcH - 1)) * lw; A = (int)((unsigned)A + al[i] * (unsigned)lf[j]); }
                    if (A & 0xF8000000) A = clip_uintp2(A, 27);
                    if (SH != 22) ((uint16_t *)da)[i] = (uint16_t)(A >> (SH - 3)); else da[i] = (uint8_t)(A >> 19);
                } else {
                    A = -0x40000000;
                    for (j = 0; j < lfs; j++) { al = P->alp + RING(P->lmask, ORMIN(firstLum + j, srcH - 1)) * lw; A = (int)((unsigned)A + al[i] * (unsigned)lf[j]); }
                    A >>= 1; A += 0x20002000;
                    if (isf) ((float *)da)[i] = (1.0f / 65535.0f) * (float)(clip_uintp2(A, 30) >> 14);
                    else ((uint16_t *)da)[i] = (uint16_t)(clip_uintp2(A, 30) >> 14);
                }
            } else if (isf) { const uint32_t one = 0x3f800000; memcpy(da + 4 * i, &one, 4); }          /* fillPlane32 */
            else if (depth > 8) ((uint16_t *)da)[i] = (uint16_t)(0xFFFF >> (16 - abits));               /* fillPlane16 */
            else da[i] = 255;                                                                           /* fillPlane */
        }
    }
#undef L
#undef CU
#undef CV
}

/* ---- ff_swscale (swscale.c:263-567) for a whole frame, in the reference's own order of events ----
 * The reference pulls destination rows: for every dstY it first makes sure the horizontal ring holds the source lines the row needs
 * (running the line converters and the horizontal scaler over BATCHES of lines, ahead of need as far as the ring allows), then runs the
 * vertical scaler for that one row.  For almost every context the result equals "h-scale the whole picture, then v-scale it"; the order
 * matters where a stage has a side effect or a per-batch term:
 *   - gamma_convert (gamma.c:31-58), the first luma descriptor of the gamma cascade's scaling step (slice.c:325-328), rewrites the source
 *     lines of a batch IN PLACE.  A line that was converted ahead of need and is then pulled again after a "hole" (a jump of the vertical
 *     filter position beyond lastInLumBuf + 1 re-bases the ring, swscale.c:404-417, :444-451) is converted a second time;
 *   - chr_convert (hscale.c:211-225) derives plane 0's line index from the start of the batch, see read_chr_line().
 * So the lines are produced here batch by batch exactly as swscale.c:388-535 schedules them; the h-scaled lines live in whole-frame arrays
 * indexed by source line (what a ring slot holds is always the latest version of its line). */
static void lum_line(OrSws *c, const uint8_t *const src[], const int srcStride[], int y, Planes *P, uint8_t *t0)
{   /* lum_convert + lum_h_scale, hscale.c:39-131; plane 3 with the LUMA filter and no range conversion when needAlpha (desc->alpha) */
    const int sf = c->o.src_format, srcW = c->o.src_w, dstW = c->o.dst_w;
    const uint8_t *line = read_lum_line(c, src, srcStride, y, t0);
    int32_t *d = P->lum + RING(P->lmask, y) * dstW;
    hscale_line(c, d, dstW, line, c->hLumFilter, c->hLumFilterPos, c->hLumFilterSize);
    if (c->range_active) range_line(c, d, dstW, 0);
    if (!c->needAlpha) return;
    const uint8_t *aline;
    if (sf == ORF_YA8) { /* uyvyToY_c on the alpha byte (input.c:2773-2775) */
        for (int i = 0; i < srcW; i++) t0[i] = src[0][(ptrdiff_t)y * srcStride[0] + 2 * i + 1];
        aline = t0;
    } else if (sf == ORF_YA16LE) { /* read_ya16le_alpha_c input.c:639-645 */
        uint16_t *d16 = (uint16_t *)t0;
        for (int i = 0; i < srcW; i++) memcpy(&d16[i], src[0] + (ptrdiff_t)y * srcStride[0] + 4 * i + 2, 2);
        aline = t0;
    } else if (sf == ORF_PAL8) { /* palToA_c input.c:474-484 */
        int16_t *d16 = (int16_t *)t0;
        for (int i = 0; i < srcW; i++) { const uint32_t p = c->pal_yuv[src[0][(ptrdiff_t)y * srcStride[0] + i]]; d16[i] = (int16_t)((p >> 24) << 6 | p >> 26); }
        aline = t0;
    } else if (sf == ORF_YAF32LE || sf == ORF_YAF16LE || sf == ORF_RGBAF16LE) { /* read_yaf32_alpha_c (input.c:1422-1431), read_yaf16_alpha_c (:1620-1627),
                                                                                 * rgbaf16ToA_endian (:1680-1687): the last element of the pixel */
        const Desc *dsd = desc_get(sf);
        const int half = dsd->c[0].depth == 16, ai = dsd->nb - 1;
        const uint8_t *sp = src[0] + (ptrdiff_t)y * srcStride[0] + dsd->c[ai].offset;
        uint16_t *d16 = (uint16_t *)t0;
        for (int i = 0; i < srcW; i++) d16[i] = (uint16_t)rdf16(sp + dsd->c[ai].step * i, half);
        aline = t0;
    } else if (sf == ORF_RGBA64LE || sf == ORF_BGRA64LE) { /* rgba64leToA_c: the 16-bit A sample as is */
        const uint16_t *sp = (const uint16_t *)(src[0] + (ptrdiff_t)y * srcStride[0]) + 3;
        uint16_t *d16 = (uint16_t *)t0;
        for (int i = 0; i < srcW; i++) d16[i] = sp[4 * i];
        aline = t0;
    } else if (sf == ORF_AYUV64LE) { /* read_ayuv64le_A_c input.c:715-721 */
        const uint8_t *sp = src[0] + (ptrdiff_t)y * srcStride[0];
        uint16_t *d16 = (uint16_t *)t0;
        for (int i = 0; i < srcW; i++) memcpy(&d16[i], sp + 8 * i, 2);
        aline = t0;
    } else if (isPacked444(sf)) { /* read_vuya_A_c / read_ayuv_A_c input.c:749-781 */
        const Desc *dsd = desc_get(sf);
        const uint8_t *sp = src[0] + (ptrdiff_t)y * srcStride[0] + dsd->c[3].offset;
        for (int i = 0; i < srcW; i++) t0[i] = sp[4 * i];
        aline = t0;
    } else if (isPlanarRGB(sf)) { /* planar_rgb_to_a (input.c:1188-1194), planar_rgb16_s16_to_a (:1235-1247), planar_rgbf32_to_a (:1289-1298) */
        const Desc *dsd = desc_get(sf);
        uint16_t *d16 = (uint16_t *)t0;
        const uint8_t *sp = src[3] + (ptrdiff_t)y * srcStride[3];
        if ((dsd->flags & PF_FLOAT) && dsd->c[0].depth == 16) for (int i = 0; i < srcW; i++) d16[i] = (uint16_t)rdf16(sp + 2 * i, 1);   /* planar_rgbf16_to_a :1561-1568 */
        else if (dsd->flags & PF_FLOAT) for (int i = 0; i < srcW; i++) d16[i] = (uint16_t)f2u16(((const float *)sp)[i]);
        else if (dsd->c[0].depth == 8) for (int i = 0; i < srcW; i++) d16[i] = (uint16_t)(sp[i] << 6);
        else { const int bpc = dsd->c[0].depth, sh = 14 - (bpc < 16 ? bpc : 14); for (int i = 0; i < srcW; i++) d16[i] = (uint16_t)(((const uint16_t *)sp)[i] << sh); }
        aline = t0;
    } else if (isAnyRGB(sf)) { /* rgbaToA_c / abgrToA_c input.c:454-472 */
        const Desc *dsd = desc_get(sf);
        const uint8_t *sp = src[0] + (ptrdiff_t)y * srcStride[0] + dsd->c[3].offset;
        int16_t *d16 = (int16_t *)t0;
        const int opaque = c->src0Alpha && !c->dst0Alpha; /* rgb0-style source: swscale.c:1106-1124 sets the X byte to 255 first */
        for (int i = 0; i < srcW; i++) { const int a = opaque ? 255 : sp[4 * i]; d16[i] = (int16_t)(a << 6 | a >> 2); }
        aline = t0;
    } else aline = src[3] + (ptrdiff_t)y * srcStride[3];
    hscale_line(c, P->alp + RING(P->lmask, y) * dstW, dstW, aline, c->hLumFilter, c->hLumFilterPos, c->hLumFilterSize);
}

static void chr_line(OrSws *c, const uint8_t *const src[], const int srcStride[], int y, int lum_row, Planes *P, uint8_t *t0, uint8_t *t1)
{   /* chr_convert + chr_h_scale, hscale.c:168-245 */
    const uint8_t *pu, *pv;
    int32_t *du = P->chrU + RING(P->cmask, y) * c->chrDstW, *dv = P->chrV + RING(P->cmask, y) * c->chrDstW;
    read_chr_line(c, src, srcStride, y, lum_row, t0, t1, &pu, &pv);
    hscale_line(c, du, c->chrDstW, pu, c->hChrFilter, c->hChrFilterPos, c->hChrFilterSize);
    hscale_line(c, dv, c->chrDstW, pv, c->hChrFilter, c->hChrFilterPos, c->hChrFilterSize);
    if (c->range_active) { range_line(c, du, c->chrDstW, 1); range_line(c, dv, c->chrDstW, 1); }
}

/* the vertical descriptors for destination row y (vscale.c, output.c) */
static void out_row(OrSws *c, const Planes *P, uint8_t *const dst[], const int dstStride[], int y)
{
    const int srcH = c->o.src_h, dstW = c->o.dst_w;
    const int df = c->o.dst_format, sf = c->o.src_format;
    const int should_dither = isNBPS(sf) || is16BPS(sf);
    const int chrDstY = y >> c->chrDstVSub;
    const uint8_t *lumDither = should_dither ? dither_8x8_128[y & 7] : pb_64;       /* swscale.c:385-387, :519-522 */
    const uint8_t *chrDither = should_dither ? dither_8x8_128[chrDstY & 7] : pb_64;
    if (isGray(df) && !isYA(df)) { /* vscale.c:219-233: luma only */
        int firstLum = ORMAX(1 - c->vLumFilterSize, c->vLumFilterPos[y]);
        write_planar_line(c, dst[0] + (size_t)y * dstStride[0], dstW, P->lum, P->lmask, dstW, srcH, firstLum,
                          c->vLumFilter + y * c->vLumFilterSize, c->vLumFilterSize, lumDither, 0, 1);
    } else if (isPlanarYUV(df)) {
        const Desc *dd = desc_get(df);
        int firstLum = ORMAX(1 - c->vLumFilterSize, c->vLumFilterPos[y]);
        write_planar_line(c, dst[0] + (size_t)y * dstStride[0], dstW, P->lum, P->lmask, dstW, srcH, firstLum,
                          c->vLumFilter + y * c->vLumFilterSize, c->vLumFilterSize, lumDither, 0, 1);
        if (isALPHA(df)) {
            if (c->needAlpha) /* lum_planar_vscale vscale.c:59-71: same writer, luma filter, luma dither */
                write_planar_line(c, dst[3] + (size_t)y * dstStride[3], dstW, P->alp, P->lmask, dstW, srcH, firstLum,
                                  c->vLumFilter + y * c->vLumFilterSize, c->vLumFilterSize, lumDither, 0, 1);
            else if (dd->c[0].depth > 8) { /* fillPlane16 (swscale.c:536-552, swscale_internal.h fillPlane16): 0xFFFF >> (16 - bits) */
                uint16_t *a16 = (uint16_t *)(dst[3] + (size_t)y * dstStride[3]);
                for (int i = 0; i < dstW; i++) a16[i] = (uint16_t)(0xFFFF >> (16 - dd->c[3].depth));
            } else memset(dst[3] + (size_t)y * dstStride[3], 255, dstW); /* fillPlane swscale.c:536-552 */
        }
        if (!(y & ((1 << c->chrDstVSub) - 1))) { /* chr_planar_vscale vscale.c:74-107 */
            int firstChr = ORMAX(1 - c->vChrFilterSize, c->vChrFilterPos[chrDstY]);
            const int16_t *cf = c->vChrFilter + chrDstY * c->vChrFilterSize;
            if (isSemiPlanarYUV(df)) {
                write_nv_chroma_line(c, dst[1] + (size_t)chrDstY * dstStride[1], c->chrDstW, P->chrU, P->chrV, P->cmask,
                                     c->chrDstW, c->chrSrcH, firstChr, cf, c->vChrFilterSize, chrDither);
            } else {
                write_planar_line(c, dst[dd->c[1].plane] + (size_t)chrDstY * dstStride[dd->c[1].plane], c->chrDstW, P->chrU, P->cmask,
                                  c->chrDstW, c->chrSrcH, firstChr, cf, c->vChrFilterSize, chrDither, 0, 0);
                write_planar_line(c, dst[dd->c[2].plane] + (size_t)chrDstY * dstStride[dd->c[2].plane], c->chrDstW, P->chrV, P->cmask,
                                  c->chrDstW, c->chrSrcH, firstChr, cf, c->vChrFilterSize, chrDither, 3, 0);
            }
        }
    } else if (isYA(df)) {
        write_ya_line(c, P, dst[0] + (size_t)y * dstStride[0], y);
    } else if (isMono(df)) {
        write_mono_line(c, P, dst[0] + (size_t)y * dstStride[0], y);
    } else if (isPackedHi(df)) {
        write_packedhi_line(c, P, dst[0] + (size_t)y * dstStride[0], y);
    } else if (isPacked444(df)) {
        write_packed444_line(c, P, dst[0] + (size_t)y * dstStride[0], y);
    } else if (df == ORF_YUYV422 || df == ORF_UYVY422 || df == ORF_YVYU422) {
        write_packed422_line(c, P, dst[0] + (size_t)y * dstStride[0], y);
    } else if (df == ORF_RGB48LE || df == ORF_BGR48LE || df == ORF_RGBA64LE || df == ORF_BGRA64LE) {
        write_packed_rgb16_line(c, P, dst[0] + (size_t)y * dstStride[0], y);
    } else if (isAnyRGB(df) && !isPlanarRGB(df)) {
        write_packed_rgb_line(c, P, dst[0] + (size_t)y * dstStride[0], y);
    } else {
        write_planar_rgb_line(c, P, dst, dstStride, y);
    }
}

/* get_min_buffer_size (slice.c:217-243) and the floor of slice.c:266-267 (MAX_LINES_AHEAD = 4): lines per plane of the horizontal ring */
static void ring_sizes(const OrSws *c, int *lum, int *chr)
{
    const int dstH = c->o.dst_h, sub = c->chrSrcVSub;
    *lum = c->vLumFilterSize; *chr = c->vChrFilterSize;
    for (int lumY = 0; lumY < dstH; lumY++) {
        const int chrY = (int)((int64_t)lumY * c->chrDstH / dstH);
        int nextSlice = ORMAX(c->vLumFilterPos[lumY] + c->vLumFilterSize - 1, (c->vChrFilterPos[chrY] + c->vChrFilterSize - 1) << sub);
        nextSlice >>= sub; nextSlice <<= sub;
        *lum = ORMAX(*lum, nextSlice - c->vLumFilterPos[lumY]);
        *chr = ORMAX(*chr, (nextSlice >> sub) - c->vChrFilterPos[chrY]);
    }
    *lum = ORMAX(*lum, c->vLumFilterSize + 4);
    *chr = ORMAX(*chr, c->vChrFilterSize + 4);
}

static int main_path(OrSws *c, const uint8_t *const src[], const int srcStride[],
                     uint8_t *const dst[], const int dstStride[])
{
    const int srcW = c->o.src_w, srcH = c->o.src_h, dstW = c->o.dst_w, dstH = c->o.dst_h;
    const int df = c->o.dst_format, sf = c->o.src_format;
    const int vsub = c->chrSrcVSub, chrSrcSliceEnd = CEIL_RSHIFT(srcH, vsub);
    const int needs_hcscale = !(isGray(sf) || isGray(df) || isMono(sf));   /* swscale.c:692-694 */
    Planes P;
    uint8_t *t0 = malloc((size_t)srcW * 4 + 128), *t1 = malloc((size_t)srcW * 4 + 128);
    int y, lumAvail, chrAvail;
    /* the cursor of swscale.c:294-300, :372-381 for a frame that arrives in one slice */
    int lastInLumBuf = -1, lastInChrBuf = -1, hasLumHoles = 1, hasChrHoles = 1;
    int lumSliceY = 0, lumSliceH = 0, chrSliceY = 0, chrSliceH = 0;   /* plane 0 / plane 1 of the horizontal scaler's output slice */

    /* scale_internal (swscale.c:1084-1086): a bit-exact context starts every frame from a clean error line; any other one carries it on */
    if ((c->o.flags & OR_SWS_BITEXACT) && c->o.dither == 3 && c->dither_error[0])
        for (y = 0; y < 3; y++) memset(c->dither_error[y], 0, sizeof(int) * ((size_t)dstW + 2));

    ring_sizes(c, &lumAvail, &chrAvail);
    {   /* line rings: twice what is ever alive at once (the lines of an output row's window plus what the schedule below converts ahead), a power of two */
        int lr = 1, cr = 1;
        while (lr < 2 * lumAvail + 2) lr <<= 1;
        while (cr < 2 * chrAvail + 2) cr <<= 1;
        P.lmask = lr - 1; P.cmask = cr - 1;
        P.lum = calloc((size_t)lr * dstW, sizeof(int32_t));
        P.chrU = malloc((size_t)cr * c->chrDstW * sizeof(int32_t));
        P.chrV = malloc((size_t)cr * c->chrDstW * sizeof(int32_t));
        P.alp = c->needAlpha ? calloc((size_t)lr * dstW, sizeof(int32_t)) : NULL;
        /* fill_ones() (slice.c:190-208, :311): what a ring line holds before it is written -- the value a chroma line keeps for good when
         * ff_init_desc_no_chr stands in for the chroma scaler (:358-361) */
        const int32_t neutral = c->dstBpc >= 16 ? 1 << 18 : 1 << 14;
        for (size_t k = 0; k < (size_t)cr * c->chrDstW; k++) P.chrU[k] = P.chrV[k] = neutral;
    }

    for (y = 0; y < dstH; y++) {   /* swscale.c:388-535 */
        const int chrDstY = y >> c->chrDstVSub;
        const int firstLumSrcY = ORMAX(1 - c->vLumFilterSize, c->vLumFilterPos[y]);
        const int firstChrSrcY = ORMAX(1 - c->vChrFilterSize, c->vChrFilterPos[chrDstY]);
        const int lastLumSrcY = ORMIN(srcH, firstLumSrcY + c->vLumFilterSize) - 1;
        const int lastChrSrcY = ORMIN(c->chrSrcH, firstChrSrcY + c->vChrFilterSize) - 1;
        int posY, cPosY, firstPosY, lastPosY, firstCPosY, lastCPosY, k;
        if (firstLumSrcY > lastInLumBuf) {   /* "handle holes (FAST_BILINEAR & weird filters)" */
            hasLumHoles = lastInLumBuf != firstLumSrcY - 1;
            if (hasLumHoles) { lumSliceY = firstLumSrcY; lumSliceH = 0; }
            lastInLumBuf = firstLumSrcY - 1;
        }
        if (firstChrSrcY > lastInChrBuf) {
            hasChrHoles = lastInChrBuf != firstChrSrcY - 1;
            if (hasChrHoles) { chrSliceY = firstChrSrcY; chrSliceH = 0; }
            lastInChrBuf = firstChrSrcY - 1;
        }
        /* (a whole frame always has "enough_lines") */
        posY = lumSliceY + lumSliceH;
        if (posY <= lastLumSrcY && !hasLumHoles) { firstPosY = ORMAX(firstLumSrcY, posY); lastPosY = ORMIN(firstLumSrcY + lumAvail - 1, srcH - 1); }
        else { firstPosY = posY; lastPosY = lastLumSrcY; }
        cPosY = chrSliceY + chrSliceH;
        if (cPosY <= lastChrSrcY && !hasChrHoles) { firstCPosY = ORMAX(firstChrSrcY, cPosY); lastCPosY = ORMIN(firstChrSrcY + chrAvail - 1, chrSrcSliceEnd - 1); }
        else { firstCPosY = cPosY; lastCPosY = lastChrSrcY; }
        /* (ff_rotate_slice moves sliceY / sliceH by the ring size together: their sum, all that is used here, stays) */
        if (posY < lastLumSrcY + 1) {
            for (k = firstPosY; k <= lastPosY; k++) {
                if (c->internal_gamma_tab) {   /* gamma_convert on the source line, in place (gamma.c:31-58) */
                    uint16_t *row = (uint16_t *)(src[0] + (ptrdiff_t)k * srcStride[0]);
                    for (int x = 0; x < srcW; x++) for (int q = 0; q < 3; q++) row[4 * x + q] = c->internal_gamma_tab[row[4 * x + q]];
                }
                lum_line(c, src, srcStride, k, &P, t0);
            }
            lumSliceH += lastPosY - firstPosY + 1;
        }
        lastInLumBuf = lastLumSrcY;
        if (cPosY < lastChrSrcY + 1) {
            if (needs_hcscale)
                for (k = firstCPosY; k <= lastCPosY; k++) chr_line(c, src, srcStride, k, (firstCPosY << vsub) + (k - firstCPosY), &P, t0, t1);
            chrSliceH += lastCPosY - firstCPosY + 1;   /* (no_chr_scale keeps its own books, hscale.c:271-278; nothing reads them) */
        }
        lastInChrBuf = lastChrSrcY;
        out_row(c, &P, dst, dstStride, y);
    }
    free(P.lum); free(P.chrU); free(P.chrV); free(P.alp); free(t0); free(t1);
    return dstH;
}

static int scale_le(OrSws *c, const uint8_t *const src[4], const int srcStride[4], int srcSliceY, int srcSliceH,
                    uint8_t *const dst[4], const int dstStride[4]);

/* init_xyz_tables / ff_sws_fill_xyztables (utils.c:709-770): gamma LUTs from libm pow(), Q12 matrices */
static uint16_t xyzgamma_tab[4096], rgbgammainv_tab[4096], rgbgamma_tab[65536], xyzgammainv_tab[65536];
static void init_xyz_tables(void)
{
    static int done;
    if (done) return;
    for (int i = 0; i < 4096; i++) {
        xyzgamma_tab[i]    = (uint16_t)lrint(pow(i / 4095.0, 2.6) * 65535.0);
        rgbgammainv_tab[i] = (uint16_t)lrint(pow(i / 4095.0, 2.2) * 65535.0);
    }
    for (int i = 0; i < 65536; i++) {
        rgbgamma_tab[i]    = (uint16_t)lrint(pow(i / 65535.0, 1.0 / 2.2) * 4095.0);
        xyzgammainv_tab[i] = (uint16_t)lrint(pow(i / 65535.0, 1.0 / 2.6) * 4095.0);
    }
    done = 1;
}

/* xyz12Torgb48_c (swscale.c:745-802) when to_rgb, rgb48Toxyz12_c (:804-861) otherwise; little-endian words */
static void xyz_convert(uint8_t *dst, int dstStride, const uint8_t *src, int srcStride, int w, int h, int to_rgb)
{
    static const int16_t xyz2rgb[3][3] = { { 13270, -6295, -2041 }, { -3969, 7682, 170 }, { 228, -835, 4329 } };
    static const int16_t rgb2xyz[3][3] = { { 1689, 1464, 739 }, { 871, 2929, 296 }, { 79, 488, 3891 } };
    const int16_t (*m)[3] = to_rgb ? xyz2rgb : rgb2xyz;
    const uint16_t *gin = to_rgb ? xyzgamma_tab : rgbgammainv_tab, *gout = to_rgb ? rgbgamma_tab : xyzgammainv_tab;
    init_xyz_tables();
    for (int yp = 0; yp < h; yp++) {
        const uint16_t *s = (const uint16_t *)(src + (ptrdiff_t)yp * srcStride);
        uint16_t *d = (uint16_t *)(dst + (ptrdiff_t)yp * dstStride);
        for (int xp = 0; xp < 3 * w; xp += 3) {
            const int a = gin[s[xp] >> 4], b = gin[s[xp + 1] >> 4], e = gin[s[xp + 2] >> 4];
            const int o0 = clip_u16((m[0][0] * a + m[0][1] * b + m[0][2] * e) >> 12);
            const int o1 = clip_u16((m[1][0] * a + m[1][1] * b + m[1][2] * e) >> 12);
            const int o2 = clip_u16((m[2][0] * a + m[2][1] * b + m[2][2] * e) >> 12);
            d[xp] = (uint16_t)(gout[o0] << 4); d[xp + 1] = (uint16_t)(gout[o1] << 4); d[xp + 2] = (uint16_t)(gout[o2] << 4);
        }
    }
}
/* sws_scale's XYZ stages (swscale.c:1126-1139, :1194-1210): the source slice is converted into a scratch picture first, the
 * written destination rows in place afterwards; both are skipped for an xyz12 -> xyz12 conversion at equal sizes */
static int scale_xyz(OrSws *c, const uint8_t *const src[4], const int srcStride[4], int srcSliceY, int srcSliceH,
                     uint8_t *const dst[4], const int dstStride[4])
{
    const int same = c->src_xyz && c->dst_xyz && c->o.src_w == c->o.dst_w && c->o.src_h == c->o.dst_h;
    const uint8_t *sp[4] = { src[0], src[1], src[2], src[3] };
    uint8_t *scratch = NULL;
    int ret;
    /* scale_internal (swscale.c:1076-1082) hands a cascaded context over to scale_gamma / scale_cascaded BEFORE it reaches its XYZ passes
     * (:1126-1139, :1194-1210), and the children were created with the formats handle_formats() had already aliased to rgb48
     * (utils.c:1461-1522, :1524-1550, :1565-1601, :1803-1833): an xyz12 picture on either side of a cascade is treated as rgb48 */
    if (c->cascade[0]) return scale_le(c, src, srcStride, srcSliceY, srcSliceH, dst, dstStride);
    if (c->src_xyz && !same) {
        const int st = srcStride[0] < 0 ? -srcStride[0] : srcStride[0];
        uint8_t *base;
        scratch = malloc((size_t)st * srcSliceH + 32);
        base = srcStride[0] < 0 ? scratch + (ptrdiff_t)st * (srcSliceH - 1) : scratch;
        xyz_convert(base, srcStride[0], src[0], srcStride[0], c->o.src_w, srcSliceH, 1);
        sp[0] = base;
    }
    ret = scale_le(c, sp, srcStride, srcSliceY, srcSliceH, dst, dstStride);
    free(scratch);
    if (ret >= 0 && c->dst_xyz && !same)
        xyz_convert(dst[0], dstStride[0], dst[0], dstStride[0], c->o.dst_w, c->o.dst_h, 0);   /* (whole frames only here) */
    return ret;
}

/* rows and visible bytes per row of plane k */
static void plane_geom(const Desc *d, int w, int h, int k, int *rows, int *row_bytes)
{
    int maxb = 0, chroma = 0, used = 0;
    for (int i = 0; i < d->nb; i++) {
        if (d->c[i].plane != k) continue;
        used = 1;
        if ((i == 1 || i == 2) && !(d->flags & PF_RGB)) chroma = 1;
    }
    if (!used) { *rows = 0; *row_bytes = 0; return; }
    {
        const int pw = chroma ? -((-w) >> d->lw) : w;
        for (int i = 0; i < d->nb; i++)
            if (d->c[i].plane == k) { const int b = d->c[i].step * pw; if (b > maxb) maxb = b; }
        if (!(d->flags & PF_PLANAR) && d->nb >= 3 && !(d->flags & PF_RGB)) maxb = d->c[0].step * w;   /* packed 4:2:2 */
    }
    if (isMono(d->fmt)) maxb = (w + 7) >> 3;
    if (d->fmt == ORF_UYYVYY411) maxb = 6 * ((w + 3) >> 2);   /* av_image_get_linesize: the widest step (6) over the chroma-shifted width */
    *rows = chroma ? -((-h) >> d->lh) : h;
    *row_bytes = maxb;
}
static void bswap_rows(uint8_t *dst, int dstStride, const uint8_t *src, int srcStride, int rows, int row_bytes, int unit)
{
    for (int y = 0; y < rows; y++) {
        const uint8_t *s = src + (ptrdiff_t)y * srcStride;
        uint8_t *d = dst + (ptrdiff_t)y * dstStride;
        for (int x = 0; x + unit <= row_bytes; x += unit) {
            uint8_t t[4];
            for (int b = 0; b < unit; b++) t[b] = s[x + unit - 1 - b];
            for (int b = 0; b < unit; b++) d[x + b] = t[b];
        }
    }
}

int or_sws_scale(OrSws *c, const uint8_t *const src[4], const int srcStride[4], int srcSliceY, int srcSliceH,
                 uint8_t *const dst[4], const int dstStride[4])
{
    if (!c || !src || !dst || !srcStride || !dstStride) return -22;
    if (srcSliceY != 0 || srcSliceH != c->o.src_h) return -22; /* oracle: whole frames only */
    if (c->bswap16 && !c->cascade[0] && !c->casc_gamma) {   /* bswap_16bpc (swscale_unscaled.c:545-570) on the caller's planes: min(|strides|) / 2 words of srcSliceH >> chrDstVSubSample rows of every plane */
        for (int p = 0; p < 4; p++) {
            const int srcstr = srcStride[p] / 2, dststr = dstStride[p] / 2;
            const int min_stride = ORMIN(abs(srcstr), abs(dststr));
            const uint16_t *sp = (const uint16_t *)src[p];
            uint16_t *dp = (uint16_t *)dst[p];
            if (!sp || !dp) continue;
            dp += (srcSliceY >> c->chrDstVSub) * dststr;
            for (int i = 0; i < (srcSliceH >> c->chrDstVSub); i++) {
                for (int j = 0; j < min_stride; j++) dp[j] = (uint16_t)((sp[j] << 8) | (sp[j] >> 8));
                sp += srcstr; dp += dststr;
            }
        }
        return srcSliceH;
    }
    if (c->src_be || c->dst_be) {
        const Desc *ds = desc_get(c->o.src_format), *dd = desc_get(c->o.dst_format);
        const uint8_t *sp[4] = { src[0], src[1], src[2], src[3] };
        int ss[4] = { srcStride[0], srcStride[1], srcStride[2], srcStride[3] };
        uint8_t *tmp[4] = { NULL, NULL, NULL, NULL };
        int ret;
        if (c->src_be) {
            const int unit = ds->c[0].depth == 32 ? 4 : 2;
            for (int k = 0; k < 4; k++) {
                int rows, rb;
                plane_geom(ds, c->o.src_w, c->o.src_h, k, &rows, &rb);
                if (!rows || !src[k]) continue;
                tmp[k] = malloc((size_t)rows * rb + 16);
                bswap_rows(tmp[k], rb, src[k], srcStride[k], rows, rb, unit);
                sp[k] = tmp[k]; ss[k] = rb;
            }
        }
        ret = scale_xyz(c, sp, ss, srcSliceY, srcSliceH, dst, dstStride);
        for (int k = 0; k < 4; k++) free(tmp[k]);
        if (ret >= 0 && c->dst_be && !c->casc_child_dst_be) {
            const int unit = dd->c[0].depth == 32 ? 4 : 2;
            for (int k = 0; k < 4; k++) {
                int rows, rb;
                plane_geom(dd, c->o.dst_w, c->o.dst_h, k, &rows, &rb);
                if (!rows || !dst[k]) continue;
                bswap_rows(dst[k], dstStride[k], dst[k], dstStride[k], rows, rb, unit);
            }
        }
        return ret;
    }
    return scale_xyz(c, src, srcStride, srcSliceY, srcSliceH, dst, dstStride);
}

static int scale_le(OrSws *c, const uint8_t *const src[4], const int srcStride[4], int srcSliceY, int srcSliceH,
                    uint8_t *const dst[4], const int dstStride[4])
{
    if (c->casc_gamma) { /* scale_gamma, swscale.c:959-990; gamma_convert (gamma.c:31-58) in place on the RGB words of the RGBA64 pictures */
        uint8_t *t0[4] = { c->casc_tmp[0], NULL, NULL, NULL }, *t1[4] = { c->casc_tmp2, NULL, NULL, NULL };
        int s0[4] = { c->casc_stride[0], 0, 0, 0 }, s1[4] = { c->casc_stride2, 0, 0, 0 };
        uint8_t *const *out1 = c->cascade[2] ? t1 : dst;
        const int *os1 = c->cascade[2] ? s1 : dstStride;
        int ret = or_sws_scale(c->cascade[0], src, srcStride, 0, srcSliceH, t0, s0), y, x, k;
        if (ret < 0) return ret;
        /* (the inverse table is applied by the scaling step itself, line by line as its ring pulls them: main_path()) */
        c->cascade[1]->internal_gamma_tab = c->inv_gamma_tab;
        ret = or_sws_scale(c->cascade[1], (const uint8_t *const *)t0, s0, 0, c->o.src_h, out1, os1);
        if (ret < 0) return ret;
        for (y = 0; y < c->o.dst_h; y++) {
            uint16_t *row = (uint16_t *)(out1[0] + (ptrdiff_t)y * os1[0]);
            for (x = 0; x < c->o.dst_w; x++) for (k = 0; k < 3; k++) row[4 * x + k] = c->gamma_tab[row[4 * x + k]];
        }
        if (c->cascade[2]) ret = or_sws_scale(c->cascade[2], (const uint8_t *const *)t1, s1, 0, c->o.dst_h, dst, dstStride);
        return ret;
    }
    if (c->cascade[0]) { /* scale_cascaded, swscale.c:992-1018 */
        uint8_t *tmp[4] = { c->casc_tmp[0], c->casc_tmp[1], c->casc_tmp[2], c->casc_tmp[3] };
        int ret = or_sws_scale(c->cascade[0], src, srcStride, 0, srcSliceH, tmp, c->casc_stride);
        if (ret < 0) return ret;
        return or_sws_scale(c->cascade[1], (const uint8_t *const *)tmp, c->casc_stride, 0, c->cascade[0]->o.dst_h, dst, dstStride);
    }
    {   /* swscale.c:1106-1124: an rgb0-style source feeding a real alpha channel is made opaque first */
        const int opaque = c->src0Alpha && !c->dst0Alpha && isALPHA(c->o.dst_format);
        if (c->unscaled_kind == UNSC_RGB2RGB) return unscaled_rgb2rgb(c, src, srcStride, 0, srcSliceH, dst, dstStride, opaque);
        if (c->unscaled_kind == UNSC_RGBLOW) return unscaled_rgblow(c, src, srcStride, 0, srcSliceH, dst, dstStride);
        if (c->unscaled_kind == UNSC_PACKEDCOPY) return unscaled_packedcopy(c, src, srcStride, 0, srcSliceH, dst, dstStride, opaque);
        if (c->unscaled_kind == UNSC_PACKED2GBRP) return unscaled_packed2gbrp(c, src, srcStride, 0, srcSliceH, dst, dstStride, opaque);
        if (opaque && c->unscaled_kind) return -22; /* no other special converter takes an rgb0-style source to an alpha destination */
    }
    /* bgr24ToYv12Wrapper (:2062-2077), yvu9ToYv12Wrapper (:2079-2093), yuyv/uyvyToYuv420Wrapper (:423-470) with a yuva420p
     * destination: fillPlane(dst[3], ..., 255) after the converter */
    if (c->o.dst_format == ORF_YUVA420P && dst[3] &&
        (c->unscaled_kind == UNSC_BGR24_YV12 || c->unscaled_kind == UNSC_YVU9_YV12 || c->unscaled_kind == UNSC_P4222PLANAR))
        for (int y = 0; y < srcSliceH; y++) memset(dst[3] + (ptrdiff_t)y * dstStride[3], 255, c->o.src_w);
    if (usePal(c->o.src_format)) update_palette(c, src[1]);   /* scale_internal, swscale.c:1088-1089 */
    switch (c->unscaled_kind) {
    case UNSC_PAL2RGB: return unscaled_pal2rgb(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_BAYER: return unscaled_bayer(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_YUV2RGB: return unscaled_yuv2rgb(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_P01X: return unscaled_p01x(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_8_P01X: return unscaled_8_p01x(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_PLANAR2NV12: return unscaled_planar2nv12(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_NV122PLANAR: return unscaled_nv122planar(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_PLANARCOPY: return unscaled_planarcopy(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_BGR24_YV12: return unscaled_bgr24_yv12(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_GBRP2PACKED: return unscaled_gbrp2packed(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_PLANAR2NV24: return unscaled_planar2nv24(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_NV242PLANAR: return unscaled_nv242planar(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_NV242YUV420: return unscaled_nv242yuv420(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_YVU9_YV12: return unscaled_yvu9_yv12(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_PACKED2GBRP: return unscaled_packed2gbrp(c, src, srcStride, 0, srcSliceH, dst, dstStride, 0);
    case UNSC_U8_TO_F32: /* uint_y_to_float_y_wrapper (:2095-2113): uint2float_lut[i] = (float)i * (1 / 255) (utils.c:1552-1556) */
        for (int y = 0; y < srcSliceH; y++) {
            const uint8_t *s = src[0] + (ptrdiff_t)y * srcStride[0]; float *d = (float *)(dst[0] + (ptrdiff_t)y * dstStride[0]);
            for (int x = 0; x < c->o.src_w; x++) d[x] = (float)s[x] * (1.0f / 255.0f);
        }
        return srcSliceH;
    case UNSC_F32_TO_U8: /* float_y_to_uint_y_wrapper (:2115-2135) */
        for (int y = 0; y < srcSliceH; y++) {
            const float *s = (const float *)(src[0] + (ptrdiff_t)y * srcStride[0]); uint8_t *d = dst[0] + (ptrdiff_t)y * dstStride[0];
            for (int x = 0; x < c->o.src_w; x++) d[x] = (uint8_t)clip_u8((int)lrintf(255.0f * s[x]));
        }
        return srcSliceH;
    case UNSC_YUV2MONO: return unscaled_yuv2mono(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_RGB30_TO_16: return unscaled_rgb30_to_16(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_RGB30_TO_GBRP: return unscaled_rgb30_to_gbrp(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_GBRP_TO_RGB30: return unscaled_gbrp_to_rgb30(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_PLANAR2P422: return unscaled_planar2p422(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_P4222PLANAR: return unscaled_p4222planar(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_RGB16SHUFFLE: return unscaled_rgb16shuffle(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_PACKED16_TO_GBRP16: return unscaled_packed16_gbrp16(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_GBRP16_TO_PACKED16: return unscaled_gbrp16_packed16(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_ALPHABLEND: return unscaled_alphablend(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    case UNSC_PLANARRGB_PLANARRGB: return unscaled_planarrgb_planarrgb(c, src, srcStride, 0, srcSliceH, dst, dstStride);
    }
    {   /* ff_swscale (swscale.c:333-334): "srcStride2[1] *= 1 << c->vChrDrop; srcStride2[2] *= 1 << c->vChrDrop;" -- the chroma planes are read
         * every 2^vChrDrop-th row (packed sources reach the same rows through `y << chrSrcVSub` in their readers) */
        const int drop = (c->o.flags & 0x30000) >> 16;
        int ss[4] = { srcStride[0], srcStride[1] << drop, srcStride[2] << drop, srcStride[3] };
        return main_path(c, src, ss, dst, dstStride);
    }
}

/* ------------------------------------------------------------------ */
/* introspection                                                       */
/* ------------------------------------------------------------------ */
int or_sws_get_filter(const OrSws *c, int which, const int16_t **filter, const int32_t **pos, int *count)
{
    switch (which) {
    case 0: *filter = c->hLumFilter; *pos = c->hLumFilterPos; *count = c->o.dst_w; return c->hLumFilterSize;
    case 1: *filter = c->hChrFilter; *pos = c->hChrFilterPos; *count = c->chrDstW; return c->hChrFilterSize;
    case 2: *filter = c->vLumFilter; *pos = c->vLumFilterPos; *count = c->o.dst_h; return c->vLumFilterSize;
    case 3: *filter = c->vChrFilter; *pos = c->vChrFilterPos; *count = c->chrDstH; return c->vChrFilterSize;
    }
    return 0;
}
int or_sws_path(const OrSws *c) { return c->cascade[0] ? 2 : c->unscaled_kind ? 1 : 0; }
const char *or_sws_path_name(const OrSws *c)
{
    static const char *n[] = { "main", "yuv2rgb_c", "planarToP01x", "planar8ToP01xle", "planarToNv12", "nv12ToPlanar", "planarCopy",
                               "rgbToRgb", "rgbToRgb", "packedCopy", "bgr24ToYv12", "planarRgbToRgb",
                               "planarToNv24", "nv24ToPlanar", "nv24ToYuv420", "yvu9ToYv12", "rgbToPlanarRgb", "rgbToRgb", "Rgb16ToPlanarRgb16", "planarRgb16ToRgb16", "yuv2rgb_c", "uint_y_to_float_y", "float_y_to_uint_y",
                               "planarToYuy2", "yuyvToPlanar",
                               "rgb16Shuffle", "Rgb16ToPlanarRgb16", "planarRgb16ToRgb16", "alphablendaway", "planarRgbToplanarRgb", "palToRgb", "bayer", "bswap_16bpc" };
    return c->cascade[0] ? "cascade" : n[c->unscaled_kind];
}
const int32_t *or_sws_rgb2yuv_table(const OrSws *c) { return c->rgb2yuv; }
void or_sws_yuv2rgb_coeffs(const OrSws *c, int out[6])
{
    out[0] = c->yuv2rgb_y_offset; out[1] = c->yuv2rgb_y_coeff; out[2] = c->yuv2rgb_v2r;
    out[3] = c->yuv2rgb_v2g; out[4] = c->yuv2rgb_u2g; out[5] = c->yuv2rgb_u2b;
}
void or_sws_range_consts(const OrSws *c, uint32_t coeff[2], int64_t offset[2], int *active)
{
    coeff[0] = c->lumCoeff; coeff[1] = c->chrCoeff; offset[0] = c->lumOffset; offset[1] = c->chrOffset;
    *active = c->range_active;
}
uint32_t or_sws_lut_rgb(const OrSws *c, int Y, int U, int V, int comp)
{
    int idx;
    if (!c->has_lut) return 0;
    if (comp == 0) idx = c->table_rV[V + HEADROOM];
    else if (comp == 1) idx = c->table_gU[U + HEADROOM] + c->table_gV[V + HEADROOM];
    else idx = c->table_bU[U + HEADROOM];
    return lut_at(c, idx + Y);
}
int or_sws_chroma_dims(const OrSws *c, int out[8])
{
    out[0] = c->chrSrcW; out[1] = c->chrSrcH; out[2] = c->chrDstW; out[3] = c->chrDstH;
    out[4] = c->chrSrcHSub; out[5] = c->chrSrcVSub; out[6] = c->chrDstHSub; out[7] = c->chrDstVSub;
    return 0;
}
