// Pixel-format descriptor rows and predicates for the formats on the hot path.
// Rows follow libavutil/pixdesc.c (:204 yuv420p, :261 rgb24, :567 nv12, :1440 yuv420p10le,
// :1656 yuv444p16le, :2352 p010le, :2529 gbrpf32le ...); predicates follow
// libswscale/swscale_internal.h:746-988.
#include "swsint.hpp"

namespace swship {

static const PixDesc g_descs[] = {
    { AV_PIX_FMT_YUV420P,  "yuv420p",  3, 1, 1, {{0,1,0,0,8},{1,1,0,0,8},{2,1,0,0,8},{0,0,0,0,0}}, PIXFLAG_PLANAR },
    { AV_PIX_FMT_YUVJ420P, "yuvj420p", 3, 1, 1, {{0,1,0,0,8},{1,1,0,0,8},{2,1,0,0,8},{0,0,0,0,0}}, PIXFLAG_PLANAR },
    { AV_PIX_FMT_YUV422P,  "yuv422p",  3, 1, 0, {{0,1,0,0,8},{1,1,0,0,8},{2,1,0,0,8},{0,0,0,0,0}}, PIXFLAG_PLANAR },
    { AV_PIX_FMT_YUV444P,  "yuv444p",  3, 0, 0, {{0,1,0,0,8},{1,1,0,0,8},{2,1,0,0,8},{0,0,0,0,0}}, PIXFLAG_PLANAR },
    { AV_PIX_FMT_NV12,     "nv12",     3, 1, 1, {{0,1,0,0,8},{1,2,0,0,8},{1,2,1,0,8},{0,0,0,0,0}}, PIXFLAG_PLANAR },
    { AV_PIX_FMT_NV21,     "nv21",     3, 1, 1, {{0,1,0,0,8},{1,2,1,0,8},{1,2,0,0,8},{0,0,0,0,0}}, PIXFLAG_PLANAR },
    { AV_PIX_FMT_YUV420P10LE, "yuv420p10le", 3, 1, 1, {{0,2,0,0,10},{1,2,0,0,10},{2,2,0,0,10},{0,0,0,0,0}}, PIXFLAG_PLANAR },
    { AV_PIX_FMT_YUV444P10LE, "yuv444p10le", 3, 0, 0, {{0,2,0,0,10},{1,2,0,0,10},{2,2,0,0,10},{0,0,0,0,0}}, PIXFLAG_PLANAR },
    { AV_PIX_FMT_YUV420P16LE, "yuv420p16le", 3, 1, 1, {{0,2,0,0,16},{1,2,0,0,16},{2,2,0,0,16},{0,0,0,0,0}}, PIXFLAG_PLANAR },
    { AV_PIX_FMT_YUV444P16LE, "yuv444p16le", 3, 0, 0, {{0,2,0,0,16},{1,2,0,0,16},{2,2,0,0,16},{0,0,0,0,0}}, PIXFLAG_PLANAR },
    { AV_PIX_FMT_P010LE,   "p010le",   3, 1, 1, {{0,2,0,6,10},{1,4,0,6,10},{1,4,2,6,10},{0,0,0,0,0}}, PIXFLAG_PLANAR },
#define PL8(F, N, LW, LH)     { F, N, 3, LW, LH, {{0,1,0,0,8},{1,1,0,0,8},{2,1,0,0,8},{0,0,0,0,0}}, PIXFLAG_PLANAR }
#define PLN(F, N, LW, LH, D)  { F, N, 3, LW, LH, {{0,2,0,0,D},{1,2,0,0,D},{2,2,0,0,D},{0,0,0,0,0}}, PIXFLAG_PLANAR }
#define SP8(F, N, LW, LH, UO) { F, N, 3, LW, LH, {{0,1,0,0,8},{1,2,UO,0,8},{1,2,1-(UO),0,8},{0,0,0,0,0}}, PIXFLAG_PLANAR }
#define SPN(F, N, LW, LH, D)  { F, N, 3, LW, LH, {{0,2,0,16-(D),D},{1,4,0,16-(D),D},{1,4,2,16-(D),D},{0,0,0,0,0}}, PIXFLAG_PLANAR }
    { AV_PIX_FMT_RGB48LE,  "rgb48le",  3, 0, 0, {{0,6,0,0,16},{0,6,2,0,16},{0,6,4,0,16},{0,0,0,0,0}}, PIXFLAG_RGB },
    { AV_PIX_FMT_BGR48LE,  "bgr48le",  3, 0, 0, {{0,6,4,0,16},{0,6,2,0,16},{0,6,0,0,16},{0,0,0,0,0}}, PIXFLAG_RGB },
    { AV_PIX_FMT_RGBA64LE, "rgba64le", 4, 0, 0, {{0,8,0,0,16},{0,8,2,0,16},{0,8,4,0,16},{0,8,6,0,16}}, PIXFLAG_RGB | PIXFLAG_ALPHA },
    { AV_PIX_FMT_BGRA64LE, "bgra64le", 4, 0, 0, {{0,8,4,0,16},{0,8,2,0,16},{0,8,0,0,16},{0,8,6,0,16}}, PIXFLAG_RGB | PIXFLAG_ALPHA },
    { AV_PIX_FMT_YUYV422, "yuyv422", 3, 1, 0, {{0,2,0,0,8},{0,4,1,0,8},{0,4,3,0,8},{0,0,0,0,0}}, 0 },
    { AV_PIX_FMT_UYVY422, "uyvy422", 3, 1, 0, {{0,2,1,0,8},{0,4,0,0,8},{0,4,2,0,8},{0,0,0,0,0}}, 0 },
    { AV_PIX_FMT_YVYU422, "yvyu422", 3, 1, 0, {{0,2,0,0,8},{0,4,3,0,8},{0,4,1,0,8},{0,0,0,0,0}}, 0 },
#define PLA(F, N, LW, LH)     { F, N, 4, LW, LH, {{0,1,0,0,8},{1,1,0,0,8},{2,1,0,0,8},{3,1,0,0,8}}, PIXFLAG_PLANAR | PIXFLAG_ALPHA }
#define PLAN_(F, N, LW, LH, D) { F, N, 4, LW, LH, {{0,2,0,0,D},{1,2,0,0,D},{2,2,0,0,D},{3,2,0,0,D}}, PIXFLAG_PLANAR | PIXFLAG_ALPHA }
    PLAN_(AV_PIX_FMT_YUVA420P9LE, "yuva420p9le", 1, 1, 9),
    PLAN_(AV_PIX_FMT_YUVA420P10LE, "yuva420p10le", 1, 1, 10),
    PLAN_(AV_PIX_FMT_YUVA420P16LE, "yuva420p16le", 1, 1, 16),
    PLAN_(AV_PIX_FMT_YUVA422P9LE, "yuva422p9le", 1, 0, 9),
    PLAN_(AV_PIX_FMT_YUVA422P10LE, "yuva422p10le", 1, 0, 10),
    PLAN_(AV_PIX_FMT_YUVA422P12LE, "yuva422p12le", 1, 0, 12),
    PLAN_(AV_PIX_FMT_YUVA422P16LE, "yuva422p16le", 1, 0, 16),
    PLAN_(AV_PIX_FMT_YUVA444P9LE, "yuva444p9le", 0, 0, 9),
    PLAN_(AV_PIX_FMT_YUVA444P10LE, "yuva444p10le", 0, 0, 10),
    PLAN_(AV_PIX_FMT_YUVA444P12LE, "yuva444p12le", 0, 0, 12),
    PLAN_(AV_PIX_FMT_YUVA444P16LE, "yuva444p16le", 0, 0, 16),
    PLA(AV_PIX_FMT_YUVA420P, "yuva420p", 1, 1), PLA(AV_PIX_FMT_YUVA422P, "yuva422p", 1, 0), PLA(AV_PIX_FMT_YUVA444P, "yuva444p", 0, 0),
    PL8(AV_PIX_FMT_YUV410P, "yuv410p", 2, 2), PL8(AV_PIX_FMT_YUV411P, "yuv411p", 2, 0), PL8(AV_PIX_FMT_YUV440P, "yuv440p", 0, 1),
    PL8(AV_PIX_FMT_YUVJ422P, "yuvj422p", 1, 0), PL8(AV_PIX_FMT_YUVJ444P, "yuvj444p", 0, 0), PL8(AV_PIX_FMT_YUVJ440P, "yuvj440p", 0, 1),
    PLN(AV_PIX_FMT_YUV420P9LE, "yuv420p9le", 1, 1, 9), PLN(AV_PIX_FMT_YUV422P9LE, "yuv422p9le", 1, 0, 9),
    PLN(AV_PIX_FMT_YUV444P9LE, "yuv444p9le", 0, 0, 9),
    PLN(AV_PIX_FMT_YUV422P10LE, "yuv422p10le", 1, 0, 10), PLN(AV_PIX_FMT_YUV440P10LE, "yuv440p10le", 0, 1, 10),
    PLN(AV_PIX_FMT_YUV420P12LE, "yuv420p12le", 1, 1, 12), PLN(AV_PIX_FMT_YUV422P12LE, "yuv422p12le", 1, 0, 12),
    PLN(AV_PIX_FMT_YUV444P12LE, "yuv444p12le", 0, 0, 12), PLN(AV_PIX_FMT_YUV440P12LE, "yuv440p12le", 0, 1, 12),
    PLN(AV_PIX_FMT_YUV420P14LE, "yuv420p14le", 1, 1, 14), PLN(AV_PIX_FMT_YUV422P14LE, "yuv422p14le", 1, 0, 14),
    PLN(AV_PIX_FMT_YUV444P14LE, "yuv444p14le", 0, 0, 14), PLN(AV_PIX_FMT_YUV422P16LE, "yuv422p16le", 1, 0, 16),
    SP8(AV_PIX_FMT_NV16, "nv16", 1, 0, 0), SP8(AV_PIX_FMT_NV24, "nv24", 0, 0, 0), SP8(AV_PIX_FMT_NV42, "nv42", 0, 0, 1),
    SPN(AV_PIX_FMT_P210LE, "p210le", 1, 0, 10), SPN(AV_PIX_FMT_P410LE, "p410le", 0, 0, 10),
    SPN(AV_PIX_FMT_P012LE, "p012le", 1, 1, 12), SPN(AV_PIX_FMT_P212LE, "p212le", 1, 0, 12), SPN(AV_PIX_FMT_P412LE, "p412le", 0, 0, 12),
    SPN(AV_PIX_FMT_P016LE, "p016le", 1, 1, 16), SPN(AV_PIX_FMT_P216LE, "p216le", 1, 0, 16), SPN(AV_PIX_FMT_P416LE, "p416le", 0, 0, 16),
    { AV_PIX_FMT_RGB24,    "rgb24",    3, 0, 0, {{0,3,0,0,8},{0,3,1,0,8},{0,3,2,0,8},{0,0,0,0,0}}, PIXFLAG_RGB },
    { AV_PIX_FMT_BGR24,    "bgr24",    3, 0, 0, {{0,3,2,0,8},{0,3,1,0,8},{0,3,0,0,8},{0,0,0,0,0}}, PIXFLAG_RGB },
    { AV_PIX_FMT_ARGB,     "argb",     4, 0, 0, {{0,4,1,0,8},{0,4,2,0,8},{0,4,3,0,8},{0,4,0,0,8}}, PIXFLAG_RGB | PIXFLAG_ALPHA },
    { AV_PIX_FMT_RGBA,     "rgba",     4, 0, 0, {{0,4,0,0,8},{0,4,1,0,8},{0,4,2,0,8},{0,4,3,0,8}}, PIXFLAG_RGB | PIXFLAG_ALPHA },
    { AV_PIX_FMT_ABGR,     "abgr",     4, 0, 0, {{0,4,3,0,8},{0,4,2,0,8},{0,4,1,0,8},{0,4,0,0,8}}, PIXFLAG_RGB | PIXFLAG_ALPHA },
    { AV_PIX_FMT_BGRA,     "bgra",     4, 0, 0, {{0,4,2,0,8},{0,4,1,0,8},{0,4,0,0,8},{0,4,3,0,8}}, PIXFLAG_RGB | PIXFLAG_ALPHA },
    { AV_PIX_FMT_0RGB,     "0rgb",     3, 0, 0, {{0,4,1,0,8},{0,4,2,0,8},{0,4,3,0,8},{0,0,0,0,0}}, PIXFLAG_RGB },
    { AV_PIX_FMT_RGB0,     "rgb0",     3, 0, 0, {{0,4,0,0,8},{0,4,1,0,8},{0,4,2,0,8},{0,0,0,0,0}}, PIXFLAG_RGB },
    { AV_PIX_FMT_0BGR,     "0bgr",     3, 0, 0, {{0,4,3,0,8},{0,4,2,0,8},{0,4,1,0,8},{0,0,0,0,0}}, PIXFLAG_RGB },
    { AV_PIX_FMT_BGR0,     "bgr0",     3, 0, 0, {{0,4,2,0,8},{0,4,1,0,8},{0,4,0,0,8},{0,0,0,0,0}}, PIXFLAG_RGB },
    { AV_PIX_FMT_GBRP,     "gbrp",     3, 0, 0, {{2,1,0,0,8},{0,1,0,0,8},{1,1,0,0,8},{0,0,0,0,0}}, PIXFLAG_PLANAR | PIXFLAG_RGB },
    { AV_PIX_FMT_GRAY8, "gray", 1, 0, 0, {{0,1,0,0,8},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0}}, 0 },
#define GRAYN(F, N, D) { F, N, 1, 0, 0, {{0,2,0,0,D},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0}}, 0 }
    GRAYN(AV_PIX_FMT_GRAY9LE, "gray9le", 9), GRAYN(AV_PIX_FMT_GRAY10LE, "gray10le", 10), GRAYN(AV_PIX_FMT_GRAY12LE, "gray12le", 12),
    GRAYN(AV_PIX_FMT_GRAY14LE, "gray14le", 14), GRAYN(AV_PIX_FMT_GRAY16LE, "gray16le", 16),
#define GBRN(F, N, D) { F, N, 3, 0, 0, {{2,2,0,0,D},{0,2,0,0,D},{1,2,0,0,D},{0,0,0,0,0}}, PIXFLAG_PLANAR | PIXFLAG_RGB }
    GBRN(AV_PIX_FMT_GBRP9LE, "gbrp9le", 9), GBRN(AV_PIX_FMT_GBRP10LE, "gbrp10le", 10), GBRN(AV_PIX_FMT_GBRP12LE, "gbrp12le", 12),
    GBRN(AV_PIX_FMT_GBRP14LE, "gbrp14le", 14), GBRN(AV_PIX_FMT_GBRP16LE, "gbrp16le", 16),
    { AV_PIX_FMT_GBRPF32LE,"gbrpf32le",3, 0, 0, {{2,4,0,0,32},{0,4,0,0,32},{1,4,0,0,32},{0,0,0,0,0}}, PIXFLAG_PLANAR | PIXFLAG_RGB | PIXFLAG_FLOAT },
    // planar RGB + alpha plane (pixdesc.c: gbrap, gbrap10/12/14/16, gbrapf32)
    { AV_PIX_FMT_GBRAP, "gbrap", 4, 0, 0, {{2,1,0,0,8},{0,1,0,0,8},{1,1,0,0,8},{3,1,0,0,8}}, PIXFLAG_PLANAR | PIXFLAG_RGB | PIXFLAG_ALPHA },
#define GBRAN(F, N, D) { F, N, 4, 0, 0, {{2,2,0,0,D},{0,2,0,0,D},{1,2,0,0,D},{3,2,0,0,D}}, PIXFLAG_PLANAR | PIXFLAG_RGB | PIXFLAG_ALPHA }
    GBRAN(AV_PIX_FMT_GBRAP10LE, "gbrap10le", 10), GBRAN(AV_PIX_FMT_GBRAP12LE, "gbrap12le", 12), GBRAN(AV_PIX_FMT_GBRAP14LE, "gbrap14le", 14),
    GBRAN(AV_PIX_FMT_GBRAP16LE, "gbrap16le", 16),
    { AV_PIX_FMT_GBRAPF32LE, "gbrapf32le", 4, 0, 0, {{2,4,0,0,32},{0,4,0,0,32},{1,4,0,0,32},{3,4,0,0,32}}, PIXFLAG_PLANAR | PIXFLAG_RGB | PIXFLAG_FLOAT | PIXFLAG_ALPHA },
    // packed YUV with 10..16-bit samples (pixdesc.c:239-262, :2327-2350, :2973-3090, :3248-3270); X fields are not components
    { AV_PIX_FMT_Y210LE, "y210le", 3, 1, 0, {{0,4,0,6,10},{0,8,2,6,10},{0,8,6,6,10},{0,0,0,0,0}}, 0 },
    { AV_PIX_FMT_Y212LE, "y212le", 3, 1, 0, {{0,4,0,4,12},{0,8,2,4,12},{0,8,6,4,12},{0,0,0,0,0}}, 0 },
    { AV_PIX_FMT_Y216LE, "y216le", 3, 1, 0, {{0,4,0,0,16},{0,8,2,0,16},{0,8,6,0,16},{0,0,0,0,0}}, 0 },
    { AV_PIX_FMT_XV30LE, "xv30le", 3, 0, 0, {{0,4,1,2,10},{0,4,0,0,10},{0,4,2,4,10},{0,0,0,0,0}}, 0 },
    { AV_PIX_FMT_V30XLE, "v30xle", 3, 0, 0, {{0,4,1,4,10},{0,4,0,2,10},{0,4,2,6,10},{0,0,0,0,0}}, 0 },
    { AV_PIX_FMT_XV36LE, "xv36le", 3, 0, 0, {{0,8,2,4,12},{0,8,0,4,12},{0,8,4,4,12},{0,0,0,0,0}}, 0 },
    { AV_PIX_FMT_XV48LE, "xv48le", 3, 0, 0, {{0,8,2,0,16},{0,8,0,0,16},{0,8,4,0,16},{0,0,0,0,0}}, 0 },
    { AV_PIX_FMT_AYUV64LE, "ayuv64le", 4, 0, 0, {{0,8,2,0,16},{0,8,4,0,16},{0,8,6,0,16},{0,8,0,0,16}}, PIXFLAG_ALPHA },
    // packed 4:4:4, 8 bit (libavutil/pixdesc.c:2290-2324, :2895-2917)
    { AV_PIX_FMT_VYU444, "vyu444", 3, 0, 0, {{0,3,1,0,8},{0,3,2,0,8},{0,3,0,0,8},{0,0,0,0,0}}, 0 },
    { AV_PIX_FMT_UYVA, "uyva", 4, 0, 0, {{0,4,1,0,8},{0,4,0,0,8},{0,4,2,0,8},{0,4,3,0,8}}, PIXFLAG_ALPHA },
    { AV_PIX_FMT_AYUV, "ayuv", 4, 0, 0, {{0,4,1,0,8},{0,4,2,0,8},{0,4,3,0,8},{0,4,0,0,8}}, PIXFLAG_ALPHA },
    { AV_PIX_FMT_VUYA, "vuya", 4, 0, 0, {{0,4,2,0,8},{0,4,1,0,8},{0,4,0,0,8},{0,4,3,0,8}}, PIXFLAG_ALPHA },
    { AV_PIX_FMT_VUYX, "vuyx", 4, 0, 0, {{0,4,2,0,8},{0,4,1,0,8},{0,4,0,0,8},{0,4,3,0,8}}, 0 },
    // planar 4:4:4 with the samples in the high bits of each 16-bit word
    { AV_PIX_FMT_YUV444P10MSBLE, "yuv444p10msble", 3, 0, 0, {{0,2,0,6,10},{1,2,0,6,10},{2,2,0,6,10},{0,0,0,0,0}}, PIXFLAG_PLANAR },
    { AV_PIX_FMT_YUV444P12MSBLE, "yuv444p12msble", 3, 0, 0, {{0,2,0,4,12},{1,2,0,4,12},{2,2,0,4,12},{0,0,0,0,0}}, PIXFLAG_PLANAR },
    // 16 bits per pixel packed RGB (libavutil/pixdesc.c:1229-1420)
    PL8(AV_PIX_FMT_YUVJ411P, "yuvj411p", 2, 0),
    { AV_PIX_FMT_NV20LE, "nv20le", 3, 1, 0, {{0,2,0,0,10},{1,4,0,0,10},{1,4,2,0,10},{0,0,0,0,0}}, PIXFLAG_PLANAR },
    { AV_PIX_FMT_GBRP10MSBLE, "gbrp10msble", 3, 0, 0, {{2,2,0,6,10},{0,2,0,6,10},{1,2,0,6,10},{0,0,0,0,0}}, PIXFLAG_PLANAR | PIXFLAG_RGB },
    { AV_PIX_FMT_GBRP12MSBLE, "gbrp12msble", 3, 0, 0, {{2,2,0,4,12},{0,2,0,4,12},{1,2,0,4,12},{0,0,0,0,0}}, PIXFLAG_PLANAR | PIXFLAG_RGB },
    { AV_PIX_FMT_YA8, "ya8", 2, 0, 0, {{0,2,0,0,8},{0,2,1,0,8},{0,0,0,0,0},{0,0,0,0,0}}, PIXFLAG_ALPHA },
    { AV_PIX_FMT_YA16LE, "ya16le", 2, 0, 0, {{0,4,0,0,16},{0,4,2,0,16},{0,0,0,0,0},{0,0,0,0,0}}, PIXFLAG_ALPHA },
    { AV_PIX_FMT_GRAYF32LE, "grayf32le", 1, 0, 0, {{0,4,0,0,32},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0}}, PIXFLAG_FLOAT },
    { AV_PIX_FMT_MONOWHITE, "monow", 1, 0, 0, {{0,1,0,0,1},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0}}, PIXFLAG_RGB },   // 1 bit per pixel, MSB first; isAnyRGB() counts them in
    { AV_PIX_FMT_MONOBLACK, "monob", 1, 0, 0, {{0,1,0,7,1},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0}}, PIXFLAG_RGB },
    // bayer mosaics (libavutil/pixdesc.c:2149-2250): one plane of 8- or 16-bit samples; inputs only
#define BAYER8(F, N)  { F, N, 3, 0, 0, {{0,1,0,0,2},{0,1,0,0,4},{0,1,0,0,2},{0,0,0,0,0}}, PIXFLAG_RGB | PIXFLAG_BAYER }
#define BAYER16(F, N) { F, N, 3, 0, 0, {{0,2,0,0,4},{0,2,0,0,8},{0,2,0,0,4},{0,0,0,0,0}}, PIXFLAG_RGB | PIXFLAG_BAYER }
    BAYER8(AV_PIX_FMT_BAYER_BGGR8, "bayer_bggr8"), BAYER8(AV_PIX_FMT_BAYER_RGGB8, "bayer_rggb8"), BAYER8(AV_PIX_FMT_BAYER_GBRG8, "bayer_gbrg8"), BAYER8(AV_PIX_FMT_BAYER_GRBG8, "bayer_grbg8"),
    BAYER16(AV_PIX_FMT_BAYER_BGGR16LE, "bayer_bggr16le"), BAYER16(AV_PIX_FMT_BAYER_RGGB16LE, "bayer_rggb16le"), BAYER16(AV_PIX_FMT_BAYER_GBRG16LE, "bayer_gbrg16le"),
    BAYER16(AV_PIX_FMT_BAYER_GRBG16LE, "bayer_grbg16le"),
    { AV_PIX_FMT_PAL8, "pal8", 1, 0, 0, {{0,1,0,0,8},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0}}, PIXFLAG_PAL | PIXFLAG_ALPHA },   // one index plane + the palette in data[1]; input only
    // float and half-float sources (libavutil/pixdesc.c:2583-2717, :2932-2971, :3108-3119) and the packed 4:1:1 source (:484-494): inputs only
    { AV_PIX_FMT_RGBF32LE, "rgbf32le", 3, 0, 0, {{0,12,0,0,32},{0,12,4,0,32},{0,12,8,0,32},{0,0,0,0,0}}, PIXFLAG_RGB | PIXFLAG_FLOAT },
    { AV_PIX_FMT_RGBF16LE, "rgbf16le", 3, 0, 0, {{0,6,0,0,16},{0,6,2,0,16},{0,6,4,0,16},{0,0,0,0,0}}, PIXFLAG_RGB | PIXFLAG_FLOAT },
    { AV_PIX_FMT_RGBAF16LE, "rgbaf16le", 4, 0, 0, {{0,8,0,0,16},{0,8,2,0,16},{0,8,4,0,16},{0,8,6,0,16}}, PIXFLAG_RGB | PIXFLAG_FLOAT | PIXFLAG_ALPHA },
    { AV_PIX_FMT_GRAYF16LE, "grayf16le", 1, 0, 0, {{0,2,0,0,16},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0}}, PIXFLAG_FLOAT },
    { AV_PIX_FMT_YAF32LE, "yaf32le", 2, 0, 0, {{0,8,0,0,32},{0,8,4,0,32},{0,0,0,0,0},{0,0,0,0,0}}, PIXFLAG_FLOAT | PIXFLAG_ALPHA },
    { AV_PIX_FMT_YAF16LE, "yaf16le", 2, 0, 0, {{0,4,0,0,16},{0,4,2,0,16},{0,0,0,0,0},{0,0,0,0,0}}, PIXFLAG_FLOAT | PIXFLAG_ALPHA },
    { AV_PIX_FMT_GBRPF16LE, "gbrpf16le", 3, 0, 0, {{2,2,0,0,16},{0,2,0,0,16},{1,2,0,0,16},{0,0,0,0,0}}, PIXFLAG_PLANAR | PIXFLAG_RGB | PIXFLAG_FLOAT },
    { AV_PIX_FMT_GBRAPF16LE, "gbrapf16le", 4, 0, 0, {{2,2,0,0,16},{0,2,0,0,16},{1,2,0,0,16},{3,2,0,0,16}}, PIXFLAG_PLANAR | PIXFLAG_RGB | PIXFLAG_FLOAT | PIXFLAG_ALPHA },
    { AV_PIX_FMT_UYYVYY411, "uyyvyy411", 3, 2, 0, {{0,4,1,0,8},{0,6,0,0,8},{0,6,3,0,8},{0,0,0,0,0}}, 0 },
    // 8 / 4 bpp RGB (libavutil/pixdesc.c:495-566), destinations only; rgb4 / bgr4 are bit streams (step and offset count bits)
    { AV_PIX_FMT_BGR8, "bgr8", 3, 0, 0, {{0,1,0,0,3},{0,1,0,3,3},{0,1,0,6,2},{0,0,0,0,0}}, PIXFLAG_RGB },
    { AV_PIX_FMT_BGR4, "bgr4", 3, 0, 0, {{0,4,3,0,1},{0,4,1,0,2},{0,4,0,0,1},{0,0,0,0,0}}, PIXFLAG_RGB },
    { AV_PIX_FMT_BGR4_BYTE, "bgr4_byte", 3, 0, 0, {{0,1,0,0,1},{0,1,0,1,2},{0,1,0,3,1},{0,0,0,0,0}}, PIXFLAG_RGB },
    { AV_PIX_FMT_RGB8, "rgb8", 3, 0, 0, {{0,1,0,5,3},{0,1,0,2,3},{0,1,0,0,2},{0,0,0,0,0}}, PIXFLAG_RGB },
    { AV_PIX_FMT_RGB4, "rgb4", 3, 0, 0, {{0,4,0,0,1},{0,4,1,0,2},{0,4,3,0,1},{0,0,0,0,0}}, PIXFLAG_RGB },
    { AV_PIX_FMT_RGB4_BYTE, "rgb4_byte", 3, 0, 0, {{0,1,0,3,1},{0,1,0,1,2},{0,1,0,0,1},{0,0,0,0,0}}, PIXFLAG_RGB },
    { AV_PIX_FMT_XYZ12LE, "xyz12le", 3, 0, 0, {{0,6,0,4,12},{0,6,2,4,12},{0,6,4,4,12},{0,0,0,0,0}}, 0 },   // only ever seen before the handle_xyz() aliasing
    { AV_PIX_FMT_X2RGB10LE, "x2rgb10le", 3, 0, 0, {{0,4,2,4,10},{0,4,1,2,10},{0,4,0,0,10},{0,0,0,0,0}}, PIXFLAG_RGB },
    { AV_PIX_FMT_X2BGR10LE, "x2bgr10le", 3, 0, 0, {{0,4,0,0,10},{0,4,1,2,10},{0,4,2,4,10},{0,0,0,0,0}}, PIXFLAG_RGB },
    { AV_PIX_FMT_RGB565LE, "rgb565le", 3, 0, 0, {{0,2,1,3,5},{0,2,0,5,6},{0,2,0,0,5},{0,0,0,0,0}}, PIXFLAG_RGB },
    { AV_PIX_FMT_RGB555LE, "rgb555le", 3, 0, 0, {{0,2,1,2,5},{0,2,0,5,5},{0,2,0,0,5},{0,0,0,0,0}}, PIXFLAG_RGB },
    { AV_PIX_FMT_RGB444LE, "rgb444le", 3, 0, 0, {{0,2,1,0,4},{0,2,0,4,4},{0,2,0,0,4},{0,0,0,0,0}}, PIXFLAG_RGB },
    { AV_PIX_FMT_BGR565LE, "bgr565le", 3, 0, 0, {{0,2,0,0,5},{0,2,0,5,6},{0,2,1,3,5},{0,0,0,0,0}}, PIXFLAG_RGB },
    { AV_PIX_FMT_BGR555LE, "bgr555le", 3, 0, 0, {{0,2,0,0,5},{0,2,0,5,5},{0,2,1,2,5},{0,0,0,0,0}}, PIXFLAG_RGB },
    { AV_PIX_FMT_BGR444LE, "bgr444le", 3, 0, 0, {{0,2,0,0,4},{0,2,0,4,4},{0,2,1,0,4},{0,0,0,0,0}}, PIXFLAG_RGB },
};

const PixDesc *pix_desc(int fmt)
{
    for (const PixDesc &d : g_descs)
        if (d.fmt == fmt) return &d;
    const int twin = pix_be_twin(fmt);   // a big-endian format has the layout of its little-endian twin
    if (twin >= 0)
        for (const PixDesc &d : g_descs)
            if (d.fmt == twin) return &d;
    return nullptr;
}

int pix_bits_per_pixel(const PixDesc *d) // av_get_bits_per_pixel
{
    int bits = 0;
    const int log2_pixels = d->log2_chroma_w + d->log2_chroma_h;
    for (int c = 0; c < d->nb_components; c++)
        bits += d->comp[c].depth << ((c == 1 || c == 2) ? 0 : log2_pixels);
    return bits >> log2_pixels;
}

int pix_nb_planes(const PixDesc *d)
{
    int n = 0;
    for (int c = 0; c < d->nb_components; c++)
        if (d->comp[c].plane + 1 > n) n = d->comp[c].plane + 1;
    if (d->flags & PIXFLAG_PAL) n = 2;   // data[1]: the palette
    return n;
}

bool is16BPS(int f) { return pix_desc(f)->comp[0].depth == 16; }
bool isNBPS(int f) { const int d = pix_desc(f)->comp[0].depth; return d >= 9 && d <= 14; }
bool isYUV(int f) { const PixDesc *d = pix_desc(f); return !(d->flags & PIXFLAG_RGB) && d->nb_components >= 2; }
bool isPlanarYUV(int f) { return (pix_desc(f)->flags & PIXFLAG_PLANAR) && isYUV(f); }
bool isSemiPlanarYUV(int f) { const PixDesc *d = pix_desc(f); return isPlanarYUV(f) && d->comp[1].plane == d->comp[2].plane; }
bool isAnyRGB(int f) { return (pix_desc(f)->flags & PIXFLAG_RGB) != 0; }
static bool isMonoFmt(int f) { return f == AV_PIX_FMT_MONOWHITE || f == AV_PIX_FMT_MONOBLACK; }
bool isGray(int f) { return pix_desc(f)->nb_components <= 2 && !isMonoFmt(f) && !(pix_desc(f)->flags & PIXFLAG_PAL); }   // swscale_internal.h:805-815
bool isFloatFmt(int f) { return (pix_desc(f)->flags & PIXFLAG_FLOAT) != 0; }
bool isBayerFmt(int f) { return (pix_desc(f)->flags & PIXFLAG_BAYER) != 0; }
bool isFloat16Fmt(int f) { const PixDesc *d = pix_desc(f); return (d->flags & PIXFLAG_FLOAT) && d->comp[0].depth == 16; }   // swscale_internal.h:890-895
bool isALPHA(int f) { return (pix_desc(f)->flags & PIXFLAG_ALPHA) != 0; }
bool isPlanarRGB(int f) { return (pix_desc(f)->flags & (PIXFLAG_PLANAR | PIXFLAG_RGB)) == (PIXFLAG_PLANAR | PIXFLAG_RGB); }
bool isPackedFmt(int f) { const PixDesc *d = pix_desc(f); return (d->nb_components >= 2 && !(d->flags & PIXFLAG_PLANAR)) || isMonoFmt(f) || f == AV_PIX_FMT_PAL8; }   // swscale_internal.h:906-914
bool isPlanarFmt(int f) { const PixDesc *d = pix_desc(f); return d->nb_components >= 2 && (d->flags & PIXFLAG_PLANAR); }
bool isSwappedChroma(int f)
{
    const PixDesc *d = pix_desc(f);
    if (!isYUV(f) || d->nb_components < 3) return false;
    if (!isPlanarYUV(f) || isSemiPlanarYUV(f)) return d->comp[1].offset > d->comp[2].offset;
    return d->comp[1].plane > d->comp[2].plane;
}
bool isDataInHighBits(int f)
{
    const PixDesc *d = pix_desc(f);
    for (int i = 0; i < d->nb_components; i++) {
        if (!d->comp[i].shift) return false;
        if ((d->comp[i].shift + d->comp[i].depth) & 7) return false;
    }
    return true;
}

// big-endian formats are converted through their little-endian twin (the reference's BE readers / writers are the LE ones behind
// AV_RB16 / AV_WB16: input.c:608-629, the output_pixel macros of output.c); returns the twin or -1
int pix_be_twin(int fmt)
{
    static const int pairs[][2] = {
    { AV_PIX_FMT_XV36BE, AV_PIX_FMT_XV36LE }, { AV_PIX_FMT_XV48BE, AV_PIX_FMT_XV48LE }, { AV_PIX_FMT_AYUV64BE, AV_PIX_FMT_AYUV64LE },
    { AV_PIX_FMT_YUVA420P9BE, AV_PIX_FMT_YUVA420P9LE }, { AV_PIX_FMT_YUVA420P10BE, AV_PIX_FMT_YUVA420P10LE }, { AV_PIX_FMT_YUVA420P16BE, AV_PIX_FMT_YUVA420P16LE }, { AV_PIX_FMT_YUVA422P9BE, AV_PIX_FMT_YUVA422P9LE }, { AV_PIX_FMT_YUVA422P10BE, AV_PIX_FMT_YUVA422P10LE }, { AV_PIX_FMT_YUVA422P12BE, AV_PIX_FMT_YUVA422P12LE }, { AV_PIX_FMT_YUVA422P16BE, AV_PIX_FMT_YUVA422P16LE }, { AV_PIX_FMT_YUVA444P9BE, AV_PIX_FMT_YUVA444P9LE }, { AV_PIX_FMT_YUVA444P10BE, AV_PIX_FMT_YUVA444P10LE }, { AV_PIX_FMT_YUVA444P12BE, AV_PIX_FMT_YUVA444P12LE }, { AV_PIX_FMT_YUVA444P16BE, AV_PIX_FMT_YUVA444P16LE },
    { AV_PIX_FMT_BAYER_BGGR16BE, AV_PIX_FMT_BAYER_BGGR16LE }, { AV_PIX_FMT_BAYER_RGGB16BE, AV_PIX_FMT_BAYER_RGGB16LE }, { AV_PIX_FMT_BAYER_GBRG16BE, AV_PIX_FMT_BAYER_GBRG16LE },
    { AV_PIX_FMT_BAYER_GRBG16BE, AV_PIX_FMT_BAYER_GRBG16LE },
    { AV_PIX_FMT_RGBF32BE, AV_PIX_FMT_RGBF32LE }, { AV_PIX_FMT_RGBF16BE, AV_PIX_FMT_RGBF16LE }, { AV_PIX_FMT_RGBAF16BE, AV_PIX_FMT_RGBAF16LE }, { AV_PIX_FMT_GRAYF16BE, AV_PIX_FMT_GRAYF16LE },
    { AV_PIX_FMT_YAF32BE, AV_PIX_FMT_YAF32LE }, { AV_PIX_FMT_YAF16BE, AV_PIX_FMT_YAF16LE }, { AV_PIX_FMT_GBRPF16BE, AV_PIX_FMT_GBRPF16LE }, { AV_PIX_FMT_GBRAPF16BE, AV_PIX_FMT_GBRAPF16LE },
    { AV_PIX_FMT_YA16BE, AV_PIX_FMT_YA16LE }, { AV_PIX_FMT_GRAYF32BE, AV_PIX_FMT_GRAYF32LE }, { AV_PIX_FMT_XYZ12BE, AV_PIX_FMT_XYZ12LE }, { AV_PIX_FMT_NV20BE, AV_PIX_FMT_NV20LE }, { AV_PIX_FMT_GBRP10MSBBE, AV_PIX_FMT_GBRP10MSBLE }, { AV_PIX_FMT_GBRP12MSBBE, AV_PIX_FMT_GBRP12MSBLE },
    { AV_PIX_FMT_YUV444P10MSBBE, AV_PIX_FMT_YUV444P10MSBLE }, { AV_PIX_FMT_YUV444P12MSBBE, AV_PIX_FMT_YUV444P12MSBLE },
    { AV_PIX_FMT_RGB565BE, AV_PIX_FMT_RGB565LE }, { AV_PIX_FMT_RGB555BE, AV_PIX_FMT_RGB555LE }, { AV_PIX_FMT_RGB444BE, AV_PIX_FMT_RGB444LE },
    { AV_PIX_FMT_BGR565BE, AV_PIX_FMT_BGR565LE }, { AV_PIX_FMT_BGR555BE, AV_PIX_FMT_BGR555LE }, { AV_PIX_FMT_BGR444BE, AV_PIX_FMT_BGR444LE },
    { AV_PIX_FMT_YUV420P9BE, AV_PIX_FMT_YUV420P9LE },
    { AV_PIX_FMT_YUV420P10BE, AV_PIX_FMT_YUV420P10LE },
    { AV_PIX_FMT_YUV420P12BE, AV_PIX_FMT_YUV420P12LE },
    { AV_PIX_FMT_YUV420P14BE, AV_PIX_FMT_YUV420P14LE },
    { AV_PIX_FMT_YUV420P16BE, AV_PIX_FMT_YUV420P16LE },
    { AV_PIX_FMT_YUV422P9BE, AV_PIX_FMT_YUV422P9LE },
    { AV_PIX_FMT_YUV422P10BE, AV_PIX_FMT_YUV422P10LE },
    { AV_PIX_FMT_YUV422P12BE, AV_PIX_FMT_YUV422P12LE },
    { AV_PIX_FMT_YUV422P14BE, AV_PIX_FMT_YUV422P14LE },
    { AV_PIX_FMT_YUV422P16BE, AV_PIX_FMT_YUV422P16LE },
    { AV_PIX_FMT_YUV444P9BE, AV_PIX_FMT_YUV444P9LE },
    { AV_PIX_FMT_YUV444P10BE, AV_PIX_FMT_YUV444P10LE },
    { AV_PIX_FMT_YUV444P12BE, AV_PIX_FMT_YUV444P12LE },
    { AV_PIX_FMT_YUV444P14BE, AV_PIX_FMT_YUV444P14LE },
    { AV_PIX_FMT_YUV444P16BE, AV_PIX_FMT_YUV444P16LE },
    { AV_PIX_FMT_YUV440P10BE, AV_PIX_FMT_YUV440P10LE },
    { AV_PIX_FMT_YUV440P12BE, AV_PIX_FMT_YUV440P12LE },
    { AV_PIX_FMT_GRAY9BE, AV_PIX_FMT_GRAY9LE },
    { AV_PIX_FMT_GRAY10BE, AV_PIX_FMT_GRAY10LE },
    { AV_PIX_FMT_GRAY12BE, AV_PIX_FMT_GRAY12LE },
    { AV_PIX_FMT_GRAY14BE, AV_PIX_FMT_GRAY14LE },
    { AV_PIX_FMT_GRAY16BE, AV_PIX_FMT_GRAY16LE },
    { AV_PIX_FMT_GBRP9BE, AV_PIX_FMT_GBRP9LE },
    { AV_PIX_FMT_GBRP10BE, AV_PIX_FMT_GBRP10LE },
    { AV_PIX_FMT_GBRP12BE, AV_PIX_FMT_GBRP12LE },
    { AV_PIX_FMT_GBRP14BE, AV_PIX_FMT_GBRP14LE },
    { AV_PIX_FMT_GBRP16BE, AV_PIX_FMT_GBRP16LE },
    { AV_PIX_FMT_GBRPF32BE, AV_PIX_FMT_GBRPF32LE },
    { AV_PIX_FMT_GBRAP10BE, AV_PIX_FMT_GBRAP10LE }, { AV_PIX_FMT_GBRAP12BE, AV_PIX_FMT_GBRAP12LE }, { AV_PIX_FMT_GBRAP14BE, AV_PIX_FMT_GBRAP14LE },
    { AV_PIX_FMT_GBRAP16BE, AV_PIX_FMT_GBRAP16LE }, { AV_PIX_FMT_GBRAPF32BE, AV_PIX_FMT_GBRAPF32LE },
    { AV_PIX_FMT_P010BE, AV_PIX_FMT_P010LE },
    { AV_PIX_FMT_P012BE, AV_PIX_FMT_P012LE },
    { AV_PIX_FMT_P016BE, AV_PIX_FMT_P016LE },
    { AV_PIX_FMT_P210BE, AV_PIX_FMT_P210LE },
    { AV_PIX_FMT_P212BE, AV_PIX_FMT_P212LE },
    { AV_PIX_FMT_P216BE, AV_PIX_FMT_P216LE },
    { AV_PIX_FMT_P410BE, AV_PIX_FMT_P410LE },
    { AV_PIX_FMT_P412BE, AV_PIX_FMT_P412LE },
    { AV_PIX_FMT_P416BE, AV_PIX_FMT_P416LE },
    { AV_PIX_FMT_RGB48BE, AV_PIX_FMT_RGB48LE },
    { AV_PIX_FMT_BGR48BE, AV_PIX_FMT_BGR48LE },
    { AV_PIX_FMT_RGBA64BE, AV_PIX_FMT_RGBA64LE },
    { AV_PIX_FMT_BGRA64BE, AV_PIX_FMT_BGRA64LE }
    };
    for (const auto &pr : pairs) if (pr[0] == fmt) return pr[1];
    return -1;
}

} // namespace swship
