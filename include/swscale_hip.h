/*
 * swscale_hip.h -- C-ABI of libswscale_hip.so, the MI355X-native drop-in for the
 * librempeg libswscale scaler / colour-conversion hot path.
 *
 * Every entry point below replaces the reference symbol of the same name; the
 * reference declaration it mirrors is cited as (libswscale/swscale.h:LINE) or
 * (libavutil/...:LINE), paths relative to the reference tree.  Signatures,
 * argument meaning, return values and error codes follow the reference, so a
 * caller compiled against the reference's <libswscale/swscale.h> can link
 * against this library unchanged.  Plain pointers and sizes only -- no torch,
 * no C++ types.
 *
 * Pixel buffers passed to sws_scale() may be HOST pointers (staged to HBM and
 * back, synchronous, like the reference) or DEVICE (HBM) pointers obtained
 * from sws_hip_malloc()/hipMalloc()/torch (zero-copy, stream-ordered).
 */
#ifndef SWSCALE_HIP_H
#define SWSCALE_HIP_H

#include <stddef.h>
#include <stdint.h>
#ifdef SWS_HIP_PREFIXED      /* libswship.so: the same entry points as swship_* (swscale_hip_prefix.h) */
#include "swscale_hip_prefix.h"
#endif

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden and the reference's version script (libswscale/libswscale.v:
 * only sws_* and swscale_* are exported); everything declared in this header is public */
#pragma GCC visibility push(default)

/* ---- libavutil/pixfmt.h enum AVPixelFormat (numeric values are ABI) ---- */
#ifndef AVUTIL_PIXFMT_H
#ifdef __cplusplus
/* a C caller may hand ANY int through this type (the reference answers EINVAL / NULL for an unknown one, utils.c:1234-1246); with a fixed
 * underlying type every int is a valid value on the C++ side of the boundary too (found by -fsanitize=enum, round 6) */
enum AVPixelFormat : int {
#else
enum AVPixelFormat {
#endif
    AV_PIX_FMT_NONE = -1,
    AV_PIX_FMT_YUV420P = 0,
    AV_PIX_FMT_RGB24 = 2,
    AV_PIX_FMT_BGR24 = 3,
    AV_PIX_FMT_YUV422P = 4,
    AV_PIX_FMT_YUV444P = 5,
    AV_PIX_FMT_GRAY8 = 8,
    AV_PIX_FMT_YUVJ420P = 12,
    AV_PIX_FMT_NV12 = 23,
    AV_PIX_FMT_NV21 = 24,
    AV_PIX_FMT_ARGB = 25,
    AV_PIX_FMT_RGBA = 26,
    AV_PIX_FMT_ABGR = 27,
    AV_PIX_FMT_BGRA = 28,
    AV_PIX_FMT_YUV420P16LE = 45,
    AV_PIX_FMT_YUV444P16LE = 49,
    AV_PIX_FMT_YUV420P10LE = 62,
    AV_PIX_FMT_YUV444P10LE = 68,
    AV_PIX_FMT_GBRP = 71,
    AV_PIX_FMT_0RGB = 118,
    AV_PIX_FMT_RGB0 = 119,
    AV_PIX_FMT_0BGR = 120,
    AV_PIX_FMT_BGR0 = 121,
    AV_PIX_FMT_P010LE = 158,
    AV_PIX_FMT_GBRPF32LE = 175,
    /* the wider planar / semi-planar YUV family the same readers, writers and wrappers cover */
    AV_PIX_FMT_YUV410P = 6, AV_PIX_FMT_YUV411P = 7, AV_PIX_FMT_YUVJ422P = 13, AV_PIX_FMT_YUVJ444P = 14,
    AV_PIX_FMT_YUV440P = 31, AV_PIX_FMT_YUVJ440P = 32, AV_PIX_FMT_YUV422P16LE = 47, AV_PIX_FMT_YUV420P9LE = 60,
    AV_PIX_FMT_YUV422P10LE = 64, AV_PIX_FMT_YUV444P9LE = 66, AV_PIX_FMT_YUV422P9LE = 70, AV_PIX_FMT_NV16 = 101,
    AV_PIX_FMT_YUV420P12LE = 123, AV_PIX_FMT_YUV420P14LE = 125, AV_PIX_FMT_YUV422P12LE = 127,
    AV_PIX_FMT_YUV422P14LE = 129, AV_PIX_FMT_YUV444P12LE = 131, AV_PIX_FMT_YUV444P14LE = 133,
    AV_PIX_FMT_YUV440P10LE = 151, AV_PIX_FMT_YUV440P12LE = 153, AV_PIX_FMT_P016LE = 169, AV_PIX_FMT_NV24 = 188,
    AV_PIX_FMT_NV42 = 189, AV_PIX_FMT_P210LE = 198, AV_PIX_FMT_P410LE = 200, AV_PIX_FMT_P216LE = 202,
    AV_PIX_FMT_P416LE = 204, AV_PIX_FMT_P012LE = 209, AV_PIX_FMT_P212LE = 222, AV_PIX_FMT_P412LE = 224,
    /* packed 16-bit RGB */
    AV_PIX_FMT_RGB48LE = 35, AV_PIX_FMT_BGR48LE = 58, AV_PIX_FMT_RGBA64LE = 105, AV_PIX_FMT_BGRA64LE = 107,
    /* packed 4:2:2 */
    AV_PIX_FMT_YUYV422 = 1, AV_PIX_FMT_UYVY422 = 15, AV_PIX_FMT_YVYU422 = 108,
    /* planar YUV with an alpha plane */
    AV_PIX_FMT_YUVA420P9LE = 81, AV_PIX_FMT_YUVA420P9BE = 80, AV_PIX_FMT_YUVA420P10LE = 87, AV_PIX_FMT_YUVA420P10BE = 86, AV_PIX_FMT_YUVA420P16LE = 93, AV_PIX_FMT_YUVA420P16BE = 92, AV_PIX_FMT_YUVA422P9LE = 83, AV_PIX_FMT_YUVA422P9BE = 82, AV_PIX_FMT_YUVA422P10LE = 89, AV_PIX_FMT_YUVA422P10BE = 88, AV_PIX_FMT_YUVA422P12LE = 185, AV_PIX_FMT_YUVA422P12BE = 184, AV_PIX_FMT_YUVA422P16LE = 95, AV_PIX_FMT_YUVA422P16BE = 94, AV_PIX_FMT_YUVA444P9LE = 85, AV_PIX_FMT_YUVA444P9BE = 84, AV_PIX_FMT_YUVA444P10LE = 91, AV_PIX_FMT_YUVA444P10BE = 90, AV_PIX_FMT_YUVA444P12LE = 187, AV_PIX_FMT_YUVA444P12BE = 186, AV_PIX_FMT_YUVA444P16LE = 97, AV_PIX_FMT_YUVA444P16BE = 96,
    AV_PIX_FMT_YUVA420P = 33, AV_PIX_FMT_YUVA422P = 78, AV_PIX_FMT_YUVA444P = 79,
    /* gray (always full range, utils.c:791-805) */
    AV_PIX_FMT_GRAY16LE = 30, AV_PIX_FMT_GRAY12LE = 166, AV_PIX_FMT_GRAY10LE = 168, AV_PIX_FMT_GRAY9LE = 173,
    AV_PIX_FMT_GRAY14LE = 181,
    /* planar RGB 9..16 bit */
    AV_PIX_FMT_GBRP9LE = 73, AV_PIX_FMT_GBRP10LE = 75, AV_PIX_FMT_GBRP16LE = 77, AV_PIX_FMT_GBRP12LE = 135,
    AV_PIX_FMT_GBRP14LE = 137,
    AV_PIX_FMT_AYUV64LE = 155, AV_PIX_FMT_AYUV64BE = 156, AV_PIX_FMT_Y210LE = 192, AV_PIX_FMT_Y212LE = 212, AV_PIX_FMT_Y216LE = 240,
    AV_PIX_FMT_X2RGB10LE = 193, AV_PIX_FMT_X2BGR10LE = 195,
    AV_PIX_FMT_YA8 = 56, AV_PIX_FMT_YA16BE = 109, AV_PIX_FMT_YA16LE = 110,
    AV_PIX_FMT_GRAYF32BE = 182, AV_PIX_FMT_GRAYF32LE = 183,
    /* bayer mosaics: inputs only, through their own unscaled converters or a cascade over rgb24 / rgb48 */
    AV_PIX_FMT_BAYER_BGGR8 = 139, AV_PIX_FMT_BAYER_RGGB8 = 140, AV_PIX_FMT_BAYER_GBRG8 = 141, AV_PIX_FMT_BAYER_GRBG8 = 142,
    AV_PIX_FMT_BAYER_BGGR16LE = 143, AV_PIX_FMT_BAYER_BGGR16BE = 144, AV_PIX_FMT_BAYER_RGGB16LE = 145, AV_PIX_FMT_BAYER_RGGB16BE = 146,
    AV_PIX_FMT_BAYER_GBRG16LE = 147, AV_PIX_FMT_BAYER_GBRG16BE = 148, AV_PIX_FMT_BAYER_GRBG16LE = 149, AV_PIX_FMT_BAYER_GRBG16BE = 150,
    AV_PIX_FMT_PAL8 = 11,   /* input only: data[1] holds 256 native-endian 0xAARRGGBB words */
    /* inputs only (like the reference's format table): float / half-float pictures and the packed 4:1:1 layout */
    AV_PIX_FMT_UYYVYY411 = 16, AV_PIX_FMT_RGBAF16BE = 206, AV_PIX_FMT_RGBAF16LE = 207, AV_PIX_FMT_RGBF32BE = 217, AV_PIX_FMT_RGBF32LE = 218,
    AV_PIX_FMT_RGBF16BE = 233, AV_PIX_FMT_RGBF16LE = 234, AV_PIX_FMT_GBRPF16BE = 243, AV_PIX_FMT_GBRPF16LE = 244, AV_PIX_FMT_GBRAPF16BE = 245,
    AV_PIX_FMT_GBRAPF16LE = 246, AV_PIX_FMT_GRAYF16BE = 247, AV_PIX_FMT_GRAYF16LE = 248, AV_PIX_FMT_YAF32BE = 252, AV_PIX_FMT_YAF32LE = 253,
    AV_PIX_FMT_YAF16BE = 254, AV_PIX_FMT_YAF16LE = 255,
    /* 8 / 4 bits per pixel RGB: destinations only (sources need the palette path, swscale_internal.h:936-953 usePal) */
    AV_PIX_FMT_BGR8 = 17, AV_PIX_FMT_BGR4 = 18, AV_PIX_FMT_BGR4_BYTE = 19, AV_PIX_FMT_RGB8 = 20, AV_PIX_FMT_RGB4 = 21, AV_PIX_FMT_RGB4_BYTE = 22,
    AV_PIX_FMT_MONOWHITE = 9, AV_PIX_FMT_MONOBLACK = 10, AV_PIX_FMT_XYZ12LE = 99, AV_PIX_FMT_XYZ12BE = 100,
    AV_PIX_FMT_YUVJ411P = 138, AV_PIX_FMT_NV20LE = 102, AV_PIX_FMT_NV20BE = 103,
    AV_PIX_FMT_GBRP10MSBBE = 262, AV_PIX_FMT_GBRP10MSBLE = 263, AV_PIX_FMT_GBRP12MSBBE = 264, AV_PIX_FMT_GBRP12MSBLE = 265,
    AV_PIX_FMT_XV30LE = 214, AV_PIX_FMT_V30XLE = 232, AV_PIX_FMT_XV36BE = 215, AV_PIX_FMT_XV36LE = 216, AV_PIX_FMT_XV48BE = 241, AV_PIX_FMT_XV48LE = 242,
    AV_PIX_FMT_VUYA = 205, AV_PIX_FMT_VUYX = 208, AV_PIX_FMT_AYUV = 228, AV_PIX_FMT_UYVA = 229, AV_PIX_FMT_VYU444 = 230,
    AV_PIX_FMT_YUV444P10MSBBE = 258, AV_PIX_FMT_YUV444P10MSBLE = 259, AV_PIX_FMT_YUV444P12MSBBE = 260, AV_PIX_FMT_YUV444P12MSBLE = 261,
    /* 16 bits per pixel packed RGB (ordered-dither writers, output.c:1714-1748) */
    AV_PIX_FMT_RGB565BE = 36, AV_PIX_FMT_RGB565LE = 37, AV_PIX_FMT_RGB555BE = 38, AV_PIX_FMT_RGB555LE = 39, AV_PIX_FMT_BGR565BE = 40, AV_PIX_FMT_BGR565LE = 41,
    AV_PIX_FMT_BGR555BE = 42, AV_PIX_FMT_BGR555LE = 43, AV_PIX_FMT_RGB444LE = 52, AV_PIX_FMT_RGB444BE = 53, AV_PIX_FMT_BGR444LE = 54, AV_PIX_FMT_BGR444BE = 55,
    /* big-endian twins of the 16-bit / float formats above (converted through the little-endian twin + a byte-swap pass) */
    AV_PIX_FMT_YUV420P9BE = 59, AV_PIX_FMT_YUV420P10BE = 61, AV_PIX_FMT_YUV420P12BE = 122, AV_PIX_FMT_YUV420P14BE = 124,
    AV_PIX_FMT_YUV420P16BE = 46, AV_PIX_FMT_YUV422P9BE = 69, AV_PIX_FMT_YUV422P10BE = 63, AV_PIX_FMT_YUV422P12BE = 126,
    AV_PIX_FMT_YUV422P14BE = 128, AV_PIX_FMT_YUV422P16BE = 48, AV_PIX_FMT_YUV444P9BE = 65, AV_PIX_FMT_YUV444P10BE = 67,
    AV_PIX_FMT_YUV444P12BE = 130, AV_PIX_FMT_YUV444P14BE = 132, AV_PIX_FMT_YUV444P16BE = 50, AV_PIX_FMT_YUV440P10BE = 152,
    AV_PIX_FMT_YUV440P12BE = 154, AV_PIX_FMT_GRAY9BE = 172, AV_PIX_FMT_GRAY10BE = 167, AV_PIX_FMT_GRAY12BE = 165,
    AV_PIX_FMT_GRAY14BE = 180, AV_PIX_FMT_GRAY16BE = 29, AV_PIX_FMT_GBRP9BE = 72, AV_PIX_FMT_GBRP10BE = 74,
    AV_PIX_FMT_GBRP12BE = 134, AV_PIX_FMT_GBRP14BE = 136, AV_PIX_FMT_GBRP16BE = 76, AV_PIX_FMT_GBRPF32BE = 174,
    AV_PIX_FMT_P010BE = 159, AV_PIX_FMT_P012BE = 210, AV_PIX_FMT_P016BE = 170, AV_PIX_FMT_P210BE = 197, AV_PIX_FMT_P212BE = 221,
    AV_PIX_FMT_P216BE = 201, AV_PIX_FMT_P410BE = 199, AV_PIX_FMT_P412BE = 223, AV_PIX_FMT_P416BE = 203, AV_PIX_FMT_RGB48BE = 34,
    AV_PIX_FMT_BGR48BE = 57, AV_PIX_FMT_RGBA64BE = 104, AV_PIX_FMT_BGRA64BE = 106,
    /* planar RGB with an alpha plane */
    AV_PIX_FMT_GBRAP = 111, AV_PIX_FMT_GBRAP16BE = 112, AV_PIX_FMT_GBRAP16LE = 113, AV_PIX_FMT_GBRAP12BE = 160, AV_PIX_FMT_GBRAP12LE = 161,
    AV_PIX_FMT_GBRAP10BE = 162, AV_PIX_FMT_GBRAP10LE = 163, AV_PIX_FMT_GBRAPF32BE = 176, AV_PIX_FMT_GBRAPF32LE = 177, AV_PIX_FMT_GBRAP14BE = 225, AV_PIX_FMT_GBRAP14LE = 226,
    /* NEW: hardware surface format of the HIP hwcontext slot; takes the value of the
     * reference's AV_PIX_FMT_NB (268, libavutil/pixfmt.h:508), i.e. it is appended as the last format */
    AV_PIX_FMT_HIP = 268,
};
#endif

/* ---- libavutil/hwcontext.h:27-44 enum AVHWDeviceType: the new slot is appended
 *      after AV_HWDEVICE_TYPE_OHCODEC ---- */
#define AV_HWDEVICE_TYPE_HIP 15

/* ---- libavutil/error.h ---- */
#define SWS_AVERROR(e) (-(e))

/* ---- flags (libswscale/swscale.h:131-208) ---- */
#define SWS_FAST_BILINEAR (1 << 0)
#define SWS_BILINEAR      (1 << 1)
#define SWS_BICUBIC       (1 << 2)
#define SWS_X             (1 << 3)
#define SWS_POINT         (1 << 4)
#define SWS_AREA          (1 << 5)
#define SWS_BICUBLIN      (1 << 6)
#define SWS_GAUSS         (1 << 7)
#define SWS_SINC          (1 << 8)
#define SWS_LANCZOS       (1 << 9)
#define SWS_SPLINE        (1 << 10)
#define SWS_STRICT        (1 << 11)
#define SWS_PRINT_INFO    (1 << 12)
#define SWS_FULL_CHR_H_INT (1 << 13)
#define SWS_FULL_CHR_H_INP (1 << 14)
#define SWS_DIRECT_BGR    (1 << 15)
#define SWS_ACCURATE_RND  (1 << 18)
#define SWS_BITEXACT      (1 << 19)
#define SWS_UNSTABLE      (1 << 20)
#define SWS_ERROR_DIFFUSION (1 << 23)

#define SWS_SRC_V_CHR_DROP_MASK  0x30000   /* swscale.h:453 */
#define SWS_SRC_V_CHR_DROP_SHIFT 16
#define SWS_PARAM_DEFAULT 123456           /* swscale.h:456 */
#define SWS_MAX_REDUCE_CUTOFF 0.002        /* swscale.h:447 */

#define SWS_CS_ITU709    1                 /* swscale.h:458-465 */
#define SWS_CS_FCC       4
#define SWS_CS_ITU601    5
#define SWS_CS_ITU624    5
#define SWS_CS_SMPTE170M 5
#define SWS_CS_SMPTE240M 7
#define SWS_CS_DEFAULT   5
#define SWS_CS_BT2020    9

typedef enum SwsDither {                    /* swscale.h:77-86 */
    SWS_DITHER_NONE = 0, SWS_DITHER_AUTO, SWS_DITHER_BAYER, SWS_DITHER_ED,
    SWS_DITHER_A_DITHER, SWS_DITHER_X_DITHER, SWS_DITHER_NB,
    SWS_DITHER_MAX_ENUM = 0x7FFFFFFF,
} SwsDither;

typedef enum SwsAlphaBlend {                /* swscale.h:88-94 */
    SWS_ALPHA_BLEND_NONE = 0, SWS_ALPHA_BLEND_UNIFORM, SWS_ALPHA_BLEND_CHECKERBOARD,
    SWS_ALPHA_BLEND_NB, SWS_ALPHA_BLEND_MAX_ENUM = 0x7FFFFFFF,
} SwsAlphaBlend;

typedef enum SwsScaler {                    /* swscale.h:96-108 */
    SWS_SCALE_AUTO = 0, SWS_SCALE_BILINEAR, SWS_SCALE_BICUBIC, SWS_SCALE_POINT, SWS_SCALE_AREA,
    SWS_SCALE_GAUSSIAN, SWS_SCALE_SINC, SWS_SCALE_LANCZOS, SWS_SCALE_SPLINE, SWS_SCALE_NB,
    SWS_SCALE_MAX_ENUM = 0x7FFFFFFF,
} SwsScaler;

typedef enum SwsBackend {                   /* swscale.h:110-128 + NEW HIP bit */
    SWS_BACKEND_LEGACY = (1 << 0),
    SWS_BACKEND_HIP    = (1 << 8),          /* NEW: this library */
    SWS_BACKEND_MAX_ENUM = 0x7FFFFFFF,
} SwsBackend;

/* Main external API structure (libswscale/swscale.h:227-315): identical field
 * order and types, so code that pokes the public fields keeps working. */
typedef struct SwsContext {
    const void *av_class;
    void *opaque;
    unsigned flags;
#define SWS_NUM_SCALER_PARAMS 2
    double scaler_params[SWS_NUM_SCALER_PARAMS];
    int threads;
    SwsDither dither;
    SwsAlphaBlend alpha_blend;
    int gamma_flag;
    int src_w, src_h;
    int dst_w, dst_h;
    int src_format;
    int dst_format;
    int src_range;
    int dst_range;
    int src_v_chr_pos;
    int src_h_chr_pos;
    int dst_v_chr_pos;
    int dst_h_chr_pos;
    int intent;
    SwsScaler scaler;
    SwsScaler scaler_sub;
    SwsBackend backends;
} SwsContext;

typedef struct SwsVector { double *coeff; int length; } SwsVector;           /* swscale.h:478-481 */
typedef struct SwsFilter { SwsVector *lumH, *lumV, *chrH, *chrV; } SwsFilter; /* swscale.h:484-489 */

/* ---- version (swscale.h:50-63, version.h) ---- */
unsigned    swscale_version(void);
const char *swscale_configuration(void);
const char *swscale_license(void);

/* ---- context management ---- */
SwsContext *sws_alloc_context(void);                                   /* swscale.h:320 */
void        sws_free_context(SwsContext **ctx);                        /* swscale.h:326 */
int         sws_init_context(SwsContext *c, SwsFilter *srcFilter, SwsFilter *dstFilter); /* swscale.h:522 */
void        sws_freeContext(SwsContext *c);                            /* swscale.h:528 */
SwsContext *sws_getContext(int srcW, int srcH, enum AVPixelFormat srcFormat,
                           int dstW, int dstH, enum AVPixelFormat dstFormat,
                           int flags, SwsFilter *srcFilter, SwsFilter *dstFilter,
                           const double *param);                       /* swscale.h:551 */
SwsContext *sws_getCachedContext(SwsContext *context, int srcW, int srcH, enum AVPixelFormat srcFormat,
                                 int dstW, int dstH, enum AVPixelFormat dstFormat, int flags,
                                 SwsFilter *srcFilter, SwsFilter *dstFilter, const double *param); /* swscale.h:737 */

int sws_isSupportedInput(enum AVPixelFormat pix_fmt);                  /* swscale.h:495 */
int sws_isSupportedOutput(enum AVPixelFormat pix_fmt);                 /* swscale.h:501 */
int sws_isSupportedEndiannessConversion(enum AVPixelFormat pix_fmt);   /* swscale.h:508 */

/* ---- colourspace ---- */
const int *sws_getCoefficients(int colorspace);                        /* swscale.h:474 */
int sws_setColorspaceDetails(SwsContext *c, const int inv_table[4], int srcRange,
                             const int table[4], int dstRange,
                             int brightness, int contrast, int saturation); /* swscale.h:684 */
int sws_getColorspaceDetails(SwsContext *c, int **inv_table, int *srcRange, int **table,
                             int *dstRange, int *brightness, int *contrast, int *saturation); /* swscale.h:692 */

/* ---- the hot path ---- */
/* swscale.h:583.  Returns the number of output rows written (>= 0) or a negative
 * AVERROR: EINVAL for NULL arguments / bad slice geometry / bad plane pointers
 * (libswscale/swscale.c:1041-1070), AVERROR_EXTERNAL for HIP failures.  Slices are
 * accepted in order, top-down or bottom-up, on every path (scaled path: the slices
 * are assembled on the device and each call returns the row count the reference's
 * cursor gives, swscale.c:372-381, :566). */
int sws_scale(SwsContext *c, const uint8_t *const srcSlice[], const int srcStride[],
              int srcSliceY, int srcSliceH, uint8_t *const dst[], const int dstStride[]);

/* libavutil/buffer.h:82-95 struct AVBufferRef, libavutil/rational.h:58-61, libavutil/channel_layout.h (AVChannelLayout):
 * the member types struct AVFrame is made of. */
typedef struct SwsBufferRef { void *buffer; uint8_t *data; size_t size; } SwsBufferRef;
typedef struct SwsRational { int num, den; } SwsRational;
typedef struct SwsChannelLayout { int order; int nb_channels; union { uint64_t mask; void *map; } u; void *opaque; } SwsChannelLayout;
/* libavutil/frame.h:236-251 struct AVFrameSideData */
typedef struct SwsFrameSideData { int type; uint8_t *data; size_t size; void *metadata; SwsBufferRef *buf; } SwsFrameSideData;

/* Field-for-field mirror of libavutil/frame.h:472-828 struct AVFrame (same order, same types,
 * sizeof == sizeof(AVFrame)): a real AVFrame* may be passed wherever SwsFrameView* is expected.
 * The library reads data, linesize, width, height, format, flags (AV_FRAME_FLAG_INTERLACED),
 * color_range, color_primaries, color_trc, colorspace, chroma_location, side_data (compared only)
 * and hw_frames_ctx; it writes nothing but the pixels data[] points at. */
typedef struct SwsFrameView {
    uint8_t *data[8];
    int linesize[8];
    uint8_t **extended_data;
    int width, height;
    int nb_samples;
    int format;
    int pict_type;
    SwsRational sample_aspect_ratio;
    int64_t pts;
    int64_t pkt_dts;
    SwsRational time_base;
    int quality;
    void *opaque;
    int repeat_pict;
    int sample_rate;
    SwsBufferRef *buf[8];
    SwsBufferRef **extended_buf;
    int nb_extended_buf;
    SwsFrameSideData **side_data;
    int nb_side_data;
    int flags;
    int color_range;          /* enum AVColorRange */
    int color_primaries;      /* enum AVColorPrimaries */
    int color_trc;            /* enum AVColorTransferCharacteristic */
    int colorspace;           /* enum AVColorSpace */
    int chroma_location;      /* enum AVChromaLocation */
    int64_t best_effort_timestamp;
    void *metadata;
    int decode_error_flags;
    SwsBufferRef *hw_frames_ctx;   /* -> struct AVHWFramesContext (include/hwcontext_hip.h) when format == AV_PIX_FMT_HIP */
    SwsBufferRef *opaque_ref;
    size_t crop_top, crop_bottom, crop_left, crop_right;
    void *private_ref;
    SwsChannelLayout ch_layout;
    int64_t duration;
    int alpha_mode;
} SwsFrameView;

#define SWS_FRAME_FLAG_INTERLACED (1 << 3)   /* AV_FRAME_FLAG_INTERLACED, frame.h:695 */
/* libavutil/pixfmt.h enum AVColorRange / AVChromaLocation values the frame fields carry */
#define SWS_COL_RANGE_UNSPECIFIED 0
#define SWS_COL_RANGE_MPEG 1
#define SWS_COL_RANGE_JPEG 2
#define SWS_CHROMA_LOC_UNSPECIFIED 0
#define SWS_CHROMA_LOC_LEFT 1
#define SWS_CHROMA_LOC_CENTER 2
#define SWS_CHROMA_LOC_TOPLEFT 3
#define SWS_CHROMA_LOC_TOP 4
#define SWS_CHROMA_LOC_BOTTOMLEFT 5
#define SWS_CHROMA_LOC_BOTTOM 6

/* swscale.h:439 (legacy-initialised contexts only: frame props must match the context) */
int sws_scale_frame(SwsContext *c, SwsFrameView *dst, const SwsFrameView *src);

/* NEW: nb_frames independent sws_scale_frame() calls with identical
 * results, sharded over the GPUs of the node inside the library: a frame that lives
 * in HBM is converted on the GPU that holds it (ONE batched launch set per GPU, on
 * that GPU's stream, launches issued on every GPU before anything is waited for);
 * frames in host memory are dealt round-robin over the GPUs and staged by one host
 * thread per GPU.  Each GPU gets its own copy of the context's tables at first use.
 * Asynchronous for HBM-resident frames: sws_hip_sync() waits for all GPUs.
 * Returns nb_frames on success or a negative AVERROR. */
int sws_scale_frames(SwsContext *c, SwsFrameView *const dst[], const SwsFrameView *const src[], int nb_frames);
/* the partition rule of sws_scale_frames() as a pure function: src_device[i] / dst_device[i] = GPU that holds the
 * frame's source / destination (-1 = host memory); out_device[i] = GPU that converts frame i */
int sws_hip_plan_shards(int nb_frames, const int *src_device, const int *dst_device, int nb_devices, int home, int *out_device);

/* ---- the rest of the reference's exported API (swscale.h) ---- */
const void *sws_get_class(void);                                            /* swscale.h:71 (const AVClass *) */
int sws_test_format(enum AVPixelFormat format, int output);                 /* :342 */
int sws_test_hw_format(enum AVPixelFormat format);                          /* :352 */
int sws_test_colorspace(int colorspace, int output);                        /* :363 (enum AVColorSpace) */
int sws_test_primaries(int primaries, int output);                          /* :374 (enum AVColorPrimaries) */
int sws_test_transfer(int trc, int output);                                 /* :385 (enum AVColorTransferCharacteristic) */
int sws_test_frame(const SwsFrameView *frame, int output);                  /* :392 */
int sws_frame_setup(SwsContext *ctx, const SwsFrameView *dst, const SwsFrameView *src); /* :405 */
int sws_is_noop(const SwsFrameView *dst, const SwsFrameView *src);          /* :415 */
int sws_frame_start(SwsContext *c, SwsFrameView *dst, const SwsFrameView *src);  /* :613 */
void sws_frame_end(SwsContext *c);                                          /* :623 */
int sws_send_slice(SwsContext *c, unsigned int slice_start, unsigned int slice_height);    /* :637 */
int sws_receive_slice(SwsContext *c, unsigned int slice_start, unsigned int slice_height); /* :657 */
unsigned int sws_receive_slice_alignment(const SwsContext *c);              /* :669 */
SwsVector *sws_allocVec(int length);                                        /* :699 */
SwsVector *sws_getGaussianVec(double variance, double quality);             /* :705 */
void sws_scaleVec(SwsVector *a, double scalar);                             /* :710 */
void sws_normalizeVec(SwsVector *a, double height);                         /* :715 */
void sws_freeVec(SwsVector *a);                                             /* :717 */
SwsFilter *sws_getDefaultFilter(float lumaGBlur, float chromaGBlur, float lumaSharpen, float chromaSharpen,
                                float chromaHShift, float chromaVShift, int verbose); /* :719 */
void sws_freeFilter(SwsFilter *filter);                                     /* :723 */
void sws_convertPalette8ToPacked32(const uint8_t *src, uint8_t *dst, int num_pixels, const uint8_t *palette); /* :753 */
void sws_convertPalette8ToPacked24(const uint8_t *src, uint8_t *dst, int num_pixels, const uint8_t *palette); /* :765 */

/* ---- HIP device plumbing (the AV_HWDEVICE_TYPE_HIP slot; shape of
 *      libavutil/hwcontext_internal.h:29-99 HWContextType, model
 *      libavutil/hwcontext_cuda.c:132-197, :523-655) ---- */
int   sws_hip_device_count(void);
int   sws_hip_set_device(SwsContext *c, int device);       /* device_create/derive: the context's home GPU */
int   sws_hip_get_device(SwsContext *c);
int   sws_hip_set_stream(SwsContext *c, void *hip_stream); /* use caller's hipStream_t (NULL = context-owned) */
void *sws_hip_get_stream(SwsContext *c);
int   sws_hip_sync(SwsContext *c);
/* frames_get_buffer: one linear HBM allocation per frame, planes at 256-byte aligned
 * offsets, linesize aligned to 256 bytes (hwcontext_cuda.c:132-197 analogue). */
int   sws_hip_frame_alloc(SwsFrameView *f, int format, int width, int height, int device);
void  sws_hip_frame_free(SwsFrameView *f);
/* transfer_data_to / transfer_data_from: per-plane 2-D async copies on the stream */
int   sws_hip_frame_upload(SwsContext *c, SwsFrameView *dev, const SwsFrameView *host);
int   sws_hip_frame_download(SwsContext *c, SwsFrameView *host, const SwsFrameView *dev);
int   sws_hip_image_layout(int format, int width, int height, int align, int linesize[4], size_t offset[4], size_t *total);

/* ---- table blob: what rank 0 broadcasts to the other GPUs (RCCL) once per context.
 *      export on the rank that ran the host-side init, import on the others. ---- */
size_t sws_hip_tables_size(const SwsContext *c);
int    sws_hip_tables_export(const SwsContext *c, void *buf, size_t size);
int    sws_hip_tables_import(SwsContext *c, const void *buf, size_t size);

/* ---- introspection used by tests and the bench ---- */
const char *sws_hip_path_name(const SwsContext *c);   /* e.g. "unscaled:yuv2rgb", "fused:rgb_lut", "generic:hv" */
const char *sws_hip_kernel_name(const SwsContext *c); /* dominant kernel symbol for rocprof matching */
int    sws_hip_get_filter(const SwsContext *c, int which, const int16_t **filter, const int32_t **pos, int *count);
int    sws_hip_get_tables(const SwsContext *c, int32_t rgb2yuv[9], int yuv2rgb[6], uint32_t range_coeff[2], int64_t range_offset[2]);
double sws_hip_last_kernel_ms(SwsContext *c);         /* HIP-event time of the last timed launch (see below) */
int    sws_hip_set_timing(SwsContext *c, int enable); /* record hipEvents around each launch on the context's stream */
/* launch heuristics ("strip_min_w", "rgb_march_waves", "max_devices", "no_strip", ...): every setting gives the same bytes */
int    sws_hip_set_option(SwsContext *c, const char *name, int value);
/* plans the context now (sws_scale() does it on first use) and digests the plan: digest[0] over the table blocks the planner built, digest[1] over the kernel
 * parameters, digest[2] over the host-side launch state (which kernels, on what strip / tile geometry).  With the option "dry_plan" set before, no GPU is touched (the context can name its path -- sws_hip_path_name -- but not convert). */
int    sws_hip_plan(SwsContext *c, uint64_t digest[3]);
/* debugging aid: reads every device table block of the context back and compares it with what was uploaded (and the host-side kernel parameters with
 * what context preparation left); returns the number of anomalies (0 = intact, < 0 = HIP error), a short text per anomaly goes to buf */
int    sws_hip_debug_check(SwsContext *c, char *buf, int cap);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* SWSCALE_HIP_H */
