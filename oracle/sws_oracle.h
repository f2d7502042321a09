/*
 * sws_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar) of the librempeg libswscale legacy
 * scaler / colour-conversion path, used ONLY as the parity checker by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
 * (librempeg_amd/csrc) never links, loads or calls anything in this directory.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference tree).  Pinning status: see oracle/README.md.
 */
#ifndef SWS_ORACLE_H
#define SWS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* AVPixelFormat numeric values, libavutil/pixfmt.h (enum order) */
enum {
    ORF_NONE = -1,
    ORF_YUV420P = 0, ORF_RGB24 = 2, ORF_BGR24 = 3, ORF_YUV422P = 4, ORF_YUV444P = 5,
    ORF_GRAY8 = 8, ORF_YUVJ420P = 12, ORF_NV12 = 23, ORF_NV21 = 24,
    ORF_ARGB = 25, ORF_RGBA = 26, ORF_ABGR = 27, ORF_BGRA = 28,
    ORF_YUV420P16LE = 45, ORF_YUV444P16LE = 49, ORF_YUV420P10LE = 62,
    ORF_YUV444P10LE = 68, ORF_GBRP = 71,
    ORF_0RGB = 118, ORF_RGB0 = 119, ORF_0BGR = 120, ORF_BGR0 = 121,
    ORF_P010LE = 158, ORF_GBRPF32LE = 175,
    /* wider planar / semi-planar YUV family (same readers, writers and wrappers, other depth or subsampling) */
    ORF_YUV410P = 6, ORF_YUV411P = 7, ORF_YUVJ422P = 13, ORF_YUVJ444P = 14, ORF_YUV440P = 31, ORF_YUVJ440P = 32,
    ORF_YUV422P16LE = 47, ORF_YUV420P9LE = 60, ORF_YUV422P10LE = 64, ORF_YUV444P9LE = 66, ORF_YUV422P9LE = 70,
    ORF_NV16 = 101, ORF_YUV420P12LE = 123, ORF_YUV420P14LE = 125, ORF_YUV422P12LE = 127, ORF_YUV422P14LE = 129,
    ORF_YUV444P12LE = 131, ORF_YUV444P14LE = 133, ORF_YUV440P10LE = 151, ORF_YUV440P12LE = 153,
    ORF_P016LE = 169, ORF_NV24 = 188, ORF_NV42 = 189, ORF_P210LE = 198, ORF_P410LE = 200, ORF_P216LE = 202,
    ORF_P416LE = 204, ORF_P012LE = 209, ORF_P212LE = 222, ORF_P412LE = 224,
    /* planar RGB 9..16 bit (readers input.c:1213-1283, writers output.c:2342-2530) */
    ORF_RGB48LE = 35, ORF_BGR48LE = 58, ORF_RGBA64LE = 105, ORF_BGRA64LE = 107,
    ORF_YUYV422 = 1, ORF_UYVY422 = 15, ORF_YVYU422 = 108,
    ORF_YUVA420P = 33, ORF_YUVA422P = 78, ORF_YUVA444P = 79,
    ORF_GRAY16LE = 30, ORF_GRAY12LE = 166, ORF_GRAY10LE = 168, ORF_GRAY9LE = 173, ORF_GRAY14LE = 181,
    /* big-endian twins (value, see be_pairs in sws_oracle.c) */
    ORF_YUV420P9BE = 59, ORF_YUV420P10BE = 61, ORF_YUV420P12BE = 122, ORF_YUV420P14BE = 124, ORF_YUV420P16BE = 46, ORF_YUV422P9BE = 69, ORF_YUV422P10BE = 63, ORF_YUV422P12BE = 126, ORF_YUV422P14BE = 128, ORF_YUV422P16BE = 48, ORF_YUV444P9BE = 65, ORF_YUV444P10BE = 67, ORF_YUV444P12BE = 130, ORF_YUV444P14BE = 132, ORF_YUV444P16BE = 50, ORF_YUV440P10BE = 152, ORF_YUV440P12BE = 154, ORF_GRAY9BE = 172, ORF_GRAY10BE = 167, ORF_GRAY12BE = 165, ORF_GRAY14BE = 180, ORF_GRAY16BE = 29, ORF_GBRP9BE = 72, ORF_GBRP10BE = 74, ORF_GBRP12BE = 134, ORF_GBRP14BE = 136, ORF_GBRP16BE = 76, ORF_GBRPF32BE = 174, ORF_P010BE = 159, ORF_P012BE = 210, ORF_P016BE = 170, ORF_P210BE = 197, ORF_P212BE = 221, ORF_P216BE = 201, ORF_P410BE = 199, ORF_P412BE = 223, ORF_P416BE = 203, ORF_RGB48BE = 34, ORF_BGR48BE = 57, ORF_RGBA64BE = 104, ORF_BGRA64BE = 106,
    ORF_RGB565BE = 36, ORF_RGB565LE = 37, ORF_RGB555BE = 38, ORF_RGB555LE = 39, ORF_BGR565BE = 40, ORF_BGR565LE = 41,
    ORF_BGR555BE = 42, ORF_BGR555LE = 43, ORF_RGB444LE = 52, ORF_RGB444BE = 53, ORF_BGR444LE = 54, ORF_BGR444BE = 55,
    ORF_YUV444P10MSBBE = 258, ORF_YUV444P10MSBLE = 259, ORF_YUV444P12MSBBE = 260, ORF_YUV444P12MSBLE = 261,
    ORF_VUYA = 205, ORF_VUYX = 208, ORF_AYUV = 228, ORF_UYVA = 229, ORF_VYU444 = 230,
    ORF_AYUV64LE = 155, ORF_AYUV64BE = 156, ORF_Y210LE = 192, ORF_Y212LE = 212, ORF_Y216LE = 240, ORF_XV30LE = 214, ORF_V30XLE = 232,
    ORF_XV36BE = 215, ORF_XV36LE = 216, ORF_XV48BE = 241, ORF_XV48LE = 242,
    ORF_X2RGB10LE = 193, ORF_X2BGR10LE = 195,
    ORF_YUVA420P9LE = 81, ORF_YUVA420P9BE = 80, ORF_YUVA420P10LE = 87, ORF_YUVA420P10BE = 86, ORF_YUVA420P16LE = 93, ORF_YUVA420P16BE = 92, ORF_YUVA422P9LE = 83, ORF_YUVA422P9BE = 82, ORF_YUVA422P10LE = 89, ORF_YUVA422P10BE = 88, ORF_YUVA422P12LE = 185, ORF_YUVA422P12BE = 184, ORF_YUVA422P16LE = 95, ORF_YUVA422P16BE = 94, ORF_YUVA444P9LE = 85, ORF_YUVA444P9BE = 84, ORF_YUVA444P10LE = 91, ORF_YUVA444P10BE = 90, ORF_YUVA444P12LE = 187, ORF_YUVA444P12BE = 186, ORF_YUVA444P16LE = 97, ORF_YUVA444P16BE = 96,
    ORF_YA8 = 56, ORF_YA16BE = 109, ORF_YA16LE = 110,
    ORF_GRAYF32BE = 182, ORF_GRAYF32LE = 183,
    /* bayer mosaics: inputs only, through their own unscaled converters or a cascade over rgb24 / rgb48 */
    ORF_BAYER_BGGR8 = 139, ORF_BAYER_RGGB8 = 140, ORF_BAYER_GBRG8 = 141, ORF_BAYER_GRBG8 = 142, ORF_BAYER_BGGR16LE = 143, ORF_BAYER_BGGR16BE = 144,
    ORF_BAYER_RGGB16LE = 145, ORF_BAYER_RGGB16BE = 146, ORF_BAYER_GBRG16LE = 147, ORF_BAYER_GBRG16BE = 148, ORF_BAYER_GRBG16LE = 149, ORF_BAYER_GRBG16BE = 150,
    ORF_PAL8 = 11,   /* input only: data[1] holds 256 native-endian 0xAARRGGBB words */
    /* float / half-float sources and the packed 4:1:1 source (inputs only in the reference's format table) */
    ORF_UYYVYY411 = 16, ORF_RGBAF16BE = 206, ORF_RGBAF16LE = 207, ORF_RGBF32BE = 217, ORF_RGBF32LE = 218, ORF_RGBF16BE = 233, ORF_RGBF16LE = 234,
    ORF_GBRPF16BE = 243, ORF_GBRPF16LE = 244, ORF_GBRAPF16BE = 245, ORF_GBRAPF16LE = 246, ORF_GRAYF16BE = 247, ORF_GRAYF16LE = 248,
    ORF_YAF32BE = 252, ORF_YAF32LE = 253, ORF_YAF16BE = 254, ORF_YAF16LE = 255,
    /* 8 / 4 bits per pixel RGB (destinations only) */
    ORF_BGR8 = 17, ORF_BGR4 = 18, ORF_BGR4_BYTE = 19, ORF_RGB8 = 20, ORF_RGB4 = 21, ORF_RGB4_BYTE = 22,
    ORF_MONOWHITE = 9, ORF_MONOBLACK = 10, ORF_XYZ12LE = 99, ORF_XYZ12BE = 100,
    ORF_YUVJ411P = 138, ORF_NV20LE = 102, ORF_NV20BE = 103, ORF_GBRP10MSBBE = 262, ORF_GBRP10MSBLE = 263, ORF_GBRP12MSBBE = 264, ORF_GBRP12MSBLE = 265,
    ORF_GBRP9LE = 73, ORF_GBRP10LE = 75, ORF_GBRP16LE = 77, ORF_GBRP12LE = 135, ORF_GBRP14LE = 137,
    /* planar RGB with an alpha plane */
    ORF_GBRAP = 111, ORF_GBRAP16BE = 112, ORF_GBRAP16LE = 113, ORF_GBRAP12BE = 160, ORF_GBRAP12LE = 161, ORF_GBRAP10BE = 162, ORF_GBRAP10LE = 163,
    ORF_GBRAPF32BE = 176, ORF_GBRAPF32LE = 177, ORF_GBRAP14BE = 225, ORF_GBRAP14LE = 226,
};

/* libswscale/swscale.h:131-208 */
#define OR_SWS_FAST_BILINEAR (1 << 0)
#define OR_SWS_BILINEAR      (1 << 1)
#define OR_SWS_BICUBIC       (1 << 2)
#define OR_SWS_X             (1 << 3)
#define OR_SWS_POINT         (1 << 4)
#define OR_SWS_AREA          (1 << 5)
#define OR_SWS_BICUBLIN      (1 << 6)
#define OR_SWS_GAUSS         (1 << 7)
#define OR_SWS_SINC          (1 << 8)
#define OR_SWS_LANCZOS       (1 << 9)
#define OR_SWS_SPLINE        (1 << 10)
#define OR_SWS_PRINT_INFO    (1 << 12)
#define OR_SWS_FULL_CHR_H_INT (1 << 13)
#define OR_SWS_FULL_CHR_H_INP (1 << 14)
#define OR_SWS_ACCURATE_RND  (1 << 18)
#define OR_SWS_BITEXACT      (1 << 19)
#define OR_SWS_ERROR_DIFFUSION (1 << 23)
#define OR_SWS_PARAM_DEFAULT 123456

typedef struct OrSws OrSws;

/* options block mirroring the public SwsContext fields (swscale.h:227-315) */
typedef struct OrSwsOpts {
    int src_w, src_h, src_format;
    int dst_w, dst_h, dst_format;
    unsigned flags;
    double scaler_params[2];
    int dither;            /* SwsDither: 0 none, 1 auto, 2 bayer, 3 ed ... */
    int src_range, dst_range;
    int src_v_chr_pos, src_h_chr_pos, dst_v_chr_pos, dst_h_chr_pos; /* -513 = default */
    /* SwsFilter arguments of sws_init_context (order lumH, lumV, chrH, chrV): source vectors are convolved into the
     * taps, destination vectors only widen the rows (utils.c:385-415); NULL / 0 = none */
    const double *src_vec[4];
    int src_vec_len[4];
    int dst_vec_len[4];
    int gamma_flag;        /* SwsContext.gamma_flag: gamma-correct scaling through RGBA64 (utils.c:1461-1522) */
    int alpha_blend;       /* SwsAlphaBlend: 0 none, 1 uniform, 2 checkerboard (utils.c:1565-1615, alphablend.c) */
} OrSwsOpts;

void   or_sws_default_opts(OrSwsOpts *o);
/* = sws_getContext(): legacy init, ranges 0, chroma pos -513 (utils.c:1919) */
OrSws *or_sws_get_context(int srcW, int srcH, int srcFmt, int dstW, int dstH,
                          int dstFmt, int flags, const double *param);
/* = sws_alloc_context + set fields + sws_init_context */
OrSws *or_sws_create(const OrSwsOpts *o);
/* = sws_setColorspaceDetails (utils.c:849) */
int    or_sws_set_colorspace(OrSws *c, const int inv_table[4], int srcRange,
                             const int table[4], int dstRange,
                             int brightness, int contrast, int saturation);
const int *or_sws_get_coefficients(int colorspace); /* yuv2rgb.c:61 */
/* whole-frame sws_scale (srcSliceY must be 0 and srcSliceH == src_h).
 * returns number of output rows, <0 on error. */
int    or_sws_scale(OrSws *c, const uint8_t *const src[4], const int srcStride[4],
                    int srcSliceY, int srcSliceH,
                    uint8_t *const dst[4], const int dstStride[4]);
void   or_sws_free(OrSws *c);

/* ---- introspection for table tests ---- */
/* which: 0 hLum, 1 hChr, 2 vLum, 3 vChr.  returns filterSize (0 if none) */
int    or_sws_get_filter(const OrSws *c, int which, const int16_t **filter,
                         const int32_t **pos, int *count);
/* path: 0 = main (scaled) path, 1 = unscaled special converter, 2 = cascade */
int    or_sws_path(const OrSws *c);
const char *or_sws_path_name(const OrSws *c);
const int32_t *or_sws_rgb2yuv_table(const OrSws *c);           /* 9 ints RY..BV */
void   or_sws_yuv2rgb_coeffs(const OrSws *c, int out[6]);      /* y_offset,y_coeff,v2r,v2g,u2g,u2b */
void   or_sws_range_consts(const OrSws *c, uint32_t coeff[2], int64_t offset[2], int *active);
/* LUT access as the reference uses it: value at table_X[idx] + Y (bytes for 24bpp, dword for 32bpp) */
uint32_t or_sws_lut_rgb(const OrSws *c, int Y, int U, int V, int comp /*0 r,1 g,2 b*/);
int    or_sws_chroma_dims(const OrSws *c, int out[8]); /* chrSrcW,chrSrcH,chrDstW,chrDstH,hsubS,vsubS,hsubD,vsubD */

#ifdef __cplusplus
}
#endif
#endif
