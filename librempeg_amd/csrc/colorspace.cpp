// Host-side colour constants: the closed form of the reference's yuv->rgb look-up tables,
// the 15-bit rgb->yuv matrix and the range-conversion constants.
#include <algorithm>
#include <cstring>

#include "swsint.hpp"

namespace swship {

static const int32_t g_yuv2rgb_coeffs[11][4] = { // libswscale/yuv2rgb.c:47-59
    { 104597, 132201, 25675, 53279 }, { 117489, 138438, 13975, 34925 },
    { 104597, 132201, 25675, 53279 }, { 104597, 132201, 25675, 53279 },
    { 104448, 132798, 24759, 53109 }, { 104597, 132201, 25675, 53279 },
    { 104597, 132201, 25675, 53279 }, { 117579, 136230, 16907, 35559 },
    { 0, 0, 0, 0 },                   { 110013, 140363, 12277, 42626 },
    { 110013, 140363, 12277, 42626 },
};

const int *yuv2rgb_coeffs(int colorspace) // sws_getCoefficients, yuv2rgb.c:61-66
{
    if (colorspace > 10 || colorspace < 0 || colorspace == 8) colorspace = SWS_CS_DEFAULT;
    return g_yuv2rgb_coeffs[colorspace];
}

static int sat_round16(int64_t f) // roundToInt16, yuv2rgb.c:705-715 (value as int16)
{
    const int r = (int)((f + (1 << 15)) >> 16);
    if (r < -0x7FFF) return -0x8000;
    if (r > 0x7FFF) return 0x7FFF;
    return r;
}

// ff_yuv2rgb_c_init_tables (yuv2rgb.c:717-973) builds a 2048-entry luma ramp
//     y_table[k] = clip_u8((yb0 + k*cy + 0x8000) >> 16)
// and per-chroma index tables that are affine in clip_u8(chroma):
//     table_X[c + 512] = &y_table[yoffs - (inc >> 9) + ((clip_u8(c) * inc) >> 16)]
// The kernels evaluate those two formulas directly; tests prove the closed
// form equals the oracle's real tables for every (Y, U, V) index that can occur.
void build_yuv2rgb(Yuv2RgbLut &l, const int inv_table[4], int fullRange, int brightness, int contrast, int saturation)
{
    int64_t crv = inv_table[0], cbu = inv_table[1], cgu = -(int64_t)inv_table[2], cgv = -(int64_t)inv_table[3];
    int64_t cy = 1 << 16, oy = 0;
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

    l.y_coeff  = (int16_t)sat_round16(cy * (1 << 13));   // :786-791
    l.y_offset = (int16_t)sat_round16(oy * (1 << 9));
    l.v2r = (int16_t)sat_round16(crv * (1 << 13));
    l.v2g = (int16_t)sat_round16(cgv * (1 << 13));
    l.u2g = (int16_t)sat_round16(cgu * (1 << 13));
    l.u2b = (int16_t)sat_round16(cbu * (1 << 13));

    const int64_t den = std::max<int64_t>(cy, 1);          // :794-797
    l.crv = ((crv * (1 << 16)) + 0x8000) / den;
    l.cbu = ((cbu * (1 << 16)) + 0x8000) / den;
    l.cgu = ((cgu * (1 << 16)) + 0x8000) / den;
    l.cgv = ((cgv * (1 << 16)) + 0x8000) / den;
    l.cy = cy;
    l.yb0 = -(384LL << 16) - 512 * cy - oy;                // :901-909
    l.yoffs = (fullRange ? 384 : 326) + 512;               // :749
    l.valid = true;
}

void build_rgb2yuv(int32_t out[9], const int table[4]) // fill_rgb2yuv_table, utils.c:614-706
{
    enum { RY, GY, BY, RU, GU, BU, RV, GV, BV };
    const int64_t ONE = 65536;
    const int64_t vr = table[0], ub = table[1], ug = -(int64_t)table[2], vg = -(int64_t)table[3];
    auto rdiv = [](int64_t a, int64_t b) { return (a >= 0 ? a + (b >> 1) : a - (b >> 1)) / b; };
    const int64_t cy = ONE * 255 / 219;                    // dstRange is forced to 0 (:663)
    const int64_t W = rdiv(ONE * ONE * ug, ub), V = rdiv(ONE * ONE * vg, vr), Z = ONE * ONE - W - V;
    const int64_t Cy = rdiv(cy * Z, ONE), Cu = rdiv(ub * Z, ONE), Cv = rdiv(vr * Z, ONE);
    const int64_t S = 1 << 15;                             // RGB2YUV_SHIFT
    out[RY] = (int32_t)-rdiv(S * V, Cy);
    out[GY] = (int32_t) rdiv(S * ONE * ONE, Cy);
    out[BY] = (int32_t)-rdiv(S * W, Cy);
    out[RU] = (int32_t) rdiv(S * V, Cu);
    out[GU] = (int32_t)-rdiv(S * ONE * ONE, Cu);
    out[BU] = (int32_t) rdiv(S * (Z + W), Cu);
    out[RV] = (int32_t) rdiv(S * (V + Z), Cv);
    out[GV] = (int32_t)-rdiv(S * ONE * ONE, Cv);
    out[BV] = (int32_t) rdiv(S * W, Cv);
    if (!std::memcmp(table, g_yuv2rgb_coeffs[SWS_CS_DEFAULT], sizeof(int) * 4)) { // BT.601 literals (:693-703)
        out[BY] =  (int)(0.114 * 219 / 255 * (1 << 15) + 0.5);
        out[BV] = -(int)(0.081 * 224 / 255 * (1 << 15) + 0.5);
        out[BU] =  (int)(0.500 * 224 / 255 * (1 << 15) + 0.5);
        out[GY] =  (int)(0.587 * 219 / 255 * (1 << 15) + 0.5);
        out[GV] = -(int)(0.419 * 224 / 255 * (1 << 15) + 0.5);
        out[GU] = -(int)(0.331 * 224 / 255 * (1 << 15) + 0.5);
        out[RY] =  (int)(0.299 * 219 / 255 * (1 << 15) + 0.5);
        out[RV] =  (int)(0.500 * 224 / 255 * (1 << 15) + 0.5);
        out[RU] = -(int)(0.169 * 224 / 255 * (1 << 15) + 0.5);
    }
}

// solve_range_convert / init_range_convert_constants / ff_sws_init_range_convert (swscale.c:577-660)
void build_range_conv(RangeConv &r, int src_range, int dst_range, int dstFormat, int dstBpc)
{
    r = RangeConv();
    if (src_range == dst_range || isAnyRGB(dstFormat) || dstBpc >= 32) return;
    const int depth = dstBpc ? std::min(dstBpc, 16) : 8;
    const int src_bits = depth <= 14 ? 15 : 19;
    const int src_shift = src_bits - depth, mult_shift = depth <= 14 ? 14 : 18;
    const uint16_t mpeg_min = 16U << (depth - 8), mpeg_max_lum = 235U << (depth - 8);
    const uint16_t mpeg_max_chr = 240U << (depth - 8), jpeg_max = (1U << depth) - 1;
    auto solve = [&](uint16_t smin, uint16_t smax, uint16_t dmin, uint16_t dmax, uint32_t &coeff, int64_t &offset) {
        const uint16_t srange = smax - smin, drange = dmax - dmin;
        const int total = mult_shift + src_shift;
        const int64_t q = (int64_t)(((uint64_t)drange << total) / srange);
        coeff = (uint32_t)(-((-q) >> src_shift));            // AV_CEIL_RSHIFT
        offset = ((int64_t)dmax << total) - ((int64_t)smax << src_shift) * coeff + (1U << (mult_shift - 1));
    };
    if (src_range) {
        solve(0, jpeg_max, mpeg_min, mpeg_max_lum, r.lumCoeff, r.lumOffset);
        solve(0, jpeg_max, mpeg_min, mpeg_max_chr, r.chrCoeff, r.chrOffset);
    } else {
        solve(mpeg_min, mpeg_max_lum, 0, jpeg_max, r.lumCoeff, r.lumOffset);
        solve(mpeg_min, mpeg_max_chr, 0, jpeg_max, r.chrCoeff, r.chrOffset);
    }
    r.active = true;
}

} // namespace swship
