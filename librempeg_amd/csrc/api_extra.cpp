// Remaining public entry points of libswscale's ABI (libswscale/swscale.h, libswscale.v exports sws_* / swscale_*):
// SwsVector / SwsFilter helpers, palette expanders, capability tests and the frame/slice API.  Host code only.
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

#include "swsint.hpp"

using namespace swship;

namespace {

SwsVector *const_vec(double c, int length)   // sws_getConstVec (utils.c:2003-2017)
{
    SwsVector *v = sws_allocVec(length);
    if (!v) return nullptr;
    for (int i = 0; i < length; i++) v->coeff[i] = c;
    return v;
}
SwsVector *identity_vec() { return const_vec(1.0, 1); }   // sws_getIdentityVec (:2019-2027)

double vec_dc(const SwsVector *a) { double s = 0; for (int i = 0; i < a->length; i++) s += a->coeff[i]; return s; }  // sws_dcVec

void add_vec(SwsVector *a, const SwsVector *b)   // sws_addVec via sws_sumVec (:2052-2067, :2100-2112): centred sum
{
    const int length = a->length > b->length ? a->length : b->length;
    SwsVector *sum = const_vec(0.0, length);
    if (!sum) { for (int i = 0; i < a->length; i++) a->coeff[i] = NAN; return; }
    for (int i = 0; i < a->length; i++) sum->coeff[i + (length - 1) / 2 - (a->length - 1) / 2] += a->coeff[i];
    for (int i = 0; i < b->length; i++) sum->coeff[i + (length - 1) / 2 - (b->length - 1) / 2] += b->coeff[i];
    std::free(a->coeff);
    a->coeff = sum->coeff; a->length = sum->length;
    std::free(sum);
}
void shift_vec(SwsVector *a, int shift)          // sws_shiftVec / sws_getShiftedVec (:2069-2099)
{
    const int length = a->length + std::abs(shift) * 2;
    SwsVector *s = const_vec(0.0, length);
    if (!s) { for (int i = 0; i < a->length; i++) a->coeff[i] = NAN; return; }
    for (int i = 0; i < a->length; i++) s->coeff[i + (length - 1) / 2 - (a->length - 1) / 2 - shift] = a->coeff[i];
    std::free(a->coeff);
    a->coeff = s->coeff; a->length = s->length;
    std::free(s);
}
bool has_nan(const SwsVector *a) { for (int i = 0; i < a->length; i++) if (std::isnan(a->coeff[i])) return true; return false; }

} // namespace

extern "C" {

// ---- SwsVector / SwsFilter (utils.c:1956-2260) ----
SwsVector *sws_allocVec(int length)
{
    if (length <= 0 || (size_t)length > (size_t)INT32_MAX / sizeof(double)) return nullptr;
    SwsVector *v = (SwsVector *)std::malloc(sizeof(SwsVector));
    if (!v) return nullptr;
    v->length = length;
    v->coeff = (double *)std::malloc(sizeof(double) * (size_t)length);
    if (!v->coeff) { std::free(v); return nullptr; }
    return v;
}

void sws_freeVec(SwsVector *a)
{
    if (!a) return;
    std::free(a->coeff);
    a->coeff = nullptr; a->length = 0;
    std::free(a);
}

void sws_scaleVec(SwsVector *a, double scalar) { for (int i = 0; i < a->length; i++) a->coeff[i] *= scalar; }
void sws_normalizeVec(SwsVector *a, double height) { sws_scaleVec(a, height / vec_dc(a)); }

SwsVector *sws_getGaussianVec(double variance, double quality)
{
    const int length = (int)(variance * quality + 0.5) | 1;
    const double middle = (length - 1) * 0.5;
    if (variance < 0 || quality < 0) return nullptr;
    SwsVector *v = sws_allocVec(length);
    if (!v) return nullptr;
    for (int i = 0; i < length; i++) {
        const double dist = i - middle;
        v->coeff[i] = std::exp(-dist * dist / (2 * variance * variance)) / std::sqrt(2 * variance * M_PI);
    }
    sws_normalizeVec(v, 1.0);
    return v;
}

void sws_freeFilter(SwsFilter *f)
{
    if (!f) return;
    sws_freeVec(f->lumH); sws_freeVec(f->lumV); sws_freeVec(f->chrH); sws_freeVec(f->chrV);
    std::free(f);
}

SwsFilter *sws_getDefaultFilter(float lumaGBlur, float chromaGBlur, float lumaSharpen, float chromaSharpen,
                                float chromaHShift, float chromaVShift, int verbose)
{
    (void)verbose;
    SwsFilter *f = (SwsFilter *)std::calloc(1, sizeof(SwsFilter));
    if (!f) return nullptr;
    if (lumaGBlur != 0.0) { f->lumH = sws_getGaussianVec(lumaGBlur, 3.0); f->lumV = sws_getGaussianVec(lumaGBlur, 3.0); }
    else { f->lumH = identity_vec(); f->lumV = identity_vec(); }
    if (chromaGBlur != 0.0) { f->chrH = sws_getGaussianVec(chromaGBlur, 3.0); f->chrV = sws_getGaussianVec(chromaGBlur, 3.0); }
    else { f->chrH = identity_vec(); f->chrV = identity_vec(); }
    if (!f->lumH || !f->lumV || !f->chrH || !f->chrV) { sws_freeFilter(f); return nullptr; }
    if (chromaSharpen != 0.0) {
        SwsVector *id = identity_vec();
        if (!id) { sws_freeFilter(f); return nullptr; }
        sws_scaleVec(f->chrH, -chromaSharpen); sws_scaleVec(f->chrV, -chromaSharpen);
        add_vec(f->chrH, id); add_vec(f->chrV, id);
        sws_freeVec(id);
    }
    if (lumaSharpen != 0.0) {
        SwsVector *id = identity_vec();
        if (!id) { sws_freeFilter(f); return nullptr; }
        sws_scaleVec(f->lumH, -lumaSharpen); sws_scaleVec(f->lumV, -lumaSharpen);
        add_vec(f->lumH, id); add_vec(f->lumV, id);
        sws_freeVec(id);
    }
    if (chromaHShift != 0.0) shift_vec(f->chrH, (int)(chromaHShift + 0.5));
    if (chromaVShift != 0.0) shift_vec(f->chrV, (int)(chromaVShift + 0.5));
    sws_normalizeVec(f->chrH, 1.0); sws_normalizeVec(f->chrV, 1.0);
    sws_normalizeVec(f->lumH, 1.0); sws_normalizeVec(f->lumV, 1.0);
    if (has_nan(f->chrH) || has_nan(f->chrV) || has_nan(f->lumH) || has_nan(f->lumV)) { sws_freeFilter(f); return nullptr; }
    return f;
}

// ---- palette expanders (swscale_unscaled.c:2709-2740): host helpers, not part of the scaling path ----
void sws_convertPalette8ToPacked32(const uint8_t *src, uint8_t *dst, int num_pixels, const uint8_t *palette)
{
    for (int i = 0; i < num_pixels; i++) std::memcpy(dst + 4 * i, palette + 4 * src[i], 4);
}
void sws_convertPalette8ToPacked24(const uint8_t *src, uint8_t *dst, int num_pixels, const uint8_t *palette)
{
    for (int i = 0; i < num_pixels; i++) {
        dst[0] = palette[src[i] * 4 + 0]; dst[1] = palette[src[i] * 4 + 1]; dst[2] = palette[src[i] * 4 + 2];
        dst += 3;
    }
}

// ---- AVClass (utils.c / options.c): only the leading, layout-stable members are populated ----
struct SwsHipClass { const char *class_name; const char *(*item_name)(void *); const void *option; int version; };
static const char *sws_item_name(void *) { return "swscaler-hip"; }
const void *sws_get_class(void)
{
    static const SwsHipClass cls = { "SWScaler", sws_item_name, nullptr, (59 << 16) | (8 << 8) | 100 };
    return &cls;
}

// ---- capability tests (format.c:611-705) ----
int sws_test_format(enum AVPixelFormat format, int output) { return output ? sws_isSupportedOutput(format) : sws_isSupportedInput(format); }
int sws_test_hw_format(enum AVPixelFormat format) { return format == AV_PIX_FMT_NONE || format == AV_PIX_FMT_HIP; }
int sws_test_colorspace(int csp, int output)   // enum AVColorSpace
{
    (void)output;
    switch (csp) { case 0: /* RGB */ case 1: /* BT709 */ case 2: /* UNSPECIFIED */ case 4: /* FCC */ case 5: /* BT470BG */
                   case 6: /* SMPTE170M */ case 7: /* SMPTE240M */ case 9: /* BT2020_NCL */ return 1; }
    return 0;
}
int sws_test_primaries(int prim, int output)   // enum AVColorPrimaries: (0, NB) or the extension range, except RESERVED (3)
{
    (void)output;
    return ((prim > 0 && prim < 23) || prim == 256) && prim != 3;
}
int sws_test_transfer(int trc, int output)     // enum AVColorTransferCharacteristic with an EOTF in libavutil/csp.c
{
    (void)output;
    switch (trc) { case 2: /* UNSPECIFIED */ case 1: case 4: case 5: case 6: case 7: case 8: case 11: case 12: case 13: case 14: case 15:
                   case 16: case 17: case 18: case 256: return 1; }
    return 0;
}
} // extern "C"
