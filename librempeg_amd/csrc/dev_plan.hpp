// The planner's stages (dev_plan*.hip) and what they hand on to each other.  dev_prepare_on() runs them in order:
//   plan_common()   kernel parameters every path reads, and the helper passes around a packed / semi-planar / RGB side of the scaler (dev_plan.hip)
//   plan_tables()   PLAN_MAIN: the filter banks as one device blob, the line-schedule contexts (virtual lines), the same-size RGB -> YUV plans (dev_plan_tables.hip)
//   plan_strip()    PLAN_MAIN: the dot2 tile kernel and the whole marching-strip family -- geometry, plan rows, the fallbacks of the helper passes (dev_plan_strip.hip)
//   plan_tile()     PLAN_MAIN: the LDS-tile kernel's geometry, the packed writers' per-row forms, the same-size packed-RGB march (dev_plan_tile.hip)
//   plan_name()     the path / dominant-kernel names tests and profiles match on (dev_plan.hip)
#pragma once
#include "dev_internal.hpp"

namespace swship {

struct PlanBuild {
    SwsInternal *c; DeviceState *d;
    const PixDesc *ds, *dd;
    FilterBank fastL, fastC;           // SWS_FAST_BILINEAR as two-tap banks (plan_common)
    FilterBank vChrJ;                  // the chroma bank of a packed 4:2:2 destination whose rows take yuv2422_1_c_template's blend
    const FilterBank *hLumB = nullptr, *hChrB = nullptr, *vChrB = nullptr;   // the banks every plan is made on
    bool fast_banks = false, join_short = false, striprgb_short = false, rgb2rgb_short = false, fast_flag = false, gray_any = false, fc_alpha = false,
         alpha_planar = false, long_taps = false, fc_plain = false, lut_gray = false, lut_u16 = false, lut_kind = false, lut_rgbsrc = false, wide_gbrp = false;
    int strip_min_w_eff = 0;
};

// the handles every stage starts from ...
#define PLAN_HANDLES(B) \
    SwsInternal *c = (B).c; DeviceState *d = (B).d; SwsDevParams &p = (B).d->params; const SwsContext &o = (B).c->opts; const PixDesc *ds = (B).ds, *dd = (B).dd; \
    (void)ds; (void)dd; (void)o; (void)p
// ... and the names plan_common() decided under, as the later stages use them
#define PLAN_LOCALS(B) \
    PLAN_HANDLES(B); \
    const FilterBank &hLumB = *(B).hLumB, &hChrB = *(B).hChrB, &vChrB = *(B).vChrB; \
    const bool fast_banks = (B).fast_banks, fast_flag = (B).fast_flag, gray_any = (B).gray_any, fc_alpha = (B).fc_alpha, alpha_planar = (B).alpha_planar, long_taps = (B).long_taps, \
               fc_plain = (B).fc_plain, lut_gray = (B).lut_gray, lut_u16 = (B).lut_u16, lut_kind = (B).lut_kind, lut_rgbsrc = (B).lut_rgbsrc, wide_gbrp = (B).wide_gbrp; \
    const int strip_min_w_eff = (B).strip_min_w_eff; \
    bool &join_short = (B).join_short, &striprgb_short = (B).striprgb_short, &rgb2rgb_short = (B).rgb2rgb_short; \
    (void)hLumB; (void)hChrB; (void)vChrB; (void)fast_banks; (void)fast_flag; (void)gray_any; (void)fc_alpha; (void)alpha_planar; (void)long_taps; (void)fc_plain; (void)lut_gray; \
    (void)lut_u16; (void)lut_kind; (void)lut_rgbsrc; (void)wide_gbrp; (void)strip_min_w_eff; (void)join_short; (void)striprgb_short; (void)rgb2rgb_short

int plan_common(PlanBuild &B);
int plan_tables(PlanBuild &B);
int plan_strip(PlanBuild &B);
int plan_tile(PlanBuild &B);
void plan_name(PlanBuild &B);

// dev_plan.hip
struct VLines { std::vector<int32_t> lum, chr, lumPos, chrPos; bool uniform = true; };
void build_vlines(const SwsInternal *c, int mode, VLines &out);
bool bank_is_identity(const FilterBank &b, int one);
bool bank_taps_all(const FilterBank &b, int one);

} // namespace swship
