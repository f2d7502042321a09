// Translation unit of pass 2 of the two-pass element-per-thread path (vertical stage + writer over the h-scaled scratch planes), ONE DESTINATION KIND
// per compilation (-DGK_DK=n, the Makefile's GDST_PARTS): sws_k_vscale_rgb / _planar / _nvchroma<false, T, -1, DK> for 15-bit (int16) and 19-bit (int32)
// intermediates.  Same routines as the all-kinds forms in k_generic.hip, with `p.dstKind` a compile-time constant (DstKindView): one writer per kernel,
// its tap loops unrolled with the scratch loads of a column in flight together.
#include "generic_kinds.hpp"
#include "kernels_generic.hpp"

#ifndef GK_DK
#error "compile with -DGK_DK=<destination kind>"
#endif

namespace swship {

#define GK_CAT_(a, b) a##b
#define GK_CAT(a, b) GK_CAT_(a, b)
void GK_CAT(generic_dst_fns_, GK_DK)(GenericDstFns *t)
{
    constexpr int DK = GK_DK;
    constexpr bool rgb = DK == DSTK_RGB24 || DK == DSTK_RGB32 || DK == DSTK_GBRP || DK == DSTK_GBRP16 || DK == DSTK_GBRPF32 || DK == DSTK_PACKED422 || DK == DSTK_PACKED444 ||
        DK == DSTK_PACKEDHI || DK == DSTK_RGB48 || DK == DSTK_RGB16 || DK == DSTK_RGB30 || DK == DSTK_MONO || DK == DSTK_RGB8 || DK == DSTK_RGB4 || DK == DSTK_YA;
    constexpr bool nv = DK == DSTK_NV12 || DK == DSTK_P010 || DK == DSTK_P016;
    if constexpr (rgb) {
        t->rgb16 = swsk::sws_k_vscale_rgb<false, int16_t, -1, DK>;
        t->rgb32 = swsk::sws_k_vscale_rgb<false, int32_t, -1, DK>;
    } else {
        t->planar16 = swsk::sws_k_vscale_planar<false, int16_t, -1, DK>;
        t->planar32 = swsk::sws_k_vscale_planar<false, int32_t, -1, DK>;
        if constexpr (nv) {
            t->nvchroma16 = swsk::sws_k_vscale_nvchroma<false, int16_t, -1, DK>;
            t->nvchroma32 = swsk::sws_k_vscale_nvchroma<false, int32_t, -1, DK>;
        }
    }
}

} // namespace swship
