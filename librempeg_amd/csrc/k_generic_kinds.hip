// Translation unit of the element-per-thread kernels that hold a READER, ONE SOURCE KIND per compilation (-DGK_SK=n, the Makefile's GKIND_PARTS):
//   * the single-pass kernels (identity horizontal filters) sws_k_vscale_rgb / _planar / _nvchroma<true, int16_t, SK, DK> for every destination kind,
//   * pass 1 of the two-pass path, sws_k_hscale<T, SK>,
//   * the fused h + v tile kernel sws_k_tile_planar<HT, CHROMA, SK> (its staging phase is a loop around the reader).
// Same routines as the all-kinds forms in k_generic.hip (kernels_generic.hpp), with `p.srcKind` / `p.dstKind` compile-time constants (KindView): a
// pair's single-pass kernel is 1 500 - 3 000 instructions in 30 - 110 registers instead of 24 000 in 256 + 256 with spills, i.e. 4 - 8 waves per SIMD
// instead of one, and the tap loops around a reader of a few instructions are unrolled with their loads in flight together.
#include "generic_kinds.hpp"
#include "kernels_generic.hpp"
#include "kernels_tile.hpp"

#ifndef GK_SK
#error "compile with -DGK_SK=<source kind>"
#endif

namespace swship {

template <int DK> static constexpr bool gk_rgb_kind =
    DK == DSTK_RGB24 || DK == DSTK_RGB32 || DK == DSTK_GBRP || DK == DSTK_GBRP16 || DK == DSTK_GBRPF32 || DK == DSTK_PACKED422 || DK == DSTK_PACKED444 ||
    DK == DSTK_PACKEDHI || DK == DSTK_RGB48 || DK == DSTK_RGB16 || DK == DSTK_RGB30 || DK == DSTK_MONO || DK == DSTK_RGB8 || DK == DSTK_RGB4 || DK == DSTK_YA;
template <int DK> static constexpr bool gk_nv_kind = DK == DSTK_NV12 || DK == DSTK_P010 || DK == DSTK_P016;

template <int DK>
static void gk_fill(GenericKindFns::Direct *t)
{
    if constexpr (DK <= DSTK_RAW32) {
        if constexpr (DK != DSTK_RAW32) {   // (the strip kernels' int32 sum planes are not a destination of these kernels)
            if constexpr (gk_rgb_kind<DK>) t[DK].rgb = swsk::sws_k_vscale_rgb<true, int16_t, GK_SK, DK>;
            else {
                t[DK].planar = swsk::sws_k_vscale_planar<true, int16_t, GK_SK, DK>;
                if constexpr (gk_nv_kind<DK>) t[DK].nvchroma = swsk::sws_k_vscale_nvchroma<true, int16_t, GK_SK, DK>;
            }
        }
        gk_fill<DK + 1>(t);
    }
}

#define GK_CAT_(a, b) a##b
#define GK_CAT(a, b) GK_CAT_(a, b)
void GK_CAT(generic_kind_fns_, GK_SK)(GenericKindFns *t)
{
    gk_fill<0>(t->direct);
    t->hscale16 = swsk::sws_k_hscale<int16_t, GK_SK>;
    t->hscale32 = swsk::sws_k_hscale<int32_t, GK_SK>;
    t->tile[0][0] = swsk::sws_k_tile_planar<int16_t, false, GK_SK>; t->tile[0][1] = swsk::sws_k_tile_planar<int16_t, true, GK_SK>;
    t->tile[1][0] = swsk::sws_k_tile_planar<int32_t, false, GK_SK>; t->tile[1][1] = swsk::sws_k_tile_planar<int32_t, true, GK_SK>;
}

} // namespace swship
