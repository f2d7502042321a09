// Translation unit of pass 2 of the two-pass element-per-thread path (vertical stage + writer over the h-scaled scratch planes), ONE DESTINATION KIND
// per compilation (-DGK_DK=n, the Makefile's GDST_PARTS): sws_k_vscale_rgb / _planar / _nvchroma<false, T, -1, DK> for 15-bit (int16) and 19-bit (int32)
// intermediates.  Same routines as the all-kinds forms in k_generic.hip, with `p.dstKind` a compile-time constant (DstKindView): one writer per kernel,
// its tap loops unrolled with the scratch loads of a column in flight together.
#include "generic_kinds.hpp"
#include "kernels_generic.hpp"

#ifndef GK_DK
#error "compile with -DGK_DK=<destination kind>"
#endif

namespace swsk {

// The packed writers behind the STRIP kernels (dev_prepare_on: fullchr_on == 4): the strip kernels leave the vertical sums of Y, U and V as int32 planes
// (DSTK_RAW32, chroma at the writer's chroma width), and this epilogue runs the writer's X form -- init + sum of taps x lines, output.c -- over them with a
// bank of the two taps {1, 0} at position y, whose "lines" are the sum planes themselves: init + S * 1 + S' * 0.  No new arithmetic: the
// routine is rgb_write_unit, the kind folded at compile time.
struct SumSampler {
    const uint8_t *pl[3]; int st[3];
    __device__ __forceinline__ int get(int comp, int row, int x) const
    {
        const int k = comp > 2 ? 0 : comp;
        return ((const int32_t *)(pl[k] + (int64_t)row * st[k]))[x];
    }
};

template <int DK>
__global__ void __launch_bounds__(256) sws_k_sum_writer(SwsFrameSet fs, SwsDevParams pa)
{
    const auto &p = kind_view<-1, DK>(pa);
    const int fi = blockIdx.z;
    const int i = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    const int units = p.full_chr ? p.dstW : (p.dstW + 1) >> 1;
    if (i >= units || y >= p.dstH) return;
    const SwsFramePtrs f = frame_copy(fs, fi);
    const SumSampler smp{{f.src[0], f.src[1], f.src[2]}, {f.srcStride[0], f.srcStride[1], f.srcStride[2]}};
    rgb_write_unit(p, smp, f, i, y);
}

} // namespace swsk

namespace swship {

#define GK_CAT_(a, b) a##b
#define GK_CAT(a, b) GK_CAT_(a, b)
void GK_CAT(generic_dst_fns_, GK_DK)(GenericDstFns *t)
{
    constexpr int DK = GK_DK;
    constexpr bool rgb = DK == DSTK_RGB24 || DK == DSTK_RGB32 || DK == DSTK_GBRP || DK == DSTK_GBRP16 || DK == DSTK_GBRPF32 || DK == DSTK_PACKED422 || DK == DSTK_PACKED444 ||
        DK == DSTK_PACKEDHI || DK == DSTK_RGB48 || DK == DSTK_RGB16 || DK == DSTK_RGB30 || DK == DSTK_MONO || DK == DSTK_RGB8 || DK == DSTK_RGB4 || DK == DSTK_YA;
    constexpr bool nv = DK == DSTK_NV12 || DK == DSTK_P010 || DK == DSTK_P016;
    if constexpr (DK == DSTK_RGB16 || DK == DSTK_RGB30 || DK == DSTK_PACKED444 || DK == DSTK_PACKEDHI || DK == DSTK_GBRP16 || DK == DSTK_GBRPF32 || DK == DSTK_RGB48) t->sum_writer = swsk::sws_k_sum_writer<DK>;
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
