// Interface between k_generic.hip (the launcher of the element-per-thread path) and k_generic_kinds.hip (its kernels, one object per source kind).
#pragma once
#include "devstate.hpp"

namespace swship {

// where the reader pre-pass of a scaled source puts its 16-bit Y / U / V lines (k_strip.hip launch_rgbread_strip: one working picture per frame)
struct Read16Layout { uint8_t *base; int64_t frame_bytes, offU, offV; int32_t strideY, strideC;
                      int32_t vec; };   // vec: pixels per thread of the kind's vector form (8 / 4; the host checked widths and alignment and sized the grid for it), 0: one chroma column per thread

// the kernels of ONE source kind
struct GenericKindFns {
    // single pass (identity horizontal filters), per destination kind; null where the kernel does not write that kind
    struct Direct {
        void (*rgb)(SwsFrameSet, SwsDevParams, const int16_t *, int64_t);
        void (*planar)(SwsFrameSet, SwsDevParams, const int16_t *, int64_t, int);
        void (*nvchroma)(SwsFrameSet, SwsDevParams, const int16_t *, int64_t);
    } direct[DSTK_RAW32 + 1];
    // pass 1 of the two-pass path (reader + horizontal stage + range conversion -> scratch planes): 15-bit and 19-bit intermediates
    void (*hscale16)(SwsFrameSet, SwsDevParams, int16_t *, int64_t);
    void (*hscale32)(SwsFrameSet, SwsDevParams, int32_t *, int64_t);
    // the fused h + v tile kernel (kernels_tile.hpp sws_k_tile_planar): [19-bit intermediates][chroma]
    void (*tile[2][2])(SwsFrameSet, SwsDevParams, SwsTileGeom);
    // the reader pre-pass in front of the strip kernels for the source kinds without a vector form of it (sws_k_read16_kind)
    void (*read16)(SwsFrameSet, SwsDevParams, Read16Layout);
};
const GenericKindFns *generic_kind_fns(int srcKind);   // k_generic.hip; null: no per-kind kernels for this source kind
// pass 2 of the two-pass path for ONE destination kind (k_generic_dst.hip); null where the kernel does not write that kind
struct GenericDstFns {
    void (*rgb16)(SwsFrameSet, SwsDevParams, const int16_t *, int64_t);
    void (*rgb32)(SwsFrameSet, SwsDevParams, const int32_t *, int64_t);
    void (*planar16)(SwsFrameSet, SwsDevParams, const int16_t *, int64_t, int);
    void (*planar32)(SwsFrameSet, SwsDevParams, const int32_t *, int64_t, int);
    void (*nvchroma16)(SwsFrameSet, SwsDevParams, const int16_t *, int64_t);
    void (*nvchroma32)(SwsFrameSet, SwsDevParams, const int32_t *, int64_t);
    // the packed writer of this kind over the strip kernels' int32 sum planes (sws_k_sum_writer; dev_prepare_on: fullchr_on == 4); null: not built for this kind
    void (*sum_writer)(SwsFrameSet, SwsDevParams);
};
// k_generic.hip: the epilogue of fullchr_on == 4 (J.fs holds {src = the int32 sum planes, dst = the packed picture}); tab: device memory for the one-tap bank it runs with
int launch_sum_writer(const LaunchCtx &J, int dst_kind, uint8_t *tab);
size_t sum_writer_table_bytes(int dstH);
// (every DstKind but DSTK_RAW32 = 22)
#define GENERIC_DST_PARTS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21)
#define GENERIC_DST_DECL(n) void generic_dst_fns_##n(GenericDstFns *t);
GENERIC_DST_PARTS(GENERIC_DST_DECL)
#undef GENERIC_DST_DECL
// (every SrcKind with a scaler reader: SRCK_BAYER = 19 has none)
#define GENERIC_KIND_PARTS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(20) X(21)
#define GENERIC_KIND_DECL(n) void generic_kind_fns_##n(GenericKindFns *t);
GENERIC_KIND_PARTS(GENERIC_KIND_DECL)
#undef GENERIC_KIND_DECL

} // namespace swship
