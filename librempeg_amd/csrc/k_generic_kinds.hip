// Translation unit of the element-per-thread kernels that hold a READER, ONE SOURCE KIND per compilation (-DGK_SK=n, the Makefile's GKIND_PARTS):
//   * the single-pass kernels (identity horizontal filters) sws_k_vscale_rgb / _planar / _nvchroma<true, int16_t, SK, DK> for every destination kind,
//   * pass 1 of the two-pass path, sws_k_hscale<T, SK>,
//   * the fused h + v tile kernel sws_k_tile_planar<HT, CHROMA, SK> (its staging phase is a loop around the reader).
// Same routines as the all-kinds forms in k_generic.hip (kernels_generic.hpp), with `p.srcKind` / `p.dstKind` compile-time constants (KindView): a
// pair's single-pass kernel is 1 500 - 3 000 instructions in 30 - 110 registers instead of 24 000 in 256 + 256 with spills, i.e. 4 - 8 waves per SIMD
// instead of one, and the tap loops around a reader of a few instructions are unrolled with their loads in flight together.
#include <type_traits>
#include "generic_kinds.hpp"
#include "kernels_generic.hpp"
#include "kernels_tile.hpp"

#ifndef GK_SK
#error "compile with -DGK_SK=<source kind>"
#endif

namespace swsk {

// Reader pre-pass for ONE source kind: what the reference's input stage hands to the horizontal scaler -- the reader's 16-bit line of every source row,
// Y at the picture's width, U and V at the chroma width -- as planes of a working picture, element per thread (a streaming pass: the strip kernels
// that follow read these planes like a planar 16-bit source).  Used for the RGB sources beyond the 8-bit ones, which have a vector form of this
// pass (kernels_rgbsrc.hpp sws_k_rgb_read16): x2rgb10 / x2bgr10, the 16 / 15 / 12 bpp formats, planar RGB of 9 - 14 bits -- and for the packed YUV sources of
// 10 / 12 bits (y210, xv30, xv36 ...), which it turns into the planes of the planar picture with the same lines.
template <int SK>
__global__ void __launch_bounds__(256) sws_k_read16_kind(SwsFrameSet fs, SwsDevParams pa, swship::Read16Layout lay)
{
    const auto &p = kind_view<SK, -1>(pa);
    const int fi = blockIdx.z, row = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;   // chroma column: one thread delivers the Y, U and V samples under it, so a pixel is loaded once
    if (row >= p.srcH) return;
    const SwsFramePtrs f = frame_copy(fs, fi);
    uint8_t *fb = lay.base + (int64_t)fi * lay.frame_bytes;
    uint16_t *dy = (uint16_t *)(fb + (int64_t)row * lay.strideY);
    uint16_t *du = (uint16_t *)(fb + lay.offU + (int64_t)row * lay.strideC), *dv = (uint16_t *)(fb + lay.offV + (int64_t)row * lay.strideC);
    if constexpr (SK == SRCK_GBRP16) {
        // planar RGB of 9 - 14 bits: eight pixels per thread, one 16-byte load per plane.  The arithmetic stays read_sample's: the loaded words are presented to it
        // as a one-row picture in registers.  (planar_rgb16_s16_to_uv has no half-width form: chroma column x is pixel x, input.c:1249-1270.)
        const uint8_t *pg = f.src[0] + (int64_t)row * f.srcStride[0], *pb = f.src[1] + (int64_t)row * f.srcStride[1], *pr = f.src[2] + (int64_t)row * f.srcStride[2];
        if (lay.vec) {   // (the host checked the width and the alignment of every frame of the call and sized the grid: k_stream.hip launch_rgb_read16)
            const int x0 = 8 * x;
            if (x0 >= p.srcW) return;
            union V8 { uint4 q; uint16_t h[8]; };
            V8 g, bb, r, oy, ou, ov;
            g.q = *(const uint4 *)(pg + 2 * x0); bb.q = *(const uint4 *)(pb + 2 * x0); r.q = *(const uint4 *)(pr + 2 * x0);
            SwsFramePtrs fl = f;
            fl.src[0] = (const uint8_t *)g.h; fl.src[1] = (const uint8_t *)bb.h; fl.src[2] = (const uint8_t *)r.h;
            fl.srcStride[0] = fl.srcStride[1] = fl.srcStride[2] = 0;
            const auto &q = chr_half_view<0>(p);   // (this reader has one form)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                oy.h[k] = (uint16_t)read_sample(q, fl, 0, 0, k); ou.h[k] = (uint16_t)read_sample(q, fl, 1, 0, k); ov.h[k] = (uint16_t)read_sample(q, fl, 2, 0, k);
            }
            *(uint4 *)(dy + x0) = oy.q;
            if (x0 + 8 <= p.chrSrcW) { *(uint4 *)(du + x0) = ou.q; *(uint4 *)(dv + x0) = ov.q; }
            else for (int k = 0; k < 8; k++) if (x0 + k < p.chrSrcW) { du[x0 + k] = ou.h[k]; dv[x0 + k] = ov.h[k]; }
            return;
        }
    }
    if constexpr (SK == SRCK_GBRPF32) {
        // planar float RGB (round 5: gbrpf32 sources reach the strip kernels): four pixels per thread, one 16-byte load per plane (neighbouring lanes read
        // neighbouring chunks), planar_rgbf32_to_y / _uv (input.c:1300-1334) written out -- lrintf(av_clipf(65535 x, 0, 65535)) per component, then the 15-bit matrix as
        // v_mad_i32_i24 (a 16-bit sample times a 15-bit coefficient: the low 32 bits are the reference's wrap-around sum) and the arithmetic shift; no half-width form
        const uint8_t *pg = f.src[0] + (int64_t)row * f.srcStride[0], *pb = f.src[1] + (int64_t)row * f.srcStride[1], *pr = f.src[2] + (int64_t)row * f.srcStride[2];
        if (lay.vec) {   // (host-checked, see above)
            const int x0 = 4 * x;
            if (x0 >= p.srcW) return;
            const float4 g = *(const float4 *)(pg + 4 * x0), bb = *(const float4 *)(pb + 4 * x0), r = *(const float4 *)(pr + 4 * x0);
            const float gv[4] = { g.x, g.y, g.z, g.w }, bv[4] = { bb.x, bb.y, bb.z, bb.w }, rv[4] = { r.x, r.y, r.z, r.w };
            const Rgb2YuvRow ty = rgb2yuv_row(p.rgb2yuv, 0), tu = rgb2yuv_row(p.rgb2yuv, 3), tv = rgb2yuv_row(p.rgb2yuv, 6);
            uint32_t oy[4], ou[4], ov[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int R = f32_to_u16(rv[k]), G = f32_to_u16(gv[k]), B = f32_to_u16(bv[k]);
                oy[k] = (uint32_t)(uint16_t)(mad24(R, ty.r, mad24(G, ty.g, mad24(B, ty.b, (int)(0x2001u << 14)))) >> 15);
                ou[k] = (uint32_t)(uint16_t)(mad24(R, tu.r, mad24(G, tu.g, mad24(B, tu.b, (int)(0x10001u << 14)))) >> 15);
                ov[k] = (uint32_t)(uint16_t)(mad24(R, tv.r, mad24(G, tv.g, mad24(B, tv.b, (int)(0x10001u << 14)))) >> 15);
            }
            *(uint2 *)(dy + x0) = make_uint2(oy[0] | oy[1] << 16, oy[2] | oy[3] << 16);
            *(uint2 *)(du + x0) = make_uint2(ou[0] | ou[1] << 16, ou[2] | ou[3] << 16);
            *(uint2 *)(dv + x0) = make_uint2(ov[0] | ov[1] << 16, ov[2] | ov[3] << 16);
            return;
        }
    }
    if constexpr (SK == SRCK_RGB48) {
        // rgb48 / bgr48 / rgba64 / bgra64 (round 5): eight pixels per thread, three or four 16-byte loads; rgb48ToY / UV(_half)_c_template and the 64-bit twins
        // (input.c:45-203) written over the pixel's WORDS -- the component order is a coefficient permutation (word j of a pixel times the coefficient of whichever of
        // r, g, b sits there; the alpha word times 0), so nothing is indexed at run time; 16-bit word x 15-bit coefficient as v_mad_i32_i24, whose low 32 bits are the
        // reference's unsigned wrap-around product
        const uint8_t *ps = f.src[0] + (int64_t)row * f.srcStride[0];
        const int st = U(p.s16_step);
        if (lay.vec) {   // (host-checked, see above)
            const int x0 = 8 * x;
            if (x0 >= p.srcW) return;
            const int rp = U(p.s16_r), gp = U(p.s16_g), bp = U(p.s16_b);
            const Rgb2YuvRow ty = rgb2yuv_row(p.rgb2yuv, 0), tu = rgb2yuv_row(p.rgb2yuv, 3), tv = rgb2yuv_row(p.rgb2yuv, 6);
            auto coef = [&](const Rgb2YuvRow &w, int j) { return j == rp ? w.r : j == gp ? w.g : j == bp ? w.b : 0; };
            const int cy[4] = { coef(ty, 0), coef(ty, 1), coef(ty, 2), coef(ty, 3) }, cu[4] = { coef(tu, 0), coef(tu, 1), coef(tu, 2), coef(tu, 3) },
                      cv[4] = { coef(tv, 0), coef(tv, 1), coef(tv, 2), coef(tv, 3) };
            union V8 { uint4 q; uint16_t h[8]; };
            V8 oy, ou, ov;
            auto run = [&](auto ST_) {
                constexpr int ST = decltype(ST_)::value;          // words per pixel: 3 or 4
                union W { uint4 q[ST]; uint16_t h[8 * ST]; };
                W w;
#pragma unroll
                for (int j = 0; j < ST; j++) w.q[j] = *(const uint4 *)(ps + 2 * ST * x0 + 16 * j);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    int sy = (int)(0x2001u << 14);
#pragma unroll
                    for (int j = 0; j < ST; j++) sy = mad24((int)w.h[ST * k + j], cy[j], sy);
                    oy.h[k] = (uint16_t)((unsigned)sy >> 15);
                }
                if (p.chr_half) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        int su = (int)(0x10001u << 14), sv = su;
#pragma unroll
                        for (int j = 0; j < ST; j++) {
                            const int a = (int)(((unsigned)w.h[ST * 2 * k + j] + (unsigned)w.h[ST * (2 * k + 1) + j] + 1u) >> 1);
                            su = mad24(a, cu[j], su); sv = mad24(a, cv[j], sv);
                        }
                        ou.h[k] = (uint16_t)((unsigned)su >> 15); ov.h[k] = (uint16_t)((unsigned)sv >> 15);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        int su = (int)(0x10001u << 14), sv = su;
#pragma unroll
                        for (int j = 0; j < ST; j++) { su = mad24((int)w.h[ST * k + j], cu[j], su); sv = mad24((int)w.h[ST * k + j], cv[j], sv); }
                        ou.h[k] = (uint16_t)((unsigned)su >> 15); ov.h[k] = (uint16_t)((unsigned)sv >> 15);
                    }
                }
            };
            if (st == 3) run(std::integral_constant<int, 3>{}); else run(std::integral_constant<int, 4>{});
            *(uint4 *)(dy + x0) = oy.q;
            if (p.chr_half) { *(uint2 *)(du + (x0 >> 1)) = make_uint2(ou.q.x, ou.q.y); *(uint2 *)(dv + (x0 >> 1)) = make_uint2(ov.q.x, ov.q.y); }
            else { *(uint4 *)(du + x0) = ou.q; *(uint4 *)(dv + x0) = ov.q; }
            return;
        }
    }
    if (x >= p.chrSrcW) return;
    auto body = [&](const auto &q) {
        if (p.chrSrcHSub) {   // half-width chroma (the RGB readers' half forms, packed 4:2:2): the pixel pair over chroma column x (the planner asks for chrSrcW == srcW / 2)
            const int y0 = read_sample(q, f, 0, row, 2 * x), y1 = read_sample(q, f, 0, row, 2 * x + 1);
            const int u = read_sample(q, f, 1, row, x), v = read_sample(q, f, 2, row, x);
            *(uint32_t *)(dy + 2 * x) = (uint32_t)(uint16_t)y0 | ((uint32_t)(uint16_t)y1 << 16);
            du[x] = (uint16_t)u; dv[x] = (uint16_t)v;
        } else {              // chrSrcW == srcW
            dy[x] = (uint16_t)read_sample(q, f, 0, row, x); du[x] = (uint16_t)read_sample(q, f, 1, row, x); dv[x] = (uint16_t)read_sample(q, f, 2, row, x);
        }
    };
    if (p.chr_half) body(chr_half_view<1>(p)); else body(chr_half_view<0>(p));
}

} // namespace swsk

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
    t->read16 = swsk::sws_k_read16_kind<GK_SK>;
    t->tile[1][0] = swsk::sws_k_tile_planar<int32_t, false, GK_SK>; t->tile[1][1] = swsk::sws_k_tile_planar<int32_t, true, GK_SK>;
}

} // namespace swship
