// Marching strip kernel for SCALED packed 24 / 32 bpp RGB sources into planar / semi-planar YUV with half-width chroma (the capture / render ->
// encoder shape: bgra 4K -> yuv420p 1080p, rgb24 1080p -> yuv420p 720p, bgra -> nv12 / p010): the reference's whole chain in one wave --
// rgb24ToY_c / rgb24ToUV_half_c (input.c:1068-1172) or rgb16_32ToY / UV_half_c_template with the 32-bit rows (:264-393), hScale16To15_c for luma and
// chroma (swscale.c:106-131), the vertical filters and the planar writers (yuv2planeX_8_c / yuv2planeX_10 / yuv2nv12cX_c / yuv2p01xcX_c,
// output.c:438-528 and twins; kernels_strip.hpp has the arithmetic).
//
// What it replaces: the reader pre-pass sws_k_rgb_read16 (k_stream.hip) wrote 16-bit Y / U / V planes of the SOURCE size to a working picture
// (2 + 1 + 1 bytes per source pixel written, then read again by two strip launches): 24 MB of traffic for the 7.6 MB a 1080p -> 720p frame needs.
// Here the 15-bit reader values only ever exist in the wave's LDS rows.
//
// Schedule: a wave owns a strip of 256 luma columns and the 128 chroma columns under them and walks down a band of output rows.  Luma and chroma
// march in LOCKSTEP over the source row pairs (the half readers give chroma planes of the source height: both plane classes consume every source
// row), so one set of loads feeds both:
//  * per step: the window's pixels of the two rows of a pair in groups of FOUR per lane (one 12- or 16-byte buffer load per row and group: group g
//    belongs to lane g % 64 in turn g / 64, so a turn is one fully coalesced 768- or 1024-byte wave load; a 1.5:1 strip's 400 pixels take two turns,
//    a 2:1 strip's 528 three -- the turn count is wave-uniform and the code of unused turns is branched over; lanes of 16 pixels, the first form,
//    left 60 % of the lanes converting zeros), the next pair's loads in flight while this one is computed -> per group 4 Y + 2 U + 2 V reader
//    values per row (v_perm byte pairs, v_dot2_i32_i16 against per-byte-position coefficient pairs with the rounding constant as the addend),
//    written as u16 rows of the strip's window (the same rows the 16-bit strip kernel stages);
//  * the horizontal stage (strip_hstage: v_dot2_i32_i16 against taps held in registers) turns the pair into one packed dword per column and
//    component and pushes it into the register rings (luma RL deep, chroma RC deep);
//  * every luma / chroma output row whose last source pair this was leaves through the vertical stage (scalar tap pairs, 64-byte plan entries)
//    and the writers of kernels_strip.hpp; rows are emitted in the order their last pair arrives, so each ring's newest entry is always the
//    emitted row's last pair.
// The plan tables are the ones the two-pass form uses (d->stripL / d->stripC: windows in reader samples); the pixel window of a strip is the union
// of its luma window and twice its chroma window, rounded to 16 pixels.
// Bound: instruction issue (rocprofv3 PMC, E1 = rgb24 1080p -> yuv420p 720p: 1.66 M vector instructions per frame, two thirds of them the reader
// arithmetic before the group form; VALU busy 68 % at two waves per SIMD).  What moved it: groups of four pixels (-45 % reader instructions),
// instantiations with two turns for windows of up to 512 pixels (162 VGPRs: three waves per SIMD), constants folded into the dot2 addends.
// What did not: two row pairs in flight (+32 VGPRs, same time -- it is not latency).
#pragma once
#include <type_traits>

#include "kernels_strip.hpp"

namespace swsk {

typedef uint32_t rsrc_u32x3 __attribute__((ext_vector_type(3)));

// one plane class's destination: descriptors, per-lane offsets, the pending row (strip_body's writers)
template <bool CHROMA, int COLS>
struct StripOut {
    static constexpr int NCOMP = CHROMA ? 2 : 1;
    sws_rsrc_t rd[NCOMP];
    int dstr[NCOMP], doff[COLS], kind, pend_y, xs;
    uint32_t pend[NCOMP][COLS];
};

template <bool CHROMA, int COLS>
__device__ __forceinline__ void so_init(StripOut<CHROMA, COLS> &O, const FrameRegs &f, const SwsDevParams &p, int xs, int lane, int tw)
{
    constexpr int NCOMP = CHROMA ? 2 : 1;
    const int W = CHROMA ? p.chrDstW : p.dstW, H = CHROMA ? p.chrDstH : p.dstH;
    const bool semi = CHROMA && (p.dstKind == DSTK_NV12 || p.dstKind == DSTK_P010);
    const bool d8 = p.dstKind == DSTK_PLANAR8 || p.dstKind == DSTK_NV12;
    O.kind = semi ? (d8 ? 2 : 3) : (d8 ? 0 : 1);               // store form: b8 / b16 per component, or b16 / b32 of an interleaved pair
    const int dbytes = (d8 ? 1 : 2) * (semi ? 2 : 1);
#pragma unroll
    for (int ci = 0; ci < NCOMP; ci++) {
        const int pl = !CHROMA ? 0 : semi ? 1 : (ci == 0 ? p.u_plane_dst : p.v_plane_dst);
        uint8_t *db = pl == 0 ? U(f.dst[0]) : pl == 1 ? U(f.dst[1]) : U(f.dst[2]);
        O.dstr[ci] = pl == 0 ? U(f.dstStride[0]) : pl == 1 ? U(f.dstStride[1]) : U(f.dstStride[2]);
        O.rd[ci] = make_rsrc(db, (uint32_t)O.dstr[ci] * (uint32_t)(H - 1) + (uint32_t)W * (uint32_t)dbytes);
    }
#pragma unroll
    for (int c = 0; c < COLS; c++) {
        const int x = xs + 64 * c + lane;
        O.doff[c] = (x < W && 64 * c + lane < tw) ? x * dbytes : 0x7fffffff;     // (columns beyond the strip belong to the next one)
    }
    O.pend_y = -1; O.xs = xs;
}

template <bool CHROMA, int COLS>
__device__ __forceinline__ void so_flush(StripOut<CHROMA, COLS> &O)
{
    constexpr int NCOMP = CHROMA ? 2 : 1;
    if (O.pend_y < 0) return;
    switch (O.kind) {
    case 0:
#pragma unroll
        for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
            for (int c = 0; c < COLS; c++) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)O.pend[ci][c], O.rd[ci], O.doff[c], O.pend_y * O.dstr[ci], 0);
        break;
    case 1:
#pragma unroll
        for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
            for (int c = 0; c < COLS; c++) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)O.pend[ci][c], O.rd[ci], O.doff[c], O.pend_y * O.dstr[ci], 0);
        break;
    case 2:
#pragma unroll
        for (int c = 0; c < COLS; c++)
            __builtin_amdgcn_raw_buffer_store_b16((uint16_t)(O.pend[0][c] | (O.pend[NCOMP - 1][c] << 8)), O.rd[0], O.doff[c], O.pend_y * O.dstr[0], 0);
        break;
    default:
#pragma unroll
        for (int c = 0; c < COLS; c++)
            __builtin_amdgcn_raw_buffer_store_b32(O.pend[0][c] | (O.pend[NCOMP - 1][c] << 16), O.rd[0], O.doff[c], O.pend_y * O.dstr[0], 0);
        break;
    }
    O.pend_y = -1;
}

// the vertical sums of output row y through the "X" writers into the pending row
template <bool CHROMA, int COLS>
__device__ __forceinline__ void so_put(StripOut<CHROMA, COLS> &O, const SwsDevParams &p, const int (&acc)[CHROMA ? 2 : 1][COLS], int y, int lane)
{
    constexpr int NCOMP = CHROMA ? 2 : 1;
    if (O.kind == 0 || O.kind == 2) {
#pragma unroll
        for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
            for (int c = 0; c < COLS; c++) {
                const int x = O.xs + 64 * c + lane;
                const int off = (CHROMA && ci == 1) ? 3 : 0;
                O.pend[ci][c] = (uint32_t)clip_u8_shr((dither8(p.should_dither, y, x + off) << 12) + acc[ci][c], 19);
            }
    } else {
        const int bits = p.dst_bits, shift = 11 + 16 - bits, osh = p.dst_shift;
#pragma unroll
        for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
            for (int c = 0; c < COLS; c++)
                O.pend[ci][c] = (uint32_t)(clip_uintp2(((1 << (shift - 1)) + acc[ci][c]) >> shift, bits) << osh);
    }
    if (CHROMA && O.kind >= 2 && p.uv_swap_dst) {
#pragma unroll
        for (int c = 0; c < COLS; c++) { const uint32_t t = O.pend[0][c]; O.pend[0][c] = O.pend[NCOMP - 1][c]; O.pend[NCOMP - 1][c] = t; }
    }
    O.pend_y = y;
}

// vertical stage: the npv newest ring entries against the row's tap pairs
template <int NCOMP, int COLS, int RD>
__device__ __forceinline__ void strip_vstage_n(const uint32_t (&ring)[NCOMP][COLS][RD], const StripRowN<RD> &e, int npv, int (&acc)[NCOMP][COLS])
{
    switch (npv) {
#define SWS_SVN(N) case N: if constexpr (N < RD) { \
        _Pragma("unroll") for (int ci = 0; ci < NCOMP; ci++) _Pragma("unroll") for (int c = 0; c < COLS; c++) { \
            acc[ci][c] = sdot2_first_s(ring[ci][c][RD - N < 0 ? 0 : RD - N], e.vt[0]); \
            _Pragma("unroll") for (int k = 1; k < N; k++) acc[ci][c] = sdot2(ring[ci][c][RD - N + k < 0 ? 0 : RD - N + k], e.vt[k], acc[ci][c]); } \
        break; } [[fallthrough]];
    SWS_SVN(1) SWS_SVN(2) SWS_SVN(3) SWS_SVN(4) SWS_SVN(5) SWS_SVN(6) SWS_SVN(7) SWS_SVN(8) SWS_SVN(9) SWS_SVN(10) SWS_SVN(11)
#undef SWS_SVN
    default:
#pragma unroll
        for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
            for (int c = 0; c < COLS; c++) {
                acc[ci][c] = sdot2_first_s(ring[ci][c][0], e.vt[0]);
#pragma unroll
                for (int k = 1; k < RD; k++) acc[ci][c] = sdot2(ring[ci][c][k], e.vt[k], acc[ci][c]);
            }
        break;
    }
}

// four pixels of one source row (one 12- or 16-byte load) -> 2 dwords of Y pairs, one dword of U, one of V (the readers' 16-bit values)
struct RgbReadCoefs { uint32_t yA, yB, uA, uB, vA, vB; int ky, kc; };

template <int BPP, typename GT>
__device__ __forceinline__ void rgb4px_read(const GT &d, const RgbReadCoefs &k, u32x2 &yo, uint32_t &uo, uint32_t &vo)
{
    if constexpr (BPP == 2) {       // packed 8-bit 4:2:2 (yuyv422 / uyvy422 / yvyu422: yuy2ToY_c / yuy2ToUV_c / uyvyTo*_c / yvy2ToUV_c, input.c:550-578, :890-907): bytes, picked by selectors
        yo[0] = __builtin_amdgcn_perm(0, d[0], k.yA); yo[1] = __builtin_amdgcn_perm(0, d[1], k.yA);
        uo = __builtin_amdgcn_perm(d[1], d[0], k.uA); vo = __builtin_amdgcn_perm(d[1], d[0], k.vA);
        return;
    } else if constexpr (BPP == 30) {
        // x2rgb10le / x2bgr10le (round 5: HDR desktop capture): rgb16_32ToY_c_template / rgb16_32ToUV_half_c_template with the rgb30le / bgr30le rows (input.c:264-372,
        // :411-412; S = RGB2YUV_SHIFT + 6).  A pixel's fields: T = bits 29:20, G = bits 19:10, B = bits 9:0; the reference takes T and G shifted left by four with the
        // plain coefficient and the low field as it is with 16 x its coefficient -- {T << 4, G << 4} are v_dot2 operands (14 bits; 15 for the half readers' pair sums),
        // the low field goes through v_mad_i32_i24 (10 / 11 bits x 19 bits).  k.yA / uA / vA = {coef(T), coef(G)} packed, k.yB / uB / vB = 16 x coef(low field);
        // x2rgb10: T is red, x2bgr10: T is blue.  32-bit wrap-around sums like the C code, the results are the low 16 bits of the shifted sums.
        uint32_t yv[4], uu[2], vv[2];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t px = d[i];
            const uint32_t tg = ((px >> 16) & 0x3FF0u) | (((px >> 6) & 0x3FF0u) << 16);
            const int S = mad24((int)(px & 0x3FFu), (int)k.yB, sdot2(tg, k.yA, k.ky));
            yv[i] = (uint32_t)S >> 15;                                        // (sum + (32 << 20) + (1 << 14)) >> 15
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const uint32_t p0 = d[2 * j], p1 = d[2 * j + 1];
            const uint32_t g2 = (p0 & 0xC00FFC00u) + (p1 & 0xC00FFC00u);       // maskgx = ~(maskr | maskb): the G field and the two X bits
            const uint32_t rb = p0 + p1 - g2;
            const uint32_t tg = ((rb >> 16) & 0x7FF0u) | (((g2 >> 6) & 0x7FF0u) << 16);   // (rb & (mask | mask << 1)) >> 16, (g & (maskg | maskg << 1)) >> 6
            const int lowf = (int)(rb & 0x7FFu);
            const int Su = mad24(lowf, (int)k.uB, sdot2(tg, k.uA, k.kc)), Sv = mad24(lowf, (int)k.vB, sdot2(tg, k.vA, k.kc));
            uu[j] = (uint32_t)Su >> 16; vv[j] = (uint32_t)Sv >> 16;          // (sum + (256 << 21) + (1 << 15)) >> 16
        }
        yo[0] = __builtin_amdgcn_perm(yv[1], yv[0], 0x05040100u); yo[1] = __builtin_amdgcn_perm(yv[3], yv[2], 0x05040100u);
        uo = __builtin_amdgcn_perm(uu[1], uu[0], 0x05040100u); vo = __builtin_amdgcn_perm(vv[1], vv[0], 0x05040100u);
        return;
    } else {
    uint32_t lo[4], hi[4];          // per pixel: {byte 0, byte 2} and {byte 1, byte 3 (0 for 24 bpp)} as 16-bit halves
    if constexpr (BPP == 4) {
#pragma unroll
        for (int i = 0; i < 4; i++) { lo[i] = d[i] & 0x00FF00FFu; hi[i] = __builtin_amdgcn_perm(0, d[i], 0x0c030c01u); }
    } else if constexpr (BPP == 0) {   // planar 8-bit G, B, R planes (gbrp, gbrap without its alpha: planar_rgb_to_y / gbr24pToUV_half_c, input.c:1174-1186, :414-432): d[0] = four G, d[1] = four B, d[2] = four R; a pixel is assembled as {R, B} / {G} halves, i.e. r, g, b at bytes 0, 1, 2
        lo[0] = __builtin_amdgcn_perm(d[1], d[2], 0x0c040c00u); hi[0] = __builtin_amdgcn_perm(d[0], d[0], 0x0c0c0c00u);
        lo[1] = __builtin_amdgcn_perm(d[1], d[2], 0x0c050c01u); hi[1] = __builtin_amdgcn_perm(d[0], d[0], 0x0c0c0c01u);
        lo[2] = __builtin_amdgcn_perm(d[1], d[2], 0x0c060c02u); hi[2] = __builtin_amdgcn_perm(d[0], d[0], 0x0c0c0c02u);
        lo[3] = __builtin_amdgcn_perm(d[1], d[2], 0x0c070c03u); hi[3] = __builtin_amdgcn_perm(d[0], d[0], 0x0c0c0c03u);
    } else {
        lo[0] = __builtin_amdgcn_perm(d[0], d[0], 0x0c020c00u); hi[0] = __builtin_amdgcn_perm(d[0], d[0], 0x0c0c0c01u);
        lo[1] = __builtin_amdgcn_perm(d[1], d[0], 0x0c050c03u); hi[1] = __builtin_amdgcn_perm(d[1], d[0], 0x0c0c0c04u);
        lo[2] = __builtin_amdgcn_perm(d[2], d[1], 0x0c040c02u); hi[2] = __builtin_amdgcn_perm(d[2], d[1], 0x0c0c0c03u);
        lo[3] = __builtin_amdgcn_perm(d[2], d[2], 0x0c030c01u); hi[3] = __builtin_amdgcn_perm(d[2], d[2], 0x0c0c0c02u);
    }
    // the readers' rounding constants ride in the first dot2's addend (k.ky, k.kc); the results are 16-bit values (uint16_t stores in the
    // reference: the low halves of the shifted sums, which is what the v_perm packing takes)
    uint32_t yv[4], uu[2], vv[2];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int S = sdot2(lo[i], k.yA, sdot2(hi[i], k.yB, k.ky));
        if constexpr (BPP != 4) yv[i] = (uint32_t)S >> 9;                     // (S + (32 << 14) + (1 << 8)) >> 9
        else yv[i] = __builtin_amdgcn_ubfe((uint32_t)S, 9, 15);               // (((unsigned)S << 8) + ((32u << 22) + (1u << 16))) >> 17: the same constant, times 256
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {      // the half readers: sums of a pixel pair stay inside their 16-bit halves
        const uint32_t al = lo[2 * j] + lo[2 * j + 1], ah = hi[2 * j] + hi[2 * j + 1];
        const int Su = sdot2(al, k.uA, sdot2(ah, k.uB, k.kc)), Sv = sdot2(al, k.vA, sdot2(ah, k.vB, k.kc));
        if constexpr (BPP != 4) { uu[j] = (uint32_t)Su >> 10; vv[j] = (uint32_t)Sv >> 10; }          // (S + (256 << 15) + (1 << 9)) >> 10
        else { uu[j] = __builtin_amdgcn_ubfe((uint32_t)Su, 10, 14); vv[j] = __builtin_amdgcn_ubfe((uint32_t)Sv, 10, 14); }   // (((unsigned)S << 8) + (256u << 23) + (1u << 17)) >> 18
    }
    yo[0] = __builtin_amdgcn_perm(yv[1], yv[0], 0x05040100u); yo[1] = __builtin_amdgcn_perm(yv[3], yv[2], 0x05040100u);
    uo = __builtin_amdgcn_perm(uu[1], uu[0], 0x05040100u); vo = __builtin_amdgcn_perm(vv[1], vv[0], 0x05040100u);
    }
}

template <int BPP, int NPH, int RL, int RC, int NG>
__device__ __forceinline__ void strip_rgbsrc_body(const FrameRegs &f, const SwsDevParams &p, const SwsStripGeom &gl, const SwsStripGeom &gc,
                                                  int strip, int y0, int y1, uint32_t *lds, int npx_max, int lane)
{
    constexpr int CL = 4, CC = 2;
    // NG: groups of four pixels per lane and source row the instantiation takes (group g of the window: lane g % 64, turn g / 64): 2 = windows of
    // up to 512 pixels (ratios up to about 1.9:1), 4 = up to 1024; the staging registers are 2 * NG * 3 or 4 dwords
    typedef typename std::conditional<BPP == 4 || BPP == 30, u32x4, typename std::conditional<BPP == 2, u32x2, rsrc_u32x3>::type>::type GT;
    const int W = p.dstW, H = p.dstH, cW = p.chrDstW, cH = p.chrDstH, sH = p.srcH, sh = p.hshift;
    const StripRange rngL = strip_range_of(p, false), rngC = strip_range_of(p, true);
    const int vs = p.chrDstVSub;
    const int cy0 = y0 >> vs, cy1 = min(cH, (y1 + (1 << vs) - 1) >> vs);
    // the strip's pixel window: luma window and twice the chroma window, from a multiple of 16 pixels on, in groups of four pixels
    const int csL = gl.colStart[strip], eL = csL + gl.colCount[strip], csC = gc.colStart[strip], eC = csC + gc.colCount[strip];
    const int w0 = min(csL, 2 * csC) & ~15;
    const int n4 = (max(eL, 2 * eC) - w0 + 3) >> 2;             // (<= 256: host check)
    const int ng = __builtin_amdgcn_readfirstlane((n4 + 63) >> 6);                              // turns this strip takes (wave-uniform): 384 + 16 pixels of a 1.5:1 strip are two, 512 + 16 of a 2:1 strip three
    StripLds LL, LC;
    LL.row_dw = (npx_max + 16) >> 1; LC.row_dw = (npx_max + 16) >> 2;     // one spare lane's worth per row: the dump slot of idle lanes
    LL.S = lds; LC.S = lds + 2 * LL.row_dw;

    int spdL[CL], spdC[CC];
    uint32_t htL[CL][NPH], htC[CC][NPH];
    {
        const int ndL = gl.hfs2 >> 1, ndC = gc.hfs2 >> 1;       // dwords per tap row
#pragma unroll
        for (int c = 0; c < CL; c++) {
            const int x = min(min(strip * gl.TW + 64 * c + lane, (strip + 1) * gl.TW - 1), W - 1);     // (gl.TW <= 256, gc.TW = gl.TW / 2: the host narrows the strips where that saves a reader turn)
            spdL[c] = ((p.hLumPos[x] & ~1) - w0) >> 1;
            const uint32_t *tp = (const uint32_t *)(gl.hT2 + (int64_t)x * gl.hfs2);
#pragma unroll
            for (int k = 0; k < NPH; k++) htL[c][k] = k < ndL ? tp[k] : 0u;
        }
#pragma unroll
        for (int c = 0; c < CC; c++) {
            const int x = min(min(strip * gc.TW + 64 * c + lane, (strip + 1) * gc.TW - 1), cW - 1);
            spdC[c] = ((p.hChrPos[x] & ~1) - (w0 >> 1)) >> 1;
            const uint32_t *tp = (const uint32_t *)(gc.hT2 + (int64_t)x * gc.hfs2);
#pragma unroll
            for (int k = 0; k < NPH; k++) htC[c][k] = k < ndC ? tp[k] : 0u;
        }
    }
    // reader coefficients per byte position, packed for v_dot2_i32_i16 against {byte 0, byte 2} / {byte 1, byte 3} halves (kernels_rgbsrc.hpp)
    RgbReadCoefs rk;
    if constexpr (BPP == 2) {   // byte selectors: Y0 | Y1 << 16 of a dword; U (V) of two neighbouring dwords
        const uint32_t y = (uint32_t)U(p.s422_y), u = (uint32_t)U(p.s422_u), v = (uint32_t)U(p.s422_v);
        rk.yA = 0x0c000c00u | y | (y + 2) << 16;
        rk.uA = 0x0c000c00u | u | (u + 4) << 16; rk.vA = 0x0c000c00u | v | (v + 4) << 16;
        rk.yB = rk.uB = rk.vB = 0;
    } else if constexpr (BPP == 30) {
        const Rgb2YuvRow ty = rgb2yuv_row(p.rgb2yuv, 0), tu = rgb2yuv_row(p.rgb2yuv, 3), tv = rgb2yuv_row(p.rgb2yuv, 6);
        const bool x2rgb = U(p.s16_is565) != 0;                 // (x2rgb10le: red in the top field; x2bgr10le: blue)
        auto top = [&](const Rgb2YuvRow &w) { return (uint32_t)(uint16_t)(x2rgb ? w.r : w.b); };
        auto low = [&](const Rgb2YuvRow &w) { return (uint32_t)(16 * (x2rgb ? w.b : w.r)); };
        rk.yA = top(ty) | (uint32_t)(uint16_t)ty.g << 16; rk.yB = low(ty);
        rk.uA = top(tu) | (uint32_t)(uint16_t)tu.g << 16; rk.uB = low(tu);
        rk.vA = top(tv) | (uint32_t)(uint16_t)tv.g << 16; rk.vB = low(tv);
    } else {
        const Rgb2YuvRow ty = rgb2yuv_row(p.rgb2yuv, 0), tu = rgb2yuv_row(p.rgb2yuv, 3), tv = rgb2yuv_row(p.rgb2yuv, 6);
        const int rp = BPP == 0 ? 0 : U(p.src_r_pos), gp = BPP == 4 ? U(p.src_g_pos) : 1, bp = BPP == 0 ? 2 : U(p.src_b_pos);
        auto coef = [&](const Rgb2YuvRow &w, int k) { return (uint32_t)(uint16_t)(k == rp ? w.r : k == gp ? w.g : k == bp ? w.b : 0); };
        rk.yA = coef(ty, 0) | coef(ty, 2) << 16; rk.yB = coef(ty, 1) | coef(ty, 3) << 16;
        rk.uA = coef(tu, 0) | coef(tu, 2) << 16; rk.uB = coef(tu, 1) | coef(tu, 3) << 16;
        rk.vA = coef(tv, 0) | coef(tv, 2) << 16; rk.vB = coef(tv, 1) | coef(tv, 3) << 16;
    }
    const int sst = f.srcStride[0];
    const sws_rsrc_t rs = make_rsrc(f.src[0], (uint32_t)sst * (uint32_t)sH);
    // (planar RGB: the B and R planes; packed sources never touch these)
    const int sst1 = BPP == 0 ? U(f.srcStride[1]) : sst, sst2 = BPP == 0 ? U(f.srcStride[2]) : sst;
    const sws_rsrc_t rs1 = BPP == 0 ? make_rsrc(f.src[1], (uint32_t)sst1 * (uint32_t)sH) : rs, rs2 = BPP == 0 ? make_rsrc(f.src[2], (uint32_t)sst2 * (uint32_t)sH) : rs;
    constexpr int PXB = BPP == 30 ? 4 : BPP ? BPP : 1;          // bytes per pixel and plane
    const int vbase = (w0 + 4 * lane) * PXB;
    rk.ky = (32 << 14) + (1 << 8); rk.kc = (256 << 15) + (1 << 9);
    if constexpr (BPP == 30) { rk.ky = (int)((32u << 20) + (1u << 14)); rk.kc = (int)((256u << 21) + (1u << 15)); }

    // ONE row pair in flight per wave, requested before the pair in LDS is h-scaled (two in flight, measured: no gain -- at 2 - 3 waves per SIMD the
    // kernel is bound by instruction issue, not by latency -- and 24 - 32 registers)
    GT pre[2][NG];
    auto prefetch = [&](int q) {
        const int r0 = min(max(2 * q, 0), sH - 1), r1 = min(max(2 * q + 1, 0), sH - 1);
#pragma unroll
        for (int j = 0; j < NG; j++)
            if (j < ng) {
                // (beyond the window: the descriptor answers 0 without touching memory)
                const int vo = lane + 64 * j < n4 ? vbase + j * (256 * PXB) : 0x7fffffff;
                if constexpr (BPP == 4 || BPP == 30) {
                    pre[0][j] = bload16(rs, vo, r0 * sst);
                    pre[1][j] = bload16(rs, vo, r1 * sst);
                } else if constexpr (BPP == 2) {
                    pre[0][j] = bload8(rs, vo, r0 * sst);
                    pre[1][j] = bload8(rs, vo, r1 * sst);
                } else if constexpr (BPP == 0) {
                    pre[0][j][0] = __builtin_amdgcn_raw_buffer_load_b32(rs, vo, r0 * sst, 0); pre[0][j][1] = __builtin_amdgcn_raw_buffer_load_b32(rs1, vo, r0 * sst1, 0);
                    pre[0][j][2] = __builtin_amdgcn_raw_buffer_load_b32(rs2, vo, r0 * sst2, 0);
                    pre[1][j][0] = __builtin_amdgcn_raw_buffer_load_b32(rs, vo, r1 * sst, 0); pre[1][j][1] = __builtin_amdgcn_raw_buffer_load_b32(rs1, vo, r1 * sst1, 0);
                    pre[1][j][2] = __builtin_amdgcn_raw_buffer_load_b32(rs2, vo, r1 * sst2, 0);
                } else {
                    pre[0][j] = __builtin_bit_cast(rsrc_u32x3, __builtin_amdgcn_raw_buffer_load_b96(rs, vo, r0 * sst, 0));
                    pre[1][j] = __builtin_bit_cast(rsrc_u32x3, __builtin_amdgcn_raw_buffer_load_b96(rs, vo, r1 * sst, 0));
                }
            }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < NG; j++)
            if (j < ng) {
                const int gs = min(lane + 64 * j, n4);           // (idle lanes: the spare slot behind the window)
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    u32x2 yo; uint32_t uo, vo;
                    rgb4px_read<BPP>(pre[r][j], rk, yo, uo, vo);
                    *(u32x2 *)(LL.S + r * LL.row_dw + 2 * gs) = yo;
                    LC.S[r * LC.row_dw + gs] = uo; LC.S[(2 + r) * LC.row_dw + gs] = vo;
                }
            }
    };

    StripOut<false, CL> OL;
    StripOut<true, CC> OC;
    so_init(OL, f, p, strip * gl.TW, lane, gl.TW);
    so_init(OC, f, p, strip * gc.TW, lane, gc.TW);

    uint32_t ringL[1][CL][RL], ringC[2][CC][RC];
#pragma unroll
    for (int c = 0; c < CL; c++)
#pragma unroll
        for (int k = 0; k < RL; k++) ringL[0][c][k] = 0;
#pragma unroll
    for (int ci = 0; ci < 2; ci++)
#pragma unroll
        for (int c = 0; c < CC; c++)
#pragma unroll
            for (int k = 0; k < RC; k++) ringC[ci][c][k] = 0;

    const SwsStripRow *rowsL = gl.rows, *rowsC = gc.rows;
    const int npvL = gl.npv, npvC = gc.npv;
    StripRowN<RL> el = load_strip_row_n<RL>(rowsL, y0);
    StripRowN<RC> ec = load_strip_row_n<RC>(rowsC, min(cy0, cH - 1));
    int y = y0, cy = cy0;
    int qnext = cy < cy1 ? min(el.pf, ec.pf) : el.pf;            // next source-row pair to h-scale == the pair staged in LDS
    auto refill = [&]() {                                        // pair qnext staged, pair qnext + 1 requested
        prefetch(qnext);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        stage();
        prefetch(qnext + 1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    refill();
    while (y < y1 || cy < cy1) {
        const int big = 0x3fffffff;
        const int pfl = y < y1 ? el.pf : big, pfc = cy < cy1 ? ec.pf : big;
        const int needL = y < y1 ? pfl + npvL - 1 : big, needC = cy < cy1 ? pfc + npvC - 1 : big;
        const int need = min(needL, needC);
        if (qnext < min(pfl, pfc)) {                             // pairs neither plane class needs (steep down-scaling with short filters): skip
            qnext = min(pfl, pfc);
            refill();
        }
        while (qnext <= need) {
            uint32_t npL[1][CL], npC[2][CC];
            // (a basic block of its own: see strip_body -- straight-line code makes hipcc interleave the stage with the rings until the loop spills)
            if (gl.hfs2 < 0) {
#pragma unroll
                for (int c = 0; c < CL; c++) npL[0][c] = LL.S[spdL[c]];
#pragma unroll
                for (int ci = 0; ci < 2; ci++)
#pragma unroll
                    for (int c = 0; c < CC; c++) npC[ci][c] = LC.S[(ci * 2) * LC.row_dw + spdC[c]];
            } else {
                strip_hstage<NPH, 1, CL>(LL, spdL, htL, sh, npL);
                strip_hstage<NPH, 2, CC>(LC, spdC, htC, sh, npC);
            }
            if (rngL.on) { strip_range<1, CL>(npL, rngL); strip_range<2, CC>(npC, rngC); }
#pragma unroll
            for (int c = 0; c < CL; c++) {
#pragma unroll
                for (int k = 0; k < RL - 1; k++) ringL[0][c][k] = ringL[0][c][k + 1];
                ringL[0][c][RL - 1] = npL[0][c];
            }
#pragma unroll
            for (int ci = 0; ci < 2; ci++)
#pragma unroll
                for (int c = 0; c < CC; c++) {
#pragma unroll
                    for (int k = 0; k < RC - 1; k++) ringC[ci][c][k] = ringC[ci][c][k + 1];
                    ringC[ci][c][RC - 1] = npC[ci][c];
                }
            qnext++;
            // LDS rows are consumed: convert the prefetched pair into them, release the pending rows, request the pair after it
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            stage();
            so_flush(OL); so_flush(OC);
            prefetch(qnext + 1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // the newest ring entry is pair `need`: every row whose last pair that is
        if (needL == need) {
            so_flush(OL);
            int acc[1][CL];
            strip_vstage_n<1, CL, RL>(ringL, el, npvL, acc);
            so_put(OL, p, acc, y, lane);
            y++;
            if (y < y1) el = load_strip_row_n<RL>(rowsL, y);
        }
        if (needC == need) {
            so_flush(OC);
            int acc[2][CC];
            strip_vstage_n<2, CC, RC>(ringC, ec, npvC, acc);
            so_put(OC, p, acc, cy, lane);
            cy++;
            if (cy < cy1) ec = load_strip_row_n<RC>(rowsC, cy);
        }
    }
    so_flush(OL); so_flush(OC);
}

#ifndef RSRC_ATTR
#define RSRC_ATTR
#endif
template <int BPP, int NPH, int RL, int RC, int NG>
__global__ void __launch_bounds__(256) RSRC_ATTR sws_k_strip_rgbsrc(SwsFrameSet fs, SwsDevParams p, SwsStripGeom gl, SwsStripGeom gc, int npx_max, int wave_lds_dw)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wid = blockIdx.x * 4 + wib;
    if (wid >= gl.strips * gl.bands) return;
    const int strip = wid % gl.strips, band = wid / gl.strips;
    const int y0 = band * gl.band_rows, y1 = min(p.dstH, y0 + gl.band_rows);
    if (y0 >= y1) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    strip_rgbsrc_body<BPP, NPH, RL, RC, NG>(f, p, gl, gc, strip, y0, y1, (uint32_t *)smem + wib * wave_lds_dw, npx_max, lane);
}

} // namespace swsk
