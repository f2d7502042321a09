// The wave-march form of the same-size packed / planar 8-bit RGB -> 8-bit 4:2:0 / 4:2:2 YUV conversion (capture -> encoder); kernels_rgbsrc.hpp holds the
// first form, which stays for widths that are not multiples of 4.
#pragma once
#include "kernels_strip.hpp"

namespace swsk {

// ------------------------------------------------------------------------------------------
// Round 4: the same conversion (packed 24 / 32 bpp or planar 8-bit RGB -> 8-bit 4:2:0 / 4:2:2 YUV of the same size) as a WAVE march with the
// vertical chroma filter on v_dot2.  sws_k_rgbsrc_unity above spends 250 of its 350 instructions per wave and row on scalar bookkeeping (run-time
// ring indices, 64-bit row addresses) and does the vertical filter with one multiply per tap and value out of a lane-private LDS ring.  Here:
//  * a wave owns G groups of 256 pixels (lane l: pixels 4 l .. 4 l + 3 of each group: every load / store instruction of a row covers a contiguous
//    piece) and walks down a band TWO source rows per step -- the rows of one chroma row pair;
//  * the two rows' 15-bit chroma samples are packed as {row 2q, row 2q + 1} dwords, the operand layout of v_dot2_i32_i16, and pushed into a REGISTER
//    ring of RD pairs (RD = 1, 3, 5, 8: the host lays a row's tap pairs out against the newest slots, older slots get zero taps), so an output
//    chroma row is RD dot2 per value instead of two multiplies, two extracts and two adds per tap;
//  * rows are addressed through buffer descriptors with scalar row offsets; the plan entry of a chroma row (first pair, tap pairs) is a 64-byte
//    scalar load (SwsStripRow); the next pair's loads are issued before this pair's arithmetic.
// Arithmetic as above: rgb24ToY_c / rgb24ToUV_half_c and the 32-bit rows (input.c:264-393, :1068-1172), identity hScale16To15_c, yuv2plane1_8_c for
// luma, yuv2planeX_8_c / yuv2nv12cX_c for chroma ((64 << 12) + sum >> 19; the planar one-tap form (s + 64) >> 7 is that with the tap 4096).
// Widths that are multiples of 4 (the host keeps the kernel above for the others).
// ------------------------------------------------------------------------------------------
struct RgbSrc2Geom { int32_t band_rows, bands, strips, npv; const SwsStripRow *rows; };

// RNG: the destination is full range (yuvj*: RGB sources are limited range behind the readers, utils.c:877-880) -- lum / chrRangeToJpeg_c on the 15-bit
// samples between the (identity) horizontal scaler and the vertical stage (swscale.c:163-209), an instantiation of its own so that the capture ->
// encoder form pays nothing for it.
template <int BPP, bool NV, int RD, int G, bool RNG = false>
__global__ void __launch_bounds__(256) sws_k_rgbsrc_unity2(SwsFrameSet fs, SwsDevParams p, RgbSrc2Geom g)
{
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wid = blockIdx.x * 4 + wib;
    if (wid >= g.strips * g.bands) return;
    const int strip = wid % g.strips, band = wid / g.strips;
    const int W = p.dstW, H = p.dstH, vs = p.chrDstVSub, cH = p.chrDstH, cW = p.chrDstW;
    const int y0 = band * g.band_rows, y1 = min(H, y0 + g.band_rows);
    if (y0 >= y1) return;
    int cy = y0 >> vs;
    const int cy1 = min(cH, (y1 + (1 << vs) - 1) >> vs);
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int hshift = p.hshift, hclip = p.hclip, npv = g.npv;
    const StripRange rngL = strip_range_of(p, false), rngC = strip_range_of(p, true);
    auto rconv = [](int v, const StripRange &r) { return (int)(int16_t)min(mad24(v, r.coeff, r.offset) >> 14, r.clipmax); };

    const Rgb2YuvRow ty = rgb2yuv_row(p.rgb2yuv, 0), tu = rgb2yuv_row(p.rgb2yuv, 3), tv = rgb2yuv_row(p.rgb2yuv, 6);
    const int rp = BPP == 0 ? 0 : U(p.src_r_pos), gp = BPP == 4 ? U(p.src_g_pos) : 1, bp = BPP == 0 ? 2 : U(p.src_b_pos);
    auto coef = [&](const Rgb2YuvRow &w, int k) { return (uint32_t)(uint16_t)(k == rp ? w.r : k == gp ? w.g : k == bp ? w.b : 0); };
    const uint32_t cyA = coef(ty, 0) | coef(ty, 2) << 16, cyB = coef(ty, 1) | coef(ty, 3) << 16;
    const uint32_t cuA = coef(tu, 0) | coef(tu, 2) << 16, cuB = coef(tu, 1) | coef(tu, 3) << 16;
    const uint32_t cvA = coef(tv, 0) | coef(tv, 2) << 16, cvB = coef(tv, 1) | coef(tv, 3) << 16;

    // ---- descriptors and per-lane offsets (a group beyond the width: out-of-range offsets, nothing read or written) ----
    constexpr int NPL = BPP == 0 ? 3 : 1;
    sws_rsrc_t rs[NPL];
    int sst[NPL];
#pragma unroll
    for (int k = 0; k < NPL; k++) {
        const uint8_t *sb = k == 0 ? f.src[0] : k == 1 ? U(f.src[1]) : U(f.src[2]);
        sst[k] = k == 0 ? f.srcStride[0] : k == 1 ? U(f.srcStride[1]) : U(f.srcStride[2]);
        rs[k] = make_rsrc(sb, (uint32_t)sst[k] * (uint32_t)H);
    }
    const sws_rsrc_t rdY = make_rsrc(f.dst[0], (uint32_t)f.dstStride[0] * (uint32_t)(H - 1) + (uint32_t)W);
    const int dsY = f.dstStride[0];
    const int upl = NV ? 1 : p.u_plane_dst, vpl = NV ? 1 : p.v_plane_dst;
    uint8_t *dbu = upl == 1 ? U(f.dst[1]) : U(f.dst[2]), *dbv = vpl == 1 ? U(f.dst[1]) : U(f.dst[2]);
    const int dsU = upl == 1 ? U(f.dstStride[1]) : U(f.dstStride[2]), dsV = vpl == 1 ? U(f.dstStride[1]) : U(f.dstStride[2]);
    const sws_rsrc_t rdU = make_rsrc(dbu, (uint32_t)dsU * (uint32_t)(cH - 1) + (uint32_t)cW * (NV ? 2u : 1u));
    const sws_rsrc_t rdV = make_rsrc(dbv, (uint32_t)dsV * (uint32_t)(cH - 1) + (uint32_t)cW * (NV ? 2u : 1u));
    int voff[G], doffY[G], doffC[G];
#pragma unroll
    for (int k = 0; k < G; k++) {
        const int x = (strip * G + k) * 256 + 4 * lane;
        const bool in = x < W;
        voff[k] = in ? x * (BPP ? BPP : 1) : 0x7fffffff;
        doffY[k] = in ? x : 0x7fffffff;
        doffC[k] = in ? (NV ? x : x >> 1) : 0x7fffffff;
    }

    uint32_t ring[G][2][2][RD];      // [group][chroma column of the lane][U, V][row pair]
#pragma unroll
    for (int a = 0; a < G; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int cc = 0; cc < 2; cc++)
#pragma unroll
                for (int k = 0; k < RD; k++) ring[a][b][cc][k] = 0;

    SwsStripRow e = load_strip_row(g.rows, min(cy, cH - 1));
    const bool have_c = cy < cy1;
    const int qlo = have_c ? min(e.pf, y0 >> 1) : (y0 >> 1);
    const int qhi = have_c ? max((y1 - 1) >> 1, load_strip_row(g.rows, cy1 - 1).pf + npv - 1) : ((y1 - 1) >> 1);

    // two pairs of rows in flight (registers are cheap here: 8 waves per SIMD at under 64): the pair after next is requested before this one is touched
    constexpr int NDW = BPP == 4 ? 4 : 3;
    uint32_t nxa[G][2][NDW], nxb[G][2][NDW];
    auto fetch = [&](int q, uint32_t (&nx)[G][2][NDW]) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int row = min(max(2 * q + r, 0), H - 1);
#pragma unroll
            for (int k = 0; k < G; k++) {
                if constexpr (BPP == 0) {
                    nx[k][r][0] = __builtin_amdgcn_raw_buffer_load_b32(rs[0], voff[k], row * sst[0], 0);
                    nx[k][r][1] = __builtin_amdgcn_raw_buffer_load_b32(rs[1], voff[k], row * sst[1], 0);
                    nx[k][r][2] = __builtin_amdgcn_raw_buffer_load_b32(rs[2], voff[k], row * sst[2], 0);
                } else if constexpr (BPP == 3) {
                    typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
                    const u32x3 v = __builtin_bit_cast(u32x3, __builtin_amdgcn_raw_buffer_load_b96(rs[0], voff[k], row * sst[0], 0));
                    nx[k][r][0] = v[0]; nx[k][r][1] = v[1]; nx[k][r][2] = v[2];
                } else {
                    const u32x4 v = bload16(rs[0], voff[k], row * sst[0]);
                    nx[k][r][0] = v[0]; nx[k][r][1] = v[1]; nx[k][r][2] = v[2]; nx[k][r][3] = v[3];
                }
            }
        }
    };
    auto step = [&](int q, uint32_t (&nx)[G][2][NDW]) {
        uint32_t d[G][2][4];
#pragma unroll
        for (int k = 0; k < G; k++)
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int j = 0; j < NDW; j++) d[k][r][j] = nx[k][r][j];
        if (q + 2 <= qhi) fetch(q + 2, nx);
        uint32_t uvp[G][2][2][2];      // [group][column][U, V][row of the pair]: 15-bit samples
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int yr = 2 * q + r;
            const bool luma = yr >= y0 && yr < y1;     // (wave-uniform)
#pragma unroll
            for (int k = 0; k < G; k++) {
                uint32_t lo[4], hi[4];     // per pixel: {byte 0, byte 2} and {byte 1, byte 3 (0 for 24 bpp)} as 16-bit halves
                const uint32_t *dd = d[k][r];
                if constexpr (BPP == 0) {   // dd[0] = four G, dd[1] = four B, dd[2] = four R
                    lo[0] = __builtin_amdgcn_perm(dd[1], dd[2], 0x0c040c00u); hi[0] = __builtin_amdgcn_perm(dd[0], dd[0], 0x0c0c0c00u);
                    lo[1] = __builtin_amdgcn_perm(dd[1], dd[2], 0x0c050c01u); hi[1] = __builtin_amdgcn_perm(dd[0], dd[0], 0x0c0c0c01u);
                    lo[2] = __builtin_amdgcn_perm(dd[1], dd[2], 0x0c060c02u); hi[2] = __builtin_amdgcn_perm(dd[0], dd[0], 0x0c0c0c02u);
                    lo[3] = __builtin_amdgcn_perm(dd[1], dd[2], 0x0c070c03u); hi[3] = __builtin_amdgcn_perm(dd[0], dd[0], 0x0c0c0c03u);
                } else if constexpr (BPP == 3) {
                    lo[0] = __builtin_amdgcn_perm(dd[0], dd[0], 0x0c020c00u); hi[0] = __builtin_amdgcn_perm(dd[0], dd[0], 0x0c0c0c01u);
                    lo[1] = __builtin_amdgcn_perm(dd[1], dd[0], 0x0c050c03u); hi[1] = __builtin_amdgcn_perm(dd[1], dd[0], 0x0c0c0c04u);
                    lo[2] = __builtin_amdgcn_perm(dd[2], dd[1], 0x0c040c02u); hi[2] = __builtin_amdgcn_perm(dd[2], dd[1], 0x0c0c0c03u);
                    lo[3] = __builtin_amdgcn_perm(dd[2], dd[2], 0x0c030c01u); hi[3] = __builtin_amdgcn_perm(dd[2], dd[2], 0x0c0c0c02u);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) { lo[j] = dd[j] & 0x00FF00FFu; hi[j] = (dd[j] >> 8) & 0x00FF00FFu; }
                }
                if (luma) {
                    uint32_t out = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int S = sdot2(lo[j], cyA, sdot2_first_s(hi[j], cyB));
                        int yv;
                        if (BPP != 4) yv = (uint16_t)((S + (32 << 14) + (1 << 8)) >> 9);
                        else yv = (uint16_t)((((unsigned)S << 8) + ((32u << 22) + (1u << 16))) >> 17);
                        int y15 = (int16_t)min((yv * 16384) >> hshift, hclip);
                        if constexpr (RNG) y15 = rconv(y15, rngL);
                        out |= (uint32_t)clip_u8_shr(y15 + 64, 7) << (8 * j);
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(out, rdY, doffY[k], yr * dsY, 0);
                }
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const uint32_t al = lo[2 * j] + lo[2 * j + 1], ah = hi[2 * j] + hi[2 * j + 1];
                    const int Su = sdot2(al, cuA, sdot2_first_s(ah, cuB)), Sv = sdot2(al, cvA, sdot2_first_s(ah, cvB));
                    int ur, vr;
                    if (BPP != 4) {
                        ur = (uint16_t)((Su + (256 << 15) + (1 << 9)) >> 10);
                        vr = (uint16_t)((Sv + (256 << 15) + (1 << 9)) >> 10);
                    } else {
                        const unsigned rnd = (256u << 23) + (1u << 17);
                        ur = (uint16_t)((((unsigned)Su << 8) + rnd) >> 18);
                        vr = (uint16_t)((((unsigned)Sv << 8) + rnd) >> 18);
                    }
                    int u15 = (int16_t)min((ur * 16384) >> hshift, hclip), v15 = (int16_t)min((vr * 16384) >> hshift, hclip);
                    if constexpr (RNG) { u15 = rconv(u15, rngC); v15 = rconv(v15, rngC); }
                    uvp[k][j][0][r] = (uint32_t)(uint16_t)(int16_t)u15;
                    uvp[k][j][1][r] = (uint32_t)(uint16_t)(int16_t)v15;
                }
            }
        }
        // ---- the pair enters the ring ----
#pragma unroll
        for (int k = 0; k < G; k++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int cc = 0; cc < 2; cc++) {
#pragma unroll
                    for (int s = 0; s < RD - 1; s++) ring[k][j][cc][s] = ring[k][j][cc][s + 1];
                    ring[k][j][cc][RD - 1] = uvp[k][j][cc][0] | uvp[k][j][cc][1] << 16;
                }
        // ---- every chroma output row whose last pair this was ----
        while (cy < cy1 && e.pf + npv - 1 <= q) {
#pragma unroll
            for (int k = 0; k < G; k++) {
                uint32_t o[2][2];
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int cc = 0; cc < 2; cc++) {
                        int acc = sdot2_first_s(ring[k][j][cc][0], e.vt[0]);
#pragma unroll
                        for (int s = 1; s < RD; s++) acc = sdot2(ring[k][j][cc][s], e.vt[s], acc);
                        o[j][cc] = (uint32_t)clip_u8_shr(acc + (64 << 12), 19);
                    }
                if constexpr (NV) {
                    const int sw = p.uv_swap_dst;
                    const uint32_t p0 = sw ? (o[0][1] | o[0][0] << 8) : (o[0][0] | o[0][1] << 8), p1 = sw ? (o[1][1] | o[1][0] << 8) : (o[1][0] | o[1][1] << 8);
                    __builtin_amdgcn_raw_buffer_store_b32(p0 | p1 << 16, rdU, doffC[k], cy * dsU, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b16((uint16_t)(o[0][0] | o[1][0] << 8), rdU, doffC[k], cy * dsU, 0);
                    __builtin_amdgcn_raw_buffer_store_b16((uint16_t)(o[0][1] | o[1][1] << 8), rdV, doffC[k], cy * dsV, 0);
                }
            }
            cy++;
            if (cy < cy1) e = load_strip_row(g.rows, cy);
        }
    };
    fetch(qlo, nxa);
    if (qlo + 1 <= qhi) fetch(qlo + 1, nxb);
    for (int q = qlo; q <= qhi; q += 2) {
        step(q, nxa);
        if (q + 1 <= qhi) step(q + 1, nxb);
    }
}

} // namespace swsk
