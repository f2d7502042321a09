// Marching strip kernel with the packed-RGB epilogue: the general h+v polyphase chain from planar 8-bit YUV to 24 / 32 bpp RGB
// (the "downscale 4K to 1080p and hand it to the display / encoder as RGB" shape): hScale8To15_c for luma and chroma
// (swscale.c:128-142) + packed_vscale + yuv2rgb24_X_c / yuv2rgbx32_X_c (vscale.c:109-171, output.c:1788-1840; the LUTs of
// yuv2rgb.c:901-961 in closed form).
//
// Schedule (kernels_strip.hpp describes the planar form this extends):
//  * a wave owns a strip of 256 output columns -- 256 luma columns and the 128 chroma columns under them -- and walks down a band
//    of output rows.  Luma and chroma are two instances of the same machine (StripPlane): wave-private LDS staging of ONE PAIR of
//    source rows, horizontal taps in registers, a register ring of h-scaled row pairs, v_dot2_i32_i16 for both stages.  Nothing
//    h-scaled ever leaves the wave: the reference's 15-bit intermediate planes (2 x 4 bytes per output pixel written and read
//    back by the two-pass form) do not exist;
//  * lane l owns luma columns l, l + 64, l + 128, l + 192 and chroma columns l, l + 64 of the strip (bank-conflict-free LDS reads in
//    the horizontal stage).  The pixel pair (2j, 2j + 1) needs chroma column j: the vertical stage's luma results cross lanes once
//    per row through 512 bytes of wave-private LDS (4 ds_write_b16, 2 ds_read_b32), after which lane l holds the pairs l and
//    l + 64 and writes their 6 or 8 bytes each;
//  * the chroma side of the LUT is two 256-entry tables in LDS built per workgroup (kernels_rgbmarch.hpp), a pixel costs one
//    multiply-add per channel plus the clamping pack;
//  * per output row the scalars (first ring pair, vertical tap pairs) of both planes arrive through one 64-byte scalar load each,
//    one row ahead; the next source row pair of each plane is prefetched into registers while the current one is computed;
//  * buffer descriptors everywhere (out-of-range lanes are dropped by the hardware), stores issued after the wait for the
//    prefetched rows and before the next prefetch (vmcnt counts loads and stores together).
#pragma once
#include "kernels_strip.hpp"
#include "kernels_rgbmarch.hpp"

namespace swsk {

// S16: the source planes hold 9 .. 15-bit samples in 16-bit words (eight samples per 16-byte chunk, staged as they are); else bytes
// P01X (with S16): a p010-style source -- words with the samples in the HIGH bits, the two chroma components interleaved in plane 1
// (p010LEToY_c / p010LEToUV_c, input.c:950-1008): every word is shifted down while it is staged, a chroma row is two chunks per lane
// (8 {U, V} word pairs) de-interleaved with v_perm into one chunk per component
// the rounding constant of the packed writers for one output row: 1 << 18 (yuv2rgb_X_c_template, yuv2rgb_1_c_template: output.c:1795-1850, :1897-1960) minus what
// the host wrote into the row's entry -- 1 << 18 for the rows packed_vscale gives to yuv2rgb_2_c_template (two vertical taps each that sum to 4096: bilinear
// up-scaling; (buf0 * (4096 - a) + buf1 * a) >> 19, no rounding: output.c:1853-1895, vscale.c:146-157).  Round 5.
__device__ __forceinline__ int strip_row_rnd(const SwsStripRow *rows, int idx)
{
    typedef const uint32_t __attribute__((address_space(4))) *cptr;
    cptr q = (cptr)(uintptr_t)(rows + idx);
    return (1 << 18) - (int)q[1];
}

template <int NCOMP, int COLS, int NPH, int RD, bool S16 = false, bool P01X = false>
struct StripPlane {
    StripLds L;
    int spd[COLS];
    uint32_t ht[COLS][NPH];
    sws_rsrc_t rs[NCOMP];
    int sst[NCOMP], sH, voff, slot, qnext, sshift;
    u32x4 pre[NCOMP * 2];                 // [component][row of the pair]; P01X chroma: [row of the pair][first / second chunk]
    uint32_t ring[NCOMP][COLS][RD];
};

template <int NCOMP, int COLS, int NPH, int RD, bool S16, bool P01X>
__device__ __forceinline__ void sp_prefetch(StripPlane<NCOMP, COLS, NPH, RD, S16, P01X> &P, int q)
{
    const int r0 = min(max(2 * q, 0), P.sH - 1), r1 = min(max(2 * q + 1, 0), P.sH - 1);
    if constexpr (P01X && NCOMP == 2) {       // one interleaved plane: 32 bytes per lane and row
        P.pre[0] = bload16(P.rs[0], P.voff, r0 * P.sst[0]); P.pre[1] = bload16(P.rs[0], P.voff == 0x7fffffff ? P.voff : P.voff + 16, r0 * P.sst[0]);
        P.pre[2] = bload16(P.rs[0], P.voff, r1 * P.sst[0]); P.pre[3] = bload16(P.rs[0], P.voff == 0x7fffffff ? P.voff : P.voff + 16, r1 * P.sst[0]);
        return;
    }
#pragma unroll
    for (int ci = 0; ci < NCOMP; ci++) {
        P.pre[2 * ci + 0] = bload16(P.rs[ci], P.voff, r0 * P.sst[ci]);
        P.pre[2 * ci + 1] = bload16(P.rs[ci], P.voff, r1 * P.sst[ci]);
    }
}

__device__ __forceinline__ void sp_put8(uint32_t *dst, const u32x4 &v)   // 16 bytes -> 8 dwords of u16 sample pairs
{
    u32x4 lo, hi;
    lo[0] = __builtin_amdgcn_perm(0, v[0], 0x0c010c00u); lo[1] = __builtin_amdgcn_perm(0, v[0], 0x0c030c02u);
    lo[2] = __builtin_amdgcn_perm(0, v[1], 0x0c010c00u); lo[3] = __builtin_amdgcn_perm(0, v[1], 0x0c030c02u);
    hi[0] = __builtin_amdgcn_perm(0, v[2], 0x0c010c00u); hi[1] = __builtin_amdgcn_perm(0, v[2], 0x0c030c02u);
    hi[2] = __builtin_amdgcn_perm(0, v[3], 0x0c010c00u); hi[3] = __builtin_amdgcn_perm(0, v[3], 0x0c030c02u);
    *(u32x4 *)dst = lo; *(u32x4 *)(dst + 4) = hi;
}

template <int NCOMP, int COLS, int NPH, int RD, bool S16, bool P01X>
__device__ __forceinline__ void sp_stage(StripPlane<NCOMP, COLS, NPH, RD, S16, P01X> &P)
{
    if constexpr (P01X) {
        const uint32_t mask = (0xFFFFu >> P.sshift) * 0x10001u;
        auto down = [&](const u32x4 &v) { u32x4 w; w[0] = (v[0] >> P.sshift) & mask; w[1] = (v[1] >> P.sshift) & mask; w[2] = (v[2] >> P.sshift) & mask; w[3] = (v[3] >> P.sshift) & mask; return w; };
        if constexpr (NCOMP == 1) {
            *(u32x4 *)(P.L.S + P.slot) = down(P.pre[0]); *(u32x4 *)(P.L.S + P.L.row_dw + P.slot) = down(P.pre[1]);
        } else {
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const u32x4 a = P.pre[2 * r], b = P.pre[2 * r + 1];
                u32x4 u, v;
                u[0] = __builtin_amdgcn_perm(a[1], a[0], 0x05040100u); u[1] = __builtin_amdgcn_perm(a[3], a[2], 0x05040100u);
                u[2] = __builtin_amdgcn_perm(b[1], b[0], 0x05040100u); u[3] = __builtin_amdgcn_perm(b[3], b[2], 0x05040100u);
                v[0] = __builtin_amdgcn_perm(a[1], a[0], 0x07060302u); v[1] = __builtin_amdgcn_perm(a[3], a[2], 0x07060302u);
                v[2] = __builtin_amdgcn_perm(b[1], b[0], 0x07060302u); v[3] = __builtin_amdgcn_perm(b[3], b[2], 0x07060302u);
                *(u32x4 *)(P.L.S + (0 + r) * P.L.row_dw + P.slot) = down(u);
                *(u32x4 *)(P.L.S + (2 + r) * P.L.row_dw + P.slot) = down(v);
            }
        }
        return;
    }
#pragma unroll
    for (int ci = 0; ci < NCOMP; ci++) {
        uint32_t *row0 = P.L.S + (ci * 2) * P.L.row_dw, *row1 = row0 + P.L.row_dw;
        if constexpr (S16) { *(u32x4 *)(row0 + P.slot) = P.pre[2 * ci + 0]; *(u32x4 *)(row1 + P.slot) = P.pre[2 * ci + 1]; }
        else { sp_put8(row0 + P.slot, P.pre[2 * ci + 0]); sp_put8(row1 + P.slot, P.pre[2 * ci + 1]); }
    }
}

// per-lane column state of one plane class: window offsets, horizontal taps, source descriptors
template <int NCOMP, int COLS, int NPH, int RD, bool S16, bool P01X>
__device__ __forceinline__ void sp_init(StripPlane<NCOMP, COLS, NPH, RD, S16, P01X> &P, const SwsStripGeom &g, int strip, int W, int sH, const int32_t *hpos,
                                        const uint8_t *const (&sb)[NCOMP], const int (&sst)[NCOMP], uint32_t *lds, int lane, int sshift = 0)
{
    P.sshift = sshift;
    constexpr int SPC = S16 ? 8 : 16;      // samples per 16-byte source chunk
    const int xs = strip * g.TW, cs = g.colStart[strip], chunks = g.colCount[strip] / SPC;
    P.L.row_dw = (g.NCmax + SPC) >> 1;     // one spare chunk per row: the dump slot of idle lanes
    P.L.S = lds;
    P.sH = sH;
    const int nd = g.hfs2 >> 1;            // dwords per tap row (rows are padded to hfs2 taps: beyond that the next column's taps begin)
#pragma unroll
    for (int c = 0; c < COLS; c++) {
        const int x = min(xs + 64 * c + lane, W - 1);
        P.spd[c] = ((hpos[x] & ~1) - cs) >> 1;
        const uint32_t *tp = (const uint32_t *)(g.hT2 + (int64_t)x * g.hfs2);
#pragma unroll
        for (int k = 0; k < NPH; k++) P.ht[c][k] = k < nd ? tp[k] : 0u;
    }
#pragma unroll
    for (int ci = 0; ci < NCOMP; ci++) { P.sst[ci] = sst[ci]; P.rs[ci] = make_rsrc(sb[ci], (uint32_t)sst[ci] * (uint32_t)sH); }
    // one 16-byte chunk per lane and source row, unconditionally: a lane beyond the strip's window gets an out-of-range offset (the
    // descriptor answers 0 without touching memory) and dumps into the spare chunk
    P.voff = lane < chunks ? ((P01X && NCOMP == 2) ? cs * 4 + lane * 32 : cs * (S16 ? 2 : 1) + lane * 16) : 0x7fffffff;
    P.slot = min(lane, chunks) * (SPC / 2);
#pragma unroll
    for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
        for (int c = 0; c < COLS; c++)
#pragma unroll
            for (int k = 0; k < RD; k++) P.ring[ci][c][k] = 0;
}

// h-scale the staged row pair into the ring, then stage the prefetched pair and request the one after it.  `flush` releases the
// pending output row between the wait for the prefetched pair and the next request (see the header of kernels_strip.hpp).
template <int NCOMP, int COLS, int NPH, int RD, bool S16, bool P01X, typename F>
__device__ __forceinline__ void sp_step(StripPlane<NCOMP, COLS, NPH, RD, S16, P01X> &P, int sh, int opaque_neg, F &&flush)
{
    uint32_t np[NCOMP][COLS];
    // (a basic block of its own: see strip_body in kernels_strip.hpp -- straight-line code makes hipcc spill inside the loop)
    if (opaque_neg < 0) {
#pragma unroll
        for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
            for (int c = 0; c < COLS; c++) np[ci][c] = P.L.S[(ci * 2) * P.L.row_dw + P.spd[c]];
    } else
    strip_hstage<NPH, NCOMP, COLS>(P.L, P.spd, P.ht, sh, np);
#pragma unroll
    for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
        for (int c = 0; c < COLS; c++) {
#pragma unroll
            for (int k = 0; k < RD - 1; k++) P.ring[ci][c][k] = P.ring[ci][c][k + 1];
            P.ring[ci][c][RD - 1] = np[ci][c];
        }
    P.qnext++;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    sp_stage(P);
    flush();
    sp_prefetch(P, P.qnext + 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int NCOMP, int COLS, int NPH, int RD, bool S16, bool P01X>
__device__ __forceinline__ void sp_restart(StripPlane<NCOMP, COLS, NPH, RD, S16, P01X> &P, int q)   // (re)fill: pair q staged, pair q + 1 requested
{
    P.qnext = q;
    sp_prefetch(P, q);
    sp_stage(P);
    sp_prefetch(P, q + 1);
}

// vertical stage over the whole ring: the host lays the row's tap pairs out against the ring slots (slots older than the row's
// window carry zero taps), so there is nothing to decide here.  Sums of sample * tap, no rounding constant.
template <int NCOMP, int COLS, int NPH, int RD, bool S16, bool P01X>
__device__ __forceinline__ void sp_vstage(const StripPlane<NCOMP, COLS, NPH, RD, S16, P01X> &P, const SwsStripRow &e, int (&acc)[NCOMP][COLS])
{
#pragma unroll
    for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
        for (int c = 0; c < COLS; c++) {
            acc[ci][c] = sdot2_first_s(P.ring[ci][c][0], e.vt[0]);
#pragma unroll
            for (int k = 1; k < RD; k++) acc[ci][c] = sdot2(P.ring[ci][c][k], e.vt[k], acc[ci][c]);
        }
}

// one pixel pair through the LUT: Y1, Y2, U, V as the reference's ">> 19" results -> BPP == 3: w[0] = bytes 0..3, w[1] = bytes 4..5;
// BPP == 4: w[0], w[1] = the two pixels
template <int BPP>
__device__ __forceinline__ void lut_pair(const SwsLutParams &L, const LutTabs &T, bool swap_rb, int Y1, int Y2, int U, int V, uint32_t (&w)[2])
{
    const uint32_t ou = (uint32_t)clip_u8(U) << 3, ov = (uint32_t)clip_u8(V) << 3;
    const u32x2 ev = *(const u32x2 *)(T.v + ov), eu = *(const u32x2 *)(T.u + ou);
    const int A0 = (int)(swap_rb ? eu[0] : ev[0]), A1 = (int)(ev[1] + eu[1]), A2 = (int)(swap_rb ? ev[0] : eu[0]);
    const int a0 = mad24(Y1, L.cy, A0), a1 = mad24(Y1, L.cy, A1), a2 = mad24(Y1, L.cy, A2);
    const int b0 = mad24(Y2, L.cy, A0), b1 = mad24(Y2, L.cy, A1), b2 = mad24(Y2, L.cy, A2);
    if constexpr (BPP == 4) {
        const int ta = 255 << 16;
        // canonical bytes {first, g, third, alpha}; L.perm32 moves them to the format's order (alpha-first formats)
        w[0] = __builtin_amdgcn_perm(0, pack4_u8_shr16(a0, a1, a2, ta), L.perm32);
        w[1] = __builtin_amdgcn_perm(0, pack4_u8_shr16(b0, b1, b2, ta), L.perm32);
    } else {
        w[0] = pack4_u8_shr16(a0, a1, a2, b0);
        w[1] = pack4_u8_shr16(b1, b2, 0, 0);
    }
}

template <int BPP, int NPH, int RL, int RC, int CL, bool S16, bool P01X>
__device__ __forceinline__ void strip_rgb_body(const FrameRegs &f, const SwsDevParams &p, const SwsStripGeom &gl, const SwsStripGeom &gc,
                                               int strip, int y0, int y1, uint32_t *lds, const LutTabs &T, int lane)
{
    const int W = p.dstW, H = p.dstH;
    constexpr int CC = CL / 2;             // lane l: luma columns l + 64 c (c < CL), chroma columns and pixel pairs l + 64 c (c < CC)
    StripPlane<1, CL, NPH, RL, S16, P01X> PL;
    StripPlane<2, CC, NPH, RC, S16, P01X> PC;
    constexpr int SPC = S16 ? 8 : 16;
    const int ldw = (gl.NCmax + SPC) >> 1, cdw = (gc.NCmax + SPC) >> 1;
    uint32_t *ldsL = lds, *ldsC = lds + 2 * ldw, *ldsX = ldsC + 4 * cdw;       // luma rows, chroma rows, 256 x int16 exchange row
    {
        const uint8_t *const sb[1] = { f.src[0] };
        const int st[1] = { f.srcStride[0] };
        sp_init(PL, gl, strip, W, p.srcH, p.hLumPos, sb, st, ldsL, lane, p.src_shift);
    }
    if constexpr (P01X) {     // (both components out of plane 1; the second descriptor is not used)
        const uint8_t *const sb[2] = { f.src[1], f.src[1] };
        const int st[2] = { f.srcStride[1], f.srcStride[1] };
        sp_init(PC, gc, strip, p.chrDstW, p.chrSrcH, p.hChrPos, sb, st, ldsC, lane, p.src_shift);
    } else {
        const bool u1 = p.u_plane_src == 1;
        const uint8_t *const sb[2] = { u1 ? f.src[1] : f.src[2], u1 ? f.src[2] : f.src[1] };
        const int st[2] = { u1 ? f.srcStride[1] : f.srcStride[2], u1 ? f.srcStride[2] : f.srcStride[1] };
        sp_init(PC, gc, strip, p.chrDstW, p.chrSrcH, p.hChrPos, sb, st, ldsC, lane);
    }
    // destination: whole picture, pair j of the strip at byte (256 * strip + 2 * j) * BPP of a row
    const sws_rsrc_t rd = make_rsrc(f.dst[0], (uint32_t)f.dstStride[0] * (uint32_t)(H - 1) + (uint32_t)W * (uint32_t)BPP);
    const int dstr = f.dstStride[0];
    int doff[CC];
#pragma unroll
    for (int c = 0; c < CC; c++) {
        const int x = strip * (64 * CL) + 2 * (64 * c + lane);
        doff[c] = x < W ? x * BPP : 0x7fffffff;
    }
    uint32_t pend[CC][2];
    int pend_y = -1;
    auto flush = [&]() {
        if (pend_y >= 0) {
            const int ro = pend_y * dstr;
#pragma unroll
            for (int c = 0; c < CC; c++) {
                if constexpr (BPP == 4) {
                    u32x2 v = { pend[c][0], pend[c][1] };
                    __builtin_amdgcn_raw_buffer_store_b64(v, rd, doff[c], ro, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(pend[c][0], rd, doff[c], ro, 0);
                    __builtin_amdgcn_raw_buffer_store_b16((uint16_t)pend[c][1], rd, doff[c] + 4, ro, 0);
                }
            }
            pend_y = -1;
        }
    };

    const SwsStripRow *rowsL = gl.rows, *rowsC = gc.rows;
    const int npvL = gl.npv, npvC = gc.npv, sh = p.hshift;
    const bool swap_rb = BPP == 4 ? p.lut.swap_rb32 != 0 : p.lut.rgb_order != 0;
    SwsStripRow el = load_strip_row(rowsL, y0), ec = load_strip_row(rowsC, y0);
    int rnd = strip_row_rnd(rowsL, y0);
    sp_restart(PL, el.pf);
    sp_restart(PC, ec.pf);
    for (int y = y0; y < y1; y++) {
        const int yn = min(y + 1, H - 1);
        const SwsStripRow eln = load_strip_row(rowsL, yn), ecn = load_strip_row(rowsC, yn);     // next row's scalars, one row ahead
        const int rndn = strip_row_rnd(rowsL, yn);
        if (PL.qnext < el.pf) sp_restart(PL, el.pf);            // pairs nobody needs (steep down-scaling with short filters)
        if (PC.qnext < ec.pf) sp_restart(PC, ec.pf);
        while (PL.qnext <= el.pf + npvL - 1) sp_step(PL, sh, gl.hfs2, flush);
        while (PC.qnext <= ec.pf + npvC - 1) sp_step(PC, sh, gc.hfs2, flush);
        flush();                                                // (a row that needed no new pair still has to release the previous one)
        int aL[1][CL], aC[2][CC];
        sp_vstage(PL, el, aL);
        sp_vstage(PC, ec, aC);
        // luma across lanes: column 64 c + l (lane l) -> pairs (2j, 2j + 1) for j = l, l + 64
        int16_t *X = (int16_t *)ldsX;
#pragma unroll
        for (int c = 0; c < CL; c++) X[64 * c + lane] = (int16_t)((aL[0][c] + rnd) >> 19);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int c = 0; c < CC; c++) {
            const uint32_t yy = ((const uint32_t *)ldsX)[64 * c + lane];
            const int Y1 = (int)(int16_t)(yy & 0xFFFFu), Y2 = (int)yy >> 16;
            const int U = (aC[0][c] + rnd) >> 19, V = (aC[1][c] + rnd) >> 19;
            lut_pair<BPP>(p.lut, T, swap_rb, Y1, Y2, U, V, pend[c]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the exchange row is rewritten by the next output row
        __builtin_amdgcn_wave_barrier();
        pend_y = y;
        el = eln; ec = ecn; rnd = rndn;
    }
    flush();
}

// ------------------------------------------------------------------------------------------
// LDS-DMA form for 8-bit planar sources (round 4; kernels_strip8.hpp describes the planar twin): the raw bytes of a source row's window go
// HBM -> LDS with one `buffer_load_dwordx4 ... lds` per row into a ring of D = 4 row pairs per plane class, the horizontal stage unpacks byte
// pairs with v_perm_b32 while it reads (windows start at any byte: tap rows without alignment padding -- bicubic at 2:1 is 4 tap pairs, not 5).
// Gone: the staging registers (24 VGPRs), 8 v_perm + 2 ds_write_b128 per staged row, and the single row pair in flight per plane.
// Both plane classes share the one in-order vmcnt counter.  Before plane X reads its pair q it waits for `vmcnt((D - 1) * P_X)` (P_X = DMA
// instructions per pair of X): when q is due, the D - 1 later pairs of X have been requested after it, whatever the other plane requested in
// between, so "at most (D - 1) * P_X operations outstanding" implies q has landed (requests return in order; stores only make the wait longer).
// ------------------------------------------------------------------------------------------
#ifndef SWS_RGB8_DEPTH
#define SWS_RGB8_DEPTH 2   // row pairs per plane class in the LDS ring (a power of two)
#endif
// NV: the two chroma components are ONE plane of interleaved byte pairs (nv12 / nv21 / nv16 / nv24: nvXXtoUV_c, input.c:926-948), DMA'd as it is
// (one request per row); the selectors pick a component's bytes (kernels_strip8.hpp strip_body_dma8<..., NV>)
template <int NCOMP, int COLS, int NPH, int RD, bool NV = false>
struct StripPlane8 {
    static constexpr int NSRC = NV ? 1 : NCOMP;
    uint32_t *ringS; uint32_t lds_base;
    int row_dw, pair_dw;
    int spd[COLS];
    uint32_t sel0[COLS], sel1[COLS];
    uint32_t ht[COLS][NPH];
    i32x4s rs[NSRC];
    int sst[NSRC], sH, voff, qnext, qdma;
    uint32_t mlo, mhi;
    uint32_t ring[NCOMP][COLS][RD];
};

template <int NCOMP, int COLS, int NPH, int RD, bool NV>
__device__ __forceinline__ void sp8_dma(StripPlane8<NCOMP, COLS, NPH, RD, NV> &P)     // request the next row pair into its ring slot
{
    constexpr int NSRC = NV ? 1 : NCOMP;
    constexpr int D = SWS_RGB8_DEPTH;
    const int q = P.qdma++;
    const int r0 = min(max(2 * q, 0), P.sH - 1), r1 = min(max(2 * q + 1, 0), P.sH - 1);
    const uint32_t slot = P.lds_base + (uint32_t)((q & (D - 1)) * P.pair_dw) * 4u;
#pragma unroll
    for (int ci = 0; ci < NSRC; ci++)
#pragma unroll
        for (int r = 0; r < 2; r++)
            strip_dma16(slot + (uint32_t)((ci * 2 + r) * P.row_dw) * 4u, P.voff, P.rs[ci], (r ? r1 : r0) * P.sst[ci], P.mlo, P.mhi);
}

template <int NCOMP, int COLS, int NPH, int RD, bool NV, int NS>
__device__ __forceinline__ void sp8_init(StripPlane8<NCOMP, COLS, NPH, RD, NV> &P, const SwsStripGeom &g, int strip, int W, int sH, const int32_t *hpos,
                                         const uint8_t *const (&sb)[NS], const int (&sst)[NS], uint32_t *lds, int lane, int nv_swap = 0)
{
    constexpr int NSRC = NV ? 1 : NCOMP;
    static_assert(NS == NSRC, "one base pointer per source plane");
    const int xs = strip * g.TW, cs = g.colStart[strip], chunks = (NV ? 2 : 1) * g.colCount[strip] / 16;
    P.row_dw = ((NV ? 2 : 1) * g.NCmax + 16) >> 2;        // bytes; one spare chunk: the last column's aligned reads
    P.pair_dw = NSRC * 2 * P.row_dw;
    P.ringS = lds;
    P.lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)lds);
    P.sH = sH;
    const int nd = g.nph8;                 // dwords per tap row of this plane (the kernel's NPH is the larger of the two planes': zero-extended)
#pragma unroll
    for (int c = 0; c < COLS; c++) {
        const int x = min(xs + 64 * c + lane, W - 1);
        const int o = (NV ? 2 : 1) * (hpos[x] - cs);
        P.spd[c] = o >> 2;
        const uint32_t b = (uint32_t)(o & 3);
        if constexpr (NV) {                    // sel0 / sel1: the first / second component's bytes of two interleaved pairs
            const uint32_t b0 = b + (nv_swap ? 1u : 0u), b1 = b + (nv_swap ? 0u : 1u);
            P.sel0[c] = 0x0c000c00u | b0 | ((b0 + 2) << 16);
            P.sel1[c] = 0x0c000c00u | b1 | ((b1 + 2) << 16);
        } else {
            P.sel0[c] = 0x0c000c00u | b | ((b + 1) << 16);
            P.sel1[c] = 0x0c000c00u | (b + 2) | ((b + 3) << 16);
        }
        const uint32_t *tp = (const uint32_t *)(g.hT8 + (int64_t)x * (2 * nd));
#pragma unroll
        for (int k = 0; k < NPH; k++) P.ht[c][k] = k < nd ? tp[k] : 0u;
    }
#pragma unroll
    for (int ci = 0; ci < NSRC; ci++) {
        P.sst[ci] = sst[ci];
        const uint64_t a = uniform_u64((uint64_t)sb[ci]);
        P.rs[ci][0] = (int)(uint32_t)a; P.rs[ci][1] = (int)(uint32_t)(a >> 32);
        P.rs[ci][2] = __builtin_amdgcn_readfirstlane((int)((uint32_t)sst[ci] * (uint32_t)sH)); P.rs[ci][3] = 0x00020000;
    }
    P.voff = (NV ? 2 : 1) * cs + lane * 16;
    const int n0 = min(chunks, 64);
    P.mlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n0 >= 32 ? 0xffffffffu : ((1u << n0) - 1u)));
    P.mhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n0 >= 64 ? 0xffffffffu : (n0 > 32 ? ((1u << (n0 - 32)) - 1u) : 0u)));
#pragma unroll
    for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
        for (int c = 0; c < COLS; c++)
#pragma unroll
            for (int k = 0; k < RD; k++) P.ring[ci][c][k] = 0;
}

// h-scale the oldest requested pair into the ring, release the pending output row, re-request the slot
template <int NCOMP, int COLS, int NPH, int RD, bool NV, typename F>
__device__ __forceinline__ void sp8_step(StripPlane8<NCOMP, COLS, NPH, RD, NV> &P, int sh, int opaque_neg, F &&flush)
{
    constexpr int D = SWS_RGB8_DEPTH;
    constexpr int NSRC = NV ? 1 : NCOMP;
    constexpr int NDW = NV ? NPH + 1 : (NPH - 1) / 2 + 2;
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 1) * NSRC * 2) : "memory");     // (D - 1) * P
    const uint32_t *S = P.ringS + (P.qnext & (D - 1)) * P.pair_dw;
    uint32_t np[NCOMP][COLS];
    if (opaque_neg < 0) {      // (never: a basic block of its own for the horizontal stage, see strip_body in kernels_strip.hpp)
#pragma unroll
        for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
            for (int c = 0; c < COLS; c++) np[ci][c] = S[(NV ? 0 : ci * 2) * P.row_dw + P.spd[c]];
    } else if constexpr (NV) {
#pragma unroll
        for (int c = 0; c < COLS; c++) {
            const uint32_t *s0 = S + P.spd[c], *s1 = s0 + P.row_dw;
            uint32_t d0[NDW], d1[NDW];
#pragma unroll
            for (int j = 0; j < NDW; j++) { d0[j] = s0[j]; d1[j] = s1[j]; }
#pragma unroll
            for (int ci = 0; ci < 2; ci++) {
                const uint32_t sl = ci ? P.sel1[c] : P.sel0[c];
                int a = sdot2_first(__builtin_amdgcn_perm(d0[1], d0[0], sl), P.ht[c][0]);
                int b = sdot2_first(__builtin_amdgcn_perm(d1[1], d1[0], sl), P.ht[c][0]);
#pragma unroll
                for (int k = 1; k < NPH; k++) {
                    a = sdot2(__builtin_amdgcn_perm(d0[k + 1], d0[k], sl), P.ht[c][k], a);
                    b = sdot2(__builtin_amdgcn_perm(d1[k + 1], d1[k], sl), P.ht[c][k], b);
                }
                np[ci][c] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(a >> sh, b >> sh));
            }
        }
    } else {
#pragma unroll
        for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
            for (int c = 0; c < COLS; c++) {
                const uint32_t *s0 = S + (ci * 2) * P.row_dw + P.spd[c], *s1 = s0 + P.row_dw;
                uint32_t d0[NDW], d1[NDW];
#pragma unroll
                for (int j = 0; j < NDW; j++) { d0[j] = s0[j]; d1[j] = s1[j]; }
                int a = sdot2_first(__builtin_amdgcn_perm(d0[1], d0[0], P.sel0[c]), P.ht[c][0]);
                int b = sdot2_first(__builtin_amdgcn_perm(d1[1], d1[0], P.sel0[c]), P.ht[c][0]);
#pragma unroll
                for (int k = 1; k < NPH; k++) {
                    const uint32_t sl = (k & 1) ? P.sel1[c] : P.sel0[c];
                    a = sdot2(__builtin_amdgcn_perm(d0[(k >> 1) + 1], d0[k >> 1], sl), P.ht[c][k], a);
                    b = sdot2(__builtin_amdgcn_perm(d1[(k >> 1) + 1], d1[k >> 1], sl), P.ht[c][k], b);
                }
                np[ci][c] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(a >> sh, b >> sh));
            }
    }
#pragma unroll
    for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
        for (int c = 0; c < COLS; c++) {
#pragma unroll
            for (int k = 0; k < RD - 1; k++) P.ring[ci][c][k] = P.ring[ci][c][k + 1];
            P.ring[ci][c][RD - 1] = np[ci][c];
        }
    P.qnext++;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the slot's reads have returned: it may be requested again
    __builtin_amdgcn_wave_barrier();
    flush();
    sp8_dma(P);
}

template <int NCOMP, int COLS, int NPH, int RD, bool NV>
__device__ __forceinline__ void sp8_vstage(const StripPlane8<NCOMP, COLS, NPH, RD, NV> &P, const SwsStripRow &e, int (&acc)[NCOMP][COLS])
{
#pragma unroll
    for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
        for (int c = 0; c < COLS; c++) {
            acc[ci][c] = sdot2_first_s(P.ring[ci][c][0], e.vt[0]);
#pragma unroll
            for (int k = 1; k < RD; k++) acc[ci][c] = sdot2(P.ring[ci][c][k], e.vt[k], acc[ci][c]);
        }
}

template <int BPP, int NPH, int RL, int RC, int CL, bool NV>
__device__ __forceinline__ void strip_rgb8_body(const FrameRegs &f, const SwsDevParams &p, const SwsStripGeom &gl, const SwsStripGeom &gc,
                                                int strip, int y0, int y1, uint32_t *lds, const LutTabs &T, int lane)
{
    constexpr int D = SWS_RGB8_DEPTH;
    const int W = p.dstW, H = p.dstH;
    constexpr int CC = CL / 2;
    StripPlane8<1, CL, NPH, RL> PL;
    StripPlane8<2, CC, NPH, RC, NV> PC;
    const int ldw = (gl.NCmax + 16) >> 2, cdw = ((NV ? 2 : 1) * gc.NCmax + 16) >> 2;
    uint32_t *ldsL = lds, *ldsC = lds + D * 2 * ldw, *ldsX = ldsC + D * (NV ? 2 : 4) * cdw;       // luma ring, chroma ring, 64 * CL x int16 exchange row
    {
        const uint8_t *const sb[1] = { f.src[0] };
        const int st[1] = { f.srcStride[0] };
        sp8_init(PL, gl, strip, W, p.srcH, p.hLumPos, sb, st, ldsL, lane);
    }
    if constexpr (NV) {     // (the planner keeps the interleaved plane in src[1]; nv_swap: V first -- nv21 / nv42)
        const uint8_t *const sb[1] = { f.src[1] };
        const int st[1] = { f.srcStride[1] };
        sp8_init(PC, gc, strip, p.chrDstW, p.chrSrcH, p.hChrPos, sb, st, ldsC, lane, p.uv_swap_src);
    } else {
        const bool u1 = p.u_plane_src == 1;
        const uint8_t *const sb[2] = { u1 ? f.src[1] : f.src[2], u1 ? f.src[2] : f.src[1] };
        const int st[2] = { u1 ? f.srcStride[1] : f.srcStride[2], u1 ? f.srcStride[2] : f.srcStride[1] };
        sp8_init(PC, gc, strip, p.chrDstW, p.chrSrcH, p.hChrPos, sb, st, ldsC, lane);
    }
    const sws_rsrc_t rd = make_rsrc(f.dst[0], (uint32_t)f.dstStride[0] * (uint32_t)(H - 1) + (uint32_t)W * (uint32_t)BPP);
    const int dstr = f.dstStride[0];
    int doff[CC];
#pragma unroll
    for (int c = 0; c < CC; c++) {
        const int x = strip * (64 * CL) + 2 * (64 * c + lane);
        doff[c] = x < W ? x * BPP : 0x7fffffff;
    }
    uint32_t pend[CC][2];
    int pend_y = -1;
    auto flush = [&]() {
        if (pend_y >= 0) {
            const int ro = pend_y * dstr;
#pragma unroll
            for (int c = 0; c < CC; c++) {
                if constexpr (BPP == 4) {
                    u32x2 v = { pend[c][0], pend[c][1] };
                    __builtin_amdgcn_raw_buffer_store_b64(v, rd, doff[c], ro, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(pend[c][0], rd, doff[c], ro, 0);
                    __builtin_amdgcn_raw_buffer_store_b16((uint16_t)pend[c][1], rd, doff[c] + 4, ro, 0);
                }
            }
            pend_y = -1;
        }
    };

    const SwsStripRow *rowsL = gl.rows, *rowsC = gc.rows;
    const int npvL = gl.npv, npvC = gc.npv, sh = p.hshift;
    const bool swap_rb = BPP == 4 ? p.lut.swap_rb32 != 0 : p.lut.rgb_order != 0;
    SwsStripRow el = load_strip_row(rowsL, y0), ec = load_strip_row(rowsC, y0);
    int rnd = strip_row_rnd(rowsL, y0);
    PL.qnext = PL.qdma = el.pf; PC.qnext = PC.qdma = ec.pf;
    // (the column state came through vector loads the compiler counts, the requests below are asm statements it does not see: everything loaded so far
    //  is consumed here, while the compiler's own vmcnt arithmetic is still exact -- kernels_strip8.hpp)
#pragma unroll
    for (int c = 0; c < CL; c++) { asm volatile("" : "+v"(PL.spd[c]), "+v"(PL.sel0[c]), "+v"(PL.sel1[c]));
#pragma unroll
        for (int k = 0; k < NPH; k++) asm volatile("" : "+v"(PL.ht[c][k])); }
#pragma unroll
    for (int c = 0; c < CC; c++) { asm volatile("" : "+v"(PC.spd[c]), "+v"(PC.sel0[c]), "+v"(PC.sel1[c]));
#pragma unroll
        for (int k = 0; k < NPH; k++) asm volatile("" : "+v"(PC.ht[c][k])); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < D; i++) sp8_dma(PL);
#pragma unroll
    for (int i = 0; i < D; i++) sp8_dma(PC);
    for (int y = y0; y < y1; y++) {
        const int yn = min(y + 1, H - 1);
        const SwsStripRow eln = load_strip_row(rowsL, yn), ecn = load_strip_row(rowsC, yn);
        const int rndn = strip_row_rnd(rowsL, yn);
        while (PL.qnext <= el.pf + npvL - 1) sp8_step(PL, sh, gl.hfs2, flush);
        while (PC.qnext <= ec.pf + npvC - 1) sp8_step(PC, sh, gc.hfs2, flush);
        flush();
        int aL[1][CL], aC[2][CC];
        sp8_vstage(PL, el, aL);
        sp8_vstage(PC, ec, aC);
        int16_t *X = (int16_t *)ldsX;
#pragma unroll
        for (int c = 0; c < CL; c++) X[64 * c + lane] = (int16_t)((aL[0][c] + rnd) >> 19);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int c = 0; c < CC; c++) {
            const uint32_t yy = ((const uint32_t *)ldsX)[64 * c + lane];
            const int Y1 = (int)(int16_t)(yy & 0xFFFFu), Y2 = (int)yy >> 16;
            const int U = (aC[0][c] + rnd) >> 19, V = (aC[1][c] + rnd) >> 19;
            lut_pair<BPP>(p.lut, T, swap_rb, Y1, Y2, U, V, pend[c]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the exchange row is rewritten by the next output row
        __builtin_amdgcn_wave_barrier();
        pend_y = y;
        el = eln; ec = ecn; rnd = rndn;
    }
    flush();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no DMA write may land after the wave has given up its LDS
}

template <int BPP, int RL, int RC, int NPH, int CL, bool NV = false>
__global__ void __launch_bounds__(256) sws_k_strip_rgb8(SwsFrameSet fs, SwsDevParams p, SwsStripGeom gl, SwsStripGeom gc, int wave_lds_dw)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ __attribute__((aligned(16))) u32x2 lds_tab[2][256];
    build_lut_tabs(p.lut, lds_tab[0], lds_tab[1], (int)threadIdx.x);
    __syncthreads();
    const LutTabs T = { (const uint8_t *)lds_tab[0], (const uint8_t *)lds_tab[1] };
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wid = blockIdx.x * 4 + wib;
    if (wid >= gl.strips * gl.bands) return;
    const int strip = wid % gl.strips, band = wid / gl.strips;
    const int y0 = band * gl.band_rows, y1 = min(p.dstH, y0 + gl.band_rows);
    if (y0 >= y1) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    uint32_t *lds = (uint32_t *)smem + wib * wave_lds_dw;
    strip_rgb8_body<BPP, NPH, RL, RC, CL, NV>(f, p, gl, gc, strip, y0, y1, lds, T, lane);
}

// RL / RC: ring depths (row pairs) of the luma / chroma plane = the vertical tap pairs the kernel multiplies per output sample
#ifndef SWS_SRGB_ATTR
#define SWS_SRGB_ATTR
#endif
template <int BPP, int RL, int RC, int NPH, int CL, bool S16 = false, bool P01X = false>      // one kernel per ring form and horizontal tap-pair count: each gets the register allocation it needs
__global__ void __launch_bounds__(256) SWS_SRGB_ATTR sws_k_strip_rgb(SwsFrameSet fs, SwsDevParams p, SwsStripGeom gl, SwsStripGeom gc, int wave_lds_dw)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ __attribute__((aligned(16))) u32x2 lds_tab[2][256];
    build_lut_tabs(p.lut, lds_tab[0], lds_tab[1], (int)threadIdx.x);
    __syncthreads();
    const LutTabs T = { (const uint8_t *)lds_tab[0], (const uint8_t *)lds_tab[1] };
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wid = blockIdx.x * 4 + wib;
    if (wid >= gl.strips * gl.bands) return;
    const int strip = wid % gl.strips, band = wid / gl.strips;
    const int y0 = band * gl.band_rows, y1 = min(p.dstH, y0 + gl.band_rows);
    if (y0 >= y1) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    uint32_t *lds = (uint32_t *)smem + wib * wave_lds_dw;
    strip_rgb_body<BPP, NPH, RL, RC, CL, S16, P01X>(f, p, gl, gc, strip, y0, y1, lds, T, lane);
}

} // namespace swsk
