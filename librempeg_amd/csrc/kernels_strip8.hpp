// LDS-DMA form of the marching strip kernel for 8-bit planar sources with short filters (C1; the upper rungs of an ABR ladder).
// Same arithmetic as strip_body (kernels_strip.hpp: hScale8To15_c, swscale.c:127-142; yuv2planeX_8_c / yuv2plane1_8_c and the N-bit / semi-planar
// writers, output.c:327-357, :468-528), same lane / column mapping, same register ring, same 64-byte plan entries.  What differs is how the
// source rows reach the horizontal stage:
//
//  * `buffer_load_dwordx4 ... lds` writes the RAW BYTES of a row's window straight into the wave's LDS ring (one instruction per row: windows
//    of at most 64 chunks of 16 bytes).  The register-staged form spends 8 v_perm + 2 ds_write_b128 per row on the byte -> u16 expansion and
//    keeps one row pair in flight per wave (the staging registers); here D - 1 = 3 pairs are in flight and the staging instructions are gone;
//  * the horizontal stage unpacks while it reads: a column whose window starts at byte o reads the aligned dwords from (o & ~3) on and builds
//    each {sample 2k, sample 2k + 1} operand of v_dot2_i32_i16 with ONE v_perm_b32 (selector bytes o & 3 .. , zero-extending).  Pair k lies
//    in dwords (k >> 1, (k >> 1) + 1) with one of TWO per-column selectors (k even / odd), whatever o is -- so a window may start at an odd
//    sample and the tap rows need no alignment padding: bilinear at 2:1 (4 taps from an odd position) is 2 tap pairs instead of 3;
//  * LDS traffic per column and row: 8 bytes (NPH <= 2) or 12 (<= 4) instead of 12 / 16 / 20 of expanded samples, and no LDS writes by the wave.
//
// The waits are written by hand as in strip_body_dma: `s_waitcnt vmcnt((D - 1) * P)` before pair q is read (P = DMA instructions per pair;
// loads return in order among themselves), `lgkmcnt(0)` before a slot is re-requested, `vmcnt(0)` before the wave ends.
// The host selects the form for planar 8-bit sources (not semi-planar: their chroma bytes are interleaved) whose plan skips no source row pair
// inside a band (dev_plan*.hip plan3_alt: dma8_ok), on 16-byte aligned frames.
#pragma once
#include "kernels_strip.hpp"

namespace swsk {

#ifndef STRIP_DMA8_DEPTH_C
#define STRIP_DMA8_DEPTH_C 2
#endif
constexpr int strip_dma8_depth_c = STRIP_DMA8_DEPTH_C;     // (dev_plan*.hip sizes the chroma launch's LDS with it)

// NV: the chroma planes of a semi-planar source (nv12 / nv21 / nv16 / nv24: nvXXtoUV_c, input.c:926-948) -- ONE plane of interleaved {U, V} byte pairs.
// Its rows land in LDS as they are (one request per row instead of two planes' worth); sample j of component ci is byte 2 j + ci, so pair k of a
// window starting at sample o lies in the aligned dwords (k, k + 1) from byte (2 o & ~3) on, with ONE selector per component whatever k is
// ({r + ci, 0, r + ci + 2, 0}, r = 2 o & 3): both components are unpacked from the same NPH + 1 dwords.
template <bool CHROMA, int COLS, int NPH, int RD, bool NV = false>
__device__ __forceinline__ void strip_body_dma8(const FrameRegs &f, const SwsDevParams &p, const SwsStripGeom &g, int strip, int y0, int y1,
                                                uint8_t *smem, int wib, int lane)
{
    static_assert(CHROMA || !NV, "only the chroma planes are interleaved");
    // ring depth in row pairs: 4 for luma; 2 for the chroma planes, whose rings (two components, strips of up to 320 columns) would otherwise hold the
    // kernel at 3 waves per SIMD through their LDS footprint (43 KB per block on C1).  Measured level with depth 4 (C1 x256 0.439 vs 0.436 - 0.446 of the
    // HBM peak, x64 0.31 vs 0.30): the waves gained and the pairs in flight lost cancel; the smaller footprint is kept
    constexpr int NCOMP = CHROMA ? 2 : 1, D = CHROMA ? STRIP_DMA8_DEPTH_C : STRIP_DMA_DEPTH;
    constexpr int NSRC = NV ? 1 : NCOMP;                      // source planes = rows requested per row of a pair
    constexpr int NDW = NV ? NPH + 1 : (NPH - 1) / 2 + 2;     // aligned dwords a column's window can touch
    const int W = CHROMA ? p.chrDstW : p.dstW, H = CHROMA ? p.chrDstH : p.dstH;
    const int sH = CHROMA ? p.chrSrcH : p.srcH;
    const int xs = strip * g.TW;
    const int cs = g.colStart[strip], chunks = (NV ? 2 : 1) * g.colCount[strip] / 16;
    const int32_t *hpos = CHROMA ? p.hChrPos : p.hLumPos;
    const int npv = g.npv, sh = p.hshift;
    const StripRange rng = strip_range_of(p, CHROMA);
    const int row_dw = ((NV ? 2 : 1) * g.NCmax + 16) >> 2;    // dwords of a staged row of bytes (one spare chunk: the last column's aligned reads)
    const int pair_dw = NSRC * 2 * row_dw;
    uint32_t *ringS = (uint32_t *)smem + wib * (D * pair_dw);
    const uint32_t lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)ringS);

    // ---- per-lane column state: window dword, the two byte selectors, horizontal taps (registers for the whole band) ----
    int spd[COLS];
    uint32_t sel0[COLS], sel1[COLS];
    uint32_t ht[COLS][NPH];
#pragma unroll
    for (int c = 0; c < COLS; c++) {
        const int x = min(xs + 64 * c + lane, W - 1);
        const int o = (NV ? 2 : 1) * (hpos[x] - cs);
        spd[c] = o >> 2;
        const uint32_t b = (uint32_t)(o & 3);
        if constexpr (NV) {                                   // sel0 / sel1: the first / second component's bytes of an interleaved pair of pairs
            const uint32_t b0 = b + (p.uv_swap_src ? 1u : 0u), b1 = b + (p.uv_swap_src ? 0u : 1u);
            sel0[c] = 0x0c000c00u | b0 | ((b0 + 2) << 16);
            sel1[c] = 0x0c000c00u | b1 | ((b1 + 2) << 16);
        } else {
            sel0[c] = 0x0c000c00u | b | ((b + 1) << 16);      // {byte b, 0, byte b + 1, 0}
            sel1[c] = 0x0c000c00u | (b + 2) | ((b + 3) << 16);
        }
        const uint32_t *tp = (const uint32_t *)(g.hT8 + (int64_t)x * (2 * NPH));
#pragma unroll
        for (int k = 0; k < NPH; k++) ht[c][k] = tp[k];
    }
    // ---- source descriptors (whole rows including their padding), as plain dwords for the asm statements ----
    const bool u1 = p.u_plane_src == 1;
    i32x4s rs[NSRC];
    int sst[NSRC];
#pragma unroll
    for (int ci = 0; ci < NSRC; ci++) {
        const bool first = !CHROMA || NV || ((ci == 0) == u1);
        const uint8_t *sb = !CHROMA ? f.src[0] : (first ? U(f.src[1]) : U(f.src[2]));
        sst[ci] = !CHROMA ? f.srcStride[0] : (first ? U(f.srcStride[1]) : U(f.srcStride[2]));
        const uint64_t a = uniform_u64((uint64_t)sb);
        rs[ci][0] = (int)(uint32_t)a; rs[ci][1] = (int)(uint32_t)(a >> 32);
        rs[ci][2] = __builtin_amdgcn_readfirstlane((int)((uint32_t)sst[ci] * (uint32_t)sH)); rs[ci][3] = 0x00020000;
    }
    const int voff = (NV ? 2 : 1) * cs + lane * 16;
    const int n0 = min(chunks, 64);
    const uint32_t m0lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n0 >= 32 ? 0xffffffffu : ((1u << n0) - 1u)));
    const uint32_t m0hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n0 >= 64 ? 0xffffffffu : (n0 > 32 ? ((1u << (n0 - 32)) - 1u) : 0u)));
    auto dma = [&](int q) {                                    // request source rows 2q, 2q + 1 (clamped) into ring slot q % D
        const int r0 = min(max(2 * q, 0), sH - 1), r1 = min(max(2 * q + 1, 0), sH - 1);
        const uint32_t slot = lds_base + (uint32_t)((q & (D - 1)) * pair_dw) * 4u;
#pragma unroll
        for (int ci = 0; ci < NSRC; ci++)
#pragma unroll
            for (int r = 0; r < 2; r++)
                strip_dma16(slot + (uint32_t)((ci * 2 + r) * row_dw) * 4u, voff, rs[ci], (r ? r1 : r0) * sst[ci], m0lo, m0hi);
    };
    // vmcnt counts the wave's loads AND stores, and only loads return in order among themselves (stores are acknowledged independently): the one safe
    // bound is the younger LOADS -- if pair q were outstanding so would the (D - 1) * P requests behind it, hence "at most (D - 1) * P operations
    // outstanding" implies that q has landed, whatever the stores of earlier rows are doing.  (Round 4 tried to allow for the stores issued behind the
    // request as well, (D - 1) * (P + S): wrong -- with the stores acknowledged early, q and its younger requests alone stay below that bound; the
    // band-start rows of a 6:1 conversion showed it.)
    constexpr int P = NSRC * 2;
    auto wait_pair = [&]() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 1) * P) : "memory"); };
    // ---- destination descriptors and per-lane offsets (columns beyond the plane get an out-of-range offset) ----
    const bool semi = CHROMA && (p.dstKind == DSTK_NV12 || p.dstKind == DSTK_P010);
    const bool d8 = p.dstKind == DSTK_PLANAR8 || p.dstKind == DSTK_NV12;
    const bool raw = p.dstKind == DSTK_RAW32;
    const int kind = raw ? 4 : semi ? (d8 ? 2 : 3) : (d8 ? 0 : 1);
    const int dbytes = raw ? 4 : (d8 ? 1 : 2) * (semi ? 2 : 1);
    sws_rsrc_t rd[NCOMP];
    int dstr[NCOMP];
#pragma unroll
    for (int ci = 0; ci < NCOMP; ci++) {
        const int pl = !CHROMA ? 0 : semi ? 1 : (ci == 0 ? p.u_plane_dst : p.v_plane_dst);
        uint8_t *db = pl == 0 ? U(f.dst[0]) : pl == 1 ? U(f.dst[1]) : U(f.dst[2]);
        dstr[ci] = pl == 0 ? U(f.dstStride[0]) : pl == 1 ? U(f.dstStride[1]) : U(f.dstStride[2]);
        rd[ci] = make_rsrc(db, (uint32_t)dstr[ci] * (uint32_t)(H - 1) + (uint32_t)W * (uint32_t)dbytes);
    }
    int doff[COLS];
#pragma unroll
    for (int c = 0; c < COLS; c++) {
        const int x = xs + 64 * c + lane;
        doff[c] = x < W ? x * dbytes : 0x7fffffff;
    }

    uint32_t ring[NCOMP][COLS][RD];
#pragma unroll
    for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
        for (int c = 0; c < COLS; c++)
#pragma unroll
            for (int k = 0; k < RD; k++) ring[ci][c][k] = 0;

    uint32_t pend[NCOMP][COLS];
    int pend_y = -1;
    auto flush = [&]() {
        if (pend_y >= 0) {
            switch (kind) {
            case 0:
#pragma unroll
                for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                    for (int c = 0; c < COLS; c++) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)pend[ci][c], rd[ci], doff[c], pend_y * dstr[ci], 0);
                break;
            case 1:
#pragma unroll
                for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                    for (int c = 0; c < COLS; c++) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)pend[ci][c], rd[ci], doff[c], pend_y * dstr[ci], 0);
                break;
            case 2:
#pragma unroll
                for (int c = 0; c < COLS; c++)
                    __builtin_amdgcn_raw_buffer_store_b16((uint16_t)(pend[0][c] | (pend[NCOMP - 1][c] << 8)), rd[0], doff[c], pend_y * dstr[0], 0);
                break;
            case 4:
#pragma unroll
                for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                    for (int c = 0; c < COLS; c++) __builtin_amdgcn_raw_buffer_store_b32(pend[ci][c], rd[ci], doff[c], pend_y * dstr[ci], 0);
                break;
            default:
#pragma unroll
                for (int c = 0; c < COLS; c++)
                    __builtin_amdgcn_raw_buffer_store_b32(pend[0][c] | (pend[NCOMP - 1][c] << 16), rd[0], doff[c], pend_y * dstr[0], 0);
                break;
            }
            pend_y = -1;
        }
    };

    // ---- march ----
    const SwsStripRow *rows = g.rows;
    StripRowN<RD> e = load_strip_row_n<RD>(rows, y0);
    int qnext = e.pf;                                          // next source-row pair to h-scale
    int qdma = qnext;                                          // next pair to request
    // The column state above came through vector loads the COMPILER counts; the DMA requests below are asm statements it does not see.  Its own
    // `s_waitcnt vmcnt(n)` for those loads allows for the n operations it knows to be younger (none today: it drains everything) -- with the unseen requests
    // in between, "at most n outstanding" would no longer imply that the loads have returned.  So they are consumed HERE, before the first request:
    // the empty asm reads every loaded register, which makes the compiler wait for them while its count is still exact.
#pragma unroll
    for (int c = 0; c < COLS; c++) {
        asm volatile("" : "+v"(spd[c]), "+v"(sel0[c]), "+v"(sel1[c]));
#pragma unroll
        for (int k = 0; k < NPH; k++) asm volatile("" : "+v"(ht[c][k]));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < D; i++) dma(qdma++);
    const int bits = p.dst_bits;
    for (int y = y0; y < y1; y++) {
        const StripRowN<RD> en = load_strip_row_n<RD>(rows, min(y + 1, H - 1));
        const int pfy = e.pf;
        while (qnext <= pfy + npv - 1) {
            wait_pair();
            const uint32_t *S = ringS + (qnext & (D - 1)) * pair_dw;
            uint32_t np[NCOMP][COLS];
            if (g.hfs2 < 0) {          // (never: keeps the horizontal stage in a basic block of its own, see strip_body)
#pragma unroll
                for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                    for (int c = 0; c < COLS; c++) np[ci][c] = S[(NV ? 0 : ci * 2) * row_dw + spd[c]];
            } else if constexpr (NV) {
#pragma unroll
                for (int c = 0; c < COLS; c++) {
                    const uint32_t *s0 = S + spd[c], *s1 = s0 + row_dw;
                    uint32_t d0[NDW], d1[NDW];
#pragma unroll
                    for (int j = 0; j < NDW; j++) { d0[j] = s0[j]; d1[j] = s1[j]; }
#pragma unroll
                    for (int ci = 0; ci < 2; ci++) {
                        const uint32_t sl = ci ? sel1[c] : sel0[c];
                        int a = sdot2_first(__builtin_amdgcn_perm(d0[1], d0[0], sl), ht[c][0]);
                        int b = sdot2_first(__builtin_amdgcn_perm(d1[1], d1[0], sl), ht[c][0]);
#pragma unroll
                        for (int k = 1; k < NPH; k++) {
                            a = sdot2(__builtin_amdgcn_perm(d0[k + 1], d0[k], sl), ht[c][k], a);
                            b = sdot2(__builtin_amdgcn_perm(d1[k + 1], d1[k], sl), ht[c][k], b);
                        }
                        np[ci][c] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(a >> sh, b >> sh));
                    }
                }
            } else {
#pragma unroll
                for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                    for (int c = 0; c < COLS; c++) {
                        const uint32_t *s0 = S + (ci * 2) * row_dw + spd[c], *s1 = s0 + row_dw;
                        uint32_t d0[NDW], d1[NDW];
#pragma unroll
                        for (int j = 0; j < NDW; j++) { d0[j] = s0[j]; d1[j] = s1[j]; }
                        int a = sdot2_first(__builtin_amdgcn_perm(d0[1], d0[0], sel0[c]), ht[c][0]);
                        int b = sdot2_first(__builtin_amdgcn_perm(d1[1], d1[0], sel0[c]), ht[c][0]);
#pragma unroll
                        for (int k = 1; k < NPH; k++) {
                            const uint32_t sl = (k & 1) ? sel1[c] : sel0[c];
                            a = sdot2(__builtin_amdgcn_perm(d0[(k >> 1) + 1], d0[k >> 1], sl), ht[c][k], a);
                            b = sdot2(__builtin_amdgcn_perm(d1[(k >> 1) + 1], d1[k >> 1], sl), ht[c][k], b);
                        }
                        // min(v >> sh, 32767) + int16 store == v_cvt_pk_i16_i32's saturation (see strip_hstage)
                        np[ci][c] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(a >> sh, b >> sh));
                    }
            }
            if (rng.on) strip_range<NCOMP, COLS>(np, rng);
#pragma unroll
            for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                for (int c = 0; c < COLS; c++) {
#pragma unroll
                    for (int k = 0; k < RD - 1; k++) ring[ci][c][k] = ring[ci][c][k + 1];
                    ring[ci][c][RD - 1] = np[ci][c];
                }
            qnext++;
            // the slot's ds_reads have returned (their results are in np): pin that, release the pending row, re-request the slot
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            flush();
            dma(qdma++);
        }
        flush();
        // ---- vertical stage: the npv newest ring entries are pairs pfy .. pfy + npv - 1 ----
        int acc[NCOMP][COLS];
        switch (npv) {
#define SWS_SV8(N) case N: if constexpr (N <= RD) { \
            _Pragma("unroll") for (int ci = 0; ci < NCOMP; ci++) _Pragma("unroll") for (int c = 0; c < COLS; c++) { \
                acc[ci][c] = sdot2_first_s(ring[ci][c][RD - N < 0 ? 0 : RD - N], e.vt[0]); \
                _Pragma("unroll") for (int k = 1; k < N; k++) acc[ci][c] = sdot2(ring[ci][c][RD - N + k < 0 ? 0 : RD - N + k], e.vt[k], acc[ci][c]); } } \
            break;
        SWS_SV8(1) SWS_SV8(2) SWS_SV8(3) SWS_SV8(4) SWS_SV8(5) SWS_SV8(6) SWS_SV8(7)
#undef SWS_SV8
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
        // ---- writers ("X" forms) ----
        if (raw) {
#pragma unroll
            for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                for (int c = 0; c < COLS; c++) pend[ci][c] = (uint32_t)acc[ci][c];
        } else if (d8) {
#pragma unroll
            for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                for (int c = 0; c < COLS; c++) {
                    const int x = xs + 64 * c + lane;
                    const int off = (CHROMA && ci == 1) ? 3 : 0;
                    pend[ci][c] = (uint32_t)clip_u8_shr((dither8(p.should_dither, y, x + off) << 12) + acc[ci][c], 19);
                }
        } else {
            const int shift = 11 + 16 - bits, osh = p.dst_shift;
#pragma unroll
            for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                for (int c = 0; c < COLS; c++)
                    pend[ci][c] = (uint32_t)(clip_uintp2(((1 << (shift - 1)) + acc[ci][c]) >> shift, bits) << osh);
        }
        if (semi && p.uv_swap_dst) {
#pragma unroll
            for (int c = 0; c < COLS; c++) { const uint32_t t = pend[0][c]; pend[0][c] = pend[NCOMP - 1][c]; pend[NCOMP - 1][c] = t; }
        }
        pend_y = y;
        e = en;
    }
    flush();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no DMA write may land after the wave has given up its LDS
}

template <bool CHROMA, int COLS, int NPH, int RD, bool NV = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) sws_k_strip_dma8(SwsFrameSet fs, SwsDevParams p, SwsStripGeom g)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wid = blockIdx.x * 4 + wib;
    if (wid >= g.strips * g.bands) return;
    const int strip = wid % g.strips, band = wid / g.strips;
    const int H = CHROMA ? p.chrDstH : p.dstH;
    const int y0 = band * g.band_rows, y1 = min(H, y0 + g.band_rows);
    if (y0 >= y1) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    strip_body_dma8<CHROMA, COLS, NPH, RD, NV>(f, p, g, strip, y0, y1, smem, wib, lane);
}

// EXPERIMENT (round 6, unmeasured: `sws_hip_set_option("exp3", 1)`): the luma and the chroma launch of the byte-DMA form as ONE grid -- blocks [0, blocksL) walk luma
// bands, the rest chroma bands, both bodies unchanged (the pattern of sws_k_strip_dma_lc, kernels_strip.hpp).  C1 at batch sizes pays two dependent launches and
// their tails per call (70 + 46 us for 256 frames).
template <int COLS_L, int COLS_C, int NPH_L, int NPH_C, int RD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) sws_k_strip_dma8_lc(SwsFrameSet fs, SwsDevParams p, SwsStripGeom gl, SwsStripGeom gc, int blocksL)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool chroma = (int)blockIdx.x >= blocksL;
    const SwsStripGeom &g = chroma ? gc : gl;
    const int wid = ((int)blockIdx.x - (chroma ? blocksL : 0)) * 4 + wib;
    if (wid >= g.strips * g.bands) return;
    const int strip = wid % g.strips, band = wid / g.strips;
    const int H = chroma ? p.chrDstH : p.dstH;
    const int y0 = band * g.band_rows, y1 = min(H, y0 + g.band_rows);
    if (y0 >= y1) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    if (chroma) strip_body_dma8<true, COLS_C, NPH_C, RD, false>(f, p, gc, strip, y0, y1, smem, wib, lane);
    else strip_body_dma8<false, COLS_L, NPH_L, RD, false>(f, p, gl, strip, y0, y1, smem, wib, lane);
}

} // namespace swsk
