// Marching strip kernel for the general h+v polyphase chain on planar sources (the C1 / C3b shapes):
// readers + hScale8To15_c / hScale16To15_c (swscale.c:99-142) + yuv2planeX_8_c / yuv2planeX_10_c / yuv2nv12cX_c /
// yuv2p01x*X_c (output.c:327-357, :468-528, :538-589).  Same arithmetic as sws_k_tile_dot2, different schedule:
//
//  * a wave owns a strip of 64 * COLS output columns (lane l: columns l, l + 64, ...; neighbouring lanes read neighbouring
//    LDS dwords, so the horizontal stage is bank-conflict free at 2:1) and walks down a band of output rows;
//  * per step it h-scales ONE PAIR of source rows (rows 2q, 2q+1 -> one dword {even row, odd row} per column, which is the
//    operand layout of v_dot2_i32_i16 for the vertical stage) and keeps the last pairs in a REGISTER ring: the vertical
//    stage is lane-local, h-scaled samples never touch LDS or HBM, and no source row is h-scaled twice inside a band
//    (the tile kernel re-did its vertical halo: (2 * 32 + 12) / 64 rows per output row at 2:1 Lanczos);
//  * the only LDS traffic is the wave-private staging of the two source rows (written once, read by every tap);
//  * horizontal taps live in registers for the whole band, vertical taps and row positions arrive through scalar loads
//    one row ahead, the next pair of source rows is prefetched into registers while the current one is computed;
//  * no block-level barrier anywhere: the 4 waves of a block are independent;
//  * all memory goes through buffer descriptors (constant per-lane offsets, scalar row offsets, out-of-range lanes dropped
//    by the hardware); vmcnt counts loads and stores together and the compiler must assume they retire out of order, so a
//    row's stores are issued AFTER the wait for the prefetched rows and BEFORE the next prefetch is issued.
#pragma once
#include "wave_util.hpp"

namespace swsk {

// vertical stage over the N newest ring entries (the body of one case of the switch over the row-pair count)
#define SWS_SVB(N) \
    _Pragma("unroll") for (int ci = 0; ci < NCOMP; ci++) _Pragma("unroll") for (int c = 0; c < COLS; c++) { \
        acc[ci][c] = sdot2_first_s(ring[ci][c][RD - (N)], e.vt[0]); \
        _Pragma("unroll") for (int k = 1; k < (N); k++) acc[ci][c] = sdot2(ring[ci][c][RD - (N) + k], e.vt[k], acc[ci][c]); }

// g.debug switches stages off for profiling (results are WRONG): only in -DSWS_HIP_PROFILING builds
#ifdef SWS_HIP_PROFILING
#define SWS_DBG(g, bit) ((g).debug & (bit))
#else
#define SWS_DBG(g, bit) false
#endif

struct StripLds { uint32_t *S; int row_dw; };

// One plan entry through the scalar data cache (s_load_dwordx*, lgkmcnt).  Written as loads from the constant address space because
// hipcc turns a plain `rows[y]` inside the march loop into global_load + v_readfirstlane: the kernel stores to global memory, so the
// loop's loads are not provably unclobbered and lose their scalar form.  A vector load there is poison for the schedule: vmcnt is one
// in-order counter, so the `s_waitcnt vmcnt(n)` that guards the entry also waits for every source row requested before it -- the
// prefetch distance collapses to nothing once per output row (measured: C3b 0.289 -> see DESIGN.md 6).
__device__ __forceinline__ SwsStripRow load_strip_row(const SwsStripRow *rows, int idx)
{
    typedef const uint32_t __attribute__((address_space(4))) *cptr;
    cptr q = (cptr)(uintptr_t)(rows + idx);
    SwsStripRow e;
    e.pf = (int)q[0];
#pragma unroll
    for (int k = 0; k < 8; k++) e.vt[k] = q[4 + k];
    return e;
}
// The same entry with RD tap pairs: the host writes pairs 8 .. 11 of a long vertical filter (chroma at 4:1: 17 bicubic taps) into the four
// spare dwords behind vt[8] of the 64-byte entry
template <int RD> struct StripRowN { int pf; uint32_t vt[RD]; };
// (RD == 16, the long-filter form: two 64-byte entries per row, the tap pairs run on from the first into the second)
template <int RD>
__device__ __forceinline__ StripRowN<RD> load_strip_row_n(const SwsStripRow *rows, int idx)
{
    typedef const uint32_t __attribute__((address_space(4))) *cptr;
    cptr q = (cptr)(uintptr_t)(rows + (RD > 24 ? 4 : RD > 12 ? 2 : 1) * idx);     // (two entries hold pf + up to 28 pairs, four as many as any form takes)
    StripRowN<RD> e;
    e.pf = (int)q[0];
#pragma unroll
    for (int k = 0; k < RD; k++) e.vt[k] = q[4 + k];
    return e;
}

// horizontal stage of one row pair for COLS columns of NCOMP components -> one packed dword per (component, column)
template <int NP, int NCOMP, int COLS>
__device__ __forceinline__ void strip_hstage(const StripLds &L, const int (&spd)[COLS], const uint32_t (&ht)[COLS][NP], int sh,
                                             uint32_t (&out)[NCOMP][COLS])
{
#pragma unroll
    for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
        for (int c = 0; c < COLS; c++) {
            const uint32_t *s0 = L.S + (ci * 2) * L.row_dw + spd[c], *s1 = s0 + L.row_dw;
            int a = sdot2_first(s0[0], ht[c][0]), b = sdot2_first(s1[0], ht[c][0]);
#pragma unroll
            for (int k = 1; k < NP; k++) { a = sdot2(s0[k], ht[c][k], a); b = sdot2(s1[k], ht[c][k], b); }
            // min(v >> sh, 32767) + int16 store == v_cvt_pk_i16_i32's saturation (see hscale_pairs_impl in kernels_tile.hpp)
            out[ci][c] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(a >> sh, b >> sh));
        }
}

// The same stage for SRC16 instantiations, with a per-column addend.  Samples of 16 significant bits (yuv4xxp16, p016, the reader planes of 16-bit RGB
// sources) are no v_dot2_i32_i16 operands as they are: the staging step flips their top bit (s' = s - 32768 as a signed 16-bit value) and the addend gives
// the difference back -- Σ s t = Σ s' t + 32768 Σ t, with Σ t the column's own tap sum (initFilter leaves it at 16384 give or take its error
// diffusion: summed from the tap registers, not assumed).  |Σ s' t| <= 2^15 Σ|t| and the reference's own int sum Σ s t both stay inside 32 bits.
// Sources of up to 15 bits run the same code with a zero addend.
template <int NP, int NCOMP, int COLS>
__device__ __forceinline__ void strip_hstage_b(const StripLds &L, const int (&spd)[COLS], const uint32_t (&ht)[COLS][NP], const int (&hb)[COLS], int sh,
                                               uint32_t (&out)[NCOMP][COLS])
{
#pragma unroll
    for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
        for (int c = 0; c < COLS; c++) {
            const uint32_t *s0 = L.S + (ci * 2) * L.row_dw + spd[c], *s1 = s0 + L.row_dw;
            int a = hb[c], b = hb[c];
#pragma unroll
            for (int k = 0; k < NP; k++) { a = sdot2(s0[k], ht[c][k], a); b = sdot2(s1[k], ht[c][k], b); }
            out[ci][c] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(a >> sh, b >> sh));
        }
}
// the addend of column taps ht[c][0 .. NP): 32768 * (sum of the 2 NP taps) for 16-bit samples, else 0
template <int NP, int COLS>
__device__ __forceinline__ void strip_u16_bias(const uint32_t (&ht)[COLS][NP], bool u16, int (&hb)[COLS])
{
#pragma unroll
    for (int c = 0; c < COLS; c++) {
        int sum = 0;
#pragma unroll
        for (int k = 0; k < NP; k++) sum += (int)(int16_t)(uint16_t)ht[c][k] + ((int)ht[c][k] >> 16);
        hb[c] = u16 ? sum << 15 : 0;
    }
}

// MPEG <-> JPEG range conversion of the h-scaled lines (lum / chrRange{To,From}Jpeg_c, swscale.c:163-209; applied per line behind the horizontal
// scaler, hscale.c:61-63, :195-197): dst = (dst * coeff + offset) >> 14 in int arithmetic on the int16 line, the ToJpeg forms clip to 2^15 - 1, the
// store truncates to int16.  Here the line is the packed {even row, odd row} dword of a column on its way into the ring.  Wave-uniform; a block of
// its own behind the stage, so that contexts without range conversion pay one scalar branch per step.
struct StripRange { int on, coeff, offset, clipmax; };
__device__ __forceinline__ StripRange strip_range_of(const SwsDevParams &p, bool chroma)
{
    StripRange r;
    r.on = p.range_active;
    r.coeff = (int)(uint16_t)(chroma ? p.chrCoeff : p.lumCoeff);
    r.offset = (int32_t)(chroma ? p.chrOffset : p.lumOffset);
    r.clipmax = p.range_to_jpeg ? (1 << 15) - 1 : 0x7fffffff;
    return r;
}
template <int NCOMP, int COLS>
__device__ __forceinline__ void strip_range(uint32_t (&np)[NCOMP][COLS], const StripRange &r)
{
#pragma unroll
    for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
        for (int c = 0; c < COLS; c++) {
            const int lo = min(mad24((int)(int16_t)(uint16_t)np[ci][c], r.coeff, r.offset) >> 14, r.clipmax);
            const int hi = min(mad24((int)np[ci][c] >> 16, r.coeff, r.offset) >> 14, r.clipmax);
            np[ci][c] = __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u);      // {(int16)lo, (int16)hi}
        }
}

// PARTS: 16-byte chunks per lane and staged source row (1: windows of up to 64 chunks -- what 8-bit sources need at ratios up to about 3:1 on 320-column
// strips; the second chunk's loads, byte expansion and LDS writes are straight-line code that costs the same whether any lane uses them or not)
template <bool SRC16, bool CHROMA, int COLS, int NPH, int RD = 8, int PARTS = (CHROMA ? 1 : 2)>
__device__ __forceinline__ void strip_body(const FrameRegs &f, const SwsDevParams &p, const SwsStripGeom &g, int strip, int y0, int y1,
                                           uint8_t *smem, int wib, int lane)
{
    constexpr int NCOMP = CHROMA ? 2 : 1;
    constexpr int SPC = SRC16 ? 8 : 16;                       // samples per 16-byte source chunk
    const int W = CHROMA ? p.chrDstW : p.dstW, H = CHROMA ? p.chrDstH : p.dstH;
    const int sH = CHROMA ? p.chrSrcH : p.srcH;
    const int xs = strip * g.TW;
    const int cs = g.colStart[strip], chunks = g.colCount[strip] / SPC;
    const int32_t *hpos = CHROMA ? p.hChrPos : p.hLumPos;
    const int npv = g.npv, sh = p.hshift;
    const StripRange rng = strip_range_of(p, CHROMA);
    StripLds L;
    L.row_dw = (g.NCmax + SPC) >> 1;                          // one spare chunk per row: the dump slot of idle lanes
    L.S = (uint32_t *)smem + wib * (NCOMP * 2 * L.row_dw);

    // ---- per-lane column state: window offsets and horizontal taps (registers for the whole band) ----
    int spd[COLS];
    uint32_t ht[COLS][NPH];
#pragma unroll
    for (int c = 0; c < COLS; c++) {
        const int x = min(xs + 64 * c + lane, W - 1);
        spd[c] = ((hpos[x] & ~1) - cs) >> 1;
        const uint32_t *tp = (const uint32_t *)(g.hT2 + (int64_t)x * g.hfs2);
#pragma unroll
        for (int k = 0; k < NPH; k++) ht[c][k] = tp[k];
    }
    // (16-bit samples: see strip_hstage_b)
    const bool u16 = SRC16 && p.src_depth >= 16;
    const uint32_t sxor = u16 ? 0x80008000u : 0u;
    int hb[COLS];
    strip_u16_bias<NPH, COLS>(ht, u16, hb);
    // ---- source descriptors (whole rows including their padding) ----
    const bool u1 = p.u_plane_src == 1;
    // nv12 / nv21 sources (8-bit): both chroma components come out of plane 1, de-interleaved on the way into LDS (nvXXtoUV_c, input.c:926-948)
    // p010 / p012 sources (16-bit words, samples in the high bits): the same for 16-bit pairs, and every sample >> src_shift (p010LEToY_c / p010LEToUV_c, :950-1008)
    const bool nvsrc = CHROMA && (SRC16 ? p.srcKind == SRCK_P010 : p.srcKind == SRCK_NV12);
    const int sshift = (SRC16 && p.srcKind == SRCK_P010) ? p.src_shift : 0;
    const uint32_t smask = (0xFFFFu >> sshift) * 0x10001u;
    sws_rsrc_t rs[NCOMP];
    int sst[NCOMP];
#pragma unroll
    for (int ci = 0; ci < NCOMP; ci++) {
        const bool first = !CHROMA || nvsrc || ((ci == 0) == u1);
        const uint8_t *sb = !CHROMA ? f.src[0] : (first ? U(f.src[1]) : U(f.src[2]));   // (U(): keeps the frame descriptor out of scratch memory, kernels_common.hpp)
        sst[ci] = !CHROMA ? f.srcStride[0] : (first ? U(f.srcStride[1]) : U(f.srcStride[2]));
        rs[ci] = make_rsrc(sb, (uint32_t)sst[ci] * (uint32_t)sH);
    }
    // Two 16-byte chunks per lane and source row, unconditionally (straight-line code): a lane whose chunk lies beyond the strip's
    // window gets an out-of-range offset (the descriptor answers 0 without touching memory) and dumps into the spare LDS chunk.
    const int sbase = cs * (SRC16 ? 2 : 1) + lane * 16;
    // (chroma strips are 128 columns wide and take one chunk per lane: windows of up to 64 chunks, checked on the host)
    constexpr bool TWO = PARTS == 2;
    const int voff0 = lane < chunks ? sbase : 0x7fffffff, voff1 = 64 + lane < chunks ? sbase + 1024 : 0x7fffffff;
    const int slot0 = min(lane, chunks) * (SPC / 2), slot1 = min(64 + lane, chunks) * (SPC / 2);

    u32x4 pre[NCOMP * 4];                                      // [component][row of the pair][chunk]
    const int nvbase = cs * (SRC16 ? 4 : 2) + lane * 32;      // a pair is 2 or 4 bytes; a lane takes 32 bytes = one chunk of each component
    const int nvoff0 = lane < chunks ? nvbase : 0x7fffffff, nvoff1 = lane < chunks ? nvbase + 16 : 0x7fffffff;
    auto prefetch = [&](int q) {                               // source rows 2q, 2q+1 (clamped) -> registers
        const int r0 = min(max(2 * q, 0), sH - 1), r1 = min(max(2 * q + 1, 0), sH - 1);
        if constexpr (CHROMA) {
            if (nvsrc) {   // 32 bytes of interleaved pairs per lane and row -> one 16-byte chunk of each component (16 bytes or 8 words)
                const uint32_t s0 = SRC16 ? 0x05040100u : 0x06040200u, s1 = SRC16 ? 0x07060302u : 0x07050301u;
                const uint32_t se = p.uv_swap_src ? s1 : s0, so = p.uv_swap_src ? s0 : s1;
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const int ro = (r ? r1 : r0) * sst[0];
                    const u32x4 a = bload16(rs[0], nvoff0, ro), b = bload16(rs[0], nvoff1, ro);
                    u32x4 u, v;
                    u[0] = __builtin_amdgcn_perm(a[1], a[0], se); u[1] = __builtin_amdgcn_perm(a[3], a[2], se);
                    u[2] = __builtin_amdgcn_perm(b[1], b[0], se); u[3] = __builtin_amdgcn_perm(b[3], b[2], se);
                    v[0] = __builtin_amdgcn_perm(a[1], a[0], so); v[1] = __builtin_amdgcn_perm(a[3], a[2], so);
                    v[2] = __builtin_amdgcn_perm(b[1], b[0], so); v[3] = __builtin_amdgcn_perm(b[3], b[2], so);
                    pre[2 * r] = u; pre[4 + 2 * r] = v;
                }
                return;
            }
        }
#pragma unroll
        for (int ci = 0; ci < NCOMP; ci++) {
            pre[4 * ci + 0] = bload16(rs[ci], voff0, r0 * sst[ci]);
            if constexpr (TWO) pre[4 * ci + 1] = bload16(rs[ci], voff1, r0 * sst[ci]);
            pre[4 * ci + 2] = bload16(rs[ci], voff0, r1 * sst[ci]);
            if constexpr (TWO) pre[4 * ci + 3] = bload16(rs[ci], voff1, r1 * sst[ci]);
        }
    };
    auto put = [&](uint32_t *dst, const u32x4 &v) {
        if constexpr (SRC16) {
            if (sshift) { u32x4 w; w[0] = (v[0] >> sshift) & smask; w[1] = (v[1] >> sshift) & smask; w[2] = (v[2] >> sshift) & smask; w[3] = (v[3] >> sshift) & smask; *(u32x4 *)dst = w; }
            else { u32x4 w; w[0] = v[0] ^ sxor; w[1] = v[1] ^ sxor; w[2] = v[2] ^ sxor; w[3] = v[3] ^ sxor; *(u32x4 *)dst = w; }
        } else {
            u32x4 lo, hi;                                      // bytes -> u16 pairs
            lo[0] = __builtin_amdgcn_perm(0, v[0], 0x0c010c00u); lo[1] = __builtin_amdgcn_perm(0, v[0], 0x0c030c02u);
            lo[2] = __builtin_amdgcn_perm(0, v[1], 0x0c010c00u); lo[3] = __builtin_amdgcn_perm(0, v[1], 0x0c030c02u);
            hi[0] = __builtin_amdgcn_perm(0, v[2], 0x0c010c00u); hi[1] = __builtin_amdgcn_perm(0, v[2], 0x0c030c02u);
            hi[2] = __builtin_amdgcn_perm(0, v[3], 0x0c010c00u); hi[3] = __builtin_amdgcn_perm(0, v[3], 0x0c030c02u);
            *(u32x4 *)dst = lo; *(u32x4 *)(dst + 4) = hi;
        }
    };
    auto stage = [&]() {                                       // registers -> the wave's LDS rows (u16 sample pairs)
#pragma unroll
        for (int ci = 0; ci < NCOMP; ci++) {
            uint32_t *row0 = L.S + (ci * 2) * L.row_dw, *row1 = row0 + L.row_dw;
            put(row0 + slot0, pre[4 * ci + 0]); put(row1 + slot0, pre[4 * ci + 2]);
            if constexpr (TWO) { put(row0 + slot1, pre[4 * ci + 1]); put(row1 + slot1, pre[4 * ci + 3]); }
        }
    };

    // ---- destination descriptors and per-lane offsets (columns beyond the plane get an out-of-range offset) ----
    const bool semi = CHROMA && (p.dstKind == DSTK_NV12 || p.dstKind == DSTK_P010);
    const bool d8 = p.dstKind == DSTK_PLANAR8 || p.dstKind == DSTK_NV12;
    const bool raw = p.dstKind == DSTK_RAW32;                  // the vertical sums themselves (int32 planes: the full-chroma RGB epilogue follows)
    const int kind = raw ? 4 : semi ? (d8 ? 2 : 3) : (d8 ? 0 : 1);       // store form: b8 / b16 per component, or b16 / b32 of an interleaved pair; b32 sums
    const int dbytes = raw ? 4 : (d8 ? 1 : 2) * (semi ? 2 : 1);          // bytes per stored element
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

    // pending output row (its stores are issued after the next wait for prefetched rows, see the header)
    uint32_t pend[NCOMP][COLS];
    int pend_y = -1;
    auto flush = [&]() {
        if (pend_y >= 0 && !SWS_DBG(g, 4)) {
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
    StripRowN<RD> e = load_strip_row_n<RD>(rows, y0);                  // scalar loads: first ring pair and vertical tap pairs of the row
    int qnext = e.pf;                                          // next source-row pair to h-scale == the pair staged in LDS
    prefetch(qnext);
    __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0): keep the fill out of the loop's wait arithmetic
    stage();
    prefetch(qnext + 1);
    const int bits = p.dst_bits;
    for (int y = y0; y < y1; y++) {
        const StripRowN<RD> en = load_strip_row_n<RD>(rows, min(y + 1, H - 1));   // next row's scalars, one row ahead
        const int pfy = e.pf;
        if (qnext < pfy) {                                     // rows nobody needs (steep down-scaling with short filters): skip
            qnext = pfy;
            prefetch(qnext);
            stage();
            prefetch(qnext + 1);
        }
        while (qnext <= pfy + npv - 1) {
            uint32_t np[NCOMP][COLS];
            // The horizontal stage sits in a basic block of its own: straight-line code lets hipcc interleave it with the ring and the
            // vertical stage until the live set overshoots the 128 VGPRs of 4 waves per SIMD and the loop spills (4x slower, measured).
            // g.hfs2 is never negative; the compiler cannot know that.  (-DSWS_HIP_PROFILING builds switch the stage off with debug bit 1.)
            if (g.hfs2 < 0 || SWS_DBG(g, 1)) {
#pragma unroll
                for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                    for (int c = 0; c < COLS; c++) np[ci][c] = L.S[(ci * 2) * L.row_dw + spd[c]];
            } else
            if constexpr (SRC16) strip_hstage_b<NPH, NCOMP, COLS>(L, spd, ht, hb, sh, np);
            else strip_hstage<NPH, NCOMP, COLS>(L, spd, ht, sh, np);
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
            // LDS rows are consumed: wait for the prefetched pair, stage it, release the pending row, prefetch the next pair
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            stage();
            flush();
            prefetch(qnext + 1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        flush();                                               // (a row that needed no new pair still has to release the previous one)
        // ---- vertical stage: the npv newest ring entries are pairs pfy .. pfy + npv - 1 ----
        int acc[NCOMP][COLS];
        if (SWS_DBG(g, 2)) {
#pragma unroll
            for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                for (int c = 0; c < COLS; c++) acc[ci][c] = (int)ring[ci][c][RD - 1];
        } else
        if constexpr (RD > 16) {   // (the long form's chroma ring: the host lays every row's tap pairs out against the whole ring, older slots get zero taps)
            SWS_SVB(RD)
        } else
        switch (npv) {
#define SWS_SV(N) case N: if constexpr (N <= RD) { \
            _Pragma("unroll") for (int ci = 0; ci < NCOMP; ci++) _Pragma("unroll") for (int c = 0; c < COLS; c++) { \
                acc[ci][c] = sdot2_first_s(ring[ci][c][RD - N < 0 ? 0 : RD - N], e.vt[0]); \
                _Pragma("unroll") for (int k = 1; k < N; k++) acc[ci][c] = sdot2(ring[ci][c][RD - N + k < 0 ? 0 : RD - N + k], e.vt[k], acc[ci][c]); } } \
            break;
        SWS_SV(1) SWS_SV(2) SWS_SV(3) SWS_SV(4) SWS_SV(5) SWS_SV(6) SWS_SV(7)
        case 8: if constexpr (RD > 8) { SWS_SVB(8) } else if constexpr (RD == 8) { SWS_SVB(RD) } break;
        case 9: if constexpr (RD > 8) { SWS_SVB(9) } break;
        case 10: if constexpr (RD > 8) { SWS_SVB(10) } break;
        case 11: if constexpr (RD > 8) { SWS_SVB(11) } break;
        case 12: if constexpr (RD > 12) { SWS_SVB(12) } else { SWS_SVB(RD) } break;
        case 13: if constexpr (RD > 12) { SWS_SVB(13) } break;
        case 14: if constexpr (RD > 12) { SWS_SVB(14) } break;
        case 15: if constexpr (RD > 12) { SWS_SVB(15) } break;
#undef SWS_SV
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
            const int shift = 11 + 16 - bits, osh = p.dst_shift;   // p010-style and msb planar formats keep the samples in the high bits
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
}

// RD: depth of the register ring in row pairs = the longest vertical filter the instantiation takes (8: 16 taps; 12: the chroma planes of a
// 4:1 vertical step -- packed RGB or 4:2:2 sources into a 4:2:0 picture of half the size -- with up to 24)
template <bool SRC16, bool CHROMA, int COLS, int RD = 8>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) sws_k_strip_march(SwsFrameSet fs, SwsDevParams p, SwsStripGeom g)
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
    switch (g.nph) {        // the whole march is instantiated per horizontal tap-pair count: the loop body is branch-free
#define SWS_SB(N) case N: strip_body<SRC16, CHROMA, COLS, N, RD>(f, p, g, strip, y0, y1, smem, wib, lane); break;
    SWS_SB(1) SWS_SB(2) SWS_SB(3) SWS_SB(4) SWS_SB(5) SWS_SB(6) SWS_SB(7) SWS_SB(8)
#undef SWS_SB
    }
}

// Long filters (ratios of 4:1 and more: the lower rungs of an ABR ladder, thumbnails -- bicubic at 4:1 has 17 taps, at 6:1 25): the same march with
// up to 16 horizontal tap pairs and a ring of 16 row pairs (24 for the chroma planes: a packed RGB source into a 4:2:0 picture doubles the vertical
// chroma ratio), on narrower strips (luma 128, chroma 64 columns: the taps and the ring live in registers per column).  The host pads the horizontal
// tap rows to 10 / 12 / 14 / 16 pairs and lays the vertical taps of the chroma planes out against the whole ring (dev_plan*.hip).
template <bool SRC16, bool CHROMA>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) sws_k_strip_long(SwsFrameSet fs, SwsDevParams p, SwsStripGeom g)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int COLS = CHROMA ? 1 : 2;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wid = blockIdx.x * 4 + wib;
    if (wid >= g.strips * g.bands) return;
    const int strip = wid % g.strips, band = wid / g.strips;
    const int H = CHROMA ? p.chrDstH : p.dstH;
    const int y0 = band * g.band_rows, y1 = min(H, y0 + g.band_rows);
    if (y0 >= y1) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    switch (g.nph) {
#define SWS_SB(N) case N: strip_body<SRC16, CHROMA, COLS, N, (CHROMA ? 24 : 16)>(f, p, g, strip, y0, y1, smem, wib, lane); break;
    SWS_SB(10) SWS_SB(12) SWS_SB(14) SWS_SB(16)
#undef SWS_SB
    }
}

// ... and filters of 33 .. 62 taps (ratios of 8:1 to 15:1: thumbnails, preview sprites): 20 / 24 / 28 / 32 horizontal tap pairs, rings of 32 row pairs
// (every row's vertical taps laid out against the whole ring), strips of 64 columns for both plane classes; two waves per SIMD (the taps and the
// rings of the chroma planes alone are 96 registers).
template <bool SRC16, bool CHROMA>
__global__ void __launch_bounds__(256) sws_k_strip_xlong(SwsFrameSet fs, SwsDevParams p, SwsStripGeom g)
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
    switch (g.nph) {
#define SWS_SB(N) case N: strip_body<SRC16, CHROMA, 1, N, 32>(f, p, g, strip, y0, y1, smem, wib, lane); break;
    SWS_SB(20) SWS_SB(24) SWS_SB(28) SWS_SB(32)
#undef SWS_SB
    }
}

// ------------------------------------------------------------------------------------------
// LDS-DMA form of the march for 16-bit sources (C3b).  Same arithmetic, same lane / column mapping, same register ring; the
// source rows take another way into LDS:
//  * `buffer_load_dwordx4 ... lds` writes a row's 16-byte chunks straight into the wave's LDS ring (destination = M0 + lane * 16,
//    lanes outside the strip's window are switched off in EXEC): no staging registers, no ds_write pass, no v_perm;
//  * the ring holds D row pairs: up to D - 1 pairs (>= 6 KB of luma) are in flight per wave while one is being h-scaled -- the
//    register-staged form has one pair (2 KB) in flight, which is what bounds it;
//  * the requests are asm statements the compiler does not count, so every wait is written by hand: before pair q is read,
//    `s_waitcnt vmcnt((D - 1) * P)` (P = DMA instructions per pair).  Loads return in order among themselves: if pair q were
//    still outstanding so would the (D - 1) * P requests behind it, hence at most (D - 1) * P outstanding operations implies pair
//    q has landed -- whatever the stores of earlier rows (same counter, unordered against loads) are doing;
//  * a slot is re-requested only after `lgkmcnt(0)`: every ds_read of the h-stage that consumed it has returned;
//  * the wave drains its requests (vmcnt(0)) before it ends: a late DMA write must not hit LDS that belongs to another wave by then.
// The host selects it when no source row pair inside a band is skipped (pf(y + 1) <= pf(y) + npv: always true for the polyphase
// scalers at ratios up to the filter length).
// ------------------------------------------------------------------------------------------
typedef int i32x4s __attribute__((ext_vector_type(4)));
constexpr int STRIP_DMA_DEPTH = 4;

__device__ __forceinline__ void strip_dma16(uint32_t lds_dst, int voff, const i32x4s &rsrc, int soff, uint32_t mask_lo, uint32_t mask_hi)
{
    // (s_nop 4: the descriptor / offset SGPRs may come straight from v_readfirstlane; M0 is written in the statement that reads it)
    uint64_t keep;
    asm volatile("s_nop 4\n\t"
                 "s_mov_b64 %0, exec\n\t"
                 "s_mov_b32 exec_lo, %5\n\t"
                 "s_mov_b32 exec_hi, %6\n\t"
                 "s_mov_b32 m0, %1\n\t"
                 "s_nop 0\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff), "s"(mask_lo), "s"(mask_hi) : "memory", "m0");
}

// D: row pairs of the LDS ring (pairs in flight + 1).  ABS: row pair q lives in register-ring slot q & (RD - 1) for the whole band -- a new pair overwrites
// ONE slot (selected by a wave-uniform switch) instead of shifting the ring down by one (RD - 1 moves per column and pair: a fifth of the vector
// instructions of the C3b luma launch), and the vertical stage picks its rotation of the taps by the row's first pair (pf & (RD - 1)): eight
// straight-line variants over all RD slots, the plan entry's taps beyond the filter being zero (devparams.h SwsStripRow).
template <bool CHROMA, int COLS, int NPH, int RD = 8, int D = STRIP_DMA_DEPTH, bool ABS = false>
__device__ __forceinline__ void strip_body_dma(const FrameRegs &f, const SwsDevParams &p, const SwsStripGeom &g, int strip, int y0, int y1,
                                               uint8_t *smem, int wib, int lane)
{
    constexpr int NCOMP = CHROMA ? 2 : 1, SPC = 8;
    static_assert(!ABS || RD == 8, "the absolute-slot form is written for a ring of 8 row pairs");
    auto slot_of = [](int q) { return (D & (D - 1)) == 0 ? (q & (D - 1)) : ((q % D) + D) % D; };
    constexpr int PARTS = CHROMA ? 1 : 2;                     // 1 KiB pieces per staged row (chroma windows are <= 64 chunks: host)
    const int W = CHROMA ? p.chrDstW : p.dstW, H = CHROMA ? p.chrDstH : p.dstH;
    const int sH = CHROMA ? p.chrSrcH : p.srcH;
    const int xs = strip * g.TW;
    const int cs = g.colStart[strip], chunks = g.colCount[strip] / SPC;
    const int32_t *hpos = CHROMA ? p.hChrPos : p.hLumPos;
    const int npv = g.npv, sh = p.hshift;
    const StripRange rng = strip_range_of(p, CHROMA);
    const int row_dw = g.NCmax >> 1;                          // dwords of a staged row (NCmax is a multiple of the 8-sample chunk)
    const int pair_dw = NCOMP * 2 * row_dw;
    uint32_t *ringS = (uint32_t *)smem + wib * (D * pair_dw);
    const uint32_t lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)ringS);

    // ---- per-lane column state: window offsets and horizontal taps (registers for the whole band) ----
    int spd[COLS];
    uint32_t ht[COLS][NPH];
#pragma unroll
    for (int c = 0; c < COLS; c++) {
        const int x = min(xs + 64 * c + lane, W - 1);
        spd[c] = ((hpos[x] & ~1) - cs) >> 1;
        const uint32_t *tp = (const uint32_t *)(g.hT2 + (int64_t)x * g.hfs2);
#pragma unroll
        for (int k = 0; k < NPH; k++) ht[c][k] = tp[k];
    }
    // ---- source descriptors (whole rows including their padding), as plain dwords for the asm statements ----
    const bool u1 = p.u_plane_src == 1;
    i32x4s rs[NCOMP];
    int sst[NCOMP];
#pragma unroll
    for (int ci = 0; ci < NCOMP; ci++) {
        const bool first = !CHROMA || ((ci == 0) == u1);
        const uint8_t *sb = !CHROMA ? f.src[0] : (first ? U(f.src[1]) : U(f.src[2]));   // (U(): keeps the frame descriptor out of scratch memory, kernels_common.hpp)
        sst[ci] = !CHROMA ? f.srcStride[0] : (first ? U(f.srcStride[1]) : U(f.srcStride[2]));
        const uint64_t a = uniform_u64((uint64_t)sb);
        rs[ci][0] = (int)(uint32_t)a; rs[ci][1] = (int)(uint32_t)(a >> 32);
        rs[ci][2] = __builtin_amdgcn_readfirstlane((int)((uint32_t)sst[ci] * (uint32_t)sH)); rs[ci][3] = 0x00020000;
    }
    const int voff = cs * 2 + lane * 16;
    const int nparts = (PARTS == 2 && chunks > 64) ? 2 : 1;   // wave-uniform: the second piece is not issued when the window fits one
    const int n0 = min(chunks, 64), n1 = max(chunks - 64, 0);
    auto lanes_lo = [](int n) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)(n >= 32 ? 0xffffffffu : ((1u << n) - 1u))); };
    auto lanes_hi = [](int n) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)(n >= 64 ? 0xffffffffu : (n > 32 ? ((1u << (n - 32)) - 1u) : 0u))); };
    const uint32_t m0lo = lanes_lo(n0), m0hi = lanes_hi(n0), m1lo = lanes_lo(n1), m1hi = lanes_hi(n1);   // EXEC masks of the two pieces
    auto dma = [&](int q) {                                    // request source rows 2q, 2q + 1 (clamped) into ring slot q % D
        const int r0 = min(max(2 * q, 0), sH - 1), r1 = min(max(2 * q + 1, 0), sH - 1);
        const uint32_t slot = lds_base + (uint32_t)(slot_of(q) * pair_dw) * 4u;
#pragma unroll
        for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const uint32_t dst = slot + (uint32_t)((ci * 2 + r) * row_dw) * 4u;
                const int soff = (r ? r1 : r0) * sst[ci];
                strip_dma16(dst, voff, rs[ci], soff, m0lo, m0hi);
                if (PARTS == 2 && nparts == 2) strip_dma16(dst + 1024u, voff + 1024, rs[ci], soff, m1lo, m1hi);
            }
    };

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

    // at most (D - 1) * P operations outstanding.  (Only the younger LOADS bound the wait: stores are acknowledged independently of the loads, so
    // allowing for the stores issued behind the request as well -- tried in round 4 -- lets the wait pass with pair q still in flight.)
    auto wait_pair = [&]() {
        if (NCOMP * 2 * nparts == 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 1) * 4) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 1) * 2) : "memory");
    };
    uint32_t pend[NCOMP][COLS];
    int pend_y = -1;
    auto flush = [&]() {
        if (pend_y >= 0 && !SWS_DBG(g, 4)) {
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
        asm volatile("" : "+v"(spd[c]));
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
            StripLds L;
            L.row_dw = row_dw;
            L.S = ringS + slot_of(qnext) * pair_dw;
            uint32_t np[NCOMP][COLS];
            // The horizontal stage sits in a basic block of its own: straight-line code lets hipcc interleave it with the ring and the
            // vertical stage until the live set overshoots the 128 VGPRs of 4 waves per SIMD and the loop spills (4x slower, measured).
            // g.hfs2 is never negative; the compiler cannot know that.  (-DSWS_HIP_PROFILING builds switch the stage off with debug bit 1.)
            if (g.hfs2 < 0 || SWS_DBG(g, 1)) {
#pragma unroll
                for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                    for (int c = 0; c < COLS; c++) np[ci][c] = L.S[(ci * 2) * L.row_dw + spd[c]];
            } else
            strip_hstage<NPH, NCOMP, COLS>(L, spd, ht, sh, np);
            if (rng.on) strip_range<NCOMP, COLS>(np, rng);
            if constexpr (ABS) {
                switch (qnext & (RD - 1)) {
#define SWS_RING_PUT(K) case K: _Pragma("unroll") for (int ci = 0; ci < NCOMP; ci++) _Pragma("unroll") for (int c = 0; c < COLS; c++) ring[ci][c][K] = np[ci][c]; break;
                SWS_RING_PUT(0) SWS_RING_PUT(1) SWS_RING_PUT(2) SWS_RING_PUT(3) SWS_RING_PUT(4) SWS_RING_PUT(5) SWS_RING_PUT(6) SWS_RING_PUT(7)
#undef SWS_RING_PUT
                }
            } else {
#pragma unroll
            for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                for (int c = 0; c < COLS; c++) {
#pragma unroll
                    for (int k = 0; k < RD - 1; k++) ring[ci][c][k] = ring[ci][c][k + 1];
                    ring[ci][c][RD - 1] = np[ci][c];
                }
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
        if (SWS_DBG(g, 2)) {
#pragma unroll
            for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                for (int c = 0; c < COLS; c++) acc[ci][c] = (int)ring[ci][c][RD - 1];
        } else
        if constexpr (ABS) {       // pair pfy + j sits in slot (pfy + j) & 7: one straight-line rotation per value of pfy & 7
            switch (pfy & (RD - 1)) {
#define SWS_SVA(R) case R: \
                _Pragma("unroll") for (int ci = 0; ci < NCOMP; ci++) _Pragma("unroll") for (int c = 0; c < COLS; c++) { \
                    acc[ci][c] = sdot2_first_s(ring[ci][c][R], e.vt[0]); \
                    _Pragma("unroll") for (int k = 1; k < RD; k++) acc[ci][c] = sdot2(ring[ci][c][(R + k) & (RD - 1)], e.vt[k], acc[ci][c]); } \
                break;
            SWS_SVA(0) SWS_SVA(1) SWS_SVA(2) SWS_SVA(3) SWS_SVA(4) SWS_SVA(5) SWS_SVA(6) SWS_SVA(7)
#undef SWS_SVA
            }
        } else
        if constexpr (RD > 16) {   // (the long form's chroma ring: the host lays every row's tap pairs out against the whole ring, older slots get zero taps)
            SWS_SVB(RD)
        } else
        switch (npv) {
#define SWS_SV(N) case N: \
            _Pragma("unroll") for (int ci = 0; ci < NCOMP; ci++) _Pragma("unroll") for (int c = 0; c < COLS; c++) { \
                acc[ci][c] = sdot2_first_s(ring[ci][c][RD - N], e.vt[0]); \
                _Pragma("unroll") for (int k = 1; k < N; k++) acc[ci][c] = sdot2(ring[ci][c][RD - N + k], e.vt[k], acc[ci][c]); } \
            break;
        SWS_SV(1) SWS_SV(2) SWS_SV(3) SWS_SV(4) SWS_SV(5) SWS_SV(6) SWS_SV(7)
        case 8: if constexpr (RD > 8) { SWS_SVB(8) } else { SWS_SVB(RD) } break;
        case 9: if constexpr (RD > 8) { SWS_SVB(9) } break;
        case 10: if constexpr (RD > 8) { SWS_SVB(10) } break;
        case 11: if constexpr (RD > 8) { SWS_SVB(11) } break;
        case 12: if constexpr (RD > 12) { SWS_SVB(12) } else { SWS_SVB(RD) } break;
        case 13: if constexpr (RD > 12) { SWS_SVB(13) } break;
        case 14: if constexpr (RD > 12) { SWS_SVB(14) } break;
        case 15: if constexpr (RD > 12) { SWS_SVB(15) } break;
#undef SWS_SV
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

template <bool CHROMA, int COLS, int RD = 8>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) sws_k_strip_dma(SwsFrameSet fs, SwsDevParams p, SwsStripGeom g)
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
    switch (g.nph) {
#define SWS_SB(N) case N: strip_body_dma<CHROMA, COLS, N, RD>(f, p, g, strip, y0, y1, smem, wib, lane); break;
    SWS_SB(1) SWS_SB(2) SWS_SB(3) SWS_SB(4) SWS_SB(5) SWS_SB(6) SWS_SB(7) SWS_SB(8)
#undef SWS_SB
    }
}

// The same body with the ring depth, the register-ring form and the occupancy as template arguments (k_strip.hip picks by `strip_dma_depth` / `strip_ring_abs`)
template <bool CHROMA, int COLS, int D, bool ABS, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) sws_k_strip_dma_v(SwsFrameSet fs, SwsDevParams p, SwsStripGeom g)
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
    switch (g.nph) {
#define SWS_SB(N) case N: strip_body_dma<CHROMA, COLS, N, 8, D, ABS>(f, p, g, strip, y0, y1, smem, wib, lane); break;
    SWS_SB(1) SWS_SB(2) SWS_SB(3) SWS_SB(4) SWS_SB(5) SWS_SB(6) SWS_SB(7) SWS_SB(8)
#undef SWS_SB
    }
}

// ------------------------------------------------------------------------------------------
// Luma and chroma launch of the planar strip kernels as ONE grid (blocks [0, blocksL) walk luma bands, the rest chroma bands).  Two reasons:
// a call with few frames is launch- and tail-bound (one frame: two launches of about 8 us each around 10 - 20 us of work), and with one grid the
// bands of both plane classes can be cut to the same length (the separate launches each filled the machine on their own: a 4:2:0 chroma plane
// got bands half as long as the luma plane's, i.e. twice the ring-fill share).
// ------------------------------------------------------------------------------------------
template <bool SRC16, int COLS_L, int COLS_C>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) sws_k_strip_march_lc(SwsFrameSet fs, SwsDevParams p, SwsStripGeom gl, SwsStripGeom gc, int blocksL)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool chroma = (int)blockIdx.x >= blocksL;
    const SwsStripGeom &g = chroma ? gc : gl;
    const int wid = ((int)blockIdx.x - (chroma ? blocksL : 0)) * 4 + wib;
    if (wid >= U(g.strips) * U(g.bands)) return;
    const int strip = wid % U(g.strips), band = wid / U(g.strips);
    const int H = chroma ? p.chrDstH : p.dstH;
    const int y0 = band * U(g.band_rows), y1 = min(H, y0 + U(g.band_rows));
    if (y0 >= y1) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    if (!chroma) {
        switch (gl.nph) {
#define SWS_SB(N) case N: strip_body<SRC16, false, COLS_L, N>(f, p, gl, strip, y0, y1, smem, wib, lane); break;
        SWS_SB(1) SWS_SB(2) SWS_SB(3) SWS_SB(4) SWS_SB(5) SWS_SB(6) SWS_SB(7) SWS_SB(8)
#undef SWS_SB
        }
    } else {
        switch (gc.nph) {
#define SWS_SB(N) case N: strip_body<SRC16, true, COLS_C, N>(f, p, gc, strip, y0, y1, smem, wib, lane); break;
        SWS_SB(1) SWS_SB(2) SWS_SB(3) SWS_SB(4) SWS_SB(5) SWS_SB(6) SWS_SB(7) SWS_SB(8)
#undef SWS_SB
        }
    }
}

template <int COLS_L, int COLS_C>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) sws_k_strip_dma_lc(SwsFrameSet fs, SwsDevParams p, SwsStripGeom gl, SwsStripGeom gc, int blocksL)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool chroma = (int)blockIdx.x >= blocksL;
    const SwsStripGeom &g = chroma ? gc : gl;
    const int wid = ((int)blockIdx.x - (chroma ? blocksL : 0)) * 4 + wib;
    if (wid >= U(g.strips) * U(g.bands)) return;
    const int strip = wid % U(g.strips), band = wid / U(g.strips);
    const int H = chroma ? p.chrDstH : p.dstH;
    const int y0 = band * U(g.band_rows), y1 = min(H, y0 + U(g.band_rows));
    if (y0 >= y1) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    if (!chroma) {
        switch (gl.nph) {
#define SWS_SB(N) case N: strip_body_dma<false, COLS_L, N>(f, p, gl, strip, y0, y1, smem, wib, lane); break;
        SWS_SB(1) SWS_SB(2) SWS_SB(3) SWS_SB(4) SWS_SB(5) SWS_SB(6) SWS_SB(7) SWS_SB(8)
#undef SWS_SB
        }
    } else {
        switch (gc.nph) {
#define SWS_SB(N) case N: strip_body_dma<true, COLS_C, N>(f, p, gc, strip, y0, y1, smem, wib, lane); break;
        SWS_SB(1) SWS_SB(2) SWS_SB(3) SWS_SB(4) SWS_SB(5) SWS_SB(6) SWS_SB(7) SWS_SB(8)
#undef SWS_SB
        }
    }
}

} // namespace swsk
