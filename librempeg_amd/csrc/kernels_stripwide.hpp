// Marching strip kernel for 19-bit intermediates (round 5): the h+v polyphase chain into destinations of 16 bits per component --
// yuv4xxp16, gray16, p016 / p216 / p416 directly, and planar RGB of 16 bits / float32, rgb48 / rgba64 through the int32 sum planes and the generic
// writer's X form (k_generic_dst.hip sws_k_sum_writer).  hScale8To19_c / hScale16To19_c (swscale.c:69-97, :144-159: Σ src · filter >> sh, clipped to
// 2^19 - 1, int32 lines), yuv2planeX_16_c / yuv2nv12cX_16_c (output.c:163-217: val = (1 << 14) - 0x40000000 + Σ line · (unsigned)tap in 32-bit wrap-around
// arithmetic, 0x8000 + clip_int16(val >> 15)); the one-tap form yuv2plane1_16_c ((s + 4) >> 3, clip_uint16: :149-161) is the X arithmetic with the tap 4096
// (floor((4096 s + 2^14) / 2^15) - 2^15 = ((s + 4) >> 3) - 2^15, and 0x8000 + clip_int16(v - 2^15) = clip_uint16(v)): the host enters one-tap banks as 4096.
//
// Same march as strip_body (kernels_strip.hpp): a wave owns a strip of 64 * COLS output columns and walks down a band, one PAIR of source rows
// h-scaled per step with v_dot2_i32_i16 (8-bit samples and samples of up to 15 bits are dot2 operands as they are; taps in registers for the band).
// What differs: a 19-bit sample does not fit a dot2 operand, so the ring keeps the two rows of a pair as two int32 registers and the vertical
// stage is one v_mad_i32_i24 per row and column (19-bit sample x 13-bit tap: both inside 24 bits, the product's low 32 bits are the reference's
// wrap-around product) against taps the scalar unit unpacks from the plan entry's packed pairs.  The host lays a row's tap pairs out against the
// WHOLE ring (older slots get zero taps: plan3's ring_of), so there is one vertical loop of RD pairs: instantiations RD = 2 (one- and two-row vertical banks: a plane that is
// not scaled vertically, 2x up-sampled chroma), RD = 4 (bilinear, bicubic up to 2:1 ...) and RD = 8.
#pragma once
#include "kernels_strip.hpp"

namespace swsk {

// MPEG <-> JPEG range conversion of the 19-bit lines (lum / chrRange{To,From}Jpeg16_c, swscale.c:211-255): ((int64_t)dst * coeff + offset) >> 18 stored as int, the
// ToJpeg forms clip to 2^19 - 1.  Wave-uniform parameters; a block of its own behind the horizontal stage (contexts without it pay one scalar branch per step).
struct StripRangeW { int on; uint32_t coeff; int64_t offset; int clipmax; };
__device__ __forceinline__ StripRangeW strip_range_wide_of(const SwsDevParams &p, bool chroma)
{
    StripRangeW r;
    r.on = p.range_active;
    r.coeff = chroma ? p.chrCoeff : p.lumCoeff;
    r.offset = chroma ? p.chrOffset : p.lumOffset;
    r.clipmax = p.range_to_jpeg ? (1 << 19) - 1 : 0x7fffffff;
    return r;
}
__device__ __forceinline__ int strip_range_wide(int v, const StripRangeW &r)
{
    return min((int)(((int64_t)v * (int64_t)r.coeff + r.offset) >> 18), r.clipmax);
}

template <bool SRC16, bool CHROMA, int COLS, int NPH, int RD>
__device__ __forceinline__ void strip_body_wide(const FrameRegs &f, const SwsDevParams &p, const SwsStripGeom &g, int strip, int y0, int y1,
                                                uint8_t *smem, int wib, int lane)
{
    constexpr int NCOMP = CHROMA ? 2 : 1;
    constexpr int SPC = SRC16 ? 8 : 16;                       // samples per 16-byte source chunk
    const int W = CHROMA ? p.chrDstW : p.dstW, H = CHROMA ? p.chrDstH : p.dstH;
    const int sH = CHROMA ? p.chrSrcH : p.srcH;
    const int xs = strip * g.TW;
    const int cs = g.colStart[strip], chunks = g.colCount[strip] / SPC;
    const int32_t *hpos = CHROMA ? p.hChrPos : p.hLumPos;
    const int npv = g.npv, sh = p.hshift, hclip = p.hclip;
    const StripRangeW rng = strip_range_wide_of(p, CHROMA);
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
    // (16-bit samples: top bit flipped while staging, 32768 * (the column's tap sum) as the chains' addend -- strip_hstage_b, kernels_strip.hpp)
    const bool u16 = SRC16 && p.src_depth >= 16;
    const uint32_t sxor = u16 ? 0x80008000u : 0u;
    int hb[COLS];
    strip_u16_bias<NPH, COLS>(ht, u16, hb);
    // ---- source descriptors and staging: strip_body's (one 16-byte chunk per lane and staged row: windows of at most 64 chunks, checked on the host) ----
    const bool u1 = p.u_plane_src == 1;
    const bool nvsrc = CHROMA && (SRC16 ? p.srcKind == SRCK_P010 : p.srcKind == SRCK_NV12);
    const int sshift = (SRC16 && p.srcKind == SRCK_P010) ? p.src_shift : 0;
    const uint32_t smask = (0xFFFFu >> sshift) * 0x10001u;
    sws_rsrc_t rs[NCOMP];
    int sst[NCOMP];
#pragma unroll
    for (int ci = 0; ci < NCOMP; ci++) {
        const bool first = !CHROMA || nvsrc || ((ci == 0) == u1);
        const uint8_t *sb = !CHROMA ? f.src[0] : (first ? U(f.src[1]) : U(f.src[2]));
        sst[ci] = !CHROMA ? f.srcStride[0] : (first ? U(f.srcStride[1]) : U(f.srcStride[2]));
        rs[ci] = make_rsrc(sb, (uint32_t)sst[ci] * (uint32_t)sH);
    }
    const int sbase = cs * (SRC16 ? 2 : 1) + lane * 16;
    const int voff0 = lane < chunks ? sbase : 0x7fffffff;
    const int slot0 = min(lane, chunks) * (SPC / 2);
    u32x4 pre[NCOMP * 2];                                      // [component][row of the pair]
    const int nvbase = cs * (SRC16 ? 4 : 2) + lane * 32;
    const int nvoff0 = lane < chunks ? nvbase : 0x7fffffff, nvoff1 = lane < chunks ? nvbase + 16 : 0x7fffffff;
    auto prefetch = [&](int q) {                               // source rows 2q, 2q + 1 (clamped) -> registers
        const int r0 = min(max(2 * q, 0), sH - 1), r1 = min(max(2 * q + 1, 0), sH - 1);
        if constexpr (CHROMA) {
            if (nvsrc) {   // 32 bytes of interleaved pairs per lane and row -> one 16-byte chunk of each component (nvXXtoUV_c, p010LEToUV_c: input.c:926-1008)
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
                    pre[r] = u; pre[2 + r] = v;
                }
                return;
            }
        }
#pragma unroll
        for (int ci = 0; ci < NCOMP; ci++) {
            pre[2 * ci + 0] = bload16(rs[ci], voff0, r0 * sst[ci]);
            pre[2 * ci + 1] = bload16(rs[ci], voff0, r1 * sst[ci]);
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
            put(row0 + slot0, pre[2 * ci + 0]); put(row1 + slot0, pre[2 * ci + 1]);
        }
    };

    // ---- destination: 16-bit words per component, interleaved word pairs (p016 chroma), or the raw int32 sums ----
    const bool semi = CHROMA && p.dstKind == DSTK_P016;
    const bool raw = p.dstKind == DSTK_RAW32;
    const int dbytes = raw ? 4 : semi ? 4 : 2;
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

    // ---- fused planar RGB writer (CHROMA launch only; p.dstKind names the real destination: DSTK_GBRP16 / DSTK_GBRPF32): the luma launch left its int32 sums in a
    //      working plane (f.src[3], stride f.srcStride[3]); this launch holds the U and V sums of the same destination columns and rows (planar RGB forces full chroma:
    //      the chroma planes are scaled to the destination size), so it finishes yuv2gbrp16_full_X_c / yuv2gbrpf32_full_X_c itself (output.c:2467-2605, arithmetic as
    //      in sws_k_fullchr_gbrp16) and writes the three destination planes: the U / V sums never travel through memory ----
    const int fuse = !CHROMA ? 0 : p.dstKind == DSTK_GBRP16 ? 1 : p.dstKind == DSTK_GBRPF32 ? 2 : 0;
    sws_rsrc_t rdP[3], rsY = rd[0];
    int dstrP[3] = { 0, 0, 0 }, strY = 0, doffP[COLS], doffY[COLS];
#pragma unroll
    for (int k = 0; k < 3; k++) rdP[k] = rd[0];
#pragma unroll
    for (int c = 0; c < COLS; c++) { doffP[c] = 0x7fffffff; doffY[c] = 0x7fffffff; }
    const SwsLutParams &LUT = p.lut;
    const int y_offset = U(LUT.y_offset), y_coeff = U(LUT.y_coeff), v2r = U(LUT.v2r), v2g = U(LUT.v2g), u2g = U(LUT.u2g), u2b = U(LUT.u2b);
    if (fuse) {
        const int eb = fuse == 2 ? 4 : 2;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            uint8_t *db = k == 0 ? U(f.dst[0]) : k == 1 ? U(f.dst[1]) : U(f.dst[2]);
            dstrP[k] = k == 0 ? U(f.dstStride[0]) : k == 1 ? U(f.dstStride[1]) : U(f.dstStride[2]);
            rdP[k] = make_rsrc(db, (uint32_t)dstrP[k] * (uint32_t)(H - 1) + (uint32_t)W * (uint32_t)eb);
        }
        strY = U(f.srcStride[3]);
        rsY = make_rsrc(U(f.src[3]), (uint32_t)strY * (uint32_t)(H - 1) + (uint32_t)W * 4u);
#pragma unroll
        for (int c = 0; c < COLS; c++) {
            const int x = xs + 64 * c + lane;
            doffP[c] = x < W ? x * eb : 0x7fffffff; doffY[c] = x < W ? x * 4 : 0x7fffffff;
        }
    }
    uint32_t pend3[3][COLS];

    int ringA[NCOMP][COLS][RD], ringB[NCOMP][COLS][RD];        // rows 2q and 2q + 1 of the last RD pairs
#pragma unroll
    for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
        for (int c = 0; c < COLS; c++)
#pragma unroll
            for (int k = 0; k < RD; k++) { ringA[ci][c][k] = 0; ringB[ci][c][k] = 0; }

    uint32_t pend[NCOMP][COLS];
    int pend_y = -1;
    auto flush = [&]() {
        if (pend_y >= 0) {
            if (fuse) {
#pragma unroll
                for (int k = 0; k < 3; k++)
#pragma unroll
                    for (int c = 0; c < COLS; c++) {
                        if (fuse == 2) __builtin_amdgcn_raw_buffer_store_b32(pend3[k][c], rdP[k], doffP[c], pend_y * dstrP[k], 0);
                        else __builtin_amdgcn_raw_buffer_store_b16((uint16_t)pend3[k][c], rdP[k], doffP[c], pend_y * dstrP[k], 0);
                    }
            } else
            if (raw) {
#pragma unroll
                for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                    for (int c = 0; c < COLS; c++) __builtin_amdgcn_raw_buffer_store_b32(pend[ci][c], rd[ci], doff[c], pend_y * dstr[ci], 0);
            } else if (semi) {
#pragma unroll
                for (int c = 0; c < COLS; c++)
                    __builtin_amdgcn_raw_buffer_store_b32(pend[0][c] | (pend[NCOMP - 1][c] << 16), rd[0], doff[c], pend_y * dstr[0], 0);
            } else {
#pragma unroll
                for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                    for (int c = 0; c < COLS; c++) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)pend[ci][c], rd[ci], doff[c], pend_y * dstr[ci], 0);
            }
            pend_y = -1;
        }
    };

    // ---- march ----
    const SwsStripRow *rows = g.rows;
    StripRowN<RD> e = load_strip_row_n<RD>(rows, y0);
    int qnext = e.pf;
    prefetch(qnext);
    stage();
    prefetch(qnext + 1);
    for (int y = y0; y < y1; y++) {
        const StripRowN<RD> en = load_strip_row_n<RD>(rows, min(y + 1, H - 1));
        const int pfy = e.pf;
        int ysum[COLS];
#pragma unroll
        for (int c = 0; c < COLS; c++) ysum[c] = fuse ? __builtin_amdgcn_raw_buffer_load_b32(rsY, doffY[c], y * strY, 0) : 0;     // (the luma launch's sums of this row: in flight over the march)
        if (qnext < pfy) {                                     // rows nobody needs: skip
            qnext = pfy;
            prefetch(qnext);
            stage();
            prefetch(qnext + 1);
        }
        while (qnext <= pfy + npv - 1) {
            int na[NCOMP][COLS], nb[NCOMP][COLS];
            if (g.hfs2 < 0) {          // (never: keeps the horizontal stage in a basic block of its own, see strip_body)
#pragma unroll
                for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                    for (int c = 0; c < COLS; c++) { na[ci][c] = (int)L.S[(ci * 2) * L.row_dw + spd[c]]; nb[ci][c] = na[ci][c]; }
            } else {
#pragma unroll
                for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                    for (int c = 0; c < COLS; c++) {
                        const uint32_t *s0 = L.S + (ci * 2) * L.row_dw + spd[c], *s1 = s0 + L.row_dw;
                        int a = SRC16 ? hb[c] : 0, b = a;
#pragma unroll
                        for (int k = 0; k < NPH; k++) { a = sdot2(s0[k], ht[c][k], a); b = sdot2(s1[k], ht[c][k], b); }
                        na[ci][c] = min(a >> sh, hclip); nb[ci][c] = min(b >> sh, hclip);      // FFMIN(val >> sh, (1 << 19) - 1)
                    }
            }
            if (rng.on) {
#pragma unroll
                for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                    for (int c = 0; c < COLS; c++) { na[ci][c] = strip_range_wide(na[ci][c], rng); nb[ci][c] = strip_range_wide(nb[ci][c], rng); }
            }
#pragma unroll
            for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                for (int c = 0; c < COLS; c++) {
#pragma unroll
                    for (int k = 0; k < RD - 1; k++) { ringA[ci][c][k] = ringA[ci][c][k + 1]; ringB[ci][c][k] = ringB[ci][c][k + 1]; }
                    ringA[ci][c][RD - 1] = na[ci][c]; ringB[ci][c][RD - 1] = nb[ci][c];
                }
            qnext++;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            stage();
            flush();
            prefetch(qnext + 1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        flush();
        // ---- vertical stage over the whole ring (the host laid the row's tap pairs out against it) ----
        int acc[NCOMP][COLS];
#pragma unroll
        for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
            for (int c = 0; c < COLS; c++) acc[ci][c] = 0;
#pragma unroll
        for (int k = 0; k < RD; k++) {
            const int t0 = (int)(int16_t)(uint16_t)e.vt[k], t1 = (int)e.vt[k] >> 16;      // (scalar unit)
#pragma unroll
            for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                for (int c = 0; c < COLS; c++) acc[ci][c] = mad24(ringA[ci][c][k], t0, mad24(ringB[ci][c][k], t1, acc[ci][c]));
        }
        // ---- writers ----
        if (fuse) {
            if constexpr (NCOMP == 2) {
#pragma unroll
                for (int c = 0; c < COLS; c++) {
                    int Y = ((int)((unsigned)ysum[c] - 0x40000000u) >> 14) + 0x10000;
                    const int Uc = (int)((unsigned)acc[0][c] - (unsigned)(128 << 23)) >> 14, Vc = (int)((unsigned)acc[NCOMP - 1][c] - (unsigned)(128 << 23)) >> 14;
                    Y -= y_offset;
                    Y = (int)((unsigned)Y * (unsigned)y_coeff);
                    Y = (int)((unsigned)Y + (unsigned)((1 << 13) - (1 << 29)));
                    const int R = (int)((unsigned)Vc * (unsigned)v2r);
                    const int G = (int)((unsigned)Vc * (unsigned)v2g + (unsigned)Uc * (unsigned)u2g);
                    const int B = (int)((unsigned)Uc * (unsigned)u2b);
                    int r, g2, b;
                    if (fuse == 2) {
                        r = clip_uintp2(((int)((unsigned)Y + (unsigned)R) >> 14) + (1 << 15), 16);
                        g2 = clip_uintp2(((int)((unsigned)Y + (unsigned)G) >> 14) + (1 << 15), 16);
                        b = clip_uintp2(((int)((unsigned)Y + (unsigned)B) >> 14) + (1 << 15), 16);
                        const float float_mult = 1.0f / 65535.0f;
                        pend3[0][c] = __float_as_uint(__fmul_rn(float_mult, (float)g2)); pend3[1][c] = __float_as_uint(__fmul_rn(float_mult, (float)b));
                        pend3[2][c] = __float_as_uint(__fmul_rn(float_mult, (float)r));
                    } else {
                        r = clip_uintp2((int)(((int64_t)Y + R) >> 14) + (1 << 15), 16);
                        g2 = clip_uintp2((int)(((int64_t)Y + G) >> 14) + (1 << 15), 16);
                        b = clip_uintp2((int)(((int64_t)Y + B) >> 14) + (1 << 15), 16);
                        pend3[0][c] = (uint32_t)g2; pend3[1][c] = (uint32_t)b; pend3[2][c] = (uint32_t)r;
                    }
                }
            }
        } else
        if (raw) {
#pragma unroll
            for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                for (int c = 0; c < COLS; c++) pend[ci][c] = (uint32_t)acc[ci][c];
        } else {
#pragma unroll
            for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
                for (int c = 0; c < COLS; c++) {
                    const int val = (int)((uint32_t)acc[ci][c] + (uint32_t)((1 << 14) - 0x40000000));
                    pend[ci][c] = (uint32_t)(0x8000 + min(max(val >> 15, -32768), 32767));
                }
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

template <bool SRC16, bool CHROMA, int COLS, int RD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) sws_k_strip_wide(SwsFrameSet fs, SwsDevParams p, SwsStripGeom g)
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
#define SWS_SB(N) case N: strip_body_wide<SRC16, CHROMA, COLS, N, RD>(f, p, g, strip, y0, y1, smem, wib, lane); break;
    SWS_SB(1) SWS_SB(2) SWS_SB(3) SWS_SB(4) SWS_SB(5) SWS_SB(6) SWS_SB(7) SWS_SB(8)
#undef SWS_SB
    }
}

} // namespace swsk
