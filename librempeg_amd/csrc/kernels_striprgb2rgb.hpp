// Marching strip kernel for SCALED packed 24 / 32 bpp RGB -> packed 24 / 32 bpp RGB (screen capture / render resize, thumbnails of RGB pictures):
// the reference's whole chain in one wave -- the readers (rgb24ToY_c / rgb24ToUV_c / rgb24ToUV_half_c, the 32-bit rows of rgb16_32To*_c_template
// and rgbaToA_c / abgrToA_c: input.c:264-393, :454-472, :1068-1172), hScale16To15_c for Y, U, V and A (swscale.c:106-131), the vertical filters and
// yuv2rgb_full_X_c_template + yuv2rgb_write_full (output.c:2005-2070, :2163-2207; an RGB destination of an RGB source always takes the full-chroma
// writers, utils.c:1270-1286).
//
// What it replaces: reader pre-pass (16-bit Y / U / V / A planes of the source size) -> strip launches for luma, chroma and alpha storing int32 sums ->
// sws_k_fullchr_rgb: 207 MB of traffic for the 41 MB a bgra 4K -> 1080p frame needs.  Here nothing between the source pixels and the destination pixels
// leaves the wave.
//
// Schedule (kernels_striprgbsrc.hpp describes the machine this extends): a wave owns a strip of 128 destination columns, lane l the columns l and
// l + 64 -- of ALL components, so the matrix at the end needs no exchange between lanes.  The vertical filters of luma and chroma are the same bank (both
// plane classes have the source's and the destination's heights; the host checks it), so luma {Y, A} and chroma {U, V} march in lockstep over the source
// row pairs and every output row leaves when its last pair has entered the rings.  Reader: groups of four pixels per lane and turn; chroma either the
// pair sums of the half readers (destinations at most half as wide as the source) or per-pixel values; alpha bytes as the 14-bit samples a << 6 | a >> 2.
#pragma once
#include "kernels_striprgbsrc.hpp"

namespace swsk {

// four pixels of one source row -> Y pairs (2 dwords), A pairs (2 dwords), U / V (HALF: one dword each in uo[0] / vo[0]; else two)
template <int BPP, bool HALF, bool ALPHA>
__device__ __forceinline__ void rgb4px_read_full(const uint32_t (&d)[4], const RgbReadCoefs &k, int kcf, uint32_t a_sel, uint32_t a_or,
                                                 u32x2 &yo, u32x2 &ao, u32x2 &uo, u32x2 &vo)
{
    uint32_t lo[4], hi[4];          // per pixel: {byte 0, byte 2} and {byte 1, byte 3 (0 for 24 bpp)} as 16-bit halves
    if constexpr (BPP == 4) {
#pragma unroll
        for (int i = 0; i < 4; i++) { lo[i] = d[i] & 0x00FF00FFu; hi[i] = __builtin_amdgcn_perm(0, d[i], 0x0c030c01u); }
    } else {
        lo[0] = __builtin_amdgcn_perm(d[0], d[0], 0x0c020c00u); hi[0] = __builtin_amdgcn_perm(d[0], d[0], 0x0c0c0c01u);
        lo[1] = __builtin_amdgcn_perm(d[1], d[0], 0x0c050c03u); hi[1] = __builtin_amdgcn_perm(d[1], d[0], 0x0c0c0c04u);
        lo[2] = __builtin_amdgcn_perm(d[2], d[1], 0x0c040c02u); hi[2] = __builtin_amdgcn_perm(d[2], d[1], 0x0c0c0c03u);
        lo[3] = __builtin_amdgcn_perm(d[2], d[2], 0x0c030c01u); hi[3] = __builtin_amdgcn_perm(d[2], d[2], 0x0c0c0c02u);
    }
    uint32_t yv[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int S = sdot2(lo[i], k.yA, sdot2(hi[i], k.yB, k.ky));
        if constexpr (BPP != 4) yv[i] = (uint32_t)S >> 9;
        else yv[i] = __builtin_amdgcn_ubfe((uint32_t)S, 9, 15);
    }
    yo[0] = __builtin_amdgcn_perm(yv[1], yv[0], 0x05040100u); yo[1] = __builtin_amdgcn_perm(yv[3], yv[2], 0x05040100u);
    if constexpr (HALF) {
        uint32_t uu[2], vv[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const uint32_t al = lo[2 * j] + lo[2 * j + 1], ah = hi[2 * j] + hi[2 * j + 1];
            const int Su = sdot2(al, k.uA, sdot2(ah, k.uB, k.kc)), Sv = sdot2(al, k.vA, sdot2(ah, k.vB, k.kc));
            if constexpr (BPP != 4) { uu[j] = (uint32_t)Su >> 10; vv[j] = (uint32_t)Sv >> 10; }
            else { uu[j] = __builtin_amdgcn_ubfe((uint32_t)Su, 10, 14); vv[j] = __builtin_amdgcn_ubfe((uint32_t)Sv, 10, 14); }
        }
        uo[0] = __builtin_amdgcn_perm(uu[1], uu[0], 0x05040100u); vo[0] = __builtin_amdgcn_perm(vv[1], vv[0], 0x05040100u);
        uo[1] = vo[1] = 0;
    } else {   // rgb24ToUV_c and the full-width 32-bit rows: (S + (256 << 14) + (1 << 8)) >> 9 per pixel (kcf)
        uint32_t uu[4], vv[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int Su = sdot2(lo[i], k.uA, sdot2(hi[i], k.uB, kcf)), Sv = sdot2(lo[i], k.vA, sdot2(hi[i], k.vB, kcf));
            if constexpr (BPP != 4) { uu[i] = (uint32_t)Su >> 9; vv[i] = (uint32_t)Sv >> 9; }
            else { uu[i] = __builtin_amdgcn_ubfe((uint32_t)Su, 9, 15); vv[i] = __builtin_amdgcn_ubfe((uint32_t)Sv, 9, 15); }
        }
        uo[0] = __builtin_amdgcn_perm(uu[1], uu[0], 0x05040100u); uo[1] = __builtin_amdgcn_perm(uu[3], uu[2], 0x05040100u);
        vo[0] = __builtin_amdgcn_perm(vv[1], vv[0], 0x05040100u); vo[1] = __builtin_amdgcn_perm(vv[3], vv[2], 0x05040100u);
    }
    if constexpr (ALPHA) {          // rgbaToA_c / abgrToA_c: a << 6 | a >> 2 (an rgb0-style source feeding a real alpha channel counts as 255: a_or)
        uint32_t av[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { const uint32_t a = __builtin_amdgcn_perm(0, d[i], a_sel) | a_or; av[i] = (a << 6) | (a >> 2); }
        ao[0] = av[0] | av[1] << 16; ao[1] = av[2] | av[3] << 16;
    }
}

template <int BPPS, int BPPD, bool HALF, bool ALPHA, int NPH, int RD, int NG>
__device__ __forceinline__ void strip_rgb2rgb_body(const FrameRegs &f, const SwsDevParams &p, const SwsStripGeom &gl, const SwsStripGeom &gc,
                                                   int strip, int y0, int y1, uint32_t *lds, int npx_max, int lane)
{
    constexpr int CL = 2;                                       // columns per lane (luma and chroma alike)
    constexpr int NL = ALPHA ? 2 : 1;                           // luma-class components: Y (, A)
    const int W = p.dstW, H = p.dstH, sH = p.srcH, sh = p.hshift;
    // the strip's pixel window: luma window and the chroma window (in pixels: twice the sample window of the half readers), in groups of four pixels
    const int csL = gl.colStart[strip], eL = csL + gl.colCount[strip], csC = gc.colStart[strip], eC = csC + gc.colCount[strip];
    const int w0 = min(csL, HALF ? 2 * csC : csC) & ~15;
    const int n4 = (max(eL, HALF ? 2 * eC : eC) - w0 + 3) >> 2;
    const int ng = __builtin_amdgcn_readfirstlane((n4 + 63) >> 6);
    StripLds LL, LC;
    LL.row_dw = (npx_max + 16) >> 1; LC.row_dw = HALF ? (npx_max + 16) >> 2 : (npx_max + 16) >> 1;
    LL.S = lds; LC.S = lds + NL * 2 * LL.row_dw;

    int spdL[CL], spdC[CL];
    uint32_t htL[CL][NPH], htC[CL][NPH];
    {
        const int ndL = gl.hfs2 >> 1, ndC = gc.hfs2 >> 1;
#pragma unroll
        for (int c = 0; c < CL; c++) {
            const int x = min(min(strip * gl.TW + 64 * c + lane, (strip + 1) * gl.TW - 1), W - 1);     // (gl.TW <= 128: the host narrows the strips where that saves a reader turn)
            spdL[c] = ((p.hLumPos[x] & ~1) - w0) >> 1;
            spdC[c] = ((p.hChrPos[x] & ~1) - (HALF ? w0 >> 1 : w0)) >> 1;
            const uint32_t *tl = (const uint32_t *)(gl.hT2 + (int64_t)x * gl.hfs2), *tc = (const uint32_t *)(gc.hT2 + (int64_t)x * gc.hfs2);
#pragma unroll
            for (int k = 0; k < NPH; k++) { htL[c][k] = k < ndL ? tl[k] : 0u; htC[c][k] = k < ndC ? tc[k] : 0u; }
        }
    }
    RgbReadCoefs rk;
    {
        const Rgb2YuvRow ty = rgb2yuv_row(p.rgb2yuv, 0), tu = rgb2yuv_row(p.rgb2yuv, 3), tv = rgb2yuv_row(p.rgb2yuv, 6);
        const int rp = U(p.src_r_pos), gp = BPPS == 4 ? U(p.src_g_pos) : 1, bp = U(p.src_b_pos);
        auto coef = [&](const Rgb2YuvRow &w, int k) { return (uint32_t)(uint16_t)(k == rp ? w.r : k == gp ? w.g : k == bp ? w.b : 0); };
        rk.yA = coef(ty, 0) | coef(ty, 2) << 16; rk.yB = coef(ty, 1) | coef(ty, 3) << 16;
        rk.uA = coef(tu, 0) | coef(tu, 2) << 16; rk.uB = coef(tu, 1) | coef(tu, 3) << 16;
        rk.vA = coef(tv, 0) | coef(tv, 2) << 16; rk.vB = coef(tv, 1) | coef(tv, 3) << 16;
        rk.ky = (32 << 14) + (1 << 8); rk.kc = (256 << 15) + (1 << 9);
    }
    const int kcf = (256 << 14) + (1 << 8);
    const uint32_t a_sel = 0x0c0c0c00u | (uint32_t)(U(p.src_a_pos) & 3), a_or = U(p.src_alpha_opaque) ? 0xFFu : 0u;
    const int sst = f.srcStride[0];
    const sws_rsrc_t rs = make_rsrc(f.src[0], (uint32_t)sst * (uint32_t)sH);
    const int vbase = (w0 + 4 * lane) * BPPS;

    uint32_t pre[2][NG][BPPS];      // [row of the pair][turn][dwords of a group]
    auto prefetch = [&](int q) {
        const int r0 = min(max(2 * q, 0), sH - 1), r1 = min(max(2 * q + 1, 0), sH - 1);
#pragma unroll
        for (int j = 0; j < NG; j++)
            if (j < ng) {
                const int vo = lane + 64 * j < n4 ? vbase + j * (256 * BPPS) : 0x7fffffff;   // (beyond the window: the descriptor answers 0 without touching memory)
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const int ro = (r ? r1 : r0) * sst;
                    if constexpr (BPPS == 4) {
                        const u32x4 t = bload16(rs, vo, ro);
                        pre[r][j][0] = t[0]; pre[r][j][1] = t[1]; pre[r][j][2] = t[2]; pre[r][j][3] = t[3];
                    } else {
                        const rsrc_u32x3 t = __builtin_bit_cast(rsrc_u32x3, __builtin_amdgcn_raw_buffer_load_b96(rs, vo, ro, 0));
                        pre[r][j][0] = t[0]; pre[r][j][1] = t[1]; pre[r][j][2] = t[2];
                    }
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
                    uint32_t dd[4] = { pre[r][j][0], pre[r][j][1], pre[r][j][2], BPPS == 4 ? pre[r][j][BPPS - 1] : 0u };
                    u32x2 yo, ao, uo, vo;
                    rgb4px_read_full<BPPS, HALF, ALPHA>(dd, rk, kcf, a_sel, a_or, yo, ao, uo, vo);
                    *(u32x2 *)(LL.S + r * LL.row_dw + 2 * gs) = yo;
                    if constexpr (ALPHA) *(u32x2 *)(LL.S + (2 + r) * LL.row_dw + 2 * gs) = ao;
                    if constexpr (HALF) { LC.S[r * LC.row_dw + gs] = uo[0]; LC.S[(2 + r) * LC.row_dw + gs] = vo[0]; }
                    else { *(u32x2 *)(LC.S + r * LC.row_dw + 2 * gs) = uo; *(u32x2 *)(LC.S + (2 + r) * LC.row_dw + 2 * gs) = vo; }
                }
            }
    };

    // destination: pixel x of a row at byte x * BPPD
    const sws_rsrc_t rd = make_rsrc(f.dst[0], (uint32_t)f.dstStride[0] * (uint32_t)(H - 1) + (uint32_t)W * (uint32_t)BPPD);
    const int dstr = f.dstStride[0];
    int doff[CL];
#pragma unroll
    for (int c = 0; c < CL; c++) { const int x = strip * gl.TW + 64 * c + lane; doff[c] = (x < W && 64 * c + lane < gl.TW) ? x * BPPD : 0x7fffffff; }
    uint32_t pend[CL];
    int pend_y = -1;
    auto flush = [&]() {
        if (pend_y >= 0) {
            const int ro = pend_y * dstr;
#pragma unroll
            for (int c = 0; c < CL; c++) {
                if constexpr (BPPD == 4) __builtin_amdgcn_raw_buffer_store_b32(pend[c], rd, doff[c], ro, 0);
                else {
                    __builtin_amdgcn_raw_buffer_store_b16((uint16_t)pend[c], rd, doff[c], ro, 0);
                    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(pend[c] >> 16), rd, doff[c] + 2, ro, 0);
                }
            }
            pend_y = -1;
        }
    };

    uint32_t ringL[NL][CL][RD], ringC[2][CL][RD];
#pragma unroll
    for (int c = 0; c < CL; c++)
#pragma unroll
        for (int k = 0; k < RD; k++) {
#pragma unroll
            for (int ci = 0; ci < NL; ci++) ringL[ci][c][k] = 0;
            ringC[0][c][k] = 0; ringC[1][c][k] = 0;
        }

    const SwsLutParams &L = p.lut;
    const int y_offset = U(L.y_offset), y_coeff = U(L.y_coeff), v2r = U(L.v2r), v2g = U(L.v2g), u2g = U(L.u2g), u2b = U(L.u2b);
    const int r_pos = U(L.r_pos), g_pos = U(L.g_pos), b_pos = U(L.b_pos), a_pos = U(L.a_pos);

    const SwsStripRow *rowsL = gl.rows, *rowsC = gc.rows;
    const int npv = gl.npv;                                      // (== gc.npv, and every row's first pair is the same for both: host check)
    StripRowN<RD> el = load_strip_row_n<RD>(rowsL, y0), ec = load_strip_row_n<RD>(rowsC, y0);
    // (the writers' rounding constant of a row: 1 << 9 minus what the host wrote for the rows of yuv2rgb_full_2_c_template / the chroma blend of yuv2rgb_full_1_c_template,
    //  which have none -- SwsStripRow::rnd_off, dev_plan*.hip; round 5)
    auto row_rnd = [&](int yy) -> unsigned {
        typedef const uint32_t __attribute__((address_space(4))) *cptr;
        cptr q = (cptr)(uintptr_t)(rowsL + (RD > 24 ? 4 : RD > 12 ? 2 : 1) * yy);
        return (1u << 9) - q[1];
    };
    unsigned rq = row_rnd(y0);
    int qnext = el.pf;
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
    for (int y = y0; y < y1; y++) {
        const int yn = min(y + 1, H - 1);
        const StripRowN<RD> eln = load_strip_row_n<RD>(rowsL, yn), ecn = load_strip_row_n<RD>(rowsC, yn);     // next row's scalars, one row ahead
        const unsigned rqn = row_rnd(yn);
        const int need = el.pf + npv - 1;
        if (qnext < el.pf) { qnext = el.pf; refill(); }          // pairs nobody needs (steep down-scaling with short filters)
        while (qnext <= need) {
            uint32_t npL[NL][CL], npC[2][CL];
            // (a basic block of its own: see strip_body -- straight-line code makes hipcc interleave the stage with the rings until the loop spills)
            if (gl.hfs2 < 0) {
#pragma unroll
                for (int c = 0; c < CL; c++) {
#pragma unroll
                    for (int ci = 0; ci < NL; ci++) npL[ci][c] = LL.S[(ci * 2) * LL.row_dw + spdL[c]];
                    npC[0][c] = LC.S[spdC[c]]; npC[1][c] = LC.S[2 * LC.row_dw + spdC[c]];
                }
            } else {
                strip_hstage<NPH, NL, CL>(LL, spdL, htL, sh, npL);
                strip_hstage<NPH, 2, CL>(LC, spdC, htC, sh, npC);
            }
#pragma unroll
            for (int c = 0; c < CL; c++) {
#pragma unroll
                for (int k = 0; k < RD - 1; k++) {
#pragma unroll
                    for (int ci = 0; ci < NL; ci++) ringL[ci][c][k] = ringL[ci][c][k + 1];
                    ringC[0][c][k] = ringC[0][c][k + 1]; ringC[1][c][k] = ringC[1][c][k + 1];
                }
#pragma unroll
                for (int ci = 0; ci < NL; ci++) ringL[ci][c][RD - 1] = npL[ci][c];
                ringC[0][c][RD - 1] = npC[0][c]; ringC[1][c][RD - 1] = npC[1][c];
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
        int aL[NL][CL], aC[2][CL];
        strip_vstage_n<NL, CL, RD>(ringL, el, npv, aL);
        strip_vstage_n<2, CL, RD>(ringC, ec, npv, aC);
        // yuv2rgb_full_X_c_template's tail on the sums (kernels_stream.hpp sws_k_fullchr_rgb has the same arithmetic on the int32 planes of the two-pass form)
#pragma unroll
        for (int c = 0; c < CL; c++) {
            int Y = (int)((unsigned)aL[0][c] + rq) >> 10;
            const int Uc = (int)((unsigned)aC[0][c] + rq - (unsigned)(128 << 19)) >> 10, Vc = (int)((unsigned)aC[1][c] + rq - (unsigned)(128 << 19)) >> 10;
            Y -= y_offset;
            Y = (int)((unsigned)Y * (unsigned)y_coeff);
            Y = (int)((unsigned)Y + (1u << 21));
            int R = (int)((unsigned)Y + (unsigned)Vc * (unsigned)v2r);
            int G = (int)((unsigned)Y + (unsigned)Vc * (unsigned)v2g + (unsigned)Uc * (unsigned)u2g);
            int B = (int)((unsigned)Y + (unsigned)Uc * (unsigned)u2b);
            if ((R | G | B) & 0xC0000000) { R = clip_uintp2(R, 30); G = clip_uintp2(G, 30); B = clip_uintp2(B, 30); }
            uint32_t px = ((uint32_t)(R >> 22) << (8 * r_pos)) | ((uint32_t)(G >> 22) << (8 * g_pos)) | ((uint32_t)(B >> 22) << (8 * b_pos));
            if constexpr (BPPD == 4) {
                int A = 255;
                if constexpr (ALPHA) { A = (int)((unsigned)aL[NL - 1][c] + (1u << 18)) >> 19; if (A & 0x100) A = clip_u8(A); }
                px |= ((uint32_t)A & 0xFFu) << (8 * a_pos);
            }
            pend[c] = px;
        }
        pend_y = y;
        el = eln; ec = ecn; rq = rqn;
    }
    flush();
}

#ifndef R2R_ATTR
#define R2R_ATTR
#endif
template <int BPPS, int BPPD, bool HALF, bool ALPHA, int NPH, int RD, int NG>
__global__ void __launch_bounds__(256) R2R_ATTR sws_k_strip_rgb2rgb(SwsFrameSet fs, SwsDevParams p, SwsStripGeom gl, SwsStripGeom gc, int npx_max, int wave_lds_dw)
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
    strip_rgb2rgb_body<BPPS, BPPD, HALF, ALPHA, NPH, RD, NG>(f, p, gl, gc, strip, y0, y1, (uint32_t *)smem + wib * wave_lds_dw, npx_max, lane);
}

} // namespace swsk
