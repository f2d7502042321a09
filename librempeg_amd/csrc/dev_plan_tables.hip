// Planner stage 2 (PLAN_MAIN): the four filter banks as one device blob (initFilter's output as the kernels read it: libswscale/utils.c:197-612), the contexts whose
// result depends on the reference's line schedule (virtual lines: swscale.c:388-535), and the plans of the same-size RGB -> YUV kernels.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <map>
#include <mutex>
#include <thread>
#include <vector>
#include "generic_kinds.hpp"
#include "../../include/hwcontext_hip.h"
#include "dev_plan.hpp"

namespace swship {

int plan_tables(PlanBuild &B)
{
    PLAN_LOCALS(B);
    const FilterBank *banks[4] = { &hLumB, &hChrB, &c->vLum, &vChrB };
    size_t off = 0, offs_t[4], offs_p[4];
    auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
    for (int i = 0; i < 4; i++) {
        offs_t[i] = off; off = align(off + banks[i]->taps.size() * sizeof(int16_t));
        offs_p[i] = off; off = align(off + banks[i]->pos.size() * sizeof(int32_t));
    }
    { int r_ = table_alloc(c, d, &d->d_tables, &d->tables_bytes, off); if (r_ < 0) return r_; }
    std::vector<uint8_t> host(off, 0);
    for (int i = 0; i < 4; i++) {
        std::memcpy(host.data() + offs_t[i], banks[i]->taps.data(), banks[i]->taps.size() * sizeof(int16_t));
        std::memcpy(host.data() + offs_p[i], banks[i]->pos.data(), banks[i]->pos.size() * sizeof(int32_t));
    }
    { int r_ = table_put(c, d, d->d_tables, host.data(), off); if (r_ < 0) return r_; }
    uint8_t *b = (uint8_t *)d->d_tables;
    p.hLumF = (const int16_t *)(b + offs_t[0]); p.hLumPos = (const int32_t *)(b + offs_p[0]); p.hLumFs = hLumB.size;
    p.hChrF = (const int16_t *)(b + offs_t[1]); p.hChrPos = (const int32_t *)(b + offs_p[1]); p.hChrFs = hChrB.size;
    p.vLumF = (const int16_t *)(b + offs_t[2]); p.vLumPos = (const int32_t *)(b + offs_p[2]); p.vLumFs = c->vLum.size;
    p.vChrF = (const int16_t *)(b + offs_t[3]); p.vChrPos = (const int32_t *)(b + offs_p[3]); p.vChrFs = vChrB.size;
    // the fast-bilinear chroma function weighs with (xalpha ^ 127): not the identity even at equal widths
    d->unity_h = bank_is_identity(hLumB, 1 << 14) && bank_is_identity(hChrB, 1 << 14) && !p.fast_bilinear;
    d->unity_v = bank_is_identity(c->vLum, 1 << 12) && bank_is_identity(vChrB, 1 << 12);
    // ---- contexts whose result depends on the reference's line schedule (build_vlines): the two-pass path over virtual lines ----
    d->vlines_on = false; c->gamma_in_reader = false; d->mixed_ok = false;
    {
        const int drop = (o.flags & SWS_SRC_V_CHR_DROP_MASK) >> SWS_SRC_V_CHR_DROP_SHIFT;
        const int mode = c->internal_gamma ? 1 : (drop && isPlanarRGB(o.src_format) && !p.no_chroma) ? 2 : 0;
        VLines vlx;
        if (mode) build_vlines(c, mode, vlx);
        if (mode == 2 || (mode == 1 && !vlx.uniform)) {
            const size_t nL = vlx.lum.size() / 2, nC = vlx.chr.size() / 2;
            std::vector<int32_t> hostv;
            hostv.insert(hostv.end(), vlx.lum.begin(), vlx.lum.end());
            hostv.insert(hostv.end(), vlx.chr.begin(), vlx.chr.end());
            const size_t o_lp = hostv.size(); hostv.insert(hostv.end(), vlx.lumPos.begin(), vlx.lumPos.end());
            const size_t o_cp = hostv.size(); hostv.insert(hostv.end(), vlx.chrPos.begin(), vlx.chrPos.end());
            const size_t bytes = hostv.size() * sizeof(int32_t);
            { int r_ = table_alloc(c, d, &d->d_vlines, &d->vlines_bytes, bytes); if (r_ < 0) return r_; }
            { int r_ = table_put(c, d, d->d_vlines, hostv.data(), bytes); if (r_ < 0) return r_; }
            const int32_t *bv = (const int32_t *)d->d_vlines;
            p.vlines = bv; p.nVL = (int32_t)nL; p.vline_mode = mode;
            p.vLumPos = bv + o_lp; p.vChrPos = bv + o_cp;
            p.srcH = (int32_t)nL; p.chrSrcH = (int32_t)nC;   // what the two passes see: one scratch row per (destination row, tap)
            d->vlines_on = true; d->unity_h = false; d->unity_v = false;
            c->gamma_in_reader = mode == 1;
            log_msg(c, 2, "line-schedule dependent context (mode %d): %zu + %zu virtual lines\n", mode, nL, nC);
        }
    }
    // ---- packed 24 / 32 bpp RGB into 8-bit 4:2:0 / 4:2:2 YUV of the same size (sws_k_rgbsrc_unity): identity horizontal filters and luma
    //      vertical filter, chroma of the "half" readers through a vertical filter of up to 16 taps whose positions only move forward ----
    d->rgbsrc_ok = false; d->rgbsrc2_rows = nullptr;
    if (d->unity_h && !d->vlines_on && (p.srcKind == SRCK_RGB24 || p.srcKind == SRCK_RGB32 || p.srcKind == SRCK_GBRP) && p.chr_half && (!p.range_active || (!(p.dstW & 3) && p.range_to_jpeg && !c->tune.no_strip_range && !c->tune.no_rgbsrc2)) && !p.need_alpha && !p.no_chroma &&
        !p.wide && !p.dst_alpha_fill && (p.dstKind == DSTK_PLANAR8 || p.dstKind == DSTK_NV12) && p.dst_bits == 8 && p.chrDstHSub == 1 && p.chrDstVSub <= 1 &&
        p.chrSrcVSub == 0 && p.srcH == p.dstH && bank_is_identity(c->vLum, 1 << 12) && vChrB.size <= 16 && p.chrDstW == ((p.dstW + 1) >> 1) &&
        !p.should_dither && !c->tune.no_rgbsrc) {
        bool fwd = true;
        for (int k = 0; k < 9; k++) fwd = fwd && p.rgb2yuv[k] > -32768 && p.rgb2yuv[k] < 32768;   // (v_dot2_i32_i16 operands)
        for (int y = 0; y < vChrB.count && fwd; y++) fwd = vChrB.pos[y] >= 0 && (y == 0 || vChrB.pos[y] >= vChrB.pos[y - 1]);
        if (fwd) {
            const bool x_form = p.dstKind == DSTK_NV12 || vChrB.size > 1;   // yuv2nv12cX_c has no one-tap form
            std::vector<SwsRgbSrcRow> rows((size_t)vChrB.count);
            for (int y = 0; y < vChrB.count; y++) {
                SwsRgbSrcRow &e = rows[(size_t)y];
                std::memset(&e, 0, sizeof(e));
                e.first = std::max(1 - vChrB.size, vChrB.pos[y]);
                e.last = e.first + vChrB.size - 1;
                for (int j = 0; j < vChrB.size; j++)
                    e.vt[j >> 1] |= (uint32_t)(uint16_t)(x_form ? vChrB.taps[(size_t)y * vChrB.size + j] : 1) << (16 * (j & 1));
            }
            // the wave-march form (sws_k_rgbsrc_unity2): per chroma row the first source-row PAIR and the tap pairs aligned to even source rows, laid out
            // against the newest slots of its register ring (1 / 3 / 5 / 8 pairs); the planar one-tap form enters as the tap 4096
            std::vector<SwsStripRow> rows2;
            int npv2 = 1;
            for (int y = 0; y < vChrB.count; y++) npv2 = std::max(npv2, ((vChrB.pos[y] & 1) + vChrB.size + 1) / 2);
            const int rd2 = npv2 <= 1 ? 1 : npv2 <= 3 ? 3 : npv2 <= 5 ? 5 : 8;
            if (npv2 <= 8) {
                rows2.resize((size_t)vChrB.count);
                std::memset(rows2.data(), 0, rows2.size() * sizeof(SwsStripRow));
                for (int y = 0; y < vChrB.count; y++) {
                    SwsStripRow &e2 = rows2[(size_t)y];
                    e2.pf = (vChrB.pos[y] & ~1) >> 1;
                    for (int j = 0; j < vChrB.size; j++) {
                        const int k = (vChrB.pos[y] & 1) + j + 2 * (rd2 - npv2);
                        const int16_t tap = x_form ? vChrB.taps[(size_t)y * vChrB.size + j] : (int16_t)4096;
                        e2.vt[k >> 1] |= (uint32_t)(uint16_t)tap << (16 * (k & 1));
                    }
                }
            }
            const size_t bytes1 = (rows.size() * sizeof(SwsRgbSrcRow) + 63) & ~(size_t)63;
            const size_t bytes = bytes1 + rows2.size() * sizeof(SwsStripRow);
            { int r_ = table_alloc(c, d, &d->d_dot2, &d->dot2_bytes, bytes); if (r_ < 0) return r_; }
            { int r_ = table_put(c, d, d->d_dot2, rows.data(), rows.size() * sizeof(SwsRgbSrcRow)); if (r_ < 0) return r_; }
            if (!rows2.empty()) { int r_ = table_put(c, d, (uint8_t *)d->d_dot2 + bytes1, rows2.data(), rows2.size() * sizeof(SwsStripRow)); if (r_ < 0) return r_; }
            d->rgbsrc_rows = (const SwsRgbSrcRow *)d->d_dot2;
            d->rgbsrc2_rows = rows2.empty() ? nullptr : (const SwsStripRow *)((const uint8_t *)d->d_dot2 + bytes1);
            d->rgbsrc2_npv = npv2;
            d->rgbsrc_ok = true;
        }
    }
    // ---- 8-bit RGB (packed 24 / 32 bpp, planar) into planar 8-bit 4:4:4 YUV of the same size: all four banks the identity, full chroma readers ----
    d->rgb444_ok = false;
    if (d->unity_h && d->unity_v && !d->vlines_on && (p.srcKind == SRCK_RGB24 || p.srcKind == SRCK_RGB32 || p.srcKind == SRCK_GBRP) && !p.chr_half && !p.range_active &&
        !p.need_alpha && !p.no_chroma && !p.wide && !p.dst_alpha_fill && p.dstKind == DSTK_PLANAR8 && p.dst_bits == 8 && p.chrDstHSub == 0 && p.chrDstVSub == 0 &&
        p.chrSrcVSub == 0 && !p.should_dither && !c->tune.no_rgbsrc) {
        bool fits = true;
        for (int k = 0; k < 9; k++) fits = fits && p.rgb2yuv[k] > -32768 && p.rgb2yuv[k] < 32768;   // (v_dot2_i32_i16 operands)
        d->rgb444_ok = fits;
    }
    return 0;
}

} // namespace swship
