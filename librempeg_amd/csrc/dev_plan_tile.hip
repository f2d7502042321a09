// Planner stage 4 (PLAN_MAIN): the LDS-tile kernel's geometry, the packed writers' per-row forms (packed_vscale, vscale.c:135-157) and the plan of the
// same-size packed-RGB march (sws_k_rgb_march: yuv2rgb*_X over identity horizontal filters).
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

int plan_tile(PlanBuild &B)
{
    PLAN_LOCALS(B);
    // ---- fused h+v tile kernel geometry (planar / semi-planar YUV outputs, non-identity horizontal filters) ----
    d->tile_ok = false;
    if (!d->unity_h && !p.fast_bilinear && !gray_any && (p.dstKind == DSTK_PLANAR8 || p.dstKind == DSTK_PLANARN || p.dstKind == DSTK_PLANAR16 ||
                        p.dstKind == DSTK_NV12 || p.dstKind == DSTK_P010) && !c->tune.no_tile) {
        const size_t hsz = p.wide ? 4 : 2;
        auto plan = [&](const FilterBank &hb, const FilterBank &vb, int W, int H, int sW, int sH, int ncomp,
                        SwsTileGeom &g, std::vector<int32_t> &arr) -> bool {
            const int TW = 128;
            for (int TH : { 32, 16, 8, 4, 2, 1 }) {
                const int tX = (W + TW - 1) / TW, tY = (H + TH - 1) / TH;
                std::vector<int32_t> rs(tY), rc(tY), cs(tX), cc(tX);
                int nrmax = 0, ncmax = 0;
                for (int t = 0; t < tY; t++) {
                    int lo = INT32_MAX, hi = -1;
                    for (int y = t * TH; y < std::min(H, (t + 1) * TH); y++) {
                        lo = std::min(lo, vb.pos[y]); hi = std::max(hi, std::min(vb.pos[y] + vb.size - 1, sH - 1));
                    }
                    rs[t] = lo; rc[t] = hi - lo + 1; nrmax = std::max(nrmax, rc[t]);
                }
                for (int t = 0; t < tX; t++) {
                    int lo = INT32_MAX, hi = -1;
                    for (int x = t * TW; x < std::min(W, (t + 1) * TW); x++) {
                        lo = std::min(lo, hb.pos[x]); hi = std::max(hi, hb.pos[x] + hb.size - 1);
                    }
                    cs[t] = lo; cc[t] = hi - lo + 1; ncmax = std::max(ncmax, cc[t]);
                }
                const size_t lds = (((size_t)nrmax * ncmax * 2 + 15) & ~(size_t)15) + (size_t)ncomp * nrmax * TW * hsz;
                if (lds > 64 * 1024) continue;
                g.TW = TW; g.TH = TH; g.tilesX = tX; g.tilesY = tY; g.NRmax = nrmax; g.NCmax = ncmax; g.lds_bytes = (int32_t)lds;
                arr.clear();
                arr.insert(arr.end(), rs.begin(), rs.end()); arr.insert(arr.end(), rc.begin(), rc.end());
                arr.insert(arr.end(), cs.begin(), cs.end()); arr.insert(arr.end(), cc.begin(), cc.end());
                return true;
            }
            return false;
        };
        std::vector<int32_t> aL, aC;
        if (plan(hLumB, c->vLum, p.dstW, p.dstH, p.srcW, p.srcH, 1, d->tileL, aL) &&
            plan(hChrB, vChrB, p.chrDstW, p.chrDstH, p.chrSrcW, p.chrSrcH, 2, d->tileC, aC)) {
            const size_t bytes = (aL.size() + aC.size()) * sizeof(int32_t);
            { int r_ = table_alloc(c, d, &d->d_tilegeom, &d->tilegeom_bytes, bytes); if (r_ < 0) return r_; }
            std::vector<int32_t> all(aL); all.insert(all.end(), aC.begin(), aC.end());
            { int r_ = table_put(c, d, d->d_tilegeom, all.data(), bytes); if (r_ < 0) return r_; }
            const int32_t *bL = (const int32_t *)d->d_tilegeom, *bC = bL + aL.size();
            auto bind = [](SwsTileGeom &g, const int32_t *b) {
                g.rowStart = b; g.rowCount = b + g.tilesY; g.colStart = b + 2 * g.tilesY; g.colCount = b + 2 * g.tilesY + g.tilesX;
            };
            bind(d->tileL, bL); bind(d->tileC, bC);
            d->tile_ok = true;
        }
    }
    {   // packed_vscale picks yuv2packed1 / yuv2packed2 per row from (lfs, cfs, taps): vscale.c:135-157
        const int lfs = c->vLum.size, cfs = vChrB.size;
        bool all_x = !(lfs == 1 && cfs == 1);
        for (int y = 0; y < o.dst_h && all_x; y++) {
            const int cy = y >> c->chrDstVSubSample;
            const int16_t *lf = &c->vLum.taps[(size_t)y * lfs], *cf = &vChrB.taps[(size_t)cy * cfs];
            if (lfs == 1 && cfs == 2 && (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U) all_x = false;
            if (lfs == 2 && cfs == 2 && (uint16_t)lf[1] + (uint16_t)lf[0] == 4096 && (uint16_t)lf[1] <= 4096U &&
                (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U) all_x = false;
        }
        d->all_x_mode = all_x;
        d->striprgb_ok = d->striprgb_ok && (all_x || (lfs == 1 && cfs == 1) || striprgb_short);     // (striprgb_short: the plan's taps and rounding offsets carry the short forms)
        if (d->alpha_launch == 2 && !d->striprgb_ok) d->alpha_launch = 0;
        const bool kind_x = d->fullchr_kind == DSTK_GBRP || d->fullchr_kind == DSTK_PACKEDHI || d->fullchr_kind == DSTK_GBRP16 || d->fullchr_kind == DSTK_GBRPF32;   // (writers with the X form only)
        // (round 5: sws_k_fullchr_rgb tells the rows of the short forms by their taps and leaves the rounding constant out there -- yuv2rgb_full_2_c_template, the chroma
        //  blend of yuv2rgb_full_1_c_template; the one-launch RGB -> RGB kernel reads the same decision from its row entries)
        const bool short_full = !all_x && !c->tune.no_short_forms && (d->fullchr_on == 1 || d->fullchr_on == 2 || d->fullchr_on == 3) && (d->fullchr_kind == DSTK_RGB24 || d->fullchr_kind == DSTK_RGB32) &&
                                (lfs == 1 || lfs == 2) && cfs == 2;      // (3: sws_k_lut_rgb likewise, for yuv2rgb_2 rows)
        if (!all_x && !(lfs == 1 && cfs == 1) && !(short_full && rgb2rgb_short)) d->rgb2rgb_ok = false;
        if (d->fullchr_on && ((!all_x && !(lfs == 1 && cfs == 1) && !kind_x && !short_full) ||
                              (d->fullchr_on == 4 && lfs == 1 && cfs == 1 && d->fullchr_kind != DSTK_PACKEDHI && d->fullchr_kind != DSTK_GBRP16 && d->fullchr_kind != DSTK_GBRPF32) || !d->strip_ok)) {   // (planar RGB: any_vscale, always the X form)   // no strip plan, or a row in one of the short writer forms: the generic full-chroma writer keeps it
            d->fullchr_on = 0; d->strip_ok = false; d->rgbread_on = false; d->striprgbsrc_ok = false; d->rgb2rgb_ok = false;
            p.dstKind = d->fullchr_kind; p.u_plane_dst = dd->comp[1].plane; p.v_plane_dst = dd->comp[2].plane;
        }
        // a 4:4:4 planar source at the same size into a full-chroma destination: four identity filters, so the epilogue reads the source planes itself
        // (sws_k_fullchr_rgb / sws_k_fullchr_gbrp with SRCM 1 / 2) -- no strip launch, no working picture
        d->fullchr_direct = 0;
        if ((d->fullchr_on == 1 || d->fullchr_on == 2) && d->strip_ok && d->unity_h && d->unity_v && !d->rgbread_on && !d->split_mode && isPlanarYUV(o.src_format) &&
            ((p.srcKind == SRCK_PLANAR8 && c->srcBpc == 8) || (p.srcKind == SRCK_PLANAR16 && p.src_depth <= 15 && p.src_shift == 0)) &&
            p.chrSrcW == p.srcW && p.chrSrcH == p.srcH && c->vLum.size == 1 && vChrB.size == 1 && !c->tune.no_mixed)
            d->fullchr_direct = p.srcKind == SRCK_PLANAR8 ? 1 : 2;
        int win = 0;
        for (int y = 0; y < o.dst_h; y += 2) {
            const int c0 = y >> c->chrDstVSubSample, c1 = std::min(y + 1, o.dst_h - 1) >> c->chrDstVSubSample;
            const int lo = std::min(vChrB.pos[c0], vChrB.pos[c1]), hi = std::max(vChrB.pos[c0], vChrB.pos[c1]) + cfs - 1;
            win = std::max(win, hi - lo + 1);
        }
        d->chr_window2 = win;
        // plan of the marching packed-RGB kernel: identity vertical luma filter, chroma window of a row pair <= 8 rows,
        // ring advance of at most 1 row per step
        d->rgb_march_ok = false;
        bool lum_unity = lfs == 1;
        for (int y = 0; y < o.dst_h && lum_unity; y++) lum_unity = c->vLum.taps[y] == 4096;
        // (round 5: one luma tap with two chroma taps -- SWS_BILINEAR at the same size from 4:2:0: yuv2rgb_1_c_template with its chroma blend, which rounds with 128 << 11 and IS the
        //  X arithmetic on the bank's taps, output.c:1913-1937; its one-row form (u0 + 64) >> 7 is the X arithmetic over {4096, 0}.  Not with an alpha plane: other formulas there)
        const bool lut_rows_x = all_x || (lfs == 1 && cfs == 2 && !c->needAlpha && !c->tune.no_short_forms);
        if (lut_rows_x && lum_unity && win <= 8 && cfs <= 8) {
            const int groups = (o.dst_h + 1) / 2;
            std::vector<SwsRgbGroupPlan> plan((size_t)groups);
            bool ok = true;
            int prev = INT32_MIN, maxspan = 0;
            for (int g = 0; g < groups && ok; g++) {
                SwsRgbGroupPlan &e = plan[(size_t)g];
                std::memset(&e, 0, sizeof(e));
                int first[2], yy[2];
                for (int r = 0; r < 2; r++) {
                    yy[r] = std::min(2 * g + r, o.dst_h - 1);
                    first[r] = std::max(1 - cfs, vChrB.pos[yy[r] >> c->chrDstVSubSample]);
                }
                e.cbase = std::min(first[0], first[1]);
                if (prev != INT32_MIN && (e.cbase < prev || e.cbase - prev > 1)) ok = false;
                prev = e.cbase;
                e.ylum0 = std::min(std::max(c->vLum.pos[yy[0]], 0), o.src_h - 1);
                e.ylum1 = std::min(std::max(c->vLum.pos[yy[1]], 0), o.src_h - 1);
                for (int r = 0; r < 2; r++) {
                    const int16_t *cf = &vChrB.taps[(size_t)(yy[r] >> c->chrDstVSubSample) * cfs];
                    if (first[r] + cfs - 1 - e.cbase >= 8) ok = false;
                    maxspan = std::max(maxspan, first[r] + cfs - 1 - e.cbase);
                    for (int ip = 0; ip < 4; ip++) {
                        const int j0 = e.cbase + 2 * ip - first[r], j1 = j0 + 1;
                        const uint32_t lo = (j0 >= 0 && j0 < cfs) ? (uint16_t)cf[j0] : 0u, hi = (j1 >= 0 && j1 < cfs) ? (uint16_t)cf[j1] : 0u;
                        e.wp[r][ip] = lo | (hi << 16);
                    }
                }
            }
            // ring rows of the kernel instantiation: 5 (2x chroma up-sampling with 4 taps, the common case), 6 or 8.  With 5 rows the
            // sixth slot of the three v_dot2 pairs is free: it carries the rounding constant (sample 1 x tap 2048)
            d->rgb_ncr = maxspan <= 4 ? 5 : maxspan <= 5 ? 6 : 8;
            if (ok && d->rgb_ncr == 5)
                for (auto &e : plan)
                    for (int r = 0; r < 2; r++) e.wp[r][2] = (e.wp[r][2] & 0xFFFFu) | (2048u << 16);
            if (ok) {
                const size_t bytes = plan.size() * sizeof(SwsRgbGroupPlan);
                { int r_ = table_alloc(c, d, &d->d_rgbplan, &d->rgbplan_bytes, bytes); if (r_ < 0) return r_; }
                { int r_ = table_put(c, d, d->d_rgbplan, plan.data(), bytes); if (r_ < 0) return r_; }
                d->rgb_groups = groups;
                d->rgb_march_ok = true;
            }
        }
    }
    return 0;
}

} // namespace swship
