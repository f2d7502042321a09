// Translation unit of the marching strip kernel (C3b and every wide h+v polyphase conversion to planar / semi-planar YUV).
#include <algorithm>

#include "devstate.hpp"
#include "kernels_strip.hpp"

namespace swship {

// which: 1 = the luma launch, 2 = the chroma launch, 3 = both
int launch_strip_planes(const LaunchCtx &L, int which)
{
    SwsInternal *c = L.c; DeviceState *d = L.d; const SwsDevParams &p = *L.p; hipStream_t st = L.st; const SwsFrameSet &fs = L.fs;
    const SwsFramePtrs *frames = L.frames; const int n = L.n, sliceY = L.sliceY, sliceH = L.sliceH; const bool vec = L.vec;
    const dim3 blk(256);
    (void)c; (void)d; (void)frames; (void)vec; (void)sliceY; (void)sliceH;
    if (p.wide) return launch_strip_wide(L, which);     // 19-bit intermediates: k_stripwide.hip
            const int target = c->tune.strip_waves;
            const bool s16 = p.srcKind == SRCK_PLANAR16 || p.srcKind == SRCK_P010;   // (launch_rgbread_strip passes its reader planes as SRCK_PLANAR16)
            auto launch = [&](SwsStripGeom g, int H, bool chroma) {
                // one resident round of waves; bands of at least `minrows` output rows.  The floor matters for ONE frame per call (what a filter
                // chain issues): the kernel's time is then the serial walk of a wave (ring fill + one step per source row pair), not throughput,
                // and the machine is otherwise idle -- 4-row bands: a 1080p plane is 2000 waves of 7 steps instead of 540 waves of 19
                const int minrows = std::max(1, c->tune.strip_min_rows);
                int bands = std::max(1, std::min(target / std::max(1, g.strips * n), (H + minrows - 1) / minrows));
                g.debug = c->tune.debug;
                g.band_rows = (H + bands - 1) / bands;
                g.bands = (H + g.band_rows - 1) / g.band_rows;
                const dim3 grid(cdiv((int64_t)g.strips * g.bands, 4), 1, n);
#define SWS_STRIP(S, C, K) hipLaunchKernelGGL((swsk::sws_k_strip_march<S, C, K>), grid, blk, g.lds_bytes, st, fs, p, g)
#define SWS_STRIP_DMA(C, K) hipLaunchKernelGGL((swsk::sws_k_strip_dma<C, K>), grid, blk, g.lds_dma_bytes, st, fs, p, g)
                const int cols = g.TW / 64;
                if (g.nph > 16 || g.npv > (chroma ? 24 : 16)) {   // filters of 33 .. 62 taps (ratios of 8:1 and more): 32 tap pairs each way, strips of 64 columns
                    if (cols != 1) { log_msg(c, 0, "internal error: extra-long-filter strip plan with %d columns per lane\n", cols); return; }
                    if (chroma) { if (s16) hipLaunchKernelGGL((swsk::sws_k_strip_xlong<true, true>), grid, blk, g.lds_bytes, st, fs, p, g);
                                  else hipLaunchKernelGGL((swsk::sws_k_strip_xlong<false, true>), grid, blk, g.lds_bytes, st, fs, p, g); }
                    else        { if (s16) hipLaunchKernelGGL((swsk::sws_k_strip_xlong<true, false>), grid, blk, g.lds_bytes, st, fs, p, g);
                                  else hipLaunchKernelGGL((swsk::sws_k_strip_xlong<false, false>), grid, blk, g.lds_bytes, st, fs, p, g); }
                    return;
                }
                if (g.nph > 8 || g.npv > (chroma ? 12 : 8)) {   // long filters (ratios of 4:1 and more): 16 tap pairs each way, strips of 128 / 64 columns
                    if (cols != (chroma ? 1 : 2)) { log_msg(c, 0, "internal error: long-filter strip plan with %d columns per lane\n", cols); return; }
                    if (chroma) { if (s16) hipLaunchKernelGGL((swsk::sws_k_strip_long<true, true>), grid, blk, g.lds_bytes, st, fs, p, g);
                                  else hipLaunchKernelGGL((swsk::sws_k_strip_long<false, true>), grid, blk, g.lds_bytes, st, fs, p, g); }
                    else        { if (s16) hipLaunchKernelGGL((swsk::sws_k_strip_long<true, false>), grid, blk, g.lds_bytes, st, fs, p, g);
                                  else hipLaunchKernelGGL((swsk::sws_k_strip_long<false, false>), grid, blk, g.lds_bytes, st, fs, p, g); }
                    return;
                }
                if (chroma && g.npv > 8) {   // long vertical chroma filters (4:1 steps): the instantiations with a ring of 12 row pairs
                    if (s16 && g.dma_ok && !c->tune.no_strip_dma) {
                        if (cols == 1) hipLaunchKernelGGL((swsk::sws_k_strip_dma<true, 1, 12>), grid, blk, g.lds_dma_bytes, st, fs, p, g);
                        else hipLaunchKernelGGL((swsk::sws_k_strip_dma<true, 2, 12>), grid, blk, g.lds_dma_bytes, st, fs, p, g);
                    } else if (s16) {
                        if (cols == 1) hipLaunchKernelGGL((swsk::sws_k_strip_march<true, true, 1, 12>), grid, blk, g.lds_bytes, st, fs, p, g);
                        else hipLaunchKernelGGL((swsk::sws_k_strip_march<true, true, 2, 12>), grid, blk, g.lds_bytes, st, fs, p, g);
                    } else {
                        if (cols == 1) hipLaunchKernelGGL((swsk::sws_k_strip_march<false, true, 1, 12>), grid, blk, g.lds_bytes, st, fs, p, g);
                        else hipLaunchKernelGGL((swsk::sws_k_strip_march<false, true, 2, 12>), grid, blk, g.lds_bytes, st, fs, p, g);
                    }
                    return;
                }
                // experiments (sws_hip_set_option "exp0" = ring depth D in row pairs (2 / 3; 0 = the shipped 4), "exp1" = absolute register-ring slots,
                // "exp2" = waves per SIMD the variant is compiled for): C3b's two launches only
                if (!chroma && cols == 3) {       // experiment "exp4" (dev_plan_strip.hip): 192-column luma strips, planar 9 .. 15-bit sources only
                    if (!s16) { log_msg(c, 0, "internal error: 192-column strip plan for an 8-bit source\n"); return; }
                    const size_t lds = (size_t)g.lds_dma_bytes;
                    if (g.dma_ok && !c->tune.no_strip_dma) {
                        if (c->tune.exp[1]) hipLaunchKernelGGL((swsk::sws_k_strip_dma_v<false, 3, 4, true, 5>), grid, blk, lds, st, fs, p, g);
                        else hipLaunchKernelGGL((swsk::sws_k_strip_dma_v<false, 3, 4, false, 5>), grid, blk, lds, st, fs, p, g);
                    } else hipLaunchKernelGGL((swsk::sws_k_strip_march<true, false, 3>), grid, blk, g.lds_bytes, st, fs, p, g);
                    return;
                }
                if (s16 && g.dma_ok && !c->tune.no_strip_dma && c->tune.exp[2] == 7 && cols == (chroma ? 1 : 2)) {     // 128 / 64-column strips (strip_cols_l=2 strip_cols_c=1) compiled for 7 waves per SIMD
                    const bool abs = c->tune.exp[1] != 0;
                    if (chroma) { if (abs) hipLaunchKernelGGL((swsk::sws_k_strip_dma_v<true, 1, 4, true, 7>), grid, blk, g.lds_dma_bytes, st, fs, p, g);
                                  else hipLaunchKernelGGL((swsk::sws_k_strip_dma_v<true, 1, 4, false, 7>), grid, blk, g.lds_dma_bytes, st, fs, p, g); }
                    else        { if (abs) hipLaunchKernelGGL((swsk::sws_k_strip_dma_v<false, 2, 4, true, 7>), grid, blk, g.lds_dma_bytes, st, fs, p, g);
                                  else hipLaunchKernelGGL((swsk::sws_k_strip_dma_v<false, 2, 4, false, 7>), grid, blk, g.lds_dma_bytes, st, fs, p, g); }
                    return;
                }
                if (s16 && g.dma_ok && !c->tune.no_strip_dma && (c->tune.exp[0] || c->tune.exp[1]) && cols == (chroma ? 2 : 4)) {
                    const int D = c->tune.exp[0] ? c->tune.exp[0] : 4, wpe = c->tune.exp[2] ? c->tune.exp[2] : 4;
                    const bool abs = c->tune.exp[1] != 0;
                    const size_t lds = (size_t)g.lds_dma_bytes * D / 4;
#define SWS_V(C, K, DD, A, W) hipLaunchKernelGGL((swsk::sws_k_strip_dma_v<C, K, DD, A, W>), grid, blk, lds, st, fs, p, g)
#define SWS_VC(DD, A, W) do { if (chroma) SWS_V(true, 2, DD, A, W); else SWS_V(false, 4, DD, A, W); } while (0)
                    if (D == 4 && abs && wpe == 4) SWS_VC(4, true, 4);
                    else if (D == 2 && !abs && wpe == 8) SWS_VC(2, false, 8);
                    else if (D == 2 && abs && wpe == 8) SWS_VC(2, true, 8);
                    else if (D == 3 && abs && wpe == 5) SWS_VC(3, true, 5);
                    else if (D == 2 && abs && wpe == 6) SWS_VC(2, true, 6);
                    else { log_msg(c, 0, "no such strip_dma experiment variant\n"); }
#undef SWS_VC
#undef SWS_V
                    return;
                }
                if (s16 && g.dma_ok && !c->tune.no_strip_dma) {   // 16-bit sources: LDS-DMA ring, 3 row pairs in flight per wave
                    if (chroma) { if (cols == 1) SWS_STRIP_DMA(true, 1); else SWS_STRIP_DMA(true, 2); }
                    else        { if (cols == 2) SWS_STRIP_DMA(false, 2); else SWS_STRIP_DMA(false, 4); }
                    return;
                }
                if (chroma) { if (s16) { if (cols == 1) SWS_STRIP(true, true, 1); else SWS_STRIP(true, true, 2); }
                              else     { if (cols == 1) SWS_STRIP(false, true, 1); else SWS_STRIP(false, true, 2); } }
                else        { if (s16) { if (cols == 2) SWS_STRIP(true, false, 2); else SWS_STRIP(true, false, 4); }
                              else     { if (cols == 2) SWS_STRIP(false, false, 2); else SWS_STRIP(false, false, 4); } }
#undef SWS_STRIP
#undef SWS_STRIP_DMA
            };
            // Both plane classes in one grid (sws_k_strip_*_lc) when the call is small -- bands of at most six times the minimum length
            // in one resident round, i.e. a few frames: such a call is launch- and tail-bound (two launches of about 8 us around 10 - 20 us of work; one 4K -> 1080p
            // frame 34 -> 25 us, 1080p -> 720p 22 -> 15 us).  Larger calls keep the two launches: the combined kernel carries both bodies' registers
            // (128 VGPRs and a few spills) and measured 15 - 35 % slower per frame on batches (C3b x32: 1.217 vs 0.898 ms).
            const int minrows_f = std::max(1, c->tune.strip_min_rows);
            const int64_t wave_rows = ((int64_t)d->stripL.strips * p.dstH + (int64_t)d->stripC.strips * p.chrDstH) * n;     // rows of 256 samples, all waves together
            if (which == 3 && d->stripL.TW == 256 && d->stripC.TW == 128 && d->stripC.npv <= 8 && !c->tune.no_strip_fuse &&
                (wave_rows + target - 1) / target <= 6 * minrows_f) {
                SwsStripGeom gl = d->stripL, gc = d->stripC;
                const int minrows = minrows_f;
                const int rows = (int)std::max<int64_t>(minrows, (wave_rows + target - 1) / target);
                gl.band_rows = gc.band_rows = rows;
                gl.bands = cdiv(p.dstH, rows); gc.bands = cdiv(p.chrDstH, rows);
                gl.debug = gc.debug = c->tune.debug;
                const int blocksL = (int)cdiv((int64_t)gl.strips * gl.bands, 4), blocksC = (int)cdiv((int64_t)gc.strips * gc.bands, 4);
                const dim3 grid(blocksL + blocksC, 1, n);
                const bool dma = s16 && gl.dma_ok && gc.dma_ok && !c->tune.no_strip_dma;
                if (dma) hipLaunchKernelGGL((swsk::sws_k_strip_dma_lc<4, 2>), grid, blk, std::max(gl.lds_dma_bytes, gc.lds_dma_bytes), st, fs, p, gl, gc, blocksL);
                else if (s16) hipLaunchKernelGGL((swsk::sws_k_strip_march_lc<true, 4, 2>), grid, blk, std::max(gl.lds_bytes, gc.lds_bytes), st, fs, p, gl, gc, blocksL);
                else hipLaunchKernelGGL((swsk::sws_k_strip_march_lc<false, 4, 2>), grid, blk, std::max(gl.lds_bytes, gc.lds_bytes), st, fs, p, gl, gc, blocksL);
                return 0;
            }
            if (which == 3 && launch_strip_short_lc(L)) return 0;       // (experiment "exp3": off unless asked for)
            // (8-bit sources with short filters: the short instantiations on the plan's own strip widths, k_strip2.hip)
            if ((which & 1) && !launch_strip_short(L, d->stripLs_ok ? d->stripLs : d->stripL, p.dstH, false)) launch(d->stripL, p.dstH, false);
            if ((which & 2) && !launch_strip_short(L, d->stripCs_ok ? d->stripCs : d->stripC, p.chrDstH, true)) launch(d->stripC, p.chrDstH, true);
    return 0;
}

int launch_strip(const LaunchCtx &L) { return launch_strip_planes(L, L.p->no_chroma ? 1 : 3); }
int launch_strip_luma(const LaunchCtx &L) { return launch_strip_planes(L, 1); }   // (gray -> gray: the luma launch alone)

// Same-size planar YUV -> planar / semi-planar YUV whose luma filters are the identity in both directions and whose chroma is scaled
// (yuv422p -> yuv420p, yuv444p -> yuv420p / nv12, 10-bit -> 8-bit twins ...): the luma plane is a streaming per-sample pass (the scaler's own
// arithmetic with one tap: k_layout.hip launch_layout_plane1), the chroma planes are the strip kernel's chroma launch.
int launch_mixed(const LaunchCtx &L)
{
    int r = launch_layout_plane1(L);
    if (r < 0) return r;
    if (isGray(L.c->opts.src_format)) { launch_gray_chroma(L); return 0; }     // (a gray source: constant chroma planes, k_stream.hip)
    return launch_strip_planes(L, 2);
}

// Scaled packed 24 / 32 bpp RGB sources (dev_prepare_on: rgbread_on): the reader pre-pass writes what the reference's input stage hands to
// hScale16To15_c -- 16-bit Y[srcH][srcW] and U / V[srcH][srcW / 2] -- into a per-frame working picture, and the strip kernel runs on those planes
// as on a planar 16-bit source (same filters, same hshift: the context was initialised for the RGB source).
int launch_rgbread_strip(const LaunchCtx &L)
{
    SwsInternal *c = L.c; DeviceState *d = L.d; const SwsDevParams &p = *L.p; hipStream_t st = L.st;
    const int n = L.n;
    if (L.vec && launch_strip_rgbsrc(L)) return 0;      // half-width-chroma YUV destinations: one launch, no working picture (k_striprgbsrc.hip)
    auto a256 = [](int64_t v) { return (v + 255) & ~(int64_t)255; };
    const int strideY = (int)a256(2 * (int64_t)p.srcW), strideC = (int)a256(2 * (int64_t)p.chrSrcW);
    const bool alpha = (d->fullchr_on == 2 || d->alpha_launch) && p.srcKind == SRCK_RGB32;     // (a scaled alpha plane behind the strip kernels: dev_plan*.hip)
    const int64_t offU = (int64_t)strideY * p.srcH, offV = offU + (int64_t)strideC * p.srcH, offA = a256(offV + (int64_t)strideC * p.srcH);
    const int64_t frame_bytes = a256(offA + (alpha ? (int64_t)strideY * p.srcH : 0));
    d->rgbread_frame_bytes = frame_bytes; d->rgbread_offA = alpha ? offA : -1; d->rgbread_strideY = strideY;
    int r = grow(c, &d->rgbread_img, &d->rgbread_bytes, (size_t)frame_bytes * (size_t)n);
    if (r < 0) return r;
    uint8_t *base = (uint8_t *)d->rgbread_img;
    r = launch_rgb_read16(L, base, frame_bytes, offU, offV, strideY, strideC, offA, alpha ? (p.src_a_pos | (p.src_alpha_opaque ? 8 : 0)) : -1);
    if (r < 0) return r;
    std::vector<SwsFramePtrs> fr(L.frames, L.frames + n);
    for (int i = 0; i < n; i++) {
        uint8_t *fb = base + (size_t)i * (size_t)frame_bytes;
        fr[(size_t)i].src[0] = fb; fr[(size_t)i].src[1] = fb + offU; fr[(size_t)i].src[2] = fb + offV; fr[(size_t)i].src[3] = nullptr;
        fr[(size_t)i].srcStride[0] = strideY; fr[(size_t)i].srcStride[1] = fr[(size_t)i].srcStride[2] = strideC; fr[(size_t)i].srcStride[3] = 0;
    }
    LaunchCtx L2 = L;
    SwsDevParams p2 = p;
    p2.srcKind = SRCK_PLANAR16; p2.u_plane_src = 1; p2.src_shift = 0;
    L2.p = &p2; L2.frames = fr.data();
    if (n == 1) { L2.fs.table = nullptr; L2.fs.one = fr[0]; }
    else {
        L2.fs.table = table_upload(c, d, st, TAB_FRAMES2, fr.data(), n);
        if (!L2.fs.table) return d->ring.last_err;
    }
    return launch_strip_planes(L2, p.no_chroma ? 1 : 3);     // (a gray destination: the luma launch alone)
}

} // namespace swship
