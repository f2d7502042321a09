// Translation unit of the SHORT-filter instantiations of the marching strip kernel (kernels_strip.hpp): 8-bit planar / semi-planar sources whose
// vertical filters span at most 6 source-row pairs, whose horizontal filters have at most 6 tap pairs and whose strip windows fit one 16-byte
// chunk per lane -- bilinear / bicubic / area scaling at ratios up to about 2.5:1, i.e. C1 (720p -> 360p) and the upper rungs of an ABR ladder.
// Same body (strip_body), other template parameters:
//   * RD = 3 / 4 / 6 ring slots instead of 8: the per-step ring shift costs RD - 1 v_mov per column (28 of about 130 vector instructions per
//     step of C1 at RD = 8), and the ring's registers decide how many waves fit a SIMD;
//   * PARTS = 1: one chunk per lane and staged row -- the general luma form issues the loads, the byte -> u16 expansion (8 v_perm) and the two
//     ds_write_b128 of a second chunk per row whether or not any lane has one (straight-line code);
//   * strips of 64 * COLS columns with COLS chosen per picture width (dev_plan*.hip): 640 columns are 2 strips of 320 (COLS = 5) instead of 3
//     strips of 256 with the last one half empty;
//   * one kernel per horizontal tap-pair count, so that each gets its own register allocation (the general form's switch over 1 .. 8 pairs
//     allocates for 8): 5 to 8 waves per SIMD instead of 4.  With one source-row pair in flight per wave the march is bound by memory latency
//     (C1: 4096 waves x 1 KB in flight = 2 TB/s at 2 us), so more resident waves is more bytes in flight; the bands of a launch are cut for the
//     occupancy the runtime reports for the instantiation.
#include <algorithm>
#include <map>
#include <mutex>

#include "devstate.hpp"
#include "kernels_strip.hpp"
#include "kernels_strip8.hpp"

namespace swsk {

template <bool CHROMA, int COLS, int NPH, int RD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) sws_k_strip_short(SwsFrameSet fs, SwsDevParams p, SwsStripGeom g)
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
    strip_body<false, CHROMA, COLS, NPH, RD, 1>(f, p, g, strip, y0, y1, smem, wib, lane);
}

} // namespace swsk

namespace swship {

typedef void (*StripShortFn)(SwsFrameSet, SwsDevParams, SwsStripGeom);

template <bool CH, int K, int R>
static StripShortFn short_fn_nph(int nph)
{
    switch (nph) {
    case 1: return swsk::sws_k_strip_short<CH, K, 1, R>;
    case 2: return swsk::sws_k_strip_short<CH, K, 2, R>;
    case 3: return swsk::sws_k_strip_short<CH, K, 3, R>;
    case 4: return swsk::sws_k_strip_short<CH, K, 4, R>;
    case 5: return swsk::sws_k_strip_short<CH, K, 5, R>;
    default: return swsk::sws_k_strip_short<CH, K, 6, R>;
    }
}
template <bool CH, int K>
static StripShortFn short_fn_rd(int rd, int nph)
{
    return rd == 3 ? short_fn_nph<CH, K, 3>(nph) : rd == 4 ? short_fn_nph<CH, K, 4>(nph) : short_fn_nph<CH, K, 6>(nph);
}

template <bool CH, int K, int R, bool NV>
static StripShortFn dma8_fn_nph(int nph)
{
    switch (nph) {
    case 1: return swsk::sws_k_strip_dma8<CH, K, 1, R, NV>;
    case 2: return swsk::sws_k_strip_dma8<CH, K, 2, R, NV>;
    case 3: return swsk::sws_k_strip_dma8<CH, K, 3, R, NV>;
    case 4: return swsk::sws_k_strip_dma8<CH, K, 4, R, NV>;
    case 5: return swsk::sws_k_strip_dma8<CH, K, 5, R, NV>;
    default: return swsk::sws_k_strip_dma8<CH, K, 6, R, NV>;
    }
}
template <bool CH, int K, bool NV = false>
static StripShortFn dma8_fn_rd(int rd, int nph)
{
    return rd == 3 ? dma8_fn_nph<CH, K, 3, NV>(nph) : rd == 4 ? dma8_fn_nph<CH, K, 4, NV>(nph) : rd == 6 ? dma8_fn_nph<CH, K, 6, NV>(nph) : dma8_fn_nph<CH, K, 8, NV>(nph);
}

// resident waves per SIMD of a kernel at its LDS size (asked once per kernel and device)
static int waves_per_simd(StripShortFn fn, int lds, int device)
{
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, int> cache;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair((const void *)fn, device * 4096 + lds / 64);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, (const void *)fn, 256, (size_t)lds) != hipSuccess) { (void)hipGetLastError(); blocks = 4; }
    const int w = std::max(1, std::min(8, blocks));     // a block is 4 waves, one per SIMD
    cache[key] = w;
    return w;
}

// 1 = launched, 0 = not a shape of the short family (the caller goes on to the general instantiations)
int launch_strip_short(const LaunchCtx &L, const SwsStripGeom &g0, int H, bool chroma)
{
    SwsInternal *c = L.c; DeviceState *d = L.d; const SwsDevParams &p = *L.p; hipStream_t st = L.st; const SwsFrameSet &fs = L.fs;
    const int n = L.n;
    if (c->tune.no_strip_short) return 0;
    if (p.srcKind != SRCK_PLANAR8 && p.srcKind != SRCK_NV12) return 0;
    const int cols = g0.TW / 64;
    const bool nv = chroma && p.srcKind == SRCK_NV12;          // interleaved chroma bytes: the NV instantiations (strips of up to 192 columns: windows of 2 bytes per sample)
    const bool dma8 = g0.dma8_ok && g0.hT8 && L.vec && !c->tune.no_strip_dma8 && (!nv || cols <= 3);
    // (the register-staged short instantiations stop at 6 tap pairs each way; the LDS-DMA form takes vertical filters of up to 8 row pairs: Lanczos at 2:1)
    if (g0.npv > (dma8 ? 8 : 6) || (dma8 ? g0.nph8 : g0.nph) > 6 || g0.NCmax / 16 > 64) return 0;
    // (the LDS-DMA form needs fewer registers per column: strips of up to 448 luma / 320 chroma columns)
    if (dma8 ? (chroma ? (cols < 1 || cols > 5) : (cols < 3 || cols > 7)) : (chroma ? (cols < 1 || cols > 3) : (cols < 3 || cols > 5))) return 0;
    SwsStripGeom g = g0;
    const int rd = g.npv <= 3 ? 3 : g.npv <= 4 ? 4 : g.npv <= 6 ? 6 : 8;
    StripShortFn fn;
    int lds = (dma8 ? g.lds_dma8_bytes : g.lds_bytes) + (c->tune.strip_lds_pad_kb > 0 ? c->tune.strip_lds_pad_kb * 1024 : 0);
    if (dma8) {
        if (nv) fn = cols == 1 ? dma8_fn_rd<true, 1, true>(rd, g.nph8) : cols == 2 ? dma8_fn_rd<true, 2, true>(rd, g.nph8) : dma8_fn_rd<true, 3, true>(rd, g.nph8);
        else if (chroma) fn = cols == 1 ? dma8_fn_rd<true, 1>(rd, g.nph8) : cols == 2 ? dma8_fn_rd<true, 2>(rd, g.nph8) : cols == 3 ? dma8_fn_rd<true, 3>(rd, g.nph8) :
                         cols == 4 ? dma8_fn_rd<true, 4>(rd, g.nph8) : dma8_fn_rd<true, 5>(rd, g.nph8);
        else fn = cols == 3 ? dma8_fn_rd<false, 3>(rd, g.nph8) : cols == 4 ? dma8_fn_rd<false, 4>(rd, g.nph8) : cols == 5 ? dma8_fn_rd<false, 5>(rd, g.nph8) :
                  cols == 6 ? dma8_fn_rd<false, 6>(rd, g.nph8) : dma8_fn_rd<false, 7>(rd, g.nph8);
    } else
    if (chroma) fn = cols == 1 ? short_fn_rd<true, 1>(rd, g.nph) : cols == 2 ? short_fn_rd<true, 2>(rd, g.nph) : short_fn_rd<true, 3>(rd, g.nph);
    else fn = cols == 3 ? short_fn_rd<false, 3>(rd, g.nph) : cols == 4 ? short_fn_rd<false, 4>(rd, g.nph) : short_fn_rd<false, 5>(rd, g.nph);
    // one resident round of waves at the instantiation's occupancy; bands of at least `minrows` output rows (k_strip.hip)
    const int wps = c->tune.strip_short_waves > 0 ? c->tune.strip_short_waves : waves_per_simd(fn, lds, d->device);
    const int target = (int)((int64_t)c->tune.strip_waves * wps / 4);
    const int minrows = std::max(1, c->tune.strip_min_rows);
    const int bands = std::max(1, std::min(target / std::max(1, g.strips * n), (H + minrows - 1) / minrows));
    g.debug = c->tune.debug;
    g.band_rows = (H + bands - 1) / bands;
    g.bands = (H + g.band_rows - 1) / g.band_rows;
    const dim3 grid(cdiv((int64_t)g.strips * g.bands, 4), 1, n), blk(256);
    hipLaunchKernelGGL(fn, grid, blk, lds, st, fs, p, g);
    return 1;
}

// EXPERIMENT ("exp3", unmeasured): both plane classes of the byte-DMA form in one grid, for the shapes whose luma and chroma plans agree in ring depth -- C1
// (640 + 320 columns: 5 columns per lane both, 2 tap pairs, ring of 3).  1 = launched, 0 = not such a shape (the caller runs the two launches).
int launch_strip_short_lc(const LaunchCtx &L)
{
    SwsInternal *c = L.c; DeviceState *d = L.d; const SwsDevParams &p = *L.p;
    if (!c->tune.exp[3] || c->tune.no_strip_short || c->tune.no_strip_dma8 || p.srcKind != SRCK_PLANAR8 || !L.vec || p.no_chroma) return 0;
    SwsStripGeom gl = d->stripLs_ok ? d->stripLs : d->stripL, gc = d->stripCs_ok ? d->stripCs : d->stripC;
    if (!gl.dma8_ok || !gc.dma8_ok || !gl.hT8 || !gc.hT8 || gl.NCmax / 16 > 64 || gc.NCmax / 16 > 64) return 0;
    const int rdl = gl.npv <= 3 ? 3 : gl.npv <= 4 ? 4 : gl.npv <= 6 ? 6 : 8, rdc = gc.npv <= 3 ? 3 : gc.npv <= 4 ? 4 : gc.npv <= 6 ? 6 : 8;
    if (rdl != 3 || rdc != 3 || gl.TW != 320 || gc.TW != 320 || gl.nph8 != 2 || gc.nph8 != 2) return 0;          // (the one instantiation built for the measurement)
    auto fn = swsk::sws_k_strip_dma8_lc<5, 5, 2, 2, 3>;
    const int lds = std::max(gl.lds_dma8_bytes, gc.lds_dma8_bytes);
    int blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, (const void *)fn, 256, (size_t)lds) != hipSuccess) { (void)hipGetLastError(); blocks = 4; }
    const int wps = std::max(1, std::min(8, blocks));
    const int target = (int)((int64_t)c->tune.strip_waves * wps / 4), minrows = std::max(1, c->tune.strip_min_rows), n = L.n;
    const int64_t wave_rows = ((int64_t)gl.strips * p.dstH + (int64_t)gc.strips * p.chrDstH) * n;
    const int rows = (int)std::max<int64_t>(minrows, (wave_rows + target - 1) / target);
    gl.band_rows = gc.band_rows = rows;
    gl.bands = cdiv(p.dstH, rows); gc.bands = cdiv(p.chrDstH, rows);
    gl.debug = gc.debug = c->tune.debug;
    const int blocksL = (int)cdiv((int64_t)gl.strips * gl.bands, 4), blocksC = (int)cdiv((int64_t)gc.strips * gc.bands, 4);
    hipLaunchKernelGGL(fn, dim3(blocksL + blocksC, 1, n), dim3(256), lds, L.st, L.fs, p, gl, gc, blocksL);
    return 1;
}

} // namespace swship
