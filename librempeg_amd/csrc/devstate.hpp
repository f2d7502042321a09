// Host-side device state of a context and the interface between the launch planner (dev_plan*.hip) and the translation units
// that hold the kernels (k_*.hip).  One DeviceState per (context, GPU): tables, plans, scratch and the stream live on that GPU.
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "swsint.hpp"

#define AVERROR_EXTERNAL_ (-0x20545845) /* FFERRTAG('E','X','T',' '), libavutil/error.h */

namespace swship {

// Frame tables (SwsFramePtrs arrays: fs.table of a batched launch) live in ONE pinned host block and ONE device block per DeviceState, used as a
// ring: every upload takes the next span, a span is recycled only when the launch sets that read it have passed an event recorded behind them,
// so neither a call with new frame pointers nor the sub-batches of one call ever wait for the stream (table_upload / table_batch_end, dev_state.hip).
struct TableRing {
    SwsFramePtrs *dev = nullptr, *host = nullptr;
    int cap = 0, head = 0;
    struct Span { int off, n; };
    struct Batch { std::vector<Span> spans; hipEvent_t ev = nullptr; };
    std::vector<Batch> inflight;     // launch sets whose event has not been seen complete yet, oldest first
    std::vector<Span> cur;           // spans handed out (or reused) since the last table_batch_end()
    std::vector<hipEvent_t> pool;    // spare events
    struct Cached { int off = -1, n = 0; } cache[8];   // last span of a slot: an identical table is not uploaded again
    // blocks a regrow replaced while a launch set was still being built: table pointers already handed out for that set stay valid until the next
    // regrow (behind its waits) or the end of the state
    struct Retired { SwsFramePtrs *dev, *host; };
    std::vector<Retired> retired;
    int last_err = 0;                // why the last table_upload() returned nullptr
};
enum { TAB_MAIN = 0, TAB_AUX0 = 1 /* .. TAB_AUX0 + 4 */, TAB_FRAMES2 = 6 };

// What table_put() uploaded into a device table block: kept so that sws_hip_debug_check() can read the block back at any later time and say whether the
// context's device tables are still what the host built (the rare events whose context stays wrong for its lifetime)
struct TableRecord { const void *dst; size_t bytes; uint64_t hash; uint64_t serial; };   // serial: the planning run that wrote it (DeviceState::plan_serial)

// a peer GPU's table upload held back for the RCCL broadcast of dev_rccl.hip (the host copy stays for the fallback)
struct TableDeferred { void *dst; size_t bytes; uint64_t hash; std::vector<uint8_t> data; };

struct DeviceState {
    int device = 0;
    bool defer_uploads = false;            // table_put() records instead of copying: rccl_deliver_tables() delivers
    std::vector<TableDeferred> deferred;
    bool dry = false;          // Tuning::dry_plan: planned without a GPU (fake table addresses, nothing uploaded, never launched)
    std::vector<TableRecord> tab_recs;
    uint64_t plan_serial = 0;       // counts planning runs (reset_plan_state): a block a re-plan no longer writes keeps its old serial and stays out of the plan digest
    uint64_t params_hash = 0;  // hash of `params` as dev_prepare_on() left it
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint64_t epoch = 0;        // SwsInternal::tables_epoch this state was built for (0 = never)
    void *d_tables = nullptr; size_t tables_bytes = 0;
    SwsDevParams params;
    bool unity_h = false, unity_v = false;
    bool all_x_mode = false;   // every output row uses the general yuv2rgb_X writer (vscale.c:135-169)
    int chr_window2 = 0;       // max chroma source rows spanned by a pair of output rows (wave kernel register budget)
    bool tile_ok = false; SwsTileGeom tileL, tileC; void *d_tilegeom = nullptr; size_t tilegeom_bytes = 0;
    bool rgb_march_ok = false; void *d_rgbplan = nullptr; size_t rgbplan_bytes = 0; int rgb_groups = 0, rgb_ncr = 6;   // sws_k_rgb_march
    void *d_be = nullptr; size_t be_bytes = 0;   // little-endian copies of big-endian source pictures
    void *d_xyz = nullptr; size_t xyz_bytes = 0; void *d_xyz_tab = nullptr;   // rgb48 copies of xyz12 source pictures; the four gamma LUTs
    bool dot2_ok = false; SwsTileGeom dotL, dotC; void *d_dot2 = nullptr; size_t dot2_bytes = 0;
    const SwsRgbSrcRow *rgbsrc_rows = nullptr;             // (in d_dot2)
    const SwsStripRow *rgbsrc2_rows = nullptr; int rgbsrc2_npv = 0;   // sws_k_rgbsrc_unity2: the chroma rows' plan entries, taps laid out against its register ring (in d_dot2, behind rgbsrc_rows)
    bool rgb444_ok = false;                                // sws_k_rgb_yuv444_unity (8-bit RGB -> planar 8-bit 4:4:4 YUV of the same size, all filters the identity)
    bool rgbsrc_ok = false;                                // sws_k_rgbsrc_unity (packed 24 / 32 bpp RGB -> 8-bit 4:2:x YUV of the same size)
    bool mixed_ok = false;                                 // identity luma (streaming plane pass) + strip kernel on the chroma planes only (launch_mixed)
    bool strip_ok = false; SwsStripGeom stripL, stripC;    // sws_k_strip_march (tables live in d_dot2)
    // the same plan on strips of another width for the short-filter instantiations (k_strip2.hip): own colStart / colCount, everything else shared
    bool stripLs_ok = false, stripCs_ok = false; SwsStripGeom stripLs, stripCs;
    bool striprgb_ok = false; SwsStripGeom stripRL, stripRC; bool striprgb_long = false;   // sws_k_strip_rgb: scaled planar 8-bit YUV -> 24 / 32 bpp RGB
    // a semi-planar source the strip-RGB kernels read THEMSELVES on 16-byte aligned frames (no split pass): 1 nv12-like (sws_k_strip_rgb8<..., NV>: chroma
    // bytes de-interleaved by the h-stage), 2 p010-like (sws_k_strip_rgb<..., S16, P01X>: words shifted down and de-interleaved while staging);
    // swap: V first (nv21 / nv42); shift: the words' right shift; now: launch_plan_le skipped the split for the call in flight
    int striprgb_direct = 0, striprgb_direct_swap = 0, striprgb_direct_shift = 0; bool striprgb_direct_now = false;
    void *scratch = nullptr; size_t scratch_bytes = 0;
    void *stage_src = nullptr; size_t stage_src_bytes = 0;
    void *stage_dst = nullptr; size_t stage_dst_bytes = 0;
    TableRing ring;            // frame tables of the batched launches
    void *casc_img = nullptr; size_t casc_bytes = 0; int casc_stride = 0;
    void *d_pal = nullptr; size_t d_pal_bytes = 0;   // palette-expanded sources: per frame pal_yuv[256] + pal_rgb[256]
    void *d_ed_err = nullptr;   // error-diffusion line of an 8 / 4 bpp destination: 3 x (dst_w + 3) ints, zeroed once, carried between frames
    void *casc_img2 = nullptr; size_t casc_bytes2 = 0; void *d_gamma_tab = nullptr;   // gamma cascade: second RGBA64 intermediate, the two 65536-entry tables
    void *slice_img = nullptr; size_t slice_bytes = 0;   // source image assembled from sws_scale() slices (scaled path)
    void *d_vlines = nullptr; size_t vlines_bytes = 0; bool vlines_on = false;   // virtual source lines of the two-pass path (SwsDevParams::vlines) + their vertical positions
    // scaled packed-RGB sources: the reader pre-pass writes 16-bit Y / U / V planes per frame (k_stream.hip launch_rgb_read16), the strip kernel
    // reads them through a second frame table (k_strip.hip launch_rgbread_strip)
    bool rgb2rgb_ok = false, rgb2rgb_now = false; int rgb2rgb_npx = 0; SwsStripGeom stripL2{}, stripC2{};   // ... and packed RGB destinations through the full-chroma writers in the same launch (k_striprgb2rgb.hip): plans on strips of 128 columns
    bool striprgbsrc_ok = false; int striprgbsrc_npx = 0;   // ... or, for half-width-chroma YUV destinations, one launch that reads the RGB rows itself (k_striprgbsrc.hip); widest strip window in pixels
    bool rgbread_on = false; void *rgbread_img = nullptr; size_t rgbread_bytes = 0;
    int64_t rgbread_frame_bytes = 0, rgbread_offA = -1; int rgbread_strideY = 0;   // layout of the last reader pre-pass (the alpha launch of a full-chroma RGB destination reads its A plane)
    // helper passes around a packed / semi-planar side of the scaler (dev_prepare_on decides, launch_plan_le runs them):
    int fullchr_on = 0, fullchr_kind = 0;              // full-chroma packed RGB destination (2: with a scaled alpha plane): the strip kernels write int32 sum planes (DSTK_RAW32), sws_k_fullchr_rgb follows; the real dstKind
    int fullchr_direct = 0;                            // 1 / 2: the epilogue reads the 8 / 16-bit planes of a same-size 4:4:4 planar source itself (four identity filters): no strip launch, no working picture
    int alpha_launch = 0;                              // planar YUV destination with a scaled alpha plane (needAlpha): one more luma launch of the strip kernel, A samples -> dst[3]
    int join422 = 0;                                   // packed 4:2:2 destination through the planar writers + interleave: 1 yuyv-like, 2 uyvy
    void *join_img = nullptr; size_t join_bytes = 0;   //   its planar 4:2:2 working pictures (one per frame of the call)
    int split_mode = 0, split_shift = 0;               // source split into planar working planes: 1 / 2 packed 4:2:2 (yuyv-like / uyvy) | 4 V first; 8 semi-planar 8-bit
                                                       //   chroma | 16 V first; 32 p010-style planes, every word >> split_shift
    void *split_img = nullptr; size_t split_bytes = 0;
    // (frame tables of the helper passes: ring slots TAB_AUX0 + 0 join / RGB epilogue, 1 split, 2 alpha launch, 3 / 4 staging in / out)
    void *stage_img = nullptr; size_t stage_bytes = 0; // 16-byte aligned copies of the planes of pictures that are not (launch_plan_le: only under the helper passes)
    hipEvent_t ev0 = nullptr, ev1 = nullptr; bool timing = false; bool timed = false;
    hipEvent_t ev_loan = nullptr;   // orders the context's own stream against a borrowed frames stream (dev_borrow_stream)
};

#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    log_msg(c, 0, "HIP error %s at %s:%d: %s\n", hipGetErrorName(e_), __FILE__, __LINE__, #expr); \
    (void)hipGetLastError(); return AVERROR_EXTERNAL_; } } while (0)

// saves the caller's current device and restores it on scope exit: library entry points never retarget the caller's HIP calls
struct DeviceGuard {
    int saved = -1;
    DeviceGuard() { if (hipGetDevice(&saved) != hipSuccess) { (void)hipGetLastError(); saved = -1; } }
    ~DeviceGuard() { if (saved >= 0) (void)hipSetDevice(saved); }
};

static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

// everything a kernel-holding translation unit needs to launch its part of a plan over `n` device-resident frames
struct LaunchCtx {
    SwsInternal *c;
    DeviceState *d;
    const SwsDevParams *p;
    hipStream_t st;
    SwsFrameSet fs;                 // frame table (device) or the single frame
    const SwsFramePtrs *frames;     // host copy of the descriptors
    int n, sliceY, sliceH;
    bool vec;                       // all planes and strides 16-byte aligned
};

// buffer-descriptor kernels address a plane as base + 32-bit offset: strides must be positive and planes below 2 GiB
static inline bool frames_desc_ok(const SwsFramePtrs *fr, int n, int srcH, int dstH)
{
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 4; k++) {
            if (fr[i].src[k] && (fr[i].srcStride[k] <= 0 || (int64_t)fr[i].srcStride[k] * srcH >= (int64_t)1 << 31)) return false;
            if (fr[i].dst[k] && (fr[i].dstStride[k] <= 0 || (int64_t)fr[i].dstStride[k] * dstH >= (int64_t)1 << 31)) return false;
        }
    return true;
}

// The context-level half of sws_k_mixed_join422's shape test: ONE predicate for the planner's path name (dev_plan*.hip) and the launcher (k_stream.hip), which adds
// the per-call alignment test of the frames -- a context whose frames are not 16-byte aligned still runs the three passes under the fused kernel's name
static inline bool mixed_join422_shape(const SwsInternal *c, const DeviceState *d, const SwsDevParams &p)
{
    return d->unity_h && c->srcBpc == 8 && !p.range_active && !c->tune.no_wave && (p.srcKind == SRCK_PLANAR8 || p.srcKind == SRCK_NV12) && !(p.dstW & 7) && p.vChrFs <= 16;
}

// ---- k_misc.hip: element-per-thread unscaled converters and helper passes ----
int  launch_misc(const LaunchCtx &L);                                   // every PLAN_UNSC_* plan not named below
void launch_fill_alpha(const LaunchCtx &L, int w, int y0, int rows, int bits);
void launch_update_palette(const LaunchCtx &L);
int launch_layout_plane1(const LaunchCtx &L);   // k_layout.hip: the luma plane of a context with identity luma filters (launch_mixed)
int launch_mixed(const LaunchCtx &L);           // k_strip.hip
int launch_rgb444(const LaunchCtx &L);          // k_stream.hip
int launch_layout(const LaunchCtx &L);   // k_layout.hip: 1 = launched, 0 = not a shape of the streaming family
void launch_ed_mono(hipStream_t st, const uint8_t *lum, int64_t lumStride, uint8_t *dst, int64_t dstStride, int n, int h, int *errline, int white);
void launch_alpha_merge(const LaunchCtx &L, int npix, int y0, int rows, int a_pos);
void launch_bswap(hipStream_t st, const uint8_t *src, int64_t sstride, uint8_t *dst, int64_t dstride, int rows, int row_bytes, int unit);
void launch_gamma_rgba64(hipStream_t st, uint8_t *img, int64_t stride, int w, int rows, const uint16_t *table);
void launch_ed_rgb8(hipStream_t st, const uint8_t *rgb, int64_t rgbStride, uint8_t *dst, int64_t dstStride, int w, int h, int *errline,
                    int bpp8, int r8, int g8, int b8);
void launch_xyz12(hipStream_t st, const uint8_t *src, int64_t sstride, uint8_t *dst, int64_t dstride, int w, int rows,
                  const uint16_t *gamma_in, const uint16_t *gamma_out, int to_rgb);
// ---- k_yuv2rgb.hip: PLAN_UNSC_YUV2RGB (C2a) ----
int  launch_yuv2rgb(const LaunchCtx &L);
// ---- k_rgb_unity.hip: PLAN_MAIN, identity horizontal filters, 8-bit planar / nv12 -> 24/32 bpp LUT writers (C2b, C4) ----
int  launch_rgb_unity(const LaunchCtx &L);
// ---- k_stream.hip: planarToP01x (C3a), planar float RGB -> yuv444 (C5) ----
int  launch_p01x(const LaunchCtx &L);
int  launch_f32rgb(const LaunchCtx &L);
// ---- k_strip.hip / k_tile.hip: fused h+v polyphase kernels for planar / semi-planar outputs (C3b, C1) ----
int  launch_strip(const LaunchCtx &L);
int  launch_rgbsrc(const LaunchCtx &L);
int  launch_strip_rgb2rgb(const LaunchCtx &L);   // k_striprgb2rgb.hip: scaled packed RGB -> packed RGB, one launch; 1 = launched, 0 = not its shape
int  launch_strip_rgbsrc(const LaunchCtx &L);    // k_striprgbsrc.hip: scaled packed-RGB source, one launch; 1 = launched, 0 = not its shape
int  launch_rgbsrc2(const LaunchCtx &L);         // k_rgbsrc2.hip: the wave-march form; 1 = launched, 0 = not its shape
int  launch_rgbread_strip(const LaunchCtx &L);   // k_strip.hip: scaled packed 24 / 32 bpp RGB source: reader pre-pass + strip kernel on its 16-bit planes
void launch_fullchr_rgb(const LaunchCtx &L);   // k_stream.hip
void launch_stage_planes(const LaunchCtx &L, const int row_bytes[4], const int rows[4], bool in);   // k_stream.hip: src[k] -> dst[k] plane copies, one side 16-byte aligned
int  launch_rgb_read16(const LaunchCtx &L, uint8_t *base, int64_t frame_bytes, int64_t offU, int64_t offV, int strideY, int strideC, int64_t offA, int a_pos);   // k_stream.hip
int  launch_strip_wide(const LaunchCtx &L, int which);   // k_stripwide.hip: the 19-bit form (destinations of 16 bits per component, int32 sums of the wide planar RGB route)
void launch_gray_chroma(const LaunchCtx &L);                    // k_stream.hip: the chroma planes of a gray source in a YUV destination (or the chroma sums behind an RGB epilogue)
bool fullchr_gray_const(const LaunchCtx &L);                    // k_stream.hip: ... or no launch: the full-chroma RGB epilogue computes the constants itself
int  launch_mixed_join422(const LaunchCtx &L, bool uyvy);       // k_stream.hip: 1 = the mixed plan and its interleave ran as one pass
int  launch_strip_short(const LaunchCtx &L, const SwsStripGeom &g, int H, bool chroma);
int  launch_strip_short_lc(const LaunchCtx &L);   // k_strip2.hip, experiment "exp3": luma and chroma of the byte-DMA form as one grid; 1 = launched   // k_strip2.hip: 1 = launched, 0 = not a shape of the short family
int  launch_strip_luma(const LaunchCtx &L);      // k_strip.hip: the luma launch alone (the alpha plane of a full-chroma RGB destination goes through the luma filters)
int  launch_striprgb(const LaunchCtx &L);
int  launch_tile_dot2(const LaunchCtx &L);
int  launch_tile(const LaunchCtx &L);
// ---- k_generic.hip: two-pass / direct element-per-thread path ----
int  launch_generic(const LaunchCtx &L);

// dev_state.hip
int grow(SwsInternal *c, void **buf, size_t *cap, size_t need);
// a device copy of v[0 .. n) for launches on `st` (nullptr: a HIP error, logged); `slot` names the table's role for the identical-table cache
const SwsFramePtrs *table_upload(SwsInternal *c, DeviceState *d, hipStream_t st, int slot, const SwsFramePtrs *v, int n);
// closes the launch set that used the tables uploaded since the last call: records the event their spans are recycled behind
int table_batch_end(SwsInternal *c, DeviceState *d, hipStream_t st);

} // namespace swship
