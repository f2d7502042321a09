// Internal declarations of libswscale_hip (host side).  Product code: never includes or
// links anything from oracle/.
#pragma once
#include <cstdint>
#include <cstddef>
#include <string>
#include <vector>

#include "../../include/swscale_hip.h"
#include "devparams.h"

namespace swship {

// ---- pixel format descriptors (subset of libavutil/pixdesc.c) ----
struct CompDesc { int plane, step, offset, shift, depth; };
struct PixDesc {
    int fmt; const char *name; int nb_components; int log2_chroma_w, log2_chroma_h;
    CompDesc comp[4]; unsigned flags;
};
enum : unsigned { PIXFLAG_BE = 1u << 0, PIXFLAG_PLANAR = 1u << 4, PIXFLAG_RGB = 1u << 5,
                  PIXFLAG_ALPHA = 1u << 7, PIXFLAG_FLOAT = 1u << 9, PIXFLAG_PAL = 1u << 1, PIXFLAG_BAYER = 1u << 8 };
const PixDesc *pix_desc(int fmt);
int pix_be_twin(int fmt);
inline bool pix_is_xyz(int fmt) { return fmt == AV_PIX_FMT_XYZ12LE || fmt == AV_PIX_FMT_XYZ12BE; }   // little-endian twin of a big-endian format, or -1
int  pix_bits_per_pixel(const PixDesc *d);
int  pix_nb_planes(const PixDesc *d);
// predicates, libswscale/swscale_internal.h:746-988
bool is16BPS(int f); bool isNBPS(int f); bool isYUV(int f); bool isPlanarYUV(int f);
bool isSemiPlanarYUV(int f); bool isAnyRGB(int f); bool isGray(int f); bool isFloatFmt(int f); bool isFloat16Fmt(int f); bool isBayerFmt(int f);
bool isALPHA(int f); bool isPlanarRGB(int f); bool isPackedFmt(int f); bool isPlanarFmt(int f);
bool isSwappedChroma(int f); bool isDataInHighBits(int f);

// ---- polyphase filter bank (libswscale/utils.c:197-612) ----
struct FilterBank {
    std::vector<int16_t> taps;   // count * size (+3 replicated rows like the reference)
    std::vector<int32_t> pos;    // count (+3)
    int size = 0, count = 0;
};
enum { FILTER_OK = 0, FILTER_ERR = -1, FILTER_USE_CASCADE = -12345 };
struct FilterVec { const double *coeff = nullptr; int length = 0; int dst_length = 0; };   // SwsFilter vectors of one axis/plane class
int build_filter_bank(FilterBank &out, int xInc, int srcW, int dstW, int filterAlign, int one,
                      int scaler, int flags, const double param[2], int srcPos, int dstPos, const FilterVec &vec = FilterVec());

// ---- colour tables ----
struct Yuv2RgbLut {          // closed form of ff_yuv2rgb_c_init_tables (yuv2rgb.c:717-973)
    bool valid = false;
    int64_t cy = 0, yb0 = 0;  // y_table[k] = clip8((yb0 + k*cy + 0x8000) >> 16)
    int64_t crv = 0, cbu = 0, cgu = 0, cgv = 0; // already divided by cy
    int yoffs = 0;
    int y_offset = 0, y_coeff = 0, v2r = 0, v2g = 0, u2g = 0, u2b = 0; // 13-bit coeffs for full-chroma writers
};
void build_yuv2rgb(Yuv2RgbLut &l, const int inv_table[4], int fullRange, int brightness, int contrast, int saturation);
void build_rgb2yuv(int32_t out[9], const int table[4]);   // utils.c:614-706
struct RangeConv { bool active = false; uint32_t lumCoeff = 0, chrCoeff = 0; int64_t lumOffset = 0, chrOffset = 0; };
void build_range_conv(RangeConv &r, int src_range, int dst_range, int dstFormat, int dstBpc); // swscale.c:577-660
const int *yuv2rgb_coeffs(int colorspace);                 // yuv2rgb.c:47-66

// ---- execution plan chosen at init (which kernels run) ----
enum PlanKind {
    PLAN_NONE = 0,
    PLAN_UNSC_YUV2RGB,     // yuv2rgb_c_* (yuv2rgb.c:68-559)
    PLAN_UNSC_P01X,        // planarToP01xWrapper
    PLAN_UNSC_8_P01X,      // planar8ToP01xleWrapper
    PLAN_UNSC_PLANAR2NV12, // planarToNv12Wrapper
    PLAN_UNSC_NV122PLANAR, // nv12ToPlanarWrapper
    PLAN_UNSC_PLANARCOPY,  // planarCopyWrapper
    PLAN_UNSC_RGB2RGB,     // rgbToRgbWrapper (8-bit 24/32 bpp byte shuffles)
    PLAN_UNSC_PACKEDCOPY,  // packedCopyWrapper
    PLAN_UNSC_BGR24_YV12,  // bgr24ToYv12Wrapper -> ff_rgb24toyv12_c
    PLAN_UNSC_GBRP_PACKED, // planarRgbToRgbWrapper (gbrp -> 24/32 bpp packed)
    PLAN_UNSC_PLANAR2NV24, // planarToNv24Wrapper
    PLAN_UNSC_NV242PLANAR, // nv24ToPlanarWrapper
    PLAN_UNSC_NV242YUV420, // nv24ToYuv420Wrapper
    PLAN_UNSC_YVU9_YV12,   // yvu9ToYv12Wrapper -> planar2x_c
    PLAN_UNSC_YUV2GBRP,    // yuv420p_gbrp_c / yuv422p_gbrp_c (yuv2rgb.c:532,553)
    PLAN_UNSC_PACKED_GBRP, // rgbToPlanarRgbWrapper (8-bit packed RGB -> gbrp)
    PLAN_UNSC_PLANAR2P422, // yuv422pToYuy2/UyvyWrapper, planarToYuy2/UyvyWrapper
    PLAN_UNSC_P4222PLANAR, // yuyv/uyvy ToYuv420/422Wrapper
    PLAN_UNSC_ALPHABLEND,  // ff_sws_alphablendaway (alphablend.c)
    PLAN_UNSC_PLANARRGB_PLANARRGB,   // planarRgbToplanarRgbWrapper: gbrp <-> gbrap
    PLAN_UNSC_RGB16SHUFFLE,   // rgb48tobgr48 / rgb48to64 / rgb48tobgr64 / rgb64to48 / rgb64tobgr48 (rgb2rgb.c:322-413)
    PLAN_UNSC_PACKED16_GBRP16,// Rgb16ToPlanarRgb16Wrapper
    PLAN_UNSC_GBRP16_PACKED16,// planarRgb16ToRgb16Wrapper
    PLAN_UNSC_U8_TO_F32,      // uint_y_to_float_y_wrapper (swscale_unscaled.c:2095-2113)
    PLAN_UNSC_F32_TO_U8,      // float_y_to_uint_y_wrapper (:2115-2135)
    PLAN_UNSC_YUV2MONO,       // yuv2rgb_c_1_ordered_dither (yuv2rgb.c:457-517)
    PLAN_UNSC_RGB30_TO_16,    // x2rgb10to48 / x2rgb10to64 / x2rgb10tobgr48 / x2rgb10tobgr64 (rgb2rgb.c:415-471)
    PLAN_UNSC_RGB30_TO_GBRP,  // Rgb16ToPlanarRgb16Wrapper + packed30togbra10
    PLAN_UNSC_GBRP_TO_RGB30,  // planarRgb16ToRgb16Wrapper + gbr16ptopacked30
    PLAN_UNSC_YUV2RGB48, PLAN_UNSC_YUV2RGB16, PLAN_UNSC_YUV2RGB8, PLAN_UNSC_PAL2RGB, PLAN_UNSC_BAYER, PLAN_UNSC_RGBLOW,      // yuv2rgb_c_48 / yuv2rgb_c_bgr48 (yuv2rgb.c:107-125, :505-508)
    PLAN_MAIN,             // ff_swscale chain
    PLAN_CASCADE,          // two contexts through an intermediate image
};

int canonical_pix_fmt(int fmt); // handle_jpeg + handle_0alpha aliases (utils.c:773-842)

struct DeviceState;  // HIP side (devstate.hpp)

// Launch heuristics of a context.  sws_hip_set_option() changes them (A/B measurements, tests that force a kernel onto shapes
// the planner would give to another one); every combination produces the same bytes.  `debug` is read only by
// -DSWS_HIP_PROFILING builds (stage switches whose results are wrong).
struct Tuning {
    int strip_min_w = 320;         // narrower pictures stay on the LDS-tile kernel (measured: tools/narrow_shapes_times.py -- the strip kernels are level or ahead from 320 columns on)
    int strip_cols_l = 4, strip_cols_c = 2, strip_waves = 4096;
    int strip_min_rows = 4;        // shortest band of a strip-kernel launch (few frames per call: the serial walk of a wave is what a call waits for)
    int no_strip_u16 = 0;          // on: sources with samples of 16 significant bits keep the tile / element-per-thread kernels (round 4 behaviour)
    int no_wide_epilogue = 0;      // on: planar RGB of 16 bits / float32 behind the 19-bit strip kernel through the generic writer (sws_k_sum_writer) instead of sws_k_fullchr_gbrp16
    int no_strip_wide = 0;         // on: destinations of 16 bits per component (19-bit intermediates) keep the tile / element-per-thread kernels (round 4 behaviour)
    int no_strip_range = 0;        // on: conversions with MPEG <-> JPEG range conversion keep the tile / element-per-thread kernels (round 4 behaviour)
    int no_strip_fuse = 0;         // off: small calls put the luma and the chroma launch into one grid
    int strip_rgb_cols = 4;        // luma columns per lane of the strip kernel with the RGB epilogue (4: 256-pixel strips, 2: 128-pixel strips)
    int rgb_march_waves = 12288;   // resident waves the packed-RGB march kernel is banded for
    int tile_lds_kb = 40, tile_threads = 256;
    int p01x_ch = 1;
    int no_mixed = 0;               // off: identity-luma contexts keep the element-per-thread / tile kernels instead of plane pass + chroma strip
    int layout_ch = 1, no_layout_stream = 0;   // streaming layout converters (kernels_layout.hpp): 16-byte chunks per lane; off = the element-per-thread kernels
    int no_wave = 0, no_march = 0, no_rgbsrc = 0, no_strip = 0, no_strip_dma = 0, no_dot2 = 0, no_tile = 0;
    int no_strip_short = 0;        // off: 8-bit sources with short filters take the short instantiations of the strip kernel (k_strip2.hip)
    int no_generic_kinds = 0;      // off: single-pass element-per-thread launches take the kernel of their (source kind, destination kind) pair (k_generic_kinds.hip)
    int no_rgbread_kinds = 0;      // off: scaled x2rgb10 / rgb565-family / 9 - 14-bit planar RGB sources reach the strip kernels through the per-kind reader pre-pass
    int strip_cols_auto = 1;       // strip widths of the short family chosen per picture width (3 .. 5 luma, 1 .. 3 chroma columns per lane)
    int no_strip_dma8 = 0;         // off: 8-bit planar sources of the short family take the LDS-DMA form (kernels_strip8.hpp)
    int strip_lds_pad_kb = 0;      // experiments: LDS pad in KB per block of the short / dma8 kernels (lowers the occupancy; results never change)
    int no_strip_rgb2rgb = 0;      // off: scaled packed RGB -> packed RGB (full-chroma writers) takes the one-launch form (sws_k_strip_rgb2rgb) where it applies
    int no_fast_banks = 0;         // on: SWS_FAST_BILINEAR contexts keep ff_hyscale_fast_c / ff_hcscale_fast_c in the element-per-thread readers (round 4 behaviour) instead of two-tap banks under every plan
    int no_short_forms = 0;        // on: packed destinations whose rows take yuv2packed1 (blended chroma) / yuv2packed2 keep the element-per-thread writers (round 4 behaviour)
    int no_strip_rgbsrc = 0;       // off: scaled packed-RGB sources into half-width-chroma YUV take the one-launch strip form (sws_k_strip_rgbsrc) where it applies
    int no_rgbsrc2 = 0;            // off: same-size 8-bit RGB -> 4:2:0 / 4:2:2 YUV takes the wave-march form (sws_k_rgbsrc_unity2) where it applies
    int no_striprgb_direct = 0;    // off: semi-planar sources (nv12 / p010 families) are read by the strip-RGB kernels themselves instead of through a split pass
    int strip_short_waves = 0;     // experiments: waves per SIMD the short family is banded for (0: strip_waves scaled by the instantiation)
    int max_devices = 0;           // sws_scale_frames(): GPUs to shard over (0 = all visible)
    int work_mb = 2048;            // budget for the helper passes' per-frame working pictures: larger batches are cut into sub-batches (dev_exec.hip launch_plan_le)
    int rccl_tables = 0;           // on: sws_scale_frames() over several GPUs uploads the tables to the home GPU only and ncclBroadcast()s them to the peers (dev_rccl.hip; also SWS_HIP_RCCL=1)
    int dry_plan = 0;              // on: the context plans WITHOUT a GPU -- no HIP call, table blocks get fixed fake addresses, uploads are hashed instead of copied; such a context
                                   //     names its path and digests its plan (sws_hip_plan) and refuses to run.  The planner test of the CPU suite (tests/test_planner_table.py)
    int exp[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };   // "exp0" .. "exp7": free knobs for A/B measurements of a kernel variant under development (no header change per experiment)
    int debug = 0;
};

// What sws_scale_frame() on a dynamic context knows about one field of a frame (libswscale/format.h:79-110 SwsFormat)
struct SwsFmt {
    int width, height, format, hw_format;
    int range, csp, loc, prim, trc;      // enum AVColorRange / AVColorSpace / AVChromaLocation / AVColorPrimaries / AVColorTransferCharacteristic
    int interlaced, field;
    uint64_t side_hash;                  // fingerprint of the mastering-display / HDR10+ side data (compared, never interpreted)
    int hip_device; void *hip_stream;    // AV_PIX_FMT_HIP frames: AVHIPDeviceContext of their device
    const void *device_ref;              // AVHWDeviceContext the frames context belongs to
};
// One field's conversion of a dynamic context (libswscale/graph.h SwsGraph, reduced to what the legacy back-end needs)
struct FrameGraph {
    bool valid = false, noop = false, incomplete = false;
    SwsFmt src{}, dst{};
    SwsContext opts_copy{};
    SwsContext *legacy = nullptr;        // sws_init_context()ed child that does the work
};

struct SwsInternal {
    SwsContext opts;          // MUST be first: the public struct (swscale_internal.h:337-340 idiom)
    uint32_t magic;
    bool legacy_init = false;
    std::vector<double> srcVec[4];   // copies of the SwsFilter vectors given to sws_init_context: lumH, lumV, chrH, chrV
    int dstVecLen[4] = {0, 0, 0, 0};
    const SwsFrameView *frame_src = nullptr; SwsFrameView *frame_dst = nullptr; std::vector<std::pair<unsigned, unsigned>> frame_ranges; bool frame_done = false;   // sws_frame_start .. sws_frame_end: source rows sent so far (ff_range_add), frame converted
    bool srcXYZ = false, dstXYZ = false; // handle_xyz (utils.c:822-842): the caller's formats were xyz12, opts.*_format hold rgb48le
    bool srcBE = false, dstBE = false;   // the caller's formats were big-endian: opts.src_format / dst_format hold the LE twins
    FrameGraph graph[2];          // dynamic mode (sws_alloc_context() only): top / bottom field conversions built by sws_frame_setup()
    int sliceDir = 0;             // 0 = no slice sequence in progress, 1 = top-down, -1 = bottom-up (swscale.c:1096-1104)
    int slice_dstY = 0;           // ff_swscale's dstY cursor (swscale.c:372-381, :566)
    int src0Alpha = 0, dst0Alpha = 0;
    int brightness = 0, contrast = 0, saturation = 0;
    int srcColorspaceTable[4] = {0}, dstColorspaceTable[4] = {0};
    int dstFormatBpp = 0, srcFormatBpp = 0;
    int chrSrcHSubSample = 0, chrSrcVSubSample = 0, chrDstHSubSample = 0, chrDstVSubSample = 0;
    int chrSrcW = 0, chrSrcH = 0, chrDstW = 0, chrDstH = 0;
    int srcBpc = 0, dstBpc = 0;
    int lumXInc = 0, lumYInc = 0, chrXInc = 0, chrYInc = 0;
    int dst_slice_align = 1;
    int needAlpha = 0;
    PlanKind plan = PLAN_NONE;
    FilterBank hLum, hChr, vLum, vChr;
    int32_t rgb2yuv[9] = {0};
    Yuv2RgbLut lut;
    RangeConv range;
    SwsInternal *cascade[3] = {nullptr, nullptr, nullptr};   // [2]: third step of the gamma cascade (RGBA64LE -> destination format)
    int cascade_mainindex = 0;    // the child sws_setColorspaceDetails() is forwarded to (utils.c:909-910): 1 for the alpha-blend cascade
    bool cascade_gamma = false;   // gamma-correct scaling (utils.c:1461-1522): cascade[1] scales RGBA64 between two in-place table passes
    bool cascade_ed = false;      // 8 / 4 bpp destination with error diffusion: cascade[0] writes rgb24 at the destination size, a diffusion pass follows
    bool mono_y16 = false;        // (inner context of a 1 bpp error-diffusion context) the mono writer stores luma words instead of bits
    bool internal_gamma = false;  // the scaling step of the gamma cascade (is_internal_gamma, utils.c:1493-1497): gamma_convert is its first luma descriptor
    bool gamma_in_reader = false; // (set by dev_prepare_on) the line schedule converts some line more than once: pass 1 applies the table per virtual line
    bool force_scaler = false;    // (inner context of the above) never take an unscaled special converter
    int cascade_fmt = -1, cascade_w = 0, cascade_h = 0;
    std::string path_name, kernel_name;
    DeviceState *dev = nullptr;             // state on the context's home GPU
    std::vector<DeviceState *> peers;       // states on the other GPUs sws_scale_frames() shards over (index = HIP device id, may hold nullptr)
    uint64_t tables_epoch = 1;              // bumped whenever the host tables change: device copies with another epoch are rebuilt
    Tuning tune;
};
inline void mark_tables_dirty(SwsInternal *c) { c->tables_epoch++; }

inline SwsInternal *internal(SwsContext *c) { return reinterpret_cast<SwsInternal *>(c); }
inline const SwsInternal *internal(const SwsContext *c) { return reinterpret_cast<const SwsInternal *>(c); }

int  init_single_context(SwsInternal *c);   // utils.c:1137-1835
void choose_unscaled(SwsInternal *c);       // swscale_unscaled.c:2392-2706
void log_msg(const SwsInternal *c, int level, const char *fmt, ...);

// ---- device side (device.cpp / kernels.hip) ----
int  dev_prepare(SwsInternal *c);                         // upload tables, build DevParams (home GPU)
void dev_release(SwsInternal *c);                         // frees the device state on every GPU
int  dev_run(SwsInternal *c, const uint8_t *const src[4], const int srcStride[4], int srcSliceY, int srcSliceH,
             uint8_t *const dst[4], const int dstStride[4], int nb_frames,
             const SwsFrameView *const *srcFrames, SwsFrameView *const *dstFrames);
struct StreamLoan { void *prev = nullptr; bool prev_own = false; bool active = false; };
int  dev_borrow_stream(SwsInternal *c, void *stream, StreamLoan *loan);   // the frames' hwdevice stream, for one call
void dev_return_stream(SwsInternal *c, const StreamLoan &loan);
int  dev_inherit(SwsInternal *child, SwsInternal *parent, bool have_stream, void *stream, int device = -1);  // a dynamic context's child runs on its GPU / stream / tuning
int  dev_use_stream(SwsInternal *c, void *stream);        // run on the stream of the frames' AVHIPDeviceContext
int  dev_copy_frame(SwsInternal *c, SwsFrameView *dst, const SwsFrameView *src, bool have_stream, void *stream);   // no-op conversion: plane copies
void frames_release(SwsInternal *c);                      // frees the per-field conversions of a dynamic context
bool check_image_pointers(const uint8_t *const data[4], int fmt, const int linesizes[4]);
size_t tables_blob_size(const SwsInternal *c);
int  tables_blob_export(const SwsInternal *c, void *buf, size_t size);
int  tables_blob_import(SwsInternal *c, const void *buf, size_t size);

// the byte order and XYZ-ness of the caller's formats live in flags next to the canonicalised opts.*_format
inline bool src_tags_match(const SwsInternal *c, int fmt) { return c->srcBE == (pix_be_twin(fmt) >= 0) && c->srcXYZ == pix_is_xyz(fmt); }
inline bool dst_tags_match(const SwsInternal *c, int fmt) { return c->dstBE == (pix_be_twin(fmt) >= 0) && c->dstXYZ == pix_is_xyz(fmt); }

} // namespace swship
