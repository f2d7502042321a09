// Translation unit of the generic element-per-thread path: every reader (input.c), hScale*_c, every writer (output.c) in their
// _1 / _2 / _X forms.  Two passes through an HBM scratch, or one pass when both horizontal banks are the identity.
#include "generic_kinds.hpp"
#include "kernels_generic.hpp"

namespace swship {

// the kernels of one source kind (k_generic_kinds.hip); null: the all-kinds forms below
const GenericKindFns *generic_kind_fns(int srcKind)
{
    static GenericKindFns tab[SRCK_RGB16 + 1];
    static const bool once = [] {
#define GENERIC_KIND_FILL(n) generic_kind_fns_##n(&tab[n]);
        GENERIC_KIND_PARTS(GENERIC_KIND_FILL)
#undef GENERIC_KIND_FILL
        return true; }();
    (void)once;
    if (srcKind < 0 || srcKind > SRCK_RGB16 || srcKind == SRCK_BAYER) return nullptr;
    return &tab[srcKind];
}

static const GenericDstFns *dst_fns(int dstKind)
{
    static GenericDstFns tab[DSTK_RAW32];
    static const bool once = [] {
#define GENERIC_DST_FILL(n) generic_dst_fns_##n(&tab[n]);
        GENERIC_DST_PARTS(GENERIC_DST_FILL)
#undef GENERIC_DST_FILL
        return true; }();
    (void)once;
    if (dstKind < 0 || dstKind >= DSTK_RAW32) return nullptr;
    return &tab[dstKind];
}

static __global__ void sws_k_sum_table(int16_t *taps, int32_t *pos, int rows)
{
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= rows) return;
    taps[2 * y] = 1; taps[2 * y + 1] = 0;   // (two taps that do not add up to 4096: packed_vscale's X form, vscale.c:135-157)
    pos[y] = y;
}

size_t sum_writer_table_bytes(int dstH) { return (((size_t)4 * dstH + 255) & ~(size_t)255) + (((size_t)4 * dstH + 255) & ~(size_t)255); }

// the packed writer of `dst_kind` over the strip kernels' sum planes (k_generic_dst.hip sws_k_sum_writer): the context's parameter block with the real
// destination kind and a vertical bank of the taps {1, 0} at position y over "source" pictures of the destination's height -- the sum planes
int launch_sum_writer(const LaunchCtx &J, int dst_kind, uint8_t *tab)
{
    const SwsDevParams &p = *J.p;
    const GenericDstFns *kd = dst_fns(dst_kind);
    if (!kd || !kd->sum_writer) { log_msg(J.c, 0, "internal error: no sum writer for destination kind %d\n", dst_kind); return SWS_AVERROR(EINVAL); }
    int16_t *taps = (int16_t *)tab;
    int32_t *pos = (int32_t *)(tab + (((size_t)4 * p.dstH + 255) & ~(size_t)255));
    hipLaunchKernelGGL(sws_k_sum_table, dim3(cdiv(p.dstH, 256)), dim3(256), 0, J.st, taps, pos, p.dstH);
    SwsDevParams pe = p;
    pe.dstKind = dst_kind;
    pe.vLumF = pe.vChrF = taps; pe.vLumPos = pe.vChrPos = pos; pe.vLumFs = pe.vChrFs = 2;
    pe.srcH = p.dstH; pe.chrSrcH = p.chrDstH;
    pe.need_alpha = 0;
    const int units = p.full_chr ? p.dstW : (p.dstW + 1) >> 1;
    hipLaunchKernelGGL(kd->sum_writer, dim3(cdiv(units, 256), p.dstH, J.n), dim3(256), 0, J.st, J.fs, pe);
    return 0;
}

int launch_generic(const LaunchCtx &L)
{
    SwsInternal *c = L.c; DeviceState *d = L.d; const SwsDevParams &p = *L.p; hipStream_t st = L.st; const SwsFrameSet &fs = L.fs;
    const SwsFramePtrs *frames = L.frames; const int n = L.n, sliceY = L.sliceY, sliceH = L.sliceH; const bool vec = L.vec;
    const dim3 blk(256);
    (void)c; (void)d; (void)frames; (void)vec; (void)sliceY; (void)sliceH;
    const bool rgb = p.dstKind == DSTK_RGB24 || p.dstKind == DSTK_RGB32 || p.dstKind == DSTK_GBRP || p.dstKind == DSTK_GBRP16 ||
                     p.dstKind == DSTK_GBRPF32 || p.dstKind == DSTK_PACKED422 || p.dstKind == DSTK_PACKED444 || p.dstKind == DSTK_PACKEDHI ||
                     p.dstKind == DSTK_RGB48 || p.dstKind == DSTK_RGB16 || p.dstKind == DSTK_RGB30 || p.dstKind == DSTK_MONO ||
                     p.dstKind == DSTK_RGB8 || p.dstKind == DSTK_RGB4 ||
                     p.dstKind == DSTK_YA;   // all written by sws_k_vscale_rgb (packed_vscale / any_vscale)
        const int64_t lumElems = (int64_t)p.srcH * p.dstW, chrElems = (int64_t)p.chrSrcH * p.chrDstW;
        const int64_t frame_elems = lumElems + 2 * chrElems + (p.need_alpha ? lumElems : 0);
        const size_t esz = p.wide ? 4 : 2;
        const bool direct = d->unity_h;
        const GenericKindFns *ks = c->tune.no_generic_kinds ? nullptr : generic_kind_fns(p.srcKind);
        const GenericKindFns::Direct *kf = (ks && direct && p.dstKind >= 0 && p.dstKind <= DSTK_RAW32) ? &ks->direct[p.dstKind] : nullptr;
        const GenericDstFns *kd = (direct || c->tune.no_generic_kinds) ? nullptr : dst_fns(p.dstKind);
#define KIND_MEMBER_sws_k_vscale_rgb rgb
#define KIND_MEMBER_sws_k_vscale_planar planar
#define KIND_MEMBER_sws_k_vscale_nvchroma nvchroma
#define KIND_MEMBER16_sws_k_vscale_rgb rgb16
#define KIND_MEMBER32_sws_k_vscale_rgb rgb32
#define KIND_MEMBER16_sws_k_vscale_planar planar16
#define KIND_MEMBER32_sws_k_vscale_planar planar32
#define KIND_MEMBER16_sws_k_vscale_nvchroma nvchroma16
#define KIND_MEMBER32_sws_k_vscale_nvchroma nvchroma32
        int chunk = n;
        if (!direct) {
            const size_t budget = (size_t)2 << 30; // scratch budget per launch group
            chunk = (int)std::max<int64_t>(1, std::min<int64_t>(n, (int64_t)(budget / (frame_elems * esz))));
            int r = grow(c, &d->scratch, &d->scratch_bytes, (size_t)frame_elems * esz * chunk);
            if (r < 0) return r;
        }
        for (int f0 = 0; f0 < n; f0 += chunk) {
            const int m = std::min(chunk, n - f0);
            SwsFrameSet sub = fs;
            sub.count = m;
            if (n == 1) sub.one = frames[0]; else sub.table = fs.table + f0;
            if (!direct) {
                const int maxW = std::max(p.dstW, p.chrDstW), maxH = std::max(p.srcH, p.chrSrcH);
                const dim3 g1(cdiv(maxW, 256), maxH, (p.need_alpha ? 4 : 3) * m);
                if (p.wide) hipLaunchKernelGGL(ks ? ks->hscale32 : swsk::sws_k_hscale<int32_t>, g1, blk, 0, st, sub, p, (int32_t *)d->scratch, frame_elems);
                else hipLaunchKernelGGL(ks ? ks->hscale16 : swsk::sws_k_hscale<int16_t>, g1, blk, 0, st, sub, p, (int16_t *)d->scratch, frame_elems);
            }
#define LAUNCH_W(K, G, ...) do { \
    if (direct && kf && kf->KIND_MEMBER_##K) hipLaunchKernelGGL(kf->KIND_MEMBER_##K, G, blk, 0, st, sub, p, (const int16_t *)nullptr, frame_elems, ##__VA_ARGS__); \
    else if (direct) hipLaunchKernelGGL((swsk::K<true, int16_t>), G, blk, 0, st, sub, p, (const int16_t *)nullptr, frame_elems, ##__VA_ARGS__); \
    else if (kd && p.wide && kd->KIND_MEMBER32_##K) hipLaunchKernelGGL(kd->KIND_MEMBER32_##K, G, blk, 0, st, sub, p, (const int32_t *)d->scratch, frame_elems, ##__VA_ARGS__); \
    else if (kd && !p.wide && kd->KIND_MEMBER16_##K) hipLaunchKernelGGL(kd->KIND_MEMBER16_##K, G, blk, 0, st, sub, p, (const int16_t *)d->scratch, frame_elems, ##__VA_ARGS__); \
    else if (p.wide) hipLaunchKernelGGL((swsk::K<false, int32_t>), G, blk, 0, st, sub, p, (const int32_t *)d->scratch, frame_elems, ##__VA_ARGS__); \
    else hipLaunchKernelGGL((swsk::K<false, int16_t>), G, blk, 0, st, sub, p, (const int16_t *)d->scratch, frame_elems, ##__VA_ARGS__); } while (0)
            if (rgb) {
                const int units = p.dstKind == DSTK_MONO ? (p.dstW + 7) >> 3 : (p.full_chr || p.dstKind == DSTK_YA) ? p.dstW : (p.dstW + 1) >> 1;
                const dim3 g(cdiv(units, 256), p.dstH, m);
                LAUNCH_W(sws_k_vscale_rgb, g);
            } else if (p.dstKind == DSTK_NV12 || p.dstKind == DSTK_P010 || p.dstKind == DSTK_P016) {
                const dim3 gl(cdiv(p.dstW, 256), p.dstH, m);
                LAUNCH_W(sws_k_vscale_planar, gl, 1);
                const dim3 gc(cdiv(p.chrDstW, 256), p.chrDstH, m);
                LAUNCH_W(sws_k_vscale_nvchroma, gc);
            } else {
                const int ncomp = isGray(c->opts.dst_format) ? 1 : (p.need_alpha ? 4 : 3); // vscale.c:219-233 (gray: luma only), :59-71 (alpha)
                const dim3 g(cdiv(std::max(p.dstW, p.chrDstW), 256), std::max(p.dstH, p.chrDstH), ncomp * m);
                LAUNCH_W(sws_k_vscale_planar, g, ncomp);
            }
#undef LAUNCH_W
        }
    return 0;
}

} // namespace swship
