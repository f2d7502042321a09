// Table blob: everything the host-side init derived for a context, packed into one flat buffer.
// sws_scale_frames() across GPUs: rank 0 runs sws_getContext() (+ sws_setColorspaceDetails()), exports
// the blob, the blob is broadcast once (RCCL over xGMI, a few hundred KB at most), the other ranks
// import it into an sws_alloc_context() shell.  No per-frame collective exists on this path.
#include <cstring>

#include "swsint.hpp"

namespace swship {

namespace {
struct Writer {
    uint8_t *p; size_t cap, off = 0; bool dry;
    void put(const void *src, size_t n) { if (!dry && off + n <= cap) std::memcpy(p + off, src, n); off += n; }
    template <typename T> void pod(const T &v) { put(&v, sizeof(T)); }
    template <typename T> void vec(const std::vector<T> &v) { uint64_t n = v.size(); pod(n); if (n) put(v.data(), n * sizeof(T)); }
};
struct Reader {
    const uint8_t *p; size_t cap, off = 0; bool ok = true;
    void get(void *dst, size_t n) { if (off + n > cap) { ok = false; return; } std::memcpy(dst, p + off, n); off += n; }
    template <typename T> void pod(T &v) { get(&v, sizeof(T)); }
    template <typename T> void vec(std::vector<T> &v) { uint64_t n = 0; pod(n); if (!ok || n > (1u << 28)) { ok = false; return; } v.resize(n); if (n) get(v.data(), n * sizeof(T)); }
};
const uint32_t kBlobMagic = 0x53574254; // 'SWBT'
const uint32_t kBlobVersion = 2;        // layout version of the records below; exporter and importer must be the same build family

void write_ctx(Writer &w, const SwsInternal *c)
{
    w.pod(kBlobMagic);
    const uint32_t hdr[4] = { kBlobVersion, (uint32_t)sizeof(SwsContext), (uint32_t)sizeof(Yuv2RgbLut), (uint32_t)sizeof(RangeConv) };
    w.put(hdr, sizeof(hdr));
    w.pod(c->opts);
    int32_t ints[] = { c->src0Alpha, c->dst0Alpha, c->brightness, c->contrast, c->saturation, c->dstFormatBpp, c->srcFormatBpp,
                       c->chrSrcHSubSample, c->chrSrcVSubSample, c->chrDstHSubSample, c->chrDstVSubSample,
                       c->chrSrcW, c->chrSrcH, c->chrDstW, c->chrDstH, c->srcBpc, c->dstBpc,
                       c->lumXInc, c->lumYInc, c->chrXInc, c->chrYInc, c->dst_slice_align, c->needAlpha, (int32_t)c->plan,
                       c->cascade_fmt, c->cascade_w, c->cascade_h, c->legacy_init ? 1 : 0, (c->srcBE ? 1 : 0) | (c->srcXYZ ? 2 : 0), (c->dstBE ? 1 : 0) | (c->dstXYZ ? 2 : 0) };
    w.put(ints, sizeof(ints));
    w.put(c->srcColorspaceTable, sizeof(c->srcColorspaceTable));
    w.put(c->dstColorspaceTable, sizeof(c->dstColorspaceTable));
    for (const FilterBank *b : { &c->hLum, &c->hChr, &c->vLum, &c->vChr }) {
        int32_t sc[2] = { b->size, b->count };
        w.put(sc, sizeof(sc)); w.vec(b->taps); w.vec(b->pos);
    }
    w.put(c->rgb2yuv, sizeof(c->rgb2yuv));
    w.pod(c->lut);
    w.pod(c->range);
    uint8_t has[2] = { (uint8_t)(c->cascade[0] != nullptr), (uint8_t)(c->cascade[1] != nullptr) };
    w.put(has, 2);
    for (int i = 0; i < 2; i++) if (c->cascade[i]) write_ctx(w, c->cascade[i]);
}

bool read_ctx(Reader &r, SwsInternal *c)
{
    uint32_t magic = 0; r.pod(magic);
    if (!r.ok || magic != kBlobMagic) return false;
    uint32_t hdr[4] = { 0, 0, 0, 0 };
    r.get(hdr, sizeof(hdr));
    if (!r.ok || hdr[0] != kBlobVersion || hdr[1] != sizeof(SwsContext) || hdr[2] != sizeof(Yuv2RgbLut) || hdr[3] != sizeof(RangeConv)) return false;
    const void *cls = c->opts.av_class; void *opaque = c->opts.opaque;
    r.pod(c->opts);
    c->opts.av_class = cls; c->opts.opaque = opaque;   // process-local pointers are not transported
    int32_t ints[30];
    r.get(ints, sizeof(ints));
    int k = 0;
    c->src0Alpha = ints[k++]; c->dst0Alpha = ints[k++]; c->brightness = ints[k++]; c->contrast = ints[k++]; c->saturation = ints[k++];
    c->dstFormatBpp = ints[k++]; c->srcFormatBpp = ints[k++];
    c->chrSrcHSubSample = ints[k++]; c->chrSrcVSubSample = ints[k++]; c->chrDstHSubSample = ints[k++]; c->chrDstVSubSample = ints[k++];
    c->chrSrcW = ints[k++]; c->chrSrcH = ints[k++]; c->chrDstW = ints[k++]; c->chrDstH = ints[k++]; c->srcBpc = ints[k++]; c->dstBpc = ints[k++];
    c->lumXInc = ints[k++]; c->lumYInc = ints[k++]; c->chrXInc = ints[k++]; c->chrYInc = ints[k++]; c->dst_slice_align = ints[k++];
    c->needAlpha = ints[k++]; c->plan = (PlanKind)ints[k++]; c->cascade_fmt = ints[k++]; c->cascade_w = ints[k++]; c->cascade_h = ints[k++];
    c->legacy_init = ints[k++] != 0;
    c->srcBE = (ints[k] & 1) != 0; c->srcXYZ = (ints[k++] & 2) != 0; c->dstBE = (ints[k] & 1) != 0; c->dstXYZ = (ints[k++] & 2) != 0;
    r.get(c->srcColorspaceTable, sizeof(c->srcColorspaceTable));
    r.get(c->dstColorspaceTable, sizeof(c->dstColorspaceTable));
    for (FilterBank *b : { &c->hLum, &c->hChr, &c->vLum, &c->vChr }) {
        int32_t sc[2] = { 0, 0 };
        r.get(sc, sizeof(sc)); b->size = sc[0]; b->count = sc[1]; r.vec(b->taps); r.vec(b->pos);
    }
    r.get(c->rgb2yuv, sizeof(c->rgb2yuv));
    r.pod(c->lut);
    r.pod(c->range);
    uint8_t has[2] = { 0, 0 };
    r.get(has, 2);
    if (!r.ok) return false;
    // a truncated or foreign blob must not turn into out-of-bounds reads in the kernels or in scale_slice(): check every invariant
    // the launch planner relies on
    if ((int)c->plan < PLAN_NONE || (int)c->plan > PLAN_CASCADE) return false;
    if (!pix_desc(c->opts.src_format) || !pix_desc(c->opts.dst_format)) return false;
    if (c->opts.src_w < 1 || c->opts.src_h < 1 || c->opts.dst_w < 1 || c->opts.dst_h < 1) return false;
    if (c->plan == PLAN_MAIN) {
        struct { const FilterBank *b; int count, src; bool horizontal; } want[4] = {
            { &c->hLum, c->opts.dst_w, c->opts.src_w, true }, { &c->hChr, c->chrDstW, c->chrSrcW, true },
            { &c->vLum, c->opts.dst_h, c->opts.src_h, false }, { &c->vChr, c->chrDstH, c->chrSrcH, false } };
        if (c->chrSrcW < 1 || c->chrSrcH < 1 || c->chrDstW < 1 || c->chrDstH < 1) return false;
        for (const auto &e : want) {
            const FilterBank &b = *e.b;
            if (b.size < 1 || b.size > 4096 || b.count != e.count) return false;
            if (b.taps.size() < (size_t)b.size * (size_t)b.count || b.pos.size() < (size_t)b.count) return false;
            for (int i = 0; i < b.count; i++) {
                const int pos = b.pos[(size_t)i];
                // horizontal windows are read unclamped: inside the line (initFilter folds border taps, utils.c:519-560); vertical
                // windows are clamped row by row (max(1 - size, pos), min(row, last)): they only have to touch the picture
                if (e.horizontal ? (pos < 0 || pos + b.size > e.src) : (pos < 1 - b.size || pos > e.src - 1)) return false;
            }
        }
    }
    if (c->plan == PLAN_CASCADE && (!has[0] || !has[1] || !pix_desc(c->cascade_fmt) || c->cascade_w < 1 || c->cascade_h < 1)) return false;
    for (int i = 0; i < 2; i++) {
        if (c->cascade[i]) { sws_freeContext(&c->cascade[i]->opts); c->cascade[i] = nullptr; }
        if (has[i]) {
            SwsContext *child = sws_alloc_context();
            if (!child) return false;
            c->cascade[i] = internal(child);
            if (!read_ctx(r, c->cascade[i])) return false;
        }
    }
    mark_tables_dirty(c);
    return r.ok;
}
} // namespace

// (a gamma cascade is three contexts and two tables, an error-diffusion context carries a running error line: every rank builds those
// itself, they are not shipped)
size_t tables_blob_size(const SwsInternal *c) { if (c->cascade_gamma || c->cascade_ed) return 0; Writer w{nullptr, 0, 0, true}; write_ctx(w, c); return w.off; }
int tables_blob_export(const SwsInternal *c, void *buf, size_t size)
{
    if (c->cascade_gamma || c->cascade_ed) return -95;   // AVERROR(ENOTSUP)
    Writer w{(uint8_t *)buf, size, 0, false};
    write_ctx(w, c);
    return w.off <= size ? 0 : -22;
}
int tables_blob_import(SwsInternal *c, const void *buf, size_t size)
{
    Reader r{(const uint8_t *)buf, size};
    return read_ctx(r, c) ? 0 : -22;
}

} // namespace swship

extern "C" {
size_t sws_hip_tables_size(const SwsContext *c) { return c ? swship::tables_blob_size(swship::internal(c)) : 0; }
int sws_hip_tables_export(const SwsContext *c, void *buf, size_t size) { return c && buf ? swship::tables_blob_export(swship::internal(c), buf, size) : -22; }
int sws_hip_tables_import(SwsContext *c, const void *buf, size_t size) { return c && buf ? swship::tables_blob_import(swship::internal(c), buf, size) : -22; }
}
