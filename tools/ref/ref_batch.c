/* Cross-check of oracle/ against the REAL reference on arbitrary conversions (build container only; links the C-only /tmp build made by tools/ref_vs_port.sh --
 * nothing of it is kept in the repo).  Reads cases from stdin, writes the destination planes to stdout; tools/ref/ref_crosscheck.py drives it.
 *   per case in:  "CASE sw sh srcFmtName dw dh dstFmtName flags prefill  opts dither src_range dst_range shp svp dhp dvp  cs inv srcRange tab dstRange b c s  alpha_blend gamma_flag p0 p1\n"
 *                 (opts 0: sws_getContext(); 1: sws_alloc_context() + the public fields + sws_init_context(); cs 1: sws_setColorspaceDetails() with the seven values),
 *                 then the source planes' visible rows (tight), top to bottom, plane after plane
 *   per case out: "RET <ret> <nbytes>\n" then nbytes of destination planes (visible rows, tight); RET -1 0 when the reference refuses the context
 * Uses only the public API (libswscale/swscale.h:424-457, 522-548) and libavutil's image helpers. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "libswscale/swscale.h"
#include "libavutil/imgutils.h"
#include "libavutil/pixdesc.h"
#include "libavutil/log.h"

static int plane_rows(const AVPixFmtDescriptor *d, enum AVPixelFormat f, int h, int p)
{
    if (f == AV_PIX_FMT_PAL8 && p == 1) return 1;
    return (p == 1 || p == 2) ? AV_CEIL_RSHIFT(h, d->log2_chroma_h) : h;
}

int main(void)
{
    char sfn[64], dfn[64];
    int sw, sh, dw, dh, flags, prefill;
    av_log_set_level(AV_LOG_QUIET);
    int useo, dith, sr, dr, shp, svp, dhp, dvp, usecs, cs[7], ablend, gam;
    double p0, p1;
    while (scanf(" CASE %d %d %63s %d %d %63s %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %lf %lf", &sw, &sh, sfn, &dw, &dh, dfn, &flags, &prefill,
                 &useo, &dith, &sr, &dr, &shp, &svp, &dhp, &dvp, &usecs, &cs[0], &cs[1], &cs[2], &cs[3], &cs[4], &cs[5], &cs[6], &ablend, &gam, &p0, &p1) == 28) {
        getchar();      /* the newline behind the header */
        enum AVPixelFormat sf = av_get_pix_fmt(sfn), df = av_get_pix_fmt(dfn);
        const AVPixFmtDescriptor *sd = av_pix_fmt_desc_get(sf), *dd = av_pix_fmt_desc_get(df);
        uint8_t *src[4] = { 0 }, *dst[4] = { 0 };
        int sls[4] = { 0 }, dls[4] = { 0 }, sbw[4] = { 0 }, dbw[4] = { 0 };
        if (!sd || !dd || av_image_alloc(src, sls, sw, sh, sf, 64) < 0 || av_image_alloc(dst, dls, dw, dh, df, 64) < 0) return 3;
        av_image_fill_linesizes(sbw, sf, sw); av_image_fill_linesizes(dbw, df, dw);
        if (sf == AV_PIX_FMT_PAL8) sbw[1] = 1024;
        if (df == AV_PIX_FMT_PAL8) dbw[1] = 1024;
        for (int p = 0; p < 4 && src[p]; p++)
            for (int y = 0; y < plane_rows(sd, sf, sh, p); y++)
                if (fread(src[p] + (long)y * sls[p], 1, sbw[p], stdin) != (size_t)sbw[p]) return 4;
        for (int p = 0; p < 4 && dst[p]; p++) memset(dst[p], prefill, (size_t)dls[p] * plane_rows(dd, df, dh, p));
        SwsContext *c;
        if (!useo) c = sws_getContext(sw, sh, sf, dw, dh, df, flags, NULL, NULL, NULL);
        else {
            c = sws_alloc_context();
            c->src_w = sw; c->src_h = sh; c->src_format = sf; c->dst_w = dw; c->dst_h = dh; c->dst_format = df; c->flags = (unsigned)flags;
            c->dither = dith; c->src_range = sr; c->dst_range = dr; c->src_h_chr_pos = shp; c->src_v_chr_pos = svp; c->dst_h_chr_pos = dhp; c->dst_v_chr_pos = dvp;
            c->threads = 1; c->alpha_blend = ablend; c->gamma_flag = gam; c->scaler_params[0] = p0; c->scaler_params[1] = p1;
            if (sws_init_context(c, NULL, NULL) < 0) { sws_freeContext(c); c = NULL; }
        }
        if (c && usecs && sws_setColorspaceDetails(c, sws_getCoefficients(cs[0]), cs[1], sws_getCoefficients(cs[2]), cs[3], cs[4], cs[5], cs[6]) < 0) { sws_freeContext(c); c = NULL; }
        if (!c) { printf("RET -1 0\n"); fflush(stdout); }
        else {
            const int ret = sws_scale(c, (const uint8_t *const *)src, sls, 0, sh, dst, dls);
            long n = 0;
            for (int p = 0; p < 4 && dst[p]; p++) n += (long)dbw[p] * plane_rows(dd, df, dh, p);
            printf("RET %d %ld\n", ret, n);
            for (int p = 0; p < 4 && dst[p]; p++)
                for (int y = 0; y < plane_rows(dd, df, dh, p); y++) fwrite(dst[p] + (long)y * dls[p], 1, dbw[p], stdout);
            fflush(stdout);
            sws_freeContext(c);
        }
        av_freep(&src[0]); av_freep(&dst[0]);
    }
    return 0;
}
