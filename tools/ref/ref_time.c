/* Times the REAL reference's C path (a C-only, --disable-asm build of /root/reference made by tools/ref_vs_port.sh under /tmp; nothing
 * of that build is kept in the repo) on the BASELINE configurations, one thread, best of N -- the denominator of bench.py's
 * "port / reference" ratios.  Uses only the public API (libswscale/swscale.h:424-457, 522-548).
 *   usage: ref_time <name> <srcW> <srcH> <srcFmtName> <dstW> <dstH> <dstFmtName> <flags> <bt2020 0|1> <reps> [source planes file] */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "libswscale/swscale.h"
#include "libavutil/imgutils.h"
#include "libavutil/pixdesc.h"
#include "libavutil/mem.h"

static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

int main(int argc, char **argv)
{
    if (argc < 11) return 2;
    int sw = atoi(argv[2]), sh = atoi(argv[3]), dw = atoi(argv[5]), dh = atoi(argv[6]), flags = atoi(argv[8]), bt2020 = atoi(argv[9]), reps = atoi(argv[10]);
    enum AVPixelFormat sf = av_get_pix_fmt(argv[4]), df = av_get_pix_fmt(argv[7]);
    uint8_t *src[4], *dst[4]; int sls[4], dls[4];
    if (av_image_alloc(src, sls, sw, sh, sf, 64) < 0 || av_image_alloc(dst, dls, dw, dh, df, 64) < 0) return 3;
    const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(sf);
    unsigned s = 12345;
    for (int p = 0; p < 4 && src[p]; p++) {
        int rows = (p == 1 || p == 2) ? AV_CEIL_RSHIFT(sh, d->log2_chroma_h) : sh;
        if (d->flags & AV_PIX_FMT_FLAG_FLOAT) { float *f = (float *)src[p]; for (long i = 0; i < (long)sls[p] * rows / 4; i++) { s = s * 1664525u + 1013904223u; f[i] = (s >> 8) / 16777216.0f * 1.5f - 0.25f; } }
        else if (d->comp[0].depth > 8) { uint16_t *w = (uint16_t *)src[p]; for (long i = 0; i < (long)sls[p] * rows / 2; i++) { s = s * 1664525u + 1013904223u; w[i] = (s >> 16) & ((1 << d->comp[0].depth) - 1); } }
        else for (long i = 0; i < (long)sls[p] * rows; i++) { s = s * 1664525u + 1013904223u; src[p][i] = s >> 24; }
    }
    if (argc > 11) {   /* the SAME picture the port is timed on: planes as tools/ref/ref_vs_port.py wrote them (visible bytes of each row, top to bottom) */
        FILE *f = fopen(argv[11], "rb");
        if (!f) return 5;
        int ls[4]; av_image_fill_linesizes(ls, sf, sw);
        for (int p = 0; p < 4 && src[p]; p++) {
            int rows = (p == 1 || p == 2) ? AV_CEIL_RSHIFT(sh, d->log2_chroma_h) : sh;
            for (int y = 0; y < rows; y++) if (fread(src[p] + (long)y * sls[p], 1, ls[p], f) != (size_t)ls[p]) return 6;
        }
        fclose(f);
    }
    SwsContext *c = sws_getContext(sw, sh, sf, dw, dh, df, flags, NULL, NULL, NULL);
    if (!c) return 4;
    if (bt2020) sws_setColorspaceDetails(c, sws_getCoefficients(SWS_CS_BT2020), 1, sws_getCoefficients(SWS_CS_BT2020), 1, 0, 1 << 16, 1 << 16);
    sws_scale(c, (const uint8_t *const *)src, sls, 0, sh, dst, dls);
    double best = 1e30;
    for (int r = 0; r < reps; r++) { double t0 = now_ms(); sws_scale(c, (const uint8_t *const *)src, sls, 0, sh, dst, dls); double t = now_ms() - t0; if (t < best) best = t; }
    printf("%s %.3f\n", argv[1], best);
    return 0;
}
