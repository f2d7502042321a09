"""What the reference's dynamic sws_scale_frame() path makes of a frame's properties, restated for the tests.

Test infrastructure (like oracle/): turns (pixel format, AVFrame colour properties, field) into the explicit legacy-scaler
configuration libswscale/graph.c:558-661 add_legacy_sws_pass() builds, so that the oracle (which only knows the explicit
sws_alloc_context() + fields + sws_init_context() + sws_setColorspaceDetails() construction) can be asked for the same conversion.

  sanitize()   libswscale/format.c:305-342 sanitize_fmt
  chroma_pos() libswscale/format.c:554-592 ff_sws_chroma_pos, libavutil/pixdesc.c:3902-3912 av_chroma_location_enum_to_pos
  legacy_config() graph.c:596-657
"""
import numpy as np

import oracle_lib as OL

RANGE = {"unspecified": 0, "mpeg": 1, "jpeg": 2}
LOC = {"unspecified": 0, "left": 1, "center": 2, "topleft": 3, "top": 4, "bottomleft": 5, "bottom": 6}
CSP = {"rgb": 0, "bt709": 1, "unspecified": 2, "fcc": 4, "bt470bg": 5, "smpte170m": 6, "smpte240m": 7, "bt2020nc": 9}


def _is_rgb(fmt):
    return fmt.startswith(("rgb", "bgr", "gbr", "argb", "abgr", "0rgb", "0bgr", "x2rgb", "x2bgr", "pal8", "bayer_"))   # AV_PIX_FMT_FLAG_RGB | PAL | BAYER


def _is_gray(fmt):
    return fmt.startswith(("gray", "ya", "mono"))


def _subsampling(fmt):
    """(log2_chroma_w, log2_chroma_h)"""
    if _is_rgb(fmt) or _is_gray(fmt) or fmt.startswith("xyz"):
        return 0, 0
    for key, sub in (("p41", (0, 0)), ("420", (1, 1)), ("nv12", (1, 1)), ("nv21", (1, 1)), ("p010", (1, 1)), ("p012", (1, 1)), ("p016", (1, 1)),
                     ("422", (1, 0)), ("nv16", (1, 0)), ("nv20", (1, 0)), ("p21", (1, 0)), ("yuyv", (1, 0)), ("uyvy", (1, 0)),
                     ("yvyu", (1, 0)), ("y21", (1, 0)), ("411", (2, 0)), ("410", (2, 2)), ("440", (0, 1))):
        if key in fmt:
            return sub
    return 0, 0


def sanitize(fmt, props):
    p = {"color_range": 0, "colorspace": 2, "chroma_location": 0}
    for k, v in (props or {}).items():
        if k in p:
            p[k] = {"color_range": RANGE, "colorspace": CSP, "chroma_location": LOC}[k][v] if isinstance(v, str) else v
    if _is_rgb(fmt):
        p["colorspace"], p["color_range"] = 0, 2
    elif fmt.startswith("xyz"):
        p["colorspace"] = 2
    elif _is_gray(fmt):
        p["colorspace"] = 2
        p["color_range"] = 0 if ("f32" in fmt or "f16" in fmt) else 2
    if fmt.startswith("yuvj"):
        p["color_range"] = 2
    if _subsampling(fmt) == (0, 0):
        p["chroma_location"] = 0
    return p


def chroma_pos(fmt, loc, interlaced=False, field=0):
    sub_x, sub_y = _subsampling(fmt)
    if loc == 0:
        loc = 2
    pos = loc - 1
    x, y = (pos & 1) * 128, ((pos >> 1) ^ (1 if pos < 4 else 0)) * 128
    x *= (1 << sub_x) - 1
    y *= (1 << sub_y) - 1
    if sub_y and interlaced:
        if field == 1:
            y += (256 << sub_y) - 256
        y >>= 1
    return (x if sub_x else -513), (y if sub_y else -513)


def legacy_config(sfmt, sprops, dfmt, dprops, interlaced=False, field=0, overrides=None):
    """-> (oracle keyword options, (inv_cs, src_range, cs, dst_range)) of the conversion the reference builds for this field"""
    s, d = sanitize(sfmt, sprops), sanitize(dfmt, dprops)
    shx, svx = chroma_pos(sfmt, s["chroma_location"], interlaced, field)
    dhx, dvx = chroma_pos(dfmt, d["chroma_location"], interlaced, field)
    opts = dict(src_range=int(s["color_range"] == 2), dst_range=int(d["color_range"] == 2),
                src_h_chr_pos=shx, src_v_chr_pos=svx, dst_h_chr_pos=dhx, dst_v_chr_pos=dvx)
    for k, v in (overrides or {}).items():        # legacy_chr_pos(): the context's own *_chr_pos fields win, except where stripped
        if v != -513 and opts[k] != -513:
            opts[k] = v
    return opts, (s["colorspace"], opts["src_range"], d["colorspace"], opts["dst_range"])


class FieldView:
    """one field of an oracle_lib.Frame (graph.c:998-1024 get_field): every second row, doubled stride"""

    def __init__(self, frame, field):
        self.fmt, self.w = frame.fmt, frame.w
        self.planes = [a[field::2] for a in frame.planes]
        self.h = (frame.h + (field == 0)) >> 1
        self.row_bytes = frame.row_bytes

    def ptrs(self):
        import ctypes as C
        p = (C.c_void_p * 4)()
        s = (C.c_int * 4)()
        for i, a in enumerate(self.planes):
            p[i] = a.ctypes.data
            s[i] = a.strides[0]
        return p, s


def oracle_convert(src, sprops, dst, dprops, flags, interlaced=False, overrides=None, **ctx_opts):
    """run the oracle the way the reference's dynamic path would: one explicit conversion per field"""
    for field in range(2 if interlaced else 1):
        sv = FieldView(src, field) if interlaced else src
        dv = FieldView(dst, field) if interlaced else dst
        opts, cs = legacy_config(src.fmt, sprops, dst.fmt, dprops, interlaced, field, overrides)
        opts.update(ctx_opts)
        o = OL.Oracle(src.w, sv.h, src.fmt, dst.w, dv.h, dst.fmt, flags, **opts)
        o.set_colorspace(*cs)
        sp, ss = sv.ptrs()
        dp, ds = dv.ptrs()
        assert OL.lib().or_sws_scale(o.c, sp, ss, 0, sv.h, dp, ds) == dv.h
    return dst
