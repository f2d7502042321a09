"""The conversions whose PLAN the CPU suite pins (tests/test_planner_table.py, tests/golden/planner_table.txt.xz): a planner-rule interaction -- a
format pair, size class or flag sent down another path, a table built differently -- shows up on the GPU-less box as a changed line, before a
random hunt on the GPU finds the wrong pixels (the rules restated: libswscale/utils.c:1137-1835, vscale.c:109-171).

Two families: EVERY pair of a 44-format set (one or two members of every reader / writer kind of devparams.h) at three size classes with the
default flags, and a 12-format core at every size class under the flag sets callers pass."""
import librempeg_amd as A

BX = A.SWS_BITEXACT
# one or two members of every source / destination kind
FORMATS = ["yuv420p", "yuvj420p", "yuv422p", "yuv444p", "yuv410p", "yuva420p", "yuv420p10le", "yuv422p10le", "yuv444p12le", "yuv420p16le", "yuv444p16le",
           "yuv420p10be", "nv12", "nv21", "nv16", "nv24", "p010le", "p016le", "p210le", "gray", "gray10le", "gray16le", "ya8", "yuyv422", "uyvy422",
           "rgb24", "bgr24", "rgba", "bgra", "bgr0", "rgb565le", "rgb555le", "x2rgb10le", "rgb48le", "rgba64le", "gbrp", "gbrap", "gbrp10le", "gbrp16le",
           "gbrpf32le", "ayuv", "vuya", "y210le", "xv30le", "xv36le", "rgb8", "pal8", "monob", "xyz12le"]
CORE = ["yuv420p", "nv12", "yuv422p", "yuv444p", "yuv420p10le", "p010le", "yuyv422", "rgb24", "bgra", "gbrp", "gray", "yuva420p"]
# (name, srcW, srcH, dstW, dstH)
SIZES_ALL = [("same1080", 1920, 1080, 1920, 1080), ("down4k", 3840, 2160, 1920, 1080), ("up720", 1280, 720, 1920, 1080)]
SIZES_CORE = SIZES_ALL + [("small", 64, 36, 128, 72), ("tiny", 18, 10, 6, 33), ("odd", 1366, 768, 1282, 720), ("oddw", 641, 361, 321, 181),
                          ("ratio12", 1920, 1080, 160, 90), ("c1", 1280, 720, 640, 360), ("vonly", 1920, 1080, 1920, 540), ("honly", 1920, 1080, 960, 1080)]
FLAGS_CORE = [("bicubic", A.SWS_BICUBIC | BX), ("bilinear", A.SWS_BILINEAR | BX), ("fastbilinear", A.SWS_FAST_BILINEAR), ("point", A.SWS_POINT | BX),
              ("lanczos", A.SWS_LANCZOS | BX), ("bicubic+accurate", A.SWS_BICUBIC | BX | A.SWS_ACCURATE_RND),
              ("bicubic+fullchr", A.SWS_BICUBIC | BX | A.SWS_FULL_CHR_H_INT), ("area", A.SWS_AREA | BX)]


def cases():
    """[(key, srcW, srcH, srcFmt, dstW, dstH, dstFmt, flags)], in a fixed order"""
    out, seen = [], set()

    def add(sz, sf, df, fl):
        key = f"{sf}>{df}|{sz[0]}|{fl[0]}"
        if key not in seen:
            seen.add(key)
            out.append((key, sz[1], sz[2], sf, sz[3], sz[4], df, fl[1]))
    for sz in SIZES_ALL:
        for sf in FORMATS:
            for df in FORMATS:
                add(sz, sf, df, FLAGS_CORE[0])
    for sz in SIZES_CORE:
        for fl in FLAGS_CORE:
            for sf in CORE:
                for df in CORE:
                    add(sz, sf, df, fl)
    return out


def plan_line(L, key, sw, sh, sf, dw, dh, df, flags):
    """one line of the table: what the planner makes of a conversion, without a GPU (option dry_plan)"""
    import ctypes as C
    from librempeg_amd import swscale as S
    try:
        ctx = S.SwsContext(sw, sh, sf, dw, dh, df, flags)
    except Exception:
        return f"{key} refused"
    try:
        if ctx.set_option("dry_plan", 1) != 0:
            return f"{key} no-dry-plan"
        dg = (C.c_uint64 * 3)()
        r = L.sws_hip_plan(ctx.c, dg)
        if r < 0:
            return f"{key} plan-error {r}"
        return f"{key} {ctx.path()} {ctx.kernel_name() or '-'} {dg[0]:016x} {dg[1]:016x} {dg[2]:016x}"
    finally:
        ctx.close()
