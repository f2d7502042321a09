"""-m gpu parity on randomly drawn conversions: random format pairs, sizes (1..260 in each direction, independent ratios), scalers and
flag combinations; seeded, so every run draws the same cases.  Complements the fixed-size format matrix of test_gpu_parity.py."""
import random

import pytest

import oracle_lib as OL
from librempeg_amd import (SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_BITEXACT, SWS_ACCURATE_RND, SWS_POINT, SWS_AREA, SWS_GAUSS,
                           SWS_SPLINE, SWS_FULL_CHR_H_INT, SWS_FAST_BILINEAR)
from test_gpu_parity import run_case, FORMAT_MATRIX_SRC, FORMAT_MATRIX_DST

pytestmark = pytest.mark.gpu

SCALERS = [SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS, SWS_POINT, SWS_AREA, SWS_GAUSS, SWS_SPLINE, SWS_FAST_BILINEAR]
EXTRA = [0, SWS_BITEXACT, SWS_ACCURATE_RND, SWS_BITEXACT | SWS_ACCURATE_RND, SWS_FULL_CHR_H_INT, SWS_FULL_CHR_H_INT | SWS_ACCURATE_RND]


def _cases(n, seed):
    rng = random.Random(seed)
    out = []
    for k in range(n):
        sf, df = rng.choice(FORMAT_MATRIX_SRC), rng.choice(FORMAT_MATRIX_DST)
        mode = rng.random()
        if mode < 0.25:       # same size: the unscaled converters and their ragged edges
            sw = dw = rng.randint(1, 200); sh = dh = rng.randint(1, 120)
        elif mode < 0.5:      # tiny pictures
            sw, sh, dw, dh = (rng.randint(1, 12) for _ in range(4))
        else:
            sw, dw = rng.randint(1, 260), rng.randint(1, 260)
            sh, dh = rng.randint(1, 140), rng.randint(1, 140)
        flags = rng.choice(SCALERS) | rng.choice(EXTRA)
        out.append((sw, sh, sf, dw, dh, df, flags, k))
    return out


import os
# (SWS_RANDOM_N / SWS_RANDOM_SEED: a longer or different draw of every generator of this file for a bug hunt; the committed suite runs the defaults)
_HUNT_N, _HUNT_SEED = os.environ.get("SWS_RANDOM_N"), os.environ.get("SWS_RANDOM_SEED")


@pytest.mark.parametrize("case", _cases(int(_HUNT_N or 6000), int(_HUNT_SEED or 20260928)), ids=lambda c: f"{c[7]}-{c[2]}_{c[0]}x{c[1]}-{c[5]}_{c[3]}x{c[4]}-{c[6]:x}")
def test_random_conversions(case):
    sw, sh, sf, dw, dh, df, flags, k = case
    try:
        o = OL.Oracle(sw, sh, sf, dw, dh, df, flags)
    except Exception:
        pytest.skip("the oracle refuses this context (the product refuses it too: tests/test_refusals_agree.py)")
    del o
    run_case(sw, sh, sf, dw, dh, df, flags, seed=k + 1, device_frames=bool(k & 1))


def _opt_cases(n, seed):
    rng = random.Random(seed)
    out = []
    for k in range(n):
        sf, df = rng.choice(FORMAT_MATRIX_SRC), rng.choice(FORMAT_MATRIX_DST)
        if rng.random() < 0.4:
            sw = dw = rng.randint(2, 160); sh = dh = rng.randint(2, 100)
        else:
            sw, dw, sh, dh = rng.randint(2, 200), rng.randint(2, 200), rng.randint(2, 120), rng.randint(2, 120)
        flags = rng.choice(SCALERS) | rng.choice(EXTRA)
        opts = {"dither": rng.choice([0, 1, 2]), "src_range": rng.choice([0, 1]), "dst_range": rng.choice([0, 1]), "threads": 1}
        if rng.random() < 0.5:
            opts.update(src_h_chr_pos=rng.choice([-513, 0, 128, 256]), src_v_chr_pos=rng.choice([-513, 0, 128, 256]),
                        dst_h_chr_pos=rng.choice([-513, 0, 128, 256]), dst_v_chr_pos=rng.choice([-513, 0, 128, 256]))
        cs = None
        if rng.random() < 0.5:
            cs = (rng.choice([1, 5, 9]), rng.choice([0, 1]), rng.choice([1, 5, 9]), rng.choice([0, 1]),
                  rng.choice([0, 0, 1 << 12]), rng.choice([1 << 16, 1 << 16, 3 << 15]), rng.choice([1 << 16, 1 << 16, 1 << 15]))
        out.append((sw, sh, sf, dw, dh, df, flags, k, opts, cs))
    return out


@pytest.mark.parametrize("case", _opt_cases(int(_HUNT_N or 3000), int(_HUNT_SEED or 777)), ids=lambda c: f"{c[7]}-{c[2]}_{c[0]}x{c[1]}-{c[5]}_{c[3]}x{c[4]}-{c[6]:x}")
def test_random_conversions_with_options(case):
    """the same with the sws_alloc_context() + public fields + sws_init_context() construction: dither mode, ranges, chroma positions,
    and sws_setColorspaceDetails() with brightness / contrast / saturation."""
    sw, sh, sf, dw, dh, df, flags, k, opts, cs = case
    try:
        o = OL.Oracle(sw, sh, sf, dw, dh, df, flags, **opts)
    except Exception:
        pytest.skip("the oracle refuses this context")
    del o
    run_case(sw, sh, sf, dw, dh, df, flags, seed=k + 7, colorspace=cs, device_frames=bool(k & 1), opts=opts)


# formats around which round 3 added helper passes and planner rules (reader pre-pass, 4:2:2 / semi-planar splits, packed 4:2:2 join, 16-bit sources
# into the RGB epilogue, gray -> gray, one-tap vertical filters, long vertical chroma filters): drawn more densely, on pictures the planner's
# width threshold would keep on the tile kernels (strip_min_w = 0), sizes from degenerate to a few strips wide
STRIP_SRC = ["yuv420p", "yuv422p", "yuv444p", "yuv410p", "yuv411p", "yuv440p", "yuvj420p", "yuvj444p", "yuv420p10le", "yuv422p10le", "yuv444p12le", "yuv420p9le", "yuv420p14le",
             "yuv420p16le", "yuv420p10be", "nv12", "nv21", "nv16", "nv24", "p010le", "p012le", "p210le", "p010be", "p016le", "yuyv422", "uyvy422", "yvyu422", "rgb24", "bgr24",
             "rgba", "bgra", "argb", "abgr", "rgb0", "gbrp", "gbrap", "gray8", "gray10le", "gray12le", "gray16le", "yuva420p", "rgb48le", "gbrp10le", "yuva444p", "yuva422p10le", "0bgr",
             "yuva420p16le"]
STRIP_DST = ["yuv420p", "yuv422p", "yuv444p", "yuv411p", "yuvj420p", "yuv420p10le", "yuv422p12le", "yuv444p9le", "yuv420p16le", "nv12", "nv21", "nv16", "p010le", "p012le", "p016le",
             "yuyv422", "uyvy422", "yvyu422", "rgb24", "bgr24", "rgba", "bgra", "argb", "abgr", "bgr0", "gray8", "gray10le", "yuva420p", "gbrp", "rgb48le", "yuv420p10be",
             "yuva444p10le", "yuva422p", "gbrap", "gbrp10le", "gbrap12le", "gbrp12msble", "rgb0", "gbrp16le"]


def _strip_cases(n, seed, srcs=None, dsts=None):
    rng = random.Random(seed)
    out = []
    srcs, dsts = srcs or STRIP_SRC, dsts or STRIP_DST
    for k in range(n):
        sf, df = rng.choice(srcs), rng.choice(dsts)
        mode = rng.random()
        if mode < 0.2:        # same size
            sw = dw = rng.choice([rng.randint(2, 400), 4 * rng.randint(1, 120)]); sh = dh = rng.randint(2, 90)
        elif mode < 0.35:     # one direction unscaled (one-tap filters), or exact 2:1 / 4:1 steps
            sw = 4 * rng.randint(8, 160); sh = 2 * rng.randint(4, 60)
            dw, dh = rng.choice([(sw, max(2, sh // 2)), (max(2, sw // 2), sh), (sw // 2, sh // 2), (sw // 4, sh // 4), (sw * 2, sh), (sw, sh * 2), (sw * 2, sh * 2)])
        else:
            sw, dw = rng.choice([rng.randint(2, 700), 4 * rng.randint(4, 170)]), rng.choice([rng.randint(2, 700), 2 * rng.randint(4, 340)])
            sh, dh = rng.randint(2, 100), rng.randint(2, 100)
        flags = rng.choice(SCALERS) | rng.choice(EXTRA)
        opts, cs = {}, None
        if rng.random() < 0.4:
            opts = {"dither": rng.choice([0, 1, 2]), "src_range": rng.choice([0, 1]), "dst_range": rng.choice([0, 1])}
            if rng.random() < 0.5:
                opts.update(src_h_chr_pos=rng.choice([-513, 0, 128, 256]), src_v_chr_pos=rng.choice([-513, 0, 128, 256]),
                            dst_h_chr_pos=rng.choice([-513, 0, 128, 256]), dst_v_chr_pos=rng.choice([-513, 0, 128, 256]))
            if rng.random() < 0.4:
                cs = (rng.choice([1, 5, 9]), rng.choice([0, 1]), rng.choice([1, 5, 9]), rng.choice([0, 1]),
                      rng.choice([0, 0, 1 << 12]), rng.choice([1 << 16, 1 << 16, 3 << 15]), rng.choice([1 << 16, 1 << 16, 1 << 15]))
        tune = {"strip_min_w": 0}
        if rng.random() < 0.3:
            tune.update(strip_cols_l=rng.choice([2, 4]), strip_cols_c=rng.choice([1, 2]), strip_rgb_cols=rng.choice([2, 4]))
        if rng.random() < 0.2:      # wide pictures with the planner's own thresholds
            sw, dw = rng.choice([(rng.randint(1024, 2100), rng.randint(1024, 1700)), (4 * rng.randint(256, 520), 2 * rng.randint(512, 900))])
            sh, dh = rng.randint(2, 40), rng.randint(2, 40)
            tune = {}
        out.append((sw, sh, sf, dw, dh, df, flags, k, opts, cs, tune))
    return out


@pytest.mark.parametrize("case", _strip_cases(int(_HUNT_N or 4000), int(_HUNT_SEED or 31337)), ids=lambda c: f"{c[7]}-{c[2]}_{c[0]}x{c[1]}-{c[5]}_{c[3]}x{c[4]}-{c[6]:x}")
def test_random_conversions_on_the_strip_family(case):
    sw, sh, sf, dw, dh, df, flags, k, opts, cs, tune = case
    try:
        o = OL.Oracle(sw, sh, sf, dw, dh, df, flags, **opts)
    except Exception:
        pytest.skip("the oracle refuses this context")
    del o
    run_case(sw, sh, sf, dw, dh, df, flags, seed=k + 3, colorspace=cs, device_frames=bool(k % 3), opts=opts or None, tune=tune)


# the formats round 4 moved behind the strip kernels (reader pre-pass per source kind: sws_k_read16_kind; packed destinations through the int32 sums:
# sws_k_sum_writer; scaled packed RGB -> packed RGB in one launch: sws_k_strip_rgb2rgb; packed 4:2:2 sources without the split pass), drawn against
# each other and against the plain planar / semi-planar formats with the strip generator's sizes, options, colourspace details and strip-width tunes
R4_SRC = ["x2rgb10le", "x2bgr10le", "rgb565le", "bgr565le", "rgb555le", "bgr555be", "rgb444le", "bgr444le", "rgb565be", "gbrp9le", "gbrp10le", "gbrp12le", "gbrp14le", "gbrp10be",
          "gbrap10le", "gbrap12le", "y210le", "y212le", "xv30le", "v30xle", "xv36le", "xv36be", "ayuv", "vuya", "vuyx", "uyva", "vyu444", "yuva420p", "yuva444p", "rgba", "bgra",
          "rgb24", "bgr24", "argb", "0rgb", "yuyv422", "uyvy422", "yvyu422", "yuv420p", "nv12", "yuv444p", "yuv422p10le", "p010le", "gbrp", "gbrap"]
R4_DST = ["rgb565le", "bgr565le", "rgb555le", "bgr555le", "rgb444le", "bgr444le", "rgb565be", "bgr555be", "x2rgb10le", "x2bgr10le", "ayuv", "vuya", "vuyx", "uyva", "vyu444",
          "y210le", "y212le", "xv30le", "v30xle", "xv36le", "xv36be", "yuv420p", "yuv422p", "yuv444p", "nv12", "nv21", "nv16", "p010le", "yuv420p10le", "yuyv422", "uyvy422",
          "rgb24", "bgr24", "rgba", "bgra", "argb", "abgr", "rgb0", "gbrp", "gbrap", "gray8", "yuva420p"]


@pytest.mark.parametrize("case", _strip_cases(int(_HUNT_N or 3000), int(_HUNT_SEED or 40404), R4_SRC, R4_DST),
                         ids=lambda c: f"{c[7]}-{c[2]}_{c[0]}x{c[1]}-{c[5]}_{c[3]}x{c[4]}-{c[6]:x}")
def test_random_conversions_on_the_round4_routes(case):
    sw, sh, sf, dw, dh, df, flags, k, opts, cs, tune = case
    try:
        o = OL.Oracle(sw, sh, sf, dw, dh, df, flags, **opts)
    except Exception:
        pytest.skip("the oracle refuses this context")
    del o
    run_case(sw, sh, sf, dw, dh, df, flags, seed=k + 5, colorspace=cs, device_frames=bool(k % 3), opts=opts or None, tune=tune)


# sources of one to four rows (initFilter gives their vertical banks one tap, with zero-vector rows under a shifted chroma position) and destinations of a few
# rows, wide enough for the strip family: the corner the one-tap rules of the planner (plane1 forms, the packed writers' _1 / _2 / X choice per row) live in
def _short_cases(n, seed):
    rng = random.Random(seed)
    srcs, dsts = sorted(set(STRIP_SRC + R4_SRC)), sorted(set(STRIP_DST + R4_DST))
    if os.environ.get("SWS_RANDOM_ALL_FORMATS"):      # (hunts: every format of the matrix instead of the strip family's)
        srcs, dsts = sorted(FORMAT_MATRIX_SRC), sorted(FORMAT_MATRIX_DST)
    out = []
    for k in range(n):
        sf, df = rng.choice(srcs), rng.choice(dsts)
        sw = rng.choice([4 * rng.randint(80, 480), rng.randint(320, 2000)]); dw = rng.choice([sw, 4 * rng.randint(80, 480), rng.randint(320, 2000)])
        sh = rng.randint(1, 4); dh = rng.choice([sh, rng.randint(1, 6), rng.randint(1, 60)])
        if rng.random() < 0.25:
            sh, dh = dh, sh
        flags = rng.choice(SCALERS) | rng.choice(EXTRA)
        opts = {}
        if rng.random() < 0.6:
            opts = dict(src_v_chr_pos=rng.choice([-513, 0, 128, 256, 384, 512]), dst_v_chr_pos=rng.choice([-513, 0, 128, 256, 512]))
            if rng.random() < 0.3:
                opts.update(dither=rng.choice([0, 1, 2]), src_range=rng.choice([0, 1]), dst_range=rng.choice([0, 1]))
        tune = {"strip_min_w": 0} if rng.random() < 0.5 else {}
        out.append((sw, sh, sf, dw, dh, df, flags, k, opts, None, tune))
    return out


@pytest.mark.parametrize("case", _short_cases(int(_HUNT_N or 2500), int(_HUNT_SEED or 60606)), ids=lambda c: f"{c[7]}-{c[2]}_{c[0]}x{c[1]}-{c[5]}_{c[3]}x{c[4]}-{c[6]:x}")
def test_random_conversions_of_pictures_of_a_few_rows(case):
    sw, sh, sf, dw, dh, df, flags, k, opts, cs, tune = case
    try:
        o = OL.Oracle(sw, sh, sf, dw, dh, df, flags, **opts)
    except Exception:
        pytest.skip("the oracle refuses this context")
    del o
    run_case(sw, sh, sf, dw, dh, df, flags, seed=k + 9, device_frames=bool(k % 3), opts=opts or None, tune=tune)


def _slice_cases(n, seed):
    rng = random.Random(seed)
    out = []
    for k in range(n):
        sf, df = rng.choice(FORMAT_MATRIX_SRC + STRIP_SRC), rng.choice(FORMAT_MATRIX_DST + STRIP_DST)
        if rng.random() < 0.45:
            sw = dw = rng.randint(2, 300); sh = dh = 2 * rng.randint(4, 48)
        else:
            sw, dw, sh, dh = rng.randint(2, 400), rng.randint(2, 400), 2 * rng.randint(4, 48), rng.randint(2, 96)
        flags = rng.choice(SCALERS) | rng.choice(EXTRA)
        tune = {"strip_min_w": 0} if rng.random() < 0.5 else {}
        out.append((sw, sh, sf, dw, dh, df, flags, k, tune, rng.randint(2, 4), rng.random()))
    return out


@pytest.mark.parametrize("case", _slice_cases(int(_HUNT_N or 2500), int(_HUNT_SEED or 8088)), ids=lambda c: f"{c[7]}-{c[2]}_{c[0]}x{c[1]}-{c[5]}_{c[3]}x{c[4]}-{c[6]:x}")
def test_random_slice_sequences(case):
    """top-down slice sequences with random cuts (multiples of four rows: whole chroma rows of every format, whole Bayer blocks) through whatever the
    context is -- special converter, scaler, cascade: the assembled picture is the whole-frame result, no call fails"""
    import numpy as np
    import torch
    from librempeg_amd import SwsContext, HostFrame, DeviceFrame
    from test_gpu_parity import _slice_ptrs
    sw, sh, sf, dw, dh, df, flags, k, tune, nsl, r = case
    try:
        o = OL.Oracle(sw, sh, sf, dw, dh, df, flags)
    except Exception:
        pytest.skip("the oracle refuses this context")
    src = OL.fill_random(OL.Frame(sf, sw, sh), k + 11)
    ref = OL.Frame(df, dw, dh, fill=0x5A)
    whole = o.scale(src, ref)
    assert whole >= 0
    p = SwsContext(sw, sh, sf, dw, dh, df, flags)
    for kk, v in tune.items():
        p.set_option(kk, v)
    rng = random.Random(k)
    cuts = sorted({4 * rng.randint(1, max(1, sh // 4 - 1)) for _ in range(nsl - 1)} | {0, sh})
    cuts = [c for c in cuts if c <= sh]
    hs = HostFrame(sf, sw, sh)
    for a, b in zip(hs.planes, src.planes):
        a[:] = b
    ds = DeviceFrame(sf, sw, sh).upload(hs)
    dd = DeviceFrame(df, dw, dh)
    dd.buf.fill_(0x5A)
    torch.cuda.synchronize()
    dp, dstr = dd.ptrs()
    rets = []
    for y0, y1 in zip(cuts[:-1], cuts[1:]):
        sp, ss = _slice_ptrs(ds, sf, y0)
        rets.append(p.L.sws_scale(p.c, sp, ss, y0, y1 - y0, dp, dstr))
    p.sync()
    assert all(x >= 0 for x in rets), (rets, cuts, p.path())
    # (planarRgbToplanarRgbWrapper: ff_copyPlane with equal strides is one memcpy of (sliceH - 1) * stride + src_w BYTES per slice, swscale_unscaled.c:126-145 --
    #  the last row of every slice of a 16-bit picture is copied in part only)
    if p.path() in ("unscaled:planarCopy", "unscaled:yvu9ToYv12", "unscaled:planarRgbToplanarRgb"):      # (yvu9ToYv12Wrapper: planar2x_c interpolates between the chroma rows of ONE slice, swscale_unscaled.c:2079-2093)
        # DITHER_COPY (swscale_unscaled.c:2159-2218) indexes its dither rows with the row number INSIDE the slice ("dithers[shift-1][i&7]", i from 0 per call):
        # the reference's result depends on the cuts.  planarCopyWrapper works row by row otherwise, so the expectation is every slice converted as a
        # picture of its own
        for y0, y1 in zip(cuts[:-1], cuts[1:]):
            sub_s, sub_d = OL.Frame(sf, sw, y1 - y0), OL.Frame(df, dw, y1 - y0, fill=0x5A)
            lay_s, lay_d = OL._FORMATS[sf], OL._FORMATS[df]
            for i, pl in enumerate(sub_s.planes):
                r0 = y0 if (i == 0 or i == 3 or lay_s[1] in ("rgbp", "packed", "gray")) else (y0 >> lay_s[3])
                pl[:] = src.planes[i][r0:r0 + pl.shape[0]]
            assert OL.Oracle(sw, y1 - y0, sf, dw, y1 - y0, df, flags).scale(sub_s, sub_d) >= 0
            for i, pl in enumerate(sub_d.planes):
                r0 = y0 if (i == 0 or i == 3 or lay_d[1] in ("rgbp", "packed", "gray")) else (y0 >> lay_d[3])
                ref.planes[i][r0:r0 + pl.shape[0]] = pl
    out = dd.download()
    for i, (a, b) in enumerate(zip(out.planes, ref.planes)):
        rb = out.row_bytes[i]
        if df in ("monob", "monow") and (dw & 7):
            a, b = a.copy(), b.copy()
            m = (0xFF00 >> (dw & 7)) & 0xFF
            a[:, rb - 1] &= m; b[:, rb - 1] &= m
        assert np.array_equal(a[:, :rb], b[:, :rb]), (case[:7], cuts, rets, p.path(), i, int(np.count_nonzero(a[:, :rb] != b[:, :rb])))


def _batch_cases(n, seed, srcs=None, dsts=None):
    rng = random.Random(seed)
    out = []
    for k in range(n):
        sf, df = rng.choice(srcs or STRIP_SRC), rng.choice(dsts or STRIP_DST)
        if rng.random() < 0.3:
            sw = dw = 4 * rng.randint(2, 90); sh = dh = 2 * rng.randint(2, 30)
        else:
            sw, dw, sh, dh = rng.choice([rng.randint(2, 400), 4 * rng.randint(4, 100)]), rng.choice([rng.randint(2, 400), 2 * rng.randint(4, 200)]), rng.randint(2, 64), rng.randint(2, 64)
        flags = rng.choice(SCALERS[:6]) | rng.choice(EXTRA)
        tune = {"strip_min_w": 0} if rng.random() < 0.7 else {}
        out.append((sw, sh, sf, dw, dh, df, flags, k, tune, [rng.randint(1, 5) for _ in range(3)]))
    return out


@pytest.mark.parametrize("case", _batch_cases(int(_HUNT_N or 1200), int(_HUNT_SEED or 4711)), ids=lambda c: f"{c[7]}-{c[2]}_{c[0]}x{c[1]}-{c[5]}_{c[3]}x{c[4]}-{c[6]:x}")
def test_random_batches(case):
    _run_batches(case)


@pytest.mark.parametrize("case", _batch_cases(int(_HUNT_N or 800), int(_HUNT_SEED or 4712), R4_SRC, R4_DST), ids=lambda c: f"{c[7]}-{c[2]}_{c[0]}x{c[1]}-{c[5]}_{c[3]}x{c[4]}-{c[6]:x}")
def test_random_batches_on_the_round4_routes(case):
    """the same over the formats of round 4's helper passes: per-kind reader planes (sws_k_read16_kind) and the int32 sum planes of the packed writers"""
    _run_batches(case)


def _run_batches(case):
    """sws_scale_frames() three times on one context with batches of different sizes and different frames (HBM frames, then a mix with host frames):
    the per-frame working pictures and cached frame tables of the helper passes (reader pre-pass, 4:2:2 / semi-planar splits, 4:2:2 join) must follow"""
    import numpy as np
    import torch
    from librempeg_amd import SwsContext, HostFrame, DeviceFrame
    sw, sh, sf, dw, dh, df, flags, k, tune, sizes = case
    try:
        o = OL.Oracle(sw, sh, sf, dw, dh, df, flags)
    except Exception:
        pytest.skip("the oracle refuses this context")
    p = SwsContext(sw, sh, sf, dw, dh, df, flags)
    for kk, v in tune.items():
        p.set_option(kk, v)
    seed = 1000 * k
    for rnd, n in enumerate(sizes):
        refs, srcs, dsts = [], [], []
        for i in range(n):
            seed += 1
            s = OL.fill_random(OL.Frame(sf, sw, sh), seed)
            ref = OL.Frame(df, dw, dh, fill=0x33)
            assert o.scale(s, ref) >= 0
            refs.append(ref)
            hs = HostFrame(sf, sw, sh)
            for a, b in zip(hs.planes, s.planes):
                a[:] = b
            host = rnd == 2 and (i & 1)
            if host:
                hd = HostFrame(df, dw, dh)
                for a in hd.planes:
                    a[:] = 0x33
                srcs.append(hs); dsts.append(hd)
            else:
                dd = DeviceFrame(df, dw, dh)
                dd.buf.fill_(0x33)
                srcs.append(DeviceFrame(sf, sw, sh).upload(hs)); dsts.append(dd)
        torch.cuda.synchronize()
        assert p.scale_frames(srcs, dsts) == n, (rnd, n, p.path())
        p.sync()
        for i in range(n):
            out = dsts[i].download() if isinstance(dsts[i], DeviceFrame) else dsts[i]
            for pl, (a, b) in enumerate(zip(out.planes, refs[i].planes)):
                rb = out.row_bytes[pl]
                if not np.array_equal(a[:, :rb], b[:, :rb]):
                    raise AssertionError(f"{case[:7]} path={p.path()} call {rnd} of {sizes}, frame {i} of {n}, plane {pl}: {int(np.count_nonzero(a[:, :rb] != b[:, :rb]))} bytes differ"
                                         + _batch_forensics(p, o, case, srcs[i], dsts[i], refs[i], seed - n + i + 1))


def _batch_forensics(p, o, case, src_frame, dst_frame, ref, frame_seed):
    """what a failing frame of a batch says on a second look (the rare events of DESIGN.md 8): the frame alone on the same context, on a fresh context, the oracle again"""
    import numpy as np
    from librempeg_amd import SwsContext, HostFrame, DeviceFrame
    sw, sh, sf, dw, dh, df, flags, k, tune, sizes = case
    notes = []
    try:
        def planes_equal(x, y):
            return all(np.array_equal(a[:, :rb], b[:, :rb]) for a, b, rb in zip(x.planes, y.planes, x.row_bytes))
        try:      # before anything else runs on it: are the context's device tables still what the host uploaded?
            nbad, text = p.debug_check()
            notes.append(f"device tables of the failed context: {nbad} anomalies ({text.strip()})")
        except Exception as e:
            notes.append(f"device table check unavailable: {e!r}")
        s = OL.fill_random(OL.Frame(sf, sw, sh), frame_seed)
        ref2 = OL.Frame(df, dw, dh, fill=0x33)
        o.scale(s, ref2)
        notes.append(f"oracle recomputed equals its first answer: {all(np.array_equal(a, b) for a, b in zip(ref2.planes, ref.planes))}")
        hs = HostFrame(sf, sw, sh)
        for a, b in zip(hs.planes, s.planes):
            a[:] = b
        if isinstance(src_frame, DeviceFrame):
            back = src_frame.download()
            notes.append(f"source on the GPU intact: {planes_equal(back, hs)}")
        hd = HostFrame(df, dw, dh)
        p.scale(hs, hd)
        notes.append(f"the frame alone on the same context equals the oracle: {planes_equal(hd, ref)}")
        if isinstance(src_frame, DeviceFrame) and isinstance(dst_frame, DeviceFrame):      # context state or frame addresses? (tests/test_gpu_parity.py _forensics)
            import torch
            notes.append(f"device addresses: src {[hex(a) for a in src_frame.ptrs()[0][:4] if a]} dst {[hex(a) for a in dst_frame.ptrs()[0][:4] if a]}")
            dst_frame.buf.fill_(0x33)
            torch.cuda.synchronize()
            p.scale(src_frame, dst_frame); p.sync()
            notes.append(f"the frame alone on the same context and the SAME device frames equals the oracle: {planes_equal(dst_frame.download(), ref)}")
        p2 = SwsContext(sw, sh, sf, dw, dh, df, flags)
        for kk, v in tune.items():
            p2.set_option(kk, v)
        hd2 = HostFrame(df, dw, dh)
        p2.scale(hs, hd2)
        notes.append(f"a fresh context equals the oracle: {planes_equal(hd2, ref)}")
        if isinstance(src_frame, DeviceFrame) and isinstance(dst_frame, DeviceFrame):
            dst_frame.buf.fill_(0x33)
            torch.cuda.synchronize()
            p2.scale(src_frame, dst_frame); p2.sync()
            notes.append(f"a FRESH context on the SAME device frames equals the oracle: {planes_equal(dst_frame.download(), ref)}")
        p2.close()
    except Exception as e:   # (forensics must never hide the failure itself)
        notes.append(f"forensics stopped: {e!r}")
    return " || forensics: " + "; ".join(notes)


# HBM-resident frames with odd plane pointers and line sizes, or stored bottom-up (tests/test_gpu_unaligned_frames.py): the strip family's draws on
# such frames -- the helper passes go through aligned working copies, everything else through the per-sample kernels
def _odd_cases(n, seed, srcs=None, dsts=None):
    rng = random.Random(seed ^ 0x5EED)
    out = []
    for c in _strip_cases(n, seed + 77, srcs, dsts):
        out.append(c + ((rng.choice([0, 1, 2, 6, 13]), rng.choice([0, 1, 2, 3, 6]), rng.choice([0, 0, 1, 2, 3])),))
    return out


@pytest.mark.parametrize("case", _odd_cases(int(_HUNT_N or 1500), int(_HUNT_SEED or 4242)), ids=lambda c: f"{c[7]}-{c[2]}_{c[0]}x{c[1]}-{c[5]}_{c[3]}x{c[4]}-{c[6]:x}")
def test_random_conversions_on_unaligned_frames(case):
    _run_odd_case(case)


@pytest.mark.parametrize("case", _odd_cases(int(_HUNT_N or 1000), int(_HUNT_SEED or 4343), R4_SRC, R4_DST), ids=lambda c: f"{c[7]}-{c[2]}_{c[0]}x{c[1]}-{c[5]}_{c[3]}x{c[4]}-{c[6]:x}")
def test_random_conversions_on_unaligned_frames_of_the_round4_formats(case):
    _run_odd_case(case)


def _run_odd_case(case):
    from test_gpu_unaligned_frames import run_odd
    sw, sh, sf, dw, dh, df, flags, k, opts, cs, tune, (pad, shift, flip) = case
    try:
        o = OL.Oracle(sw, sh, sf, dw, dh, df, flags, **opts)
    except Exception:
        pytest.skip("the oracle refuses this context")
    del o
    run_odd(sw, sh, sf, dw, dh, df, flags, pad, shift, flip, nframes=1 + k % 3, opts=opts, colorspace=cs, tune=tune, seed=k + 11)
