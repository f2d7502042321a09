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


@pytest.mark.parametrize("case", _cases(6000, 20260928), ids=lambda c: f"{c[7]}-{c[2]}_{c[0]}x{c[1]}-{c[5]}_{c[3]}x{c[4]}-{c[6]:x}")
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


@pytest.mark.parametrize("case", _opt_cases(3000, 777), ids=lambda c: f"{c[7]}-{c[2]}_{c[0]}x{c[1]}-{c[5]}_{c[3]}x{c[4]}-{c[6]:x}")
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
