"""-m gpu tests at BASELINE.json's full sizes.

Every BASELINE configuration is converted once at its full size and compared with the oracle over the WHOLE frame (the scalar C
oracle needs 8 ms .. 1 s per frame); the real C4 batch (512 frames through one sws_scale_frames() call) is
checked frame by frame; widths that straddle the 1024-pixel wave strips are covered at small heights for every wave / march /
strip / stream kernel.  Size-independent properties on top: batch == N single calls, idempotence, constant image.
"""
import hashlib

import numpy as np
import pytest
import torch

import oracle_lib as OL
from librempeg_amd import (SwsContext, HostFrame, DeviceFrame, plane_layout, SWS_BICUBIC, SWS_BILINEAR, SWS_LANCZOS,
                           SWS_BITEXACT, SWS_ACCURATE_RND, SWS_CS_BT2020)
from test_gpu_parity import run_case

pytestmark = pytest.mark.gpu
BX, AR = SWS_BITEXACT, SWS_ACCURATE_RND


def random_device_frame(fmt, w, h, seed):
    host = OL.fill_random(OL.Frame(fmt, w, h, align=256), seed)
    hf = HostFrame(fmt, w, h, align=256)
    for a, b in zip(hf.planes, host.planes):
        a[:] = b
    return DeviceFrame(fmt, w, h).upload(hf), hf


def digest(frame):
    return hashlib.sha256(frame.visible()).hexdigest()


def oracle_frame(sfmt, dfmt, sw, sh, dw, dh, flags, hsrc, cs=None):
    """the oracle's whole output frame for the host copy `hsrc` of the source"""
    o = OL.Oracle(sw, sh, sfmt, dw, dh, dfmt, flags)
    if cs:
        assert o.set_colorspace(*cs) == 0
    osrc = OL.Frame(sfmt, sw, sh)
    for a, b, (rb, _) in zip(osrc.planes, hsrc.planes, plane_layout(sfmt, sw, sh)):
        a[:, :rb] = b[:, :rb]
    ref = OL.Frame(dfmt, dw, dh)
    assert o.scale(osrc, ref) == dh
    return ref


def assert_frames_equal(out, ref, dfmt, w, h, what):
    for pi, (a, b, (rb, _)) in enumerate(zip(out.planes, ref.planes, plane_layout(dfmt, w, h))):
        if not np.array_equal(a[:, :rb], b[:, :rb]):
            bad = np.argwhere(a[:, :rb] != b[:, :rb])
            raise AssertionError(f"{what}: plane {pi}: {len(bad)} bytes differ, first at row {bad[0][0]} byte {bad[0][1]}")


FULL = {
    "c2a": ("yuv420p", "rgb24", 3840, 2160, 3840, 2160, SWS_BICUBIC | BX, None, "unscaled:yuv2rgb"),
    "c2b": ("yuv420p", "rgb24", 3840, 2160, 3840, 2160, SWS_BICUBIC | BX | AR, None, "main:fused_rgb_unity"),
    "c3a": ("yuv420p10le", "p010le", 7680, 4320, 7680, 4320, SWS_LANCZOS | BX, None, "unscaled:planarToP01x"),
    "c3b": ("yuv420p10le", "p010le", 7680, 4320, 3840, 2160, SWS_LANCZOS | BX, None, "main:strip_march"),
    "c4": ("nv12", "bgr0", 1920, 1080, 1920, 1080, SWS_BICUBIC | BX, None, "main:fused_rgb_unity"),
    "c5": ("gbrpf32le", "yuv444p16le", 3840, 2160, 3840, 2160, SWS_BICUBIC | BX, (SWS_CS_BT2020, 1, SWS_CS_BT2020, 1), "main:fused_f32rgb_yuv444"),
    "c1": ("yuv420p", "yuv420p", 1280, 720, 640, 360, SWS_BILINEAR | BX, None, None),
    # not BASELINE configurations: the downscale-to-packed-RGB shapes of bench.py's d1 / d2 variants
    "d1": ("yuv420p", "rgb24", 3840, 2160, 1920, 1080, SWS_BICUBIC | BX, None, "main:strip_rgb"),
    "d2": ("yuv420p", "bgra", 3840, 2160, 1920, 1080, SWS_BICUBIC | BX, None, "main:strip_rgb"),
    "d3": ("yuv422p", "argb", 1920, 1080, 2560, 1440, SWS_LANCZOS | BX | AR, None, "main:strip_rgb"),
}


@pytest.mark.parametrize("name", list(FULL))
def test_fullsize_whole_frame_equals_oracle(name):
    """every BASELINE configuration at its full size, every byte of the output against the oracle"""
    sfmt, dfmt, sw, sh, dw, dh, flags, cs, path = FULL[name]
    ctx = SwsContext(sw, sh, sfmt, dw, dh, dfmt, flags)
    if cs:
        assert ctx.set_colorspace(*cs) == 0
    src, hsrc = random_device_frame(sfmt, sw, sh, 11)
    dst = DeviceFrame(dfmt, dw, dh)
    torch.cuda.synchronize()
    assert ctx.scale(src, dst) == dh
    ctx.sync()
    if path:
        assert ctx.path() == path
    assert_frames_equal(dst.download(), oracle_frame(sfmt, dfmt, sw, sh, dw, dh, flags, hsrc, cs), dfmt, dw, dh, name)


# widths around the 1024-pixel strips of the wave / march kernels (one wave = 1024 pixels; 4 waves per block = adjacent segments),
# and around the 256- / 128-column strips of the planar strip kernel
EVEN_W = [1022, 1024, 1026, 2046, 2050, 3074, 4098]
ANY_W = [1023, 1025, 2047, 2050, 3074]


@pytest.mark.parametrize("w", EVEN_W)
@pytest.mark.parametrize("case", [("yuv420p", "rgb24", SWS_BICUBIC | BX, "unscaled:yuv2rgb"), ("yuv422p", "bgra", SWS_BICUBIC | BX, "unscaled:yuv2rgb"),
                                  ("yuv420p", "rgb24", SWS_BICUBIC | BX | AR, "main:fused_rgb_unity"), ("yuv420p", "argb", SWS_BICUBIC | BX | AR, "main:fused_rgb_unity"),
                                  ("nv12", "bgr0", SWS_BICUBIC | BX, "main:fused_rgb_unity"), ("nv21", "bgr24", SWS_BICUBIC | BX, "main:fused_rgb_unity"),
                                  ("yuv422p", "rgba", SWS_BICUBIC | BX | AR, "main:fused_rgb_unity")],
                         ids=lambda c: f"{c[0]}-{c[1]}-{c[2]:x}")
def test_wave_strip_boundaries_packed_rgb(case, w):
    sfmt, dfmt, flags, path = case
    for h in (34, 6):
        got, _ = run_case(w, h, sfmt, w, h, dfmt, flags, seed=w + h)
        assert got == path


@pytest.mark.parametrize("w", ANY_W)
def test_wave_strip_boundaries_streaming_kernels(w):
    assert run_case(w, 18, "yuv420p10le", w, 18, "p010le", SWS_LANCZOS | BX, seed=w)[0] == "unscaled:planarToP01x"
    assert run_case(w, 18, "yuv420p16le", w, 18, "p016le", SWS_LANCZOS | BX, seed=w)[0] == "unscaled:planarToP01x"
    assert run_case(w, 10, "gbrpf32le", w, 10, "yuv444p16le", SWS_BICUBIC | BX, seed=w, colorspace=(SWS_CS_BT2020, 1, SWS_CS_BT2020, 1))[0] == "main:fused_f32rgb_yuv444"
    assert run_case(w, 10, "gbrpf32le", w, 10, "yuv444p10le", SWS_BICUBIC | BX, seed=w)[0] == "main:fused_f32rgb_yuv444"


@pytest.mark.parametrize("dw", ANY_W)
@pytest.mark.parametrize("case", [("yuv420p10le", "p010le", 2, SWS_LANCZOS), ("yuv420p", "yuv420p", 2, SWS_BILINEAR), ("yuv420p", "nv12", 1.5, SWS_BICUBIC),
                                  ("yuv444p12le", "yuv422p10le", 0.75, SWS_BICUBIC)], ids=lambda c: f"{c[0]}-{c[1]}-x{c[2]}")
def test_strip_kernel_boundaries(case, dw):
    """the planar strip kernel (256-column luma / 128-column chroma strips, 4 waves per block) at widths around its strips"""
    sfmt, dfmt, ratio, scaler = case
    sw = int(dw * ratio)
    got, _ = run_case(sw, 40, sfmt, dw, 22, dfmt, scaler | BX, seed=dw, tune={"strip_min_w": 0})
    assert got == "main:strip_march"


@pytest.mark.parametrize("dw", [254, 256, 258, 510, 512, 514, 1022, 1026, 1918, 2050])
@pytest.mark.parametrize("case", [("yuv420p", "rgb24", 2, SWS_BICUBIC), ("yuv420p", "bgra", 1.5, SWS_BILINEAR), ("yuv422p", "abgr", 0.75, SWS_LANCZOS),
                                  ("yuv440p", "bgr24", 1.5, SWS_BICUBIC), ("yuv420p", "rgb0", 2, SWS_LANCZOS | AR)], ids=lambda c: f"{c[0]}-{c[1]}-x{c[2]}")
def test_strip_rgb_kernel_boundaries(case, dw):
    """the strip kernel with the packed-RGB epilogue (256 luma + 128 chroma columns per wave) at widths around its strips, short and
    long ring forms, bands of different heights"""
    sfmt, dfmt, ratio, scaler = case
    sw = int(dw * ratio) & ~1
    for sh, dh in ((40, 22), (91, 37)):     # (vertical up-scaling with two taps would select the reference's yuv2packed2 rows: not this kernel)
        got, _ = run_case(sw, sh, sfmt, dw, dh, dfmt, scaler | BX, seed=dw + dh)
        assert got == "main:strip_rgb"


@pytest.mark.parametrize("dw", [126, 128, 130, 254, 258, 514, 1026, 1918])
@pytest.mark.parametrize("case", [("yuv420p10le", "rgb24", 2, SWS_BICUBIC), ("yuv420p10le", "bgra", 1.5, SWS_BILINEAR), ("yuv422p12le", "abgr", 0.75, SWS_LANCZOS),
                                  ("yuv420p9le", "bgr24", 1.5, SWS_BICUBIC), ("yuv420p14le", "rgb0", 2, SWS_LANCZOS | AR), ("yuv444p10le", "argb", 2, SWS_BICUBIC)],
                         ids=lambda c: f"{c[0]}-{c[1]}-x{c[2]}")
def test_strip_rgb_kernel_16bit_sources(case, dw):
    """9 .. 15-bit planar sources into the packed-RGB LUT writers (decoded HDR pictures for display): the strip kernel with the RGB epilogue on
    128-column strips, eight samples per 16-byte chunk staged as they are (hScale16To15_c, sh = depth - 1)"""
    sfmt, dfmt, ratio, scaler = case
    sw = int(dw * ratio) & ~1
    for sh, dh in ((40, 22), (91, 37)):
        got, _ = run_case(sw, sh, sfmt, dw, dh, dfmt, scaler | BX, seed=dw + dh)
        if "444" not in sfmt:      # (a 4:4:4 source forces the full-chroma writers: not this kernel)
            assert got == "main:strip_rgb", (got, sfmt, dfmt, sw, dw)
    if dw == 1918:
        assert run_case(3840, 2160, "yuv420p10le", 1920, 1080, "bgra", SWS_BICUBIC | BX, seed=4)[0] == "main:strip_rgb"


def test_c4_batch_of_512_frames_through_one_call():
    """BASELINE config 4 as written: 512 1080p nv12 frames -> bgr0 through ONE sws_scale_frames() call.  Frames 0..7 are random and
    compared with the oracle over the whole frame; frames 8..511 are distinct variations of them made on the GPU (luma xor a
    per-frame byte) and compared with what a single sws_scale() call gives for the same frame."""
    sfmt, dfmt, w, h, flags = "nv12", "bgr0", 1920, 1080, SWS_BICUBIC | BX
    n, nbase = 512, 8
    ctx = SwsContext(w, h, sfmt, w, h, dfmt, flags)
    base = [random_device_frame(sfmt, w, h, 500 + i) for i in range(nbase)]
    srcs = [b[0] for b in base]
    for i in range(nbase, n):
        f = DeviceFrame(sfmt, w, h)
        for k in range(f.nplanes):
            f.plane_tensor(k).copy_(srcs[i % nbase].plane_tensor(k))
        f.plane_tensor(0).bitwise_xor_(i // nbase)
        srcs.append(f)
    dsts = [DeviceFrame(dfmt, w, h) for _ in range(n)]
    torch.cuda.synchronize()
    assert ctx.scale_frames(srcs, dsts) == n
    ctx.sync()
    for i in range(nbase):
        assert_frames_equal(dsts[i].download(), oracle_frame(sfmt, dfmt, w, h, w, h, flags, base[i][1]), dfmt, w, h, f"frame {i}")
    single = DeviceFrame(dfmt, w, h)
    rb = plane_layout(dfmt, w, h)[0][0]
    seen, sampled = set(), 0
    for i in range(nbase, n):
        assert ctx.scale(srcs[i], single) == h
        ctx.sync()
        assert torch.equal(dsts[i].plane_tensor(0)[:, :rb], single.plane_tensor(0)[:, :rb]), f"frame {i}"
        if i % 37 == 0:
            sampled += 1
            seen.add(digest(dsts[i].download()))
    assert len(seen) == sampled   # distinct inputs -> distinct outputs


@pytest.mark.parametrize("name", ["c2a", "c2b", "c4"])
def test_batched_frames_equal_single_calls_checksum_of_checksums(name):
    """sws_scale_frames(N) == N x sws_scale(): digest of per-frame digests; also idempotence (second run identical)."""
    cfg = {"c2a": ("yuv420p", "rgb24", 3840, 2160, SWS_BICUBIC | BX), "c2b": ("yuv420p", "rgb24", 3840, 2160, SWS_BICUBIC | BX | AR),
           "c4": ("nv12", "bgr0", 1920, 1080, SWS_BICUBIC | BX)}[name]
    sfmt, dfmt, w, h, flags = cfg
    n = 5
    ctx = SwsContext(w, h, sfmt, w, h, dfmt, flags)
    srcs = [random_device_frame(sfmt, w, h, 100 + i)[0] for i in range(n)]
    d_batch = [DeviceFrame(dfmt, w, h) for _ in range(n)]
    d_single = [DeviceFrame(dfmt, w, h) for _ in range(n)]
    torch.cuda.synchronize()
    assert ctx.scale_frames(srcs, d_batch) == n
    for s, d in zip(srcs, d_single):
        assert ctx.scale(s, d) == h
    ctx.sync()
    a = hashlib.sha256("".join(digest(d.download()) for d in d_batch).encode()).hexdigest()
    b = hashlib.sha256("".join(digest(d.download()) for d in d_single).encode()).hexdigest()
    assert a == b
    assert ctx.scale_frames(srcs, d_batch) == n
    ctx.sync()
    assert hashlib.sha256("".join(digest(d.download()) for d in d_batch).encode()).hexdigest() == a
    assert len({digest(d.download()) for d in d_batch}) == n   # different inputs -> different outputs


def test_constant_image_maps_to_the_oracles_constant():
    """a flat grey 4K frame must give the value the oracle gives for a flat grey 64x64 frame, everywhere."""
    w, h = 3840, 2160
    for flags in (SWS_BICUBIC | BX, SWS_BICUBIC | BX | AR):
        ctx = SwsContext(w, h, "yuv420p", w, h, "rgb24", flags)
        src = DeviceFrame("yuv420p", w, h)
        src.plane_tensor(0).fill_(81)
        src.plane_tensor(1).fill_(90)
        src.plane_tensor(2).fill_(240)
        dst = DeviceFrame("rgb24", w, h)
        torch.cuda.synchronize()
        assert ctx.scale(src, dst) == h
        ctx.sync()
        o = OL.Oracle(64, 64, "yuv420p", 64, 64, "rgb24", flags)
        s = OL.Frame("yuv420p", 64, 64)
        s.planes[0][:] = 81
        s.planes[1][:] = 90
        s.planes[2][:] = 240
        r = OL.Frame("rgb24", 64, 64)
        o.scale(s, r)
        px = r.planes[0][32, 96:99].tolist()
        t = dst.plane_tensor(0)[:, :3 * w].view(h, w, 3)
        assert all(bool((t[:, :, k] == px[k]).all()) for k in range(3)), px
