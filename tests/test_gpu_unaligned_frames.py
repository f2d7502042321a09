"""-m gpu parity for HBM-resident frames whose plane pointers and line sizes are NOT multiples of 16 bytes (a cropped view of a larger picture, a
tightly packed rgb24 row) or that are stored bottom-up (negative line sizes): sws_scale_frames() takes them like the reference takes any data[] / linesize[] -- the vector kernels and the helper passes
around them (reader pre-pass, 4:2:2 split / join, the full-chroma RGB epilogue) need 16-byte granules, so such frames must still come out bit-exact
through whatever path the library falls back to."""
import numpy as np
import pytest

from librempeg_amd import SWS_BICUBIC, SWS_BILINEAR, SWS_BITEXACT, SWS_FULL_CHR_H_INT, SWS_LANCZOS
from librempeg_amd.swscale import plane_layout

pytestmark = pytest.mark.gpu
BX = SWS_BITEXACT


def _odd_frame(fmt, w, h, pad, shift):
    """DeviceFrame whose planes start `shift` bytes past a 256-byte boundary with line sizes `row_bytes + pad`"""
    import torch
    from librempeg_amd import DeviceFrame
    f = DeviceFrame(fmt, w, h)
    lay = plane_layout(fmt, w, h)
    off, ls, o = [], [], shift
    for rb, rows in lay:
        off.append(o); ls.append(rb + pad)
        o = (o + (rb + pad) * rows + 255) // 256 * 256 + shift
    f.linesize, f.offset, f.total = ls + [0] * (4 - len(ls)), off + [0] * (4 - len(off)), o
    f._allocate(o + 256, "cuda:0")           # (guard bands on either side: download() checks them)
    base = f.buf.data_ptr()
    f.base = (base + 255) // 256 * 256
    f._shift = f.base - base
    return f


class _BottomUp:
    """the same picture stored bottom-up: data[] point at the last stored row, line sizes are negative"""

    def __init__(self, f):
        self.f = f
        self.nplanes, self.row_bytes, self.rows = f.nplanes, f.row_bytes, f.rows

    def view(self):
        v = self.f.view()
        for i in range(self.f.nplanes):
            v.data[i] = self.f.base + self.f.offset[i] + (self.f.rows[i] - 1) * self.f.linesize[i]
            v.linesize[i] = -self.f.linesize[i]
        return v

    def upload(self, host):
        import torch
        for i, a in enumerate(host.planes):
            rb = self.f.row_bytes[i]
            self.f.plane_tensor(i)[:, :rb].copy_(torch.from_numpy(a[::-1, :rb].copy()))
        torch.cuda.synchronize()
        return self

    def download(self):
        out = self.f.download()
        for i in range(len(out.planes)):
            out.planes[i][:] = out.planes[i][::-1].copy()
        return out

    def plane_tensor(self, i):
        return self.f.plane_tensor(i)

    @property
    def buf(self):
        return self.f.buf


CASES = [
    # (src, sw, sh, dst, dw, dh, flags)                       the aligned-frame path of the pair
    ("rgb24", 256, 64, "bgr24", 192, 48, SWS_BICUBIC),        # rgbread + strip + fullchr_rgb
    ("bgra", 256, 64, "bgra", 192, 48, SWS_BICUBIC),          # ... with the alpha launch
    ("yuva420p", 256, 64, "rgba", 192, 48, SWS_BICUBIC | SWS_FULL_CHR_H_INT),
    ("bgra", 256, 64, "yuva420p", 192, 48, SWS_BICUBIC),      # rgbread + strip + alpha
    ("yuva420p", 256, 64, "yuva444p", 192, 48, SWS_BICUBIC),  # strip + alpha
    ("yuv444p", 256, 64, "bgra", 256, 64, SWS_BICUBIC),       # the full-chroma epilogue on the source planes
    ("yuva444p10le", 256, 64, "gbrap", 256, 64, SWS_BICUBIC),
    ("yuv420p", 256, 64, "rgb24", 191, 48, SWS_BICUBIC),      # strip + fullchr_rgb
    ("yuv420p", 256, 64, "bgra", 192, 48, SWS_BICUBIC),       # strip_rgb
    ("yuv420p", 256, 64, "yuyv422", 192, 48, SWS_BICUBIC),    # strip + join422
    ("uyvy422", 256, 64, "yuv420p", 192, 48, SWS_BICUBIC),    # split422 + strip
    ("nv12", 256, 64, "bgra", 192, 48, SWS_BICUBIC),          # splitnv + strip_rgb
    ("p010le", 256, 64, "bgra", 192, 48, SWS_BICUBIC),        # split p01x + strip_rgb
    ("rgb24", 256, 64, "yuv420p", 192, 48, SWS_BICUBIC),      # rgbread + strip
    ("rgb24", 256, 64, "yuv420p", 256, 64, SWS_BICUBIC),      # same-size packed RGB source
    ("yuv420p", 256, 64, "yuv420p", 192, 48, SWS_LANCZOS),    # strip
    ("yuv422p", 256, 64, "yuv420p", 256, 64, SWS_BILINEAR),   # mixed
    ("yuv420p10le", 256, 64, "bgra", 256, 64, SWS_BICUBIC),   # 16-bit strip_rgb, same size
    ("yuyv422", 256, 64, "uyvy422", 256, 64, SWS_BICUBIC),    # layout converter
    ("yuv420p", 256, 64, "nv12", 256, 64, SWS_BICUBIC),       # layout converter
]


def _wide_samples(fmt):
    """formats whose samples are 16-bit (or wider) words: the reference's readers need their natural alignment too"""
    import re
    return bool(re.search(r"(le|be)$|^p0|^p2|^p4|^y2|^xv|^x2|f32|f16|48|64", fmt))


def run_odd(sw, sh, src, dw, dh, dst, flags, pad, shift, flip, nframes=3, opts=None, colorspace=None, tune=None, seed=90):
    import torch
    import oracle_lib as OL
    from librempeg_amd import SwsContext, HostFrame
    if _wide_samples(src) or _wide_samples(dst):
        pad, shift = pad & ~3, shift & ~3
    o = OL.Oracle(sw, sh, src, dw, dh, dst, flags, **(opts or {}))
    p = SwsContext(sw, sh, src, dw, dh, dst, flags, **(opts or {}))
    for k, v in (tune or {}).items():
        p.set_option(k, v)
    if colorspace:
        rc = o.set_colorspace(*colorspace)
        assert rc == p.set_colorspace(*colorspace)
        if rc < 0:
            return None
    n = nframes
    refs, srcs, dsts = [], [], []
    for k in range(n):
        s = OL.fill_random(OL.Frame(src, sw, sh), seed + k)
        ref = OL.Frame(dst, dw, dh, fill=0x5A)
        assert o.scale(s, ref) >= 0
        refs.append(ref)
        hs = HostFrame(src, sw, sh)
        for a, b in zip(hs.planes, s.planes):
            a[:] = b
        sf = _odd_frame(src, sw, sh, pad, shift)
        srcs.append((_BottomUp(sf) if flip & 1 else sf).upload(hs))     # flip: 1 the sources, 2 the destinations are bottom-up pictures
        d = _odd_frame(dst, dw, dh, pad, shift)
        d.buf.fill_(0x5A)
        dsts.append(_BottomUp(d) if flip & 2 else d)
    torch.cuda.synchronize()
    for nn in sorted({n, 1}, reverse=True):
        assert p.scale_frames(srcs[:nn], dsts[:nn]) == nn
        p.sync()
        for k in range(nn):
            out = dsts[k].download()
            for i, (a, b, rb) in enumerate(zip(out.planes, refs[k].planes, out.row_bytes)):
                if dst in ("monob", "monow") and (dw & 7):
                    continue
                assert np.array_equal(a[:, :rb], b[:, :rb]), (src, dst, sw, sh, dw, dh, hex(flags), pad, shift, flip, k, i, p.path())
            # the padding bytes between rows stay untouched
            for i in range(dsts[k].nplanes):
                t = dsts[k].plane_tensor(i)
                if pad:
                    assert bool((t[:-1, dsts[k].row_bytes[i]:] == 0x5A).all()), (src, dst, k, i, p.path())
    return p.path()


@pytest.mark.parametrize("src,sw,sh,dst,dw,dh,flags", CASES)
@pytest.mark.parametrize("pad,shift,flip", [(2, 0, 0), (6, 2, 0), (0, 6, 0), (1, 3, 0), (0, 0, 1), (0, 0, 2), (6, 2, 3)])
def test_unaligned_device_frames(src, sw, sh, dst, dw, dh, flags, pad, shift, flip):
    run_odd(sw, sh, src, dw, dh, dst, flags | BX, pad, shift, flip, tune=dict(strip_min_w=0))
