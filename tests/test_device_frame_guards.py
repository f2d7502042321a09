"""CPU box: the guard bands every DeviceFrame of the GPU tests carries (librempeg_amd/swscale.py) -- with CPU tensors standing in for HBM.  download() must pass on an
untouched frame whatever the test wrote inside `buf`, and name a write on either side of it."""
import pytest

from librempeg_amd.swscale import DeviceFrame


@pytest.mark.parametrize("fmt,w,h", [("yuv420p", 64, 36), ("rgb24", 33, 7), ("p010le", 130, 18), ("pal8", 17, 5)])
def test_guards(fmt, w, h):
    f = DeviceFrame(fmt, w, h, device="cpu")
    assert f.base % 256 == 0 and f.buf.numel() == f.total + 256
    f.buf.fill_(0x5A)                      # what the tests' prefills do: inside the bands
    f.download()
    f._alloc[f.GUARD - 1] = 0              # the byte just before the frame's storage
    with pytest.raises(AssertionError, match="1 guard bytes BEFORE"):
        f.download()
    f._alloc[f.GUARD - 1] = f.GUARD_BYTE
    f._alloc[f.GUARD + f.buf.numel()] = 7  # the byte just after it
    with pytest.raises(AssertionError, match="1 AFTER"):
        f.check_guards()


def test_replaced_storage_is_left_alone():
    import torch
    f = DeviceFrame("gray8", 16, 16, device="cpu")
    f.buf = torch.zeros(16 * 256 + 256, dtype=torch.uint8)      # (tests/test_gpu_guard_bands.py builds its own surroundings)
    f.check_guards()
