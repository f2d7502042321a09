import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hiplib():
    """The product C-ABI library.  Built in-tree if missing (hipcc cross-compiles without a GPU)."""
    import librempeg_amd
    if not os.path.exists(librempeg_amd.library_path()):
        librempeg_amd.build_library()
    return librempeg_amd.load_library()


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


# SWS_SUITE_ON_EMU=1 (CPU box, by hand): the -m gpu tests against the x86 EMULATION build of the library (tests/hipemu/README.md) -- their "device" frames become
# pageable host tensors, which the library stages through its own (test-double) device buffers like any host frame.  Not a GPU result; it holds the suite's specific
# cases (reference goldens, slices, cascades, filters ...) to the oracle while no GPU is at hand:
#   LD_PRELOAD=tests/hipstub/libhipstub.so SWS_HIP_LIBRARY=tests/hipemu/libswscale_hip_emu.so SWS_SUITE_ON_EMU=1 python -m pytest tests -m gpu -q -n 6
if os.environ.get("SWS_SUITE_ON_EMU") == "1":
    import torch
    from librempeg_amd import swscale as _S
    torch.cuda.synchronize = lambda *a, **k: None
    _device_frame_init = _S.DeviceFrame.__init__
    _device_frame_alloc = _S.DeviceFrame._allocate
    _S.DeviceFrame.__init__ = lambda self, fmt, w, h, device="cpu": _device_frame_init(self, fmt, w, h, "cpu")
    _S.DeviceFrame._allocate = lambda self, nbytes, device: _device_frame_alloc(self, nbytes, "cpu")
