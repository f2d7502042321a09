"""CPU box: the x86 emulation build of the product (tests/hipemu/README.md: the library's own sources, kernels included, run thread by thread over the HIP runtime test
double) against the oracle, over the random generators of the GPU tests, in both fiber schedules.  Test infrastructure: it holds the host side and the kernels' C++ to the
oracle's pictures while no GPU is at hand; it is not a GPU parity result and replaces none.  Skipped when the emulation library has not been built
(`make -C tests/hipstub && make -C tests/hipemu`, about four minutes)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "hipemu", "libswscale_hip_emu.so")
STUB = os.path.join(ROOT, "tests", "hipstub", "libhipstub.so")

pytestmark = pytest.mark.skipif(not (os.path.exists(EMU) and os.path.exists(STUB)), reason="tests/hipemu/libswscale_hip_emu.so is not built (make -C tests/hipstub && make -C tests/hipemu)")


def _fresh():
    """the emulation library must be at least as new as the product sources it was made from"""
    src = os.path.join(ROOT, "librempeg_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(src, f)) for f in os.listdir(src) if f.endswith((".hip", ".hpp", ".cpp", ".h")))
    return os.path.getmtime(EMU) >= newest


@pytest.mark.parametrize("reverse,seed", [(0, 20260930), (1, 20261001)])
def test_emulated_library_gives_the_oracles_pictures(reverse, seed):
    if not _fresh():
        pytest.skip("tests/hipemu/libswscale_hip_emu.so is older than librempeg_amd/csrc: make -C tests/hipemu")
    env = dict(os.environ, LD_PRELOAD=STUB, SWS_HIP_LIBRARY=EMU, SWS_HIP_NO_TORCH="1", HIPEMU_REVERSE=str(reverse), HIPSTUB_DEVICES="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipemu_parity.py"), "30", str(seed)], capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert "'different': 0" in last and "'product call fails': 0" in last, last
    assert int(last.split("'compared': ")[1].split(",")[0]) >= 180, last
