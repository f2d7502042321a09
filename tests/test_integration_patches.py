"""integration/patches/*.patch apply to the reference tree they were cut against (checked here, where /root/reference exists;
`patch --dry-run` writes nothing)."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF) or not shutil.which("patch"), reason="needs the reference tree and patch(1)")
def test_patches_apply_to_the_reference():
    patches = sorted(glob.glob(os.path.join(ROOT, "integration", "patches", "*.patch")))
    assert len(patches) == 3
    for p in patches:
        r = subprocess.run(["patch", "-p1", "--dry-run", "--force", "-d", REF, "-i", p], capture_output=True, text=True)
        assert r.returncode == 0 and "FAILED" not in r.stdout and "fuzz" not in r.stdout, (p, r.stdout, r.stderr)


def test_patch_set_touches_what_integration_md_says():
    text = "".join(open(p).read() for p in sorted(glob.glob(os.path.join(ROOT, "integration", "patches", "*.patch"))))
    for needle in ("AV_HWDEVICE_TYPE_HIP", "AV_PIX_FMT_HIP", "ff_hwcontext_type_hip", "swship_scale_frame", "swship_frame_setup",
                   "ff_vf_scale_hip", "hwcontext_hip.o", "vf_scale_hip.o"):
        assert needle in text, needle
