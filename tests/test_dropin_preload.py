"""The drop-in claim of INTEGRATION.md section 1, literally: a binary that was linked against a `libswscale.so.10` exporting
`sws_*@LIBSWSCALE_10` (the reference's SONAME and version node: libswscale/libswscale.v, ffbuild/library.mak:83) binds to this library
(a) when librempeg_amd/lib/libswscale_hip.so is LD_PRELOADed and (b) when librempeg_amd/lib/dropin/libswscale.so.10 is found first on the
library path.  The stand-in for the reference's shared library is a stub built here (it only has to carry the same SONAME, symbol names
and version node; its answers are markers).  No GPU: the caller only makes host-side calls."""
import os
import subprocess
import textwrap

import pytest

import librempeg_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "librempeg_amd", "lib")

STUB_C = """
/* stand-in for the reference's libswscale.so.10: same exported names, marker answers */
unsigned swscale_version(void) { return 0xDEAD; }
const char *swscale_configuration(void) { return "stub"; }
int sws_isSupportedInput(int f) { (void)f; return -77; }
void *sws_getContext(int a, int b, int c, int d, int e, int f, int g, void *h, void *i, const double *j)
{ (void)a; (void)b; (void)c; (void)d; (void)e; (void)f; (void)g; (void)h; (void)i; (void)j; return 0; }
void sws_freeContext(void *c) { (void)c; }
int sws_scale(void *c, const unsigned char *const s[], const int ss[], int y, int h, unsigned char *const d[], const int ds[])
{ (void)c; (void)s; (void)ss; (void)y; (void)h; (void)d; (void)ds; return -77; }
"""

STUB_V = "LIBSWSCALE_10 { global: swscale_*; sws_*; local: *; };\n"

CALLER_C = """
#include <stdio.h>
#include "swscale_hip.h"   /* the declarations of libswscale/swscale.h (include/swscale_hip.h cites each one) */
int main(void)
{
    SwsContext *c = sws_getContext(64, 48, AV_PIX_FMT_YUV420P, 32, 24, AV_PIX_FMT_RGB24, SWS_BICUBIC | SWS_BITEXACT, NULL, NULL, NULL);
    printf("version=%u config=%s in=%d ctx=%d\\n", swscale_version(), swscale_configuration(), sws_isSupportedInput(AV_PIX_FMT_YUV420P), c != NULL);
    sws_freeContext(c);
    return 0;
}
"""


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    librempeg_amd.load_library()   # builds the library (and lib/dropin/libswscale.so.10) when it is missing
    d = tmp_path_factory.mktemp("dropin")
    (d / "stub.c").write_text(STUB_C)
    (d / "stub.v").write_text(STUB_V)
    (d / "caller.c").write_text(textwrap.dedent(CALLER_C))
    os.makedirs(d / "stublib")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-Wl,-soname,libswscale.so.10", f"-Wl,--version-script={d / 'stub.v'}",
                           "-o", str(d / "stublib" / "libswscale.so.10"), str(d / "stub.c")])
    os.symlink("libswscale.so.10", d / "stublib" / "libswscale.so")
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(d / "caller"), str(d / "caller.c"),
                           "-L", str(d / "stublib"), "-lswscale"])
    return d


def _run(d, **env):
    e = dict(os.environ)
    e.update(env)
    return subprocess.run([str(d / "caller")], env=e, capture_output=True, text=True, timeout=120)


def test_caller_asks_for_the_reference_names(built):
    out = subprocess.check_output(["readelf", "-W", "-d", "--dyn-syms", str(built / "caller")], text=True)
    assert "libswscale.so.10" in out                      # DT_NEEDED = the reference's SONAME
    assert "sws_getContext@LIBSWSCALE_10" in out          # versioned reference symbol


def test_stub_answers_without_substitution(built):
    r = _run(built, LD_LIBRARY_PATH=str(built / "stublib"))
    assert r.returncode == 0 and "version=57005 config=stub in=-77 ctx=0" in r.stdout, r.stdout + r.stderr


def test_ld_preload_takes_the_calls(built):
    r = _run(built, LD_LIBRARY_PATH=str(built / "stublib"), LD_PRELOAD=os.path.join(LIBDIR, "libswscale_hip.so"))
    assert r.returncode == 0, r.stderr
    assert f"version={(10 << 16) | (2 << 8) | 100} config=hip gfx950" in r.stdout, r.stdout + r.stderr
    assert "in=1 ctx=1" in r.stdout     # host-side init runs without a GPU; only compute calls need the device


def test_library_path_substitution(built):
    so10 = os.path.join(LIBDIR, "dropin", "libswscale.so.10")
    assert os.path.exists(so10)
    out = subprocess.check_output(["readelf", "-d", so10], text=True)
    assert "Library soname: [libswscale.so.10]" in out
    r = _run(built, LD_LIBRARY_PATH=os.path.join(LIBDIR, "dropin"))
    assert r.returncode == 0, r.stderr
    assert "config=hip gfx950" in r.stdout and "ctx=1" in r.stdout, r.stdout + r.stderr
