#!/usr/bin/env python3
"""Collect the reference's fate-pixfmt known answers (tests/ref/pixfmt/*, data files of the reference's own test
suite) for the pixel formats in SURVEY.md §8's scope into tests/golden/fate_pixfmt_md5.json.

Each entry is the MD5 + byte count of a raw video file the reference's FATE run produces
(tests/fate-run.sh:575-597 pixfmt_conversion / pixfmt_conversion_ext, tests/fate/pixfmt.mak):
  "<fmt>"          vsynth1 frame 0 (yuv420p 352x288) -> <fmt> -> yuv444p, 1 frame
  "<base>-<fmt>"   {yuv,rgb}testsrc 352x288 in <base> -> <fmt> (sws_dither=none) -> <base>, 25 identical frames
all with sws flags bicubic+accurate_rnd+bitexact.  Run in the build container (needs /root/reference)."""
import json
import os
import sys

REF = os.path.join(os.environ.get("SWS_REFERENCE_ROOT", sys.argv[1] if len(sys.argv) > 1 else "/root/reference"), "tests/ref/pixfmt")   # reference tree: argv[1] or $SWS_REFERENCE_ROOT
SCOPE = ["bgr24", "nv12", "rgb24", "rgb32", "yuv420p", "yuv422p", "yuv444p", "yuvj420p", "yuv420p16le",
         "yuv444p16le", "yuv420p10le", "yuv444p10le", "p010le",
         "nv16", "nv24", "yuv410p", "yuv411p", "yuv440p", "yuvj422p", "yuvj440p", "yuvj444p",
         "yuv422p10le", "yuv440p10le", "yuv420p12le", "yuv422p12le", "yuv440p12le", "yuv444p12le", "yuv422p16le",
         "p210le", "p410le", "p012le", "p212le", "p412le", "p016le", "p216le", "p416le",
         "gbrp10le", "gbrp12le", "gbrp16le", "gray", "gray10le", "gray12le", "gray16le", "yuyv422", "yvyu422", "uyvy422", "rgb48",
         "rgb565", "rgb555", "vuyx", "vyu444", "y210le", "y212le", "y216le", "xv30le", "v30xle", "xv36le", "xv36be", "xv48le", "xv48be", "yuv444p10msble", "yuv444p10msbbe", "yuv444p12msble", "yuv444p12msbbe", "x2rgb10le", "x2bgr10le", "xyz12le", "monob", "monow",
         # big-endian twins
         "yuv420p10be", "yuv420p12be", "yuv420p16be", "yuv422p10be", "yuv422p12be", "yuv422p16be", "yuv440p10be", "yuv440p12be",
         "yuv444p10be", "yuv444p12be", "yuv444p16be", "gbrp10be", "gbrp12be", "gbrp16be", "gray10be", "gray12be", "gray16be",
         "p010be", "p012be", "p016be", "p210be", "p212be", "p216be", "p410be", "p412be", "p416be"]
BASES = ["", "yuv444p-", "rgb24-", "yuv444p10-", "yuv444p12-", "yuv444p16-", "nv24-", "p410-", "p412-", "p416-",
         "gbrp-", "gbrp10-", "gbrp12-", "gbrp16-", "rgb48-"]

out = {}
for b in BASES:
    for f in SCOPE:
        p = os.path.join(REF, b + f)
        if os.path.exists(p):
            tok = open(p).read().split()
            out[b + f] = {"md5": tok[0], "bytes": int(tok[2])}
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fate_pixfmt_md5.json")
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(len(out), "entries ->", dst, file=sys.stderr)
