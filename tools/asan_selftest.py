#!/usr/bin/env python3
"""Does the sanitizer set-up of this process SEE an overflow made by the library's host code?  Hands sws_hip_image_layout() a three-entry offset array where it writes four
(24 bytes: ctypes keeps objects of up to 16 bytes inline, and python's small-object arenas hide a buffer from the sanitizer -- hence PYTHONMALLOC=malloc in tools/asan_env.sh):
under tools/asan_env.sh (or the gcc-runtime variant) the run must end in an AddressSanitizer report, not in "returned"."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from librempeg_amd import swscale as S  # noqa: E402

L = S.load_library()
ls = (C.c_int * 4)()
offs = (C.c_size_t * 3)()
tot = C.c_size_t()
L.sws_hip_image_layout.argtypes = [C.c_int] * 4 + [C.c_void_p, C.c_void_p, C.c_void_p]
print("calling with an undersized array ...", flush=True)
L.sws_hip_image_layout(0, 64, 36, 256, C.addressof(ls), C.addressof(offs), C.addressof(tot))
print("returned (no report)")
