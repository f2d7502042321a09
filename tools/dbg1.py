import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from test_gpu_parity import run_case
from librempeg_amd import *
import traceback
for args in ((224,36,"nv21",224,36,"yuv420p10be",2), (89,49,"yuv444p",226,8,"yuyv422",4), (6,33,"rgb24",224,49,"yuyv422",0x200), (224,36,"nv12",224,36,"yuv420p10le",2), (224,36,"yuv420p",224,36,"yuv420p10be",2),(224,36,"nv21",224,36,"yuv420p10be",4)):
    for tune in (None, dict(no_short_forms=1), dict(no_fast_banks=1), dict(no_wave=1), dict(no_mixed=1)):
        try:
            r = run_case(*args[:6], args[6], seed=3, tune=tune)
            print(args, tune, "OK", r[0])
        except AssertionError as e:
            print(args, tune, "FAIL", str(e)[:260])
        except Exception as e:
            print(args, tune, "EXC", repr(e)[:200])
