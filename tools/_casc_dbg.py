import sys
sys.path.insert(0, 'tests')
from test_gpu_parity import run_case
for (w, h) in ((25, 93), (24, 93), (25, 92), (26, 94)):
  for sf in ("yuv420p14le", "yuv420p"):
    try:
        run_case(w, h, sf, w, h, "p010le", 524320, seed=5, device_frames=False, opts={'dither': 0, 'src_range': 1, 'dst_range': 1, 'threads': 1})
        print(w, h, sf, "ok")
    except AssertionError as e:
        print(w, h, sf, "FAIL", str(e)[:300])
