"""bug-hunt helper: run variations of one failing parity case on the GPU and print which pass (usage: python tools/diag_case.py)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from test_gpu_parity import run_case

BASE = dict(sw=1836, sh=3, sf="gbrp14le", dw=1512, dh=39, df="y212le", flags=270338, seed=9759 + 5,
            opts={'dither': 1, 'src_range': 1, 'dst_range': 0, 'src_h_chr_pos': -513, 'src_v_chr_pos': 256, 'dst_h_chr_pos': -513, 'dst_v_chr_pos': 256}, tune={}, dev=bool(9759 % 3))


def go(name, **kw):
    c = dict(BASE); c.update(kw)
    try:
        r = run_case(c["sw"], c["sh"], c["sf"], c["dw"], c["dh"], c["df"], c["flags"], seed=c["seed"], device_frames=c["dev"], opts=c["opts"] or None, tune=c["tune"])
        print(f"ok    {name}: {r[0] if r else r}")
    except AssertionError as e:
        print(f"FAIL  {name}: {str(e)[:260]}")
    except Exception as e:
        print(f"ERR   {name}: {type(e).__name__} {str(e)[:160]}")
    sys.stdout.flush()


O = {'src_v_chr_pos': 256}
go("base2", opts=O)
for t in ("no_strip_fuse", "no_strip_short", "no_strip_dma", "no_strip_dma8", "no_mixed"):
    go("tune " + t, opts=O, tune={t: 1})
go("tune strip_min_rows 16", opts=O, tune={"strip_min_rows": 16})
go("tune strip_cols_l2 c1", opts=O, tune={"strip_cols_l": 2, "strip_cols_c": 1})
for sf in ("yuv420p", "yuv422p", "nv12", "yuv444p", "yuv422p10le", "bgra"):
    go("src " + sf + " sh4->39", sf=sf, sh=4, opts=O)
    go("src " + sf + " sh2->39", sf=sf, sh=2, opts=O)
for df in ("bgra", "rgb24", "gbrp", "rgb565le", "vuya", "yuv422p12le", "yuv444p"):
    go("rgb24 -> " + df + " flags bilinear", sf="rgb24", df=df, flags=2 | 0x80000, opts=O)
    go("rgb24 -> " + df + " flags bilinear|fullchr", sf="rgb24", df=df, flags=2 | 0x80000 | 0x2000, opts=O)
for v in (128, 384, 512, -256):
    go(f"src_v_chr_pos {v}", opts={'src_v_chr_pos': v})
    go(f"src_v_chr_pos {v} sh 6", sh=6, opts={'src_v_chr_pos': v})
    go(f"src_v_chr_pos {v} sh 12 dh 7", sh=12, dh=7, opts={'src_v_chr_pos': v})
go("dst_v_chr_pos 512 sh3", opts={'dst_v_chr_pos': 512})
go("sh 3 dh 39 lanczos", flags=0x200 | 0x80000, opts=O)
