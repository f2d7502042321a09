#!/usr/bin/env python3
"""The ORACLE side of the random GPU tests (tests/test_gpu_random.py: every generator) on the CPU box, meant for the sanitizer build:

    make -C oracle asan ; tools/asan_env.sh python tools/oracle_hunt.py <N per generator> <seed>

In a GPU parity test the oracle converts in the SAME process as the product, between the product context's host-side init (filter banks on the heap) and its first
sws_scale() (which uploads them): an oracle that writes a few bytes past one of its line buffers for some odd geometry would damage the product's banks -- wrong for that
context's lifetime, right in a fresh one, never alone -- i.e. the forensics of DESIGN.md 8.  The goldens exercise the oracle at a handful of sizes only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("SWS_RANDOM_N", "1")
import test_gpu_random as R  # noqa: E402
import oracle_lib as OL  # noqa: E402


def main():
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    gens = [("conversions", R._cases(n, seed)), ("options", R._opt_cases(n, seed + 1)), ("strip family", R._strip_cases(n, seed + 2)),
            ("round-4 routes", R._strip_cases(n, seed + 3, R.R4_SRC, R.R4_DST)), ("few rows", R._short_cases(n, seed + 4)),
            ("batches", R._batch_cases(n, seed + 5)), ("unaligned", R._odd_cases(n, seed + 6)), ("slice sequences (whole frames)", R._slice_cases(n, seed + 7))]
    if os.environ.get("SWS_HUNT_ONLY"):
        gens = [g for g in gens if os.environ["SWS_HUNT_ONLY"] in g[0]]
    total = refused = 0
    import hashlib
    dump = open(os.environ["SWS_ORACLE_DUMP"], "w") if os.environ.get("SWS_ORACLE_DUMP") else None      # one digest per conversion: two oracle builds (SWS_ORACLE_LIBRARY) must agree
    for name, cases in gens:
        for c in cases:
            sw, sh, sf, dw, dh, df, flags = c[:7]
            opts = next((x for x in c[7:] if isinstance(x, dict) and ("dither" in x or "src_range" in x or "threads" in x)), None)
            cs = next((x for x in c[7:] if isinstance(x, tuple) and len(x) == 7 and all(isinstance(v, int) for v in x)), None)
            total += 1
            try:
                o = OL.Oracle(sw, sh, sf, dw, dh, df, flags, **(opts or {}))
            except Exception:
                refused += 1
                continue
            if cs and o.set_colorspace(*cs) < 0:
                refused += 1
                continue
            src = OL.fill_random(OL.Frame(sf, sw, sh), total)
            ref = OL.Frame(df, dw, dh, fill=0xA5)
            o.scale(src, ref)
            if dump:
                digest = hashlib.md5(b"".join(a.tobytes() for a in ref.planes)).hexdigest()
                dump.write(f"{name} {c[:7]} {digest}\n")
            if total % 5 == 0:      # twice on one context: the answer must repeat
                ref2 = OL.Frame(df, dw, dh, fill=0xA5)
                o.scale(src, ref2)
                if any((a != b).any() for a, b in zip(ref.planes, ref2.planes)):
                    print("ORACLE ANSWER CHANGED", c[:7], flush=True)
            del o
        print(f"{name}: {len(cases)} cases", flush=True)
    print(f"total {total}, refused {refused}")


if __name__ == "__main__":
    main()
