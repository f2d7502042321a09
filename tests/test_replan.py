"""CPU box, no GPU: a context that is planned again under other launch options ends up with the plan of a fresh context (tools/replan_hunt.py in small).

Round 6 found -- with exactly this comparison -- that a re-plan kept flags and geometries of the plan before (`strip_ok` under `no_strip` ...): a context whose options
change after its first conversion then ran a path the new plan had not built, over tables the new plan had rewritten."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_replanned_contexts_equal_fresh_ones():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "replan_hunt.py"), "700", "20260930"], capture_output=True, text=True, timeout=600)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]
    assert r.returncode == 0 and " 0 differ" in tail, r.stdout[-3000:]
