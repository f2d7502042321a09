#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (--kernel-trace [--stats] or --pmc) as a markdown table.

usage: tools/rocpd_summary.py results.db [title] > profiles/xxx.md
Per kernel: calls, total / avg / min / max duration (from the dispatch start/end timestamps), grid,
workgroup, VGPR/SGPR/LDS where the DB has them, and PMC counter sums per dispatch when present.
"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    cur = db.cursor()
    cols = [d[1] for d in cur.execute("pragma table_info('kernels')")]
    rows = cur.execute("select * from kernels").fetchall()
    ix = {c: i for i, c in enumerate(cols)}
    stats = defaultdict(list)
    meta = {}
    for r in rows:
        name = r[ix["name"]] if "name" in ix else r[ix.get("kernel_name", 0)]
        short = name.split("(")[0]
        dur = r[ix["end"]] - r[ix["start"]]
        stats[short].append(dur)
        meta[short] = {k: r[ix[k]] for k in ("grid_x", "grid_y", "grid_z", "workgroup_x", "vgpr_count", "accum_vgpr_count",
                                                "sgpr_count", "lds_size", "scratch_size") if k in ix}
    print(f"# {title}\n")
    print("| kernel | calls | total ms | avg us | min us | max us | grid | wg | vgpr | sgpr | lds |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for k, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        m = meta[k]
        print(f"| `{k}` | {len(v)} | {sum(v)/1e6:.3f} | {sum(v)/len(v)/1e3:.2f} | {min(v)/1e3:.2f} | {max(v)/1e3:.2f} | "
              f"{m.get('grid_x')}x{m.get('grid_y')}x{m.get('grid_z')} | {m.get('workgroup_x')} | {m.get('vgpr_count')} | "
              f"{m.get('sgpr_count')} | {m.get('lds_size')} |")
    # PMC counters, if any
    try:
        pc = [d[1] for d in cur.execute("pragma table_info('counters_collection')")]
        prow = cur.execute("select * from counters_collection").fetchall()
    except sqlite3.Error:
        prow = []
    if prow:
        pi = {c: i for i, c in enumerate(pc)}
        agg = defaultdict(lambda: defaultdict(list))
        for r in prow:
            kn = str(r[pi.get("kernel_name", pi.get("name", 0))]).split("(")[0]
            cn = r[pi.get("counter_name", pi.get("pmc_name", 0))]
            val = r[pi.get("value", pi.get("counter_value", 0))]
            did = r[pi.get("dispatch_id", 0)]
            agg[kn][cn].append((did, val))
        print("\n## PMC counters (sum over dimensions, averaged per dispatch)\n")
        print("| kernel | counter | dispatches | avg per dispatch |")
        print("|---|---|---|---|")
        for kn, cs in agg.items():
            for cn, vals in cs.items():
                per = defaultdict(float)
                for did, v in vals:
                    per[did] += v
                print(f"| `{kn}` | {cn} | {len(per)} | {sum(per.values())/len(per):.1f} |")


if __name__ == "__main__":
    main()
