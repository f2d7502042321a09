#!/bin/bash
# copies what tools/r03_profiles.sh left under gpurun_out/r03p/ into the tracked profiles/r03_* files
set -u
R=gpurun_out/r03p; P=profiles
cp $R/bench_lines.jsonl $P/r03_bench.jsonl
for k in c1 c2a c2b c3a c3b c4 c5 d1 d2 common_shapes layout ladder; do [ -f $R/kernel_stats_$k.md ] && cp $R/kernel_stats_$k.md $P/r03_kernel_stats_$k.md; done
cp $R/common.md $P/r03_common_shapes.md; cp $R/conv.txt $P/r03_common_conversions.txt; cp $R/aux.txt $P/r03_aux_kernels.md; cp $R/layout.md $P/r03_layout_times.md; cp $R/single.md $P/r03_single_frame.md
cp $R/narrow.md $P/r03_narrow_shapes.md
for m in same down up; do grep "^|" $R/survey_$m.md > $P/r03_survey_$m.md; done
{ echo "# Ratios of 3:1 and more (16 frames per call, wall time per frame), tools/common_shapes_times.py with SWS_SHAPES_SET=ladder"; echo
  echo "## before the strip kernel's long forms (filters of more than 16 taps on the tile / two-pass kernels; the library of commit 178d04a, and of later commits with the long forms switched off -- SWSOPT_NO_MIXED=1 -- for the last seven rows)"
  grep "^|" gpurun_out/a5/ladder_before.txt; grep "^|" gpurun_out/b5_before.txt 2>/dev/null
  echo; echo "## with sws_k_strip_long / sws_k_strip_xlong and the sws_k_lut_rgb epilogue (final library)"; grep "^|" $R/ladder.md; } > $P/r03_ladder.md
python - <<'PY'
import json,subprocess
txt=open("gpurun_out/r03p/pmc_traffic.txt").read()
d=json.loads(txt[txt.index("{"):txt.rindex("}")+1])
commit=subprocess.check_output(["git","rev-parse","--short","HEAD"]).decode().strip()
d["_source"]=f"tools/r03_profiles.sh (tools/pmc_traffic.sh) on 1x MI355X, library at commit {commit}"
json.dump(d,open("profiles/r03_pmc_traffic.json","w"),indent=1); json.dump(d,open("profiles/pmc_latest.json","w"),indent=1)
PY
