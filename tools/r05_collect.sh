#!/bin/bash
# copies what tools/r05_profiles.sh left under gpurun_out/r05p/ into the tracked profiles/r05_* files
set -u
R=gpurun_out/r05p; P=profiles
cp $R/bench_lines.jsonl $P/r05_bench.jsonl
for k in c1 c2a c2b c3a c3b c4 c5 d1 d2 e1 e2 r1 r2 w1 f1 u1 common_shapes layout ladder; do [ -f $R/kernel_stats_$k.md ] && cp $R/kernel_stats_$k.md $P/r05_kernel_stats_$k.md; done
cp $R/common.md $P/r05_common_shapes.md; cp $R/conv.txt $P/r05_common_conversions.txt; cp $R/aux.txt $P/r05_aux_kernels.md; cp $R/layout.md $P/r05_layout_times.md; cp $R/single.md $P/r05_single_frame.md
cp $R/narrow.md $P/r05_narrow_shapes.md
cp $R/rgb2rgb.md $P/r05_rgb2rgb.md
for m in same down up same4k; do grep "^|" $R/survey_$m.md > $P/r05_survey_$m.md; done
for f in $R/flags_*.md; do grep "^|" $f > $P/r05_$(basename $f); done
grep "^|" $R/hdr_capture.md > $P/r05_hdr_capture.md
cp $R/ladder.md $P/r05_ladder.md
cp $R/range.md $P/r05_range_shapes.md; cp $R/wide.md $P/r05_wide_shapes.md; cp $R/u16.md $P/r05_u16_shapes.md
cp $R/bench_default.json $P/r05_bench_default.json 2>/dev/null
python - <<'PY'
import json,subprocess
txt=open("gpurun_out/r05p/pmc_traffic.txt").read()
d=json.loads(txt[txt.index("{"):txt.rindex("}")+1])
commit=subprocess.check_output(["git","rev-parse","--short","HEAD"]).decode().strip()
d["_source"]=f"tools/r05_profiles.sh (tools/pmc_traffic.sh) on 1x MI355X, library at commit {commit}"
json.dump(d,open("profiles/r05_pmc_traffic.json","w"),indent=1); json.dump(d,open("profiles/pmc_latest.json","w"),indent=1)
PY
