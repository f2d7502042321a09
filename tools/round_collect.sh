#!/bin/bash
# copies what tools/round_profiles.sh <tag> left under gpurun_out/<tag>p/ into the tracked profiles/<tag>_* files.  usage: tools/round_collect.sh <tag>
set -u
TAG=${1:?round tag, e.g. r06}
R=gpurun_out/${TAG}p; P=profiles
cp $R/bench_lines.jsonl $P/${TAG}_bench.jsonl
for k in c1 c2a c2b c3a c3b c4 c5 d1 d2 e1 e2 r1 r2 w1 f1 u1 common_shapes layout ladder; do [ -f $R/kernel_stats_$k.md ] && cp $R/kernel_stats_$k.md $P/${TAG}_kernel_stats_$k.md; done
cp $R/common.md $P/${TAG}_common_shapes.md; cp $R/conv.txt $P/${TAG}_common_conversions.txt; cp $R/aux.txt $P/${TAG}_aux_kernels.md; cp $R/layout.md $P/${TAG}_layout_times.md; cp $R/single.md $P/${TAG}_single_frame.md
cp $R/narrow.md $P/${TAG}_narrow_shapes.md
cp $R/rgb2rgb.md $P/${TAG}_rgb2rgb.md
for m in same down up same4k; do grep "^|" $R/survey_$m.md > $P/${TAG}_survey_$m.md; done
for f in $R/flags_*.md; do grep "^|" $f > $P/${TAG}_$(basename $f); done
grep "^|" $R/hdr_capture.md > $P/${TAG}_hdr_capture.md
cp $R/ladder.md $P/${TAG}_ladder.md
cp $R/range.md $P/${TAG}_range_shapes.md; cp $R/wide.md $P/${TAG}_wide_shapes.md; cp $R/u16.md $P/${TAG}_u16_shapes.md
cp $R/bench_default.json $P/${TAG}_bench_default.json 2>/dev/null
python - "$TAG" <<'PY'
import json,subprocess,sys
TAG=sys.argv[1]
d=json.load(open("gpurun_out/"+TAG+"p/pmc_latest.json"))      # tools/pmc_traffic.sh: per workload, all sws_k* dispatches summed per sws_scale_frames() call
commit=subprocess.check_output(["git","rev-parse","--short","HEAD"]).decode().strip()
d["_tool_version"]=2
d["_source"]=f"tools/round_profiles.sh (tools/pmc_traffic.sh) on 1x MI355X, library at commit {commit}"
json.dump(d,open("profiles/"+TAG+"_pmc_traffic.json","w"),indent=1); json.dump(d,open("profiles/pmc_latest.json","w"),indent=1)
PY
