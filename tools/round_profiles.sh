#!/bin/bash
# the profile set of a round (tag r05, r06 ...; one parameterised script since round 6, the per-round copies of rounds 3 - 5 are gone): rocprofv3 kernel stats of every bench workload + the bench lines, PMC HBM traffic, layout / common-shape / conversion
# tables and the kernel stats of the common-shape run.  usage: tools/round_profiles.sh <tag>  -> gpurun_out/<tag>p/, then tools/round_collect.sh <tag> copies the summaries into profiles/<tag>_*
set -u
TAG=${1:?round tag, e.g. r06}
OUT=$PWD/gpurun_out/${TAG}p; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
tools/profile_all.sh ${TAG}p > $OUT/bench_lines.jsonl 2>$OUT/profile_all.err
tools/pmc_traffic.sh ${TAG}p "c2a c2b c4 c3a c3b c5 c1 d1 e2 r1 w1 f1 u1" > $OUT/pmc_traffic.txt 2>&1
python tools/layout_times.py > $OUT/layout.md 2>$OUT/layout.err
python tools/common_shapes_times.py > $OUT/common.md 2>$OUT/common.err
python tools/rgb2rgb_times.py > $OUT/rgb2rgb.md 2>$OUT/rgb2rgb.err
python tools/common_conversions_times.py > $OUT/conv.txt 2>$OUT/conv.err
python tools/aux_kernel_times.py > $OUT/aux.txt 2>$OUT/aux.err
python tools/single_frame_times.py > $OUT/single.md 2>$OUT/single.err
SWS_SHAPES_SET=ladder python tools/common_shapes_times.py > $OUT/ladder.md 2>$OUT/ladder.err
SWS_SHAPES_SET=range python tools/common_shapes_times.py > $OUT/range.md 2>$OUT/range.err
SWS_SHAPES_SET=wide python tools/common_shapes_times.py > $OUT/wide.md 2>$OUT/wide.err
SWS_SHAPES_SET=u16 python tools/common_shapes_times.py > $OUT/u16.md 2>$OUT/u16.err
{ python tools/narrow_shapes_times.py; SWS_NARROW_SET=small python tools/narrow_shapes_times.py; } 2>$OUT/narrow.err | grep '^|' > $OUT/narrow.md
for m in same down up same4k; do python tools/format_survey.py $m > $OUT/survey_$m.md 2>$OUT/survey_$m.err; done
# the same surveys with the flags players and capture tools pass (SWS_BILINEAR = 2, SWS_FAST_BILINEAR = 1, SWS_POINT = 0x10), and the shapes with the options switched off
for m in same4k down up; do
  SWS_SURVEY_FLAGS=2 python tools/format_survey.py $m > $OUT/flags_bilinear_$m.md 2>/dev/null
  SWS_SURVEY_FLAGS=1 python tools/format_survey.py $m > $OUT/flags_fastbilinear_$m.md 2>/dev/null
done
SWS_SURVEY_FLAGS=16 python tools/format_survey.py down > $OUT/flags_point_down.md 2>/dev/null
python tools/hdr_capture_times.py > $OUT/hdr_capture.md 2>$OUT/hdr_capture.err
python tools/flags_before_after.py > $OUT/flags_before_after.md 2>$OUT/flags_before_after.err
python bench.py > $OUT/bench_default.json 2>$OUT/bench_default.err
(cd /tmp && SWS_SHAPES_SET=ladder rocprofv3 --kernel-trace --stats -d $OUT/prof_ladder -o res -- python $ROOT/tools/common_shapes_times.py > $OUT/prof_ladder.log 2>&1)
db=$(find $OUT/prof_ladder -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py "$db" "r05 ladder shapes (ratios of 3:1 and more): SWS_SHAPES_SET=ladder rocprofv3 --kernel-trace --stats -- python tools/common_shapes_times.py" > $OUT/kernel_stats_ladder.md
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_common -o res -- python $ROOT/tools/common_shapes_times.py > $OUT/prof_common.log 2>&1)
db=$(find $OUT/prof_common -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py "$db" "r05 common shapes: rocprofv3 --kernel-trace --stats -- python tools/common_shapes_times.py" > $OUT/kernel_stats_common_shapes.md
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_layout -o res -- python $ROOT/tools/layout_times.py > $OUT/prof_layout.log 2>&1)
db=$(find $OUT/prof_layout -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py "$db" "r05 layout converters (4K): rocprofv3 --kernel-trace --stats -- python tools/layout_times.py" > $OUT/kernel_stats_layout.md
rm -rf $OUT/prof_common $OUT/prof_layout $OUT/prof_ladder $OUT/traffic_*_FETCH_SIZE $OUT/traffic_*_WRITE_SIZE
ls $OUT | head -60
