mkdir -p gpurun_out/stress
for t in a b c; do python tools/stress_batches.py 400 $t exact > gpurun_out/stress/exact_$t.log 2>&1 & done
python tools/stress_batches.py 400 d near > gpurun_out/stress/near_d.log 2>&1 &
wait
tail -n 3 gpurun_out/stress/*.log
grep -h MISMATCH gpurun_out/stress/*.log | head -20
