for dbg in 0 8; do
for w in c3b c1; do
    echo -n "dbg=$dbg $w: "
    SWS_HIP_TILE_DEBUG=$dbg python bench.py --workload $w --variants none --no-cpu --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_avg'], d['ms_per_step'])"
done; done
