for ch in 0 1 2 4; do
    echo -n "P01X_CH=$ch c3a: "
    SWS_HIP_P01X_CH=$ch python bench.py --workload c3a --variants none --no-cpu --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_avg'], d['ms_per_step'], d['roofline']['frac'])"
done
echo -n "c5: "; python bench.py --workload c5 --variants none --no-cpu --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_avg'], d['ms_per_step'], d['roofline']['frac'])"
