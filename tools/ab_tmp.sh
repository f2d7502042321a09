for w in c3b c3b c1 c2b; do
    python bench.py --workload $w --variants none --no-cpu --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$w', d['ms_per_step'], round(d['roofline']['frac'],3))"
done
