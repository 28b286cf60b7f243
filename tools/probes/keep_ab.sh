# A/B of k_resample_polar's output (all 600 planes x 3 arrays against the rows the later stages read), 100-step regions, interleaved
for k in 1 2 3; do for y in 1 0; do
SH_BENCH_KEEP_PRODUCTS=$y python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('keep_products $y', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done
