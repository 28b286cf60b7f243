# A/B of the yielding reserve (SHOULDER_CU_YIELD), interleaved runs of the headline leg
for k in 1 2 3; do for y in 0 1; do
SHOULDER_CU_YIELD=$y python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('yield $y', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done
