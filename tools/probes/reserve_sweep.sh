# the fixed CU reserve of the two-lane schedule (SHOULDER_CU_RESERVE), 100-step regions, interleaved
for k in 1 2; do for r in ${RESERVES:-24 28 32 36}; do
SHOULDER_CU_RESERVE=$r python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reserve $r', d['value'], d['ms_per_step'])"
done; done
