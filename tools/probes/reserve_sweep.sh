for r in 16 24 32 40; do for k in 1 2; do
SHOULDER_CU_RESERVE=$r python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reserve $r', d['value'], d['ms_per_step'])"
done; done
