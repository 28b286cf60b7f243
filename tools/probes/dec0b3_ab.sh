# A/B of the three-buffer dec0b + head kernel (SHOULDER_DEC0B3), interleaved 100-step regions
for k in 1 2 3; do for y in 0 1; do
SHOULDER_DEC0B3=$y timeout -k 10 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=dict((k,v) for k,v in d['device_ms_per_step_top']); print('dec0b3 $y', d['value'], d['ms_per_step'], {k:v for k,v in t.items() if 'dma16' in k or 'dec0b' in k})"
done; done
