# A/B of the grouped decoder tail (SHOULDER_UNET_GROUP=G images per group; 0 = whole batch per layer), interleaved 100-step regions
for k in 1 2; do for g in ${GROUPS_:-0 8 16 32}; do
SHOULDER_UNET_GROUP=$g timeout -k 10 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('group $g', d['value'], d['ms_per_step'], d['config'].get('meshes_with_error_status'), d['parity'].get('records_equal_across_lanes') if isinstance(d.get('parity'),dict) else '')"
done; done
