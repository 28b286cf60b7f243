# A/B of the CU-masked lane streams (SHOULDER_CU_MASK), interleaved runs of the headline leg
for k in 1 2 3; do for y in 0 1; do
SHOULDER_CU_MASK=$y python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs 2>gpurun_out/mask_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mask $y', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('meshes_with_error_status'))"
done; done
