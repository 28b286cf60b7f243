# A/B of two builds on one GPU box: SHOULDER_LIB=<alt .so> against the tree's library, interleaved 100-step regions of the headline leg
ALT=${1:-shoulder_amd/lib/alt_prev.so}
for k in 1 2 3 4; do for v in alt tree; do
if [ $v = alt ]; then export SHOULDER_LIB=$PWD/$ALT; else unset SHOULDER_LIB; fi
python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
done; done
