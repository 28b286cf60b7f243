cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SHOULDER_CU_MASK=1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tlm -o tl -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs > gpurun_out/tlm_bench.json 2>> gpurun_out/tl_err.log
python tools/lane_timeline.py gpurun_out/tlm/tl_kernel_trace.csv 6 > gpurun_out/tl_mask1.txt
python tools/unet_gaps.py gpurun_out/tlm/tl_kernel_trace.csv 6 >> gpurun_out/tl_mask1.txt
python - <<'P' >> gpurun_out/tl_mask1.txt
import csv
rows=list(csv.DictReader(open("gpurun_out/tlm/tl_kernel_trace.csv")))
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][-40:],r["Queue_Id"]) for r in rows)
t0=ev[len(ev)//2][0]
for e in ev[len(ev)//2:len(ev)//2+140]:
    print(f"{(e[0]-t0)/1e3:9.1f} {(e[1]-t0)/1e3:9.1f} q{e[3]} {e[2]}")
P
rm -f gpurun_out/tlm/*.csv
