#!/bin/bash
# kernel trace of a short bench run -> per-launch durations of the engine kernels (gpurun_out/trace)
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out/trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o spf -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/run.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "hspf" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step: find the last k_init
idx = [i for i, r in enumerate(rows) if "k_init_fused" in r["Kernel_Name"]]
last = rows[idx[-2] - 1:idx[-1] + 1]            # one whole step: from the fill before the last-but-one root init to the next one
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print(f'{r["Kernel_Name"].split("(")[0][-22:]:24s} start {(int(r["Start_Timestamp"])-t0)/1e3:9.1f} us  dur {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} us  vgpr {r.get("VGPR_Count","?")} sgpr {r.get("SGPR_Count","?")}')
PY
cp gpurun_out/trace/*kernel_stats.csv gpurun_out/trace/kernel_stats.csv 2>/dev/null
find gpurun_out/trace -name "*kernel_trace.csv" -size +4M -delete
