#!/bin/bash
# kernel trace of tools/gpu_config_timing.py (configs[3] / configs[4]) -> per-kernel totals and the launches of the last fat-tree run
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out/cfgtrace; rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o cfg -- python $R/tools/gpu_config_timing.py > $OUT/run.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/cfgtrace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "hspf" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last run = everything after the last k_init_roots / k_init_fused pair preceding the final emit: take the last 60 launches
idx = [i for i, r in enumerate(rows) if "k_init_fused" in r["Kernel_Name"]]
last = rows[idx[-1]:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print(f'{r["Kernel_Name"].split("(")[0][-34:]:36s} start {(int(r["Start_Timestamp"])-t0)/1e3:9.1f} us  dur {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} us')
PY
find gpurun_out/cfgtrace -name "*kernel_trace.csv" -size +4M -delete
