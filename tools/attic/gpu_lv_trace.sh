#!/bin/bash
# kernel trace of one-root runs on the lane = vertex kernel -> per-launch start / duration (gpurun_out/lvtrace)
#   tools/gpu_lv_trace.sh [graph] [roots]
export TMPDIR=/tmp
G=${1:-isis-100k}; RR=${2:-1}
R=$(pwd); OUT=$R/gpurun_out/lvtrace; rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o lv -- python $R/tools/gpu_lv_threshold.py --graph=$G --roots=$RR > $OUT/run.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/lvtrace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "hspf" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_init_lv" in r["Kernel_Name"]]
last = rows[idx[-2]:idx[-1]] if len(idx) > 1 else rows[idx[-1]:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print(f'{r["Kernel_Name"].split("(")[0][-22:]:24s} start {(int(r["Start_Timestamp"])-t0)/1e3:9.1f} us  dur {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} us  vgpr {r.get("VGPR_Count","?")} sgpr {r.get("SGPR_Count","?")}')
PY
find gpurun_out/lvtrace -name "*kernel_trace.csv" -size +4M -delete
