#!/bin/bash
# kernel trace of tools/gpu_fattree.py -> per-launch durations of the last run
# usage: bash tools/gpu_trace_fattree.sh <tag>     (HSPF_VARIANT is passed through)
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out/ft_$1; rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ft -- python $R/tools/gpu_fattree.py 5 > $OUT/run.log 2>&1
cd $R
grep device_ms $OUT/run.log | cut -c1-300
python - "$OUT" "${2:-1}" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "hspf" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_init_fw" in r["Kernel_Name"] or "k_init_fused" in r["Kernel_Name"] or "k_init_roots" in r["Kernel_Name"]]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 1
last = rows[max(idx[-n_last] - 2, 0):]
t0 = int(last[0]["Start_Timestamp"])
with open(out + "/one_run.txt", "w") as fo:
    for r in last:
        line = f'{r["Kernel_Name"].split("(")[0][-40:]:42s} start {(int(r["Start_Timestamp"])-t0)/1e3:9.1f} us  dur {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} us  grid {r.get("Grid_Size_X","?")}x{r.get("Grid_Size_Y","?")} vgpr {r.get("VGPR_Count","?")}'
        print(line); fo.write(line + "\n")
PY
cp $OUT/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
find $OUT -name "*kernel_trace.csv" -size +4M -delete
