#!/bin/bash
# kernel trace of tools/gpu_schedule_variants.py (one variant, spread roots) -> per-launch durations of one 64-root run
# usage: bash tools/gpu_trace_variants.sh <HSPF_VARIANT value> <tag>
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out/trace_$2; rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o spf -- python $R/tools/gpu_schedule_variants.py --variants $1 --reps 6 --no-oracle --only spread64 > $OUT/run.log 2>&1
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "hspf" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_init_fused" in r["Kernel_Name"]]
last = rows[idx[-3]:idx[-2]]
t0 = int(last[0]["Start_Timestamp"])
with open(out + "/one_run.txt", "w") as fo:
    for r in last:
        line = f'{r["Kernel_Name"].split("(")[0][-26:]:28s} start {(int(r["Start_Timestamp"])-t0)/1e3:9.1f} us  dur {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} us  vgpr {r.get("VGPR_Count","?")} sgpr {r.get("SGPR_Count","?")} lds {r.get("LDS_Block_Size","?")}'
        print(line); fo.write(line + "\n")
PY
cp $OUT/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
find $OUT -name "*kernel_trace.csv" -size +4M -delete
