#!/bin/bash
# PMC passes (counters only) on a short bench run; aggregates per kernel (full sweeps only = top 50% by value)
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out/pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line --output-format csv -d $OUT/p$i -o q -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1
done <<'PASSES'
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
PASSES
cd $R
python - <<'PY'
import csv, glob, collections, json
out = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/pmc/p*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "hspf" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        for c, x in v.items():
            x = sorted(x); top = x[len(x)//2:]
            out[k][c] = {"launches": len(x), "mean_all": sum(x)/len(x), "mean_top_half": sum(top)/len(top), "max": x[-1]}
json.dump(out, open("gpurun_out/pmc/summary.json", "w"), indent=1)
for k, v in out.items():
    print("==", k)
    for c, d in v.items(): print(f"   {c:34s} n={d['launches']:4d} mean={d['mean_all']:14.1f} top-half={d['mean_top_half']:14.1f} max={d['max']:14.1f}")
PY
rm -rf $OUT/p*/
