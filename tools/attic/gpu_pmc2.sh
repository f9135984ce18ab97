#!/bin/bash
# PMC passes (counters only; every pass under its own `timeout`: a counter set the hardware cannot collect makes
# rocprofv3 abort and then hang in its finaliser) on a short bench run; aggregates per kernel (full sweeps only = top 50% by value)
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out/pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 5 45 rocprofv3 --pmc $line --output-format csv -d $OUT/p$i -o q -- python $R/bench.py --steps 3 --warmup 1 --min-timed-ms 0 --no-cpu-baseline > $OUT/p$i.log 2>&1
done <<'PASSES'
GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCC_TAG_STALL_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_CYCLE_sum
TCC_EA0_RDREQ_LEVEL_sum TCC_LATENCY_FIFO_FULL_sum TCC_SRC_FIFO_FULL_sum
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES
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
