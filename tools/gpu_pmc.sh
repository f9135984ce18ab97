#!/bin/bash
# PMC diagnosis passes on tools/quick_gpu.py (counters only; no tracing flags).
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out/pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line --output-format csv -d $OUT/p$i -o q -- python $R/tools/quick_gpu.py > $OUT/p$i.log 2>&1
done <<'PASSES'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum
TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_sum TCC_CYCLE_sum
PASSES
cd $R
python - <<'PY'
import csv, glob, collections, os
out = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/pmc/p*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "hspf" not in k: continue
        for c, x in v.items():
            x = sorted(x)
            # drop the early-exit launches: take the mean of the top 60% values
            top = x[int(len(x)*0.4):]
            out[k][c] = (len(x), sum(top)/len(top))
for k, v in out.items():
    print("==", k)
    for c, (n, m) in v.items():
        print(f"   {c:40s} n={n:4d} mean_top={m:14.1f}")
PY
