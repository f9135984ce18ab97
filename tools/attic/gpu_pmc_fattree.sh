#!/bin/bash
# PMC passes over tools/gpu_fattree.py: per-kernel means of k_fw / k_emit (top half of the launches = the full sweeps).
# usage: bash tools/gpu_pmc_fattree.sh <tag>
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out/pmcft_$1; rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 5 90 rocprofv3 --pmc $line --output-format csv -d $OUT/p$i -o q -- python $R/tools/gpu_fattree.py 3 > $OUT/p$i.log 2>&1
done <<'PASSES'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum
GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum
SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LEVEL_WAVES
PASSES
cd $R
python - "$OUT" <<'PY'
import csv, glob, collections, json, sys
o = sys.argv[1]
out = collections.defaultdict(dict)
for f in sorted(glob.glob(o + "/p*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_fw" in k or "k_emit<" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        for c, x in v.items():
            x = sorted(x); top = x[len(x)//2:]
            out[k][c] = {"launches": len(x), "mean_all": sum(x)/len(x), "mean_top_half": sum(top)/len(top), "max": x[-1]}
json.dump(out, open(o + "/summary.json", "w"), indent=1)
for k, v in out.items():
    print("==", k)
    for c, d in v.items(): print(f"   {c:34s} n={d['launches']:4d} mean={d['mean_all']:14.1f} top-half={d['mean_top_half']:14.1f} max={d['max']:14.1f}")
PY
rm -rf $OUT/p*/
