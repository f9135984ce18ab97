#!/bin/bash
# kernel trace of the headline run (isis-100k, 64 roots, one context, 12 runs): per-launch durations of the last run
# usage: bash tools/gpu_trace_headline.sh <tag>     (HSPF_VARIANT / HSPF_DENSE_PCT are passed through)
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out/hl_$1; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/hl_child.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R")
from holo_amd import synth, engine as E
ctx = E.SpfContext(0); dev = torch.device("cuda:0")
g = synth.isis_100k(); n = g.n
roots = ((np.arange(64, dtype=np.int64) * n) // 64).astype(np.uint32)
G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
d = torch.empty((64, n), dtype=torch.int32, device=dev); h = torch.empty((64, n), dtype=torch.int16, device=dev)
f = torch.empty((64, n), dtype=torch.int16, device=dev); m = torch.empty((64, n, 1), dtype=torch.int64, device=dev)
for i in range(12):
    st = ctx.run_device(G, roots, 0, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(), mask_ptr=m.data_ptr(), mask_words=1)
print(st["ms_total"], st["n_relax_launches"])
PY
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o hl -- python /tmp/hl_child.py > $OUT/run.log 2>&1
cd $R
tail -1 $OUT/run.log
python - "$OUT" <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
f = glob.glob(out + "/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "hspf" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_init_fused" in r["Kernel_Name"]]
last = rows[idx[-1]:]
t0 = int(last[0]["Start_Timestamp"])
with open(out + "/one_run.txt", "w") as fo:
    for r in last:
        nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("hspf::", "")
        line = f'{nm:40s} start {(int(r["Start_Timestamp"])-t0)/1e3:9.1f} us  dur {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} us'
        print(line); fo.write(line + "\n")
PY
find $OUT -name "*kernel_trace.csv" -size +2M -delete
