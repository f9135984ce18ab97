#!/bin/bash
# kernel trace of the asynchronous pipeline (isis-100k, 64-root runs in flight on the lanes of one context): a 1.5 ms
# window of the steady state, every launch with its queue — who overlaps whom
# usage: bash tools/gpu_trace_lanes.sh <tag> <lanes> <token 0|1>
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out/ln_$1; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/ln_child.py <<PY
import os, sys, numpy as np
os.environ["HSPF_ASYNC_LANES"] = "$2"; os.environ["HSPF_DENSE_STREAM"] = "$3"
import torch
sys.path.insert(0, "$R")
from holo_amd import synth, engine as E
ctx = E.SpfContext(0); dev = torch.device("cuda:0")
g = synth.isis_100k(); n = g.n
roots = ((np.arange(64, dtype=np.int64) * n) // 64).astype(np.uint32)
G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
L = int("$2"); nb = L + 1
bufs = [dict(dist_ptr=torch.empty((64, n), dtype=torch.int32, device=dev), hops_ptr=torch.empty((64, n), dtype=torch.int16, device=dev),
             flags_ptr=torch.empty((64, n), dtype=torch.int16, device=dev), mask_ptr=torch.empty((64, n, 1), dtype=torch.int64, device=dev)) for _ in range(nb)]
tk = []
for i in range(40):
    b = bufs[i % nb]
    tk.append(ctx.run_device_async(G, roots, 0, **{k: v.data_ptr() for k, v in b.items()}, mask_words=1))
    if len(tk) >= L:
        ctx.wait(tk.pop(0))
while tk:
    ctx.wait(tk.pop(0))
PY
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o ln -- python /tmp/ln_child.py > $OUT/run.log 2>&1
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
f = glob.glob(out + "/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "hspf" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tend = int(rows[-1]["End_Timestamp"])
win = [r for r in rows if tend - 2_600_000 <= int(r["Start_Timestamp"]) <= tend - 1_100_000]
t0 = int(win[0]["Start_Timestamp"])
qs = sorted({r["Queue_Id"] for r in win})
with open(out + "/timeline.txt", "w") as fo:
    for r in win:
        nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("hspf::", "")
        line = f'q{qs.index(r["Queue_Id"])} {nm:34s} start {(int(r["Start_Timestamp"])-t0)/1e3:8.1f} end {(int(r["End_Timestamp"])-t0)/1e3:8.1f} dur {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:7.1f}'
        print(line); fo.write(line + "\n")
PY
find $OUT -name "*kernel_trace.csv" -size +2M -delete
