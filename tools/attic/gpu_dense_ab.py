"""A/B of the lean sweep's learned mode schedule (dense sweeps without activation stamps) on the headline workload:
device ms per 64-root run with the schedule (default) and without (HSPF_VARIANT bit19), per HSPF_DENSE_PCT.  Run on the GPU
box; results of the last run of every setting are compared with each other bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import json, os, sys, numpy as np, torch, hashlib
sys.path.insert(0, %r)
from holo_amd import synth, engine as E
ctx = E.SpfContext(0); dev = torch.device("cuda:0")
g = synth.isis_100k(); n = g.n
roots = ((np.arange(64, dtype=np.int64) * n) // 64).astype(np.uint32)
G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
d = torch.empty((64, n), dtype=torch.int32, device=dev); h = torch.empty((64, n), dtype=torch.int16, device=dev)
f = torch.empty((64, n), dtype=torch.int16, device=dev); m = torch.empty((64, n, 1), dtype=torch.int64, device=dev)
ms, launches = [], []
for i in range(40):
    st = ctx.run_device(G, roots, 0, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(), mask_ptr=m.data_ptr(), mask_words=1)
    ms.append(st["ms_total"]); launches.append(st["n_relax_launches"])
sig = hashlib.sha1(d.cpu().numpy().tobytes() + h.cpu().numpy().tobytes() + m.cpu().numpy().tobytes()).hexdigest()
print(json.dumps({"first3_ms": [round(x, 4) for x in ms[:3]], "median_ms": round(float(np.median(ms[5:])), 4), "launches": launches[-1], "lean": st["dbg"][0], "sig": sig}))
""" % ROOT

if __name__ == "__main__":
    sets = [("schedule off", {"HSPF_VARIANT": "524288"}), ("passes 1", {"HSPF_DENSE_PASSES": "1"}), ("default (passes 16)", {})]
    sets += [(f"passes {k}", {"HSPF_DENSE_PASSES": str(k)}) for k in (2, 4, 8)]
    if len(sys.argv) > 1 and sys.argv[1] == "pct":
        sets = [("pct 90", {}), ("pct 70", {"HSPF_DENSE_PCT": "70"}), ("pct 50", {"HSPF_DENSE_PCT": "50"}), ("pct 30", {"HSPF_DENSE_PCT": "30"}), ("pct 15", {"HSPF_DENSE_PCT": "15"})]
    for name, env in sets:
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(name, line[-1] if line else out.stderr[-400:], flush=True)
