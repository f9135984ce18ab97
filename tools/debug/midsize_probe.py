"""ospf-10k x 64 roots (frac 0.021, launch-bound) under the lean sweep's plan switches: one context per setting."""
import os, sys, json, numpy as np
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tools"))
from holo_amd import engine as E, synth
import gpu_dynamic_probe as P
g = synth.ospf_10k()
roots = (np.arange(64, dtype=np.uint64) * g.n // 64).astype(np.uint32)
for env in ({}, {"HSPF_DENSE_MIN_WGS": "256", "HSPF_DENSE_STAY_PCT": "0"}, {"HSPF_DENSE_MIN_WGS": "256", "HSPF_DENSE_STAY_PCT": "0", "HSPF_DENSE_PASSES": "32"},
            {"HSPF_DENSE_MIN_WGS": "256", "HSPF_DENSE_STAY_PCT": "0", "HSPF_DENSE_PASSES": "8"}, {"HSPF_DENSE_MIN_WGS": "256", "HSPF_DENSE_STAY_PCT": "0", "HSPF_LEAN_HEAD": "2", "HSPF_DENSE_PCT": "5"}):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    ctx = E.SpfContext(0)
    for k, v in old.items():
        if v is None: del os.environ[k]
        else: os.environ[k] = v
    ms, st = P.timed(ctx, g, roots, reps=20)
    ms, st = P.timed(ctx, g, roots, reps=50)
    print(json.dumps({"env": env, "ms_call": round(ms, 4), "ms_device": round(st["ms_total"], 4), "launches": st["n_relax_launches"], "dbg1": hex(st["dbg"][1])}), flush=True)
    ctx.close()
