"""isis-100k with a share of zero-cost link entries, 64 roots, a few runs: full stats (for rocprofv3 --kernel-trace --stats)."""
import os, sys, json, numpy as np
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tools"))
from holo_amd import engine as E, synth
import gpu_dynamic_probe as P
share = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
ctx = E.SpfContext(0)
g = P.zero_links(synth.isis_100k(), share, 11)
roots = (np.arange(64, dtype=np.uint64) * g.n // 64).astype(np.uint32)
ms, st = P.timed(ctx, g, roots, reps=5)
print(json.dumps({"share": share, "ms": ms, "stats": st}))
