import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import engine as E, synth
import gpu_dynamic_probe as P
os.environ["HSPF_REPAIR_PROF"]="1"
ctx = E.SpfContext(0)
g0 = synth.isis_100k()
for share in (0.001, 0.01):
    g = P.zero_links(g0, share, 11)
    for roots in ([0], (np.arange(64, dtype=np.uint64) * g.n // 64).astype(np.uint32)):
        ms, st = P.timed(ctx, g, np.asarray(roots, np.uint32), reps=2)
        print(share, len(roots), ms, file=sys.stderr)
