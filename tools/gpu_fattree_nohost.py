import json, os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from holo_amd import synth, engine as E
g = synth.isis_fattree(100)
host0 = 12500
rp = g.row_ptr.astype(np.int64)
src = np.repeat(np.arange(g.n), np.diff(rp))
keep = (src < host0) & (g.col < host0)
col = g.col[keep]; met = g.metric[keep]
cnt = np.bincount(src[keep], minlength=host0)[:host0]
nrp = np.zeros(host0 + 1, np.uint32); nrp[1:] = np.cumsum(cnt)
vf = g.vflags[:host0].copy()
ctx = E.SpfContext(0); dev = torch.device("cuda:0")
roots = np.array([r if r < host0 else 7500 + 50 + i for i, r in enumerate(g.meta["roots"])], np.uint32)
G = ctx.upload(nrp, col, met, vf, g.max_path_metric)
W = G.mask_words(roots); R = len(roots); n = host0
d = torch.empty((R, n), dtype=torch.int32, device=dev); h = torch.empty((R, n), dtype=torch.int16, device=dev)
f = torch.empty((R, n), dtype=torch.int16, device=dev); m = torch.empty((R, n, W), dtype=torch.int64, device=dev)
for _ in range(5):
    st = ctx.run_device(G, roots, 0, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(), mask_ptr=m.data_ptr(), mask_words=W)
print(json.dumps({k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in st.items()}))
