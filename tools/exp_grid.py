import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import synth, engine as E
import torch
ctx = E.SpfContext(0)
def run(g, name):
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    n = g.n
    roots = np.array([(i*n)//64 for i in range(64)], np.uint32)
    d = torch.empty((64, n), dtype=torch.int32, device='cuda'); h = torch.empty((64, n), dtype=torch.int16, device='cuda')
    f = torch.empty((64, n), dtype=torch.int16, device='cuda'); m = torch.empty((64, n, 1), dtype=torch.int64, device='cuda')
    for i in range(4):
        st = ctx.run_device(G, roots, 0, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(), mask_ptr=m.data_ptr(), mask_words=1)
    print(name, 'E', g.e, 'launches', st['n_relax_launches'], 'ms', round(st['ms_relax'],3), 'us/launch', round(st['ms_relax']/st['n_relax_launches']*1e3,1))
    G.free()
g = synth.isis_100k(); run(g, 'isis-100k (grid+chords)')
n = 100000
links = synth._grid8_links(250, 400)
g2 = synth._routers_only(n, links, synth.SEED, 1, 100, synth.MAX_PATH_METRIC_WIDE, 'grid-only', {})
run(g2, 'grid only')
# chords only: random graph same E
links3 = synth._add_chords(n, links[:0], 500000, synth.SEED)
g3 = synth._routers_only(n, links3, synth.SEED, 1, 100, synth.MAX_PATH_METRIC_WIDE, 'random-only', {})
run(g3, 'random only')
