import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import synth, engine as E
ctx = E.SpfContext(0)
g = synth.isis_100k()
G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
roots = np.asarray(g.meta['roots'], np.uint32)
for i in range(4):
    t = time.time(); res = ctx.run(G, roots, 0); dt = time.time() - t
    print(i, 'wall', round(dt*1e3,2), 'ms', res.stats)
