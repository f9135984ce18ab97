"""bench.py's latency_1root block alone (one root per run on ospf-500 / ospf-10k / isis-100k), with HSPF_VARIANT passed
through: run on the GPU box."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                   # noqa: E402
from holo_amd import engine as E               # noqa: E402

if __name__ == "__main__":
    ctx = E.SpfContext(0)
    print(json.dumps(bench.latency_1root(ctx, torch.device("cuda:0"))))
