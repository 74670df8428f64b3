import os, sys
if os.environ.get("PROBE_TORCH"): import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from opmgpu import capi, decks, partition
from opmgpu.model import GpuBlackoilModel
from test_gpu_dist import _periodic_pair, _Dom
gridA, gridB, src, halo = _periodic_pair()
tab = decks.satfunc_standard_tables()
B = GpuBlackoilModel(gridB, tab, capi.default_params())
dom = _Dom()
for k, v in halo.items(): setattr(dom, k, v)
partition.attach_comm(B, dom, 0, 1, partition.make_unique_id())
print("comm ok")
