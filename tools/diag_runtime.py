"""Which HIP / HSA runtime libraries end up mapped into the process depending on the import order (torch first or librbl first):
    python tools/diag_runtime.py torch_first | rbl_first   (diagnostic of round 1's duplicate-runtime question)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def maps(tag):
    libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l or 'hsa-runtime' in l})
    print(tag, libs, flush=True)
order = sys.argv[1]
import numpy as np
if order == "torch_first":
    import torch
    print("torch avail", torch.cuda.is_available(), flush=True)
    maps("after torch init")
from rigid_body_light_amd import RigidBody, make_config
maps("after rbl import")
c = make_config(2, 12, False)
rb = RigidBody(c["cfg"], c["X"], c["Q"], c["a"], 1.0, 0.01)
try:
    r = rb.get_blob_positions(); print("rbl ok", r.shape, flush=True)
except Exception as e:
    print("rbl FAILED", e, flush=True)
maps("after rbl compute")
if order != "torch_first":
    import torch
    print("torch avail", torch.cuda.is_available(), flush=True)
    x = torch.ones(4, device="cuda"); print("torch ok", float(x.sum()), flush=True)
    maps("after torch init")
