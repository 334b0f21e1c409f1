"""One-vector against two-vector symmetric product at cfg 3 (k_apply_M_sym<true,2,4> / k_apply_M_sym2<true,2,4>): wall time of each and
their ratio (the instruction counts of librbl.isa.json say 95 : 75 VALU per pair); the script the counter passes of
tools/run_profile_two_vector.sh run.  usage: bench_two_vector.py [bodies blobs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext
nb, nblb = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 642)
c = make_config(nb, nblb, True); N = nb * nblb
dev = torch.device("cuda:0"); st = torch.cuda.current_stream()
ctx = DeviceContext(c["a"], c["eta"], True, cfg=c["cfg"], stream_ptr=st.cuda_stream); ctx.set_config(c["X"], c["Q"])
r = torch.empty(3 * N, dtype=torch.float64, device=dev); ctx.blob_positions(0, nb, r.data_ptr())
F = torch.from_numpy(np.random.default_rng(2).standard_normal(2 * 3 * N)).to(dev)
U = torch.empty_like(F)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = {}
for nv in (1, 2, 1, 2):
    fn = (lambda: ctx.apply_M(F.data_ptr(), r.data_ptr(), N, 0, N, U.data_ptr())) if nv == 1 else (lambda: ctx.apply_M_multi(F.data_ptr(), r.data_ptr(), N, 2, U.data_ptr()))
    fn(); ctx.sync_check()
    e0.record(st)
    for _ in range(10):
        fn()
    e1.record(st); ctx.sync_check()
    res.setdefault(nv, []).append(e0.elapsed_time(e1) / 10)
print("%d x shell_N_%d wall: one vector %s ms, two vectors %s ms, ratio %.3f" % (nb, nblb, ["%.2f" % t for t in res[1]], ["%.2f" % t for t in res[2]], min(res[2]) / min(res[1])))
ctx.close()
