"""apply_M time against the chunk-length override of the symmetric kernel (RBL_OPT_MATVEC_KERNEL = 2, RBL_OPT_SYM_CHUNK = chunk) at one system size.
usage: bench_chunk_sweep.py bodies blobs wall"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext
nb, nblb, wall = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3]))
dev = torch.device("cuda:0")
c = make_config(nb, nblb, wall)
ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], dt=c["dt"], stream_ptr=torch.cuda.current_stream().cuda_stream)
ctx.set_config(c["X"], c["Q"])
N = nb * nblb
r = torch.empty(3 * N, dtype=torch.float64, device=dev); ctx.blob_positions(0, nb, r.data_ptr())
x = torch.randn(3 * N, dtype=torch.float64, device=dev); o = torch.empty_like(x)
for chunk in (0, 1, 2, 3, 4, 6, 8, 12, 16):
    ctx.set_option("matvec_kernel", 2); ctx.set_option("sym_chunk", chunk)
    for _ in range(5): ctx.apply_M(x.data_ptr(), r.data_ptr(), N, 0, N, o.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): ctx.apply_M(x.data_ptr(), r.data_ptr(), N, 0, N, o.data_ptr())
    e1.record(); torch.cuda.synchronize()
    print("chunk %2d: %.1f us / apply_M  info %s" % (chunk, e0.elapsed_time(e1) * 10.0, ctx.apply_M_sym_info(N) if chunk == 0 else ""), flush=True)
