"""Large-N robustness of apply_M on the GPU: sizes beyond the BASELINE configs (256 800 and 513 600 wall-corrected
blobs, 256 200 free-space ones), a few rows of each against the CPU oracle.  The largest one needs 56 GB of slab
workspace for the symmetric kernel (a quarter of the card is its budget)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nb,nblb,wall", [(400, 642, True), (800, 642, True), (100, 2562, False)])
def test_apply_M_large_N_rows_vs_oracle(orc, nb, nblb, wall):
    import torch
    from rigid_body_light_amd import make_config
    from rigid_body_light_amd._lib import DeviceContext
    dev = torch.device("cuda:0")
    c = make_config(nb, nblb, wall)
    N = nb * nblb
    ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], stream_ptr=torch.cuda.current_stream().cuda_stream)
    ctx.set_config(c["X"], c["Q"])
    r = torch.empty(3 * N, dtype=torch.float64, device=dev)
    ctx.blob_positions(0, nb, r.data_ptr())
    x = torch.from_numpy(np.random.default_rng(9).standard_normal(3 * N)).to(dev)
    out = torch.empty_like(x)
    ctx.apply_M(x.data_ptr(), r.data_ptr(), N, 0, N, out.data_ptr())
    ctx.sync_check()
    rh, xh, oh = r.cpu().numpy(), x.cpu().numpy(), out.cpu().numpy()
    for b in (0, N // 2 + 7, N - 5):
        Uo = orc.apply_M_rows(xh, rh, b, b + 4, c["a"], c["eta"], wall, nthreads=16)
        assert np.linalg.norm(oh[3 * b:3 * b + 12] - Uo) / np.linalg.norm(Uo) < 1e-11
    ctx.close()
    del r, x, out
    torch.cuda.empty_cache()
