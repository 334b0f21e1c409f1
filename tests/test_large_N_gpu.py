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


def test_cfg5_full_size_dense_cholesky(orc):
    """BASELINE configs[4] at FULL size: 20 x shell_N_2562 = 51 240 blobs, n = 153 720, 2.36e10 matrix entries
    (> 2^31: 64-bit indexing everywhere), 189 GB.  The reference's M_half_W (c_rigid_obj.cpp:661-675): B Mob B
    assembled (k_build_M), factored in place (blocked MFMA Cholesky), L W.  Checks: sampled 3x3 blocks beyond flat
    index 2^31 bit-equal to the oracle's blocks; diag(L L^T) = diag(M); L L^T x = (B M B) x with the right-hand side
    from the matrix-free kernel; L W rows against row sums of the stored factor."""
    import time
    import torch
    from rigid_body_light_amd import make_config
    from rigid_body_light_amd._lib import DeviceContext
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    nb, nblb, wall = 20, 2562, False
    N = nb * nblb; n = 3 * N
    if free < 8 * n * n + (12 << 30):
        pytest.skip("needs %.0f GB of free HBM, %.0f GB free" % ((8 * n * n + (12 << 30)) / 1e9, free / 1e9))
    dev = torch.device("cuda:0")
    c = make_config(nb, nblb, wall)
    a, eta = c["a"], c["eta"]
    ctx = DeviceContext(a, eta, wall, cfg=c["cfg"], stream_ptr=torch.cuda.current_stream().cuda_stream)
    ctx.set_config(c["X"], c["Q"])
    r = torch.empty(n, dtype=torch.float64, device=dev)
    ctx.blob_positions(0, nb, r.data_ptr())
    rh = r.cpu().numpy().reshape(-1, 3)
    M = torch.empty(n * n, dtype=torch.float64, device=dev)
    assert M.numel() > 2 ** 31
    ctx.build_M(r.data_ptr(), N, True, M.data_ptr())            # B Mob B, :667-669
    ctx.sync_check()
    V = M.view(n, n)                                             # column-major storage: V[col][row]
    # -- (1) sampled blocks, all at flat offsets beyond 2^31 (column index > 2^31 / n ~ 13 970)
    damp = np.where(rh[:, 2] >= a, 1.0, rh[:, 2] / a)            # make_damp_mat :618-639
    nf = 1.0 / (8.0 * np.pi * eta * a)
    rng = np.random.default_rng(5)
    pairs = [(N - 1, N - 1), (0, N - 1), (N - 1, 4700), (N - 2, N - 1), (123, 40000)]
    pairs += [(int(rng.integers(0, N)), int(rng.integers(4700, N))) for _ in range(59)]
    for (i, j) in pairs:
        assert (3 * j) * n + 3 * i > 2 ** 31
        got = V[3 * j:3 * j + 3, 3 * i:3 * i + 3].cpu().numpy().T           # rows 3i.., columns 3j..
        blk = orc.pair_block(rh[i], rh[j], i, j, a, eta, wall) if i <= j else orc.pair_block(rh[j], rh[i], j, i, a, eta, wall).T
        assert np.array_equal(got, (damp[i] * blk) * damp[j]), (i, j)       # (di * v) * dj, the kernel's order
    d0 = M[:: n + 1].clone()
    x = torch.from_numpy(np.random.default_rng(4).standard_normal(n)).to(dev)
    W = torch.from_numpy(np.random.default_rng(3).standard_normal(n)).to(dev)
    Bd = torch.from_numpy(np.repeat(damp, 3)).to(dev)
    Mx = torch.empty_like(x)
    Bx = (Bd * x).contiguous()
    ctx.apply_M(Bx.data_ptr(), r.data_ptr(), N, 0, N, Mx.data_ptr())      # free-space M (no wall): damping applied here
    Mx = Bd * Mx
    # -- (2) in-place lower Cholesky, :670-671
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.cholesky(M.data_ptr(), n, zero_upper=True)
    ctx.sync_check()
    t_chol = time.perf_counter() - t0
    acc = torch.zeros(n, dtype=torch.float64, device=dev)
    CH = 2048
    for a0 in range(0, n, CH):
        acc += (V[a0:a0 + CH] ** 2).sum(0)                      # sum_j L_ij^2
    assert float(((acc - d0).abs() / d0.abs()).max()) < 1e-12
    # -- (3) L (L^T x) = (B M B) x ; V = L^T as a row-major matrix
    t = torch.empty_like(x)
    for a0 in range(0, n, CH):
        t[a0:a0 + CH] = V[a0:a0 + CH] @ x                       # (L^T x)_a = sum_b L[b][a] x[b]
    LLtx = torch.empty_like(x)
    ctx.trmv_lower(M.data_ptr(), n, t.data_ptr(), LLtx.data_ptr())       # our kernel: L t
    ctx.sync_check()
    assert float(torch.linalg.norm(LLtx - Mx) / torch.linalg.norm(Mx)) < 1e-10
    # -- (4) L W (:672) against row sums of the stored factor on sampled rows
    LW = torch.empty_like(W)
    ctx.trmv_lower(M.data_ptr(), n, W.data_ptr(), LW.data_ptr())
    ctx.sync_check()
    rows = torch.tensor([0, 1, n // 3, n // 2, n - 2, n - 1], device=dev)
    ref = (V[:, rows] * W[:, None]).sum(0)
    assert float(((LW[rows] - ref).abs() / ref.abs()).max()) < 1e-11
    print("cfg5 full size: Cholesky n=%d in %.2f s = %.1f TFLOP/s" % (n, t_chol, n ** 3 / 3.0 / t_chol / 1e12))
    ctx.close()
    del M, V, acc, t, LLtx, LW
    torch.cuda.empty_cache()


@pytest.mark.parametrize("wall", [False, True])
def test_more_bodies_than_a_grid_dimension(orc, wall):
    """70 000 four-blob bodies (280 000 blobs): the per-body kernels carry the body index in gridDim.y / .z (limit 65 535)
    and must go in rounds -- batched build + Cholesky + substitution with the wall term, the body-frame forms without.
    (L L^T)^-1 v of sampled bodies from both ends of the range against dense numpy blocks; the block preconditioner is
    finite everywhere and, in free space, the same with per-configuration factors."""
    import torch
    from rigid_body_light_amd._lib import DeviceContext, lib
    nb, nblb = 70000, 4
    a = 0.25
    cfg = 0.6 * np.array([[1.0, 1.0, 1.0], [1.0, -1.0, -1.0], [-1.0, 1.0, -1.0], [-1.0, -1.0, 1.0]])
    side = 42                                                   # 42^3 = 74 088 lattice sites, spacing 3
    idx = np.arange(nb)
    X = 3.0 * np.stack([idx % side, (idx // side) % side, idx // (side * side)], axis=1).astype(np.float64)
    X[:, 2] += 2.0
    rng = np.random.default_rng(9)
    Q = rng.standard_normal((nb, 4)); Q /= np.linalg.norm(Q, axis=1)[:, None]
    dev = torch.device("cuda:0")
    m = 3 * nblb
    ctx = DeviceContext(a, 1.0, wall, cfg=cfg, dt=0.01, stream_ptr=torch.cuda.current_stream().cuda_stream)
    ctx.set_config(X, Q)
    v = torch.from_numpy(rng.standard_normal(m * nb)).to(dev)
    o = torch.empty_like(v)
    ctx.block_solve(v.data_ptr(), o.data_ptr(), 0); ctx.sync_check()
    r = torch.empty(m * nb, dtype=torch.float64, device=dev)
    ctx.blob_positions(0, nb, r.data_ptr()); ctx.sync_check()
    rh, vh, oh = r.cpu().numpy(), v.cpu().numpy(), o.cpu().numpy()
    for b in (0, 1, 65534, 65535, 65536, nb - 1):
        sl = slice(m * b, m * (b + 1))
        Mb = orc.rotne_prager_tensor(rh[sl], a, 1.0, wall)
        ref = np.linalg.solve(Mb, vh[sl])
        assert np.linalg.norm(oh[sl] - ref) / np.linalg.norm(ref) < 1e-11, b
    lib().rbl_set_blk_pc(ctx.h, 1)
    z = torch.from_numpy(rng.standard_normal(m * nb + 6 * nb)).to(dev)
    p1 = torch.empty_like(z)
    ctx.apply_PC(z.data_ptr(), p1.data_ptr()); ctx.sync_check()
    assert bool(torch.isfinite(p1).all())
    if not wall:
        ctx.set_option("bodyframe_factor", 0)
        p2 = torch.empty_like(z)
        ctx.apply_PC(z.data_ptr(), p2.data_ptr()); ctx.sync_check()
        assert float(torch.linalg.norm(p1 - p2) / torch.linalg.norm(p2)) < 1e-10
    ctx.close()


def test_gmres_beyond_one_pass_of_the_arnoldi_kernels():
    """545 x shell_N_642 = 349 890 blobs: the saddle system has 1 052 940 unknowns, more than one pass of the Arnoldi
    update kernels covers (1 048 576 = 1024 blocks x 256 threads x 4 entries), so they walk their entries in two chunks.
    Six GMRES iterations (diagonal PC): the residual the solver reports (from its Hessenberg least squares) must be the
    true residual of the returned iterate -- that holds only if the basis is orthonormal and H = V^T A P^-1 V."""
    import torch
    from rigid_body_light_amd import make_config
    from rigid_body_light_amd._lib import DeviceContext
    nb, nblb, wall = 545, 642, False
    c = make_config(nb, nblb, wall)
    dev = torch.device("cuda:0")
    n3 = 3 * nb * nblb; nsys = n3 + 6 * nb
    assert nsys > 1024 * 256 * 4
    ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], dt=c["dt"], stream_ptr=torch.cuda.current_stream().cuda_stream)
    ctx.set_config(c["X"], c["Q"])
    rng = np.random.default_rng(12)
    b = torch.from_numpy(np.concatenate([rng.standard_normal(n3), np.tile([0.0, 0, -1.0, 0, 0, 0], nb)])).to(dev)
    x = torch.empty_like(b)
    m, res = ctx.gmres_saddle(b.data_ptr(), 6, 0.0, x.data_ptr())
    out = torch.empty_like(b)
    ctx.apply_saddle(x.data_ptr(), out.data_ptr()); ctx.sync_check()
    true_res = float(torch.linalg.norm(out - b) / torch.linalg.norm(b))
    assert m == 6 and 0.0 < res < 1.0
    assert abs(true_res - res) < 1e-6 * max(res, 1e-3), (res, true_res)
    ctx.close()

