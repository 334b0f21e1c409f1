"""Round 4: the host-pointer entry points a user of the reference's wrapper meets -- apply_saddle in one boundary crossing, the
reference's real usage model (an external SciPy Krylov solver over apply_saddle / apply_PC, src/Rigid.py:69-80), the library's own
solver for host vectors, and the C++-only members of the RFD family (c_rigid_obj.cpp:798-863, 880-893) against their numpy
restatements in oracle/oracle.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _body(nb, nblb, wall, block=False, dt=0.01):
    from rigid_body_light_amd import RigidBody, make_config
    c = make_config(nb, nblb, wall)
    return c, RigidBody(c["cfg"], c["X"], c["Q"], c["a"], c["eta"], dt, wall_PC=wall, block_PC=block)


@pytest.mark.parametrize("nb,nblb,wall", [(10, 12, False), (10, 12, True), (5, 162, True)])
def test_apply_saddle_one_crossing_equals_the_reference_composition(orc, nb, nblb, wall):
    """rbl_apply_saddle == the four-call composition of src/Rigid.py:73-80 == the oracle's dense saddle product"""
    from oracle import oracle as O
    c, rb = _body(nb, nblb, wall)
    n3 = 3 * nb * nblb
    x = np.random.default_rng(4).standard_normal(n3 + 6 * nb)
    got = rb.apply_saddle(x)
    lam, U = x[:n3], x[n3:]
    comp = np.concatenate((rb.apply_M(lam, rb.get_blob_positions()) - rb.K_dot(U).reshape(-1), rb.KT_dot(lam).reshape(-1)))
    assert np.linalg.norm(got - comp) <= 1e-13 * np.linalg.norm(comp)
    cfg = c["cfg"] - c["cfg"].mean(axis=0)
    r = orc.multi_body_pos(c["X"], c["Q"], cfg)
    K = O.K_matrix(c["X"], c["Q"], cfg)
    ref = np.concatenate((orc.apply_M(lam, r, c["a"], c["eta"], wall, mode="dense") - K @ U, K.T @ lam))
    assert np.linalg.norm(got - ref) <= 1e-12 * np.linalg.norm(ref)
    with pytest.raises(RuntimeError):
        rb.apply_saddle(x[:-1])


@pytest.mark.parametrize("nb,nblb,wall,block", [(10, 12, False, False), (8, 162, True, True)])
def test_scipy_gmres_over_the_dropin_operators_equals_the_native_solver(nb, nblb, wall, block):
    """the reference's usage model: scipy.sparse.linalg.gmres over RigidBody.apply_saddle with apply_PC as (right)
    preconditioner; its solution and the library's own device-resident GMRES (solve_saddle) agree to the tolerance, and
    both satisfy the system"""
    import scipy.sparse.linalg as spla
    c, rb = _body(nb, nblb, wall, block)
    n3, nsys = 3 * nb * nblb, 3 * nb * nblb + 6 * nb
    rhs = np.concatenate([0.1 * np.random.default_rng(9).standard_normal(n3), -np.tile([0.0, 0.0, -1.0, 0.3, 0.0, 0.0], nb)])
    A = spla.LinearOperator((nsys, nsys), matvec=lambda y: rb.apply_saddle(rb.apply_PC(y)), dtype=np.float64)
    y, info = spla.gmres(A, rhs, rtol=1e-10, atol=0.0, restart=200, maxiter=1)
    assert info == 0
    xs = rb.apply_PC(y)
    xn, its, res = rb.solve_saddle(rhs, max_iter=200, rtol=1e-10)
    bn = np.linalg.norm(rhs)
    assert 0 < its < 200 and res < 1e-10
    assert np.linalg.norm(rb.apply_saddle(xs) - rhs) < 1e-9 * bn and np.linalg.norm(rb.apply_saddle(xn) - rhs) < 1e-9 * bn
    assert np.linalg.norm(xs - xn) < 1e-7 * np.linalg.norm(xn)
    # warm start from the solution: nothing left to do
    xw, itw, resw = rb.solve_saddle(rhs, max_iter=200, rtol=1e-8, x0=xn)
    assert itw <= 1 and np.linalg.norm(xw - xn) < 1e-7 * np.linalg.norm(xn)


@pytest.mark.parametrize("wall", [False, True])
def test_rfd_family_vs_numpy_restatements(orc, wall):
    """M_RFD_cfgs (:798), M_RFD_from_U (:820), KT_RFD_from_U (:844), evolve_X_Q_RFD (:880) -- C++-only in the reference"""
    from oracle import oracle as O
    nb, nblb = 6, 12
    c, rb = _body(nb, nblb, wall, block=True)
    cfg = c["cfg"] - c["cfg"].mean(axis=0)
    rng = np.random.default_rng(21)
    U = rng.standard_normal(6 * nb)
    W = rng.standard_normal(3 * nb * nblb)
    rp, rm = rb.M_RFD_cfgs(U, 1e-2)
    op, om = O.M_RFD_cfgs(orc, U, c["X"], c["Q"], cfg, 1e-2)
    assert np.abs(rp - op).max() < 1e-13 and np.abs(rm - om).max() < 1e-13 and np.abs(rp - rm).max() > 1e-4
    got = rb.M_RFD_from_U(U, W)                                   # delta = 1e-3 as the reference hard-codes (:822)
    ref = O.M_RFD_from_U(orc, U, W, c["X"], c["Q"], cfg, c["a"], c["eta"], wall)
    assert np.linalg.norm(got - ref) < 1e-8 * np.linalg.norm(ref)  # a difference quotient: 1e-16 / 1e-3 of the products
    got = rb.KT_RFD_from_U(U, W)
    ref = O.KT_RFD_from_U(U, W, c["X"], c["Q"], cfg)
    assert np.linalg.norm(got - ref) < 1e-9 * np.linalg.norm(ref)
    X0, Q0 = rb.get_config()
    assert np.array_equal(X0.reshape(-1), np.asarray(c["X"]).reshape(-1))       # none of the above commits anything
    # evolve_X_Q_RFD: commits q displaced by U (no dt), K follows, the preconditioner is kept
    x = rng.standard_normal(3 * nb * nblb + 6 * nb)
    pc_before = rb.apply_PC(x)
    d = 1e-4 * U
    rb.evolve_rigid_bodies_RFD(d)
    Xn, Qn = rb.get_config()
    Xo, Qo = O.update_X_Q(c["X"], c["Q"], d)
    assert np.abs(Xn.reshape(-1, 3) - Xo).max() < 1e-14 and np.abs(Qn.reshape(-1, 4) - Qo).max() < 1e-14
    assert np.abs(rb.get_blob_positions().reshape(-1) - orc.multi_body_pos(Xo, Qo, cfg)).max() < 1e-13
    pc_after = rb.apply_PC(x)
    rel = np.linalg.norm(pc_after - pc_before) / np.linalg.norm(pc_before)
    assert rel < 1e-2                                             # the kept factors of q serve q + delta U ...
    rb.set_config(Xn, Qn)                                         # ... while a fresh build at the same configuration differs from
    fresh = rb.apply_PC(x)                                        #     the kept one only by O(delta)
    assert np.linalg.norm(fresh - pc_after) / np.linalg.norm(fresh) < 1e-2
    with pytest.raises(RuntimeError):
        rb.M_RFD_from_U(U[:-1], W)


@pytest.mark.parametrize("wall", [True, False])
def test_overlapped_convergence_test_changes_nothing_but_the_schedule(wall):
    """RBL_OPT_GMRES_OVERLAP_CHECK (large systems: the host reads iteration j's Hessenberg column while the GPU already applies
    iteration j + 1's preconditioner): same iterations, bitwise the same solution as the drain-then-continue schedule, and the
    solve ends at the first iteration that passes (36 x shell_N_642 = 23 112 blobs, block PC, 1e-9)."""
    import torch
    from rigid_body_light_amd import make_config
    from rigid_body_light_amd._lib import DeviceContext, lib
    nb, nblb = 36, 642
    c = make_config(nb, nblb, wall)
    dev = torch.device("cuda:0")
    nsys = 3 * nb * nblb + 6 * nb
    rhs = torch.zeros(nsys, dtype=torch.float64, device=dev)
    rhs[3 * nb * nblb:] = torch.from_numpy(-np.tile([0.0, 0.1, -1.0, 0.0, 0.0, 0.3], nb)).to(dev)
    out = {}
    for overlap in (0, 1):
        ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], dt=c["dt"], stream_ptr=torch.cuda.current_stream().cuda_stream)
        lib().rbl_set_blk_pc(ctx.h, 1)
        ctx.set_config(c["X"], c["Q"])
        ctx.set_option("gmres_overlap_check", overlap)
        assert ctx.get_option("gmres_overlap_check") == overlap
        x = torch.empty_like(rhs)
        m, res = ctx.gmres_saddle(rhs.data_ptr(), 100, 1e-9, x.data_ptr())
        ctx.sync_check()
        chk = torch.empty_like(rhs)
        ctx.apply_saddle(x.data_ptr(), chk.data_ptr()); ctx.sync_check()
        out[overlap] = (m, res, x.clone(), float(torch.linalg.norm(chk - rhs) / torch.linalg.norm(rhs)))
        ctx.close()
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1] and 3 < out[1][0] < 60
    assert torch.equal(out[0][2], out[1][2])
    assert out[1][3] < 2e-9


@pytest.mark.parametrize("nb,nblb", [(50, 162), (7, 162), (23, 42)])
def test_shared_matrix_product_on_the_matrix_cores_equals_the_batched_form(nb, nblb):
    """Free space: ONE body-frame inverse / preconditioner table serves every body, so a sweep over all bodies is a matrix-matrix
    product (k_shared_gemm on the fp64 MFMA, RBL_OPT_SHARED_GEMM = 1, default) instead of a matrix-vector product per body that
    re-reads the table (= 0).  Every user of it -- the three factor operations, the block preconditioner, the preconditioned
    Lanczos root (two vectors in lock step), a GMRES solve -- must give the same answer either way (column groups of 16 that end
    inside a body, a last row tile of 6 rows at 162 blobs, 42-blob bodies)."""
    import torch
    from rigid_body_light_amd import make_config
    from rigid_body_light_amd._lib import DeviceContext, lib
    c = make_config(nb, nblb, False)
    N = nb * nblb; n3 = 3 * N; nsys = n3 + 6 * nb
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    v = torch.from_numpy(rng.standard_normal(n3)).to(dev)
    xs = torch.from_numpy(rng.standard_normal(nsys)).to(dev)
    W = torch.from_numpy(rng.standard_normal(n3)).to(dev)
    res = {}
    for gemm in (0, 1):
        ctx = DeviceContext(c["a"], c["eta"], False, cfg=c["cfg"], dt=c["dt"], stream_ptr=torch.cuda.current_stream().cuda_stream)
        lib().rbl_set_blk_pc(ctx.h, 1)
        ctx.set_config(c["X"], c["Q"])
        ctx.set_option("shared_gemm", gemm)
        out = []
        for mode in (0, 1, 2):
            o = torch.empty_like(v)
            ctx.block_solve(v.data_ptr(), o.data_ptr(), mode); ctx.sync_check()
            out.append(o)
        o = torch.empty_like(xs)
        ctx.apply_PC(xs.data_ptr(), o.data_ptr()); ctx.sync_check()
        out.append(o)
        r = torch.empty(n3, dtype=torch.float64, device=dev)
        ctx.blob_positions(0, nb, r.data_ptr())
        ctx.set_lanczos(100, 1e-10)
        o = torch.empty_like(W)
        ctx.M_half_W(r.data_ptr(), N, W.data_ptr(), "lanczos_pc", o.data_ptr()); ctx.sync_check()
        out.append(o)
        sol = torch.empty_like(xs)
        m, rr = ctx.gmres_saddle(xs.data_ptr(), 100, 1e-10, sol.data_ptr()); ctx.sync_check()
        out.append(sol)
        res[gemm] = (out, m)
        ctx.close()
    names = ("(G G^T)^-1 v", "G^-1 v", "G^-T v", "apply_PC", "M_half_W lanczos_pc", "gmres solution")
    for name, a, b in zip(names, res[0][0], res[1][0]):
        err = float(torch.linalg.norm(a - b) / torch.linalg.norm(a))
        assert err < (1e-9 if name in ("M_half_W lanczos_pc", "gmres solution") else 1e-12), (name, err)
    assert abs(res[0][1] - res[1][1]) <= 1


@pytest.mark.parametrize("nb,nblb,shared_gemm,wall,block", [(50, 162, 1, False, True), (9, 42, 1, False, True), (70, 42, 0, False, True),
                                                             (27, 162, 1, True, True), (30, 42, 1, False, False), (30, 42, 1, True, False)])
def test_fused_krylov_iteration_equals_one_kernel_per_operation(nb, nblb, shared_gemm, wall, block):
    """RBL_OPT_FUSED_KRYLOV on launch-bound free-space systems with the block preconditioner: the product's slab reduction writes the
    saddle tail and the partial sums of the first Gram-Schmidt pass, and the normalisation V_{j+1} = w / |w| is folded into the next
    preconditioner application (RblNormFold: its kernels read w and the partial sums of |w|^2; body-frame tables in free space, the
    per-body factors' tail kernel above a wall, the diagonal preconditioner) -- against one kernel per operation:
    same iteration count, solutions equal to 1e-11, the reported residual is the true one (to tolerance: converged solve; and for a
    FIXED number of iterations, where the last iteration normalises the old way and a test-less solve reads H from the device)."""
    import torch
    from rigid_body_light_amd import make_config
    from rigid_body_light_amd._lib import DeviceContext, lib
    c = make_config(nb, nblb, wall)
    dev = torch.device("cuda:0")
    n3 = 3 * nb * nblb; nsys = n3 + 6 * nb
    rng = np.random.default_rng(nb)
    rhs = torch.from_numpy(np.concatenate([0.1 * rng.standard_normal(n3), -np.tile([0.0, 0.1, -1.0, 0.0, 0.0, 0.3], nb)])).to(dev)
    out = {}
    for fused in (0, 1):
        ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], dt=c["dt"], stream_ptr=torch.cuda.current_stream().cuda_stream)
        lib().rbl_set_blk_pc(ctx.h, 1 if block else 0)
        ctx.set_config(c["X"], c["Q"])
        ctx.set_option("fused_krylov", fused); ctx.set_option("shared_gemm", shared_gemm)
        ctx.set_option("gmres_one_kernel", 0)                               # (small systems: the general solver is what is under test)
        res = []
        for max_iter, rtol in ((200, 1e-9), (7, 0.0), (200, 1e-9)):       # (the third solve: convergence tests placed by the first one's count)
            x = torch.empty_like(rhs)
            m, r = ctx.gmres_saddle(rhs.data_ptr(), max_iter, rtol, x.data_ptr())
            ctx.sync_check()
            chk = torch.empty_like(rhs)
            ctx.apply_saddle(x.data_ptr(), chk.data_ptr()); ctx.sync_check()
            res.append((m, r, x.clone(), float(torch.linalg.norm(chk - rhs) / torch.linalg.norm(rhs))))
        out[fused] = res
        ctx.close()
    for a, b in zip(out[0], out[1]):
        assert a[0] == b[0], (a[0], b[0])
        assert float(torch.linalg.norm(a[2] - b[2]) / torch.linalg.norm(a[2])) < 1e-11
        assert abs(a[1] - b[1]) < 1e-9 * max(a[1], 1e-12) + 1e-15
    assert 3 < out[1][0][0] < 200 and out[1][0][3] < 2e-9 and out[1][2][3] < 2e-9
    assert out[1][1][0] == 7 and abs(out[1][1][3] - out[1][1][1]) < 1e-6 * out[1][1][1]      # fixed work: estimate == true residual


@pytest.mark.parametrize("nb,nblb,wall,block", [(10, 12, False, False), (6, 162, True, True), (6, 162, False, True)])
def test_lock_step_gmres_for_many_right_hand_sides(orc, nb, nblb, wall, block):
    """`solve_saddle_multi` (rbl_gmres_saddle_multi_dev): k right-hand sides of one configuration advance in lock step -- ONE
    multi-vector mobility product per iteration on the fp64 matrix cores, shared passes over the per-body factors -- and every
    column must be the solve `solve_saddle` gives it alone: same iteration count (+-1), same solution to 1e-10, its own residual
    below the tolerance and the TRUE residual through apply_saddle too.  19 columns = two passes (16 + 3) of the product; columns
    of very different scale converge at different iterations.  Then the customer: the body mobility matrix N = (K^T M^-1 K)^-1
    from 6 N_bod unit loads against the oracle's dense M and K (the operator is the reference's
    src/Rigid.py:69-80)."""
    from oracle import oracle as O
    c, rb = _body(nb, nblb, wall, block)
    n3, nsys = 3 * nb * nblb, 3 * nb * nblb + 6 * nb
    rng = np.random.default_rng(21)
    k = 19
    rhs = rng.standard_normal((k, nsys))
    rhs[:, :n3] *= 0.1
    rhs[3] *= 1e-6; rhs[7] *= 1e4                      # scale must not matter: the tolerance is relative, column by column
    rhs[11, :n3] = 0.0                                 # a pure body load, as in a time step
    x, its, res = rb.solve_saddle_multi(rhs, max_iter=200, rtol=1e-10)
    assert x.shape == (k, nsys) and its.shape == (k,) and res.shape == (k,)
    for col in range(k):
        xs, it1, res1 = rb.solve_saddle(rhs[col], max_iter=200, rtol=1e-10)
        bn = np.linalg.norm(rhs[col])
        assert abs(int(its[col]) - it1) <= 1 and res[col] < 1e-10, (col, its[col], it1, res[col])
        assert np.linalg.norm(x[col] - xs) <= 1e-9 * np.linalg.norm(xs), col          # (both are 1e-10 solves of the same system)
        assert np.linalg.norm(rb.apply_saddle(x[col]) - rhs[col]) < 1e-9 * bn, col
    # fixed work (rtol <= 0): exactly max_iter iterations for every column, no host test inside
    xf, itf, _ = rb.solve_saddle_multi(rhs[:5], max_iter=12, rtol=0.0)
    assert list(itf) == [12] * 5
    for col in range(5):
        x1, _, _ = rb.solve_saddle(rhs[col], max_iter=12, rtol=0.0)
        assert np.linalg.norm(xf[col] - x1) <= 1e-9 * np.linalg.norm(x1)
    with pytest.raises(RuntimeError):
        rb.solve_saddle_multi(rhs[:, :-1])
    # the body mobility matrix against dense numpy on the oracle's matrices
    Nmat, itn = rb.body_mobility_matrix(rtol=1e-11)
    cfg = c["cfg"] - c["cfg"].mean(axis=0)
    r = orc.multi_body_pos(c["X"], c["Q"], cfg)
    K = O.K_matrix(c["X"], c["Q"], cfg)
    M = orc.rotne_prager_tensor(r, c["a"], c["eta"], wall)
    if wall:                                            # apply_M's wall form is B M B (one flag for the wall term AND the damping, :641-659)
        B = orc.damp(r, c["a"])
        M = B[:, None] * M * B[None, :]
    Nref = np.linalg.inv(K.T @ np.linalg.solve(M, K))
    assert Nmat.shape == (6 * nb, 6 * nb) and np.all(itn > 0)
    assert np.linalg.norm(Nmat - Nref) <= 1e-8 * np.linalg.norm(Nref)
    assert np.linalg.norm(Nmat - Nmat.T) <= 1e-8 * np.linalg.norm(Nmat)            # symmetric positive definite, as a mobility must be
