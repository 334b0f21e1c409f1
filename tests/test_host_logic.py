"""CPU tests of the host-side mirror of the reference interface (no GPU compute):
the behaviour checks of reference tests/test_interface.py + tests/test_precision.py
that do not touch the mobility, plus numeric checks of K, K^T, K^-1, apply_PC (diagonal
PC) and evolve against the numpy restatement in oracle/oracle.py."""
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from conftest import create_solver, random_positions
from oracle import oracle as onp


def test_import_surface():
    from Rigid import RigidBody, c_rigid                       # reference tests/test_import.py
    assert c_rigid.CManyBodies.precision == "double"
    for name in ("getConfig", "setParameters", "setBlkPC", "setWallPC", "set_K_mats", "K_x_U",
                 "KT_x_Lam", "multi_body_pos", "apply_PC", "setConfig", "get_K", "get_Kinv",
                 "evolve_X_Q", "apply_M"):                      # c_rigid_obj.cpp:1002-1023
        assert hasattr(c_rigid.CManyBodies, name)
    for name in ("get_config", "set_config", "get_blob_positions", "KT_dot", "K_dot", "apply_PC",
                 "apply_saddle", "apply_M", "get_K", "get_Kinv", "evolve_rigid_bodies"):
        assert hasattr(RigidBody, name)


def test_create(shell12):                                      # tests/test_interface.py:8-23
    from rigid_body_light_amd import RigidBody
    rng = np.random.default_rng(0)
    X = rng.standard_normal((10, 3)); Q = rng.standard_normal((10, 4))
    RigidBody(shell12, X, Q, 1.0, 1.0, dt=0.01)
    RigidBody(shell12, X, Q, 1.0, 1.0, dt=0.01, wall_PC=True)
    RigidBody(shell12, X, Q, 1.0, 1.0, dt=0.01, block_PC=True)
    with pytest.raises(RuntimeError):
        RigidBody(shell12.flatten()[:-1], X, Q, 1.0, 1.0, dt=0.01)


def test_config_roundtrip_and_shapes():                        # :26-38
    rng = np.random.default_rng(1)
    X0 = rng.random((10, 3)); Q0 = rng.random((10, 4))
    cb = create_solver(X0, Q0)
    cb.set_config(X0, Q0)
    X, Q = cb.get_config()
    assert np.allclose(X, X0)
    assert np.allclose(Q, Rotation.from_quat(Q0).as_quat())    # normalised
    assert X.shape == (10, 3) and Q.shape == (10, 4)
    cb2 = create_solver(X0.flatten(), Q0.flatten())            # flat in -> flat out (src/Rigid.py:54)
    assert cb2.get_config()[0].shape == (30,)
    assert cb2.K_dot(np.ones(60)).shape == (360,)


def test_bad_config():                                         # :41-52
    rng = np.random.default_rng(2)
    X0 = rng.random((10, 3)); Q0 = rng.random((10, 4))
    cb = create_solver(X0, Q0)
    with pytest.raises(RuntimeError):
        cb.set_config(X0, Q0[:9])
    with pytest.raises(RuntimeError):
        cb.set_config(X0[:9], Q0)


def test_inputs_not_mutated(shell12):
    cfg = shell12 + 0.3                                        # non-zero mean: reference removes it in place
    cfg0 = cfg.copy()
    X, Q = random_positions(3, seed=3)
    cb = create_solver(X, Q, rigid_config=cfg)
    U = np.ones(18); U0 = U.copy()
    cb.evolve_rigid_bodies(U)
    assert np.array_equal(cfg, cfg0) and np.array_equal(U, U0)


@pytest.mark.parametrize("precision", [np.float32, np.float64])   # tests/test_precision.py:7-25
def test_K_ops_numeric(shell12, precision):
    n = 5
    X, Q = random_positions(n, seed=4)
    Xp = X.astype(precision); Qp = Q.astype(precision)
    cb = create_solver(Xp, Qp)
    cfg = onp.remove_mean(shell12)
    Qn = onp.normalize_quats(Qp)
    K = onp.K_matrix(Xp, Qn, cfg)
    U = np.random.default_rng(5).standard_normal(6 * n).astype(precision)
    lam = np.random.default_rng(6).standard_normal(3 * 12 * n).astype(precision)
    ku = cb.K_dot(U); ktl = cb.KT_dot(lam)
    assert ku.shape == (n * 12, 3) and ktl.shape == (2 * n, 3)     # :76-109
    np.testing.assert_allclose(ku.ravel(), K @ U.astype(np.float64), atol=1e-13)
    np.testing.assert_allclose(ktl.ravel(), K.T @ lam.astype(np.float64), atol=1e-12)
    with pytest.raises(RuntimeError):
        cb.K_dot(np.zeros(6 * n - 3))
    with pytest.raises(RuntimeError):
        cb.KT_dot(np.zeros(3 * 12 * n - 5))


def test_get_K_Kinv(shell12):                                  # :112-122 + numeric
    n = 3
    X, Q = random_positions(n, seed=7)
    cb = create_solver(X, Q)
    cfg = onp.remove_mean(shell12); Qn = onp.normalize_quats(Q)
    K = cb.get_K(); Ki = cb.get_Kinv()
    assert K.shape == (3 * 12 * n, 6 * n) and Ki.shape == (6 * n, 3 * 12 * n)
    np.testing.assert_allclose(K.toarray(), onp.K_matrix(X, Qn, cfg), atol=1e-14)
    np.testing.assert_allclose(Ki.toarray(), onp.Kinv_matrix(X, Qn, cfg), atol=1e-12)
    np.testing.assert_allclose((Ki @ K).toarray(), np.eye(6 * n), atol=1e-12)   # pseudo-inverse


@pytest.mark.parametrize("precision", [np.float32, np.float64])  # tests/test_precision.py:28-44
def test_apply_PC_diag_numeric(orc, shell12, precision):       # :125-147, (block_PC=False) cases
    n = 3
    for wall in (False, True):
        X, Q = random_positions(n, wall=wall, seed=8)
        X[:, 2] += 1.0 if wall else 0.0
        cb = create_solver(X, Q, wall_PC=wall)
        size = 3 * 12 * n + 6 * n
        b = np.random.default_rng(9).standard_normal(size).astype(precision)
        out = cb.apply_PC(b)
        assert out.shape == (size,) and np.linalg.norm(out) > 0
        ref = onp.apply_PC(orc, b.astype(np.float64), X, onp.normalize_quats(Q), onp.remove_mean(shell12), 1.0, 1.0, wall, False)
        np.testing.assert_allclose(out, ref, rtol=1e-11, atol=1e-11)
        with pytest.raises(RuntimeError):
            cb.apply_PC(np.zeros(size - 4))


def test_apply_PC_under_wall_raises():                         # tests/test_wall.py:33-36
    cb = create_solver(np.array([[0.0, 0.0, 0.0]]), np.array([[1.0, 0, 0, 0]]), wall_PC=True)
    with pytest.raises(RuntimeError, match="below the wall"):
        cb.apply_PC(np.ones(42))


def test_evolve_numeric(shell12):                              # :199-211 + numeric
    n = 3
    X, Q = random_positions(n, seed=10)
    cb = create_solver(X, Q, dt=0.37)
    U = np.random.default_rng(11).standard_normal(6 * n)
    cb.evolve_rigid_bodies(U)
    Xn, Qn = cb.get_config()
    assert np.linalg.norm(Xn - X) > 0 and np.linalg.norm(Qn - Q) > 0
    Xr, Qr = onp.evolve(X, onp.normalize_quats(Q), U, 0.37)
    np.testing.assert_allclose(Xn, Xr, atol=1e-14)
    np.testing.assert_allclose(Qn, Qr, atol=1e-14)
    # K follows the new configuration (set_K_mats inside evolve, :876)
    np.testing.assert_allclose(cb.get_K().toarray(), onp.K_matrix(Xr, Qr, onp.remove_mean(shell12)), atol=1e-13)
    with pytest.raises(RuntimeError):
        cb.evolve_rigid_bodies(np.zeros(6 * n - 1))


def test_update_X_Q_numeric_and_non_committing(shell12):       # c_rigid_obj.cpp:798-863 (C++ only in the reference)
    n = 4
    X, Q = random_positions(n, seed=14)
    cb = create_solver(X, Q, dt=0.2)
    U = np.random.default_rng(15).standard_normal(6 * n) * 0.3
    U0 = U.copy()
    Xu, Qu = cb.update_X_Q(U)
    Xr, Qr = onp.update_X_Q(X, onp.normalize_quats(Q), U)          # displacement units: no dt
    np.testing.assert_allclose(Xu.reshape(-1, 3), Xr, atol=1e-14)
    np.testing.assert_allclose(Qu.reshape(-1, 4), Qr, atol=1e-14)
    assert np.array_equal(U, U0)
    Xc, Qc = cb.get_config()
    np.testing.assert_allclose(Xc, X, atol=0)                     # nothing committed
    with pytest.raises(RuntimeError):
        cb.update_X_Q(np.zeros(6 * n + 1))


def test_size_errors_without_gpu():
    X, Q = random_positions(2, seed=12)
    cb = create_solver(X, Q)
    F = np.ones(72)
    with pytest.raises(RuntimeError):
        cb.apply_M(F[:-4], np.ones(72))                         # :158-163
    with pytest.raises(RuntimeError):
        cb.apply_M(F[:-1], np.ones(71))
    with pytest.raises(RuntimeError):
        cb.apply_saddle(np.ones(72 + 12 - 2))                   # :192-196


def test_dimer_is_singular_error_not_exit():
    # reference exit()s on singular K^T K (c_rigid_obj.cpp:313-316); we raise
    cfg = np.array([[0, 0, -0.5], [0, 0, 0.5]])
    from rigid_body_light_amd import RigidBody
    with pytest.raises(RuntimeError, match="singular"):
        RigidBody(cfg, np.zeros((1, 3)), np.array([[1.0, 0, 0, 0]]), 1.0, 1.0, 0.1)


def test_KTinv_RFD_numeric():
    """reference c_rigid_obj.cpp:743-767 (host-only arithmetic) vs the numpy restatement.  An
    irregular body: for the symmetric shells the quantity vanishes identically."""
    n = 3
    cfg = np.random.default_rng(5).uniform(-1, 1, (9, 3)) * np.array([2.0, 1.0, 0.5])
    X, Q = random_positions(n, seed=20)
    cb = create_solver(X, Q, rigid_config=cfg)
    W = np.random.default_rng(21).standard_normal(6 * n)
    for delta in (1e-2, 1e-4):
        out = cb.KTinv_RFD(W, delta=delta)
        ref = onp.KTinv_RFD(W, X, onp.normalize_quats(Q), onp.remove_mean(cfg), delta)
        assert out.shape == (6 * n,) and np.abs(ref).max() > 0.1
        np.testing.assert_allclose(out, ref, rtol=0, atol=1e-14 / delta * 50)   # difference quotient: rounding / delta
    with pytest.raises(RuntimeError):
        cb.KTinv_RFD(W[:-1])


def test_bench_refuses_more_ranks_than_gpus():
    """`bench.py --gpus N` must never run on fewer devices and print n_gpus N (or 1): with RCCL and fewer than N
    visible GPUs it exits non-zero with a message -- here, on the CPU-only container, for N = 2."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0
    assert "needs 2 visible GPUs" in (p.stdout + p.stderr)
    assert not any(l.startswith("{") for l in p.stdout.splitlines())


def test_bench_line_helpers():
    """the pieces of bench.py's line that are pure host logic: the top-level `timesteps_per_sec` / `cpu_baseline_timestep` blocks
    (BASELINE's metric is timesteps/sec + M.F GFLOP/s), the product count of a Brownian step, the --opt parser"""
    import bench
    b = {"timesteps_per_sec": 1.6, "gmres_iterations": [17, 17, 18], "lanczos_iterations_last_step": [5]}
    t = {"deterministic_fixed": {"timesteps_per_sec": 2.4, "gmres_residual_max": 2e-4},
         "converged": {"timesteps_per_sec": 15.0, "gmres_iterations": [2, 1, 1]},
         "brownian_converged": {"lanczos_0.001": b}}
    assert abs(bench.brownian_products(b) - (52.0 / 3 + 2 + 10)) < 1e-12
    h = bench.headline_timesteps(t)
    assert h["deterministic_fixed_work"] == 2.4 and h["deterministic_converged"] == 15.0 and h["brownian_converged"] == 1.6
    assert h["deterministic_fixed_work_residual"] == 2e-4 and "SURVEY" in h["definition"]
    cb = {"1core": {"seconds_per_step": 500.0, "cores": 1}, "allcores": {"seconds_per_step": 30.0, "cores": 16}}
    c = bench.cpu_timestep_baseline(t, cb)
    assert abs(c["1core"]["deterministic_fixed_work"] - 1.0 / (21 * 500.0)) < 1e-18
    assert abs(c["allcores"]["brownian_converged"] - 1.0 / (bench.brownian_products(b) * 30.0)) < 1e-18
    assert abs(c["allcores"]["deterministic_converged"] - 1.0 / ((4.0 / 3 + 1) * 30.0)) < 1e-15 and c["kind"] == "port"

    class Ctx:
        def __init__(self): self.seen = []
        def set_option(self, k, v): self.seen.append((k, v))

    class Args:
        opt = ["comm_split=1", " sym_work_queue = 0"]
    ctx = Ctx()
    bench.apply_opts(ctx, Args())
    assert ctx.seen == [("comm_split", 1), ("sym_work_queue", 0)]


def test_bench_slim_line_ends_with_the_whole_metric():
    """The driver records the parsed contract keys and the LAST 2 000 characters of stdout: the one line must be
    small, carry no prose, and END with a `summary` that holds both halves of BASELINE's metric -- time steps per second with the CPU
    port's figures beside them and the M.F rate -- plus every configuration's roofline fraction.  Replayed on round 4's full record."""
    import json
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    full = json.loads(open(os.path.join(root, "profiles", "r04_bench_line_final.json")).read().strip().splitlines()[-1])
    line = bench.slim_line(full)
    text = json.dumps(line)
    assert len(text) < 4500 and list(line)[-1] == "summary"
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert line[k] == full[k]
    assert set(line["roofline"]) == set(bench.ROOFLINE_KEYS) and 0.0 < line["roofline"]["frac"] <= 1.0
    assert set(line["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample", "seconds_per_step"}
    tail = text[-2000:]
    start = tail.index('"summary"')
    s = json.loads(tail[start + len('"summary": '):-1])       # the whole summary sits inside the recorded tail
    ts = s["timesteps_per_sec"]
    assert ts["deterministic_fixed_work"] > 2.0 and ts["deterministic_converged"] > 10.0 and 1.0 < ts["brownian_converged"] < 3.0
    cpu = s["cpu_timesteps_per_sec"]
    assert cpu["1core"]["cores"] == 1 and cpu["allcores"]["brownian_converged"] < 1.0
    assert s["mf_gflops"] > 1.0e4 and set(s["roofline_frac"]) >= {"cfg3_apply_M", "cfg1_apply_M", "cfg2_apply_M", "cfg5_build_hbm",
                                                                  "cfg5_cholesky_mfma", "cfg5_LW_hbm"}
    assert s["dropin_cfg3_ms"]["scipy_gmres"] > s["dropin_cfg3_ms"]["rbl_gmres_saddle"]
    assert not any(isinstance(v, str) and len(v) > 120 for v in json.loads(text)["summary"].values())
    # ... and on round 5's own record, which carries every block the summary can hold (matched tolerances, multi-RHS solve, both SciPy figures)
    full5 = json.load(open(os.path.join(root, "profiles", "r05_bench_detail_final.json")))
    text5 = json.dumps(bench.slim_line(full5))
    assert len(text5) < 4500 and text5[-2000:].index('"summary"') >= 0
    s5 = json.loads(text5)["summary"]
    assert s5["multi_rhs"]["ratio_to_sequential"] < 0.35 and s5["multi_rhs"]["column_vs_sequential_solve"] < 1e-10
    assert s5["brownian_gmres_rtol_matched_to_root"]["tol_1e-3"]["U_err_vs_1e-8_solve"] < 1e-2
    assert s5["timesteps_per_sec"]["fixed_work_residual"] > 1e-5                      # (the fixed-work step says it is not converged)
    # every prose key of the old line now lives in the notes, and the notes explain every block of the summary
    for k in ("timesteps_per_sec", "cpu_timesteps_per_sec", "roofline_frac", "dropin_cfg3_ms", "brownian_gmres_rtol_matched_to_root"):
        assert "summary." + k in bench.NOTES


@pytest.mark.parametrize("NT", [1, 2, 3, 5, 16, 17, 61])
def test_tile_factorisation_queue_order_is_a_topological_order(NT):
    """The dataflow tile factorisation (csrc/rbl_tilechol.hip) claims its tasks in a fixed order per body and lets a claimed task WAIT
    for the tiles it reads.  That cannot deadlock iff every task comes after everything it waits for -- checked here on the host, on
    the very function the kernel decodes its queue slots with: every CHOL(i, j) and INV(j, i) tile exactly once, the K-panel tiles,
    the diagonal tile of the column and (for the inverse) the complete row of L before it."""
    import ctypes as C
    from rigid_body_light_amd._lib import lib
    L = lib()
    out = (C.c_int * (3 * (NT + 1) * (NT + 1)))()
    assert L.rbl_debug_tile_order(NT, out) == 0
    order = [tuple(out[3 * k: 3 * k + 3]) for k in range((NT + 1) * (NT + 1))]
    order = [t for t in order if t[0] != 0]
    pos = {t: k for k, t in enumerate(order)}
    assert len(pos) == len(order)
    expect = {(1, i, j) for j in range(NT) for i in range(j, NT)} | {(2, jj, i) for i in range(NT) for jj in range(i + 1)}
    assert set(order) == expect
    for (kind, a, b), p in pos.items():
        if kind == 1:       # CHOL(i = a, j = b): rows i and j of L in the columns before j, then the diagonal tile of column j
            deps = [(1, a, k) for k in range(b)] + [(1, b, k) for k in range(b)] + ([(1, b, b)] if a != b else [])
        else:               # INV(row jj = a, column i = b): row i of L complete, Y(jj, jj .. i - 1)
            deps = [(1, b, k) for k in range(b + 1)] + [(2, a, k) for k in range(a, b)]
        assert all(pos[d] < p for d in deps)
