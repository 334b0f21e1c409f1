"""Multi-rank rehearsal of the sharded paths on ONE GPU: two processes (gloo process group, both on cuda:0)
run the same code the 8-GPU job runs over RCCL -- tile-pair-sharded apply_M with its all-gather / all-reduce,
and the sharded stochastic midpoint step -- and compare with the oracle / the single-process result."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(nproc, script_args, timeout=300):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = None
    for attempt in range(2):       # a second try on another port if the rendezvous itself could not be set up
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + script_args
        p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
        if p.returncode == 0 or not any(k in p.stderr for k in ("EADDRINUSE", "address already in use", "RendezvousConnectionError")):
            break
    return p


def test_two_rank_sharded_apply_M_matches_oracle(orc, tmp_path):
    """bench.py's N = 2 path (cfg 2 size): every rank's partial-sum result, all-reduced, against the CPU oracle"""
    import json
    import numpy as np
    from rigid_body_light_amd import make_config
    dump = str(tmp_path / "chk")
    p = _torchrun(2, ["bench.py", "--gpus", "2", "--backend", "gloo", "--config", "cfg2", "--steps", "3", "--warmup", "1",
                      "--dump-check", dump, "--cpu-budget", "0", "--timestep-steps", "1", "--detail", dump + ".json"])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    full = json.load(open(dump + ".json"))                 # the sidecar: the full record behind the slim line
    assert d["n_gpus"] == 2 and full["timestep"]["apply_M_per_timestep"] == 21
    assert list(d)[-1] == "summary" and d["summary"]["timesteps_per_sec"]["brownian_converged"] > 0.0
    nb, nblb, wall = 50, 162, False                       # cfg2, as bench.py builds it
    c = make_config(nb, nblb, wall)
    F = np.random.default_rng(2).standard_normal(3 * nb * nblb)
    r = orc.multi_body_pos(c["X"], c["Q"], c["cfg"] - c["cfg"].mean(axis=0))
    for rank in range(2):
        z = np.load("%s.rank%d.npz" % (dump, rank))
        b0 = int(z["row0"])
        Uo = orc.apply_M_rows(F, r, b0, b0 + 8, c["a"], c["eta"], wall, nthreads=8)
        assert np.linalg.norm(z["values"] - Uo) / np.linalg.norm(Uo) < 1e-11


@pytest.mark.parametrize("timestep_steps", ["0", "1"])
def test_bench_self_launches_two_ranks(timestep_steps, tmp_path):
    """`python bench.py --gpus 2` WITHOUT torchrun (the way the driver calls it): bench.py starts the two ranks itself
    (a child torchrun, before touching the GPU) and the ONE line it prints says n_gpus = 2.  With time steps requested they
    run as a second 2-rank job (native sharded GMRES / Lanczos through rbl_set_comm) whose result is merged into the line."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RBL_BENCH_PHASE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--config", "cfg2", "--steps", "3",
                        "--warmup", "1", "--cpu-budget", "0", "--timestep-steps", timestep_steps, "--detail", str(tmp_path / "d.json")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "x2" in d["config"]["parallelism"]
    assert 0.0 < d["roofline"]["frac"] <= 1.0
    assert len(lines[0]) < 4500 and list(d)[-1] == "summary"
    if timestep_steps != "0":
        t = json.load(open(str(tmp_path / "d.json")))["timestep"]
        assert d["summary"]["timesteps_per_sec"]["deterministic_fixed_work"] > 0.0
        assert "error" not in t and t["apply_M_per_timestep"] == 21
        assert t["brownian_converged"]["lanczos_0.001"]["gmres_residual_max"] < 1e-8


@pytest.mark.parametrize("failing_rank", ["0", "1"])
def test_headline_line_survives_a_failing_time_step_part(monkeypatch, failing_rank):
    """the driver's scaling runs start bench.py under ONE torchrun job: if the time-step part fails on any rank after the hot
    path was timed, rank 0 still prints the one line (with the reason in `timestep.error`) and the job's status is non-zero"""
    import json
    monkeypatch.setenv("RBL_BENCH_INJECT_FAILURE", failing_rank)
    p = _torchrun(2, ["bench.py", "--gpus", "2", "--backend", "gloo", "--config", "cfg2", "--steps", "3", "--warmup", "1",
                      "--cpu-budget", "0", "--timestep-steps", "1"])
    assert p.returncode != 0
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:] + p.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and 0.0 < d["roofline"]["frac"] <= 1.0
    assert "rank %s" % failing_rank in d["timestep_error"] and "injected failure" in d["timestep_error"]
    assert d["summary"]["failed_parts"] == ["timestep"]


def _max_diff(stdout, world):
    line = [l for l in stdout.splitlines() if l.startswith("world %d:" % world)][-1]
    return float(line.split("=")[1].split(",")[0])


@pytest.mark.parametrize("native", ["1", "0"])
def test_two_rank_sharded_brownian_step_matches_single_process(monkeypatch, native):
    """native = 1: librbl's own Lanczos / GMRES loops with the communicator callback (rbl_set_comm); 0: the torch loops"""
    monkeypatch.setenv("RBL_CHECK_NATIVE", native)
    p = _torchrun(2, ["tools/check_sharded_brownian.py"])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "world 2" in p.stdout and _max_diff(p.stdout, 2) < 1e-10


@pytest.mark.parametrize("world,split", [(2, "0"), (3, "0"), (2, "1")])
def test_lock_step_multi_rhs_gmres_on_a_sharded_context(monkeypatch, world, split):
    """rbl_gmres_saddle_multi_dev on a multi-rank context (gloo rehearsal, several ranks on one GPU; 7 bodies: ragged shares at 2 and 3
    ranks): every product of every column is the rank's share + one all-reduce (rows + all-gather with split 1), every preconditioner
    application the owners' bodies + one all-gather; the five columns equal the single-process solves to 1e-9."""
    monkeypatch.setenv("RBL_CHECK_SPLIT", split)
    p = _torchrun(world, ["tools/check_sharded_multi_rhs.py"])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "world %d" % world in p.stdout


@pytest.mark.parametrize("split,allreduce_only", [("1", "0"), ("0", "1"), ("1", "1")])
def test_two_rank_step_with_row_split_and_with_the_allreduce_only_callbacks(monkeypatch, split, allreduce_only):
    """the library's sharded step with the ROW split (own bodies' geometry + all-gather of positions, ordered-pair kernel on own
    rows + all-gather of U; per-body results completed by the all-gather callback) and with the round-2/3 callback form
    (rbl_set_comm: all-reduce only, per-body results zero-padded and summed) against the single-process step"""
    monkeypatch.setenv("RBL_CHECK_SPLIT", split)
    monkeypatch.setenv("RBL_CHECK_ALLREDUCE_ONLY", allreduce_only)
    monkeypatch.setenv("RBL_CHECK_BLOCK_PC", "1")
    monkeypatch.setenv("RBL_CHECK_BODIES", "7")
    p = _torchrun(2, ["tools/check_sharded_brownian.py"])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "world 2" in p.stdout and _max_diff(p.stdout, 2) < 1e-10


@pytest.mark.parametrize("block_pc", ["0", "1"])
def test_three_rank_uneven_split_brownian_step(monkeypatch, block_pc):
    """7 bodies over 3 ranks (3 + 2 + 2): per-rank body ranges of the block factors -- of the preconditioned square root
    and (block_pc = 1) of the block-diagonal preconditioner, each rank substituting through its own bodies only"""
    monkeypatch.setenv("RBL_CHECK_BODIES", "7")
    monkeypatch.setenv("RBL_CHECK_BLOCK_PC", block_pc)
    p = _torchrun(3, ["tools/check_sharded_brownian.py"])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "world 3" in p.stdout and _max_diff(p.stdout, 3) < 1e-10


@pytest.mark.parametrize("world", [1, 2])
def test_cfg4_size_sharded_brownian_step(monkeypatch, world, orc, tmp_path):
    """BASELINE configs[3] at full size (200 x shell_N_642, wall-corrected + Brownian), ONE stochastic midpoint step:
    the multi-GPU driver (ShardedBrownianStepper: tile-pair-sharded products, block-Jacobi preconditioned Lanczos with
    per-rank body factors, block-PC GMRES) at world size 1 and as a 2-rank gloo rehearsal on one GPU, against the
    single-process BrownianStepper with the same injected noise -- and against the CPU ORACLE: the saddle system the
    sharded solve was given, evaluated by the oracle on every blob row of two whole bodies (one from each rank's share)
    and on their force / torque rows, is satisfied by the solution the sharded solve returned."""
    import numpy as np
    from rigid_body_light_amd import make_config
    dump = str(tmp_path / "cfg4_solve.npz")
    monkeypatch.setenv("RBL_CHECK_BODIES", "200")
    monkeypatch.setenv("RBL_CHECK_BLOBS", "642")
    monkeypatch.setenv("RBL_CHECK_BLOCK_PC", "1")
    monkeypatch.setenv("RBL_CHECK_LANCZOS_TOL", "1e-10")
    monkeypatch.setenv("RBL_CHECK_GMRES_TOL", "1e-10")
    monkeypatch.setenv("RBL_CHECK_DUMP", dump)
    p = _torchrun(world, ["tools/check_sharded_brownian.py"], timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "world %d" % world in p.stdout and _max_diff(p.stdout, world) < 1e-8
    nb, nblb = 200, 642
    c = make_config(nb, nblb, True)
    z = np.load(dump)
    n3 = 3 * nb * nblb
    lam, U, rhs = z["x"][:n3], z["x"][n3:].reshape(nb, 6), z["rhs"]
    r = orc.multi_body_pos(z["X"], z["Q"], c["cfg"] - c["cfg"].mean(axis=0))
    bnorm = np.linalg.norm(rhs)
    for b in (37, 163):                                   # [M lambda - K U ; K^T lambda] = rhs   (src/Rigid.py:73-80)
        rows = slice(3 * nblb * b, 3 * nblb * (b + 1))
        Ml = orc.apply_M_rows(lam, r, nblb * b, nblb * (b + 1), c["a"], c["eta"], True, nthreads=8)
        lever = r[rows].reshape(nblb, 3) - z["X"][b]
        KU = U[b, :3] + np.cross(U[b, 3:], lever)
        res_blob = Ml - KU.reshape(-1) - rhs[rows]
        lb = lam[rows].reshape(nblb, 3)
        ktl = np.concatenate([lb.sum(axis=0), np.cross(lever, lb).sum(axis=0)])
        res_body = ktl - rhs[n3 + 6 * b:n3 + 6 * b + 6]
        # a body's rows carry 1/200 of the residual on average: GMRES stopped at 1e-10 |rhs| for the whole vector
        assert np.linalg.norm(res_blob) < 2e-9 * bnorm and np.linalg.norm(res_body) < 2e-9 * bnorm


def test_world1_nccl_group_runs_the_rccl_code_path():
    """RCCL INSIDE librbl (rbl_comm_init_rccl: ncclAllReduce / ncclAllGather on the context's stream) and the callback form
    (rbl_set_comm_ops over torch.distributed `nccl` on device buffers) have only ever run under gloo with host staging in
    the rehearsals above.  A communicator of ONE rank drives exactly what N ranks run on one GPU -- rbl_gmres_saddle_dev with
    the body-sharded block preconditioner, the preconditioned Lanczos square root and a whole stochastic step, with both
    work splits (tile pairs + all-reduce; rows by body index + all-gather of positions and U), on torch's current stream and
    on a side stream -- equal to the un-sharded results (tools/check_nccl_world1.py; its own process)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "tools/check_nccl_world1.py"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "ALL OK" in p.stdout and "side stream" in p.stdout and "FAILED" not in p.stdout
    assert "RCCL in librbl, split 0" in p.stdout and "RCCL in librbl, split 1" in p.stdout and "callbacks, split 1" in p.stdout


def test_cpp_only_host_drives_rccl_inside_librbl(tmp_path):
    """examples/host_rccl_step.cpp: a C++ host with no Python and no PyTorch in the process (what the reference itself is,
    c_rigid_obj.cpp:997-1027) creates the unique id, initialises RCCL inside librbl and runs Brownian steps through the C ABI;
    the steps of its communicator context must reproduce the same steps on a plain single-GPU context (one rank here: a
    one-GPU box; the same binary takes N ranks, one process per GPU)."""
    exe = os.path.join(ROOT, "examples", "host_rccl_step")
    assert os.path.exists(exe), "examples/host_rccl_step is not built: run rigid_body_light_amd/build.py"
    csv = os.path.join(ROOT, "rigid_body_light_amd", "structures", "shell_N_162.csv")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for split in ("0", "1"):
        p = subprocess.run([exe, "0", "1", str(tmp_path / ("id%s" % split)), csv, "8", "3", split], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
        assert "HOST OK" in p.stdout and "communicator kind 2" in p.stdout and p.stdout.count("GMRES iterations") == 3


def test_bench_n_rank_code_path_on_one_rank_over_rccl(orc, tmp_path):
    """`bench.py --force-comm`: the code path of the driver's N = 2, 4, 8 runs -- nccl process group, RCCL inside librbl, both work
    splits timed, the sharded time steps -- with ONE rank on the one GPU there is: the line carries both partitionings with their
    rooflines and per-rank phases, and the product it timed matches the CPU oracle."""
    import json
    import numpy as np
    from rigid_body_light_amd import make_config
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "RBL_BENCH_PHASE"):
        env.pop(k, None)
    dump = str(tmp_path / "chk")
    p = subprocess.run([sys.executable, "bench.py", "--force-comm", "--config", "cfg2", "--steps", "3", "--warmup", "1", "--cpu-budget", "0",
                        "--timestep-steps", "1", "--dump-check", dump, "--detail", dump + ".json"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    slim = json.loads(lines[0])
    assert slim["n_gpus"] == 1 and "RCCL inside librbl" in slim["config"]["parallelism"] and len(lines[0]) < 4500
    assert slim["partitionings"]["rows"]["kernel"].startswith("k_apply_M<false>") and slim["partitionings"]["tile_pairs"]["collectives_per_step"] >= 1.0
    d = json.load(open(dump + ".json"))                    # the full record
    for name, kernel in (("tile_pairs", "k_apply_M_symw<false"), ("rows", "k_apply_M<false>")):   # (the wave-unit kernel, whichever rows per lane)
        part = d["partitionings"][name]
        assert part["roofline"]["kernel"].startswith(kernel) and 0.0 < part["roofline"]["frac"] <= 1.0
        assert part["per_rank"]["kernel_ms"]["max"] > 0.0 and part["per_rank"]["collectives_per_step"] >= 1.0
    t = d["timestep"]
    assert "error" not in t and t["brownian_converged"]["lanczos_0.001"]["gmres_residual_max"] < 1e-8
    assert t["brownian_converged"]["lanczos_0.001_rows"]["gmres_residual_max"] < 1e-8
    assert d["timesteps_per_sec"]["brownian_converged"] > 0.0
    nb, nblb, wall = 50, 162, False
    c = make_config(nb, nblb, wall)
    F = np.random.default_rng(2).standard_normal(3 * nb * nblb)
    r = orc.multi_body_pos(c["X"], c["Q"], c["cfg"] - c["cfg"].mean(axis=0))
    z = np.load("%s.rank0.npz" % dump)
    b0 = int(z["row0"])
    Uo = orc.apply_M_rows(F, r, b0, b0 + 8, c["a"], c["eta"], wall, nthreads=8)
    assert np.linalg.norm(z["values"] - Uo) / np.linalg.norm(Uo) < 1e-11
