"""N>1 path on CPU: two gloo ranks exercise the body partition + all-gather exchange of
rigid_body_light_amd.dist with the ORACLE standing in for the per-rank HIP kernel."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_bodies, wall, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import Oracle
        from rigid_body_light_amd.dist import ShardedMobility
        from rigid_body_light_amd.synth import make_config
        orc = Oracle()
        c = make_config(n_bodies, 12, wall)
        a, eta = c["a"], 1.0
        cfg = c["cfg"] - c["cfg"].mean(axis=0)

        class OracleSharded(ShardedMobility):
            """partition + exchange logic of the product class, the CPU oracle standing in for the two kernel calls"""

            def apply_M_rows(self, F_full):
                return torch.from_numpy(orc.apply_M_rows(F_full.numpy(), self.r_full.numpy(), self.row0, self.row1, a, eta, wall))

            def apply_M_sym_part(self, F_full):
                # semantics of rbl_apply_M_sym_dev: unordered 64-blob tile pairs {I, J>=I}, I % step == first
                first, step = self.rank, self.world
                rr, FF = self.r_full.numpy(), F_full.numpy()
                B = orc.damp(rr, a) if wall else np.ones(rr.size)
                M = (B[:, None] * orc.rotne_prager_tensor(rr, a, eta, wall)) * B[None, :]
                n = rr.size // 3; T = (n + 63) // 64
                part = np.zeros(3 * n)
                for I in range(first, T, step):
                    ri = slice(192 * I, min(192 * (I + 1), 3 * n))
                    for J in range(I, T):
                        cj = slice(192 * J, min(192 * (J + 1), 3 * n))
                        part[ri] += M[ri, cj] @ FF[cj]
                        if J > I:
                            part[cj] += M[cj, ri] @ FF[ri]
                return torch.from_numpy(part)

        sm = OracleSharded(n_bodies, 12)
        # each rank computes ONLY its own bodies' blob positions (a8) ...
        r_local = torch.from_numpy(orc.multi_body_pos(c["X"][sm.b0:sm.b1], c["Q"][sm.b0:sm.b1], cfg))
        r_full = sm.set_positions_local(r_local)                      # ... one all-gather per configuration
        F = np.random.default_rng(2).standard_normal(3 * 12 * n_bodies)
        U_local = sm.apply_M_local(torch.from_numpy(F[3 * sm.row0:3 * sm.row1]))
        U_full = sm.apply_M_local(torch.from_numpy(F[3 * sm.row0:3 * sm.row1]), gather_output=True)
        U_sym = sm.apply_M_allreduce(torch.from_numpy(F[3 * sm.row0:3 * sm.row1]))
        if rank == 0:
            r_ref = orc.multi_body_pos(c["X"], c["Q"], cfg)
            U_ref = orc.apply_M(F, r_ref, a, eta, wall, mode="dense")
            ret["pos_ok"] = bool(np.array_equal(r_full.numpy(), r_ref))
            ret["err_full"] = float(np.abs(U_full.numpy() - U_ref).max() / np.abs(U_ref).max())
            ret["err_local"] = float(np.abs(U_local.numpy() - U_ref[3 * sm.row0:3 * sm.row1]).max() / np.abs(U_ref).max())
            ret["err_sym"] = float(np.abs(U_sym.numpy() - U_ref).max() / np.abs(U_ref).max())
            ret["parts"] = sm.parts
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_bodies,wall", [(6, False), (7, True), (30, True)])   # 7: uneven shards; 30: 6 tiles
def test_two_rank_body_sharded_apply_M(n_bodies, wall):
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), n_bodies, wall, ret), nprocs=2, join=True)
    assert ret["pos_ok"]
    assert ret["err_full"] < 1e-13 and ret["err_local"] < 1e-13 and ret["err_sym"] < 1e-13
    assert ret["parts"][0][0] == 0 and ret["parts"][-1][1] == n_bodies


def _gather_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rigid_body_light_amd.dist import ShardedMobility
        sm = ShardedMobility(7, 12)                        # 7 bodies over 2 ranks: 4 + 3 (ragged segments)
        n = 7 * 36
        buf = torch.full((n + 10,), -1.0, dtype=torch.float64)
        offs = [5 + 36 * b for b, _ in sm.parts]
        cnts = [36 * (e - b) for b, e in sm.parts]
        buf[offs[rank]:offs[rank] + cnts[rank]] = torch.arange(cnts[rank], dtype=torch.float64) + 1000.0 * (rank + 1)
        sm.all_gather_segments(lambda a, k: buf[a:a + k], offs, cnts)      # what librbl's all-gather callback does
        want = torch.full((n + 10,), -1.0, dtype=torch.float64)
        for r_ in range(world):
            want[offs[r_]:offs[r_] + cnts[r_]] = torch.arange(cnts[r_], dtype=torch.float64) + 1000.0 * (r_ + 1)
        ret[rank] = bool(torch.equal(buf, want)) and sm.n_all_gather == 1
    finally:
        dist.destroy_process_group()


def test_two_rank_in_place_all_gather_of_ragged_segments():
    """the all-gather librbl asks its communicator for (owners' segments of ONE vector, in place, ragged: rbl_allgatherv_fn) as the
    gloo rehearsals implement it (ShardedMobility.all_gather_segments); the native communicator does it with ncclAllGather /
    grouped ncclBroadcast on the GPU box"""
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_gather_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret[0] and ret[1]


def test_partition():
    from rigid_body_light_amd.dist import body_partition
    assert body_partition(200, 8) == [(25 * i, 25 * i + 25) for i in range(8)]
    p = body_partition(10, 4)
    assert [e - b for b, e in p] == [3, 3, 2, 2] and p[0][0] == 0 and p[-1][1] == 10
