"""Body-sharded multi-GPU apply_M: one process per GPU,
bodies split contiguously by body index, ONE exchange step -- an all-gather of
blob positions (once per configuration) and of the force vector (once per
matvec) over torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in CPU tests).

Two per-rank work splits:
  rows      (apply_M_local)     rank owns its bodies' rows, ordered-pair kernel, no reduction;
  symmetric (apply_M_allreduce) rank owns the unordered blob-tile pairs {I,J>=I} of one row tile I in every
            `world` consecutive ones (offset rank, mirrored in odd groups -> balanced triangle), symmetric kernel (each
            unordered pair once, ~1.5x less arithmetic), then ONE all-reduce of the 24 N-byte
            partial U.

The per-rank compute is the HIP kernel (DeviceContext.apply_M / apply_M_sym on the rank's share): the two methods
`apply_M_rows` and `apply_M_sym_part` are the only places that touch it.
"""
import torch
import torch.distributed as dist


def body_partition(n_bodies, world_size):
    """Contiguous split, sizes differ by at most one: -> list of (begin, end)."""
    base, rem = divmod(n_bodies, world_size)
    out, b = [], 0
    for r in range(world_size):
        e = b + base + (1 if r < rem else 0)
        out.append((b, e))
        b = e
    return out


class ShardedMobility:
    def __init__(self, n_bodies, blobs_per_body, group=None, device=None, ctx=None, force_collectives=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_bodies, self.nblb = n_bodies, blobs_per_body
        self.parts = body_partition(n_bodies, self.world)
        self.b0, self.b1 = self.parts[self.rank]
        self.row0, self.row1 = self.b0 * blobs_per_body, self.b1 * blobs_per_body
        self.n_blobs = n_bodies * blobs_per_body
        self.max_rows = max(e - b for b, e in self.parts) * blobs_per_body
        self.device = device if device is not None else torch.device("cpu")
        self.ctx = ctx
        self.r_full = None
        # force_collectives: issue every collective even in a group of ONE rank (a one-GPU box can then drive the RCCL
        # code path -- all_gather_into_tensor / all_reduce on device buffers -- that N ranks run; world == 1 otherwise
        # short-circuits them)
        self.collectives = bool(dist.is_initialized() and (self.world > 1 or force_collectives))
        # CPU-staged collectives when the process group cannot move device tensors (gloo rehearsal
        # of the multi-rank path on a single GPU); RCCL ("nccl") moves device buffers directly.
        self.stage_cpu = bool(dist.is_initialized() and dist.get_backend(group) == "gloo"
                              and self.device.type == "cuda")
        self.n_all_gather = self.n_all_reduce = 0      # collectives issued (tests, bench's per_rank block)

    # -- exchange -----------------------------------------------------------
    def all_gather_rows(self, local):
        """local: (rows_local*3,) -> (n_blobs*3,) in body order.  Shards are padded to the
        largest shard so a single all_gather_into_tensor moves everything."""
        if not self.collectives:
            return local.clone()
        self.n_all_gather += 1
        if local.numel() == self.max_rows * 3:      # equal shards (the usual case): no staging copy
            pad = local
        else:
            pad = torch.zeros(self.max_rows * 3, dtype=local.dtype, device=local.device)
            pad[: local.numel()] = local
        if self.stage_cpu:
            hbuf = torch.empty(self.world * self.max_rows * 3, dtype=local.dtype)
            dist.all_gather_into_tensor(hbuf, pad.cpu(), group=self.group)
            buf = hbuf.to(local.device)
        else:
            buf = torch.empty(self.world * self.max_rows * 3, dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(buf, pad, group=self.group)
        if all((e - b) * self.nblb == self.max_rows for b, e in self.parts):
            return buf
        chunks = [buf[r * self.max_rows * 3: r * self.max_rows * 3 + (e - b) * self.nblb * 3]
                  for r, (b, e) in enumerate(self.parts)]
        return torch.cat(chunks)

    def set_positions_local(self, r_local):
        """r_local: this rank's blob positions (rows_local*3,).  One all-gather per configuration."""
        self.r_full = self.all_gather_rows(r_local.contiguous())
        return self.r_full

    # -- matvec ---------------------------------------------------------------
    def apply_M_local(self, F_local, gather_output=False):
        """F_local: this rank's slice of the force vector.  Returns this rank's slice of
        U = M F (or the full vector if gather_output)."""
        F_full = self.all_gather_rows(F_local.contiguous())
        U_local = self.apply_M_rows(F_full)
        return self.all_gather_rows(U_local) if gather_output else U_local

    def apply_M_allreduce(self, F_local):
        """Symmetric-kernel sharding: every rank evaluates the unordered tile pairs of its share of
        the row tiles (rbl_apply_M_sym_dev: i_first = rank, i_step = world) and produces a PARTIAL full-length U; one
        all-reduce (sum) completes it.  Returns the full U on every rank."""
        F_full = self.all_gather_rows(F_local.contiguous())
        return self.all_reduce_sum(self.apply_M_sym_part(F_full))

    def apply_M_sym_part(self, F_full):
        """this rank's share of the unordered tile pairs -> partial full-length U (rbl_apply_M_sym_dev)"""
        if self.ctx is None:
            raise RuntimeError("ShardedMobility needs a DeviceContext (HIP) to compute with")
        part = torch.empty(self.n_blobs * 3, dtype=torch.float64, device=self.device)
        self.ctx.apply_M_sym(F_full.data_ptr(), self.r_full.data_ptr(), self.n_blobs, self.rank, self.world, part.data_ptr())
        return part

    def agree(self, flag):
        """rank 0's value of a control-flow decision on every rank: data-dependent loop exits (Lanczos convergence)
        must not depend on every rank having bitwise the same numbers, or one rank would leave a collective behind"""
        if not self.collectives:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=torch.device("cpu") if self.stage_cpu else self.device)
        dist.broadcast(t, src=0, group=self.group)
        return bool(int(t.item()))

    def all_reduce_sum(self, part):
        if self.collectives:
            self.n_all_reduce += 1
            if self.stage_cpu:
                h = part.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                part.copy_(h)
            else:
                dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)
        return part

    def all_gather_segments(self, view, offs, cnts):
        """in-place all-gather of per-rank segments of ONE device vector: rank r owns [offs[r], offs[r] + cnts[r]);
        view(a, k) returns the tensor over elements [a, a + k).  (librbl's all-gather callback in the rehearsals; the
        native communicator does this with ncclAllGather / grouped ncclBroadcast itself.)"""
        if not self.collectives:
            return
        self.n_all_gather += 1
        for r in range(self.world):
            if cnts[r] <= 0:
                continue
            seg = view(offs[r], cnts[r])
            src = dist.get_global_rank(self.group, r) if self.group is not None else r
            if self.stage_cpu:
                h = seg.cpu()
                dist.broadcast(h, src=src, group=self.group)
                if r != self.rank:
                    seg.copy_(h)
            else:
                dist.broadcast(seg, src=src, group=self.group)

    def apply_M_rows(self, F_full):
        """this rank's rows of U = M F with the ordered-pair kernel (rbl_apply_M_dev)"""
        if self.ctx is None:
            raise RuntimeError("ShardedMobility needs a DeviceContext (HIP) to compute with")
        nrows = self.row1 - self.row0
        out = torch.empty(nrows * 3, dtype=torch.float64, device=self.device)
        self.ctx.apply_M(F_full.data_ptr(), self.r_full.data_ptr(), self.n_blobs, self.row0, self.row1,
                         out.data_ptr())
        return out
