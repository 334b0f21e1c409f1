"""Emulated per-rank budget of ONE converged Brownian time step of BASELINE configs[3] (200 x shell_N_642, wall) at P = 8,
measured on ONE GPU (no 8-GPU node is available to the builder; the driver measures the real curve):
  1. the real single-GPU step with librbl's phase timings (rbl_get_timings): products, per-body work, factor build and
     what is left (Krylov vector work, K operators, host tests) -- the part that is REPLICATED on every rank;
  2. rank by rank (r = 0 .. P-1), the pieces a rank executes: its share of the tile pairs of a one- and of a two-vector
     product (rbl_apply_M_sym[_multi]_dev(r, P)), the applications of ITS 25 bodies' factors / inverses, their build;
  3. budget = counts of the real step x the slowest rank's piece + the replicated part, before collectives.
usage: bench_step_budget_p8.py [P] [option=value ...]   (e.g. 8 block_inverse_f32=1)"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from rigid_body_light_amd import make_config
from rigid_body_light_amd._lib import DeviceContext, lib
from rigid_body_light_amd.krylov import BrownianStepper
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tunes = list(sys.argv[2:])
nb, nblb, wall = 200, 642, True
dev = torch.device("cuda:0")
c = make_config(nb, nblb, wall)
N = nb * nblb; n3 = 3 * N; m_ = 3 * nblb
Fb = np.tile([0.0, 0.0, -1.0, 0.0, 0.0, 0.0], nb)
st = torch.cuda.current_stream()


def new_ctx(extra=()):
    ctx = DeviceContext(c["a"], c["eta"], wall, cfg=c["cfg"], dt=c["dt"], kBT=1.0, stream_ptr=st.cuda_stream)
    lib().rbl_set_blk_pc(ctx.h, 1)
    ctx.set_config(c["X"], c["Q"]); ctx.set_lanczos(200, 1e-3); ctx.set_block_refresh(2)
    for t in tuple(tunes) + tuple(extra):
        ctx.set_option(str(t).split("=")[0], int(str(t).split("=")[1]))
    return ctx


def ev_ms(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):                      # (the first launches after a pause run ~10 % slow: clocks)
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# ---- 1. the real step on one GPU
ctx = new_ctx()
bst = BrownianStepper(ctx, nb, nblb, dev)
bst.step(Fb, seed=0, method=2, iters=200, rtol=1e-8)
ctx.set_timing(True); ctx.reset_timings()
K = 3
torch.cuda.synchronize(); t0 = time.perf_counter()
its = []
for k in range(K):
    m, r = bst.step(Fb, seed=1 + k, method=2, iters=200, rtol=1e-8); its.append(m)
torch.cuda.synchronize(); T1 = (time.perf_counter() - t0) / K * 1e3
tm = ctx.timings(); ctx.set_timing(False)
L = ctx.lanczos_report()[0]
mg = int(round(sum(its) / len(its)))
ph = {k: tm[k][0] / K for k in tm}
calls = {k: tm[k][1] / K for k in tm}
replicated = T1 - ph["product"] - ph["per_body"] - ph["factor"]
print("# Brownian step of cfg 4 at P = %d: emulated per-rank budget (one GPU; tuning %s)\n" % (P, tunes or "default"))
print("single-GPU step: **%.1f ms** wall (GMRES %s iterations, Lanczos %d); librbl phases per step: product %.1f ms (%d brackets), "
      "per-body %.1f ms (%d), factor build %.1f ms (%d), solver calls in total %.1f ms; the rest of the step -- Krylov vector work, "
      "K operators, uploads, host convergence tests, evolve: **%.1f ms, replicated on every rank**\n"
      % (T1, its, L, ph["product"], calls["product"], ph["per_body"], calls["per_body"], ph["factor"], calls["factor"], ph["total"], replicated))
ctx.close()

# ---- 2. the pieces, rank by rank
ctx = new_ctx(("block_explicit_large=1",))                        # the multi-GPU default: explicit inverses of the rank's bodies
r = torch.empty(n3, dtype=torch.float64, device=dev)
ctx.blob_positions(0, nb, r.data_ptr())
x2 = torch.randn(2 * n3, dtype=torch.float64, device=dev)
o2 = torch.empty_like(x2)
v = torch.randn(n3, dtype=torch.float64, device=dev); o = torch.empty_like(v)
base, rem = divmod(nb, P)
ctx.set_option("block_explicit_large", 1)
ctx.block_solve(v.data_ptr(), o.data_ptr(), 0, 0, base + (1 if rem else 0))        # first build: also sizes the factor / inverse storage (12 GB of hipMalloc)
ctx.sync_check()
rows = []
for rk in range(P):
    b0 = rk * base + min(rk, rem); b1 = b0 + base + (1 if rk < rem else 0)
    p1 = ev_ms(lambda: ctx.apply_M_sym(x2.data_ptr(), r.data_ptr(), N, rk, P, o2.data_ptr()), 10)
    p2 = ev_ms(lambda: ctx.apply_M_sym_multi(x2.data_ptr(), r.data_ptr(), N, 2, rk, P, o2.data_ptr()), 10)
    ctx.set_option("block_explicit_large", 1)                                        # invalidates the factors: the next call builds this rank's only
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    ctx.block_solve(v.data_ptr(), o.data_ptr(), 0, b0, b1)
    e1.record(); torch.cuda.synchronize()
    pb = [ev_ms(lambda md=md: ctx.block_solve(v.data_ptr(), o.data_ptr(), md, b0, b1), 10) for md in (0, 1, 2, 3)]
    fb = e0.elapsed_time(e1) - pb[0]
    rows.append((rk, b1 - b0, p1, p2, fb, pb))
# the two-level factor of the Lanczos root is rebuilt per configuration on EVERY rank (replicated, apart from Z = L^-1 K_t)
def tl_rebuild():
    ctx.set_option("lanczos_two_level", 1)                                        # invalidates the two-level factor only
    ctx.block_solve(v.data_ptr(), o.data_ptr(), 5)
t_tl = ev_ms(tl_rebuild, 5) - ev_ms(lambda: ctx.block_solve(v.data_ptr(), o.data_ptr(), 5), 5)
ctx.close()
print("| rank | bodies | 1-vector product share | 2-vector product share | factor + inverse build | (L L^T)^-1 v | L^-1 v | L^-T v | L v |")
print("|---|---|---|---|---|---|---|---|---|")
for rk, nbr, p1, p2, fb, pb in rows:
    print("| %d | %d | %.3f | %.3f | %.2f | %.3f | %.3f | %.3f | %.3f |" % (rk, nbr, p1, p2, fb, pb[0], pb[1], pb[2], pb[3]))
mx = lambda f: max(f(rw) for rw in rows)
p1, p2, fb = mx(lambda rw: rw[2]), mx(lambda rw: rw[3]), mx(lambda rw: rw[4])
pb = [mx(lambda rw, i=i: rw[5][i]) for i in range(4)]
print("| **max** | | **%.3f** | **%.3f** | **%.2f** | **%.3f** | **%.3f** | **%.3f** | **%.3f** |\n" % (p1, p2, fb, pb[0], pb[1], pb[2], pb[3]))
print("(milliseconds; the per-body pieces read ONE vector -- the two Lanczos vectors and the three columns of M^-1 K share a pass over the matrix)\n")

# ---- 3. the budget
n_single = mg + 2                        # GMRES iterations + the two products of M_RFD
n_pair = L                               # lock-step Lanczos iterations
n_pc = mg + 1 + 2                        # preconditioner applications (iterations + x = P^-1 z) + the two passes of M^-1 K
items = [("two-vector products (Lanczos)", n_pair, p2), ("one-vector products (GMRES + M_RFD)", n_single, p1),
         ("(L L^T)^-1 applications (preconditioner, M^-1 K)", n_pc, pb[0]), ("L^-T and L^-1 applications (Lanczos)", 2 * L, 0.5 * (pb[1] + pb[2])),
         ("L v (the two increments)", 2, pb[3]), ("factor + inverse build (every 2nd configuration of two per step)", 1, fb),
         ("two-level factor of the root: sphere tensor, 3 N_bod-square Cholesky + inverse (replicated)", 1, t_tl),
         ("replicated: vector work, K operators, host", 1, replicated)]
print("| piece | count per step | slowest rank, ms each | ms per step |")
print("|---|---|---|---|")
tot = 0.0
for name, cnt, each in items:
    print("| %s | %d | %.3f | %.2f |" % (name, cnt, each, cnt * each)); tot += cnt * each
print("| **per-rank budget before collectives** | | | **%.1f** |" % tot)
print("\nsingle-GPU step / budget = %.1f / %.1f = **%.2fx** at P = %d (target of item 1d: >= 6.5x before collectives; a product then adds one "
      "3.1 MB all-reduce, a per-body application one more)" % (T1, tot, T1 / tot, P))
