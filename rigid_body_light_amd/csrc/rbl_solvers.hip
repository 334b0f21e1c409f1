// rbl_solvers.hip -- right-preconditioned GMRES on the saddle operator.
// Part of the implementation of the C ABI in include/rbl.h (split from the former rbl_api.hip along its sections);
// shared internals are declared in rbl_api_internal.hpp.  Nothing here falls back to a CPU path.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "rbl_api_internal.hpp"

// ---- GMRES on the saddle operator --------------------------------------------
// Right-preconditioned GMRES(max_iter), no restart:  A = apply_saddle (src/Rigid.py:73-80), P^-1 = apply_PC
// (c_rigid_obj.cpp:589-616), Arnoldi with classical Gram-Schmidt applied twice.  Everything stays on the
// device and the stream is not drained inside the loop: the Hessenberg matrix lives in HBM and is read back
// once at the end (fixed work, rtol <= 0) or, for the convergence test, every iteration (large systems) /
// every 4th (small, launch-bound ones).

static int gmres_saddle_core(rbl_ctx *c, const double *d_rhs, int max_iter, double rtol, double *d_x, int *iters_out,
                             double *resid_out);

// Small systems (<= 256 blobs, diagonal PC): geometry, preconditioner build and the whole Arnoldi / Givens loop in ONE
// kernel launch on one CU (rbl_small.hip) -- launch-bound otherwise (cfg 1: ~6 launches per iteration).
static int gmres_small(rbl_ctx *c, const double *d_rhs, const double *d_x0, int max_iter, double rtol, double *d_x,
                       int *iters_out, double *resid_out)
{
  int rc = ensure_xq_dev(c); if (rc) return rc;
  const RblBodyState &S = c->S;
  const size_t wd = rbl_gmres_small_work_doubles(S.N_blb, S.N_bod, max_iter);
  if ((rc = rbl_dev_reserve(c, c->d_gm, sizeof(double) * (wd + 2)))) return rc;
  double *work = (double *)c->d_gm.p, *scal = work + wd;
  const double *dX = (const double *)c->d_XQ.p, *dQ = dX + 3 * (size_t)S.N_bod;
  rc = rbl_launch_gmres_small(c->stream, rbl_make_params(S.a, S.eta), S.wall, dX, dQ, (const double *)c->d_cfg.p, S.N_blb,
                              S.N_bod, d_rhs, d_x0, d_x, max_iter, rtol, c->gmres_pc_sign_fix ? 1.0 : c->pc_fsign, work, scal,
                              c->d_err);
  if (rc == RBL_ERR_SIZE) return rc;       // the caller falls back to the general solver
  if (rc) return rbl_fail(c, rc, "gmres (one-kernel solver): launch failed");
  double hs[2] = {0.0, 0.0};
  RBL_HIP(c, hipMemcpyAsync(hs, scal, sizeof(hs), hipMemcpyDeviceToHost, c->stream));
  if ((rc = finish_and_check(c))) return rc;
  int it = 0;
  std::memcpy(&it, &hs[0], sizeof(int));
  if (iters_out) *iters_out = it;
  if (resid_out) *resid_out = hs[1];
  return RBL_OK;
}

// use_x0 != 0: d_x holds an initial guess (e.g. the previous time step's solution): the solver iterates on the
// residual b - A x0 (one extra product) and the tolerance stays relative to |b|.
int rbl_gmres_saddle_dev(rbl_ctx *c, const double *d_rhs, int max_iter, double rtol, double *d_x, int use_x0,
                         int *iters_out, double *resid_out)
{
  RblPhase ph_total(c, RBL_T_TOTAL);
  {
    int rc = need_config(c); if (rc) return rc;
    if ((rc = rbl_dev_init(c))) return rc;
    if (!d_rhs || !d_x || max_iter < 1) return rbl_fail(c, RBL_ERR_ARG, "gmres: bad arguments");
    if (c->gmres_small && !comm_on(c) && rbl_gmres_small_fits(c->S.N_blb, c->S.N_bod, max_iter, c->S.block_pc)) {
      rc = gmres_small(c, d_rhs, use_x0 ? d_x : nullptr, max_iter, rtol, d_x, iters_out, resid_out);
      if (rc != RBL_ERR_SIZE) return rc;
      c->gmres_small = false;                // this runtime does not grant the LDS the one-kernel solver needs
    }
  }
  if (!use_x0) return gmres_saddle_core(c, d_rhs, max_iter, rtol, d_x, iters_out, resid_out);
  int rc = sync_bodies(c); if (rc) return rc;
  if (!d_rhs || !d_x) return rbl_fail(c, RBL_ERR_ARG, "gmres: bad arguments");
  const int64_t nsys = (int64_t)3 * c->S.N_bod * c->S.N_blb + (int64_t)6 * c->S.N_bod;
  const size_t vb = sizeof(double) * (size_t)nsys;
  if ((rc = rbl_dev_reserve(c, c->d_bd2, 3 * vb + sizeof(double) * (2 + 2 * 512)))) return rc;   // + dot2 scratch
  double *r0 = (double *)c->d_bd2.p, *dx = r0 + nsys, *x0 = dx + nsys, *dn = x0 + nsys;
  RBL_HIP(c, hipMemcpyAsync(x0, d_x, vb, hipMemcpyDeviceToDevice, c->stream));
  if ((rc = rbl_apply_saddle_dev(c, x0, r0))) return rc;
  rbl_launch_axpby(c->stream, nsys, 1.0, d_rhs, -1.0, r0, r0);                          // r0 = b - A x0
  double nn[2] = {0.0, 0.0};
  rbl_launch_dot2(c->stream, d_rhs, d_rhs, nullptr, nsys, dn);                          // |b|^2
  if ((rc = read_back(c, &nn[0], dn, sizeof(double)))) return rc;
  rbl_launch_dot2(c->stream, r0, r0, nullptr, nsys, dn);                                // |r0|^2
  if ((rc = read_back(c, &nn[1], dn, sizeof(double)))) return rc;
  const double nb2 = nn[0], nr2 = nn[1];
  const double scale = (nb2 > 0.0 && nr2 > 0.0) ? std::sqrt(nb2 / nr2) : 1.0;          // |b| / |r0|
  if (nr2 == 0.0) { if (iters_out) *iters_out = 0; if (resid_out) *resid_out = 0.0; return RBL_OK; }   // x0 already solves it
  double resid = 0.0;
  if ((rc = gmres_saddle_core(c, r0, max_iter, rtol > 0.0 ? rtol * scale : rtol, dx, iters_out, &resid))) return rc;
  rbl_launch_axpby(c->stream, nsys, 1.0, x0, 1.0, dx, d_x);                             // x = x0 + dx
  if (resid_out) *resid_out = resid / scale;
  return finish_and_check(c);
}

static int gmres_saddle_core_(rbl_ctx *c, const double *d_rhs, int max_iter, double rtol, double *d_x, int *iters_out,
                              double *resid_out);

// any invertible right preconditioner leaves the solution unchanged: inside the solve the force block of apply_PC
// takes the sign that makes A P^-1 ~ I (see rbl_ctx::pc_fsign); the bound apply_PC keeps the reference's convention
static int gmres_saddle_core(rbl_ctx *c, const double *d_rhs, int max_iter, double rtol, double *d_x, int *iters_out,
                             double *resid_out)
{
  const double keep = c->pc_fsign;
  if (c->gmres_pc_sign_fix) c->pc_fsign = 1.0;
  const int rc = gmres_saddle_core_(c, d_rhs, max_iter, rtol, d_x, iters_out, resid_out);
  c->pc_fsign = keep;
  return rc;
}

static int gmres_saddle_core_(rbl_ctx *c, const double *d_rhs, int max_iter, double rtol, double *d_x, int *iters_out,
                              double *resid_out)
{
  int rc = sync_bodies(c); if (rc) return rc;
  if (!d_rhs || !d_x || max_iter < 1) return rbl_fail(c, RBL_ERR_ARG, "gmres: bad arguments");
  if (max_iter + 1 > rbl_gmres_max_vectors()) return rbl_fail(c, RBL_ERR_ARG, "gmres: at most 255 iterations (no restart)");
  const RblBodyState &S = c->S;
  const int64_t nsys = (int64_t)3 * S.N_bod * S.N_blb + (int64_t)6 * S.N_bod;
  const int m = max_iter, ldh = m + 1;
  const size_t vb = sizeof(double) * (size_t)nsys;
  // workspace: V[(m+1)][nsys] | w | z | H[(m+1) x m] column-major | beta | y[m] | partial sums
  const size_t need = vb * (size_t)(m + 3) + sizeof(double) * ((size_t)ldh * m + 1 + m + rbl_gmres_part_doubles() +
                                                               rbl_lanczos_part_doubles());
  if ((rc = rbl_dev_reserve(c, c->d_gm, need))) return rc;
  // (beta sits in FRONT of H: a convergence test reads back 1 + ldh * used doubles, not the whole ldh x m array)
  double *V = (double *)c->d_gm.p, *w = V + (size_t)(m + 1) * nsys, *z = w + nsys, *d_beta = z + nsys, *H = d_beta + 1,
         *d_y = H + (size_t)ldh * m, *part = d_y + m, *part2 = part + rbl_gmres_part_doubles();
  // (H is not cleared: the least-squares solve reads rows 0 .. j + 1 of column j only, all written by the Arnoldi step)
  rbl_launch_lanczos_init(c->stream, nsys, d_rhs, d_beta, V, part2);                    // V_0 = b/|b|, beta = |b|
  std::vector<double> Hh((size_t)ldh * m + 1), y;
  // convergence test: every iteration when an iteration is expensive.  When it is launch-bound a test (copy + stream
  // drain) costs as much as half an iteration: the first one waits until two iterations before the count the previous
  // solve needed (time steps repeat), later ones follow the residual's rate, at most four iterations apart; a test
  // looks at every iteration since the one before, so the solve still ends at the first iteration that passes.
  const int check_every = ((int64_t)S.N_bod * S.N_blb > 20000) ? 1 : 4;
  int next_check = check_every, last_checked = 0;
  if (check_every > 1 && c->gmres_predict && c->gmres_last_used > 0) next_check = c->gmres_last_used >= 8 ? c->gmres_last_used - 2 : c->gmres_last_used;
  int used = 0;
  double resid = 1.0;
  // least squares min |beta e1 - H_k y| by Givens rotations on a host copy; returns the residual estimate
  auto solve_ls = [&](int k, std::vector<double> &yout) -> double {
    std::vector<double> R(Hh.begin() + 1, Hh.begin() + 1 + (size_t)ldh * k), g((size_t)k + 1, 0.0);
    const double beta = Hh[0];
    g[0] = beta;
    std::vector<double> cs((size_t)k), sn((size_t)k);
    for (int j = 0; j < k; ++j) {
      double *col = R.data() + (size_t)j * ldh;
      for (int i = 0; i < j; ++i) {
        const double t = cs[i] * col[i] + sn[i] * col[i + 1];
        col[i + 1] = -sn[i] * col[i] + cs[i] * col[i + 1];
        col[i] = t;
      }
      const double den = std::hypot(col[j], col[j + 1]);
      cs[j] = den > 0.0 ? col[j] / den : 1.0;
      sn[j] = den > 0.0 ? col[j + 1] / den : 0.0;
      col[j] = den; col[j + 1] = 0.0;
      g[j + 1] = -sn[j] * g[j];
      g[j] = cs[j] * g[j];
    }
    yout.assign((size_t)k, 0.0);
    for (int i = k - 1; i >= 0; --i) {
      double v = g[i];
      for (int j = i + 1; j < k; ++j) v -= R[(size_t)j * ldh + i] * yout[j];
      const double d = R[(size_t)i * ldh + i];
      yout[i] = d != 0.0 ? v / d : 0.0;
    }
    return beta > 0.0 ? std::fabs(g[k]) / beta : 0.0;
  };
  // Overlapped convergence test (large systems, RBL_OPT_GMRES_OVERLAP_CHECK): the Hessenberg columns of iteration j go to the
  // host by an asynchronous copy; the preconditioner of iteration j + 1 is enqueued BEFORE the host waits for that copy, so
  // the stream never runs dry at a test -- on N GPUs no rank drains per iteration.  A solve that ends at j has applied one
  // preconditioner too many (0.2 ms at cfg 3 against a 20 ms product; enqueuing the whole next iteration would waste a product).
  const size_t pin_need = sizeof(double) * (1 + (size_t)ldh * m);
  // (launch-bound systems too, round 4: a test's round trip costs 40-50 us there, the preconditioner it hides behind 18)
  const bool overlap_ok = c->gmres_overlap && pin_need <= ((size_t)1 << 20);
  if (overlap_ok && !c->ev_check) RBL_HIP(c, hipEventCreateWithFlags(&c->ev_check, hipEventDisableTiming));
  if (overlap_ok && !c->h_pin) RBL_HIP(c, hipHostMalloc(&c->h_pin, (size_t)1 << 20, hipHostMallocDefault));
  bool z_ready = false;                                // z = P^-1 V_j is already enqueued (by the previous iteration's test)
  // Normalisation folded into the next preconditioner (launch-bound systems whose preconditioner takes it, pc_can_fold): the Arnoldi
  // step leaves w and the partial sums of |w|^2; P^-1 is linear, so the kernels that apply it to V_{j+1} = w / |w| read w, scale by
  // 1 / |w| themselves and store V_{j+1} and H[j+1][j] on the side -- one launch fewer per iteration.  A convergence test that
  // falls between the two gets |w| from the same partial sums on the host.  The last possible iteration normalises as before.
  RblNormFold pend;                                    // set: V_j and H[j][j-1] are still to be written, from w
  auto apply_pc = [&](int jv) -> int {                 // z = P^-1 V_jv
    c->ktl_arm = true;                                 // the PC's K^T Lambda by-product feeds the product that follows
    const double *src = V + (size_t)jv * nsys;
    if (pend.part) { c->pc_fold = pend; src = w; pend = RblNormFold(); }
    const int r = rbl_apply_PC_dev(c, src, z);
    if (r) c->ktl_arm = false;
    return r;
  };
  for (int j = 0; j < m; ++j) {
    if (!z_ready && (rc = apply_pc(j))) return rc;
    z_ready = false;
    // inexact Krylov: the j-th product may be in error by ~ rtol / |r_{j-1}| (relative); the relaxed kernel's ~1e-6 is
    // admissible once the residual estimate is below rtol x 1e5 (an order of magnitude in hand)
    c->sym_tune.relaxed = (c->gmres_relax == 1 && rtol > 0.0 && check_every == 1 && resid <= rtol * 1.0e5) ? 1 : 0;
    c->fuse_dotV = V; c->fuse_dotK = j + 1; c->fuse_dotPart = part;   // (small systems: the product's last kernel starts the Gram-Schmidt pass)
    rc = rbl_apply_saddle_dev(c, z, w);
    const int fused_np = c->fuse_dots_np;
    c->fuse_dotV = nullptr; c->fuse_dotK = 0; c->fuse_dotPart = nullptr; c->fuse_dots_np = 0;
    c->sym_tune.relaxed = 0;
    c->ktl_arm = false; c->ktl_of = nullptr;
    if (rc) return rc;
    double *Hcol = H + (size_t)j * ldh;
    // classical Gram-Schmidt twice, H[j+1][j] = |w|, V_{j+1} = w / |w|: four launches (three when the product left the first sums)
    const bool fold = j + 1 < m && pc_can_fold(c);
    const double *npart = nullptr; int nnp = 0;
    rbl_launch_arnoldi_step(c->stream, V, nsys, j + 1, w, Hcol, V + (size_t)(j + 1) * nsys, part, fused_np, fold, &npart, &nnp);
    if (fold) { pend.part = npart; pend.np = nnp; pend.vnext = V + (size_t)(j + 1) * nsys; pend.hout = Hcol + (j + 1); }
    used = j + 1;
    if (rtol > 0.0 && (used >= next_check || used == m)) {
      const size_t hb = sizeof(double) * (1 + (size_t)ldh * used);
      const size_t pb = fold ? sizeof(double) * (size_t)nnp : 0;       // H[j+1][j] is not on the device yet: its partial sums come along
      std::vector<double> hp((size_t)(fold ? nnp : 0));
      if (overlap_ok && used < m && hb + pb <= ((size_t)1 << 20)) {
        RBL_HIP(c, hipMemcpyAsync(c->h_pin, d_beta, hb, hipMemcpyDeviceToHost, c->stream));
        if (fold) RBL_HIP(c, hipMemcpyAsync((char *)c->h_pin + hb, npart, pb, hipMemcpyDeviceToHost, c->stream));
        RBL_HIP(c, hipEventRecord(c->ev_check, c->stream));
        if ((rc = apply_pc(j + 1))) return rc;         // iteration j + 1's preconditioner, ahead of the host's wait
        z_ready = true;
        RBL_HIP(c, hipEventSynchronize(c->ev_check));
        std::memcpy(Hh.data(), c->h_pin, hb);
        if (fold) std::memcpy(hp.data(), (const char *)c->h_pin + hb, pb);
      } else {
        if (fold && (rc = read_back(c, hp.data(), npart, pb))) return rc;
        if ((rc = read_back(c, Hh.data(), d_beta, hb))) return rc;
      }
      if (fold) {
        double s2 = 0.0;
        for (double v : hp) s2 += v;
        Hh[1 + (size_t)j * ldh + (size_t)(j + 1)] = std::sqrt(s2);
      }
      // the test may have become true anywhere since the last look: take the first k that passes
      int hit = 0;
      double r_before = resid;
      for (int k = last_checked + 1; k <= used; ++k) {
        r_before = resid;
        resid = solve_ls(k, y);
        if (resid < rtol) { hit = k; break; }
      }
      if (hit) { used = hit; c->ktl_arm = false; c->ktl_of = nullptr; break; }
      last_checked = used;
      int ahead = 1;
      if (check_every > 1 && !c->gmres_predict) ahead = check_every - (used % check_every);
      else if (check_every > 1) {                        // iterations the residual still needs at its current rate
        ahead = check_every;
        if (resid > 0.0 && r_before > resid) {
          const double rem = std::log(rtol / resid) / std::log(resid / r_before);
          ahead = rem < 1.0 ? 1 : (rem > (double)check_every ? check_every : (int)rem);
        }
      }
      next_check = used + ahead;
    }
  }
  if (!(rtol > 0.0) || y.size() != (size_t)used) {
    if ((rc = read_back(c, Hh.data(), d_beta, sizeof(double) * (1 + (size_t)ldh * used)))) return rc;
    resid = solve_ls(used, y);
  }
  for (int k = 0; k < used; ++k)
    if (!std::isfinite(y[k])) return rbl_fail(c, RBL_ERR_NONFINITE, "gmres: non-finite Hessenberg solve");
  if ((rc = upload_coef(c, d_y, y.data(), used, 0))) return rc;
  rbl_launch_lanczos_combine(c->stream, nsys, V, d_y, used, z);                        // z = V y
  if ((rc = rbl_apply_PC_dev(c, z, d_x))) return rc;                                   // x = P^-1 z
  if (iters_out) *iters_out = used;
  if (resid_out) *resid_out = resid;
  if (rtol > 0.0) c->gmres_last_used = used;
  return finish_and_check(c);
}

// ---- k right-hand sides in lock step: the mobility product of every iteration on the fp64 matrix cores ---------------------------
// The reference exposes the saddle operator for an outer Krylov loop (src/Rigid.py:69-80); with MANY right-hand sides for one
// configuration -- the 6 N_bod unit loads of the body mobility matrix, several noise realisations -- the loops can advance
// together: k independent right-preconditioned GMRES recurrences (each with its own Krylov basis, Hessenberg matrix and
// stopping test: exactly the iterates rbl_gmres_saddle_dev would produce one by one), whose k products per iteration are ONE
// launch of k_apply_M_mrhs (16 right-hand sides per pass through v_mfma_f64_16x16x4: 6.3 ms a vector at cfg 3 against 20.4 for
// the one-vector kernel) and whose k block-preconditioner applications share passes over the per-body factors.  Columns that
// have converged stop iterating (their slots ride along in the product, which costs the same for 13 as for 16).
static int gmres_multi_batch(rbl_ctx *c, const double *d_rhs, int k, int m, double rtol, double *d_x, int *iters_out, double *resid_out)
{
  const RblBodyState &S = c->S;
  const int64_t N = (int64_t)S.N_bod * S.N_blb, n3 = 3 * N, nsys = n3 + (int64_t)6 * S.N_bod;
  const int ldh = m + 1;
  const int64_t slots = m + 4;                          // per column: V_0 .. V_m | z | w | scratch
  const int64_t pitch = slots * nsys;
  const size_t hcol = 1 + (size_t)ldh * m;              // beta | H (column-major) of one column
  int rc;
  const size_t need = sizeof(double) * ((size_t)pitch * k + hcol * k + (size_t)m * k + rbl_gmres_part_doubles() + rbl_lanczos_part_doubles());
  if ((rc = rbl_dev_reserve(c, c->d_gm, need))) return rc;
  double *base = (double *)c->d_gm.p, *Hall = base + (size_t)pitch * k, *d_y = Hall + hcol * k, *part = d_y + (size_t)m * k,
         *part2 = part + rbl_gmres_part_doubles();
  auto Vc = [&](int col, int j) { return base + (size_t)col * (size_t)pitch + (size_t)j * (size_t)nsys; };
  auto Zc = [&](int col) { return Vc(col, m + 1); };
  auto Wc = [&](int col) { return Vc(col, m + 2); };
  auto Sc = [&](int col) { return Vc(col, m + 3); };
  for (int col = 0; col < k; ++col)
    rbl_launch_lanczos_init(c->stream, nsys, d_rhs + (size_t)col * (size_t)nsys, Hall + hcol * col, Vc(col, 0), part2);   // V_0 = b/|b|, beta = |b|
  std::vector<double> Hh(hcol * (size_t)k);
  std::vector<std::vector<double>> ys((size_t)k);
  std::vector<int> used((size_t)k, 0), done((size_t)k, 0);
  std::vector<double> resid((size_t)k, 1.0);
  auto solve_ls = [&](int col, int kk, std::vector<double> &yout) -> double {       // Givens on a host copy, as in the one-vector solver
    const double *hc = Hh.data() + hcol * (size_t)col;
    std::vector<double> R(hc + 1, hc + 1 + (size_t)ldh * kk), g((size_t)kk + 1, 0.0), cs((size_t)kk), sn((size_t)kk);
    const double beta = hc[0];
    g[0] = beta;
    for (int j = 0; j < kk; ++j) {
      double *cl = R.data() + (size_t)j * ldh;
      for (int i = 0; i < j; ++i) {
        const double t = cs[i] * cl[i] + sn[i] * cl[i + 1];
        cl[i + 1] = -sn[i] * cl[i] + cs[i] * cl[i + 1];
        cl[i] = t;
      }
      const double den = std::hypot(cl[j], cl[j + 1]);
      cs[j] = den > 0.0 ? cl[j] / den : 1.0;
      sn[j] = den > 0.0 ? cl[j + 1] / den : 0.0;
      cl[j] = den; cl[j + 1] = 0.0;
      g[j + 1] = -sn[j] * g[j];
      g[j] = cs[j] * g[j];
    }
    yout.assign((size_t)kk, 0.0);
    for (int i = kk - 1; i >= 0; --i) {
      double v = g[i];
      for (int j = i + 1; j < kk; ++j) v -= R[(size_t)j * ldh + i] * yout[j];
      const double d = R[(size_t)i * ldh + i];
      yout[i] = d != 0.0 ? v / d : 0.0;
    }
    return beta > 0.0 ? std::fabs(g[kk]) / beta : 0.0;
  };
  int n_done = 0;
  for (int j = 0; j < m && n_done < k; ++j) {
    // z_c = P^-1 V_c,j : all columns together (converged ones ride along: their slots are never read again)
    if ((rc = apply_PC_multi_dev(c, Vc(0, j), Zc(0), Sc(0), k, pitch))) return rc;
    // w_c = [M lambda - K U ; K^T lambda] : ONE multi-vector product, then the O(N) body terms column by column
    if ((rc = rbl_dev_reserve(c, c->d_sad, sizeof(double) * (size_t)n3 * (size_t)k))) return rc;
    if ((rc = apply_M_multi_enqueue(c, S.wall, Zc(0), (const double *)c->d_pos.p, N, k, (double *)c->d_sad.p, pitch, n3))) return rc;
    for (int col = 0; col < k; ++col) {
      if (done[(size_t)col]) continue;
      rbl_launch_K_x_U(c->stream, (const double *)c->d_lever.p, Zc(col) + n3, S.N_blb, N, Wc(col), (const double *)c->d_sad.p + (size_t)col * (size_t)n3, -1.0);
      rbl_launch_KT_x_Lam(c->stream, (const double *)c->d_lever.p, Zc(col), S.N_blb, S.N_bod, Wc(col) + n3);
      rbl_launch_arnoldi_step(c->stream, Vc(col, 0), nsys, j + 1, Wc(col), Hall + hcol * col + 1 + (size_t)j * ldh, Vc(col, j + 1), part);
      used[(size_t)col] = j + 1;
    }
    if (rtol > 0.0) {
      if ((rc = read_back(c, Hh.data(), Hall, sizeof(double) * hcol * (size_t)k))) return rc;
      for (int col = 0; col < k; ++col) {
        if (done[(size_t)col]) continue;
        resid[(size_t)col] = solve_ls(col, j + 1, ys[(size_t)col]);
        if (resid[(size_t)col] < rtol) { done[(size_t)col] = 1; ++n_done; }
      }
    }
  }
  if (!(rtol > 0.0) || n_done < k) {
    if ((rc = read_back(c, Hh.data(), Hall, sizeof(double) * hcol * (size_t)k))) return rc;
    for (int col = 0; col < k; ++col)
      if (!done[(size_t)col]) resid[(size_t)col] = solve_ls(col, used[(size_t)col], ys[(size_t)col]);
  }
  for (int col = 0; col < k; ++col) {
    const int u = used[(size_t)col];
    for (int q = 0; q < u; ++q)
      if (!std::isfinite(ys[(size_t)col][(size_t)q])) return rbl_fail(c, RBL_ERR_NONFINITE, "gmres (multi): non-finite Hessenberg solve");
    if ((rc = upload_coef(c, d_y + (size_t)m * col, ys[(size_t)col].data(), u, -1))) return rc;       // (synchronous: once per column and solve)
    rbl_launch_lanczos_combine(c->stream, nsys, Vc(col, 0), d_y + (size_t)m * col, u, Zc(col));      // z = V y
    if (iters_out) iters_out[col] = u;
    if (resid_out) resid_out[col] = resid[(size_t)col];
  }
  if ((rc = apply_PC_multi_dev(c, Zc(0), Wc(0), Sc(0), k, pitch))) return rc;                        // x = P^-1 z
  for (int col = 0; col < k; ++col)
    RBL_HIP(c, hipMemcpyAsync(d_x + (size_t)col * (size_t)nsys, Wc(col), sizeof(double) * (size_t)nsys, hipMemcpyDeviceToDevice, c->stream));
  return RBL_OK;
}

int rbl_gmres_saddle_multi_dev(rbl_ctx *c, const double *d_rhs, int nrhs, int max_iter, double rtol, double *d_x, int *iters_out,
                               double *resid_out)
{
  RblPhase ph_total(c, RBL_T_TOTAL);
  int rc = need_config(c); if (rc) return rc;
  if ((rc = rbl_dev_init(c))) return rc;
  if (!d_rhs || !d_x || nrhs < 1 || max_iter < 1) return rbl_fail(c, RBL_ERR_ARG, "gmres_saddle_multi: bad arguments");
  if (max_iter + 1 > rbl_gmres_max_vectors()) return rbl_fail(c, RBL_ERR_ARG, "gmres: at most 255 iterations (no restart)");
  if ((rc = sync_bodies(c))) return rc;
  const int64_t nsys = (int64_t)3 * c->S.N_bod * c->S.N_blb + (int64_t)6 * c->S.N_bod;
  const double keep = c->pc_fsign;
  if (c->gmres_pc_sign_fix) c->pc_fsign = 1.0;          // one eigenvalue cluster instead of two, as in the one-vector solver
  for (int k0 = 0; k0 < nrhs && !rc; k0 += 16) {
    const int kb = nrhs - k0 < 16 ? nrhs - k0 : 16;
    rc = gmres_multi_batch(c, d_rhs + (size_t)k0 * (size_t)nsys, kb, max_iter, rtol, d_x + (size_t)k0 * (size_t)nsys,
                           iters_out ? iters_out + k0 : nullptr, resid_out ? resid_out + k0 : nullptr);
  }
  c->pc_fsign = keep;
  if (rc) return rc;
  return finish_and_check(c);
}

// host vectors: rhs, x = nrhs vectors of n3 + 6 N_bod doubles, one after the other
int rbl_gmres_saddle_multi(rbl_ctx *c, const double *rhs, int nrhs, int max_iter, double rtol, double *x, int *iters, double *resid)
{
  int rc = need_K(c); if (rc) return rc;
  if ((rc = rbl_dev_init(c))) return rc;
  if (!rhs || !x || nrhs < 1 || max_iter < 1) return rbl_fail(c, RBL_ERR_ARG, "gmres_saddle_multi: bad arguments");
  const size_t nsys = (size_t)3 * c->S.N_bod * c->S.N_blb + (size_t)6 * c->S.N_bod;
  if ((rc = rbl_dev_reserve(c, c->d_step, sizeof(double) * 2 * nsys * (size_t)nrhs))) return rc;
  double *d_x = (double *)c->d_step.p, *d_rhs = d_x + nsys * (size_t)nrhs;
  c->step_hist_n = 0; c->step_x_size = 0;                 // (d_step is shared with the time-step entry points' history)
  if ((rc = copy_h2d(c, d_rhs, rhs, sizeof(double) * nsys * (size_t)nrhs))) return rc;
  if ((rc = rbl_gmres_saddle_multi_dev(c, d_rhs, nrhs, max_iter, rtol, d_x, iters, resid))) return rc;
  if ((rc = copy_d2h(c, x, d_x, sizeof(double) * nsys * (size_t)nrhs))) return rc;
  return finish_and_check(c);
}

// host-vector form: one upload, the device-resident solve, one download
int rbl_gmres_saddle(rbl_ctx *c, const double *rhs, int max_iter, double rtol, double *x, int use_x0, int *iters, double *resid)
{
  int rc = need_K(c); if (rc) return rc;
  if ((rc = rbl_dev_init(c))) return rc;
  if (!rhs || !x || max_iter < 1) return rbl_fail(c, RBL_ERR_ARG, "gmres_saddle: bad arguments");
  const size_t nsys = (size_t)3 * c->S.N_bod * c->S.N_blb + (size_t)6 * c->S.N_bod;
  if ((rc = rbl_dev_reserve(c, c->d_step, sizeof(double) * 2 * nsys))) return rc;
  double *d_x = (double *)c->d_step.p, *d_rhs = d_x + nsys;
  c->step_hist_n = 0; c->step_x_size = 0;                 // (d_step is shared with the time-step entry points' history)
  if ((rc = copy_h2d(c, d_rhs, rhs, sizeof(double) * nsys))) return rc;
  if (use_x0 && (rc = copy_h2d(c, d_x, x, sizeof(double) * nsys))) return rc;
  if ((rc = rbl_gmres_saddle_dev(c, d_rhs, max_iter, rtol, d_x, use_x0, iters, resid))) return rc;
  if ((rc = copy_d2h(c, x, d_x, sizeof(double) * nsys))) return rc;
  return finish_and_check(c);
}
