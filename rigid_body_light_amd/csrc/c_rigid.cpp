// c_rigid.cpp -- pybind11 module `c_rigid` exposing class CManyBodies with the
// same Python-visible names as the reference's nanobind module
// (reference src/c_rigid_obj.cpp:997-1027).  It is a thin shim: every method
// calls one extern "C" entry point of include/rbl.h (librbl.so) and nothing else.
//
// Differences a caller can observe (all listed in DESIGN.md):
//   * multi_body_pos() returns a numpy array instead of a Python list (the
//     reference wrapper wraps it in np.array() anyway, src/Rigid.py:55);
//   * error conditions that make the reference exit() raise RuntimeError;
//   * inputs are never modified (the reference mutates `cfg` and `U` C++-side).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rbl.h"

namespace py = pybind11;
using darr = py::array_t<double, py::array::c_style | py::array::forcecast>;

namespace {

static int mhalf_method(const std::string &method)
{
  if (method == "cholesky") return RBL_MHALF_CHOLESKY;
  if (method == "lanczos") return RBL_MHALF_LANCZOS;
  if (method == "lanczos_pc") return RBL_MHALF_LANCZOS_PC;
  throw std::runtime_error("M_half_W: method must be 'cholesky', 'lanczos' or 'lanczos_pc'");
}

struct CManyBodies {
  rbl_ctx *ctx;
  CManyBodies() : ctx(rbl_create())
  {
    if (!ctx) throw std::runtime_error("rbl_create failed");
  }
  ~CManyBodies() { rbl_destroy(ctx); }
  CManyBodies(const CManyBodies &) = delete;
  CManyBodies &operator=(const CManyBodies &) = delete;

  void check(int rc) const
  {
    if (rc != RBL_OK) throw std::runtime_error(std::string(rbl_last_error(ctx)) + " [rbl status " + std::to_string(rc) + "]");
  }
  int n_bod() const { int a = 0, b = 0; rbl_get_sizes(ctx, &a, &b); return a; }
  int n_blb() const { int a = 0, b = 0; rbl_get_sizes(ctx, &a, &b); return b; }
  py::ssize_t n3() const { return (py::ssize_t)3 * n_bod() * n_blb(); }

  // setParameters(a, dt, kBT, eta, cfg)   reference :183
  void setParameters(double a, double dt, double kBT, double eta, darr cfg)
  {
    if (cfg.size() % 3 != 0) throw std::runtime_error("Rigid config must have length 3N");
    check(rbl_set_parameters(ctx, a, dt, kBT, eta, cfg.data(), (int)(cfg.size() / 3)));
  }
  void setBlkPC(bool v) { check(rbl_set_blk_pc(ctx, v)); }      // :197
  void setWallPC(bool v) { check(rbl_set_wall_pc(ctx, v)); }    // :199

  void setConfig(darr X, darr Q)                                 // :201
  {
    if (X.size() % 3 != 0 || Q.size() != 4 * (X.size() / 3))
      throw std::runtime_error("setConfig: X must have length 3*N_bod and Q length 4*N_bod");
    check(rbl_set_config(ctx, X.data(), Q.data(), (int)(X.size() / 3)));
  }

  py::tuple getConfig()                                          // :235
  {
    const int nb = n_bod();
    darr X(3 * (py::ssize_t)nb), Q(4 * (py::ssize_t)nb);
    check(rbl_get_config(ctx, X.mutable_data(), Q.mutable_data()));
    return py::make_tuple(X, Q);
  }

  void set_K_mats() { check(rbl_set_K_mats(ctx)); }              // :395

  darr K_x_U(darr U)                                             // :404
  {
    if (U.size() != 6 * (py::ssize_t)n_bod()) throw std::runtime_error("K_x_U: U must have length 6*N_bod");
    darr out(n3());
    check(rbl_K_x_U(ctx, U.data(), out.mutable_data()));
    return out;
  }

  darr KT_x_Lam(darr lam)                                        // :410
  {
    if (lam.size() != n3()) throw std::runtime_error("KT_x_Lam: lambda must have length 3*N_blobs");
    darr out(6 * (py::ssize_t)n_bod());
    check(rbl_KT_x_Lam(ctx, lam.data(), out.mutable_data()));
    return out;
  }

  darr multi_body_pos()                                          // :295
  {
    darr out(n3());
    int rc;
    {
      py::gil_scoped_release rel;
      rc = rbl_multi_body_pos(ctx, out.mutable_data());
    }
    check(rc);
    return out;
  }

  darr apply_PC(darr in)                                         // :589
  {
    const py::ssize_t n = n3() + 6 * (py::ssize_t)n_bod();
    if (in.size() != n) throw std::runtime_error("apply_PC: input must have length 3*N_blobs + 6*N_bod");
    darr out(n);
    check(rbl_apply_PC(ctx, in.data(), out.mutable_data()));
    return out;
  }

  py::object csc(bool inverse)                                   // get_K :978 / get_Kinv :986
  {
    int64_t nnz = 0, nr = 0, nc = 0;
    auto fn = inverse ? rbl_get_Kinv_csc : rbl_get_K_csc;
    check(fn(ctx, &nnz, &nr, &nc, nullptr, nullptr, nullptr));
    py::array_t<double> data(nnz);
    py::array_t<int32_t> indices(nnz), indptr(nc + 1);
    check(fn(ctx, &nnz, &nr, &nc, data.mutable_data(), indices.mutable_data(), indptr.mutable_data()));
    py::object csc_matrix = py::module_::import("scipy.sparse").attr("csc_matrix");
    return csc_matrix(py::make_tuple(data, indices, indptr), py::arg("shape") = py::make_tuple(nr, nc));
  }
  py::object get_K() { return csc(false); }
  py::object get_Kinv() { return csc(true); }

  void evolve_X_Q(darr U)                                        // :865
  {
    if (U.size() != 6 * (py::ssize_t)n_bod()) throw std::runtime_error("evolve_X_Q: U must have length 6*N_bod");
    check(rbl_evolve_X_Q(ctx, U.data()));
  }

  darr apply_M(darr F, darr r_vecs)                              // :641
  {
    if (F.size() != r_vecs.size()) throw std::runtime_error("Positions and forces must be of the same size");
    darr out(F.size());
    int rc;
    {
      py::gil_scoped_release rel;
      rc = rbl_apply_M(ctx, F.data(), r_vecs.data(), (int64_t)F.size(), out.mutable_data());
    }
    check(rc);
    return out;
  }

  // ---- extensions (not in the reference's Python surface) -------------------
  darr apply_M_multi(darr F, darr r_vecs)  // F: (nrhs, n3) C-order = n3 x nrhs column-major
  {
    if (F.ndim() != 2 || F.shape(1) != r_vecs.size()) throw std::runtime_error("apply_M_multi: F must be (nrhs, 3N)");
    darr out({F.shape(0), F.shape(1)});
    int rc;
    {
      py::gil_scoped_release rel;
      rc = rbl_apply_M_multi(ctx, F.data(), r_vecs.data(), (int64_t)r_vecs.size(), (int)F.shape(0), out.mutable_data());
    }
    check(rc);
    return out;
  }

  darr M_half_W(py::object W, uint64_t seed, const std::string &method)   // :661 (unbound in the reference)
  {
    const int m = mhalf_method(method);
    darr out(n3());
    int rc;
    if (W.is_none()) {
      py::gil_scoped_release rel;
      rc = rbl_M_half_W(ctx, nullptr, seed, m, out.mutable_data());
    } else {
      darr Wa = W.cast<darr>();
      if (Wa.size() != n3()) throw std::runtime_error("M_half_W: W must have length 3*N_blobs");
      py::gil_scoped_release rel;
      rc = rbl_M_half_W(ctx, Wa.data(), seed, m, out.mutable_data());
    }
    check(rc);
    return out;
  }

  darr M_half_W_r(darr r_vecs, darr W, const std::string &method)
  {
    const int m = mhalf_method(method);
    if (W.size() != r_vecs.size()) throw std::runtime_error("M_half_W_r: W and r_vecs must have the same length");
    darr out(W.size());
    int rc;
    {
      py::gil_scoped_release rel;
      rc = rbl_M_half_W_r(ctx, r_vecs.data(), (int64_t)r_vecs.size(), W.data(), 0, m, out.mutable_data());
    }
    check(rc);
    return out;
  }

  darr M_RFD(py::object W, uint64_t seed, double delta)             // :769 (unbound in the reference)
  {
    darr out(n3());
    int rc;
    if (W.is_none()) {
      py::gil_scoped_release rel;
      rc = rbl_M_RFD(ctx, nullptr, seed, delta, out.mutable_data());
    } else {
      darr Wa = W.cast<darr>();
      if (Wa.size() != n3()) throw std::runtime_error("M_RFD: W must have length 3*N_blobs");
      py::gil_scoped_release rel;
      rc = rbl_M_RFD(ctx, Wa.data(), seed, delta, out.mutable_data());
    }
    check(rc);
    return out;
  }

  darr KTinv_RFD(darr W, double delta)                              // :743 (unbound in the reference)
  {
    if (W.size() != 6 * (py::ssize_t)n_bod()) throw std::runtime_error("KTinv_RFD: W must have length 6*N_bod");
    darr out(6 * (py::ssize_t)n_bod());
    check(rbl_KTinv_RFD(ctx, W.data(), delta, out.mutable_data()));
    return out;
  }

  py::tuple update_X_Q(darr U)                                      // :798 (unbound in the reference)
  {
    if (U.size() != 6 * (py::ssize_t)n_bod()) throw std::runtime_error("update_X_Q: U must have length 6*N_bod");
    darr X(3 * (py::ssize_t)n_bod()), Q(4 * (py::ssize_t)n_bod());
    check(rbl_update_X_Q(ctx, U.data(), X.mutable_data(), Q.mutable_data()));
    return py::make_tuple(X, Q);
  }

  // whole time steps inside librbl (no counterpart in the reference, which ships no driver) -> (iterations, residual)
  py::tuple step_deterministic(darr F, py::object slip, int max_iter, double rtol, int warm_start)
  {
    if (F.size() != 6 * (py::ssize_t)n_bod()) throw std::runtime_error("step_deterministic: F must have length 6*N_bod");
    darr sl;
    const double *sp = nullptr;
    if (!slip.is_none()) {
      sl = slip.cast<darr>();
      if (sl.size() != n3()) throw std::runtime_error("step_deterministic: slip must have length 3*N_blobs");
      sp = sl.data();
    }
    int it = 0, rc; double res = 0.0;
    {
      py::gil_scoped_release rel;
      rc = rbl_step_deterministic(ctx, F.data(), sp, max_iter, rtol, warm_start, &it, &res);
    }
    check(rc);
    return py::make_tuple(it, res);
  }

  py::tuple step_brownian(darr F, py::object slip, py::object W, uint64_t seed, const std::string &method, bool split_rand,
                          double delta, int max_iter, double rtol)
  {
    if (F.size() != 6 * (py::ssize_t)n_bod()) throw std::runtime_error("step_brownian: F must have length 6*N_bod");
    darr sl, Wa;
    const double *sp = nullptr, *wp = nullptr;
    if (!slip.is_none()) {
      sl = slip.cast<darr>();
      if (sl.size() != n3()) throw std::runtime_error("step_brownian: slip must have length 3*N_blobs");
      sp = sl.data();
    }
    if (!W.is_none()) {
      Wa = W.cast<darr>();
      if (Wa.size() != 3 * n3()) throw std::runtime_error("step_brownian: W must have length 9*N_blobs (W1|W2|W_rfd)");
      wp = Wa.data();
    }
    const int m = mhalf_method(method);
    int it = 0, rc; double res = 0.0;
    {
      py::gil_scoped_release rel;
      rc = rbl_step_brownian(ctx, F.data(), sp, wp, seed, m, split_rand ? 1 : 0, delta, max_iter, rtol, &it, &res);
    }
    check(rc);
    return py::make_tuple(it, res);
  }

  // RHS_and_Midpoint(Slip, Force) :917 (unbound in the reference) -> (RHS, X_half, Q_half)
  py::tuple RHS_and_Midpoint(darr Slip, darr Force, py::object W, uint64_t seed, const std::string &method,
                             bool split_rand, double delta)
  {
    const py::ssize_t nb = n_bod();
    if (Slip.size() != n3()) throw std::runtime_error("RHS_and_Midpoint: Slip must have length 3*N_blobs");
    if (Force.size() != 6 * nb) throw std::runtime_error("RHS_and_Midpoint: Force must have length 6*N_bod");
    const int m = mhalf_method(method);
    darr RHS(n3() + 6 * nb), X(3 * nb), Q(4 * nb);
    darr Wa;
    const double *Wp = nullptr;
    if (!W.is_none()) {
      Wa = W.cast<darr>();
      if (Wa.size() != 3 * n3()) throw std::runtime_error("RHS_and_Midpoint: W must have length 9*N_blobs (W1|W2|W_rfd)");
      Wp = Wa.data();
    }
    int rc;
    {
      py::gil_scoped_release rel;
      rc = rbl_RHS_and_Midpoint(ctx, Slip.data(), Force.data(), Wp, seed, m, split_rand ? 1 : 0, delta,
                                RHS.mutable_data(), X.mutable_data(), Q.mutable_data());
    }
    check(rc);
    return py::make_tuple(RHS, X, Q);
  }

  py::tuple lanczos_report()
  {
    int it = 0; double res = 0;
    rbl_get_lanczos_report(ctx, &it, &res);
    return py::make_tuple(it, res);
  }
  void set_lanczos(int max_iter, double tol) { check(rbl_set_lanczos(ctx, max_iter, tol)); }

  darr rotne_prager_tensor(darr r_vecs, bool scale_damp)       // :413 -> (n3, n3) column-major
  {
    const py::ssize_t n = r_vecs.size();
    py::array_t<double, py::array::f_style> out({n, n});
    int rc;
    {
      py::gil_scoped_release rel;
      rc = rbl_rotne_prager_tensor(ctx, r_vecs.data(), (int64_t)n, scale_damp, out.mutable_data());
    }
    check(rc);
    return out;
  }

  py::array cholesky_lower(py::array_t<double, py::array::f_style | py::array::forcecast> M)
  {
    if (M.ndim() != 2 || M.shape(0) != M.shape(1)) throw std::runtime_error("cholesky_lower: square matrix expected");
    py::array_t<double, py::array::f_style> A({M.shape(0), M.shape(1)});
    std::memcpy(A.mutable_data(), M.data(), sizeof(double) * (size_t)M.size());
    int rc;
    {
      py::gil_scoped_release rel;
      rc = rbl_cholesky_lower(ctx, A.mutable_data(), (int64_t)M.shape(0));
    }
    check(rc);
    return A;
  }

  darr pair_blocks(darr ri, darr rj, py::array_t<int32_t, py::array::c_style | py::array::forcecast> ii,
                   py::array_t<int32_t, py::array::c_style | py::array::forcecast> jj, bool wall, int mode)
  {
    const py::ssize_t n = ii.size();
    if (ri.size() != 3 * n || rj.size() != 3 * n || jj.size() != n) throw std::runtime_error("pair_blocks: size mismatch");
    darr out({n, (py::ssize_t)3, (py::ssize_t)3});
    check(rbl_debug_pair_blocks(ctx, ri.data(), rj.data(), ii.data(), jj.data(), (int64_t)n, wall, mode, out.mutable_data()));
    return out;
  }

  // [M lambda - K U ; K^T lambda], src/Rigid.py:73-80 in ONE boundary crossing
  darr apply_saddle(darr x)
  {
    const py::ssize_t n = n3() + 6 * (py::ssize_t)n_bod();
    if (x.size() != n) throw std::runtime_error("apply_saddle: input must have length 3*N_blobs + 6*N_bod");
    darr out(n);
    check(rbl_apply_saddle(ctx, x.data(), out.mutable_data()));
    return out;
  }

  // right-preconditioned GMRES on the saddle operator inside the library (host vectors) -> (x, iterations, residual estimate)
  py::tuple solve_saddle(darr rhs, int max_iter, double rtol, py::object x0)
  {
    const py::ssize_t n = n3() + 6 * (py::ssize_t)n_bod();
    if (rhs.size() != n) throw std::runtime_error("solve_saddle: rhs must have length 3*N_blobs + 6*N_bod");
    darr x(n);
    int use_x0 = 0;
    if (!x0.is_none()) {
      darr g = x0.cast<darr>();
      if (g.size() != n) throw std::runtime_error("solve_saddle: x0 must have length 3*N_blobs + 6*N_bod");
      std::memcpy(x.mutable_data(), g.data(), sizeof(double) * (size_t)n);
      use_x0 = 1;
    }
    int it = 0;
    double res = 0.0;
    check(rbl_gmres_saddle(ctx, rhs.data(), max_iter, rtol, x.mutable_data(), use_x0, &it, &res));
    return py::make_tuple(x, it, res);
  }

  // nrhs right-hand sides in lock step (rhs: (nrhs, 3 N_blobs + 6 N_bod)): products on the fp64 matrix cores, 16 at a time
  py::tuple solve_saddle_multi(darr rhs, int max_iter, double rtol)
  {
    const py::ssize_t n = n3() + 6 * (py::ssize_t)n_bod();
    if (rhs.ndim() != 2 || rhs.shape(1) != n || rhs.shape(0) < 1)
      throw std::runtime_error("solve_saddle_multi: rhs must have shape (nrhs, 3*N_blobs + 6*N_bod)");
    const int nrhs = (int)rhs.shape(0);
    darr x({(py::ssize_t)nrhs, n});
    py::array_t<int> its(nrhs);
    darr res(nrhs);
    check(rbl_gmres_saddle_multi(ctx, rhs.data(), nrhs, max_iter, rtol, x.mutable_data(), its.mutable_data(), res.mutable_data()));
    return py::make_tuple(x, its, res);
  }

  py::tuple M_RFD_cfgs(darr U, double delta)                        // :798 (unbound in the reference)
  {
    if (U.size() != 6 * (py::ssize_t)n_bod()) throw std::runtime_error("M_RFD_cfgs: U must have length 6*N_bod");
    darr rp(n3()), rm(n3());
    check(rbl_M_RFD_cfgs(ctx, U.data(), delta, rp.mutable_data(), rm.mutable_data()));
    return py::make_tuple(rp, rm);
  }

  darr M_RFD_from_U(darr U, darr W, double delta)                   // :820 (unbound in the reference)
  {
    if (U.size() != 6 * (py::ssize_t)n_bod()) throw std::runtime_error("M_RFD_from_U: U must have length 6*N_bod");
    if (W.size() != n3()) throw std::runtime_error("M_RFD_from_U: W must have length 3*N_blobs");
    darr out(n3());
    check(rbl_M_RFD_from_U(ctx, U.data(), W.data(), delta, out.mutable_data()));
    return out;
  }

  darr KT_RFD_from_U(darr U, darr W, double delta)                  // :844 (unbound in the reference)
  {
    if (U.size() != 6 * (py::ssize_t)n_bod()) throw std::runtime_error("KT_RFD_from_U: U must have length 6*N_bod");
    if (W.size() != n3()) throw std::runtime_error("KT_RFD_from_U: W must have length 3*N_blobs");
    darr out(6 * (py::ssize_t)n_bod());
    check(rbl_KT_RFD_from_U(ctx, U.data(), W.data(), delta, out.mutable_data()));
    return out;
  }

  void evolve_X_Q_RFD(darr U)                                       // :880 (unbound in the reference)
  {
    if (U.size() != 6 * (py::ssize_t)n_bod()) throw std::runtime_error("evolve_X_Q_RFD: U must have length 6*N_bod");
    check(rbl_evolve_X_Q_RFD(ctx, U.data()));
  }

  void set_option(const std::string &name, int64_t value)
  {
    const int key = rbl_option_key(name.c_str());
    if (!key) throw std::runtime_error("set_option: unknown option '" + name + "'");
    check(rbl_set_option(ctx, key, value));
  }
  int64_t get_option(const std::string &name) const
  {
    const int key = rbl_option_key(name.c_str());
    int64_t v = 0;
    if (!key || rbl_get_option(ctx, key, &v)) throw std::runtime_error("get_option: unknown option '" + name + "'");
    return v;
  }
  uintptr_t handle() const { return (uintptr_t)ctx; }
};

}  // namespace

PYBIND11_MODULE(c_rigid, m)
{
  m.doc() = "Rigid code (MI355X-native librbl behind the reference's CManyBodies surface)";
  py::class_<CManyBodies>(m, "CManyBodies")
      .def(py::init<>())
      .def("getConfig", &CManyBodies::getConfig, "get the X and Q vectors for the current position")
      .def("setParameters", &CManyBodies::setParameters, "Set parameters for the module")
      .def("setBlkPC", &CManyBodies::setBlkPC, "set PC type")
      .def("setWallPC", &CManyBodies::setWallPC, "use wall corrections")
      .def("set_K_mats", &CManyBodies::set_K_mats, "Set the K,K^T,K^-1 matrices for the module")
      .def("K_x_U", &CManyBodies::K_x_U, "Multiply K by U", py::arg("U"))
      .def("KT_x_Lam", &CManyBodies::KT_x_Lam, "Multiply K^T by lambda", py::arg("lambda"))
      .def("multi_body_pos", &CManyBodies::multi_body_pos, "Get the blob positions")
      .def("apply_PC", &CManyBodies::apply_PC, "apply for PC")
      .def("setConfig", &CManyBodies::setConfig, "Set the X and Q vectors for the current position",
           py::arg("X"), py::arg("Q"))
      .def("get_K", &CManyBodies::get_K, "get K")
      .def("get_Kinv", &CManyBodies::get_Kinv, "get Kinv")
      .def("evolve_X_Q", &CManyBodies::evolve_X_Q, "evolve rigid bodies", py::arg("U"))
      .def("apply_M", &CManyBodies::apply_M, "mobility matrix mult", py::arg("F"), py::arg("r_vecs"))
      // extensions
      .def("apply_M_multi", &CManyBodies::apply_M_multi, py::arg("F"), py::arg("r_vecs"))
      .def("M_half_W", &CManyBodies::M_half_W, py::arg("W") = py::none(), py::arg("seed") = 0,
           py::arg("method") = "cholesky")
      .def("M_half_W_r", &CManyBodies::M_half_W_r, py::arg("r_vecs"), py::arg("W"), py::arg("method") = "cholesky")
      .def("M_RFD", &CManyBodies::M_RFD, py::arg("W") = py::none(), py::arg("seed") = 0, py::arg("delta") = 1.0e-4)
      .def("KTinv_RFD", &CManyBodies::KTinv_RFD, py::arg("W"), py::arg("delta") = 1.0e-4)
      .def("update_X_Q", &CManyBodies::update_X_Q, py::arg("U"))
      .def("step_deterministic", &CManyBodies::step_deterministic, py::arg("F"), py::arg("slip") = py::none(),
           py::arg("max_iter") = 50, py::arg("rtol") = 1.0e-8, py::arg("warm_start") = 0)
      .def("step_brownian", &CManyBodies::step_brownian, py::arg("F"), py::arg("slip") = py::none(),
           py::arg("W") = py::none(), py::arg("seed") = 0, py::arg("method") = "lanczos_pc", py::arg("split_rand") = true,
           py::arg("delta") = 1.0e-4, py::arg("max_iter") = 50, py::arg("rtol") = 1.0e-8)
      .def("RHS_and_Midpoint", &CManyBodies::RHS_and_Midpoint, py::arg("Slip"), py::arg("Force"),
           py::arg("W") = py::none(), py::arg("seed") = 0, py::arg("method") = "cholesky",
           py::arg("split_rand") = true, py::arg("delta") = 1.0e-4)
      .def("lanczos_report", &CManyBodies::lanczos_report)
      .def("set_lanczos", &CManyBodies::set_lanczos, py::arg("max_iter"), py::arg("tol"))
      .def("rotne_prager_tensor", &CManyBodies::rotne_prager_tensor, py::arg("r_vecs"), py::arg("scale_damp") = false)
      .def("cholesky_lower", &CManyBodies::cholesky_lower, py::arg("M"))
      .def("pair_blocks", &CManyBodies::pair_blocks)
      .def("apply_saddle", &CManyBodies::apply_saddle, py::arg("x"))
      .def("solve_saddle", &CManyBodies::solve_saddle, py::arg("rhs"), py::arg("max_iter") = 100, py::arg("rtol") = 1.0e-8,
           py::arg("x0") = py::none())
      .def("solve_saddle_multi", &CManyBodies::solve_saddle_multi, py::arg("rhs"), py::arg("max_iter") = 100, py::arg("rtol") = 1.0e-8)
      .def("M_RFD_cfgs", &CManyBodies::M_RFD_cfgs, py::arg("U"), py::arg("delta") = 1.0e-4)
      .def("M_RFD_from_U", &CManyBodies::M_RFD_from_U, py::arg("U"), py::arg("W"), py::arg("delta") = 1.0e-3)
      .def("KT_RFD_from_U", &CManyBodies::KT_RFD_from_U, py::arg("U"), py::arg("W"), py::arg("delta") = 1.0e-3)
      .def("evolve_X_Q_RFD", &CManyBodies::evolve_X_Q_RFD, py::arg("U"))
      .def("set_option", &CManyBodies::set_option, py::arg("name"), py::arg("value"))
      .def("get_option", &CManyBodies::get_option, py::arg("name"))
      .def("handle", &CManyBodies::handle, "address of the underlying rbl_ctx (for the ctypes device API)")
      .def_property_readonly_static("precision", [](py::object) { return std::string(rbl_precision()); },
                                    "Compilation precision, a string holding either single or double.");
}
