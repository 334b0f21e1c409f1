// host_rccl_step.cpp -- a C++-only host (no Python, no PyTorch) driving Brownian time steps on N GPUs through librbl's
// C ABI, one process per GPU, RCCL inside the library.  This is the shape of host the reference itself is
// (src/c_rigid_obj.cpp:997-1027 is a C++ class; its Python layer is a thin wrapper): it shows that nothing above
// include/rbl.h is needed to run the multi-GPU hot path.
//
//   build:  hipcc -O2 -std=c++17 examples/host_rccl_step.cpp -Iinclude -Lrigid_body_light_amd -lrbl \
//                 -Wl,-rpath,$PWD/rigid_body_light_amd -o examples/host_rccl_step        (rigid_body_light_amd/build.py does it)
//   run:    for r in 0 .. N-1:  examples/host_rccl_step <rank> <N> <id_file> <shell_N_*.csv> <n_bodies> <steps> [split]
//           (split: 0 unordered tile pairs + all-reduce, 1 rows by body index + all-gather; one process per GPU)
//
// Rank 0 creates the RCCL unique id and publishes it through <id_file> (any transport would do: MPI_Bcast, a socket); the
// ranks hold the same replicated body state and call the same entry points.  Every rank also runs the SAME steps on a
// second, single-GPU context and prints the largest difference of the body positions: the sharded solve must reproduce
// the un-sharded one (bitwise at N = 1, to solver tolerance otherwise).
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "rbl.h"

static void die(rbl_ctx *c, const char *what, int rc)
{
  std::fprintf(stderr, "%s failed: status %d (%s)\n", what, rc, c ? rbl_last_error(c) : "");
  std::exit(1);
}
#define CHECK(c, call) do { int rc__ = (call); if (rc__ != RBL_OK) die(c, #call, rc__); } while (0)

// structure file of the reference's tests (tests/utils.py:9-19): "# sep,N,rg,rh", "# values", then N rows x y z
static bool read_structure(const std::string &path, double *sep, std::vector<double> &cfg)
{
  std::ifstream f(path);
  if (!f) return false;
  std::string line;
  std::getline(f, line);
  std::getline(f, line);
  *sep = std::atof(line.c_str() + line.find_first_not_of("# "));
  double x;
  while (f >> x) cfg.push_back(x);
  return !cfg.empty() && cfg.size() % 3 == 0;
}

int main(int argc, char **argv)
{
  if (argc < 7) {
    std::fprintf(stderr, "usage: %s rank world id_file structure.csv n_bodies steps [split]\n", argv[0]);
    return 2;
  }
  const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
  const std::string id_file = argv[3];
  const int nb = std::atoi(argv[5]), steps = std::atoi(argv[6]);
  const int split = argc > 7 ? std::atoi(argv[7]) : 0;
  double sep = 0.0;
  std::vector<double> cfg;
  if (!read_structure(argv[4], &sep, cfg)) { std::fprintf(stderr, "cannot read %s\n", argv[4]); return 2; }
  const int nblb = (int)(cfg.size() / 3);
  const double a = 0.5 * sep, eta = 1.0, dt = 0.01, kBT = 1.0;

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { std::fprintf(stderr, "no GPU\n"); return 3; }
  if (world > ndev) { std::fprintf(stderr, "%d ranks need %d GPUs, %d visible (RCCL wants one device per rank)\n", world, world, ndev); return 3; }
  if (hipSetDevice(rank) != hipSuccess) return 3;

  // bodies on a simple cubic lattice above a wall, orientations from a fixed linear congruential sequence
  std::vector<double> X((size_t)3 * nb), Q((size_t)4 * nb), F((size_t)6 * nb, 0.0);
  const int side = (int)std::ceil(std::cbrt((double)nb) - 1e-9);
  const double spacing = 2.0 * (1.0 + a) + 0.5;
  unsigned long long s = 88172645463325252ull;
  auto uni = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0 - 0.5; };
  for (int b = 0; b < nb; ++b) {
    X[3 * b] = (b % side) * spacing; X[3 * b + 1] = ((b / side) % side) * spacing; X[3 * b + 2] = (b / (side * side)) * spacing + 1.3 + a;
    double q[4], n2 = 0.0;
    for (double &v : q) { v = uni(); n2 += v * v; }
    for (int k = 0; k < 4; ++k) Q[4 * b + k] = q[k] / std::sqrt(n2);
    F[6 * b + 2] = -1.0;
  }

  rbl_ctx *ctx[2] = {rbl_create(), rbl_create()};       // [0] this rank of the N-GPU job, [1] the same steps on one GPU
  for (rbl_ctx *c : ctx) {
    if (!c) return 4;
    CHECK(c, rbl_set_parameters(c, a, dt, kBT, eta, cfg.data(), nblb));
    CHECK(c, rbl_set_wall_pc(c, 1));
    CHECK(c, rbl_set_blk_pc(c, 1));
    CHECK(c, rbl_set_config(c, X.data(), Q.data(), nb));
    CHECK(c, rbl_set_lanczos(c, 100, 1e-8));
  }

  unsigned char id[RBL_COMM_ID_BYTES];
  if (rank == 0) {
    CHECK(ctx[0], rbl_comm_unique_id(id));
    const std::string tmp = id_file + ".tmp";
    std::ofstream(tmp, std::ios::binary).write((const char *)id, sizeof(id));
    std::rename(tmp.c_str(), id_file.c_str());
  } else {
    for (int tries = 0;; ++tries) {
      std::ifstream f(id_file, std::ios::binary);
      if (f && f.read((char *)id, sizeof(id))) break;
      if (tries > 600) { std::fprintf(stderr, "rank %d: no unique id in %s\n", rank, id_file.c_str()); return 5; }
      usleep(100000);
    }
  }
  CHECK(ctx[0], rbl_comm_init_rccl(ctx[0], id, rank, world));
  CHECK(ctx[0], rbl_set_option(ctx[0], RBL_OPT_COMM_SPLIT, split));
  int r_ = -1, w_ = -1, kind = -1;
  CHECK(ctx[0], rbl_comm_info(ctx[0], &r_, &w_, &kind));
  if (rank == 0) std::printf("host_rccl_step: %d ranks, %d x %d blobs, communicator kind %d (2 = RCCL inside librbl), split %d\n", w_, nb, nblb, kind, split);

  CHECK(ctx[0], rbl_set_timing(ctx[0], 1));
  double worst = 0.0;
  std::vector<double> Xa((size_t)3 * nb), Xb((size_t)3 * nb), Qa((size_t)4 * nb), Qb((size_t)4 * nb);
  for (int k = 0; k < steps; ++k) {
    int it[2] = {0, 0};
    double res[2] = {0.0, 0.0};
    for (int v = 0; v < 2; ++v)
      CHECK(ctx[v], rbl_step_brownian(ctx[v], F.data(), nullptr, nullptr, 1000 + k, RBL_MHALF_LANCZOS_PC, 1, 1e-4, 100, 1e-9, &it[v], &res[v]));
    CHECK(ctx[0], rbl_get_config(ctx[0], Xa.data(), Qa.data()));
    CHECK(ctx[1], rbl_get_config(ctx[1], Xb.data(), Qb.data()));
    double d = 0.0;
    for (size_t i = 0; i < Xa.size(); ++i) d = std::fmax(d, std::fabs(Xa[i] - Xb[i]));
    worst = std::fmax(worst, d);
    if (rank == 0) std::printf("step %2d on %d rank(s): %d GMRES iterations (%.2e), single-GPU run %d (%.2e), max |dX| %.3e\n", k, world, it[0], res[0], it[1], res[1], d);
  }
  double ms[RBL_T_COUNT];
  int64_t calls[RBL_T_COUNT];
  CHECK(ctx[0], rbl_get_timings(ctx[0], ms, calls));
  if (rank == 0)
    std::printf("per step: products %.2f ms, per-body %.2f ms, factors %.2f ms, collectives %.2f ms in %lld calls; worst |dX| %.3e\n",
                ms[RBL_T_PRODUCT] / steps, ms[RBL_T_PERBODY] / steps, ms[RBL_T_FACTOR] / steps, ms[RBL_T_COLLECTIVE] / steps,
                (long long)(calls[RBL_T_COLLECTIVE] / (steps > 0 ? steps : 1)), worst);
  CHECK(ctx[0], rbl_comm_finalize(ctx[0]));
  rbl_destroy(ctx[0]);
  rbl_destroy(ctx[1]);
  if (rank == 0) std::remove(id_file.c_str());
  if (!(worst < 1e-6)) { std::fprintf(stderr, "sharded and single-GPU steps differ: %.3e\n", worst); return 6; }
  if (rank == 0) std::printf("HOST OK\n");
  return 0;
}
