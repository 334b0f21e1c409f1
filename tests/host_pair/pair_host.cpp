// Host build of the device pair arithmetic (rbl_pair.hpp) for CPU-side algebra checks -- see hip/hip_runtime.h here.
//   g++ -O2 -ffp-contract=fast -mfma -shared -fPIC -Itests/host_pair -o /tmp/libpair_host.so tests/host_pair/pair_host.cpp
#include "../../rigid_body_light_amd/csrc/rbl_pair.hpp"

static RblParams make_params(double a)
{
  RblParams P;
  P.a = a; P.inv_a = 1.0 / a; P.nf = 1.0; P.four_a2 = 4.0 * a * a; P.tiny2 = (1e-12 * a) * (1e-12 * a);
  P.c_near_A = -0.375 / a; P.c_near_B = 0.125 / a; P.no_damp = 0;
  return P;
}

extern "C" {
// ordered block (i <- j, h = z_j) by the fast accumulation form, unscaled, row-major
void fast_block(const double *ri, const double *rj, int i, int j, double a, int wall, double *out9)
{
  const RblParams P = make_params(a);
  unsigned flags = 0;
  for (int c = 0; c < 3; ++c) {
    double ux = 0, uy = 0, uz = 0;
    const double fx = c == 0, fy = c == 1, fz = c == 2;
    if (wall) rbl_pair_accum<true, true>(P, ri[0], ri[1], ri[2], rj[0], rj[1], rj[2], fx, fy, fz, i == j, ux, uy, uz, flags);
    else rbl_pair_accum<false, true>(P, ri[0], ri[1], ri[2], rj[0], rj[1], rj[2], fx, fy, fz, i == j, ux, uy, uz, flags);
    out9[c] = ux; out9[3 + c] = uy; out9[6 + c] = uz;
  }
}
// symmetric form in radius-scaled coordinates (what k_apply_M_sym runs): M_ij (from U_i += M F_j) and M_ji
// (from U_j += M^T F_i), both row-major, unscaled
void sym_blocks(const double *ri, const double *rj, double a, int wall, int nearchk, double *Mij, double *Mji)
{
  RblParams P = make_params(1.0);
  const double xi = ri[0] / a, yi = ri[1] / a, zi = ri[2] / a, xj = rj[0] / a, yj = rj[1] / a, zj = rj[2] / a;
  unsigned flags = 0;
  for (int c = 0; c < 3; ++c) {
    const double fx = c == 0, fy = c == 1, fz = c == 2;
    double ui[3] = {0, 0, 0}, uj[3] = {0, 0, 0};
    if (wall) {
      if (nearchk) rbl_pair_sym<true, true, true>(P, xi, yi, zi, fx, fy, fz, xj, yj, zj, fx, fy, fz, ui[0], ui[1], ui[2], uj[0], uj[1], uj[2], flags);
      else rbl_pair_sym<true, true, false>(P, xi, yi, zi, fx, fy, fz, xj, yj, zj, fx, fy, fz, ui[0], ui[1], ui[2], uj[0], uj[1], uj[2], flags);
    } else {
      if (nearchk) rbl_pair_sym<false, true, true>(P, xi, yi, zi, fx, fy, fz, xj, yj, zj, fx, fy, fz, ui[0], ui[1], ui[2], uj[0], uj[1], uj[2], flags);
      else rbl_pair_sym<false, true, false>(P, xi, yi, zi, fx, fy, fz, xj, yj, zj, fx, fy, fz, ui[0], ui[1], ui[2], uj[0], uj[1], uj[2], flags);
    }
    for (int p = 0; p < 3; ++p) { Mij[3 * p + c] = ui[p]; Mji[3 * p + c] = uj[p]; }
  }
}
}
