/*
 * rbl_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 * See rbl_oracle.h for scope, the parity pin and who may load this.
 *
 * Compiled with -ffp-contract=off so every expression rounds exactly like the
 * reference's scalar C++ (no FMA contraction); the operation ORDER inside the
 * two pair kernels therefore follows the reference's expressions term by term.
 */
#include "rbl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- a1: free-space RPY pair block, c_rigid_obj.cpp:31-83 ------------------ */
int orc_mobilityUFRPY(double rx, double ry, double rz, double *o, int i, int j,
                      double inv_a)
{
  const double four3 = 4.0 / 3.0;
  if (i == j) { /* :40-46  self term decided by index equality */
    o[0] = four3; o[1] = 0.0; o[2] = 0.0;
    o[3] = four3; o[4] = 0.0; o[5] = four3;
    return ORC_OK;
  }
  rx = rx * inv_a; ry = ry * inv_a; rz = rz * inv_a;         /* :48-50 */
  const double r2 = rx * rx + ry * ry + rz * rz;              /* :51 */
  const double r = sqrt(r2);                                  /* :52 */
  if (r < 1e-12) return ORC_ERR_OVERLAP;                      /* :53-58 exit() */
  const double invr = 1.0 / r;
  const double invr2 = invr * invr;
  if (r >= 2.0) {                                             /* :62-70 */
    const double c1 = 1.0 + 2.0 / (3.0 * r2);
    const double c2 = (1.0 - 2.0 * invr2) * invr2;
    o[0] = (c1 + c2 * rx * rx) * invr;
    o[1] = (c2 * rx * ry) * invr;
    o[2] = (c2 * rx * rz) * invr;
    o[3] = (c1 + c2 * ry * ry) * invr;
    o[4] = (c2 * ry * rz) * invr;
    o[5] = (c1 + c2 * rz * rz) * invr;
  } else {                                                    /* :71-79 overlap */
    const double c1 = four3 * (1.0 - 0.28125 * r);
    const double c2 = four3 * 0.09375 * invr;
    o[0] = c1 + c2 * rx * rx;
    o[1] = c2 * rx * ry;
    o[2] = c2 * rx * rz;
    o[3] = c1 + c2 * ry * ry;
    o[4] = c2 * ry * rz;
    o[5] = c1 + c2 * rz * rz;
  }
  return ORC_OK;
}

/* ---- a2: Swan-Brady single-wall correction, c_rigid_obj.cpp:85-142 --------- */
int orc_mobilityUFSingleWallCorrection(double rx, double ry, double rz, double *M,
                                       int i, int j, double hj)
{
  if (hj < 0.0) return ORC_ERR_BELOW_WALL;                    /* :95-97 throw */
  if (i == j) {                                               /* :98-104 */
    const double iz = 1.0 / hj;
    const double iz3 = iz * iz * iz;
    const double iz5 = iz3 * iz * iz;
    M[0] += -(9 * iz - 2 * iz3 + iz5) / 12.0;
    M[4] += -(9 * iz - 2 * iz3 + iz5) / 12.0;
    M[8] += -(9 * iz - 4 * iz3 + iz5) / 6.0;
    return ORC_OK;
  }
  const double hh = hj / rz;                                  /* :106 h_hat */
  const double iR = 1.0 / sqrt(rx * rx + ry * ry + rz * rz);  /* :107 */
  const double ex = rx * iR, ey = ry * iR, ez = rz * iR;
  const double iR3 = iR * iR * iR;
  const double iR5 = iR3 * iR * iR;

  const double f1 = -(3 * (1 + 2 * hh * (1 - hh) * ez * ez) * iR +
                      2 * (1 - 3 * ez * ez) * iR3 - 2 * (1 - 5 * ez * ez) * iR5) / 3.0;
  const double f2 = -(3 * (1 - 6 * hh * (1 - hh) * ez * ez) * iR -
                      6 * (1 - 5 * ez * ez) * iR3 + 10 * (1 - 7 * ez * ez) * iR5) / 3.0;
  const double f3 = ez *
                    (3 * hh * (1 - 6 * (1 - hh) * ez * ez) * iR -
                     6 * (1 - 5 * ez * ez) * iR3 + 10 * (2 - 7 * ez * ez) * iR5) *
                    2.0 / 3.0;
  const double f4 = ez * (3 * hh * iR - 10 * iR5) * 2.0 / 3.0;
  const double f5 = -(3 * hh * hh * ez * ez * iR + 3 * ez * ez * iR3 +
                      (2 - 15 * ez * ez) * iR5) * 4.0 / 3.0;

  M[0] += f1 + f2 * ex * ex;                                  /* :132-140 */
  M[1] += f2 * ex * ey;
  M[2] += f2 * ex * ez + f3 * ex;
  M[3] += f2 * ey * ex;
  M[4] += f1 + f2 * ey * ey;
  M[5] += f2 * ey * ez + f3 * ey;
  M[6] += f2 * ez * ex + f4 * ex;
  M[7] += f2 * ez * ey + f4 * ey;
  M[8] += f1 + f2 * ez * ez + f3 * ez + f4 * ez + f5;
  return ORC_OK;
}

/* One (i<=j) block as the assembly loop forms it: c_rigid_obj.cpp:432-447,456 */
static inline int pair_block_unscaled(const double *ri, const double *rj, int i, int j,
                                      double a, double inv_a, int wall, double *b)
{
  const double rx = ri[0] - rj[0], ry = ri[1] - rj[1], rz = ri[2] - rj[2];
  double s[6];
  int rc = orc_mobilityUFRPY(rx, ry, rz, s, i, j, inv_a);
  if (rc) return rc;
  b[0] = s[0]; b[1] = s[1]; b[2] = s[2];
  b[3] = s[1]; b[4] = s[3]; b[5] = s[4];
  b[6] = s[2]; b[7] = s[4]; b[8] = s[5];
  if (wall) { /* :440-445  image vector, h = z_j / a */
    rc = orc_mobilityUFSingleWallCorrection(rx / a, ry / a, (rz + 2 * rj[2]) / a, b, i,
                                            j, rj[2] / a);
    if (rc) return rc;
  }
  return ORC_OK;
}

int orc_pair_block(const double *ri, const double *rj, int i, int j, double a,
                   double eta, int wall, double *blk9)
{
  const double nf = 1.0 / (8.0 * M_PI * eta * a);             /* :415 */
  int rc = pair_block_unscaled(ri, rj, i, j, a, 1.0 / a, wall, blk9);
  if (rc) return rc;
  for (int k = 0; k < 9; ++k) blk9[k] *= nf;                  /* :456 */
  return ORC_OK;
}

/* ---- a3: dense assembly, c_rigid_obj.cpp:413-459 --------------------------- */
int orc_rotne_prager_tensor(const double *r, long n3, double a, double eta, int wall,
                            double *Mob)
{
  const double nf = 1.0 / (8.0 * M_PI * eta * a);
  const long np = n3 / 3;
  const double inv_a = 1.0 / a;
  double b[9];
  for (long i = 0; i < np; ++i) {
    for (long j = i; j < np; ++j) {
      int rc = pair_block_unscaled(r + 3 * i, r + 3 * j, (int)i, (int)j, a, inv_a, wall, b);
      if (rc) return rc;
      for (int p = 0; p < 3; ++p)
        for (int q = 0; q < 3; ++q) {
          Mob[(3 * j + q) * n3 + (3 * i + p)] = b[3 * p + q];      /* (3i+p,3j+q) :449 */
          if (j != i) Mob[(3 * i + p) * n3 + (3 * j + q)] = b[3 * p + q]; /* transpose :451 */
        }
    }
  }
  for (long k = 0; k < n3 * n3; ++k) Mob[k] *= nf;                 /* :456 */
  return ORC_OK;
}

/* ---- a4: wall damping diagonal, c_rigid_obj.cpp:618-639 -------------------- */
void orc_make_damp(const double *r, long n3, double a, double *B)
{
  const long np = n3 / 3;
  for (long i = 0; i < np; ++i) {
    const double z = r[3 * i + 2];
    const double d = (z >= a) ? 1.0 : z / a;
    B[3 * i] = d; B[3 * i + 1] = d; B[3 * i + 2] = d;
  }
}

/* ---- a5: apply_M, literal dense form, c_rigid_obj.cpp:641-659 -------------- */
int orc_apply_M_dense(const double *F, const double *r, long n3, double a, double eta,
                      int wall, double *U)
{
  double *M = (double *)malloc(sizeof(double) * (size_t)n3 * (size_t)n3);
  double *x = (double *)malloc(sizeof(double) * (size_t)n3);
  double *B = (double *)malloc(sizeof(double) * (size_t)n3);
  if (!M || !x || !B) { free(M); free(x); free(B); return -1; }
  int rc = orc_rotne_prager_tensor(r, n3, a, eta, wall, M);
  if (rc) { free(M); free(x); free(B); return rc; }
  if (wall) { /* U = B (M (B F)): the one flag switches wall term AND damping */
    orc_make_damp(r, n3, a, B);
    for (long k = 0; k < n3; ++k) x[k] = B[k] * F[k];
  } else {
    for (long k = 0; k < n3; ++k) { B[k] = 1.0; x[k] = F[k]; }
  }
  memset(U, 0, sizeof(double) * (size_t)n3);
  for (long c = 0; c < n3; ++c) { /* column-major GEMV as column axpys */
    const double xc = x[c];
    const double *col = M + (size_t)c * (size_t)n3;
    for (long k = 0; k < n3; ++k) U[k] += col[k] * xc;
  }
  if (wall) for (long k = 0; k < n3; ++k) U[k] = B[k] * U[k];
  free(M); free(x); free(B);
  return ORC_OK;
}

static inline void blk_apply(const double *b, const double *f, double *u)
{
  u[0] += b[0] * f[0] + b[1] * f[1] + b[2] * f[2];
  u[1] += b[3] * f[0] + b[4] * f[1] + b[5] * f[2];
  u[2] += b[6] * f[0] + b[7] * f[1] + b[8] * f[2];
}
static inline void blk_applyT(const double *b, const double *f, double *u)
{
  u[0] += b[0] * f[0] + b[3] * f[1] + b[6] * f[2];
  u[1] += b[1] * f[0] + b[4] * f[1] + b[7] * f[2];
  u[2] += b[2] * f[0] + b[5] * f[1] + b[8] * f[2];
}

int orc_apply_M_matfree(const double *F, const double *r, long n3, double a, double eta,
                        int wall, double *U)
{
  const double nf = 1.0 / (8.0 * M_PI * eta * a);
  const long np = n3 / 3;
  const double inv_a = 1.0 / a;
  double *x = (double *)malloc(sizeof(double) * (size_t)n3);
  double *B = (double *)malloc(sizeof(double) * (size_t)n3);
  if (!x || !B) { free(x); free(B); return -1; }
  if (wall) {
    orc_make_damp(r, n3, a, B);
    for (long k = 0; k < n3; ++k) x[k] = B[k] * F[k];
  } else {
    for (long k = 0; k < n3; ++k) { B[k] = 1.0; x[k] = F[k]; }
  }
  memset(U, 0, sizeof(double) * (size_t)n3);
  double b[9];
  for (long i = 0; i < np; ++i) {
    for (long j = i; j < np; ++j) {
      int rc = pair_block_unscaled(r + 3 * i, r + 3 * j, (int)i, (int)j, a, inv_a, wall, b);
      if (rc) { free(x); free(B); return rc; }
      for (int k = 0; k < 9; ++k) b[k] *= nf;
      blk_apply(b, x + 3 * j, U + 3 * i);
      if (j != i) blk_applyT(b, x + 3 * i, U + 3 * j);
    }
  }
  if (wall) for (long k = 0; k < n3; ++k) U[k] = B[k] * U[k];
  free(x); free(B);
  return ORC_OK;
}

int orc_apply_M_rows(const double *F, const double *r, long n3, long row_begin,
                     long row_end, double a, double eta, int wall, int nthreads,
                     double *U)
{
  const double nf = 1.0 / (8.0 * M_PI * eta * a);
  const long np = n3 / 3;
  const double inv_a = 1.0 / a;
  int err = 0;
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 8) num_threads(nthreads > 0 ? nthreads : 1) reduction(max : err)
#endif
  for (long i = row_begin; i < row_end; ++i) {
    double u[3] = {0.0, 0.0, 0.0};
    double b[9], f[3];
    for (long j = 0; j < np; ++j) {
      const double dj = (wall && r[3 * j + 2] < a) ? r[3 * j + 2] / a : 1.0;
      f[0] = dj * F[3 * j]; f[1] = dj * F[3 * j + 1]; f[2] = dj * F[3 * j + 2];
      int rc;
      if (i <= j) {
        rc = pair_block_unscaled(r + 3 * i, r + 3 * j, (int)i, (int)j, a, inv_a, wall, b);
        for (int k = 0; k < 9; ++k) b[k] *= nf;
        blk_apply(b, f, u);
      } else { /* block (i,j) = transpose of the stored block (j,i) */
        rc = pair_block_unscaled(r + 3 * j, r + 3 * i, (int)j, (int)i, a, inv_a, wall, b);
        for (int k = 0; k < 9; ++k) b[k] *= nf;
        blk_applyT(b, f, u);
      }
      if (rc > err) err = rc;
    }
    const double di = (wall && r[3 * i + 2] < a) ? r[3 * i + 2] / a : 1.0;
    U[3 * (i - row_begin)] = di * u[0];
    U[3 * (i - row_begin) + 1] = di * u[1];
    U[3 * (i - row_begin) + 2] = di * u[2];
  }
  return err;
}

/* ---- lower Cholesky (what Eigen::LLT yields, :670-671), column-major ------- */
int orc_cholesky_lower(double *M, long n)
{
  /* left-looking, column by column: unit-stride inner loops on column-major */
  for (long j = 0; j < n; ++j) {
    double *cj = M + (size_t)j * (size_t)n;
    for (long k = 0; k < j; ++k) {
      const double *ck = M + (size_t)k * (size_t)n;
      const double ljk = ck[j];
      if (ljk != 0.0)
        for (long i = j; i < n; ++i) cj[i] -= ck[i] * ljk;
    }
    const double d = cj[j];
    if (!(d > 0.0)) return ORC_ERR_NOT_SPD;
    const double s = sqrt(d);
    cj[j] = s;
    for (long i = j + 1; i < n; ++i) cj[i] /= s;
    for (long i = 0; i < j; ++i) cj[i] = 0.0;
  }
  return ORC_OK;
}

/* ---- a6: M_half_W with injected noise, c_rigid_obj.cpp:661-675 ------------- */
int orc_M_half_W(const double *r, long n3, double a, double eta, int wall,
                 const double *W, double *out, double *Lout)
{
  double *M = (double *)malloc(sizeof(double) * (size_t)n3 * (size_t)n3);
  double *B = (double *)malloc(sizeof(double) * (size_t)n3);
  if (!M || !B) { free(M); free(B); return -1; }
  int rc = orc_rotne_prager_tensor(r, n3, a, eta, wall, M);         /* :667 */
  if (rc) { free(M); free(B); return rc; }
  orc_make_damp(r, n3, a, B);                                       /* :668 always */
  for (long c = 0; c < n3; ++c)                                     /* :669 B*Mob*B */
    for (long k = 0; k < n3; ++k) M[(size_t)c * n3 + k] = (B[k] * M[(size_t)c * n3 + k]) * B[c];
  rc = orc_cholesky_lower(M, n3);                                   /* :670-671 */
  if (rc) { free(M); free(B); return rc; }
  memset(out, 0, sizeof(double) * (size_t)n3);
  for (long c = 0; c < n3; ++c) {                                   /* :672 L*W */
    const double w = W[c];
    const double *col = M + (size_t)c * (size_t)n3;
    for (long k = c; k < n3; ++k) out[k] += col[k] * w;
  }
  if (Lout) memcpy(Lout, M, sizeof(double) * (size_t)n3 * (size_t)n3);
  free(M); free(B);
  return ORC_OK;
}

/* ---- a8: blob kinematics, c_rigid_obj.cpp:201-233 + 257-300 ---------------- */
void orc_multi_body_pos(const double *X, const double *Q, const double *ref_cfg,
                        int N_bod, int N_blb, double *out)
{
  for (int b = 0; b < N_bod; ++b) {
    /* setConfig: scalar-first in, normalise (:212-216) */
    double w = Q[4 * b], x = Q[4 * b + 1], y = Q[4 * b + 2], z = Q[4 * b + 3];
    const double nrm = sqrt(w * w + x * x + y * y + z * z);
    w /= nrm; x /= nrm; y /= nrm; z /= nrm;
    /* unit quaternion -> rotation matrix (Eigen's toRotationMatrix, :258) */
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    const double R[9] = {1 - (tyy + tzz), txy - twz, txz + twy,
                         txy + twz, 1 - (txx + tzz), tyz - twx,
                         txz - twy, tyz + twx, 1 - (txx + tyy)};
    for (int k = 0; k < N_blb; ++k) { /* r = ref_cfg * R^T + X  (:259-263) */
      const double *c = ref_cfg + 3 * k;
      double *o = out + 3 * ((long)b * N_blb + k);
      o[0] = c[0] * R[0] + c[1] * R[1] + c[2] * R[2] + X[3 * b];
      o[1] = c[0] * R[3] + c[1] * R[4] + c[2] * R[5] + X[3 * b + 1];
      o[2] = c[0] * R[6] + c[1] * R[7] + c[2] * R[8] + X[3 * b + 2];
    }
  }
}
