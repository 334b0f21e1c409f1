/*
 * rbl_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the blob-level RPY mobility hot path of
 * brennansprinkle/Rigid_Body_Light (reference src/c_rigid_obj.cpp).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (rigid_body_light_amd/) never does.
 *
 * Parity pin: the two pair kernels are checked against the reference's own
 * mobilityUFRPY / mobilityUFSingleWallCorrection compiled from
 * /root/reference/src/c_rigid_obj.cpp:31-142 (oracle/build_ref.sh ->
 * oracle/_ref/libref_pair.so) and against tests/golden/pair_kernels.json
 * generated from that build.  Everything above the pair kernels (assembly,
 * damping, matvec, Cholesky) has no golden vector in the reference's tests
 *  and is pinned by analytic known answers + numpy.
 *
 * All entry points are double precision; all matrices are column-major like
 * Eigen's default (reference src/eigen_defines.h:28-37).
 *
 * Return codes: 0 ok, 1 two blobs closer than 1e-12 a (the reference calls
 * exit(), c_rigid_obj.cpp:53-58), 2 blob centre below the wall (the reference
 * throws std::runtime_error, c_rigid_obj.cpp:95-97), 3 matrix not SPD.
 */
#ifndef RBL_ORACLE_H
#define RBL_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_OK 0
#define ORC_ERR_OVERLAP 1
#define ORC_ERR_BELOW_WALL 2
#define ORC_ERR_NOT_SPD 3

/* c_rigid_obj.cpp:31-83.  out6 = {Mxx,Mxy,Mxz,Myy,Myz,Mzz}, units 1/(8 pi eta a). */
int orc_mobilityUFRPY(double rx, double ry, double rz, double *out6, int i, int j,
                      double inv_a);

/* c_rigid_obj.cpp:85-142.  M9 (row-major xx,xy,xz,yx,..,zz) is updated in place. */
int orc_mobilityUFSingleWallCorrection(double rx, double ry, double rz, double *M9,
                                       int i, int j, double hj);

/* One 3x3 block exactly as rotne_prager_tensor forms it for i<=j
 * (c_rigid_obj.cpp:432-447), already scaled by 1/(8 pi eta a) (:456).
 * blk9 row-major.  ri, rj are the two blob positions. */
int orc_pair_block(const double *ri, const double *rj, int i, int j, double a,
                   double eta, int wall, double *blk9);

/* c_rigid_obj.cpp:413-459.  r: n3 = 3*Nblobs interleaved xyz.  Mob: n3*n3
 * column-major, fully written (both triangles). */
int orc_rotne_prager_tensor(const double *r, long n3, double a, double eta, int wall,
                            double *Mob);

/* c_rigid_obj.cpp:618-639.  B: n3 diagonal entries. */
void orc_make_damp(const double *r, long n3, double a, double *B);

/* c_rigid_obj.cpp:641-659, literal: dense build then GEMV. */
int orc_apply_M_dense(const double *F, const double *r, long n3, double a, double eta,
                      int wall, double *U);

/* Same block arithmetic as the dense build (i<=j roles, mirrored transpose,
 * scaled entries) but never stores the matrix: U_i += blk F_j, U_j += blk^T F_i.
 * Used at sizes where 8 n3^2 bytes cannot exist (cfg 3). */
int orc_apply_M_matfree(const double *F, const double *r, long n3, double a, double eta,
                        int wall, double *U);

/* Rows [row_begin,row_end) (blob indices) of apply_M, matrix-free, each ordered
 * pair evaluated with the reference's (min,max) roles and transposed when
 * i>j; OpenMP over rows when nthreads>1.  U holds 3*(row_end-row_begin). */
int orc_apply_M_rows(const double *F, const double *r, long n3, long row_begin,
                     long row_end, double a, double eta, int wall, int nthreads,
                     double *U);

/* In-place lower Cholesky of a column-major n x n SPD matrix (what Eigen::LLT
 * computes, c_rigid_obj.cpp:670-671).  The strict upper triangle is zeroed. */
int orc_cholesky_lower(double *M, long n);

/* c_rigid_obj.cpp:661-675 with the noise W injected instead of clock-seeded:
 * Mob = rotne_prager_tensor(r); B = damp(r) (ALWAYS applied); M = B Mob B;
 * L = chol(M); out = L W.  If Lout != NULL the factor is copied there. */
int orc_M_half_W(const double *r, long n3, double a, double eta, int wall,
                 const double *W, double *out, double *Lout);

/* c_rigid_obj.cpp:201-233,257-300.  X[3Nb], Q[4Nb] scalar-first (normalised
 * here, as setConfig does), ref_cfg[N_blb*3] row-major, already mean-removed.
 * out[3*Nb*N_blb] body-major interleaved xyz. */
void orc_multi_body_pos(const double *X, const double *Q, const double *ref_cfg,
                        int N_bod, int N_blb, double *out);

#ifdef __cplusplus
}
#endif
#endif
