/*
 * rbl.h -- C ABI of the MI355X-native blob-mobility hot path (librbl.so).
 *
 * Drop-in boundary for brennansprinkle/Rigid_Body_Light: every entry point in
 * section 1 replaces ONE method the reference binds on `c_rigid.CManyBodies`
 * (reference src/c_rigid_obj.cpp:997-1027, nanobind).  The reference-side
 * binding a maintainer would write is in INTEGRATION.md; our own pybind11
 * shim (rigid_body_light_amd/csrc/c_rigid.cpp) calls nothing but this header.
 *
 * Conventions
 *   - plain pointers + sizes, double precision only, no C++/torch types;
 *   - host-pointer entry points (section 1,2) take caller-owned contiguous
 *     arrays, never modify inputs, and are synchronous;
 *   - device-pointer entry points (section 3) take HIP device pointers, enqueue
 *     on the context's stream (rbl_set_stream) and do NOT synchronise;
 *   - every function returns an int status (0 = RBL_OK) and never throws or
 *     exit()s; rbl_last_error(ctx) gives the message for the last failure;
 *   - vectors over blobs are xyz-interleaved, body-major (reference
 *     c_rigid_obj.cpp:281-293); quaternions are scalar-first (:212-215);
 *   - a context is not thread-safe (neither is the reference object, :151-154).
 */
#ifndef RBL_H
#define RBL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RBL_OK 0
#define RBL_ERR_OVERLAP 1     /* two blobs closer than 1e-12 a; reference exit()s, :53-58 */
#define RBL_ERR_BELOW_WALL 2  /* blob centre below the wall; reference throws, :95-97    */
#define RBL_ERR_NOT_SPD 3     /* Cholesky met a non-positive pivot                        */
#define RBL_ERR_SIZE 4        /* bad length / shape                                       */
#define RBL_ERR_NO_DEVICE 5   /* no HIP device / HIP runtime failure at init              */
#define RBL_ERR_HIP 6         /* a HIP call failed                                        */
#define RBL_ERR_STATE 7       /* parameters or configuration not set                      */
#define RBL_ERR_SINGULAR 8    /* K^T K singular (reference exit()s, :313-316)             */
#define RBL_ERR_ALLOC 9
#define RBL_ERR_NONFINITE 10  /* result contains inf/nan                                  */
#define RBL_ERR_ARG 11
#define RBL_ERR_COMM 12       /* a collective failed (RCCL error, librccl not loadable, callback returned non-zero) */

typedef struct rbl_ctx rbl_ctx;

/* ===================================================================== */
/* 1. One entry point per method bound in the reference (file:line)       */
/* ===================================================================== */

/* CManyBodies()  -- nb::init<>(), c_rigid_obj.cpp:1001.  Host-only; the HIP
 * device is initialised lazily by the first compute call. */
rbl_ctx *rbl_create(void);
void rbl_destroy(rbl_ctx *ctx);

/* static `precision`, c_rigid_obj.cpp:170-174,1024-1026.  Always "double". */
const char *rbl_precision(void);

/* setParameters(a, dt, kBT, eta, cfg), c_rigid_obj.cpp:183-195.
 * cfg: N_blb x 3 row-major; its mean is removed in a private copy (:188). */
int rbl_set_parameters(rbl_ctx *ctx, double a, double dt, double kBT, double eta,
                       const double *cfg, int N_blb);

/* setBlkPC(bool) :197, setWallPC(bool) :199 */
int rbl_set_blk_pc(rbl_ctx *ctx, int block_diag_pc);
int rbl_set_wall_pc(rbl_ctx *ctx, int wall);

/* setConfig(X[3Nb], Q[4Nb]), c_rigid_obj.cpp:201-233 (Q normalised). */
int rbl_set_config(rbl_ctx *ctx, const double *X, const double *Q, int N_bod);

/* getConfig() -> (X[3Nb], Q[4Nb]), c_rigid_obj.cpp:235-255 */
int rbl_get_config(const rbl_ctx *ctx, double *X, double *Q);

/* set_K_mats(), c_rigid_obj.cpp:395-402 */
int rbl_set_K_mats(rbl_ctx *ctx);

/* K_x_U(U[6Nb]) -> [3N], :404 ; KT_x_Lam(lambda[3N]) -> [6Nb], :410 */
int rbl_K_x_U(rbl_ctx *ctx, const double *U, double *out);
int rbl_KT_x_Lam(rbl_ctx *ctx, const double *lambda, double *out);

/* multi_body_pos() -> [3N], c_rigid_obj.cpp:295-300 (computed on the GPU) */
int rbl_multi_body_pos(rbl_ctx *ctx, double *out);

/* apply_PC(IN[3N+6Nb]) -> [3N+6Nb], c_rigid_obj.cpp:589-616.  Diagonal PC (diag_invM :489): host
 * arithmetic, O(N).  Block-diagonal PC (Block_diag_invM :461): GPU -- batched per-body mobility,
 * batched Cholesky on the matrix cores, substitution instead of the reference's explicit inverse. */
int rbl_apply_PC(rbl_ctx *ctx, const double *in, double *out);

/* get_K() / get_Kinv(), c_rigid_obj.cpp:978-992: CSC arrays.  Call once with
 * NULL arrays to get nnz, then with arrays of that size (indptr: ncols+1). */
int rbl_get_K_csc(rbl_ctx *ctx, int64_t *nnz, int64_t *nrows, int64_t *ncols,
                  double *data, int32_t *indices, int32_t *indptr);
int rbl_get_Kinv_csc(rbl_ctx *ctx, int64_t *nnz, int64_t *nrows, int64_t *ncols,
                     double *data, int32_t *indices, int32_t *indptr);

/* evolve_X_Q(U[6Nb]), c_rigid_obj.cpp:865-878 (multiplies by dt internally,
 * rebuilds K, invalidates the preconditioner).  U is not modified. */
int rbl_evolve_X_Q(rbl_ctx *ctx, const double *U);

/* apply_M(F[n3], r_vecs[n3]) -> [n3], c_rigid_obj.cpp:641-659.
 * n3 = 3 * (number of blobs in r_vecs); it need not equal 3*N_bod*N_blb
 * (reference tests/test_interface.py:171-177).  U = M F, or B (M (B F)) with
 * the wall term when wall_PC is set.  Matrix-free on the GPU. */
int rbl_apply_M(rbl_ctx *ctx, const double *F, const double *r_vecs, int64_t n3,
                double *out);

/* ===================================================================== */
/* 2. Reference C++ members that are NOT bound to Python, + extensions    */
/* ===================================================================== */

/* Kinv_x_V :406, KTinv_x_F :408 */
int rbl_Kinv_x_V(rbl_ctx *ctx, const double *V, double *out);
int rbl_KTinv_x_F(rbl_ctx *ctx, const double *F, double *out);

/* apply_M for nrhs right-hand sides, F/out column-major n3 x nrhs. */
int rbl_apply_M_multi(rbl_ctx *ctx, const double *F, const double *r_vecs, int64_t n3,
                      int nrhs, double *out);

/* rotne_prager_tensor(r) -> dense column-major n3 x n3, c_rigid_obj.cpp:413-459
 * (wall term per wall_PC).  scale_damp != 0 returns B Mob B (:668-669). */
int rbl_rotne_prager_tensor(rbl_ctx *ctx, const double *r_vecs, int64_t n3,
                            int scale_damp, double *out);

/* M_half_W(), c_rigid_obj.cpp:661-675, on the object's own configuration.
 * W == NULL: standard normal noise from `seed` (counter-based, reproducible;
 * the reference seeds from the clock, :731).  method: RBL_MHALF_CHOLESKY is the
 * reference algorithm (dense B Mob B, lower Cholesky, L W);
 * RBL_MHALF_LANCZOS is matrix-free (a different square root of the same M). */
#define RBL_MHALF_CHOLESKY 0
#define RBL_MHALF_LANCZOS 1
/* Lanczos on the block-Jacobi preconditioned matrix S = L^-1 M L^-T (L L^T = per-body mobility), then
 * x = B L S^{1/2} W: covariance B M B exactly, a handful of iterations instead of ~30 (own configuration only) */
#define RBL_MHALF_LANCZOS_PC 2
int rbl_M_half_W(rbl_ctx *ctx, const double *W, uint64_t seed, int method, double *out);

/* same on caller-supplied positions (n3 free, like apply_M) */
int rbl_M_half_W_r(rbl_ctx *ctx, const double *r_vecs, int64_t n3, const double *W,
                   uint64_t seed, int method, double *out);

/* M_RFD(), c_rigid_obj.cpp:769-796 (C++ only in the reference): random finite difference of the
 * mobility, (1/delta)[M(q + delta/2 Kinv W) - M(q - delta/2 Kinv W)] W, out[3N].  W == NULL draws
 * N(0,1) from `seed`.  The reference hard-codes delta = 1e-4. */
int rbl_M_RFD(rbl_ctx *ctx, const double *W, uint64_t seed, double delta, double *out);

/* KTinv_RFD(), c_rigid_obj.cpp:743-767: K^T (1/delta)[Kinv(q+)^T - Kinv(q-)^T] W, W[6Nb] -> out[6Nb] */
int rbl_KTinv_RFD(rbl_ctx *ctx, const double *W, double delta, double *out);

/* M_RFD_cfgs(U, delta), c_rigid_obj.cpp:798-818: blob positions of the two configurations displaced by +-(delta/2) U
 * (U[6Nb], displacement units) -> r_plus[3N], r_minus[3N] (computed on the GPU; nothing is committed).  The reference
 * also draws a noise vector there that it never uses (:801). */
int rbl_M_RFD_cfgs(rbl_ctx *ctx, const double *U, double delta, double *r_plus, double *r_minus);

/* M_RFD_from_U(U, W), c_rigid_obj.cpp:820-842: (1/delta) [M(q + delta/2 U) - M(q - delta/2 U)] W with the caller's
 * displacement U[6Nb] and vector W[3N] -> out[3N].  The reference hard-codes delta = 1e-3 here (:822). */
int rbl_M_RFD_from_U(rbl_ctx *ctx, const double *U, const double *W, double delta, double *out);

/* KT_RFD_from_U(U, W), c_rigid_obj.cpp:844-863: (1/delta) [K(q + delta/2 U)^T - K(q - delta/2 U)^T] W, W[3N] -> out[6Nb]
 * (delta = 1e-3 in the reference, :846). */
int rbl_KT_RFD_from_U(rbl_ctx *ctx, const double *U, const double *W, double delta, double *out);

/* evolve_X_Q_RFD(U), c_rigid_obj.cpp:880-893: commit the configuration displaced by U[6Nb] (displacement units: NOT
 * multiplied by dt), rebuild K, and KEEP the preconditioner of the configuration it was built for (:892 sets
 * PC_mat_Set = true: an RFD displacement is of size delta, the factors of q serve q + delta U). */
int rbl_evolve_X_Q_RFD(rbl_ctx *ctx, const double *U);

/* The saddle operator a caller's Krylov solver applies, src/Rigid.py:73-80:  x = [lambda (3N) ; U (6Nb)]  ->
 * [M lambda - K U ; K^T lambda] on the object's own configuration, host vectors, ONE upload and ONE download (the
 * wrapper composes it from multi_body_pos + apply_M + K_x_U + KT_x_Lam: four round trips). */
int rbl_apply_saddle(rbl_ctx *ctx, const double *x, double *out);

/* update_X_Q(U), c_rigid_obj.cpp:798-863: the configuration displaced by U[6Nb] (translation and rotation
 * vector per body, displacement units) -> X_out[3Nb], Q_out[4Nb] (scalar-first); nothing is committed. */
int rbl_update_X_Q(rbl_ctx *ctx, const double *U, double *X_out, double *Q_out);

/* RHS_and_Midpoint(Slip, Force), c_rigid_obj.cpp:917-976 (C++ only in the reference): right-hand side and
 * predictor configuration of the stochastic midpoint step.
 *   in : Slip[n3], Force[6Nb] (NOT modified -- the reference mutates its arguments, :963,972);
 *        W = [W1 | W2 | W_rfd], 3*n3 standard normals, or NULL to draw them from `seed` (:730-741 seeds
 *        from the clock); method = RBL_MHALF_*; split_rand as the reference member (:150, default 1);
 *        delta = RFD step (the reference hard-codes 1e-4, :771)
 *   out: RHS[n3 + 6Nb] = [ Slip - (kBT*M_RFD + BI) ; -Force ],
 *        BI = c2 (M^1/2 W1 - M^1/2 W2), c1 = 2 sqrt(kBT/dt), c2 = sqrt(kBT/dt)   (split_rand)
 *        BI = c2  M^1/2 W1,             c1 = c2 = sqrt(2 kBT/dt)                 (otherwise)
 *        X_half[3Nb], Q_half[4Nb] = update_X_Q((dt/2) Kinv c1 M^1/2 W1)           (:955-959)
 *   kBT <= 1e-10: RHS = [Slip ; -Force], X_half/Q_half = current configuration     (:967-970)
 * With RBL_MHALF_CHOLESKY the dense factor is computed once and applied to W1 and W2. */
int rbl_RHS_and_Midpoint(rbl_ctx *ctx, const double *Slip, const double *Force, const double *W,
                         uint64_t seed, int method, int split_rand, double delta, double *RHS,
                         double *X_half, double *Q_half);

/* Lanczos controls / report.  max_iter: the basis is kept (max_iter + 1 vectors of 3 N doubles per recurrence).  tol: the recurrence stops when the ERROR ESTIMATE of the increment is below tol (relative):
 * the last correction d_m = |x_m - x_{m-1}| / |x_m| extrapolated geometrically, d_m rho / (1 - rho), rho = d_m / d_{m-1}.
 * For RBL_MHALF_LANCZOS_PC the estimate is taken in the Euclidean norm of the increment itself (RBL_OPT_LANCZOS_EUCLID_NORM = 0:
 * in its energy norm).  The report returns the iterations used by the last call and that estimate. */
int rbl_set_lanczos(rbl_ctx *ctx, int max_iter, double tol);
int rbl_get_lanczos_report(const rbl_ctx *ctx, int *iters, double *resid);

/* dense lower Cholesky of a caller matrix (column-major n x n, in place on the
 * device, result copied back; strict upper triangle zeroed). */
int rbl_cholesky_lower(rbl_ctx *ctx, double *M, int64_t n);

/* sizes / flags */
int rbl_get_sizes(const rbl_ctx *ctx, int *N_bod, int *N_blb);
const char *rbl_last_error(const rbl_ctx *ctx);

/* test hook: n independent 3x3 blocks (row-major, scaled by 1/(8 pi eta a)) of
 * pairs (ri[k], rj[k]) with indices (ii[k], jj[k]); mode 0 = reference-order
 * arithmetic (dense-build kernel), 1 = fast matvec arithmetic. */
int rbl_debug_pair_blocks(rbl_ctx *ctx, const double *ri, const double *rj,
                          const int32_t *ii, const int32_t *jj, int64_t n, int wall,
                          int mode, double *out9);

/* ===================================================================== */
/* 3. Device-pointer API (resident data, multi-GPU shards, benchmarks)    */
/* ===================================================================== */

/* Bind to the calling thread's current HIP device and a stream handle
 * (hipStream_t as void*; NULL = the default stream). */
int rbl_set_stream(rbl_ctx *ctx, void *hip_stream);

/* rows [row_begin,row_end) of apply_M over n_blobs blobs.  d_F, d_r: n_blobs*3
 * doubles on the device; d_out: 3*(row_end-row_begin).  Error conditions are
 * latched in a device word, read with rbl_sync_check. */
int rbl_apply_M_dev(rbl_ctx *ctx, const double *d_F, const double *d_r, int64_t n_blobs,
                    int64_t row_begin, int64_t row_end, double *d_out);

/* apply_M for nrhs right-hand sides resident on the device (d_F, d_out column-major
 * 3*n_blobs x nrhs).  Four or more vectors run on the fp64 matrix cores, 16 per pass. */
int rbl_apply_M_multi_dev(rbl_ctx *ctx, const double *d_F, const double *d_r, int64_t n_blobs,
                          int nrhs, double *d_out);

/* Symmetric-kernel shard of apply_M for multi-GPU strong scaling: this call evaluates the
 * unordered blob-tile pairs {I,J}, J >= I, of share i_first out of i_step: one row UNIT out of every i_step consecutive
 * ones (a unit = the rows one workgroup sweeps: 64 blobs, or 512 for large systems), at offset i_first in even blocks of
 * units and i_step - 1 - i_first in odd ones, so that the triangular work is the same for every share; it writes the PARTIAL sum of U = [B] M [B] F over all 3*n_blobs entries
 * to d_out; the sum of the i_step partial vectors (an all-reduce) is the full product.
 * i_first = 0, i_step = 1 is the whole product. */
int rbl_apply_M_sym_dev(rbl_ctx *ctx, const double *d_F, const double *d_r, int64_t n_blobs,
                        int i_first, int i_step, double *d_out);
/* the same for nrhs = 1 or 2 force vectors at once (d_F, d_out: [nrhs][3 n_blobs]); with two vectors the pair
 * coefficients are evaluated once for both (the two Brownian increments of the stochastic step) */
int rbl_apply_M_sym_multi_dev(rbl_ctx *ctx, const double *d_F, const double *d_r, int64_t n_blobs, int nrhs,
                              int i_first, int i_step, double *d_out);

/* launch geometry rbl_apply_M_sym[_multi]_dev would use (reporting: bench.py names the kernel instantiation it
 * times -- k_apply_M_sym<wall, rows_per_lane> -- and the slab workspace it needs) */
int rbl_apply_M_sym_info(rbl_ctx *ctx, int64_t n_blobs, int i_step, int nrhs, int *rows_per_lane, int *chunk_tiles,
                         int64_t *workspace_bytes);

/* ... and the kernel instantiation a one- or two-vector symmetric product of that size launches under the context's options
 * ("k_apply_M_sym<true,2>", "k_apply_M_symw<false>", ...; name: >= 40 bytes) */
int rbl_apply_M_sym_kernel(rbl_ctx *ctx, int64_t n_blobs, int i_step, int nrhs, int wall, char *name, int name_len);

/* blob positions of bodies [body_begin, body_end) into d_out
 * (3*N_blb*(body_end-body_begin)); uses the host-side configuration. */
int rbl_blob_positions_dev(rbl_ctx *ctx, int body_begin, int body_end, double *d_out);

/* multi_body_pos() into a device vector d_out[3 N] according to the context's communicator: all bodies on this rank (single
 * GPU, tile-pair split), or -- RBL_OPT_COMM_SPLIT = 1 -- this rank's bodies followed by ONE all-gather over the ranks. */
int rbl_multi_body_pos_dev(rbl_ctx *ctx, double *d_out);

/* dense build on the device: d_out column-major n3 x n3 (ld = n3) */
int rbl_rotne_prager_tensor_dev(rbl_ctx *ctx, const double *d_r, int64_t n_blobs,
                                int scale_damp, double *d_out);

/* in-place lower Cholesky on the device (strict upper left untouched unless
 * zero_upper), then out = L W */
int rbl_cholesky_lower_dev(rbl_ctx *ctx, double *d_M, int64_t n, int zero_upper);
int rbl_trmv_lower_dev(rbl_ctx *ctx, const double *d_L, int64_t n, const double *d_W,
                       double *d_out);

/* M^{1/2} W on the device with positions d_r (n_blobs) and noise d_W (3 n_blobs) */
int rbl_M_half_W_dev(rbl_ctx *ctx, const double *d_r, int64_t n_blobs, const double *d_W,
                     int method, double *d_out);

/* ---- device-resident rigid-body operators: a Krylov iteration without host round trips ----
 * rbl_sync_bodies_dev uploads (X, Q, ref_cfg) and computes lever arms + blob positions on the
 * GPU; the functions below call it themselves when the configuration has changed.
 * Vectors: U/F 6*N_bod, lambda/slip 3*N, saddle/PC vectors 3*N + 6*N_bod (reference layout). */
int rbl_sync_bodies_dev(rbl_ctx *ctx);
/* uploads + workspace growth + preconditioner build for the current configuration, so that the
 * operator calls below are launch-only afterwards (capturable in a hipGraph) */
int rbl_prepare_dev(rbl_ctx *ctx);
int rbl_positions_dev(rbl_ctx *ctx, const double **d_pos, int64_t *n_blobs);  /* multi_body_pos, resident */
int rbl_K_x_U_dev(rbl_ctx *ctx, const double *d_U, double *d_out);            /* K_x_U    :404 */
int rbl_KT_x_Lam_dev(rbl_ctx *ctx, const double *d_lambda, double *d_out);    /* KT_x_Lam :410 */
int rbl_apply_PC_dev(rbl_ctx *ctx, const double *d_in, double *d_out);        /* apply_PC :589, diagonal PC */
int rbl_apply_saddle_dev(rbl_ctx *ctx, const double *d_x, double *d_out);     /* src/Rigid.py:73-80 */
/* Per-body factors L L^T = M_body of the object's own configuration (wall term per wall_PC, undamped; the
 * block-diagonal preconditioner's factors), applied to a blob vector d_in[n3] -> d_out[n3]:
 * mode 0: (L L^T)^-1 x, 1: L^-1 x, 2: L^-T x, 3: L x (mode 3 not in place).  With the wall term L is the lower Cholesky
 * factor of the body's block, rebuilt per configuration.  In free space every body's block is ONE body-frame matrix seen
 * through the body's rotation, M_b = (I x R_b) M_body (I x R_b)^T, so the factor is L = (I x R_b) chol(M_body): built once per
 * rbl_set_parameters, exact for every configuration, not triangular (mode 0 is the same operator either way;
 * RBL_OPT_BODYFRAME_FACTOR = 0 restores per-configuration Cholesky factors).  rbl_set_no_damp(ctx, 1) makes the matvec entry points
 * apply the plain wall-corrected M (no damping B) until switched off again: together they let a caller compose
 * the preconditioned square root  B L (L^-1 M L^-T)^{1/2} W  around its own (e.g. sharded) product.
 * Modes 5, 6, 7 (all bodies, not in place, single GPU) apply the WHOLE factor G the library's preconditioned Lanczos root
 * uses on this configuration -- G^-1 x, G^-T x, G x with G = L (I + Q (L_E - I) Q^T), the two-level factor (RBL_OPT_LANCZOS_TWO_LEVEL,
 * default), or G = L: what a test needs to check the root identities  s = G^-1 B^-1 x,  v = G^-T W,  |s|^2 = v^T M v,
 * root(s) = B M v  whatever the factor. */
int rbl_block_solve_dev(rbl_ctx *ctx, const double *d_in, double *d_out, int mode);
/* the same for the bodies [body_begin, body_end) only (body_end < 0: to the last body): d_in / d_out are still
 * full-length blob vectors, only the entries of those bodies are read and written, and only those bodies are
 * factored -- a multi-GPU driver gives every rank its own bodies and all-gathers the result. */
int rbl_block_solve_range_dev(rbl_ctx *ctx, const double *d_in, double *d_out, int mode, int body_begin, int body_end);
int rbl_set_no_damp(rbl_ctx *ctx, int on);
/* Keep the per-body Cholesky factors for `every` configuration changes (default 1: rebuilt after each change) before
 * they are rebuilt: both of their uses -- the block-diagonal preconditioner and the L of the preconditioned square
 * root -- stay exact with factors of a nearby configuration, only the iteration counts move.  M^-1 K and
 * (K^T M^-1 K)^-1 are still rebuilt for every configuration (with the kept factors). */
int rbl_set_block_refresh(rbl_ctx *ctx, int every);

/* Right-preconditioned GMRES(max_iter <= 255, no restart) on the saddle operator of the object's own
 * configuration (src/Rigid.py:73-80 is what a caller's Krylov solver applies; the reference ships no solver):
 * solves [M -K; K^T 0] x = rhs with P^-1 = apply_PC, all vectors in HBM.  rtol <= 0: exactly max_iter
 * iterations, no host round trip inside the loop; rtol > 0: stops at the first iteration whose residual
 * estimate is below rtol (tested every iteration, or every 4th for small launch-bound systems).  d_rhs, d_x: n3 + 6 N_bod doubles. */
int rbl_gmres_saddle_dev(rbl_ctx *ctx, const double *d_rhs, int max_iter, double rtol, double *d_x,
                         int use_x0 /* d_x holds an initial guess, e.g. the previous step's solution */,
                         int *iters, double *resid);
/* the same solve for host vectors (one upload of rhs [and x0], one download of x): what a caller of the wrapper's apply_saddle /
 * apply_PC would otherwise loop over from outside (src/Rigid.py:69-80) */
int rbl_gmres_saddle(rbl_ctx *ctx, const double *rhs, int max_iter, double rtol, double *x, int use_x0, int *iters, double *resid);
/* nrhs right-hand sides of the SAME configuration in lock step (rhs, x: nrhs vectors of n3 + 6 N_bod doubles, one after the other):
 * nrhs independent GMRES recurrences -- each column gets exactly the iterates rbl_gmres_saddle_dev would give it, its own iteration
 * count and residual (iters, resid: nrhs entries, either may be NULL) -- whose mobility products run 16 at a time on the fp64
 * matrix cores (the multi-vector kernel of rbl_apply_M_multi_dev) and whose block-preconditioner applications share passes over
 * the per-body factors.  The customer: the body mobility matrix (6 N_bod unit loads), several noise realisations of one
 * configuration.  The operator such loops iterate is the reference's src/Rigid.py:69-80. */
int rbl_gmres_saddle_multi_dev(rbl_ctx *ctx, const double *d_rhs, int nrhs, int max_iter, double rtol, double *d_x, int *iters,
                               double *resid);
int rbl_gmres_saddle_multi(rbl_ctx *ctx, const double *rhs, int nrhs, int max_iter, double rtol, double *x, int *iters, double *resid);
/* Whole time steps in one call, on the object's own configuration (the reference has no driver; these are what
 * rigid_body_light_amd/krylov.py's steppers do, for hosts without a Python loop).
 *   rbl_step_deterministic: solve [M -K; K^T 0][lambda; U] = [slip; -F_body] (rbl_gmres_saddle_dev; slip NULL = 0;
 *       warm_start: 0 cold, 1 begin from the previous call's solution, 2 / 3 from the linear / quadratic extrapolation
 *       of the last two / three solutions), then evolve_X_Q(U).
 *   rbl_step_brownian: stochastic midpoint step -- RHS_and_Midpoint at q^n (W = [W1|W2|W_rfd] host, or NULL for
 *       seeded device noise; method RBL_MHALF_*), saddle solve at q^{n+1/2}, update from q^n.
 * F_body: host, 6 N_bod; slip: host, 3 N_blobs.  iters / resid report the GMRES run. */
int rbl_step_deterministic(rbl_ctx *ctx, const double *F_body, const double *slip, int max_iter, double rtol,
                           int warm_start, int *iters, double *resid);
int rbl_step_brownian(rbl_ctx *ctx, const double *F_body, const double *slip, const double *W, uint64_t seed,
                      int method, int split_rand, double delta, int max_iter, double rtol, int *iters,
                      double *resid);

/* RHS_and_Midpoint on device vectors (d_Slip[n3], d_Force[6Nb], d_W[3 n3] or NULL, d_RHS[n3+6Nb]);
 * X_half / Q_half are host arrays (O(N_bod)). */
int rbl_RHS_and_Midpoint_dev(rbl_ctx *ctx, const double *d_Slip, const double *d_Force, const double *d_W,
                             uint64_t seed, int method, int split_rand, double delta, double *d_RHS,
                             double *X_half, double *Q_half);

/* ---- multi-GPU inside the library's own solvers --------------------------------------------------------------
 * One process per GPU, every rank holds the same (replicated) body state and calls the same entry points with the
 * same arguments.  Once a context has a communicator, every FULL mobility product the library evaluates for itself --
 * the iterations of rbl_gmres_saddle_dev, of the Lanczos square roots, M_RFD, rbl_apply_saddle_dev, the whole-step
 * entry points, and rbl_apply_M_dev over the full row range -- is this rank's share of the work followed by one
 * collective (RBL_OPT_COMM_SPLIT):
 *   0 (default)  unordered blob-tile pairs, dealt as rbl_apply_M_sym_dev(rank, world) deals them; the partial
 *                full-length U is completed by ONE sum all-reduce (24 N bytes);
 *   1            rows by body index (SURVEY.md 8e / north_star): each rank computes the geometry of ITS bodies, the blob
 *                positions are all-gathered once per configuration, a product is the ordered-pair kernel on the rank's own
 *                rows followed by ONE all-gather of U.
 * Per-body work (block factors, their applications) is done by the body's owner -- bodies are split contiguously by
 * index, sizes differing by at most one -- and completed by an in-place all-gather of the owners' segments.  All vectors
 * of the Krylov recurrences stay replicated and bitwise identical on every rank, so the ranks take the same convergence
 * decisions.
 *
 * (a) RCCL inside the library -- what a C / C++ host (the reference is one, c_rigid_obj.cpp:997-1027) uses: rank 0 calls
 *     rbl_comm_unique_id and hands the RBL_COMM_ID_BYTES bytes to the other ranks by whatever means the host has (MPI_Bcast,
 *     a file, torch.distributed); every rank then calls rbl_comm_init_rccl (collective: ncclCommInitRank on the context's
 *     device).  The collectives are ncclAllReduce / ncclAllGather / grouped ncclBroadcast enqueued on the context's stream;
 *     no host code runs between two products of a solve.  librccl is opened at run time (dlopen "librccl.so.1"): the copy
 *     already in the process (PyTorch's) or ROCm's.  world == 1 is allowed and keeps the multi-GPU code path on with one
 *     share: a one-GPU rehearsal of what N ranks run.
 * (b) the caller's callbacks (rbl_set_comm / rbl_set_comm_ops) -- rehearsals over other transports (gloo with host staging).
 *     `allreduce` must leave the sum over all ranks in d_buf[0..count) on every rank; `allgatherv` (optional) must leave
 *     rank r's segment d_buf[offsets[r] .. offsets[r] + counts[r]) on every rank, in place; both ordered after the work
 *     already enqueued on the context's stream and before whatever is enqueued next.  Without `allgatherv` the library
 *     zero-pads and sums instead.  allreduce == NULL switches back to single-GPU products.
 * rbl_comm_finalize destroys the communicator (rbl_destroy does it too). */
typedef int (*rbl_allreduce_fn)(void *user, double *d_buf, int64_t count);
typedef int (*rbl_allgatherv_fn)(void *user, double *d_buf, const int64_t *offsets, const int64_t *counts);
int rbl_set_comm(rbl_ctx *ctx, int rank, int world, rbl_allreduce_fn allreduce, void *user);
int rbl_set_comm_ops(rbl_ctx *ctx, int rank, int world, rbl_allreduce_fn allreduce, rbl_allgatherv_fn allgatherv, void *user);
#define RBL_COMM_ID_BYTES 128
int rbl_comm_unique_id(void *id_out /* RBL_COMM_ID_BYTES */);
int rbl_comm_init_rccl(rbl_ctx *ctx, const void *unique_id, int rank, int world);
int rbl_comm_finalize(rbl_ctx *ctx);
/* kind: 0 single GPU, 1 callbacks, 2 RCCL inside the library */
int rbl_comm_info(const rbl_ctx *ctx, int *rank, int *world, int *kind);
/* the context's collectives themselves on a device buffer (tests, benchmarks): sum all-reduce; in-place all-gather of the
 * per-rank segments d_buf[offsets[r] .. + counts[r]) */
int rbl_comm_allreduce_dev(rbl_ctx *ctx, double *d_buf, int64_t count);
int rbl_comm_allgatherv_dev(rbl_ctx *ctx, double *d_buf, const int64_t *offsets, const int64_t *counts);

/* ---- per-phase timings of the library's own solvers (the reference has one gettimeofday helper,
 * c_rigid_obj.cpp:22-29, and one printf around M_half_W, :929-932) -----------------------------------------------------
 * rbl_set_timing(ctx, 1): from now on the phases below are bracketed by hipEvents on the context's stream (a few
 * microseconds of host time per bracket; off by default).  rbl_get_timings synchronises the stream and returns, per
 * phase, the GPU time in milliseconds and the number of brackets accumulated since the last rbl_reset_timings (arrays of
 * RBL_T_COUNT entries; either may be NULL).  RBL_T_TOTAL spans the solver entry points (rbl_gmres_saddle_dev, the
 * Lanczos square roots, M_RFD); what it holds beyond the other phases is Krylov vector work, K operators, launch gaps and
 * the host's convergence tests.  On a multi-GPU context (rbl_set_comm) RBL_T_COLLECTIVE is the time the stream spent in
 * the caller's all-reduce, including the wait for the slowest rank. */
#define RBL_T_PRODUCT 0     /* mobility products: pair kernels + their slab reduction                         */
#define RBL_T_PERBODY 1     /* applications of the per-body factors / inverses, preconditioner tail            */
#define RBL_T_FACTOR 2      /* per-body dense blocks, batched Cholesky, explicit inverses, M^-1 K, (K^T M^-1 K) */
#define RBL_T_COLLECTIVE 3  /* the all-reduce callback of rbl_set_comm                                          */
#define RBL_T_DENSE 4       /* dense B M B build, Cholesky, L W (RBL_MHALF_CHOLESKY)                            */
#define RBL_T_TOTAL 5       /* whole solver calls                                                               */
#define RBL_T_COUNT 6
int rbl_set_timing(rbl_ctx *ctx, int on);
int rbl_reset_timings(rbl_ctx *ctx);
int rbl_get_timings(rbl_ctx *ctx, double *ms, int64_t *calls);

/* stream-synchronise, read and clear the latched device error word */
int rbl_sync_check(rbl_ctx *ctx);

/* ---- named per-context options ---------------------------------------------------------------------------------
 * rbl_set_option(ctx, RBL_OPT_*, value) / rbl_get_option: an unknown key or a value outside the option's range returns
 * RBL_ERR_ARG and changes nothing.  rbl_option_info gives name, range and default of a key (tests enumerate
 * 1 .. RBL_OPT_COUNT - 1), rbl_option_key the key of a name (0: unknown).  All per context; defaults in brackets. */
enum {
  RBL_OPT_MATVEC_KERNEL = 1,       /* [0] 0 heuristic (symmetric kernel for full products, MFMA kernel for >= 4 vectors), 1 ordered-rows
                                      kernel, 2 symmetric kernel, 3 MFMA multi-RHS kernel also for few vectors                        */
  RBL_OPT_ORDERED_JSPLIT = 2,      /* [0] j-split of the ordered kernel (0 = heuristic)                                               */
  RBL_OPT_SYM_CHUNK = 3,           /* [0] column tiles per work unit of the symmetric kernels (0 = heuristic)                         */
  RBL_OPT_SYM_ROWS_PER_LANE = 4,   /* [0] rows per lane of the one-vector symmetric kernel: 0 heuristic, 1, 2 (experiments)            */
  RBL_OPT_SYM2_ROWS_PER_LANE = 5,  /* [0] the same for the two-vector kernel                                                          */
  RBL_OPT_SYM_WAVES = 6,           /* [0] waves per workgroup of the symmetric kernels: 0 heuristic, 1 or 4 (4 needs two rows per lane; any other
                                      value, or a combination no kernel has, is RBL_ERR_ARG -- at the call or at the product)            */
  RBL_OPT_SYM_WORK_QUEUE = 7,      /* [1] large systems (four-wave workgroups): 1 a fixed set of resident workgroups draws work units
                                      from a counter (an XCD that runs faster takes more), 0 one unit per workgroup in launch order;
                                      the slabs are addressed by unit, so results are bitwise the same either way                   */
  RBL_OPT_GMRES_PC_SIGN_FIX = 8,   /* [1] rbl_gmres_saddle_dev applies apply_PC with the sign of its force block restored (one
                                      eigenvalue cluster at +1; fewer iterations, same solution); 0: the reference's sign (:601)    */
  RBL_OPT_GMRES_ONE_KERNEL = 9,    /* [1] small systems (<= 256 blobs, diagonal PC, <= 255 iterations): whole solve in ONE launch    */
  RBL_OPT_GMRES_PREDICT_CHECKS = 10, /* [1] launch-bound systems (<= 20 000 blobs): convergence tests placed by the previous solve's
                                      count and the residual's rate; 0: every 4th iteration                                         */
  RBL_OPT_GMRES_OVERLAP_CHECK = 11, /* [1] the host reads the Hessenberg columns of a convergence test while the GPU already applies the
                                      preconditioner of the next iteration (no idle stream at the test); 0: drain, then go on          */
  RBL_OPT_RELAXED_KRYLOV = 12,     /* [0] inexact Krylov: 1 = once GMRES's residual estimate is below rtol x 1e5 (and in Lanczos runs to
                                      tolerances >= 1e-4) far tile pairs are evaluated in packed single precision (relative product
                                      error <= 3e-6, ~1.8x faster); the solution still satisfies the fp64 system to rtol.  2 = in those
                                      Lanczos runs only (a root asked for to 1e-3 does not see a product error of 1e-6): every GMRES
                                      product stays fp64                                                                             */
  RBL_OPT_RELAXED_ALWAYS = 13,     /* [0] test hook: every full product through that relaxed kernel                                   */
  RBL_OPT_BLOCK_EXPLICIT_SMALL = 14, /* [1] per-body factors of bodies with <= 170 blobs applied through explicit inverses L^-1       */
  RBL_OPT_BLOCK_EXPLICIT_LARGE = 15, /* [2] explicit inverses of larger bodies (the reference's own form of Block_diag_invM,
                                      :461-487): 0 never, 1 always, 2 when it pays (multi-GPU contexts; the shared body-frame factor) */
  RBL_OPT_BLOCK_INVERSE_F32 = 16,  /* [0] keep (also) a single-precision copy of the large inverses: the preconditioner and Lanczos
                                      runs to tolerances >= 1e-5 read half the bytes (sums stay fp64)                               */
  RBL_OPT_BODYFRAME_FACTOR = 17,   /* [1] free space: ONE body-frame factor rotated with each body; 0: per-configuration factors     */
  RBL_OPT_BODYFRAME_WALL_APPROX = 18, /* [0] with the wall term: the free-space body-frame factor as an APPROXIMATE block factor      */
  RBL_OPT_BLOCK_REFRESH = 19,      /* [1] keep the per-body factors for this many configuration changes (rbl_set_block_refresh)      */
  RBL_OPT_LANCZOS_TWO_LEVEL = 20,  /* [1] preconditioned root: two-level factor L (I + Q (L_E - I) Q^T); 0: block-Jacobi factor L    */
  RBL_OPT_LANCZOS_EUCLID_NORM = 21, /* [1] preconditioned root stops on the error estimate of x itself (Euclidean norm);
                                      0: of z = (L^-1 M L^-T)^{1/2} W, i.e. of x in the energy norm x^T (B M B)^-1 x                 */
  RBL_OPT_LANCZOS_REORTH = 22,     /* [1] every new Lanczos vector re-orthogonalised against the whole basis; 0: three-term recurrence */
  RBL_OPT_NO_DAMP = 23,            /* [0] transient: the matvec entry points apply the plain wall-corrected M, no damping B          */
  RBL_OPT_COMM_SPLIT = 24,         /* [0] multi-GPU contexts: 0 unordered tile pairs + all-reduce(U), 1 rows by body index +
                                      all-gather(positions, U) -- see "multi-GPU" above                                             */
  RBL_OPT_FUSED_KRYLOV = 25,       /* [1] launch-bound systems: the product's slab reduction also writes the saddle tail and the partial
                                      sums of the Arnoldi step's first Gram-Schmidt pass, and (free-space body-frame preconditioner) the
                                      normalisation of the new basis vector is done by the preconditioner kernels that consume it
                                      (three launches fewer per GMRES iteration); 0: one kernel per operation                       */
  RBL_OPT_RELAXED_GAP_RATIO = 26,  /* [0] relaxed product: a far tile pair is swept in single precision when the extents of its boxes,
                                      d_I + 2 d_J, are at most this many times their gap (0 = the library's default); smaller = fewer
                                      pairs relaxed, smaller product error (6e-8 (1 + ratio) of a separation)                        */
  RBL_OPT_SYM_WAVE_UNITS = 27,     /* [1] mid-size systems (one row per lane, < 128 blob tiles): work units owned by single waves, four
                                      independent waves per workgroup, column sums rotating through the lanes in registers
                                      (k_apply_M_symw); 0: the round-3 kernel (one workgroup per unit, column sums by LDS atomics).
                                      Same slabs and reduction either way                                                          */
  RBL_OPT_SHARED_GEMM = 28,        /* [1] free space, bodies of <= 170 blobs: the ONE body-frame inverse / preconditioner table is applied to
                                      all bodies' vectors as a matrix-matrix product on the fp64 matrix cores (the table read once);
                                      0: batched matrix-vector products (every body re-reads it)                                    */
  RBL_OPT_TWO_LEVEL_REFRESH = 29,  /* [1] two-level factor of the preconditioned root: keep the factored coarse (body-centre) operator for this
                                      many configuration changes; the bodies' coarse basis Q follows every change.  The root stays exact
                                      for ANY coarse operator (H^-1 is the exact inverse of H whatever L_E is); a stale one costs at most
                                      a Lanczos iteration and saves its 3 N_bod-square Cholesky factor + inverse per step             */
  RBL_OPT_BLOCK_TILE_FACTOR = 30,  /* [1] per-body factors (and explicit inverses) of bodies with 3 N_blb > 512 built by ONE dataflow launch over
                                      128 x 128 tiles (rbl_tilechol.hip); 0: the batched panel kernels of rounds 1-4 (12 + 12 launches)      */
  RBL_OPT_COMM_FORCE_STAGED = 31,  /* [0] test hook: the in-place all-gathers of a native (RCCL) communicator take the STAGED form (segments
                                      padded to the largest share, one ncclAllGather, unpacked) even when the shares are equal -- what a job
                                      with N_bod % world != 0 runs, exercised with one rank                                                 */
  RBL_OPT_BLOCK_SOLVE_PIPE = 32,   /* [1] substitution through the factors of bodies with 3 N_blb > 512: ONE software pipeline per body (the
                                      dependent chain of diagonal solves in one wave, fifteen waves streaming the factor with the next
                                      step's loads already in flight, one barrier a step: k_block_solve_pipe); 0: the two-barrier kernel
                                      of rounds 1-4 (k_block_solve).  Same sums per row in another association                            */
  RBL_OPT_COUNT = 33
};
int rbl_set_option(rbl_ctx *ctx, int option, int64_t value);
int rbl_get_option(const rbl_ctx *ctx, int option, int64_t *value);
int rbl_option_info(int option, const char **name, int64_t *min_value, int64_t *max_value, int64_t *default_value);
int rbl_option_key(const char *name);

#ifdef __cplusplus
}
#endif
#endif /* RBL_H */
