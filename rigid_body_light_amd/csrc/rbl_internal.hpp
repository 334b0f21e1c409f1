// rbl_internal.hpp -- private declarations shared by the librbl translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rbl.h"
#include "rbl_pair.hpp"

#pragma GCC visibility push(hidden)   // nothing declared here is part of librbl.so's ABI (that is include/rbl.h alone)

// ----------------------------------------------------------------------------
// Host-side rigid-body state: the O(N_bod) bookkeeping of CManyBodies
// (reference c_rigid_obj.cpp:144-168 members).  Plain C++, no Eigen.
// ----------------------------------------------------------------------------
struct RblBodyState {
  double a = 0, dt = 0, kBT = 0, eta = 0;
  bool wall = false;          // PC_wall   :147
  bool block_pc = false;      // block_diag_PC :148
  double M_scale = 1.0;       // :149,194
  bool pc_set = false;        // PC_mat_Set :151
  bool cfg_set = false;       // :157
  bool params_set = false;    // :166
  bool K_set = false;
  int N_bod = 0, N_blb = 0;
  std::vector<double> ref_cfg;  // N_blb x 3 row-major, mean removed (:188)
  std::vector<double> X;        // 3 N_bod
  std::vector<double> Q;        // 4 N_bod, scalar-first, unit
  // K is stored implicitly as the body-frame-rotated lever arms r_k = R(Q_b) c_k
  std::vector<double> lever;    // 3 N   (r_k - X_b), :374
  std::vector<double> KTKinv;   // 36 N_bod  (6x6 row-major per body), :302-326
  // preconditioner caches (:152-154)
  std::vector<double> invM_diag;   // 9 N      (3x3 row-major per blob) when !block_pc
  std::vector<double> Ninv_chol;   // 36 N_bod  lower Cholesky of K^T invM K blocks
};

void rbl_quat_to_rot(const double *q, double *R9);
int rbl_body_set_K(RblBodyState &S, std::string &err);
void rbl_body_K_x_U(const RblBodyState &S, const double *U, double *out);
void rbl_body_KT_x_Lam(const RblBodyState &S, const double *lam, double *out);
void rbl_body_Kinv_x_V(const RblBodyState &S, const double *V, double *out);
void rbl_body_KTinv_x_F(const RblBodyState &S, const double *F, double *out);
void rbl_body_update_X_Q(const RblBodyState &S, const double *U, std::vector<double> &Xo,
                         std::vector<double> &Qo);
int rbl_body_apply_PC(RblBodyState &S, const double *in, double *out, std::string &err);
int rbl_chol6(double *A);  // in-place lower Cholesky of a 6x6 row-major block

// ----------------------------------------------------------------------------
// Device buffers + context
// ----------------------------------------------------------------------------
struct RblDevBuf {
  void *p = nullptr;
  size_t bytes = 0;
};

// saddle epilogue fused into the slab reduction of a symmetric product (launch-bound systems, RBL_OPT_FUSED_KRYLOV): besides
// out = B M B F the reduction also leaves  w_top = out - K U  and  w_bot = ktl  (the K^T lambda the block preconditioner left)
struct RblSaddleFuse {
  const double *lever = nullptr;   // nullptr: off
  const double *U = nullptr;       // 6 N_bod
  const double *ktl = nullptr;     // 6 N_bod
  double *w = nullptr;             // 3 N + 6 N_bod
  int N_blb = 0, nb6 = 0;
  // ... and the first Gram-Schmidt pass of the Arnoldi step that follows: partial sums of V_k . w over each block's 64 entries,
  // dotPart[k][dotNp] (block order; slot dotNp - 1 = the body rows) -- what k_mdot_partial would leave for k_arnoldi_upd
  const double *dotV = nullptr;    // basis, dotStride doubles between vectors; dotK of them (0: off)
  long dotStride = 0;
  int dotK = 0, dotNp = 0;
  double *dotPart = nullptr;
};

// normalisation of an Arnoldi vector folded into the (linear) preconditioner that consumes it: the kernels take w, the partial sums
// of |w|^2 its second Gram-Schmidt pass left, apply P^-1 (w / |w|) and store w / |w| and |w| on the side (GMRES, free-space tables)
struct RblNormFold {
  const double *part = nullptr;    // np partial sums (nullptr: off)
  int np = 0;
  double *vnext = nullptr;         // w / |w|  (3 N + 6 N_bod)
  double *hout = nullptr;          // |w|
};

struct RblSymTune {        // per-context tuning of the symmetric matvec kernels (RBL_OPT_SYM_*)
  int chunk = 0;           // > 0: column tiles per work unit (0 = heuristic)
  int ni2 = 0;             // > 0: rows per lane of the two-vector kernel (0 = same rule as one vector)
  int ni1 = 0;             // > 0: rows per lane of the one-vector kernel (0 = heuristic; experiments)
  int sw = 0;              // > 0: waves per workgroup (0 = heuristic; experiments)
  int queue = 0;           // < 0: one unit per workgroup in launch order also for large systems (RBL_OPT_SYM_WORK_QUEUE = 0); 0: work queue there
  int gap_ratio = 0;       // relaxed product: a tile pair is swept in single precision when (d_I + 2 d_J) <= gap_ratio x gap (0 = default; RBL_OPT_RELAXED_GAP_RATIO)
  int wave_units = 0;      // < 0: mid-size systems on the round-3 kernel (one workgroup per unit, column sums by LDS atomics); 0: wave-owned units (RBL_OPT_SYM_WAVE_UNITS)
  RblSaddleFuse fuse;      // transient: see RblSaddleFuse
  int relaxed = 0;         // transient: far tile pairs in packed single precision (inexact Krylov iterations only)
};

struct RblCholAux {        // second stream + events for the one-panel lookahead
  hipStream_t stream = nullptr;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
};

struct rbl_ctx {
  RblBodyState S;
  std::string last_error;
  // device
  bool dev_ready = false;
  int device = -1;
  int n_cu = 0;
  hipStream_t stream = nullptr;
  RblDevBuf d_r, d_F, d_U, d_part, d_W, d_cfg, d_XQ, d_mat, d_tmp, d_tmp2, d_chol;
  RblDevBuf d_lever, d_pos, d_invM2, d_NL, d_sad;   // device-resident body state (rbl_sync_bodies_dev)
  RblDevBuf d_blkL, d_blkLinv, d_pcw, d_pcMK;       // block-diagonal PC: per-body Cholesky factors, work, invM K
  RblDevBuf d_blkX, d_blkTmp;                       // explicit L_b^-1 (two layouts), scratch of their application
  RblDevBuf d_commStage;                            // padded slots of a ragged in-place all-gather (rbl_comm.hip)
  RblDevBuf d_blkXf, d_blkAug;                      // large bodies: single-precision copy of the inverses; scratch of their inversion
  bool comm_force_staged = false;                   // RBL_OPT_COMM_FORCE_STAGED (test hook)
  bool blk_pipe = true;                             // substitution through large bodies' factors: the one-barrier pipeline (k_block_solve_pipe); false: k_block_solve
  bool blk_tile = true;                             // RBL_OPT_BLOCK_TILE_FACTOR: large bodies factored (and inverted) by the dataflow tile kernel
  int blk_large = 2;                                // explicit inverses of bodies with 3 N_blb > 512: 0 never, 1 always, 2 when it pays
                                                    // (multi-GPU contexts: few bodies per rank; shared body-frame factor: built once) -- RBL_OPT_BLOCK_EXPLICIT_LARGE
  bool blk_f32 = false, blk_f32_valid = false;      // RBL_OPT_BLOCK_INVERSE_F32: apply them from a single-precision copy (half the bytes)
  bool bf_tables = false;                           // d_bfPC holds the body-frame preconditioner tables (small bodies)
  // two-level factor of the preconditioned Lanczos root (rbl_roots.hip: tl_build): G = L (I + Q (L_E - I) Q^T)
  RblDevBuf d_tlQ, d_tlCb, d_tlCs, d_tlA, d_tlLinv, d_tlX, d_tlT, d_tlZ;
  int tl_refresh = 1, tl_age = 0;     // RBL_OPT_TWO_LEVEL_REFRESH: configuration changes the factored coarse operator is kept for
  bool tl_q_stale = false;            // ... while its basis Q (rotations, per-body factors) is rebuilt at every change
  bool tl_on = true, tl_valid = false, tl_ok = false;   // RBL_OPT_LANCZOS_TWO_LEVEL; built for the current configuration; usable (SPD)
  unsigned *d_err2 = nullptr;                       // error word of the two-level build: a failure there is not an error, only "not usable"
  double body_radius = 0.0;                         // max |c_k| + a: the sphere the far-field model gives a body
  // free space: M_b = (I x R_b) M_body (I x R_b)^T with ONE body-frame matrix for all bodies and all time: its factor
  // (d_bfL, d_bfLinv; d_bfX = explicit inverse when the body is small) is built once per rbl_set_parameters
  RblDevBuf d_bfL, d_bfLinv, d_bfX;
  RblDevBuf d_bfPC;                                 // small bodies: M_body^-1 (n^2) | M_body^-1 K_body (6 n) | chol(N_body) (36)
  bool blk_bodyframe = true, bf_valid = false, bf_inv = false;   // RBL_OPT_BODYFRAME_FACTOR
  bool bf_wall_approx = false;                                   // RBL_OPT_BODYFRAME_WALL_APPROX (experiment)
  RblDevBuf d_ktl;                                  // K^T Lambda of the last block-PC output (GMRES: the saddle product re-uses it)
  bool ktl_arm = false; const double *ktl_of = nullptr;   // armed by the GMRES loop only; ktl_of = the vector d_ktl belongs to
  bool shared_gemm = true;                          // RBL_OPT_SHARED_GEMM: the ONE body-frame matrix of free space applied to all bodies' vectors as a matrix-matrix product (MFMA)
  bool blk_explicit = true, blk_inv_valid = false;  // RBL_OPT_BLOCK_EXPLICIT_SMALL; d_blkX matches d_blkL for bodies blk_b0 .. blk_b1
  RblDevBuf d_bd, d_bd2;                            // RHS_and_Midpoint workspaces
  RblDevBuf d_gm;                                   // GMRES: Krylov basis, Hessenberg, scratch
  RblDevBuf d_step;                                 // time-step entry points: solution, rhs, slip, force
  RblDevBuf d_hist;                                 // the last (up to 3) deterministic-step solutions, a ring: warm start
  int step_hist_n = 0, step_hist_head = 0;          // entries held, slot of the newest one
  int64_t step_x_size = 0;
  bool dev_bodies_valid = false, dev_pc_valid = false, dev_xq_valid = false;
  bool pc_keep_once = false;    // rbl_evolve_X_Q_RFD: the next configuration re-synchronisation keeps the device preconditioner
  bool dev_blk_valid = false;   // per-body Cholesky factors (d_blkL, d_blkLinv) match the current configuration ...
  int blk_b0 = 0, blk_b1 = 0;   // ... for the bodies [blk_b0, blk_b1) (a multi-GPU driver factors only its own bodies)
  int blk_refresh = 1, blk_age = 0;   // rbl_set_block_refresh: keep the factors for blk_refresh configuration changes
  unsigned *d_err = nullptr;
  unsigned *h_err = nullptr;  // pinned
  void *h_stage = nullptr;    // pinned staging for large pageable host copies
  void *h_pin = nullptr;      // pinned, 1 MB: target of the small device-to-host reads of the solver loops (read_back)
  std::vector<double> h_xq;   // [X | Q] packed for one upload
  bool dev_cfg_valid = false; // d_cfg holds the current reference configuration
  double *h_coef = nullptr;   // pinned, 2 x 512 doubles: Krylov coefficients go up without draining the stream (the next write is a solve later)
  RblCholAux chol_aux;
  // apply_PC as the reference defines it (:601-608) answers [M -K; K^T 0] x = [slip; -F]: with the saddle operator of
  // src/Rigid.py:73-80 the preconditioned operator then has its 6 N_bod body eigenvalues at -1 and the rest at +1.  The
  // library's own GMRES may apply the preconditioner with the force block's sign restored (pc_fsign = +1: one cluster).
  double pc_fsign = -1.0;
  bool gmres_pc_sign_fix = true;
  bool gmres_small = true;
  // inexact-Krylov relaxation (off by default): once GMRES's residual estimate is below rtol x 1e5 its products may carry
  // a relative error of ~1e-6 without the solution losing accuracy -- far tile pairs then run in packed single precision
  int gmres_relax = 0;                              // RBL_OPT_RELAXED_KRYLOV: 0 fp64 throughout, 1 inexact GMRES + Lanczos products, 2 the Lanczos roots only
  bool force_relaxed = false;   // test / benchmark hook: every full product through the relaxed kernel
  // multi-GPU (rbl_set_comm): this context is rank comm_rank of comm_world; every full mobility product inside the
  // library becomes this rank's share of the unordered tile pairs followed by comm_fn (sum all-reduce over the ranks)
  int comm_rank = 0, comm_world = 1;
  int comm_kind = 0;                                // 0 single GPU, 1 the caller's callbacks (rbl_set_comm / rbl_set_comm_ops), 2 RCCL inside the library (rbl_comm_init_rccl)
  rbl_allreduce_fn comm_fn = nullptr;
  rbl_allgatherv_fn comm_gather_fn = nullptr;       // optional: in-place all-gather of per-rank segments (NULL: zero-padded sum all-reduce)
  void *comm_user = nullptr;
  void *comm_nccl = nullptr;                        // ncclComm_t of the native communicator
  int comm_split = 0;                               // RBL_OPT_COMM_SPLIT: 0 unordered tile pairs + all-reduce(U), 1 rows by body index + all-gather (north_star)
  std::vector<int64_t> comm_offs, comm_cnts;        // scratch of the all-gather calls
  bool fuse_done = false; // the last full product honoured sym_tune.fuse (rbl_apply_saddle_dev)
  RblNormFold pc_fold;              // transient, GMRES -> the next rbl_apply_PC_dev (body-frame tables only; see pc_can_fold)
  bool gmres_fold_norm = true;      // ... unless switched off with RBL_OPT_FUSED_KRYLOV = 0
  const double *fuse_dotV = nullptr; double *fuse_dotPart = nullptr;   // GMRES -> rbl_apply_saddle_dev: also leave the partials of V^T w ...
  int fuse_dotK = 0, fuse_dots_np = 0;                                  // ... for dotK basis vectors; answer: partials per vector (0 = not done)
  bool no_damp = false;   // transient: matvec kernels skip the damping B (preconditioned square root)
  // tuning
  size_t sym_workspace_budget = (size_t)24 << 30;   // bytes the symmetric kernel may use for its slabs
  int tune_jsplit = 0;
  int tune_variant = 0;
  RblSymTune sym_tune;
  // per-phase timings (rbl_set_timing / rbl_get_timings): hipEvent pairs recorded on the context's stream around the phases of
  // the library's own solvers; resolved (hipEventElapsedTime) when the caller asks
  bool timing_on = false;
  int timing_open = -1;                             // the phase (other than RBL_T_TOTAL) whose bracket is open: inner brackets are part of it
  bool timing_total_open = false;
  std::vector<hipEvent_t> ev_pool;
  struct TimedSpan { int phase; hipEvent_t a, b; };
  std::vector<TimedSpan> ev_spans;
  double t_ms[RBL_T_COUNT] = {0, 0, 0, 0, 0, 0};
  int64_t t_calls[RBL_T_COUNT] = {0, 0, 0, 0, 0, 0};
  bool gmres_predict = true;    // RBL_OPT_GMRES_PREDICT_CHECKS
  bool gmres_overlap = true;    // RBL_OPT_GMRES_OVERLAP_CHECK: next iteration's preconditioner enqueued before the host reads the Hessenberg column
  bool fused_krylov = true;     // RBL_OPT_FUSED_KRYLOV
  hipEvent_t ev_check = nullptr;   // recorded behind the asynchronous copy of the Hessenberg columns (overlapped convergence test)
  int gmres_last_used = 0;      // iterations of the previous converged solve: where the next one looks first (launch-bound systems)
  // lanczos
  int lanczos_max_iter = 100;
  bool lanczos_out_norm = true;  // preconditioned root: final stopping test in the Euclidean norm of the increment (RBL_OPT_LANCZOS_EUCLID_NORM)
  bool lanczos_reorth = true;   // full re-orthogonalisation of the Lanczos basis (RBL_OPT_LANCZOS_REORTH)
  double lanczos_tol = 1e-10;
  int lanczos_iters = 0;
  double lanczos_resid = 0.0;
};

// RAII bracket of one timed phase (no-op unless rbl_set_timing(ctx, 1)); a bracket opened inside another phase's bracket
// belongs to the outer one, RBL_T_TOTAL may contain the others
struct RblPhase {
  rbl_ctx *c; int phase; hipEvent_t a = nullptr; bool live = false;
  RblPhase(rbl_ctx *ctx, int ph);
  ~RblPhase();
  RblPhase(const RblPhase &) = delete;
  RblPhase &operator=(const RblPhase &) = delete;
};

RblParams rbl_make_params(double a, double eta);
int rbl_dev_init(rbl_ctx *c);
int rbl_dev_reserve(rbl_ctx *c, RblDevBuf &b, size_t bytes);
int rbl_fail(rbl_ctx *c, int code, const std::string &msg);
int rbl_hip_fail(rbl_ctx *c, hipError_t e, const char *what);
int rbl_flags_to_status(rbl_ctx *c, unsigned flags);

#define RBL_HIP(c, call)                                   \
  do {                                                     \
    hipError_t e__ = (call);                               \
    if (e__ != hipSuccess) return rbl_hip_fail(c, e__, #call); \
  } while (0)

// ----------------------------------------------------------------------------
// Kernel launchers (rbl_kernels.hip).  All enqueue on `st`, none synchronise.
// ----------------------------------------------------------------------------
size_t rbl_apply_M_part_bytes(int64_t n_blobs, int64_t nrows, int n_cu, int jsplit_override,
                              int *jsplit_out);
void rbl_launch_apply_M(hipStream_t st, const RblParams &P, bool wall, const double *d_F,
                        const double *d_r, int64_t n_blobs, int64_t row_begin,
                        int64_t row_end, double *d_out, double *d_part, int jsplit,
                        int variant, unsigned *d_err);
size_t rbl_apply_M_sym_bytes(int64_t n_blobs, int n_cu, int i_step, int nrhs, const RblSymTune &tune,
                             int *NI_out = nullptr, int *C_out = nullptr);
void rbl_apply_M_sym_kernel_name(int64_t n_blobs, int n_cu, int i_step, int nrhs, const RblSymTune &tune, bool wall, char *out, size_t len);
// nrhs = 1 or 2 vectors ([nrhs][3 n_blobs]); workspace rbl_apply_M_sym_bytes(..., nrhs)
int rbl_launch_apply_M_sym(hipStream_t st, const RblParams &P, bool wall, const double *d_F,
                           const double *d_r, int64_t n_blobs, int i_first, int i_step,
                           double *d_out, double *d_work, int n_cu, unsigned *d_err, int nrhs, const RblSymTune &tune);
size_t rbl_apply_M_mrhs_bytes(int64_t n_blobs, int n_cu);
void rbl_launch_apply_M_mrhs(hipStream_t st, const RblParams &P, bool wall, const double *d_F,
                             const double *d_r, int64_t n_blobs, int nrhs, double *d_out,
                             double *d_work, int n_cu, unsigned *d_err, int64_t ldF = 0, int64_t ldO = 0);
void rbl_launch_blob_positions(hipStream_t st, const double *d_X, const double *d_Q,
                               const double *d_cfg, int N_blb, int body_begin, int body_end,
                               double *d_out);
void rbl_launch_build_M(hipStream_t st, const RblParams &P, bool wall, bool scale_damp,
                        const double *d_r, int64_t n_blobs, double *d_M, unsigned *d_err);
void rbl_launch_pair_blocks(hipStream_t st, const RblParams &P, bool wall, int mode,
                            const double *d_ri, const double *d_rj, const int32_t *d_ii,
                            const int32_t *d_jj, int64_t n, double *d_out9, unsigned *d_err);
void rbl_launch_normal(hipStream_t st, uint64_t seed, uint64_t offset, int64_t n, double *d_out);

// dense linear algebra (rbl_dense.hip)
int rbl_launch_cholesky(hipStream_t st, double *d_M, int64_t n, bool zero_upper, unsigned *d_err,
                        double *d_work, size_t work_bytes, const RblCholAux *aux);
size_t rbl_cholesky_work_bytes(int64_t n);
size_t rbl_cholesky_batched_work_bytes(int64_t n, int batch);
int rbl_launch_cholesky_batched(hipStream_t st, double *d_M, int64_t n, int batch, int64_t strideA,
                                unsigned *d_err, double *d_Linv);
int rbl_launch_block_solve_multi(hipStream_t st, const double *d_L, int64_t n, int batch, int64_t strideA,
                                 const double *d_Linv, const double *d_in, double *d_out, int64_t vec_stride, int nv,
                                 int64_t rhs_pitch, int mode, const double *d_Q = nullptr);
int rbl_launch_block_trmv(hipStream_t st, const double *d_L, int64_t n, int batch, int64_t strideA, const double *d_in,
                          double *d_out, int64_t vec_stride);
// explicit per-body inverses for small bodies (3 N_blb <= 512): substitution sweeps become triangular matrix-vector products
bool rbl_block_inverse_fits(int64_t n);
bool rbl_block_inverse_large_fits(int64_t n);
// one matrix shared by all bodies applied to N_bod x nv vectors as a matrix-matrix product on the fp64 matrix cores
bool rbl_shared_gemm_fits(int64_t n);
int rbl_launch_shared_gemm(hipStream_t st, const double *d_A, int64_t n, int64_t lda, int tri, const double *d_in, double *d_out,
                           int64_t vec_stride, int64_t rhs_pitch, int nbod, int nv, const double *d_Q, int rot);
size_t rbl_block_inverse_bytes(int64_t n, int batch);
int64_t rbl_block_inverse_ld(int64_t n);
int rbl_launch_block_inverse(hipStream_t st, const double *d_L, int64_t n, int batch, int64_t strideA, const double *d_Linv,
                             double *d_X);
int rbl_launch_block_inv_apply(hipStream_t st, const double *d_X, int64_t n, int batch, const double *d_in, double *d_out,
                               int64_t vec_stride, int nv, int64_t rhs_pitch, int mode, double *d_tmp,
                               const double *d_Q = nullptr, int f32 = 0);
// explicit inverses of large bodies (3 N_blb > 512) with the factorisation's own MFMA kernels on an augmented matrix
size_t rbl_block_inverse_large_aug_bytes(int64_t n, int batch, int *chunk_out);
int rbl_launch_block_inverse_large(hipStream_t st, const double *d_L, int64_t n, int batch, int64_t strideA, const double *d_Linv,
                                   double *d_X, float *d_Xf, double *d_aug);
// per-body factors and explicit inverses of large bodies as ONE dataflow launch over 128 x 128 tiles (rbl_tilechol.hip)
bool rbl_tile_cholesky_fits(int64_t n);
size_t rbl_tile_cholesky_work_bytes(int64_t n, int batch);
int rbl_launch_tile_cholesky(hipStream_t st, double *d_M, int64_t n, int batch, int64_t strideA, unsigned *d_err, double *d_Linv,
                             double *d_X, float *d_Xf, void *d_work, int n_cu);
int rbl_launch_block_trmv_small(hipStream_t st, const double *d_L, int64_t n, int batch, int64_t strideA, const double *d_in,
                                double *d_out, int64_t vec_stride, const double *d_Q);
void rbl_launch_rotate_bodies(hipStream_t st, const double *d_Q, const double *d_in, double *d_out, int N_blb, int batch, int nv,
                              int64_t rhs_pitch, int transpose);
void rbl_launch_trmv_lower(hipStream_t st, const double *d_L, int64_t n, const double *d_W,
                           double *d_out, double *d_part);
size_t rbl_trmv_part_bytes(int64_t n);

// small vector kernels (rbl_kernels.hip) used by Lanczos
void rbl_launch_dot2(hipStream_t st, const double *x, const double *y, const double *z,
                     int64_t n, double *d_out2);  // out[0]=x.y out[1]=x.z (z may be null)
int rbl_gmres_max_vectors(void);
size_t rbl_gmres_part_doubles(void);
int rbl_gmres_p1_capacity(void);          // partial sums per vector the first Gram-Schmidt pass may be handed
// skip_norm: leave out the normalisation launch; *norm_part / *norm_np then say where the partial sums of |w|^2 lie (RblNormFold)
void rbl_launch_arnoldi_step(hipStream_t st, const double *V, int64_t n, int k, double *w, double *Hcol, double *vnext,
                             double *part, int fused_np = 0, bool skip_norm = false, const double **norm_part = nullptr,
                             int *norm_np = nullptr);
size_t rbl_lanczos_part_doubles(void);
void rbl_launch_lanczos_init(hipStream_t st, int64_t n, const double *d_W, double *wnorm_out, double *V0,
                             double *part, int nvec = 1, int64_t scal_stride = 0);
void rbl_launch_lanczos_step(hipStream_t st, int64_t n, double *u, const double *v, const double *vprev,
                             const double *beta_prev, double *alpha_out, double *beta_out, double *vnext,
                             double *part, int nvec = 1, int64_t vec_stride = 0, int64_t scal_stride = 0);
void rbl_launch_lanczos_step_reorth(hipStream_t st, int64_t n, int k, double *u, const double *V, double *vnext, double *alpha_out,
                                    double *beta_out, int64_t scal_stride, double *hcol, int64_t hcol_stride, double *part, int nvec);
void rbl_launch_lanczos_combine(hipStream_t st, int64_t n, const double *V, const double *coef, int m,
                                double *out, int64_t stride = 0);
void rbl_launch_lanczos_combine_xd(hipStream_t st, int64_t n, const double *V, int64_t stride, int64_t vsep, const double *coef,
                                   int64_t csep, int m, double *zx, double *zd, int64_t osep, int nvec);
constexpr int RBL_SQNORM_BLOCKS = 128;
int rbl_launch_damp_sqnorm(hipStream_t st, const RblParams &P, const double *d_r, int64_t n_blobs, double *o, int64_t pitch, int nv,
                           double *part);
void rbl_launch_axpby(hipStream_t st, int64_t n, double a, const double *x, double b,
                      const double *y, double *out);
void rbl_launch_rhs_combine(hipStream_t st, int64_t n, const double *x, double a, const double *y, double b, const double *z,
                            const double *w, double *out);
void rbl_launch_scale_by_damp(hipStream_t st, const RblParams &P, const double *d_r,
                              int64_t n_blobs, const double *in, double *out);

// whole GMRES solve of a small system in one kernel (rbl_small.hip)
bool rbl_gmres_small_fits(int N_blb, int N_bod, int max_iter, bool block_pc);
size_t rbl_gmres_small_work_doubles(int N_blb, int N_bod, int max_iter);
int rbl_launch_gmres_small(hipStream_t st, const RblParams &P, bool wall, const double *dX, const double *dQ, const double *dcfg,
                           int N_blb, int N_bod, const double *d_rhs, const double *d_x0, double *d_x, int max_iter, double rtol,
                           double fsign, double *d_work, double *d_scal, unsigned *d_err);

// per-body geometric operators on the device (rbl_body_dev.hip)
void rbl_launch_body_geom(hipStream_t st, const double *dX, const double *dQ, const double *dcfg,
                          int N_blb, int64_t N, double *d_lever, double *d_pos);
void rbl_launch_K_x_U(hipStream_t st, const double *d_lever, const double *d_U, int N_blb, int64_t N,
                      double *d_out, const double *d_sub, double alpha);
void rbl_launch_KT_x_Lam(hipStream_t st, const double *d_lever, const double *d_lam, int N_blb, int N_bod,
                         double *d_out);
void rbl_launch_pc_diag_build(hipStream_t st, const RblParams &P, bool wall, const double *d_lever,
                              const double *d_pos, int N_blb, int N_bod, double *d_invM2, double *d_NL,
                              unsigned *d_err);
void rbl_launch_pc_diag_apply(hipStream_t st, const double *d_lever, const double *d_invM2, const double *d_NL,
                              int N_blb, int N_bod, const double *d_in, double *d_out, double fsign, const RblNormFold *fold = nullptr);
void rbl_launch_pc_block_ninv(hipStream_t st, const double *d_cols, int N_bod, double *d_NL, unsigned *d_err, int b_begin = 0,
                              int b_end = -1);
void rbl_launch_pc_block_tail(hipStream_t st, const double *d_lever, const double *d_y1, const double *d_MK, int64_t stride,
                              const double *d_NL, const double *d_F, int N_blb, int b_begin, int b_count, double fsign,
                              double *d_U, double *d_lam, double *d_ktl, const RblNormFold *fold = nullptr, const double *d_win = nullptr);
void rbl_launch_saddle_tail(hipStream_t st, const double *d_lever, const double *d_U, int N_blb, int64_t N, int N_bod,
                            double *d_out, const double *d_sub, const double *d_ktl);
void rbl_launch_bf_tables(hipStream_t st, const double *d_XU, const double *d_cfg, int64_t n, double *d_Minv, double *d_MK,
                          double *d_NL, unsigned *d_err);
int rbl_launch_pc_bodyframe(hipStream_t st, const double *d_Minv, const double *d_MK, const double *d_NL, const double *d_cfg,
                            const double *d_Q, int64_t n, int b_begin, int b_count, const double *d_in, int64_t n3, double fsign,
                            double *d_out, double *d_ktl, double *d_y1, int gemm, const RblNormFold *fold = nullptr);
bool rbl_pc_bodyframe_folds(int b_count, int gemm);
void rbl_launch_unit_U(hipStream_t st, int N_bod, int c, double *d_U);
// two-level factor of the preconditioned Lanczos root (rbl_body_dev.hip: k_tl_*)
void rbl_launch_tl_unit(hipStream_t st, int64_t n3, double *d_out);
void rbl_launch_tl_orth(hipStream_t st, double *d_Z, int64_t n3, int N_blb, int N_bod, double *d_Cb, unsigned *d_err);
void rbl_launch_tl_E(hipStream_t st, const double *d_Cs, const double *d_Cb, int N_bod, double *d_A);
void rbl_launch_tl_qt(hipStream_t st, const double *d_Q, int64_t n3, int N_blb, int N_bod, const double *d_w, int64_t wpitch, int nvec,
                      double *d_t, int64_t tpitch);
void rbl_launch_tl_eaddq(hipStream_t st, const double *d_Q, int64_t n3, int N_blb, int N_bod, const double *d_Op, int64_t nt, int64_t ld,
                         int kind, const double *d_t, int64_t tpitch, const double *d_w, double *d_wo, int64_t wpitch, int nvec);
void rbl_launch_tl_addq(hipStream_t st, const double *d_Q, int64_t n3, int N_blb, const double *d_s, const double *d_t, int64_t tpitch,
                        const double *d_w, double *d_wo, int64_t wpitch, int nvec);
void rbl_launch_build_M_batched(hipStream_t st, const RblParams &P, bool wall, const double *d_r,
                                int64_t n_blobs, int batch, double *d_M, int64_t strideM, unsigned *d_err, int lower_tiles = 0);

// ---- ONE wave-wide sum of doubles, in ONE order --------------------------------------------------------------------------------
// The bitwise-equal-everywhere claims of the fused reductions (RblNormFold: every wave of every workgroup must get the SAME |w| from
// the same partial sums; k_reduce_sym's Gram-Schmidt partials; k_arnoldi_upd) rest on every user adding in the same order: the four
// 16-lane rows by DPP (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror), then the rows' sums as (r0 + r1) + (r2 + r3)
// through scalar reads.  Three copies of this lived in two translation units; they are these two functions now.
#if defined(__HIPCC__)
template <int CTRL>
__device__ __forceinline__ double rbl_dpp_row(double v)      // DPP move of both halves of a double inside a 16-lane row
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rbl_row_sum16(double v)    // every lane of a 16-lane row gets the row's sum
{
  v += rbl_dpp_row<0xB1>(v);       // quad_perm [1,0,3,2]
  v += rbl_dpp_row<0x4E>(v);       // quad_perm [2,3,0,1]
  v += rbl_dpp_row<0x141>(v);      // row_half_mirror
  v += rbl_dpp_row<0x140>(v);      // row_mirror
  return v;
}
__device__ __forceinline__ double rbl_wave_sum64(double v)   // every lane gets the sum over the 64 lanes
{
  v = rbl_row_sum16(v);
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
  const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
  return (r0 + r1) + (r2 + r3);
}
#endif

#pragma GCC visibility pop
