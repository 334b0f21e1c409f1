// rbl_api_internal.hpp -- internals shared by the translation units that implement include/rbl.h:
//   rbl_core.hip      context, errors, device buffers, copies, timings, parameters / configuration
//   rbl_options.hip   named options (rbl_set_option / rbl_get_option)
//   rbl_comm.hip      multi-GPU communicator: RCCL inside the library, or the caller's callbacks
//   rbl_products.hip  mobility products (kernel choice, launches), positions, dense entry points
//   rbl_bodies.hip    K operators, preconditioners, per-body factors, saddle operator
//   rbl_roots.hip     M^{1/2} W: dense Cholesky path and the Lanczos roots
//   rbl_solvers.hip   GMRES on the saddle operator
//   rbl_steps.hip     whole time steps, random finite differences
// None of these symbols is exported from librbl.so.
#pragma once
#include "rbl_internal.hpp"

#pragma GCC visibility push(hidden)

// ---- rbl_core.hip -------------------------------------------------------------------------------------------------
int need_params(rbl_ctx *c);
int need_config(rbl_ctx *c);
int need_K(rbl_ctx *c);                       // (rbl_bodies.hip)
// Krylov coefficients (<= 512 doubles, slot 0 or 1) to the device through a pinned buffer: no stream drain
int upload_coef(rbl_ctx *c, double *d_dst, const double *src, int count, int slot);
int copy_h2d(rbl_ctx *c, void *dst, const void *src, size_t bytes);   // synchronous for large pageable sources
int copy_d2h(rbl_ctx *c, void *dst, const void *src, size_t bytes);
int read_back(rbl_ctx *c, void *dst, const void *d_src, size_t bytes);   // small device -> host read the host needs NOW
int finish_and_check(rbl_ctx *c);             // drain the stream, read + clear the latched device flags
RblParams ctx_params(const rbl_ctx *c);

// ---- rbl_comm.hip -------------------------------------------------------------------------------------------------
bool comm_on(const rbl_ctx *c);
void comm_body_range(const rbl_ctx *c, int *b0, int *b1);              // this rank's bodies
void comm_body_range_of(const rbl_ctx *c, int rank, int *b0, int *b1);
int comm_allreduce(rbl_ctx *c, double *d_buf, int64_t count);
// Complete per-body results in place: every rank has written the entries of ITS bodies; with a native communicator / an
// all-gather callback the owners' segments are gathered, otherwise the caller must have ZEROED what it does not own
// (comm_gather_needs_zero) and a sum all-reduce over the span completes it.
bool comm_gather_needs_zero(const rbl_ctx *c);
// per_body doubles per body (body-major, first body at offset base) of nvec vectors `pitch` apart in d_buf
int comm_allgather_bodies(rbl_ctx *c, double *d_buf, int64_t base, int64_t per_body, int nvec, int64_t pitch);
// two per-body parts in ONE fused collective (the preconditioner's [lambda ; U]; lever arms + positions)
int comm_allgather_bodies2(rbl_ctx *c, double *d_buf1, int64_t base1, int64_t per_body1, double *d_buf2, int64_t base2, int64_t per_body2);
// rows [row_bounds[r], row_bounds[r + 1]) x `width` doubles of every rank r, in place in one vector (the row split's product)
int comm_allgather_rows(rbl_ctx *c, double *d_buf, const int64_t *row_bounds, int64_t width);
void comm_release(rbl_ctx *c);                // destroy a native communicator (rbl_destroy)

// ---- rbl_products.hip ---------------------------------------------------------------------------------------------
int apply_M_enqueue(rbl_ctx *c, bool wall, const double *d_F, const double *d_r, int64_t nbl, int64_t row_begin, int64_t row_end,
                    double *d_out);
int apply_M_multi_enqueue(rbl_ctx *c, bool wall, const double *d_F, const double *d_r, int64_t nbl, int nrhs, double *d_out,
                          int64_t ldF = 0, int64_t ldO = 0);
int apply_PC_multi_dev(rbl_ctx *c, const double *d_in, double *d_out, double *d_scratch, int nv, int64_t pitch);
int ensure_xq_dev(rbl_ctx *c);
int positions_dev(rbl_ctx *c, int b0, int b1, double *d_out);

// ---- rbl_bodies.hip -----------------------------------------------------------------------------------------------
int sync_bodies(rbl_ctx *c);
int pc_block_factors(rbl_ctx *c, int b0 = 0, int b1 = -1);
bool bf_on(const rbl_ctx *c);
int bf_build(rbl_ctx *c);
int blk_prepare(rbl_ctx *c, int b0, int b1);
int blk_solve(rbl_ctx *c, int b0, int nbo, const double *in, double *out, int nv, int64_t pitch, int mode, bool allow_f32 = true);
int blk_trmv(rbl_ctx *c, int b0, int nbo, const double *in, double *out);
bool pc_can_fold(rbl_ctx *c);
int blk_trmv_multi(rbl_ctx *c, int b0, int nbo, const double *in, double *out, int nv, int64_t pitch);

// ---- rbl_roots.hip ------------------------------------------------------------------------------------------------
int tl_build(rbl_ctx *c);
int tl_apply(rbl_ctx *c, const double *w, double *wo, int nvec, int64_t pitch, int op);
int mhalf_dev_multi(rbl_ctx *c, const double *d_r, int64_t nbl, const double *d_W, int nvec, int method, double *d_out);

// ---- rbl_steps.hip ------------------------------------------------------------------------------------------------
int m_rfd_core(rbl_ctx *c, const double *d_W, const double *Wh, double delta, double *d_out, double *d_r, double *d_work);

#pragma GCC visibility pop
