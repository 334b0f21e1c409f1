// rbl_options.hip -- named per-context options (include/rbl.h: rbl_set_option / rbl_get_option / rbl_option_info).
//
// One table row per option: key, name, range (and, where not every value of the range selects something, a predicate), default,
// and what else a change invalidates (cached factors, preconditioner state).  An unknown key or an inadmissible value is
// RBL_ERR_ARG and changes nothing.
#include <cstring>

#include "rbl_api_internal.hpp"

namespace {

struct OptRow {
  int key;
  const char *name;
  int64_t lo, hi, dflt;
  int64_t (*get)(const rbl_ctx *);
  void (*set)(rbl_ctx *, int64_t);
  bool (*ok)(int64_t) = nullptr;        // values of [lo, hi] that select something (nullptr: all of them)
};

void drop_factors(rbl_ctx *c) { c->dev_blk_valid = false; c->blk_inv_valid = false; c->bf_valid = false; c->dev_pc_valid = false; }

const OptRow kOptions[] = {
    {RBL_OPT_MATVEC_KERNEL, "matvec_kernel", 0, 3, 0, [](const rbl_ctx *c) -> int64_t { return c->tune_variant; },
     [](rbl_ctx *c, int64_t v) { c->tune_variant = (int)v; }},
    {RBL_OPT_ORDERED_JSPLIT, "ordered_jsplit", 0, 4096, 0, [](const rbl_ctx *c) -> int64_t { return c->tune_jsplit; },
     [](rbl_ctx *c, int64_t v) { c->tune_jsplit = (int)v; }},
    {RBL_OPT_SYM_CHUNK, "sym_chunk", 0, 4096, 0, [](const rbl_ctx *c) -> int64_t { return c->sym_tune.chunk; },
     [](rbl_ctx *c, int64_t v) { c->sym_tune.chunk = (int)v; }},
    {RBL_OPT_SYM_ROWS_PER_LANE, "sym_rows_per_lane", 0, 2, 0, [](const rbl_ctx *c) -> int64_t { return c->sym_tune.ni1; },
     [](rbl_ctx *c, int64_t v) { c->sym_tune.ni1 = (int)v; }},
    {RBL_OPT_SYM2_ROWS_PER_LANE, "sym2_rows_per_lane", 0, 2, 0, [](const rbl_ctx *c) -> int64_t { return c->sym_tune.ni2; },
     [](rbl_ctx *c, int64_t v) { c->sym_tune.ni2 = (int)v; }},
    {RBL_OPT_SYM_WAVES, "sym_waves", 0, 4, 0, [](const rbl_ctx *c) -> int64_t { return c->sym_tune.sw; },
     [](rbl_ctx *c, int64_t v) { c->sym_tune.sw = (int)v; }, [](int64_t v) { return v == 0 || v == 1 || v == 4; }},
    {RBL_OPT_SYM_WORK_QUEUE, "sym_work_queue", 0, 1, 1, [](const rbl_ctx *c) -> int64_t { return c->sym_tune.queue < 0 ? 0 : 1; },
     [](rbl_ctx *c, int64_t v) { c->sym_tune.queue = v ? 0 : -1; }},
    {RBL_OPT_SYM_WAVE_UNITS, "sym_wave_units", 0, 1, 1, [](const rbl_ctx *c) -> int64_t { return c->sym_tune.wave_units < 0 ? 0 : 1; },
     [](rbl_ctx *c, int64_t v) { c->sym_tune.wave_units = v ? 0 : -1; }},
    {RBL_OPT_GMRES_PC_SIGN_FIX, "gmres_pc_sign_fix", 0, 1, 1, [](const rbl_ctx *c) -> int64_t { return c->gmres_pc_sign_fix; },
     [](rbl_ctx *c, int64_t v) { c->gmres_pc_sign_fix = v != 0; }},
    {RBL_OPT_GMRES_ONE_KERNEL, "gmres_one_kernel", 0, 1, 1, [](const rbl_ctx *c) -> int64_t { return c->gmres_small; },
     [](rbl_ctx *c, int64_t v) { c->gmres_small = v != 0; }},
    {RBL_OPT_GMRES_PREDICT_CHECKS, "gmres_predict_checks", 0, 1, 1, [](const rbl_ctx *c) -> int64_t { return c->gmres_predict; },
     [](rbl_ctx *c, int64_t v) { c->gmres_predict = v != 0; c->gmres_last_used = 0; }},
    {RBL_OPT_GMRES_OVERLAP_CHECK, "gmres_overlap_check", 0, 1, 1, [](const rbl_ctx *c) -> int64_t { return c->gmres_overlap; },
     [](rbl_ctx *c, int64_t v) { c->gmres_overlap = v != 0; }},
    {RBL_OPT_RELAXED_KRYLOV, "relaxed_krylov", 0, 2, 0, [](const rbl_ctx *c) -> int64_t { return c->gmres_relax; },
     [](rbl_ctx *c, int64_t v) { c->gmres_relax = (int)v; }},
    {RBL_OPT_RELAXED_ALWAYS, "relaxed_always", 0, 1, 0, [](const rbl_ctx *c) -> int64_t { return c->force_relaxed; },
     [](rbl_ctx *c, int64_t v) { c->force_relaxed = v != 0; }},
    {RBL_OPT_RELAXED_GAP_RATIO, "relaxed_gap_ratio", 0, 64, 0, [](const rbl_ctx *c) -> int64_t { return c->sym_tune.gap_ratio; },
     [](rbl_ctx *c, int64_t v) { c->sym_tune.gap_ratio = (int)v; }},
    {RBL_OPT_BLOCK_EXPLICIT_SMALL, "block_explicit_small", 0, 1, 1, [](const rbl_ctx *c) -> int64_t { return c->blk_explicit; },
     [](rbl_ctx *c, int64_t v) { c->blk_explicit = v != 0; drop_factors(c); }},
    {RBL_OPT_BLOCK_EXPLICIT_LARGE, "block_explicit_large", 0, 2, 2, [](const rbl_ctx *c) -> int64_t { return c->blk_large; },
     [](rbl_ctx *c, int64_t v) { c->blk_large = (int)v; drop_factors(c); }},
    {RBL_OPT_BLOCK_INVERSE_F32, "block_inverse_f32", 0, 1, 0, [](const rbl_ctx *c) -> int64_t { return c->blk_f32; },
     [](rbl_ctx *c, int64_t v) { c->blk_f32 = v != 0; c->dev_blk_valid = false; c->blk_inv_valid = false; c->dev_pc_valid = false; }},
    {RBL_OPT_BODYFRAME_FACTOR, "bodyframe_factor", 0, 1, 1, [](const rbl_ctx *c) -> int64_t { return c->blk_bodyframe; },
     [](rbl_ctx *c, int64_t v) { c->blk_bodyframe = v != 0; drop_factors(c); }},
    {RBL_OPT_BODYFRAME_WALL_APPROX, "bodyframe_wall_approx", 0, 1, 0, [](const rbl_ctx *c) -> int64_t { return c->bf_wall_approx; },
     [](rbl_ctx *c, int64_t v) { c->bf_wall_approx = v != 0; c->dev_pc_valid = false; c->dev_blk_valid = false; c->blk_inv_valid = false; }},
    {RBL_OPT_BLOCK_REFRESH, "block_refresh", 1, 1 << 20, 1, [](const rbl_ctx *c) -> int64_t { return c->blk_refresh; },
     [](rbl_ctx *c, int64_t v) { c->blk_refresh = (int)v; c->blk_age = 0; c->dev_blk_valid = false; }},
    {RBL_OPT_LANCZOS_TWO_LEVEL, "lanczos_two_level", 0, 1, 1, [](const rbl_ctx *c) -> int64_t { return c->tl_on; },
     [](rbl_ctx *c, int64_t v) { c->tl_on = v != 0; c->tl_valid = false; }},
    {RBL_OPT_LANCZOS_EUCLID_NORM, "lanczos_euclid_norm", 0, 1, 1, [](const rbl_ctx *c) -> int64_t { return c->lanczos_out_norm; },
     [](rbl_ctx *c, int64_t v) { c->lanczos_out_norm = v != 0; }},
    {RBL_OPT_LANCZOS_REORTH, "lanczos_reorth", 0, 1, 1, [](const rbl_ctx *c) -> int64_t { return c->lanczos_reorth; },
     [](rbl_ctx *c, int64_t v) { c->lanczos_reorth = v != 0; }},
    {RBL_OPT_NO_DAMP, "no_damp", 0, 1, 0, [](const rbl_ctx *c) -> int64_t { return c->no_damp; },
     [](rbl_ctx *c, int64_t v) { c->no_damp = v != 0; }},
    {RBL_OPT_COMM_SPLIT, "comm_split", 0, 1, 0, [](const rbl_ctx *c) -> int64_t { return c->comm_split; },
     [](rbl_ctx *c, int64_t v) { c->comm_split = (int)v; c->dev_bodies_valid = false; }},
    {RBL_OPT_SHARED_GEMM, "shared_gemm", 0, 1, 1, [](const rbl_ctx *c) -> int64_t { return c->shared_gemm; },
     [](rbl_ctx *c, int64_t v) { c->shared_gemm = v != 0; }},
    {RBL_OPT_TWO_LEVEL_REFRESH, "two_level_refresh", 1, 1 << 20, 1, [](const rbl_ctx *c) -> int64_t { return c->tl_refresh; },
     [](rbl_ctx *c, int64_t v) { c->tl_refresh = (int)v; c->tl_age = 0; c->tl_valid = false; }},
    {RBL_OPT_BLOCK_TILE_FACTOR, "block_tile_factor", 0, 1, 1, [](const rbl_ctx *c) -> int64_t { return c->blk_tile; },
     [](rbl_ctx *c, int64_t v) { c->blk_tile = v != 0; drop_factors(c); c->tl_valid = false; }},
    {RBL_OPT_BLOCK_SOLVE_PIPE, "block_solve_pipe", 0, 1, 1, [](const rbl_ctx *c) -> int64_t { return c->blk_pipe; },
     [](rbl_ctx *c, int64_t v) { c->blk_pipe = v != 0; }},
    {RBL_OPT_COMM_FORCE_STAGED, "comm_force_staged", 0, 1, 0, [](const rbl_ctx *c) -> int64_t { return c->comm_force_staged; },
     [](rbl_ctx *c, int64_t v) { c->comm_force_staged = v != 0; }},
    {RBL_OPT_FUSED_KRYLOV, "fused_krylov", 0, 1, 1, [](const rbl_ctx *c) -> int64_t { return c->fused_krylov; },
     [](rbl_ctx *c, int64_t v) { c->fused_krylov = v != 0; }},
};

const OptRow *find_option(int key)
{
  for (const OptRow &r : kOptions)
    if (r.key == key) return &r;
  return nullptr;
}

}  // namespace

static_assert(sizeof(kOptions) / sizeof(kOptions[0]) == RBL_OPT_COUNT - 1, "every RBL_OPT_* key of include/rbl.h needs a row here");

int rbl_set_option(rbl_ctx *c, int option, int64_t value)
{
  if (!c) return RBL_ERR_ARG;
  const OptRow *r = find_option(option);
  if (!r) return rbl_fail(c, RBL_ERR_ARG, "set_option: unknown option key " + std::to_string(option));
  if (value < r->lo || value > r->hi)
    return rbl_fail(c, RBL_ERR_ARG, std::string("set_option: ") + r->name + " takes " + std::to_string(r->lo) + " .. " + std::to_string(r->hi));
  if (r->ok && !r->ok(value))
    return rbl_fail(c, RBL_ERR_ARG, std::string("set_option: ") + r->name + " = " + std::to_string(value) + " selects nothing (see include/rbl.h)");
  r->set(c, value);
  return RBL_OK;
}

int rbl_get_option(const rbl_ctx *c, int option, int64_t *value)
{
  if (!c || !value) return RBL_ERR_ARG;
  const OptRow *r = find_option(option);
  if (!r) return RBL_ERR_ARG;
  *value = r->get(c);
  return RBL_OK;
}

int rbl_option_info(int option, const char **name, int64_t *lo, int64_t *hi, int64_t *dflt)
{
  const OptRow *r = find_option(option);
  if (!r) return RBL_ERR_ARG;
  if (name) *name = r->name;
  if (lo) *lo = r->lo;
  if (hi) *hi = r->hi;
  if (dflt) *dflt = r->dflt;
  return RBL_OK;
}

int rbl_option_key(const char *name)
{
  if (!name) return 0;
  for (const OptRow &r : kOptions)
    if (std::strcmp(r.name, name) == 0) return r.key;
  return 0;
}

int rbl_set_block_refresh(rbl_ctx *c, int every)
{
  if (!c || every < 1) return rbl_fail(c, RBL_ERR_ARG, "set_block_refresh: every >= 1");
  return rbl_set_option(c, RBL_OPT_BLOCK_REFRESH, every);
}

// transient switch: the matvec entry points skip the damping B (plain, wall-corrected M) while it is on
int rbl_set_no_damp(rbl_ctx *c, int on) { return c ? rbl_set_option(c, RBL_OPT_NO_DAMP, on != 0) : RBL_ERR_ARG; }
