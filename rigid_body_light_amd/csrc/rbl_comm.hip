// rbl_comm.hip -- the multi-GPU communicator of a context (include/rbl.h "multi-GPU inside the library's own solvers").
//
// One process per GPU.  Two ways to give a context its collectives:
//   * rbl_comm_init_rccl: RCCL inside the library.  ncclCommInitRank from a unique id the host distributes by whatever means
//     it has (MPI, a file, torch.distributed), then ncclAllReduce / ncclAllGather (ragged segments through a padded staging buffer) on the CONTEXT'S
//     stream -- a C or C++ host (which is what the reference is, c_rigid_obj.cpp:997-1027) runs N GPUs without Python, and
//     no host frame sits between two products of a solve.  librccl is opened at run time (dlopen by soname): in a process
//     that already holds PyTorch's bundled copy that very copy is found, so the process keeps ONE RCCL and one HIP runtime;
//     a host without PyTorch gets /opt/rocm's through librbl's own RUNPATH.  Nothing links against it at build time.
//   * rbl_set_comm / rbl_set_comm_ops: the caller's callbacks (torch.distributed over gloo in the CPU-staged rehearsals).
//
// Who owns what: bodies are split contiguously by body index, sizes differing by at most one (the partition of
// rigid_body_light_amd/dist.py).  Per-body results are written by their owner and completed by an in-place all-gather of
// the owners' segments; only when neither a native communicator nor an all-gather callback exists does the library fall
// back to zero-padding + sum all-reduce.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

#include "rbl_api_internal.hpp"

static_assert(RBL_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "rbl.h: RBL_COMM_ID_BYTES must be the size of ncclUniqueId");

namespace {

struct RcclApi {
  void *handle = nullptr;
  bool tried = false;
  std::string error;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_rccl;
std::mutex g_rccl_mutex;

template <class F> bool load_sym(void *h, const char *name, F &fn)
{
  fn = reinterpret_cast<F>(dlsym(h, name));
  return fn != nullptr;
}

// the process-wide RCCL entry points, or nullptr (g_rccl.error says why)
const RcclApi *rccl()
{
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.tried) return g_rccl.handle ? &g_rccl : nullptr;
  g_rccl.tried = true;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) { const char *e = dlerror(); g_rccl.error = std::string("cannot open librccl: ") + (e ? e : "?"); return nullptr; }
  const bool ok = load_sym(h, "ncclGetUniqueId", g_rccl.GetUniqueId) && load_sym(h, "ncclCommInitRank", g_rccl.CommInitRank) &&
                  load_sym(h, "ncclCommDestroy", g_rccl.CommDestroy) && load_sym(h, "ncclAllReduce", g_rccl.AllReduce) &&
                  load_sym(h, "ncclAllGather", g_rccl.AllGather) &&
                  load_sym(h, "ncclGroupStart", g_rccl.GroupStart) && load_sym(h, "ncclGroupEnd", g_rccl.GroupEnd) &&
                  load_sym(h, "ncclGetErrorString", g_rccl.GetErrorString);
  if (!ok) { g_rccl.error = "librccl lacks an expected entry point"; dlclose(h); return nullptr; }
  g_rccl.handle = h;
  return &g_rccl;
}

int nccl_fail(rbl_ctx *c, const RcclApi *R, ncclResult_t r, const char *what)
{
  return rbl_fail(c, RBL_ERR_COMM, std::string("RCCL error: ") + (R && R->GetErrorString ? R->GetErrorString(r) : "?") + " in " + what);
}

#define RBL_NCCL(c, R, call)                                        \
  do {                                                              \
    ncclResult_t r__ = (call);                                      \
    if (r__ != ncclSuccess) return nccl_fail(c, R, r__, #call);     \
  } while (0)

}  // namespace

bool comm_on(const rbl_ctx *c) { return c->comm_kind != 0; }

void comm_body_range_of(const rbl_ctx *c, int r, int *b0, int *b1)
{
  const int nb = c->S.N_bod, base = nb / c->comm_world, rem = nb % c->comm_world;
  *b0 = r * base + (r < rem ? r : rem);
  *b1 = *b0 + base + (r < rem ? 1 : 0);
}

void comm_body_range(const rbl_ctx *c, int *b0, int *b1) { comm_body_range_of(c, c->comm_rank, b0, b1); }

int comm_allreduce(rbl_ctx *c, double *d_buf, int64_t count)
{
  RblPhase ph(c, RBL_T_COLLECTIVE);
  if (c->comm_kind == 2) {
    const RcclApi *R = rccl();
    if (!R) return rbl_fail(c, RBL_ERR_COMM, g_rccl.error);
    RBL_NCCL(c, R, R->AllReduce(d_buf, d_buf, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)c->comm_nccl, c->stream));
    return RBL_OK;
  }
  if (!c->comm_fn || c->comm_fn(c->comm_user, d_buf, count)) return rbl_fail(c, RBL_ERR_COMM, "all-reduce callback failed");
  return RBL_OK;
}

bool comm_gather_needs_zero(const rbl_ctx *c) { return c->comm_kind == 1 && c->comm_gather_fn == nullptr; }

// ---- in-place all-gather of per-rank segments: ONE RCCL shape ----------------------------------------------------------------
// Equal, back-to-back segments are an in-place ncclAllGather.  Ragged ones (N_bod % world != 0, uneven row bounds) take the SAME
// collective through a staging buffer: every rank's segment padded to the largest share, own part packed in, one in-place
// ncclAllGather of the padded slots, everything unpacked to where it belongs.  Round 4 issued a group of in-place ncclBroadcasts
// there -- a branch that no run had ever executed (a communicator of one rank is always "even"); the staged form is exercised
// with one rank by RBL_OPT_COMM_FORCE_STAGED.  Pack and unpack are one small kernel each per set.
namespace {

constexpr int SEG_MAX_RANKS = 64;
struct SegCopy {
  int W;
  long slot;                     // doubles per padded slot of the staging buffer
  long off[SEG_MAX_RANKS];       // offsets of the ranks' segments in the vector
  long cnt[SEG_MAX_RANKS];
};

// unpack = 1: vec[off[r] + e] = stage[r * slot + e] for every rank r (blockIdx.y) ; unpack = 0: the reverse, rank `only` alone
__global__ __launch_bounds__(256) void k_seg_copy(double *__restrict__ vec, double *__restrict__ stage, SegCopy S, int unpack, int only)
{
  const int r = unpack ? (int)blockIdx.y : only;
  const long n = S.cnt[r];
  double *v = vec + S.off[r], *s = stage + (size_t)r * (size_t)S.slot;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    if (unpack) v[e] = s[e]; else s[e] = v[e];
  }
}

}  // namespace

// `nsets` sets of segments: in set s rank r owns bufs[s][offs[s * world + r] .. + cnts[s * world + r]); every set is
// completed in place, all of them in ONE fused RCCL group.  Fallback (callbacks without an all-gather): the caller has
// zeroed what it does not own, one sum all-reduce per set over the span of its segments.
static int comm_allgatherv(rbl_ctx *c, int nsets, double *const *bufs, const int64_t *offs, const int64_t *cnts)
{
  RblPhase ph(c, RBL_T_COLLECTIVE);
  const int W = c->comm_world;
  if (c->comm_kind == 2) {
    const RcclApi *R = rccl();
    if (!R) return rbl_fail(c, RBL_ERR_COMM, g_rccl.error);
    ncclComm_t comm = (ncclComm_t)c->comm_nccl;
    // which sets are staged, and where in the staging buffer
    std::vector<char> staged((size_t)nsets, 0);
    std::vector<int64_t> slot((size_t)nsets, 0), base((size_t)nsets, 0);
    int64_t stage_doubles = 0;
    for (int s = 0; s < nsets; ++s) {
      const int64_t *o = offs + (size_t)s * W, *n = cnts + (size_t)s * W;
      bool even = !c->comm_force_staged;                   // equal, back-to-back segments: the plain in-place all-gather
      for (int r = 0; r < W; ++r) even = even && n[r] == n[0] && o[r] == o[0] + (int64_t)r * n[0];
      if (even) continue;
      if (W > SEG_MAX_RANKS) return rbl_fail(c, RBL_ERR_COMM, "ragged all-gather: more than 64 ranks");
      staged[(size_t)s] = 1;
      for (int r = 0; r < W; ++r) slot[(size_t)s] = std::max(slot[(size_t)s], n[r]);
      base[(size_t)s] = stage_doubles;
      stage_doubles += slot[(size_t)s] * W;
    }
    int rc;
    if (stage_doubles > 0 && (rc = rbl_dev_reserve(c, c->d_commStage, sizeof(double) * (size_t)stage_doubles))) return rc;
    double *stage = (double *)c->d_commStage.p;
    auto seg = [&](int s) {
      SegCopy S; S.W = W; S.slot = (long)slot[(size_t)s];
      for (int r = 0; r < W; ++r) { S.off[r] = (long)offs[(size_t)s * W + r]; S.cnt[r] = (long)cnts[(size_t)s * W + r]; }
      return S;
    };
    for (int s = 0; s < nsets; ++s)                        // pack the own segment of every staged set
      if (staged[(size_t)s] && cnts[(size_t)s * W + c->comm_rank] > 0) {
        const unsigned g = (unsigned)std::min<int64_t>((cnts[(size_t)s * W + c->comm_rank] + 255) / 256, 1024);
        hipLaunchKernelGGL(k_seg_copy, dim3(g, 1), dim3(256), 0, c->stream, bufs[s], stage + base[(size_t)s], seg(s), 0, c->comm_rank);
      }
    // ONE group, and the group is always closed: a failure inside is reported after ncclGroupEnd (a return between
    // GroupStart and GroupEnd would leave the thread's group open and every later RCCL call of this thread queued in it)
    ncclResult_t first = R->GroupStart();
    const char *what = "ncclGroupStart";
    if (first == ncclSuccess) {
      for (int s = 0; s < nsets && first == ncclSuccess; ++s) {
        const int64_t *o = offs + (size_t)s * W, *n = cnts + (size_t)s * W;
        if (staged[(size_t)s]) {
          double *sb = stage + base[(size_t)s];
          first = R->AllGather(sb + (size_t)c->comm_rank * (size_t)slot[(size_t)s], sb, (size_t)slot[(size_t)s], ncclDouble, comm, c->stream);
        } else {
          first = R->AllGather(bufs[s] + o[c->comm_rank], bufs[s] + o[0], (size_t)n[0], ncclDouble, comm, c->stream);
        }
        if (first != ncclSuccess) what = "ncclAllGather";
      }
      const ncclResult_t end = R->GroupEnd();
      if (first == ncclSuccess && end != ncclSuccess) { first = end; what = "ncclGroupEnd"; }
    }
    if (first != ncclSuccess) return nccl_fail(c, R, first, what);
    for (int s = 0; s < nsets; ++s)                        // unpack every rank's segment of every staged set
      if (staged[(size_t)s] && slot[(size_t)s] > 0) {
        const unsigned g = (unsigned)std::min<int64_t>((slot[(size_t)s] + 255) / 256, 256);
        hipLaunchKernelGGL(k_seg_copy, dim3(g, (unsigned)W), dim3(256), 0, c->stream, bufs[s], stage + base[(size_t)s], seg(s), 1, 0);
      }
    return RBL_OK;
  }
  for (int s = 0; s < nsets; ++s) {
    const int64_t *o = offs + (size_t)s * W, *n = cnts + (size_t)s * W;
    if (c->comm_gather_fn) {
      if (c->comm_gather_fn(c->comm_user, bufs[s], o, n)) return rbl_fail(c, RBL_ERR_COMM, "all-gather callback failed");
      continue;
    }
    int64_t lo = o[0], hi = o[0] + n[0];
    for (int r = 1; r < W; ++r) { lo = std::min(lo, o[r]); hi = std::max(hi, o[r] + n[r]); }
    if (hi > lo && (!c->comm_fn || c->comm_fn(c->comm_user, bufs[s] + lo, hi - lo))) return rbl_fail(c, RBL_ERR_COMM, "all-reduce callback failed");
  }
  return RBL_OK;
}

// per-body segments (per_body doubles per body, body-major, first body at offset base) of nvec vectors `pitch` apart in d_buf
int comm_allgather_bodies(rbl_ctx *c, double *d_buf, int64_t base, int64_t per_body, int nvec, int64_t pitch)
{
  const int W = c->comm_world;
  c->comm_offs.resize((size_t)nvec * W); c->comm_cnts.resize((size_t)nvec * W);
  std::vector<double *> bufs((size_t)nvec, d_buf);
  for (int v = 0; v < nvec; ++v)
    for (int r = 0; r < W; ++r) {
      int b0, b1; comm_body_range_of(c, r, &b0, &b1);
      c->comm_offs[(size_t)v * W + r] = (int64_t)v * pitch + base + (int64_t)b0 * per_body;
      c->comm_cnts[(size_t)v * W + r] = (int64_t)(b1 - b0) * per_body;
    }
  return comm_allgatherv(c, nvec, bufs.data(), c->comm_offs.data(), c->comm_cnts.data());
}

// two per-body parts, each in its own buffer (or at its own offset of one): [lambda (3 N_blb per body) ; U (6 per body)] of the
// preconditioner, lever arms + positions of the row split's geometry
int comm_allgather_bodies2(rbl_ctx *c, double *d_buf1, int64_t base1, int64_t per_body1, double *d_buf2, int64_t base2, int64_t per_body2)
{
  const int W = c->comm_world;
  c->comm_offs.resize((size_t)2 * W); c->comm_cnts.resize((size_t)2 * W);
  double *bufs[2] = {d_buf1, d_buf2};
  for (int r = 0; r < W; ++r) {
    int b0, b1; comm_body_range_of(c, r, &b0, &b1);
    c->comm_offs[r] = base1 + (int64_t)b0 * per_body1;     c->comm_cnts[r] = (int64_t)(b1 - b0) * per_body1;
    c->comm_offs[W + r] = base2 + (int64_t)b0 * per_body2; c->comm_cnts[W + r] = (int64_t)(b1 - b0) * per_body2;
  }
  return comm_allgatherv(c, 2, bufs, c->comm_offs.data(), c->comm_cnts.data());
}

// rows [row_bounds[r], row_bounds[r + 1]) x `width` doubles of every rank r, in place in one vector (the row split's product)
int comm_allgather_rows(rbl_ctx *c, double *d_buf, const int64_t *row_bounds, int64_t width)
{
  const int W = c->comm_world;
  c->comm_offs.resize((size_t)W); c->comm_cnts.resize((size_t)W);
  for (int r = 0; r < W; ++r) { c->comm_offs[r] = row_bounds[r] * width; c->comm_cnts[r] = (row_bounds[r + 1] - row_bounds[r]) * width; }
  return comm_allgatherv(c, 1, &d_buf, c->comm_offs.data(), c->comm_cnts.data());
}

void comm_release(rbl_ctx *c)
{
  if (c->comm_kind == 2 && c->comm_nccl) {
    const RcclApi *R = rccl();
    if (R) (void)R->CommDestroy((ncclComm_t)c->comm_nccl);
  }
  c->comm_nccl = nullptr; c->comm_kind = 0; c->comm_rank = 0; c->comm_world = 1;
  c->comm_fn = nullptr; c->comm_gather_fn = nullptr; c->comm_user = nullptr;
}

// a change of communicator changes who factors which bodies
static void comm_invalidate(rbl_ctx *c)
{
  c->dev_blk_valid = false; c->blk_inv_valid = false; c->dev_pc_valid = false; c->tl_valid = false; c->dev_bodies_valid = false;
  c->pc_keep_once = false;
}

int rbl_set_comm_ops(rbl_ctx *c, int rank, int world, rbl_allreduce_fn allreduce, rbl_allgatherv_fn allgatherv, void *user)
{
  if (!c || world < 1 || rank < 0 || rank >= world) return rbl_fail(c, RBL_ERR_ARG, "set_comm: need 0 <= rank < world");
  if (c->dev_ready) (void)hipStreamSynchronize(c->stream);
  comm_release(c);
  comm_invalidate(c);
  if (!allreduce) return RBL_OK;                         // back to single-GPU products
  c->comm_kind = 1; c->comm_rank = rank; c->comm_world = world;
  c->comm_fn = allreduce; c->comm_gather_fn = allgatherv; c->comm_user = user;
  return RBL_OK;
}

int rbl_set_comm(rbl_ctx *c, int rank, int world, rbl_allreduce_fn fn, void *user)
{
  return rbl_set_comm_ops(c, rank, world, fn, nullptr, user);
}

int rbl_comm_unique_id(void *id_out)
{
  if (!id_out) return RBL_ERR_ARG;
  const RcclApi *R = rccl();
  if (!R) return RBL_ERR_COMM;
  ncclUniqueId id;
  if (R->GetUniqueId(&id) != ncclSuccess) return RBL_ERR_COMM;
  std::memcpy(id_out, &id, sizeof(id));
  return RBL_OK;
}

int rbl_comm_init_rccl(rbl_ctx *c, const void *unique_id, int rank, int world)
{
  if (!c || !unique_id || world < 1 || rank < 0 || rank >= world) return rbl_fail(c, RBL_ERR_ARG, "comm_init_rccl: need an id and 0 <= rank < world");
  int rc = rbl_dev_init(c); if (rc) return rc;
  const RcclApi *R = rccl();
  if (!R) return rbl_fail(c, RBL_ERR_COMM, g_rccl.error);
  RBL_HIP(c, hipStreamSynchronize(c->stream));
  comm_release(c);
  comm_invalidate(c);
  RBL_HIP(c, hipSetDevice(c->device));                    // the communicator binds to the context's device
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof(id));
  ncclComm_t comm = nullptr;
  RBL_NCCL(c, R, R->CommInitRank(&comm, world, id, rank));
  c->comm_nccl = comm; c->comm_kind = 2; c->comm_rank = rank; c->comm_world = world;
  return RBL_OK;
}

int rbl_comm_finalize(rbl_ctx *c)
{
  if (!c) return RBL_ERR_ARG;
  if (c->dev_ready) RBL_HIP(c, hipStreamSynchronize(c->stream));
  comm_release(c);
  comm_invalidate(c);
  return RBL_OK;
}

int rbl_comm_info(const rbl_ctx *c, int *rank, int *world, int *kind)
{
  if (!c) return RBL_ERR_ARG;
  if (rank) *rank = c->comm_rank;
  if (world) *world = c->comm_world;
  if (kind) *kind = c->comm_kind;
  return RBL_OK;
}

// test / benchmark hooks on device buffers: the context's collectives themselves (sum all-reduce; in-place all-gather of
// per-rank segments d_buf[offsets[r] .. + counts[r]))
int rbl_comm_allreduce_dev(rbl_ctx *c, double *d_buf, int64_t count)
{
  if (!c || !d_buf || count < 0) return rbl_fail(c, RBL_ERR_ARG, "comm_allreduce_dev: bad arguments");
  if (!comm_on(c)) return RBL_OK;
  return comm_allreduce(c, d_buf, count);
}

int rbl_comm_allgatherv_dev(rbl_ctx *c, double *d_buf, const int64_t *offsets, const int64_t *counts)
{
  if (!c || !d_buf || !offsets || !counts) return rbl_fail(c, RBL_ERR_ARG, "comm_allgatherv_dev: bad arguments");
  if (!comm_on(c)) return RBL_OK;
  return comm_allgatherv(c, 1, &d_buf, offsets, counts);
}
