// The whole in-place SGD training step of a BPR two-tower model as two host calls (rsa_bpr_sgd_prepare / rsa_bpr_sgd_apply).
//
// Replaces, per batch of recstudio/model/basemodel/recommender.py:596-646 (training_step -> loss.backward() ->
// optimizer.step()) for the stock BPR configuration (mf/bpr.py:7-25: nn.Embedding towers, InnerProductScorer, BPRLoss,
// Uniform / Popular sampler): the sampler call (ann/sampler.py:86-111, :243-258), both tower look-ups, the scorer, the loss,
// autograd's embedding_dense_backward and torch.optim.SGD.step().
//
// No new kernels: the calls issue the launches of rsa_sample_uniform / rsa_sample_popular, rsa_sort_step_elements,
// rsa_fused_sample_gather_score (ids given, BPR epilogue, in-forward update of the solo rows) and rsa_rows_update_presorted
// in the order recstudio_amd/fused.py issued them one ctypes call at a time.  What this saves is host time: ~25 us of
// Python per entry point (tensor checks, argument marshalling, output allocation) x 7 entry points per step, more than the
// kernels of a B = 4096 step run on the GPU.  Results are bit-identical to the call-by-call sequence.
#include "rsa_common.hpp"
#include "rsa_internal.hpp"

using namespace rsa;

static int check_step(const rsa_bpr_sgd_args& a, const char* who) {
  RSA_CHECK_ARG(a.n_queries >= 0 && a.n_items >= 2 && a.n_users >= 1 && a.n_items < (1ll << 31) && a.n_users < (1ll << 31),
                "%s: bad sizes", who);
  RSA_CHECK_ARG(a.num_neg == 64, "%s: num_neg = %d: the in-forward update is built for one 64-negative tile per query", who,
                a.num_neg);
  if (a.dim != 64 && a.dim != 128 && a.dim != 256) {
    rsa::set_error("%s: dim=%d: built for dim in {64, 128, 256}", who, a.dim);
    return RSA_ERR_UNSUPPORTED;
  }
  RSA_CHECK_ARG(a.user_ids && a.pos_ids && a.neg_ids && a.solo && a.item_workspace && a.user_workspace,
                "%s: null pointer (user_ids / pos_ids / neg_ids / solo / workspaces)", who);
  return RSA_OK;
}

extern "C" int rsa_bpr_sgd_prepare(const rsa_bpr_sgd_args* args, rsa_stream_t stream) {
  rsa_bpr_sgd_args a;
  if (int rc = load_args(a, args, "rsa_bpr_sgd_prepare")) return rc;
  if (int rc = check_step(a, "rsa_bpr_sgd_prepare")) return rc;
  if (a.n_queries == 0) return RSA_OK;
  const int64_t numel = a.n_queries * (int64_t)a.num_neg;
  // 1. the negatives: what Sampler.forward draws (same stream, same generator consumption as the in-kernel samplers)
  if (a.sampler == RSA_SAMPLER_UNIFORM) {
    const int64_t high = a.uniform_high > 0 ? a.uniform_high : a.n_items;
    RSA_CHECK_ARG(high >= 2 && high <= a.n_items, "rsa_bpr_sgd_prepare: uniform_high = %lld outside [2, n_items]", (long long)high);
    if (int rc = rsa_sample_uniform(a.neg_ids, numel, 1, high, a.seed, a.offset, a.grid_threads, a.elem_base, stream)) return rc;
  } else if (a.sampler == RSA_SAMPLER_POPULAR) {
    rsa_popular_args p;
    if (int rc = load_args(p, a.pop, "rsa_bpr_sgd_prepare(pop)")) return rc;
    RSA_CHECK_ARG(p.n_items == a.n_items, "rsa_bpr_sgd_prepare: the popularity tables cover %lld items, the item table has %lld rows",
                  (long long)p.n_items, (long long)a.n_items);
    if (int rc = sample_popular_impl(p.table, p.pop_prob, p.guide, p.n_items, p.guide_log2, a.neg_ids, nullptr, nullptr, numel, a.seed,
                                     a.offset, a.grid_threads, a.elem_base, p.cdf_lut, p.cdf_lines, p.lines_log2, stream))
      return rc;
  } else {
    rsa::set_error("rsa_bpr_sgd_prepare: sampler must be RSA_SAMPLER_UNIFORM or RSA_SAMPLER_POPULAR");
    return RSA_ERR_ARG;
  }
  // 2. the step's (item id, element) pairs sorted by id, and which elements are alone on their row
  rsa_rows_update_args r;
  __builtin_memset(&r, 0, sizeof r);
  r.size = sizeof r;
  r.pos_ids = a.pos_ids;
  r.neg_ids = a.neg_ids;
  r.n_queries = a.n_queries;
  r.num_neg = a.num_neg;
  r.n_items = a.n_items;
  r.pad_row = 0;
  r.solo = a.solo;
  r.workspace = a.item_workspace;
  r.workspace_bytes = a.item_workspace_bytes;
  if (int rc = rsa_sort_step_elements(&r, stream)) return rc;
  // 3. the (user id, query) pairs: one "negative" per query, no positives, nothing flagged
  r.pos_ids = nullptr;
  r.neg_ids = a.user_ids;
  r.num_neg = 1;
  r.n_items = a.n_users;
  r.solo = nullptr;
  r.workspace = a.user_workspace;
  r.workspace_bytes = a.user_workspace_bytes;
  return rsa_sort_step_elements(&r, stream);
}

extern "C" int rsa_bpr_sgd_apply(const rsa_bpr_sgd_args* args, rsa_stream_t stream) {
  rsa_bpr_sgd_args a;
  if (int rc = load_args(a, args, "rsa_bpr_sgd_apply")) return rc;
  if (int rc = check_step(a, "rsa_bpr_sgd_apply")) return rc;
  if (a.n_queries == 0) return RSA_OK;
  RSA_CHECK_ARG(a.item_table && a.user_table && a.step_scale && a.pos_score && a.neg_score && a.row_loss && a.dpos && a.dneg &&
                    a.query_grad && a.ones && a.loss_out && a.reduce_scratch,
                "rsa_bpr_sgd_apply: null pointer");
  // forward + BPR epilogue + user gradient; item rows only one element touches are rewritten by the wave that holds them
  rsa_fused_args f;
  __builtin_memset(&f, 0, sizeof f);
  f.item_table = a.item_table;
  f.n_items = a.n_items;
  f.dim = a.dim;
  f.score_mode = RSA_SCORE_IP;
  f.query = a.user_table;
  f.query_index = a.user_ids;
  f.n_query_rows = a.n_users;
  f.pos_ids = a.pos_ids;
  f.n_queries = a.n_queries;
  f.num_neg = a.num_neg;
  f.sampler = RSA_SAMPLER_GIVEN;
  f.neg_ids = a.neg_ids;
  f.pos_score = a.pos_score;
  f.neg_score = a.neg_score;
  f.fused_loss = RSA_LOSS_BPR + 1;
  f.row_loss = a.row_loss;
  f.loss_out = a.loss_out;
  f.dpos = a.dpos;
  f.dneg = a.dneg;
  f.query_grad = a.query_grad;
  f.reduce_scratch = a.reduce_scratch;
  f.solo_flags = a.solo;
  f.upd_scale = a.step_scale;
  if (int rc = rsa_fused_sample_gather_score(&f, stream)) return rc;
  // the shared item rows: every such row read-modified-written once, in sorted order (needs the PRE-update user rows)
  rsa_rows_update_args r;
  __builtin_memset(&r, 0, sizeof r);
  r.size = sizeof r;
  r.query = a.user_table;
  r.query_index = a.user_ids;
  r.n_query_rows = a.n_users;
  r.dim = a.dim;
  r.has_pos = 1;
  r.n_queries = a.n_queries;
  r.num_neg = a.num_neg;
  r.dpos = a.dpos;
  r.dneg = a.dneg;
  r.upstream = a.step_scale;
  r.n_items = a.n_items;
  r.pad_row = 0;
  r.target = a.item_table;
  r.workspace = a.item_workspace;
  r.workspace_bytes = a.item_workspace_bytes;
  if (int rc = rsa_rows_update_presorted(&r, stream)) return rc;
  // the user rows: user[uid] += step_scale * query_grad, duplicates of a user summed in sorted order
  r.query = a.query_grad;
  r.query_index = nullptr;
  r.n_query_rows = a.n_queries;
  r.has_pos = 0;
  r.num_neg = 1;
  r.dpos = nullptr;
  r.dneg = a.ones;
  r.n_items = a.n_users;
  r.target = a.user_table;
  r.workspace = a.user_workspace;
  r.workspace_bytes = a.user_workspace_bytes;
  return rsa_rows_update_presorted(&r, stream);
}
