// The whole in-place SGD training step of a BPR two-tower model as two host calls (rsa_bpr_sgd_prepare / rsa_bpr_sgd_apply).
//
// Replaces, per batch of recstudio/model/basemodel/recommender.py:596-646 (training_step -> loss.backward() ->
// optimizer.step()) for the stock BPR configuration (mf/bpr.py:7-25: nn.Embedding towers, InnerProductScorer, BPRLoss,
// Uniform / Popular sampler): the sampler call (ann/sampler.py:86-111, :243-258), both tower look-ups, the scorer, the loss,
// autograd's embedding_dense_backward and torch.optim.SGD.step().
//
// The kernels are those of rsa_sample_uniform / rsa_sample_popular, rsa_sort_step_elements, rsa_fused_sample_gather_score
// (ids given, BPR epilogue, in-forward update of the solo rows) and rsa_rows_update_presorted, with two fusions the
// entry-point-by-entry-point sequence cannot have (round 6):
//   * ONE radix sort for the step's item elements AND its user elements (rsa_radix.hpp SrcStepAll: user keys behind the item
//     keys) -- the 65 536 (user id, query) pairs of the headline step cost nine launches of their own before;
//   * the negatives are drawn INSIDE pass 0's histogram launch of that sort (StepDraw: the same Philox elements, the same
//     bucket-line look-up as the stand-alone samplers, ids stored for the later launches) -- no sampler launch, no id read.
// Same ids, same sorted orders, same chunking of the apply passes: results are bit-identical to the call-by-call sequence
// (tests/test_gpu_round4.py, test_gpu_round6.py).  Host time: two calls instead of seven (~25 us of Python each).
#include "rsa_common.hpp"
#include "rsa_radix.hpp"
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
  RSA_CHECK_ARG(a.user_ids && a.pos_ids && a.neg_ids && a.solo && a.item_workspace,
                "%s: null pointer (user_ids / pos_ids / neg_ids / solo / item_workspace)", who);
  RSA_CHECK_ARG(a.item_workspace_bytes >= step_all_workspace_bytes(a.n_queries, a.num_neg),
                "%s: item_workspace holds %lld bytes, the step needs rsa_scatter_rows_sorted_workspace_bytes(n_queries, num_neg + 1, "
                "n_items) = %lld (ABI 11: the user rows are sorted with the item rows; user_workspace is no longer used)",
                who, (long long)a.item_workspace_bytes, (long long)step_all_workspace_bytes(a.n_queries, a.num_neg));
  return RSA_OK;
}

extern "C" int rsa_bpr_sgd_prepare(const rsa_bpr_sgd_args* args, rsa_stream_t stream) {
  rsa_bpr_sgd_args a;
  if (int rc = load_args(a, args, "rsa_bpr_sgd_prepare")) return rc;
  if (int rc = check_step(a, "rsa_bpr_sgd_prepare")) return rc;
  if (a.n_queries == 0) return RSA_OK;
  const int64_t numel = a.n_queries * (int64_t)a.num_neg;
  // 1. the negatives: what Sampler.forward draws (same stream, same generator consumption as the in-kernel samplers).  Drawn by
  //    the sort's first launch where that launch exists (more than one tile of elements) and the look-up is the bucket-line one;
  //    by the stand-alone sampler otherwise.  RSA_SAMPLER_GIVEN: neg_ids is an input.
  StepDraw draw{a.neg_ids, 0, 0, PhiloxCall{a.seed, a.offset >> 2, a.grid_threads, a.elem_base}, 0, 0, nullptr, nullptr, nullptr};
  const bool in_sort = a.n_queries * (int64_t)(a.num_neg + 2) > RDX_TILE;
  if (a.sampler == RSA_SAMPLER_UNIFORM) {
    const int64_t high = a.uniform_high > 0 ? a.uniform_high : a.n_items;
    RSA_CHECK_ARG(high >= 2 && high <= a.n_items, "rsa_bpr_sgd_prepare: uniform_high = %lld outside [2, n_items]", (long long)high);
    RSA_CHECK_ARG(a.grid_threads > 0 && (a.offset & 3) == 0, "rsa_bpr_sgd_prepare: bad philox state");
    if (in_sort) {
      draw.kind = 1;
      draw.range = (uint64_t)(high - 1);
      draw.low = 1;
    } else if (int rc = rsa_sample_uniform(a.neg_ids, numel, 1, high, a.seed, a.offset, a.grid_threads, a.elem_base, stream)) {
      return rc;
    }
  } else if (a.sampler == RSA_SAMPLER_POPULAR) {
    rsa_popular_args p;
    if (int rc = load_args(p, a.pop, "rsa_bpr_sgd_prepare(pop)")) return rc;
    RSA_CHECK_ARG(p.n_items == a.n_items, "rsa_bpr_sgd_prepare: the popularity tables cover %lld items, the item table has %lld rows",
                  (long long)p.n_items, (long long)a.n_items);
    RSA_CHECK_ARG(a.grid_threads > 0 && (a.offset & 3) == 0, "rsa_bpr_sgd_prepare: bad philox state");
    if (in_sort && p.cdf_lines != nullptr && p.table && p.pop_prob && p.lines_log2 >= 0 && p.lines_log2 <= 28 &&
        ((uintptr_t)p.cdf_lines & 127) == 0) {
      draw.kind = 2;
      draw.lines = p.cdf_lines;
      draw.lines_log2 = p.lines_log2;
      draw.table = p.table;
      draw.pop_prob = p.pop_prob;
    } else if (int rc = sample_popular_impl(p.table, p.pop_prob, p.guide, p.n_items, p.guide_log2, a.neg_ids, nullptr, nullptr, numel,
                                            a.seed, a.offset, a.grid_threads, a.elem_base, p.cdf_lut, p.cdf_lines, p.lines_log2, stream)) {
      return rc;
    }
  } else if (a.sampler != RSA_SAMPLER_GIVEN) {
    rsa::set_error("rsa_bpr_sgd_prepare: sampler must be RSA_SAMPLER_GIVEN, RSA_SAMPLER_UNIFORM or RSA_SAMPLER_POPULAR");
    return RSA_ERR_ARG;
  }
  // 2. ONE sort: the step's (item id, element) pairs by id followed by its (user id, query) pairs by user; which item elements
  //    are alone on their row
  return sort_step_all(a.pos_ids, a.neg_ids, a.user_ids, a.n_queries, a.num_neg, a.n_items, a.n_users, a.solo, a.item_workspace,
                       a.item_workspace_bytes, &draw, (hipStream_t)stream, "rsa_bpr_sgd_prepare");
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
  if (int rc = apply_step_all(false, a.user_table, a.user_ids, a.dim, a.n_queries, a.num_neg, a.dpos, a.dneg, a.step_scale, a.n_items,
                              a.n_users, a.item_table, a.item_workspace, (hipStream_t)stream))
    return rc;
  // the user rows: user[uid] += step_scale * query_grad, duplicates of a user summed in sorted order
  return apply_step_all(true, a.query_grad, nullptr, a.dim, a.n_queries, a.num_neg, nullptr, a.ones, a.step_scale, a.n_items, a.n_users,
                        a.user_table, a.item_workspace, (hipStream_t)stream);
}
