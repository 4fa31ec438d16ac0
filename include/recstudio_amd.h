/*
 * recstudio_amd -- C ABI of the MI355X (gfx950) retriever hot path.
 *
 * One shared library, librecstudio_amd.so, built from recstudio_amd/csrc/ by
 * hipcc --offload-arch=gfx950.  Plain pointers and sizes only: every pointer
 * is a DEVICE pointer unless stated otherwise, `stream` is a hipStream_t
 * passed as void*.  Every entry point returns 0 on success or a negative
 * rsa_status; rsa_last_error() returns the thread-local message of the last
 * failure.  No entry point synchronises or allocates: the few reductions that need device scratch (mean of the
 * row losses, valid-row count of the BCE losses, arrival counter + per-workgroup partials of the in-kernel loss
 * reduction) take a caller-owned block of rsa_scratch_bytes() bytes, zero-filled ONCE by the caller and then
 * left alone (the kernels restore the words they use); one block per device and stream -- calls sharing a block
 * must be stream-ordered.
 *
 * The reference (ustcml/RecStudio) is pure Python on PyTorch; it has no native
 * boundary of its own.  Each entry point below therefore replaces a SEQUENCE
 * of ATen ops in the reference, cited as file:line under /root/reference.
 * The Python binding a maintainer would add is in INTEGRATION.md; the binding
 * this repo ships is recstudio_amd/_native.py (ctypes).
 *
 * Index convention (same as the reference): tables are row-major fp32
 * [rows, dim]; row 0 is the padding row; ids are int64.
 */
#ifndef RECSTUDIO_AMD_H
#define RECSTUDIO_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSA_ABI_VERSION 11  /* 11: rsa_bpr_sgd_prepare / _apply sort the step's user rows WITH its item rows (one radix sort; item_workspace = rsa_scatter_rows_sorted_workspace_bytes(n_queries, num_neg + 1, n_items), user_workspace unused) and draw the negatives inside that sort's first launch; sampler RSA_SAMPLER_GIVEN accepted (neg_ids is an input);
                               10: rsa_fullscore_lse_grad (flash forward: logsumexp + d/d query in one pass); rsa_fullscore_softmax_dw (d/d items of the full softmax with the softmax tile recomputed on the matrix cores: no
                               [B, N] matrix anywhere in the backward); rsa_fullscore_softmax_dq: probs may be NULL (not written);
                               9: every entry point that took more than 12 positional arguments takes ONE argument block whose first field is its
                               own size (see "Versioned argument blocks"): rsa_popular_args (rsa_sample_popular, rsa_popular_lookup), rsa_loss_args
                               (rsa_pairwise_loss -- which now covers the kinds of the former rsa_pairwise_loss_ex --, rsa_ssm_shared_loss),
                               rsa_seg_gather_args, rsa_fullscore_args, rsa_rows_update_args (rsa_sort_step_elements, rsa_rows_update_sorted,
                               rsa_rows_update_presorted: the former rsa_scatter_rows_sorted / _presorted / rsa_adam_rows_sorted / _presorted);
                               rsa_bpr_sgd_prepare / rsa_bpr_sgd_apply (the whole in-place SGD step of a BPR two-tower model as two calls);
                               8: rsa_shard_owner_bpr_args.finish_parts / forward_parts (the owner-side step in parts, for a caller with two streams);
                               7: rsa_shard_pos_score / rsa_shard_owner_bpr_forward / _finish (the BPR step evaluated on the owners: rows read
                               once per step); rsa_shard_sample_route: route_pos; rsa_shard_backward_segments (the owner side of the sharded backward in one call: in-tree radix sorts
                               straight from the received segments, one walk over the query runs that reads every item row once and
                               updates solo rows in place, sorted apply for the shared rows); the sorted scatters no longer call rocPRIM;
                               6: rsa_adam_rows_presorted; rows_per_shard == 0 = interleaved row ownership in the shard routing entry points;
                               5: version 2 of the fixed-capacity shard exchange (rsa_shard_sample_route: sampling fused into the routing
                               pass, self-describing segments with {count, dropped} headers, 32-bit slots; rsa_shard_score_segments;
                               rsa_shard_home: scatter + loss + mean + routed-order gradient in one launch; rsa_shard_unpack_segments;
                               rsa_shard_scatter_slots); rsa_sample_masked_uniform takes elem_base;
                               4: rsa_fullscore_softmax_dq (d/d query of the full softmax on the matrix cores, in the
                               recompute pass); 3: caller-owned reduction scratch (rsa_scratch_bytes; the library allocates nothing);
                               Philox element base (G-invariant sampling across ranks); bucket-line inverse CDF
                               (cdf_lines); SampledSoftmax epilogue (fused_loss = 2); cosine / Euclidean full-catalog
                               scores (rsa_row_sqnorm, rsa_fullscore score_mode); fixed-capacity shard routing
                               (rsa_shard_route_fixed, negative keys / positions = empty slot).
                               2: rsa_fused_args grew query_grad, packed_keys, offset_dev; new entry points
                               (rsa_row_topk, rsa_fullscore_softmax, rsa_pairwise_loss_ex, rsa_scatter_rows_sorted,
                               rsa_rng_advance) */

typedef void* rsa_stream_t; /* hipStream_t */

enum rsa_status {
  RSA_OK = 0,
  RSA_ERR_ARG = -1,         /* bad argument (null pointer, size, unsupported dim) */
  RSA_ERR_HIP = -2,         /* a HIP runtime call / kernel launch failed */
  RSA_ERR_UNSUPPORTED = -3  /* valid request this build does not implement */
};

enum rsa_score_mode { RSA_SCORE_IP = 0, RSA_SCORE_COS = 1, RSA_SCORE_EUC = 2 /* EuclideanScorer, scorer.py:28-34 */ };
enum rsa_sampler_kind { RSA_SAMPLER_GIVEN = 0, RSA_SAMPLER_UNIFORM = 1, RSA_SAMPLER_POPULAR = 2 };
enum rsa_loss_kind { RSA_LOSS_BPR = 0, RSA_LOSS_SSM = 1, RSA_LOSS_BCE = 2,
                     /* rsa_pairwise_loss: */ RSA_LOSS_WBPR = 3, RSA_LOSS_WBCE = 4, RSA_LOSS_HINGE = 5, RSA_LOSS_NCE = 6,
                     RSA_LOSS_CCL = 7 };

/* ---- Versioned argument blocks (ABI 9) --------------------------------------------------------------------------
 * The structs below whose FIRST field is `int64_t size` are read as follows: the caller sets size = sizeof(the struct)
 * as ITS header declares it; the library copies min(size, its own sizeof) bytes into a zero-filled block of its own.  Fields
 * are only ever appended, and 0 / NULL always means "not given": a caller compiled against an older header keeps working
 * against a newer library (the new fields read as 0), and a newer caller against an older library has its extra fields
 * ignored.  rsa_fused_args / rsa_backward_args / the rsa_shard_* blocks predate this and are covered by RSA_ABI_VERSION. */

const char* rsa_last_error(void);
int rsa_abi_version(void);
/* Size of the caller-owned reduction scratch (see above). */
int64_t rsa_scratch_bytes(void);

/* Device properties torch's distribution kernels size their grid from
 * (multiProcessorCount, maxThreadsPerMultiProcessor) -- needed to reproduce
 * the device random stream.  Host pointers. */
int rsa_device_info(int device, int32_t* cu_count, int32_t* max_threads_per_cu, int32_t* wave_size);

/* ---- Philox state ---------------------------------------------------------
 * (seed, offset) are the torch CUDA/HIP generator's philox seed and offset at
 * the moment the reference would have called torch.randint / torch.rand;
 * grid_threads is the thread count of torch's distribution grid for `numel`
 * outputs (256 * min(CUs * maxThreadsPerCU/256, ceil(numel/256))).  With
 * these, element i of the output is bit-identical to element i of
 * torch.randint(low, high, (numel,), device='cuda') / torch.rand(numel).
 * elem_base: output element i draws what element elem_base + i of the torch call would -- a rank that owns
 * rows [r*B, (r+1)*B) of a global [G*B, n] id tensor passes elem_base = r*B*n and the grid_threads of the GLOBAL
 * call, so that the negatives of a run do not depend on the number of GPUs; 0 otherwise. */

/* UniformSampler.forward -- recstudio/ann/sampler.py:86-111
 * (torch.randint(1, num_items+1, (num_queries, num_neg)), :102-104).
 * neg_ids[numel] <- low + philox % (high - low). */
int rsa_sample_uniform(int64_t* neg_ids, int64_t numel, int64_t low, int64_t high,
                       uint64_t seed, uint64_t offset, uint32_t grid_threads, uint64_t elem_base, rsa_stream_t stream);

/* MaskedUniformSampler.forward / uniform_sample_masked_hist -- recstudio/ann/sampler.py:117-147, :187-214:
 * rejection-free uniform negatives over the items NOT in each user's 0-padded history
 * user_hist [n_rows, hist_len] (hist_len <= 2048).  num_items excludes the padding id.  neg_ids
 * [n_rows, per_row] (per_row = queries-per-user x num_neg); u = torch.rand(n_rows, per_row) on the
 * device stream (elem_base as in "Philox state": a rank's rows of one job-wide call). */
int rsa_sample_masked_uniform(const int64_t* user_hist, int64_t n_rows, int32_t hist_len, int64_t num_items,
                              int32_t per_row, int64_t* neg_ids, uint64_t seed, uint64_t offset,
                              uint32_t grid_threads, uint64_t elem_base, rsa_stream_t stream);

/* PopularSamplerModel.forward -- recstudio/ann/sampler.py:243-258
 * (u = torch.rand; ids = torch.searchsorted(table, u); logp = log(pop_prob[ids])).
 * table/pop_prob: fp32 [n_items] exactly as the reference's registered buffers
 * (:239-241).  guide: int32 [2^guide_log2 + 1], guide[j] = first index with
 * table[i] >= j / 2^guide_log2, guide[2^guide_log2] = n_items (cut-point
 * acceleration, guide_log2 <= 28; returns the same index searchsorted does).  An id that would be
 * n_items (u above table[-1]) is clamped to n_items-1.  neg_logp / u_out may be
 * null.  cdf_lut / cdf_lines (nullable, see rsa_fused_args) are faster forms of the same lookup; with
 * cdf_lines the guide may be null. */
typedef struct rsa_popular_args {
  int64_t size;                /* sizeof(rsa_popular_args) */
  const float* table;          /* [n_items] CDF */
  const float* pop_prob;       /* [n_items] */
  const int32_t* guide;        /* nullable with cdf_lines */
  int64_t n_items;
  int32_t guide_log2;
  int32_t lines_log2;
  const float* cdf_lut;        /* nullable */
  const float* cdf_lines;      /* nullable */
  const float* u_in;           /* rsa_popular_lookup: the uniforms [numel]; ignored by rsa_sample_popular */
  int64_t* ids;                /* [numel] out */
  float* logp;                 /* nullable [numel] out */
  float* u_out;                /* nullable [numel] out (rsa_sample_popular: the uniforms drawn) */
  int64_t numel;
  uint64_t seed, offset;       /* "Philox state" (rsa_sample_popular) */
  uint32_t grid_threads;
  uint32_t _pad;
  uint64_t elem_base;
} rsa_popular_args;
int rsa_sample_popular(const rsa_popular_args* args, rsa_stream_t stream);

/* The same inverse-CDF lookup for caller-supplied uniforms u_in[numel] (used by the
 * parity tests to hit exact table edges). */
int rsa_popular_lookup(const rsa_popular_args* args, rsa_stream_t stream);

/* PopularSamplerModel.compute_item_p -- recstudio/ann/sampler.py:257-258:
 * logp[i] = log(pop_prob[ids[i]]). */
int rsa_item_logp(const float* pop_prob, int64_t n_items, const int64_t* ids, int64_t numel,
                  float* logp, rsa_stream_t stream);

/* torch.nn.Embedding forward (F.embedding) -- baseretriever.py:153-154, :167-168,
 * :211, seq/sasrec.py:42.  out[numel, dim] <- table[ids]. */
int rsa_embedding_gather(const float* table, int64_t n_rows, int32_t dim, const int64_t* ids,
                         int64_t numel, float* out, rsa_stream_t stream);

/* Fused BaseRetriever.forward body for an Embedding item tower --
 * recstudio/model/basemodel/baseretriever.py:153-171 : sampler (S1/S2 above, or
 * ids given) -> item_encoder(neg ids) -> score_func(query, pos) and
 * score_func(query, neg) (recstudio/model/scorer.py:5-25), without materialising
 * the [M, n, dim] negative rows.
 *
 * M "queries" (B, or B*L for sequence targets), n negatives each.  Flat element
 * e = m*n + j is element e of the reference's [M, n] id tensor, so sampled ids
 * are bit-identical to the reference's device stream. */
typedef struct rsa_fused_args {
  const float* item_table;     /* [n_items, dim] */
  int64_t n_items;
  int32_t dim;                 /* multiple of 4, <= 1024 */
  int32_t score_mode;          /* rsa_score_mode */
  const float* query;          /* [n_query_rows, dim]: query vectors, or a user table */
  const int64_t* query_index;  /* nullable [M]: row of `query` for query m (user-id gather,
                                  baseretriever.py:211); null => row m */
  int64_t n_query_rows;
  const int64_t* pos_ids;      /* nullable [M] */
  int64_t n_queries;           /* M */
  int32_t num_neg;             /* n */
  int32_t sampler;             /* rsa_sampler_kind */
  int32_t mask_pad_pos;        /* !=0: pos_score = -inf where pos_ids == 0 (baseretriever.py:164-165) */
  int32_t guide_log2;
  uint64_t seed, offset;       /* philox state (sampler != GIVEN) */
  uint32_t grid_threads;
  uint32_t _pad;
  const float* table;          /* POPULAR: cdf / pop_prob / guide as in rsa_sample_popular */
  const float* pop_prob;
  const int32_t* guide;
  int64_t* neg_ids;            /* [M, n]: INPUT if sampler == GIVEN, else OUTPUT */
  float* neg_logp;             /* nullable [M, n] out (POPULAR); INPUT log-probabilities when sampler == GIVEN and
                                  fused_loss == 2 (null = 0) */
  float* pos_logp;             /* nullable [M] out (POPULAR, needs pos_ids); INPUT like neg_logp for GIVEN + fused_loss 2 */
  float* pos_score;            /* nullable [M] out (needs pos_ids) */
  float* neg_score;            /* [M, n] out; nullable with fused_loss != 0 (a training forward that keeps no scores: 4 B/triplet less) */
  const float* table_prob;     /* nullable [n_items][2]: interleaved copy {table[i], pop_prob[i]}.  When given,
                                  the CDF probes and the log-prob read share cache lines (one Infinity-Cache
                                  round trip fewer per sampled id); results are identical. */
  int32_t fused_loss;          /* 0 = none; 1 = BPRLoss (loss_func.py:55-59); 2 = SampledSoftmaxLoss (loss_func.py:80-90,
                                  one positive per row; logsumexp over the positive and all num_neg negatives of a query
                                  carried across its num_neg/64 tiles by ONE wave; inner product, dim in {32,64,128,256};
                                  a -inf positive yields NaN like the reference).  Evaluated in the kernel's epilogue:
                                  needs num_neg % 64 == 0, pos_ids, pos_score.  Outputs below. */
  int32_t _pad2;
  float* row_loss;             /* [M] out: per-query loss */
  float* loss_out;             /* nullable [1] out: mean over queries, reduced in the same launch by the last workgroup
                                  to finish (fixed summation order: reproducible run to run) */
  float* dpos;                 /* nullable [M] out: d loss_out / d pos_score */
  float* dneg;                 /* nullable [M, n] out: d loss_out / d neg_score */
  const float* cdf_lut;        /* nullable [2^guide_log2 + 1][4] fp32, one self-contained entry per guide bucket b with
                                  lo = guide[b], hi = guide[b+1]: {lo as int32 bits, bit 31 set when hi - lo >= 2;
                                  table[lo] (+inf when hi == lo); pop_prob[min(lo, n_items-1)];
                                  pop_prob[min(lo+1, n_items-1)]}.  Direct-lookup form of the inverse CDF: one 16-byte
                                  read resolves id and probability for every bucket holding <= 1 CDF boundary
                                  (one HBM line instead of three dependent round trips); identical results. */
  float* query_grad;           /* nullable [M, dim] out, fused_loss != 0 only (inner product, dim in {32,64,128,256}):
                                  d loss_out / d query row m = sum_j dneg[m,j] * item[neg_ids[m,j]] + dpos[m] * item[pos],
                                  accumulated while the rows are in registers, so that rsa_fused_backward can be
                                  called without query_grad / query_table_grad and then never reads an item row. */
  const int64_t* packed_keys;  /* nullable [M]: sampler GIVEN, num_neg == 1, no positives: element m scores query row
                                  (key >> 32) against item row (key & 0xffffffff) -- the owner side of the sharded
                                  exchange (rsa_shard_route keys) without unpacking; neg_ids / query_index unused.
                                  A negative key is an empty slot: its score is 0. */
  const uint64_t* offset_dev;  /* nullable device word: when set, the Philox offset is read from it at run time instead
                                  of `offset` (a multiple of 4, like torch's) -- lets a captured HIP graph draw fresh
                                  numbers on every replay; advance it with rsa_rng_advance in the same graph. */
  uint64_t elem_base;          /* Philox element index of flat element 0 (see "Philox state"); 0 on one GPU */
  void* reduce_scratch;        /* loss_out != null: caller-owned scratch of rsa_scratch_bytes() bytes (zeroed once) */
  const float* cdf_lines;      /* nullable [2^lines_log2 + 1][32] fp32 (128-byte aligned): BUCKET LINES, the one-HBM-line form
                                  of the inverse CDF.  Line b describes u in [b, b+1) / 2^lines_log2 through the DISTINCT
                                  CDF values inside it (items whose CDF equals their predecessor's can never be returned
                                  by searchsorted and are left out), 12 slots:
                                    words 0..11   cdf of the bucket's first distinct values, then (if fewer than 12) of the
                                                  first distinct entry ABOVE the bucket, then +inf
                                    words 12..23  pop_prob of those entries
                                    words 24..29  their ids as uint16 offsets from `base` (even slot in the low half)
                                    word  30      base id (int32 bits);  word 31: number of distinct values in the bucket
                                                  (int32; -1 = ids too far apart for 16-bit offsets)
                                  id = base + delta[#{cdf[i] < u}] -- the same comparisons torch.searchsorted makes; a
                                  draw above 12 in-bucket values (or a -1 line) is searched in `table` between this
                                  line's and the next line's base (line 2^lines_log2 is a sentinel).  Takes precedence
                                  over cdf_lut / guide. */
  int32_t lines_log2;
  int32_t _pad3;
  /* In-forward SGD (fused_loss = 1, num_neg == 64, ids GIVEN, query_grad given; ABI v5).  solo_flags: the classification
   * of the step's elements by rsa_sort_step_elements (element order m * (num_neg + 1) + c, c = 0 the positive: 1 <=> no
   * other element of the step touches that item row, and the row is not the padding row).  Such a row is rewritten by
   * the wave that has it in registers: item[id] += upd_scale[0] * d * q (d = d loss/d score of that element; upd_scale =
   * -lr is plain SGD) -- nothing else reads or writes that row in the step, so the result is the one a separate update
   * pass gives.  The other elements are applied by rsa_rows_update_presorted on the same workspace, which skips
   * exactly the flagged ones.  item_table is written. */
  const uint8_t* solo_flags;   /* nullable [M * (num_neg + 1)] */
  const float* upd_scale;      /* device scalar */
  /* A QUEUE of independent batches consumed by one resident grid (ABI 9; forward-only scoring / loss evaluation of many
   * batches: a small batch's launch is a sampling phase followed by a row phase, and in a stream of such launches the
   * first has nothing to hide under -- inside ONE grid the waves of different batches are out of phase, so one batch's
   * draw -> bucket line -> slot chain runs under another's row reads).  n_batches > 1: the n_queries queries are n_batches
   * consecutive batches of n_queries / n_batches; batch k's negatives are those of ITS OWN torch call -- (seed, offset +
   * k * batch_offset_step, grid_threads of ONE batch's numel) -- i.e. exactly what n_batches consecutive launches draw.
   * In-kernel samplers, num_neg % 64 == 0, loss_out null (row_loss holds every query's loss; the per-batch means are the
   * caller's), the BPR epilogue with num_neg == 64. */
  int32_t n_batches;           /* 0 / 1: one batch */
  int32_t _pad4;
  uint64_t batch_offset_step;  /* Philox offset consumed by one batch's torch call (multiple of 4) */
} rsa_fused_args;

int rsa_fused_sample_gather_score(const rsa_fused_args* args, rsa_stream_t stream);

/* *offset_dev += increment (one thread).  Together with rsa_fused_args.offset_dev this keeps the device copy of the
 * torch generator offset moving inside a captured graph; `increment` is what torch's distribution call would
 * consume ((numel-1)/(grid_threads*unroll)+1)*4, see "Philox state"). */
int rsa_rng_advance(uint64_t* offset_dev, uint64_t increment, rsa_stream_t stream);

/* BPRLoss.forward (recstudio/model/loss_func.py:55-59) / SampledSoftmaxLoss.forward
 * (:80-90) / BinaryCrossEntropyLoss.forward (:105-127, dns=False) on pos_score [M], neg_score [M, n] (each positive with its own n
 * negatives), value AND gradient in one pass.  pos_logp / neg_logp are nullable
 * (treated as 0; the reference's UniformSampler hands int64 zeros).  loss_out[1]
 * = mean over rows; row_loss [M] scratch/out; dpos [M] and dneg [M, n] (nullable)
 * = d loss_out / d score. */
typedef struct rsa_loss_args {
  int64_t size;                /* sizeof(rsa_loss_args) */
  int32_t loss_kind;           /* rsa_loss_kind */
  int32_t n_pos;               /* rsa_ssm_shared_loss: positives per row (L); ignored by rsa_pairwise_loss */
  const float* pos_score;      /* [n_rows] ([n_rows, n_pos] for rsa_ssm_shared_loss) */
  const float* neg_score;      /* [n_rows, num_neg] */
  const float* pos_logp;       /* nullable, like pos_score */
  const float* neg_logp;       /* nullable, like neg_score */
  int64_t n_rows;
  int32_t num_neg;
  int32_t _pad;
  float param0, param1;        /* RSA_LOSS_HINGE: margin; RSA_LOSS_CCL: margin, neg_weight */
  float* row_loss;             /* [n_rows] scratch / out */
  float* loss_out;             /* [1] out: the mean over rows */
  float* dpos;                 /* nullable out, like pos_score */
  float* dneg;                 /* nullable out, like neg_score */
  void* scratch;               /* rsa_scratch_bytes() */
} rsa_loss_args;
int rsa_pairwise_loss(const rsa_loss_args* args, rsa_stream_t stream);

/* loss_kind >= RSA_LOSS_WBPR -- the other PairwiseLoss classes of recstudio/model/loss_func.py, value + d loss/d score in
 * one pass:  RSA_LOSS_WBPR WeightedBPRLoss (:93-97; the softmax(neg - logQ) weights are differentiated
 * through, as in the reference), RSA_LOSS_WBCE WeightedBinaryCrossEntropyLoss (:135-137 on :105-127, padded -inf
 * positives dropped), RSA_LOSS_HINGE HingeLoss (:140-154, num_items=None; param0 = margin), RSA_LOSS_NCE NCELoss
 * (:163-168), RSA_LOSS_CCL CCLLoss (:171-186; param0 = margin, param1 = neg_weight).  InfoNCELoss (:157-160) is
 * RSA_LOSS_SSM with null log-probabilities.  Top1Loss (:66-78) is not offered: the reference's
 * forward modifies a sigmoid output in place and cannot be back-propagated.
 *
 * rsa_ssm_shared_loss: SampledSoftmaxLoss.forward when pos_score [B, L] and neg_score [B, n] have the SAME rank
 * (recstudio/model/loss_func.py:84-89): the L = n_pos positives of a row share its n negatives; padded
 * positives (+-inf) contribute nothing and are not counted.  Value + gradients in one pass. */
int rsa_ssm_shared_loss(const rsa_loss_args* args, rsa_stream_t stream);

/* out[0] = mean(row_loss[0..n_rows)) in a fixed summation order (two tiny launches). */
int rsa_mean_rows(const float* row_loss, int64_t n_rows, float* out, void* scratch, rsa_stream_t stream);

/* SoftmaxLoss.forward on a MATERIALISED score matrix -- recstudio/model/loss_func.py:41:
 * lse[m] = logsumexp(x[m, :]); softmax_scaled (nullable) [n_rows, n_cols] = softmax(x) * scale
 * (= d mean(lse) / d x for scale = 1/n_rows). */
int rsa_row_lse(const float* x, int64_t n_rows, int64_t n_cols, float* lse, float* softmax_scaled,
                float scale, rsa_stream_t stream);

/* Backward of the fused forward (shown for the inner-product scorer; RSA_SCORE_COS applies the cosine
 * chain rule instead) == what autograd
 * produces at recommender.py:636-639 (embedding_dense_backward + bmm backward):
 *   item_grad[neg_ids[m,j]] += up * dneg[m,j] * q_m     (skipped for id == item_pad_row: padding_idx)
 *   item_grad[pos_ids[m]]   += up * dpos[m]   * q_m
 *   query_grad[m]            = up * (dpos[m] * item[pos] + sum_j dneg[m,j] * item[neg_j])
 * item_grad: dense [n_items, dim], accumulated with atomics, caller zero-fills
 * (nullable).  item_grad_rows: nullable [M*(n+1), dim] row-sparse values, row
 * m*(n+1) = positive, m*(n+1)+1+j = negative j (COO values for ids
 * cat(pos, neg); no atomics).  query_grad: nullable [M, dim], overwritten.
 * upstream: nullable device scalar (d final / d loss), default 1. */
typedef struct rsa_backward_args {
  const float* item_table;
  int64_t n_items;
  int32_t dim;
  int32_t num_neg;
  const float* query;
  const int64_t* query_index;
  int64_t n_query_rows;
  const int64_t* pos_ids;      /* nullable */
  const int64_t* neg_ids;      /* [M, n] */
  int64_t n_queries;
  const float* dpos;           /* [M] (nullable iff pos_ids null) */
  const float* dneg;           /* [M, n] */
  const float* upstream;
  float* item_grad;
  float* item_grad_rows;
  float* query_grad;
  float* query_table_grad;     /* nullable [n_query_rows, dim], atomics, caller-zeroed: the query gradient accumulated
                                  at row query_index[m] -- the dense grad of a user-embedding query
                                  tower in the same launch; also the per-query accumulation of the sharded backward */
  int32_t query_table_pad_row; /* row of query_table_grad that receives no gradient (padding_idx; 0 for a RecStudio
                                  user table), -1 = none */
  int32_t item_pad_row;        /* item row that receives no gradient (padding_idx = 0 in RecStudio); -1 = none (an
                                  item-table shard that does not hold the global row 0) */
  int32_t score_mode;          /* rsa_score_mode: RSA_SCORE_IP (tuned kernels) or RSA_SCORE_COS / RSA_SCORE_EUC (plain kernel) */
  int32_t _pad;
} rsa_backward_args;

int rsa_fused_backward(const rsa_backward_args* args, rsa_stream_t stream);

/* Atomics-free, bit-reproducible form of the item-side scatter-add of a step:
 *     target[id] += upstream * sum_{e : id_e = id} d_e * query[qrow_e]      (rows id != pad_row)
 * over the elements of the step (with pos_ids: the positive of query m first, then its num_neg negatives; without: the
 * negatives only; d = dpos / dneg).
 * The (id, element) pairs are radix-sorted (the in-tree stable LSD sort, csrc/rsa_radix.hpp) and every run of equal ids is summed by one wave in
 * element order, then the row is read-modified-written once.  `target` [n_items, dim] is a zeroed dense gradient
 * (== the reference's weight.grad, recommender.py:636-639) or the weight table itself with upstream = -lr (plain
 * SGD in place).  dim in {64, 128, 256}; pos_ids / dpos nullable; query_index nullable (query row m);
 * upstream: nullable device scalar; pad_row < 0: none.  Elements whose id is NEGATIVE are dropped (empty slots of
 * the fixed-capacity shard exchange): nothing is read or written for them.  Workspace from the _workspace_bytes call. */
int64_t rsa_scatter_rows_sorted_workspace_bytes(int64_t n_queries, int32_t num_neg, int64_t n_items);

/* With exp_avg / exp_avg_sq given the accumulate becomes a LAZY ADAM update of every touched row (the update rule of
 * torch.optim.SparseAdam on the coalesced gradient g[id] = upstream * sum_e d_e * query[qrow_e]):
 *     m += (g - m)(1 - beta1);  v += (g^2 - v)(1 - beta2);  weight -= lr * sqrt(1 - beta2^step) / (1 - beta1^step) * m / (sqrt(v) + eps)
 * exp_avg / exp_avg_sq: [n_items, dim] optimizer state, updated in place; rows not in the step are untouched (lazy).
 * The row sums never leave registers: no gradient tensor, no coalesce pass.  step >= 1 is the 1-based step count. */
typedef struct rsa_rows_update_args {
  int64_t size;                /* sizeof(rsa_rows_update_args) */
  const float* query;          /* [n_query_rows, dim] */
  const int64_t* query_index;  /* nullable [n_queries] (null: query row m) */
  int64_t n_query_rows;
  int32_t dim;                 /* 64, 128 or 256 */
  int32_t has_pos;             /* rsa_rows_update_presorted: the workspace was sorted WITH positives (pos_ids given to the sort) */
  const int64_t* pos_ids;      /* nullable [n_queries] */
  const int64_t* neg_ids;      /* [n_queries, num_neg] */
  int64_t n_queries;
  int32_t num_neg;
  int32_t _pad;
  const float* dpos;           /* [n_queries] (nullable iff no positives) */
  const float* dneg;           /* [n_queries, num_neg] */
  const float* upstream;       /* nullable device scalar */
  int64_t n_items;             /* rows of target */
  int64_t pad_row;             /* < 0: none */
  float* target;               /* [n_items, dim]: a zeroed dense gradient, or the weight table (upstream = -lr: SGD in place) */
  float* exp_avg;              /* nullable: lazy Adam state (then target = the weights) */
  float* exp_avg_sq;
  float lr, beta1, beta2, eps;
  int64_t step;
  uint8_t* solo;               /* rsa_sort_step_elements: nullable [n_queries * (num_neg + has_pos)] out */
  void* workspace;
  int64_t workspace_bytes;     /* >= rsa_scatter_rows_sorted_workspace_bytes(n_queries, num_neg, n_items) */
} rsa_rows_update_args;

/* sort + apply in one call (the former rsa_scatter_rows_sorted / rsa_adam_rows_sorted) */
int rsa_rows_update_sorted(const rsa_rows_update_args* args, rsa_stream_t stream);

/* The same in two steps, for a forward that updates the rows only one element touches itself
 * (rsa_fused_args.solo_flags):
 *   rsa_sort_step_elements   sorts the step's (item id, element) pairs into `workspace` (reads pos_ids, neg_ids, n_queries,
 *     num_neg, n_items, pad_row) and, with `solo` (uint8 [n_queries * (num_neg + has_pos)], element order
 *     m * (num_neg + has_pos) + c), classifies them: solo[e] = 1 iff element e is the only one on its row and the row is
 *     neither pad_row nor a negative (dropped) id; those elements are also flagged inside the workspace;
 *   rsa_rows_update_presorted  is the apply pass over that workspace: every run of equal ids
 *     summed by one wave, the row read-modified-written once -- except the flagged elements, whose rows it leaves alone
 *     (has_pos says whether the sort saw positives; pos_ids / neg_ids are not read).  With the lazy-Adam fields: over a
 *     workspace sorted WITHOUT `solo` -- the sort does not read the weights, so a trainer can issue it, and the sampling in
 *     front of it, for the next batch on another stream while the current step runs.
 * Nothing may touch the workspace in between.  dim in {64, 128, 256}. */
int rsa_sort_step_elements(const rsa_rows_update_args* args, rsa_stream_t stream);
int rsa_rows_update_presorted(const rsa_rows_update_args* args, rsa_stream_t stream);

/* ---- The whole in-place SGD training step of a BPR two-tower model (nn.Embedding user and item tables, inner product,
 * BPRLoss, num_neg == 64) as TWO calls -- what `fit(train.fused_optimizer: 'sgd')` issues per batch.  Replaces, per step of
 * recstudio/model/basemodel/recommender.py:596-646: sampler.forward (ann/sampler.py:86-111 / :243-258), the two tower
 * look-ups, score_func, BPRLoss, loss.backward() and optimizer.step() of torch.optim.SGD (no momentum / weight decay).
 *   rsa_bpr_sgd_prepare  the part that does not read the weights: the negatives (the device random stream, "Philox
 *     state"; sampler RSA_SAMPLER_UNIFORM: ids in [1, n_items); RSA_SAMPLER_POPULAR: `pop`; RSA_SAMPLER_GIVEN: neg_ids is
 *     an INPUT) and ONE radix sort of the step's (item id, element) pairs followed by its (user id, query) pairs (user keys
 *     behind the item keys; item_workspace), + the solo classification of the item part (solo).  The negatives are drawn by
 *     the sort's first launch (same ids as rsa_sample_uniform / rsa_sample_popular, stored to neg_ids).
 *     A trainer issues it for batch k + 1 on a second stream while batch k's apply runs.
 *   rsa_bpr_sgd_apply    forward + loss + update: rsa_fused_sample_gather_score with the ids given, fused BPR epilogue, user
 *     gradient accumulated in the forward, solo item rows updated in the forward; the shared item rows and the user rows by
 *     the sorted apply pass over the two parts of the sorted pairs.  loss_out = the batch's mean BPR loss.
 * Bit-identical to the same sequence issued entry point by entry point.  dim in {64, 128, 256}. */
typedef struct rsa_bpr_sgd_args {
  int64_t size;                /* sizeof(rsa_bpr_sgd_args) */
  float* item_table;           /* [n_items, dim], updated in place */
  int64_t n_items;
  float* user_table;           /* [n_users, dim], updated in place */
  int64_t n_users;
  int32_t dim;
  int32_t num_neg;             /* 64 */
  const int64_t* user_ids;     /* [n_queries] */
  const int64_t* pos_ids;      /* [n_queries] */
  int64_t n_queries;
  int32_t sampler;             /* RSA_SAMPLER_UNIFORM / RSA_SAMPLER_POPULAR / RSA_SAMPLER_GIVEN (prepare) */
  int32_t _pad;
  const rsa_popular_args* pop; /* HOST pointer, RSA_SAMPLER_POPULAR: the tables (ids / logp / numel / philox fields ignored) */
  uint64_t seed, offset;       /* "Philox state" of the draw (prepare) */
  uint32_t grid_threads;
  uint32_t _pad2;
  uint64_t elem_base;
  const float* step_scale;     /* device scalar: -lr */
  int64_t* neg_ids;            /* [n_queries, num_neg]: out of prepare (in with RSA_SAMPLER_GIVEN), in of apply */
  uint8_t* solo;               /* [n_queries, 1 + num_neg]: out of prepare, in of apply */
  void* item_workspace;        /* rsa_scatter_rows_sorted_workspace_bytes(n_queries, num_neg + 1, n_items): item AND user elements (ABI 11) */
  int64_t item_workspace_bytes;
  void* user_workspace;        /* unused since ABI 11 (nullable) */
  int64_t user_workspace_bytes;
  float* pos_score;            /* [n_queries] out */
  float* neg_score;            /* [n_queries, num_neg] out */
  float* row_loss;             /* [n_queries] out */
  float* dpos;                 /* [n_queries] out */
  float* dneg;                 /* [n_queries, num_neg] out */
  float* query_grad;           /* [n_queries, dim] out */
  const float* ones;           /* [n_queries] of 1.0f (the coefficients of the user-row apply) */
  float* loss_out;             /* [1] out */
  void* reduce_scratch;        /* rsa_scratch_bytes(), of the stream apply runs on */
  int64_t uniform_high;        /* RSA_SAMPLER_UNIFORM: ids drawn in [1, uniform_high); 0 = n_items (UniformSampler(num_items) of a
                                  model whose table has num_items rows, mf/bpr.py:24-25) */
} rsa_bpr_sgd_args;
int rsa_bpr_sgd_prepare(const rsa_bpr_sgd_args* args, rsa_stream_t stream);
int rsa_bpr_sgd_apply(const rsa_bpr_sgd_args* args, rsa_stream_t stream);

/* embedding_dense_backward: dst[ids[i]] += src[i] for ids != 0 (padding_idx=0).
 * Used for the user-table gradient.  dst [n_rows, dim] caller-zeroed. */
int rsa_scatter_add_rows(const float* src, const int64_t* ids, int64_t numel, int32_t dim,
                         float* dst, int64_t n_rows, rsa_stream_t stream);

/* SeqDataset batch materialisation + item_encoder(in_item_id) --
 * recstudio/data/dataset.py:1418-1439 (ragged [start,end) slices of the user-sorted
 * item column, right-padded with 0) + recstudio/model/seq/sasrec.py:42.
 * out_ids [B, max_len] int64 (nullable), out_rows [B, max_len, dim] (nullable),
 * out_len [B] int64 (nullable).  Segments longer than max_len keep their LAST
 * max_len items (dataset.py:1400,1409 windows to the most recent max_seq_len). */
typedef struct rsa_seg_gather_args {
  int64_t size;                /* sizeof(rsa_seg_gather_args) */
  const float* item_table;     /* nullable when out_rows is null */
  int64_t n_items;
  int32_t dim;
  int32_t max_len;
  const int64_t* flat_item_ids;
  int64_t n_flat;
  const int64_t* seg_start;
  const int64_t* seg_end;
  int64_t n_seg;
  int64_t* out_ids;            /* nullable [n_seg, max_len] */
  float* out_rows;             /* nullable [n_seg, max_len, dim] */
  int64_t* out_len;            /* nullable [n_seg] */
} rsa_seg_gather_args;
int rsa_seg_gather(const rsa_seg_gather_args* args, rsa_stream_t stream);

/* aux[r] of every row of a [n_rows, dim] table, the per-row operand of the cosine / Euclidean full-catalog scores:
 * RSA_SCORE_COS -> 1 / ||row||_2 (inf for a zero row: the reference divides by zero too, scorer.py:21-24),
 * RSA_SCORE_EUC -> ||row||_2^2 (scorer.py:28-34).  dim % 4 == 0. */
int rsa_row_sqnorm(const float* table, int64_t n_rows, int32_t dim, int32_t score_mode, float* out, rsa_stream_t stream);

/* Full-catalog scoring -- InnerProductScorer / CosineScorer / EuclideanScorer ([B,d],[N,d]) case, scorer.py:16, :19-34,
 * called from baseretriever.py:183-186 (training, FullScoreLoss) and :384 (topk).
 * Scores query [B, dim] against item_table rows [1, n_items) (item_vector =
 * weight[1:], baseretriever.py:122-123) with the fp32 MFMA
 * (v_mfma_f32_32x32x2_f32), never materialising [B, N-1] unless `scores` is given.
 *   scores : nullable [B, n_items-1] out (reference-style materialisation)
 *   lse    : nullable [B] out, logsumexp over the catalog (SoftmaxLoss, loss_func.py:41)
 *   topk_val / topk_idx : nullable [B, k] out, k <= 1024: largest k scores in descending order,
 *     ids are ITEM ids (1-based, baseretriever.py:385); equal scores -> smaller id first.
 *   dim must be 32, 64 or 128 (RSA_ERR_UNSUPPORTED otherwise).
 *   score_mode RSA_SCORE_COS / RSA_SCORE_EUC: the dot products of a tile are turned into cosine / negative squared
 *     distance in the tile epilogue, before logsumexp / filter / store:  cos = dot * item_aux * query_aux,
 *     euc = 2 dot - item_aux - query_aux, with item_aux [n_items - 1 + 64] (entry i-1 belongs to item row i; 16-byte
 *     aligned; the 64 trailing floats are padding the last tile of an item range may read -- any values) and
 *     query_aux [B] from rsa_row_sqnorm of the same mode; both null for RSA_SCORE_IP.
 *   workspace: device scratch of rsa_fullscore_workspace_bytes(B, n_items, k) bytes. */
int64_t rsa_fullscore_workspace_bytes(int64_t n_query, int64_t n_items, int32_t k);
typedef struct rsa_fullscore_args {
  int64_t size;                /* sizeof(rsa_fullscore_args) */
  const float* item_table;
  int64_t n_items;
  int32_t dim;
  int32_t score_mode;
  const float* query;
  int64_t n_query;
  float* scores;               /* nullable */
  float* lse;                  /* nullable */
  float* topk_val;             /* nullable */
  int64_t* topk_idx;           /* nullable */
  int32_t k;
  int32_t _pad;
  const float* item_aux;       /* nullable */
  const float* query_aux;      /* nullable */
  void* workspace;
  int64_t workspace_bytes;
} rsa_fullscore_args;
int rsa_fullscore(const rsa_fullscore_args* args, rsa_stream_t stream);

/* Backward of logsumexp over the full catalog (SoftmaxLoss, loss_func.py:39-47, when the forward was
 * rsa_fullscore(..., lse) and [B, N-1] was never written):  probs[b, i-1] = row_scale[b] * exp(<q_b, item_i> - lse[b])
 * = row_scale[b] * softmax_i, recomputed by the same MFMA kernel.  d lse/d query = probs @ items[1:],
 * d lse/d items[1:] = probs^T @ query are then plain library GEMMs on the caller's side.
 * row_scale may be null (1).  probs: [n_query, n_items-1]. */
int rsa_fullscore_softmax(const float* item_table, int64_t n_items, int32_t dim, const float* query,
                          int64_t n_query, const float* lse, const float* row_scale, float* probs,
                          rsa_stream_t stream);

/* The same pass that ALSO leaves query_grad [n_query, dim] = probs @ items[1:] (d lse/d query times row_scale): the
 * tile of probs the epilogue has just produced is multiplied with the item tile staged in LDS on the matrix cores, so
 * one of the two backward GEMMs needs no second pass over the 4 * n_query * n_items bytes of probs.  The per-item-range
 * partial sums go through `workspace` (rsa_fullscore_softmax_dq_workspace_bytes) and are added in range order
 * (reproducible).  probs is still written (d lse/d items = probs^T @ query remains a library GEMM on the caller's side). */
int64_t rsa_fullscore_softmax_dq_workspace_bytes(int64_t n_query, int64_t n_items, int32_t dim);
int rsa_fullscore_softmax_dq(const float* item_table, int64_t n_items, int32_t dim, const float* query,
                             int64_t n_query, const float* lse, const float* row_scale, float* probs,
                             float* query_grad, void* workspace, int64_t workspace_bytes, rsa_stream_t stream);

/* probs == NULL (ABI 10): the softmax tile feeds the second product and is NOT written -- together with
 * rsa_fullscore_softmax_dw below a backward that never holds [B, N] ("only if [B, N] is never written"). */

/* FLASH forward of the full softmax (ABI 10): lse[q] = logsumexp_i <q, item_i> over rows 1 .. n_items - 1 AND
 * query_grad[q, :] = d lse[q] / d q = softmax_q @ items[1:] in ONE pass over the catalog -- the softmax tile is formed against a
 * running per-query reference that only moves when a tile's maximum exceeds it by more than 8 (the dQ accumulators and the running
 * sum are then rescaled: rare after the first tiles), per-item-range records (reference, sum, unnormalised sum P * item) are merged
 * in range order (reproducible).  A training step of SoftmaxLoss (loss_func.py:39-47 over scorer.py:16) is then this call (two
 * products of 2 B N d flop) + rsa_fullscore_softmax_dw (two more): d loss/d query = upstream[q] * query_grad[q], no [B, N] matrix,
 * no second pass for the query gradient.  workspace: rsa_fullscore_lse_grad_workspace_bytes.  dim in {32, 64, 128}. */
int64_t rsa_fullscore_lse_grad_workspace_bytes(int64_t n_query, int64_t n_items, int32_t dim);
int rsa_fullscore_lse_grad(const float* item_table, int64_t n_items, int32_t dim, const float* query, int64_t n_query,
                           float* lse, float* query_grad, void* workspace, int64_t workspace_bytes, rsa_stream_t stream);

/* d lse / d items WITHOUT probs (ABI 10): item_grad[i, :] = sum_b row_scale[b] * exp(<q_b, item_i> - lse[b]) * q_b for rows
 * i = 1 .. n_items - 1, item_grad[0, :] = 0 -- the reference's autograd through loss_func.py:39-47 over scorer.py:16.  Item-
 * stationary: a wave keeps its 32 item rows in registers, recomputes the [32 batch rows x 32 items] score tile per batch chunk
 * on the fp32 matrix cores, forms the scaled softmax in its accumulator registers and feeds them back as the operand of
 * item_grad^T += P^T Q (the accumulator layout IS the operand layout when the chunk's rows are taken in that order) -- the
 * [B, N] matrix exists nowhere, every output row is written once, no atomics.  flops 4 B N d, HBM bytes 8 N d.  row_scale may
 * be NULL (1).  item_table / query / item_grad 16-byte aligned, dim in {32, 64, 128}. */
int rsa_fullscore_softmax_dw(const float* item_table, int64_t n_items, int32_t dim, const float* query, int64_t n_query,
                             const float* lse, const float* row_scale, float* item_grad, rsa_stream_t stream);

/* The other backward GEMM, d lse / d items[1:] = probs^T @ query (ATen's mm backward of scorer.py:16 under loss_func.py:39-47),
 * item-stationary on the fp32 matrix cores: out[i, :] = sum_b probs[b, i] * query[b, :] for i in [0, n_cols) -- every output row
 * written once, no atomics, no split-K partials.  probs [n_query, ld] (ld >= n_cols: its row stride), query [n_query, dim],
 * out [n_cols, dim] (the caller passes item_grad + dim: row 1 of the table gradient; row 0 stays zero), dim in {32, 64, 128}. */
int rsa_probs_t_query(const float* probs, int64_t n_query, int64_t n_cols, int64_t ld, const float* query, int32_t dim,
                      float* out, rsa_stream_t stream);

/* torch.topk(values, k) over the last dim of a [n_rows, n_cols] matrix (k <= 1024): values in descending
 * order and their COLUMN indices (equal values -> smaller column first).  Used by the 'dns' sampling method
 * (baseretriever.py:343-347: the hardest num_neg[1] of a sampled pool of num_neg[0] negatives). */
int rsa_row_topk(const float* values, int64_t n_rows, int64_t n_cols, int32_t k, float* out_val,
                 int64_t* out_col, rsa_stream_t stream);

/* History exclusion of BaseRetriever.topk -- baseretriever.py:386-392: candidates (sorted
 * descending, [n_query, n_cand], item ids) that appear in user_hist [n_query, hist_len] (0-padded)
 * get score -inf; out = the k best survivors in order (masked ones, as -inf, fill the tail only
 * when fewer than k survive). */
int rsa_topk_mask_history(const float* cand_val, const int64_t* cand_idx, int32_t n_cand,
                          const int64_t* user_hist, int32_t hist_len, int64_t n_query, int32_t k,
                          float* out_val, int64_t* out_idx, rsa_stream_t stream);

/* ---- Row-sharded item table (BASELINE.json configs[3]; no counterpart in the reference, whose
 * only multi-device mode re-broadcasts whole tables every step, utils/data_parallel.py:106-159).
 * Shard g owns item rows [g*rows_per_shard, (g+1)*rows_per_shard); with rows_per_shard == 0 the rows
 * are INTERLEAVED instead: owner(id) = id % n_shards, local row = id / n_shards (every owner holds
 * 1/n_shards of any id range, so ids ordered by popularity do not make a hot shard).  An "element" is one (query,
 * item) pair of the [n_queries, 1+num_neg] matrix whose column 0 is pos_ids and columns 1.. are
 * neg_ids.  Elements are counting-sorted by owner into contiguous per-owner segments and travel
 * as one packed key = (query_base + query) << 32 | local_row; the fp32 scores come back in the
 * same order; `positions[i]` is the slot of element i's score in the home buffer
 * [pos_score (n_queries) | neg_score (n_queries x num_neg)], which rsa_scatter_f32 fills. */
int rsa_shard_count(const int64_t* pos_ids, const int64_t* neg_ids, int64_t n_queries, int32_t num_neg,
                    int64_t rows_per_shard, int32_t n_shards, int32_t* counts /* [n_shards] out */,
                    rsa_stream_t stream);
/* cursor [n_shards]: IN the exclusive prefix sum of counts (segment starts), clobbered. */
int rsa_shard_route(const int64_t* pos_ids, const int64_t* neg_ids, int64_t n_queries, int32_t num_neg,
                    int64_t rows_per_shard, int32_t n_shards, int64_t query_base, int32_t* cursor,
                    int64_t* keys, int64_t* positions, rsa_stream_t stream);
int rsa_shard_unpack(const int64_t* keys, int64_t numel, int64_t* local_rows, int64_t* query_index,
                     rsa_stream_t stream);
/* dst[positions[i]] = src[i] for positions[i] >= 0 (exact exchange) */
int rsa_scatter_f32(const float* src, const int64_t* positions, int64_t numel, float* dst, rsa_stream_t stream);
/* dst[i] = positions[i] >= 0 ? src[positions[i]] : 0: d loss/d score in routed order for the gradient exchange. */
int rsa_gather_f32(const float* src, const int64_t* positions, int64_t numel, float* dst, rsa_stream_t stream);

/* ---- Version 2 of the fixed-capacity exchange (ABI v5).  What changed against rsa_shard_route_fixed + rsa_scatter_f32 +
 * rsa_pairwise_loss: (a) the negatives are drawn INSIDE the routing pass (same Philox element mapping as
 * rsa_sample_uniform / rsa_sample_popular, so the ids are the ones the stand-alone samplers -- and the reference's
 * torch.randint / torch.rand + searchsorted, sampler.py:102-104, :246-258 -- produce); the [n_queries, num_neg] id tensor
 * is written only on request; (b) a segment is SELF-DESCRIBING: RSA_SHARD_HDR 8-byte header words {live keys, elements
 * the source rank dropped in this step -- in bank 0 of every (slice, owner), 0 elsewhere} precede its keys, so the slack is never filled, owners skip the tiles past the
 * count, and after the key all-to-all every rank knows the job-wide number of dropped elements of the step without a
 * collective of its own; (c) the home side keeps ONE int32 per element (slot_of: where its key sits in the send buffer
 * == where its score sits in the returned buffer; -1 = dropped) instead of an 8-byte position per slot, and gathers
 * through it inside the loss kernel (rsa_shard_home) -- no scatter pass, no separate loss / mean launches; (d) one
 * routing launch covers all pipelined query slices.
 * A DROPPED element (its owner's segment was full) has a defined outcome: no key, no score (reads as 0 where scores are
 * materialised), no term in a fused loss, d loss / d score = 0 (nothing is sent for it); and the owner-side update
 * scale of a step in which ANY rank dropped anything is 0 (rsa_shard_unpack_segments), so such a step leaves every
 * weight untouched. */
#define RSA_SHARD_HDR 2   /* 8-byte header words per segment: [0] live keys, [1] the source's dropped total of the step */

typedef struct rsa_shard_route_args {
  const int64_t* pos_ids;      /* [n_queries] global item ids (column 0 of the element matrix) */
  int64_t* neg_ids;            /* [n_queries, num_neg]: INPUT when sampler == RSA_SAMPLER_GIVEN, else nullable OUTPUT */
  float* neg_logp;             /* nullable [n_queries, num_neg] out (POPULAR): log pop_prob[id] */
  float* pos_logp;             /* nullable [n_queries] out: log pop_prob[pos_ids] (needs pop_prob) */
  int64_t n_queries;
  int32_t num_neg;
  int32_t sampler;             /* rsa_sampler_kind */
  int32_t n_slices;            /* pipelined slices (>= 1): slice c = the c-th of n_slices contiguous ranges of the launch's
                                  workgroups -- an arbitrary, run-independent partition of the step's elements */
  int32_t n_shards;            /* G <= 64 */
  int32_t n_banks;             /* >= 1: what goes to one owner in one slice is split over n_banks segments, each filled through its
                                  own cursor by its own share of the launch's workgroups (one cursor per owner makes every
                                  workgroup queue up on one atomic address); n_slices * n_shards * n_banks <= 4096 */
  int32_t _pad0;
  int64_t rows_per_shard;      /* owner(id) = min(id / rows_per_shard, G - 1); 0: interleaved, owner = id % G, row = id / G */
  int64_t query_base;          /* global index of query 0 (rank * n_queries): goes into the keys */
  int64_t capacity;            /* keys per (slice, owner, bank) segment */
  int64_t n_items;             /* catalog size: uniform ids are drawn from [1, n_items) */
  uint64_t seed, offset;       /* Philox state, see "Philox state" (sampler != GIVEN) */
  uint32_t grid_threads;
  uint32_t _pad;
  uint64_t elem_base;
  const float* table;          /* POPULAR: as in rsa_fused_args */
  const float* pop_prob;
  const int32_t* guide;
  const float* table_prob;
  const float* cdf_lut;
  const float* cdf_lines;
  int32_t guide_log2;
  int32_t lines_log2;
  int64_t* send_keys;          /* [n_slices][n_shards][n_banks][RSA_SHARD_HDR + capacity] out; key = (query_base + query) << 32 | local row.
                                  NULL: count only (nothing is sampled into the outputs, only counts_out is written) */
  int32_t* slot_of;            /* [n_queries * (1 + num_neg)] out, element e = query * (1 + num_neg) + column */
  int32_t* cursors;            /* [(n_slices * n_shards * n_banks + 33) * 32] int32 device scratch (one 128-byte line per cursor
                                  and ticket), zeroed ONCE by the caller (self-resetting) */
  int32_t* counts_out;         /* nullable [n_slices * n_shards * n_banks]: exact element counts per segment -- calibration */
  int32_t skip_pos;            /* != 0: the positives are NOT routed (slot_of of column 0 = -1): the owner-side BPR step
                                  (rsa_shard_owner_bpr_forward) scores them from the gathered ids instead */
  int32_t group_by_query;      /* != 0 (needs skip_pos, an in-kernel sampler and rsa_shard_route_query_groups(...) > 0): a workgroup's
                                  share of a segment is written query by query -- every query's elements for an owner form ONE
                                  contiguous run of the segment (rsa_shard_owner_bpr_args.keys_grouped: no sort by query there) */
  /* DETERMINISTIC slots (ABI 9): with deterministic != 0 no atomic decides where an element's key lands -- a count pass, an
   * exclusive prefix over the workgroups in launch order and the routing pass (three launches instead of one: +20-45 us at
   * 4.2 M elements): the send buffer, and with it every sum the owners form in slot order, is then bit-identical run to run.
   * wg_scratch: caller-owned, 2 * rsa_shard_route_workgroups(args) * n_shards ints (contents irrelevant). */
  int32_t deterministic;
  int32_t _pad1;
  int32_t* wg_scratch;
  int64_t wg_scratch_ints;
  int32_t* extra_dropped;      /* nullable device word: elements of this step that were dropped OUTSIDE this call (the fixed-capacity row
                                  look-up of an item-tower query encoder).  Added to the dropped total the segment headers carry -- the
                                  owners gate the step like any routing overflow -- and reset to 0 */
} rsa_shard_route_args;
int64_t rsa_shard_segment_stride(int64_t capacity);    /* RSA_SHARD_HDR + capacity: 8-byte words per segment */
int rsa_shard_sample_route(const rsa_shard_route_args* args, rsa_stream_t stream);
/* workgroups the call above launches for these arguments (host arithmetic only) */
int64_t rsa_shard_route_workgroups(const rsa_shard_route_args* args);
/* Queries per workgroup if a call with these parameters can route query-grouped (num_neg a divisor of 1024 and >= 64, the
 * random call's grid a multiple of 1024 threads, the rank's element base a multiple of num_neg; unroll = 4, or 2 for the
 * 64-bit draws of catalogs beyond 2^28 items), else 0. */
int32_t rsa_shard_route_query_groups(int32_t num_neg, uint32_t grid_threads, uint64_t elem_base, int32_t unroll,
                                     int32_t n_shards);

/* Owner side: scores[i] = <query[key_i >> 32], item_table[key_i & 0xffffffff]> for every live slot of the received
 * segments keys [n_segments][stride] (the fused gather+score kernel of rsa_fused_sample_gather_score reading the
 * segments directly; slots past a segment's count are not read and their score is not written; whole tiles past the
 * count are skipped).  step_dropped (nullable int32 device word) <- sum over the segments of header word 1 (the job-wide
 * dropped count of the step when the segments come from all ranks); overflow_sticky (nullable) += the same. */
int rsa_shard_score_segments(const float* item_table, int64_t n_rows, int32_t dim, const float* query,
                             int64_t n_query_rows, const int64_t* keys, int64_t n_segments, int64_t stride, float* scores,
                             int32_t* step_dropped, int32_t* overflow_sticky, rsa_stream_t stream);

/* Home side: gather the returned scores through slot_of and evaluate the loss in the same launch (one wave per query).
 * loss 0: pos_score / neg_score only.  loss 1: BPRLoss (loss_func.py:55-59).  loss 2: SampledSoftmaxLoss (:80-90) with
 * nullable log-probabilities.  loss_out = sum(row_loss) / mean_den (mean_den = n_queries for the local mean, the job-wide
 * query count for this rank's share of the global mean); dpos / dneg / d_send = d loss_out / d score, d_send in routed
 * order (index = slot) ready for the gradient all-to-all.  Every output is nullable except neg_score for loss 0. */
typedef struct rsa_shard_home_args {
  const float* scores;         /* returned score buffer, indexed by slot */
  const int32_t* slot_of;      /* [n_queries * (1 + num_neg)] */
  int64_t n_queries;
  int32_t num_neg;
  int32_t loss;
  const float* pos_logp;       /* loss 2, nullable [n_queries] */
  const float* neg_logp;       /* loss 2, nullable [n_queries, num_neg] */
  int64_t mean_den;
  float* pos_score;            /* nullable [n_queries] out */
  float* neg_score;            /* nullable [n_queries, num_neg] out */
  float* row_loss;             /* nullable [n_queries] out */
  float* loss_out;             /* nullable [1] out (needs reduce_scratch) */
  float* dpos;                 /* nullable [n_queries] out */
  float* dneg;                 /* nullable [n_queries, num_neg] out */
  float* d_send;               /* nullable, slot-indexed out */
  void* reduce_scratch;        /* rsa_scratch_bytes() bytes, zeroed once */
} rsa_shard_home_args;
int rsa_shard_home(const rsa_shard_home_args* args, rsa_stream_t stream);

/* d_send[slot_of[e]] = (dpos | dneg)[e] for a loss evaluated outside rsa_shard_home (dropped elements send nothing). */
int rsa_shard_scatter_slots(const float* dpos, const float* dneg, const int32_t* slot_of, int64_t n_queries,
                            int32_t num_neg, float* d_send, rsa_stream_t stream);

/* Owner side of the backward: (local row, query index) per slot of the received segments, -1 past a segment's count
 * (rsa_rows_update_sorted drops negative ids); scale_out (nullable, 2 floats) = {gate * (scale_in ? scale_in[0] : 1),
 * gate} with gate = step_dropped[0] != 0 ? 0 : 1 -- the `upstream` scalars of the item-side and the query-side
 * sorted scatter, so that a step in which any rank dropped an element updates nothing. */
int rsa_shard_unpack_segments(const int64_t* keys, int64_t n_segments, int64_t stride, int64_t* local_rows,
                              int64_t* query_index, const float* scale_in, const int32_t* step_dropped, float* scale_out,
                              rsa_stream_t stream);

/* The whole owner side of the sharded backward on received segments (replaces rsa_shard_unpack_segments + two
 * rsa_rows_update_sorted calls, i.e. the ATen sequence autograd runs for the reference's
 * item_encoder(neg ids) / score_func backward, recommender.py:636-639, on the rows this rank owns):
 *     qgrad_all[q]   += gate * sum_{slots of query q} d[slot] * item_local[row(slot)]
 *     item_target[r] += gate * item_scale * sum_{slots on row r} d[slot] * q_all[query(slot)]      (r != item_pad_row)
 * with gate = step_dropped[0] != 0 ? 0 : 1 (scale_out <- {gate * item_scale, gate}).  item_target == item_local: plain
 * SGD in place (item_scale = -lr) -- rows that ONE slot of the step touches are rewritten by the walk that reads them
 * for the query gradient, the others by the sorted apply pass; otherwise item_target is this rank's block of the dense
 * gradient.  Deterministic (no float atomics).  dim in {64, 128, 256}.  `workspace`: caller-owned,
 * rsa_shard_backward_workspace_bytes(n_segments, stride, n_query_rows) bytes. */
typedef struct rsa_shard_backward_args {
  const float* item_local;     /* [n_rows, dim] this rank's rows */
  int64_t n_rows;
  int32_t dim;
  const float* q_all;          /* [n_query_rows, dim] the gathered queries of all ranks */
  int64_t n_query_rows;
  const int64_t* keys;         /* [n_segments, stride] received segments (RSA_SHARD_HDR header words + keys) */
  int64_t n_segments;
  int64_t stride;
  const float* d_owner;        /* [n_segments * stride] d loss / d score in slot order (slack: never read) */
  float* item_target;          /* item_local itself (in-place SGD) or the [n_rows, dim] gradient block */
  const float* item_scale;     /* device scalar, nullable (1) */
  const int32_t* step_dropped; /* device word, nullable (no gate) */
  float* scale_out;            /* [2] device: {gate * item_scale, gate} */
  float* qgrad_all;            /* [n_query_rows, dim], accumulated into */
  int64_t item_pad_row;        /* row that never receives gradient (-1: none) */
  void* workspace;
  int64_t workspace_bytes;
} rsa_shard_backward_args;
int64_t rsa_shard_backward_workspace_bytes(int64_t n_segments, int64_t stride, int64_t n_query_rows);
int rsa_shard_backward_segments(const rsa_shard_backward_args* args, rsa_stream_t stream);

/* The stock BPR training step (loss_func.py:55-59 on the scores of baseretriever.py:153-171, and its backward) evaluated
 * ON THE OWNERS of the negatives -- the item rows of a step are then read once instead of twice (scoring pass + backward
 * pass), and neither scores nor score gradients cross the fabric:
 *   rsa_shard_pos_score          out[i] = q_all[i] . item_local[pos_rows[i]] for the positives this rank owns (pos_rows[i]
 *                                >= 0: the local row; < 0: another rank's; derived from the gathered global ids in the same
 *                                launch when pos_ids is given), 0 for the others -- summed over the ranks
 *                                (a 4-byte-per-query all-reduce) it is every positive's score on every rank;
 *   rsa_shard_owner_bpr_forward  over the negatives this rank received (the positives are NOT routed in this protocol):
 *                                score, loss term -logsigmoid(pos - neg) / num_neg, d = sigmoid(neg - pos) / (num_neg *
 *                                mean_den) (-> d_slots[slot]), qgrad_all[q] += gate * sum d * row, rows that one element of
 *                                the step touches updated in place (item_target == item_local, scale = gate * item_scale);
 *                                dsum_part[q] = the query's sum of d here, loss_part = this rank's share of the mean loss.
 *                                Also publishes the step's dropped total (the received headers' word 1) into step_dropped /
 *                                overflow_sticky and the update scales into scale_out, like rsa_shard_score_segments +
 *                                rsa_shard_backward_segments do in the score-at-home protocol;
 *   rsa_shard_owner_bpr_finish   with dsum_all = the ranks' dsum_part summed (second small all-reduce): the positives this
 *                                rank owns -- d loss/d pos = -dsum_all[q] (-> d_slots[slots + q]), qgrad_all[q] += gate *
 *                                dpos * row, the row updated in place when it is the positive's alone -- and then the
 *                                sorted apply pass for every row several elements (negatives or positives) touch.
 * Same workspace (rsa_shard_backward_workspace_bytes) and the same args for both calls of a step.  dim in {64, 128, 256}. */
typedef struct rsa_shard_owner_bpr_args {
  const float* item_local;     /* [n_rows, dim] */
  int64_t n_rows;
  int32_t dim;
  int32_t num_neg;             /* negatives per query of the step (the 1/n of BPRLoss) */
  const float* q_all;          /* [n_query_rows, dim] */
  int64_t n_query_rows;
  const int64_t* keys;         /* [n_segments, stride] received segments (negatives only) */
  int64_t n_segments;
  int64_t stride;
  const int64_t* pos_rows;     /* [n_query_rows] local row of each query's positive on this rank, < 0: not owned */
  const float* pos_score;      /* [n_query_rows] */
  int64_t mean_den;            /* queries of the step over all ranks (the mean of BPRLoss) */
  float* item_target;          /* item_local (SGD in place) or the [n_rows, dim] gradient block */
  const float* item_scale;     /* device scalar, nullable (1) */
  int32_t* step_dropped;       /* device word, written */
  int32_t* overflow_sticky;    /* device word, accumulated */
  float* scale_out;            /* [2] device */
  float* qgrad_all;            /* [n_query_rows, dim], accumulated into */
  float* d_slots;              /* [n_segments * stride + n_query_rows] OUT */
  float* dsum_part;            /* [n_query_rows] OUT */
  float* loss_part;            /* [1] OUT, nullable */
  void* reduce_scratch;        /* rsa_scratch_bytes() bytes, zeroed once (with loss_part) */
  int64_t item_pad_row;
  void* workspace;
  int64_t workspace_bytes;
  int32_t keys_grouped;        /* != 0: the segments were routed with group_by_query (every query's elements for this owner are
                                  one contiguous run of one segment): the sort by query is skipped */
  int32_t finish_parts;        /* rsa_shard_owner_bpr_finish only: 0 = all of it; 1 = the positives only (qgrad_all is complete
                                  after this call); 2 = only the sorted apply pass of the shared rows (after a call with 1).  A
                                  caller can then issue what depends on qgrad_all -- the reduce-scatter, the query tower's
                                  update -- on another stream, beside the apply pass */
  int32_t forward_parts;       /* rsa_shard_owner_bpr_forward only: 0 = all of it; 1 = only what reads nothing but the received
                                  keys and pos_rows -- the sort by row, the solo classification, the queries' runs -- into the
                                  workspace (q_all, pos_score, qgrad_all and the outputs may be null): a caller can issue it
                                  for the NEXT step, behind that step's key exchange, on another stream; 2 = the rest (update
                                  scales from the headers, then the pass over the rows) over a workspace a call with 1 filled */
  /* rsa_shard_owner_ssm_forward / _finish (ABI 9): */
  const float* logq_rows;      /* nullable [n_rows]: log-probability under the sampler of each LOCAL row's item (null: 0, the uniform sampler) */
  float* run_max;              /* [n_query_rows] OUT of the forward: max over this owner's slots of z = score - log q (-inf: none) */
  float* run_sum;              /* [n_query_rows] OUT: sum exp(z - run_max) */
  float* run_acc;              /* [n_query_rows, dim] OUT: sum exp(z - run_max) * row */
} rsa_shard_owner_bpr_args;
int rsa_shard_pos_score(const float* item_local, int64_t n_rows, int32_t dim, const float* q_all, int64_t n_query_rows,
                        int64_t* pos_rows, float* out, const int64_t* pos_ids, int64_t rows_per_shard, int32_t n_shards,
                        int32_t rank, rsa_stream_t stream);   /* pos_ids != NULL: GLOBAL ids in, pos_rows is written (the owner /
                                                                  local-row rule of rsa_shard_sample_route: rows_per_shard == 0 =
                                                                  interleaved rows); NULL: pos_rows is the input */
int rsa_shard_owner_bpr_forward(const rsa_shard_owner_bpr_args* args, rsa_stream_t stream);
int rsa_shard_owner_bpr_finish(const rsa_shard_owner_bpr_args* args, const float* dsum_all, rsa_stream_t stream);

/* SampledSoftmaxLoss (recstudio/model/loss_func.py:80-90, one positive per row) evaluated ON THE OWNERS of the negatives, for the
 * same routed step (positives not routed, pos_rows per rank): loss_q = logsumexp(z_pos, z_1 .. z_n) - z_pos with z = score -
 * log q spans all owners of a query's negatives, so the owner pass has two phases around an 8-byte-per-query all-reduce:
 *   rsa_shard_owner_ssm_forward  sort by row (in place, i.e. item_target == item_local: + the solo classification), the queries'
 *     runs, and ONE walk over the received negatives' rows: z per slot (d_slots), and per query over this owner's slots
 *     run_max = max z, run_sum = sum exp(z - run_max), run_acc = sum exp(z - run_max) * row (flash-attention style partials);
 *   [caller: m = max(z_pos, all-reduce-max(run_max)); s = all-reduce-sum(run_sum * exp(run_max - m));
 *            lse = m + log(s + exp(z_pos - m))]
 *   rsa_shard_owner_ssm_finish   (args.pos_score = z_pos) d = exp(z - lse) / mean_den per slot; qgrad_all += gate *
 *     exp(run_max - lse) / mean_den * run_acc -- the query gradient needs NO second pass over the rows --; the positives'
 *     owners add (softmax_pos - 1) / mean_den * row_pos; every touched item row gets gate * item_scale * sum d * q: into a
 *     gradient block through the sorted apply pass; in place, the rows ONE element of the step touches from a second walk by
 *     query (the query row in registers, d from the z phase 1 left per slot: no dot product, no query row per element) and
 *     only the rows several elements touch through the sorted apply pass -- 3 row transfers per negative and B query rows
 *     instead of one per element.
 * The loss itself is lse - z_pos on the caller's side.  No scores travel home, no score gradients travel back (8 instead of
 * 16 bytes per triplet over xGMI).  dims in {64, 128, 256}; same workspace as the BPR form. */
int rsa_shard_owner_ssm_forward(const rsa_shard_owner_bpr_args* args, rsa_stream_t stream);
int rsa_shard_owner_ssm_finish(const rsa_shard_owner_bpr_args* args, const float* lse_all, rsa_stream_t stream);

/* Placement probe (no reference counterpart: the reference allocates through torch and never looks).  On MI355X the same
 * launch runs 10-15 % slower when the buffers it WRITES live in certain physical regions of the HBM (per allocation, stable,
 * visible only under concurrent read load -- DESIGN.md 6, profiles/r05_output_placement.json).  This launch reproduces the
 * access pattern of the fused forward without its arithmetic: per tile of 64 elements 64 random 512-byte rows read from
 * `source` (any resident buffer much larger than the 256 MB Infinity Cache; its contents do not matter) and 512 + 3 x 256 + 16
 * bytes written over `region` (four arrays at quarter offsets, at most 65536 tiles).  A caller times it (events on `stream`)
 * on candidate allocations and keeps the fastest: recstudio_amd/placement.py.  region: 256-byte aligned, >= 1 MiB, OVERWRITTEN. */
int rsa_placement_probe(void* region, int64_t region_bytes, const void* source, int64_t source_bytes, uint32_t salt,
                        rsa_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RECSTUDIO_AMD_H */
