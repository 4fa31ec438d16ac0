// Declarations shared between the translation units of librecstudio_amd.so (not part of the C ABI).
#pragma once
#include "rsa_common.hpp"

namespace rsa {

struct AdamArgs {            // exp_avg == nullptr: plain accumulate (target[id] += scale * sum)
  float* exp_avg;
  float* exp_avg_sq;
  float one_minus_beta1, one_minus_beta2, eps, step_size;
};

struct SortedLayout {        // a sorted scatter's workspace (sorted_workspace_bytes)
  uint64_t *pairs_a, *pairs_b;     // packed (key << 32 | element) pairs: the radix sort's ping-pong buffers
  void* temp;                      // its digit counters
  float *lead_part, *trail_part;   // per 64-element chunk: partial sums of the runs that cross its borders
  int32_t* meta;
};

int64_t sorted_workspace_bytes(int64_t max_total);
SortedLayout sorted_layout(void* workspace, int64_t max_total);

// flags[e] <- 1 for the elements that are alone on their row (and bit 31 of the pair's payload is set)
int classify_solo(uint64_t* pairs, int64_t total, int64_t pad_row, int64_t drop_key, uint8_t* solo, hipStream_t s, const char* who);

// target[row] += upstream * sum d[e] * query[qrow(e)] over row-sorted pairs of received exchange segments (e = slot;
// elements e >= slots: the positive of query e - slots, coefficient d[e])
int apply_sorted_segments(const uint64_t* pairs, int64_t total, int64_t slots, const float* query, int32_t dim,
                          const int64_t* keys, const float* d, const float* upstream, int64_t n_rows, int64_t pad_row,
                          float* target, const SortedLayout& L, hipStream_t s);

// The in-place SGD step's sort (rsa_sorted.hip): the step's M * (num_neg + 1) item elements and its M user elements in ONE
// radix sort (user keys behind the item keys), then the solo classification of the item part.  `draw` (nullable; kind 0 = no):
// pass 0's histogram launch also DRAWS the negatives into neg_ids (rsa_radix.hpp StepDraw).
struct StepDraw;
int64_t step_all_workspace_bytes(int64_t n_queries, int32_t num_neg);
int sort_step_all(const int64_t* pos_ids, int64_t* neg_ids, const int64_t* user_ids, int64_t n_queries, int32_t num_neg,
                  int64_t n_items, int64_t n_users, uint8_t* solo, void* workspace, int64_t workspace_bytes, const StepDraw* draw,
                  hipStream_t s, const char* who);
int apply_step_all(bool users, const float* query, const int64_t* query_index, int32_t dim, int64_t n_queries, int32_t num_neg,
                   const float* dpos, const float* dneg, const float* upstream, int64_t n_items, int64_t n_users, float* target,
                   void* workspace, hipStream_t s);

// PopularSamplerModel.forward on explicit arguments (rsa_sample.hip; the body behind rsa_sample_popular)
int sample_popular_impl(const float* table, const float* pop_prob, const int32_t* guide, int64_t n_items,
                        int32_t guide_log2, int64_t* neg_ids, float* neg_logp, float* u_out, int64_t numel,
                        uint64_t seed, uint64_t offset, uint32_t grid_threads, uint64_t elem_base,
                        const float* cdf_lut, const float* cdf_lines, int32_t lines_log2, rsa_stream_t stream);

}  // namespace rsa
