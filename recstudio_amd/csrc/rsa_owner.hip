// Owner side of the sharded BACKWARD in one call (include/recstudio_amd.h,
// rsa_shard_backward_segments).  The owner holds the step's received key segments (query << 32 | local row), the
// gathered queries and, after the gradient all-to-all, d loss/d score per slot.  What has to happen:
//     qgrad_all[query] += gate * sum d * item[row]              (partial query gradients, reduce-scattered afterwards)
//     item_target[row] += gate * item_scale * sum d * q[query]  (the dense gradient block, or SGD in place)
// Round 3 ran two full sorted scatters for this (one keyed by query that re-read the 4.2 M item rows, one keyed by row
// that re-read 4.2 M query rows and read-modify-wrote every row: 0.42 + 0.8 ms at configs[3]'s per-GPU shape) behind five
// library sort passes and an unpack kernel that materialised two int64 arrays.  Now:
//   1. the slots are radix-sorted by ROW straight from the segments (rsa_radix.hpp) and, in place, classified: a row
//      that ONE element of the step touches is "solo";
//   2. the slots are radix-sorted by QUERY (12-15 key bits: two passes) and every query's run is located;
//   3. ONE walk over the query runs reads every item row once: the run's query-gradient partial accumulates in
//      registers (a run = one query, so its query row is wave-uniform), and a solo row is rewritten on the spot as
//      row + scale * d * q while row and query fragments are in registers -- the single-GPU step's in-forward update;
//   4. the rows several elements touch (29 % of the elements at that shape) go through the sorted apply pass, which
//      skips everything flagged solo.
// All of it deterministic: no float atomics, fixed summation orders.
//
// The same walk is also the FORWARD of the stock BPR training step (rsa_shard_owner_bpr_forward / _finish): BPR's
// d loss/d neg = sigmoid(neg - pos) / (n M) needs nothing but the query's positive score, so once every owner holds the
// positives' scores (rsa_shard_pos_score on the positive's owner + a 4-byte-per-query all-reduce) it evaluates scores,
// loss terms and gradients of the negatives it received while their rows are in registers: no scores travel home, no
// gradients travel back, and the item rows are read ONCE per step instead of twice.
#include "rsa_common.hpp"
#include "rsa_internal.hpp"
#include "rsa_radix.hpp"
#include "rsa_tile.hpp"

#ifndef RSA_OWN_BATCH
#define RSA_OWN_BATCH 4      // rows per load batch (double-buffered, order pinned by data dependences like tile_rows_qg)
#endif
#ifndef RSA_OWN_MIN_WAVES
#define RSA_OWN_MIN_WAVES 1
#endif
#ifndef RSA_SSM_UPDATE_WALK
#define RSA_SSM_UPDATE_WALK 1   // SampledSoftmax on the owners, in place: solo rows from the second walk by query (0: every element through the sorted apply)
#endif

namespace rsa {

struct OwnArgs {
  const float* item;           // [n_rows, D]
  float* item_rw;              // UPD: the same table, written for solo rows
  const float* q_all;          // [n_queries, D]
  float* qgrad_all;            // [n_queries, D], accumulated into
  const int64_t* keys;         // received segments
  const float* d;              // d loss/d score per slot
  const uint64_t* qpairs;      // (query, slot) sorted by query; null: the segments are query-grouped, a run is a run of slots
  const int32_t* run_start;    // [n_queries] first / one-past-last sorted position of each query's run (0, 0: none)
  const int32_t* run_end;
  const uint8_t* solo;         // [slots] or null
  const float* scale;          // {gate * item_scale, gate}
  int64_t n_rows;
  int32_t n_queries;
  // SCORE (the BPR training step evaluated on the owner): d is an OUTPUT
  float* d_out;                // [slots] d loss/d score of every live slot
  const float* pos_score;      // [n_queries] score of every query's positive
  float* dsum;                 // [n_queries] <- sum of the query's d over this owner's slots
  float* loss_out;             // <- this owner's share of the mean loss
  unsigned int* done_counter;
  float* loss_partials;
  float bw, binv;              // 1 / num_neg, 1 / mean_den
  int64_t mean_den;
  // SampledSoftmax on the owners, phase 1 (owner_ssm_walk_kernel): z = dot - logq per slot into d_out, and per query over this
  // owner's slots the running maximum, sum of exp(z - max) and sum of exp(z - max) * row
  const float* logq_rows;      // nullable [n_rows]: the sampler's log-probability of each LOCAL row's item
  float* run_max;              // [n_queries]
  float* run_sum;              // [n_queries]
  float* run_acc;              // [n_queries, D]
};

// keys != nullptr: step_dropped / overflow_sticky are first PUBLISHED from the received segments' header word 1 (what each
// source could not place this step) -- in the score-at-home protocol rsa_shard_score_segments does that
__global__ void owner_scale_kernel(const float* __restrict__ scale_in, int32_t* __restrict__ step_dropped,
                                   float* __restrict__ scale_out, const int64_t* __restrict__ keys, int64_t n_seg,
                                   int64_t stride, int32_t* __restrict__ overflow_sticky) {
  if (keys != nullptr && step_dropped != nullptr) {
    int64_t total = 0;
    for (int64_t sg = 0; sg < n_seg; ++sg) total += keys[sg * stride + 1];
    const int32_t t32 = total > 0x7fffffffll ? 0x7fffffff : (int32_t)total;
    *step_dropped = t32;
    if (overflow_sticky && t32) atomicAdd(overflow_sticky, t32);
  }
  const float gate = (step_dropped != nullptr && step_dropped[0] != 0) ? 0.f : 1.f;
  scale_out[0] = gate * (scale_in ? scale_in[0] : 1.f);
  scale_out[1] = gate;
}

// run_start / run_end of every query over the query-sorted pairs (both arrays zeroed before: a query without elements
// keeps the empty run [0, 0)); keys >= n_queries are dead slots
__global__ __launch_bounds__(256) void query_runs_kernel(const uint64_t* __restrict__ pairs, int64_t total, int32_t n_queries,
                                                         int32_t* __restrict__ run_start, int32_t* __restrict__ run_end) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
    const uint32_t k = rdx_key(pairs[i]);
    if (k >= (uint32_t)n_queries) continue;
    const uint32_t before = i > 0 ? rdx_key(pairs[i - 1]) : 0xffffffffu;
    const uint32_t after = i + 1 < total ? rdx_key(pairs[i + 1]) : 0xffffffffu;
    if (k != before) run_start[k] = (int32_t)i;
    if (k != after) run_end[k] = (int32_t)(i + 1);
  }
}

// The same from QUERY-GROUPED segments (rsa_shard_route_args.group_by_query): every query's elements for this owner are one
// contiguous run of one segment already, so the runs are read off the slots in place -- no sort by query
__global__ __launch_bounds__(256) void query_runs_segments_kernel(const int64_t* __restrict__ keys, int64_t slots, RdxDiv32 by_stride,
                                                                  int32_t n_queries, int32_t* __restrict__ run_start,
                                                                  int32_t* __restrict__ run_end) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += step) {
    const uint32_t seg = by_stride.div((uint32_t)i), within = (uint32_t)i - seg * by_stride.d;
    if (within < RSA_SHARD_HDR) continue;
    const int64_t live = keys[(size_t)seg * by_stride.d];
    const int64_t at = (int64_t)(within - RSA_SHARD_HDR);
    if (at >= live) continue;
    const uint32_t k = (uint32_t)((keys[i] >> 32) & 0x7fffffffll);
    if (k >= (uint32_t)n_queries) continue;
    const uint32_t before = at > 0 ? (uint32_t)((keys[i - 1] >> 32) & 0x7fffffffll) : 0xffffffffu;
    const uint32_t after = at + 1 < live ? (uint32_t)((keys[i + 1] >> 32) & 0x7fffffffll) : 0xffffffffu;
    if (k != before) run_start[k] = (int32_t)i;
    if (k != after) run_end[k] = (int32_t)(i + 1);
  }
}

// 64 elements of ONE query: lane r holds element r's row (bit 31: solo) and coefficient; every lane group streams its
// rows in batches, qacc += d * row, and a solo row is rewritten as row + upd * (d * q)
// SCORE: d_lane carries the BPR weight of the lane's element instead (1 / n, 0 for an idle lane); every row's dot with
// the query is completed while its fragment is in registers and turned into d = w * sigmoid(dot - pos_s) / M; on return
// dot_out is the dot of the lane's own row
template <int LPR, bool NT, bool UPD, bool SCORE>
__device__ __forceinline__ void tile_rows_own(const float* table, int32_t id_lane, float d_lane, const Frag<LPR, false>& qf,
                                              float4& qacc, float* item_rw, float upd, float pos_s, float binv,
                                              float& dot_out) {
  using F = Frag<LPR, false>;
  constexpr int D = LPR * 4;
  constexpr int BATCH = LPR < RSA_OWN_BATCH ? LPR : RSA_OWN_BATCH;
  constexpr int NB = LPR / BATCH;
  const int lane = lane_id();
  const int sub = lane % LPR;
  F x[2][BATCH];
  int gb = lane - sub;
  auto request = [&](int b) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      const int32_t rid = __shfl(id_lane, gb + b * BATCH + k, 64) & 0x7fffffff;
      frag_load<LPR, false, NT>(x[b & 1][k], table + (size_t)rid * D, sub, D);
    }
  };
  request(0);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (b + 1 < NB) request(b + 1);
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      const int r = gb + b * BATCH + k;
      float g = __shfl(d_lane, r, 64);
      if constexpr (SCORE) {
        const float dk = group_sum<LPR>(frag_dot<LPR, false>(x[b & 1][k], qf));
        g = bpr_dneg(pos_s, dk, g, binv);
        dot_out = sub == b * BATCH + k ? dk : dot_out;
      }
      const float4 xv = x[b & 1][k].v[0];
      qacc.x = __fmaf_rn(g, xv.x, qacc.x);
      qacc.y = __fmaf_rn(g, xv.y, qacc.y);
      qacc.z = __fmaf_rn(g, xv.z, qacc.z);
      qacc.w = __fmaf_rn(g, xv.w, qacc.w);
      if constexpr (UPD) {
        const int32_t idf = __shfl(id_lane, r, 64);
        if (idf < 0) {       // lane-group uniform: the row belongs to this element alone
          // the apply pass's arithmetic, rounding for rounding: acc = d * q, row + scale * acc
          const float4 qv = qf.v[0];
          typedef float v4f __attribute__((ext_vector_type(4)));
          v4f nv = {__fadd_rn(xv.x, __fmul_rn(upd, __fmul_rn(g, qv.x))), __fadd_rn(xv.y, __fmul_rn(upd, __fmul_rn(g, qv.y))),
                    __fadd_rn(xv.z, __fmul_rn(upd, __fmul_rn(g, qv.z))), __fadd_rn(xv.w, __fmul_rn(upd, __fmul_rn(g, qv.w)))};
          __builtin_nontemporal_store(nv, reinterpret_cast<v4f*>(item_rw + (size_t)(idf & 0x7fffffff) * D + sub * 4));
        }
      }
    }
    asm volatile("" : "+v"(gb), "+v"(qacc.x), "+v"(qacc.y), "+v"(qacc.z), "+v"(qacc.w));
  }
}

// A query belongs to ONE workgroup: wpq = 1 << wpq_log2 of its waves share the run's tiles round-robin and wave 0 of
// the query adds the partials in wave order through LDS (fixed order, no atomics).
template <int LPR, bool NT, bool UPD, bool SCORE>
__global__ __launch_bounds__(256, RSA_OWN_MIN_WAVES) void owner_backward_walk_kernel(const OwnArgs a, const int wpq_log2) {
  using F = Frag<LPR, false>;
  constexpr int D = LPR * 4;
  __shared__ float s_q[4][D];
  __shared__ float s_sum[4];
  float wave_loss = 0.f;
  const int lane = lane_id();
  const int sub = lane % LPR;
  const int wave = threadIdx.x >> 6;
  const int wpq = 1 << wpq_log2, qpb = 4 >> wpq_log2;
  const int qslot = wave >> wpq_log2, part = wave & (wpq - 1);
  const float upd = a.scale[0], gate = a.scale[1];
  const int64_t groups = ((int64_t)a.n_queries + qpb - 1) / qpb;
  for (int64_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const int64_t m = grp * qpb + qslot;
    const bool valid = m < a.n_queries;                      // wave-uniform
    int32_t rs = 0, re = 0;
    if (valid) {
      rs = a.run_start[m];
      re = a.run_end[m];
    }
    float4 qacc = make_float4(0.f, 0.f, 0.f, 0.f);
    float gsum = 0.f;            // SCORE: the run's sum of d (this wave's tiles)
    if (re > rs) {
      F qf;
      frag_load<LPR, false>(qf, a.q_all + (size_t)m * D, sub, D);
      const float pos_s = SCORE ? a.pos_score[m] : 0.f;
      const int T = (re - rs + 63) >> 6;
#pragma unroll 1
      for (int t = part; t < T; t += wpq) {
        const int32_t i = rs + (t << 6) + lane;
        const bool act = i < re;
        int32_t id = 0;
        float dv = 0.f;
        uint32_t slot = 0;
        if (act) {
          slot = a.qpairs ? rdx_val(a.qpairs[i]) : (uint32_t)i;      // (query-grouped segments: the runs are runs of slots)
          int64_t row = a.keys[slot] & 0xffffffffll;
          row = row >= a.n_rows ? a.n_rows - 1 : row;        // never fault on a bad key
          id = (int32_t)row;
          dv = SCORE ? a.bw : a.d[slot];
          if (UPD && a.solo[slot]) id |= (int32_t)0x80000000;
        }
        float dot = 0.f;
        tile_rows_own<LPR, NT, UPD, SCORE>(a.item, id, dv, qf, qacc, a.item_rw, upd, pos_s, a.binv, dot);
        if constexpr (SCORE) {
          // the lane's own element: the same float operations as inside the tile, so d_out is the value the updates used
          const float g = bpr_dneg(pos_s, dot, dv, a.binv);
          if (act) st_out(&a.d_out[slot], g);
          const float xd = pos_s - dot;
          const float tt = __expf(-fabsf(xd));
          wave_loss -= group_sum<64>(act ? (fminf(xd, 0.f) - __logf(1.f + tt)) * a.bw : 0.f);
          gsum += group_sum<64>(g);
        }
      }
#pragma unroll
      for (int mk = LPR; mk < 64; mk <<= 1) {
        qacc.x += __shfl_xor(qacc.x, mk, 64); qacc.y += __shfl_xor(qacc.y, mk, 64);
        qacc.z += __shfl_xor(qacc.z, mk, 64); qacc.w += __shfl_xor(qacc.w, mk, 64);
      }
    }
    if (wpq > 1) {         // block-uniform: the partials of a query's waves meet in LDS, added in wave order
      if (part != 0 && lane < LPR) *reinterpret_cast<float4*>(&s_q[wave][sub * 4]) = qacc;
      if (SCORE && part != 0 && lane == 0) s_sum[wave] = gsum;
      __syncthreads();
      if (part == 0) {
        for (int k = 1; k < wpq; ++k) {
          if (lane < LPR) {
            const float4 o = *reinterpret_cast<const float4*>(&s_q[wave + k][sub * 4]);
            qacc.x += o.x; qacc.y += o.y; qacc.z += o.z; qacc.w += o.w;
          }
          if constexpr (SCORE) gsum += s_sum[wave + k];
        }
      }
      __syncthreads();     // the slots are rewritten in the next iteration
    }
    if constexpr (SCORE) {
      if (valid && part == 0 && lane == 0) a.dsum[m] = gsum;      // (0 for a query without slots here)
    }
    if (re > rs && part == 0 && lane < LPR) {
      float4* gp = reinterpret_cast<float4*>(a.qgrad_all + (size_t)m * D + sub * 4);
      float4 o = *gp;
      o.x = __fmaf_rn(gate, qacc.x, o.x); o.y = __fmaf_rn(gate, qacc.y, o.y);
      o.z = __fmaf_rn(gate, qacc.z, o.z); o.w = __fmaf_rn(gate, qacc.w, o.w);
      *gp = o;
    }
  }
  if constexpr (SCORE) {
    if (a.loss_out != nullptr) reduce_mean_loss(wave_loss, a.loss_out, a.done_counter, a.loss_partials, a.mean_den);
  }
}

// ---- SampledSoftmaxLoss evaluated on the owners (loss_func.py:80-90 across shards) ------------------------------------------
// loss_q = logsumexp(z_pos, z_1 .. z_n) - z_pos with z = score - log q spans ALL owners of a query's negatives, so the step
// takes two phases.  Phase 1 (this walk): every owner reads the rows of the negatives it received ONCE and leaves, per query,
// the flash-attention style partials over ITS slots -- m = max z, s = sum exp(z - m), acc = sum exp(z - m) * row -- plus z
// per slot.  After an 8-byte-per-query all-reduce (max, then the rescaled sums) every rank knows logsumexp_q, and phase 2
// needs no row for the QUERY gradient any more: qgrad += exp(m - lse) / M * acc.  The item rows get their update
// d_j * q, d_j = exp(z_j - lse) / M, in place: rows ONE element touches from a second walk by query (the query row in
// registers, owner_ssm_update_walk_kernel), the shared ones from the sorted apply pass -- one read-modify-write per touched row.
// 64 elements of one query: lane r holds element r's row and log q (+inf for an idle lane: z = -inf, weight 0).
template <int LPR, bool NT>
__device__ __forceinline__ void tile_rows_ssm(const float* table, int32_t id_lane, float lq_lane, const Frag<LPR, false>& qf,
                                              float4& acc, float& mg, float& sg, float& z_out) {
  using F = Frag<LPR, false>;
  constexpr int D = LPR * 4;
  constexpr int BATCH = LPR < RSA_OWN_BATCH ? LPR : RSA_OWN_BATCH;
  constexpr int NB = LPR / BATCH;
  const int lane = lane_id();
  const int sub = lane % LPR;
  F x[2][BATCH];
  int gb = lane - sub;
  auto request = [&](int b) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      const int32_t rid = __shfl(id_lane, gb + b * BATCH + k, 64);
      frag_load<LPR, false, NT>(x[b & 1][k], table + (size_t)rid * D, sub, D);
    }
  };
  request(0);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (b + 1 < NB) request(b + 1);
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      const int r = gb + b * BATCH + k;
      const float lq = __shfl(lq_lane, r, 64);
      const float dk = group_sum<LPR>(frag_dot<LPR, false>(x[b & 1][k], qf));
      const float z = dk - lq;                       // idle lane: -inf
      z_out = sub == b * BATCH + k ? z : z_out;
      const float mn = fmaxf(mg, z);
      const float ms = mn == -INFINITY ? 0.f : mn;   // nothing seen yet: exp(-inf - 0) = 0 everywhere
      const float so = __expf(mg - ms), e = __expf(z - ms);
      const float4 xv = x[b & 1][k].v[0];
      acc.x = __fmaf_rn(e, xv.x, acc.x * so);
      acc.y = __fmaf_rn(e, xv.y, acc.y * so);
      acc.z = __fmaf_rn(e, xv.z, acc.z * so);
      acc.w = __fmaf_rn(e, xv.w, acc.w * so);
      sg = __fmaf_rn(sg, so, e);
      mg = mn;
    }
    asm volatile("" : "+v"(gb), "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w), "+v"(sg), "+v"(mg));
  }
}

// (m, s, acc) <- the merge of two partials; symmetric in its operands bit for bit (both sides of a shuffle get the same value)
__device__ __forceinline__ void ssm_merge(float& m, float& s, float4& a, float mo, float so, const float4& ao) {
  const float mn = fmaxf(m, mo);
  const float ms = mn == -INFINITY ? 0.f : mn;
  const float ea = __expf(m - ms), eb = __expf(mo - ms);
  a.x = __fmaf_rn(a.x, ea, ao.x * eb); a.y = __fmaf_rn(a.y, ea, ao.y * eb);
  a.z = __fmaf_rn(a.z, ea, ao.z * eb); a.w = __fmaf_rn(a.w, ea, ao.w * eb);
  s = __fmaf_rn(s, ea, so * eb);
  m = mn;
}

template <int LPR, bool NT>
__global__ __launch_bounds__(256, RSA_OWN_MIN_WAVES) void owner_ssm_walk_kernel(const OwnArgs a, const int wpq_log2) {
  using F = Frag<LPR, false>;
  constexpr int D = LPR * 4;
  __shared__ float s_q[4][D];
  __shared__ float s_ms[4][2];
  const int lane = lane_id();
  const int sub = lane % LPR;
  const int wave = threadIdx.x >> 6;
  const int wpq = 1 << wpq_log2, qpb = 4 >> wpq_log2;
  const int qslot = wave >> wpq_log2, part = wave & (wpq - 1);
  const int64_t groups = ((int64_t)a.n_queries + qpb - 1) / qpb;
  for (int64_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const int64_t m = grp * qpb + qslot;
    const bool valid = m < a.n_queries;                      // wave-uniform
    int32_t rs = 0, re = 0;
    if (valid) {
      rs = a.run_start[m];
      re = a.run_end[m];
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float mg = -INFINITY, sg = 0.f;
    if (re > rs) {
      F qf;
      frag_load<LPR, false>(qf, a.q_all + (size_t)m * D, sub, D);
      const int T = (re - rs + 63) >> 6;
#pragma unroll 1
      for (int t = part; t < T; t += wpq) {
        const int32_t i = rs + (t << 6) + lane;
        const bool act = i < re;
        int32_t id = 0;
        float lq = INFINITY;
        uint32_t slot = 0;
        if (act) {
          slot = a.qpairs ? rdx_val(a.qpairs[i]) : (uint32_t)i;
          int64_t row = a.keys[slot] & 0xffffffffll;
          row = row >= a.n_rows ? a.n_rows - 1 : row;        // never fault on a bad key
          id = (int32_t)row;
          lq = a.logq_rows ? a.logq_rows[row] : 0.f;
        }
        float z = -INFINITY;
        tile_rows_ssm<LPR, NT>(a.item, id, lq, qf, acc, mg, sg, z);
        if (act) st_out(&a.d_out[slot], z);
      }
      // the lane groups' partials (each group walked its own rows of every tile): xor-butterfly over the group bits.  The
      // float4 of a lane is its 4 columns of the row: lanes with the same `sub` hold the same columns
#pragma unroll
      for (int mk = LPR; mk < 64; mk <<= 1) {
        const float mo = __shfl_xor(mg, mk, 64), so = __shfl_xor(sg, mk, 64);
        const float4 ao = make_float4(__shfl_xor(acc.x, mk, 64), __shfl_xor(acc.y, mk, 64), __shfl_xor(acc.z, mk, 64),
                                      __shfl_xor(acc.w, mk, 64));
        ssm_merge(mg, sg, acc, mo, so, ao);
      }
    }
    if (wpq > 1) {         // block-uniform: the partials of a query's waves meet in LDS, merged in wave order
      if (part != 0 && lane < LPR) *reinterpret_cast<float4*>(&s_q[wave][sub * 4]) = acc;
      if (part != 0 && lane == 0) {
        s_ms[wave][0] = mg;
        s_ms[wave][1] = sg;
      }
      __syncthreads();
      if (part == 0) {
        for (int k = 1; k < wpq; ++k) {
          const float4 ao = *reinterpret_cast<const float4*>(&s_q[wave + k][(lane % LPR) * 4]);
          ssm_merge(mg, sg, acc, s_ms[wave + k][0], s_ms[wave + k][1], ao);
        }
      }
      __syncthreads();     // the slots are rewritten in the next iteration
    }
    if (valid && part == 0) {
      if (lane == 0) {
        a.run_max[m] = mg;               // (-inf, 0, 0 for a query without slots here)
        a.run_sum[m] = sg;
      }
      if (lane < LPR) *reinterpret_cast<float4*>(a.run_acc + (size_t)m * D + sub * 4) = acc;
    }
  }
}

// phase 2, per slot: z -> d loss/d score = exp(z - lse_query) / M for live slots, 0 for the slack (the apply pass reads d of
// every sorted element, and dead slots sort behind every real row)
__global__ __launch_bounds__(256) void owner_ssm_d_kernel(const int64_t* __restrict__ keys, int64_t slots, RdxDiv32 by_stride,
                                                          int32_t n_queries, const float* __restrict__ lse, float binv,
                                                          float* __restrict__ d_slots) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += step) {
    const uint32_t seg = by_stride.div((uint32_t)i), within = (uint32_t)i - seg * by_stride.d;
    float d = 0.f;
    if (within >= RSA_SHARD_HDR && (int64_t)(within - RSA_SHARD_HDR) < keys[(size_t)seg * by_stride.d]) {
      const uint32_t k = (uint32_t)((keys[i] >> 32) & 0x7fffffffll);
      if (k < (uint32_t)n_queries) d = __expf(d_slots[i] - lse[k]) * binv;
    }
    d_slots[i] = d;
  }
}

// phase 2, per query: qgrad_all[q] += gate * exp(run_max - lse) / M * run_acc[q];  dsum[q] = (1 - exp(z_pos - lse)) / M, the
// value owner_pos_finish_kernel negates into d loss/d pos = (softmax_pos - 1) / M
template <int LPR>
__global__ __launch_bounds__(256) void owner_ssm_query_kernel(const float* __restrict__ run_max, const float* __restrict__ run_acc,
                                                              const float* __restrict__ lse, const float* __restrict__ z_pos,
                                                              float binv, const float* __restrict__ scale,
                                                              float* __restrict__ qgrad_all, float* __restrict__ dsum, int32_t n_queries) {
  constexpr int D = LPR * 4, GPB = 256 / LPR;
  const int sub = threadIdx.x % LPR;
  const float gate = scale[1];
  for (int64_t i = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR; i < n_queries; i += (int64_t)gridDim.x * GPB) {
    const float l = lse[i], mx = run_max[i];
    if (sub == 0) dsum[i] = (1.f - __expf(z_pos[i] - l)) * binv;
    if (mx == -INFINITY) continue;
    const float c = gate * binv * __expf(mx - l);
    const float4 av = *reinterpret_cast<const float4*>(run_acc + (size_t)i * D + sub * 4);
    float4* gp = reinterpret_cast<float4*>(qgrad_all + (size_t)i * D + sub * 4);
    float4 o = *gp;
    o.x = __fmaf_rn(c, av.x, o.x); o.y = __fmaf_rn(c, av.y, o.y); o.z = __fmaf_rn(c, av.z, o.z); o.w = __fmaf_rn(c, av.w, o.w);
    *gp = o;
  }
}

// phase 2, the rows ONE element of the step touches (in-place SGD): a second walk by query.  The sorted apply pass reads a
// query row per ELEMENT (elements in row order hit random query rows: 4.2 M x 512 B at configs[3]'s per-GPU shape, as much
// again as the rows themselves); here a run is one query, its row sits in registers, and a solo row is rewritten as
// row + upd * (d * q) by the lane group that read it -- the in-forward update of the BPR walk, the apply pass's arithmetic
// rounding for rounding.  d = exp(z - lse) / M comes from the z phase 1 left per slot (no second dot product); the rows
// several elements touch get their d written back and go through the sorted apply pass, which skips everything flagged solo.
// The tile's solo elements are compacted first (ds_permute: solo lane -> lane rank) so that every row request is a real one.
template <int LPR, bool NT>
__global__ __launch_bounds__(256, RSA_OWN_MIN_WAVES) void owner_ssm_update_walk_kernel(const OwnArgs a, const float* __restrict__ lse,
                                                                                       const int wpq_log2) {
  using F = Frag<LPR, false>;
  constexpr int D = LPR * 4;
  constexpr int G = 64 / LPR;                                    // lane groups per wave: rows in flight per request slot
  constexpr int BATCH = RSA_OWN_BATCH;
  constexpr int NB = (LPR + BATCH - 1) / BATCH;                  // batches that cover a full tile (64 solo elements)
  const int lane = lane_id();
  const int sub = lane % LPR, grp = lane / LPR;
  const int wave = threadIdx.x >> 6;
  const int wpq = 1 << wpq_log2, qpb = 4 >> wpq_log2;
  const int qslot = wave >> wpq_log2, part = wave & (wpq - 1);
  const float upd = a.scale[0];
  const int64_t groups = ((int64_t)a.n_queries + qpb - 1) / qpb;
  for (int64_t gq = blockIdx.x; gq < groups; gq += gridDim.x) {
    const int64_t m = gq * qpb + qslot;
    if (m >= a.n_queries) continue;                              // wave-uniform (no workgroup barrier in this kernel)
    const int32_t rs = a.run_start[m], re = a.run_end[m];
    if (re <= rs) continue;
    F qf;
    frag_load<LPR, false>(qf, a.q_all + (size_t)m * D, sub, D);
    const float4 qv = qf.v[0];
    const float lse_m = lse[m];
    const int T = (re - rs + 63) >> 6;
#pragma unroll 1
    for (int t = part; t < T; t += wpq) {
      const int32_t i = rs + (t << 6) + lane;
      const bool act = i < re;
      int32_t id = 0;
      float dv = 0.f;
      bool solo = false;
      if (act) {
        const uint32_t slot = a.qpairs ? rdx_val(a.qpairs[i]) : (uint32_t)i;
        int64_t row = a.keys[slot] & 0xffffffffll;
        row = row >= a.n_rows ? a.n_rows - 1 : row;              // never fault on a bad key
        id = (int32_t)row;
        dv = __expf(a.d_out[slot] - lse_m) * a.binv;
        solo = a.solo[slot] != 0;
        if (!solo) st_out(&a.d_out[slot], dv);                   // the apply pass's coefficient
      }
      const uint64_t mask = __ballot(solo);
      const int cnt = __popcll(mask);                            // wave-uniform
      if (cnt == 0) continue;
      // compaction: solo lane -> lane (its rank among the solo lanes), the others behind them
      const uint64_t below = mask & ((1ull << lane) - 1ull);
      const int rank = solo ? __popcll(below) : cnt + (lane - __popcll(below));
      const int32_t idc = __builtin_amdgcn_ds_permute(rank << 2, id);
      const float dc = __int_as_float(__builtin_amdgcn_ds_permute(rank << 2, __float_as_int(dv)));
      // compacted position p = (b * BATCH + k) * G + grp: the lane groups take alternate positions
      F x[2][BATCH];
      auto request = [&](int b) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
          const int p = (b * BATCH + k) * G + grp;
          const int32_t got = __shfl(idc, p, 64);                 // (unconditional: a shuffle under a lane predicate reads 0 from
          const int32_t rid = p < cnt ? got : 0;                  //  the switched-off source lanes); past the last: row 0, always readable
          frag_load<LPR, false, NT>(x[b & 1][k], a.item + (size_t)rid * D, sub, D);
        }
      };
      request(0);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (b * BATCH * G >= cnt) break;                         // wave-uniform
        if (b + 1 < NB && (b + 1) * BATCH * G < cnt) request(b + 1);
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
          const int p = (b * BATCH + k) * G + grp;
          const int32_t rid = __shfl(idc, p, 64);
          const float g = __shfl(dc, p, 64);
          if (p < cnt) {                                          // lane-group uniform
            const float4 xv = x[b & 1][k].v[0];
            typedef float v4f __attribute__((ext_vector_type(4)));
            v4f nv = {__fadd_rn(xv.x, __fmul_rn(upd, __fmul_rn(g, qv.x))), __fadd_rn(xv.y, __fmul_rn(upd, __fmul_rn(g, qv.y))),
                      __fadd_rn(xv.z, __fmul_rn(upd, __fmul_rn(g, qv.z))), __fadd_rn(xv.w, __fmul_rn(upd, __fmul_rn(g, qv.w)))};
            __builtin_nontemporal_store(nv, reinterpret_cast<v4f*>(a.item_rw + (size_t)rid * D + sub * 4));
          }
        }
      }
    }
  }
}

// One lane group per query i: the POSITIVE's part of the step on the rank that owns its row (pos_rows[i] >= 0):
// d loss/d pos = -(sum over all owners of the query's d) -> the coefficient slot behind the segments' (for the sorted
// apply), qgrad_all[i] += gate * dpos * row, and -- when the row is the positive's alone -- the row's update in place.
template <int LPR>
__global__ __launch_bounds__(256) void owner_pos_finish_kernel(const float* item, float* item_rw, const float* __restrict__ q_all,
                                                               float* __restrict__ qgrad_all, const int64_t* __restrict__ pos_rows,
                                                               const float* __restrict__ dsum_all, float* __restrict__ d_pos_out,
                                                               const uint8_t* __restrict__ solo_pos, const float* __restrict__ scale,
                                                               int64_t n_rows, int32_t n_queries) {
  constexpr int D = LPR * 4, GPB = 256 / LPR;
  const int sub = threadIdx.x % LPR;
  const float upd = scale[0], gate = scale[1];
  for (int64_t i = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR; i < n_queries; i += (int64_t)gridDim.x * GPB) {
    int64_t row = pos_rows[i];
    const float dp = row >= 0 ? -dsum_all[i] : 0.f;
    if (sub == 0) d_pos_out[i] = dp;
    if (row < 0) continue;
    row = row >= n_rows ? n_rows - 1 : row;
    const float4 xv = *reinterpret_cast<const float4*>(item + (size_t)row * D + sub * 4);
    const float4 qv = *reinterpret_cast<const float4*>(q_all + (size_t)i * D + sub * 4);
    float4* gp = reinterpret_cast<float4*>(qgrad_all + (size_t)i * D + sub * 4);
    float4 o = *gp;
    const float gd = gate * dp;
    o.x = __fmaf_rn(gd, xv.x, o.x); o.y = __fmaf_rn(gd, xv.y, o.y);
    o.z = __fmaf_rn(gd, xv.z, o.z); o.w = __fmaf_rn(gd, xv.w, o.w);
    *gp = o;
    if (item_rw != nullptr && solo_pos[i]) {
      *reinterpret_cast<float4*>(item_rw + (size_t)row * D + sub * 4) =
          make_float4(__fadd_rn(xv.x, __fmul_rn(upd, __fmul_rn(dp, qv.x))), __fadd_rn(xv.y, __fmul_rn(upd, __fmul_rn(dp, qv.y))),
                      __fadd_rn(xv.z, __fmul_rn(upd, __fmul_rn(dp, qv.z))), __fadd_rn(xv.w, __fmul_rn(upd, __fmul_rn(dp, qv.w))));
    }
  }
}

// out[i] = q_all[i] . item[pos_rows[i]] for the positives this rank owns, 0 for the others (summed over the ranks: the
// score of every positive everywhere)
// pos_ids != nullptr: pos_rows is an OUTPUT -- the local row of every positive this rank owns (contiguous row blocks of
// rows_per_shard rows, or rows_per_shard == 0: rows interleaved, owner = id % n_shards), -1 for the others
template <int LPR>
__global__ __launch_bounds__(256) void shard_pos_score_kernel(const float* __restrict__ item, const float* __restrict__ q_all,
                                                              int64_t* __restrict__ pos_rows, int64_t n_rows,
                                                              int32_t n_queries, float* __restrict__ out,
                                                              const int64_t* __restrict__ pos_ids, int64_t rows_per_shard,
                                                              int32_t n_shards, int32_t rank) {
  constexpr int D = LPR * 4, GPB = 256 / LPR;
  const int sub = threadIdx.x % LPR;
  for (int64_t i = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR; i < n_queries; i += (int64_t)gridDim.x * GPB) {
    int64_t row;
    if (pos_ids != nullptr) {
      int64_t id = pos_ids[i];
      id = id < 0 ? 0 : id;
      int64_t g, loc;
      if (rows_per_shard == 0) {
        g = id % n_shards;
        loc = id / n_shards;
      } else {
        g = id / rows_per_shard;
        g = g >= n_shards ? n_shards - 1 : g;
        loc = id - g * rows_per_shard;
      }
      row = g == rank ? loc : -1;
      if (sub == 0) pos_rows[i] = row;
    } else {
      row = pos_rows[i];
    }
    const bool own = row >= 0;
    row = row < 0 ? 0 : (row >= n_rows ? n_rows - 1 : row);
    const float4 xv = *reinterpret_cast<const float4*>(item + (size_t)row * D + sub * 4);
    const float4 qv = *reinterpret_cast<const float4*>(q_all + (size_t)i * D + sub * 4);
    const float s = group_sum<LPR>(dot4(xv, qv));
    if (sub == 0) out[i] = own ? s : 0.f;
  }
}

template <int LPR, bool SCORE>
static void launch_walk(const OwnArgs& a, bool upd, int64_t slots, hipStream_t s) {
  const bool nt = (size_t)a.n_rows * LPR * 16 > (512ull << 20);
  const int64_t tiles_per_query = slots / (a.n_queries > 0 ? a.n_queries : 1) / 64;
  const int wpq_log2 = tiles_per_query >= 8 ? 2 : (tiles_per_query >= 3 ? 1 : 0);
  const int qpb = 4 >> wpq_log2;
  int64_t blocks = ((int64_t)a.n_queries + qpb - 1) / qpb;
  if (blocks > 4096) blocks = 4096;
  dim3 grid((unsigned)blocks), block(256);
  if (upd) {
    if (nt) hipLaunchKernelGGL((owner_backward_walk_kernel<LPR, true, true, SCORE>), grid, block, 0, s, a, wpq_log2);
    else hipLaunchKernelGGL((owner_backward_walk_kernel<LPR, false, true, SCORE>), grid, block, 0, s, a, wpq_log2);
  } else {
    if (nt) hipLaunchKernelGGL((owner_backward_walk_kernel<LPR, true, false, SCORE>), grid, block, 0, s, a, wpq_log2);
    else hipLaunchKernelGGL((owner_backward_walk_kernel<LPR, false, false, SCORE>), grid, block, 0, s, a, wpq_log2);
  }
}

static inline int64_t align256o(int64_t b) { return (b + 255) / 256 * 256; }

struct OwnLayout {
  void* sorted_ws;              // the row sort + sorted apply workspace (sorted_workspace_bytes(slots + positives))
  uint64_t *qa, *qb;            // the query sort's ping-pong buffers
  void* qtemp;
  int32_t *run_start, *run_end;
  uint8_t* solo;                // [slots + positives]
};

static OwnLayout own_layout(void* workspace, int64_t slots, int64_t n_queries) {
  char* ws = reinterpret_cast<char*>(workspace);
  OwnLayout L;
  L.sorted_ws = ws;
  ws += align256o(sorted_workspace_bytes(slots + n_queries));
  L.qa = reinterpret_cast<uint64_t*>(ws);
  ws += align256o(slots * 8);
  L.qb = reinterpret_cast<uint64_t*>(ws);
  ws += align256o(slots * 8);
  L.qtemp = ws;
  ws += align256o(radix_temp_bytes(slots));
  L.run_start = reinterpret_cast<int32_t*>(ws);
  ws += align256o(n_queries * 4);
  L.run_end = reinterpret_cast<int32_t*>(ws);
  ws += align256o(n_queries * 4);
  L.solo = reinterpret_cast<uint8_t*>(ws);
  return L;
}

static int64_t own_workspace_bytes(int64_t slots, int64_t n_queries) {
  return align256o(sorted_workspace_bytes(slots + n_queries)) + 2 * align256o(slots * 8) + align256o(radix_temp_bytes(slots)) +
         2 * align256o(n_queries * 4) + align256o(slots + n_queries) + 256;
}

// What both forms of the owner pass share: the update scales, the row sort (+ classification), the query sort, the runs
struct OwnCommon {
  const float* item_local;
  int64_t n_rows;
  int32_t dim;
  const float* q_all;
  int64_t n_query_rows;
  const int64_t* keys;
  int64_t n_segments, stride;
  const int64_t* pos_rows;      // nullable: the step's positives as extra elements of the row sort
  float* item_target;
  const float* item_scale;
  int32_t* step_dropped;
  int32_t* overflow_sticky;     // != nullptr: publish the dropped total from the segment headers first
  float* scale_out;
  int64_t item_pad_row;
  void* workspace;
  int64_t workspace_bytes;
  int keys_grouped;             // the router wrote query-grouped segments: no sort by query
  int no_solo = 0;              // never classify: every element goes through the sorted apply pass
};

struct OwnPrepared {
  OwnLayout W;
  SortedLayout L;
  uint64_t* row_sorted;
  const uint64_t* q_sorted;
  int64_t slots, row_total;
  bool inplace;
};

// which part of the preparation a call issues: everything (one-call steps), the layout only (a later entry point of the
// same step), the sorts / classification / runs only (they read the received keys, nothing else: a caller may issue them a
// step ahead, on another stream), or the update scales only (the step's own half of such a split)
enum OwnPrep { OWN_PREP_ALL = 0, OWN_PREP_LAYOUT = 1, OWN_PREP_SORTS = 2, OWN_PREP_SCALE = 3 };

static int owner_prepare(const OwnCommon& c, OwnPrepared& P, OwnPrep mode, hipStream_t s, const char* who) {
  RSA_CHECK_ARG(c.n_segments >= 0 && c.stride > RSA_SHARD_HDR, "%s: bad sizes", who);
  P.slots = c.n_segments * c.stride;
  RSA_CHECK_ARG(c.scale_out != nullptr, "%s: scale_out is null", who);
  RSA_CHECK_ARG(P.slots < (1ll << 31) - c.n_query_rows, "%s: more than 2^31 slots", who);
  RSA_CHECK_ARG(c.item_local && (c.q_all || mode == OWN_PREP_SORTS) && (c.keys || P.slots == 0) && c.item_target, "%s: null pointer", who);
  RSA_CHECK_ARG(c.n_rows >= 1 && c.n_rows < (1ll << 31) && c.n_query_rows >= 1 && c.n_query_rows < (1ll << 31),
                "%s: table sizes out of range", who);
  if (c.dim != 64 && c.dim != 128 && c.dim != 256) {
    rsa::set_error("%s: dim=%d: built for dim in {64, 128, 256}", who, c.dim);
    return RSA_ERR_UNSUPPORTED;
  }
  const int64_t need = own_workspace_bytes(P.slots, c.n_query_rows);
  RSA_CHECK_ARG(c.workspace && c.workspace_bytes >= need, "%s: workspace too small (%lld < %lld)", who,
                (long long)c.workspace_bytes, (long long)need);
  P.W = own_layout(c.workspace, P.slots, c.n_query_rows);
  P.row_total = P.slots + (c.pos_rows ? c.n_query_rows : 0);
  P.L = sorted_layout(P.W.sorted_ws, P.slots + c.n_query_rows);
  P.inplace = c.item_target == c.item_local;
  const unsigned row_bits = radix_key_bits(c.n_rows + 1);
  P.row_sorted = radix_result(P.L.pairs_a, P.L.pairs_b, row_bits);
  const unsigned q_bits = radix_key_bits(c.n_query_rows + 1);
  P.q_sorted = c.keys_grouped ? nullptr : radix_result(P.W.qa, P.W.qb, q_bits);
  if (mode == OWN_PREP_LAYOUT) return RSA_OK;
  if (mode != OWN_PREP_SORTS) {
    hipLaunchKernelGGL(owner_scale_kernel, dim3(1), dim3(1), 0, s, c.item_scale, c.step_dropped, c.scale_out,
                       c.overflow_sticky ? c.keys : nullptr, c.n_segments, c.stride, c.overflow_sticky);
    RSA_CHECK_LAUNCH(who);
  }
  if (mode == OWN_PREP_SCALE) return RSA_OK;
  if (P.row_total == 0) return RSA_OK;
  // 1. elements by row (dead slots: key n_rows, behind every real row), solo classification for the in-place update
  const RdxDiv32 by_stride = rdx_make_div32((uint64_t)c.stride);
  const SrcSegments<false> by_row{c.keys, c.pos_rows, P.slots, by_stride, (uint32_t)c.n_rows};
  if (radix_sort_pairs(by_row, P.L.pairs_a, P.L.pairs_b, P.row_total, row_bits, P.L.temp, s) != hipSuccess) {
    rsa::set_error("%s: row sort failed: %s", who, hipGetErrorString(hipGetLastError()));
    return RSA_ERR_HIP;
  }
  if (P.inplace && !c.no_solo) {
    const int rc = classify_solo(P.row_sorted, P.row_total, c.item_pad_row, c.n_rows, P.W.solo, s, who);
    if (rc != RSA_OK) return rc;
  }
  if (P.slots == 0) return RSA_OK;
  if (c.keys_grouped) {       // 2'. the router already grouped the segments by query: read the runs off the slots
    if (hipMemsetAsync(P.W.run_start, 0, (size_t)(2 * align256o(c.n_query_rows * 4)), s) != hipSuccess) {
      rsa::set_error("%s: memset failed", who);
      return RSA_ERR_HIP;
    }
    int64_t gblocks = (P.slots + 255) / 256;
    if (gblocks > 8192) gblocks = 8192;
    hipLaunchKernelGGL(query_runs_segments_kernel, dim3((unsigned)gblocks), dim3(256), 0, s, c.keys, P.slots, by_stride,
                       (int32_t)c.n_query_rows, P.W.run_start, P.W.run_end);
    RSA_CHECK_LAUNCH(who);
    P.q_sorted = nullptr;
    return RSA_OK;
  }
  // 2. slots by query, the queries' runs
  const SrcSegments<true> by_query{c.keys, nullptr, P.slots, by_stride, (uint32_t)c.n_query_rows};
  if (radix_sort_pairs(by_query, P.W.qa, P.W.qb, P.slots, q_bits, P.W.qtemp, s) != hipSuccess) {
    rsa::set_error("%s: query sort failed: %s", who, hipGetErrorString(hipGetLastError()));
    return RSA_ERR_HIP;
  }
  if (hipMemsetAsync(P.W.run_start, 0, (size_t)(2 * align256o(c.n_query_rows * 4)), s) != hipSuccess) {
    rsa::set_error("%s: memset failed", who);
    return RSA_ERR_HIP;
  }
  int64_t blocks = (P.slots + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(query_runs_kernel, dim3((unsigned)blocks), dim3(256), 0, s, P.q_sorted, P.slots, (int32_t)c.n_query_rows,
                     P.W.run_start, P.W.run_end);
  RSA_CHECK_LAUNCH(who);
  return RSA_OK;
}

static OwnArgs walk_args(const OwnCommon& c, const OwnPrepared& P, float* qgrad_all) {
  OwnArgs o = {};
  o.item = c.item_local;
  o.item_rw = P.inplace ? c.item_target : nullptr;
  o.q_all = c.q_all;
  o.qgrad_all = qgrad_all;
  o.keys = c.keys;
  o.qpairs = P.q_sorted;
  o.run_start = P.W.run_start;
  o.run_end = P.W.run_end;
  o.solo = P.inplace ? P.W.solo : nullptr;
  o.scale = c.scale_out;
  o.n_rows = c.n_rows;
  o.n_queries = (int32_t)c.n_query_rows;
  return o;
}

}  // namespace rsa

using namespace rsa;

extern "C" int64_t rsa_shard_backward_workspace_bytes(int64_t n_segments, int64_t stride, int64_t n_query_rows) {
  if (n_segments < 0 || stride <= 0 || n_query_rows <= 0) return 0;
  return own_workspace_bytes(n_segments * stride, n_query_rows);
}

extern "C" int rsa_shard_backward_segments(const rsa_shard_backward_args* a, rsa_stream_t stream) {
  RSA_CHECK_ARG(a != nullptr, "rsa_shard_backward_segments: args is null");
  hipStream_t s = (hipStream_t)stream;
  const OwnCommon c{a->item_local, a->n_rows,       a->dim,        a->q_all,       a->n_query_rows, a->keys,
                    a->n_segments, a->stride,       nullptr,       a->item_target, a->item_scale,   const_cast<int32_t*>(a->step_dropped),
                    nullptr,       a->scale_out,    a->item_pad_row, a->workspace, a->workspace_bytes, 0};
  OwnPrepared P;
  int rc = owner_prepare(c, P, OWN_PREP_ALL, s, "rsa_shard_backward_segments");
  if (rc != RSA_OK || P.slots == 0) return rc;
  RSA_CHECK_ARG(a->d_owner && a->qgrad_all, "rsa_shard_backward_segments: null pointer");
  // 3. the walk: query gradients, solo rows in place
  OwnArgs o = walk_args(c, P, a->qgrad_all);
  o.d = a->d_owner;
  switch (a->dim) {
    case 64: launch_walk<16, false>(o, P.inplace, P.slots, s); break;
    case 128: launch_walk<32, false>(o, P.inplace, P.slots, s); break;
    default: launch_walk<64, false>(o, P.inplace, P.slots, s); break;
  }
  RSA_CHECK_LAUNCH("rsa_shard_backward_segments(walk)");
  // 4. the rows that several elements touch (or, for a gradient block, every row): sorted apply
  return apply_sorted_segments(P.row_sorted, P.slots, P.slots, a->q_all, a->dim, a->keys, a->d_owner, a->scale_out, a->n_rows,
                               a->item_pad_row, a->item_target, P.L, s);
}

extern "C" int rsa_shard_pos_score(const float* item_local, int64_t n_rows, int32_t dim, const float* q_all,
                                   int64_t n_query_rows, int64_t* pos_rows, float* out, const int64_t* pos_ids,
                                   int64_t rows_per_shard, int32_t n_shards, int32_t rank, rsa_stream_t stream) {
  RSA_CHECK_ARG(n_query_rows >= 0 && n_rows >= 1, "rsa_shard_pos_score: bad sizes");
  RSA_CHECK_ARG(pos_ids == nullptr || (rows_per_shard >= 0 && n_shards >= 1 && rank >= 0 && rank < n_shards),
                "rsa_shard_pos_score: bad shard geometry");
  if (n_query_rows == 0) return RSA_OK;
  RSA_CHECK_ARG(item_local && q_all && pos_rows && out, "rsa_shard_pos_score: null pointer");
  hipStream_t s = (hipStream_t)stream;
  int64_t blocks;
#define RSA_POS_LAUNCH(LPR)                                                                                       \
  blocks = (n_query_rows + 256 / LPR - 1) / (256 / LPR);                                                          \
  if (blocks > 4096) blocks = 4096;                                                                               \
  hipLaunchKernelGGL(shard_pos_score_kernel<LPR>, dim3((unsigned)blocks), dim3(256), 0, s, item_local, q_all, pos_rows, \
                     n_rows, (int32_t)n_query_rows, out, pos_ids, rows_per_shard, n_shards, rank)
  switch (dim) {
    case 64: RSA_POS_LAUNCH(16); break;
    case 128: RSA_POS_LAUNCH(32); break;
    case 256: RSA_POS_LAUNCH(64); break;
    default:
      rsa::set_error("rsa_shard_pos_score: dim=%d: built for dim in {64, 128, 256}", dim);
      return RSA_ERR_UNSUPPORTED;
  }
#undef RSA_POS_LAUNCH
  RSA_CHECK_LAUNCH("rsa_shard_pos_score");
  return RSA_OK;
}

static OwnCommon bpr_common(const rsa_shard_owner_bpr_args* a) {
  return OwnCommon{a->item_local, a->n_rows,  a->dim,      a->q_all,       a->n_query_rows, a->keys,
                   a->n_segments, a->stride,  a->pos_rows, a->item_target, a->item_scale,   a->step_dropped,
                   a->overflow_sticky, a->scale_out, a->item_pad_row, a->workspace, a->workspace_bytes, a->keys_grouped};
}

extern "C" int rsa_shard_owner_bpr_forward(const rsa_shard_owner_bpr_args* a, rsa_stream_t stream) {
  RSA_CHECK_ARG(a != nullptr, "rsa_shard_owner_bpr_forward: args is null");
  RSA_CHECK_ARG(a->forward_parts >= 0 && a->forward_parts <= 2, "rsa_shard_owner_bpr_forward: forward_parts must be 0, 1 or 2");
  RSA_CHECK_ARG(a->pos_rows && a->num_neg >= 1 && a->mean_den >= 1 &&
                    (a->forward_parts == 1 || (a->pos_score && a->d_slots && a->dsum_part && a->qgrad_all)),
                "rsa_shard_owner_bpr_forward: null pointer / bad sizes");
  hipStream_t s = (hipStream_t)stream;
  const OwnCommon c = bpr_common(a);
  OwnPrepared P;
  int rc = owner_prepare(c, P, a->forward_parts == 1 ? OWN_PREP_SORTS : a->forward_parts == 2 ? OWN_PREP_SCALE : OWN_PREP_ALL, s,
                         "rsa_shard_owner_bpr_forward");
  if (rc != RSA_OK || a->forward_parts == 1) return rc;
  if (hipMemsetAsync(a->dsum_part, 0, (size_t)a->n_query_rows * 4, s) != hipSuccess ||
      (a->loss_part && hipMemsetAsync(a->loss_part, 0, 4, s) != hipSuccess)) {
    rsa::set_error("rsa_shard_owner_bpr_forward: memset failed");
    return RSA_ERR_HIP;
  }
  if (P.slots == 0) return RSA_OK;
  OwnArgs o = walk_args(c, P, a->qgrad_all);
  o.d_out = a->d_slots;
  o.pos_score = a->pos_score;
  o.dsum = a->dsum_part;
  o.bw = 1.f / (float)a->num_neg;
  o.binv = 1.f / (float)a->mean_den;
  o.mean_den = a->mean_den;
  if (a->loss_part != nullptr) {
    RSA_CHECK_ARG(a->reduce_scratch != nullptr, "rsa_shard_owner_bpr_forward: loss_part needs reduce_scratch");
    char* sc = reinterpret_cast<char*>(a->reduce_scratch);
    o.loss_out = a->loss_part;
    o.done_counter = reinterpret_cast<unsigned int*>(sc + SCRATCH_COUNTER);
    o.loss_partials = reinterpret_cast<float*>(sc + SCRATCH_FUSED_PARTIALS);
  }
  switch (a->dim) {
    case 64: launch_walk<16, true>(o, P.inplace, P.slots, s); break;
    case 128: launch_walk<32, true>(o, P.inplace, P.slots, s); break;
    default: launch_walk<64, true>(o, P.inplace, P.slots, s); break;
  }
  RSA_CHECK_LAUNCH("rsa_shard_owner_bpr_forward(walk)");
  return RSA_OK;
}

extern "C" int rsa_shard_owner_bpr_finish(const rsa_shard_owner_bpr_args* a, const float* dsum_all, rsa_stream_t stream) {
  RSA_CHECK_ARG(a != nullptr && dsum_all != nullptr, "rsa_shard_owner_bpr_finish: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const OwnCommon c = bpr_common(a);
  OwnPrepared P;
  int rc = owner_prepare(c, P, OWN_PREP_LAYOUT, s, "rsa_shard_owner_bpr_finish");
  if (rc != RSA_OK) return rc;
  RSA_CHECK_ARG(a->finish_parts >= 0 && a->finish_parts <= 2, "rsa_shard_owner_bpr_finish: finish_parts must be 0, 1 or 2");
  if (a->finish_parts == 2)
    return apply_sorted_segments(P.row_sorted, P.row_total, P.slots, a->q_all, a->dim, a->keys, a->d_slots, a->scale_out, a->n_rows,
                                 a->item_pad_row, a->item_target, P.L, s);
  const int64_t Q = a->n_query_rows;
  int64_t blocks;
#define RSA_FIN_LAUNCH(LPR)                                                                                          \
  blocks = (Q + 256 / LPR - 1) / (256 / LPR);                                                                        \
  if (blocks > 4096) blocks = 4096;                                                                                  \
  hipLaunchKernelGGL(owner_pos_finish_kernel<LPR>, dim3((unsigned)blocks), dim3(256), 0, s, a->item_local,           \
                     P.inplace ? a->item_target : nullptr, a->q_all, a->qgrad_all, a->pos_rows, dsum_all,             \
                     a->d_slots + P.slots, P.W.solo + P.slots, a->scale_out, a->n_rows, (int32_t)Q)
  switch (a->dim) {
    case 64: RSA_FIN_LAUNCH(16); break;
    case 128: RSA_FIN_LAUNCH(32); break;
    default: RSA_FIN_LAUNCH(64); break;
  }
#undef RSA_FIN_LAUNCH
  RSA_CHECK_LAUNCH("rsa_shard_owner_bpr_finish(positives)");
  if (a->finish_parts == 1) return RSA_OK;
  return apply_sorted_segments(P.row_sorted, P.row_total, P.slots, a->q_all, a->dim, a->keys, a->d_slots, a->scale_out, a->n_rows,
                               a->item_pad_row, a->item_target, P.L, s);
}

// ---- SampledSoftmaxLoss on the owners (two phases around an 8-byte-per-query all-reduce; see owner_ssm_walk_kernel)
extern "C" int rsa_shard_owner_ssm_forward(const rsa_shard_owner_bpr_args* a, rsa_stream_t stream) {
  RSA_CHECK_ARG(a != nullptr, "rsa_shard_owner_ssm_forward: args is null");
  RSA_CHECK_ARG(a->pos_rows && a->mean_den >= 1 && a->d_slots && a->run_max && a->run_sum && a->run_acc && a->q_all,
                "rsa_shard_owner_ssm_forward: null pointer / bad sizes");
  hipStream_t s = (hipStream_t)stream;
  OwnCommon c = bpr_common(a);            // (in place: the row sort also classifies the solo rows for phase 2)
  c.no_solo = !RSA_SSM_UPDATE_WALK;
  OwnPrepared P;
  int rc = owner_prepare(c, P, OWN_PREP_ALL, s, "rsa_shard_owner_ssm_forward");
  if (rc != RSA_OK) return rc;
  const int64_t Q = a->n_query_rows;
  OwnArgs o = walk_args(c, P, nullptr);
  o.solo = nullptr;
  o.item_rw = nullptr;
  o.d_out = a->d_slots;
  o.logq_rows = a->logq_rows;
  o.run_max = a->run_max;
  o.run_sum = a->run_sum;
  o.run_acc = a->run_acc;
  // (the walk writes the three per-query outputs of EVERY query, with or without slots -- also when there is no slot at all:
  // its runs are then all empty)
  if (P.slots == 0) {
    if (hipMemsetAsync(P.W.run_start, 0, (size_t)(2 * align256o(Q * 4)), s) != hipSuccess) {
      rsa::set_error("rsa_shard_owner_ssm_forward: memset failed");
      return RSA_ERR_HIP;
    }
  }
  const bool nt = (size_t)a->n_rows * a->dim * 4 > (512ull << 20);
  const int64_t tiles_per_query = P.slots / (Q > 0 ? Q : 1) / 64;
  const int wpq_log2 = tiles_per_query >= 8 ? 2 : (tiles_per_query >= 3 ? 1 : 0);
  const int qpb = 4 >> wpq_log2;
  int64_t blocks = (Q + qpb - 1) / qpb;
  if (blocks > 4096) blocks = 4096;
#define RSA_SSM_WALK(LPR)                                                                                                \
  if (nt) hipLaunchKernelGGL((owner_ssm_walk_kernel<LPR, true>), dim3((unsigned)blocks), dim3(256), 0, s, o, wpq_log2);   \
  else hipLaunchKernelGGL((owner_ssm_walk_kernel<LPR, false>), dim3((unsigned)blocks), dim3(256), 0, s, o, wpq_log2)
  switch (a->dim) {
    case 64: RSA_SSM_WALK(16); break;
    case 128: RSA_SSM_WALK(32); break;
    default: RSA_SSM_WALK(64); break;
  }
#undef RSA_SSM_WALK
  RSA_CHECK_LAUNCH("rsa_shard_owner_ssm_forward(walk)");
  return RSA_OK;
}

extern "C" int rsa_shard_owner_ssm_finish(const rsa_shard_owner_bpr_args* a, const float* lse_all, rsa_stream_t stream) {
  RSA_CHECK_ARG(a != nullptr && lse_all != nullptr, "rsa_shard_owner_ssm_finish: null pointer");
  RSA_CHECK_ARG(a->pos_score && a->qgrad_all && a->dsum_part && a->run_max && a->run_acc && a->d_slots,
                "rsa_shard_owner_ssm_finish: null pointer (pos_score = z_pos, qgrad_all, dsum_part, run_max, run_acc, d_slots)");
  hipStream_t s = (hipStream_t)stream;
  const OwnCommon c = bpr_common(a);
  OwnPrepared P;
  int rc = owner_prepare(c, P, OWN_PREP_LAYOUT, s, "rsa_shard_owner_ssm_finish");
  if (rc != RSA_OK) return rc;
  if (!RSA_SSM_UPDATE_WALK) P.inplace = false;      // (A/B builds: nothing was classified, every element through the apply pass)
  const int64_t Q = a->n_query_rows;
  const float binv = 1.f / (float)a->mean_den;
  // In place (SGD applied by the kernels) the forward's row sort has flagged the rows one element touches: the positives'
  // owners update such a row themselves, the second walk below the negatives'.  A gradient BLOCK takes every element through
  // the apply pass (nothing was classified: the flags behind the slots' are cleared for the positives' kernel).
  if (!P.inplace && hipMemsetAsync(P.W.solo + P.slots, 0, (size_t)Q, s) != hipSuccess) {
    rsa::set_error("rsa_shard_owner_ssm_finish: memset failed");
    return RSA_ERR_HIP;
  }
  int64_t blocks;
  // per query: the query-gradient partials from the phase-1 accumulators, then the positives' terms (d loss/d pos behind the
  // slots' coefficients for the apply pass; a positive alone on its row is updated here).  Both read rows as they were
  // BEFORE this step's updates: a solo row belongs to one element, and the shared rows change in the apply pass only.
#define RSA_SSM_FIN(LPR)                                                                                                  \
  blocks = (Q + 256 / LPR - 1) / (256 / LPR);                                                                             \
  if (blocks > 4096) blocks = 4096;                                                                                       \
  hipLaunchKernelGGL(owner_ssm_query_kernel<LPR>, dim3((unsigned)blocks), dim3(256), 0, s, a->run_max, a->run_acc, lse_all, \
                     a->pos_score, binv, a->scale_out, a->qgrad_all, a->dsum_part, (int32_t)Q);                           \
  hipLaunchKernelGGL(owner_pos_finish_kernel<LPR>, dim3((unsigned)blocks), dim3(256), 0, s, a->item_local,                \
                     P.inplace ? a->item_target : (float*)nullptr, a->q_all, a->qgrad_all, a->pos_rows, a->dsum_part,      \
                     a->d_slots + P.slots, P.W.solo + P.slots, a->scale_out, a->n_rows, (int32_t)Q)
  switch (a->dim) {
    case 64: RSA_SSM_FIN(16); break;
    case 128: RSA_SSM_FIN(32); break;
    default: RSA_SSM_FIN(64); break;
  }
#undef RSA_SSM_FIN
  RSA_CHECK_LAUNCH("rsa_shard_owner_ssm_finish");
  if (P.slots > 0 && !P.inplace) {          // z -> d for every slot
    int64_t dblocks = (P.slots + 255) / 256;
    if (dblocks > 8192) dblocks = 8192;
    hipLaunchKernelGGL(owner_ssm_d_kernel, dim3((unsigned)dblocks), dim3(256), 0, s, a->keys, P.slots, rdx_make_div32((uint64_t)a->stride),
                       (int32_t)Q, lse_all, binv, a->d_slots);
    RSA_CHECK_LAUNCH("rsa_shard_owner_ssm_finish(d)");
  } else if (P.slots > 0) {                 // the second walk by query: z -> d, solo rows rewritten with the query row in registers
    OwnArgs o = walk_args(c, P, a->qgrad_all);
    o.d_out = a->d_slots;
    o.binv = binv;
    const bool nt = (size_t)a->n_rows * a->dim * 4 > (512ull << 20);
    const int64_t tiles_per_query = P.slots / (Q > 0 ? Q : 1) / 64;
    const int wpq_log2 = tiles_per_query >= 8 ? 2 : (tiles_per_query >= 3 ? 1 : 0);
    const int qpb = 4 >> wpq_log2;
    int64_t wblocks = (Q + qpb - 1) / qpb;
    if (wblocks > 4096) wblocks = 4096;
#define RSA_SSM_UPD(LPR)                                                                                                          \
  if (nt) hipLaunchKernelGGL((owner_ssm_update_walk_kernel<LPR, true>), dim3((unsigned)wblocks), dim3(256), 0, s, o, lse_all, wpq_log2); \
  else hipLaunchKernelGGL((owner_ssm_update_walk_kernel<LPR, false>), dim3((unsigned)wblocks), dim3(256), 0, s, o, lse_all, wpq_log2)
    switch (a->dim) {
      case 64: RSA_SSM_UPD(16); break;
      case 128: RSA_SSM_UPD(32); break;
      default: RSA_SSM_UPD(64); break;
    }
#undef RSA_SSM_UPD
    RSA_CHECK_LAUNCH("rsa_shard_owner_ssm_finish(update walk)");
  }
  // the rows several elements touch (or, for a gradient block, every row): sorted apply
  return apply_sorted_segments(P.row_sorted, P.row_total, P.slots, a->q_all, a->dim, a->keys, a->d_slots, a->scale_out, a->n_rows,
                               a->item_pad_row, a->item_target, P.L, s);
}
