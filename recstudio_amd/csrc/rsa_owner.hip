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

namespace rsa {

struct OwnArgs {
  const float* item;           // [n_rows, D]
  float* item_rw;              // UPD: the same table, written for solo rows
  const float* q_all;          // [n_queries, D]
  float* qgrad_all;            // [n_queries, D], accumulated into
  const int64_t* keys;         // received segments
  const float* d;              // d loss/d score per slot
  const uint64_t* qpairs;      // (query, slot) sorted by query
  const int32_t* run_start;    // [n_queries] first / one-past-last sorted position of each query's run (0, 0: none)
  const int32_t* run_end;
  const uint8_t* solo;         // [slots] or null
  const float* scale;          // {gate * item_scale, gate}
  int64_t n_rows;
  int32_t n_queries;
};

__global__ void owner_scale_kernel(const float* __restrict__ scale_in, const int32_t* __restrict__ step_dropped,
                                   float* __restrict__ scale_out) {
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

// 64 elements of ONE query: lane r holds element r's row (bit 31: solo) and coefficient; every lane group streams its
// rows in batches, qacc += d * row, and a solo row is rewritten as row + upd * (d * q)
template <int LPR, bool NT, bool UPD>
__device__ __forceinline__ void tile_rows_own(const float* table, int32_t id_lane, float d_lane, const Frag<LPR, false>& qf,
                                              float4& qacc, float* item_rw, float upd) {
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
      const float g = __shfl(d_lane, r, 64);
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
template <int LPR, bool NT, bool UPD>
__global__ __launch_bounds__(256, RSA_OWN_MIN_WAVES) void owner_backward_walk_kernel(const OwnArgs a, const int wpq_log2) {
  using F = Frag<LPR, false>;
  constexpr int D = LPR * 4;
  __shared__ float s_q[4][D];
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
    if (re > rs) {
      F qf;
      frag_load<LPR, false>(qf, a.q_all + (size_t)m * D, sub, D);
      const int T = (re - rs + 63) >> 6;
#pragma unroll 1
      for (int t = part; t < T; t += wpq) {
        const int32_t i = rs + (t << 6) + lane;
        const bool act = i < re;
        int32_t id = 0;
        float dv = 0.f;
        if (act) {
          const uint32_t slot = rdx_val(a.qpairs[i]);
          int64_t row = a.keys[slot] & 0xffffffffll;
          row = row >= a.n_rows ? a.n_rows - 1 : row;        // never fault on a bad key
          id = (int32_t)row;
          dv = a.d[slot];
          if (UPD && a.solo[slot]) id |= (int32_t)0x80000000;
        }
        tile_rows_own<LPR, NT, UPD>(a.item, id, dv, qf, qacc, a.item_rw, upd);
      }
#pragma unroll
      for (int mk = LPR; mk < 64; mk <<= 1) {
        qacc.x += __shfl_xor(qacc.x, mk, 64); qacc.y += __shfl_xor(qacc.y, mk, 64);
        qacc.z += __shfl_xor(qacc.z, mk, 64); qacc.w += __shfl_xor(qacc.w, mk, 64);
      }
    }
    if (wpq > 1) {         // block-uniform: the partials of a query's waves meet in LDS, added in wave order
      if (part != 0 && lane < LPR) *reinterpret_cast<float4*>(&s_q[wave][sub * 4]) = qacc;
      __syncthreads();
      if (part == 0 && lane < LPR) {
        for (int k = 1; k < wpq; ++k) {
          const float4 o = *reinterpret_cast<const float4*>(&s_q[wave + k][sub * 4]);
          qacc.x += o.x; qacc.y += o.y; qacc.z += o.z; qacc.w += o.w;
        }
      }
      __syncthreads();     // the slots are rewritten in the next iteration
    }
    if (re > rs && part == 0 && lane < LPR) {
      float4* gp = reinterpret_cast<float4*>(a.qgrad_all + (size_t)m * D + sub * 4);
      float4 o = *gp;
      o.x = __fmaf_rn(gate, qacc.x, o.x); o.y = __fmaf_rn(gate, qacc.y, o.y);
      o.z = __fmaf_rn(gate, qacc.z, o.z); o.w = __fmaf_rn(gate, qacc.w, o.w);
      *gp = o;
    }
  }
}

template <int LPR>
static void launch_walk(const OwnArgs& a, bool upd, int64_t slots, hipStream_t s) {
  const bool nt = (size_t)a.n_rows * LPR * 16 > (512ull << 20);
  const int64_t tiles_per_query = slots / (a.n_queries > 0 ? a.n_queries : 1) / 64;
  const int wpq_log2 = tiles_per_query >= 8 ? 2 : (tiles_per_query >= 3 ? 1 : 0);
  const int qpb = 4 >> wpq_log2;
  int64_t blocks = ((int64_t)a.n_queries + qpb - 1) / qpb;
  if (blocks > 4096) blocks = 4096;
  dim3 grid((unsigned)blocks), block(256);
  if (upd) {
    if (nt) hipLaunchKernelGGL((owner_backward_walk_kernel<LPR, true, true>), grid, block, 0, s, a, wpq_log2);
    else hipLaunchKernelGGL((owner_backward_walk_kernel<LPR, false, true>), grid, block, 0, s, a, wpq_log2);
  } else {
    if (nt) hipLaunchKernelGGL((owner_backward_walk_kernel<LPR, true, false>), grid, block, 0, s, a, wpq_log2);
    else hipLaunchKernelGGL((owner_backward_walk_kernel<LPR, false, false>), grid, block, 0, s, a, wpq_log2);
  }
}

static inline int64_t align256o(int64_t b) { return (b + 255) / 256 * 256; }

struct OwnLayout {
  void* sorted_ws;              // the row sort + sorted apply workspace (sorted_workspace_bytes(slots))
  uint64_t *qa, *qb;            // the query sort's ping-pong buffers
  void* qtemp;
  int32_t *run_start, *run_end;
  uint8_t* solo;
};

static OwnLayout own_layout(void* workspace, int64_t slots, int64_t n_queries) {
  char* ws = reinterpret_cast<char*>(workspace);
  OwnLayout L;
  L.sorted_ws = ws;
  ws += align256o(sorted_workspace_bytes(slots));
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

}  // namespace rsa

using namespace rsa;

extern "C" int64_t rsa_shard_backward_workspace_bytes(int64_t n_segments, int64_t stride, int64_t n_query_rows) {
  if (n_segments <= 0 || stride <= 0 || n_query_rows <= 0) return 0;
  const int64_t slots = n_segments * stride;
  return align256o(sorted_workspace_bytes(slots)) + 2 * align256o(slots * 8) + align256o(radix_temp_bytes(slots)) +
         2 * align256o(n_query_rows * 4) + align256o(slots) + 256;
}

extern "C" int rsa_shard_backward_segments(const rsa_shard_backward_args* a, rsa_stream_t stream) {
  RSA_CHECK_ARG(a != nullptr, "rsa_shard_backward_segments: args is null");
  RSA_CHECK_ARG(a->n_segments >= 0 && a->stride > RSA_SHARD_HDR, "rsa_shard_backward_segments: bad sizes");
  const int64_t slots = a->n_segments * a->stride;
  hipStream_t s = (hipStream_t)stream;
  RSA_CHECK_ARG(a->scale_out != nullptr, "rsa_shard_backward_segments: scale_out is null");
  hipLaunchKernelGGL(owner_scale_kernel, dim3(1), dim3(1), 0, s, a->item_scale, a->step_dropped, a->scale_out);
  RSA_CHECK_LAUNCH("rsa_shard_backward_segments(scale)");
  if (slots == 0) return RSA_OK;
  RSA_CHECK_ARG(slots < (1ll << 31), "rsa_shard_backward_segments: more than 2^31 slots");
  RSA_CHECK_ARG(a->item_local && a->q_all && a->keys && a->d_owner && a->item_target && a->qgrad_all,
                "rsa_shard_backward_segments: null pointer");
  RSA_CHECK_ARG(a->n_rows >= 1 && a->n_rows < (1ll << 31) && a->n_query_rows >= 1 && a->n_query_rows < (1ll << 31),
                "rsa_shard_backward_segments: table sizes out of range");
  if (a->dim != 64 && a->dim != 128 && a->dim != 256) {
    rsa::set_error("rsa_shard_backward_segments: dim=%d: built for dim in {64, 128, 256}", a->dim);
    return RSA_ERR_UNSUPPORTED;
  }
  const int64_t need = rsa_shard_backward_workspace_bytes(a->n_segments, a->stride, a->n_query_rows);
  RSA_CHECK_ARG(a->workspace && a->workspace_bytes >= need, "rsa_shard_backward_segments: workspace too small (%lld < %lld)",
                (long long)a->workspace_bytes, (long long)need);
  const OwnLayout W = own_layout(a->workspace, slots, a->n_query_rows);
  const SortedLayout L = sorted_layout(W.sorted_ws, slots);
  const bool inplace = a->item_target == a->item_local;
  // 1. slots by row (dead slots: key n_rows, behind every real row), solo classification for the in-place update
  const unsigned row_bits = radix_key_bits(a->n_rows + 1);
  const SrcSegments<false> by_row{a->keys, (uint32_t)a->stride, (uint32_t)a->n_rows};
  if (radix_sort_pairs(by_row, L.pairs_a, L.pairs_b, slots, row_bits, L.temp, s) != hipSuccess) {
    rsa::set_error("rsa_shard_backward_segments: row sort failed: %s", hipGetErrorString(hipGetLastError()));
    return RSA_ERR_HIP;
  }
  uint64_t* row_sorted = radix_result(L.pairs_a, L.pairs_b, row_bits);
  if (inplace) {
    const int rc = classify_solo(row_sorted, slots, a->item_pad_row, a->n_rows, W.solo, s, "rsa_shard_backward_segments");
    if (rc != RSA_OK) return rc;
  }
  // 2. slots by query, the queries' runs
  const unsigned q_bits = radix_key_bits(a->n_query_rows + 1);
  const SrcSegments<true> by_query{a->keys, (uint32_t)a->stride, (uint32_t)a->n_query_rows};
  if (radix_sort_pairs(by_query, W.qa, W.qb, slots, q_bits, W.qtemp, s) != hipSuccess) {
    rsa::set_error("rsa_shard_backward_segments: query sort failed: %s", hipGetErrorString(hipGetLastError()));
    return RSA_ERR_HIP;
  }
  const uint64_t* q_sorted = radix_result(W.qa, W.qb, q_bits);
  if (hipMemsetAsync(W.run_start, 0, (size_t)(2 * align256o(a->n_query_rows * 4)), s) != hipSuccess) {
    rsa::set_error("rsa_shard_backward_segments: memset failed");
    return RSA_ERR_HIP;
  }
  int64_t blocks = (slots + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(query_runs_kernel, dim3((unsigned)blocks), dim3(256), 0, s, q_sorted, slots, (int32_t)a->n_query_rows,
                     W.run_start, W.run_end);
  // 3. the walk: query gradients, solo rows in place
  OwnArgs o;
  o.item = a->item_local;
  o.item_rw = inplace ? a->item_target : nullptr;
  o.q_all = a->q_all;
  o.qgrad_all = a->qgrad_all;
  o.keys = a->keys;
  o.d = a->d_owner;
  o.qpairs = q_sorted;
  o.run_start = W.run_start;
  o.run_end = W.run_end;
  o.solo = inplace ? W.solo : nullptr;
  o.scale = a->scale_out;
  o.n_rows = a->n_rows;
  o.n_queries = (int32_t)a->n_query_rows;
  switch (a->dim) {
    case 64: launch_walk<16>(o, inplace, slots, s); break;
    case 128: launch_walk<32>(o, inplace, slots, s); break;
    default: launch_walk<64>(o, inplace, slots, s); break;
  }
  RSA_CHECK_LAUNCH("rsa_shard_backward_segments(walk)");
  // 4. the rows that several elements touch (or, for a gradient block, every row): sorted apply
  return apply_sorted_segments(row_sorted, slots, a->q_all, a->dim, a->keys, a->d_owner, a->scale_out, a->n_rows, a->item_pad_row,
                               a->item_target, L, s);
}
