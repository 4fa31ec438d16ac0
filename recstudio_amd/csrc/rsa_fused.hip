// Fused sample -> gather -> score forward for an Embedding item tower (gfx950, wave64).
//
// Replaces, in one launch, the ATen sequence of BaseRetriever.forward
// (recstudio/model/basemodel/baseretriever.py:153-171):
//   sampler(...)            sampler.py:102-104 (uniform) / :246-258 (popularity)
//   item_encoder(neg ids)   F.embedding           -> [M, n, d] rows (never materialised here)
//   score_func(q, pos/neg)  scorer.py:9-25        -> [M], [M, n]
//
// Work decomposition: a TILE is 64 consecutive elements of the flat [M*n] id
// tensor and belongs to one wave.  Lane r of the wave owns element r: it draws /
// reads the id and, at the end, holds and stores that element's score.  The row
// reads are spread the other way: a D-float row is read by LPR = D/4 lanes as one
// 16-byte load each (D = 128: 32 lanes x 16 B = 512 B, two rows per wave
// instruction, every 128-B line fully used).  Lane group g (LPR lanes) walks rows
// g*LPR .. g*LPR+LPR-1 of the tile, so after LPR steps each lane holds LPR partial
// dot products that a transpose-reduce (LPR-1 shuffles for LPR rows) turns into
// "lane r holds the full dot of row r".
//
// The kernel is HBM-bound: 4*D bytes of random row traffic per element against
// ~2*D flops; see DESIGN.md for the byte model.
#include "rsa_common.hpp"
#include "rsa_tile.hpp"

#ifndef RSA_QG_BATCH
#define RSA_QG_BATCH 4     // rows per load batch of the training forward (double-buffered, order pinned by data dependences: 120 VGPRs = 4 waves/SIMD instead of 171)
#endif

namespace rsa {

struct FwdParams {
  const float* item_table;
  const float* query;
  const int64_t* query_index;
  const int64_t* pos_ids;
  const float* table;
  const float* pop_prob;
  const float* table_prob;
  const float* lut;
  const float* lines;          // bucket lines [2^lines_log2][32] (nullable)
  const int32_t* guide;
  int64_t* neg_ids;
  float* neg_logp;
  float* pos_logp;
  float* pos_score;
  float* neg_score;
  float* row_loss;   // fused BPR epilogue (nullable): per-query loss, d loss/d pos, d loss/d neg
  float* dpos;
  float* dneg;
  float* loss_out;              // fused loss epilogue: mean of row_loss, reduced in the same launch (reduce_mean_loss)
  unsigned int* done_counter;   //   NaN / inf flag word and the {arrivals, fixed-point sum} words, both in the CALLER's
  float* loss_partials;         //   reduce_scratch (rsa_common.hpp, "caller-owned reduction scratch")
  const uint64_t* offset_dev;   // nullable: Philox offset read at run time (graph replays)
  const int64_t* packed_keys;   // num_neg == 1, GIVEN: element e = (query row << 32) | item row (sharded owner side)
  // In-forward SGD (rsa_fused_args.solo_flags): a row that exactly ONE element of the step touches is updated by the wave
  // that has it in registers; the elements on rows touched more than once are left to the sorted scatter
  const uint8_t* solo_flags;
  const float* upd_scale;
  float* item_rw;
  int32_t* step_dropped;        // segment form (seg_stride != 0): <- sum of the segments' header word 1 (nullable)
  int32_t* overflow_sticky;     //   += the same (nullable)
  float* qgrad;      // fused BPR epilogue (nullable): [M, dim] d loss / d query row, accumulated from the rows in flight
  int64_t n_items, n_query_rows, n_queries, numel;
  int64_t mean_queries;         // queries the fused loss is the mean over: n_queries, or one batch of a queue
  PhiloxCall pc;
  int32_t dim, num_neg, sampler, mask_pad_pos, guide_log2, score_mode, lines_log2;
  int32_t seg_stride;           // != 0: packed_keys is [n_segments][seg_stride] with RSA_SHARD_HDR header words per segment
  // a QUEUE of independent batches in one launch (rsa_fused_args.n_batches > 1): tile t belongs to batch t / tiles_per_batch,
  // whose elements draw from their own torch call -- element e - batch * batch_numel at Philox offset + batch * batch_offset4
  uint32_t tiles_per_batch;     // 0: one batch
  int64_t batch_numel;
  uint64_t batch_offset4;
};

// In-place partial transpose-reduce over the lane-mask bits {step, 2*step, ..., step*L/2}:
// on entry d[k] belongs to tile row (base + k*step); on return d[0] is the sum, over the lanes
// that differ only in those mask bits, of the row selected by this lane's own bits.
template <int L>
__device__ __forceinline__ void fold(float (&d)[L], int sub, int step) {
#pragma unroll
  for (int mm = L / 2; mm >= 1; mm >>= 1) {
    const bool upper = (sub & (mm * step)) != 0;
#pragma unroll
    for (int k = 0; k < mm; ++k) {
      const float send = upper ? d[k] : d[k + mm];
      const float keep = upper ? d[k + mm] : d[k];
      d[k] = keep + __shfl_xor(send, mm * step, 64);
    }
  }
}

// Dot products (and squared norms for the cosine scorer) of the tile's 64 rows.
// id_lane / qrow_lane: lane r's row id and query row; lanes whose element is out of range pass
// row 0 (always readable) and ignore their result.  On return lane r holds the results of row r.
//
// Rows are visited BATCH at a time, batch b = rows {b, b+NB, b+2NB, ...} of the lane group, and
// each batch is folded to one value right away (the depth-first order of the transpose-reduce
// tree), so only BATCH row fragments + NB partials are live instead of LPR of each.
#ifndef RSA_FWD_BATCH
#define RSA_FWD_BATCH 8
#endif
#ifndef RSA_SEG_BATCH
#define RSA_SEG_BATCH 4      // rows per batch when every row has its own query row (QU = false: the sharded step's scoring kernel): 123 VGPRs =
                             // 4 waves/SIMD.  With 8 (the query-uniform form's batch) the second fragment per row made it 129 VGPRs = 3 waves:
                             // in-process A/B (tools/exp_seg.py) 382.5 -> 354.0 us at n = 1024, B = 4096; 399.5 -> 369.8 at n = 64, B = 65536;
                             // 4 waves forced on the 8-row form (24 bytes of scratch): 372; 2-row batches: 460; 5 waves (96 VGPRs, scratch): 445
#endif
template <int LPR, bool GENERIC, bool COS, bool QU, bool NT>
__device__ __forceinline__ void tile_rows(const float* __restrict__ table, int D, int32_t id_lane,
                                          const float* __restrict__ query, int32_t qrow_lane,
                                          const Frag<LPR, GENERIC>& qf_uniform, float& dot, float& inorm2,
                                          float& qnorm2) {
  using F = Frag<LPR, GENERIC>;
  constexpr int B0 = QU ? RSA_FWD_BATCH : RSA_SEG_BATCH;     // per-row queries: every row brings a second fragment
  constexpr int BATCH = GENERIC ? 2 : (LPR < B0 ? LPR : B0);
  constexpr int NB = LPR / BATCH;
  constexpr int NBUF = 1;
  const int lane = lane_id();
  const int sub = lane % LPR;
  int gbase = lane - sub;   // first tile row of this lane group
  float top[NB];
  float top2[COS ? NB : 1];
  float top3[(COS && !QU) ? NB : 1];
  F x[NBUF][BATCH];
  F qx[NBUF][QU ? 1 : BATCH];
  auto request = [&](int b) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      const int r = gbase + b + k * NB;
      const int32_t rid = __shfl(id_lane, r, 64);
      frag_load<LPR, GENERIC, NT>(x[b % NBUF][k], table + (size_t)rid * D, sub, D);
      if constexpr (!QU) {
        const int32_t qr = __shfl(qrow_lane, r, 64);
        frag_load<LPR, GENERIC>(qx[b % NBUF][k], query + (size_t)qr * D, sub, D);
      }
    }
  };
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    request(b);
    float d[BATCH];
    float d2[COS ? BATCH : 1];
    float d3[(COS && !QU) ? BATCH : 1];
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      const F& q = QU ? qf_uniform : qx[b % NBUF][QU ? 0 : k];
      d[k] = frag_dot<LPR, GENERIC>(x[b % NBUF][k], q);
      if constexpr (COS) {
        d2[k] = frag_dot<LPR, GENERIC>(x[b % NBUF][k], x[b % NBUF][k]);
        if constexpr (!QU) d3[k] = frag_dot<LPR, GENERIC>(q, q);
      }
    }
    fold<BATCH>(d, sub, NB);
    top[b] = d[0];
    if constexpr (COS) {
      fold<BATCH>(d2, sub, NB);
      top2[b] = d2[0];
      if constexpr (!QU) {
        fold<BATCH>(d3, sub, NB);
        top3[b] = d3[0];
      }
    }
  }
  fold<NB>(top, sub, 1);
  dot = top[0];
  if constexpr (COS) {
    fold<NB>(top2, sub, 1);
    inorm2 = top2[0];
    if constexpr (!QU) {
      fold<NB>(top3, sub, 1);
      qnorm2 = top3[0];
    }
  }
}

// Training variant of tile_rows (query-uniform, inner product, BPR): while a batch's row fragments are still
// in registers every lane of the group gets the batch's complete dot products (plain butterfly sums -- no
// transpose tree, whose long-lived select masks cost > 100 extra VGPRs here), turns them into d loss/d neg and
// accumulates the query-gradient fragment qacc += dneg_row * row.  The backward pass then never re-reads the
// negative rows.  On return lane r holds the dot of row r, as in tile_rows.
// UPD: bit 31 of id_lane marks an element whose row no other element of the step touches: its row is rewritten right
// here as row + upd * dneg * q (SGD in place: upd = -lr) while row and query fragment are in registers.
template <int LPR, bool NT, bool UPD = false>
__device__ __forceinline__ void tile_rows_qg(const float* table, int32_t id_lane,
                                             const Frag<LPR, false>& qf, float pos_s, float bw, float binv, float& dot,
                                             float4& qacc, float* item_rw = nullptr, float upd = 0.f) {
  using F = Frag<LPR, false>;
  constexpr int D = LPR * 4;
  constexpr int BATCH = LPR < RSA_QG_BATCH ? LPR : RSA_QG_BATCH;
  constexpr int NB = LPR / BATCH;
  const int lane = lane_id();
  const int sub = lane % LPR;
  const int gbase = lane - sub;
  dot = 0.f;
  // batch b+1 requested while batch b is consumed; order pinned by data dependences (see tile_rows_ssm)
  F x[2][BATCH];
  int gb = gbase;
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
      const float dk = group_sum<LPR>(frag_dot<LPR, false>(x[b & 1][k], qf));
      const float g = bpr_dneg(pos_s, dk, bw, binv);
      const float4 xv = x[b & 1][k].v[0];
      qacc.x = __fmaf_rn(g, xv.x, qacc.x);
      qacc.y = __fmaf_rn(g, xv.y, qacc.y);
      qacc.z = __fmaf_rn(g, xv.z, qacc.z);
      qacc.w = __fmaf_rn(g, xv.w, qacc.w);
      dot = sub == b * BATCH + k ? dk : dot;
      if constexpr (UPD) {
        const int32_t idf = __shfl(id_lane, gb + b * BATCH + k, 64);
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

// Plain (no loss) counterpart of the pipelined training tile: batches of RSA_QG_BATCH rows, batch b+1 requested while
// batch b is reduced with butterfly sums, order pinned by data dependences.  Query-uniform inner product only.
template <int LPR, bool NT, int PB = RSA_QG_BATCH>
__device__ __forceinline__ void tile_rows_pipe(const float* __restrict__ table, int32_t id_lane,
                                               const Frag<LPR, false>& qf, float& dot) {
  using F = Frag<LPR, false>;
  constexpr int D = LPR * 4;
  constexpr int BATCH = LPR < PB ? LPR : PB;
  constexpr int NB = LPR / BATCH;
  const int lane = lane_id();
  const int sub = lane % LPR;
  dot = 0.f;
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
      const float dk = group_sum<LPR>(frag_dot<LPR, false>(x[b & 1][k], qf));
      dot = sub == b * BATCH + k ? dk : dot;
    }
    asm volatile("" : "+v"(gb), "+v"(dot));
  }
}

__device__ __forceinline__ float finish_score(int mode, float dot, float inorm2, float qnorm2) {
  // CosineScorer: dot / ||item|| / ||query||, two divisions, no epsilon (scorer.py:21-24)
  // EuclideanScorer: -(-2 dot + ||item||^2 + ||query||^2)                (scorer.py:28-34)
  if (mode == RSA_SCORE_COS) return (dot / sqrtf(inorm2)) / sqrtf(qnorm2);
  if (mode == RSA_SCORE_EUC) return -((-2.f * dot + inorm2) + qnorm2);
  return dot;
}

#ifndef RSA_FWD_GRID_CAP
#define RSA_FWD_GRID_CAP (256 * 8)
#endif
#ifndef RSA_QG_MIN_WAVES
#define RSA_QG_MIN_WAVES 1
#endif
#ifndef RSA_FWD_MIN_WAVES
#define RSA_FWD_MIN_WAVES 1
#endif
#ifndef RSA_SSM_BATCH
#define RSA_SSM_BATCH 2     // double-buffered: 2 * 2 row loads in flight per wave, 122 VGPRs at d = 128 = 4 waves/SIMD (4: 142 = 3 waves;
                            // in-process A/B at n = 256, B = 8192 with the query gradient: 197.4 vs 205.6 us; 8: 180)
#endif

// Small and medium launches (up to RSA_PIPE_MAX_TILES tiles: B = 32768 at n = 64) run the butterfly tile with 8-row batches
// (tile_rows_pipe<.., 8>) instead of the transposed fold: B = 4096: 37.8 -> 35.5 us, B = 16384: 120 -> 111 us per launch
// (popularity sampler; uniform and given ids alike), equal within the run-to-run noise at B = 65536, where the transposed
// fold stays (tools/run_r3f.sh).  With few tiles per wave slot what counts is how early a tile's first rows are
// requested, not the shuffle count.
#ifndef RSA_PIPE_MAX_TILES
#define RSA_PIPE_MAX_TILES 32768
#endif
#ifndef RSA_PIPE_BATCH
#define RSA_PIPE_BATCH 8
#endif
#ifndef RSA_UPD_MIN_WAVES
#define RSA_UPD_MIN_WAVES 4      // 128 VGPRs (48 bytes of scratch at d = 128): 1-1.5 % faster than 145 VGPRs at 3 waves/SIMD
#endif
#ifndef RSA_SEG_MIN_WAVES
#define RSA_SEG_MIN_WAVES 1
#endif
#ifndef RSA_PIPE_MIN_WAVES
#define RSA_PIPE_MIN_WAVES 1
#endif
template <int LPR, bool GENERIC, bool COS, bool QU, bool NT, bool QG = false, bool UPD = false, int PIPE = 0>
__global__ __launch_bounds__(256, UPD ? RSA_UPD_MIN_WAVES : (QG ? RSA_QG_MIN_WAVES : ((!QU && !COS && !GENERIC) ? RSA_SEG_MIN_WAVES : (PIPE > 0 ? RSA_PIPE_MIN_WAVES : RSA_FWD_MIN_WAVES))))
void fused_fwd_kernel(const FwdParams p) {
  using F = Frag<LPR, GENERIC>;
  const int lane = lane_id();
  const int sub = lane % LPR;
  const int D = GENERIC ? p.dim : LPR * 4;
  const int64_t n = p.num_neg;
  const int64_t n_tiles = (p.numel + 63) >> 6;
  PhiloxCall pc = p.pc;
  if (p.offset_dev != nullptr) pc.offset4 = *p.offset_dev >> 2;     // graph replay: the offset lives on the device
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t wstride = (int64_t)gridDim.x * (blockDim.x >> 6);

  // Popularity sampler with the direct-lookup table: the draw and the LUT entry of the wave's NEXT tile are
  // fetched one tile ahead (5 VGPRs), so that a tile's row loads no longer wait behind the LUT round trip.
  // (bucket lines are NOT fetched ahead: carrying a line's 12 values across a tile measured 2-4 % slower)
  const bool ahead = QU && p.sampler == RSA_SAMPLER_POPULAR && p.lut != nullptr && p.lines == nullptr;
  float u_next = 0.f;
  float4 lut_next = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch_ahead = [&](int64_t t) {
    const int64_t e2 = (t << 6) + lane;
    if (e2 < p.numel) {
      u_next = torch_rand_element(pc, (uint64_t)e2);
      lut_next = reinterpret_cast<const float4*>(p.lut)[lut_bucket(p.guide_log2, u_next)];
    }
  };
  if (ahead && wave0 < n_tiles) fetch_ahead(wave0);

  if constexpr (!QU) {
    if (p.seg_stride && blockIdx.x == 0 && threadIdx.x == 0 && (p.step_dropped || p.overflow_sticky)) {
      // header word 1 of every received segment = what its source could not place this step: the job-wide total
      const int64_t n_seg = p.numel / p.seg_stride;
      int64_t total = 0;
      for (int64_t sgm = 0; sgm < n_seg; ++sgm) total += p.packed_keys[sgm * p.seg_stride + 1];
      const int32_t t32 = total > 0x7fffffffll ? 0x7fffffff : (int32_t)total;
      if (p.step_dropped) *p.step_dropped = t32;
      if (p.overflow_sticky && t32) atomicAdd(p.overflow_sticky, t32);
    }
  }
  float wave_loss = 0.f;     // fused BPR epilogue: sum of this wave's tile losses (fixed tile order)
  for (int64_t tile = wave0; tile < n_tiles; tile += wstride) {
    const int64_t e = (tile << 6) + lane;
    const int act = e < p.numel;
    const float u_cur = u_next;
    const float4 lut_cur = lut_next;

    // ---- 0. (query-uniform path) the two scalar loads everything else hangs off -- query row index and
    // positive id -- are issued first so that their latency hides under the sampling chain below
    int64_t m_lane = 0, qrow_u = 0, pid_u = 0;
    bool want_pos = false, first = false, pad = false;
    if constexpr (QU) {
      m_lane = (tile << 6) / n;   // wave-uniform: n % 64 == 0
      first = (tile << 6) % n == 0;
      qrow_u = p.query_index ? p.query_index[m_lane] : m_lane;
      want_pos = p.pos_ids != nullptr && (p.pos_score != nullptr || p.pos_logp != nullptr) &&
                 (first || p.row_loss != nullptr);
      if (want_pos) pid_u = p.pos_ids[m_lane];
    }

    // ---- 1. the id of element e (lane-parallel: one Philox / one CDF search per lane)
    int32_t id = 0;
    int64_t key_lane = -1;       // packed_keys: this lane's key, -1 = empty slot
    // queue of batches: this tile's batch draws from its own torch call (wave-uniform arithmetic)
    PhiloxCall pcb = pc;
    uint64_t eb = (uint64_t)e;
    if constexpr (QU) {
      if (p.tiles_per_batch) {
        const uint32_t k = (uint32_t)tile / p.tiles_per_batch;
        pcb.offset4 += (uint64_t)k * p.batch_offset4;
        eb -= (uint64_t)k * (uint64_t)p.batch_numel;
      }
    }
    if (act) {
      if (p.sampler == RSA_SAMPLER_UNIFORM) {
        id = (int32_t)torch_randint_element(pcb, eb, (uint64_t)(p.n_items - 1), 1);
        st_out(&p.neg_ids[e], (int64_t)id);
      } else if (p.sampler == RSA_SAMPLER_POPULAR) {
        const float u = ahead ? u_cur : torch_rand_element(pcb, eb);
        float pr;
        if (ahead) {
          if (p.table_prob)
            id = cdf_resolve_lut<2>(lut_cur, reinterpret_cast<const float4*>(p.lut), p.table_prob, p.table_prob + 1, 2,
                                    p.n_items, p.guide_log2, u, pr);
          else
            id = cdf_resolve_lut<1>(lut_cur, reinterpret_cast<const float4*>(p.lut), p.table, p.pop_prob, 1,
                                    p.n_items, p.guide_log2, u, pr);
        } else {
          id = lookup_popular(p, u, pr);
        }
        st_out(&p.neg_ids[e], (int64_t)id);
        if (p.neg_logp) st_out(&p.neg_logp[e], logf(pr));
      } else {
        int64_t g = p.packed_keys ? p.packed_keys[e] : p.neg_ids[e];
        if (p.packed_keys) {
          if (p.seg_stride) {     // self-describing segments: live iff behind the header and below the segment's count
            const uint32_t seg = (uint32_t)e / (uint32_t)p.seg_stride;
            const uint32_t within = (uint32_t)e - seg * (uint32_t)p.seg_stride;
            const int64_t live = p.packed_keys[(size_t)seg * p.seg_stride];
            if (within < RSA_SHARD_HDR || (int64_t)(within - RSA_SHARD_HDR) >= live) g = -1;
          }
          key_lane = g;
          g = g < 0 ? 0 : (g & 0xffffffffll);   // a negative key is an empty slot (row 0, query 0)
        }
        g = g < 0 ? 0 : (g >= p.n_items ? p.n_items - 1 : g);   // clamp: never fault on a bad id
        id = (int32_t)g;
      }
    }
    if constexpr (!QU) {
      // segment form: a tile that lies entirely in a segment's slack (or past the end) has nothing to score
      if (p.seg_stride && __ballot(key_lane >= 0) == 0ull) continue;
    }

    if (ahead && tile + wstride < n_tiles) fetch_ahead(tile + wstride);

    // ---- 2. query fragment (and, query-uniform path, the positive row): loads issued back to back here,
    // consumed after the negative rows are in flight
    F qf, px;
    int32_t qrow_lane = 0;
    float qn2_u = 0.f;
    bool empty_slot = false;     // packed_keys < 0: the slot's score is 0
    if constexpr (QU) {
      frag_load<LPR, GENERIC>(qf, p.query + (size_t)qrow_u * D, sub, D);
      pad = pid_u == 0;
      pid_u = pid_u < 0 ? 0 : (pid_u >= p.n_items ? p.n_items - 1 : pid_u);
      frag_load<LPR, GENERIC>(px, p.item_table + (size_t)pid_u * D, sub, D);   // row 0 when there is no positive
      if constexpr (COS) qn2_u = group_sum<LPR>(frag_dot<LPR, GENERIC>(qf, qf));
    } else {
      m_lane = act ? e / n : 0;
      qrow_lane = (int32_t)(p.query_index ? (act ? p.query_index[m_lane] : 0) : m_lane);
      if (p.packed_keys) {
        qrow_lane = key_lane < 0 ? 0 : (int32_t)(key_lane >> 32);
        if (qrow_lane >= p.n_query_rows) qrow_lane = 0;      // never fault on a bad key
        if (key_lane < 0) empty_slot = true;
      }
      frag_load<LPR, GENERIC>(qf, p.query, sub, D);   // unused in this path
      px = qf;
    }

    // ---- 3. negatives: gather + dot
    float dot = 0.f, in2 = 1.f, qn2 = 1.f;
    float pos_early = 0.f;
    float4 qacc = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (QG) {
      // the positive score first (its loads were issued before everything else): the negatives' d loss/d score
      // is needed while their rows are in registers
      pos_early = group_sum<LPR>(frag_dot<LPR, GENERIC>(px, qf));
      if (p.mask_pad_pos && pad) pos_early = -INFINITY;
      const float bw = 1.f / (float)n, binv = 1.f / (float)p.mean_queries;
      if constexpr (UPD) {
        // rows touched by exactly one element of the step (rsa_sort_step_elements' classification, element order
        // m * (n + 1) + 1 + j) are updated in the tile; the sorted scatter skips exactly those
        const float upd = p.upd_scale[0];
        const bool solo = act && p.solo_flags[e + m_lane + 1] != 0;
        tile_rows_qg<LPR, NT, true>(p.item_table, solo ? (id | (int32_t)0x80000000) : id, qf, pos_early, bw, binv, dot, qacc,
                                    p.item_rw, upd);
      } else {
        tile_rows_qg<LPR, NT>(p.item_table, id, qf, pos_early, bw, binv, dot, qacc);
      }
      // d loss/d query row = sum_j dneg_j * item_j + dpos * item_pos, dpos = -sum_j dneg_j (this tile's share).
      // Written right here, unconditionally (the host only selects this variant with the BPR epilogue on):
      // under the epilogue's run-time conditions the compiler sinks the whole accumulation below the row
      // loads and keeps all LPR row fragments alive (> 220 VGPRs).
      const float tg = group_sum<64>(bpr_dneg(pos_early, dot, bw, binv));
#pragma unroll
      for (int mk = LPR; mk < 64; mk <<= 1) {
        qacc.x += __shfl_xor(qacc.x, mk, 64); qacc.y += __shfl_xor(qacc.y, mk, 64);
        qacc.z += __shfl_xor(qacc.z, mk, 64); qacc.w += __shfl_xor(qacc.w, mk, 64);
      }
      const float4 pv = px.v[0];
      qacc.x = __fmaf_rn(-tg, pv.x, qacc.x); qacc.y = __fmaf_rn(-tg, pv.y, qacc.y);
      qacc.z = __fmaf_rn(-tg, pv.z, qacc.z); qacc.w = __fmaf_rn(-tg, pv.w, qacc.w);
      if (lane < LPR) *reinterpret_cast<float4*>(p.qgrad + (size_t)m_lane * D + sub * 4) = qacc;    // n == 64: one tile per query
      if constexpr (UPD) {
        // the positive row: d loss/d pos = -sum_j dneg_j, all of it known here (one tile per query)
        const int64_t pid = pid_u;
        const bool solo_p = p.solo_flags[m_lane * (n + 1)] != 0;     // wave-uniform
        if (solo_p && lane < LPR) {
          const float us = p.upd_scale[0], dp = -tg;
          const float4 qv = qf.v[0];
          *reinterpret_cast<float4*>(p.item_rw + (size_t)pid * D + sub * 4) =
              make_float4(__fadd_rn(pv.x, __fmul_rn(us, __fmul_rn(dp, qv.x))), __fadd_rn(pv.y, __fmul_rn(us, __fmul_rn(dp, qv.y))),
                          __fadd_rn(pv.z, __fmul_rn(us, __fmul_rn(dp, qv.z))), __fadd_rn(pv.w, __fmul_rn(us, __fmul_rn(dp, qv.w))));
        }
      }
    } else if constexpr (PIPE > 0 && QU && !COS && !GENERIC) {
      tile_rows_pipe<LPR, NT, PIPE>(p.item_table, id, qf, dot);
    } else {
      tile_rows<LPR, GENERIC, COS, QU, NT>(p.item_table, D, id, p.query, qrow_lane, qf, dot, in2, qn2);
    }
    if constexpr (COS && QU) qn2 = qn2_u;
    if (p.neg_score && act && !(p.seg_stride && empty_slot))       // (null: a training forward whose loss is fused keeps no scores)
      st_out(&p.neg_score[e], empty_slot ? 0.f : finish_score(COS ? p.score_mode : RSA_SCORE_IP, dot, in2, qn2));

    // ---- 4. positives (+ the fused BPR epilogue: every tile of a query needs the positive score)
    const float neg_s = finish_score(COS ? p.score_mode : RSA_SCORE_IP, dot, in2, qn2);
    if (p.pos_ids != nullptr && (p.pos_score != nullptr || p.pos_logp != nullptr)) {
      if constexpr (QU) {
        const bool fuse = p.row_loss != nullptr;
        if (want_pos) {
          const int64_t pid = pid_u;
          float s = 0.f;
          if (p.pos_score) {
            const F& x = px;
            float pd = group_sum<LPR>(frag_dot<LPR, GENERIC>(x, qf));
            float pi2 = 1.f;
            if constexpr (COS) pi2 = group_sum<LPR>(frag_dot<LPR, GENERIC>(x, x));
            s = finish_score(COS ? p.score_mode : RSA_SCORE_IP, pd, pi2, qn2_u);
            if (p.mask_pad_pos && pad) s = -INFINITY;
            if (first && lane == 0) p.pos_score[m_lane] = s;
          }
          if (first && p.pos_logp && lane == 0) p.pos_logp[m_lane] = logf(p.pop_prob[pid]);
          if (fuse) {
            // BPRLoss (loss_func.py:55-59): -mean_m (1/n) sum_j logsigmoid(pos - neg_j), with its gradient
            const float w = 1.f / (float)n, inv_m = 1.f / (float)p.mean_queries;
            // one hardware exp + one log per element: t = exp(-|x|) serves logsigmoid and sigmoid(-x)
            // (absolute error ~1e-7 on terms of O(1), far inside the 1e-4 contract)
            const float xd = s - neg_s;
            const float t = __expf(-fabsf(xd));
            const float ls = fminf(xd, 0.f) - __logf(1.f + t);
            const float sg = bpr_dneg(s, neg_s, w, inv_m);
            if (p.dneg) st_out(&p.dneg[e], sg);
            const float tl = group_sum<64>(ls * w), tg = group_sum<64>(sg);
            wave_loss -= tl;
            if (lane == 0) {   // one tile per query (n == 64; longer queries run fused_bpr_walk_kernel): plain stores
              p.row_loss[m_lane] = -tl;
              if (p.dpos) p.dpos[m_lane] = -tg;
            }
          }
        }
      } else {
        const int owner = act && (e - m_lane * n == 0);
        if (__ballot(owner) != 0ull) {
          int64_t pid64 = owner ? p.pos_ids[m_lane] : 0;
          const bool pad = pid64 == 0;
          pid64 = pid64 < 0 ? 0 : (pid64 >= p.n_items ? p.n_items - 1 : pid64);
          if (p.pos_score) {
            float pd = 0.f, pi2 = 1.f, pq2 = 1.f;
            tile_rows<LPR, GENERIC, COS, false, false>(p.item_table, D, (int32_t)pid64, p.query, qrow_lane, qf, pd,
                                                       pi2, pq2);
            float s = finish_score(COS ? p.score_mode : RSA_SCORE_IP, pd, pi2, pq2);
            if (p.mask_pad_pos && pad) s = -INFINITY;
            if (owner) p.pos_score[m_lane] = s;
          }
          if (p.pos_logp && owner) p.pos_logp[m_lane] = logf(p.pop_prob[pid64]);
        }
      }
    }
  }

  // ---- 5. (fused BPR epilogue) loss = mean of the per-query losses, in the same launch
  if constexpr (QU) {
    if (p.loss_out != nullptr) reduce_mean_loss(wave_loss, p.loss_out, p.done_counter, p.loss_partials, p.n_queries);
  }
}

// ------------------------------------------------------------------ SampledSoftmax epilogue (fused_loss = 2)
// SampledSoftmaxLoss.forward (recstudio/model/loss_func.py:80-90) for pos_score [M], neg_score [M, n], n % 64 == 0:
//   z_pos = pos - logQ_pos, z_j = neg_j - logQ_j;  row = logsumexp(z_pos, z_1..z_n) - z_pos;  loss = mean_m row
//   d loss/d neg_j = softmax_j / M,  d loss/d pos = (softmax_pos - 1) / M   (NaN for a -inf positive, like the reference)
// The logsumexp spans the n/64 tiles of a query, so ONE wave owns a query and walks its tiles in order, carrying the
// running (max, sum) -- no cross-wave exchange, no atomics, bit-reproducible.  z is staged in the dneg buffer and
// rewritten as the gradient once the query's logsumexp is known (same lane, same address).
// QG: d loss/d query = (1/M) sum_j softmax_j item_j + dpos * item_pos is accumulated flash-style while a batch's rows
// are in registers: every lane group keeps sum_j exp(z_j - g) row_j with its own running reference g (rescaled when
// g moves, once per batch of 8 rows) and the groups are merged with the final logsumexp -- the backward of configs[2]
// then never reads an item row.

template <int LPR, bool NT>
__device__ __forceinline__ void tile_rows_ssm(const float* __restrict__ table, int32_t id_lane, float lq_lane, bool has_lq,
                                              const Frag<LPR, false>& qf, float& dot, float& gm, float4& qacc) {
  using F = Frag<LPR, false>;
  constexpr int D = LPR * 4;
  constexpr int BATCH = LPR < RSA_SSM_BATCH ? LPR : RSA_SSM_BATCH;
  constexpr int NB = LPR / BATCH;
  const int lane = lane_id();
  const int sub = lane % LPR;
  const int gbase = lane - sub;
  dot = 0.f;
  // Batch b+1's rows are requested while batch b is being consumed (two batches of fragments live, 2 * BATCH row
  // loads in flight per wave).  The order is pinned through DATA dependences -- an empty asm ties the source-lane
  // number of batch b+2's shuffles to the accumulator after batch b: sched_barrier alone does not hold, the
  // selection DAG moved every batch's accumulation behind the last batch and kept all LPR row fragments alive
  // (277 VGPRs at d = 128).
  F x[2][BATCH];
  float lqk[2][BATCH];
  int gb = gbase;
  auto request = [&](int b) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      const int32_t rid = __shfl(id_lane, gb + b * BATCH + k, 64);
      frag_load<LPR, false, NT>(x[b & 1][k], table + (size_t)rid * D, sub, D);
      lqk[b & 1][k] = has_lq ? __shfl(lq_lane, gb + b * BATCH + k, 64) : 0.f;
    }
  };
  request(0);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (b + 1 < NB) request(b + 1);
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      const float dk = group_sum<LPR>(frag_dot<LPR, false>(x[b & 1][k], qf));     // all LPR lanes hold the row's dot
      dot = sub == b * BATCH + k ? dk : dot;
      // online softmax accumulation: move the group's reference to max(gm, z), rescale, add this row
      const float z = dk - lqk[b & 1][k];
      const float g_new = fmaxf(gm, z);
      const float r = __expf(gm - g_new);      // gm = -inf on the first row: r = 0, qacc is 0 anyway
      const float w = __expf(z - g_new);
      const float4 xv = x[b & 1][k].v[0];
      qacc.x = __fmaf_rn(w, xv.x, qacc.x * r);
      qacc.y = __fmaf_rn(w, xv.y, qacc.y * r);
      qacc.z = __fmaf_rn(w, xv.z, qacc.z * r);
      qacc.w = __fmaf_rn(w, xv.w, qacc.w * r);
      gm = g_new;
    }
    asm volatile("" : "+v"(gb), "+v"(qacc.x), "+v"(qacc.y), "+v"(qacc.z), "+v"(qacc.w));
  }
}

#ifndef RSA_SSM_PIPE_BATCH
#define RSA_SSM_PIPE_BATCH RSA_QG_BATCH      // forward-only tiles of the SampledSoftmax kernel
#endif
#ifndef RSA_SSM_MIN_WAVES
#define RSA_SSM_MIN_WAVES 1      // (4 with 4-row batches: 128 VGPRs + 40 bytes of scratch, no faster than 3 waves)
#endif
template <int LPR, bool NT, bool QG>
__global__ __launch_bounds__(256, QG ? RSA_SSM_MIN_WAVES : 1) void fused_ssm_kernel(const FwdParams p) {
  using F = Frag<LPR, false>;
  constexpr int D = LPR * 4;
  const int lane = lane_id();
  const int sub = lane % LPR;
  const int64_t n = p.num_neg;
  const int T = (int)(n >> 6);
  PhiloxCall pc = p.pc;
  if (p.offset_dev != nullptr) pc.offset4 = *p.offset_dev >> 2;
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t wstride = (int64_t)gridDim.x * (blockDim.x >> 6);
  const float inv_m = 1.f / (float)p.n_queries;
  const bool popular = p.sampler == RSA_SAMPLER_POPULAR;
  const bool has_lq = popular || (p.sampler == RSA_SAMPLER_GIVEN && p.neg_logp != nullptr);
  float wave_loss = 0.f;

  for (int64_t m = wave0; m < p.n_queries; m += wstride) {
    const int64_t qrow = p.query_index ? p.query_index[m] : m;
    int64_t pid = p.pos_ids[m];
    const bool pad = pid == 0;
    pid = pid < 0 ? 0 : (pid >= p.n_items ? p.n_items - 1 : pid);
    F qf, px;
    frag_load<LPR, false>(qf, p.query + (size_t)qrow * D, sub, D);
    frag_load<LPR, false>(px, p.item_table + (size_t)pid * D, sub, D);
    float lq_pos = 0.f;
    if (popular) lq_pos = logf(p.pop_prob[pid]);
    else if (p.sampler == RSA_SAMPLER_GIVEN && p.pos_logp != nullptr) lq_pos = p.pos_logp[m];
    float pos_s = group_sum<LPR>(frag_dot<LPR, false>(px, qf));
    if (p.mask_pad_pos && pad) pos_s = -INFINITY;
    const float z_pos = pos_s - lq_pos;

    float run_m = -INFINITY, run_s = 0.f;      // wave-uniform running logsumexp state over the negatives
    float gm = -INFINITY;                      // lane-group reference of qacc
    float4 qacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
      const int64_t e = m * n + ((int64_t)t << 6) + lane;
      int32_t id;
      float lq = 0.f;
      if (p.sampler == RSA_SAMPLER_UNIFORM) {
        id = (int32_t)torch_randint_element(pc, (uint64_t)e, (uint64_t)(p.n_items - 1), 1);
        st_out(&p.neg_ids[e], (int64_t)id);
      } else if (popular) {
        float pr;
        id = lookup_popular(p, torch_rand_element(pc, (uint64_t)e), pr);
        lq = logf(pr);
        st_out(&p.neg_ids[e], (int64_t)id);
        if (p.neg_logp) st_out(&p.neg_logp[e], lq);
      } else {
        int64_t g = p.neg_ids[e];
        g = g < 0 ? 0 : (g >= p.n_items ? p.n_items - 1 : g);
        id = (int32_t)g;
        if (p.neg_logp) lq = p.neg_logp[e];
      }
      float dot;
      if constexpr (QG) {
        tile_rows_ssm<LPR, NT>(p.item_table, id, lq, has_lq, qf, dot, gm, qacc);
      } else {
        tile_rows_pipe<LPR, NT, RSA_SSM_PIPE_BATCH>(p.item_table, id, qf, dot);      // (the transposed-fold tile needs 164 VGPRs in this frame)
      }
      if (p.neg_score) st_out(&p.neg_score[e], dot);
      const float z = dot - lq;
      if (p.dneg) p.dneg[e] = z;               // staged; rewritten below once the logsumexp is known
      const float m_new = fmaxf(run_m, wave_max(z));
      run_s = run_s * __expf(run_m - m_new) + group_sum<64>(__expf(z - m_new));
      run_m = m_new;
    }

    // ---- the query's logsumexp, loss and gradients
    const float top = fmaxf(run_m, z_pos);
    const float lse = top + logf(run_s * expf(run_m - top) + expf(z_pos - top));
    const bool bad = isinf(z_pos);             // padded positive: the reference divides 0 by 0 (loss_func.py:88-89)
    const float row = bad ? NAN : lse - z_pos;
    const float dp = bad ? NAN : (expf(z_pos - lse) - 1.f) * inv_m;
    wave_loss += row;
    if (lane == 0) {
      p.pos_score[m] = pos_s;
      if (popular && p.pos_logp) p.pos_logp[m] = lq_pos;
      p.row_loss[m] = row;
      if (p.dpos) p.dpos[m] = dp;
    }
    if (p.dneg) {
#pragma unroll 2
      for (int t = 0; t < T; ++t) {
        const int64_t e = m * n + ((int64_t)t << 6) + lane;
        const float z = p.dneg[e];
        st_out(&p.dneg[e], bad ? NAN : __expf(z - lse) * inv_m);
      }
    }
    if constexpr (QG) {
      const float sc = bad ? NAN : __expf(gm - lse) * inv_m;     // this group's weights relative to the final lse
      qacc.x *= sc; qacc.y *= sc; qacc.z *= sc; qacc.w *= sc;
#pragma unroll
      for (int mk = LPR; mk < 64; mk <<= 1) {
        qacc.x += __shfl_xor(qacc.x, mk, 64); qacc.y += __shfl_xor(qacc.y, mk, 64);
        qacc.z += __shfl_xor(qacc.z, mk, 64); qacc.w += __shfl_xor(qacc.w, mk, 64);
      }
      const float4 pv = px.v[0];
      qacc.x = __fmaf_rn(dp, pv.x, qacc.x); qacc.y = __fmaf_rn(dp, pv.y, qacc.y);
      qacc.z = __fmaf_rn(dp, pv.z, qacc.z); qacc.w = __fmaf_rn(dp, pv.w, qacc.w);
      if (lane < LPR) *reinterpret_cast<float4*>(p.qgrad + (size_t)m * D + sub * 4) = qacc;
    }
  }
  if (p.loss_out != nullptr) reduce_mean_loss(wave_loss, p.loss_out, p.done_counter, p.loss_partials, p.n_queries);
}

// ------------------------------------------------------------------ BPR epilogue for queries longer than one tile
// fused_loss = 1 with num_neg = 64 * T, T > 1.  The first version ran the tile-parallel kernel above and let the T tiles
// of a query meet in float atomics on row_loss / dpos / query_grad (plus three memsets): not reproducible.  Here a
// query belongs to ONE workgroup: WPQ = min(4, largest power of two <= T) of its waves share the query's tiles
// round-robin, each carrying its sums in registers, and wave 0 of the query adds the partials in wave order through
// LDS -- fixed order, no atomics, nothing to zero.  BPR needs no second pass over the tiles (the positive score is known
// before the first negative): d loss/d neg is final when it is written.
#ifndef RSA_WALK_PIPE_BATCH
#define RSA_WALK_PIPE_BATCH RSA_QG_BATCH
#endif
#ifndef RSA_WALK_FWD_MIN_WAVES
#define RSA_WALK_FWD_MIN_WAVES 1
#endif
#ifndef RSA_WALK_MIN_WAVES
#define RSA_WALK_MIN_WAVES 4      // the query-gradient form at d = 128 with streaming loads: 130 VGPRs = 3 waves/SIMD without it
#endif
template <int LPR, bool NT, bool QG>
__global__ __launch_bounds__(256, QG ? RSA_WALK_MIN_WAVES : RSA_WALK_FWD_MIN_WAVES) void fused_bpr_walk_kernel(const FwdParams p, const int wpq_log2) {
  using F = Frag<LPR, false>;
  constexpr int D = LPR * 4;
  __shared__ float s_sum[4][2];
  __shared__ float s_q[QG ? 4 : 1][QG ? D : 1];
  const int lane = lane_id();
  const int sub = lane % LPR;
  const int wave = threadIdx.x >> 6;
  const int64_t n = p.num_neg;
  const int T = (int)(n >> 6);
  const int wpq = 1 << wpq_log2, qpb = 4 >> wpq_log2;
  const int qslot = wave >> wpq_log2, part = wave & (wpq - 1);
  PhiloxCall pc = p.pc;
  if (p.offset_dev != nullptr) pc.offset4 = *p.offset_dev >> 2;
  const float w = 1.f / (float)n, inv_m = 1.f / (float)p.n_queries;
  const bool popular = p.sampler == RSA_SAMPLER_POPULAR;
  const int64_t groups = (p.n_queries + qpb - 1) / qpb;
  float wave_loss = 0.f;
  for (int64_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const int64_t m = grp * qpb + qslot;
    const bool valid = m < p.n_queries;                      // wave-uniform
    float lsum = 0.f, gsum = 0.f, pos_s = 0.f;
    float4 qacc = make_float4(0.f, 0.f, 0.f, 0.f);
    F qf, px;
    int64_t pid = 0;
    if (valid) {
      const int64_t qrow = p.query_index ? p.query_index[m] : m;
      pid = p.pos_ids[m];
      const bool pad = pid == 0;
      pid = pid < 0 ? 0 : (pid >= p.n_items ? p.n_items - 1 : pid);
      frag_load<LPR, false>(qf, p.query + (size_t)qrow * D, sub, D);
      frag_load<LPR, false>(px, p.item_table + (size_t)pid * D, sub, D);
      pos_s = group_sum<LPR>(frag_dot<LPR, false>(px, qf));
      if (p.mask_pad_pos && pad) pos_s = -INFINITY;
#pragma unroll 1
      for (int t = part; t < T; t += wpq) {
        const int64_t e = m * n + ((int64_t)t << 6) + lane;
        int32_t id;
        if (p.sampler == RSA_SAMPLER_UNIFORM) {
          id = (int32_t)torch_randint_element(pc, (uint64_t)e, (uint64_t)(p.n_items - 1), 1);
          st_out(&p.neg_ids[e], (int64_t)id);
        } else if (popular) {
          float pr;
          id = lookup_popular(p, torch_rand_element(pc, (uint64_t)e), pr);
          st_out(&p.neg_ids[e], (int64_t)id);
          if (p.neg_logp) st_out(&p.neg_logp[e], logf(pr));
        } else {
          int64_t g = p.neg_ids[e];
          g = g < 0 ? 0 : (g >= p.n_items ? p.n_items - 1 : g);
          id = (int32_t)g;
        }
        float dot;
        if constexpr (QG) {
          tile_rows_qg<LPR, NT>(p.item_table, id, qf, pos_s, w, inv_m, dot, qacc);
        } else {
          tile_rows_pipe<LPR, NT, RSA_WALK_PIPE_BATCH>(p.item_table, id, qf, dot);
        }
        if (p.neg_score) st_out(&p.neg_score[e], dot);
        const float xd = pos_s - dot;
        const float tt = __expf(-fabsf(xd));
        lsum += (fminf(xd, 0.f) - __logf(1.f + tt)) * w;
        const float sg = bpr_dneg(pos_s, dot, w, inv_m);
        if (p.dneg) st_out(&p.dneg[e], sg);
        gsum += sg;
      }
      lsum = group_sum<64>(lsum);
      gsum = group_sum<64>(gsum);
      if constexpr (QG) {
#pragma unroll
        for (int mk = LPR; mk < 64; mk <<= 1) {
          qacc.x += __shfl_xor(qacc.x, mk, 64); qacc.y += __shfl_xor(qacc.y, mk, 64);
          qacc.z += __shfl_xor(qacc.z, mk, 64); qacc.w += __shfl_xor(qacc.w, mk, 64);
        }
      }
    }
    if (wpq > 1) {         // block-uniform: the partials of a query's waves meet in LDS, added in wave order
      if (part != 0) {
        if (lane == 0) {
          s_sum[wave][0] = lsum;
          s_sum[wave][1] = gsum;
        }
        if constexpr (QG) {
          if (lane < LPR) *reinterpret_cast<float4*>(&s_q[wave][sub * 4]) = qacc;
        }
      }
      __syncthreads();
      if (part == 0) {
        for (int k = 1; k < wpq; ++k) {
          lsum += s_sum[wave + k][0];
          gsum += s_sum[wave + k][1];
          if constexpr (QG) {
            if (lane < LPR) {
              const float4 o = *reinterpret_cast<const float4*>(&s_q[wave + k][sub * 4]);
              qacc.x += o.x; qacc.y += o.y; qacc.z += o.z; qacc.w += o.w;
            }
          }
        }
      }
      __syncthreads();     // the slots are rewritten in the next iteration
    }
    if (valid && part == 0) {
      wave_loss -= lsum;
      if (lane == 0) {
        p.pos_score[m] = pos_s;
        if (popular && p.pos_logp) p.pos_logp[m] = logf(p.pop_prob[pid]);
        p.row_loss[m] = -lsum;
        if (p.dpos) p.dpos[m] = -gsum;
      }
      if constexpr (QG) {
        const float4 pv = px.v[0];        // d loss/d query = sum_j dneg_j item_j + dpos item_pos, dpos = -sum_j dneg_j
        qacc.x = __fmaf_rn(-gsum, pv.x, qacc.x); qacc.y = __fmaf_rn(-gsum, pv.y, qacc.y);
        qacc.z = __fmaf_rn(-gsum, pv.z, qacc.z); qacc.w = __fmaf_rn(-gsum, pv.w, qacc.w);
        if (lane < LPR) *reinterpret_cast<float4*>(p.qgrad + (size_t)m * D + sub * 4) = qacc;
      }
    }
  }
  if (p.loss_out != nullptr) reduce_mean_loss(wave_loss, p.loss_out, p.done_counter, p.loss_partials, p.n_queries);
}

template <int LPR>
static int launch_bpr_walk(const FwdParams& p, hipStream_t stream) {
  const bool nt = (size_t)p.n_items * p.dim * sizeof(float) > (512ull << 20);
  const int T = p.num_neg >> 6;
  const int wpq_log2 = T >= 4 ? 2 : (T >= 2 ? 1 : 0);
  const int qpb = 4 >> wpq_log2;
  int64_t blocks = (p.n_queries + qpb - 1) / qpb;
  if (blocks > RSA_FWD_GRID_CAP) blocks = RSA_FWD_GRID_CAP;
  dim3 grid((unsigned)blocks), block(256);
  if (p.qgrad != nullptr) {
    if (nt) hipLaunchKernelGGL((fused_bpr_walk_kernel<LPR, true, true>), grid, block, 0, stream, p, wpq_log2);
    else hipLaunchKernelGGL((fused_bpr_walk_kernel<LPR, false, true>), grid, block, 0, stream, p, wpq_log2);
  } else {
    if (nt) hipLaunchKernelGGL((fused_bpr_walk_kernel<LPR, true, false>), grid, block, 0, stream, p, wpq_log2);
    else hipLaunchKernelGGL((fused_bpr_walk_kernel<LPR, false, false>), grid, block, 0, stream, p, wpq_log2);
  }
  RSA_CHECK_LAUNCH("rsa_fused_sample_gather_score(bpr walk)");
  return RSA_OK;
}

template <int LPR>
static int launch_ssm(const FwdParams& p, hipStream_t stream) {
  const bool nt = (size_t)p.n_items * p.dim * sizeof(float) > (512ull << 20);
  int64_t blocks = (p.n_queries + 3) / 4;      // one wave per query
  if (blocks > RSA_FWD_GRID_CAP) blocks = RSA_FWD_GRID_CAP;
  dim3 grid((unsigned)blocks), block(256);
  if (p.qgrad != nullptr) {
    if (nt) hipLaunchKernelGGL((fused_ssm_kernel<LPR, true, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((fused_ssm_kernel<LPR, false, true>), grid, block, 0, stream, p);
  } else {
    if (nt) hipLaunchKernelGGL((fused_ssm_kernel<LPR, true, false>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((fused_ssm_kernel<LPR, false, false>), grid, block, 0, stream, p);
  }
  RSA_CHECK_LAUNCH("rsa_fused_sample_gather_score(ssm)");
  return RSA_OK;
}

template <int LPR, bool GENERIC, bool COS, bool QU>
static void launch_fwd2(const FwdParams& p, dim3 grid, dim3 block, hipStream_t stream) {
  // streaming (nontemporal) row loads once the table cannot live in the 256 MB Infinity Cache
  const bool nt = !GENERIC && (size_t)p.n_items * p.dim * sizeof(float) > (512ull << 20);
  if constexpr (QU && !COS && !GENERIC) {
    if (p.qgrad != nullptr && p.solo_flags != nullptr) {     // ... + SGD in place for the rows one element owns
      if (nt) hipLaunchKernelGGL((fused_fwd_kernel<LPR, GENERIC, COS, QU, true, true, true>), grid, block, 0, stream, p);
      else hipLaunchKernelGGL((fused_fwd_kernel<LPR, GENERIC, COS, QU, false, true, true>), grid, block, 0, stream, p);
      return;
    }
    if (p.qgrad != nullptr) {     // training forward: query gradient accumulated from the rows in flight
      if (nt) hipLaunchKernelGGL((fused_fwd_kernel<LPR, GENERIC, COS, QU, true, true>), grid, block, 0, stream, p);
      else hipLaunchKernelGGL((fused_fwd_kernel<LPR, GENERIC, COS, QU, false, true>), grid, block, 0, stream, p);
      return;
    }
    if (((p.numel + 63) >> 6) <= RSA_PIPE_MAX_TILES) {      // small / medium launch: the butterfly tile
      if (nt) hipLaunchKernelGGL((fused_fwd_kernel<LPR, GENERIC, COS, QU, true, false, false, RSA_PIPE_BATCH>), grid, block, 0, stream, p);
      else hipLaunchKernelGGL((fused_fwd_kernel<LPR, GENERIC, COS, QU, false, false, false, RSA_PIPE_BATCH>), grid, block, 0, stream, p);
      return;
    }
  }
  if (nt) hipLaunchKernelGGL((fused_fwd_kernel<LPR, GENERIC, COS, QU, !GENERIC>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((fused_fwd_kernel<LPR, GENERIC, COS, QU, false>), grid, block, 0, stream, p);
}

template <int LPR, bool GENERIC>
static int launch_fwd(const FwdParams& p, bool cos, bool qu, hipStream_t stream) {
  const int64_t n_tiles = (p.numel + 63) >> 6;
  int64_t blocks = (n_tiles + 3) / 4;
  if (blocks > RSA_FWD_GRID_CAP) blocks = RSA_FWD_GRID_CAP;
  if (blocks < 1) blocks = 1;
  dim3 grid((unsigned)blocks), block(256);
  if (cos) {
    if (qu) launch_fwd2<LPR, GENERIC, true, true>(p, grid, block, stream);
    else launch_fwd2<LPR, GENERIC, true, false>(p, grid, block, stream);
  } else {
    if (qu) launch_fwd2<LPR, GENERIC, false, true>(p, grid, block, stream);
    else launch_fwd2<LPR, GENERIC, false, false>(p, grid, block, stream);
  }
  RSA_CHECK_LAUNCH("rsa_fused_sample_gather_score");
  return RSA_OK;
}

__global__ void rng_advance_kernel(uint64_t* offset_dev, uint64_t increment) { *offset_dev += increment; }

}  // namespace rsa

using namespace rsa;

extern "C" int rsa_rng_advance(uint64_t* offset_dev, uint64_t increment, rsa_stream_t stream) {
  RSA_CHECK_ARG(offset_dev != nullptr && (increment & 3) == 0, "rsa_rng_advance: null pointer / increment not a multiple of 4");
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, offset_dev, increment);
  RSA_CHECK_LAUNCH("rsa_rng_advance");
  return RSA_OK;
}

extern "C" int64_t rsa_scratch_bytes(void) { return SCRATCH_BYTES; }

extern "C" int rsa_fused_sample_gather_score(const rsa_fused_args* a, rsa_stream_t stream) {
  RSA_CHECK_ARG(a != nullptr, "rsa_fused_sample_gather_score: args is null");
  RSA_CHECK_ARG(a->n_queries >= 0 && a->num_neg >= 0, "rsa_fused_sample_gather_score: negative sizes");
  const int64_t numel = a->n_queries * (int64_t)a->num_neg;
  if (a->n_queries == 0) return RSA_OK;       // nothing to do (empty batch): before the table-shape checks
  RSA_CHECK_ARG(a->dim >= 4 && a->dim <= 1024 && a->dim % 4 == 0,
                "rsa_fused_sample_gather_score: dim=%d must be a multiple of 4 in [4, 1024]", a->dim);
  RSA_CHECK_ARG(a->n_items >= 2 && a->n_items < (1ll << 31), "rsa_fused_sample_gather_score: n_items out of range");
  RSA_CHECK_ARG(a->n_query_rows >= 1 && a->n_query_rows < (1ll << 31),
                "rsa_fused_sample_gather_score: n_query_rows out of range");
  RSA_CHECK_ARG(a->score_mode >= RSA_SCORE_IP && a->score_mode <= RSA_SCORE_EUC,
                "rsa_fused_sample_gather_score: unknown score_mode %d", a->score_mode);
  RSA_CHECK_ARG(a->sampler >= RSA_SAMPLER_GIVEN && a->sampler <= RSA_SAMPLER_POPULAR,
                "rsa_fused_sample_gather_score: unknown sampler %d", a->sampler);
  if (a->n_queries == 0) return RSA_OK;
  RSA_CHECK_ARG(a->item_table && a->query, "rsa_fused_sample_gather_score: item_table/query is null");
  RSA_CHECK_ARG(a->query_index != nullptr || a->packed_keys != nullptr || a->n_query_rows >= a->n_queries,
                "rsa_fused_sample_gather_score: query has fewer rows than n_queries");
  if (numel > 0) {
    RSA_CHECK_ARG((a->neg_ids || a->packed_keys) && (a->neg_score || a->fused_loss != 0),
                  "rsa_fused_sample_gather_score: neg_ids is null, or neg_score is null without a fused loss");
    RSA_CHECK_ARG(a->packed_keys == nullptr || (a->sampler == RSA_SAMPLER_GIVEN && a->num_neg == 1 && !a->pos_ids),
                  "rsa_fused_sample_gather_score: packed_keys needs sampler GIVEN, num_neg == 1 and no positives");
    if (a->sampler != RSA_SAMPLER_GIVEN)
      RSA_CHECK_ARG(a->grid_threads > 0 && (a->offset & 3) == 0, "rsa_fused_sample_gather_score: bad philox state");
    if (a->sampler == RSA_SAMPLER_POPULAR) {
      RSA_CHECK_ARG(a->table && a->pop_prob, "rsa_fused_sample_gather_score: popularity tables missing");
      if (a->cdf_lines != nullptr)
        RSA_CHECK_ARG(a->lines_log2 >= 0 && a->lines_log2 <= 28 && ((uintptr_t)a->cdf_lines & 127) == 0,
                      "rsa_fused_sample_gather_score: cdf_lines must be 128-byte aligned with lines_log2 in [0, 28]");
      else
        RSA_CHECK_ARG(a->guide && a->guide_log2 >= 0 && a->guide_log2 <= 28,
                      "rsa_fused_sample_gather_score: guide table missing");
    }
  }
  // (ids given + SampledSoftmax epilogue: pos_logp / neg_logp are INPUTS, no table needed)
  RSA_CHECK_ARG(a->pos_logp == nullptr || a->pop_prob != nullptr ||
                    (a->sampler == RSA_SAMPLER_GIVEN && a->fused_loss == RSA_LOSS_SSM + 1),
                "rsa_fused_sample_gather_score: pos_logp needs pop_prob");
  RSA_CHECK_ARG((a->pos_score == nullptr && a->pos_logp == nullptr) || a->pos_ids != nullptr,
                "rsa_fused_sample_gather_score: pos outputs need pos_ids");

  FwdParams p;
  p.item_table = a->item_table;
  p.query = a->query;
  p.query_index = a->query_index;
  p.pos_ids = a->pos_ids;
  p.table = a->table;
  p.pop_prob = a->pop_prob;
  p.table_prob = a->table_prob;
  p.lut = a->cdf_lut;
  p.lines = a->cdf_lines;
  p.lines_log2 = a->lines_log2;
  p.guide = a->guide;
  p.neg_ids = a->neg_ids;
  p.neg_logp = a->neg_logp;
  p.pos_logp = a->pos_logp;
  p.pos_score = a->pos_score;
  p.neg_score = a->neg_score;
  p.row_loss = nullptr;
  p.dpos = nullptr;
  p.dneg = nullptr;
  p.qgrad = nullptr;
  p.loss_out = nullptr;
  p.done_counter = nullptr;
  p.offset_dev = a->offset_dev;
  p.loss_partials = nullptr;
  p.packed_keys = a->packed_keys;
  p.solo_flags = nullptr;
  p.upd_scale = nullptr;
  p.item_rw = nullptr;
  p.step_dropped = nullptr;
  p.overflow_sticky = nullptr;
  p.seg_stride = 0;
  p.n_items = a->n_items;
  p.n_query_rows = a->n_query_rows;
  p.n_queries = a->n_queries;
  p.mean_queries = a->n_queries;
  p.pc = PhiloxCall{a->seed, a->offset >> 2, a->grid_threads, a->elem_base};
  p.tiles_per_batch = 0;
  p.batch_numel = 0;
  p.batch_offset4 = 0;
  p.dim = a->dim;
  p.num_neg = a->num_neg;
  p.sampler = a->sampler;
  p.mask_pad_pos = a->mask_pad_pos;
  p.guide_log2 = a->guide_log2;
  p.score_mode = a->score_mode;

  const bool cos = a->score_mode != RSA_SCORE_IP;   // cosine and Euclidean both need the squared norms
  hipStream_t s = (hipStream_t)stream;
  int rc = RSA_OK;
  if (numel == 0) {
    // no negatives: score the positives only, as one "negative" column of given ids == pos ids
    // (num_neg = 1, QU = false) writing into pos_score through the positive path.
    if (a->pos_score == nullptr && a->pos_logp == nullptr) return RSA_OK;
    rsa::set_error("rsa_fused_sample_gather_score: num_neg == 0 is not supported (use rsa_embedding_gather)");
    return RSA_ERR_UNSUPPORTED;
  }
  p.numel = numel;
  const bool qu = (a->num_neg % 64) == 0;
  const bool bpr = a->fused_loss == RSA_LOSS_BPR + 1, ssm = a->fused_loss == RSA_LOSS_SSM + 1;
  RSA_CHECK_ARG(a->fused_loss == 0 || bpr || ssm,
                "rsa_fused_sample_gather_score: fused_loss=%d not supported (0 = none, 1 = BPR, 2 = SampledSoftmax)",
                a->fused_loss);
  RSA_CHECK_ARG(a->query_grad == nullptr || bpr || ssm,
                "rsa_fused_sample_gather_score: query_grad is an output of the fused loss epilogue (fused_loss != 0)");
  if (a->n_batches > 1) {
    // a queue of n_batches independent batches of n_queries / n_batches queries each, consumed by ONE resident grid
    RSA_CHECK_ARG(qu && !ssm && !(bpr && a->num_neg != 64) && a->sampler != RSA_SAMPLER_GIVEN && a->n_queries % a->n_batches == 0 && a->offset_dev == nullptr &&
                      a->loss_out == nullptr && a->elem_base == 0 && (a->batch_offset_step & 3) == 0 &&
                      (a->sampler != RSA_SAMPLER_POPULAR || a->cdf_lines != nullptr || a->cdf_lut == nullptr),
                  "rsa_fused_sample_gather_score: n_batches > 1 needs an in-kernel sampler, num_neg %% 64 == 0, equal batches, no "
                  "loss_out (per-batch means are the caller's), no offset_dev / elem_base, and not the LUT form of the popularity lookup");
    p.batch_numel = (a->n_queries / a->n_batches) * (int64_t)a->num_neg;
    RSA_CHECK_ARG(p.batch_numel / 64 < (1ll << 31), "rsa_fused_sample_gather_score: batches of more than 2^37 elements");
    p.tiles_per_batch = (uint32_t)(p.batch_numel / 64);
    p.batch_offset4 = a->batch_offset_step >> 2;
    p.mean_queries = a->n_queries / a->n_batches;
  }
  if (bpr || ssm) {
    RSA_CHECK_ARG(qu && a->pos_ids && a->pos_score && a->row_loss,
                  "rsa_fused_sample_gather_score: the fused loss epilogue needs num_neg %% 64 == 0, pos_ids, pos_score "
                  "and row_loss");
    p.row_loss = a->row_loss;
    p.dpos = a->dpos;
    p.dneg = a->dneg;
    if (a->loss_out != nullptr) {
      RSA_CHECK_ARG(a->reduce_scratch != nullptr,
                    "rsa_fused_sample_gather_score: loss_out needs reduce_scratch (rsa_scratch_bytes() bytes, zeroed once)");
      char* sc = reinterpret_cast<char*>(a->reduce_scratch);
      p.loss_out = a->loss_out;
      p.done_counter = reinterpret_cast<unsigned int*>(sc + SCRATCH_COUNTER);
      p.loss_partials = reinterpret_cast<float*>(sc + SCRATCH_FUSED_PARTIALS);
    }
    if (a->query_grad != nullptr) {
      RSA_CHECK_ARG(!cos && (a->dim == 32 || a->dim == 64 || a->dim == 128 || a->dim == 256),
                    "rsa_fused_sample_gather_score: query_grad needs the inner-product scorer and dim in "
                    "{32, 64, 128, 256}");
      p.qgrad = a->query_grad;
    }
    if (a->solo_flags != nullptr) {
      RSA_CHECK_ARG(bpr && a->num_neg == 64 && a->query_grad && a->upd_scale && a->sampler == RSA_SAMPLER_GIVEN &&
                        a->packed_keys == nullptr && !a->mask_pad_pos,
                    "rsa_fused_sample_gather_score: the in-forward update (solo_flags) needs fused_loss = BPR with num_neg == 64, "
                    "given ids, query_grad and upd_scale");
      p.solo_flags = a->solo_flags;
      p.upd_scale = a->upd_scale;
      p.item_rw = const_cast<float*>(a->item_table);
    }
  }
  if (ssm) {
    RSA_CHECK_ARG(!cos && (a->dim == 32 || a->dim == 64 || a->dim == 128 || a->dim == 256) && a->packed_keys == nullptr,
                  "rsa_fused_sample_gather_score: the SampledSoftmax epilogue needs the inner-product scorer and dim in "
                  "{32, 64, 128, 256}");
    switch (a->dim) {
      case 32: return launch_ssm<8>(p, s);
      case 64: return launch_ssm<16>(p, s);
      case 128: return launch_ssm<32>(p, s);
      default: return launch_ssm<64>(p, s);
    }
  }
  if (bpr && a->num_neg != 64 && !cos && a->packed_keys == nullptr &&
      (a->dim == 32 || a->dim == 64 || a->dim == 128 || a->dim == 256)) {
    // queries longer than one tile: the workgroup-per-query walk (deterministic, no atomics, nothing to zero)
    switch (a->dim) {
      case 32: return launch_bpr_walk<8>(p, s);
      case 64: return launch_bpr_walk<16>(p, s);
      case 128: return launch_bpr_walk<32>(p, s);
      default: return launch_bpr_walk<64>(p, s);
    }
  }
  RSA_CHECK_ARG(!bpr || a->num_neg == 64,
                "rsa_fused_sample_gather_score: the BPR epilogue with num_neg > 64 needs the inner-product scorer and dim in "
                "{32, 64, 128, 256}");
  switch (a->dim) {
    case 32: rc = launch_fwd<8, false>(p, cos, qu, s); break;
    case 64: rc = launch_fwd<16, false>(p, cos, qu, s); break;
    case 128: rc = launch_fwd<32, false>(p, cos, qu, s); break;
    case 256: rc = launch_fwd<64, false>(p, cos, qu, s); break;
    default: rc = launch_fwd<64, true>(p, cos, qu, s); break;
  }
  return rc;
}


// Owner side of the sharded exchange, segment form (include/recstudio_amd.h, "Version 2 of the fixed-capacity exchange")
extern "C" int rsa_shard_score_segments(const float* item_table, int64_t n_rows, int32_t dim, const float* query,
                                        int64_t n_query_rows, const int64_t* keys, int64_t n_segments, int64_t stride,
                                        float* scores, int32_t* step_dropped, int32_t* overflow_sticky, rsa_stream_t stream) {
  RSA_CHECK_ARG(n_segments >= 0 && stride > RSA_SHARD_HDR, "rsa_shard_score_segments: bad sizes");
  const int64_t numel = n_segments * stride;
  if (numel == 0) return RSA_OK;
  RSA_CHECK_ARG(numel < (1ll << 31), "rsa_shard_score_segments: more than 2^31 slots");
  RSA_CHECK_ARG(item_table && query && keys && scores, "rsa_shard_score_segments: null pointer");
  RSA_CHECK_ARG(dim >= 4 && dim <= 1024 && dim % 4 == 0, "rsa_shard_score_segments: dim=%d must be a multiple of 4 in [4, 1024]", dim);
  RSA_CHECK_ARG(n_rows >= 1 && n_rows < (1ll << 32) && n_query_rows >= 1 && n_query_rows < (1ll << 31),
                "rsa_shard_score_segments: table sizes out of range");
  FwdParams p = {};
  p.item_table = item_table;
  p.query = query;
  p.packed_keys = keys;
  p.neg_score = scores;
  p.step_dropped = step_dropped;
  p.overflow_sticky = overflow_sticky;
  p.seg_stride = (int32_t)stride;
  p.n_items = n_rows;
  p.n_query_rows = n_query_rows;
  p.n_queries = numel;
  p.mean_queries = numel;
  p.numel = numel;
  p.pc = PhiloxCall{0, 0, 256, 0};
  p.dim = dim;
  p.num_neg = 1;
  p.sampler = RSA_SAMPLER_GIVEN;
  p.score_mode = RSA_SCORE_IP;
  hipStream_t s = (hipStream_t)stream;
  switch (dim) {
    case 32: return launch_fwd<8, false>(p, false, false, s);
    case 64: return launch_fwd<16, false>(p, false, false, s);
    case 128: return launch_fwd<32, false>(p, false, false, s);
    case 256: return launch_fwd<64, false>(p, false, false, s);
    default: return launch_fwd<64, true>(p, false, false, s);
  }
}
