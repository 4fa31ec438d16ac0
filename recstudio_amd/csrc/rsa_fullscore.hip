// Full-catalog scoring with the fp32 MFMA (gfx950): scores = Q [B,d] x Items[1:N]^T.
//
//   InnerProductScorer ([B,D],[N,D]) case     recstudio/model/scorer.py:16
//   called from BaseRetriever.forward         baseretriever.py:183-186  (FullScoreLoss training)
//   and BaseRetriever.topk                    baseretriever.py:384      (torch.topk(k + more))
//   SoftmaxLoss                               loss_func.py:41           (logsumexp over the catalog)
//
// GEMM kernel.  v_mfma_f32_32x32x2_f32 with the ITEM rows as the A operand and the QUERY rows as
// the B operand ("swapped" orientation): a lane then owns one query column (lane & 31) and 16 item
// rows of every 32x32 output tile, so the per-query reductions (running max / sum-exp of the
// online logsumexp) stay inside the lane -- no cross-lane traffic in the main loop.  The K index is
// permuted (lane half h owns k in [h*D/2, (h+1)*D/2)) so that each lane's A and B operands are
// CONTIGUOUS floats of one row: 16-byte LDS / global reads instead of stride-2 gathers; a sum over k
// does not care about the order.  A workgroup = 4 waves = 128 queries; the 32-item A tile is staged
// through LDS (row stride D+4 floats: conflict-free ds_read_b128) and shared by the 4 waves; the
// query operand lives in registers for the whole item range.  Exact fp32 (fmaf chain), 157 TFLOP/s peak.
//
// Top-k without materialising [B, N] (default when the caller does not ask for `scores`):
//   A. the same GEMM over an evenly spread SAMPLE of 32-item tiles (<= 32 768 items) -> per-query
//      threshold T = the j-th largest sampled score, j sized so that ~3k (+ margin) catalog items beat T;
//   B. the full GEMM with a filter epilogue: a lane appends (score, item) to its query's candidate
//      list only when score > T -- 16 compares per lane per tile, a rare atomic;
//   C. exact radix select (11/11/10 bits) + bitonic sort over the few hundred candidates of each query.
// A query whose candidate count ends up outside [k, 4096] (or that overflows a 32-slot segment) (threshold too tight / too loose: a catalog whose
// scores are far from exchangeable across the sampled tiles) is flagged and recomputed exactly by
// D. a per-row radix select that evaluates the dot products on the fly.  The result is always the exact
// top-k (ties -> smaller id).  With `scores` given, the radix select runs on the materialised rows.
#include "rsa_common.hpp"
#include <type_traits>

#ifndef RSA_FS_MIN_BLOCKS
#define RSA_FS_MIN_BLOCKS 1
#endif
#ifndef RSA_FS_DMA
#define RSA_FS_DMA 1
#endif
#ifndef RSA_FS_DQ_MIN_BLOCKS
#define RSA_FS_DQ_MIN_BLOCKS 2     // the dQ variant: at most 256 registers, two waves per SIMD
#endif

namespace rsa {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int QB = 128;   // queries per workgroup (4 waves x 32)
constexpr int TI = 32;    // items per MFMA tile
#ifndef RSA_FS_STG
#define RSA_FS_STG 1
#endif
constexpr int STG = RSA_FS_STG;   // MFMA tiles per LDS stage: one workgroup barrier per STG * TI items.  Measured at
                                  // B = 2048, N = 1e6: STG = 2 (half the barriers, 67 KB LDS) 115.1 vs 115.8 TFLOP/s for
                                  // STG = 1, and the score-writing variants lose a wave per SIMD -- the barrier is not
                                  // what limits the MFMA pipe; neither are a second accumulation chain (111 vs 117) nor
                                  // 3 waves/SIMD via launch bounds (+2 %, spills in the other variants) nor the epilogue
                                  // (GEMM core alone: same time)

// Candidate lists of the filter epilogue.  Every (query, item-range split, lane half) owns a private
// segment of SEG slots and counts in a register: plain fire-and-forget stores, no atomics (a returning
// global atomic inside the tile loop stalls the wave for a memory round trip and cost 25 % of the GEMM).
constexpr int SEG = 32;
constexpr int OVF = 1024;
struct FilterArgs {
  const float* thr;     // [n_query] per-query threshold (null: no filtering)
  int32_t* seg_cnt;     // [n_query, splits, 2] candidates seen per segment (may exceed SEG: overflow marker)
  float2* cand;         // [n_query, splits, 2, SEG] {score, item id as int32 bits}: ONE 8-byte store per candidate (every
                        // scattered store is its own 64-byte write request: two arrays cost two)
  // SCORES epilogue transform (backward of the full softmax): with sm_lse set the stored value is
  // sm_scale[q] * exp(score - sm_lse[q]) instead of the raw score
  const float* sm_lse;    // [n_query] or null
  const float* sm_scale;  // [n_query] or null (1)
  // Overflow list of a query (OVF slots): a candidate that finds its private segment full is appended here
  // with a returning atomic -- paid only on that rare path.  Scores that trend along the id axis (ids sorted
  // by popularity, say) put most of a query's candidates into a few item ranges; without this list every such
  // row fell back to the exact single-workgroup recompute (17 ms per row at N = 1e6).
  int32_t* ovf_cnt;     // [n_query], zeroed by the host
  float* ovf_val;       // [n_query, OVF]
  int32_t* ovf_idx;     // [n_query, OVF]
  // Cosine / Euclidean scores (MODE != 0): per-item and per-query operands from rsa_row_sqnorm
  const float* item_aux;   // [n_items - 1], entry i-1 for item row i
  const float* query_aux;  // [n_query]
  // DQ variant (softmax backward): this item range's share of d/d query = sum_i P[q, i] * item_i, [splits, n_query, D]
  float* dq_part;
};

// tile_stride == 1: the workgroup walks the contiguous item range [1 + bx*items_per_split, ...).
// tile_stride  > 1: SAMPLE mode -- "item" positions are sample positions; sample tile s reads catalog tile
// s*tile_stride, and scores are written to a dense [n_query, score_ld] sample matrix.
// MODE: rsa_score_mode.  For RSA_SCORE_COS / RSA_SCORE_EUC the tile's dot products are turned into the score in the
// epilogue (cos = dot * item_aux * query_aux, euc = 2 dot - item_aux - query_aux) before logsumexp / filter / store; the
// inner-product instantiation is unchanged.
// DQ (with the softmax score transform only): the tile of P = row_scale * softmax the epilogue has just produced is fed
// straight back into the matrix cores as the B operand of a second product, dQ^T[d, q] += X^T[d, item] * P[item, q] --
// a lane's 16 accumulator registers ARE its B-operand values when the 32 items of the tile are taken as the K index in
// the order the accumulator layout holds them (step r: items row(r, h = 0) and row(r, h = 1)), so nothing moves between
// lanes; the A operand X[item(r, h)][32 db + j] is a conflict-free 4-byte LDS read of the staged item tile.  dQ stays in
// 16 * D/32 accumulator registers for the whole item range and leaves the kernel once (per-range partials).
// PW = false (DQ only): the P tile feeds the second product and is NOT written -- the backward that never holds [B, N]
// (rsa_fullscore_softmax_dq with probs == NULL; d/d items then comes from rsa_fullscore_softmax_dw, rsa_dx.hip).
// FL = true (DQ, PW = false): FLASH forward of the full softmax -- logsumexp AND d lse/d query = softmax @ items in ONE pass over the
// catalog (rsa_fullscore_lse_grad).  The softmax tile is formed against a running per-query REFERENCE instead of a known
// logsumexp: P = exp(S - ref); the reference only moves when a tile's maximum exceeds it by more than FL_SLACK (then the
// dQ accumulators and the running sum are rescaled by exp(ref_old - ref_new): rare after the first tiles), so P <= e^FL_SLACK
// and nothing overflows.  Per item range the kernel leaves (ref, sum P) and the unnormalised sum P * item; the merge kernel
// brings the ranges to a common reference.  With it a training step needs FOUR products of 2 B N d flop and no [B, N] matrix:
// this pass (2) + rsa_fullscore_softmax_dw (2).
#ifndef RSA_FS_FL_SLACK
#define RSA_FS_FL_SLACK 8.f
#endif
template <int D, bool LSE, bool SCORES, bool FILTER, int MODE = 0, bool DQ = false, bool PW = true, bool FL = false>
__global__ __launch_bounds__(256, DQ ? RSA_FS_DQ_MIN_BLOCKS : RSA_FS_MIN_BLOCKS) void fullscore_kernel(const float* __restrict__ item_table, int64_t n_items,
                                                        const float* __restrict__ query, int64_t n_query,
                                                        float* __restrict__ scores, int64_t score_ld,
                                                        float2* __restrict__ lse_part, int splits,
                                                        int64_t items_per_split, int64_t tile_stride,
                                                        int64_t n_positions, FilterArgs flt) {
  constexpr int KH = D / 2;         // k values per lane half
  constexpr int LD = D + 4;         // padded LDS row stride (floats)
  constexpr int V4 = D / 4;         // float4 per row
  // two stage buffers; the dQ variant reads tile u-1 again (second product) while tile u+1 is being committed: three
  constexpr int NBUF = DQ ? 3 : 2;
  // DMA2: d = 128 tiles staged by direct HBM -> LDS loads (global_load_lds_dwordx4: no staging registers, no ds_write)
  // into STATICALLY distinct buffers -- the main loop is unrolled by the buffer count, so which buffer is read and
  // which is filled is known at compile time.  With one array indexed at run time the compiler cannot prove that the
  // asynchronous fill and this tile's LDS reads touch different buffers and waits (vmcnt(0)) in front of the MFMA
  // chain: 4.04 -> 4.75 ms; with distinct arrays 4.04 -> 3.92 ms.  Rows 2i, 2i+1 are contiguous (1024 bytes = one
  // wave-wide load) and each pair is followed by a 32-byte pad (interleaving the two rows' chunks to make "lane j
  // reads row j" conflict-free instead of 2-way measured the same).
  constexpr bool DMA2 = RSA_FS_DMA && D == 128 && STG == 1;
  constexpr int PAIR_F = 2 * D + 8;
  constexpr int TILE_F = DMA2 ? (TI / 2) * PAIR_F : TI * STG * LD;
  __shared__ __attribute__((aligned(16))) float tile0[TILE_F];
  __shared__ __attribute__((aligned(16))) float tile1[TILE_F];
  __shared__ __attribute__((aligned(16))) float tile2[NBUF == 3 ? TILE_F : 4];
  auto tbuf = [&](int b) __attribute__((always_inline)) -> float* { return b == 0 ? tile0 : (b == 1 ? tile1 : tile2); };
  auto lds_off = [](int row, int c4) __attribute__((always_inline)) -> int {
    return DMA2 ? (row >> 1) * PAIR_F + (row & 1) * D + c4 * 4 : row * LD + c4 * 4;
  };
  __shared__ float tpose[4][32][33];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int64_t q = (int64_t)blockIdx.y * QB + wave * 32 + j;

  float bq[KH];
  {
    const bool qv = q < n_query;
    const float4* src = reinterpret_cast<const float4*>(query + (size_t)(qv ? q : 0) * D + h * KH);
#pragma unroll
    for (int c = 0; c < KH / 4; ++c) {
      float4 v = src[c];
      if (!qv) v = make_float4(0.f, 0.f, 0.f, 0.f);
      bq[4 * c + 0] = v.x; bq[4 * c + 1] = v.y; bq[4 * c + 2] = v.z; bq[4 * c + 3] = v.w;
    }
  }

  // positions are 1-based like item ids: position p of a contiguous walk IS item p; in sample mode
  // position p maps to item 1 + ((p-1)/TI)*tile_stride*TI + (p-1)%TI
  const int64_t p_limit = n_positions + 1;
  const int64_t i_begin = 1 + (int64_t)blockIdx.x * items_per_split;
  int64_t i_end = i_begin + items_per_split;
  if (i_end > p_limit) i_end = p_limit;
  const int n_tiles = i_begin < i_end ? (int)((i_end - i_begin + TI - 1) / TI) : 0;
  float thr = INFINITY;
  if (flt.thr != nullptr && q < n_query) thr = flt.thr[q];
  float q_aux = 0.f;
  if constexpr (MODE != 0) q_aux = flt.query_aux[q < n_query ? q : 0];
  // the lane's 16 rows of a tile are 4 runs of 4 consecutive items: 4 aligned 16-byte loads of item_aux per tile
  auto to_score = [&](const f32x16& acc, int64_t i0) __attribute__((always_inline)) -> f32x16 {
    if constexpr (MODE == 0) {
      return acc;
    } else {
      const int64_t item0 = tile_stride == 1 ? i0 : 1 + ((i0 - 1) / TI) * tile_stride * TI;
      f32x16 out;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // (the last tile of a range may reach up to TI - 1 rows past the table: item_aux is allocated with 64 floats
        // of padding, those rows are masked by the callers)
        const int64_t a0 = item0 - 1 + 8 * g + 4 * h;
        const float4 xa = *reinterpret_cast<const float4*>(flt.item_aux + a0);
        const float xs[4] = {xa.x, xa.y, xa.z, xa.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = acc[4 * g + r];
          out[4 * g + r] = MODE == RSA_SCORE_COS ? d * xs[r] * q_aux : __fmaf_rn(2.f, d, -xs[r]) - q_aux;
        }
      }
      return out;
    }
  };
  float my_lse = 0.f, my_scale = 1.f;
  const bool softmax_out = SCORES && flt.sm_lse != nullptr;
  if (softmax_out && q < n_query) {
    my_lse = flt.sm_lse[q];
    if (flt.sm_scale != nullptr) my_scale = flt.sm_scale[q];
  }
  int32_t my_cnt = 0;
  const size_t seg = ((size_t)(q < n_query ? q : 0) * splits + blockIdx.x) * 2 + h;

  constexpr int LOADS = (TI * STG * V4) / 256;    // float4 per thread per stage (D=128, STG=2: 8)
  static_assert((TI * STG * V4) % 256 == 0, "stage must split evenly over the workgroup");
  float4 stage[LOADS];
  // Stages that lie completely inside the item range of a contiguous walk (all but the last one or two) are
  // loaded through per-thread pointers that advance by a constant: no index arithmetic, range checks or
  // zero-selects in the main loop -- VALU instructions there compete with the MFMA chain for issue slots
  // (measured: the same loop without its loads/epilogue VALU runs at 148 instead of 119 TFLOP/s).
  const int64_t n_full_tiles = tile_stride == 1 && i_begin < i_end ? (i_end - i_begin) / TI : 0;
  const int n_full_stages = (int)(n_full_tiles / STG);
  const float4* fast_src[LOADS];
#pragma unroll
  for (int f = 0; f < LOADS; ++f) {
    const int idx = f * 256 + tid;
    const int row = idx / V4, c4 = idx - row * V4;
    fast_src[f] = reinterpret_cast<const float4*>(item_table + (size_t)(n_full_stages > 0 ? i_begin + row : 0) * D) + c4;
  }
  constexpr int64_t STAGE_F4 = (int64_t)TI * STG * V4;     // float4 per stage
  // direct loads: the wave's instruction f fills pair block f*4 + wave; lane l is 16-byte slot l of it = row (l >> 5) of
  // the pair, chunk l & 31
  const float4* dma_src[DMA2 ? LOADS : 1];
  if constexpr (DMA2) {
#pragma unroll
    for (int f = 0; f < LOADS; ++f)
      dma_src[f] = reinterpret_cast<const float4*>(item_table + (size_t)(n_full_stages > 0 ? i_begin + 2 * (f * 4 + wave) + (lane >> 5) : 0) * D) + (lane & 31);
  }
  auto dma_fetch = [&](int t, float* dst) __attribute__((always_inline)) {
    if constexpr (DMA2) {
#pragma unroll
      for (int f = 0; f < LOADS; ++f)
        __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)(dma_src[f] + (int64_t)t * STAGE_F4),
                                         (__attribute__((address_space(3))) void*)(dst + (f * 4 + wave) * PAIR_F), 16, 0, 0);
    }
  };
  auto fetch = [&](int t) __attribute__((always_inline)) {
    if (t < n_full_stages) {
#pragma unroll
      for (int f = 0; f < LOADS; ++f) stage[f] = fast_src[f][(int64_t)t * STAGE_F4];
      return;
    }
#pragma unroll
    for (int f = 0; f < LOADS; ++f) {
      const int idx = f * 256 + tid;
      const int row = idx / V4, c4 = idx - row * V4;
      const int64_t pos = i_begin + (int64_t)t * (TI * STG) + row;
      int64_t item = tile_stride == 1 ? pos : 1 + ((pos - 1) / TI) * tile_stride * TI + (pos - 1) % TI;
      const bool ok = pos < i_end && item < n_items;
      item = item < n_items ? item : n_items - 1;          // unconditional load from a valid row, zeroed by select
      const float4 v = reinterpret_cast<const float4*>(item_table + (size_t)item * D)[c4];
      stage[f] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < LOADS; ++f) {
      const int idx = f * 256 + tid;
      const int row = idx / V4, c4 = idx - row * V4;
      *reinterpret_cast<float4*>(tbuf(buf) + lds_off(row, c4)) = stage[f];
    }
  };

  float run_m = -INFINITY, run_s = 0.f;
  static_assert(!DQ || (SCORES && !LSE && !FILTER && MODE == 0 && STG == 1 && D % 32 == 0), "DQ: softmax-backward variant only");
  static_assert(PW || DQ, "PW = false: only the dQ variant can do without the score store");
  static_assert(!FL || (DQ && !PW), "FL: the flash forward is the dQ variant without the score store");
  float fl_ref = -INFINITY, fl_sum = 0.f;     // FL: the query's running reference (same in both lane halves) and this half's sum of P
  constexpr int DB = D / 32;
  f32x16 dq[DQ ? DB : 1];
  if constexpr (DQ) {
#pragma unroll
    for (int db = 0; db < DB; ++db) dq[db] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  }

  // ---- per-tile epilogue pieces.  They are applied to the PREVIOUS tile's accumulators while the current
  // tile's MFMA chain (64 dependent 64-cycle instructions) is in flight.
  // (a) branch-free online logsumexp: pure VALU, interleaved with the MFMAs by the sched_group_barriers below
  float tile_max = -INFINITY;      // max of the tile's (in-range) scores, left by lse_update for the candidate filter
  auto lse_update = [&](const f32x16& acc, int64_t i0, auto masked) __attribute__((always_inline)) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      v[r] = acc[r];
      if constexpr (decltype(masked)::value) {     // only the last tiles of a range can be partial
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (!(i0 + row < i_end)) v[r] = -INFINITY;
      }
    }
    float tmax = fmaxf(fmaxf(v[0], v[1]), v[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) tmax = fmaxf(fmaxf(tmax, v[r]), v[r + 1]);    // v_max3_f32
    tmax = fmaxf(tmax, v[15]);
    tile_max = tmax;
    const float m_new = fmaxf(run_m, tmax);
    const float m_safe = m_new == -INFINITY ? 0.f : m_new;   // nothing seen yet: exp(-inf - 0) = 0 everywhere
    // exp(x - m) = 2^(x * log2(e) - m * log2(e)): one fma + one v_exp per element (VALU slots are what the
    // epilogue costs: they are stolen from the MFMA chain's issue)
    constexpr float LOG2E = 1.4426950408889634f;
    const float c = m_safe * LOG2E;
    float sum = run_s * __builtin_amdgcn_exp2f(__fmaf_rn(run_m, LOG2E, -c));
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += __builtin_amdgcn_exp2f(__fmaf_rn(v[r], LOG2E, -c));
    run_m = m_new;
    run_s = sum;
  };
  // (b) the parts with memory side effects (score rows, candidate lists)
  auto emit = [&](const f32x16& acc, int64_t i0, int pbuf, auto masked) __attribute__((always_inline)) {
    if constexpr (SCORES) {
      // transpose the wave's 32 (items) x 32 (queries) tile through LDS so that every half-wave
      // writes 128 contiguous bytes of one query's score row
      f32x16 pv;
      if constexpr (FL) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          v[r] = acc[r];
          if constexpr (decltype(masked)::value) {     // only the last tiles of a range can be partial
            if (!(i0 + (r & 3) + 8 * (r >> 2) + 4 * h < i_end)) v[r] = -INFINITY;
          }
        }
        float tmax = fmaxf(fmaxf(v[0], v[1]), v[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) tmax = fmaxf(fmaxf(tmax, v[r]), v[r + 1]);
        tmax = fmaxf(tmax, v[15]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));            // the query's other 16 items of the tile
        if (__ballot(tmax > fl_ref + RSA_FS_FL_SLACK) != 0ull) {   // wave-uniform, rare: some query's reference moves
          const float nr = tmax > fl_ref + RSA_FS_FL_SLACK ? tmax : fl_ref;
          const float sc = nr == fl_ref ? 1.f : __expf(fl_ref - nr);     // (first tile: exp(-inf) = 0 on accumulators that are 0)
#pragma unroll
          for (int db = 0; db < DB; ++db) dq[db] *= sc;
          fl_sum *= sc;
          fl_ref = nr;
        }
        const float ref = fl_ref == -INFINITY ? 0.f : fl_ref;   // (nothing seen yet: every score is -inf, P = 0)
        float ts = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pv[r] = __expf(v[r] - ref);
          ts += pv[r];
        }
        fl_sum += ts;
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pv[r] = softmax_out ? my_scale * __expf(acc[r] - my_lse) : acc[r];
          if constexpr (PW) tpose[wave][j][(r & 3) + 8 * (r >> 2) + 4 * h] = pv[r];
        }
      }
      if constexpr (DQ) {
        // rows past the item range are zero in LDS (fetch), so their (finite) P values add nothing
        // output column block db of lane j is d = DB * j + db: the DB A-operand values of a step are DB consecutive
        // floats of the item row -- one 16-byte LDS read at D = 128 instead of four 4-byte ones
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float* xr = tbuf(pbuf) + lds_off((r & 3) + 8 * (r >> 2) + 4 * h, 0) + DB * j;
          float xa[DB];
          if constexpr (DB == 4) {
            const float4 v = *reinterpret_cast<const float4*>(xr);
            xa[0] = v.x; xa[1] = v.y; xa[2] = v.z; xa[3] = v.w;
          } else if constexpr (DB == 2) {
            const float2 v = *reinterpret_cast<const float2*>(xr);
            xa[0] = v.x; xa[1] = v.y;
          } else {
#pragma unroll
            for (int db = 0; db < DB; ++db) xa[db] = xr[db];
          }
#pragma unroll
          for (int db = 0; db < DB; ++db)
            dq[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[db], pv[r], dq[db], 0, 0, 0);
        }
      }
      if constexpr (PW) {
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes have landed
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) {
          const int qq = 2 * tt + h;
          const int64_t qg = (int64_t)blockIdx.y * QB + wave * 32 + qq;
          const int64_t item = i0 + j;
          if (qg < n_query && item < i_end) scores[(size_t)qg * score_ld + (item - 1)] = tpose[wave][qq][j];
        }
      }
    }
    if constexpr (FILTER) {
      // candidate filter: almost never taken once the threshold is in place
      // with the logsumexp in the same pass the tile maximum is already there
      float best = -INFINITY;
      if constexpr (LSE) {
        best = tile_max;
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) best = fmaxf(best, acc[r]);
      }
      if (best > thr) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
          if (acc[r] > thr && i0 + row < i_end) {
            if (my_cnt < SEG) {
              flt.cand[seg * SEG + my_cnt] = make_float2(acc[r], __int_as_float((int32_t)(i0 + row)));
              ++my_cnt;
            } else {
              const int pos = atomicAdd(&flt.ovf_cnt[q], 1);
              if (pos < OVF) {
                flt.ovf_val[(size_t)q * OVF + pos] = acc[r];
                flt.ovf_idx[(size_t)q * OVF + pos] = (int32_t)(i0 + row);
              }
            }
          }
        }
      }
    }
  };
  auto mfma_tile = [&](int buf, int sub) __attribute__((always_inline)) -> f32x16 {
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float* arow = tbuf(buf) + lds_off(sub * TI + j, 0) + h * KH;   // lane's item row of this tile, its k half
#pragma unroll
    for (int c = 0; c < KH / 4; ++c) {
      const float4 a = *reinterpret_cast<const float4*>(arow + 4 * c);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[4 * c + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[4 * c + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[4 * c + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[4 * c + 3], acc, 0, 0, 0);
    }
    return acc;   // acc[r] = <item (i0 + row(r)), query q>, row(r) = (r & 3) + 8 * (r >> 2) + 4 * h
  };

  auto store_dq = [&]() __attribute__((always_inline)) {
    if constexpr (DQ) {
      if (q < n_query) {
        float* dst = flt.dq_part + ((size_t)blockIdx.x * n_query + q) * D;
#pragma unroll
        for (int r = 0; r < 16; ++r) {      // dq[db][r] = dQ[q][DB * i + db], i = row(r) of the accumulator layout
          const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
          for (int db = 0; db < DB; ++db) dst[DB * i + db] = dq[db][r];
        }
      }
    }
  };
  if (n_tiles == 0) {
    store_dq();
    if ((LSE || FL) && h == 0 && q < n_query) lse_part[(size_t)q * splits + blockIdx.x] = make_float2(-INFINITY, 0.f);
    if (FILTER && q < n_query) flt.seg_cnt[seg] = 0;
    return;
  }
  // Stage s = MFMA tiles s*STG .. s*STG+STG-1 staged together in LDS (double-buffered): one workgroup barrier
  // per STG tiles.  Tiles past n_tiles inside the last stage are all-masked (zero rows, results ignored).
  const int n_stages = (n_tiles + STG - 1) / STG;
  fetch(0);
  commit(0);
  __syncthreads();
  fetch(1);                               // out-of-range stages load a valid row and are zeroed
  f32x16 acc_prev = mfma_tile(0, 0);      // tile 0 peeled: the loop body is branch-free up to the rare filter hit
  if (STG == 1) {
    commit(1);
    __syncthreads();
  }
  // iteration u: MFMA chain of tile u with the epilogue of tile u-1 interleaved under it
  int rb_prv = 0, rb_cur = 1, rb_nxt = 2;    // dQ variant: buffers of tiles u-1, u, u+1 (rotating)
  auto iteration = [&](int u, auto masked, auto dma, auto curc) __attribute__((always_inline)) {
    constexpr bool USE_DMA = decltype(dma)::value;
    constexpr int CURC = decltype(curc)::value;                 // buffer of tile u when known at compile time, else -1
    const int st = u / STG, sub = u - st * STG;
    const int cur = CURC >= 0 ? CURC : (DQ ? rb_cur : (st & 1));
    const int nxt = CURC >= 0 ? (CURC + 1) % NBUF : (DQ ? rb_nxt : (cur ^ 1));
    const int prv = CURC >= 0 ? (CURC + NBUF - 1) % NBUF : (DQ ? rb_prv : (cur ^ 1));
    if constexpr (USE_DMA) {
      dma_fetch(st + 1, tbuf(nxt));        // lands in LDS under this stage's MFMA chain
      __builtin_amdgcn_sched_barrier(0);   // keep the loads at the top: the scheduler otherwise sinks them to the barrier
    }
    else if (sub == 0) fetch(st + 1);     // global loads fly under this stage's MFMA chains
    // dQ variant on the direct-load path: the previous tile's epilogue -- softmax transform, the score stores, the second
    // product -- goes FIRST, so that the vmcnt(0) at the bottom of the iteration (it counts stores as well as the direct
    // loads) finds the stores long completed instead of waiting a write round trip per tile: 8.88 -> 8.38 ms.  (The
    // plain score-writing variant loses by the same reordering, 4.82 -> 5.00 ms: its MFMA chain then starts behind the
    // epilogue's VALU / LDS work.)
    constexpr bool EMIT_FIRST = USE_DMA && DQ;
    if constexpr (EMIT_FIRST) emit(acc_prev, i_begin + (int64_t)(u - 1) * TI, prv, masked);
    const f32x16 acc = mfma_tile(cur, sub);
    if constexpr (MODE != 0) acc_prev = to_score(acc_prev, i_begin + (int64_t)(u - 1) * TI);
    if constexpr (LSE) {
      lse_update(acc_prev, i_begin + (int64_t)(u - 1) * TI, masked);
      // Pin the logsumexp state in front of the filter's branch: its exps are only needed by the NEXT tile's update, so
      // the compiler sank them past the branch into a block of their own at the loop head, outside the MFMA chain
      // they are meant to be interleaved with.  (Measured: no change in kernel time -- the SIMD's other wave covers
      // the gap -- but the loop now has the shape the sched_group_barriers describe.)
      if constexpr (FILTER) asm volatile("" : "+v"(run_s), "+v"(run_m));
      // one MFMA of this tile, then a couple of the previous tile's epilogue VALU ops, and so on
#pragma unroll
      for (int g = 0; g < KH; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
    }
    if constexpr (!EMIT_FIRST) emit(acc_prev, i_begin + (int64_t)(u - 1) * TI, prv, masked);
    if constexpr (USE_DMA) {
      __builtin_amdgcn_sched_barrier(0);    // ... and the wait at the bottom, behind the whole MFMA chain
      __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's direct loads have landed
      __syncthreads();
    } else if (sub == STG - 1) {
      commit(nxt);
      __syncthreads();
    }
    if constexpr (DQ && CURC < 0) {      // (the unrolled loop below runs whole rotations: the indices come back)
      rb_prv = cur;
      rb_cur = nxt;
      rb_nxt = prv;
    }
    acc_prev = acc;
  };
  const int total_tiles = n_stages * STG;
  const int fast_end = (int)(n_full_tiles < total_tiles - 1 ? n_full_tiles : total_tiles - 1);   // tiles 0..fast_end-1 are full
  int u = 1;
  using NoBuf = std::integral_constant<int, -1>;
  if constexpr (DMA2) {
    // NBUF tiles per trip, tile u in buffer u % NBUF (u starts at 1), while the stages behind them are full
    const int dma_last = (n_full_stages - 2 < fast_end ? n_full_stages - 2 : fast_end);
    for (; u + NBUF - 1 <= dma_last; u += NBUF) {
      iteration(u, std::false_type{}, std::true_type{}, std::integral_constant<int, 1>{});
      iteration(u + 1, std::false_type{}, std::true_type{}, std::integral_constant<int, 2 % NBUF>{});
      if constexpr (NBUF == 3) iteration(u + 2, std::false_type{}, std::true_type{}, std::integral_constant<int, 0>{});
    }
  }
  for (; u <= fast_end; ++u) iteration(u, std::false_type{}, std::false_type{}, NoBuf{});
  for (; u < total_tiles; ++u) iteration(u, std::true_type{}, std::false_type{}, NoBuf{});
  if constexpr (MODE != 0) acc_prev = to_score(acc_prev, i_begin + (int64_t)(total_tiles - 1) * TI);
  if constexpr (LSE) lse_update(acc_prev, i_begin + (int64_t)(total_tiles - 1) * TI, std::true_type{});
  emit(acc_prev, i_begin + (int64_t)(total_tiles - 1) * TI, DQ ? rb_prv : ((total_tiles - 1) & 1), std::true_type{});
  store_dq();
  if constexpr (FL) {       // (reference, sum of P over this item range): the same record the logsumexp variant leaves
    const float tot = fl_sum + __shfl_xor(fl_sum, 32, 64);
    if (h == 0 && q < n_query) lse_part[(size_t)q * splits + blockIdx.x] = make_float2(fl_ref, tot);
  }
  if (FILTER && q < n_query) flt.seg_cnt[seg] = my_cnt;
  if constexpr (LSE) {
    // fold the two k-halves' item subsets (lanes j and j+32 hold the same query)
    const float om = __shfl_xor(run_m, 32, 64), os = __shfl_xor(run_s, 32, 64);
    const float m = fmaxf(run_m, om);
    float s = 0.f;
    if (m > -INFINITY) s = run_s * __expf(run_m - m) + os * __expf(om - m);
    if (h == 0 && q < n_query) lse_part[(size_t)q * splits + blockIdx.x] = make_float2(m, s);
  }
}

__global__ __launch_bounds__(256) void lse_merge_kernel(const float2* __restrict__ part, int64_t n_query, int splits,
                                                        float* __restrict__ lse) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_query) return;
  float m = -INFINITY;
  for (int s = 0; s < splits; ++s) m = fmaxf(m, part[(size_t)q * splits + s].x);
  float acc = 0.f;
  for (int s = 0; s < splits; ++s) {
    const float2 p = part[(size_t)q * splits + s];
    if (p.x > -INFINITY) acc += p.y * expf(p.x - m);
  }
  lse[q] = m + logf(acc);
}

// ------------------------------------------------------------------ candidate threshold from group maxima
// thr[q] = the j-th largest of the G group maxima part[q][g].x (the running max the logsumexp instantiation of the GEMM
// leaves per (query, item range): in SAMPLE mode a range is a group of g sampled items).  The number of groups whose
// maximum beats a score T estimates the tail mass above T just as the number of SAMPLES above T does -- P(group max > T)
// = 1 - (1 - p)^g -- so the sampled scores need not be written (268 MB at B = 2048) and radix-selected (three passes
// over them): 180 + 194 us -> 150 + 6 us.  The threshold is a heuristic either way: too few / too many survivors are
// caught by the select pass and recomputed exactly.  One wave per query, rank by counting (G <= 1024).
__global__ __launch_bounds__(256) void group_threshold_kernel(const float2* __restrict__ part, int64_t n_query, int G, int ld, int j,
                                                              float* __restrict__ thr) {
  __shared__ float vals[4][1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * 4 + wave;
  const bool live = q < n_query;
  float* v = vals[wave];
  if (live)
    for (int g = lane; g < G; g += 64) v[g] = part[(size_t)q * ld + g].x;
  __syncthreads();
  float found = -INFINITY;
  if (live) {
    for (int g = lane; g < G; g += 64) {
      const float x = v[g];
      int rank = 0;                          // values ahead of x in (value desc, index asc) order
      for (int o = 0; o < G; ++o) {
        const float y = v[o];
        rank += (y > x || (y == x && o < g)) ? 1 : 0;
      }
      if (rank == j - 1) found = x;
    }
  }
#pragma unroll
  for (int off = 32; off; off >>= 1) found = fmaxf(found, __shfl_xor(found, off, 64));
  if (lane == 0 && live) thr[q] = found;
}

// ------------------------------------------------------------------ exact top-k of one row
__device__ __forceinline__ uint32_t order_key(float x) {   // monotone float -> uint
  const uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_value(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

enum { SEL_DENSE = 0, SEL_CAND = 1, SEL_RECOMPUTE = 2, SEL_THRESHOLD = 3 };

constexpr int CAND_MAX = 4096;   // candidates of one query that the select kernel compacts into LDS
struct SelectArgs {
  const float* values;      // DENSE/THRESHOLD: [n_rows, ld] score rows
  const float2* cand;       // CAND: [n_rows, n_seg, SEG] {score, item id bits}
  const int32_t* seg_cnt;   // CAND: [n_rows, n_seg] candidates seen per segment
  int32_t n_seg;            // CAND: segments per row (splits x 2)
  const int32_t* ovf_cnt;   // CAND: [n_rows] entries appended to the row's overflow list (may exceed OVF)
  const float* ovf_val;     // CAND: [n_rows, OVF]
  const int32_t* ovf_idx;   // CAND: [n_rows, OVF]
  int32_t* flags;           // CAND: out, 1 = row must be recomputed;  RECOMPUTE: in
  int64_t ld;               // row stride of `values`
  int64_t n_cols;           // DENSE/THRESHOLD/RECOMPUTE: elements per row
  const float* item_table;  // RECOMPUTE: item rows 1..n_cols, query rows
  const float* query;
  int32_t dim;
  float* thr_out;           // THRESHOLD: [n_rows] the k-th largest value of each row
  int32_t idx_base;         // DENSE: reported index = column + idx_base (1: item ids of a catalog row, 0: columns)
  int32_t score_mode;       // RECOMPUTE: rsa_score_mode with item_aux [n_cols] / query_aux [n_rows]
  const float* item_aux;
  const float* query_aux;
};

// Scans over the 2048 LDS bins of a 1024-thread workgroup, two barriers each: a thread owns bins 2 tid, 2 tid + 1, the
// pair sums are scanned inside the wave by shuffles and the 16 wave totals are added from LDS (the Hillis-Steele form
// used before took 22 barriers per scan, four scans per row; candidate select 137 -> 122 us, threshold select 206 ->
// 200 us at B = 2048 -- the latter is bound by its three passes over the 268 MB sample matrix, not by the scans).
__device__ __forceinline__ void suffix_sums_2048(uint32_t* hist, uint32_t* wave_tot) {     // hist[b] <- sum_{b' >= b} hist[b']
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t a = hist[2 * tid], b = hist[2 * tid + 1], pair = a + b;
  uint32_t v = pair;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_down(v, off, 64);
    if (lane + off < 64) v += o;
  }
  if (lane == 0) wave_tot[wave] = v;
  __syncthreads();
  uint32_t above = 0;
  for (int w = wave + 1; w < 16; ++w) above += wave_tot[w];
  const uint32_t s1 = b + (v - pair) + above;
  hist[2 * tid] = a + s1;
  hist[2 * tid + 1] = s1;
  __syncthreads();
}
__device__ __forceinline__ void prefix_sums_2048(uint32_t* hist, uint32_t* wave_tot) {     // hist[b] <- sum_{b' <= b} hist[b']
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t a = hist[2 * tid], b = hist[2 * tid + 1], pair = a + b;
  uint32_t v = pair;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(v, off, 64);
    if (lane >= off) v += o;
  }
  if (lane == 63) wave_tot[wave] = v;
  __syncthreads();
  uint32_t below = 0;
  for (int w = 0; w < wave; ++w) below += wave_tot[w];
  const uint32_t p0 = a + (v - pair) + below;
  hist[2 * tid] = p0;
  hist[2 * tid + 1] = p0 + b;
  __syncthreads();
}

// Exact top-k of one row (one 1024-thread workgroup per row): 3-pass radix select on the order-preserving
// key finds the k-th key, one more pass collects the winners, a bitonic sort orders them.
template <int MODE>
__global__ __launch_bounds__(1024) void topk_row_kernel(SelectArgs a, int k, float* __restrict__ topk_val,
                                                        int64_t* __restrict__ topk_idx) {
  __shared__ uint32_t hist[2048];
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t s_digit, s_krem, s_cnt_gt;
  __shared__ uint32_t okey[1024];
  __shared__ int32_t oidx[1024];
  __shared__ float qrow[MODE == SEL_RECOMPUTE ? 128 : 1];
  __shared__ float cval[MODE == SEL_CAND ? CAND_MAX : 1];
  __shared__ int32_t cidx[MODE == SEL_CAND ? CAND_MAX : 1];
  __shared__ int32_t s_total, s_over;
  const int tid = threadIdx.x;
  const int64_t r = blockIdx.x;
  int64_t n = a.n_cols;
  if (MODE == SEL_CAND) {
    // compact the row's private segments into LDS: exclusive prefix of the segment counts (n_seg <= 2048)
    if (tid == 0) s_over = 0;
    for (int b = tid; b < 2048; b += 1024) {
      int32_t c = b < a.n_seg ? a.seg_cnt[(size_t)r * a.n_seg + b] : 0;
      if (c > SEG) {
        s_over = 1;
        c = SEG;
      }
      hist[b] = (uint32_t)c;
    }
    __syncthreads();
    prefix_sums_2048(hist, wave_tot);               // inclusive prefix sums of the segment counts
    const int32_t n_ovf = a.ovf_cnt[r];
    if (tid == 0) s_total = (int32_t)hist[2047];
    __syncthreads();
    const int32_t seg_total = s_total;
    const int32_t total = seg_total + (n_ovf < OVF ? n_ovf : OVF);
    const bool bad = s_over != 0 || n_ovf > OVF || total < k || total > CAND_MAX;
    if (tid == 0) a.flags[r] = bad ? 1 : 0;
    if (bad) return;                       // recomputed exactly by the SEL_RECOMPUTE pass
    for (int b = tid; b < a.n_seg; b += 1024) {
      const int32_t end = (int32_t)hist[b], beg = b ? (int32_t)hist[b - 1] : 0;
      const size_t src = ((size_t)r * a.n_seg + b) * SEG;
      for (int32_t e = beg; e < end; ++e) {
        const float2 c = a.cand[src + (e - beg)];
        cval[e] = c.x;
        cidx[e] = __float_as_int(c.y);
      }
    }
    for (int32_t e = tid; e < total - seg_total; e += 1024) {
      cval[seg_total + e] = a.ovf_val[(size_t)r * OVF + e];
      cidx[seg_total + e] = a.ovf_idx[(size_t)r * OVF + e];
    }
    __syncthreads();
    n = total;
  }
  if (MODE == SEL_RECOMPUTE) {
    if (a.flags[r] == 0) return;
    for (int c = tid; c < a.dim; c += 1024) qrow[c] = a.query[(size_t)r * a.dim + c];
    __syncthreads();
  }
  const float* row = a.values + (size_t)r * a.ld;
  auto value_at = [&](int64_t i) -> float {
    if (MODE == SEL_RECOMPUTE) {
      const float* it = a.item_table + (size_t)(i + 1) * a.dim;
      float acc = 0.f;
      for (int c = 0; c < a.dim; ++c) acc = __fmaf_rn(it[c], qrow[c], acc);
      if (a.score_mode == RSA_SCORE_COS) acc = acc * a.item_aux[i] * a.query_aux[r];
      else if (a.score_mode == RSA_SCORE_EUC) acc = __fmaf_rn(2.f, acc, -a.item_aux[i]) - a.query_aux[r];
      return acc;
    }
    if (MODE == SEL_CAND) return cval[i];
    return row[i];
  };
  // CAND with at most 1024 candidates (the usual case: the threshold aims at 3k + 300): they all fit the sort below -- no radix
  // select, no winner collection (three histogram passes + six 2048-bin scans: half of this kernel's barriers)
  const bool direct = MODE == SEL_CAND && n <= 1024;
  if (direct) {
    okey[tid] = tid < n ? order_key(cval[tid]) : 0u;
    oidx[tid] = tid < n ? cidx[tid] : 0x7fffffff;
  }
  uint32_t prefix = 0, pmask = 0;
  uint32_t k_rem = (uint32_t)k;
  const int shifts[3] = {21, 10, 0}, nbits[3] = {11, 11, 10};
  for (int pass = 0; pass < (direct ? 0 : 3); ++pass) {
    const int shift = shifts[pass], nb = 1 << nbits[pass];
    for (int b = tid; b < 2048; b += 1024) hist[b] = 0;
    __syncthreads();
    for (int64_t i = tid; i < n; i += 1024) {
      const uint32_t key = order_key(value_at(i));
      if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & (nb - 1)], 1u);
    }
    __syncthreads();
    suffix_sums_2048(hist, wave_tot);               // hist[b] <- number of keys with digit >= b
    // the digit d with suffix(d) >= k_rem > suffix(d+1)
    for (int b = tid; b < nb; b += 1024) {
      const uint32_t ge = hist[b], gt = (b + 1 < 2048) ? hist[b + 1] : 0u;
      if (ge >= k_rem && gt < k_rem) {
        s_digit = (uint32_t)b;
        s_krem = k_rem - gt;
      }
    }
    __syncthreads();
    prefix |= s_digit << shift;
    pmask |= (uint32_t)(nb - 1) << shift;
    k_rem = s_krem;
    __syncthreads();
  }
  // prefix = key of the k-th largest; k_rem of the keys equal to it are needed
  if (MODE == SEL_THRESHOLD) {
    if (tid == 0) a.thr_out[r] = key_value(prefix);
    return;
  }
  // Winners: every key above the k-th key, plus the k_rem SMALLEST ids among the keys equal to it
  // (deterministic tie rule).  The tie slots [n_gt, k) start as sentinels and are filled by
  // "replace the current largest id if mine is smaller"; s_tie_max lets the (possibly very many)
  // remaining ties bail out with one LDS read.
  __shared__ int32_t s_tie_max;
  const uint32_t n_gt = (uint32_t)k - k_rem;
  if (!direct) {
    if (tid == 0) {
      s_cnt_gt = 0;
      s_tie_max = 0x7fffffff;
    }
    okey[tid] = (tid >= (int)n_gt && tid < k) ? prefix : 0u;
    oidx[tid] = 0x7fffffff;
  }
  __syncthreads();
  for (int64_t i = tid; i < (direct ? 0 : n); i += 1024) {
    const uint32_t key = order_key(value_at(i));
    const int32_t id = MODE == SEL_CAND ? cidx[i] : (int32_t)i + (MODE == SEL_DENSE ? a.idx_base : 1);   // item id
    if (key > prefix) {
      const uint32_t p = atomicAdd(&s_cnt_gt, 1u);
      okey[p] = key;
      oidx[p] = id;
    } else if (key == prefix) {
      while (id < *(volatile int32_t*)&s_tie_max) {
        uint32_t worst = n_gt;
        int32_t wv = *(volatile int32_t*)&oidx[n_gt];
        for (uint32_t t = n_gt + 1; t < (uint32_t)k; ++t) {
          const int32_t v = *(volatile int32_t*)&oidx[t];
          if (v > wv) {
            wv = v;
            worst = t;
          }
        }
        if (id >= wv) break;
        if (atomicCAS(&oidx[worst], wv, id) == wv) {
          int32_t mx = 0;
          for (uint32_t t = n_gt; t < (uint32_t)k; ++t) {
            const int32_t v = *(volatile int32_t*)&oidx[t];
            mx = v > mx ? v : mx;
          }
          atomicMin(&s_tie_max, mx);   // stale-high is harmless (extra work), never too low:
          break;                       // mx is the max of a superset-or-equal of the final kept ids
        }
      }
    }
  }
  __syncthreads();
  // bitonic sort, descending by key, ascending id among equal keys
  for (int size = 2; size <= 1024; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int partner = tid ^ stride;
      if (partner > tid) {
        const bool desc = (tid & size) == 0;
        const uint32_t ka = okey[tid], kb = okey[partner];
        const int32_t ia = oidx[tid], ib = oidx[partner];
        const bool a_first = ka > kb || (ka == kb && ia < ib);
        if (a_first != desc) {
          okey[tid] = kb; okey[partner] = ka;
          oidx[tid] = ib; oidx[partner] = ia;
        }
      }
      __syncthreads();
    }
  }
  if (tid < k) {
    topk_val[(size_t)r * k + tid] = key_value(okey[tid]);
    topk_idx[(size_t)r * k + tid] = (int64_t)oidx[tid];   // item id (baseretriever.py:385)
  }
}

// score[mask] = -inf where the item is in the user's history, then the k best survivors
// (baseretriever.py:386-392).  One wave per query; cand are sorted descending.
__global__ __launch_bounds__(256) void mask_history_kernel(const float* __restrict__ cand_val,
                                                           const int64_t* __restrict__ cand_idx, int kc,
                                                           const int64_t* __restrict__ hist, int hist_len, int k,
                                                           int64_t n_query, float* __restrict__ out_val,
                                                           int64_t* __restrict__ out_idx) {
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= n_query) return;
  const int lane = lane_id();
  const float* cv = cand_val + (size_t)q * kc;
  const int64_t* ci = cand_idx + (size_t)q * kc;
  const int64_t* hq = hist + (size_t)q * hist_len;
  int kept = 0, dropped = 0;
  for (int base = 0; base < kc; base += 64) {
    const int c = base + lane;
    bool in_hist = false, valid = c < kc;
    int64_t id = 0;
    float v = 0.f;
    if (valid) {
      id = ci[c];
      v = cv[c];
      for (int t = 0; t < hist_len; ++t) in_hist |= (hq[t] == id);
    }
    const unsigned long long keep_mask = __ballot(valid && !in_hist);
    const unsigned long long drop_mask = __ballot(valid && in_hist);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (valid && !in_hist) {
      const int p = kept + __popcll(keep_mask & below);
      if (p < k) {
        out_val[(size_t)q * k + p] = v;
        out_idx[(size_t)q * k + p] = id;
      }
    }
    // masked candidates fill the tail (as -inf) only if fewer than k survive
    if (valid && in_hist) {
      const int p = (kc - 1) - (dropped + __popcll(drop_mask & below));
      // position from the end among all kc slots; it lands in the output only when p < k
      if (p < k) {
        out_val[(size_t)q * k + p] = -INFINITY;
        out_idx[(size_t)q * k + p] = id;
      }
    }
    kept += __popcll(keep_mask);
    dropped += __popcll(drop_mask);
    if (kept >= k) break;       // (wave-uniform) the k best survivors are out; the rest of the list cannot matter
  }
}

}  // namespace rsa

using namespace rsa;

static int64_t fullscore_splits(int64_t n_query, int64_t n_positions, int64_t min_splits = 1) {
  const int64_t groups = (n_query + QB - 1) / QB;
  int64_t splits = (1024 + groups - 1) / groups;   // ~4 workgroups per CU in flight
  if (splits < min_splits) splits = min_splits;
  // XCD-aware launch: workgroups are dealt round-robin to the 8 XCDs by linear id = blockIdx.y * gridDim.x +
  // blockIdx.x.  With gridDim.x (= splits) a multiple of 8, all query groups of one item range (same blockIdx.x)
  // land on the same XCD and share its L2: the range is fetched from HBM once, not once per query group.
  if (splits >= 8) splits = (splits + 7) / 8 * 8;
  const int64_t max_splits = (n_positions + TI - 1) / TI;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  return splits;
}

constexpr int64_t SAMPLE_TILES_MAX = 1024;  // 32 768 sampled items
constexpr int64_t FILTER_MIN_ITEMS = 32768; // below this the dense radix select is cheaper

constexpr int64_t GROUP_TILES = 8;          // sampled tiles per threshold group (256 items)
struct TopkPlan {
  bool filter;
  int64_t sample_tiles, tile_stride, sample_items;
  int64_t groups;       // threshold groups per query (item ranges of the sample pass), <= 1024
  int32_t j;            // threshold = the j-th largest group maximum
  double expect;        // catalog items expected above it
  int64_t min_splits;   // item-range splits needed to keep the per-segment candidate count far below SEG
};

static TopkPlan plan_topk(int64_t n_items, int32_t k, bool scores_given) {
  TopkPlan pl{};
  const int64_t n_cols = n_items - 1;
  const int64_t tiles = (n_cols + TI - 1) / TI;
  pl.filter = !scores_given && k > 0 && n_cols >= FILTER_MIN_ITEMS;
  if (!pl.filter) return pl;
  pl.sample_tiles = tiles / 4 < SAMPLE_TILES_MAX ? tiles / 4 : SAMPLE_TILES_MAX;
  pl.tile_stride = tiles / pl.sample_tiles;
  pl.sample_items = pl.sample_tiles * TI;
  // The threshold is the j-th largest of the per-GROUP maxima of the sample (group = GROUP_TILES consecutive sampled tiles):
  // with tail mass p above a score, a group of g samples has its maximum above it with probability 1 - (1 - p)^g.  Aim at
  // E = 2k + 250 catalog items above the threshold (k = 100: 450, rank j = 14 of 128 group maxima; the relative spread of
  // the true count is ~ 1/sqrt(j): fewer than k survivors -- the exact-recompute path -- has probability ~3e-6 per query).
  pl.groups = (pl.sample_tiles + GROUP_TILES - 1) / GROUP_TILES;
  const double g_items = (double)(GROUP_TILES * TI);
#ifndef RSA_FS_EXPECT_MULT
#define RSA_FS_EXPECT_MULT 2.0     // in-process A/B at B = 2048, N = 1e6, k = 100 (ms, logsumexp + top-100): 3k + 300 -> 4.586,
#define RSA_FS_EXPECT_ADD 250.0    // 2k + 250 -> 4.528, 2k + 150 -> 4.505 (logsumexp alone 3.93)
#endif
  pl.expect = RSA_FS_EXPECT_MULT * k + RSA_FS_EXPECT_ADD;
  const double p_tail = pl.expect / (double)n_cols;
  double jj = (double)pl.groups * (1.0 - pow(1.0 - (p_tail < 1.0 ? p_tail : 1.0), g_items));
  if (jj < 12.0) {                       // too few groups above the target for a stable rank: take rank 12 and what it implies
    jj = 12.0;
    const double pg = jj / (double)pl.groups;
    pl.expect = pg < 1.0 ? (1.0 - pow(1.0 - pg, 1.0 / g_items)) * (double)n_cols : (double)n_cols;
  }
  pl.j = (int32_t)(jj + 0.5);
  if (pl.j > pl.groups / 2 || pl.expect > 0.6 * CAND_MAX || pl.expect < 2.0 * k) pl.filter = false;   // would crowd / starve the candidate lists
  // a (query, split, lane half) segment holds SEG = 32 candidates; with ~6 expected per segment an overflow
  // (-> the slow exact recompute of that row) has probability ~1e-14 for exchangeable scores.  Large k at
  // large B (few default splits) used to sit at ~16 per segment: a handful of overflowing rows per call, each
  // costing a full single-workgroup pass over the catalog (460 ms instead of 5 ms at B = 2048, k = 500).
  pl.min_splits = (int64_t)(pl.expect / (2.0 * 6.0)) + 1;
  return pl;
}

static inline int64_t align256(int64_t b) { return (b + 255) / 256 * 256; }

extern "C" int64_t rsa_fullscore_workspace_bytes(int64_t n_query, int64_t n_items, int32_t k) {
  if (n_query <= 0 || n_items <= 1) return 0;
  const TopkPlan pl = plan_topk(n_items, k, k <= 0);
  int64_t bytes = align256(n_query * fullscore_splits(n_query, n_items - 1, pl.filter ? pl.min_splits : 1) *
                           (int64_t)sizeof(float2));   // lse partials
  if (k > 0) {
    if (pl.filter) {
      bytes += align256(n_query * pl.groups * (int64_t)sizeof(float2));   // group maxima of the sample pass
      const int64_t n_seg = fullscore_splits(n_query, n_items - 1, pl.min_splits) * 2;
      bytes += 2 * align256(n_query * 4);                       // thresholds, flags
      bytes += align256(n_query * n_seg * 4);                   // per-segment counts
      bytes += align256(n_query * n_seg * (int64_t)SEG * 8);    // candidates {score, id}
      bytes += align256(n_query * 4) + 2 * align256(n_query * (int64_t)OVF * 4);   // overflow lists
    } else {
      bytes += align256(n_query * (n_items - 1) * (int64_t)sizeof(float));   // dense score rows
    }
  }
  return bytes + 256;
}

template <int D, int MODE>
static void launch_gemm(dim3 grid, hipStream_t s, const float* item_table, int64_t n_items, const float* query,
                        int64_t n_query, float* scores, int64_t score_ld, float2* lse_part, int splits, int64_t per,
                        int64_t tile_stride, int64_t n_positions, FilterArgs flt) {
  const bool L = lse_part != nullptr, S = scores != nullptr, F = flt.thr != nullptr;
#define RSA_GEMM(LL, SS, FF)                                                                                             \
  hipLaunchKernelGGL((fullscore_kernel<D, LL, SS, FF, MODE>), grid, dim3(256), 0, s, item_table, n_items, query, n_query, \
                     scores, score_ld, lse_part, splits, per, tile_stride, n_positions, flt)
  if (L && !S && !F) RSA_GEMM(true, false, false);        // training: logsumexp only
  else if (L && !S && F) RSA_GEMM(true, false, true);     // eval: logsumexp + candidate filter
  else if (!L && !S && F) RSA_GEMM(false, false, true);   // eval: candidate filter
  else if (!L && S && !F) RSA_GEMM(false, true, false);   // materialised scores / threshold sample
  else if (L && S && !F) RSA_GEMM(true, true, false);     // materialised scores + logsumexp
  else RSA_GEMM(true, true, true);
#undef RSA_GEMM
}

static void gemm_dispatch(int dim, int mode, dim3 grid, hipStream_t s, const float* item_table, int64_t n_items,
                          const float* query, int64_t n_query, float* scores, int64_t score_ld, float2* lse_part,
                          int splits, int64_t per, int64_t tile_stride, int64_t n_positions, FilterArgs flt) {
#define RSA_DIM(DD, MM) launch_gemm<DD, MM>(grid, s, item_table, n_items, query, n_query, scores, score_ld, lse_part, splits, per, tile_stride, n_positions, flt)
#define RSA_MODE(DD)                       \
  if (mode == RSA_SCORE_COS) RSA_DIM(DD, 1); \
  else if (mode == RSA_SCORE_EUC) RSA_DIM(DD, 2); \
  else RSA_DIM(DD, 0)
  switch (dim) {
    case 32: RSA_MODE(32); break;
    case 64: RSA_MODE(64); break;
    default: RSA_MODE(128); break;
  }
#undef RSA_MODE
#undef RSA_DIM
}

static int fullscore_impl(const float* item_table, int64_t n_items, int32_t dim, const float* query,
                          int64_t n_query, float* scores, float* lse, float* topk_val, int64_t* topk_idx, int32_t k,
                          int32_t score_mode, const float* item_aux, const float* query_aux,
                          void* workspace, int64_t workspace_bytes, rsa_stream_t stream);

extern "C" int rsa_fullscore(const rsa_fullscore_args* args, rsa_stream_t stream) {
  rsa_fullscore_args a;
  if (int rc = load_args(a, args, "rsa_fullscore")) return rc;
  return fullscore_impl(a.item_table, a.n_items, a.dim, a.query, a.n_query, a.scores, a.lse, a.topk_val, a.topk_idx, a.k,
                        a.score_mode, a.item_aux, a.query_aux, a.workspace, a.workspace_bytes, stream);
}

static int fullscore_impl(const float* item_table, int64_t n_items, int32_t dim, const float* query,
                          int64_t n_query, float* scores, float* lse, float* topk_val, int64_t* topk_idx, int32_t k,
                          int32_t score_mode, const float* item_aux, const float* query_aux,
                          void* workspace, int64_t workspace_bytes, rsa_stream_t stream) {
  RSA_CHECK_ARG(n_query >= 0 && n_items >= 2, "rsa_fullscore: need n_items >= 2");
  if (n_query == 0) return RSA_OK;
  RSA_CHECK_ARG(item_table && query, "rsa_fullscore: item_table/query is null");
  RSA_CHECK_ARG(scores || lse || k > 0, "rsa_fullscore: no output requested");
  RSA_CHECK_ARG(k >= 0 && k <= 1024 && (int64_t)k <= n_items - 1, "rsa_fullscore: k must be in [0, min(1024, n_items-1)]");
  RSA_CHECK_ARG(k == 0 || (topk_val && topk_idx), "rsa_fullscore: topk outputs are null");
  RSA_CHECK_ARG(score_mode >= RSA_SCORE_IP && score_mode <= RSA_SCORE_EUC, "rsa_fullscore: unknown score_mode %d", score_mode);
  RSA_CHECK_ARG(score_mode == RSA_SCORE_IP || (item_aux && query_aux && ((uintptr_t)item_aux & 15) == 0),
                "rsa_fullscore: cosine / Euclidean scores need item_aux (16-byte aligned) and query_aux");
  if (dim != 32 && dim != 64 && dim != 128) {
    rsa::set_error("rsa_fullscore: dim=%d: the MFMA full-score kernel is built for dim in {32, 64, 128}", dim);
    return RSA_ERR_UNSUPPORTED;
  }
  const int64_t need = rsa_fullscore_workspace_bytes(n_query, n_items, scores ? 0 : k);
  RSA_CHECK_ARG(workspace != nullptr && workspace_bytes >= need, "rsa_fullscore: workspace too small (%lld < %lld)",
                (long long)workspace_bytes, (long long)need);
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_cols = n_items - 1;
  const unsigned groups = (unsigned)((n_query + QB - 1) / QB);
  const TopkPlan pl = plan_topk(n_items, k, scores != nullptr);
  const int64_t splits = fullscore_splits(n_query, n_cols, pl.filter ? pl.min_splits : 1);
  const int64_t per = ((n_cols + splits - 1) / splits + TI - 1) / TI * TI;
  const int64_t splits_used = (n_cols + per - 1) / per;
  char* ws = reinterpret_cast<char*>(workspace);
  float2* part = reinterpret_cast<float2*>(ws);
  ws += align256(n_query * splits * (int64_t)sizeof(float2));
  float2* lp = lse ? part : nullptr;
  const FilterArgs no_filter{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, item_aux, query_aux, nullptr};

  if (pl.filter) {
    float2* gmax = reinterpret_cast<float2*>(ws);      ws += align256(n_query * pl.groups * (int64_t)sizeof(float2));
    const int64_t n_seg = splits_used * 2;
    float* thr = reinterpret_cast<float*>(ws);         ws += align256(n_query * 4);
    int32_t* flags = reinterpret_cast<int32_t*>(ws);   ws += align256(n_query * 4);
    int32_t* seg_cnt = reinterpret_cast<int32_t*>(ws); ws += align256(n_query * splits * 2 * 4);
    float2* cand = reinterpret_cast<float2*>(ws);      ws += align256(n_query * splits * 2 * (int64_t)SEG * 8);
    int32_t* ovf_cnt = reinterpret_cast<int32_t*>(ws);  ws += align256(n_query * 4);
    float* ovf_val = reinterpret_cast<float*>(ws);      ws += align256(n_query * (int64_t)OVF * 4);
    int32_t* ovf_idx = reinterpret_cast<int32_t*>(ws);
    // A. sample GEMM (the logsumexp instantiation: its per-range running maximum is the group maximum) + per-query threshold
    const int64_t sper = GROUP_TILES * TI;
    gemm_dispatch(dim, score_mode, dim3((unsigned)pl.groups, groups), s, item_table, n_items, query, n_query, nullptr,
                  pl.sample_items, gmax, (int)pl.groups, sper, pl.tile_stride, pl.sample_items, no_filter);
    RSA_CHECK_LAUNCH("rsa_fullscore(sample gemm)");
    hipLaunchKernelGGL(group_threshold_kernel, dim3((unsigned)((n_query + 3) / 4)), dim3(256), 0, s, gmax, n_query, (int)pl.groups,
                       (int)pl.groups, (int)pl.j, thr);
    RSA_CHECK_LAUNCH("rsa_fullscore(threshold)");
    // B. full GEMM with the filter epilogue (+ fused logsumexp)
    if (hipMemsetAsync(ovf_cnt, 0, (size_t)n_query * 4, s) != hipSuccess) {
      rsa::set_error("rsa_fullscore: memset failed");
      return RSA_ERR_HIP;
    }
    const FilterArgs flt{thr, seg_cnt, cand, nullptr, nullptr, ovf_cnt, ovf_val, ovf_idx, item_aux, query_aux, nullptr};
    gemm_dispatch(dim, score_mode, dim3((unsigned)splits_used, groups), s, item_table, n_items, query, n_query, nullptr, n_cols, lp,
                  (int)splits_used, per, 1, n_cols, flt);
    RSA_CHECK_LAUNCH("rsa_fullscore(gemm+filter)");
    // C. exact select over the candidates;  D. exact recompute of flagged rows
    SelectArgs ca{};
    ca.cand = cand; ca.seg_cnt = seg_cnt; ca.n_seg = (int32_t)n_seg; ca.flags = flags;
    ca.ovf_cnt = ovf_cnt; ca.ovf_val = ovf_val; ca.ovf_idx = ovf_idx;
    hipLaunchKernelGGL(topk_row_kernel<SEL_CAND>, dim3((unsigned)n_query), dim3(1024), 0, s, ca, (int)k, topk_val,
                       topk_idx);
    SelectArgs ra{};
    ra.flags = flags; ra.n_cols = n_cols; ra.item_table = item_table; ra.query = query; ra.dim = dim; ra.values = query;
    ra.score_mode = score_mode; ra.item_aux = item_aux; ra.query_aux = query_aux;
    hipLaunchKernelGGL(topk_row_kernel<SEL_RECOMPUTE>, dim3((unsigned)n_query), dim3(1024), 0, s, ra, (int)k, topk_val,
                       topk_idx);
    RSA_CHECK_LAUNCH("rsa_fullscore(select)");
  } else {
    float* score_rows = scores;
    if (k > 0 && scores == nullptr) score_rows = reinterpret_cast<float*>(ws);
    gemm_dispatch(dim, score_mode, dim3((unsigned)splits_used, groups), s, item_table, n_items, query, n_query, score_rows,
                  n_cols, lp, (int)splits_used, per, 1, n_cols, no_filter);
    RSA_CHECK_LAUNCH("rsa_fullscore(gemm)");
    if (k > 0) {
      SelectArgs da{};
      da.values = score_rows; da.ld = n_cols; da.n_cols = n_cols; da.idx_base = 1;
      hipLaunchKernelGGL(topk_row_kernel<SEL_DENSE>, dim3((unsigned)n_query), dim3(1024), 0, s, da, (int)k, topk_val,
                         topk_idx);
      RSA_CHECK_LAUNCH("rsa_fullscore(topk)");
    }
  }
  if (lse) {
    hipLaunchKernelGGL(lse_merge_kernel, dim3((unsigned)((n_query + 255) / 256)), dim3(256), 0, s, part, n_query,
                       (int)splits_used, lse);
    RSA_CHECK_LAUNCH("rsa_fullscore(lse)");
  }
  return RSA_OK;
}

extern "C" int rsa_fullscore_softmax(const float* item_table, int64_t n_items, int32_t dim, const float* query,
                                     int64_t n_query, const float* lse, const float* row_scale, float* probs,
                                     rsa_stream_t stream) {
  RSA_CHECK_ARG(n_query >= 0 && n_items >= 2, "rsa_fullscore_softmax: need n_items >= 2");
  if (n_query == 0) return RSA_OK;
  RSA_CHECK_ARG(item_table && query && lse && probs, "rsa_fullscore_softmax: null pointer");
  if (dim != 32 && dim != 64 && dim != 128) {
    rsa::set_error("rsa_fullscore_softmax: dim=%d: the MFMA full-score kernel is built for dim in {32, 64, 128}", dim);
    return RSA_ERR_UNSUPPORTED;
  }
  const int64_t n_cols = n_items - 1;
  const unsigned groups = (unsigned)((n_query + QB - 1) / QB);
  const int64_t splits = fullscore_splits(n_query, n_cols);
  const int64_t per = ((n_cols + splits - 1) / splits + TI - 1) / TI * TI;
  const int64_t splits_used = (n_cols + per - 1) / per;
  const FilterArgs ep{nullptr, nullptr, nullptr, lse, row_scale, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  gemm_dispatch(dim, RSA_SCORE_IP, dim3((unsigned)splits_used, groups), (hipStream_t)stream, item_table, n_items, query, n_query, probs,
                n_cols, nullptr, (int)splits_used, per, 1, n_cols, ep);
  RSA_CHECK_LAUNCH("rsa_fullscore_softmax");
  return RSA_OK;
}

// query_grad[i] = sum over the item-range partials, in range order (reproducible)
__global__ __launch_bounds__(256) void dq_reduce_kernel(const float4* __restrict__ part, int splits, int64_t n4,
                                                        float4* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 a = part[i];
  for (int s = 1; s < splits; ++s) {
    const float4 v = part[(size_t)s * n4 + i];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  out[i] = a;
}

static void softmax_plan(int64_t n_query, int64_t n_items, int64_t& per, int64_t& splits_used) {
  const int64_t n_cols = n_items - 1;
  const int64_t splits = fullscore_splits(n_query, n_cols);
  per = ((n_cols + splits - 1) / splits + TI - 1) / TI * TI;
  splits_used = (n_cols + per - 1) / per;
}

extern "C" int64_t rsa_fullscore_softmax_dq_workspace_bytes(int64_t n_query, int64_t n_items, int32_t dim) {
  if (n_query <= 0 || n_items <= 1 || dim <= 0) return 0;
  int64_t per, splits_used;
  softmax_plan(n_query, n_items, per, splits_used);
  return splits_used * n_query * (int64_t)dim * (int64_t)sizeof(float) + 256;
}

extern "C" int rsa_fullscore_softmax_dq(const float* item_table, int64_t n_items, int32_t dim, const float* query,
                                        int64_t n_query, const float* lse, const float* row_scale, float* probs,
                                        float* query_grad, void* workspace, int64_t workspace_bytes,
                                        rsa_stream_t stream) {
  RSA_CHECK_ARG(n_query >= 0 && n_items >= 2, "rsa_fullscore_softmax_dq: need n_items >= 2");
  if (n_query == 0) return RSA_OK;
  RSA_CHECK_ARG(item_table && query && lse && query_grad, "rsa_fullscore_softmax_dq: null pointer");
  if (dim != 32 && dim != 64 && dim != 128) {
    rsa::set_error("rsa_fullscore_softmax_dq: dim=%d: the MFMA full-score kernel is built for dim in {32, 64, 128}", dim);
    return RSA_ERR_UNSUPPORTED;
  }
  RSA_CHECK_ARG(workspace && workspace_bytes >= rsa_fullscore_softmax_dq_workspace_bytes(n_query, n_items, dim),
                "rsa_fullscore_softmax_dq: workspace too small (rsa_fullscore_softmax_dq_workspace_bytes)");
  const int64_t n_cols = n_items - 1;
  const unsigned groups = (unsigned)((n_query + QB - 1) / QB);
  int64_t per, splits_used;
  softmax_plan(n_query, n_items, per, splits_used);
  float* part = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
  const FilterArgs ep{nullptr, nullptr, nullptr, lse, row_scale, nullptr, nullptr, nullptr, nullptr, nullptr, part};
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)splits_used, groups);
  // probs == NULL: the softmax tile only feeds the second product (nothing of [B, N] is written)
#define RSA_DQ(DD)                                                                                                                  \
  if (probs != nullptr)                                                                                                             \
    hipLaunchKernelGGL((fullscore_kernel<DD, false, true, false, 0, true>), grid, dim3(256), 0, s, item_table, n_items, query,       \
                       n_query, probs, n_cols, (float2*)nullptr, (int)splits_used, per, (int64_t)1, n_cols, ep);                     \
  else                                                                                                                              \
    hipLaunchKernelGGL((fullscore_kernel<DD, false, true, false, 0, true, false>), grid, dim3(256), 0, s, item_table, n_items, query, \
                       n_query, probs, n_cols, (float2*)nullptr, (int)splits_used, per, (int64_t)1, n_cols, ep)
  switch (dim) {
    case 32: RSA_DQ(32); break;
    case 64: RSA_DQ(64); break;
    default: RSA_DQ(128); break;
  }
#undef RSA_DQ
  const int64_t n4 = n_query * dim / 4;
  hipLaunchKernelGGL(dq_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s,
                     reinterpret_cast<const float4*>(part), (int)splits_used, n4, reinterpret_cast<float4*>(query_grad));
  RSA_CHECK_LAUNCH("rsa_fullscore_softmax_dq");
  return RSA_OK;
}

// lse[q] and out[q, :] = softmax_q @ items from the flash forward's per-range records: ranges brought to the common
// reference M = max_s ref_s, summed in range order (reproducible)
__global__ __launch_bounds__(256) void flash_merge_kernel(const float2* __restrict__ part, const float* __restrict__ dq_part, int splits,
                                                          int64_t n_query, int dim, float* __restrict__ lse, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (query, 4 columns)
  const int d4 = dim / 4;
  if (i >= n_query * d4) return;
  const int64_t q = i / d4;
  float m = -INFINITY;
  for (int s = 0; s < splits; ++s) m = fmaxf(m, part[(size_t)q * splits + s].x);
  float den = 0.f;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < splits; ++s) {
    const float2 p = part[(size_t)q * splits + s];
    if (p.x == -INFINITY) continue;
    const float w = expf(p.x - m);
    den += p.y * w;
    const float4 v = reinterpret_cast<const float4*>(dq_part)[(size_t)s * n_query * d4 + i];
    a.x += v.x * w; a.y += v.y * w; a.z += v.z * w; a.w += v.w * w;
  }
  const float inv = 1.f / den;
  reinterpret_cast<float4*>(out)[i] = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
  if (i - q * d4 == 0) lse[q] = m + logf(den);
}

extern "C" int64_t rsa_fullscore_lse_grad_workspace_bytes(int64_t n_query, int64_t n_items, int32_t dim) {
  if (n_query <= 0 || n_items <= 1 || dim <= 0) return 0;
  int64_t per, splits_used;
  softmax_plan(n_query, n_items, per, splits_used);
  return splits_used * n_query * (int64_t)dim * (int64_t)sizeof(float) + align256(splits_used * n_query * (int64_t)sizeof(float2)) + 512;
}

extern "C" int rsa_fullscore_lse_grad(const float* item_table, int64_t n_items, int32_t dim, const float* query, int64_t n_query,
                                      float* lse, float* query_grad, void* workspace, int64_t workspace_bytes, rsa_stream_t stream) {
  RSA_CHECK_ARG(n_query >= 0 && n_items >= 2, "rsa_fullscore_lse_grad: need n_items >= 2");
  if (n_query == 0) return RSA_OK;
  RSA_CHECK_ARG(item_table && query && lse && query_grad, "rsa_fullscore_lse_grad: null pointer");
  if (dim != 32 && dim != 64 && dim != 128) {
    rsa::set_error("rsa_fullscore_lse_grad: dim=%d: the MFMA full-score kernel is built for dim in {32, 64, 128}", dim);
    return RSA_ERR_UNSUPPORTED;
  }
  RSA_CHECK_ARG(workspace && workspace_bytes >= rsa_fullscore_lse_grad_workspace_bytes(n_query, n_items, dim),
                "rsa_fullscore_lse_grad: workspace too small (rsa_fullscore_lse_grad_workspace_bytes)");
  const int64_t n_cols = n_items - 1;
  const unsigned groups = (unsigned)((n_query + QB - 1) / QB);
  int64_t per, splits_used;
  softmax_plan(n_query, n_items, per, splits_used);
  char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
  float2* part = reinterpret_cast<float2*>(ws);
  float* dq_part = reinterpret_cast<float*>(ws + align256(splits_used * n_query * (int64_t)sizeof(float2)));
  const FilterArgs ep{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, dq_part};
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)splits_used, groups);
#define RSA_FL(DD)                                                                                                                       \
  hipLaunchKernelGGL((fullscore_kernel<DD, false, true, false, 0, true, false, true>), grid, dim3(256), 0, s, item_table, n_items, query, \
                     n_query, (float*)nullptr, n_cols, part, (int)splits_used, per, (int64_t)1, n_cols, ep)
  switch (dim) {
    case 32: RSA_FL(32); break;
    case 64: RSA_FL(64); break;
    default: RSA_FL(128); break;
  }
#undef RSA_FL
  const int64_t n4 = n_query * dim / 4;
  hipLaunchKernelGGL(flash_merge_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, part, dq_part, (int)splits_used, n_query,
                     (int)dim, lse, query_grad);
  RSA_CHECK_LAUNCH("rsa_fullscore_lse_grad");
  return RSA_OK;
}

extern "C" int rsa_row_topk(const float* values, int64_t n_rows, int64_t n_cols, int32_t k, float* out_val,
                            int64_t* out_col, rsa_stream_t stream) {
  RSA_CHECK_ARG(n_rows >= 0 && n_cols >= 1 && k >= 1 && k <= 1024 && (int64_t)k <= n_cols,
                "rsa_row_topk: need 1 <= k <= min(1024, n_cols)");
  if (n_rows == 0) return RSA_OK;
  RSA_CHECK_ARG(values && out_val && out_col, "rsa_row_topk: null pointer");
  SelectArgs da{};
  da.values = values; da.ld = n_cols; da.n_cols = n_cols; da.idx_base = 0;
  hipLaunchKernelGGL(topk_row_kernel<SEL_DENSE>, dim3((unsigned)n_rows), dim3(1024), 0, (hipStream_t)stream, da, (int)k,
                     out_val, out_col);
  RSA_CHECK_LAUNCH("rsa_row_topk");
  return RSA_OK;
}

extern "C" int rsa_topk_mask_history(const float* cand_val, const int64_t* cand_idx, int32_t n_cand,
                                     const int64_t* user_hist, int32_t hist_len, int64_t n_query, int32_t k,
                                     float* out_val, int64_t* out_idx, rsa_stream_t stream) {
  RSA_CHECK_ARG(n_query >= 0 && n_cand >= 1 && k >= 1 && k <= n_cand && hist_len >= 0,
                "rsa_topk_mask_history: need 1 <= k <= n_cand");
  if (n_query == 0) return RSA_OK;
  RSA_CHECK_ARG(cand_val && cand_idx && out_val && out_idx && (user_hist || hist_len == 0),
                "rsa_topk_mask_history: null pointer");
  hipLaunchKernelGGL(mask_history_kernel, dim3((unsigned)((n_query + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     cand_val, cand_idx, (int)n_cand, user_hist, (int)hist_len, (int)k, n_query, out_val, out_idx);
  RSA_CHECK_LAUNCH("rsa_topk_mask_history");
  return RSA_OK;
}
