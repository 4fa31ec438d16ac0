// Atomics-free, deterministic row scatter-add:  target[id] += scale * sum_{e : id_e = id} d_e * query[qrow_e]
// over the B*(1+n) (positive and negative) elements of a step.
//
// embedding_dense_backward / index_add_ of the reference (recstudio/model/basemodel/recommender.py:636-639 via
// autograd) is an atomic scatter in ATen and in rsa_fused_backward's dense mode.  Float atomics from 8 XCDs to
// one table are performed at the memory side, a dword at a time: 537 M of them cost 2.5 ms per step at
// B = 65536, n = 64, d = 128.  Here the elements are sorted by item id first (the in-tree LSD radix sort of packed (id,
// element) pairs, rsa_radix.hpp, stable), every wave sums its 64 sorted elements run by run in element order, and every row is read-modified-
// written once with full-line accesses: no atomics, bit-reproducible, ~2x faster.  A run that crosses chunk
// boundaries (rare and short for item ids; the RULE when the key is a query index -- the owner side of the sharded
// backward sums ~1000 item rows per query) is not walked by one wave: each chunk leaves the partial sum of its
// leading / trailing open segment in the workspace and a second kernel adds a run's partials in chunk order.
// `target` may be a zeroed dense gradient (== the reference's weight.grad) or the weight table itself with
// scale = -lr (plain SGD applied in place).
#include "rsa_common.hpp"
#include "rsa_radix.hpp"
#include "rsa_internal.hpp"
#include <cstring>

// rows requested ahead by the apply pass.  4: 78 VGPRs at d = 128 = 6 waves/SIMD; 8: 110 = 4 waves.  In-process A/B
// (tools/exp_sorted_ab.py, us per launch, 4 vs 8): headline shape all elements 836.2 / 837.7, solo rows skipped 343.7 / 350.4;
// the sharded backward's scatters onto 4096 query rows 476.5 / 485.5 and onto item rows 839.9 / 843.0; 16 (179 VGPRs): 2-3 x slower
#ifndef RSA_SORTED_UNROLL
#define RSA_SORTED_UNROLL 4
#endif
// Target rows (and the lazy-Adam state rows) are touched ONCE per pass; the query rows are re-read once per element.  With
// streaming (nontemporal) accesses for the former the latter keep their L2 lines: the 2 MB of a 4096-query step's rows were
// otherwise re-fetched per ELEMENT under the 5 GB row stream (2.05 GB of 3.87 GB fetched).
// 0: never, 1: always, 2 (default): when the target table is larger than 512 MB
#ifndef RSA_SORTED_NT
#define RSA_SORTED_NT 2
#endif

namespace rsa {

// target[cur] (+ lazy Adam state) <- one read-modify-write of the row with the run's sum `acc`; trow / mrow_v / vrow_v
// are the row's current values (requested earlier, at the head of the run).
template <bool NT>
__device__ __forceinline__ float ld_row(const float* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  return *p;
}
template <bool NT>
__device__ __forceinline__ void st_row(float* p, float v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

template <int NDW, bool NT = false>
__device__ __forceinline__ void apply_run(float* __restrict__ target, const AdamArgs& adam, int32_t cur, float scale, int lane,
                                          const float (&acc)[NDW], const float (&trow)[NDW], const float (&mrow_v)[NDW],
                                          const float (&vrow_v)[NDW]) {
  constexpr int D = 64 * NDW;
  float* row = target + (size_t)cur * D;
  if (adam.exp_avg == nullptr) {
#pragma unroll
    for (int k = 0; k < NDW; ++k) st_row<NT>(row + k * 64 + lane, trow[k] + scale * acc[k]);
    return;
  }
  // lazy Adam on the touched row (torch.optim.SparseAdam's update, torch/optim/_functional.py sparse_adam):
  // m += (g - m)(1 - b1); v += (g^2 - v)(1 - b2); w -= step_size * m / (sqrt(v) + eps)
  float* mrow = adam.exp_avg + (size_t)cur * D;
  float* vrow = adam.exp_avg_sq + (size_t)cur * D;
#pragma unroll
  for (int k = 0; k < NDW; ++k) {
    const int c = k * 64 + lane;
    const float g = scale * acc[k];
    const float m0 = mrow_v[k], v0 = vrow_v[k];
    const float m1 = m0 + (g - m0) * adam.one_minus_beta1;
    const float v1 = v0 + (g * g - v0) * adam.one_minus_beta2;
    st_row<NT>(mrow + c, m1);
    st_row<NT>(vrow + c, v1);
    st_row<NT>(row + c, trow[k] - adam.step_size * (m1 / (sqrtf(v1) + adam.eps)));
  }
}

// Per-chunk record of the open segments: meta[4 c + 0] = key of the LEADING segment if it continues the previous chunk's
// run (else -1), [1] = 1 if that segment is the whole chunk, [2] = key of the TRAILING segment if its run goes on into
// the next chunk and started in this one (else -1).
enum { META_LEAD_KEY = 0, META_LEAD_FULL = 1, META_TRAIL_KEY = 2, META_STRIDE = 4 };

// One wave per chunk of 64 sorted elements.  Runs that begin and end inside the chunk are applied here; the open
// leading / trailing segments leave their partial sums in lead_part / trail_part [chunk, D] for sorted_finish_kernel.
#ifndef RSA_SORTED_MIN_WAVES
#define RSA_SORTED_MIN_WAVES 1
#endif
// What element e of the sorted pairs stands for: its query row and its coefficient d loss/d score.
struct DecStep {             // a step's [M, (1 +) n] element order: e = m * w + c
  const int64_t* query_index;
  const float* dpos;
  const float* dneg;
  int n, has_pos;
  __device__ __forceinline__ void operator()(int64_t e, int32_t& qrow, float& coef) const {
    const int w = n + has_pos;
    const int64_t m = (uint32_t)e / (uint32_t)w;
    const int c = (int)(e - m * w);
    qrow = (int32_t)(query_index ? query_index[m] : m);
    coef = (has_pos && c == 0) ? dpos[m] : dneg[m * (int64_t)n + (c - has_pos)];
  }
};
struct DecSegments {         // received exchange segments: e = slot, key = (query << 32 | row), d in slot order;
  const int64_t* keys;       // e >= slots: the positive of query e - slots (its coefficient sits behind the slots' in d)
  const float* d;
  int64_t slots;
  __device__ __forceinline__ void operator()(int64_t e, int32_t& qrow, float& coef) const {
    qrow = e >= slots ? (int32_t)(e - slots) : (int32_t)((keys[e] >> 32) & 0x7fffffffll);
    coef = d[e];
  }
};

template <int NDW, class DEC, bool NT>   // dwords per lane per row: D = 64 * NDW; NT: streaming accesses to the target (and state) rows
__global__ __launch_bounds__(256, RSA_SORTED_MIN_WAVES) void sorted_apply_kernel(const uint64_t* __restrict__ pairs,
                                                           int64_t total, const float* __restrict__ query, const DEC dec,
                                                           const float* __restrict__ upstream, int32_t pad_row,
                                                           int32_t drop_key, float* __restrict__ target, AdamArgs adam,
                                                           float* __restrict__ lead_part, float* __restrict__ trail_part,
                                                           int32_t* __restrict__ meta, int32_t key_base, int32_t elem_base,
                                                           int32_t chunk_elems) {
  constexpr int D = 64 * NDW;
  const int lane = lane_id();
  const int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  // chunk_elems = 64, or 16 for a small pass (sorted_chunk_elems): a wave walks its elements a few at a time, one round trip
  // per group -- the 65 536 user rows of the headline step as 1024 waves x 16 groups took 30 us of latency for 100 MB
  const int64_t begin = chunk * chunk_elems;
  if (begin >= total) return;
  const float scale = upstream ? upstream[0] : 1.f;
  // this lane's element of the chunk: key, query row, coefficient (all lanes in parallel)
  const int64_t i = begin + lane;
  const bool in = lane < chunk_elems && i < total;
  const uint64_t pr = in ? pairs[i] : 0ull;
  // key_base / elem_base (0 but for the user part of a merged step sort, sort_step_all): what the part's keys and element
  // numbers are offset by inside the common sort
  const int32_t key = in ? (int32_t)(rdx_key(pr) - (uint32_t)key_base) : -1;
  int32_t qrow = 0;
  float coef = 0.f;
  bool solo = false;                  // flagged by classify_solo_kernel: the row has been applied by the forward
  if (in && key != drop_key) {        // empty slots: nothing is read for them (their query index is -1)
    int32_t e = (int32_t)rdx_val(pr);
    solo = e < 0;
    e = (e & 0x7fffffff) - elem_base;
    if (!solo) dec((int64_t)e, qrow, coef);
  }
  const int cnt = (int)(total - begin < chunk_elems ? total - begin : chunk_elems);
  const int32_t prev = begin > 0 ? (int32_t)(rdx_key(pairs[begin - 1]) - (uint32_t)key_base) : -1;                     // key in front of the chunk
  const int32_t next = begin + cnt < total ? (int32_t)(rdx_key(pairs[begin + cnt]) - (uint32_t)key_base) : -1;         // key behind it
  float acc[NDW];
  float trow[NDW], mrow_v[NDW], vrow_v[NDW];     // the current run's target (and Adam state) row, requested at its head
#pragma unroll
  for (int k = 0; k < NDW; ++k) acc[k] = trow[k] = mrow_v[k] = vrow_v[k] = 0.f;
  int32_t cur = -1;           // id of the run being accumulated (-1: none)
  bool leading = false;       // accumulating the segment that continues the previous chunk's run
  int32_t lead_key = -1;
  auto close_segment = [&]() {        // a new run begins (or the chunk ends closed): what has been summed so far is complete
    if (leading) {
#pragma unroll
      for (int k = 0; k < NDW; ++k) lead_part[(size_t)chunk * D + k * 64 + lane] = acc[k];
      leading = false;
    } else if (cur >= 0) {
      apply_run<NDW, NT>(target, adam, cur, scale, lane, acc, trow, mrow_v, vrow_v);
    }
#pragma unroll
    for (int k = 0; k < NDW; ++k) acc[k] = 0.f;
  };
  // Only the elements with work are visited: not the dropped ones (empty slots), not the padding row's, not the ones a
  // forward has applied itself (solo) -- in a step whose solo rows are applied in the forward half of a chunk is such,
  // and a pass that walked them anyway (redirected to row 0) ran no faster than the full one.  None of them can sit
  // INSIDE a run that has work (a run is one key), so skipping them leaves every run's element order untouched.
  uint64_t work = __ballot(in && !solo && key != drop_key && key != pad_row);
  int last_t = -1;
  // The query rows AND the target rows of U elements are requested together before the (serial) run logic consumes
  // them.  Round 1 loaded one query row per iteration and read-modified-wrote the target row inside flush(): every
  // element waited for its own round trip and every run for an HBM read in the middle of the serial chain (VERDICT r1:
  // 0.9 ms of the 1.7 ms SGD step).  A target row belongs to exactly one run and a closed run to exactly one wave, so
  // reading it at the head of the run instead of at its end sees the same value.  Slots past the last element with
  // work request row 0, always readable.
  constexpr int U = RSA_SORTED_UNROLL;
  while (work) {
    int ts[U];
    int nu = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ts[u] = 63;
      if (work) {
        ts[u] = __ffsll((unsigned long long)work) - 1;
        work &= work - 1;
        nu = u + 1;
      }
    }
    float qv[U][NDW], tv[U][NDW], mv[U][NDW], vv[U][NDW];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int32_t qr = u < nu ? __builtin_amdgcn_readlane(qrow, ts[u]) : 0;
      const int32_t kr = u < nu ? __builtin_amdgcn_readlane(key, ts[u]) : 0;
      const float* qp = query + (size_t)qr * D;
      const float* tp = target + (size_t)kr * D;
#pragma unroll
      for (int k = 0; k < NDW; ++k) {
        qv[u][k] = qp[k * 64 + lane];
        tv[u][k] = ld_row<NT>(tp + k * 64 + lane);
      }
      if (adam.exp_avg != nullptr) {
        const float* mp = adam.exp_avg + (size_t)kr * D;
        const float* vp = adam.exp_avg_sq + (size_t)kr * D;
#pragma unroll
        for (int k = 0; k < NDW; ++k) {
          mv[u][k] = ld_row<NT>(mp + k * 64 + lane);
          vv[u][k] = ld_row<NT>(vp + k * 64 + lane);
        }
      } else {
#pragma unroll
        for (int k = 0; k < NDW; ++k) mv[u][k] = vv[u][k] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (u >= nu) break;
      const int t = ts[u];
      const int32_t kt = __builtin_amdgcn_readlane(key, t);
      const int32_t before = t == 0 ? prev : __builtin_amdgcn_readlane(key, t - 1);
      const bool head = kt != before;
      if (head) {
        close_segment();
        cur = kt;
#pragma unroll
        for (int k = 0; k < NDW; ++k) {
          trow[k] = tv[u][k];
          mrow_v[k] = mv[u][k];
          vrow_v[k] = vv[u][k];
        }
      } else if (t == 0) {            // the chunk opens inside a run that began in an earlier chunk
        leading = true;
        cur = lead_key = kt;
      }
      const float cf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(coef), t));   // bit pattern, not a value cast
#pragma unroll
      for (int k = 0; k < NDW; ++k) acc[k] = __fmaf_rn(cf, qv[u][k], acc[k]);
      last_t = t;
    }
  }
  // the chunk's last segment: still the leading one (the whole chunk lies inside one run), open towards the next chunk,
  // or closed
  const bool whole = leading;
  int32_t trail_key = -1;
  if (leading) {
    close_segment();                                  // -> lead_part
  } else if (cur >= 0 && last_t == cnt - 1 && next == cur) {
    trail_key = cur;                                  // the run goes on: sorted_finish_kernel owns its row
#pragma unroll
    for (int k = 0; k < NDW; ++k) trail_part[(size_t)chunk * D + k * 64 + lane] = acc[k];
  } else {
    close_segment();
  }
  if (lane == 0) {
    meta[chunk * META_STRIDE + META_LEAD_KEY] = lead_key;
    meta[chunk * META_STRIDE + META_LEAD_FULL] = whole ? 1 : 0;
    meta[chunk * META_STRIDE + META_TRAIL_KEY] = trail_key;
  }
}

// One wave per chunk whose trailing segment opened a run: adds the leading partials of the following chunks in chunk
// order (a chunk that lies wholly inside the run passes the walk on) and applies the row once.
template <int NDW>
__global__ __launch_bounds__(256) void sorted_finish_kernel(int64_t n_chunks, const float* __restrict__ upstream,
                                                            int32_t pad_row, float* __restrict__ target, AdamArgs adam,
                                                            const float* __restrict__ lead_part,
                                                            const float* __restrict__ trail_part,
                                                            const int32_t* __restrict__ meta) {
  constexpr int D = 64 * NDW;
  const int lane = lane_id();
  const int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (chunk >= n_chunks) return;
  const int32_t cur = meta[chunk * META_STRIDE + META_TRAIL_KEY];
  if (cur < 0 || cur == pad_row) return;
  const float scale = upstream ? upstream[0] : 1.f;
  float acc[NDW], trow[NDW], mrow_v[NDW], vrow_v[NDW];
#pragma unroll
  for (int k = 0; k < NDW; ++k) {
    acc[k] = trail_part[(size_t)chunk * D + k * 64 + lane];
    trow[k] = target[(size_t)cur * D + k * 64 + lane];
    mrow_v[k] = adam.exp_avg ? adam.exp_avg[(size_t)cur * D + k * 64 + lane] : 0.f;
    vrow_v[k] = adam.exp_avg ? adam.exp_avg_sq[(size_t)cur * D + k * 64 + lane] : 0.f;
  }
  for (int64_t c2 = chunk + 1; c2 < n_chunks; ++c2) {
    if (meta[c2 * META_STRIDE + META_LEAD_KEY] != cur) break;
#pragma unroll
    for (int k = 0; k < NDW; ++k) acc[k] += lead_part[(size_t)c2 * D + k * 64 + lane];
    if (!meta[c2 * META_STRIDE + META_LEAD_FULL]) break;
  }
  apply_run<NDW>(target, adam, cur, scale, lane, acc, trow, mrow_v, vrow_v);
}

// After the sort: an element whose key differs from both neighbours is the ONLY element of the step on its row ("solo").
// solo[e] (element order) <- 1 for such elements unless the row is the padding row / a dropped id, and bit 31 of the
// element number in `vals` is set so that the apply kernel leaves the row alone: a forward that updates solo rows
// itself (rsa_fused_args.solo_flags) has already applied them.
constexpr int32_t SOLO_BIT = (int32_t)0x80000000;
// MOSTLY_SOLO: the flags were preset to 1 and the elements that are NOT solo store a 0 -- the scattered byte stores are the cost
// of this pass (each its own 64-byte write request: 3 M of them 38 us), so the host picks the polarity with fewer of them.
template <bool MOSTLY_SOLO>
__global__ __launch_bounds__(256) void classify_solo_kernel(uint64_t* __restrict__ pairs, int64_t total, int32_t pad_row,
                                                            int32_t drop_key, uint8_t* __restrict__ solo) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const uint64_t pr = pairs[i];
    const int32_t k = (int32_t)rdx_key(pr);
    const int32_t before = i > 0 ? (int32_t)rdx_key(pairs[i - 1]) : -1, after = i + 1 < total ? (int32_t)rdx_key(pairs[i + 1]) : -1;
    const uint32_t e = rdx_val(pr);
    if (k != before && k != after && k != pad_row && k != drop_key) {
      if (!MOSTLY_SOLO) solo[e] = 1;                     // (the flags were zeroed: only the ones are stored)
      pairs[i] = pr | (uint64_t)(uint32_t)SOLO_BIT;      // (only the payload word changes: a neighbour reading the key races on nothing)
    } else if (MOSTLY_SOLO) {
      solo[e] = 0;
    }
  }
}

static inline int64_t align256s(int64_t b) { return (b + 255) / 256 * 256; }

// ---- the workspace of a sorted scatter over `max_total` elements: two packed-pair buffers, the radix sort's counters,
// per 64-element chunk two partial rows (sized for dim = 256) and the segment record
// Chunks of the apply pass: 64 sorted elements per wave, 16 for a pass over at most SMALL_TOTAL elements (a function of the
// pass's total ALONE, so that two passes over the same elements sum in the same order wherever they run).  The workspace holds
// per-chunk records for ANY total <= max_total (callers size it for their largest step): the bound below is monotone.
constexpr int64_t SORTED_SMALL_TOTAL = 1 << 18;
int sorted_chunk_elems(int64_t total) { return total <= SORTED_SMALL_TOTAL ? 16 : 64; }
static int64_t sorted_max_chunks(int64_t max_total) {
  const int64_t by64 = (max_total + 63) / 64;
  const int64_t small = max_total < SORTED_SMALL_TOTAL ? max_total : SORTED_SMALL_TOTAL;
  const int64_t by16 = (small + 15) / 16;
  return by64 > by16 ? by64 : by16;
}

int64_t sorted_workspace_bytes(int64_t max_total) {
  const int64_t chunks = sorted_max_chunks(max_total);
  return 2 * align256s(max_total * 8) + align256s(radix_temp_bytes(max_total)) + 2 * align256s(chunks * 256 * 4) +
         align256s(chunks * META_STRIDE * 4) + 256;
}

SortedLayout sorted_layout(void* workspace, int64_t max_total) {
  char* ws = reinterpret_cast<char*>(workspace);
  const int64_t seg = align256s(max_total * 8);
  const int64_t max_chunks = sorted_max_chunks(max_total);
  SortedLayout L;
  L.pairs_a = reinterpret_cast<uint64_t*>(ws);
  L.pairs_b = reinterpret_cast<uint64_t*>(ws + seg);
  L.temp = ws + 2 * seg;
  char* tail = ws + 2 * seg + align256s(radix_temp_bytes(max_total));
  L.lead_part = reinterpret_cast<float*>(tail);
  L.trail_part = reinterpret_cast<float*>(tail + align256s(max_chunks * 256 * 4));
  L.meta = reinterpret_cast<int32_t*>(tail + 2 * align256s(max_chunks * 256 * 4));
  return L;
}

int classify_solo(uint64_t* pairs, int64_t total, int64_t pad_row, int64_t drop_key, uint8_t* solo, hipStream_t s, const char* who) {
  const int32_t pad = (int32_t)(pad_row < 0 || pad_row >= (1ll << 31) ? -2 : pad_row);
  // `total` draws over `drop_key` rows (drop_key = the row count): under a uniform draw a row is alone with probability
  // exp(-total / rows), i.e. most elements are solo below total / rows = ln 2.  A skewed draw has fewer solo elements than that;
  // the guess only decides which of the two polarities stores fewer bytes.
  const bool mostly_solo = (double)total < 0.69 * (double)drop_key;
  if (hipMemsetAsync(solo, mostly_solo ? 1 : 0, (size_t)total, s) != hipSuccess) {
    rsa::set_error("%s: memset failed", who);
    return RSA_ERR_HIP;
  }
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (mostly_solo)
    hipLaunchKernelGGL(classify_solo_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, pairs, total, pad, (int32_t)drop_key, solo);
  else
    hipLaunchKernelGGL(classify_solo_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, pairs, total, pad, (int32_t)drop_key, solo);
  RSA_CHECK_LAUNCH(who);
  return RSA_OK;
}

// the apply + finish passes over sorted pairs
template <class DEC>
static int apply_sorted_pairs(const uint64_t* pairs, int64_t total, const float* query, int32_t dim, const DEC& dec,
                              const float* upstream, int64_t drop_key, int64_t pad_row, float* target, AdamArgs adam,
                              const SortedLayout& L, hipStream_t s, int64_t key_base = 0, int64_t elem_base = 0) {
  if (dim != 64 && dim != 128 && dim != 256) {
    rsa::set_error("rsa_rows_update_sorted: dim=%d: built for dim in {64, 128, 256}", dim);
    return RSA_ERR_UNSUPPORTED;
  }
  const int ce = sorted_chunk_elems(total);
  const unsigned chunks = (unsigned)((total + ce - 1) / ce);
  dim3 grid((chunks + 3) / 4), block(256);
  const int32_t pad = (int32_t)(pad_row < 0 || pad_row >= (1ll << 31) ? -2 : pad_row);
  const bool nt = RSA_SORTED_NT == 1 || (RSA_SORTED_NT == 2 && (size_t)drop_key * dim * sizeof(float) > (512ull << 20));      // (drop_key = the table's row count)
#define RSA_SORTED_LAUNCH(NDW)                                                                                                 \
  if (nt) hipLaunchKernelGGL((sorted_apply_kernel<NDW, DEC, true>), grid, block, 0, s, pairs, total, query, dec, upstream, pad, \
                             (int32_t)drop_key, target, adam, L.lead_part, L.trail_part, L.meta, (int32_t)key_base,            \
                             (int32_t)elem_base, (int32_t)ce);                                                                   \
  else hipLaunchKernelGGL((sorted_apply_kernel<NDW, DEC, false>), grid, block, 0, s, pairs, total, query, dec, upstream, pad,   \
                          (int32_t)drop_key, target, adam, L.lead_part, L.trail_part, L.meta, (int32_t)key_base, (int32_t)elem_base, (int32_t)ce); \
  hipLaunchKernelGGL(sorted_finish_kernel<NDW>, grid, block, 0, s, (int64_t)chunks, upstream, pad, target, adam, L.lead_part,   \
                     L.trail_part, L.meta)
  switch (dim) {
    case 64: RSA_SORTED_LAUNCH(1); break;
    case 128: RSA_SORTED_LAUNCH(2); break;
    default: RSA_SORTED_LAUNCH(4); break;
  }
#undef RSA_SORTED_LAUNCH
  RSA_CHECK_LAUNCH("rsa_rows_update_sorted(apply)");
  return RSA_OK;
}

int apply_sorted_segments(const uint64_t* pairs, int64_t total, int64_t slots, const float* query, int32_t dim,
                          const int64_t* keys, const float* d, const float* upstream, int64_t n_rows, int64_t pad_row,
                          float* target, const SortedLayout& L, hipStream_t s) {
  const AdamArgs none{nullptr, nullptr, 0.f, 0.f, 0.f, 0.f};
  const DecSegments dec{keys, d, slots};
  return apply_sorted_pairs(pairs, total, query, dim, dec, upstream, n_rows, pad_row, target, none, L, s);
}

// ---- the in-place SGD step (rsa_step.hip): ONE sort for the step's item elements AND its user elements (SrcStepAll)
static unsigned step_all_bits(int64_t n_items, int64_t n_users) { return radix_key_bits(n_items + n_users + 2); }

int64_t step_all_workspace_bytes(int64_t n_queries, int32_t num_neg) { return sorted_workspace_bytes(n_queries * (int64_t)(num_neg + 2)); }

int sort_step_all(const int64_t* pos_ids, int64_t* neg_ids, const int64_t* user_ids, int64_t n_queries, int32_t num_neg,
                  int64_t n_items, int64_t n_users, uint8_t* solo, void* workspace, int64_t workspace_bytes, const StepDraw* draw,
                  hipStream_t s, const char* who) {
  const int64_t t_items = n_queries * (int64_t)(num_neg + 1), total = t_items + n_queries;
  RSA_CHECK_ARG(total < (1ll << 31) && n_items + n_users + 2 < (1ll << 31), "%s: more than 2^31 elements / keys", who);
  const int64_t need = step_all_workspace_bytes(n_queries, num_neg);
  RSA_CHECK_ARG(workspace && workspace_bytes >= need, "%s: item_workspace too small (%lld < %lld = "
                "rsa_scatter_rows_sorted_workspace_bytes(n_queries, num_neg + 1, n_items): the user rows are sorted with the item rows)",
                who, (long long)workspace_bytes, (long long)need);
  const SortedLayout L = sorted_layout(workspace, total);
  const unsigned bits = step_all_bits(n_items, n_users);
  const SrcStepIds it{pos_ids, neg_ids, n_items, num_neg, num_neg + 1, 1};
  const StepDraw none{nullptr, 0, 0, PhiloxCall{0, 0, 1, 0}, 0, 0, nullptr, nullptr, nullptr};
  const SrcStepAll<false> rd{it, user_ids, t_items, n_users, (uint32_t)(n_items + 1), none};
  hipError_t err;
  if (draw != nullptr && draw->kind != 0) {
    const SrcStepAll<true> dw{it, user_ids, t_items, n_users, (uint32_t)(n_items + 1), *draw};
    err = radix_sort_pairs2(dw, rd, L.pairs_a, L.pairs_b, total, bits, L.temp, s);
  } else {
    err = radix_sort_pairs(rd, L.pairs_a, L.pairs_b, total, bits, L.temp, s);
  }
  if (err != hipSuccess) {
    rsa::set_error("%s: radix sort failed: %s", who, hipGetErrorString(hipGetLastError()));
    return RSA_ERR_HIP;
  }
  // the item part only: a user row is never updated by the forward
  return classify_solo(radix_result(L.pairs_a, L.pairs_b, bits), t_items, 0, n_items, solo, s, who);
}

// the shared item rows (users == false) or the user rows (users == true) of the step sorted by sort_step_all
int apply_step_all(bool users, const float* query, const int64_t* query_index, int32_t dim, int64_t n_queries, int32_t num_neg,
                   const float* dpos, const float* dneg, const float* upstream, int64_t n_items, int64_t n_users, float* target,
                   void* workspace, hipStream_t s) {
  const int64_t t_items = n_queries * (int64_t)(num_neg + 1), total = t_items + n_queries;
  const SortedLayout L = sorted_layout(workspace, total);
  const uint64_t* sorted = radix_result(L.pairs_a, L.pairs_b, step_all_bits(n_items, n_users));
  const AdamArgs none{nullptr, nullptr, 0.f, 0.f, 0.f, 0.f};
  if (!users) {
    const DecStep dec{query_index, dpos, dneg, (int)num_neg, 1};
    return apply_sorted_pairs(sorted, t_items, query, dim, dec, upstream, n_items, 0, target, none, L, s);
  }
  const DecStep dec{nullptr, nullptr, dneg, 1, 0};      // element m: query row m of `query`, coefficient dneg[m]
  return apply_sorted_pairs(sorted + t_items, n_queries, query, dim, dec, upstream, n_users, 0, target, none, L, s, n_items + 1, t_items);
}

}  // namespace rsa

using namespace rsa;

extern "C" int64_t rsa_scatter_rows_sorted_workspace_bytes(int64_t n_queries, int32_t num_neg, int64_t n_items) {
  if (n_queries <= 0 || num_neg < 0 || n_items < 1) return 0;
  return sorted_workspace_bytes(n_queries * (int64_t)(num_neg + 1));     // sized for the with-positives layout
}

// where the sorted pairs of a step live in its workspace (a function of the sizes alone: the sort and the apply are
// separate entry points)
static uint64_t* step_sorted(const SortedLayout& L, int64_t n_items) {
  return radix_result(L.pairs_a, L.pairs_b, radix_key_bits(n_items + 1));      // ids 0 .. n_items-1 and the drop key n_items
}

// (item id, element) pairs of a step, radix-sorted by id into the workspace; with `solo` also the classification pass
static int sort_elements_impl(const int64_t* pos_ids, const int64_t* neg_ids, int64_t n_queries, int32_t num_neg, int64_t n_items,
                              int64_t pad_row, uint8_t* solo, void* workspace, int64_t workspace_bytes, rsa_stream_t stream,
                              const char* who) {
  RSA_CHECK_ARG(n_queries >= 0 && num_neg >= 1 && n_items >= 1 && n_items < (1ll << 31), "%s: bad sizes", who);
  if (n_queries == 0) return RSA_OK;
  RSA_CHECK_ARG(neg_ids != nullptr, "%s: neg_ids is null", who);
  const int has_pos = pos_ids != nullptr ? 1 : 0;
  const int64_t total = n_queries * (int64_t)(num_neg + has_pos);
  RSA_CHECK_ARG(total < (1ll << 31), "%s: more than 2^31 elements", who);
  const int64_t need = rsa_scatter_rows_sorted_workspace_bytes(n_queries, num_neg, n_items);
  RSA_CHECK_ARG(workspace && workspace_bytes >= need, "%s: workspace too small (%lld < %lld)", who, (long long)workspace_bytes,
                (long long)need);
  hipStream_t s = (hipStream_t)stream;
  const SortedLayout L = sorted_layout(workspace, n_queries * (int64_t)(num_neg + 1));
  // pass 0 of the sort reads the id tensors themselves (no key-extraction launch, no id round trip)
  const SrcStepIds src{pos_ids, neg_ids, n_items, num_neg, num_neg + has_pos, has_pos};
  if (radix_sort_pairs(src, L.pairs_a, L.pairs_b, total, radix_key_bits(n_items + 1), L.temp, s) != hipSuccess) {
    rsa::set_error("%s: radix sort failed: %s", who, hipGetErrorString(hipGetLastError()));
    return RSA_ERR_HIP;
  }
  if (solo != nullptr) return classify_solo(step_sorted(L, n_items), total, pad_row, n_items, solo, s, who);
  return RSA_OK;
}

// the apply + finish passes over the sorted pairs in the workspace
static int apply_sorted_impl(const float* query, const int64_t* query_index, int64_t n_query_rows, int32_t dim, int has_pos,
                             int64_t n_queries, int32_t num_neg, const float* dpos, const float* dneg, const float* upstream,
                             int64_t n_items, int64_t pad_row, float* target, AdamArgs adam, void* workspace,
                             int64_t workspace_bytes, rsa_stream_t stream) {
  RSA_CHECK_ARG(n_queries >= 0 && num_neg >= 1 && n_items >= 1 && n_items < (1ll << 31), "rsa_rows_update_sorted: bad sizes");
  if (n_queries == 0) return RSA_OK;
  RSA_CHECK_ARG(query && dneg && target, "rsa_rows_update_sorted: null pointer");
  RSA_CHECK_ARG(!has_pos || dpos != nullptr, "rsa_rows_update_sorted: pos_ids without dpos");
  RSA_CHECK_ARG(query_index != nullptr || n_query_rows >= n_queries, "rsa_rows_update_sorted: query has fewer rows than n_queries");
  const int64_t total = n_queries * (int64_t)(num_neg + has_pos);
  const int64_t need = rsa_scatter_rows_sorted_workspace_bytes(n_queries, num_neg, n_items);
  RSA_CHECK_ARG(workspace && workspace_bytes >= need, "rsa_rows_update_sorted: workspace too small (%lld < %lld)",
                (long long)workspace_bytes, (long long)need);
  const SortedLayout L = sorted_layout(workspace, n_queries * (int64_t)(num_neg + 1));
  const DecStep dec{query_index, dpos, dneg, (int)num_neg, has_pos};
  return apply_sorted_pairs(step_sorted(L, n_items), total, query, dim, dec, upstream, n_items, pad_row, target, adam, L,
                            (hipStream_t)stream);
}

static int scatter_sorted_impl(const float* query, const int64_t* query_index, int64_t n_query_rows, int32_t dim,
                               const int64_t* pos_ids, const int64_t* neg_ids, int64_t n_queries, int32_t num_neg,
                               const float* dpos, const float* dneg, const float* upstream, int64_t n_items,
                               int64_t pad_row, float* target, AdamArgs adam, void* workspace, int64_t workspace_bytes,
                               rsa_stream_t stream) {
  RSA_CHECK_ARG(pos_ids == nullptr || dpos != nullptr, "rsa_rows_update_sorted: pos_ids without dpos");
  if (dim != 64 && dim != 128 && dim != 256) {
    rsa::set_error("rsa_rows_update_sorted: dim=%d: built for dim in {64, 128, 256}", dim);
    return RSA_ERR_UNSUPPORTED;
  }
  int rc = sort_elements_impl(pos_ids, neg_ids, n_queries, num_neg, n_items, pad_row, nullptr, workspace, workspace_bytes, stream,
                              "rsa_rows_update_sorted");
  if (rc != RSA_OK) return rc;
  return apply_sorted_impl(query, query_index, n_query_rows, dim, pos_ids != nullptr, n_queries, num_neg, dpos, dneg, upstream,
                           n_items, pad_row, target, adam, workspace, workspace_bytes, stream);
}

static AdamArgs adam_of(const rsa_rows_update_args& a) {
  if (a.exp_avg == nullptr) return AdamArgs{nullptr, nullptr, 0.f, 0.f, 0.f, 0.f};
  const double bc1 = 1.0 - pow((double)a.beta1, (double)a.step), bc2 = 1.0 - pow((double)a.beta2, (double)a.step);
  return AdamArgs{a.exp_avg, a.exp_avg_sq, 1.f - a.beta1, 1.f - a.beta2, a.eps, (float)((double)a.lr * sqrt(bc2) / bc1)};
}

static int check_adam(const rsa_rows_update_args& a, const char* who) {
  if (a.exp_avg == nullptr && a.exp_avg_sq == nullptr) return RSA_OK;
  RSA_CHECK_ARG(a.exp_avg && a.exp_avg_sq && a.step >= 1 && a.beta1 >= 0.f && a.beta1 < 1.f && a.beta2 >= 0.f && a.beta2 < 1.f,
                "%s: bad optimizer state / hyper-parameters", who);
  return RSA_OK;
}

extern "C" int rsa_sort_step_elements(const rsa_rows_update_args* args, rsa_stream_t stream) {
  rsa_rows_update_args a;
  if (int rc = load_args(a, args, "rsa_sort_step_elements")) return rc;
  return sort_elements_impl(a.pos_ids, a.neg_ids, a.n_queries, a.num_neg, a.n_items, a.pad_row, a.solo, a.workspace,
                            a.workspace_bytes, stream, "rsa_sort_step_elements");
}

extern "C" int rsa_rows_update_presorted(const rsa_rows_update_args* args, rsa_stream_t stream) {
  rsa_rows_update_args a;
  if (int rc = load_args(a, args, "rsa_rows_update_presorted")) return rc;
  if (int rc = check_adam(a, "rsa_rows_update_presorted")) return rc;
  return apply_sorted_impl(a.query, a.query_index, a.n_query_rows, a.dim, a.has_pos != 0, a.n_queries, a.num_neg, a.dpos, a.dneg,
                           a.upstream, a.n_items, a.pad_row, a.target, adam_of(a), a.workspace, a.workspace_bytes, stream);
}

extern "C" int rsa_rows_update_sorted(const rsa_rows_update_args* args, rsa_stream_t stream) {
  rsa_rows_update_args a;
  if (int rc = load_args(a, args, "rsa_rows_update_sorted")) return rc;
  if (int rc = check_adam(a, "rsa_rows_update_sorted")) return rc;
  return scatter_sorted_impl(a.query, a.query_index, a.n_query_rows, a.dim, a.pos_ids, a.neg_ids, a.n_queries, a.num_neg, a.dpos,
                             a.dneg, a.upstream, a.n_items, a.pad_row, a.target, adam_of(a), a.workspace, a.workspace_bytes, stream);
}
