// Atomics-free, deterministic row scatter-add:  target[id] += scale * sum_{e : id_e = id} d_e * query[qrow_e]
// over the B*(1+n) (positive and negative) elements of a step.
//
// embedding_dense_backward / index_add_ of the reference (recstudio/model/basemodel/recommender.py:636-639 via
// autograd) is an atomic scatter in ATen and in rsa_fused_backward's dense mode.  Float atomics from 8 XCDs to
// one table are performed at the memory side, a dword at a time: 537 M of them cost 2.5 ms per step at
// B = 65536, n = 64, d = 128.  Here the elements are sorted by item id first (rocPRIM radix sort of (id, element)
// pairs, stable), every wave sums its 64 sorted elements run by run in element order, and every row is read-modified-
// written once with full-line accesses: no atomics, bit-reproducible, ~2x faster.  A run that crosses chunk
// boundaries (rare and short for item ids; the RULE when the key is a query index -- the owner side of the sharded
// backward sums ~1000 item rows per query) is not walked by one wave: each chunk leaves the partial sum of its
// leading / trailing open segment in the workspace and a second kernel adds a run's partials in chunk order.
// `target` may be a zeroed dense gradient (== the reference's weight.grad) or the weight table itself with
// scale = -lr (plain SGD applied in place).
#include "rsa_common.hpp"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

// rows requested ahead by the apply pass.  4: 78 VGPRs at d = 128 = 6 waves/SIMD; 8: 110 = 4 waves.  In-process A/B
// (tools/exp_sorted_ab.py, us per launch, 4 vs 8): headline shape all elements 836.2 / 837.7, solo rows skipped 343.7 / 350.4;
// the sharded backward's scatters onto 4096 query rows 476.5 / 485.5 and onto item rows 839.9 / 843.0; 16 (179 VGPRs): 2-3 x slower
#ifndef RSA_SORTED_UNROLL
#define RSA_SORTED_UNROLL 4
#endif

namespace rsa {

__global__ __launch_bounds__(256) void sorted_keys_kernel(const int64_t* __restrict__ pos_ids,
                                                          const int64_t* __restrict__ neg_ids, int64_t n_queries, int n,
                                                          int64_t n_items, int32_t* __restrict__ keys,
                                                          int32_t* __restrict__ vals) {
  const int w = pos_ids ? n + 1 : n;          // elements per query: positive slot only when positives are given
  const int off = pos_ids ? 1 : 0;
  const int64_t total = n_queries * (int64_t)w;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t m = e / w;
    const int c = (int)(e - m * w);
    int64_t id = (off && c == 0) ? pos_ids[m] : neg_ids[m * (int64_t)n + (c - off)];
    // a negative id is an empty slot: key n_items sorts behind every real row and its run is skipped
    id = id < 0 ? n_items : (id >= n_items ? n_items - 1 : id);
    keys[e] = (int32_t)id;
    vals[e] = (int32_t)e;
  }
}

struct AdamArgs {            // exp_avg == nullptr: plain accumulate (target[id] += scale * sum)
  float* exp_avg;
  float* exp_avg_sq;
  float one_minus_beta1, one_minus_beta2, eps, step_size;
};

// target[cur] (+ lazy Adam state) <- one read-modify-write of the row with the run's sum `acc`; trow / mrow_v / vrow_v
// are the row's current values (requested earlier, at the head of the run).
template <int NDW>
__device__ __forceinline__ void apply_run(float* __restrict__ target, const AdamArgs& adam, int32_t cur, float scale, int lane,
                                          const float (&acc)[NDW], const float (&trow)[NDW], const float (&mrow_v)[NDW],
                                          const float (&vrow_v)[NDW]) {
  constexpr int D = 64 * NDW;
  float* row = target + (size_t)cur * D;
  if (adam.exp_avg == nullptr) {
#pragma unroll
    for (int k = 0; k < NDW; ++k) row[k * 64 + lane] = trow[k] + scale * acc[k];
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
    mrow[c] = m1;
    vrow[c] = v1;
    row[c] = trow[k] - adam.step_size * (m1 / (sqrtf(v1) + adam.eps));
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
template <int NDW>   // dwords per lane per row: D = 64 * NDW
__global__ __launch_bounds__(256, RSA_SORTED_MIN_WAVES) void sorted_apply_kernel(const int32_t* __restrict__ keys, const int32_t* __restrict__ vals,
                                                           int64_t total, const float* __restrict__ query,
                                                           const int64_t* __restrict__ query_index, int n, int has_pos,
                                                           const float* __restrict__ dpos, const float* __restrict__ dneg,
                                                           const float* __restrict__ upstream, int32_t pad_row,
                                                           int32_t drop_key, float* __restrict__ target, AdamArgs adam,
                                                           float* __restrict__ lead_part, float* __restrict__ trail_part,
                                                           int32_t* __restrict__ meta) {
  constexpr int D = 64 * NDW;
  const int lane = lane_id();
  const int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t begin = chunk * 64;
  if (begin >= total) return;
  const float scale = upstream ? upstream[0] : 1.f;
  const int w = n + has_pos;
  // this lane's element of the chunk: key, query row, coefficient (all lanes in parallel)
  const int64_t i = begin + lane;
  const bool in = i < total;
  const int32_t key = in ? keys[i] : -1;
  int32_t qrow = 0;
  float coef = 0.f;
  bool solo = false;                  // flagged by classify_solo_kernel: the row has been applied by the forward
  if (in && key != drop_key) {        // empty slots: nothing is read for them (their query index is -1)
    int64_t e = vals[i];
    solo = e < 0;
    e &= 0x7fffffff;
    const int64_t m = e / w;
    const int c = (int)(e - m * w);
    if (!solo) {
      qrow = (int32_t)(query_index ? query_index[m] : m);
      coef = (has_pos && c == 0) ? dpos[m] : dneg[m * (int64_t)n + (c - has_pos)];
    }
  }
  const int cnt = (int)(total - begin < 64 ? total - begin : 64);
  const int32_t prev = begin > 0 ? keys[begin - 1] : -1;                     // key in front of the chunk
  const int32_t next = begin + cnt < total ? keys[begin + cnt] : -1;         // key behind it
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
      apply_run<NDW>(target, adam, cur, scale, lane, acc, trow, mrow_v, vrow_v);
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
        tv[u][k] = tp[k * 64 + lane];
      }
      if (adam.exp_avg != nullptr) {
        const float* mp = adam.exp_avg + (size_t)kr * D;
        const float* vp = adam.exp_avg_sq + (size_t)kr * D;
#pragma unroll
        for (int k = 0; k < NDW; ++k) {
          mv[u][k] = mp[k * 64 + lane];
          vv[u][k] = vp[k * 64 + lane];
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
__global__ __launch_bounds__(256) void classify_solo_kernel(const int32_t* __restrict__ keys, int32_t* __restrict__ vals,
                                                            int64_t total, int32_t pad_row, int32_t drop_key,
                                                            uint8_t* __restrict__ solo) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int32_t k = keys[i];
    const int32_t before = i > 0 ? keys[i - 1] : -1, after = i + 1 < total ? keys[i + 1] : -1;
    if (k != before && k != after && k != pad_row && k != drop_key) {       // (the flags were zeroed: only the ones are stored --
      const int32_t e = vals[i];                                            //  a scattered byte store each)
      solo[e] = 1;
      vals[i] = e | SOLO_BIT;
    }
  }
}

static inline int64_t align256s(int64_t b) { return (b + 255) / 256 * 256; }

static size_t sort_temp_bytes(int64_t total, unsigned end_bit) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (const int32_t*)nullptr,
                                  (int32_t*)nullptr, (size_t)total, 0u, end_bit, (hipStream_t)0);
  return bytes;
}

static unsigned key_bits(int64_t n_items) {
  unsigned b = 1;
  while (b < 31 && (1ll << b) < n_items) ++b;
  return b;
}

}  // namespace rsa

using namespace rsa;

extern "C" int64_t rsa_scatter_rows_sorted_workspace_bytes(int64_t n_queries, int32_t num_neg, int64_t n_items) {
  if (n_queries <= 0 || num_neg < 0 || n_items < 1) return 0;
  const int64_t total = n_queries * (int64_t)(num_neg + 1);     // sized for the with-positives layout
  const int64_t chunks = (total + 63) / 64;
  // + per chunk: two partial rows (sized for dim = 256) and the segment record
  // + per chunk: two partial rows (sized for dim = 256) and the segment record
  return 4 * align256s(total * 4) + align256s((int64_t)sort_temp_bytes(total, key_bits(n_items + 1))) +
         2 * align256s(chunks * 256 * 4) + align256s(chunks * META_STRIDE * 4) + 256;
}

struct SortedLayout {        // the caller's workspace (rsa_scatter_rows_sorted_workspace_bytes)
  int32_t *k_in, *v_in, *k_out, *v_out;
  void* temp;
  float *lead_part, *trail_part;
  int32_t* meta;
};


static SortedLayout sorted_layout(void* workspace, int64_t n_queries, int32_t num_neg, int64_t n_items) {
  char* ws = reinterpret_cast<char*>(workspace);
  const int64_t max_total = n_queries * (int64_t)(num_neg + 1);
  const int64_t seg = align256s(max_total * 4);
  const int64_t max_chunks = (max_total + 63) / 64;
  SortedLayout L;
  L.k_in = reinterpret_cast<int32_t*>(ws);
  L.v_in = reinterpret_cast<int32_t*>(ws + seg);
  L.k_out = reinterpret_cast<int32_t*>(ws + 2 * seg);
  L.v_out = reinterpret_cast<int32_t*>(ws + 3 * seg);
  L.temp = ws + 4 * seg;
  char* tail = ws + 4 * seg + align256s((int64_t)sort_temp_bytes(max_total, key_bits(n_items + 1)));
  L.lead_part = reinterpret_cast<float*>(tail);
  L.trail_part = reinterpret_cast<float*>(tail + align256s(max_chunks * 256 * 4));
  L.meta = reinterpret_cast<int32_t*>(tail + 2 * align256s(max_chunks * 256 * 4));
  return L;
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
  const SortedLayout L = sorted_layout(workspace, n_queries, num_neg, n_items);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(sorted_keys_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pos_ids, neg_ids, n_queries, (int)num_neg,
                     n_items, L.k_in, L.v_in);
  RSA_CHECK_LAUNCH(who);
  const unsigned bits = key_bits(n_items + 1);      // ids 0 .. n_items-1 and the drop key n_items
  size_t temp_bytes = sort_temp_bytes(total, bits);
  if (rocprim::radix_sort_pairs(L.temp, temp_bytes, L.k_in, L.k_out, L.v_in, L.v_out, (size_t)total, 0u, bits, s) != hipSuccess) {
    rsa::set_error("%s: radix sort failed: %s", who, hipGetErrorString(hipGetLastError()));
    return RSA_ERR_HIP;
  }
  if (solo != nullptr) {
    const int32_t pad = (int32_t)(pad_row < 0 || pad_row >= (1ll << 31) ? -2 : pad_row);
    if (hipMemsetAsync(solo, 0, (size_t)total, s) != hipSuccess) {
      rsa::set_error("%s: memset failed", who);
      return RSA_ERR_HIP;
    }
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(classify_solo_kernel, dim3((unsigned)blocks), dim3(256), 0, s, L.k_out, L.v_out, total, pad,
                       (int32_t)n_items, solo);
    RSA_CHECK_LAUNCH(who);
  }
  return RSA_OK;
}

// the apply + finish passes over the sorted pairs in the workspace
static int apply_sorted_impl(const float* query, const int64_t* query_index, int64_t n_query_rows, int32_t dim, int has_pos,
                             int64_t n_queries, int32_t num_neg, const float* dpos, const float* dneg, const float* upstream,
                             int64_t n_items, int64_t pad_row, float* target, AdamArgs adam, void* workspace,
                             int64_t workspace_bytes, rsa_stream_t stream) {
  RSA_CHECK_ARG(n_queries >= 0 && num_neg >= 1 && n_items >= 1 && n_items < (1ll << 31), "rsa_scatter_rows_sorted: bad sizes");
  if (n_queries == 0) return RSA_OK;
  RSA_CHECK_ARG(query && dneg && target, "rsa_scatter_rows_sorted: null pointer");
  RSA_CHECK_ARG(!has_pos || dpos != nullptr, "rsa_scatter_rows_sorted: pos_ids without dpos");
  RSA_CHECK_ARG(query_index != nullptr || n_query_rows >= n_queries, "rsa_scatter_rows_sorted: query has fewer rows than n_queries");
  if (dim != 64 && dim != 128 && dim != 256) {
    rsa::set_error("rsa_scatter_rows_sorted: dim=%d: built for dim in {64, 128, 256}", dim);
    return RSA_ERR_UNSUPPORTED;
  }
  const int64_t total = n_queries * (int64_t)(num_neg + has_pos);
  const int64_t need = rsa_scatter_rows_sorted_workspace_bytes(n_queries, num_neg, n_items);
  RSA_CHECK_ARG(workspace && workspace_bytes >= need, "rsa_scatter_rows_sorted: workspace too small (%lld < %lld)",
                (long long)workspace_bytes, (long long)need);
  hipStream_t s = (hipStream_t)stream;
  const SortedLayout L = sorted_layout(workspace, n_queries, num_neg, n_items);
  const unsigned chunks = (unsigned)((total + 63) / 64);
  dim3 grid((chunks + 3) / 4), block(256);
  const int32_t pad = (int32_t)(pad_row < 0 || pad_row >= (1ll << 31) ? -2 : pad_row);
#define RSA_SORTED_LAUNCH(NDW)                                                                                              \
  hipLaunchKernelGGL(sorted_apply_kernel<NDW>, grid, block, 0, s, L.k_out, L.v_out, total, query, query_index, (int)num_neg,  \
                     has_pos, dpos, dneg, upstream, pad, (int32_t)n_items, target, adam, L.lead_part, L.trail_part, L.meta); \
  hipLaunchKernelGGL(sorted_finish_kernel<NDW>, grid, block, 0, s, (int64_t)chunks, upstream, pad, target, adam,              \
                     L.lead_part, L.trail_part, L.meta)
  switch (dim) {
    case 64: RSA_SORTED_LAUNCH(1); break;
    case 128: RSA_SORTED_LAUNCH(2); break;
    default: RSA_SORTED_LAUNCH(4); break;
  }
#undef RSA_SORTED_LAUNCH
  RSA_CHECK_LAUNCH("rsa_scatter_rows_sorted(apply)");
  return RSA_OK;
}

static int scatter_sorted_impl(const float* query, const int64_t* query_index, int64_t n_query_rows, int32_t dim,
                               const int64_t* pos_ids, const int64_t* neg_ids, int64_t n_queries, int32_t num_neg,
                               const float* dpos, const float* dneg, const float* upstream, int64_t n_items,
                               int64_t pad_row, float* target, AdamArgs adam, void* workspace, int64_t workspace_bytes,
                               rsa_stream_t stream) {
  RSA_CHECK_ARG(pos_ids == nullptr || dpos != nullptr, "rsa_scatter_rows_sorted: pos_ids without dpos");
  if (dim != 64 && dim != 128 && dim != 256) {
    rsa::set_error("rsa_scatter_rows_sorted: dim=%d: built for dim in {64, 128, 256}", dim);
    return RSA_ERR_UNSUPPORTED;
  }
  int rc = sort_elements_impl(pos_ids, neg_ids, n_queries, num_neg, n_items, pad_row, nullptr, workspace, workspace_bytes, stream,
                              "rsa_scatter_rows_sorted");
  if (rc != RSA_OK) return rc;
  return apply_sorted_impl(query, query_index, n_query_rows, dim, pos_ids != nullptr, n_queries, num_neg, dpos, dneg, upstream,
                           n_items, pad_row, target, adam, workspace, workspace_bytes, stream);
}

extern "C" int rsa_sort_step_elements(const int64_t* pos_ids, const int64_t* neg_ids, int64_t n_queries, int32_t num_neg,
                                      int64_t n_items, int64_t pad_row, uint8_t* solo, void* workspace, int64_t workspace_bytes,
                                      rsa_stream_t stream) {
  return sort_elements_impl(pos_ids, neg_ids, n_queries, num_neg, n_items, pad_row, solo, workspace, workspace_bytes, stream,
                            "rsa_sort_step_elements");
}

extern "C" int rsa_scatter_rows_presorted(const float* query, const int64_t* query_index, int64_t n_query_rows, int32_t dim,
                                          int32_t has_pos, int64_t n_queries, int32_t num_neg, const float* dpos, const float* dneg,
                                          const float* upstream, int64_t n_items, int64_t pad_row, float* target, void* workspace,
                                          int64_t workspace_bytes, rsa_stream_t stream) {
  const AdamArgs none{nullptr, nullptr, 0.f, 0.f, 0.f, 0.f};
  return apply_sorted_impl(query, query_index, n_query_rows, dim, has_pos != 0, n_queries, num_neg, dpos, dneg, upstream, n_items,
                           pad_row, target, none, workspace, workspace_bytes, stream);
}

extern "C" int rsa_scatter_rows_sorted(const float* query, const int64_t* query_index, int64_t n_query_rows, int32_t dim,
                                       const int64_t* pos_ids, const int64_t* neg_ids, int64_t n_queries,
                                       int32_t num_neg, const float* dpos, const float* dneg, const float* upstream,
                                       int64_t n_items, int64_t pad_row, float* target, void* workspace,
                                       int64_t workspace_bytes, rsa_stream_t stream) {
  const AdamArgs none{nullptr, nullptr, 0.f, 0.f, 0.f, 0.f};
  return scatter_sorted_impl(query, query_index, n_query_rows, dim, pos_ids, neg_ids, n_queries, num_neg, dpos, dneg,
                             upstream, n_items, pad_row, target, none, workspace, workspace_bytes, stream);
}

extern "C" int rsa_adam_rows_presorted(const float* query, const int64_t* query_index, int64_t n_query_rows, int32_t dim,
                                       int32_t has_pos, int64_t n_queries, int32_t num_neg, const float* dpos, const float* dneg,
                                       const float* upstream, int64_t n_items, int64_t pad_row, float* weight, float* exp_avg,
                                       float* exp_avg_sq, float lr, float beta1, float beta2, float eps, int64_t step,
                                       void* workspace, int64_t workspace_bytes, rsa_stream_t stream) {
  RSA_CHECK_ARG(exp_avg && exp_avg_sq && step >= 1 && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f,
                "rsa_adam_rows_presorted: bad optimizer state / hyper-parameters");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const AdamArgs adam{exp_avg, exp_avg_sq, 1.f - beta1, 1.f - beta2, eps, (float)((double)lr * sqrt(bc2) / bc1)};
  return apply_sorted_impl(query, query_index, n_query_rows, dim, has_pos != 0, n_queries, num_neg, dpos, dneg, upstream, n_items,
                           pad_row, weight, adam, workspace, workspace_bytes, stream);
}

extern "C" int rsa_adam_rows_sorted(const float* query, const int64_t* query_index, int64_t n_query_rows, int32_t dim,
                                    const int64_t* pos_ids, const int64_t* neg_ids, int64_t n_queries, int32_t num_neg,
                                    const float* dpos, const float* dneg, const float* upstream, int64_t n_items,
                                    int64_t pad_row, float* weight, float* exp_avg, float* exp_avg_sq, float lr,
                                    float beta1, float beta2, float eps, int64_t step, void* workspace,
                                    int64_t workspace_bytes, rsa_stream_t stream) {
  RSA_CHECK_ARG(exp_avg && exp_avg_sq && step >= 1 && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f,
                "rsa_adam_rows_sorted: bad optimizer state / hyper-parameters");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const AdamArgs adam{exp_avg, exp_avg_sq, 1.f - beta1, 1.f - beta2, eps, (float)((double)lr * sqrt(bc2) / bc1)};
  return scatter_sorted_impl(query, query_index, n_query_rows, dim, pos_ids, neg_ids, n_queries, num_neg, dpos, dneg,
                             upstream, n_items, pad_row, weight, adam, workspace, workspace_bytes, stream);
}
