// Routing kernels for a row-sharded item table (nothing comparable exists in
// the reference, whose only multi-device mode replicates the full tables, data_parallel.py:106-159).
//
// A rank samples negatives for its own queries, then every (query, item) element travels to the
// rank that owns the item row as one packed 64-bit key, is scored there, and its fp32 score comes
// back:  8 + 4 bytes per triplet over xGMI instead of a 512-byte row.
//
//   rsa_shard_count   : histogram of owners                     (-> all-to-all split sizes)
//   rsa_shard_route   : counting-sort scatter into per-owner segments, key = qidx << 32 | local row
//   rsa_shard_unpack  : owner side, key -> (local row int64, query index int64)
//   rsa_scatter_f32   : home side, dst[pos[i]] = src[i]  (returned scores -> [pos_score | neg_score] buffer)
#include "rsa_common.hpp"

namespace rsa {

// x / d for 0 <= x < 2^32 by one 64-bit multiply-high: magic = floor(2^64 / d) + 1 (host side, d >= 2) is exact on
// that range; d == 1 is passed as magic == 0.  (The plain 64-bit divisions of the first version -- two per element
// and pass -- were a third of the routing kernel.)
struct FastDiv {
  uint64_t magic;
  __device__ __forceinline__ uint64_t div(uint64_t x) const { return magic ? __umul64hi(x, magic) : x; }
};
static inline FastDiv make_fastdiv(uint64_t d) {
  FastDiv f;
  f.magic = d <= 1 ? 0 : (~0ull / d) + 1;     // floor((2^64 - 1) / d) + 1 == floor(2^64 / d) + 1 unless d | 2^64 (then exact too)
  return f;
}

struct RouteShape {
  const int64_t* pos_ids;
  const int64_t* neg_ids;
  int64_t n_queries;
  int n, G;
  int64_t rows_per_shard;
  FastDiv by_width, by_rows;
};

// element e of the [n_queries, 1 + n] (positive, negatives) layout -> (id, query m, column c, owner g)
__device__ __forceinline__ int64_t route_element(const RouteShape& sh, int64_t e, int64_t& m, int& c, int& g) {
  m = (int64_t)sh.by_width.div((uint64_t)e);
  c = (int)(e - m * (sh.n + 1));
  const int64_t id = c == 0 ? sh.pos_ids[m] : sh.neg_ids[m * sh.n + (c - 1)];
  const int64_t q = id < 0 ? 0 : (int64_t)sh.by_rows.div((uint64_t)id);
  g = q >= sh.G ? sh.G - 1 : (int)q;
  return id;
}

// Per-owner counters in LDS.  With two or more owners: one LDS atomic per element (the LDS unit resolves the
// same-address conflicts of 64 lanes over a few counters faster than software can: 45 vs 73 us for 8 owners, 45 vs 50
// for 2).  With ONE owner the 64-way conflict dominates (59 us) and the wave aggregates first: lanes with the same owner
// are found by ballot, the first of them adds their number and every lane takes its rank among them (46 us).
template <bool WANT_SLOT>
__device__ __forceinline__ int32_t wave_count(int32_t* cnt, bool valid, int g, bool aggregate) {
  if (!aggregate) {
    if (!valid) return 0;
    if (WANT_SLOT) return atomicAdd(&cnt[g], 1);
    atomicAdd(&cnt[g], 1);
    return 0;
  }
  const int lane = threadIdx.x & 63;
  uint64_t todo = __ballot(valid);
  int32_t slot = 0;
  while (todo) {
    const int leader = __ffsll((unsigned long long)todo) - 1;
    const int gg = __shfl(g, leader, 64);
    const uint64_t same = __ballot(valid && g == gg);
    int32_t old = 0;
    if (lane == leader) old = atomicAdd(&cnt[gg], (int32_t)__popcll(same));
    if (WANT_SLOT) {
      old = __shfl(old, leader, 64);
      if (valid && g == gg) slot = old + (int32_t)__popcll(same & ((1ull << lane) - 1ull));
    }
    todo &= ~same;
  }
  return slot;
}

// Elements per thread of the routing kernels.  A thread first issues the loads of ALL its ids (independent, all in
// flight together) and only then counts: with one id per loop iteration the kernels were a chain of dependent
// HBM round trips (16 per pass, ~50 us for 4.2 M elements that take 6 us to read).
constexpr int ROUTE_EPT = 16;
constexpr int ROUTE_CHUNK = 256 * ROUTE_EPT;

__global__ __launch_bounds__(256) void shard_count_kernel(RouteShape sh, int32_t* __restrict__ counts) {
  __shared__ int32_t h[64];
  if (threadIdx.x < 64) h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t numel = sh.n_queries * (sh.n + 1);
  for (int64_t e_lo = (int64_t)blockIdx.x * ROUTE_CHUNK; e_lo < numel; e_lo += (int64_t)gridDim.x * ROUTE_CHUNK) {
    int g[ROUTE_EPT];
#pragma unroll
    for (int k = 0; k < ROUTE_EPT; ++k) {
      const int64_t e = e_lo + k * 256 + threadIdx.x;
      int64_t m;
      int c;
      g[k] = -1;
      if (e < numel) route_element(sh, e, m, c, g[k]);
    }
#pragma unroll
    for (int k = 0; k < ROUTE_EPT; ++k) wave_count<false>(h, g[k] >= 0, g[k], sh.G == 1);
  }
  __syncthreads();
  if (threadIdx.x < sh.G && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], h[threadIdx.x]);
}

// Counting-sort scatter.  Each workgroup owns a contiguous chunk of ROUTE_CHUNK elements, held in registers as
// (owner, local row): pass 1 counts the chunk's elements per owner in LDS, ONE global atomic per (workgroup, owner)
// reserves a contiguous slot range, pass 2 hands out the slots from LDS counters.  (A first version did one global
// atomic per wave and owner on the same few cursor words and spent 0.8 ms there for 4 M elements.)
// FIXED: owner g's segment is slots [g*capacity, (g+1)*capacity), the per-owner cursor counts from 0, and an element
// that does not fit is dropped and counted in *overflow.  Otherwise the cursors arrive holding the segment starts.
template <bool FIXED>
__global__ __launch_bounds__(256) void shard_route_kernel(RouteShape sh, int64_t query_base, int64_t capacity,
                                                          int32_t* __restrict__ cursor, int64_t* __restrict__ keys,
                                                          int64_t* __restrict__ pos_out, int32_t* __restrict__ overflow) {
  __shared__ int32_t cnt[64], base[64];
  const int64_t numel = sh.n_queries * (sh.n + 1);
  const int64_t e_lo = (int64_t)blockIdx.x * ROUTE_CHUNK;
  if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
  int g[ROUTE_EPT];
  uint32_t local[ROUTE_EPT];
#pragma unroll
  for (int k = 0; k < ROUTE_EPT; ++k) {
    const int64_t e = e_lo + k * 256 + threadIdx.x;
    int64_t m;
    int c;
    g[k] = -1;
    local[k] = 0;
    if (e < numel) {
      const int64_t id = route_element(sh, e, m, c, g[k]);
      local[k] = (uint32_t)(id - (int64_t)g[k] * sh.rows_per_shard);
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ROUTE_EPT; ++k) wave_count<false>(cnt, g[k] >= 0, g[k], sh.G == 1);
  __syncthreads();
  if (threadIdx.x < sh.G) {
    base[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], cnt[threadIdx.x]) : 0;
    cnt[threadIdx.x] = 0;
  }
  __syncthreads();
  int dropped = 0;
#pragma unroll
  for (int k = 0; k < ROUTE_EPT; ++k) {
    const bool valid = g[k] >= 0;
    const int gk = valid ? g[k] : 0;
    const int64_t slot = (int64_t)base[gk] + wave_count<true>(cnt, valid, gk, sh.G == 1);
    if (!valid) continue;
    if (FIXED && slot >= capacity) {
      ++dropped;
      continue;
    }
    const int64_t e = e_lo + k * 256 + threadIdx.x;
    const int64_t m = (int64_t)sh.by_width.div((uint64_t)e);
    const int c = (int)(e - m * (sh.n + 1));
    const int64_t at = FIXED ? gk * capacity + slot : slot;
    keys[at] = ((query_base + m) << 32) | (int64_t)local[k];
    // destination of this element's score in the home buffer [pos_score (n_queries) | neg_score (n_queries x n)]
    pos_out[at] = c == 0 ? m : sh.n_queries + m * sh.n + (c - 1);
  }
  if (FIXED && dropped) atomicAdd(overflow, dropped);
}

// keys / positions of the UNUSED tail of every owner segment <- -1 (cursor[g] = elements routed to g, possibly more
// than the capacity).  Only the slack is written: filling both whole buffers first cost 5 % of the sharded step.
__global__ __launch_bounds__(256) void shard_fill_tail_kernel(const int32_t* __restrict__ cursor, int64_t capacity,
                                                              int64_t* __restrict__ keys, int64_t* __restrict__ pos_out) {
  const int g = blockIdx.y;
  const int64_t used = cursor[g] < capacity ? cursor[g] : capacity;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = used + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < capacity; i += stride) {
    keys[g * capacity + i] = -1;
    pos_out[g * capacity + i] = -1;
  }
}

__global__ __launch_bounds__(256) void shard_unpack_kernel(const int64_t* __restrict__ keys, int64_t numel,
                                                           int64_t* __restrict__ local_rows,
                                                           int64_t* __restrict__ qidx) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) {
    const int64_t k = keys[i];
    local_rows[i] = k < 0 ? -1 : (k & 0xffffffffll);      // empty slot of the fixed-capacity exchange
    qidx[i] = k < 0 ? -1 : ((k >> 32) & 0x7fffffffll);
  }
}

__global__ __launch_bounds__(256) void scatter_f32_kernel(const float* __restrict__ src,
                                                          const int64_t* __restrict__ pos, int64_t numel,
                                                          float* __restrict__ dst) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) {
    const int64_t q = pos[i];
    if (q >= 0) dst[q] = src[i];
  }
}

__global__ __launch_bounds__(256) void gather_f32_kernel(const float* __restrict__ src,
                                                         const int64_t* __restrict__ pos, int64_t numel,
                                                         float* __restrict__ dst) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) {
    const int64_t q = pos[i];
    dst[i] = q >= 0 ? src[q] : 0.f;
  }
}

static inline int grid1d(int64_t numel) {
  int64_t b = (numel + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace rsa

using namespace rsa;

extern "C" int rsa_shard_count(const int64_t* pos_ids, const int64_t* neg_ids, int64_t n_queries, int32_t num_neg,
                               int64_t rows_per_shard, int32_t n_shards, int32_t* counts, rsa_stream_t stream) {
  RSA_CHECK_ARG(n_queries >= 0 && num_neg >= 0 && rows_per_shard >= 1, "rsa_shard_count: bad sizes");
  RSA_CHECK_ARG(n_shards >= 1 && n_shards <= 64, "rsa_shard_count: n_shards must be in [1, 64]");
  RSA_CHECK_ARG(counts != nullptr, "rsa_shard_count: counts is null");
  hipError_t e = hipMemsetAsync(counts, 0, sizeof(int32_t) * n_shards, (hipStream_t)stream);
  if (e != hipSuccess) {
    rsa::set_error("rsa_shard_count: memset failed: %s", hipGetErrorString(e));
    return RSA_ERR_HIP;
  }
  if (n_queries == 0) return RSA_OK;
  RSA_CHECK_ARG(pos_ids && (neg_ids || num_neg == 0), "rsa_shard_count: null ids");
  RSA_CHECK_ARG(n_queries * (num_neg + 1) < (1ll << 32), "rsa_shard_count: more than 2^32 elements");
  const RouteShape sh{pos_ids, neg_ids, n_queries, (int)num_neg, (int)n_shards, rows_per_shard,
                      make_fastdiv((uint64_t)num_neg + 1), make_fastdiv((uint64_t)rows_per_shard)};
  int64_t count_blocks = (n_queries * (num_neg + 1) + ROUTE_CHUNK - 1) / ROUTE_CHUNK;
  if (count_blocks > 4096) count_blocks = 4096;
  hipLaunchKernelGGL(shard_count_kernel, dim3((unsigned)count_blocks), dim3(256), 0, (hipStream_t)stream, sh, counts);
  RSA_CHECK_LAUNCH("rsa_shard_count");
  return RSA_OK;
}

extern "C" int rsa_shard_route(const int64_t* pos_ids, const int64_t* neg_ids, int64_t n_queries, int32_t num_neg,
                               int64_t rows_per_shard, int32_t n_shards, int64_t query_base, int32_t* cursor,
                               int64_t* keys, int64_t* positions, rsa_stream_t stream) {
  RSA_CHECK_ARG(n_queries >= 0 && num_neg >= 0 && rows_per_shard >= 1 && rows_per_shard < (1ll << 32),
                "rsa_shard_route: bad sizes");
  RSA_CHECK_ARG(n_shards >= 1 && n_shards <= 64, "rsa_shard_route: n_shards must be in [1, 64]");
  RSA_CHECK_ARG(query_base >= 0 && query_base + n_queries < (1ll << 31), "rsa_shard_route: query index overflow");
  if (n_queries == 0) return RSA_OK;
  RSA_CHECK_ARG(pos_ids && (neg_ids || num_neg == 0) && cursor && keys && positions, "rsa_shard_route: null pointer");
  const int64_t numel = n_queries * (num_neg + 1);
  const int64_t blocks = (numel + ROUTE_CHUNK - 1) / ROUTE_CHUNK;
  RSA_CHECK_ARG(numel < (1ll << 31), "rsa_shard_route: more than 2^31 elements");
  const RouteShape sh{pos_ids, neg_ids, n_queries, (int)num_neg, (int)n_shards, rows_per_shard,
                      make_fastdiv((uint64_t)num_neg + 1), make_fastdiv((uint64_t)rows_per_shard)};
  hipLaunchKernelGGL(shard_route_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, sh, query_base,
                     (int64_t)0, cursor, keys, positions, (int32_t*)nullptr);
  RSA_CHECK_LAUNCH("rsa_shard_route");
  return RSA_OK;
}

extern "C" int rsa_shard_route_fixed(const int64_t* pos_ids, const int64_t* neg_ids, int64_t n_queries, int32_t num_neg,
                                     int64_t rows_per_shard, int32_t n_shards, int64_t query_base, int64_t capacity,
                                     int32_t* cursor, int64_t* keys, int64_t* positions, int32_t* overflow,
                                     rsa_stream_t stream) {
  RSA_CHECK_ARG(n_queries >= 0 && num_neg >= 0 && rows_per_shard >= 1 && rows_per_shard < (1ll << 32),
                "rsa_shard_route_fixed: bad sizes");
  RSA_CHECK_ARG(n_shards >= 1 && n_shards <= 64, "rsa_shard_route_fixed: n_shards must be in [1, 64]");
  RSA_CHECK_ARG(query_base >= 0 && query_base + n_queries < (1ll << 31), "rsa_shard_route_fixed: query index overflow");
  RSA_CHECK_ARG(capacity >= 1 && capacity * n_shards < (1ll << 31), "rsa_shard_route_fixed: capacity out of range");
  RSA_CHECK_ARG(cursor && keys && positions && overflow, "rsa_shard_route_fixed: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(cursor, 0, sizeof(int32_t) * n_shards, s) != hipSuccess) {
    rsa::set_error("rsa_shard_route_fixed: memset failed");
    return RSA_ERR_HIP;
  }
  int64_t tail_blocks = (capacity + 255) / 256;
  if (tail_blocks > 256) tail_blocks = 256;
  if (n_queries == 0) {
    hipLaunchKernelGGL(shard_fill_tail_kernel, dim3((unsigned)tail_blocks, (unsigned)n_shards), dim3(256), 0, s, cursor,
                       capacity, keys, positions);
    RSA_CHECK_LAUNCH("rsa_shard_route_fixed(fill)");
    return RSA_OK;
  }
  RSA_CHECK_ARG(pos_ids && (neg_ids || num_neg == 0), "rsa_shard_route_fixed: null ids");
  const int64_t numel = n_queries * (num_neg + 1);
  const int64_t blocks = (numel + ROUTE_CHUNK - 1) / ROUTE_CHUNK;
  RSA_CHECK_ARG(numel < (1ll << 31), "rsa_shard_route_fixed: more than 2^31 elements");
  const RouteShape sh{pos_ids, neg_ids, n_queries, (int)num_neg, (int)n_shards, rows_per_shard,
                      make_fastdiv((uint64_t)num_neg + 1), make_fastdiv((uint64_t)rows_per_shard)};
  hipLaunchKernelGGL(shard_route_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, sh, query_base, capacity, cursor,
                     keys, positions, overflow);
  hipLaunchKernelGGL(shard_fill_tail_kernel, dim3((unsigned)tail_blocks, (unsigned)n_shards), dim3(256), 0, s, cursor,
                     capacity, keys, positions);
  RSA_CHECK_LAUNCH("rsa_shard_route_fixed");
  return RSA_OK;
}

extern "C" int rsa_shard_unpack(const int64_t* keys, int64_t numel, int64_t* local_rows, int64_t* query_index,
                                rsa_stream_t stream) {
  RSA_CHECK_ARG(numel >= 0, "rsa_shard_unpack: numel < 0");
  if (numel == 0) return RSA_OK;
  RSA_CHECK_ARG(keys && local_rows && query_index, "rsa_shard_unpack: null pointer");
  hipLaunchKernelGGL(shard_unpack_kernel, dim3(grid1d(numel)), dim3(256), 0, (hipStream_t)stream, keys, numel,
                     local_rows, query_index);
  RSA_CHECK_LAUNCH("rsa_shard_unpack");
  return RSA_OK;
}

extern "C" int rsa_scatter_f32(const float* src, const int64_t* positions, int64_t numel, float* dst,
                               rsa_stream_t stream) {
  RSA_CHECK_ARG(numel >= 0, "rsa_scatter_f32: numel < 0");
  if (numel == 0) return RSA_OK;
  RSA_CHECK_ARG(src && positions && dst, "rsa_scatter_f32: null pointer");
  hipLaunchKernelGGL(scatter_f32_kernel, dim3(grid1d(numel)), dim3(256), 0, (hipStream_t)stream, src, positions, numel,
                     dst);
  RSA_CHECK_LAUNCH("rsa_scatter_f32");
  return RSA_OK;
}

extern "C" int rsa_gather_f32(const float* src, const int64_t* positions, int64_t numel, float* dst,
                              rsa_stream_t stream) {
  RSA_CHECK_ARG(numel >= 0, "rsa_gather_f32: numel < 0");
  if (numel == 0) return RSA_OK;
  RSA_CHECK_ARG(src && positions && dst, "rsa_gather_f32: null pointer");
  hipLaunchKernelGGL(gather_f32_kernel, dim3(grid1d(numel)), dim3(256), 0, (hipStream_t)stream, src, positions, numel,
                     dst);
  RSA_CHECK_LAUNCH("rsa_gather_f32");
  return RSA_OK;
}
