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

__device__ __forceinline__ int64_t element_id(const int64_t* __restrict__ pos_ids, const int64_t* __restrict__ neg_ids,
                                              int64_t e, int n, int64_t& m, int& c) {
  const int w = n + 1;
  m = e / w;
  c = (int)(e - m * w);
  return c == 0 ? pos_ids[m] : neg_ids[m * n + (c - 1)];
}

__global__ __launch_bounds__(256) void shard_count_kernel(const int64_t* __restrict__ pos_ids,
                                                          const int64_t* __restrict__ neg_ids, int64_t n_queries,
                                                          int n, int64_t rows_per_shard, int G,
                                                          int32_t* __restrict__ counts) {
  __shared__ int32_t h[64];
  if (threadIdx.x < 64) h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t numel = n_queries * (n + 1);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < numel; e += stride) {
    int64_t m;
    int c;
    const int64_t id = element_id(pos_ids, neg_ids, e, n, m, c);
    int g = (int)(id / rows_per_shard);
    g = g < 0 ? 0 : (g >= G ? G - 1 : g);
    atomicAdd(&h[g], 1);
  }
  __syncthreads();
  if (threadIdx.x < G && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], h[threadIdx.x]);
}

// Counting-sort scatter.  Each workgroup owns a contiguous chunk of elements: pass 1 counts the chunk's
// elements per owner in LDS, ONE global atomic per (workgroup, owner) reserves a contiguous slot range,
// pass 2 hands out the slots from LDS counters.  (A first version did one global atomic per wave and
// owner on the same few cursor words and spent 0.8 ms there for 4 M elements.)
__global__ __launch_bounds__(256) void shard_route_kernel(const int64_t* __restrict__ pos_ids,
                                                          const int64_t* __restrict__ neg_ids, int64_t n_queries,
                                                          int n, int64_t rows_per_shard, int G, int64_t query_base,
                                                          int64_t chunk, int32_t* __restrict__ cursor,
                                                          int64_t* __restrict__ keys, int64_t* __restrict__ pos_out) {
  __shared__ int32_t cnt[64], base[64];
  const int64_t numel = n_queries * (n + 1);
  const int64_t e_lo = (int64_t)blockIdx.x * chunk;
  int64_t e_hi = e_lo + chunk;
  if (e_hi > numel) e_hi = numel;
  if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
  __syncthreads();
  for (int64_t e = e_lo + threadIdx.x; e < e_hi; e += 256) {
    int64_t m;
    int c;
    const int64_t id = element_id(pos_ids, neg_ids, e, n, m, c);
    int g = (int)(id / rows_per_shard);
    g = g < 0 ? 0 : (g >= G ? G - 1 : g);
    atomicAdd(&cnt[g], 1);
  }
  __syncthreads();
  if (threadIdx.x < G) {
    base[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], cnt[threadIdx.x]) : 0;
    cnt[threadIdx.x] = 0;
  }
  __syncthreads();
  for (int64_t e = e_lo + threadIdx.x; e < e_hi; e += 256) {
    int64_t m;
    int c;
    const int64_t id = element_id(pos_ids, neg_ids, e, n, m, c);
    int g = (int)(id / rows_per_shard);
    g = g < 0 ? 0 : (g >= G ? G - 1 : g);
    const int32_t slot = base[g] + atomicAdd(&cnt[g], 1);
    const int64_t local = id - (int64_t)g * rows_per_shard;
    keys[slot] = ((query_base + m) << 32) | (local & 0xffffffffll);
    // destination of this element's score in the home buffer [pos_score (n_queries) | neg_score (n_queries x n)]
    pos_out[slot] = c == 0 ? m : n_queries + m * n + (c - 1);
  }
}

// Fixed-capacity variant of shard_route_kernel: owner g's segment is slots [g*capacity, (g+1)*capacity); the
// per-owner cursor counts from 0 and an element that does not fit is dropped and counted in *overflow.
__global__ __launch_bounds__(256) void shard_route_fixed_kernel(const int64_t* __restrict__ pos_ids,
                                                                const int64_t* __restrict__ neg_ids, int64_t n_queries,
                                                                int n, int64_t rows_per_shard, int G, int64_t query_base,
                                                                int64_t chunk, int64_t capacity, int32_t* __restrict__ cursor,
                                                                int64_t* __restrict__ keys, int64_t* __restrict__ pos_out,
                                                                int32_t* __restrict__ overflow) {
  __shared__ int32_t cnt[64], base[64];
  const int64_t numel = n_queries * (n + 1);
  const int64_t e_lo = (int64_t)blockIdx.x * chunk;
  int64_t e_hi = e_lo + chunk;
  if (e_hi > numel) e_hi = numel;
  if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
  __syncthreads();
  for (int64_t e = e_lo + threadIdx.x; e < e_hi; e += 256) {
    int64_t m;
    int c;
    const int64_t id = element_id(pos_ids, neg_ids, e, n, m, c);
    int g = (int)(id / rows_per_shard);
    g = g < 0 ? 0 : (g >= G ? G - 1 : g);
    atomicAdd(&cnt[g], 1);
  }
  __syncthreads();
  if (threadIdx.x < G) {
    base[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], cnt[threadIdx.x]) : 0;
    cnt[threadIdx.x] = 0;
  }
  __syncthreads();
  int dropped = 0;
  for (int64_t e = e_lo + threadIdx.x; e < e_hi; e += 256) {
    int64_t m;
    int c;
    const int64_t id = element_id(pos_ids, neg_ids, e, n, m, c);
    int g = (int)(id / rows_per_shard);
    g = g < 0 ? 0 : (g >= G ? G - 1 : g);
    const int64_t slot = (int64_t)base[g] + atomicAdd(&cnt[g], 1);
    if (slot >= capacity) {
      ++dropped;
      continue;
    }
    const int64_t local = id - (int64_t)g * rows_per_shard;
    keys[g * capacity + slot] = ((query_base + m) << 32) | (local & 0xffffffffll);
    pos_out[g * capacity + slot] = c == 0 ? m : n_queries + m * n + (c - 1);
  }
  if (dropped) atomicAdd(overflow, dropped);
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
  hipLaunchKernelGGL(shard_count_kernel, dim3(grid1d(n_queries * (num_neg + 1))), dim3(256), 0, (hipStream_t)stream,
                     pos_ids, neg_ids, n_queries, (int)num_neg, rows_per_shard, (int)n_shards, counts);
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
  int64_t blocks = (numel + 16383) / 16384;            // >= 16 K elements per workgroup ...
  if (blocks < 512 && numel > 512 * 1024) blocks = 512;   // ... but keep the chip busy on mid-size batches
  const int64_t chunk = (numel + blocks - 1) / blocks;
  hipLaunchKernelGGL(shard_route_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pos_ids, neg_ids,
                     n_queries, (int)num_neg, rows_per_shard, (int)n_shards, query_base, chunk, cursor, keys, positions);
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
  int64_t blocks = (numel + 16383) / 16384;
  if (blocks < 512 && numel > 512 * 1024) blocks = 512;
  const int64_t chunk = (numel + blocks - 1) / blocks;
  hipLaunchKernelGGL(shard_route_fixed_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pos_ids, neg_ids, n_queries,
                     (int)num_neg, rows_per_shard, (int)n_shards, query_base, chunk, capacity, cursor, keys, positions,
                     overflow);
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
