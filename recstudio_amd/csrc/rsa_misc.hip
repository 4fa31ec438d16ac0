// Error plumbing, device info, plain embedding gather and the SeqDataset segment gather.
#include <cstdarg>
#include <cstdio>

#include "rsa_common.hpp"

namespace rsa {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// out[i, :] = table[ids[i], :]; D/4 lanes per row, 16-B accesses.
__global__ __launch_bounds__(256) void embedding_gather_kernel(const float* __restrict__ table, int64_t n_rows,
                                                               int D, const int64_t* __restrict__ ids,
                                                               int64_t numel, float* __restrict__ out) {
  const int vec = D >> 2;   // float4 per row
  const int64_t total = numel * vec;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t row = i / vec;
    const int c = (int)(i - row * vec);
    int64_t id = ids[row];
    id = id < 0 ? 0 : (id >= n_rows ? n_rows - 1 : id);
    reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(table + (size_t)id * D)[c];
  }
}

// One wave per (segment, position) pair group: block handles one segment.
__global__ __launch_bounds__(256) void seg_gather_kernel(const float* __restrict__ table, int64_t n_items, int D,
                                                         const int64_t* __restrict__ flat, int64_t n_flat,
                                                         const int64_t* __restrict__ seg_start,
                                                         const int64_t* __restrict__ seg_end, int max_len,
                                                         int64_t* __restrict__ out_ids, float* __restrict__ out_rows,
                                                         int64_t* __restrict__ out_len) {
  const int64_t b = blockIdx.x;
  int64_t s = seg_start[b], e = seg_end[b];
  s = s < 0 ? 0 : s;
  e = e > n_flat ? n_flat : e;
  int64_t len = e > s ? e - s : 0;
  if (len > max_len) {   // keep the most recent max_len items (dataset.py:1400,1409)
    s = e - max_len;
    len = max_len;
  }
  if (threadIdx.x == 0 && out_len) out_len[b] = len;
  if (out_ids)
    for (int l = threadIdx.x; l < max_len; l += blockDim.x) out_ids[b * max_len + l] = l < len ? flat[s + l] : 0;
  if (out_rows) {
    const int vec = D >> 2;
    const int total = max_len * vec;
    float4* orow = reinterpret_cast<float4*>(out_rows + (size_t)b * max_len * D);
    const int64_t last_flat = n_flat > 0 ? n_flat - 1 : 0;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
      const int l = i / vec, c = i - l * vec;
      // unconditional loads (a load under a lane predicate becomes a branch + vmcnt(0)): padded positions read a
      // valid address and are zeroed by a select
      const int64_t fi = s + l < last_flat ? s + l : last_flat;
      int64_t id = n_flat > 0 ? flat[fi] : 0;
      id = id < 0 ? 0 : (id >= n_items ? n_items - 1 : id);
      const float4 x = reinterpret_cast<const float4*>(table + (size_t)id * D)[c];
      orow[i] = l < len ? x : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// aux[r] = 1 / ||row r|| (cosine) or ||row r||^2 (Euclidean): one lane group of D/4 lanes per row, 16-byte loads.
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* __restrict__ table, int64_t n_rows, int D, int mode,
                                                         float* __restrict__ out) {
  const int lane = lane_id();
  const int64_t w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), ws = (int64_t)gridDim.x * 4;
  const int v4 = D >> 2;
  for (int64_t r = w0; r < n_rows; r += ws) {
    const float4* row = reinterpret_cast<const float4*>(table + (size_t)r * D);
    float acc = 0.f;
    for (int c = lane; c < v4; c += 64) {
      const float4 v = row[c];
      acc += dot4(v, v);
    }
    acc = group_sum<64>(acc);
    if (lane == 0) out[r] = mode == RSA_SCORE_COS ? 1.f / sqrtf(acc) : acc;
  }
}

// Placement probe (rsa_placement_probe): the access pattern of the fused forward without its arithmetic -- per tile of 64
// elements 64 random 512-byte rows read from `source`, 512 B + 3 x 256 B written to four arrays laid over `region` at quarter
// offsets, 4 x 4 B to per-tile arrays behind them.  Reads and writes together: the writes of a stand-alone write stream never
// leave the 256 MB Infinity Cache, it is under read load that the two classes of allocations differ.
__global__ __launch_bounds__(256) void placement_probe_kernel(char* __restrict__ region, int64_t quarter, uint32_t n_tiles,
                                                               const float4* __restrict__ source, uint32_t n_rows, uint32_t salt) {
  const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63, n_waves = gridDim.x * 4;
  char* ids = region;
  char* f0 = region + quarter;
  char* f1 = region + 2 * quarter;
  char* f2 = region + 3 * quarter;
  char* s0 = f2 + (size_t)n_tiles * 256;
  auto mix = [](uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; };
  for (uint32_t t = wave; t < n_tiles; t += n_waves) {
    float acc = 0.f;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      float4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {            // 2 half-waves x 16 rows x 2 passes = 64 rows per tile
        const uint32_t r = mix((t * 64 + pass * 32 + u * 2 + (lane >> 5)) ^ salt) % n_rows;
        v[u] = source[(size_t)r * 32 + (lane & 31)];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += v[u].x + v[u].w;
    }
    reinterpret_cast<int64_t*>(ids + (size_t)t * 512)[lane] = (int64_t)lane + (int64_t)acc;
    reinterpret_cast<float*>(f0 + (size_t)t * 256)[lane] = acc;
    reinterpret_cast<float*>(f1 + (size_t)t * 256)[lane] = acc + 1.f;
    reinterpret_cast<float*>(f2 + (size_t)t * 256)[lane] = acc + 2.f;
    if (lane < 4) reinterpret_cast<float*>(s0 + (size_t)lane * n_tiles * 4)[t] = acc;
  }
}

}  // namespace rsa

using namespace rsa;

extern "C" const char* rsa_last_error(void) { return g_err; }
extern "C" int rsa_abi_version(void) { return RSA_ABI_VERSION; }

extern "C" int rsa_device_info(int device, int32_t* cu_count, int32_t* max_threads_per_cu, int32_t* wave_size) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    set_error("rsa_device_info: hipGetDeviceProperties(%d): %s", device, hipGetErrorString(e));
    return RSA_ERR_HIP;
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (max_threads_per_cu) *max_threads_per_cu = prop.maxThreadsPerMultiProcessor;
  if (wave_size) *wave_size = prop.warpSize;
  return RSA_OK;
}

extern "C" int rsa_embedding_gather(const float* table, int64_t n_rows, int32_t dim, const int64_t* ids,
                                    int64_t numel, float* out, rsa_stream_t stream) {
  RSA_CHECK_ARG(numel >= 0 && n_rows >= 1, "rsa_embedding_gather: bad sizes");
  RSA_CHECK_ARG(dim >= 4 && dim % 4 == 0, "rsa_embedding_gather: dim=%d must be a positive multiple of 4", dim);
  if (numel == 0) return RSA_OK;
  RSA_CHECK_ARG(table && ids && out, "rsa_embedding_gather: null pointer");
  const int64_t total = numel * (dim / 4);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(embedding_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, table,
                     n_rows, (int)dim, ids, numel, out);
  RSA_CHECK_LAUNCH("rsa_embedding_gather");
  return RSA_OK;
}

extern "C" int rsa_seg_gather(const rsa_seg_gather_args* args, rsa_stream_t stream) {
  rsa_seg_gather_args a;
  if (int rc = load_args(a, args, "rsa_seg_gather")) return rc;
  const float* item_table = a.item_table;
  const int64_t n_items = a.n_items, n_flat = a.n_flat, n_seg = a.n_seg;
  const int32_t dim = a.dim, max_len = a.max_len;
  const int64_t *flat_item_ids = a.flat_item_ids, *seg_start = a.seg_start, *seg_end = a.seg_end;
  int64_t *out_ids = a.out_ids, *out_len = a.out_len;
  float* out_rows = a.out_rows;
  RSA_CHECK_ARG(n_seg >= 0 && max_len >= 1 && n_flat >= 0, "rsa_seg_gather: bad sizes");
  if (n_seg == 0) return RSA_OK;
  RSA_CHECK_ARG(flat_item_ids && seg_start && seg_end, "rsa_seg_gather: null input pointer");
  RSA_CHECK_ARG(out_rows == nullptr || (item_table && dim >= 4 && dim % 4 == 0 && n_items >= 1),
                "rsa_seg_gather: out_rows needs item_table and dim % 4 == 0");
  hipLaunchKernelGGL(seg_gather_kernel, dim3((unsigned)n_seg), dim3(256), 0, (hipStream_t)stream, item_table, n_items,
                     (int)dim, flat_item_ids, n_flat, seg_start, seg_end, (int)max_len, out_ids, out_rows, out_len);
  RSA_CHECK_LAUNCH("rsa_seg_gather");
  return RSA_OK;
}

extern "C" int rsa_row_sqnorm(const float* table, int64_t n_rows, int32_t dim, int32_t score_mode, float* out,
                              rsa_stream_t stream) {
  RSA_CHECK_ARG(n_rows >= 0 && dim >= 4 && dim % 4 == 0, "rsa_row_sqnorm: bad sizes");
  RSA_CHECK_ARG(score_mode == RSA_SCORE_COS || score_mode == RSA_SCORE_EUC, "rsa_row_sqnorm: score_mode must be COS or EUC");
  if (n_rows == 0) return RSA_OK;
  RSA_CHECK_ARG(table && out, "rsa_row_sqnorm: null pointer");
  int64_t blocks = (n_rows + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(rsa::row_sqnorm_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, table, n_rows,
                     (int)dim, (int)score_mode, out);
  RSA_CHECK_LAUNCH("rsa_row_sqnorm");
  return RSA_OK;
}

extern "C" int rsa_placement_probe(void* region, int64_t region_bytes, const void* source, int64_t source_bytes, uint32_t salt,
                                   rsa_stream_t stream) {
  RSA_CHECK_ARG(region && source, "rsa_placement_probe: null pointer");
  RSA_CHECK_ARG(region_bytes >= (int64_t)1 << 20 && source_bytes >= (int64_t)1 << 20, "rsa_placement_probe: region and source must hold at least 1 MiB");
  RSA_CHECK_ARG(((uintptr_t)region & 255) == 0 && ((uintptr_t)source & 15) == 0, "rsa_placement_probe: region must be 256-byte, source 16-byte aligned");
  // four arrays at quarter offsets; the first needs 512 B per tile, the last 256 B + 16 B per tile
  const int64_t quarter = region_bytes / 4 / 256 * 256;
  int64_t tiles = quarter / 512;
  if (tiles > 65536) tiles = 65536;
  const int64_t rows = source_bytes / 512;
  RSA_CHECK_ARG(tiles >= 1 && rows >= 1 && rows <= 0xffffffffLL, "rsa_placement_probe: bad sizes");
  hipLaunchKernelGGL(placement_probe_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, (char*)region, quarter, (uint32_t)tiles,
                     (const float4*)source, (uint32_t)rows, salt);
  RSA_CHECK_LAUNCH("rsa_placement_probe");
  return RSA_OK;
}
