// Stand-alone negative samplers + small id utilities (gfx950).
//
// rsa_sample_uniform  : UniformSampler.forward      recstudio/ann/sampler.py:86-111
// rsa_sample_popular  : PopularSamplerModel.forward recstudio/ann/sampler.py:243-258
// rsa_item_logp       : compute_item_p              recstudio/ann/sampler.py:257-258
//
// The fused forward (rsa_fused.hip) draws the same stream in-kernel; these entry
// points exist for the Sampler plugin surface (a Sampler used on its own) and as
// the unit the parity tests compare with torch.randint / torch.rand on device.
#include "rsa_common.hpp"
#include "rsa_internal.hpp"

namespace rsa {

__global__ __launch_bounds__(256) void sample_uniform_kernel(int64_t* __restrict__ out, int64_t numel,
                                                             uint64_t range, int64_t low, PhiloxCall pc) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < numel; e += stride)
    out[e] = torch_randint_element(pc, (uint64_t)e, range, low);
}

template <bool FROM_U>
__global__ __launch_bounds__(256) void sample_popular_kernel(const float* __restrict__ table,
                                                             const float* __restrict__ pop_prob,
                                                             const int32_t* __restrict__ guide,
                                                             const float* __restrict__ lut,
                                                             const float* __restrict__ lines, int lines_log2,
                                                             int64_t n_items,
                                                             int guide_log2, const float* __restrict__ u_in,
                                                             int64_t* __restrict__ ids, float* __restrict__ logp,
                                                             float* __restrict__ u_out, int64_t numel,
                                                             PhiloxCall pc) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < numel; e += stride) {
    const float u = FROM_U ? u_in[e] : torch_rand_element(pc, (uint64_t)e);
    int32_t id;
    float pr;
    if (lines != nullptr) {   // bucket lines: one 128-byte line resolves id and probability
      id = cdf_lookup_line(lines, lines_log2, table, 1, pop_prob, 1, n_items, u, pr);
    } else if (lut != nullptr) {   // direct lookup: id and probability in one round trip (see rsa_common.hpp)
      id = cdf_lookup_lut<1>(reinterpret_cast<const float4*>(lut), table, pop_prob, 1, n_items, guide_log2, u, pr);
    } else {
      id = cdf_lower_bound(table, guide, n_items, guide_log2, u);
      pr = logp ? pop_prob[id] : 1.f;
    }
    ids[e] = id;
    if (logp) logp[e] = logf(pr);
    if (!FROM_U && u_out) u_out[e] = u;
  }
}

// LUT variant with E independent look-ups per thread.  A look-up is a chain of dependent random reads
// (LUT entry -> 0..log2(bucket size) CDF probes -> probability), each a full HBM line; with one element per
// thread the kernel is bound by the longest chain of every wave (205 us for 4.2 M samples at N = 1e7).
// Batching E chains per thread keeps E loads in flight per lane for the same number of round trips.
template <bool FROM_U, int E>
__global__ __launch_bounds__(256) void sample_popular_lut_kernel(const float* __restrict__ table,
                                                                 const float* __restrict__ pop_prob,
                                                                 const float4* __restrict__ lut, int64_t n_items,
                                                                 int guide_log2, const float* __restrict__ u_in,
                                                                 int64_t* __restrict__ ids, float* __restrict__ logp,
                                                                 float* __restrict__ u_out, int64_t numel,
                                                                 PhiloxCall pc) {
  const int64_t base = (int64_t)blockIdx.x * (256 * E) + threadIdx.x;
  const int32_t K = 1 << guide_log2, last = (int32_t)(n_items - 1);
  float u[E], pr[E];
  int32_t lo[E], hi[E];
  bool have_pr[E];
  float4 e0[E];
  int32_t bk[E];
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const int64_t e = base + k * 256;
    const int64_t ee = e < numel ? e : numel - 1;          // tail lanes repeat the last element (never stored)
    u[k] = FROM_U ? u_in[ee] : torch_rand_element(pc, (uint64_t)ee);
    int32_t b = (int32_t)(u[k] * (float)K);
    bk[k] = b < 0 ? 0 : (b > K - 1 ? K - 1 : b);
    e0[k] = lut[bk[k]];
  }
  constexpr int PROBE = 4;
  float c[E][PROBE];
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const uint32_t x = __float_as_uint(e0[k].x);
    lo[k] = (int32_t)(x & ~LUT_SEARCH_BIT);
    have_pr[k] = !(x & LUT_SEARCH_BIT);
    if (have_pr[k]) {                                       // same decision as cdf_lookup_lut
      const bool up = e0[k].y < u[k];
      pr[k] = up ? e0[k].w : e0[k].z;
      lo[k] += up ? 1 : 0;
    }
    // buckets with >= 2 boundaries: probe the next PROBE CDF entries at once (see cdf_lookup_lut); resolved
    // chains re-read the cached table[0] so that the loads stay unconditional and batched
#pragma unroll
    for (int i = 0; i < PROBE; ++i) {
      const int32_t j = lo[k] + i > last ? last : lo[k] + i;
      c[k][i] = table[have_pr[k] ? 0 : j];
    }
  }
#pragma unroll
  for (int k = 0; k < E; ++k) {
    hi[k] = lo[k];
    if (!have_pr[k]) {
      int cnt = 0;
#pragma unroll
      for (int i = 0; i < PROBE; ++i) cnt += (lo[k] + i <= last && c[k][i] < u[k]) ? 1 : 0;
      lo[k] += cnt;
      hi[k] = lo[k];
      if (cnt == PROBE) {                                   // rare: more than PROBE boundaries below u
        hi[k] = (int32_t)(__float_as_uint(lut[bk[k] + 1].x) & ~LUT_SEARCH_BIT);
        while (lo[k] < hi[k]) {
          const int32_t mid = lo[k] + ((hi[k] - lo[k]) >> 1);
          if (table[mid] < u[k]) lo[k] = mid + 1; else hi[k] = mid;
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < E; ++k) lo[k] = lo[k] > last ? last : lo[k];
  if (logp != nullptr) {
    float p2[E];
#pragma unroll
    for (int k = 0; k < E; ++k) p2[k] = pop_prob[have_pr[k] ? 0 : lo[k]];
#pragma unroll
    for (int k = 0; k < E; ++k) if (!have_pr[k]) pr[k] = p2[k];
  }
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const int64_t e = base + k * 256;
    if (e < numel) {
      ids[e] = lo[k];
      if (logp) logp[e] = logf(pr[k]);
      if (!FROM_U && u_out) u_out[e] = u[k];
    }
  }
}

__global__ __launch_bounds__(256) void item_logp_kernel(const float* __restrict__ pop_prob, int64_t n_items,
                                                        const int64_t* __restrict__ ids, int64_t numel,
                                                        float* __restrict__ logp) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < numel; e += stride) {
    int64_t id = ids[e];
    id = id < 0 ? 0 : (id >= n_items ? n_items - 1 : id);
    logp[e] = logf(pop_prob[id]);
  }
}

// uniform_sample_masked_hist -- recstudio/ann/sampler.py:117-147.  One workgroup per user row: the 0-padded
// history is sorted in LDS (zeros first), entry k of the sorted non-zero part is shifted down by its rank
// ("h_i - i"), and a draw from [1, num_items - |hist|] is pushed up by the number of shifted entries <= it:
// a rejection-free uniform sample over the items outside the history.
constexpr int MASK_MAX_HIST = 2048;
__global__ __launch_bounds__(256) void sample_masked_kernel(const int64_t* __restrict__ user_hist, int hist_len,
                                                            int64_t num_items, int per_row,
                                                            int64_t* __restrict__ out, PhiloxCall pc) {
  __shared__ int32_t h[MASK_MAX_HIST];
  __shared__ int32_t s_nz;
  const int64_t b = blockIdx.x;
  int P = 1;
  while (P < hist_len) P <<= 1;
  if (threadIdx.x == 0) s_nz = 0;
  __syncthreads();
  int nz_local = 0;
  for (int k = threadIdx.x; k < P; k += 256) {
    const int32_t v = k < hist_len ? (int32_t)user_hist[b * hist_len + k] : 0x7fffffff;   // pad to a power of two
    h[k] = v;
    nz_local += (k < hist_len && v != 0) ? 1 : 0;
  }
  if (nz_local) atomicAdd(&s_nz, nz_local);
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1)            // bitonic sort, ascending
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int k = threadIdx.x; k < P; k += 256) {
        const int partner = k ^ stride;
        if (partner > k) {
          const bool up = (k & size) == 0;
          const int32_t a = h[k], c = h[partner];
          if ((a > c) == up) {
            h[k] = c;
            h[partner] = a;
          }
        }
      }
      __syncthreads();
    }
  const int nz = s_nz, pad = hist_len - nz;
  for (int k = threadIdx.x; k < hist_len; k += 256) {   // sorted_hist - offset (sampler.py:135-138)
    const int off = k - pad;
    if (off > 0) h[k] -= off;
  }
  __syncthreads();
  const float span = (float)(num_items - nz);           // fp32 product, like float32 * int64 in torch
  for (int sidx = threadIdx.x; sidx < per_row; sidx += 256) {
    const int64_t e = b * per_row + sidx;
    const float u = torch_rand_element(pc, (uint64_t)e);
    const int64_t id0 = (int64_t)floorf(u * span) + 1;
    int lo = 0, hi = hist_len;                          // searchsorted(..., right=True): first k with h[k] > id0
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((int64_t)h[mid] <= id0) lo = mid + 1; else hi = mid;
    }
    out[e] = id0 + (lo - pad);
  }
}

constexpr int LUT_BATCH = 4;
static inline int grid_for(int64_t numel) {
  // one element per thread up to 64 K workgroups: the samplers are a chain of dependent loads per
  // element, so more threads in flight beat a grid-stride loop (0.23 -> 0.1x ms for 4 M ids)
  int64_t b = (numel + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}

}  // namespace rsa

using namespace rsa;

extern "C" int rsa_sample_uniform(int64_t* neg_ids, int64_t numel, int64_t low, int64_t high, uint64_t seed,
                                  uint64_t offset, uint32_t grid_threads, uint64_t elem_base, rsa_stream_t stream) {
  RSA_CHECK_ARG(numel >= 0, "rsa_sample_uniform: numel < 0");
  if (numel == 0) return RSA_OK;
  RSA_CHECK_ARG(neg_ids != nullptr, "rsa_sample_uniform: neg_ids is null");
  RSA_CHECK_ARG(high > low, "rsa_sample_uniform: empty range [%lld, %lld)", (long long)low, (long long)high);
  RSA_CHECK_ARG(grid_threads > 0 && (offset & 3) == 0, "rsa_sample_uniform: bad philox state");
  PhiloxCall pc{seed, offset >> 2, grid_threads, elem_base};
  hipLaunchKernelGGL(sample_uniform_kernel, dim3(grid_for(numel)), dim3(256), 0, (hipStream_t)stream, neg_ids, numel,
                     (uint64_t)(high - low), low, pc);
  RSA_CHECK_LAUNCH("rsa_sample_uniform");
  return RSA_OK;
}

static int check_popular(const char* fn, const float* table, const float* pop_prob, const int32_t* guide,
                         int64_t n_items, int32_t guide_log2, const float* lines, int32_t lines_log2) {
  RSA_CHECK_ARG(table && pop_prob, "%s: table/pop_prob is null", fn);
  RSA_CHECK_ARG(n_items >= 1 && n_items < (1ll << 31), "%s: n_items out of range", fn);
  if (lines != nullptr) {
    RSA_CHECK_ARG(lines_log2 >= 0 && lines_log2 <= 28 && ((uintptr_t)lines & 127) == 0,
                  "%s: cdf_lines must be 128-byte aligned with lines_log2 in [0, 28]", fn);
  } else {
    RSA_CHECK_ARG(guide != nullptr, "%s: guide is null", fn);
    RSA_CHECK_ARG(guide_log2 >= 0 && guide_log2 <= 28, "%s: guide_log2 must be in [0, 28]", fn);
  }
  return RSA_OK;
}

extern "C" int rsa_sample_popular(const rsa_popular_args* args, rsa_stream_t stream) {
  rsa_popular_args a;
  if (int rc = load_args(a, args, "rsa_sample_popular")) return rc;
  return sample_popular_impl(a.table, a.pop_prob, a.guide, a.n_items, a.guide_log2, a.ids, a.logp, a.u_out, a.numel, a.seed,
                             a.offset, a.grid_threads, a.elem_base, a.cdf_lut, a.cdf_lines, a.lines_log2, stream);
}

int rsa::sample_popular_impl(const float* table, const float* pop_prob, const int32_t* guide, int64_t n_items,
                             int32_t guide_log2, int64_t* neg_ids, float* neg_logp, float* u_out, int64_t numel,
                             uint64_t seed, uint64_t offset, uint32_t grid_threads, uint64_t elem_base,
                             const float* cdf_lut, const float* cdf_lines, int32_t lines_log2, rsa_stream_t stream) {
  RSA_CHECK_ARG(numel >= 0, "rsa_sample_popular: numel < 0");
  if (numel == 0) return RSA_OK;
  if (int rc = check_popular("rsa_sample_popular", table, pop_prob, guide, n_items, guide_log2, cdf_lines, lines_log2)) return rc;
  RSA_CHECK_ARG(neg_ids != nullptr, "rsa_sample_popular: neg_ids is null");
  RSA_CHECK_ARG(grid_threads > 0 && (offset & 3) == 0, "rsa_sample_popular: bad philox state");
  PhiloxCall pc{seed, offset >> 2, grid_threads, elem_base};
  if (cdf_lut != nullptr && cdf_lines == nullptr) {
    hipLaunchKernelGGL((sample_popular_lut_kernel<false, LUT_BATCH>), dim3((unsigned)((numel + 256 * LUT_BATCH - 1) / (256 * LUT_BATCH))),
                       dim3(256), 0, (hipStream_t)stream, table, pop_prob, reinterpret_cast<const float4*>(cdf_lut),
                       n_items, guide_log2, (const float*)nullptr, neg_ids, neg_logp, u_out, numel, pc);
    RSA_CHECK_LAUNCH("rsa_sample_popular");
    return RSA_OK;
  }
  hipLaunchKernelGGL(sample_popular_kernel<false>, dim3(grid_for(numel)), dim3(256), 0, (hipStream_t)stream, table,
                     pop_prob, guide, cdf_lut, cdf_lines, (int)lines_log2, n_items, guide_log2, (const float*)nullptr,
                     neg_ids, neg_logp, u_out, numel, pc);
  RSA_CHECK_LAUNCH("rsa_sample_popular");
  return RSA_OK;
}

static int popular_lookup_impl(const float* table, const float* pop_prob, const int32_t* guide, int64_t n_items,
                               int32_t guide_log2, const float* u, int64_t* ids, float* logp, int64_t numel,
                               const float* cdf_lut, const float* cdf_lines, int32_t lines_log2, rsa_stream_t stream);

extern "C" int rsa_popular_lookup(const rsa_popular_args* args, rsa_stream_t stream) {
  rsa_popular_args a;
  if (int rc = load_args(a, args, "rsa_popular_lookup")) return rc;
  return popular_lookup_impl(a.table, a.pop_prob, a.guide, a.n_items, a.guide_log2, a.u_in, a.ids, a.logp, a.numel, a.cdf_lut,
                             a.cdf_lines, a.lines_log2, stream);
}

static int popular_lookup_impl(const float* table, const float* pop_prob, const int32_t* guide, int64_t n_items,
                               int32_t guide_log2, const float* u, int64_t* ids, float* logp, int64_t numel,
                               const float* cdf_lut, const float* cdf_lines, int32_t lines_log2, rsa_stream_t stream) {
  RSA_CHECK_ARG(numel >= 0, "rsa_popular_lookup: numel < 0");
  if (numel == 0) return RSA_OK;
  if (int rc = check_popular("rsa_popular_lookup", table, pop_prob, guide, n_items, guide_log2, cdf_lines, lines_log2)) return rc;
  RSA_CHECK_ARG(u && ids, "rsa_popular_lookup: u/ids is null");
  PhiloxCall pc{0, 0, 1, 0};
  if (cdf_lut != nullptr && cdf_lines == nullptr) {
    hipLaunchKernelGGL((sample_popular_lut_kernel<true, LUT_BATCH>), dim3((unsigned)((numel + 256 * LUT_BATCH - 1) / (256 * LUT_BATCH))),
                       dim3(256), 0, (hipStream_t)stream, table, pop_prob, reinterpret_cast<const float4*>(cdf_lut),
                       n_items, guide_log2, u, ids, logp, (float*)nullptr, numel, pc);
    RSA_CHECK_LAUNCH("rsa_popular_lookup");
    return RSA_OK;
  }
  hipLaunchKernelGGL(sample_popular_kernel<true>, dim3(grid_for(numel)), dim3(256), 0, (hipStream_t)stream, table,
                     pop_prob, guide, cdf_lut, cdf_lines, (int)lines_log2, n_items, guide_log2, u, ids, logp,
                     (float*)nullptr, numel, pc);
  RSA_CHECK_LAUNCH("rsa_popular_lookup");
  return RSA_OK;
}

extern "C" int rsa_item_logp(const float* pop_prob, int64_t n_items, const int64_t* ids, int64_t numel, float* logp,
                             rsa_stream_t stream) {
  RSA_CHECK_ARG(numel >= 0, "rsa_item_logp: numel < 0");
  if (numel == 0) return RSA_OK;
  RSA_CHECK_ARG(pop_prob && ids && logp && n_items >= 1, "rsa_item_logp: null pointer / empty table");
  hipLaunchKernelGGL(item_logp_kernel, dim3(grid_for(numel)), dim3(256), 0, (hipStream_t)stream, pop_prob, n_items,
                     ids, numel, logp);
  RSA_CHECK_LAUNCH("rsa_item_logp");
  return RSA_OK;
}

extern "C" int rsa_sample_masked_uniform(const int64_t* user_hist, int64_t n_rows, int32_t hist_len, int64_t num_items,
                                         int32_t per_row, int64_t* neg_ids, uint64_t seed, uint64_t offset,
                                         uint32_t grid_threads, uint64_t elem_base, rsa_stream_t stream) {
  RSA_CHECK_ARG(n_rows >= 0 && per_row >= 0 && num_items >= 1, "rsa_sample_masked_uniform: bad sizes");
  if (n_rows == 0 || per_row == 0) return RSA_OK;
  RSA_CHECK_ARG(user_hist && neg_ids, "rsa_sample_masked_uniform: null pointer");
  RSA_CHECK_ARG(hist_len >= 1 && hist_len <= MASK_MAX_HIST, "rsa_sample_masked_uniform: hist_len must be in [1, %d]",
                MASK_MAX_HIST);
  RSA_CHECK_ARG(grid_threads > 0 && (offset & 3) == 0, "rsa_sample_masked_uniform: bad philox state");
  PhiloxCall pc{seed, offset >> 2, grid_threads, elem_base};
  hipLaunchKernelGGL(sample_masked_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, user_hist,
                     (int)hist_len, num_items, (int)per_row, neg_ids, pc);
  RSA_CHECK_LAUNCH("rsa_sample_masked_uniform");
  return RSA_OK;
}
