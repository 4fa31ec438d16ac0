// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of recstudio_amd.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/recstudio_amd.h"

namespace rsa {

constexpr int WAVE = 64;

// ---------------------------------------------------------------- error plumbing (host)
void set_error(const char* fmt, ...);
#define RSA_CHECK_ARG(cond, ...)                \
  do {                                          \
    if (!(cond)) {                              \
      rsa::set_error(__VA_ARGS__);              \
      return RSA_ERR_ARG;                       \
    }                                           \
  } while (0)
#define RSA_CHECK_LAUNCH(what)                                                   \
  do {                                                                           \
    hipError_t e_ = hipGetLastError();                                           \
    if (e_ != hipSuccess) {                                                      \
      rsa::set_error("%s: launch failed: %s", what, hipGetErrorString(e_));      \
      return RSA_ERR_HIP;                                                        \
    }                                                                            \
  } while (0)

// ---------------------------------------------------------------- Philox4x32-10
// Random123 Philox4x32 with 10 rounds: the generator behind torch's device
// distributions (rocRAND philox4x32_10_engine::ten_rounds).
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
    const uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += W0;
    k.y += W1;
  }
  return c;
}

// State of one torch distribution call, see include/recstudio_amd.h ("Philox state").
struct PhiloxCall {
  uint64_t seed;
  uint64_t offset4;        // generator offset / 4 (torch offsets are multiples of 4)
  uint32_t grid_threads;   // G
  uint64_t elem_base;      // element index of this launch's element 0 inside the (global) torch call
};

// ---------------------------------------------------------------- caller-owned reduction scratch
// Layout of the rsa_scratch_bytes() block (zero-filled once by the caller):
//   [0]     NaN / inf flag word of the in-kernel loss reduction (cleared by the kernel that read it)
//   [64]    valid-row counter of the BCE losses (reset by a memset before each use)
//   [256]   64-bit {arrivals, fixed-point loss sum} words of the in-kernel loss reduction: the top word, then 32
//           sub-words 128 bytes apart (self-resetting; 4.2 KB of the 16 KB reserved here)
//   [256 + 4 * SCRATCH_MAX_GRID]   256 stage-1 partials of rsa_mean_rows
constexpr int SCRATCH_MAX_GRID = 4096;
constexpr int64_t SCRATCH_COUNTER = 0, SCRATCH_BCE_COUNT = 64, SCRATCH_FUSED_PARTIALS = 256,
                  SCRATCH_MEAN_PARTIALS = 256 + 4 * SCRATCH_MAX_GRID, SCRATCH_BYTES = SCRATCH_MEAN_PARTIALS + 4 * 256;

// Raw philox draw for output element li of a call whose per-draw unroll is UNROLL
// (4: 32-bit ints and floats, 2: 64-bit ints).  Returns the 4 words and the
// component index the element owns.
template <int UNROLL>
__device__ __forceinline__ uint4 philox_for_element(const PhiloxCall& pc, uint64_t li, int& comp) {
  li += pc.elem_base;
  uint64_t idx, j;
  if (li < pc.grid_threads) {   // common case: every element on its own subsequence, first draw
    idx = li;
    j = 0;
  } else {
    j = li / pc.grid_threads;
    idx = li - j * pc.grid_threads;
  }
  const uint64_t k = j / UNROLL;
  comp = (int)(j - k * UNROLL);
  const uint64_t ctr = pc.offset4 + k;
  const uint4 c = make_uint4((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)idx, (uint32_t)(idx >> 32));
  const uint2 key = make_uint2((uint32_t)pc.seed, (uint32_t)(pc.seed >> 32));
  return philox4x32_10(c, key);
}

__device__ __forceinline__ uint32_t pick(const uint4& v, int comp) {
  return comp == 0 ? v.x : comp == 1 ? v.y : comp == 2 ? v.z : v.w;
}

// == element li of torch.randint(low, low+range, ..., device='cuda')
__device__ __forceinline__ int64_t torch_randint_element(const PhiloxCall& pc, uint64_t li, uint64_t range,
                                                         int64_t low) {
  int comp;
  if (range >= (1ull << 28)) {   // ATen random_from_to_kernel: 64-bit draws above 2^28
    const uint4 v = philox_for_element<2>(pc, li, comp);
    const uint64_t r = comp == 0 ? (((uint64_t)v.x << 32) | v.y) : (((uint64_t)v.z << 32) | v.w);
    return (int64_t)(r % range) + low;
  }
  const uint4 v = philox_for_element<4>(pc, li, comp);
  return (int64_t)((uint64_t)pick(v, comp) % range) + low;
}

// == element li of torch.rand(..., device='cuda', dtype=float32)
__device__ __forceinline__ float torch_rand_element(const PhiloxCall& pc, uint64_t li) {
  int comp;
  const uint4 v = philox_for_element<4>(pc, li, comp);
  const float inv = 2.3283064e-10f;                       // 2^-32 (rocRAND uniform_distribution)
  const float u = __fmaf_rn((float)pick(v, comp), inv, inv);   // product exact => == mul then add
  return u == 1.0f ? 0.0f : u;                            // torch uniform_kernel bound flip
}

// ---------------------------------------------------------------- inverse-CDF lookup
// First index i in [0, n_items) with table[i] >= u, found inside the cut-point
// bucket [guide[b], guide[b+1]] of b = floor(u * 2^guide_log2); clamped to n_items-1.
template <int STRIDE = 1>
__device__ __forceinline__ int32_t cdf_lower_bound(const float* __restrict__ table,
                                                   const int32_t* __restrict__ guide, int64_t n_items,
                                                   int guide_log2, float u) {
  const int32_t K = 1 << guide_log2;
  int32_t b = (int32_t)(u * (float)K);    // exact: power-of-two scale
  b = b < 0 ? 0 : (b > K - 1 ? K - 1 : b);
  int32_t lo = guide[b], hi = guide[b + 1];   // answer in [lo, hi]
  while (lo < hi) {
    const int32_t mid = lo + ((hi - lo) >> 1);
    if (table[(size_t)mid * STRIDE] < u) lo = mid + 1; else hi = mid;
  }
  return lo > (int32_t)(n_items - 1) ? (int32_t)(n_items - 1) : lo;
}

// Direct-lookup form of the same search.  One self-contained 16-byte entry per guide bucket b, with
// lo = guide[b], hi = guide[b+1] (so the answer lies in [lo, hi]):
//   x = lo, sign bit set when hi - lo >= 2 (bucket holds two or more CDF boundaries: binary search needed)
//   y = table[lo], or +inf when hi == lo          z = pop_prob[min(lo, N-1)]       w = pop_prob[min(lo+1, N-1)]
// Direct buckets (hi - lo <= 1, the common case with a fine guide) resolve id AND probability from this
// single read -- one HBM line per sampled id instead of three dependent round trips (guide -> table ->
// pop_prob): id = lo + (y < u).  Same comparisons as the binary search, same index.
constexpr uint32_t LUT_SEARCH_BIT = 0x80000000u;

__device__ __forceinline__ int32_t lut_bucket(int guide_log2, float u) {
  const int32_t K = 1 << guide_log2;
  const int32_t b = (int32_t)(u * (float)K);     // exact: power-of-two scale
  return b < 0 ? 0 : (b > K - 1 ? K - 1 : b);
}

// resolve a draw u given its (already loaded) bucket entry e0 = lut[lut_bucket(u)]
template <int STRIDE>
__device__ __forceinline__ int32_t cdf_resolve_lut(const float4 e0, const float4* __restrict__ lut,
                                                   const float* __restrict__ cdf, const float* __restrict__ prob,
                                                   int prob_stride, int64_t n_items, int guide_log2, float u,
                                                   float& pr) {
  const int32_t b = lut_bucket(guide_log2, u);
  const uint32_t x = __float_as_uint(e0.x);
  int32_t lo = (int32_t)(x & ~LUT_SEARCH_BIT);
  const int32_t last = (int32_t)(n_items - 1);
  if (!(x & LUT_SEARCH_BIT)) {
    const bool up = e0.y < u;
    pr = up ? e0.w : e0.z;
    const int32_t id = lo + (up ? 1 : 0);
    return id > last ? last : id;
  }
  // Bucket with >= 2 boundaries.  The answer is lo + #{i : cdf[lo+i] < u} (the CDF is non-decreasing, so the
  // predicate holds on a prefix): probe the next PROBE entries with loads issued together -- one more round
  // trip instead of a chain of dependent binary-search probes -- and only buckets with more than PROBE
  // boundaries below u (rare with ~1 item per bucket) continue with the binary search in [lo+PROBE, hi].
  constexpr int PROBE = 4;
  float c[PROBE], q[PROBE];
#pragma unroll
  for (int i = 0; i < PROBE; ++i) {
    const int32_t j = lo + i > last ? last : lo + i;
    c[i] = cdf[(size_t)j * STRIDE];
    q[i] = STRIDE == 2 ? cdf[(size_t)j * STRIDE + 1] : 0.f;     // interleaved {cdf, prob}: the same 8-byte load
  }
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < PROBE; ++i) cnt += (lo + i <= last && c[i] < u) ? 1 : 0;   // prefix count
  if (cnt < PROBE) {
    lo += cnt;
    lo = lo > last ? last : lo;
    if (STRIDE == 2 && prob == cdf + 1) {
      pr = cnt == 0 ? q[0] : cnt == 1 ? q[1] : cnt == 2 ? q[2] : q[3];
      if (lo == last) pr = prob[(size_t)last * prob_stride];    // clamped index: re-read (edge only)
    } else {
      pr = prob[(size_t)lo * prob_stride];
    }
    return lo;
  }
  lo += PROBE;
  int32_t hi = (int32_t)(__float_as_uint(lut[b + 1].x) & ~LUT_SEARCH_BIT);
  while (lo < hi) {
    const int32_t mid = lo + ((hi - lo) >> 1);
    if (cdf[(size_t)mid * STRIDE] < u) lo = mid + 1; else hi = mid;
  }
  lo = lo > last ? last : lo;
  pr = prob[(size_t)lo * prob_stride];
  return lo;
}

template <int STRIDE>
__device__ __forceinline__ int32_t cdf_lookup_lut(const float4* __restrict__ lut, const float* __restrict__ cdf,
                                                  const float* __restrict__ prob, int prob_stride, int64_t n_items,
                                                  int guide_log2, float u, float& pr) {
  return cdf_resolve_lut<STRIDE>(lut[lut_bucket(guide_log2, u)], lut, cdf, prob, prob_stride, n_items, guide_log2, u,
                                 pr);
}

// Bucket-line form (layout: include/recstudio_amd.h, rsa_fused_args.cdf_lines): one 128-byte line per bucket,
//   words 0..11  cdf[12]     the bucket's first distinct CDF values, then (when fewer than 12) the first entry ABOVE the
//                            bucket, then +inf
//   words 12..23 prob[12]    pop_prob of those entries
//   words 24..29 delta[12]   uint16 each: id = base + delta
//   word  30     base        id of slot 0
//   word  31     count       distinct values inside the bucket (may exceed 12); -1: ids too far apart for 16-bit deltas
// k = #{cdf[i] < u} is the answer's slot: the entry above the bucket compares >= u, so k never points past it.  Only
// k == 12 (more than 12 distinct values, all below u) and count == -1 fall back to a binary search of `table`
// between this line's and the next line's base.  Same comparisons as torch.searchsorted, same index; one HBM line.
constexpr int LINE_SLOTS = 12;
__device__ __forceinline__ int32_t lines_bucket(int lines_log2, float u) { return lut_bucket(lines_log2, u); }

__device__ __forceinline__ int32_t cdf_lookup_line(const float* __restrict__ lines, int lines_log2,
                                                   const float* __restrict__ cdf, int cdf_stride,
                                                   const float* __restrict__ prob, int prob_stride, int64_t n_items,
                                                   float u, float& pr) {
  const float* line = lines + (size_t)lines_bucket(lines_log2, u) * 32;
  const float4* l4 = reinterpret_cast<const float4*>(line);
  const float4 c0 = l4[0], c1 = l4[1], c2 = l4[2];                       // four loads of ONE line, issued together
  const int2 tail = *reinterpret_cast<const int2*>(line + 30);
  const int k = (c0.x < u) + (c0.y < u) + (c0.z < u) + (c0.w < u) + (c1.x < u) + (c1.y < u) + (c1.z < u) + (c1.w < u) +
                (c2.x < u) + (c2.y < u) + (c2.z < u) + (c2.w < u);
  if (tail.y >= 0 && k < LINE_SLOTS) {
    pr = line[12 + k];
    const uint32_t w = __float_as_uint(line[24 + (k >> 1)]);
    return tail.x + (int32_t)((k & 1) ? (w >> 16) : (w & 0xffffu));
  }
  // rare: > 12 distinct values in the bucket and u above the first 12, or a line whose ids do not fit 16-bit deltas
  int32_t lo = tail.x, hi = __float_as_int(line[32 + 30]);               // base of the next line (a sentinel line closes the table)
  const int32_t last = (int32_t)(n_items - 1);
  hi = hi > last ? last : hi;
  while (lo < hi) {
    const int32_t mid = lo + ((hi - lo) >> 1);
    if (cdf[(size_t)mid * cdf_stride] < u) lo = mid + 1; else hi = mid;
  }
  lo = lo > last ? last : lo;
  pr = prob[(size_t)lo * prob_stride];
  return lo;
}

// ---------------------------------------------------------------- wave helpers
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {   // all lanes of each WIDTH-group get the sum
#pragma unroll
  for (int m = WIDTH / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
  return __fmaf_rn(a.w, b.w, __fmaf_rn(a.z, b.z, __fmaf_rn(a.y, b.y, a.x * b.x)));
}

}  // namespace rsa
