// Experiment (not part of the product library): cost of random 16-byte reads (the CDF look-up table
// probes of the popularity sampler) as a function of table size, alone and while 512-B row gathers
// stream through the caches.
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// one random float4 read per thread
__global__ __launch_bounds__(256) void lut_kernel(const float4* __restrict__ lut, uint32_t mask, int64_t numel,
                                                  uint32_t salt, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= numel) return;
  const float4 v = lut[mix((uint32_t)e ^ salt) & mask];
  if (v.x + v.y + v.z + v.w == 123.456f) out[0] = v.x;
}

// dependent chain: DEPTH random reads, each address derived from the previous value
template <int DEPTH>
__global__ __launch_bounds__(256) void chain_kernel(const float4* __restrict__ lut, uint32_t mask, int64_t numel,
                                                    uint32_t salt, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= numel) return;
  uint32_t a = mix((uint32_t)e ^ salt);
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < DEPTH; ++k) {
    const float4 v = lut[a & mask];
    acc += v.x;
    a = mix(a + __float_as_uint(v.y));
  }
  if (acc == 123.456f) out[0] = acc;
}

// sampler-shaped: one LUT read per element, then FRAC/256 of the lanes do a dependent second read in another
// table, then 12 bytes are written per element (E elements per thread, batched)
template <int E>
__global__ __launch_bounds__(256) void sampler_like_kernel(const float4* __restrict__ lut, uint32_t mask,
                                                           const float2* __restrict__ tab2, uint32_t mask2, int frac,
                                                           int64_t numel, uint32_t salt, int64_t* __restrict__ ids,
                                                           float* __restrict__ logp) {
  const int64_t base = (int64_t)blockIdx.x * (256 * E) + threadIdx.x;
  float4 v[E];
  uint32_t a[E];
#pragma unroll
  for (int k = 0; k < E; ++k) {
    a[k] = mix((uint32_t)(base + k * 256) ^ salt);
    v[k] = lut[a[k] & mask];
  }
  float2 w[E];
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const bool second = (int)((a[k] >> 24) & 255) < frac;
    w[k] = tab2[second ? (mix(a[k] + __float_as_uint(v[k].y)) & mask2) : 0];
  }
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const int64_t e = base + k * 256;
    if (e < numel) {
      ids[e] = (int64_t)(__float_as_uint(v[k].x) & 0xffffff);
      logp[e] = __logf(v[k].z + w[k].y + 1.0f);
    }
  }
}

extern "C" int exp_sampler_like(const float4* lut, uint32_t mask, const float2* tab2, uint32_t mask2, int frac, int e,
                                int64_t numel, uint32_t salt, int64_t* ids, float* logp, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  dim3 b(256);
  if (e == 1) hipLaunchKernelGGL(sampler_like_kernel<1>, dim3((unsigned)((numel + 255) / 256)), b, 0, s, lut, mask, tab2, mask2, frac, numel, salt, ids, logp);
  else hipLaunchKernelGGL(sampler_like_kernel<4>, dim3((unsigned)((numel + 1023) / 1024)), b, 0, s, lut, mask, tab2, mask2, frac, numel, salt, ids, logp);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int exp_lut(const float4* lut, uint32_t mask, int64_t numel, uint32_t salt, int depth, float* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  dim3 g((unsigned)((numel + 255) / 256)), b(256);
  switch (depth) {
    case 0: hipLaunchKernelGGL(lut_kernel, g, b, 0, s, lut, mask, numel, salt, out); break;
    case 2: hipLaunchKernelGGL(chain_kernel<2>, g, b, 0, s, lut, mask, numel, salt, out); break;
    case 3: hipLaunchKernelGGL(chain_kernel<3>, g, b, 0, s, lut, mask, numel, salt, out); break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
