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
