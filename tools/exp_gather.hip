// Experiment (not part of the product library): ceiling of random 512-B row gathers on MI355X.
// Each wave reads rows ids[tile*64 .. +63] (32 lanes x 16 B per row, 2 rows / instruction) and
// accumulates a checksum.  Variants: loads in flight per lane (U), nontemporal loads, block size.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int U, bool NT>
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids,
                                                     int64_t numel, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, sub = lane & 31, gbase = lane - sub;
  const int64_t n_tiles = numel >> 6;
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t wstride = (int64_t)gridDim.x * (blockDim.x >> 6);
  float acc = 0.f;
  for (int64_t tile = wave0; tile < n_tiles; tile += wstride) {
    const int32_t id = ids[(tile << 6) + lane];
#pragma unroll
    for (int t0 = 0; t0 < 32; t0 += U) {
      float4 x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int32_t rid = __shfl(id, gbase + t0 + u, 64);
        const float4* p = reinterpret_cast<const float4*>(table + (size_t)rid * 128) + sub;
        if (NT) {
          x[u].x = __builtin_nontemporal_load(&p->x); x[u].y = __builtin_nontemporal_load(&p->y);
          x[u].z = __builtin_nontemporal_load(&p->z); x[u].w = __builtin_nontemporal_load(&p->w);
        } else {
          x[u] = *p;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc += x[u].x + x[u].y + x[u].z + x[u].w;
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

extern "C" int exp_gather(const float* table, const int32_t* ids, int64_t numel, int variant, int blocks, int threads,
                          float* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  dim3 g(blocks), b(threads);
  switch (variant) {
    case 0: hipLaunchKernelGGL((gather_kernel<4, false>), g, b, 0, s, table, ids, numel, out); break;
    case 1: hipLaunchKernelGGL((gather_kernel<8, false>), g, b, 0, s, table, ids, numel, out); break;
    case 2: hipLaunchKernelGGL((gather_kernel<16, false>), g, b, 0, s, table, ids, numel, out); break;
    case 3: hipLaunchKernelGGL((gather_kernel<32, false>), g, b, 0, s, table, ids, numel, out); break;
    case 4: hipLaunchKernelGGL((gather_kernel<8, true>), g, b, 0, s, table, ids, numel, out); break;
    case 5: hipLaunchKernelGGL((gather_kernel<16, true>), g, b, 0, s, table, ids, numel, out); break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// streaming read of the same number of bytes for reference
__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ src, int64_t n4, float* out) {
  float acc = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride * 4) {
    float4 a = src[i], b = i + stride < n4 ? src[i + stride] : a, c = i + 2 * stride < n4 ? src[i + 2 * stride] : a,
           d = i + 3 * stride < n4 ? src[i + 3 * stride] : a;
    acc += a.x + b.y + c.z + d.w;
  }
  if (acc == 123.456f) out[0] = acc;
}
extern "C" int exp_stream(const float* src, int64_t n4, int blocks, float* out, void* stream) {
  hipLaunchKernelGGL(stream_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)src, n4, out);
  return 0;
}
