// Experiment (not part of the product library): what does one dependent random look-up per gathered 512-B row cost,
// as a function of how many bytes / which 32-B sectors of its 128-B line are touched?  Emulates the popularity
// sampler in front of the row gather: each lane first reads from a random line of an auxiliary table, the value
// feeds the row id, then the wave gathers its 64 rows.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int MODE>
__global__ __launch_bounds__(256) void gran_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids,
                                                   const float* __restrict__ aux, const int32_t* __restrict__ aux_idx,
                                                   int64_t numel, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, sub = lane & 31, gbase = lane - sub;
  const int64_t n_tiles = numel >> 6;
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t wstride = (int64_t)gridDim.x * (blockDim.x >> 6);
  float acc = 0.f;
  for (int64_t tile = wave0; tile < n_tiles; tile += wstride) {
    int32_t id = ids[(tile << 6) + lane];
    if (MODE > 0) {
      const float* line = aux + (size_t)aux_idx[(tile << 6) + lane] * 32;
      float v = 0.f;
      if (MODE == 1) v = line[0];                                                      // 4 B
      if (MODE == 2) { float2 a = *(const float2*)line; v = a.x + a.y; }               // 8 B
      if (MODE == 3) { float4 a = *(const float4*)line; float4 b = *(const float4*)(line + 4); v = a.x + b.w; }  // one 32-B sector
      if (MODE == 4) { float4 a = *(const float4*)line; float4 b = *(const float4*)(line + 12); v = a.x + b.w; } // 64-B half
      if (MODE == 5) { float4 a = *(const float4*)line; float4 b = *(const float4*)(line + 16); v = a.x + b.w; } // both halves
      if (MODE == 6) { float4 a = *(const float4*)line; float4 b = *(const float4*)(line + 8);
                       float4 c = *(const float4*)(line + 16); float4 e = *(const float4*)(line + 24); v = a.x + b.y + c.z + e.w; }
      id += (__float_as_int(v) & 1);
    }
#pragma unroll
    for (int t0 = 0; t0 < 32; t0 += 8) {
      float4 x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int32_t rid = __shfl(id, gbase + t0 + u, 64);
        const float4* p = reinterpret_cast<const float4*>(table + (size_t)rid * 128) + sub;
        x[u].x = __builtin_nontemporal_load(&p->x); x[u].y = __builtin_nontemporal_load(&p->y);
        x[u].z = __builtin_nontemporal_load(&p->z); x[u].w = __builtin_nontemporal_load(&p->w);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += x[u].x + x[u].y + x[u].z + x[u].w;
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

// look-ups alone (no row gather): U independent random reads in flight per lane
template <int MODE>
__global__ __launch_bounds__(256) void lookup_kernel(const float* __restrict__ aux, const int32_t* __restrict__ aux_idx,
                                                     int64_t numel, float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i + 3 * stride < numel; i += 4 * stride) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* line = aux + (size_t)aux_idx[i + u * stride] * 32;
      if (MODE == 1) acc += line[0];
      if (MODE == 2) { float2 a = *(const float2*)line; acc += a.x + a.y; }
      if (MODE == 3) { float4 a = *(const float4*)line; float4 b = *(const float4*)(line + 4); acc += a.x + b.w; }
      if (MODE == 4) { float4 a = *(const float4*)line; float4 b = *(const float4*)(line + 12); acc += a.x + b.w; }
      if (MODE == 5) { float4 a = *(const float4*)line; float4 b = *(const float4*)(line + 16); acc += a.x + b.w; }
      if (MODE == 6) { float4 a = *(const float4*)line; float4 b = *(const float4*)(line + 8);
                       float4 c = *(const float4*)(line + 16); float4 e = *(const float4*)(line + 24); acc += a.x + b.y + c.z + e.w; }
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

#define CASE(K, M) case M: hipLaunchKernelGGL((K<M>), g, b, 0, s, table, ids, aux, aux_idx, numel, out); break;
extern "C" int exp_gran(const float* table, const int32_t* ids, const float* aux, const int32_t* aux_idx, int64_t numel,
                        int mode, int blocks, float* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  dim3 g(blocks), b(256);
  switch (mode) {
    CASE(gran_kernel, 0) CASE(gran_kernel, 1) CASE(gran_kernel, 2) CASE(gran_kernel, 3) CASE(gran_kernel, 4)
    CASE(gran_kernel, 5) CASE(gran_kernel, 6)
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
#define CASE2(M) case M: hipLaunchKernelGGL((lookup_kernel<M>), g, b, 0, s, aux, aux_idx, numel, out); break;
extern "C" int exp_lookup(const float* aux, const int32_t* aux_idx, int64_t numel, int mode, int blocks, float* out,
                          void* stream) {
  hipStream_t s = (hipStream_t)stream;
  dim3 g(blocks), b(256);
  switch (mode) {
    CASE2(1) CASE2(2) CASE2(3) CASE2(4) CASE2(5) CASE2(6)
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
