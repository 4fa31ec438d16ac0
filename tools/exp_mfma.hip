// Experiment: sustained v_mfma_f32_32x32x2_f32 rate with no memory traffic (dependent chain per wave, like the
// full-score kernel), for 1..4 waves per SIMD and 1 or 2 independent chains per wave.
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters, float a0, float b0) {
  f32x16 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) acc[c] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 64 / CHAINS; ++k) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  if (s == 123.456f) out[0] = s;
}

// the same chain with 64 distinct B registers and 4 rotating A registers (the full-score kernel's operand pattern)
__global__ __launch_bounds__(256) void mfma_regs_kernel(float* out, int iters, const float* __restrict__ src) {
  f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  float bq[64];
#pragma unroll
  for (int k = 0; k < 64; ++k) bq[k] = src[k * 64 + (threadIdx.x & 63)];
  float4 a = reinterpret_cast<const float4*>(src)[threadIdx.x];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[4 * c + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[4 * c + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[4 * c + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[4 * c + 3], acc, 0, 0, 0);
    }
    a.x += 1e-9f;
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc[r];
  if (s == 123.456f) out[0] = s;
}

extern "C" int exp_mfma_regs(int blocks, int iters, float* out, const float* src, void* stream) {
  hipLaunchKernelGGL(mfma_regs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, src);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int exp_mfma(int chains, int blocks, int iters, float* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (chains == 1) hipLaunchKernelGGL(mfma_kernel<1>, dim3(blocks), dim3(256), 0, s, out, iters, 1.0f, 2.0f);
  else hipLaunchKernelGGL(mfma_kernel<2>, dim3(blocks), dim3(256), 0, s, out, iters, 1.0f, 2.0f);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
