// Experiment (not part of the product library): what do the forward's OUTPUT writes cost on top of the gather + look-up
// microbenchmark (tools/exp_gran.hip), and does their shape matter?  A wave gathers the 64 rows of its tile behind one
// random 128-byte look-up per lane (mode 6 of exp_gran), then writes per-element outputs the way fused_fwd_kernel does.
//   WR = 0  no writes                         (== exp_gran mode 6)
//   WR = 1  per element: int64 id + 3 floats  (the headline's 20 B / element: 512 + 3 x 256 bytes per tile, streaming stores)
//   WR = 2  WR 1 + 4 per-QUERY floats         (4-byte stores to four arrays per tile: the headline's per-query outputs)
//   WR = 3  only the 4 per-query floats
//   WR = 4  WR 2 with the per-element outputs staged in LDS per workgroup and written as 4-tile bursts after a barrier
//   WR = 5  WR 1 with plain (cached) stores instead of streaming ones
//   WR = 7  WR 1 with a CONTIGUOUS chunk of tiles per wave (every concurrent wave writes into pages of its own)
//   WR = 8  WR 1 with the workgroups of one XCD (blockIdx % 8) covering one contiguous eighth of every window of tiles
//   WR = 6  WR 1, but a wave keeps the outputs of all its tiles (<= 8) in registers and writes them when it has read its last row
#include <hip/hip_runtime.h>
#include <stdint.h>

template <class T> __device__ __forceinline__ void st_stream(T* p, T v) { __builtin_nontemporal_store(v, p); }

template <int WR>
__global__ __launch_bounds__(256) void wr_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids,
                                                 const float* __restrict__ aux, const int32_t* __restrict__ aux_idx,
                                                 int64_t numel, int64_t* __restrict__ o_id, float* __restrict__ o_a,
                                                 float* __restrict__ o_b, float* __restrict__ o_c, float* __restrict__ q0,
                                                 float* __restrict__ q1, float* __restrict__ q2, float* __restrict__ q3,
                                                 float* __restrict__ out) {
  __shared__ int64_t s_id[4][64];
  __shared__ float s_f[3][4][64];
  const int lane = threadIdx.x & 63, sub = lane & 31, gbase = lane - sub, wave = threadIdx.x >> 6;
  const int64_t n_tiles = numel >> 6;
  int64_t bid = blockIdx.x;
  if (WR == 8) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);     // (gridDim.x % 8 == 0)
  int64_t wave0 = bid * 4 + wave;
  int64_t wstride = (int64_t)gridDim.x * 4;
  if (WR == 7) {
    const int64_t per = ((numel >> 6) + wstride - 1) / wstride;
    wave0 = wave0 * per;
    wstride = 1;
  }
  const int64_t t_last = WR == 7 ? wave0 + (((numel >> 6) + (int64_t)gridDim.x * 4 - 1) / ((int64_t)gridDim.x * 4)) : (int64_t)1 << 62;
  float acc = 0.f;
  int32_t k_id[8];
  float k_dot[8];
  int kt = 0;
  for (int64_t tile = wave0; tile < n_tiles + 3 && tile < t_last; tile += wstride) {     // (+3: whole workgroups iterate together for WR 4)
    const bool live = tile < n_tiles;
    const int64_t e = (tile << 6) + lane;
    int32_t id = 1;
    float dot = 0.f;
    if (live) {
      id = ids[e];
      const float* line = aux + (size_t)aux_idx[e] * 32;
      float4 a = *(const float4*)line; float4 b = *(const float4*)(line + 8);
      float4 c = *(const float4*)(line + 16); float4 d4 = *(const float4*)(line + 24);
      const float v = a.x + b.y + c.z + d4.w;
      id += (__float_as_int(v) & 1);
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
        for (int u = 0; u < 8; ++u) dot += x[u].x + x[u].y + x[u].z + x[u].w;
      }
      acc += dot;
    }
    if (WR == 1 || WR == 2 || WR == 7 || WR == 8) {
      if (live) {
        st_stream(&o_id[e], (int64_t)id);
        st_stream(&o_a[e], dot);
        st_stream(&o_b[e], dot * 2.f);
        st_stream(&o_c[e], dot * 3.f);
      }
    }
    if (WR == 6) {
      if (live) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k == kt) { k_id[k] = id; k_dot[k] = dot; }
        ++kt;
      }
    }
    if (WR == 5) {
      if (live) {
        o_id[e] = (int64_t)id;
        o_a[e] = dot;
        o_b[e] = dot * 2.f;
        o_c[e] = dot * 3.f;
      }
    }
    if (WR == 2 || WR == 3 || WR == 4) {
      if (live && lane == 0) {
        q0[tile] = dot;
        q1[tile] = dot + 1.f;
        q2[tile] = dot + 2.f;
        q3[tile] = dot + 3.f;
      }
    }
    if (WR == 4) {
      s_id[wave][lane] = (int64_t)id;
      s_f[0][wave][lane] = dot;
      s_f[1][wave][lane] = dot * 2.f;
      s_f[2][wave][lane] = dot * 3.f;
      __syncthreads();
      // the workgroup's 4 consecutive tiles: 2 KB of ids (one 8-byte store per thread), 1 KB per float array (wave w < 3
      // writes array w: 4 floats per lane)
      const int64_t t0 = tile - wave;                        // first tile of the workgroup's quad
      const int64_t e0 = (t0 << 6) + threadIdx.x;
      if (e0 < numel) st_stream(&o_id[e0], s_id[threadIdx.x >> 6][lane]);
      if (wave < 3) {
        float* dst = wave == 0 ? o_a : (wave == 1 ? o_b : o_c);
        const int64_t ef = (t0 << 6) + lane * 4;
        if (ef + 3 < numel) {
          typedef float v4f __attribute__((ext_vector_type(4)));
          const float* src = &s_f[wave][0][0] + lane * 4;
          v4f v = {src[0], src[1], src[2], src[3]};
          __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(dst + ef));
        }
      }
      __syncthreads();
    }
  }
  if (WR == 6) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < kt) {
        const int64_t e = ((wave0 + k * wstride) << 6) + lane;
        st_stream(&o_id[e], (int64_t)k_id[k]);
        st_stream(&o_a[e], k_dot[k]);
        st_stream(&o_b[e], k_dot[k] * 2.f);
        st_stream(&o_c[e], k_dot[k] * 3.f);
      }
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

#define CASE(M) case M: hipLaunchKernelGGL((wr_kernel<M>), g, b, 0, s, table, ids, aux, aux_idx, numel, o_id, o_a, o_b, o_c, q0, q1, q2, q3, out); break;
extern "C" int exp_wr(const float* table, const int32_t* ids, const float* aux, const int32_t* aux_idx, int64_t numel, int mode,
                      int blocks, int64_t* o_id, float* o_a, float* o_b, float* o_c, float* q0, float* q1, float* q2, float* q3,
                      float* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  dim3 g(blocks), b(256);
  switch (mode) {
    CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
