// d loss / d item rows of the full softmax:  out[i][:] = sum_b probs[b][i] * query[b][:]   (i over the catalog, b over the batch)
//
// The last library GEMM of the full-softmax training step (BASELINE.json configs[4]): the reference gets it from autograd
// -- scorer.py:16 (`query @ items.T`) under loss_func.py:39-47, i.e. ATen's mm backward `probs.T @ query` -- and rounds 1-4 of
// this package called torch.matmul -> rocBLAS for it.  Item-stationary on the fp32 matrix cores:
//
//   * a workgroup of four waves owns 128 consecutive items, wave w the 32 items i0 + 32 w .. + 31, for ALL D columns: the
//     [32 x D] block of the result lives in D/32 accumulator tiles (64 VGPRs at D = 128) for the whole batch loop -- every
//     output row is written exactly once, no atomics, no split-K partials;
//   * v_mfma_f32_32x32x2_f32 with the items as rows: step t takes two batch rows b, b + 1; lane (j, h) supplies
//     A = probs[b + h][i0 + j] -- the half-wave reads 128 contiguous bytes of one probs row -- and, for column block c,
//     B = query[b + h][NC j + c]: output column block c of lane j is column NC j + c (NC = D/32), so a lane's NC operand
//     values are ONE contiguous 16-byte read of the query row and its NC results of an item are one 16-byte store;
//   * the query chunk of a step group (32 batch rows x D floats = 16 KB) is staged once per workgroup in LDS, double buffered
//     (it is shared by the four waves and by every workgroup: 1 MB at B = 2048, L2 resident); probs -- the 4 B N bytes the
//     kernel exists to stream -- goes straight to registers, a whole step group (16 dwords per lane) ahead of its use.
//
// flops 2 B N D; bytes 4 B N (probs once) + 4 N D (result once).  Bound: fp32 matrix peak (157.3 TFLOP/s) above B ~ 64.
#include "rsa_common.hpp"

namespace rsa {

using f32x16 = float __attribute__((ext_vector_type(16)));

#ifndef RSA_DX_KB
#define RSA_DX_KB 32
#endif
constexpr int DX_KB = RSA_DX_KB;     // batch rows per staged query chunk
constexpr int DX_STEPS = DX_KB / 2;  // MFMA steps per chunk (two batch rows each)

#ifndef RSA_DX_MIN_BLOCKS
#define RSA_DX_MIN_BLOCKS 2
#endif
#ifndef RSA_DX_RT
#define RSA_DX_RT 1              // 32-item row tiles per wave (2: the query operand read from LDS feeds twice as many MFMAs)
#endif
#ifndef RSA_DX_QAHEAD
#define RSA_DX_QAHEAD 2          // LDS reads of the query operand issued this many steps ahead of their MFMAs
#endif
template <int D>
__global__ __launch_bounds__(256, RSA_DX_MIN_BLOCKS) void probs_t_query_kernel(const float* __restrict__ probs, int64_t n_cols, int64_t ld,
                                                               const float* __restrict__ query, int64_t n_query,
                                                               float* __restrict__ out) {
  constexpr int NC = D / 32;
  __shared__ __attribute__((aligned(16))) float qs[2][DX_KB * D];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  constexpr int RT = RSA_DX_RT;
  const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * (32 * RT);      // this wave's first column of probs (item i0 + 1)
  const float* pcol[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int64_t col = i0 + 32 * rt + j;
    pcol[rt] = probs + (col < n_cols ? col : 0);
  }

  f32x16 acc[RT][NC];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[rt][c] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  // cooperative stage of a query chunk: DX_KB * D / 4 float4 over 256 threads
  constexpr int QLOADS = DX_KB * D / 4 / 256;
  static_assert(DX_KB * D / 4 % 256 == 0, "query chunk must split evenly over the workgroup");
  float4 qstage[QLOADS];
  // FULL chunks (all DX_KB batch rows exist) are fetched without a single bounds check: a predicated load is a branch, and a
  // branch in front of the MFMA chain is a merge point at which the wait-count pass takes the minimum over both paths -- step t
  // of a chunk then waits for the t-th load of the NEXT chunk issued just above it (a memory round trip exposed per trip:
  // 4.69 ms = 112 TFLOP/s at B = 2048, N = 1e6 in the first version).  The odd full chunk and the partial last chunk go
  // through the checked forms after the pipelined loop.
  auto q_fetch = [&](int64_t chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < QLOADS; ++f)
      qstage[f] = reinterpret_cast<const float4*>(query + (size_t)chunk * DX_KB * D)[f * 256 + tid];
  };
  auto q_fetch_checked = [&](int64_t chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < QLOADS; ++f) {
      const int idx = f * 256 + tid;
      const int64_t b = chunk * DX_KB + idx / (D / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b < n_query) v = reinterpret_cast<const float4*>(query + (size_t)chunk * DX_KB * D)[idx];
      qstage[f] = v;
    }
  };
  auto q_commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < QLOADS; ++f) reinterpret_cast<float4*>(qs[buf])[f * 256 + tid] = qstage[f];
  };
  float pa[RT][DX_STEPS], pb[RT][DX_STEPS];
  auto p_fetch = [&](int64_t chunk, float (&dst)[RT][DX_STEPS]) __attribute__((always_inline)) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const float* src = pcol[rt] + (size_t)(chunk * DX_KB + h) * ld;
#pragma unroll
      for (int t = 0; t < DX_STEPS; ++t) dst[rt][t] = src[(size_t)(2 * t) * ld];
    }
  };
  auto p_fetch_checked = [&](int64_t chunk, float (&dst)[RT][DX_STEPS]) __attribute__((always_inline)) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int t = 0; t < DX_STEPS; ++t) {
        const int64_t b = chunk * DX_KB + 2 * t + h;
        dst[rt][t] = b < n_query ? pcol[rt][(size_t)b * ld] : 0.f;      // (the query rows staged for b >= n_query are zero as well)
      }
  };
  auto q_read = [&](int buf, int t, float (&qv)[NC]) __attribute__((always_inline)) {
    const float* qrow = qs[buf] + (2 * t + h) * D + NC * j;
    if constexpr (NC == 4) {
      const float4 v = *reinterpret_cast<const float4*>(qrow);
      qv[0] = v.x; qv[1] = v.y; qv[2] = v.z; qv[3] = v.w;
    } else if constexpr (NC == 2) {
      const float2 v = *reinterpret_cast<const float2*>(qrow);
      qv[0] = v.x; qv[1] = v.y;
    } else {
      qv[0] = qrow[0];
    }
  };
  auto run_chunk = [&](int buf, const float (&p)[RT][DX_STEPS]) __attribute__((always_inline)) {
    constexpr int AH = RSA_DX_QAHEAD;
    float qv[AH + 1][NC];
#pragma unroll
    for (int t = 0; t < AH; ++t) q_read(buf, t, qv[t]);
#pragma unroll
    for (int t = 0; t < DX_STEPS; ++t) {
      if (t + AH < DX_STEPS) q_read(buf, t + AH, qv[(t + AH) % (AH + 1)]);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int c = 0; c < NC; ++c)
          acc[rt][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(p[rt][t], qv[t % (AH + 1)][c], acc[rt][c], 0, 0, 0);
    }
  };

  const int64_t n_full = n_query / DX_KB;          // chunks with all their batch rows
  const int64_t n_pipe = n_full & ~(int64_t)1;     // ... taken two per trip by the pipelined loop
  if (n_pipe > 0) {
    const int64_t last = n_pipe - 1;
    q_fetch(0);
    p_fetch(0, pa);
    q_commit(0);
    __syncthreads();
    for (int64_t chunk = 0; chunk < n_pipe; chunk += 2) {
      // even chunk: operands in (qs[0], pa); the next chunk's go to (qs[1], pb) under this chunk's MFMA chain.  Past the last
      // chunk the last one is fetched again (and never used): no conditional anywhere in the trip.
      q_fetch(chunk + 1);
      p_fetch(chunk + 1, pb);
      __builtin_amdgcn_sched_barrier(0);     // the loads stay at the top: the scheduler otherwise sinks them behind the MFMA chain
      run_chunk(0, pa);
      q_commit(1);
      __syncthreads();
      const int64_t nxt = chunk + 2 < last ? chunk + 2 : last;
      q_fetch(nxt);
      p_fetch(nxt, pa);
      __builtin_amdgcn_sched_barrier(0);
      run_chunk(1, pb);
      q_commit(0);
      __syncthreads();
    }
  }
  for (int64_t chunk = n_pipe; chunk * DX_KB < n_query; ++chunk) {      // at most one full and one partial chunk
    q_fetch_checked(chunk);
    p_fetch_checked(chunk, pa);
    __syncthreads();
    q_commit(0);
    __syncthreads();
    run_chunk(0, pa);
  }
  // acc[c][r] = out[item row(r, h)][NC j + c], row(r, h) = (r & 3) + 8 (r >> 2) + 4 h
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t item = i0 + 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (item < n_cols) {
        float* dst = out + (size_t)item * D + NC * j;
        if constexpr (NC == 4) {
          *reinterpret_cast<float4*>(dst) = make_float4(acc[rt][0][r], acc[rt][1][r], acc[rt][2][r], acc[rt][3][r]);
        } else if constexpr (NC == 2) {
          *reinterpret_cast<float2*>(dst) = make_float2(acc[rt][0][r], acc[rt][1][r]);
        } else {
          dst[0] = acc[rt][0][r];
        }
      }
    }
}

}  // namespace rsa

using namespace rsa;

extern "C" int rsa_probs_t_query(const float* probs, int64_t n_query, int64_t n_cols, int64_t ld, const float* query, int32_t dim,
                                 float* out, rsa_stream_t stream) {
  RSA_CHECK_ARG(n_query >= 0 && n_cols >= 0 && ld >= n_cols, "rsa_probs_t_query: bad sizes");
  if (n_cols == 0) return RSA_OK;
  RSA_CHECK_ARG(out != nullptr, "rsa_probs_t_query: out is null");
  hipStream_t s = (hipStream_t)stream;
  if (n_query == 0) {
    if (hipMemsetAsync(out, 0, (size_t)n_cols * dim * sizeof(float), s) != hipSuccess) {
      rsa::set_error("rsa_probs_t_query: memset failed");
      return RSA_ERR_HIP;
    }
    return RSA_OK;
  }
  RSA_CHECK_ARG(probs && query, "rsa_probs_t_query: null pointer");
  RSA_CHECK_ARG(((uintptr_t)query & 15) == 0 && ((uintptr_t)out & 15) == 0, "rsa_probs_t_query: query / out must be 16-byte aligned");
  const dim3 grid((unsigned)((n_cols + 128 * RSA_DX_RT - 1) / (128 * RSA_DX_RT))), block(256);
  switch (dim) {
    case 32: hipLaunchKernelGGL(probs_t_query_kernel<32>, grid, block, 0, s, probs, n_cols, ld, query, n_query, out); break;
    case 64: hipLaunchKernelGGL(probs_t_query_kernel<64>, grid, block, 0, s, probs, n_cols, ld, query, n_query, out); break;
    case 128: hipLaunchKernelGGL(probs_t_query_kernel<128>, grid, block, 0, s, probs, n_cols, ld, query, n_query, out); break;
    default:
      rsa::set_error("rsa_probs_t_query: dim=%d: built for dim in {32, 64, 128}", dim);
      return RSA_ERR_UNSUPPORTED;
  }
  RSA_CHECK_LAUNCH("rsa_probs_t_query");
  return RSA_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// d loss / d item rows of the full softmax WITHOUT the [B, N] matrix (VERDICT r5 missing #4; reference: autograd through
// loss_func.py:39-47 over scorer.py:16):
//     out[i][:] = sum_b  g_b * exp(<q_b, w_i> - lse_b) * q_b[:]
// The kernel above streams P from HBM (4 B N bytes written by the recompute pass, read back here: 16.5 GB per step at B = 2048,
// N = 1e6, and 8 GB of capacity).  Here the P tile never leaves the registers: per chunk of 32 batch rows a wave
//   1. recomputes S[32 batch rows x its 32 items] = Q_chunk W^T on the matrix cores -- the wave's 32 item rows live in D/2
//      VGPRs per lane for the whole kernel (lane (j, h) holds W[i0 + j][KH h .. KH h + KH - 1], the B operand of every step), the
//      A operand Q[b0 + j][KH h + s] is a conflict-free 16-byte LDS read per four steps of the staged query chunk;
//   2. turns its 16 accumulator values into P = g_b exp(S - lse_b) -- lane (j, h), register r: batch row row(r, h), item j;
//   3. feeds them straight back as the A operand of the second product out^T: step t takes the two batch rows row(t, 0),
//      row(t, 1) as the K index -- exactly the pairing in which the accumulator layout holds them, so nothing moves between
//      lanes (the trick of the query-stationary dQ pass, rsa_fullscore.hip, mirrored) -- against B = Q[row(t, h)][NC j + c]
//      from the same LDS chunk.
// flops 4 B N D (two products); HBM bytes 4 N D (items) + 4 N D (result) + the L2-resident query block per workgroup.
// Bound: fp32 matrix peak.  It costs a fifth GEMM per step (the dQ pass recomputes the same tile on its side: the two
// gradients reduce over different axes, and one of the two reductions would have to cross workgroups -- N/128 partial [B, D]
// blocks or B/128 partial [N, D] blocks -- if both were formed from one recompute).
#ifndef RSA_DW_MIN_BLOCKS
#define RSA_DW_MIN_BLOCKS 2
#endif
#ifndef RSA_DW_AHEAD
#define RSA_DW_AHEAD 2           // 16-byte LDS reads of the query operand issued this many reads ahead of their MFMAs
#endif
namespace rsa {

template <int D>
__global__ __launch_bounds__(256, RSA_DW_MIN_BLOCKS) void softmax_dw_kernel(const float* __restrict__ items, int64_t n_cols,
                                                                            const float* __restrict__ query, int64_t n_query,
                                                                            const float* __restrict__ lse,
                                                                            const float* __restrict__ row_scale,
                                                                            float* __restrict__ out) {
  constexpr int NC = D / 32;        // output column blocks per lane
  constexpr int KH = D / 2;         // k values per lane half
  constexpr int KB = 32;            // batch rows per chunk
  constexpr int LDQ = D + 4;        // padded LDS row stride: "lane j reads row j" is conflict-free for 16-byte reads
  __shared__ __attribute__((aligned(16))) float qs[2][KB * LDQ];
  __shared__ float2 ls[2][KB];      // {lse_b, g_b} of the chunk's rows ({0, 0} past the batch: P = 0)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * 32;        // this wave's first column (item row i0 + 1 of the table)

  // the wave's item rows: B operand of the first product, resident in registers
  float wv[KH];
  {
    const int64_t col = i0 + j < n_cols ? i0 + j : n_cols - 1;      // (a valid row; its results are not stored)
    const float4* src = reinterpret_cast<const float4*>(items + (size_t)col * D + KH * h);
#pragma unroll
    for (int u = 0; u < KH / 4; ++u) {
      const float4 v = src[u];
      wv[4 * u] = v.x; wv[4 * u + 1] = v.y; wv[4 * u + 2] = v.z; wv[4 * u + 3] = v.w;
    }
  }
  f32x16 acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  constexpr int QLOADS = KB * D / 4 / 256;
  static_assert(KB * D / 4 % 256 == 0, "query chunk must split evenly over the workgroup");
  float4 qstage[QLOADS];
  float2 lstage = make_float2(0.f, 0.f);
  auto fetch = [&](int64_t chunk) __attribute__((always_inline)) {      // a FULL chunk: no bounds checks (see the kernel above)
#pragma unroll
    for (int f = 0; f < QLOADS; ++f)
      qstage[f] = reinterpret_cast<const float4*>(query + (size_t)chunk * KB * D)[f * 256 + tid];
    const int64_t b = chunk * KB + (tid & 31);                            // (every wave's lanes 0..31 hold the same 32 pairs)
    lstage = make_float2(lse[b], row_scale ? row_scale[b] : 1.f);
  };
  auto fetch_checked = [&](int64_t chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < QLOADS; ++f) {
      const int idx = f * 256 + tid;
      const int64_t b = chunk * KB + idx / (D / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b < n_query) v = reinterpret_cast<const float4*>(query + (size_t)chunk * KB * D)[idx];
      qstage[f] = v;
    }
    const int64_t b = chunk * KB + (tid & 31);
    lstage = b < n_query ? make_float2(lse[b], row_scale ? row_scale[b] : 1.f) : make_float2(0.f, 0.f);
  };
  auto commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < QLOADS; ++f) {
      const int idx = f * 256 + tid;
      const int row = idx / (D / 4), c4 = idx - row * (D / 4);
      *reinterpret_cast<float4*>(qs[buf] + row * LDQ + c4 * 4) = qstage[f];
    }
    if (tid < KB) ls[buf][tid] = lstage;
  };
  auto run_chunk = [&](int buf) __attribute__((always_inline)) {
    constexpr int AH = RSA_DW_AHEAD;
    // ---- 1. S = Q_chunk W^T: KH steps, the A operand four steps per 16-byte LDS read
    f32x16 sacc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float* arow = qs[buf] + j * LDQ + KH * h;
    float4 qa[AH + 1];
#pragma unroll
    for (int u = 0; u < AH && u < KH / 4; ++u) qa[u] = *reinterpret_cast<const float4*>(arow + 4 * u);
#pragma unroll
    for (int u = 0; u < KH / 4; ++u) {
      if (u + AH < KH / 4) qa[(u + AH) % (AH + 1)] = *reinterpret_cast<const float4*>(arow + 4 * (u + AH));
      const float4 a = qa[u % (AH + 1)];
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wv[4 * u + 0], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wv[4 * u + 1], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wv[4 * u + 2], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wv[4 * u + 3], sacc, 0, 0, 0);
    }
    // ---- 2. P = g exp(S - lse): sacc[r] belongs to batch row row(r, h) = (r & 3) + 8 (r >> 2) + 4 h and item j
    float pv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float2 lg = ls[buf][(r & 3) + 8 * (r >> 2) + 4 * h];
      pv[r] = lg.y * __expf(sacc[r] - lg.x);
    }
    // ---- 3. out^T += P^T Q_chunk: step t = the two batch rows row(t, 0), row(t, 1)
    auto q_read = [&](int t, float (&qv)[NC]) __attribute__((always_inline)) {
      const float* qrow = qs[buf] + ((t & 3) + 8 * (t >> 2) + 4 * h) * LDQ + NC * j;
      if constexpr (NC == 4) {
        const float4 v = *reinterpret_cast<const float4*>(qrow);
        qv[0] = v.x; qv[1] = v.y; qv[2] = v.z; qv[3] = v.w;
      } else if constexpr (NC == 2) {
        const float2 v = *reinterpret_cast<const float2*>(qrow);
        qv[0] = v.x; qv[1] = v.y;
      } else {
        qv[0] = qrow[0];
      }
    };
    float qv[AH + 1][NC];
#pragma unroll
    for (int t = 0; t < AH; ++t) q_read(t, qv[t]);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      if (t + AH < 16) q_read(t + AH, qv[(t + AH) % (AH + 1)]);
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv[t], qv[t % (AH + 1)][c], acc[c], 0, 0, 0);
    }
  };

  const int64_t n_full = n_query / KB;             // chunks with all their batch rows
  const int64_t n_pipe = n_full & ~(int64_t)1;     // ... taken two per trip by the pipelined loop
  if (n_pipe > 0) {
    const int64_t last = n_pipe - 1;
    fetch(0);
    commit(0);
    __syncthreads();
    for (int64_t chunk = 0; chunk < n_pipe; chunk += 2) {
      fetch(chunk + 1);
      __builtin_amdgcn_sched_barrier(0);     // the loads stay at the top of the trip, under the MFMA chains
      run_chunk(0);
      commit(1);
      __syncthreads();
      fetch(chunk + 2 < last ? chunk + 2 : last);      // (past the end: the last chunk again, never used)
      __builtin_amdgcn_sched_barrier(0);
      run_chunk(1);
      commit(0);
      __syncthreads();
    }
  }
  for (int64_t chunk = n_pipe; chunk * KB < n_query; ++chunk) {      // at most one full and one partial chunk
    fetch_checked(chunk);
    __syncthreads();
    commit(0);
    __syncthreads();
    run_chunk(0);
  }
  // acc[c][r] = out[item row(r, h)][NC j + c]
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t item = i0 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (item < n_cols) {
      float* dst = out + (size_t)item * D + NC * j;
      if constexpr (NC == 4) {
        *reinterpret_cast<float4*>(dst) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
      } else if constexpr (NC == 2) {
        *reinterpret_cast<float2*>(dst) = make_float2(acc[0][r], acc[1][r]);
      } else {
        dst[0] = acc[0][r];
      }
    }
  }
}

}  // namespace rsa

extern "C" int rsa_fullscore_softmax_dw(const float* item_table, int64_t n_items, int32_t dim, const float* query, int64_t n_query,
                                        const float* lse, const float* row_scale, float* item_grad, rsa_stream_t stream) {
  RSA_CHECK_ARG(n_query >= 0 && n_items >= 2, "rsa_fullscore_softmax_dw: need n_items >= 2");
  RSA_CHECK_ARG(item_table && item_grad, "rsa_fullscore_softmax_dw: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_cols = n_items - 1;
  // row 0 (the padding row) takes no part in the softmax: its gradient is zero; so is everything for an empty batch
  if (hipMemsetAsync(item_grad, 0, (size_t)(n_query == 0 ? n_items : 1) * dim * sizeof(float), s) != hipSuccess) {
    rsa::set_error("rsa_fullscore_softmax_dw: memset failed");
    return RSA_ERR_HIP;
  }
  if (n_query == 0) return RSA_OK;
  RSA_CHECK_ARG(query && lse, "rsa_fullscore_softmax_dw: null pointer");
  RSA_CHECK_ARG(((uintptr_t)query & 15) == 0 && ((uintptr_t)item_table & 15) == 0 && ((uintptr_t)item_grad & 15) == 0,
                "rsa_fullscore_softmax_dw: item_table / query / item_grad must be 16-byte aligned");
  const dim3 grid((unsigned)((n_cols + 127) / 128)), block(256);
  const float* items = item_table + dim;      // rows 1 .. n_items - 1
  float* out = item_grad + dim;
  switch (dim) {
    case 32: hipLaunchKernelGGL(softmax_dw_kernel<32>, grid, block, 0, s, items, n_cols, query, n_query, lse, row_scale, out); break;
    case 64: hipLaunchKernelGGL(softmax_dw_kernel<64>, grid, block, 0, s, items, n_cols, query, n_query, lse, row_scale, out); break;
    case 128: hipLaunchKernelGGL(softmax_dw_kernel<128>, grid, block, 0, s, items, n_cols, query, n_query, lse, row_scale, out); break;
    default:
      rsa::set_error("rsa_fullscore_softmax_dw: dim=%d: built for dim in {32, 64, 128}", dim);
      return RSA_ERR_UNSUPPORTED;
  }
  RSA_CHECK_LAUNCH("rsa_fullscore_softmax_dw");
  return RSA_OK;
}
