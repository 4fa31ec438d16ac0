// Backward of the fused forward for the inner-product scorer (gfx950, wave64).
//
// What autograd does in the reference at recommender.py:636-639 for
// pos_score = <q, item[pos]>, neg_score[j] = <q, item[neg_j]>:
//   item.grad[id] += dscore * q        (embedding_dense_backward; padding_idx row 0 skipped)
//   q.grad        += dscore * item[id]
// The reference materialises a dense [N, d] gradient; this kernel can produce that
// (atomics into a caller-zeroed table) and/or the row-sparse values (one d-vector
// per sampled id, no atomics) that a sparse optimiser consumes at N = 1e7..1e8.
//
// Two code paths:
//  * QU  (n % 64 == 0): one workgroup per query; each wave walks 64-element tiles of
//    that query's negatives, accumulates q.grad in registers (16-B row loads, same
//    layout as the forward), and issues the item-gradient atomics in a
//    dword-contiguous layout (lane s -> dwords s, s+LPR, ...), so that one wave
//    instruction covers whole 128-B lines instead of every 4th dword of four lines.
//  * general (any n): rows of a tile belong to different queries; q.grad is
//    accumulated with atomics into a zero-filled [M, d] buffer.
#include "rsa_common.hpp"


namespace rsa {

struct BwdParams {
  const float* item_table;
  const float* query;
  const int64_t* query_index;
  const int64_t* pos_ids;
  const int64_t* neg_ids;
  const float* dpos;
  const float* dneg;
  const float* upstream;
  float* item_grad;
  float* item_grad_rows;
  float* query_grad;
  float* query_table_grad;
  int64_t n_items, n_query_rows, n_queries;
  int32_t dim, num_neg, qpad, ipad, score_mode;
};

__device__ __forceinline__ int64_t clamp_id(int64_t id, int64_t n) { return id < 0 ? 0 : (id >= n ? n - 1 : id); }

// ------------------------------------------------------------------ general path
template <int LPR, bool GENERIC>
__device__ __forceinline__ void tile_bwd_general(const BwdParams& p, int D, int32_t id_lane, float d_lane,
                                                 int act_lane, int32_t qrow_lane, int32_t m_lane,
                                                 int64_t outrow_lane) {
  const int lane = lane_id();
  const int sub = lane % LPR;
  const int gbase = lane - sub;
  const int ndw = (D + LPR - 1) / LPR;   // dwords per lane (4 when D == 4*LPR)
  for (int t = 0; t < LPR; ++t) {
    const int r = gbase + t;
    const int32_t rid = __shfl(id_lane, r, 64);
    const float rd = __shfl(d_lane, r, 64);
    const int ract = __shfl(act_lane, r, 64);
    const int32_t qr = __shfl(qrow_lane, r, 64);
    const int32_t mr = __shfl(m_lane, r, 64);
    const int32_t orl = __shfl((int32_t)(outrow_lane & 0xffffffff), r, 64);
    const int32_t orh = __shfl((int32_t)(outrow_lane >> 32), r, 64);
    if (!ract) continue;
    const int64_t outrow = ((int64_t)orh << 32) | (uint32_t)orl;
    const float* irow = p.item_table + (size_t)rid * D;
    const float* qrow = p.query + (size_t)qr * D;
    for (int c = 0; c < ndw; ++c) {
      const int col = c * LPR + sub;
      if (GENERIC && col >= D) break;
      const float qa = qrow[col];
      const float xa = irow[col];
      const float gi = rid != p.ipad ? rd * qa : 0.f;
      if (p.item_grad && rid != p.ipad) atomicAdd(p.item_grad + (size_t)rid * D + col, gi);
      if (p.item_grad_rows) p.item_grad_rows[(size_t)outrow * D + col] = gi;
      if (p.query_grad) atomicAdd(p.query_grad + (size_t)mr * D + col, rd * xa);
      if (p.query_table_grad && qr != p.qpad) atomicAdd(p.query_table_grad + (size_t)qr * D + col, rd * xa);
    }
  }
}

template <int LPR, bool GENERIC>
__global__ __launch_bounds__(256) void bwd_general_kernel(const BwdParams p) {
  const int lane = lane_id();
  const int D = GENERIC ? p.dim : LPR * 4;
  const int64_t n = p.num_neg;
  const int64_t numel = p.n_queries * n;
  const int64_t n_tiles = (numel + 63) >> 6;
  const float up = p.upstream ? p.upstream[0] : 1.f;
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t wstride = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t tile = wave0; tile < n_tiles; tile += wstride) {
    const int64_t e = (tile << 6) + lane;
    const int act = e < numel;
    const int64_t m = act ? e / n : 0;
    const int64_t j = e - m * n;
    const int32_t id = act ? (int32_t)clamp_id(p.neg_ids[e], p.n_items) : 0;
    const float d = act ? p.dneg[e] * up : 0.f;
    const int32_t qrow = (int32_t)(p.query_index ? (act ? p.query_index[m] : 0) : m);
    tile_bwd_general<LPR, GENERIC>(p, D, id, d, act, qrow, (int32_t)m, m * (n + 1) + 1 + j);
    if (p.pos_ids) {
      const int owner = act && j == 0;
      if (__ballot(owner) != 0ull) {
        const int32_t pid = owner ? (int32_t)clamp_id(p.pos_ids[m], p.n_items) : 0;
        const float dp = owner ? p.dpos[m] * up : 0.f;
        tile_bwd_general<LPR, GENERIC>(p, D, pid, dp, owner, qrow, (int32_t)m, m * (n + 1));
      }
    }
  }
}

// ------------------------------------------------------------------ query-uniform path
template <int LPR, bool GENERIC>
__global__ __launch_bounds__(256) void bwd_qu_kernel(const BwdParams p) {
  constexpr int CH = GENERIC ? 4 : 1;
  constexpr int BATCH = GENERIC ? 2 : 8;
  __shared__ float4 red[4 * 64 * CH];
  const int lane = lane_id();
  const int sub = lane % LPR;
  const int gbase = lane - sub;
  const int wave = threadIdx.x >> 6;
  const int nwave = blockDim.x >> 6;
  const int D = GENERIC ? p.dim : LPR * 4;
  const int ndw = (D + LPR - 1) / LPR;
  const int64_t n = p.num_neg;
  const int64_t m = blockIdx.x;
  const float up = p.upstream ? p.upstream[0] : 1.f;
  const int64_t qrow = p.query_index ? p.query_index[m] : m;
  const float* qptr = p.query + (size_t)qrow * D;

  float4 qf[CH], qacc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int col = (c * LPR + sub) * 4;
    qf[c] = (!GENERIC || col < D) ? *reinterpret_cast<const float4*>(qptr + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    qacc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // the query gradient may already have been produced by the forward (rsa_fused_args.query_grad): then the
  // negative rows are not read at all and this kernel only writes dneg * q rows (or their atomics)
  const bool need_q = p.query_grad != nullptr || p.query_table_grad != nullptr;
  const int tiles = (int)(n >> 6);
  for (int tq = wave; tq < tiles; tq += nwave) {
    const int64_t j = ((int64_t)tq << 6) + lane;
    const int64_t e = m * n + j;
    const int32_t id = (int32_t)clamp_id(p.neg_ids[e], p.n_items);
    const float d = p.dneg[e] * up;
#pragma unroll
    for (int t0 = 0; t0 < LPR; t0 += BATCH) {
      float4 x[BATCH][CH];
      int32_t rid[BATCH];
      float rd[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int r = gbase + t0 + u;
        rid[u] = __shfl(id, r, 64);
        rd[u] = __shfl(d, r, 64);
        const float* irow = p.item_table + (size_t)rid[u] * D;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int col = (c * LPR + sub) * 4;
          x[u][c] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (need_q && (!GENERIC || col < D)) x[u][c] = *reinterpret_cast<const float4*>(irow + col);
        }
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          qacc[c].x = __fmaf_rn(rd[u], x[u][c].x, qacc[c].x);
          qacc[c].y = __fmaf_rn(rd[u], x[u][c].y, qacc[c].y);
          qacc[c].z = __fmaf_rn(rd[u], x[u][c].z, qacc[c].z);
          qacc[c].w = __fmaf_rn(rd[u], x[u][c].w, qacc[c].w);
        }
        const int r = gbase + t0 + u;
        if (p.item_grad_rows) {
          float* orow = p.item_grad_rows + (size_t)(m * (n + 1) + 1 + ((int64_t)tq << 6) + r) * D;
          const float s = rid[u] != p.ipad ? rd[u] : 0.f;
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const int col = (c * LPR + sub) * 4;
            if (!GENERIC || col < D) {
              typedef float v4f __attribute__((ext_vector_type(4)));
              v4f v = {s * qf[c].x, s * qf[c].y, s * qf[c].z, s * qf[c].w};
              __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(orow + col));   // written once, read by the optimizer later
            }
          }
        }
        if (p.item_grad && rid[u] != p.ipad) {
          float* grow = p.item_grad + (size_t)rid[u] * D;
          for (int c = 0; c < ndw; ++c) {
            const int col = c * LPR + sub;
            if (GENERIC && col >= D) break;
            atomicAdd(grow + col, rd[u] * qptr[col]);
          }
        }
      }
    }
  }

  // fold the lane groups of a wave (they hold the same columns), then the waves via LDS
#pragma unroll
  for (int c = 0; c < CH; ++c) {
#pragma unroll
    for (int k = LPR; k < 64; k <<= 1) {
      qacc[c].x += __shfl_xor(qacc[c].x, k, 64);
      qacc[c].y += __shfl_xor(qacc[c].y, k, 64);
      qacc[c].z += __shfl_xor(qacc[c].z, k, 64);
      qacc[c].w += __shfl_xor(qacc[c].w, k, 64);
    }
    red[(wave * CH + c) * 64 + lane] = qacc[c];
  }
  __syncthreads();
  if (wave == 0 && (need_q || p.pos_ids != nullptr)) {
    float dp = 0.f;
    int64_t pid = 0;
    if (p.pos_ids) {
      pid = clamp_id(p.pos_ids[m], p.n_items);
      dp = p.dpos[m] * up;
    }
    if (lane < LPR) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int col = (c * LPR + sub) * 4;
        if (GENERIC && col >= D) continue;
        float4 s = red[c * 64 + lane];
        for (int w = 1; w < nwave; ++w) {
          const float4 o = red[(w * CH + c) * 64 + lane];
          s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
        }
        if (p.pos_ids) {
          const float4 x = *reinterpret_cast<const float4*>(p.item_table + (size_t)pid * D + col);
          s.x = __fmaf_rn(dp, x.x, s.x); s.y = __fmaf_rn(dp, x.y, s.y);
          s.z = __fmaf_rn(dp, x.z, s.z); s.w = __fmaf_rn(dp, x.w, s.w);
          const float sp = pid != p.ipad ? dp : 0.f;
          if (p.item_grad_rows)
            *reinterpret_cast<float4*>(p.item_grad_rows + (size_t)(m * (n + 1)) * D + col) =
                make_float4(sp * qf[c].x, sp * qf[c].y, sp * qf[c].z, sp * qf[c].w);
          if (p.item_grad && pid != p.ipad) {
            float* grow = p.item_grad + (size_t)pid * D + col;
            atomicAdd(grow + 0, dp * qf[c].x); atomicAdd(grow + 1, dp * qf[c].y);
            atomicAdd(grow + 2, dp * qf[c].z); atomicAdd(grow + 3, dp * qf[c].w);
          }
        }
        if (p.query_grad) *reinterpret_cast<float4*>(p.query_grad + (size_t)m * D + col) = s;
        if (p.query_table_grad && qrow != p.qpad) {   // embedding_dense_backward of the query table, padding row skipped
          float* g = p.query_table_grad + (size_t)qrow * D + col;
          atomicAdd(g + 0, s.x); atomicAdd(g + 1, s.y); atomicAdd(g + 2, s.z); atomicAdd(g + 3, s.w);
        }
      }
    }
  }
}

// ------------------------------------------------------------------ cosine / Euclidean scorers
// score = <q, x> / (|x| |q|)  (scorer.py:19-25, no epsilon):
//   d score / d x = q / (|x||q|) - score * x / |x|^2        d score / d q = x / (|x||q|) - score * q / |q|^2
// One wave per (query, item) element, positives included as column 0; a plain kernel -- cosine training
// is not the path the byte model is built around.
__global__ __launch_bounds__(256) void bwd_cos_kernel(const BwdParams p) {
  const int lane = lane_id();
  const int D = p.dim;
  const int64_t n = p.num_neg, w = p.pos_ids ? n + 1 : n;
  const int64_t numel = p.n_queries * w;
  const float up = p.upstream ? p.upstream[0] : 1.f;
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t wstride = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t e = wave0; e < numel; e += wstride) {
    const int64_t m = e / w;
    const int c = (int)(e - m * w);
    const bool is_pos = p.pos_ids != nullptr && c == 0;
    const int64_t j = p.pos_ids ? c - 1 : c;
    const int64_t id = clamp_id(is_pos ? p.pos_ids[m] : p.neg_ids[m * n + j], p.n_items);
    const float d = (is_pos ? p.dpos[m] : p.dneg[m * n + j]) * up;
    const int64_t qrow = p.query_index ? p.query_index[m] : m;
    const float* x = p.item_table + (size_t)id * D;
    const float* q = p.query + (size_t)qrow * D;
    float dot = 0.f, nx2 = 0.f, nq2 = 0.f;
    for (int col = lane; col < D; col += 64) {
      const float xv = x[col], qv = q[col];
      dot = __fmaf_rn(xv, qv, dot);
      nx2 = __fmaf_rn(xv, xv, nx2);
      nq2 = __fmaf_rn(qv, qv, nq2);
    }
    dot = group_sum<64>(dot);
    nx2 = group_sum<64>(nx2);
    nq2 = group_sum<64>(nq2);
    const bool euc = p.score_mode == RSA_SCORE_EUC;   // score = -|x - q|^2: d/dx = 2(q - x), d/dq = 2(x - q)
    const float inv = 1.f / (sqrtf(nx2) * sqrtf(nq2));
    const float sc = dot * inv;
    const int64_t outrow = m * (n + 1) + (is_pos ? 0 : 1 + j);
    for (int col = lane; col < D; col += 64) {
      const float xv = x[col], qv = q[col];
      const float gx = id != p.ipad ? (euc ? 2.f * d * (qv - xv) : d * (qv * inv - sc * xv / nx2)) : 0.f;
      const float gq = euc ? 2.f * d * (xv - qv) : d * (xv * inv - sc * qv / nq2);
      if (p.item_grad && id != p.ipad) atomicAdd(p.item_grad + (size_t)id * D + col, gx);
      if (p.item_grad_rows) p.item_grad_rows[(size_t)outrow * D + col] = gx;
      if (p.query_grad) atomicAdd(p.query_grad + (size_t)m * D + col, gq);
      if (p.query_table_grad && qrow != p.qpad) atomicAdd(p.query_table_grad + (size_t)qrow * D + col, gq);
    }
  }
}

// ------------------------------------------------------------------ embedding_dense_backward
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ src,
                                                               const int64_t* __restrict__ ids, int64_t numel,
                                                               int D, float* __restrict__ dst, int64_t n_rows) {
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t wstride = (int64_t)gridDim.x * 4;
  const int lane = lane_id();
  for (int64_t i = wave0; i < numel; i += wstride) {
    const int64_t id = ids[i];
    if (id <= 0 || id >= n_rows) continue;   // padding_idx = 0 receives no gradient
    for (int col = lane; col < D; col += 64) atomicAdd(dst + (size_t)id * D + col, src[(size_t)i * D + col]);
  }
}

template <int LPR, bool GENERIC>
static int launch_bwd(const BwdParams& p, hipStream_t s) {
  const int64_t n = p.num_neg;
  if (n % 64 == 0) {
    const int threads = n >= 256 ? 256 : (int)n;
    hipLaunchKernelGGL((bwd_qu_kernel<LPR, GENERIC>), dim3((unsigned)p.n_queries), dim3(threads), 0, s, p);
  } else {
    if (p.query_grad) {
      hipError_t e = hipMemsetAsync(p.query_grad, 0, (size_t)p.n_queries * p.dim * sizeof(float), s);
      if (e != hipSuccess) {
        rsa::set_error("rsa_fused_backward: memset failed: %s", hipGetErrorString(e));
        return RSA_ERR_HIP;
      }
    }
    const int64_t n_tiles = (p.n_queries * n + 63) >> 6;
    int64_t blocks = (n_tiles + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((bwd_general_kernel<LPR, GENERIC>), dim3((unsigned)blocks), dim3(256), 0, s, p);
  }
  RSA_CHECK_LAUNCH("rsa_fused_backward");
  return RSA_OK;
}

}  // namespace rsa

using namespace rsa;

extern "C" int rsa_fused_backward(const rsa_backward_args* a, rsa_stream_t stream) {
  RSA_CHECK_ARG(a != nullptr, "rsa_fused_backward: args is null");
  RSA_CHECK_ARG(a->dim >= 4 && a->dim <= 1024 && a->dim % 4 == 0,
                "rsa_fused_backward: dim=%d must be a multiple of 4 in [4, 1024]", a->dim);
  RSA_CHECK_ARG(a->n_queries >= 0 && a->num_neg >= 1, "rsa_fused_backward: need num_neg >= 1");
  if (a->n_queries == 0) return RSA_OK;
  RSA_CHECK_ARG(a->n_queries < (1ll << 31) && a->n_items >= 2 && a->n_items < (1ll << 31) &&
                    a->n_query_rows >= 1 && a->n_query_rows < (1ll << 31),
                "rsa_fused_backward: sizes out of range");
  RSA_CHECK_ARG(a->item_table && a->query && a->neg_ids && a->dneg, "rsa_fused_backward: null input pointer");
  RSA_CHECK_ARG(a->pos_ids == nullptr || a->dpos != nullptr, "rsa_fused_backward: pos_ids without dpos");
  RSA_CHECK_ARG(a->item_grad || a->item_grad_rows || a->query_grad || a->query_table_grad,
                "rsa_fused_backward: no output requested");
  BwdParams p;
  p.item_table = a->item_table;
  p.query = a->query;
  p.query_index = a->query_index;
  p.pos_ids = a->pos_ids;
  p.neg_ids = a->neg_ids;
  p.dpos = a->dpos;
  p.dneg = a->dneg;
  p.upstream = a->upstream;
  p.item_grad = a->item_grad;
  p.item_grad_rows = a->item_grad_rows;
  p.query_grad = a->query_grad;
  p.query_table_grad = a->query_table_grad;
  p.n_items = a->n_items;
  p.n_query_rows = a->n_query_rows;
  p.n_queries = a->n_queries;
  p.dim = a->dim;
  p.num_neg = a->num_neg;
  p.qpad = a->query_table_pad_row;
  p.ipad = a->item_pad_row;
  hipStream_t s = (hipStream_t)stream;
  p.score_mode = a->score_mode;
  if (a->score_mode == RSA_SCORE_COS || a->score_mode == RSA_SCORE_EUC) {
    if (p.query_grad) {
      if (hipMemsetAsync(p.query_grad, 0, (size_t)p.n_queries * p.dim * sizeof(float), s) != hipSuccess) {
        rsa::set_error("rsa_fused_backward: memset failed");
        return RSA_ERR_HIP;
      }
    }
    const int64_t numel = p.n_queries * (p.num_neg + (p.pos_ids ? 1 : 0));
    int64_t blocks = (numel + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bwd_cos_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
    RSA_CHECK_LAUNCH("rsa_fused_backward(cosine)");
    return RSA_OK;
  }
  RSA_CHECK_ARG(a->score_mode == RSA_SCORE_IP, "rsa_fused_backward: unknown score_mode %d", a->score_mode);
  switch (a->dim) {
    case 32: return launch_bwd<8, false>(p, s);
    case 64: return launch_bwd<16, false>(p, s);
    case 128: return launch_bwd<32, false>(p, s);
    case 256: return launch_bwd<64, false>(p, s);
    default: return launch_bwd<64, true>(p, s);
  }
}

extern "C" int rsa_scatter_add_rows(const float* src, const int64_t* ids, int64_t numel, int32_t dim, float* dst,
                                    int64_t n_rows, rsa_stream_t stream) {
  RSA_CHECK_ARG(numel >= 0 && dim >= 1 && n_rows >= 1, "rsa_scatter_add_rows: bad sizes");
  if (numel == 0) return RSA_OK;
  RSA_CHECK_ARG(src && ids && dst, "rsa_scatter_add_rows: null pointer");
  int64_t blocks = (numel + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, ids,
                     numel, (int)dim, dst, n_rows);
  RSA_CHECK_LAUNCH("rsa_scatter_add_rows");
  return RSA_OK;
}
