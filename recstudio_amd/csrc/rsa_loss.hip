// Pairwise losses on the [M, n] score tile: value and gradient in one pass (gfx950).
//
//   BPRLoss.forward            recstudio/model/loss_func.py:55-59   (dns=False)
//   SampledSoftmaxLoss.forward recstudio/model/loss_func.py:80-90   (one positive per row)
//
// RL lanes cooperate on one row (RL in {1, 4, 16, 64} chosen from n), each lane
// striding over the row's negatives; group shuffles finish the row reductions.
// Rows are reduced to the mean by a second, single-block, fixed-order kernel so the
// scalar is deterministic.
#include "rsa_common.hpp"

namespace rsa {

__device__ __forceinline__ float log_sigmoid(float x) {   // == F.logsigmoid
  return fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoid_neg(float x) {   // sigma(-x), stable
  if (x >= 0.f) {
    const float e = expf(-x);
    return e / (1.f + e);
  }
  return 1.f / (1.f + expf(x));
}

template <int RL>
__global__ __launch_bounds__(256) void pairwise_loss_kernel(int kind, const float* __restrict__ pos,
                                                            const float* __restrict__ neg,
                                                            const float* __restrict__ pos_lp,
                                                            const float* __restrict__ neg_lp, int64_t M, int n,
                                                            float* __restrict__ row_loss, float* __restrict__ dpos,
                                                            float* __restrict__ dneg) {
  const int sub = threadIdx.x % RL;
  const int64_t g0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / RL;
  const int64_t gstride = ((int64_t)gridDim.x * blockDim.x) / RL;
  const float inv_m = 1.f / (float)M;
  // every lane iterates ceil(M / gstride) times so that the group shuffles always see full waves
  const int64_t m_end = ((M + gstride - 1) / gstride) * gstride;
  for (int64_t m = g0; m < m_end; m += gstride) {
    const bool rv = m < M;
    const float* nrow = neg + (rv ? m : 0) * (int64_t)n;
    const float* lrow = neg_lp ? neg_lp + (rv ? m : 0) * (int64_t)n : nullptr;
    float* drow = dneg ? dneg + (rv ? m : 0) * (int64_t)n : nullptr;
    const float ps = rv ? pos[m] : 0.f;
    if (kind == RSA_LOSS_BPR) {
      const float w = 1.f / (float)n;   // softmax(ones) == 1/n (loss_func.py:57)
      float acc = 0.f, gsum = 0.f;
      if (rv)
        for (int j = sub; j < n; j += RL) {
          const float x = ps - nrow[j];
          acc += log_sigmoid(x) * w;
          const float s = sigmoid_neg(x) * w * inv_m;
          gsum += s;
          if (drow) drow[j] = s;
        }
      acc = group_sum<RL>(acc);
      gsum = group_sum<RL>(gsum);
      if (rv && sub == 0) {
        row_loss[m] = -acc;
        if (dpos) dpos[m] = -gsum;
      }
    } else {
      const float zp = ps - ((rv && pos_lp) ? pos_lp[m] : 0.f);
      float mx = zp;
      if (rv)
        for (int j = sub; j < n; j += RL) mx = fmaxf(mx, nrow[j] - (lrow ? lrow[j] : 0.f));
#pragma unroll
      for (int k = RL / 2; k >= 1; k >>= 1) mx = fmaxf(mx, __shfl_xor(mx, k, 64));
      float se = 0.f;
      if (rv)
        for (int j = sub; j < n; j += RL) se += expf(nrow[j] - (lrow ? lrow[j] : 0.f) - mx);
      se = group_sum<RL>(se) + expf(zp - mx);
      const float lse = mx + logf(se);
      const bool bad = isinf(zp);   // padded positive: the reference divides 0 by 0 (loss_func.py:88-89)
      if (rv) {
        if (drow)
          for (int j = sub; j < n; j += RL)
            drow[j] = bad ? NAN : expf(nrow[j] - (lrow ? lrow[j] : 0.f) - lse) * inv_m;
        if (sub == 0) {
          row_loss[m] = bad ? NAN : (lse - zp);
          if (dpos) dpos[m] = bad ? NAN : (expf(zp - lse) - 1.f) * inv_m;
        }
      }
    }
  }
}

// loss_out[0] = sum(row_loss) / M in a fixed order: up to 256 workgroups each sum one contiguous chunk
// (kernel 1), one workgroup sums the partials (kernel 2).
__global__ __launch_bounds__(256) void mean_rows_stage1(const float* __restrict__ row_loss, int64_t M, int64_t chunk,
                                                        float* __restrict__ partial) {
  __shared__ float part[4];
  const int64_t lo = (int64_t)blockIdx.x * chunk;
  int64_t hi = lo + chunk;
  if (hi > M) hi = M;
  float acc = 0.f;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) acc += row_loss[i];
  acc = group_sum<64>(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ __launch_bounds__(256) void mean_rows_stage2(const float* __restrict__ partial, int n_part, int64_t M,
                                                        const int32_t* __restrict__ denom, float* __restrict__ out) {
  __shared__ float part[4];
  float acc = (int)threadIdx.x < n_part ? partial[threadIdx.x] : 0.f;
  acc = group_sum<64>(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ((part[0] + part[1]) + (part[2] + part[3])) / (denom ? (float)denom[0] : (float)M);
}

// BinaryCrossEntropyLoss (loss_func.py:100-132, dns=False, one negative set per positive):
//   loss = -sum_valid logsigmoid(pos) / V + sum_valid (1/n) sum_j softplus(neg_j) / V,  V = #non-padded positives.
__global__ __launch_bounds__(256) void count_valid_kernel(const float* __restrict__ pos, int64_t M,
                                                          int32_t* __restrict__ count) {
  int c = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += stride) c += isinf(pos[i]) ? 0 : 1;
  c = (int)group_sum<64>((float)c);   // exact up to 2^24 per wave-iteration sum; counts here are tiny per lane
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

__device__ __forceinline__ float softplus_f(float x) {   // == F.softplus (beta 1, threshold 20)
  return x > 20.f ? x : log1pf(expf(x));
}

template <int RL>
__global__ __launch_bounds__(256) void bce_loss_kernel(const float* __restrict__ pos, const float* __restrict__ neg,
                                                       int64_t M, int n, const int32_t* __restrict__ count,
                                                       float* __restrict__ row_loss, float* __restrict__ dpos,
                                                       float* __restrict__ dneg) {
  const int sub = threadIdx.x % RL;
  const int64_t g0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / RL;
  const int64_t gstride = ((int64_t)gridDim.x * blockDim.x) / RL;
  const float inv_v = 1.f / (float)count[0];
  const float w = 1.f / (float)n;
  const int64_t m_end = ((M + gstride - 1) / gstride) * gstride;
  for (int64_t m = g0; m < m_end; m += gstride) {
    const bool rv = m < M;
    const float ps = rv ? pos[m] : 0.f;
    const bool valid = rv && !isinf(ps);
    const float* nrow = neg + (rv ? m : 0) * (int64_t)n;
    float acc = 0.f;
    if (rv)
      for (int j = sub; j < n; j += RL) {
        const float x = nrow[j];
        acc += softplus_f(x) * w;
        if (dneg) dneg[m * (int64_t)n + j] = valid ? (1.f / (1.f + expf(-x))) * w * inv_v : 0.f;
      }
    acc = group_sum<RL>(acc);
    if (rv && sub == 0) {
      row_loss[m] = valid ? (acc - log_sigmoid(ps)) : 0.f;
      if (dpos) dpos[m] = valid ? -sigmoid_neg(ps) * inv_v : 0.f;
    }
  }
}

// The other PairwiseLoss classes of recstudio/model/loss_func.py on the same [M, n] tile, value + gradient:
//   RSA_LOSS_WBPR  WeightedBPRLoss :93-97                 w = softmax(neg - logQ) is part of the graph
//   RSA_LOSS_WBCE  WeightedBinaryCrossEntropyLoss :135-137 (on :105-127, dns=False)
//   RSA_LOSS_HINGE HingeLoss :140-154 (num_items=None)     p0 = margin
//   RSA_LOSS_NCE   NCELoss :163-168
//   RSA_LOSS_CCL   CCLLoss :171-186                        p0 = margin, p1 = neg_weight
// RL lanes per row as above; the weighted losses need the row's softmax statistics first (max, sum of
// exponentials, weighted sums), hence up to three sweeps over the row's n scores (they stay in L1/L2).
template <int RL>
__global__ __launch_bounds__(256) void pairwise_loss_ex_kernel(int kind, const float* __restrict__ pos,
                                                               const float* __restrict__ neg,
                                                               const float* __restrict__ pos_lp,
                                                               const float* __restrict__ neg_lp, int64_t M, int n,
                                                               float p0, float p1, const int32_t* __restrict__ count,
                                                               float* __restrict__ row_loss, float* __restrict__ dpos,
                                                               float* __restrict__ dneg) {
  const int sub = threadIdx.x % RL;
  const int64_t g0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / RL;
  const int64_t gstride = ((int64_t)gridDim.x * blockDim.x) / RL;
  const float inv = kind == RSA_LOSS_WBCE ? 1.f / (float)count[0] : 1.f / (float)M;
  const int64_t m_end = ((M + gstride - 1) / gstride) * gstride;
  for (int64_t m = g0; m < m_end; m += gstride) {
    const bool rv = m < M;
    const float* nrow = neg + (rv ? m : 0) * (int64_t)n;
    const float* lrow = neg_lp ? neg_lp + (rv ? m : 0) * (int64_t)n : nullptr;
    float* drow = dneg ? dneg + (rv ? m : 0) * (int64_t)n : nullptr;
    const float ps = rv ? pos[m] : 0.f;
    float rl = 0.f, dp = 0.f;
    if (kind == RSA_LOSS_WBPR || kind == RSA_LOSS_WBCE) {
      const bool valid = kind == RSA_LOSS_WBPR || !isinf(ps);       // BCE: padded (-inf) positives drop out
      float mx = -INFINITY;
      if (rv)
        for (int j = sub; j < n; j += RL) mx = fmaxf(mx, nrow[j] - (lrow ? lrow[j] : 0.f));
#pragma unroll
      for (int k = RL / 2; k >= 1; k >>= 1) mx = fmaxf(mx, __shfl_xor(mx, k, 64));
      float se = 0.f, a = 0.f, gsum = 0.f;     // sum e, sum e * f_j, sum e * f'_j  (e = exp(z - mx))
      if (rv)
        for (int j = sub; j < n; j += RL) {
          const float e = expf(nrow[j] - (lrow ? lrow[j] : 0.f) - mx);
          se += e;
          if (kind == RSA_LOSS_WBPR) {
            a += e * log_sigmoid(ps - nrow[j]);
            gsum += e * sigmoid_neg(ps - nrow[j]);
          } else {
            a += e * softplus_f(nrow[j]);
          }
        }
      se = group_sum<RL>(se);
      a = group_sum<RL>(a);
      gsum = group_sum<RL>(gsum);
      const float bar = a / se;                 // sum_j w_j f_j
      if (kind == RSA_LOSS_WBPR) {
        rl = -bar;
        dp = -(gsum / se) * inv;
      } else {
        rl = valid ? bar - log_sigmoid(ps) : 0.f;
        dp = valid ? -sigmoid_neg(ps) * inv : 0.f;
      }
      if (rv && drow)
        for (int j = sub; j < n; j += RL) {
          const float w = expf(nrow[j] - (lrow ? lrow[j] : 0.f) - mx) / se;
          float g;
          if (kind == RSA_LOSS_WBPR)           // -(w f' + f dw): f = logsigmoid(pos - neg), df/dneg = -sigmoid(neg - pos)
            g = w * sigmoid_neg(ps - nrow[j]) - w * (log_sigmoid(ps - nrow[j]) - bar);
          else                                 // w f' + f dw:   f = softplus(neg), f' = sigmoid(neg)
            g = w * (1.f / (1.f + expf(-nrow[j]))) + w * (softplus_f(nrow[j]) - bar);
          drow[j] = valid ? g * inv : 0.f;
        }
    } else if (kind == RSA_LOSS_HINGE) {
      float mx = -INFINITY;
      int arg = 0x7fffffff;
      if (rv)
        for (int j = sub; j < n; j += RL)
          if (nrow[j] > mx) {
            mx = nrow[j];
            arg = j;
          }
#pragma unroll
      for (int k = RL / 2; k >= 1; k >>= 1) {            // (max, smallest index among equals)
        const float om = __shfl_xor(mx, k, 64);
        const int oa = __shfl_xor(arg, k, 64);
        if (om > mx || (om == mx && oa < arg)) {
          mx = om;
          arg = oa;
        }
      }
      const bool active = mx - ps + p0 > 0.f;
      rl = active ? mx - ps + p0 : 0.f;
      dp = active ? -inv : 0.f;
      if (rv && drow)
        for (int j = sub; j < n; j += RL) drow[j] = (active && j == arg) ? inv : 0.f;
    } else if (kind == RSA_LOSS_NCE) {
      const float zp = ps - ((rv && pos_lp) ? pos_lp[m] : 0.f);
      float acc = 0.f;
      if (rv)
        for (int j = sub; j < n; j += RL) {
          const float z = nrow[j] - (lrow ? lrow[j] : 0.f);
          acc += z - softplus_f(z);
          if (drow) drow[j] = -(z > 20.f ? 0.f : sigmoid_neg(z)) * inv;     // d/dz (z - softplus(z)), threshold 20
        }
      acc = group_sum<RL>(acc);
      rl = -(log_sigmoid(zp) + acc);
      dp = -sigmoid_neg(zp) * inv;
    } else {                                                               // RSA_LOSS_CCL
      const float pp = 1.f / (1.f + expf(-ps));
      const float wn = p1 / (float)n;
      float acc = 0.f;
      if (rv)
        for (int j = sub; j < n; j += RL) {
          const float q = 1.f / (1.f + expf(-nrow[j]));
          const bool on = q - p0 > 0.f;
          acc += on ? q - p0 : 0.f;
          if (drow) drow[j] = on ? wn * q * (1.f - q) * inv : 0.f;
        }
      acc = group_sum<RL>(acc);
      rl = (1.f - pp) + wn * acc;
      dp = -pp * (1.f - pp) * inv;
    }
    if (rv && sub == 0) {
      row_loss[m] = rl;
      if (dpos) dpos[m] = dp;
    }
  }
}

// SampledSoftmaxLoss with SEVERAL positives per row sharing one negative set -- loss_func.py:84-89 when
// pos_score [B, L] and neg_score [B, n] have the same rank: z = cat(pos - logQ_pos, neg - logQ_neg),
// out_l = logsumexp(z) - z_pos_l, padded positives (+-inf) contribute 0 and are not counted, row = sum / count.
// One wave per row.
__global__ __launch_bounds__(256) void ssm_shared_kernel(const float* __restrict__ pos, const float* __restrict__ pos_lp,
                                                         const float* __restrict__ neg, const float* __restrict__ neg_lp,
                                                         int64_t B, int L, int n, float* __restrict__ row_loss,
                                                         float* __restrict__ dpos, float* __restrict__ dneg) {
  const int lane = lane_id();
  const int64_t w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), ws = (int64_t)gridDim.x * 4;
  const float inv_b = 1.f / (float)B;
  for (int64_t b = w0; b < B; b += ws) {
    const float* pr = pos + b * L;
    const float* nr = neg + b * n;
    const float* plp = pos_lp ? pos_lp + b * L : nullptr;
    const float* nlp = neg_lp ? neg_lp + b * n : nullptr;
    float mx = -INFINITY;
    for (int l = lane; l < L; l += 64) mx = fmaxf(mx, pr[l] - (plp ? plp[l] : 0.f));
    for (int j = lane; j < n; j += 64) mx = fmaxf(mx, nr[j] - (nlp ? nlp[j] : 0.f));
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) mx = fmaxf(mx, __shfl_xor(mx, k, 64));
    float se = 0.f, zsum = 0.f, cnt = 0.f;
    for (int l = lane; l < L; l += 64) {
      const float z = pr[l] - (plp ? plp[l] : 0.f);
      se += expf(z - mx);
      if (!isinf(z)) {
        zsum += z;
        cnt += 1.f;
      }
    }
    for (int j = lane; j < n; j += 64) se += expf(nr[j] - (nlp ? nlp[j] : 0.f) - mx);
    se = group_sum<64>(se);
    zsum = group_sum<64>(zsum);
    cnt = group_sum<64>(cnt);
    const float lse = mx + logf(se);
    if (lane == 0) row_loss[b] = (cnt * lse - zsum) / cnt;     // 0/0 = NaN like the reference when all padded
    if (dpos)
      for (int l = lane; l < L; l += 64) {
        const float z = pr[l] - (plp ? plp[l] : 0.f);
        dpos[b * L + l] = (expf(z - lse) - (isinf(z) ? 0.f : 1.f / cnt)) * inv_b;
      }
    if (dneg)
      for (int j = lane; j < n; j += 64) dneg[b * n + j] = expf(nr[j] - (nlp ? nlp[j] : 0.f) - lse) * inv_b;
  }
}

// lse[m] = logsumexp(x[m, :]); optionally grad[m, :] = softmax(x[m, :]) * scale.  One workgroup per row.
__global__ __launch_bounds__(256) void row_lse_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ lse,
                                                      float* __restrict__ grad, float scale) {
  __shared__ float red[4];
  const float* row = x + (size_t)blockIdx.x * n;
  float m = -INFINITY;
  for (int64_t i = threadIdx.x; i < n; i += 256) m = fmaxf(m, row[i]);
#pragma unroll
  for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += expf(row[i] - m);
  s = group_sum<64>(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  s = (red[0] + red[1]) + (red[2] + red[3]);
  const float l = m + logf(s);
  if (threadIdx.x == 0) lse[blockIdx.x] = l;
  if (grad != nullptr) {
    float* g = grad + (size_t)blockIdx.x * n;
    for (int64_t i = threadIdx.x; i < n; i += 256) g[i] = expf(row[i] - l) * scale;
  }
}

}  // namespace rsa

using namespace rsa;

extern "C" int rsa_row_lse(const float* x, int64_t n_rows, int64_t n_cols, float* lse, float* softmax_scaled,
                           float scale, rsa_stream_t stream) {
  RSA_CHECK_ARG(n_rows >= 0 && n_cols >= 1, "rsa_row_lse: bad sizes");
  if (n_rows == 0) return RSA_OK;
  RSA_CHECK_ARG(x && lse, "rsa_row_lse: null pointer");
  hipLaunchKernelGGL(row_lse_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, x, n_cols, lse,
                     softmax_scaled, scale);
  RSA_CHECK_LAUNCH("rsa_row_lse");
  return RSA_OK;
}

// Deterministic mean of n_rows floats.  The <= 256 stage-1 partials live in the caller-owned scratch block
// (rsa_scratch_bytes(), include/recstudio_amd.h): calls sharing a block must be stream-ordered.
static int mean_rows_impl(const float* row_loss, int64_t n_rows, const int32_t* denom, float* out, void* scratch,
                          rsa_stream_t stream);

extern "C" int rsa_mean_rows(const float* row_loss, int64_t n_rows, float* out, void* scratch, rsa_stream_t stream) {
  return mean_rows_impl(row_loss, n_rows, nullptr, out, scratch, stream);
}

static int mean_rows_impl(const float* row_loss, int64_t n_rows, const int32_t* denom, float* out, void* scratch,
                          rsa_stream_t stream) {
  RSA_CHECK_ARG(row_loss && out && n_rows >= 1, "rsa_mean_rows: bad arguments");
  RSA_CHECK_ARG(scratch != nullptr, "rsa_mean_rows: scratch is null (rsa_scratch_bytes() bytes, zeroed once by the caller)");
  hipStream_t s = (hipStream_t)stream;
  float* g_partials = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + SCRATCH_MEAN_PARTIALS);
  int blocks = (int)((n_rows + 1023) / 1024);
  if (blocks > 256) blocks = 256;
  const int64_t chunk = (n_rows + blocks - 1) / blocks;
  hipLaunchKernelGGL(mean_rows_stage1, dim3(blocks), dim3(256), 0, s, row_loss, n_rows, chunk, g_partials);
  hipLaunchKernelGGL(mean_rows_stage2, dim3(1), dim3(256), 0, s, g_partials, blocks, n_rows, denom, out);
  RSA_CHECK_LAUNCH("rsa_mean_rows");
  return RSA_OK;
}

static int pairwise_loss_ex_impl(int32_t loss_kind, const float* pos_score, const float* neg_score,
                                 const float* pos_logp, const float* neg_logp, int64_t n_rows, int32_t num_neg,
                                 float param0, float param1, float* row_loss, float* loss_out, float* dpos,
                                 float* dneg, void* scratch, rsa_stream_t stream);
static int pairwise_loss_impl(int32_t loss_kind, const float* pos_score, const float* neg_score,
                              const float* pos_logp, const float* neg_logp, int64_t n_rows, int32_t num_neg,
                              float* row_loss, float* loss_out, float* dpos, float* dneg, void* scratch,
                              rsa_stream_t stream);

extern "C" int rsa_pairwise_loss(const rsa_loss_args* args, rsa_stream_t stream) {
  rsa_loss_args a;
  if (int rc = load_args(a, args, "rsa_pairwise_loss")) return rc;
  if (a.loss_kind >= RSA_LOSS_WBPR)
    return pairwise_loss_ex_impl(a.loss_kind, a.pos_score, a.neg_score, a.pos_logp, a.neg_logp, a.n_rows, a.num_neg, a.param0,
                                 a.param1, a.row_loss, a.loss_out, a.dpos, a.dneg, a.scratch, stream);
  return pairwise_loss_impl(a.loss_kind, a.pos_score, a.neg_score, a.pos_logp, a.neg_logp, a.n_rows, a.num_neg, a.row_loss,
                            a.loss_out, a.dpos, a.dneg, a.scratch, stream);
}

static int pairwise_loss_impl(int32_t loss_kind, const float* pos_score, const float* neg_score,
                              const float* pos_logp, const float* neg_logp, int64_t n_rows, int32_t num_neg,
                              float* row_loss, float* loss_out, float* dpos, float* dneg, void* scratch,
                              rsa_stream_t stream) {
  RSA_CHECK_ARG(scratch != nullptr, "rsa_pairwise_loss: scratch is null");
  int32_t* g_count = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(scratch) + SCRATCH_BCE_COUNT);
  RSA_CHECK_ARG(loss_kind == RSA_LOSS_BPR || loss_kind == RSA_LOSS_SSM || loss_kind == RSA_LOSS_BCE,
                "rsa_pairwise_loss: unknown loss %d", loss_kind);
  RSA_CHECK_ARG(n_rows >= 1 && num_neg >= 1, "rsa_pairwise_loss: need n_rows >= 1 and num_neg >= 1");
  RSA_CHECK_ARG(pos_score && neg_score && row_loss && loss_out, "rsa_pairwise_loss: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int rl = num_neg <= 2 ? 1 : num_neg <= 8 ? 4 : num_neg <= 32 ? 16 : 64;
  int64_t blocks = (n_rows * rl + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  dim3 grid((unsigned)blocks), block(256);
  if (loss_kind == RSA_LOSS_BCE) {
    if (hipMemsetAsync(g_count, 0, sizeof(int32_t), s) != hipSuccess) {
      rsa::set_error("rsa_pairwise_loss: memset failed");
      return RSA_ERR_HIP;
    }
    int64_t cb = (n_rows + 255) / 256;
    if (cb > 1024) cb = 1024;
    hipLaunchKernelGGL(count_valid_kernel, dim3((unsigned)cb), dim3(256), 0, s, pos_score, n_rows, g_count);
#define RSA_LAUNCH_BCE(RL) \
  hipLaunchKernelGGL(bce_loss_kernel<RL>, grid, block, 0, s, pos_score, neg_score, n_rows, (int)num_neg, g_count, row_loss, dpos, dneg)
    switch (rl) {
      case 1: RSA_LAUNCH_BCE(1); break;
      case 4: RSA_LAUNCH_BCE(4); break;
      case 16: RSA_LAUNCH_BCE(16); break;
      default: RSA_LAUNCH_BCE(64); break;
    }
#undef RSA_LAUNCH_BCE
    RSA_CHECK_LAUNCH("rsa_pairwise_loss(bce)");
    return mean_rows_impl(row_loss, n_rows, g_count, loss_out, scratch, stream);
  }
#define RSA_LAUNCH_LOSS(RL)                                                                                      \
  hipLaunchKernelGGL(pairwise_loss_kernel<RL>, grid, block, 0, s, (int)loss_kind, pos_score, neg_score, pos_logp, \
                     neg_logp, n_rows, (int)num_neg, row_loss, dpos, dneg)
  switch (rl) {
    case 1: RSA_LAUNCH_LOSS(1); break;
    case 4: RSA_LAUNCH_LOSS(4); break;
    case 16: RSA_LAUNCH_LOSS(16); break;
    default: RSA_LAUNCH_LOSS(64); break;
  }
#undef RSA_LAUNCH_LOSS
  RSA_CHECK_LAUNCH("rsa_pairwise_loss");
  return rsa_mean_rows(row_loss, n_rows, loss_out, scratch, stream);
}

static int pairwise_loss_ex_impl(int32_t loss_kind, const float* pos_score, const float* neg_score,
                                 const float* pos_logp, const float* neg_logp, int64_t n_rows, int32_t num_neg,
                                 float param0, float param1, float* row_loss, float* loss_out, float* dpos,
                                 float* dneg, void* scratch, rsa_stream_t stream) {
  RSA_CHECK_ARG(scratch != nullptr, "rsa_pairwise_loss: scratch is null");
  int32_t* g_count = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(scratch) + SCRATCH_BCE_COUNT);
  RSA_CHECK_ARG(loss_kind >= RSA_LOSS_WBPR && loss_kind <= RSA_LOSS_CCL, "rsa_pairwise_loss: unknown loss %d",
                loss_kind);
  RSA_CHECK_ARG(n_rows >= 1 && num_neg >= 1, "rsa_pairwise_loss: need n_rows >= 1 and num_neg >= 1");
  RSA_CHECK_ARG(pos_score && neg_score && row_loss && loss_out, "rsa_pairwise_loss: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int rl = num_neg <= 2 ? 1 : num_neg <= 8 ? 4 : num_neg <= 32 ? 16 : 64;
  int64_t blocks = (n_rows * rl + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  dim3 grid((unsigned)blocks), block(256);
  const int32_t* count = nullptr;
  if (loss_kind == RSA_LOSS_WBCE) {
    if (hipMemsetAsync(g_count, 0, sizeof(int32_t), s) != hipSuccess) {
      rsa::set_error("rsa_pairwise_loss: memset failed");
      return RSA_ERR_HIP;
    }
    int64_t cb = (n_rows + 255) / 256;
    if (cb > 1024) cb = 1024;
    hipLaunchKernelGGL(count_valid_kernel, dim3((unsigned)cb), dim3(256), 0, s, pos_score, n_rows, g_count);
    count = g_count;
  }
#define RSA_LAUNCH_EX(RL)                                                                                         \
  hipLaunchKernelGGL(pairwise_loss_ex_kernel<RL>, grid, block, 0, s, (int)loss_kind, pos_score, neg_score, pos_logp, \
                     neg_logp, n_rows, (int)num_neg, param0, param1, count, row_loss, dpos, dneg)
  switch (rl) {
    case 1: RSA_LAUNCH_EX(1); break;
    case 4: RSA_LAUNCH_EX(4); break;
    case 16: RSA_LAUNCH_EX(16); break;
    default: RSA_LAUNCH_EX(64); break;
  }
#undef RSA_LAUNCH_EX
  RSA_CHECK_LAUNCH("rsa_pairwise_loss");
  return mean_rows_impl(row_loss, n_rows, count, loss_out, scratch, stream);
}

extern "C" int rsa_ssm_shared_loss(const rsa_loss_args* args, rsa_stream_t stream) {
  rsa_loss_args a;
  if (int rc = load_args(a, args, "rsa_ssm_shared_loss")) return rc;
  const float *pos_score = a.pos_score, *pos_logp = a.pos_logp, *neg_score = a.neg_score, *neg_logp = a.neg_logp;
  const int64_t n_rows = a.n_rows;
  const int32_t n_pos = a.n_pos, num_neg = a.num_neg;
  float *row_loss = a.row_loss, *loss_out = a.loss_out, *dpos = a.dpos, *dneg = a.dneg;
  void* scratch = a.scratch;
  RSA_CHECK_ARG(n_rows >= 1 && n_pos >= 1 && num_neg >= 1, "rsa_ssm_shared_loss: bad sizes");
  RSA_CHECK_ARG(pos_score && neg_score && row_loss && loss_out, "rsa_ssm_shared_loss: null pointer");
  int64_t blocks = (n_rows + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(ssm_shared_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pos_score, pos_logp,
                     neg_score, neg_logp, n_rows, (int)n_pos, (int)num_neg, row_loss, dpos, dneg);
  RSA_CHECK_LAUNCH("rsa_ssm_shared_loss");
  return rsa_mean_rows(row_loss, n_rows, loss_out, scratch, stream);
}
