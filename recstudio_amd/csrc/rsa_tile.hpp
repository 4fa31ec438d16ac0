// Row-fragment helpers shared by the tile kernels (rsa_fused.hip, rsa_owner.hip): a D-float row is read by LPR = D/4
// lanes as one 16-byte load each (D = 128: 32 lanes x 16 B, two rows per wave instruction, every 128-B line fully used).
#pragma once
#include "rsa_common.hpp"

namespace rsa {

// per-element outputs (ids, scores, log-probs, d loss/d score) are written once and consumed by a later kernel:
// streaming stores keep them from displacing table lines in L2
template <typename T>
__device__ __forceinline__ void st_out(T* p, T v) {
  __builtin_nontemporal_store(v, p);
}

template <int LPR, bool GENERIC>
struct Frag {
  static constexpr int CH = GENERIC ? 4 : 1;
  float4 v[CH];
};

// Unconditional 16-byte loads (a load under a lane predicate becomes a branch + vmcnt(0) per load
// and serialises the wave's row stream): rows that must not count are redirected to row 0 by the
// caller, generic-dim tails are clamped to the last in-range column and zeroed with a select.
__device__ __forceinline__ float4 load16(const float* p, bool nt) {
  if (nt) {   // streaming hint: rows of a table far larger than the 256 MB Infinity Cache are never re-read
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  return *reinterpret_cast<const float4*>(p);
}

template <int LPR, bool GENERIC, bool NT = false>
__device__ __forceinline__ void frag_load(Frag<LPR, GENERIC>& f, const float* __restrict__ row, int sub, int D) {
#pragma unroll
  for (int c = 0; c < Frag<LPR, GENERIC>::CH; ++c) {
    const int col = (c * LPR + sub) * 4;
    if constexpr (GENERIC) {
      const int cc = col < D ? col : D - 4;
      float4 v = load16(row + cc, NT);
      const float m = col < D ? 1.f : 0.f;
      f.v[c] = make_float4(v.x * m, v.y * m, v.z * m, v.w * m);
    } else {
      f.v[c] = load16(row + col, NT);
    }
  }
}

template <int LPR, bool GENERIC>
__device__ __forceinline__ float frag_dot(const Frag<LPR, GENERIC>& a, const Frag<LPR, GENERIC>& b) {
  float s = dot4(a.v[0], b.v[0]);
#pragma unroll
  for (int c = 1; c < Frag<LPR, GENERIC>::CH; ++c) s += dot4(a.v[c], b.v[c]);
  return s;
}

// d / d neg of  -w * inv_m * logsigmoid(pos - neg)  =  w * inv_m * sigmoid(neg - pos), written so that the
// in-loop (query gradient) and epilogue (dneg output) evaluations are the same float operations
__device__ __forceinline__ float bpr_dneg(float pos, float neg, float w, float inv_m) {
  const float xd = pos - neg;
  const float t = __expf(-fabsf(xd));
  const float r = __frcp_rn(1.f + t);
  return (xd >= 0.f ? t * r : r) * w * inv_m;
}

}  // namespace rsa
