// Routing kernels for a row-sharded item table (nothing comparable exists in
// the reference, whose only multi-device mode replicates the full tables, data_parallel.py:106-159).
//
// A rank samples negatives for its own queries, then every (query, item) element travels to the
// rank that owns the item row as one packed 64-bit key, is scored there, and its fp32 score comes
// back:  8 + 4 bytes per triplet over xGMI instead of a 512-byte row.
//
// Exact (variable-split) exchange:
//   rsa_shard_count   : histogram of owners                     (-> all-to-all split sizes)
//   rsa_shard_route   : counting-sort scatter into per-owner segments, key = qidx << 32 | local row
//   rsa_shard_unpack  : owner side, key -> (local row int64, query index int64)
//   rsa_scatter_f32   : home side, dst[pos[i]] = src[i]  (returned scores -> [pos_score | neg_score] buffer)
// Fixed-capacity exchange, version 2 (the default; second half of this file):
//   rsa_shard_sample_route / rsa_shard_score_segments (rsa_fused.hip) / rsa_shard_home / rsa_shard_unpack_segments
#include "rsa_common.hpp"

namespace rsa {

// x / d for 0 <= x < 2^32 by one 64-bit multiply-high: magic = floor(2^64 / d) + 1 (host side, d >= 2) is exact on
// that range; d == 1 is passed as magic == 0.  (The plain 64-bit divisions of the first version -- two per element
// and pass -- were a third of the routing kernel.)
struct FastDiv {
  uint64_t magic;
  __device__ __forceinline__ uint64_t div(uint64_t x) const { return magic ? __umul64hi(x, magic) : x; }
};
static inline FastDiv make_fastdiv(uint64_t d) {
  FastDiv f;
  f.magic = d <= 1 ? 0 : (~0ull / d) + 1;     // floor((2^64 - 1) / d) + 1 == floor(2^64 / d) + 1 unless d | 2^64 (then exact too)
  return f;
}

// x / d for 32-bit x by ONE 32-bit multiply-high and a correction: with magic = floor(2^32 / d) the estimate
// floor(x * magic / 2^32) is the quotient or one below it (x * magic / 2^32 > x / d - x / 2^32 > x / d - 1).  The
// routing kernel divides four times per element (query of an element, owner of an id, modulo of the uniform draw): the
// 64-bit forms (a 64 x 64 multiply-high is four quarter-rate multiplies, a 64-bit % a ~150-instruction subroutine)
// were most of its time.
struct Div32 {
  uint32_t d, magic;
  __device__ __forceinline__ uint32_t div(uint32_t x) const {
    const uint32_t q = __umulhi(x, magic);
    return (x - q * d >= d) ? q + 1 : q;
  }
  __device__ __forceinline__ uint32_t mod(uint32_t x) const { return x - div(x) * d; }
};
static inline Div32 make_div32(uint64_t d) {
  Div32 v;
  v.d = (uint32_t)d;
  v.magic = d <= 1 ? 0xffffffffu : (uint32_t)((1ull << 32) / d);
  return v;
}

struct RouteShape {
  const int64_t* pos_ids;
  const int64_t* neg_ids;
  int64_t n_queries;
  int n, G;
  int64_t rows_per_shard;    // 0: interleaved ownership (owner = id % G, row = id / G); by_rows then divides by G
  FastDiv by_width, by_rows;
};

// Owner and row inside the owner's table of item `id`.  Contiguous blocks: owner = id / rows_per_shard (clamped), row =
// the remainder.  Interleaved (rows_per_shard == 0 at the boundary): owner = id % G, row = id / G -- whatever the ids'
// popularity order, every owner gets 1/G of any id range.
template <class DIV>
__device__ __forceinline__ void owner_and_row(int64_t id, const DIV& by_rows, int64_t rows_per_shard, int G, int& g, int64_t& loc) {
  const int64_t q = id <= 0 ? 0 : (int64_t)by_rows.div(id);
  if (rows_per_shard == 0) {
    g = id <= 0 ? 0 : (int)(id - q * G);
    loc = q;
  } else {
    g = q >= G ? G - 1 : (int)q;
    loc = id - (int64_t)g * rows_per_shard;
    if (loc < 0) loc = 0;
  }
}

// element e of the [n_queries, 1 + n] (positive, negatives) layout -> (id, query m, column c, owner g)
__device__ __forceinline__ int64_t route_element(const RouteShape& sh, int64_t e, int64_t& m, int& c, int& g, int64_t& loc) {
  m = (int64_t)sh.by_width.div((uint64_t)e);
  c = (int)(e - m * (sh.n + 1));
  const int64_t id = c == 0 ? sh.pos_ids[m] : sh.neg_ids[m * sh.n + (c - 1)];
  owner_and_row(id, sh.by_rows, sh.rows_per_shard, sh.G, g, loc);
  return id;
}

// Per-owner counters in LDS.  With two or more owners: one LDS atomic per element (the LDS unit resolves the
// same-address conflicts of 64 lanes over a few counters faster than software can: 45 vs 73 us for 8 owners, 45 vs 50
// for 2).  With ONE owner the 64-way conflict dominates (59 us) and the wave aggregates first: lanes with the same owner
// are found by ballot, the first of them adds their number and every lane takes its rank among them (46 us).
template <bool WANT_SLOT>
__device__ __forceinline__ int32_t wave_count(int32_t* cnt, bool valid, int g, bool aggregate) {
  if (!aggregate) {
    if (!valid) return 0;
    if (WANT_SLOT) return atomicAdd(&cnt[g], 1);
    atomicAdd(&cnt[g], 1);
    return 0;
  }
  const int lane = threadIdx.x & 63;
  uint64_t todo = __ballot(valid);
  int32_t slot = 0;
  while (todo) {
    const int leader = __ffsll((unsigned long long)todo) - 1;
    const int gg = __shfl(g, leader, 64);
    const uint64_t same = __ballot(valid && g == gg);
    int32_t old = 0;
    if (lane == leader) old = atomicAdd(&cnt[gg], (int32_t)__popcll(same));
    if (WANT_SLOT) {
      old = __shfl(old, leader, 64);
      if (valid && g == gg) slot = old + (int32_t)__popcll(same & ((1ull << lane) - 1ull));
    }
    todo &= ~same;
  }
  return slot;
}

// Elements per thread of the routing kernels.  A thread first issues the loads of ALL its ids (independent, all in
// flight together) and only then counts: with one id per loop iteration the kernels were a chain of dependent
// HBM round trips (16 per pass, ~50 us for 4.2 M elements that take 6 us to read).
constexpr int ROUTE_EPT = 16;
constexpr int ROUTE_CHUNK = 256 * ROUTE_EPT;

__global__ __launch_bounds__(256) void shard_count_kernel(RouteShape sh, int32_t* __restrict__ counts) {
  __shared__ int32_t h[64];
  if (threadIdx.x < 64) h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t numel = sh.n_queries * (sh.n + 1);
  for (int64_t e_lo = (int64_t)blockIdx.x * ROUTE_CHUNK; e_lo < numel; e_lo += (int64_t)gridDim.x * ROUTE_CHUNK) {
    int g[ROUTE_EPT];
#pragma unroll
    for (int k = 0; k < ROUTE_EPT; ++k) {
      const int64_t e = e_lo + k * 256 + threadIdx.x;
      int64_t m;
      int c;
      g[k] = -1;
      int64_t loc;
      if (e < numel) route_element(sh, e, m, c, g[k], loc);
    }
#pragma unroll
    for (int k = 0; k < ROUTE_EPT; ++k) wave_count<false>(h, g[k] >= 0, g[k], sh.G == 1);
  }
  __syncthreads();
  if ((int)threadIdx.x < sh.G && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], h[threadIdx.x]);
}

// Counting-sort scatter (exact exchange).  Each workgroup owns a contiguous chunk of ROUTE_CHUNK elements, held in
// registers as (owner, local row): pass 1 counts the chunk's elements per owner in LDS, ONE global atomic per
// (workgroup, owner) reserves a contiguous slot range, pass 2 hands out the slots from LDS counters.  (A first version
// did one global atomic per wave and owner on the same few cursor words and spent 0.8 ms there for 4 M elements.)
// The cursors arrive holding the segment starts (exclusive prefix sum of the exact counts).
__global__ __launch_bounds__(256) void shard_route_kernel(RouteShape sh, int64_t query_base, int32_t* __restrict__ cursor,
                                                          int64_t* __restrict__ keys, int64_t* __restrict__ pos_out) {
  __shared__ int32_t cnt[64], base[64];
  const int64_t numel = sh.n_queries * (sh.n + 1);
  const int64_t e_lo = (int64_t)blockIdx.x * ROUTE_CHUNK;
  if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
  int g[ROUTE_EPT];
  uint32_t local[ROUTE_EPT];
#pragma unroll
  for (int k = 0; k < ROUTE_EPT; ++k) {
    const int64_t e = e_lo + k * 256 + threadIdx.x;
    int64_t m;
    int c;
    g[k] = -1;
    local[k] = 0;
    if (e < numel) {
      int64_t loc;
      route_element(sh, e, m, c, g[k], loc);
      local[k] = (uint32_t)loc;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ROUTE_EPT; ++k) wave_count<false>(cnt, g[k] >= 0, g[k], sh.G == 1);
  __syncthreads();
  if ((int)threadIdx.x < sh.G) {
    base[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], cnt[threadIdx.x]) : 0;
    cnt[threadIdx.x] = 0;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ROUTE_EPT; ++k) {
    const bool valid = g[k] >= 0;
    const int gk = valid ? g[k] : 0;
    const int64_t at = (int64_t)base[gk] + wave_count<true>(cnt, valid, gk, sh.G == 1);
    if (!valid) continue;
    const int64_t e = e_lo + k * 256 + threadIdx.x;
    const int64_t m = (int64_t)sh.by_width.div((uint64_t)e);
    const int c = (int)(e - m * (sh.n + 1));
    keys[at] = ((query_base + m) << 32) | (int64_t)local[k];
    // destination of this element's score in the home buffer [pos_score (n_queries) | neg_score (n_queries x n)]
    pos_out[at] = c == 0 ? m : sh.n_queries + m * sh.n + (c - 1);
  }
}

__global__ __launch_bounds__(256) void shard_unpack_kernel(const int64_t* __restrict__ keys, int64_t numel,
                                                           int64_t* __restrict__ local_rows,
                                                           int64_t* __restrict__ qidx) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) {
    const int64_t k = keys[i];
    local_rows[i] = k < 0 ? -1 : (k & 0xffffffffll);      // empty slot of the fixed-capacity exchange
    qidx[i] = k < 0 ? -1 : ((k >> 32) & 0x7fffffffll);
  }
}

__global__ __launch_bounds__(256) void scatter_f32_kernel(const float* __restrict__ src,
                                                          const int64_t* __restrict__ pos, int64_t numel,
                                                          float* __restrict__ dst) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) {
    const int64_t q = pos[i];
    if (q >= 0) dst[q] = src[i];
  }
}

__global__ __launch_bounds__(256) void gather_f32_kernel(const float* __restrict__ src,
                                                         const int64_t* __restrict__ pos, int64_t numel,
                                                         float* __restrict__ dst) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) {
    const int64_t q = pos[i];
    dst[i] = q >= 0 ? src[q] : 0.f;
  }
}


// =====================================================================================================================
// Version 2 of the fixed-capacity exchange (ABI v5): sampling fused into the routing pass, self-describing segments,
// 32-bit slots, one home-side kernel for scatter + loss + mean (+ d loss/d score in routed order).
//
// Send buffer of a rank: [n_slices][n_shards][n_banks] segments of stride = RSA_SHARD_HDR + capacity 8-byte words.  Word 0
// of a segment = number of keys in it (<= capacity), word 1 = elements of THIS STEP the source rank could not place
// anywhere (its dropped total over all segments; carried by bank 0 of every (slice, owner), 0 in the other banks), then
// the keys.  BANKS: what goes from one rank to one owner in one slice is split over n_banks segments, each filled by its
// own share of the workgroups through its own cursor -- one cursor per (slice, owner) made every workgroup of the
// launch queue up on one address (~10 ns per atomic: 10 us for 1028 workgroups at one owner).  The headers travel with the keys through the equal-split
// all-to-all, so every owner learns (a) how many slots of each received segment are live -- no -1 fill of the slack,
// tiles past the count are skipped -- and (b) the step's dropped totals of ALL sources: after the key exchange every
// rank holds the job-wide overflow count without a collective of its own.
//
// slot_of[e] (element order: query-major, column 0 = positive) = index of the element's key in the send buffer == index
// of its score in the returned score buffer (same geometry), or -1 when it was dropped.  The home kernel gathers through
// it; nothing is scattered, no 8-byte position list exists.

struct PopTables {
  const float* table;
  const float* pop_prob;
  const float* table_prob;
  const float* lut;
  const float* lines;
  const int32_t* guide;
  int64_t n_items;
  int32_t guide_log2, lines_log2;
};

// Which elements a workgroup routes is free (an element only needs SOME slot in its owner's segment), so the
// enumeration follows the random stream instead of the id matrix: torch's element li of a distribution call over
// grid_threads T draws component (li / T) % U of the Philox block with counter offset/4 + (li / T) / U on subsequence
// li % T (U = 4 outputs per block, 2 for 64-bit integers) -- the U elements li, li + T, ..., li + (U-1) T share ONE
// block.  A WORK ITEM is such a block: the thread evaluates Philox once (40 quarter-rate 32-bit multiplies -- 20 us of
// the first version's 48 were spent recomputing every block four times) and routes its U elements.  Consecutive
// threads own consecutive subsequences, i.e. consecutive elements of the id matrix (coalesced id / slot stores).
// Given ids use the same enumeration with a made-up T (numel / 4).  The positives are the work items behind the last
// block.  A pipelined SLICE is a contiguous range of workgroups (any partition of the elements will do: the home
// kernel gathers across all slices, the owner handles each slice's segments on their own).
constexpr int ROUTE_ITEMS = 4;                       // work items per thread
constexpr int ROUTE_ITEMS_PER_BLOCK = 256 * ROUTE_ITEMS;
constexpr int CURSOR_PAD = 32;                       // one 128-byte line per cursor: atomics on one line are served in turn
constexpr int ROUTE_CNT = 1024;                      // LDS counters: owners x queries of a workgroup (grouped routing)

struct RouteV2 {
  const int64_t* pos_ids;
  int64_t* neg_ids;
  float* neg_logp;
  float* pos_logp;
  int64_t* send;
  int32_t* slot_of;
  int32_t* cursors;        // [n_slices * G * n_banks] arrival cursors + [1 + TICKET_SUB] tickets, CURSOR_PAD ints apart: zero between launches
  int32_t* counts_out;
  int64_t n_queries, capacity, stride, rows_per_shard, query_base;
  int64_t n_neg;           // n_queries * n
  int64_t n_groups;        // work items that are Philox blocks (the positives follow)
  uint64_t k_lo;           // first Philox counter step (li / T / U) this rank's elements touch
  PhiloxCall pc;
  PopTables pop;
  Div32 by_width, by_rows, by_n, by_gt, by_range;
  int32_t n, G, sampler, n_slices, unroll, n_banks, skip_pos;
  int32_t group_ql;        // > 0: a workgroup's elements are whole queries (group_ql of them): its share of a segment is written query by query
  // DETERMINISTIC slots (rsa_shard_route_args.deterministic): no atomic decides a position.  Inside a workgroup an element's
  // rank in its (owner, query) share is (waves before it, in wave order) + (its wave's elements before it, in program order);
  // across workgroups the share bases are an exclusive prefix over the workgroups in launch order, from a COUNT pass
  // (det_phase 1: wg_cnt[workgroup][owner]) and a scan (shard_route_scan_kernel -> wg_base, segment totals into the cursors).
  int32_t* extra_dropped;  // nullable device word: elements of this step dropped OUTSIDE the routing (a tower's fixed-capacity row
                           // look-up): added to the dropped total of the headers -- the step is gated like any other overflow -- and reset
  int32_t det_phase;       // 0: atomic cursors; 1: count pass; 2: routing pass over wg_base
  int32_t* wg_cnt;         // [gridDim.x, G]
  const int32_t* wg_base;  // [gridDim.x, G]
};

// same-key lanes of a wave found by ballot; the wave's OWN counter row is read-modify-written by one lane in program order
// (no other wave touches it): the returned value is the element's rank inside its wave's share of the key
__device__ __forceinline__ int32_t wave_rank_det(int32_t* wcnt_wave, bool valid, int key) {
  const int lane = threadIdx.x & 63;
  uint64_t todo = __ballot(valid);
  int32_t slot = 0;
  while (todo) {
    const int leader = __ffsll((unsigned long long)todo) - 1;
    const int kk = __shfl(key, leader, 64);
    const uint64_t same = __ballot(valid && key == kk);
    int32_t old = 0;
    if (lane == leader) {
      old = wcnt_wave[kk];
      wcnt_wave[kk] = old + (int32_t)__popcll(same);
    }
    old = __shfl(old, leader, 64);
    if (valid && key == kk) slot = old + (int32_t)__popcll(same & ((1ull << lane) - 1ull));
    todo &= ~same;
  }
  return slot;
}

// deterministic routing, between the count pass and the routing pass: for every segment (slice, owner, bank) the exclusive
// prefix of its workgroups' counts in launch order -> wg_base, and the segment's total into its cursor (what the routing
// pass's header writer reads, and zeroes)
__global__ __launch_bounds__(256) void shard_route_scan_kernel(const int32_t* __restrict__ wg_cnt, int32_t* __restrict__ wg_base,
                                                               int32_t* __restrict__ cursors, int n_blocks, int G, int n_slices,
                                                               int n_banks) {
  // one WORKGROUP per segment (a thread per segment walked ~1000 dependent loads: 158 us): the segment's workgroups are a
  // contiguous range of block ids; every thread sums a contiguous piece of it, the 256 piece sums are scanned in LDS, and
  // the thread writes its piece's bases
  __shared__ int32_t part[256];
  const int sg = blockIdx.x;
  const int bank = sg % n_banks, g = (sg / n_banks) % G, slice = sg / (n_banks * G);
  const int want = slice * n_banks + bank, n_micro = n_slices * n_banks;
  // blocks b with floor(b * n_micro / n_blocks) == want  <=>  b in [ceil(want * n_blocks / n_micro), ceil((want + 1) * n_blocks / n_micro))
  const int lo = (int)(((int64_t)want * n_blocks + n_micro - 1) / n_micro);
  const int hi = (int)(((int64_t)(want + 1) * n_blocks + n_micro - 1) / n_micro);
  const int len = hi - lo, per = (len + 255) / 256;
  const int b0 = lo + (int)threadIdx.x * per, b1 = b0 + per < hi ? b0 + per : hi;
  int32_t sum = 0;
  for (int b = b0; b < b1; ++b) sum += wg_cnt[(size_t)b * G + g];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {          // inclusive Hillis-Steele scan of the piece sums
    const int32_t v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int32_t run = part[threadIdx.x] - sum;
  for (int b = b0; b < b1; ++b) {
    wg_base[(size_t)b * G + g] = run;
    run += wg_cnt[(size_t)b * G + g];
  }
  if (threadIdx.x == 255) cursors[(size_t)sg * CURSOR_PAD] = part[255];
}

constexpr int TICKET_SUB = 32;      // two-level completion ticket: 32 sub-words, then one top word

template <bool COUNT_ONLY>
__global__ __launch_bounds__(256) void shard_sample_route_kernel(const RouteV2 a) {
  // counters per (owner, query of this workgroup): cnt[g + G * q].  Without grouping q = 0 and cnt[g] is the owner's count.
  __shared__ int32_t cnt[ROUTE_CNT], qbase[ROUTE_CNT], base[64];
  __shared__ int32_t wcnt[4][ROUTE_CNT];      // deterministic slots: per-wave counters
  __shared__ int s_last;
  // contiguous ranges of workgroups: n_slices slices, each cut into n_banks banks
  const int micro = (int)(((int64_t)blockIdx.x * a.n_slices * a.n_banks) / gridDim.x);
  const int slice = micro / a.n_banks, bank = micro - slice * a.n_banks;
  for (int t = threadIdx.x; t < ROUTE_CNT; t += 256) cnt[t] = 0;
  if (a.det_phase)
    for (int t = threadIdx.x; t < 4 * ROUTE_CNT; t += 256) (&wcnt[0][0])[t] = 0;
  constexpr int EPT = ROUTE_ITEMS * 4;
  // Query-grouped shares (the host checked: num_neg divides 1024, the grid's threads are a multiple of 1024, this rank's
  // element base a multiple of num_neg): the 1024 consecutive subsequences of a workgroup are, per Philox component, 1024
  // consecutive elements = 1024 / n WHOLE queries, and a wave's 64 consecutive elements belong to one of them.  An element's
  // slot is then its owner's share base + the share's prefix over the workgroup's queries + its rank inside (owner, query):
  // every query's elements for an owner form ONE contiguous run of the segment, and the owner needs no sort to find them.
  const int qpc = a.group_ql > 0 ? a.group_ql / a.unroll : 0;      // queries per component
  int32_t gl[EPT];         // owner << 16 | slot inside this workgroup's range; -1: no element
  uint32_t local[EPT];     // row inside the owner's block
  int32_t el[EPT];         // element index in the [n_queries, 1 + n] matrix
  const uint64_t range = (uint64_t)(a.pop.n_items - 1);
  const uint32_t T = a.pc.grid_threads;
#pragma unroll
  for (int r = 0; r < ROUTE_ITEMS; ++r) {
    const int64_t w = (int64_t)blockIdx.x * ROUTE_ITEMS_PER_BLOCK + r * 256 + threadIdx.x;
#pragma unroll
    for (int c = 0; c < 4; ++c) gl[r * 4 + c] = -1;
    if (w < a.n_groups) {
      const uint64_t kq = a.by_gt.div((uint32_t)w);
      const uint64_t idx = (uint64_t)w - kq * T;
      const uint64_t kk = a.k_lo + kq;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (a.sampler != RSA_SAMPLER_GIVEN) {
        const uint64_t ctr = a.pc.offset4 + kk;
        v = philox4x32_10(make_uint4((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)idx, (uint32_t)(idx >> 32)),
                          make_uint2((uint32_t)a.pc.seed, (uint32_t)(a.pc.seed >> 32)));
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c >= a.unroll) break;
        const uint64_t li = (kk * a.unroll + c) * T + idx;
        if (li < a.pc.elem_base || li - a.pc.elem_base >= (uint64_t)a.n_neg) continue;
        const int64_t flat = (int64_t)(li - a.pc.elem_base);
        int64_t id;
        if (a.sampler == RSA_SAMPLER_UNIFORM) {
          if (a.unroll == 2) {
            const uint64_t r64 = c == 0 ? (((uint64_t)v.x << 32) | v.y) : (((uint64_t)v.z << 32) | v.w);
            id = (int64_t)(r64 % range) + 1;
          } else {
            id = (int64_t)a.by_range.mod(pick(v, c)) + 1;
          }
          if (!COUNT_ONLY && a.neg_ids != nullptr) a.neg_ids[flat] = id;
        } else if (a.sampler == RSA_SAMPLER_POPULAR) {
          const float inv = 2.3283064e-10f;
          float u = __fmaf_rn((float)pick(v, c), inv, inv);
          u = u == 1.0f ? 0.0f : u;
          float pr;
          id = lookup_popular(a.pop, u, pr);
          if (!COUNT_ONLY) {
            if (a.neg_ids != nullptr) a.neg_ids[flat] = id;
            if (a.neg_logp != nullptr) a.neg_logp[flat] = logf(pr);
          }
        } else {
          id = a.neg_ids[flat];
        }
        const int64_t m = (int64_t)a.by_n.div((uint32_t)flat);
        el[r * 4 + c] = (int32_t)(flat + m + 1);               // = m * (n + 1) + 1 + (flat - m * n)
        int g = 0;
        int64_t loc = id < 0 ? 0 : id;
        if (a.G > 1) owner_and_row(id, a.by_rows, a.rows_per_shard, a.G, g, loc);
        local[r * 4 + c] = (uint32_t)loc;
        gl[r * 4 + c] = g << 16;
      }
    } else if (w - a.n_groups < a.n_queries) {
      const int64_t m = w - a.n_groups;
      const int64_t id = a.pos_ids[m];
      if (!COUNT_ONLY && a.pos_logp != nullptr) {
        const int64_t pc_ = id < 0 ? 0 : (id >= a.pop.n_items ? a.pop.n_items - 1 : id);
        a.pos_logp[m] = logf(a.pop.pop_prob[pc_]);
      }
      if (a.skip_pos) {      // the positives do not travel (the owners score them from the gathered ids): no slot
        if (!COUNT_ONLY) a.slot_of[m * (a.n + 1)] = -1;
      } else {
        el[r * 4] = (int32_t)(m * (a.n + 1));
        int g = 0;
        int64_t loc = id < 0 ? 0 : id;
        if (a.G > 1) owner_and_row(id, a.by_rows, a.rows_per_shard, a.G, g, loc);
        local[r * 4] = (uint32_t)loc;
        gl[r * 4] = g << 16;
      }
    }
  }
  __syncthreads();
  // one pass of RETURNING LDS atomics: an element's slot inside this workgroup's share of its owner's segment
  const int by_n_shift = a.group_ql > 0 ? 31 - __clz(a.n) : 0;     // (n is a power of two when grouped: it divides 1024)
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const bool valid = gl[k] >= 0;
    // (wave-uniform: a wave's 64 consecutive elements of one component lie inside one query)
    const int ql = a.group_ql > 0 ? (k & 3) * qpc + (((k >> 2) * 256 + (int)threadIdx.x) >> by_n_shift) : 0;
    int32_t ls;
    if (a.det_phase) ls = wave_rank_det(wcnt[threadIdx.x >> 6], valid, valid ? (gl[k] >> 16) + a.G * ql : 0);
    else ls = wave_count<true>(cnt + a.G * ql, valid, valid ? gl[k] >> 16 : 0, a.G == 1);
    if (valid) gl[k] |= ls;
  }
  __syncthreads();
  if (a.det_phase) {       // the waves' shares of every key in wave order; the key's count
    for (int t = threadIdx.x; t < ROUTE_CNT; t += 256) {
      const int32_t c0 = wcnt[0][t], c1 = wcnt[1][t], c2 = wcnt[2][t], c3 = wcnt[3][t];
      cnt[t] = c0 + c1 + c2 + c3;
      wcnt[0][t] = 0;
      wcnt[1][t] = c0;
      wcnt[2][t] = c0 + c1;
      wcnt[3][t] = c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      if (gl[k] < 0) continue;
      const int ql = a.group_ql > 0 ? (k & 3) * qpc + (((k >> 2) * 256 + (int)threadIdx.x) >> by_n_shift) : 0;
      gl[k] += wcnt[threadIdx.x >> 6][(gl[k] >> 16) + a.G * ql];
    }
    __syncthreads();
  }
  if ((int)threadIdx.x < a.G) {
    int32_t c = 0;
    const int nq = a.group_ql > 0 ? a.group_ql : 1;
    for (int q = 0; q < nq; ++q) {          // the share's prefix over the workgroup's queries
      qbase[threadIdx.x + a.G * q] = c;
      c += cnt[threadIdx.x + a.G * q];
    }
    cnt[threadIdx.x] = c;                   // (the header pass below reads the cursors, not this)
    if (a.det_phase == 1) {
      a.wg_cnt[(size_t)blockIdx.x * a.G + threadIdx.x] = c;
      base[threadIdx.x] = 0;
    } else if (a.det_phase == 2) {
      base[threadIdx.x] = a.wg_base[(size_t)blockIdx.x * a.G + threadIdx.x];
    } else {
      base[threadIdx.x] = c ? atomicAdd(&a.cursors[((slice * a.G + threadIdx.x) * a.n_banks + bank) * CURSOR_PAD], c) : 0;
    }
  }
  __syncthreads();      // the returning cursor atomics of this workgroup have been performed
  if (a.det_phase == 1) return;
  if (!COUNT_ONLY) {
    const int64_t seg_base = (int64_t)slice * a.G * a.n_banks;     // first segment of the slice
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      if (gl[k] < 0) continue;
      const int gk = gl[k] >> 16;
      const int ql = a.group_ql > 0 ? (k & 3) * qpc + (((k >> 2) * 256 + (int)threadIdx.x) >> by_n_shift) : 0;
      const int64_t slot = (int64_t)base[gk] + qbase[gk + a.G * ql] + (gl[k] & 0xffff);
      if (slot >= a.capacity) {        // no room: the element is dropped -- no key, no score, no gradient
        a.slot_of[el[k]] = -1;
        continue;
      }
      const int64_t m = (int64_t)a.by_width.div((uint32_t)el[k]);
      const int64_t at = (seg_base + (int64_t)gk * a.n_banks + bank) * a.stride + RSA_SHARD_HDR + slot;
      a.send[at] = ((a.query_base + m) << 32) | (int64_t)local[k];
      a.slot_of[el[k]] = (int32_t)at;
    }
  }
  // The workgroup that takes the last ticket sees every cursor final: it writes the segment headers (and the exact
  // counts for the calibration step) and leaves the cursors and the tickets zeroed for the next launch.  Two levels
  // (workgroup b -> sub-ticket b % 32, the last arrival of a sub-ticket -> the top ticket), like the loss reduction of
  // the fused kernels: one ticket word made the tail of every workgroup wait in a queue of gridDim.x atomics.
  const int segs = a.n_slices * a.G * a.n_banks;
  if (threadIdx.x == 0) {
    int32_t* tick = a.cursors + (int64_t)segs * CURSOR_PAD;
    const unsigned j = blockIdx.x % TICKET_SUB;
    const unsigned members = (gridDim.x - j + TICKET_SUB - 1) / TICKET_SUB;
    const unsigned n_sub = gridDim.x < TICKET_SUB ? gridDim.x : TICKET_SUB;
    int last = 0;
    const int t = __hip_atomic_fetch_add(&tick[(1 + j) * CURSOR_PAD], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == (int)members - 1) {
      __hip_atomic_store(&tick[(1 + j) * CURSOR_PAD], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int t2 = __hip_atomic_fetch_add(&tick[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t2 == (int)n_sub - 1) {
        __hip_atomic_store(&tick[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = 1;
      }
    }
    s_last = last;
  }
  __syncthreads();
  if (s_last && threadIdx.x < 64) {
    int64_t dropped = 0;
    for (int sg = threadIdx.x; sg < segs; sg += 64) {
      const int64_t c = __hip_atomic_load(&a.cursors[sg * CURSOR_PAD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (c > a.capacity) dropped += c - a.capacity;
    }
#pragma unroll
    for (int mk = 32; mk >= 1; mk >>= 1) dropped += __shfl_xor((long long)dropped, mk, 64);
    if (!COUNT_ONLY && a.extra_dropped != nullptr) {
      dropped += *a.extra_dropped;            // (every lane reads it before lane 0 resets it: the wave runs in lock step)
      __builtin_amdgcn_wave_barrier();
      if (threadIdx.x == 0) *a.extra_dropped = 0;
    }
    for (int sg = threadIdx.x; sg < segs; sg += 64) {
      const int64_t c = __hip_atomic_load(&a.cursors[sg * CURSOR_PAD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a.counts_out != nullptr) a.counts_out[sg] = (int32_t)c;
      if (!COUNT_ONLY) {
        a.send[(int64_t)sg * a.stride] = c < a.capacity ? c : a.capacity;
        a.send[(int64_t)sg * a.stride + 1] = (sg % a.n_banks) == 0 ? dropped : 0;
      }
      __hip_atomic_store(&a.cursors[sg * CURSOR_PAD], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Owner side of the backward: received segments -> (local row, query index) per slot, -1 for slots past a segment's
// count (the sorted scatters drop negative ids), and the step's update scales scale_out[2] = {gate * scale_in (1 when
// null), gate}: gate = 0 when ANY rank dropped an element in this step (the step then changes no weight anywhere).
__global__ __launch_bounds__(256) void shard_unpack_segments_kernel(const int64_t* __restrict__ keys, int64_t numel,
                                                                    int64_t stride, int64_t* __restrict__ local_rows,
                                                                    int64_t* __restrict__ qidx,
                                                                    const float* __restrict__ scale_in,
                                                                    const int32_t* __restrict__ step_dropped,
                                                                    float* __restrict__ scale_out) {
  if (scale_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
    const float gate = (step_dropped != nullptr && step_dropped[0] != 0) ? 0.f : 1.f;
    scale_out[0] = gate * (scale_in ? scale_in[0] : 1.f);
    scale_out[1] = gate;
  }
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += step) {
    const int64_t seg = i / stride, within = i - seg * stride;
    const int64_t count = keys[seg * stride];
    const bool live = within >= RSA_SHARD_HDR && within - RSA_SHARD_HDR < count;
    const int64_t k = keys[i];
    local_rows[i] = live ? (k & 0xffffffffll) : -1;
    qidx[i] = live ? ((k >> 32) & 0x7fffffffll) : -1;
  }
}

// d_send[slot_of[e]] = d[e] (d = [dpos | dneg] of a loss evaluated outside): d loss/d score in routed order
__global__ __launch_bounds__(256) void shard_scatter_slots_kernel(const float* __restrict__ dpos,
                                                                  const float* __restrict__ dneg,
                                                                  const int32_t* __restrict__ slot_of, int64_t n_queries,
                                                                  int n, FastDiv by_width, float* __restrict__ d_send) {
  const int64_t numel = n_queries * (n + 1);
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < numel; e += step) {
    const int32_t s = slot_of[e];
    if (s < 0) continue;
    const int64_t m = (int64_t)by_width.div((uint64_t)e);
    const int c = (int)(e - m * (n + 1));
    d_send[s] = c == 0 ? dpos[m] : dneg[m * n + (c - 1)];
  }
}

// Home side of the forward: ONE wave per query walks the query's (1 + n) elements in tiles of 64, gathers their
// scores from the returned buffer through slot_of and evaluates the loss in place:
//   LOSS 0: scores only (pos_score / neg_score; a dropped element reads 0)
//   LOSS 1: BPRLoss (loss_func.py:55-59)    row = -(1/n) sum_j logsigmoid(pos - neg_j)
//   LOSS 2: SampledSoftmaxLoss (:80-90)     row = logsumexp(z_pos, z_1..z_n) - z_pos, z = score - logQ
// with d loss/d score written per element (dpos / dneg) and / or straight into the routed-order gradient buffer d_send
// (the gather + concatenation of the first version).  A dropped negative contributes nothing to the sums (the 1/n of
// BPR stays); a dropped positive zeroes the query's loss and every gradient of the query.  The mean over queries
// (divided by mean_den: the local or the job-wide query count) is reduced in the same launch (reduce_mean_loss).
struct HomeArgs {
  const float* scores;
  const int32_t* slot_of;
  const float* pos_logp;
  const float* neg_logp;
  float* pos_score;
  float* neg_score;
  float* row_loss;
  float* loss_out;
  float* dpos;
  float* dneg;
  float* d_send;
  unsigned int* flag_word;
  float* loss_partials;
  int64_t n_queries, mean_den;
  int32_t n;
};

// Tiles are handled HOME_TB at a time with the loads of a batch issued together (slots, then the scores they point
// at): one tile at a time the kernel was a chain of 2 dependent memory round trips per tile (27 us for 4 M elements).
constexpr int HOME_TB = 8;

template <int LOSS>
__global__ __launch_bounds__(256) void shard_home_kernel(const HomeArgs a) {
  const int lane = lane_id();
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t wstride = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int n = a.n;
  const int T = (n + 63) >> 6;
  const float w = 1.f / (float)n, inv_m = 1.f / (float)a.mean_den;
  float wave_loss = 0.f;
  for (int64_t m = wave0; m < a.n_queries; m += wstride) {
    const int64_t E0 = m * (n + 1);
    const int32_t* so = a.slot_of + E0 + 1;
    const int32_t sp = a.slot_of[E0];
    const float pos = sp >= 0 ? a.scores[sp] : 0.f;
    if (lane == 0 && a.pos_score != nullptr) a.pos_score[m] = pos;
    // a batch of tiles: slots and scores of this lane's element of each tile
    auto fetch = [&](int t0, int32_t (&s)[HOME_TB], float (&sc)[HOME_TB]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < HOME_TB; ++i) {
        const int j = ((t0 + i) << 6) + lane;
        s[i] = (t0 + i < T && j < n) ? so[j] : -1;
      }
#pragma unroll
      for (int i = 0; i < HOME_TB; ++i) sc[i] = s[i] >= 0 ? a.scores[s[i]] : 0.f;
    };
    if constexpr (LOSS == 0) {
      for (int t0 = 0; t0 < T; t0 += HOME_TB) {
        int32_t s[HOME_TB];
        float sc[HOME_TB];
        fetch(t0, s, sc);
#pragma unroll
        for (int i = 0; i < HOME_TB; ++i) {
          const int j = ((t0 + i) << 6) + lane;
          if (t0 + i < T && j < n) a.neg_score[m * n + j] = sc[i];
        }
      }
    } else if constexpr (LOSS == 1) {
      float lsum = 0.f, gsum = 0.f;
      for (int t0 = 0; t0 < T; t0 += HOME_TB) {
        int32_t s[HOME_TB];
        float sc[HOME_TB];
        fetch(t0, s, sc);
#pragma unroll
        for (int i = 0; i < HOME_TB; ++i) {
          const int j = ((t0 + i) << 6) + lane;
          const bool in = t0 + i < T && j < n;
          const bool live = s[i] >= 0 && sp >= 0;
          const float xd = pos - sc[i];
          const float tt = __expf(-fabsf(xd));
          const float ls = live ? fminf(xd, 0.f) - __logf(1.f + tt) : 0.f;
          const float r = __frcp_rn(1.f + tt);
          const float sg = live ? (xd >= 0.f ? tt * r : r) * w * inv_m : 0.f;
          if (in) {
            if (a.neg_score != nullptr) a.neg_score[m * n + j] = sc[i];
            if (a.dneg != nullptr) a.dneg[m * n + j] = sg;
          }
          if (s[i] >= 0 && a.d_send != nullptr) a.d_send[s[i]] = sg;
          lsum += ls;
          gsum += sg;
        }
      }
      lsum = group_sum<64>(lsum);
      gsum = group_sum<64>(gsum);
      const float row = -lsum * w;
      wave_loss += row;
      if (lane == 0) {
        if (a.row_loss != nullptr) a.row_loss[m] = row;
        if (a.dpos != nullptr) a.dpos[m] = -gsum;
        if (sp >= 0 && a.d_send != nullptr) a.d_send[sp] = -gsum;
      }
    } else {
      const float lq_pos = a.pos_logp ? a.pos_logp[m] : 0.f;
      const float z_pos = pos - lq_pos;
      const float* lq = a.neg_logp ? a.neg_logp + m * n : nullptr;
      float run_m = -INFINITY, run_s = 0.f;
      for (int t0 = 0; t0 < T; t0 += HOME_TB) {
        int32_t s[HOME_TB];
        float sc[HOME_TB], z[HOME_TB];
        fetch(t0, s, sc);
        float bm = -INFINITY;
#pragma unroll
        for (int i = 0; i < HOME_TB; ++i) {
          const int j = ((t0 + i) << 6) + lane;
          const bool in = t0 + i < T && j < n;
          z[i] = s[i] >= 0 ? sc[i] - (lq ? lq[j] : 0.f) : -INFINITY;
          if (in && a.neg_score != nullptr) a.neg_score[m * n + j] = sc[i];
          bm = fmaxf(bm, z[i]);
        }
        const float m_new = fmaxf(run_m, wave_max(bm));
        if (m_new > -INFINITY) {
          float part = 0.f;
#pragma unroll
          for (int i = 0; i < HOME_TB; ++i) part += s[i] >= 0 ? __expf(z[i] - m_new) : 0.f;
          run_s = run_s * __expf(run_m - m_new) + group_sum<64>(part);
          run_m = m_new;
        }
      }
      const bool gone = sp < 0;                   // dropped positive: the query leaves the step
      const float top = fmaxf(run_m, z_pos);
      const float lse = top + logf((run_m > -INFINITY ? run_s * expf(run_m - top) : 0.f) + expf(z_pos - top));
      const bool bad = isinf(z_pos);              // padded positive: NaN like the reference (loss_func.py:88-89)
      const float row = gone ? 0.f : (bad ? NAN : lse - z_pos);
      const float dp = gone ? 0.f : (bad ? NAN : (expf(z_pos - lse) - 1.f) * inv_m);
      wave_loss += row;
      if (lane == 0) {
        if (a.row_loss != nullptr) a.row_loss[m] = row;
        if (a.dpos != nullptr) a.dpos[m] = dp;
        if (sp >= 0 && a.d_send != nullptr) a.d_send[sp] = dp;
      }
      if (a.dneg != nullptr || a.d_send != nullptr) {
        for (int t0 = 0; t0 < T; t0 += HOME_TB) {
          int32_t s[HOME_TB];
          float sc[HOME_TB];
          fetch(t0, s, sc);
#pragma unroll
          for (int i = 0; i < HOME_TB; ++i) {
            const int j = ((t0 + i) << 6) + lane;
            const bool in = t0 + i < T && j < n;
            float dv = 0.f;
            if (s[i] >= 0 && !gone) dv = bad ? NAN : __expf(sc[i] - (lq ? lq[j] : 0.f) - lse) * inv_m;
            if (in && a.dneg != nullptr) a.dneg[m * n + j] = dv;
            if (s[i] >= 0 && a.d_send != nullptr) a.d_send[s[i]] = dv;
          }
        }
      }
    }
  }
  if constexpr (LOSS != 0) {
    if (a.loss_out != nullptr) reduce_mean_loss(wave_loss, a.loss_out, a.flag_word, a.loss_partials, a.mean_den);
  }
}

// Short queries (n <= 256): a wave handles QPW queries per iteration -- QPW * TPQ = 8 tiles in flight, the loads of all of
// them issued together -- and keeps every score in registers, so the SampledSoftmax gradient needs no second pass.
// One query per iteration was a chain of four dependent round trips per query (52 us for 65536 queries of n = 64).
template <int LOSS, int QPW>
__global__ __launch_bounds__(256) void shard_home_small_kernel(const HomeArgs a) {
  constexpr int TPQ = HOME_TB / QPW;
  const int lane = lane_id();
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t wstride = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int n = a.n;
  const float w = 1.f / (float)n, inv_m = 1.f / (float)a.mean_den;
  float wave_loss = 0.f;
  for (int64_t m0 = wave0 * QPW; m0 < a.n_queries; m0 += wstride * QPW) {
    int32_t sp[QPW], s[QPW][TPQ];
    float pos[QPW], sc[QPW][TPQ], lq[QPW][TPQ], lqp[QPW];
#pragma unroll
    for (int q = 0; q < QPW; ++q) {
      const int64_t m = m0 + q;
      const bool mv = m < a.n_queries;
      sp[q] = mv ? a.slot_of[m * (n + 1)] : -1;
#pragma unroll
      for (int t = 0; t < TPQ; ++t) {
        const int j = (t << 6) + lane;
        s[q][t] = (mv && j < n) ? a.slot_of[m * (n + 1) + 1 + j] : -1;
        lq[q][t] = (LOSS == 2 && mv && j < n && a.neg_logp) ? a.neg_logp[m * n + j] : 0.f;
      }
      lqp[q] = (LOSS == 2 && mv && a.pos_logp) ? a.pos_logp[m] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < QPW; ++q) {
      pos[q] = sp[q] >= 0 ? a.scores[sp[q]] : 0.f;
#pragma unroll
      for (int t = 0; t < TPQ; ++t) sc[q][t] = s[q][t] >= 0 ? a.scores[s[q][t]] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < QPW; ++q) {
      const int64_t m = m0 + q;
      if (m >= a.n_queries) break;                 // wave-uniform
      if (lane == 0 && a.pos_score != nullptr) a.pos_score[m] = pos[q];
      if (a.neg_score != nullptr) {
#pragma unroll
        for (int t = 0; t < TPQ; ++t) {
          const int j = (t << 6) + lane;
          if (j < n) a.neg_score[m * n + j] = sc[q][t];
        }
      }
      if constexpr (LOSS == 1) {
        float lsum = 0.f, gsum = 0.f;
#pragma unroll
        for (int t = 0; t < TPQ; ++t) {
          const int j = (t << 6) + lane;
          const bool live = s[q][t] >= 0 && sp[q] >= 0;
          const float xd = pos[q] - sc[q][t];
          const float tt = __expf(-fabsf(xd));
          const float ls = live ? fminf(xd, 0.f) - __logf(1.f + tt) : 0.f;
          const float r = __frcp_rn(1.f + tt);
          const float sg = live ? (xd >= 0.f ? tt * r : r) * w * inv_m : 0.f;
          if (j < n && a.dneg != nullptr) a.dneg[m * n + j] = sg;
          if (s[q][t] >= 0 && a.d_send != nullptr) a.d_send[s[q][t]] = sg;
          lsum += ls;
          gsum += sg;
        }
        lsum = group_sum<64>(lsum);
        gsum = group_sum<64>(gsum);
        const float row = -lsum * w;
        wave_loss += row;
        if (lane == 0) {
          if (a.row_loss != nullptr) a.row_loss[m] = row;
          if (a.dpos != nullptr) a.dpos[m] = -gsum;
          if (sp[q] >= 0 && a.d_send != nullptr) a.d_send[sp[q]] = -gsum;
        }
      } else if constexpr (LOSS == 2) {
        const float z_pos = pos[q] - lqp[q];
        float z[TPQ], bm = -INFINITY;
#pragma unroll
        for (int t = 0; t < TPQ; ++t) {
          z[t] = s[q][t] >= 0 ? sc[q][t] - lq[q][t] : -INFINITY;
          bm = fmaxf(bm, z[t]);
        }
        const float run_m = wave_max(bm);
        float part = 0.f;
        if (run_m > -INFINITY) {
#pragma unroll
          for (int t = 0; t < TPQ; ++t) part += s[q][t] >= 0 ? __expf(z[t] - run_m) : 0.f;
        }
        const float run_s = group_sum<64>(part);
        const bool gone = sp[q] < 0;
        const float top = fmaxf(run_m, z_pos);
        const float lse = top + logf((run_m > -INFINITY ? run_s * expf(run_m - top) : 0.f) + expf(z_pos - top));
        const bool bad = isinf(z_pos);
        const float row = gone ? 0.f : (bad ? NAN : lse - z_pos);
        const float dp = gone ? 0.f : (bad ? NAN : (expf(z_pos - lse) - 1.f) * inv_m);
        wave_loss += row;
        if (lane == 0) {
          if (a.row_loss != nullptr) a.row_loss[m] = row;
          if (a.dpos != nullptr) a.dpos[m] = dp;
          if (sp[q] >= 0 && a.d_send != nullptr) a.d_send[sp[q]] = dp;
        }
#pragma unroll
        for (int t = 0; t < TPQ; ++t) {
          const int j = (t << 6) + lane;
          float dv = 0.f;
          if (s[q][t] >= 0 && !gone) dv = bad ? NAN : __expf(z[t] - lse) * inv_m;
          if (j < n && a.dneg != nullptr) a.dneg[m * n + j] = dv;
          if (s[q][t] >= 0 && a.d_send != nullptr) a.d_send[s[q][t]] = dv;
        }
      }
    }
  }
  if constexpr (LOSS != 0) {
    if (a.loss_out != nullptr) reduce_mean_loss(wave_loss, a.loss_out, a.flag_word, a.loss_partials, a.mean_den);
  }
}

template <int LOSS>
static void launch_home(const HomeArgs& h, hipStream_t s) {
  const int n = h.n;
  const int qpw = n <= 64 ? 8 : (n <= 128 ? 4 : (n <= 256 ? 2 : 1));
  int64_t blocks = (h.n_queries + 4 * qpw - 1) / (4 * qpw);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  const dim3 grid((unsigned)blocks), block(256);
  if (qpw == 8) hipLaunchKernelGGL((shard_home_small_kernel<LOSS, 8>), grid, block, 0, s, h);
  else if (qpw == 4) hipLaunchKernelGGL((shard_home_small_kernel<LOSS, 4>), grid, block, 0, s, h);
  else if (qpw == 2) hipLaunchKernelGGL((shard_home_small_kernel<LOSS, 2>), grid, block, 0, s, h);
  else hipLaunchKernelGGL(shard_home_kernel<LOSS>, grid, block, 0, s, h);
}

static inline int grid1d(int64_t numel) {
  int64_t b = (numel + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace rsa

using namespace rsa;

extern "C" int rsa_shard_count(const int64_t* pos_ids, const int64_t* neg_ids, int64_t n_queries, int32_t num_neg,
                               int64_t rows_per_shard, int32_t n_shards, int32_t* counts, rsa_stream_t stream) {
  RSA_CHECK_ARG(n_queries >= 0 && num_neg >= 0 && rows_per_shard >= 0, "rsa_shard_count: bad sizes");
  RSA_CHECK_ARG(n_shards >= 1 && n_shards <= 64, "rsa_shard_count: n_shards must be in [1, 64]");
  RSA_CHECK_ARG(counts != nullptr, "rsa_shard_count: counts is null");
  hipError_t e = hipMemsetAsync(counts, 0, sizeof(int32_t) * n_shards, (hipStream_t)stream);
  if (e != hipSuccess) {
    rsa::set_error("rsa_shard_count: memset failed: %s", hipGetErrorString(e));
    return RSA_ERR_HIP;
  }
  if (n_queries == 0) return RSA_OK;
  RSA_CHECK_ARG(pos_ids && (neg_ids || num_neg == 0), "rsa_shard_count: null ids");
  RSA_CHECK_ARG(n_queries * (num_neg + 1) < (1ll << 32), "rsa_shard_count: more than 2^32 elements");
  const RouteShape sh{pos_ids, neg_ids, n_queries, (int)num_neg, (int)n_shards, rows_per_shard,
                      make_fastdiv((uint64_t)num_neg + 1), make_fastdiv((uint64_t)(rows_per_shard ? rows_per_shard : n_shards))};
  int64_t count_blocks = (n_queries * (num_neg + 1) + ROUTE_CHUNK - 1) / ROUTE_CHUNK;
  if (count_blocks > 4096) count_blocks = 4096;
  hipLaunchKernelGGL(shard_count_kernel, dim3((unsigned)count_blocks), dim3(256), 0, (hipStream_t)stream, sh, counts);
  RSA_CHECK_LAUNCH("rsa_shard_count");
  return RSA_OK;
}

extern "C" int rsa_shard_route(const int64_t* pos_ids, const int64_t* neg_ids, int64_t n_queries, int32_t num_neg,
                               int64_t rows_per_shard, int32_t n_shards, int64_t query_base, int32_t* cursor,
                               int64_t* keys, int64_t* positions, rsa_stream_t stream) {
  RSA_CHECK_ARG(n_queries >= 0 && num_neg >= 0 && rows_per_shard >= 0 && rows_per_shard < (1ll << 32),
                "rsa_shard_route: bad sizes");
  RSA_CHECK_ARG(n_shards >= 1 && n_shards <= 64, "rsa_shard_route: n_shards must be in [1, 64]");
  RSA_CHECK_ARG(query_base >= 0 && query_base + n_queries < (1ll << 31), "rsa_shard_route: query index overflow");
  if (n_queries == 0) return RSA_OK;
  RSA_CHECK_ARG(pos_ids && (neg_ids || num_neg == 0) && cursor && keys && positions, "rsa_shard_route: null pointer");
  const int64_t numel = n_queries * (num_neg + 1);
  const int64_t blocks = (numel + ROUTE_CHUNK - 1) / ROUTE_CHUNK;
  RSA_CHECK_ARG(numel < (1ll << 31), "rsa_shard_route: more than 2^31 elements");
  const RouteShape sh{pos_ids, neg_ids, n_queries, (int)num_neg, (int)n_shards, rows_per_shard,
                      make_fastdiv((uint64_t)num_neg + 1), make_fastdiv((uint64_t)(rows_per_shard ? rows_per_shard : n_shards))};
  hipLaunchKernelGGL(shard_route_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, sh, query_base, cursor, keys,
                     positions);
  RSA_CHECK_LAUNCH("rsa_shard_route");
  return RSA_OK;
}

extern "C" int rsa_shard_unpack(const int64_t* keys, int64_t numel, int64_t* local_rows, int64_t* query_index,
                                rsa_stream_t stream) {
  RSA_CHECK_ARG(numel >= 0, "rsa_shard_unpack: numel < 0");
  if (numel == 0) return RSA_OK;
  RSA_CHECK_ARG(keys && local_rows && query_index, "rsa_shard_unpack: null pointer");
  hipLaunchKernelGGL(shard_unpack_kernel, dim3(grid1d(numel)), dim3(256), 0, (hipStream_t)stream, keys, numel,
                     local_rows, query_index);
  RSA_CHECK_LAUNCH("rsa_shard_unpack");
  return RSA_OK;
}

extern "C" int rsa_scatter_f32(const float* src, const int64_t* positions, int64_t numel, float* dst,
                               rsa_stream_t stream) {
  RSA_CHECK_ARG(numel >= 0, "rsa_scatter_f32: numel < 0");
  if (numel == 0) return RSA_OK;
  RSA_CHECK_ARG(src && positions && dst, "rsa_scatter_f32: null pointer");
  hipLaunchKernelGGL(scatter_f32_kernel, dim3(grid1d(numel)), dim3(256), 0, (hipStream_t)stream, src, positions, numel,
                     dst);
  RSA_CHECK_LAUNCH("rsa_scatter_f32");
  return RSA_OK;
}

extern "C" int rsa_gather_f32(const float* src, const int64_t* positions, int64_t numel, float* dst,
                              rsa_stream_t stream) {
  RSA_CHECK_ARG(numel >= 0, "rsa_gather_f32: numel < 0");
  if (numel == 0) return RSA_OK;
  RSA_CHECK_ARG(src && positions && dst, "rsa_gather_f32: null pointer");
  hipLaunchKernelGGL(gather_f32_kernel, dim3(grid1d(numel)), dim3(256), 0, (hipStream_t)stream, src, positions, numel,
                     dst);
  RSA_CHECK_LAUNCH("rsa_gather_f32");
  return RSA_OK;
}

// ------------------------------------------------------------------------------------------------ ABI v5 entry points
extern "C" int64_t rsa_shard_segment_stride(int64_t capacity) { return capacity + RSA_SHARD_HDR; }

// Queries per workgroup when the router can write query-grouped shares (0: it cannot): see the kernel.
extern "C" int32_t rsa_shard_route_query_groups(int32_t num_neg, uint32_t grid_threads, uint64_t elem_base, int32_t unroll,
                                                int32_t n_shards) {
  if (num_neg < 64 || num_neg > 1024 || (1024 % num_neg) != 0 || (grid_threads % 1024u) != 0 || (elem_base % (uint64_t)num_neg) != 0)
    return 0;
  if (unroll != 2 && unroll != 4) return 0;
  const int32_t ql = unroll * (1024 / num_neg);
  return (int64_t)ql * n_shards <= ROUTE_CNT ? ql : 0;
}

extern "C" int rsa_shard_sample_route(const rsa_shard_route_args* a, rsa_stream_t stream) {
  RSA_CHECK_ARG(a != nullptr, "rsa_shard_sample_route: args is null");
  RSA_CHECK_ARG(a->n_queries >= 0 && a->num_neg >= 0 && a->rows_per_shard >= 0 && a->rows_per_shard < (1ll << 32),
                "rsa_shard_sample_route: bad sizes");
  RSA_CHECK_ARG(a->n_shards >= 1 && a->n_shards <= 64, "rsa_shard_sample_route: n_shards must be in [1, 64]");
  RSA_CHECK_ARG(a->n_slices >= 1 && a->n_banks >= 1 && (int64_t)a->n_slices * a->n_shards * a->n_banks <= 4096,
                "rsa_shard_sample_route: n_slices, n_banks must be >= 1 with n_slices * n_shards * n_banks <= 4096");
  RSA_CHECK_ARG(a->query_base >= 0 && a->query_base + a->n_queries < (1ll << 31), "rsa_shard_sample_route: query index overflow");
  RSA_CHECK_ARG(a->sampler >= RSA_SAMPLER_GIVEN && a->sampler <= RSA_SAMPLER_POPULAR, "rsa_shard_sample_route: unknown sampler %d",
                a->sampler);
  RSA_CHECK_ARG(a->cursors != nullptr, "rsa_shard_sample_route: cursors is null");
  const bool count_only = a->send_keys == nullptr;
  RSA_CHECK_ARG(!count_only || a->counts_out != nullptr, "rsa_shard_sample_route: nothing to do (send_keys and counts_out null)");
  if (a->n_queries == 0) {
    if (a->counts_out) {
      if (hipMemsetAsync(a->counts_out, 0, sizeof(int32_t) * a->n_slices * a->n_shards * a->n_banks, (hipStream_t)stream) != hipSuccess) {
        rsa::set_error("rsa_shard_sample_route: memset failed");
        return RSA_ERR_HIP;
      }
    }
    if (!count_only) {     // empty segments still carry their headers
      const int64_t stride = a->capacity + RSA_SHARD_HDR;
      if (hipMemset2DAsync(a->send_keys, sizeof(int64_t) * stride, 0, sizeof(int64_t) * RSA_SHARD_HDR,
                           (size_t)a->n_slices * a->n_shards * a->n_banks, (hipStream_t)stream) != hipSuccess) {
        rsa::set_error("rsa_shard_sample_route: memset failed");
        return RSA_ERR_HIP;
      }
    }
    return RSA_OK;
  }
  RSA_CHECK_ARG(a->pos_ids != nullptr, "rsa_shard_sample_route: pos_ids is null");
  RSA_CHECK_ARG(a->num_neg == 0 || a->sampler != RSA_SAMPLER_GIVEN || a->neg_ids != nullptr,
                "rsa_shard_sample_route: sampler GIVEN needs neg_ids");
  RSA_CHECK_ARG(a->n_items >= 2 && a->n_items < (1ll << 31), "rsa_shard_sample_route: n_items out of range");
  if (a->sampler != RSA_SAMPLER_GIVEN && a->num_neg > 0)
    RSA_CHECK_ARG(a->grid_threads > 0 && (a->offset & 3) == 0, "rsa_shard_sample_route: bad philox state");
  if (a->sampler == RSA_SAMPLER_POPULAR && a->num_neg > 0) {
    RSA_CHECK_ARG(a->table && a->pop_prob, "rsa_shard_sample_route: popularity tables missing");
    if (a->cdf_lines != nullptr)
      RSA_CHECK_ARG(a->lines_log2 >= 0 && a->lines_log2 <= 28 && ((uintptr_t)a->cdf_lines & 127) == 0,
                    "rsa_shard_sample_route: cdf_lines must be 128-byte aligned with lines_log2 in [0, 28]");
    else
      RSA_CHECK_ARG(a->guide && a->guide_log2 >= 0 && a->guide_log2 <= 28, "rsa_shard_sample_route: guide table missing");
  }
  RSA_CHECK_ARG(a->pos_logp == nullptr || a->pop_prob != nullptr, "rsa_shard_sample_route: pos_logp needs pop_prob");
  if (!count_only) {
    RSA_CHECK_ARG(a->slot_of != nullptr, "rsa_shard_sample_route: slot_of is null");
    RSA_CHECK_ARG(a->capacity >= 1 && (a->capacity + RSA_SHARD_HDR) * a->n_shards * a->n_slices * a->n_banks < (1ll << 31),
                  "rsa_shard_sample_route: capacity out of range");
  }
  RSA_CHECK_ARG(a->n_queries * (a->num_neg + 1) < (1ll << 31), "rsa_shard_sample_route: more than 2^31 elements");
  const int64_t n_neg = a->n_queries * a->num_neg;
  RouteV2 r;
  r.pos_ids = a->pos_ids;
  r.neg_ids = a->neg_ids;
  r.neg_logp = a->neg_logp;
  r.pos_logp = a->pos_logp;
  r.send = a->send_keys;
  r.slot_of = a->slot_of;
  r.cursors = a->cursors;
  r.counts_out = a->counts_out;
  r.n_queries = a->n_queries;
  r.capacity = count_only ? (1ll << 40) : a->capacity;
  r.stride = a->capacity + RSA_SHARD_HDR;
  r.rows_per_shard = a->rows_per_shard;
  r.query_base = a->query_base;
  r.n_neg = n_neg;
  r.pop = PopTables{a->table, a->pop_prob, a->table_prob, a->cdf_lut, a->cdf_lines, a->guide, a->n_items, a->guide_log2,
                    a->lines_log2};
  r.n = a->num_neg;
  r.G = a->n_shards;
  r.sampler = a->sampler;
  r.n_slices = a->n_slices;
  r.n_banks = a->n_banks;
  r.skip_pos = a->skip_pos;
  // the enumeration of the negatives: Philox blocks of the torch call (see the kernel), or the same shape made up for
  // given ids (four interleaved quarters)
  if (a->sampler == RSA_SAMPLER_GIVEN || n_neg == 0) {
    int64_t quarter = ((n_neg + 3) / 4 + 255) / 256 * 256;
    if (quarter < 256) quarter = 256;
    r.pc = PhiloxCall{0, 0, (uint32_t)quarter, 0};
    r.unroll = 4;
  } else {
    r.pc = PhiloxCall{a->seed, a->offset >> 2, a->grid_threads, a->elem_base};
    r.unroll = (a->sampler == RSA_SAMPLER_UNIFORM && (uint64_t)(a->n_items - 1) >= (1ull << 28)) ? 2 : 4;   // ATen: 64-bit draws
  }
  const uint64_t T = r.pc.grid_threads;
  uint64_t k_lo = 0, k_hi = 0;
  if (n_neg > 0) {
    k_lo = (r.pc.elem_base / T) / r.unroll;
    k_hi = ((r.pc.elem_base + (uint64_t)n_neg - 1) / T) / r.unroll;
  }
  r.k_lo = k_lo;
  r.group_ql = 0;
  if (a->group_by_query) {
    RSA_CHECK_ARG(a->skip_pos && a->sampler != RSA_SAMPLER_GIVEN && n_neg > 0,
                  "rsa_shard_sample_route: group_by_query needs skip_pos and an in-kernel sampler");
    r.group_ql = rsa_shard_route_query_groups(a->num_neg, r.pc.grid_threads, r.pc.elem_base, r.unroll, a->n_shards);
    RSA_CHECK_ARG(r.group_ql > 0, "rsa_shard_sample_route: this shape cannot be routed query-grouped "
                                  "(rsa_shard_route_query_groups says so beforehand)");
  }
  r.extra_dropped = a->extra_dropped;
  r.det_phase = 0;
  r.wg_cnt = nullptr;
  r.wg_base = nullptr;
  r.n_groups = n_neg > 0 ? (int64_t)((k_hi - k_lo + 1) * T) : 0;
  RSA_CHECK_ARG(r.n_groups + a->n_queries < (1ll << 32), "rsa_shard_sample_route: too many work items");
  r.by_width = make_div32((uint64_t)a->num_neg + 1);
  r.by_rows = make_div32((uint64_t)(a->rows_per_shard ? a->rows_per_shard : a->n_shards));    // 0: interleaved rows
  r.by_n = make_div32((uint64_t)(a->num_neg > 0 ? a->num_neg : 1));
  r.by_gt = make_div32(T);
  r.by_range = make_div32((uint64_t)(a->n_items - 1));        // 32-bit draws only (ranges below 2^28)
  int64_t blocks = (r.n_groups + a->n_queries + ROUTE_ITEMS_PER_BLOCK - 1) / ROUTE_ITEMS_PER_BLOCK;
  if (blocks < (int64_t)a->n_slices * a->n_banks) blocks = (int64_t)a->n_slices * a->n_banks;     // every bank owns at least one workgroup
  RSA_CHECK_ARG(blocks < (1ll << 31), "rsa_shard_sample_route: grid too large");
  const dim3 grid((unsigned)blocks), block(256);
  if (a->deterministic && !count_only) {
    // count pass -> scan over the workgroups in launch order -> routing pass: no atomic decides a slot
    RSA_CHECK_ARG(a->wg_scratch != nullptr && a->wg_scratch_ints >= 2 * blocks * a->n_shards,
                  "rsa_shard_sample_route: deterministic routing needs wg_scratch of 2 * rsa_shard_route_workgroups() * n_shards ints");
    RSA_CHECK_ARG(a->n_shards <= ROUTE_CNT && (r.group_ql == 0 || r.group_ql * a->n_shards <= ROUTE_CNT),
                  "rsa_shard_sample_route: too many (owner, query) counters for one workgroup");
    int32_t* wg_cnt = a->wg_scratch;
    int32_t* wg_base = a->wg_scratch + blocks * a->n_shards;
    const int segs = a->n_slices * a->n_shards * a->n_banks;
    RouteV2 c1 = r;
    c1.det_phase = 1;
    c1.wg_cnt = wg_cnt;
    c1.counts_out = nullptr;
    hipLaunchKernelGGL(shard_sample_route_kernel<true>, grid, block, 0, (hipStream_t)stream, c1);
    hipLaunchKernelGGL(shard_route_scan_kernel, dim3((unsigned)segs), dim3(256), 0, (hipStream_t)stream, wg_cnt, wg_base,
                       a->cursors, (int)blocks, (int)a->n_shards, (int)a->n_slices, (int)a->n_banks);
    r.det_phase = 2;
    r.wg_base = wg_base;
    hipLaunchKernelGGL(shard_sample_route_kernel<false>, grid, block, 0, (hipStream_t)stream, r);
    RSA_CHECK_LAUNCH("rsa_shard_sample_route(deterministic)");
    return RSA_OK;
  }
  if (count_only) hipLaunchKernelGGL(shard_sample_route_kernel<true>, grid, block, 0, (hipStream_t)stream, r);
  else hipLaunchKernelGGL(shard_sample_route_kernel<false>, grid, block, 0, (hipStream_t)stream, r);
  RSA_CHECK_LAUNCH("rsa_shard_sample_route");
  return RSA_OK;
}

// workgroups rsa_shard_sample_route launches for these arguments (sizes the deterministic mode's wg_scratch: 2 x this x n_shards ints)
extern "C" int64_t rsa_shard_route_workgroups(const rsa_shard_route_args* a) {
  if (a == nullptr || a->n_queries <= 0) return 0;
  const int64_t n_neg = a->n_queries * a->num_neg;
  uint64_t T;
  int unroll = 4;
  uint64_t elem_base = 0;
  if (a->sampler == RSA_SAMPLER_GIVEN || n_neg == 0) {
    int64_t quarter = ((n_neg + 3) / 4 + 255) / 256 * 256;
    if (quarter < 256) quarter = 256;
    T = (uint64_t)quarter;
  } else {
    if (a->grid_threads == 0) return 0;
    T = a->grid_threads;
    elem_base = a->elem_base;
    unroll = (a->sampler == RSA_SAMPLER_UNIFORM && (uint64_t)(a->n_items - 1) >= (1ull << 28)) ? 2 : 4;
  }
  int64_t n_groups = 0;
  if (n_neg > 0) {
    const uint64_t k_lo = (elem_base / T) / unroll, k_hi = ((elem_base + (uint64_t)n_neg - 1) / T) / unroll;
    n_groups = (int64_t)((k_hi - k_lo + 1) * T);
  }
  int64_t blocks = (n_groups + a->n_queries + ROUTE_ITEMS_PER_BLOCK - 1) / ROUTE_ITEMS_PER_BLOCK;
  if (blocks < (int64_t)a->n_slices * a->n_banks) blocks = (int64_t)a->n_slices * a->n_banks;
  return blocks;
}

extern "C" int rsa_shard_home(const rsa_shard_home_args* a, rsa_stream_t stream) {
  RSA_CHECK_ARG(a != nullptr, "rsa_shard_home: args is null");
  RSA_CHECK_ARG(a->n_queries >= 0 && a->num_neg >= 0, "rsa_shard_home: negative sizes");
  RSA_CHECK_ARG(a->loss >= 0 && a->loss <= 2, "rsa_shard_home: loss must be 0 (none), 1 (BPR) or 2 (SampledSoftmax)");
  if (a->n_queries == 0) return RSA_OK;
  RSA_CHECK_ARG(a->scores && a->slot_of, "rsa_shard_home: scores / slot_of is null");
  RSA_CHECK_ARG(a->loss != 0 || (a->neg_score != nullptr || a->num_neg == 0), "rsa_shard_home: loss 0 needs neg_score");
  RSA_CHECK_ARG(a->loss == 0 || a->num_neg >= 1, "rsa_shard_home: a loss needs num_neg >= 1");
  RSA_CHECK_ARG(a->loss == 0 || a->mean_den >= 1, "rsa_shard_home: mean_den must be >= 1");
  HomeArgs h;
  h.scores = a->scores;
  h.slot_of = a->slot_of;
  h.pos_logp = a->pos_logp;
  h.neg_logp = a->neg_logp;
  h.pos_score = a->pos_score;
  h.neg_score = a->neg_score;
  h.row_loss = a->row_loss;
  h.loss_out = a->loss_out;
  h.dpos = a->dpos;
  h.dneg = a->dneg;
  h.d_send = a->d_send;
  h.flag_word = nullptr;
  h.loss_partials = nullptr;
  h.n_queries = a->n_queries;
  h.mean_den = a->mean_den;
  h.n = a->num_neg;
  if (a->loss != 0 && a->loss_out != nullptr) {
    RSA_CHECK_ARG(a->reduce_scratch != nullptr, "rsa_shard_home: loss_out needs reduce_scratch (rsa_scratch_bytes() bytes, zeroed once)");
    char* sc = reinterpret_cast<char*>(a->reduce_scratch);
    h.flag_word = reinterpret_cast<unsigned int*>(sc + SCRATCH_COUNTER);
    h.loss_partials = reinterpret_cast<float*>(sc + SCRATCH_FUSED_PARTIALS);
  }
  hipStream_t s = (hipStream_t)stream;
  if (a->loss == 0) launch_home<0>(h, s);
  else if (a->loss == 1) launch_home<1>(h, s);
  else launch_home<2>(h, s);
  RSA_CHECK_LAUNCH("rsa_shard_home");
  return RSA_OK;
}

extern "C" int rsa_shard_scatter_slots(const float* dpos, const float* dneg, const int32_t* slot_of, int64_t n_queries,
                                       int32_t num_neg, float* d_send, rsa_stream_t stream) {
  RSA_CHECK_ARG(n_queries >= 0 && num_neg >= 0, "rsa_shard_scatter_slots: negative sizes");
  if (n_queries == 0) return RSA_OK;
  RSA_CHECK_ARG(dpos && (dneg || num_neg == 0) && slot_of && d_send, "rsa_shard_scatter_slots: null pointer");
  RSA_CHECK_ARG(n_queries * (num_neg + 1) < (1ll << 31), "rsa_shard_scatter_slots: more than 2^31 elements");
  hipLaunchKernelGGL(shard_scatter_slots_kernel, dim3(grid1d(n_queries * (num_neg + 1))), dim3(256), 0, (hipStream_t)stream,
                     dpos, dneg, slot_of, n_queries, (int)num_neg, make_fastdiv((uint64_t)num_neg + 1), d_send);
  RSA_CHECK_LAUNCH("rsa_shard_scatter_slots");
  return RSA_OK;
}

extern "C" int rsa_shard_unpack_segments(const int64_t* keys, int64_t n_segments, int64_t stride, int64_t* local_rows,
                                         int64_t* query_index, const float* scale_in, const int32_t* step_dropped,
                                         float* scale_out, rsa_stream_t stream) {
  RSA_CHECK_ARG(n_segments >= 0 && stride > RSA_SHARD_HDR, "rsa_shard_unpack_segments: bad sizes");
  const int64_t numel = n_segments * stride;
  if (numel == 0) return RSA_OK;
  RSA_CHECK_ARG(keys && local_rows && query_index, "rsa_shard_unpack_segments: null pointer");
  hipLaunchKernelGGL(shard_unpack_segments_kernel, dim3(grid1d(numel)), dim3(256), 0, (hipStream_t)stream, keys, numel, stride,
                     local_rows, query_index, scale_in, step_dropped, scale_out);
  RSA_CHECK_LAUNCH("rsa_shard_unpack_segments");
  return RSA_OK;
}
