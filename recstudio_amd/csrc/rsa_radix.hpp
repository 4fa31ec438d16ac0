// In-tree LSD radix sort of (32-bit key, 32-bit payload) pairs for gfx950 (wave64), specialised to what the training
// steps sort: a few million (row id | query index, element number) pairs whose keys have 12 .. 27 significant bits.
//
// Replaces rocprim::radix_sort_pairs on the step's hot path.  What the specialisation buys:
//   * a pair travels as ONE 8-byte word (key << 32 | payload): one load / one store per element and pass, and the
//     consumers (sorted_apply_kernel, classify_solo_kernel, the owner-side walk) read one array;
//   * pass 0 reads its keys straight from the producer's tensors through a SOURCE functor (the step's int64 id tensors,
//     the received exchange segments): no key-extraction launch, no 33 MB id round trip;
//   * only ceil(bits / 8) passes, 8-bit digits, tiles sized so that the scatter pass runs in whole rounds of the chip.
// Per pass: radix_hist_kernel (per-tile digit counts) -> radix_scan_kernel (per digit: exclusive scan over the tiles) ->
// radix_scatter_kernel (stable rank inside the tile, the tile regrouped by digit in LDS, contiguous runs written out).
// Stable and deterministic: no atomics decide a position (the LDS atomics of the histogram only count).
//
// Ranking inside a tile: a wave owns 1024 consecutive elements and walks them 64 at a time in element order; the lanes
// that hold the same digit find each other with 8 ballots (one per digit bit), the rank inside the row is a popcount of
// the lower peers, the row's count goes to the wave's own 256 LDS counters.  The four waves' counters are then prefixed
// in wave order, so positions follow the element order: stable.
#pragma once
#include "rsa_common.hpp"
#include <type_traits>

namespace rsa {

constexpr int RDX_DIGIT_BITS = 8;
constexpr int RDX_BINS = 1 << RDX_DIGIT_BITS;
constexpr int RDX_ITEMS = 16;                         // rows of 64 per wave in the one-workgroup sort (radix_small_kernel)
constexpr int RDX_WAVE_TILE = 64 * RDX_ITEMS;         // 1024
constexpr int RDX_TILE = 4 * RDX_WAVE_TILE;           // 4096 elements: the largest input of the one-workgroup sort
// The multi-pass kernels take their tile size at run time: `items` rows of 64 per wave, tile = 256 * items elements.  The
// scatter pass keeps a regrouped tile in LDS (8 bytes per element + 5 KB of counters), so 4 workgroups fit a CU up to
// items = 17 (39 952 bytes each) and an MI355X holds RDX_SLOTS = 4 x 256 of them at once.  The host picks `items` so that the
// tiles fill a whole number of such rounds (radix_plan): the headline step's 4.26 M elements are 1041 tiles of 4096 --
// two rounds, the second for 17 tiles -- but 1024 tiles of 4160.
constexpr int RDX_ITEMS_MAX = 17;
constexpr int RDX_ITEMS_MIN = 4;
constexpr int RDX_SLOTS = 1024;
constexpr int RDX_INLINE_SCAN_TILES = 8;              // up to this many tiles the scatter pass sums the tile counts itself (measured: 2 tiles 40 -> 33 us per sort, 32 tiles even, 64 tiles 41 -> 45)

__host__ __device__ inline uint64_t rdx_pack(uint32_t key, uint32_t val) { return ((uint64_t)key << 32) | val; }
__host__ __device__ inline uint32_t rdx_key(uint64_t p) { return (uint32_t)(p >> 32); }
__host__ __device__ inline uint32_t rdx_val(uint64_t p) { return (uint32_t)p; }

// ---- sources of pass 0 (element i -> packed pair)
struct SrcPairs {                       // an already packed buffer (passes 1..)
  const uint64_t* p;
  __device__ __forceinline__ uint64_t operator()(int64_t i) const { return p[i]; }
};

// (item id, element number) of a step: element e = m * w + c, c = 0 the positive (when given), else negative c - off.
// A negative id is an empty slot: key n_items sorts behind every real row (its run is skipped by the consumers).
struct SrcStepIds {
  const int64_t* pos_ids;
  const int64_t* neg_ids;
  int64_t n_items;
  int32_t n, w, off;
  __device__ __forceinline__ uint64_t operator()(int64_t e) const {
    const int64_t m = (uint32_t)e / (uint32_t)w;       // e < 2^31 (checked by the host)
    const int c = (int)(e - m * w);
    int64_t id = (off && c == 0) ? pos_ids[m] : neg_ids[m * (int64_t)n + (c - off)];
    id = id < 0 ? n_items : (id >= n_items ? n_items - 1 : id);
    return rdx_pack((uint32_t)id, (uint32_t)e);
  }
};

// ALL sorted elements of an in-place SGD step in ONE key space: the t_items = M * w item elements of SrcStepIds (keys 0 ..
// n_items, n_items = the empty slot) followed by the step's M USER elements, element t_items + m with key user_key_base +
// user id (user_key_base = n_items + 1; a negative user id -> user_key_base + n_users, behind every real user).  Every
// user key is larger than every item key, so the sorted array is the item sort followed by the user sort, each exactly
// what its own sort would have produced (stable: equal keys stay in element order) -- the user rows' sort costs no
// launches of its own (nine of ~4 us for the 65 536 users of the headline step).
// DRAW: the source also DRAWS the negatives (draw_kind 1: UniformSampler, ids in [low, low + range); 2:
// PopularSamplerModel through its bucket lines), element f = m * n + j of the torch call exactly as rsa_sample_uniform /
// rsa_sample_popular draw it, and stores them to neg_out -- handed to the sort as the source of pass 0's HISTOGRAM launch
// (radix_sort_pairs2), which visits every element once; the scatter launch then reads the stored ids.
struct StepDraw {
  int64_t* neg_out;
  int32_t kind, lines_log2;
  PhiloxCall pc;
  uint64_t range;
  int64_t low;
  const float *lines, *table, *pop_prob;
};
template <bool DRAW>
struct SrcStepAll {
  SrcStepIds it;
  const int64_t* user_ids;       // nullable: no user elements
  int64_t t_items, n_users;
  uint32_t user_key_base;
  StepDraw dr;
  __device__ __forceinline__ uint64_t operator()(int64_t e) const {
    if (e >= t_items) {
      int64_t u = user_ids[e - t_items];
      u = u < 0 ? n_users : (u >= n_users ? n_users - 1 : u);
      return rdx_pack(user_key_base + (uint32_t)u, (uint32_t)e);
    }
    if constexpr (DRAW) {
      const int64_t m = (uint32_t)e / (uint32_t)it.w;
      const int c = (int)(e - m * it.w);
      int64_t id;
      if (it.off && c == 0) {
        id = it.pos_ids[m];
      } else {
        const int64_t f = m * (int64_t)it.n + (c - it.off);
        if (dr.kind == 1) {
          id = torch_randint_element(dr.pc, (uint64_t)f, dr.range, dr.low);
        } else {
          float pr;
          id = cdf_lookup_line(dr.lines, dr.lines_log2, dr.table, 1, dr.pop_prob, 1, it.n_items, torch_rand_element(dr.pc, (uint64_t)f), pr);
        }
        dr.neg_out[f] = id;
      }
      id = id < 0 ? it.n_items : (id >= it.n_items ? it.n_items - 1 : id);
      return rdx_pack((uint32_t)id, (uint32_t)e);
    } else {
      return it(e);
    }
  }
};

// x / d for 32-bit x by one multiply-high and a correction (magic = floor(2^32 / d); d == 1: magic = 2^32 - 1)
struct RdxDiv32 {
  uint32_t d, magic;
  __device__ __forceinline__ uint32_t div(uint32_t x) const {
    const uint32_t q = __umulhi(x, magic);
    return (x - q * d >= d) ? q + 1 : q;
  }
};
inline RdxDiv32 rdx_make_div32(uint64_t d) {
  RdxDiv32 v;
  v.d = (uint32_t)d;
  v.magic = d <= 1 ? 0xffffffffu : (uint32_t)((1ull << 32) / d);
  return v;
}

// Received exchange segments [n_seg][stride] of 8-byte keys (query << 32 | local row) behind RSA_SHARD_HDR header
// words: element = slot number; BY_QUERY: key = query index, else the local row; a slot outside its segment's live
// range gets `dead_key` (sorts last).  Elements past the `slots` segment slots (row sort only) are the step's POSITIVES,
// one per query: extra_rows[i] = local row of query i's positive on this rank, or < 0 when another rank owns it.
template <bool BY_QUERY>
struct SrcSegments {
  const int64_t* keys;
  const int64_t* extra_rows;
  int64_t slots;
  RdxDiv32 by_stride;
  uint32_t dead_key;
  __device__ __forceinline__ uint64_t operator()(int64_t i) const {
    uint32_t key = dead_key;
    if (i >= slots) {
      const int64_t r = extra_rows[i - slots];
      if (r >= 0) key = (uint32_t)r > dead_key ? dead_key : (uint32_t)r;
      return rdx_pack(key, (uint32_t)i);
    }
    const uint32_t seg = by_stride.div((uint32_t)i), within = (uint32_t)i - seg * by_stride.d;
    const int64_t k = keys[i];                        // unconditional (the slack is readable): not behind the header's round trip
    const int64_t live = keys[(size_t)seg * by_stride.d];
    if (within >= RSA_SHARD_HDR && (int64_t)(within - RSA_SHARD_HDR) < live) {
      key = BY_QUERY ? (uint32_t)((k >> 32) & 0x7fffffffll) : (uint32_t)(k & 0xffffffffll);
      key = key > dead_key ? dead_key : key;          // never index past the tables on a bad key
    }
    return rdx_pack(key, (uint32_t)i);
  }
};

// exclusive scan of one value per thread over a 256-thread workgroup; returns the exclusive prefix, *total = the sum
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* s_wave /*[4]*/, uint32_t* total) {
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const uint32_t s = s_wave[w];
    if (w < wave) base += s;
    tot += s;
  }
  if (total) *total = tot;
  __syncthreads();               // s_wave may be reused by the caller
  return base + inc - v;
}

template <class SRC>
__global__ __launch_bounds__(256) void radix_hist_kernel(const SRC src, int64_t total, int shift, int64_t n_tiles, int items,
                                                         uint32_t* __restrict__ counts) {
  __shared__ uint32_t h[RDX_BINS];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * (256 * items);
#pragma unroll 4
  for (int r = 0; r < items; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    if (i < total) atomicAdd(&h[(uint32_t)(src(i) >> shift) & (RDX_BINS - 1)], 1u);
  }
  __syncthreads();
  counts[(size_t)threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];       // digit-major: the scan is linear per digit
}

// one workgroup per digit: counts[d][0 .. n_tiles) -> exclusive prefix over the tiles; totals[d] = the digit's count
static __global__ __launch_bounds__(256) void radix_scan_kernel(uint32_t* __restrict__ counts, int64_t n_tiles,
                                                         uint32_t* __restrict__ totals) {
  __shared__ uint32_t s_wave[4];
  uint32_t* c = counts + (size_t)blockIdx.x * n_tiles;
  const int64_t per = (n_tiles + 255) / 256;
  const int64_t lo = threadIdx.x * per, hi = lo + per < n_tiles ? lo + per : n_tiles;
  uint32_t sum = 0;
  for (int64_t i = lo; i < hi; ++i) sum += c[i];
  uint32_t tot;
  uint32_t run = block_excl_scan_256(sum, s_wave, &tot);
  for (int64_t i = lo; i < hi; ++i) {
    const uint32_t t = c[i];
    c[i] = run;
    run += t;
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = tot;
}

template <class SRC>
__global__ __launch_bounds__(256) void radix_scatter_kernel(const SRC src, int64_t total, int shift, int64_t n_tiles, int items,
                                                            const uint32_t* __restrict__ counts,
                                                            const uint32_t* __restrict__ totals,
                                                            uint64_t* __restrict__ out) {
  // 39 952 bytes: four workgroups per CU (see RDX_ITEMS_MAX)
  __shared__ uint32_t cnt[4][RDX_BINS];      // per-wave digit counters while ranking, then wave w's first slot of digit d in the tile
  __shared__ uint32_t gbase[RDX_BINS];       // global position of digit d's first element of this tile, minus its slot (mod 2^32)
  __shared__ uint32_t s_wave[4];
  __shared__ uint64_t stage[256 * RDX_ITEMS_MAX];      // the tile regrouped by digit
  const int lane = lane_id(), wave = threadIdx.x >> 6;
#pragma unroll
  for (int w = 0; w < 4; ++w) cnt[w][threadIdx.x] = 0;
  __syncthreads();
  const int tile_cap = 256 * items;
  const int64_t tile0 = (int64_t)blockIdx.x * tile_cap;
  const int64_t base = tile0 + (int64_t)wave * (64 * items);
  uint64_t v[RDX_ITEMS_MAX];
  uint32_t loc[RDX_ITEMS_MAX];
#pragma unroll
  for (int r = 0; r < RDX_ITEMS_MAX; ++r) {
    const int64_t i = base + r * 64 + lane;
    v[r] = (r < items && i < total) ? src(i) : ~0ull;
  }
  volatile uint32_t* wc = cnt[wave];
  const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < RDX_ITEMS_MAX; ++r) {
    if (r < items) {                         // wave-uniform
      const bool valid = base + r * 64 + lane < total;
      const uint32_t d = (uint32_t)(v[r] >> shift) & (RDX_BINS - 1);
      uint64_t peers = __ballot(valid);
#pragma unroll
      for (int b = 0; b < RDX_DIGIT_BITS; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        peers &= bit ? bal : ~bal;
      }
      loc[r] = 0;
      if (valid) {
        const uint32_t c0 = wc[d];
        const uint32_t rank = (uint32_t)__popcll(peers & lt);
        loc[r] = c0 + rank;
        if (rank == 0) wc[d] = c0 + (uint32_t)__popcll(peers);       // the row's lowest lane of the digit adds the row's count
      }
    }
  }
  __syncthreads();
  {   // thread d: the digit's count in the tile, its first slot, the waves' first slots inside it
    const int d = threadIdx.x;
    const uint32_t c0 = cnt[0][d], c1 = cnt[1][d], c2 = cnt[2][d], c3 = cnt[3][d];
    const uint32_t first = block_excl_scan_256(c0 + c1 + c2 + c3, s_wave, nullptr);
    uint32_t before, all;            // digit d: elements in the tiles in front of this one / in all tiles
    if (totals != nullptr) {         // radix_scan_kernel has run: counts holds the prefixes
      before = counts[(size_t)d * n_tiles + blockIdx.x];
      all = totals[d];
    } else {                         // a few tiles (n_tiles <= RDX_INLINE_SCAN_TILES): counts is raw, summed here -- one launch fewer
      before = all = 0;
      const uint32_t* c = counts + (size_t)d * n_tiles;
      for (int t = 0; t < (int)n_tiles; ++t) {
        const uint32_t v = c[t];
        all += v;
        if (t < (int)blockIdx.x) before += v;
      }
    }
    const uint32_t gstart = block_excl_scan_256(all, s_wave, nullptr);      // elements of smaller digits, all tiles
    cnt[0][d] = first;
    cnt[1][d] = first + c0;
    cnt[2][d] = first + c0 + c1;
    cnt[3][d] = first + c0 + c1 + c2;
    gbase[d] = gstart + before - first;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RDX_ITEMS_MAX; ++r) {
    if (r < items && base + r * 64 + lane < total) {
      const uint32_t d = (uint32_t)(v[r] >> shift) & (RDX_BINS - 1);
      stage[cnt[wave][d] + loc[r]] = v[r];
    }
  }
  __syncthreads();
  const int64_t left = total - tile0;
  const int tile_n = left < tile_cap ? (int)left : tile_cap;
#pragma unroll 4
  for (int k = 0; k < items; ++k) {
    const int j = k * 256 + threadIdx.x;
    if (j < tile_n) {
      const uint64_t p = stage[j];
      const uint32_t d = (uint32_t)(p >> shift) & (RDX_BINS - 1);
      out[(uint32_t)(gbase[d] + (uint32_t)j)] = p;
    }
  }
}


// total <= RDX_TILE: ONE workgroup sorts the whole input in LDS, all passes in one launch (a 4096-element sort through
// the three-kernels-per-pass form is nine launches of ~5 us each for microseconds of work: the user rows of a sharded
// step, every sort of a small-batch `fit`).  The same ranking as radix_scatter_kernel over 16 waves of 256 elements (four
// waves per SIMD hide the LDS round trips of the ranking; four waves of 1024 elements took 33 us for three passes);
// between passes the tile is re-read from the staging buffer in order.
constexpr int RDX_SMALL_WAVES = 16;
constexpr int RDX_SMALL_ROWS = RDX_TILE / (64 * RDX_SMALL_WAVES);      // 4 rows of 64 per wave

template <class SRC>
__global__ __launch_bounds__(64 * RDX_SMALL_WAVES) void radix_small_kernel(const SRC src, int total, int passes,
                                                                           uint64_t* __restrict__ out) {
  __shared__ uint32_t cnt[RDX_SMALL_WAVES][RDX_BINS];      // per-wave digit counters, then the waves' bases inside a digit
  __shared__ uint32_t dbase[RDX_BINS];
  __shared__ uint32_t s_wave[4];
  __shared__ uint64_t stage[RDX_TILE];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int base = wave * (64 * RDX_SMALL_ROWS);
  uint64_t v[RDX_SMALL_ROWS];
  uint32_t loc[RDX_SMALL_ROWS];
#pragma unroll
  for (int r = 0; r < RDX_SMALL_ROWS; ++r) {
    const int i = base + r * 64 + lane;
    v[r] = i < total ? src((int64_t)i) : ~0ull;
  }
  const uint64_t lt = (1ull << lane) - 1ull;
  for (int p = 0; p < passes; ++p) {
    const int shift = 32 + p * RDX_DIGIT_BITS;
#pragma unroll
    for (int k = 0; k < RDX_BINS / 64; ++k) cnt[wave][k * 64 + lane] = 0;
    __syncthreads();
    volatile uint32_t* wc = cnt[wave];
#pragma unroll
    for (int r = 0; r < RDX_SMALL_ROWS; ++r) {
      const bool valid = base + r * 64 + lane < total;
      const uint32_t d = (uint32_t)(v[r] >> shift) & (RDX_BINS - 1);
      uint64_t peers = __ballot(valid);
#pragma unroll
      for (int b = 0; b < RDX_DIGIT_BITS; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        peers &= bit ? bal : ~bal;
      }
      loc[r] = 0;
      if (valid) {
        const uint32_t c0 = wc[d];
        const uint32_t rank = (uint32_t)__popcll(peers & lt);
        loc[r] = c0 + rank;
        if (rank == 0) wc[d] = c0 + (uint32_t)__popcll(peers);
      }
    }
    __syncthreads();
    uint32_t inc = 0, tot_d = 0;
    if (threadIdx.x < RDX_BINS) {     // thread d: the waves' bases inside digit d, then the digits' first slots
      const int d = threadIdx.x;
#pragma unroll
      for (int w = 0; w < RDX_SMALL_WAVES; ++w) {
        const uint32_t c = cnt[w][d];
        cnt[w][d] = tot_d;
        tot_d += c;
      }
      inc = tot_d;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
      }
      if (lane == 63) s_wave[wave] = inc;
    }
    __syncthreads();
    if (threadIdx.x < RDX_BINS) {
      uint32_t before = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if (w < wave) before += s_wave[w];
      dbase[threadIdx.x] = before + inc - tot_d;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RDX_SMALL_ROWS; ++r) {
      if (base + r * 64 + lane < total) {
        const uint32_t d = (uint32_t)(v[r] >> shift) & (RDX_BINS - 1);
        stage[dbase[d] + cnt[wave][d] + loc[r]] = v[r];
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RDX_SMALL_ROWS; ++r) {        // the regrouped tile, in order: the next pass's input
      const int i = base + r * 64 + lane;
      v[r] = i < total ? stage[i] : ~0ull;
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < RDX_SMALL_ROWS; ++r) {
    const int i = base + r * 64 + lane;
    if (i < total) out[i] = v[r];
  }
}

// ---- host side
inline unsigned radix_key_bits(int64_t n_keys) {      // bits needed for keys 0 .. n_keys - 1
  unsigned b = 1;
  while (b < 32 && (1ll << b) < n_keys) ++b;
  return b;
}
inline int radix_passes(unsigned bits) { return (int)((bits + RDX_DIGIT_BITS - 1) / RDX_DIGIT_BITS); }
// tile size of the multi-pass kernels: the fewest rounds of RDX_SLOTS co-resident workgroups, then the smallest tile
// that still fits the elements into those rounds
inline int radix_items(int64_t total) {
  const int64_t round_cap = (int64_t)RDX_SLOTS * 256 * RDX_ITEMS_MAX;
  const int64_t rounds = total <= round_cap ? 1 : (total + round_cap - 1) / round_cap;
  const int64_t per = rounds * RDX_SLOTS * 256;
  const int64_t items = (total + per - 1) / per;
  return (int)(items < RDX_ITEMS_MIN ? RDX_ITEMS_MIN : items);
}
inline int64_t radix_tiles(int64_t total) {
  const int64_t tile = 256 * (int64_t)radix_items(total);
  return (total + tile - 1) / tile;
}
// Counter space for a sort of ANY total <= max_total.  radix_tiles() is not monotone in the total (the tile size steps up at
// every whole round of the chip), and callers size one workspace for the largest step and then sort fewer elements in it (the
// owner side of the sharded backward: slots vs slots + queries) -- so the size is taken from an upper bound of the tile count
// that IS monotone: tiles(t) <= t / (256 * RDX_ITEMS_MIN) and tiles(t) <= rounds(t) * RDX_SLOTS.
inline int64_t radix_tiles_upper(int64_t max_total) {
  if (max_total <= 0) return 1;
  const int64_t by_min_tile = (max_total + 256 * (int64_t)RDX_ITEMS_MIN - 1) / (256 * (int64_t)RDX_ITEMS_MIN);
  const int64_t round_cap = (int64_t)RDX_SLOTS * 256 * RDX_ITEMS_MAX;
  const int64_t by_rounds = ((max_total + round_cap - 1) / round_cap) * RDX_SLOTS;
  return by_min_tile < by_rounds ? by_min_tile : by_rounds;
}
inline int64_t radix_temp_bytes(int64_t max_total) {
  const int64_t c = RDX_BINS * radix_tiles_upper(max_total) * 4;
  return (c + 255) / 256 * 256 + RDX_BINS * 4;
}

// Sorts `total` pairs by their `bits` low key bits.  Pass 0 reads `src0` and writes buf_a, pass 1 reads buf_a and writes
// buf_b, ...: the result is in radix_result(buf_a, buf_b, bits).  `temp`: radix_temp_bytes(total) bytes.
inline uint64_t* radix_result(uint64_t* buf_a, uint64_t* buf_b, unsigned bits) { return (radix_passes(bits) & 1) ? buf_a : buf_b; }

// `src0h` feeds pass 0's histogram launch, `src0` everything else that reads the producer (pass 0's scatter launch, the
// one-workgroup sort): the same pairs from both -- a source with a side effect (SrcStepAll<true> DRAWS a step's negatives
// and stores them) is handed in as `src0h` and runs exactly once per element, before any launch that reads its output.
template <class SRC0H, class SRC0>
inline hipError_t radix_sort_pairs2(const SRC0H& src0h, const SRC0& src0, uint64_t* buf_a, uint64_t* buf_b, int64_t total,
                                    unsigned bits, void* temp, hipStream_t s);

template <class SRC0>
inline hipError_t radix_sort_pairs(const SRC0& src0, uint64_t* buf_a, uint64_t* buf_b, int64_t total, unsigned bits,
                                   void* temp, hipStream_t s) {
  return radix_sort_pairs2(src0, src0, buf_a, buf_b, total, bits, temp, s);
}

template <class SRC0H, class SRC0>
inline hipError_t radix_sort_pairs2(const SRC0H& src0h, const SRC0& src0, uint64_t* buf_a, uint64_t* buf_b, int64_t total,
                                    unsigned bits, void* temp, hipStream_t s) {
  if (total <= 0) return hipSuccess;
  if (total <= RDX_TILE) {       // one workgroup, one launch; the result goes where the multi-pass form would leave it
    if constexpr (!std::is_same<SRC0H, SRC0>::value) return hipErrorInvalidValue;      // (a drawing source needs the multi-pass form)
    hipLaunchKernelGGL((radix_small_kernel<SRC0>), dim3(1), dim3(64 * RDX_SMALL_WAVES), 0, s, src0, (int)total, radix_passes(bits),
                       radix_result(buf_a, buf_b, bits));
    return hipGetLastError();
  }
  const int64_t n_tiles = radix_tiles(total);
  const int items = radix_items(total);
  if (n_tiles > radix_tiles_upper(total)) return hipErrorInvalidValue;      // (cannot happen: the bound above)
  uint32_t* counts = reinterpret_cast<uint32_t*>(temp);
  uint32_t* totals = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(temp) + (RDX_BINS * n_tiles * 4 + 255) / 256 * 256);
  const int passes = radix_passes(bits);
  const dim3 grid((unsigned)n_tiles), block(256);
  const bool inline_scan = n_tiles <= RDX_INLINE_SCAN_TILES;
  const uint32_t* tot = inline_scan ? nullptr : totals;
  for (int p = 0; p < passes; ++p) {
    const int shift = 32 + p * RDX_DIGIT_BITS;
    uint64_t* dst = (p & 1) ? buf_b : buf_a;
    if (p == 0) {
      hipLaunchKernelGGL((radix_hist_kernel<SRC0H>), grid, block, 0, s, src0h, total, shift, n_tiles, items, counts);
      if (!inline_scan) hipLaunchKernelGGL(radix_scan_kernel, dim3(RDX_BINS), block, 0, s, counts, n_tiles, totals);
      hipLaunchKernelGGL((radix_scatter_kernel<SRC0>), grid, block, 0, s, src0, total, shift, n_tiles, items, counts, tot, dst);
    } else {
      const SrcPairs sp{(p & 1) ? buf_a : buf_b};
      hipLaunchKernelGGL((radix_hist_kernel<SrcPairs>), grid, block, 0, s, sp, total, shift, n_tiles, items, counts);
      if (!inline_scan) hipLaunchKernelGGL(radix_scan_kernel, dim3(RDX_BINS), block, 0, s, counts, n_tiles, totals);
      hipLaunchKernelGGL((radix_scatter_kernel<SrcPairs>), grid, block, 0, s, sp, total, shift, n_tiles, items, counts, tot, dst);
    }
  }
  return hipGetLastError();
}

}  // namespace rsa
