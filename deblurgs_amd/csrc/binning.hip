// binning.hip -- prefix sum, duplicateWithKeys, stable LSD radix sort, identifyTileRanges
// (replaces cub::DeviceScan::InclusiveSum, duplicateWithKeys, cub::DeviceRadixSort::SortPairs and
// identifyTileRanges: rasterizer_impl.cu:70-138,166,188-191,283,295-324).
//
// All integer work, HBM-bound.  Key = ((k*T + tile) << 32) | depth_bits so that ONE sort orders the
// duplicates of all K subframes; within a subframe the low 32+bits(T) bits are exactly the reference's key.
#include <stdlib.h>
#include <string.h>

#include "dgs_common.h"

// n / d and n % d for n < 2^24, 1 <= d < 2^12 (slots of a tile rectangle by its width) from a float reciprocal:
// 3 conversions / multiplies, one 24-bit multiply and a fix-up, instead of the ~22 instructions (six of them quarter-rate
// 32-bit multiplies) of a u32 division.  float(n) is exact and the quotient is below 2^12, so the estimate is off by at
// most one; the fix-up makes the result exact.
__device__ __forceinline__ void dgs_divmod_u24(uint32_t n, uint32_t d, float inv_d, uint32_t& q, uint32_t& r) {
  uint32_t qe = (uint32_t)((float)n * inv_d);
  int rr = (int)n - (int)__umul24(qe, d);
  if (rr < 0) {
    qe -= 1u;
    rr += (int)d;
  } else if (rr >= (int)d) {
    qe += 1u;
    rr -= (int)d;
  }
  q = qe;
  r = (uint32_t)rr;
}

namespace {

// ------------------------------------------------------------------------------------------------ scan
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 4096 elements per block

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t n = __shfl_up(v, d, 64);
    if (dgs_lane() >= d) v += n;
  }
  return v;
}

// exclusive scan of one value per thread across a 256-thread block; returns the exclusive prefix, *total = sum
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* total, uint32_t* lds /*[8]*/) {
  const int lane = dgs_lane(), w = threadIdx.x >> 6;
  uint32_t incl = wave_incl_scan(v);
  if (lane == 63) lds[w] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < SCAN_THREADS / 64; i++) {
    uint32_t s = lds[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

// exclusive scan of one value per thread across an NT-thread block; returns the exclusive prefix, *total = sum
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan_n(uint32_t v, uint32_t* total, uint32_t* lds /*[NT / 64]*/) {
  const int lane = dgs_lane(), w = threadIdx.x >> 6;
  uint32_t incl = wave_incl_scan(v);
  if (lane == 63) lds[w] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NT / 64; i++) {
    uint32_t s = lds[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

__global__ void __launch_bounds__(SCAN_THREADS)
scan_reduce_kernel(const uint32_t* __restrict__ in, uint64_t n, uint32_t* __restrict__ block_sums) {
  __shared__ uint32_t lds[8];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t s = 0;
  if (base + SCAN_ITEMS <= n) {
    const uint4* p = reinterpret_cast<const uint4*>(in + base);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS / 4; i++) {
      uint4 q = p[i];
      s += q.x + q.y + q.z + q.w;
    }
  } else {
    for (int i = 0; i < SCAN_ITEMS; i++)
      if (base + i < n) s += in[base + i];
  }
  uint32_t tot;
  block_excl_scan(s, &tot, lds);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single block: exclusive scan of the block sums in place, grand total to *total
__global__ void __launch_bounds__(SCAN_THREADS)
scan_top_kernel(uint32_t* __restrict__ block_sums, uint64_t nb, uint32_t* __restrict__ total) {
  __shared__ uint32_t lds[8];
  uint32_t carry = 0;
  uint64_t wide = 0;  // the grand total in 64 bits: total[1] != 0 tells the caller that the u32 offsets wrapped
  for (uint64_t start = 0; start < nb; start += SCAN_TILE) {
    const uint64_t base = start + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t vals[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
      vals[i] = (base + i < nb) ? block_sums[base + i] : 0u;
      s += vals[i];
    }
    uint32_t tot;
    uint32_t pre = block_excl_scan(s, &tot, lds) + carry;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
      if (base + i < nb) block_sums[base + i] = pre;
      pre += vals[i];
    }
    carry += tot;
    wide += tot;
  }
  if (threadIdx.x == 0 && total != nullptr) {
    total[0] = carry;
    total[1] = (uint32_t)(wide >> 32);
  }
}

__global__ void __launch_bounds__(SCAN_THREADS)
scan_apply_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n,
                  const uint32_t* __restrict__ block_offsets) {
  __shared__ uint32_t lds[8];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t vals[SCAN_ITEMS];
  uint32_t s = 0;
  if (base + SCAN_ITEMS <= n) {
    const uint4* p = reinterpret_cast<const uint4*>(in + base);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS / 4; i++) {
      uint4 q = p[i];
      vals[4 * i] = q.x;
      vals[4 * i + 1] = q.y;
      vals[4 * i + 2] = q.z;
      vals[4 * i + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) vals[i] = (base + i < n) ? in[base + i] : 0u;
  }
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) s += vals[i];
  uint32_t tot;
  uint32_t pre = block_excl_scan(s, &tot, lds) + block_offsets[blockIdx.x];
  if (base + SCAN_ITEMS <= n) {
    uint4* p = reinterpret_cast<uint4*>(out + base);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS / 4; i++) {
      uint4 q;
      q.x = pre;
      pre += vals[4 * i];
      q.y = pre;
      pre += vals[4 * i + 1];
      q.z = pre;
      pre += vals[4 * i + 2];
      q.w = pre;
      pre += vals[4 * i + 3];
      p[i] = q;
    }
  } else {
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
      if (base + i < n) out[base + i] = pre;
      pre += vals[i];
    }
  }
}

// Small and medium inputs (up to SCAN_SELF_MAX blocks = 16.7 M elements): no top kernel -- every block of the apply
// pass adds up the sums of the blocks before it by itself (a few KB from L2), the last block publishes the 64-bit grand
// total.  One launch less per scan; at DeblurGS-sized scenes a scan is three few-microsecond kernels, so that is a third
// of its cost.
constexpr uint32_t SCAN_SELF_MAX = 4096;

__global__ void __launch_bounds__(SCAN_THREADS)
scan_apply_self_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n,
                       const uint32_t* __restrict__ block_sums, uint32_t nb, uint32_t* __restrict__ total) {
  __shared__ uint32_t lds[8];
  __shared__ unsigned long long s_wide[SCAN_THREADS / 64];
  __shared__ uint32_t s_base;
  // exclusive offset of this block = sum of block_sums[0 .. blockIdx.x)
  unsigned long long mine = 0;
  const uint32_t upto = (blockIdx.x == nb - 1 && total != nullptr) ? nb : blockIdx.x;   // the last block also needs all
  for (uint32_t i = threadIdx.x; i < upto; i += SCAN_THREADS) mine += block_sums[i];
  for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor((long long)mine, d, 64);
  if (dgs_lane() == 0) s_wide[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long all = 0;
    for (int w = 0; w < SCAN_THREADS / 64; w++) all += s_wide[w];
    if (blockIdx.x == nb - 1 && total != nullptr) {
      total[0] = (uint32_t)all;
      total[1] = (uint32_t)(all >> 32);
      all -= block_sums[nb - 1];
    }
    s_base = (uint32_t)all;
  }
  __syncthreads();
  const uint32_t block_base = s_base;
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t vals[SCAN_ITEMS];
  uint32_t s = 0;
  if (base + SCAN_ITEMS <= n) {
    const uint4* p = reinterpret_cast<const uint4*>(in + base);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS / 4; i++) {
      uint4 q = p[i];
      vals[4 * i] = q.x;
      vals[4 * i + 1] = q.y;
      vals[4 * i + 2] = q.z;
      vals[4 * i + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) vals[i] = (base + i < n) ? in[base + i] : 0u;
  }
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) s += vals[i];
  uint32_t tot;
  uint32_t pre = block_excl_scan(s, &tot, lds) + block_base;
  if (base + SCAN_ITEMS <= n) {
    uint4* p = reinterpret_cast<uint4*>(out + base);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS / 4; i++) {
      uint4 q;
      q.x = pre;
      pre += vals[4 * i];
      q.y = pre;
      pre += vals[4 * i + 1];
      q.z = pre;
      pre += vals[4 * i + 2];
      q.w = pre;
      pre += vals[4 * i + 3];
      p[i] = q;
    }
  } else {
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
      if (base + i < n) out[base + i] = pre;
      pre += vals[i];
    }
  }
}

// ------------------------------------------------------------------------------ depth-ordered duplication
// The duplicates only need a STABLE sort by (k, tile) if they are generated in (k, depth, index) order: the
// low 32 key bits (depth) are then already in order inside every (k, tile) group, exactly as if the LSD passes
// over the depth bits had run.  So the K*P Gaussians are sorted by (k, depth_bits) first (their (k << 32 | depth)
// keys are written by the preprocess kernel itself) (15 M pairs instead of
// R = 63 M, 4 digit passes) and the big sort shrinks from 6 passes over all 49 bits to 2 passes over the
// bits(K*T) = 17 tile bits.  Sorted keys and point list are bit-identical to the one-big-sort result.
__global__ void __launch_bounds__(256)
gather_u32_kernel(uint64_t n, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ src,
                  uint32_t* __restrict__ dst) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < n) dst[j] = src[idx[j]];
}

// Load-balanced expansion: a wave takes 64 consecutive (k, Gaussian) pairs of the (k, depth, index) order, whose
// duplicate segments are contiguous (offs_sorted is an exclusive scan in that order), and its lanes emit the
// wave's duplicates cooperatively -- duplicate d of the wave is produced by lane d % 64, which finds its source
// pair by a binary search over the 64 segment starts held in LDS.  Every key/value store instruction therefore
// writes 64 consecutive elements (the thread-per-pair loop of the reference, rasterizer_impl.cu:92-108, writes
// 64 scattered 8-byte words per instruction and idles the lanes with small rectangles).
__global__ void __launch_bounds__(256)
duplicate_sorted_kernel(DgsView v, DgsRow* __restrict__ rows, const uint32_t* __restrict__ order,
                        const uint32_t* __restrict__ tt_sorted, const uint32_t* __restrict__ offs_sorted,
                        uint32_t* __restrict__ point_offsets, uint64_t* __restrict__ keys,
                        uint32_t* __restrict__ vals, uint32_t cap) {
  __shared__ uint32_t s_off[4][64];    // segment start relative to the wave's first duplicate
  __shared__ uint32_t s_rect[4][64];   // minx | miny << 12 | width << 24  (grid <= 4095 tiles per side, width <= 255)
  __shared__ uint32_t s_wide[4][64];   // full width for rectangles wider than 255 tiles
  __shared__ float s_invw[4][64];      // 1 / width (dgs_divmod_u24)
  __shared__ uint32_t s_tb[4][64];     // k * T
  __shared__ uint32_t s_db[4][64];     // depth bits
  __shared__ uint32_t s_g[4][64];      // Gaussian index
  const int lane = dgs_lane(), w = threadIdx.x >> 6;
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t n = (uint64_t)v.K * v.P;
  const bool in = j < n;
  uint32_t off = 0, nt = 0;
  if (in) {
    const uint32_t i = order[j];
    off = offs_sorted[j];
    nt = tt_sorted[j];
    uint32_t rect = 0, wide = 0, tb = 0, db = 0, g = 0;
    if (nt != 0) {
      const uint32_t k = i / (uint32_t)v.P;
      g = i - k * (uint32_t)v.P;
      const DgsRow* row = rows + i;
      point_offsets[i] = off;   // first duplicate of this (k, Gaussian): read by the backward (natural index)
      int minx, miny, maxx, maxy;
      dgs_get_rect(row->x, row->y, row->radius, v.gx, v.gy, minx, miny, maxx, maxy);
      wide = (uint32_t)(maxx - minx);
      rect = (uint32_t)minx | ((uint32_t)miny << 12);
      tb = k * (uint32_t)v.T;
      db = __float_as_uint(row->depth);
    }
    s_rect[w][lane] = rect;
    s_wide[w][lane] = wide;
    s_invw[w][lane] = __builtin_amdgcn_rcpf((float)wide);
    s_tb[w][lane] = tb;
    s_db[w][lane] = db;
    s_g[w][lane] = g;
  }
  // offsets are contiguous across the wave's valid lanes: base = first lane's offset, total = last end - base
  const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
  s_off[w][lane] = in ? off - base : 0xFFFFFFFFu;
  const uint64_t valid = __ballot(in);
  if (valid == 0ull) return;                      // wave entirely past the end (wave-uniform)
  const int last = 63 - __builtin_clzll(valid);
  const uint32_t total =
      (uint32_t)__builtin_amdgcn_readlane((int)(off + nt), last) - base;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (uint32_t d = (uint32_t)lane; d < total; d += 64) {
    // largest s with s_off[s] <= d (segments of pairs without tiles are empty and are skipped by the search)
    int lo = 0;
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1) {
      const int mid = lo + step;
      if (mid < 64 && s_off[w][mid] <= d) lo = mid;
    }
    const uint32_t local = d - s_off[w][lo];
    const uint32_t width = s_wide[w][lo];
    const uint32_t rect = s_rect[w][lo];
    uint32_t ry, rx;
    dgs_divmod_u24(local, width, s_invw[w][lo], ry, rx);
    const uint32_t tile = s_tb[w][lo] + __umul24((rect >> 12) + ry, (uint32_t)v.gx) + (rect & 0xFFFu) + rx;
    if (base + d < cap) {   // cap = capacity of the duplicate arrays (the exact count unless the caller sized them ahead)
      keys[base + d] = ((uint64_t)tile << 32) | s_db[w][lo];
      vals[base + d] = s_g[w][lo];
    }
  }
}

// ------------------------------------------------------------------------------- tile_cull duplication
// The reference emits one duplicate per tile of the 3-sigma bounding rectangle (rasterizer_impl.cu:76-108), but
// on the metric scene 42 % of those can never reach alpha >= 1/255 inside their tile: the compositing kernels
// skip them at every pixel (forward.cu:356-358), after they have been sorted, range-marked, gathered and tested.
// With tile_cull the exact ellipse-vs-tile test (dgs_cull_hit, conservative) runs once per rectangle slot and only the
// hits become duplicates:
//   COUNT (cull_count_kernel) walks the pairs in NATURAL order -- coalesced row reads, no global offsets: a wave's 64
//     pairs share out their rectangle slots by a wave-local scan -- and leaves one 16-byte record per pair: the
//     rectangle, the number of hits and the hit bits of the first 64 slots (94 % of the visible pairs at the metric
//     config have no more).
//   The counts are gathered into (k, depth, index) order and scanned.
//   EMIT (cull_emit_kernel) walks the pairs in that order, gathers their records and writes the surviving
//     duplicates compacted: for a pair of at most 64 slots the stored bits ARE the test; a larger rectangle is tested
//     again (identical instructions -- this file is built with -ffp-contract=off).
// The low key word carries the duplicate's own index u (its contribution-row slot for the backward) instead of the depth
// bits, which the tile-bits-only stable sort never looks at.
constexpr uint32_t CULL_BIG = 0x80000000u;   // record.x: rectangle of more than 64 slots (otherwise minx | miny << 12 | (width - 1) << 24)
constexpr int CULL_CH = 32;                  // rounds (of 64 slots) per refill of the segment-start bit table

__device__ __forceinline__ void cull_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// inclusive prefix sum over the 64 lanes: DPP row shifts inside each row of 16, row_bcast15 / row_bcast31 across rows
// (6 VALU adds; a __shfl_up ladder is 6 dependent LDS round trips -- and a wave here has only ~4 rounds of work to hide them)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t cull_dpp_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);   // no source lane: 0
}
__device__ __forceinline__ uint32_t cull_wave_incl_scan(uint32_t x, int) {
  x += cull_dpp_u32<0x111, 0xf>(x);   // row_shr:1
  x += cull_dpp_u32<0x112, 0xf>(x);   // row_shr:2
  x += cull_dpp_u32<0x114, 0xf>(x);   // row_shr:4
  x += cull_dpp_u32<0x118, 0xf>(x);   // row_shr:8
  x += cull_dpp_u32<0x142, 0xa>(x);   // row_bcast15 -> rows 1, 3
  x += cull_dpp_u32<0x143, 0xc>(x);   // row_bcast31 -> rows 2, 3
  return x;
}

// Slot -> pair mapping of one wave.  The wave's non-empty pairs are compacted (index c), pair c owns the slots
// [excl_c, excl_c + work_c) of the wave.  Instead of a binary search per slot, every pair sets the bit of its first slot
// in an LDS bit table (one 64-bit word per round of 64 slots, refilled every CULL_CH rounds); lane l of round r then owns
// pair  (pairs started before the round) + popcount(bits of the round at or below l) - 1  -- one v_mbcnt pair.
// body(d, act, pi, own, m): d = slot, act = d < total, pi = compacted pair, own = this lane holds a first slot,
// m = the round's start bits.
template <typename F>
__device__ __forceinline__ void cull_walk(unsigned long long* s_bits, int lane, bool nonempty, uint32_t excl,
                                          uint32_t total, F&& body) {
  for (uint32_t cb = 0; cb < total; cb += (uint32_t)CULL_CH * 64u) {
    if (lane < CULL_CH) s_bits[lane] = 0ull;
    cull_wave_sync();
    if (nonempty && excl >= cb && excl - cb < (uint32_t)CULL_CH * 64u) {
      const uint32_t rel = excl - cb;
      atomicOr(reinterpret_cast<uint32_t*>(s_bits) + (rel >> 5), 1u << (rel & 31u));
    }
    cull_wave_sync();
    uint32_t run_pairs = (uint32_t)__builtin_popcountll(__ballot(nonempty && excl < cb));
    const uint32_t cend = min(total, cb + (uint32_t)CULL_CH * 64u);
    for (uint32_t d0 = cb; d0 < cend; d0 += 64) {
      const unsigned long long mv = s_bits[(d0 - cb) >> 6];
      const uint32_t mlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)mv);
      const uint32_t mhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(mv >> 32));
      const uint64_t m = ((uint64_t)mhi << 32) | mlo;
      const uint32_t below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
      const bool own = ((m >> lane) & 1ull) != 0ull;
      const uint32_t d = d0 + (uint32_t)lane;
      body(d, d < total, (int)(run_pairs + below + (own ? 1u : 0u)) - 1, own, m);
      run_pairs += (uint32_t)__builtin_popcountll(m);
    }
  }
}

__global__ void __launch_bounds__(256)
cull_count_kernel(DgsView v, const DgsRow* __restrict__ rows, const uint32_t* __restrict__ tiles_touched,
                  uint4* __restrict__ recs, uint32_t* __restrict__ cnts) {
  __shared__ unsigned long long s_bits[4][CULL_CH];
  __shared__ uint4 s_a[4][64];    // first slot, minx | miny << 12, width | flags << 16 (1 = always, 2 = never), 1 / width
  __shared__ float4 s_q[4][64];   // x, y, a, b
  __shared__ float4 s_r[4][64];   // c, 1/a, 1/c, r2
  __shared__ uint32_t s_cnt[4][64];
  __shared__ unsigned long long s_mask[4][64];
  const int lane = dgs_lane(), w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t n = (uint64_t)v.K * v.P;
  const bool in = i < n;
  const uint32_t nt = in ? tiles_touched[i] : 0u;
  const bool nonempty = nt != 0u;
  const uint64_t nm = __ballot(nonempty);
  if (nm == 0ull) {   // wave-uniform: nothing visible here
    if (in) {
      recs[i] = make_uint4(0u, 0u, 0u, 0u);
      cnts[i] = 0u;
    }
    return;
  }
  const uint32_t c = __builtin_amdgcn_mbcnt_hi((uint32_t)(nm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nm, 0u));
  const uint32_t incl = cull_wave_incl_scan(nt, lane);
  const uint32_t excl = incl - nt;
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  uint32_t desc = 0;
  if (nonempty) {
    const DgsRow* row = rows + i;
    const float4 A = reinterpret_cast<const float4*>(row)[0];  // x, y, cx, cy
    const float4 B = reinterpret_cast<const float4*>(row)[1];  // cz, op, ...
    int minx, miny, maxx, maxy;
    dgs_get_rect(A.x, A.y, row->radius, v.gx, v.gy, minx, miny, maxx, maxy);
    const uint32_t wide = (uint32_t)(maxx - minx);
    const uint32_t rect = (uint32_t)minx | ((uint32_t)miny << 12);
    const DgsCull cg = dgs_make_cull(A.z, A.w, B.x, B.y);
    const uint32_t fl = (cg.always ? 1u : 0u) | (cg.never ? 2u : 0u);
    s_a[w][c] = make_uint4(excl, rect, wide | (fl << 16), __float_as_uint(__builtin_amdgcn_rcpf((float)wide)));
    s_q[w][c] = make_float4(A.x, A.y, cg.a, cg.b);
    s_r[w][c] = make_float4(cg.c, cg.inv_a, cg.inv_c, cg.r2);
    s_cnt[w][c] = 0u;
    s_mask[w][c] = 0ull;
    desc = (nt > 64u) ? CULL_BIG : (rect | ((wide - 1u) << 24));
  }
  cull_walk(s_bits[w], lane, nonempty, excl, total, [&](uint32_t d, bool act, int pi, bool own, uint64_t m) {
    bool hit = false;
    uint32_t local = 0;
    if (act) {
      const uint4 a = s_a[w][pi];
      const float4 q = s_q[w][pi], r = s_r[w][pi];
      local = d - a.x;
      uint32_t ry, rx;
      dgs_divmod_u24(local, a.z & 0xFFFFu, __uint_as_float(a.w), ry, rx);
      const uint32_t tx = (a.y & 0xFFFu) + rx, ty = (a.y >> 12) + ry;
      DgsCull cg;
      cg.a = q.z; cg.b = q.w; cg.c = r.x; cg.inv_a = r.y; cg.inv_c = r.z; cg.r2 = r.w;
      cg.always = (a.z & 0x10000u) != 0u; cg.never = (a.z & 0x20000u) != 0u;
      // d = mean - pixel over the tile's pixel centres [16 t, 16 t + 15]
      const float ex = q.x - (float)(tx * DGS_TILE), ey = q.y - (float)(ty * DGS_TILE);
      hit = dgs_cull_hit(cg, ex - (float)(DGS_TILE - 1), ex, ey - (float)(DGS_TILE - 1), ey);
    }
    const uint64_t hm = __ballot(hit);
    // one lane per pair and round (the pair's first slot of the round) books the pair's hits of the round
    if (act && (lane == 0 || own)) {
      const uint64_t rest = (lane == 63) ? 0ull : (m >> (lane + 1));
      const int len = rest ? (__builtin_ctzll(rest) + 1) : (64 - lane);
      const uint64_t seg = (len == 64) ? ~0ull : ((1ull << len) - 1ull);
      const uint64_t chunk = (hm >> lane) & seg;
      s_cnt[w][pi] += (uint32_t)__builtin_popcountll(chunk);
      if (local < 64u) s_mask[w][pi] |= chunk << local;
    }
  });
  cull_wave_sync();
  if (in) {
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    if (nonempty) {
      const uint64_t mk = s_mask[w][c];
      rec = make_uint4(desc, s_cnt[w][c], (uint32_t)mk, (uint32_t)(mk >> 32));
    }
    recs[i] = rec;
    cnts[i] = rec.y;   // dense copy: the input of the depth-order gather + scan
  }
}

template <bool ANYBIG>
__device__ __forceinline__ void cull_emit_rounds(const DgsView& v, unsigned long long* s_bits, const uint4* s_a,
                                                 const uint4* s_b, const float4* s_q, const float4* s_r, int lane,
                                                 bool live, uint32_t excl, uint32_t total, uint32_t obase, uint32_t cap,
                                                 uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  uint32_t run = 0;   // hits emitted by earlier rounds (wave-uniform)
  cull_walk(s_bits, lane, live, excl, total, [&](uint32_t d, bool act, int pi, bool own, uint64_t m) {
    bool hit = false;
    uint32_t tile = 0, g = 0;
    if (act) {
      const uint4 a = s_a[pi];
      const uint4 b = s_b[pi];
      const uint32_t local = d - a.x;
      const bool big = (a.z & 0x40000u) != 0u;
      g = b.y;
      uint32_t ry = 0, rx = 0;
      if (!ANYBIG || !big) {
        const uint64_t mk = ((uint64_t)b.w << 32) | b.z;
        hit = ((mk >> local) & 1ull) != 0ull;
        if (hit) dgs_divmod_u24(local, a.z & 0xFFFFu, __uint_as_float(a.w), ry, rx);
      } else {
        dgs_divmod_u24(local, a.z & 0xFFFFu, __uint_as_float(a.w), ry, rx);
        const float4 q = s_q[pi], r = s_r[pi];
        DgsCull cg;
        cg.a = q.z; cg.b = q.w; cg.c = r.x; cg.inv_a = r.y; cg.inv_c = r.z; cg.r2 = r.w;
        cg.always = (a.z & 0x10000u) != 0u; cg.never = (a.z & 0x20000u) != 0u;
        const uint32_t tx = (a.y & 0xFFFu) + rx, ty = (a.y >> 12) + ry;
        const float ex = q.x - (float)(tx * DGS_TILE), ey = q.y - (float)(ty * DGS_TILE);
        hit = dgs_cull_hit(cg, ex - (float)(DGS_TILE - 1), ex, ey - (float)(DGS_TILE - 1), ey);
      }
      tile = b.x + __umul24((a.y >> 12) + ry, (uint32_t)v.gx) + (a.y & 0xFFFu) + rx;
    }
    const uint64_t hm = __ballot(hit);
    if (hit) {
      const uint32_t pos = obase + run + __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
      if (pos < cap) {
        if (v.pack_tile_shift > 0) {   // compact keys: the whole record in the key, no value array
          keys[pos] = ((uint64_t)tile << v.pack_tile_shift) | ((uint64_t)g << v.pack_g_shift) | pos;
        } else {
          keys[pos] = ((uint64_t)tile << 32) | pos;
          vals[pos] = g;
        }
      }
    }
    run += (uint32_t)__builtin_popcountll(hm);
  });
}

__global__ void __launch_bounds__(256)
cull_emit_kernel(DgsView v, const DgsRow* __restrict__ rows, const uint32_t* __restrict__ order,
                 const uint32_t* __restrict__ tt_tight, const uint4* __restrict__ recs,
                 const uint32_t* __restrict__ offs_tight, const uint32_t* __restrict__ status,
                 uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t cap) {
  if (status[3] != 0u) return;   // the surviving total overflowed 32 bits: the offsets are meaningless (the host raises)
  __shared__ unsigned long long s_bits[4][CULL_CH];
  __shared__ uint4 s_a[4][64];    // first slot, minx | miny << 12, width | flags << 16 (1 always, 2 never, 4 big), 1 / width
  __shared__ uint4 s_b[4][64];    // k * T, Gaussian, hit bits of the first 64 slots
  __shared__ float4 s_q[4][64];   // big rectangles only: x, y, a, b
  __shared__ float4 s_r[4][64];   //                      c, 1/a, 1/c, r2
  const int lane = dgs_lane(), w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t n = (uint64_t)v.K * v.P;
  const bool in = j < n;
  const bool live = in && tt_tight[j] != 0u;
  const uint64_t lm = __ballot(live);
  if (lm == 0ull) return;   // wave-uniform
  const uint32_t i = in ? order[j] : 0u;
  const uint4 rec = live ? recs[i] : make_uint4(0u, 0u, 0u, 0u);
  const uint32_t otight = in ? offs_tight[j] : 0u;
  const bool big = live && (rec.x & CULL_BIG) != 0u;
  const bool any_big = __ballot(big) != 0ull;
  const uint32_t c = __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u));
  uint32_t work = 0, rect = 0, wide = 1, fl = 0, tb = 0, g = 0;
  float4 q = make_float4(0, 0, 0, 0), r = q;
  if (live) {
    const uint32_t k = i / (uint32_t)v.P;
    g = i - k * (uint32_t)v.P;
    tb = k * (uint32_t)v.T;
    if (!big) {   // the stored bits are the test: slots up to the last hit
      rect = rec.x & 0xFFFFFFu;
      wide = ((rec.x >> 24) & 63u) + 1u;
      work = 64u - (uint32_t)__builtin_clzll(((uint64_t)rec.w << 32) | rec.z);
    } else {      // more than 64 slots: the whole rectangle again, from the row
      const DgsRow* row = rows + i;
      const float4 A = reinterpret_cast<const float4*>(row)[0];
      const float4 B = reinterpret_cast<const float4*>(row)[1];
      int minx, miny, maxx, maxy;
      dgs_get_rect(A.x, A.y, row->radius, v.gx, v.gy, minx, miny, maxx, maxy);
      wide = (uint32_t)(maxx - minx);
      rect = (uint32_t)minx | ((uint32_t)miny << 12);
      work = wide * (uint32_t)(maxy - miny);
      const DgsCull cg = dgs_make_cull(A.z, A.w, B.x, B.y);
      fl = (cg.always ? 1u : 0u) | (cg.never ? 2u : 0u) | 4u;
      q = make_float4(A.x, A.y, cg.a, cg.b);
      r = make_float4(cg.c, cg.inv_a, cg.inv_c, cg.r2);
    }
  }
  const uint32_t incl = cull_wave_incl_scan(work, lane);
  const uint32_t excl = incl - work;
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  if (live) {
    s_a[w][c] = make_uint4(excl, rect, wide | (fl << 16), __float_as_uint(__builtin_amdgcn_rcpf((float)wide)));
    s_b[w][c] = make_uint4(tb, g, rec.z, rec.w);
    if (big) {
      s_q[w][c] = q;
      s_r[w][c] = r;
    }
  }
  // the wave's duplicates are contiguous from its first pair's offset (offs_tight is the exclusive scan in this order)
  const uint32_t obase = (uint32_t)__builtin_amdgcn_readfirstlane((int)otight);
  if (any_big)
    cull_emit_rounds<true>(v, s_bits[w], s_a[w], s_b[w], s_q[w], s_r[w], lane, live, excl, total, obase, cap, keys, vals);
  else
    cull_emit_rounds<false>(v, s_bits[w], s_a[w], s_b[w], s_q[w], s_r[w], lane, live, excl, total, obase, cap, keys, vals);
}

// Capacity mode (dgs_forward): the duplicate arrays were sized by the caller before the count was known.  Words of
// c.num_rendered: [0] rectangle total, [1] its high half (u32 overflow), [2], [3] the same for the surviving total
// (tile_cull; [0], [1] stay 0 then), [4] the count
// the sort / ranges kernels use = min(count, capacity), [5] overflow flag (the lists are truncated: every consumer that
// indexes by duplicate offset returns early, the caller re-runs with a larger capacity).
__global__ void finalize_count_kernel(uint32_t* __restrict__ nr, int cull, uint32_t cap, uint32_t* __restrict__ drops,
                                      uint32_t* __restrict__ status, uint32_t* __restrict__ host_words,
                                      const unsigned long long* __restrict__ host_indirect) {
  const uint32_t n = cull ? nr[2] : nr[0];
  const uint32_t hi = cull ? nr[3] : nr[1];
  const bool bad = (hi != 0u) || (n > cap);
  nr[4] = bad ? 0u : n;
  nr[5] = bad ? 1u : 0u;
  uint32_t dropped = 0u;
  if (drops != nullptr) {   // the caller's running count of overflowed forwards (graph replays)
    dropped = drops[0] + (bad ? 1u : 0u);
    if (bad) drops[0] = dropped;
  }
  if (status != nullptr) {                       // DgsForwardOut.status_dev: the words of THIS forward, outside the blobs
    status[0] = n;
    status[1] = hi;
    status[2] = bad ? 1u : 0u;
    status[3] = bad ? 0u : n;
  }
  // The same words straight into PINNED HOST memory (device-accessible), instead of four 4-byte copy nodes behind this
  // kernel: a copy node in a replayed graph costs a hand-over between the compute queue and the copy engine -- tens of
  // microseconds each on a step that takes a few hundred (tools/step_gaps.py).  host_words = num_rendered_host itself;
  // host_indirect = a device word holding the address of the pinned block this REPLAY should write (the caller changes it
  // from step to step through device memory: DgsForwardOut.status_host_indirect).
  uint32_t* dst[2] = {host_words, host_indirect != nullptr ? reinterpret_cast<uint32_t*>(host_indirect[0]) : nullptr};
  for (int i = 0; i < 2; i++) {
    uint32_t* h = dst[i];
    if (h == nullptr) continue;
    __hip_atomic_store(&h[0], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&h[1], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&h[2], bad ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&h[3], bad ? 0u : n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (drops != nullptr) __hip_atomic_store(&h[4], dropped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
}

// a few words cleared / copied by a kernel instead of a memset / copy node (see finalize_count_kernel)
__global__ void clear_words_kernel(uint32_t* __restrict__ p, int n) {
  for (int i = threadIdx.x; i < n; i += 64) p[i] = 0u;
}
// (dst may be pinned host memory: system-scope stores and a system fence, like finalize_count_kernel's)
__global__ void copy_words_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, int n) {
  for (int i = threadIdx.x; i < n; i += 256) __hip_atomic_store(&dst[i], src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
}

// ---------------------------------------------------------------------------------------------- ranges
__global__ void __launch_bounds__(256)
ranges_search_kernel(uint32_t L, const uint32_t* __restrict__ n_dev, const uint64_t* __restrict__ keys,
                     uint2* __restrict__ ranges, int tile_shift, uint32_t ntiles) {
  __shared__ uint32_t s_b[257];
  if (n_dev != nullptr) L = min(L, n_dev[0]);
  const uint32_t t0 = blockIdx.x * 256;
  auto bound = [&](uint32_t tile) {   // first idx in [0, L] with tile(keys[idx]) >= tile
    uint32_t lo = 0, hi = L;
    while (lo < hi) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      if ((uint32_t)(keys[mid] >> tile_shift) < tile) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  s_b[threadIdx.x] = bound(t0 + threadIdx.x);
  if (threadIdx.x == 0) s_b[256] = bound(t0 + 256);
  __syncthreads();
  const uint32_t t = t0 + threadIdx.x;
  if (t < ntiles) {
    const uint32_t b0 = s_b[threadIdx.x], b1 = s_b[threadIdx.x + 1];
    ranges[t] = (b1 > b0) ? make_uint2(b0, b1) : make_uint2(0u, 0u);
  }
}

// ------------------------------------------------------------------------------------------ radix sort
// Stable LSD radix sort, RB-bit digits (RB <= 9; 10-bit digits were measured slower per key: 5 x 1.11 ms vs
// 6 x 0.81 ms at the metric config).  Per pass: (1) per-block digit histogram written
// digit-major [digit][block]; (2) exclusive scan of that table = global scatter bases; (3) scatter with
// stable in-block ranks.  A block owns SORT_TILE consecutive pairs; wave w owns a contiguous share of them,
// read in rounds of 64, so the stable order inside a block is (wave, round, lane).
// 512-thread blocks of 8192 pairs (round 4): the digit runs a block writes are twice as long (128 bytes on average with
// 512 digits) and, with 16-bit wave counts, two blocks = 16 waves fit a CU (the 256-thread block was held to 3 waves per SIMD
// by its 130 VGPRs): scatter + histogram of a pass 0.312 -> 0.275 ms at the metric config (-DDGS_SORT_THREADS=256: the old shape)
#ifndef DGS_SORT_THREADS
#define DGS_SORT_THREADS 512
#endif
constexpr int SORT_THREADS = DGS_SORT_THREADS;
constexpr int SORT_WAVES = SORT_THREADS / 64;
// (32 items = 8192-pair tiles would double the length of the digit runs the scatter writes, but need 256 VGPRs and 78 KB
// of LDS per block: scatter 265 vs 208 us per pass, measured with -DDGS_SORT_ITEMS=32)
#ifndef DGS_SORT_ITEMS
#define DGS_SORT_ITEMS 16
#endif
constexpr int SORT_ITEMS = DGS_SORT_ITEMS;
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;  // 8192 pairs per block
constexpr int SORT_MAX_RB = 9;
constexpr int SORT_MAX_BINS = 1 << SORT_MAX_RB;

// ----------------------------------------------------------------------- tile-major histogram table + column scan
// The classic table is digit-major ([digit][tile]) so that one flat scan yields scatter bases, but that makes
// every tile read and write 512 words at a 61 KB stride (one 64-byte line per 4 useful bytes: ~1 GB of hidden
// traffic per pass at the metric config).  Here a tile owns one contiguous 2 KB row; the bases come from a
// column-wise scan done in two small kernels whose loads are all full rows.
constexpr int CS_CHUNK = 64;  // tiles per column-scan chunk

__global__ void __launch_bounds__(SORT_THREADS)
sort_hist_rows_kernel(const uint64_t* __restrict__ keys, uint64_t n, const uint32_t* __restrict__ n_dev, int shift,
                      int rb, uint32_t* __restrict__ table) {
  // n_dev (optional): the pair count lives in device memory and n is only the capacity the grid was sized for
  if (n_dev != nullptr) n = min(n, (uint64_t)n_dev[0]);
  if ((uint64_t)blockIdx.x * SORT_TILE >= n) return;
  __shared__ uint32_t h[SORT_MAX_BINS];
  for (int i = threadIdx.x; i < SORT_MAX_BINS; i += SORT_THREADS) h[i] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * SORT_TILE;
  const uint32_t mask = (1u << rb) - 1;
#pragma unroll 4
  for (int r = 0; r < SORT_ITEMS; r++) {
    const uint64_t i = base + (uint64_t)r * SORT_THREADS + threadIdx.x;
    if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & mask], 1u);
  }
  __syncthreads();
  uint32_t* row = table + (size_t)blockIdx.x * SORT_MAX_BINS;
  for (int i = threadIdx.x; i < SORT_MAX_BINS; i += SORT_THREADS) row[i] = h[i];
}

// chunk c: per digit, exclusive running count over the chunk's tiles (in place) and the chunk total
__global__ void __launch_bounds__(256)
colscan_chunk_kernel(uint32_t* __restrict__ table, uint32_t nblocks, const uint32_t* __restrict__ n_dev,
                     uint32_t* __restrict__ ctot) {
  if (n_dev != nullptr) nblocks = min(nblocks, (uint32_t)(((uint64_t)n_dev[0] + SORT_TILE - 1) / SORT_TILE));
  const uint32_t t0 = blockIdx.x * CS_CHUNK;
  if (t0 >= nblocks) return;
  const uint32_t t1 = min(t0 + CS_CHUNK, nblocks);
  uint32_t run0 = 0, run1 = 0;
  // the rows are fetched eight at a time before the serial prefix is applied (the loads do not depend on the running
  // sums; one dependent row per iteration left the 142 blocks of the metric config latency-bound: 30 us per pass)
  constexpr int U = 8;
  for (uint32_t t = t0; t < t1; t += U) {
    uint32_t a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const bool ok = t + u < t1;
      const uint32_t* row = table + (size_t)(ok ? t + u : t) * SORT_MAX_BINS;
      a[u] = ok ? row[threadIdx.x] : 0u;
      b[u] = ok ? row[threadIdx.x + 256] : 0u;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (t + u < t1) {
        uint32_t* row = table + (size_t)(t + u) * SORT_MAX_BINS;
        row[threadIdx.x] = run0;
        row[threadIdx.x + 256] = run1;
      }
      run0 += a[u];
      run1 += b[u];
    }
  }
  ctot[(size_t)blockIdx.x * SORT_MAX_BINS + threadIdx.x] = run0;
  ctot[(size_t)blockIdx.x * SORT_MAX_BINS + threadIdx.x + 256] = run1;
}

// one block: chunk totals -> exclusive chunk bases per digit (in place), plus the exclusive base of each digit (dbase).
// The chunk loop is latency-bound (one block, dependent only through the running sums), so rows are fetched
// 32 at a time before the serial prefix is applied.
__global__ void __launch_bounds__(256)
colscan_top_kernel(uint32_t* __restrict__ ctot, uint32_t nchunks, const uint32_t* __restrict__ n_dev,
                   uint32_t* __restrict__ dbase) {
  if (n_dev != nullptr) {
    const uint32_t nb = (uint32_t)(((uint64_t)n_dev[0] + SORT_TILE - 1) / SORT_TILE);
    nchunks = min(nchunks, (nb + CS_CHUNK - 1) / CS_CHUNK);
  }
  __shared__ uint32_t lds[8];
  constexpr int U = 32;
  uint32_t run0 = 0, run1 = 0;
  for (uint32_t c0 = 0; c0 < nchunks; c0 += U) {
    uint32_t a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const bool ok = c0 + u < nchunks;
      const uint32_t* row = ctot + (size_t)(ok ? c0 + u : c0) * SORT_MAX_BINS;
      a[u] = ok ? row[threadIdx.x] : 0u;
      b[u] = ok ? row[threadIdx.x + 256] : 0u;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (c0 + u < nchunks) {
        uint32_t* row = ctot + (size_t)(c0 + u) * SORT_MAX_BINS;
        row[threadIdx.x] = run0;
        row[threadIdx.x + 256] = run1;
      }
      run0 += a[u];
      run1 += b[u];
    }
  }
  // digit order is 0..255 (run0 of thread t) then 256..511 (run1 of thread t).  The digit bases go to a row of their own
  // (dbase) that the scatter adds itself: a second sweep over the chunk rows to fold them in doubled this kernel's time.
  uint32_t tot0, tot1;
  const uint32_t pre0 = block_excl_scan(run0, &tot0, lds);
  const uint32_t pre1 = block_excl_scan(run1, &tot1, lds) + tot0;
  dbase[threadIdx.x] = pre0;
  dbase[threadIdx.x + 256] = pre1;
}

// ---------------------------------------------------------------------------------------- scatter pass
// Ranks its SORT_TILE pairs (wave-ballot match per digit bit, per-wave running counts in LDS), takes the tile's global digit
// bases from the tile-major histogram table (gbase: in-chunk exclusive counts) and the column-scanned chunk bases
// (ctot, followed by the row of digit bases), and re-orders the pairs through LDS so that the global writes are
// contiguous runs per digit instead of 64 scattered 8-byte stores per wave instruction.  (A decoupled-look-back
// "onesweep" version of this kernel -- one read of the keys for the histograms of all passes, ticketed tiles, self-tagged
// status words -- measured slower here: variants/NOTES.md.)
__global__ void __launch_bounds__(SORT_THREADS)
sort_scatter_kernel(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                    uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint64_t n, int shift, int rb,
                    const uint32_t* __restrict__ gbase, const uint32_t* __restrict__ ctot, uint32_t nblocks,
                    const uint32_t* __restrict__ n_dev) {
  if (n_dev != nullptr) {
    n = min(n, (uint64_t)n_dev[0]);
    if ((uint64_t)blockIdx.x * SORT_TILE >= n) return;  // block-uniform, before any barrier
  }
  __shared__ uint64_t lds_k[SORT_TILE];  // 8 bytes per pair; re-used for the values
  __shared__ uint16_t whist[SORT_WAVES][SORT_MAX_BINS];   // (a tile holds fewer than 65536 pairs)
  __shared__ uint32_t dstart[SORT_MAX_BINS];
  __shared__ uint32_t gb[SORT_MAX_BINS];
  __shared__ uint32_t s_scan[SORT_WAVES];
  static_assert(SORT_TILE < 65536 && SORT_MAX_BINS % SORT_THREADS == 0, "16-bit wave counts; whole digits per thread");
  const uint32_t mask = (1u << rb) - 1;
  const int lane = dgs_lane(), w = threadIdx.x >> 6;
  for (int i = lane; i < SORT_MAX_BINS; i += 64) whist[w][i] = 0;
  __syncthreads();
  const uint32_t tile = blockIdx.x;

  const uint64_t tbase = (uint64_t)tile * SORT_TILE;
  const uint64_t wbase = tbase + (uint64_t)w * (64 * SORT_ITEMS);
  uint64_t key[SORT_ITEMS];
  uint32_t val[SORT_ITEMS];
  uint32_t rank[SORT_ITEMS];
  volatile uint16_t* wh = whist[w];
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int r = 0; r < SORT_ITEMS; r++) {
    const uint64_t i = wbase + (uint64_t)r * 64 + lane;
    const bool valid = i < n;
    key[r] = valid ? keys_in[i] : ~0ull;
    val[r] = (valid && vals_in != nullptr) ? vals_in[i] : 0u;   // vals_in == nullptr: keys only (compact keys)
  }
#pragma unroll
  for (int r = 0; r < SORT_ITEMS; r++) {
    const bool valid = (wbase + (uint64_t)r * 64 + lane) < n;
    const uint32_t d = (uint32_t)(key[r] >> shift) & mask;
    uint64_t peers = __ballot(valid);
    for (int b = 0; b < rb; b++) {
      const uint64_t m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const uint32_t below = (uint32_t)__popcll(peers & lt_mask);
    uint32_t pre = 0;
    if (valid) pre = wh[d];
    __builtin_amdgcn_wave_barrier();
    if (valid && below == 0) wh[d] = (uint16_t)(pre + (uint32_t)__popcll(peers));
    __builtin_amdgcn_wave_barrier();
    rank[r] = pre + below;
  }
  __syncthreads();

  // thread t owns digits DPT t .. DPT t + DPT - 1: block totals, per-wave exclusive offsets, start of each digit inside the tile
  constexpr int DPT = SORT_MAX_BINS / SORT_THREADS;
  uint32_t cnt[DPT], sum = 0;
#pragma unroll
  for (int e = 0; e < DPT; e++) {
    const int d = DPT * threadIdx.x + e;
    uint32_t run = 0;
#pragma unroll
    for (int ww = 0; ww < SORT_WAVES; ww++) {
      const uint32_t c = whist[ww][d];
      whist[ww][d] = (uint16_t)run;
      run += c;
    }
    cnt[e] = run;
    sum += run;
  }
  uint32_t tot;
  uint32_t pre2 = block_excl_scan_n<SORT_THREADS>(sum, &tot, s_scan);
  // global position of tile-local slot i holding digit d:  gb[d] + i
  const uint32_t cap_chunks = (nblocks + CS_CHUNK - 1) / CS_CHUNK;
#pragma unroll
  for (int e = 0; e < DPT; e++) {
    const int d = DPT * threadIdx.x + e;
    dstart[d] = pre2;
    gb[d] = gbase[(size_t)tile * SORT_MAX_BINS + d] + ctot[(size_t)(tile / CS_CHUNK) * SORT_MAX_BINS + d] +
            ctot[(size_t)cap_chunks * SORT_MAX_BINS + d] - pre2;
    pre2 += cnt[e];
  }
  __syncthreads();

  // stage the keys in tile-sorted order
#pragma unroll
  for (int r = 0; r < SORT_ITEMS; r++) {
    if ((wbase + (uint64_t)r * 64 + lane) < n) {
      const uint32_t d = (uint32_t)(key[r] >> shift) & mask;
      rank[r] = dstart[d] + whist[w][d] + rank[r];  // tile-local slot
      lds_k[rank[r]] = key[r];
    }
  }
  __syncthreads();
  const uint32_t nvalid = (uint32_t)((n - tbase) < (uint64_t)SORT_TILE ? (n - tbase) : (uint64_t)SORT_TILE);
  uint32_t gpos[SORT_ITEMS];
#pragma unroll
  for (int r = 0; r < SORT_ITEMS; r++) {
    const uint32_t i = (uint32_t)r * SORT_THREADS + threadIdx.x;
    if (i < nvalid) {
      const uint64_t k = lds_k[i];
      gpos[r] = gb[(uint32_t)(k >> shift) & mask] + i;
      keys_out[gpos[r]] = k;
    }
  }
  if (vals_in == nullptr) return;   // block-uniform
  __syncthreads();
  uint32_t* lds_v = reinterpret_cast<uint32_t*>(lds_k);
#pragma unroll
  for (int r = 0; r < SORT_ITEMS; r++)
    if ((wbase + (uint64_t)r * 64 + lane) < n) lds_v[rank[r]] = val[r];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < SORT_ITEMS; r++) {
    const uint32_t i = (uint32_t)r * SORT_THREADS + threadIdx.x;
    if (i < nvalid) vals_out[gpos[r]] = lds_v[i];
  }
}

// ---------------------------------------------------------------------- depth order: segmented 27(+5)-bit sort
// The (k, depth, index) ordering of the K*P (subframe, Gaussian) pairs is K independent sorts of P 32-bit depth keys: a
// segmented stable LSD radix sort, one launch chain for all K segments.  Against sorting (k << 32 | depth, index) pairs
// with the generic 64-bit sort this moves 20 bytes per pair and pass instead of 32 (4-byte keys, the first pass
// generates the indices instead of reading them, the last pass does not write keys).
//
// THREE 9-bit passes instead of four 8-bit ones (round 4): a visible pair has view depth > 0.2 (in_frustum,
// auxiliary.h:159), and positive floats order like their bit patterns, so preprocess stores key = bits(depth) -
// bits(0.2f) -- order-preserving -- and every depth below 13107 (= the float whose bits are bits(0.2f) + 2^27) gives a
// key below 2^27: three 9-bit digits.  Invisible pairs carry 0xFFFFFFFF; their three low digits are all ones, so they
// sort behind every visible key < 2^27 - 1.  A visible key >= 2^27 - 1 (a Gaussian more than 13 km... units deep) sets
// a device flag (first histogram); the fourth pass on bits [27, 32) and a copy that puts the result where the three-pass
// result lands are always launched and return at once unless the flag is set: exact for every input, ~4 empty launches
// in the common case.  A block owns DS_TILE consecutive pairs of ONE segment; the stable order inside a block is
// (wave, round, lane) as in the generic sort.
#ifndef DGS_DS_THREADS
#define DGS_DS_THREADS 512
#endif
constexpr int DS_THREADS = DGS_DS_THREADS;
constexpr int DS_WAVES = DS_THREADS / 64;
constexpr int DS_ITEMS = 16;
constexpr int DS_TILE = DS_THREADS * DS_ITEMS;  // pairs per block
constexpr int DS_RB = 9;
constexpr int DS_BINS = 1 << DS_RB;
constexpr int DS_BPT = DS_BINS / DS_THREADS > 0 ? DS_BINS / DS_THREADS : 1;   // bins per thread where a block owns all bins
constexpr int DS_CHUNK = 32;  // blocks per column-scan chunk
constexpr uint32_t DS_INVISIBLE = 0xFFFFFFFFu;
constexpr uint32_t DS_NARROW_MAX = (1u << (3 * DS_RB)) - 1u;   // a visible key at or above this needs the fourth pass
static_assert(DS_BINS % DS_THREADS == 0 || DS_THREADS % DS_BINS == 0, "bins and threads must divide each other");
static_assert(DS_TILE <= 65535, "per-wave digit counts are kept in 16 bits");

// pass: 0..2 the 9-bit digits, 3 the optional pass on bits [27, 32) (returns unless *wide_flag)
__global__ void __launch_bounds__(DS_THREADS)
dsort_hist_kernel(const uint32_t* __restrict__ keys, uint32_t P, uint32_t nb, int pass, uint32_t* __restrict__ table,
                  uint32_t* __restrict__ wide_flag, const uint32_t* __restrict__ segcnt) {
  // segcnt (tile_cull; round 5): the first pass DROPS the invisible pairs (22 % at the metric configuration: they only
  // ever sorted to the end of their segment) -- it counts and scatters the visible ones alone and leaves their number per
  // segment in segcnt[k]; the later passes work on the first segcnt[k] pairs of every segment
  if (pass == 3 && *wide_flag == 0u) return;
  __shared__ uint32_t h[DS_BINS];
  for (int i = threadIdx.x; i < DS_BINS; i += DS_THREADS) h[i] = 0;
  __syncthreads();
  const int shift = DS_RB * pass;
  const uint32_t k = blockIdx.x / nb, b = blockIdx.x - k * nb;
  const uint32_t* seg = keys + (size_t)k * P;
  const uint32_t Pk = (segcnt != nullptr && pass > 0) ? segcnt[k] : P;
  const int lane = dgs_lane();
  bool wide = false;
#pragma unroll 4
  for (int r = 0; r < DS_ITEMS; r++) {
    const uint32_t i = b * DS_TILE + (uint32_t)r * DS_THREADS + threadIdx.x;
    bool valid = i < Pk;
    const uint32_t key = valid ? seg[i] : 0u;
    if (segcnt != nullptr && pass == 0) valid = valid && key != DS_INVISIBLE;
    const uint32_t d = (key >> shift) & (uint32_t)(DS_BINS - 1);
    if (pass == 0) wide = wide || (valid && key != DS_INVISIBLE && key >= DS_NARROW_MAX);
    // the upper digits of depth keys take few values, and after the earlier passes a wave's 64 keys often share theirs:
    // one add of the wave's count instead of 64 same-address LDS atomics
    const uint64_t vm = __ballot(valid);
    if (vm == 0ull) continue;
    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, __builtin_ctzll(vm));
    if (__ballot(valid && d != d0) == 0ull) {
      if (lane == __builtin_ctzll(vm)) atomicAdd(&h[d0], (uint32_t)__builtin_popcountll(vm));
    } else if (valid) {
      atomicAdd(&h[d], 1u);
    }
  }
  if (pass == 0 && __ballot(wide) != 0ull && lane == 0) atomicOr(wide_flag, 1u);   // (never, for sane depth ranges)
  __syncthreads();
  for (int i = threadIdx.x; i < DS_BINS; i += DS_THREADS) table[(size_t)blockIdx.x * DS_BINS + i] = h[i];
}

// (segment, chunk): per digit, exclusive running count over the chunk's blocks (in place) and the chunk total
// fuse_top (nch == 1, i.e. segments of up to DS_CHUNK * DS_TILE pairs): the chunk total IS the digit's count in
// the segment, so the digit bases are formed right here and the top kernel is not launched
__global__ void __launch_bounds__(DS_BINS)
dsort_colscan_chunk_kernel(uint32_t* __restrict__ table, uint32_t nb, uint32_t nch, uint32_t* __restrict__ ctot,
                           int fuse_top, uint32_t P, int pass, const uint32_t* __restrict__ wide_flag,
                           uint32_t* __restrict__ segcnt) {
  if (pass == 3 && *wide_flag == 0u) return;
  __shared__ uint32_t lds[DS_BINS / 64];
  const uint32_t k = blockIdx.x / nch, c = blockIdx.x - k * nch;
  const uint32_t t0 = c * DS_CHUNK, t1 = min(t0 + (uint32_t)DS_CHUNK, nb);
  uint32_t* base = table + ((size_t)k * nb) * DS_BINS + threadIdx.x;
  uint32_t run = 0;
  constexpr int U = 8;
  for (uint32_t t = t0; t < t1; t += U) {
    uint32_t v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = (t + u < t1) ? base[(size_t)(t + u) * DS_BINS] : 0u;
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (t + u < t1) base[(size_t)(t + u) * DS_BINS] = run;
      run += v[u];
    }
  }
  if (fuse_top) {
    uint32_t tot;
    run = block_excl_scan_n<DS_BINS>(run, &tot, lds) + k * P;   // (what dsort_colscan_top_kernel leaves for the only chunk)
    if (segcnt != nullptr && pass == 0 && threadIdx.x == 0) segcnt[k] = tot;   // the segment's visible pairs
  }
  ctot[(size_t)blockIdx.x * DS_BINS + threadIdx.x] = run;
}

// segment k: chunk totals -> exclusive chunk bases per digit, plus the digit's base inside the segment and k * P
__global__ void __launch_bounds__(DS_BINS)
dsort_colscan_top_kernel(uint32_t* __restrict__ ctot, uint32_t nch, uint32_t P, int pass,
                         const uint32_t* __restrict__ wide_flag, uint32_t* __restrict__ segcnt) {
  if (pass == 3 && *wide_flag == 0u) return;
  __shared__ uint32_t lds[DS_BINS / 64];
  uint32_t* base = ctot + ((size_t)blockIdx.x * nch) * DS_BINS + threadIdx.x;
  uint32_t run = 0;
  for (uint32_t c = 0; c < nch; c++) {
    const uint32_t v = base[(size_t)c * DS_BINS];
    base[(size_t)c * DS_BINS] = run;
    run += v;
  }
  uint32_t tot;
  const uint32_t pre = block_excl_scan_n<DS_BINS>(run, &tot, lds) + blockIdx.x * P;
  if (segcnt != nullptr && pass == 0 && threadIdx.x == 0) segcnt[blockIdx.x] = tot;   // the segment's visible pairs
  for (uint32_t c = 0; c < nch; c++) base[(size_t)c * DS_BINS] += pre;
}

// PASS 0: the values are generated (flat index k * P + i) instead of read.  PASS 2: the last pass unless the wide flag is
// set -- keys are then not written, and the visibility flags are.  PASS 3: only if the flag is set.
template <int PASS>
__global__ void __launch_bounds__(DS_THREADS)
dsort_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                     uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t P, uint32_t nb,
                     uint32_t nch, const uint32_t* __restrict__ table, const uint32_t* __restrict__ ctot,
                     uint32_t* __restrict__ vis_dst, const uint32_t* __restrict__ wide_flag,
                     const uint32_t* __restrict__ cnt_src, uint32_t* __restrict__ cnt_dst,
                     const uint32_t* __restrict__ segcnt) {
  // cnt_src / cnt_dst (tile_cull, K * P <= 2^24; round 5): a per-pair payload -- the surviving-tile count cull_count_kernel
  // left in NATURAL order -- rides in the top byte of the 32-bit value through all passes (the flat index needs 24 bits;
  // 255 = "look the count up", for the few pairs with more tiles) and is laid out in the final order by the last pass,
  // which also strips it from the indices.  Replaces gather_cnt_kernel: K * P random 4-byte reads, 0.12 ms at the metric
  // configuration, for one coalesced read in the first pass and one more scattered 4-byte write in the last.
  constexpr uint32_t IDX_MASK = 0x00FFFFFFu;
  const bool wide = (PASS >= 2) ? (*wide_flag != 0u) : false;
  if (PASS == 3 && !wide) return;
  constexpr int shift = DS_RB * PASS;
  constexpr bool FIRST = PASS == 0;
  const bool last = (PASS == 3) || (PASS == 2 && !wide);
  __shared__ uint32_t lds_k[DS_TILE];
  __shared__ uint32_t lds_v[DS_TILE];
  __shared__ uint16_t whist[DS_WAVES][DS_BINS];   // per-wave digit counts, then the wave's first slot of the digit in the block
  __shared__ uint32_t gb[DS_BINS];
  __shared__ uint32_t s_scan[DS_THREADS / 64];
  const int lane = dgs_lane(), w = threadIdx.x >> 6;
  const uint32_t k = blockIdx.x / nb, b = blockIdx.x - k * nb;
  // segcnt: the invisible pairs were dropped by the first pass (see dsort_hist_kernel): this pass sees the segment's first
  // Pk pairs; the last pass clears the flags and counts of the positions behind them (their indices stay undefined:
  // no consumer reads the index of a pair whose flag is 0)
  const bool drop = segcnt != nullptr;
  const uint32_t Pk = (drop && !FIRST) ? segcnt[k] : P;
  if (drop && !FIRST) {
    const uint32_t t0 = b * DS_TILE, t1 = min(t0 + (uint32_t)DS_TILE, P);
    if (last && t1 > Pk) {   // (block-uniform)
      for (uint32_t i = max(t0, Pk) + threadIdx.x; i < t1; i += DS_THREADS) {
        if (vis_dst != nullptr) vis_dst[(size_t)k * P + i] = 0u;
        if (cnt_dst != nullptr) cnt_dst[(size_t)k * P + i] = 0u;
      }
    }
    if (t0 >= Pk) return;   // nothing of this block is left (block-uniform, before any barrier)
  }
#pragma unroll
  for (int i = 0; i < DS_BINS / 64; i++) whist[w][lane + 64 * i] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const uint32_t wbase = b * DS_TILE + (uint32_t)w * (64 * DS_ITEMS);  // index inside the segment
  const size_t sbase = (size_t)k * P;
  uint32_t key[DS_ITEMS], val[DS_ITEMS];
  uint16_t rank[DS_ITEMS];
  volatile uint16_t* wh = whist[w];
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int r = 0; r < DS_ITEMS; r++) {
    const uint32_t i = wbase + (uint32_t)r * 64 + lane;
    const bool valid = i < Pk;
    key[r] = valid ? keys_in[sbase + i] : DS_INVISIBLE;
    val[r] = FIRST ? (uint32_t)(sbase + i) : (valid ? vals_in[sbase + i] : 0u);
  }
  if (FIRST && cnt_src != nullptr) {
    // the payload: loaded for every pair, whatever its key says (an invisible pair's count is 0 by construction:
    // cull_count_kernel) -- a load that waits for the key to decide whether it is needed turns the sixteen rounds into
    // sixteen dependent round trips (134 instead of 75 us for this pass)
    uint32_t cn[DS_ITEMS];
#pragma unroll
    for (int r = 0; r < DS_ITEMS; r++) {
      const uint32_t i = wbase + (uint32_t)r * 64 + lane;
      cn[r] = (i < Pk) ? cnt_src[sbase + i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < DS_ITEMS; r++) val[r] |= (cn[r] < 255u ? cn[r] : 255u) << 24;
  }
#pragma unroll
  for (int r = 0; r < DS_ITEMS; r++) {
    // (first pass with `drop`: an invisible pair takes no part -- it gets no rank and no slot)
    const bool valid = (wbase + (uint32_t)r * 64 + lane) < Pk && !(drop && FIRST && key[r] == DS_INVISIBLE);
    const uint32_t d = (key[r] >> shift) & (uint32_t)(DS_BINS - 1);
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < DS_RB; bit++) {
      const uint64_t m = __ballot((d >> bit) & 1u);
      peers &= ((d >> bit) & 1u) ? m : ~m;
    }
    const uint32_t below = (uint32_t)__popcll(peers & lt_mask);
    uint32_t pre = 0;
    if (valid) pre = wh[d];
    __builtin_amdgcn_wave_barrier();
    if (valid && below == 0) wh[d] = (uint16_t)(pre + (uint32_t)__popcll(peers));
    __builtin_amdgcn_wave_barrier();
    rank[r] = (uint16_t)(pre + below);
  }
  __syncthreads();
  // thread t owns digits t, t + DS_THREADS, ...: block total, per-wave exclusive offsets, start of the digit inside the block
  uint32_t cnt[DS_BPT], cnt_sum = 0;
#pragma unroll
  for (int i = 0; i < DS_BPT; i++) {
    const int d = threadIdx.x * DS_BPT + i;       // (consecutive digits per thread: the block scan below is then in digit order)
    uint32_t c = 0;
    if (d < DS_BINS) {
#pragma unroll
      for (int ww = 0; ww < DS_WAVES; ww++) c += whist[ww][d];
    }
    cnt[i] = c;
    cnt_sum += c;
  }
  uint32_t tot;
  uint32_t dstart = block_excl_scan_n<DS_THREADS>(cnt_sum, &tot, s_scan);
#pragma unroll
  for (int i = 0; i < DS_BPT; i++) {
    const int d = threadIdx.x * DS_BPT + i;
    if (d < DS_BINS) {
      // global position of block-local slot i holding digit d:  gb[d] + i
      gb[d] = table[(size_t)blockIdx.x * DS_BINS + d] + ctot[((size_t)k * nch + b / DS_CHUNK) * DS_BINS + d] - dstart;
      uint32_t run = dstart;
#pragma unroll
      for (int ww = 0; ww < DS_WAVES; ww++) {
        const uint32_t c = whist[ww][d];
        whist[ww][d] = (uint16_t)run;              // first block-local slot of (wave ww, digit d)
        run += c;
      }
      dstart += cnt[i];
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < DS_ITEMS; r++) {
    if ((wbase + (uint32_t)r * 64 + lane) < Pk && !(drop && FIRST && key[r] == DS_INVISIBLE)) {
      const uint32_t d = (key[r] >> shift) & (uint32_t)(DS_BINS - 1);
      const uint32_t slot = (uint32_t)whist[w][d] + rank[r];
      lds_k[slot] = key[r];
      lds_v[slot] = val[r];
    }
  }
  __syncthreads();
  // pairs of this block that take part: all of it, what the segment's visible count leaves of it, or (first pass with
  // `drop`) its visible pairs = the block total of the digit counts
  const uint32_t nvalid = (drop && FIRST) ? tot : min((uint32_t)DS_TILE, Pk - b * DS_TILE);
#pragma unroll
  for (int r = 0; r < DS_ITEMS; r++) {
    const uint32_t i = (uint32_t)r * DS_THREADS + threadIdx.x;
    if (i < nvalid) {
      const uint32_t kk = lds_k[i];
      const uint32_t pos = gb[(kk >> shift) & (uint32_t)(DS_BINS - 1)] + i;
      if (!last) keys_out[pos] = kk;
      uint32_t vv = lds_v[i];
      if (cnt_src != nullptr && last) {   // the payload leaves the index here
        uint32_t cn = vv >> 24;
        vv &= IDX_MASK;
        if (cn == 255u) cn = cnt_src[vv];
        cnt_dst[pos] = (kk != DS_INVISIBLE) ? cn : 0u;
      }
      vals_out[pos] = vv;
      // tile_cull: the last pass also writes the visibility flags in the final order (invisible pairs sort to the end of
      // their subframe); the 16-byte cull records follow in a gather kernel of their own -- done here, behind this
      // kernel's LDS exchange and at its occupancy, the dependent random reads cost 225 us instead of ~100
      if (last && vis_dst != nullptr) vis_dst[pos] = (kk != DS_INVISIBLE) ? 1u : 0u;
    }
  }
}

// only if the fourth pass ran: its result into the buffer the three-pass result lands in
__global__ void __launch_bounds__(256)
dsort_copy_if_wide_kernel(uint64_t n, const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                          const uint32_t* __restrict__ wide_flag) {
  if (*wide_flag == 0u) return;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) dst[i] = src[i];
}

// tile_cull: the surviving-tile counts (natural order) into (k, depth, index) order, the input of the scan.
// (3 M random reads cost ~100 us here whatever the block -> XCD map: an XCD-contiguous map measured 107 vs 97 us.  The
// same gather of the 16-byte records costs 226 us on its own, which is why cull_emit_kernel does that one itself, behind
// its own arithmetic.)
__global__ void __launch_bounds__(256)
gather_cnt_kernel(uint64_t n, const uint32_t* __restrict__ order, const uint32_t* __restrict__ vis,
                  const uint32_t* __restrict__ src, uint32_t* __restrict__ dst) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  dst[j] = (vis[j] != 0u) ? src[order[j]] : 0u;
}

struct PassPlan {
  int n;
  int shift[16];
  int rb[16];
};

PassPlan plan_passes(int begin_bit, int end_bit) {
  PassPlan p;
  if (begin_bit < 0) begin_bit = 0;
  if (end_bit > 64) end_bit = 64;
  if (end_bit < begin_bit + 1) end_bit = begin_bit + 1;
  const int nbits = end_bit - begin_bit;
  p.n = (nbits + SORT_MAX_RB - 1) / SORT_MAX_RB;
  int lo = nbits / p.n, extra = nbits % p.n, s = begin_bit;
  for (int i = 0; i < p.n; i++) {
    p.shift[i] = s;
    p.rb[i] = lo + (i < extra ? 1 : 0);
    s += p.rb[i];
  }
  return p;
}

}  // namespace

hipError_t dgs_launch_finalize_count(const DgsCarve& c, int cull, uint32_t cap, uint32_t* drops, uint32_t* status,
                                     uint32_t* host_words, const uint64_t* host_indirect, hipStream_t s) {
  hipLaunchKernelGGL(finalize_count_kernel, dim3(1), dim3(1), 0, s, c.num_rendered, cull, cap, drops, status, host_words,
                     reinterpret_cast<const unsigned long long*>(host_indirect));
  return hipGetLastError();
}

hipError_t dgs_launch_clear_words(uint32_t* p, int n, hipStream_t s) {
  hipLaunchKernelGGL(clear_words_kernel, dim3(1), dim3(64), 0, s, p, n);
  return hipGetLastError();
}

hipError_t dgs_launch_copy_words(uint32_t* dst, const uint32_t* src, int n, hipStream_t s) {
  hipLaunchKernelGGL(copy_words_kernel, dim3(1), dim3(256), 0, s, dst, src, n);
  return hipGetLastError();
}


size_t dgs_scan_tmp_words(uint64_t n) { return (size_t)((n + SCAN_TILE - 1) / SCAN_TILE) + 64; }

hipError_t dgs_launch_scan(const uint32_t* in, uint32_t* out, uint64_t n, uint32_t* tmp, uint32_t* total,
                           hipStream_t s) {
  if (n == 0) {
    if (total) return hipMemsetAsync(total, 0, 2 * sizeof(uint32_t), s);
    return hipSuccess;
  }
  const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL(scan_reduce_kernel, dim3((uint32_t)nb), dim3(SCAN_THREADS), 0, s, in, n, tmp);
  if (nb <= SCAN_SELF_MAX) {
    hipLaunchKernelGGL(scan_apply_self_kernel, dim3((uint32_t)nb), dim3(SCAN_THREADS), 0, s, in, out, n, tmp, (uint32_t)nb,
                       total);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, tmp, nb, total);
  hipLaunchKernelGGL(scan_apply_kernel, dim3((uint32_t)nb), dim3(SCAN_THREADS), 0, s, in, out, n, tmp);
  return hipGetLastError();
}

int dgs_sort_num_passes(int begin_bit, int end_bit) { return plan_passes(begin_bit, end_bit).n; }

size_t dgs_sort_tmp_words(uint64_t n) {
  const uint64_t nblocks = (n + SORT_TILE - 1) / SORT_TILE;
  const uint64_t chunks = (nblocks + CS_CHUNK - 1) / CS_CHUNK;
  return (size_t)(nblocks * SORT_MAX_BINS + (chunks + 1) * SORT_MAX_BINS + 256);   // histogram rows, chunk bases + digit bases
}

// n_dev (optional): device word holding the actual pair count (<= n); every launch is then sized by the capacity n and
// the kernels read the count themselves (no host read of num_rendered between duplication and sort)
hipError_t dgs_launch_sort(uint64_t* keys, uint32_t* vals, uint64_t* keys_alt, uint32_t* vals_alt, uint64_t n,
                           int begin_bit, int end_bit, uint32_t* tmp, int* result_in_alt, hipStream_t s,
                           const uint32_t* n_dev) {
  const PassPlan plan = plan_passes(begin_bit, end_bit);
  *result_in_alt = plan.n & 1;
  if (n == 0) return hipSuccess;
  const uint32_t nblocks = (uint32_t)((n + SORT_TILE - 1) / SORT_TILE);
  uint32_t* table = tmp;
  uint32_t* ctot = tmp + (uint64_t)nblocks * SORT_MAX_BINS;  // [nchunks + 1][512]
  const uint32_t nchunks = (nblocks + CS_CHUNK - 1) / CS_CHUNK;
  uint64_t* kin = keys;
  uint32_t* vin = vals;
  uint64_t* kout = keys_alt;
  uint32_t* vout = vals_alt;
  for (int p = 0; p < plan.n; p++) {
    hipLaunchKernelGGL(sort_hist_rows_kernel, dim3(nblocks), dim3(SORT_THREADS), 0, s, kin, n, n_dev, plan.shift[p],
                       plan.rb[p], table);
    hipLaunchKernelGGL(colscan_chunk_kernel, dim3(nchunks), dim3(256), 0, s, table, nblocks, n_dev, ctot);
    hipLaunchKernelGGL(colscan_top_kernel, dim3(1), dim3(256), 0, s, ctot, nchunks, n_dev,
                       ctot + (size_t)nchunks * SORT_MAX_BINS);
    hipLaunchKernelGGL(sort_scatter_kernel, dim3(nblocks), dim3(SORT_THREADS), 0, s, kin, vin, kout, vout, n,
                       plan.shift[p], plan.rb[p], table, ctot, nblocks, n_dev);
    uint64_t* tk = kin;
    kin = kout;
    kout = tk;
    uint32_t* tv = vin;
    vin = vout;
    vout = tv;
  }
  return hipGetLastError();
}

hipError_t dgs_launch_duplicate_sorted(const DgsView& v, const DgsCarve& c, const uint32_t* order, uint32_t* tt_sorted,
                                       uint32_t* offs_sorted, uint32_t* scan_tmp, uint32_t cap, hipStream_t s) {
  const uint64_t n = (uint64_t)v.K * v.P;
  const dim3 grid((uint32_t)((n + 255) / 256));
  // (laying tiles_touched out inside the last depth-sort pass was measured slower: 142 us for the gather there against
  // 98 us for this launch)
  hipLaunchKernelGGL(gather_u32_kernel, grid, dim3(256), 0, s, n, order, c.tiles_touched, tt_sorted);
  hipError_t e = dgs_launch_scan(tt_sorted, offs_sorted, n, scan_tmp, nullptr, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(duplicate_sorted_kernel, grid, dim3(256), 0, s, v, c.rows, order, tt_sorted, offs_sorted,
                     c.point_offsets, c.keys_unsorted, c.vals_unsorted, cap);
  return hipGetLastError();
}

// tile_cull, before the depth sort: per (k, Gaussian) record of rectangle, surviving-tile count and hit bits, natural order
hipError_t dgs_launch_cull_count(const DgsView& v, const DgsCarve& c, hipStream_t s) {
  const uint64_t n = (uint64_t)v.K * v.P;
  hipLaunchKernelGGL(cull_count_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, v, c.rows, c.tiles_touched,
                     c.cull_rec, c.cull_cnt);
  return hipGetLastError();
}

// tile_cull, after the depth sort: the counts into its order, offsets = their exclusive scan
// (total -> total_tight[0..1])
hipError_t dgs_launch_cull_offsets(const DgsView& v, const DgsCarve& c, uint32_t* total_tight, hipStream_t s,
                                   bool counts_in_order) {
  const uint64_t n = (uint64_t)v.K * v.P;
  if (!counts_in_order)   // (K * P > 2^24: the counts did not ride with the depth sort)
    hipLaunchKernelGGL(gather_cnt_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, n, c.gsort_vals, c.tt_sorted,
                       c.cull_cnt, c.tt_tight);
  return dgs_launch_scan(c.tt_tight, c.offs_tight, (uint64_t)v.K * v.P, c.scan_tmp, total_tight, s);
}

hipError_t dgs_launch_duplicate_tight(const DgsView& v, const DgsCarve& c, const uint32_t* order, uint32_t cap,
                                      hipStream_t s) {
  const uint64_t n = (uint64_t)v.K * v.P;
  hipLaunchKernelGGL(cull_emit_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, v, c.rows, order, c.tt_tight, c.cull_rec,
                     c.offs_tight, c.num_rendered, c.keys_unsorted, c.vals_unsorted, cap);
  return hipGetLastError();
}

hipError_t dgs_launch_ranges(const DgsView& v, const DgsCarve& c, uint32_t R, hipStream_t s, const uint32_t* n_dev,
                             int tile_shift) {
  const uint32_t ntiles = (uint32_t)v.K * (uint32_t)v.T;
  hipLaunchKernelGGL(ranges_search_kernel, dim3((ntiles + 255) / 256), dim3(256), 0, s, R, n_dev, c.keys_sorted, c.ranges,
                     tile_shift, ntiles);
  return hipGetLastError();
}

size_t dgs_depth_sort_tmp_words(int K, uint32_t P) {
  const uint64_t nb = ((uint64_t)P + DS_TILE - 1) / DS_TILE;
  const uint64_t nch = (nb + DS_CHUNK - 1) / DS_CHUNK;
  return (size_t)((uint64_t)K * (nb + nch) * DS_BINS + 256);
}

// keys [K*P] u32 = bits(depth) - DGS_DEPTH_KEY_BASE for a visible pair, 0xFFFFFFFF for an invisible one (destroyed);
// order out [K*P] u32 = flat (k, Gaussian) indices in (k, key, index) order.  Three 9-bit passes (+ the conditional
// fourth): the result always lands in `order`.  wide_flag: one device word, zero on entry.
hipError_t dgs_launch_depth_sort(uint32_t* keys, uint32_t* keys_alt, uint32_t* order, uint32_t* order_alt, int K,
                                 uint32_t P, uint32_t* tmp, uint32_t* vis_dst, uint32_t* wide_flag, hipStream_t s,
                                 const uint32_t* cnt_src, uint32_t* cnt_dst, bool drop_invisible) {
  if (K <= 0 || P == 0) return hipSuccess;
  if (drop_invisible && (vis_dst == nullptr || K > 128)) return hipErrorInvalidValue;
  if (cnt_src != nullptr && ((uint64_t)K * P > (1ull << 24) || cnt_dst == nullptr)) return hipErrorInvalidValue;
  const uint32_t nb = (P + DS_TILE - 1) / DS_TILE;
  const uint32_t nch = (nb + DS_CHUNK - 1) / DS_CHUNK;
  uint32_t* table = tmp;
  uint32_t* ctot = tmp + (size_t)K * nb * DS_BINS;
  // per-segment visible counts (drop_invisible): in the slack words behind the two tables
  uint32_t* segcnt = drop_invisible ? tmp + (size_t)K * (nb + nch) * DS_BINS + 64 : nullptr;
  const dim3 grid((uint32_t)K * nb), cgrid((uint32_t)K * nch);
  // buffers A = (keys, order), B = (keys_alt, order_alt):
  //   pass 0  A.keys          -> B.keys, A.order        pass 1  B.keys, A.order -> A.keys, B.order
  //   pass 2  A.keys, B.order -> A.order (+ B.keys if wide)     pass 3 (wide)  B.keys, A.order -> B.order, copied to A.order
  for (int pass = 0; pass < 4; pass++) {
    const uint32_t* kin = (pass & 1) ? keys_alt : keys;
    uint32_t* kout = (pass & 1) ? keys : keys_alt;
    const uint32_t* vin = (pass & 1) ? order : order_alt;
    uint32_t* vout = (pass & 1) ? order_alt : order;
    hipLaunchKernelGGL(dsort_hist_kernel, grid, dim3(DS_THREADS), 0, s, kin, P, nb, pass, table, wide_flag, segcnt);
    hipLaunchKernelGGL(dsort_colscan_chunk_kernel, cgrid, dim3(DS_BINS), 0, s, table, nb, nch, ctot, nch == 1 ? 1 : 0, P,
                       pass, wide_flag, segcnt);
    if (nch > 1)
      hipLaunchKernelGGL(dsort_colscan_top_kernel, dim3((uint32_t)K), dim3(DS_BINS), 0, s, ctot, nch, P, pass, wide_flag,
                         segcnt);
#define DGS_DS_SCATTER(PASS_)                                                                                         \
  hipLaunchKernelGGL((dsort_scatter_kernel<PASS_>), grid, dim3(DS_THREADS), 0, s, kin, vin, kout, vout, P, nb, nch, table, \
                     ctot, vis_dst, wide_flag, cnt_src, cnt_dst, segcnt)
    if (pass == 0) DGS_DS_SCATTER(0);
    else if (pass == 1) DGS_DS_SCATTER(1);
    else if (pass == 2) DGS_DS_SCATTER(2);
    else DGS_DS_SCATTER(3);
#undef DGS_DS_SCATTER
  }
  const uint64_t n = (uint64_t)K * P;
  const uint32_t cb = (uint32_t)std::min<uint64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(dsort_copy_if_wide_kernel, dim3(cb), dim3(256), 0, s, n, order_alt, order, wide_flag);
  return hipGetLastError();
}
