// binning.hip -- prefix sum, duplicateWithKeys, stable LSD radix sort, identifyTileRanges
// (replaces cub::DeviceScan::InclusiveSum, duplicateWithKeys, cub::DeviceRadixSort::SortPairs and
// identifyTileRanges: rasterizer_impl.cu:70-138,166,188-191,283,295-324).
//
// All integer work, HBM-bound.  Key = ((k*T + tile) << 32) | depth_bits so that ONE sort orders the
// duplicates of all K subframes; within a subframe the low 32+bits(T) bits are exactly the reference's key.
#include "dgs_common.h"

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
  }
  if (threadIdx.x == 0 && total != nullptr) *total = carry;
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

// ------------------------------------------------------------------------------------------- duplicate
// One thread per (subframe, Gaussian).  Also stamps the duplicate offset into the geometry row so that the
// backward compositing can address its contribution rows without an inverse permutation.
__global__ void __launch_bounds__(256)
duplicate_kernel(DgsView v, DgsRow* __restrict__ rows, const uint32_t* __restrict__ tiles_touched,
                 const uint32_t* __restrict__ offsets, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t n = (uint64_t)v.K * v.P;
  if (i >= n) return;
  if (tiles_touched[i] == 0) return;  // radii > 0 <=> tiles_touched > 0 (forward.cu:246-249,267)
  const uint32_t k = (uint32_t)(i / (uint64_t)v.P);
  const uint32_t g = (uint32_t)(i - (uint64_t)k * v.P);
  DgsRow* row = rows + i;
  const float x = row->x, y = row->y;
  const int radius = row->radius;
  uint32_t off = offsets[i];
  row->dup_offset = off;
  int minx, miny, maxx, maxy;
  dgs_get_rect(x, y, radius, v.gx, v.gy, minx, miny, maxx, maxy);
  const uint32_t dbits = __float_as_uint(row->depth);
  const uint32_t tbase = k * (uint32_t)v.T;
  for (int ty = miny; ty < maxy; ty++)
    for (int tx = minx; tx < maxx; tx++) {
      uint64_t key = (uint64_t)(tbase + (uint32_t)ty * (uint32_t)v.gx + (uint32_t)tx);
      key <<= 32;
      key |= dbits;
      keys[off] = key;
      vals[off] = g;
      off++;
    }
}

// ---------------------------------------------------------------------------------------------- ranges
__global__ void __launch_bounds__(256)
ranges_kernel(uint32_t L, const uint64_t* __restrict__ keys, uint2* __restrict__ ranges) {
  const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= L) return;
  const uint32_t currtile = (uint32_t)(keys[idx] >> 32);
  if (idx == 0)
    ranges[currtile].x = 0;
  else {
    const uint32_t prevtile = (uint32_t)(keys[idx - 1] >> 32);
    if (currtile != prevtile) {
      ranges[prevtile].y = idx;
      ranges[currtile].x = idx;
    }
  }
  if (idx == L - 1) ranges[currtile].y = L;
}

// ------------------------------------------------------------------------------------------ radix sort
// Stable LSD radix sort, RB-bit digits (RB <= 9; 10-bit digits were measured slower per key: 5 x 1.11 ms vs
// 6 x 0.81 ms at the metric config).  Per pass: (1) per-block digit histogram written
// digit-major [digit][block]; (2) exclusive scan of that table = global scatter bases; (3) scatter with
// stable in-block ranks.  A block owns SORT_TILE consecutive pairs; wave w owns a contiguous quarter of them,
// read in rounds of 64, so the stable order inside a block is (wave, round, lane).
constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS = 16;
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;  // 4096 pairs per block
constexpr int SORT_MAX_RB = 9;
constexpr int SORT_MAX_BINS = 1 << SORT_MAX_RB;

__global__ void __launch_bounds__(SORT_THREADS)
sort_hist_kernel(const uint64_t* __restrict__ keys, uint64_t n, int shift, int rb, uint32_t nblocks,
                 uint32_t* __restrict__ table) {
  __shared__ uint32_t h[SORT_MAX_BINS];
  const int bins = 1 << rb;
  for (int i = threadIdx.x; i < bins; i += SORT_THREADS) h[i] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * SORT_TILE;
  const uint32_t mask = (uint32_t)bins - 1;
#pragma unroll 4
  for (int r = 0; r < SORT_ITEMS; r++) {
    const uint64_t i = base + (uint64_t)r * SORT_THREADS + threadIdx.x;
    if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & mask], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < bins; i += SORT_THREADS) table[(uint64_t)i * nblocks + blockIdx.x] = h[i];
}

__global__ void __launch_bounds__(SORT_THREADS)
sort_scatter_kernel(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                    uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint64_t n, int shift, int rb,
                    uint32_t nblocks, const uint32_t* __restrict__ table) {
  // per-wave running digit counts, then (after the barrier) per-wave exclusive bases
  __shared__ uint32_t whist[SORT_THREADS / 64][SORT_MAX_BINS];
  const int bins = 1 << rb;
  const uint32_t mask = (uint32_t)bins - 1;
  const int lane = dgs_lane(), w = threadIdx.x >> 6;
  for (int i = lane; i < bins; i += 64) whist[w][i] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  const uint64_t wbase = (uint64_t)blockIdx.x * SORT_TILE + (uint64_t)w * (64 * SORT_ITEMS);
  uint64_t key[SORT_ITEMS];
  uint32_t val[SORT_ITEMS];
  uint32_t rank[SORT_ITEMS];
  volatile uint32_t* wh = whist[w];
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int r = 0; r < SORT_ITEMS; r++) {
    const uint64_t i = wbase + (uint64_t)r * 64 + lane;
    const bool valid = i < n;
    key[r] = valid ? keys_in[i] : ~0ull;
    val[r] = valid ? vals_in[i] : 0u;
    const uint32_t d = (uint32_t)(key[r] >> shift) & mask;
    // lanes of this wave holding the same digit (invalid lanes match nobody valid)
    uint64_t peers = __ballot(valid);
    for (int b = 0; b < rb; b++) {
      const uint64_t m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const uint32_t below = (uint32_t)__popcll(peers & lt_mask);
    uint32_t pre = 0;
    if (valid) pre = wh[d];
    __builtin_amdgcn_wave_barrier();
    if (valid && below == 0) wh[d] = pre + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    rank[r] = pre + below;
  }
  __syncthreads();
  // whist[w][d] now holds wave totals.  Turn them into global bases: table[d][block] + sum of earlier waves.
  for (int d = threadIdx.x; d < bins; d += SORT_THREADS) {
    uint32_t run = table[(uint64_t)d * nblocks + blockIdx.x];
#pragma unroll
    for (int ww = 0; ww < SORT_THREADS / 64; ww++) {
      const uint32_t c = whist[ww][d];
      whist[ww][d] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < SORT_ITEMS; r++) {
    const uint64_t i = wbase + (uint64_t)r * 64 + lane;
    if (i < n) {
      const uint32_t d = (uint32_t)(key[r] >> shift) & mask;
      const uint32_t pos = whist[w][d] + rank[r];
      keys_out[pos] = key[r];
      vals_out[pos] = val[r];
    }
  }
}

struct PassPlan {
  int n;
  int shift[16];
  int rb[16];
};

PassPlan plan_passes(int end_bit) {
  PassPlan p;
  if (end_bit < 1) end_bit = 1;
  if (end_bit > 64) end_bit = 64;
  p.n = (end_bit + SORT_MAX_RB - 1) / SORT_MAX_RB;
  int lo = end_bit / p.n, extra = end_bit % p.n, s = 0;
  for (int i = 0; i < p.n; i++) {
    p.shift[i] = s;
    p.rb[i] = lo + (i < extra ? 1 : 0);
    s += p.rb[i];
  }
  return p;
}

}  // namespace

size_t dgs_scan_tmp_words(uint64_t n) { return (size_t)((n + SCAN_TILE - 1) / SCAN_TILE) + 64; }

hipError_t dgs_launch_scan(const uint32_t* in, uint32_t* out, uint64_t n, uint32_t* tmp, uint32_t* total,
                           hipStream_t s) {
  if (n == 0) {
    if (total) return hipMemsetAsync(total, 0, sizeof(uint32_t), s);
    return hipSuccess;
  }
  const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL(scan_reduce_kernel, dim3((uint32_t)nb), dim3(SCAN_THREADS), 0, s, in, n, tmp);
  hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, tmp, nb, total);
  hipLaunchKernelGGL(scan_apply_kernel, dim3((uint32_t)nb), dim3(SCAN_THREADS), 0, s, in, out, n, tmp);
  return hipGetLastError();
}

int dgs_sort_num_passes(int end_bit) { return plan_passes(end_bit).n; }

size_t dgs_sort_tmp_words(uint64_t n) {
  const uint64_t nblocks = (n + SORT_TILE - 1) / SORT_TILE;
  const uint64_t table = nblocks * SORT_MAX_BINS;
  return (size_t)(table + dgs_scan_tmp_words(table) + 64);
}

hipError_t dgs_launch_sort(uint64_t* keys, uint32_t* vals, uint64_t* keys_alt, uint32_t* vals_alt, uint64_t n,
                           int end_bit, uint32_t* tmp, int* result_in_alt, hipStream_t s) {
  const PassPlan plan = plan_passes(end_bit);
  *result_in_alt = plan.n & 1;
  if (n == 0) return hipSuccess;
  const uint32_t nblocks = (uint32_t)((n + SORT_TILE - 1) / SORT_TILE);
  uint32_t* table = tmp;
  uint32_t* scan_tmp = tmp + (uint64_t)nblocks * SORT_MAX_BINS;
  uint64_t* kin = keys;
  uint32_t* vin = vals;
  uint64_t* kout = keys_alt;
  uint32_t* vout = vals_alt;
  for (int p = 0; p < plan.n; p++) {
    const uint64_t tn = (uint64_t)nblocks << plan.rb[p];
    hipLaunchKernelGGL(sort_hist_kernel, dim3(nblocks), dim3(SORT_THREADS), 0, s, kin, n, plan.shift[p], plan.rb[p],
                       nblocks, table);
    hipError_t e = dgs_launch_scan(table, table, tn, scan_tmp, nullptr, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sort_scatter_kernel, dim3(nblocks), dim3(SORT_THREADS), 0, s, kin, vin, kout, vout, n,
                       plan.shift[p], plan.rb[p], nblocks, table);
    uint64_t* tk = kin;
    kin = kout;
    kout = tk;
    uint32_t* tv = vin;
    vin = vout;
    vout = tv;
  }
  return hipGetLastError();
}

hipError_t dgs_launch_duplicate(const DgsView& v, const DgsCarve& c, hipStream_t s) {
  const uint64_t n = (uint64_t)v.K * v.P;
  hipLaunchKernelGGL(duplicate_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, v, c.rows, c.tiles_touched,
                     c.point_offsets, c.keys_unsorted, c.vals_unsorted);
  return hipGetLastError();
}

hipError_t dgs_launch_ranges(const DgsView& v, const DgsCarve& c, uint32_t R, hipStream_t s) {
  hipError_t e = hipMemsetAsync(c.ranges, 0, (size_t)v.K * v.T * sizeof(uint2), s);
  if (e != hipSuccess) return e;
  if (R > 0) hipLaunchKernelGGL(ranges_kernel, dim3((R + 255) / 256), dim3(256), 0, s, R, c.keys_sorted, c.ranges);
  return hipGetLastError();
}
