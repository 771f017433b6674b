// composite.hip -- per-tile alpha compositing, forward and backward, for all K subframes in one launch
// (replaces renderCUDA<3> forward, forward.cu:273-392, and backward, backward.cu:463-640).
//
// CDNA4 design (not the reference's 256-thread / one-pixel-per-thread block):
//   * ONE wave64 owns one 16x16 tile; each lane carries 4 pixels, one in each 8x8 quadrant.  No block
//     barriers, no __syncthreads_count: termination votes and culling masks are 64-bit ballots in SGPRs.
//   * The tile's sorted duplicate list is consumed in batches of 64: lane j gathers Gaussian j's 48-byte
//     geometry row with three 16-byte loads, tests its alpha>=1/255 ellipse against the four quadrants
//     (exact ellipse/box minimum with rounding slack) and publishes the row to LDS.  The per-Gaussian loop then walks only the
//     set bits of the ballot masks, so a (Gaussian, quadrant) pair that cannot contribute costs nothing.
//     Culling is conservative, hence the per-pixel tests below are exactly the reference's.
//   * Backward: no global atomics.  The 10 per-Gaussian partial sums are accumulated over a lane's pixels
//     in registers, over the wave through LDS (value-major rows, 40 lanes add 16 values each), and stored
//     as ONE 48-byte contribution row per (tile, Gaussian) duplicate:
//       [sum w*dx, sum w*dy, sum w*dx*dx, sum w*dx*dy, sum w*dy*dy, sum w, dL_dr, dL_dg, dL_db, dL_ddepth].  geometry_bwd.hip sums a Gaussian's rows in
//     duplicate order, so the whole backward is bitwise reproducible (the reference issues 10 float
//     atomicAdds per (pixel, Gaussian), backward.cu:599-637).
#include <type_traits>

#include "dgs_common.h"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));  // lowers to v_pk_{mul,add,fma}_f32 on gfx950

// Backward: the ten per-lane sums of a list entry become one contribution row through LDS -- ten conflict-free 4-byte
// stores per lane, 40 lanes read 16 values each and add them, the totals go straight to the row in global memory (one
// 40-byte store per entry).  Round 2's reduce-scatter on the VALU (v_permlane*_swap + DPP, rows staged in LDS) measured
// 5.70-5.77 ms against 5.41 ms on the same box: variants/bwd_valu_reduce_scatter.patch.
// waves (= tiles) per block; the waves never synchronise with each other.  (1 or 2 waves per block -- finer scheduling of
// tiles whose lists differ in length -- measured 5.25-5.28 ms against 5.09-5.13 for the backward, 2.39 against 2.42 for the
// forward: kept at 4)
constexpr int CW = 4;

struct TileCtx {
  int k, tx, ty;
  uint32_t r0, r1;
};

// XCD-aware block -> tile-group map: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous
// run of tiles so that neighbouring tiles, which share Gaussians, hit the same L2.  Speed only.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t per_xcd) { return (b & 7u) * per_xcd + (b >> 3); }

__device__ __forceinline__ bool load_tile_ctx(const DgsView& v, const uint2* __restrict__ ranges, uint32_t per_xcd,
                                              TileCtx& t) {
  const uint32_t logical = xcd_remap(blockIdx.x, per_xcd);
  // threadIdx.x >> 6 is wave-uniform, but the compiler only knows that through readfirstlane: without it the
  // tile range, every loop bound and the ballot masks end up in VGPRs under exec-mask control flow.
  const uint32_t gw = logical * CW + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t KT = (uint32_t)v.K * (uint32_t)v.T;
  if (gw >= KT) return false;
  t.k = (int)(gw / (uint32_t)v.T);
  const uint32_t tile = gw - (uint32_t)t.k * (uint32_t)v.T;
  t.ty = (int)(tile / (uint32_t)v.gx);
  t.tx = (int)(tile - (uint32_t)t.ty * (uint32_t)v.gx);
  const uint2 r = ranges[gw];
  t.r0 = r.x;
  t.r1 = r.y;
  return true;
}

// Conservative-but-tight test: can Gaussian (x, y, conic a/b/c, opacity op) reach alpha >= 1/255 at some pixel
// centre of the 8x8 quadrant?  alpha = op*exp(power) >= 1/255  <=>  q(d) = a dx^2 + 2 b dx dy + c dy^2 <= r2 with
// r2 = 2 ln(255 op), so the question is whether the minimum of the convex quadratic q over the box
// d in [dx_lo, dx_hi] x [dy_lo, dy_hi] (d = mean - pixel) is <= r2.  The minimum is 0 if the box contains the
// origin, otherwise it sits on one of the four edges, where q is a 1-D quadratic with a closed-form clamped
// minimiser.  A slack on r2 and a relative slack on q keep it conservative under fp32 rounding, so the exact
// per-pixel tests of the reference still decide every pair that survives.
// DGS_LOG2_CONIC=1 (rounds 2-5, now an A/B build only): the conic sits in LDS pre-multiplied by -0.5 log2(e) (-log2(e) for
// the cross term), so that the quadratic form feeds v_exp_f32 directly -- one VALU instruction fewer per pass.  But the three
// products round the COEFFICIENTS (6e-8 relative each), and for a needle -- a nearly singular conic, terms of +-5000 that
// cancel to a power of -2 -- that is 3e-4 in the exponent, the same sign at every pixel of the Gaussian: dL_dconic of one
// such Gaussian came out 2e-4 off at the metric size (1.03e-4 of its column; tools/r06_calls/find_pixel.py bisected it to
// "no pixel: all of them").  Default since round 6: the conic is scaled by the exact factors -0.5 / -1 / -0.5 only, the
// quadratic form is the reference's `power` (forward.cu:351) evaluated with the same five instructions, and ONE multiply by
// log2(e) -- whose rounding is not amplified by the cancellation -- follows it.
#ifndef DGS_LOG2_CONIC
#define DGS_LOG2_CONIC 0
#endif
#if DGS_LOG2_CONIC
constexpr float K_LOG2E = -1.4426950408889634f;        // cross term:  -b dx dy        -> log2 domain
constexpr float K_HALF_LOG2E = -0.7213475204444817f;   // square terms: -0.5 a dx^2 ...
#else
constexpr float K_LOG2E = -1.0f;                       // exact scalings: the coefficients keep their bits
constexpr float K_HALF_LOG2E = -0.5f;
#endif
// (-0.5 (a dx^2 + c dy^2) - b dx dy) [* log2(e)] from the pre-scaled conic (A, B, C) = (-0.5 a, -b, -0.5 c) [* log2(e)],
// as the argument of v_exp_f32 (2^x)
__device__ __forceinline__ float dgs_power2(float A, float B, float C, float dx, float dy) {
  const float u = fmaf(B, dy, A * dx);
#if DGS_LOG2_CONIC
  return fmaf(C * dy, dy, dx * u);
#else
  return fmaf(C * dy, dy, dx * u) * 1.4426950408889634f;
#endif
}

// DGS_EXACT_POWER=1 (A/B build only, git history: tools/r05_calls/parity_ab.sh; never the shipped library): the three places where the
// per-pair arithmetic departs from the letter of the reference are put back --
//   * `power` in the natural domain with the reference's own expression and term order, every operation rounded on its
//     own (forward.cu:351 / backward.cu:572: -0.5f * (a dx dx + c dy dy) - b dx dy), then ONE multiply by log2(e) in front
//     of v_exp_f32 (instead of the conic pre-scaled into the log2 domain: one more rounding per term);
//   * T / (1 - alpha) divided (backward.cu:579) instead of multiplied by v_rcp_f32's 1-ulp reciprocal;
//   * the `power > 0` skip kept for every entry (forward.cu:354, backward.cu:573), also for positive definite conics.
// DESIGN.md 5 has the measured effect on the unstable-pixel mask, on dL_dconic and on the step time.
#ifndef DGS_EXACT_POWER
#define DGS_EXACT_POWER 0
#endif
// DGS_DIV_REFINE=1 (A/B build only): T / (1 - alpha) with one residual correction behind the hardware reciprocal.  Round 6:
// +0.38 ms on the compositing backward (5.31 against 4.93 ms, three interleaved runs) and dL_dconic's column figure at the
// metric size does not move by a digit (1.0289e-4 both ways): the recurrence's rounding is not what that figure measures.
#ifndef DGS_DIV_REFINE
#define DGS_DIV_REFINE 0
#endif
#if DGS_EXACT_POWER
__device__ __forceinline__ float dgs_power_ref(float a, float b, float c, float dx, float dy) {
  const float t0 = __fmul_rn(__fmul_rn(a, dx), dx), t1 = __fmul_rn(__fmul_rn(c, dy), dy);
  const float t2 = __fmul_rn(__fmul_rn(b, dx), dy);
  return __fsub_rn(__fmul_rn(-0.5f, __fadd_rn(t0, t1)), t2);
}
#endif

// v_min_f32 without the canonicalising v_max_f32 x, x that fminf() of a value merged from two control-flow paths gets
// (au is never a signalling NaN: it comes out of v_mul / v_cndmask)
__device__ __forceinline__ float dgs_min_raw(float c, float x) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(c), "v"(x));
  return r;
}

// DGS_TIMELINE=1 (diagnostic build only, tools/tile_timeline.py): every wave of the compositing backward leaves its
// start and end time (s_memrealtime, 100 MHz) and where it ran in a caller-provided buffer, [K*T][3] u64 by tile -- how many waves are
// resident over the launch, i.e. how long the tail of the wave-per-tile schedule is.
#ifndef DGS_TIMELINE
#define DGS_TIMELINE 0
#endif
#if DGS_TIMELINE
__device__ unsigned long long* g_timeline = nullptr;
#endif

using CullGauss = DgsCull;
#define make_cull dgs_make_cull
#define cull_hit dgs_cull_hit

// ------------------------------------------------------------------------------------------------ forward
// WITHDEPTH = false: the caller does not consume the depth image (DgsForwardOut.out_depth == NULL; the default training
// loss never reads it): the depth channel drops out of the per-pair math and nothing is stored for it.
// KEEP = false: an inference call (DgsProblem.forward_only): final_T / n_contrib, which only the backward reads, are not
// stored (and `last` drops out of the per-pair math).
// CHK = true (parity tests only, DgsForwardOut.debug_contrib_checksum): per pixel, the wrap-around sum of
// pos * 2654435761 over the list positions of the pairs that contribute -- which per-pair decisions this traversal took.
template <bool WITHDEPTH, bool KEEP, bool CHK = false>
__global__ void __launch_bounds__(64 * CW)
composite_fwd_kernel(DgsView v, uint32_t per_xcd, const uint2* __restrict__ ranges,
                     const uint32_t* __restrict__ point_list, const uint64_t* __restrict__ keys,
                     const DgsRow* __restrict__ rows,
                     const float* __restrict__ bg, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                     float* __restrict__ out_color, float* __restrict__ out_depth, uint32_t* __restrict__ checksum = nullptr) {
  // one 48-byte LDS row per list entry: (x, y, A, B | C, op, r, g | b, depth, -, -): one address per read
  __shared__ float4 s_row[CW][64 * 3];
  TileCtx t;
  if (!load_tile_ctx(v, ranges, per_xcd, t)) return;
  const int lane = dgs_lane(), w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lx = lane & 7, ly = lane >> 3;
  const int px0 = t.tx * DGS_TILE + lx, py0 = t.ty * DGS_TILE + ly;  // quadrant 0 pixel; +8 for the others
  const float pxf0 = (float)px0, pxf1 = (float)(px0 + 8), pyf0 = (float)py0, pyf1 = (float)(py0 + 8);
  const float qx0 = (float)(t.tx * DGS_TILE), qy0 = (float)(t.ty * DGS_TILE);

  // T carries the pixel's "done" flag in its sign: positive while the pixel is live (then T >= 1e-4), and flipped to
  // -T by the pair that would take it below 1e-4 (forward.cu:362-367), after which T (1 - alpha) < 1e-4 holds for every
  // later pair by itself and |T| is the final transmittance.  No separate per-lane flag to test and update per pair.
  float T[4], C0[4], C1[4], C2[4], Dd[4];
  uint32_t last[4], chk[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    C0[q] = C1[q] = C2[q] = Dd[q] = 0.0f;
    last[q] = 0;
    chk[q] = 0;
    const int px = px0 + (q & 1) * 8, py = py0 + (q >> 1) * 8;
    T[q] = (px < v.W && py < v.H) ? 1.0f : -1.0f;
  }
  const DgsRow* krows = rows + (size_t)t.k * v.P;
  const uint32_t n = t.r1 - t.r0;
  uint64_t alive[4];
#pragma unroll
  for (int q = 0; q < 4; q++) alive[q] = __ballot(T[q] > 0.0f);

  for (uint32_t base = 0; base < n; base += 64) {
    if ((alive[0] | alive[1] | alive[2] | alive[3]) == 0) break;  // whole tile terminated (forward.cu:325)
    const bool has = base + lane < n;
    float4 A = make_float4(0, 0, 0, 0), B = A, Cc = A;
    if (has) {
      // compact keys: the Gaussian sits between the emission index and the tile in the key itself
      const uint32_t g = v.pack_tile_shift > 0
                             ? (uint32_t)(keys[t.r0 + base + lane] >> v.pack_g_shift) &
                                   ((1u << (v.pack_tile_shift - v.pack_g_shift)) - 1u)
                             : point_list[t.r0 + base + lane];
      const float4* rp = reinterpret_cast<const float4*>(krows + g);
      A = rp[0];
      B = rp[1];
      Cc = rp[2];
    }
    const CullGauss cg = make_cull(A.z, A.w, B.x, B.y);
    // entries whose conic is positive definite (as evaluated in fp32 by make_cull -- the SAME predicate in the forward
    // and the backward): power = -q/2 <= 0 for every pixel in exact arithmetic, so the reference's `power > 0` skip
    // (forward.cu:354-355) can only fire through rounding, on pixels the parity tests already classify as unstable;
    // those entries take the pass variant without the compare
    const uint64_t pdm = __ballot(!cg.always);
    uint64_t m[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      // d = mean - pixel over the quadrant's pixel centres [q0, q0+7]
      const float ex = A.x - (qx0 + (float)((q & 1) * 8)), ey = A.y - (qy0 + (float)((q >> 1) * 8));
      const bool hit = has && cull_hit(cg, ex - 7.0f, ex, ey - 7.0f, ey);
      const uint64_t bh = __ballot(hit);
      m[q] = alive[q] ? bh : 0ull;
    }
    // the conic goes to LDS pre-multiplied by -0.5 log2(e) (-log2(e) for the cross term): the per-pair exponent is
    // then a 5-instruction quadratic form that feeds v_exp_f32 directly (forward and backward use the same bits)
#if DGS_EXACT_POWER
    s_row[w][3 * lane] = make_float4(A.x, A.y, A.z, A.w);
    s_row[w][3 * lane + 1] = make_float4(B.x, B.y, B.z, B.w);
#else
    s_row[w][3 * lane] = make_float4(A.x, A.y, A.z * K_HALF_LOG2E, A.w * K_LOG2E);
    s_row[w][3 * lane + 1] = make_float4(B.x * K_HALF_LOG2E, B.y, B.z, B.w);
#endif
    s_row[w][3 * lane + 2] = make_float4(Cc.x, Cc.y, 0.0f, 0.0f);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // The batch loop exists twice: without the `power <= 0` compare when every entry that can contribute has a positive
    // definite conic (all but pathological batches), with it otherwise.  One wave-uniform decision per batch.
    const uint64_t mu_all = m[0] | m[1] | m[2] | m[3];
    auto run_batch = [&](auto checkc) {
    constexpr bool CHECK = decltype(checkc)::value;
    uint64_t mu = mu_all;
    while (mu) {
      const int j = __builtin_ctzll(mu);
      mu &= mu - 1;
      const float4* rowj = &s_row[w][3 * j];
      const float4 a = rowj[0];
      const float4 b = rowj[1];
      const float2 c = *reinterpret_cast<const float2*>(rowj + 2);
      // 1-based list position of this entry, materialised in a VGPR once per entry: as a wave-uniform (SGPR) value the
      // select below needs a v_mov per quadrant pass (a select cannot take an SGPR mask and an SGPR source)
      uint32_t posv = base + (uint32_t)j + 1u;
      asm volatile("" : "+v"(posv));
      // forward.cu:348-380, branch-free per lane: a pair that fails one of the reference's tests gets alpha = 0, which
      // leaves T, C, D and `last` untouched (T >= 1e-4 always, so alpha = 0 can never terminate).
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if ((m[q] >> j) & 1ull) {  // wave-uniform: this Gaussian can reach quadrant q
          const float dx = a.x - ((q & 1) ? pxf1 : pxf0);
          const float dy = a.y - ((q >> 1) ? pyf1 : pyf0);
#if DGS_EXACT_POWER
          const float power = dgs_power_ref(a.z, a.w, b.x, dx, dy);
          const float alpha_raw = fminf(0.99f, b.y * __builtin_amdgcn_exp2f(power * 1.4426950408889634f));
          const bool ok = (power <= 0.0f) && (alpha_raw >= 1.0f / 255.0f);
#else
          const float power = dgs_power2(a.z, a.w, b.x, dx, dy);  // log2(e) * the reference's `power`
          const float alpha_raw = fminf(0.99f, b.y * __builtin_amdgcn_exp2f(power));
          const bool ok = (!CHECK || power <= 0.0f) && (alpha_raw >= 1.0f / 255.0f);
#endif
          const float alpha = ok ? alpha_raw : 0.0f;
          const float test_T = T[q] * (1.0f - alpha);
          const bool stop = test_T < 0.0001f;      // also every pair of a pixel that is already done (T < 0)
          const float wgt = stop ? 0.0f : alpha * T[q];
          C0[q] += b.z * wgt;
          C1[q] += b.w * wgt;
          C2[q] += c.x * wgt;
          if (WITHDEPTH) Dd[q] += c.y * wgt;
          T[q] = stop ? -fabsf(T[q]) : test_T;     // alpha = 0 leaves a live T unchanged
          if (KEEP) last[q] = (ok && !stop) ? posv : last[q];
          if (CHK) chk[q] += (ok && !stop) ? posv * 2654435761u : 0u;
        }
      }
    }
    };
    if ((~pdm & mu_all) == 0ull)
      run_batch(std::false_type{});
    else
      run_batch(std::true_type{});
#pragma unroll
    for (int q = 0; q < 4; q++) alive[q] = __ballot(T[q] > 0.0f);
    __builtin_amdgcn_wave_barrier();  // LDS rows are rewritten by the next batch
  }

  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const size_t N = (size_t)v.W * v.H;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int px = px0 + (q & 1) * 8, py = py0 + (q >> 1) * 8;
    if (px < v.W && py < v.H) {
      const size_t pix = (size_t)py * v.W + px;
      const float Tf = fabsf(T[q]);
      if (KEEP) {
        final_T[(size_t)t.k * N + pix] = Tf;
        n_contrib[(size_t)t.k * N + pix] = last[q];
      }
      float* oc = out_color + (size_t)t.k * 3 * N;
      oc[pix] = C0[q] + Tf * bg0;
      oc[N + pix] = C1[q] + Tf * bg1;
      oc[2 * N + pix] = C2[q] + Tf * bg2;
      if (WITHDEPTH) out_depth[(size_t)t.k * N + pix] = Dd[q] + Tf * v.z_far;
      if (CHK) checksum[(size_t)t.k * N + pix] = chk[q];
    }
  }
}

// ----------------------------------------------------------------------------------------------- backward
// TIGHT: the duplicate index travels in the low key word (tile_cull emission).  HASDEPTH: dL_ddepth is given (the
// default training loss does not use the depth output: the depth channel then drops out of the per-pair math)
template <bool TIGHT, bool HASDEPTH>
__global__ void __launch_bounds__(64 * CW)
composite_bwd_kernel(DgsView v, uint32_t per_xcd, const uint2* __restrict__ ranges,
                     const uint32_t* __restrict__ point_list, const uint64_t* __restrict__ keys,
                     const DgsRow* __restrict__ rows,
                     const float* __restrict__ bg, const float* __restrict__ final_T,
                     const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,
                     const float* __restrict__ dL_ddepth, const uint32_t* __restrict__ dup_off,
                     float* __restrict__ contrib) {
  __shared__ float4 s_row[CW][64 * 3];  // (x, y, A, B | C, op, r, g | b, depth, -, -) per list entry
  // the ten per-lane sums of a list entry, value-major (row stride 68 floats: the 16-byte column reads of the rows start
  // in different banks)
  __shared__ __attribute__((aligned(16))) float s_part[CW][10][68];
  TileCtx t;
  if (!load_tile_ctx(v, ranges, per_xcd, t)) return;
#if DGS_TIMELINE
  const unsigned long long t_begin = wall_clock64();
#endif
  const int lane = dgs_lane(), w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lx = lane & 7, ly = lane >> 3;
  const int px0 = t.tx * DGS_TILE + lx, py0 = t.ty * DGS_TILE + ly;
  const float pxf0 = (float)px0, pxf1 = (float)(px0 + 8), pyf0 = (float)py0, pyf1 = (float)(py0 + 8);
  const float qx0 = (float)(t.tx * DGS_TILE), qy0 = (float)(t.ty * DGS_TILE);
  const size_t N = (size_t)v.W * v.H;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  // contribution-row slots: [S_wx, S_wy, S_xx, S_xy, S_yy, S_w, r, g, b, depth]
  // reduction through LDS: lane (row = lane >> 2, quarter = lane & 3), row < 10 (9 without a depth gradient), sums 16 lanes'
  // values of sum `row`; the quad's four partial sums are combined with two DPP adds and lane quarter 0 stores the total
  const int rrow = lane >> 2, rq = lane & 3;
  const bool rlane = rrow < (HASDEPTH ? 10 : 9);
  float* const contrib_lane = contrib + (rrow < 10 ? rrow : 0);   // this lane's column of every contribution row
  static_assert(sizeof(float) * 68 == 272, "the store offsets below are multiples of one s_part row");
  // LDS byte offset of this wave's s_part block (the low half of a flat LDS address is the offset inside the LDS)
  const uint32_t part_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)(&s_part[w][0][0]));
  

  // per-pixel channel state kept as (r,g) and (b,depth) pairs so the channel arithmetic issues as packed fp32
  float T[4];
  v2f gA[4], gB[4];
  // sum_ch dL_dpixel[ch] * (colour behind the current pair)[ch].  It starts at the background's dot product
  // (bg . dL_dpixel + z_far * dL_ddepthpix): the background is what lies behind the last contributor, and with that
  // start T_i * (c.g - accg_i) already contains the reference's separate background term
  // -T_final / (1 - alpha_i) * bg_dot_dpixel (backward.cu:613-618), because T_i (1 - alpha_i) prod_{k>i}(1 - alpha_k)
  // = T_final.
  float accg[4];
  uint32_t last[4];
  uint32_t maxc = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int px = px0 + (q & 1) * 8, py = py0 + (q >> 1) * 8;
    const bool inside = (px < v.W && py < v.H);
    const size_t pix = (size_t)py * v.W + px;
    const float Tfin = inside ? final_T[(size_t)t.k * N + pix] : 0.0f;
    T[q] = Tfin;
    last[q] = inside ? n_contrib[(size_t)t.k * N + pix] : 0u;
    const float* gp = dL_dpix + (size_t)t.k * 3 * N;
    const float g0 = inside ? gp[pix] : 0.0f;
    const float g1 = inside ? gp[N + pix] : 0.0f;
    const float g2 = inside ? gp[2 * N + pix] : 0.0f;
    const float gd = (inside && dL_ddepth != nullptr) ? dL_ddepth[(size_t)t.k * N + pix] : 0.0f;
    gA[q] = (v2f){g0, g1};
    gB[q] = (v2f){g2, gd};
    accg[q] = bg0 * g0 + bg1 * g1 + bg2 * g2 + v.z_far * gd;
    maxc = max(maxc, last[q]);
  }
  // wave-wide max of n_contrib: entries at or beyond it are skipped by every pixel (backward.cu:566-568); the same
  // per quadrant lets a (Gaussian, quadrant) pass be dropped once all 64 pixels of that quadrant are past their last
  // contributor
  uint32_t maxq[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    uint32_t mq = last[q];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mq = max(mq, (uint32_t)__shfl_xor((int)mq, d, 64));
    maxq[q] = __builtin_amdgcn_readfirstlane(mq);
  }
  maxc = max(max(maxq[0], maxq[1]), max(maxq[2], maxq[3]));

  const DgsRow* krows = rows + (size_t)t.k * v.P;
  const uint32_t n = t.r1 - t.r0;
  const uint32_t nbatch = (n + 63) / 64;
  for (uint32_t bi = nbatch; bi-- > 0;) {
    const uint32_t base = bi * 64;
    const bool has = base + lane < n;
    float4 A = make_float4(0, 0, 0, 0), B = A, Cc = A;
    uint32_t u = 0;
    if (has) {
      uint32_t g;
      if (TIGHT && v.pack_tile_shift > 0) {   // compact keys: tile | Gaussian | emission index in one word
        const uint64_t key = keys[t.r0 + base + lane];
        g = (uint32_t)(key >> v.pack_g_shift) & ((1u << (v.pack_tile_shift - v.pack_g_shift)) - 1u);
        u = (uint32_t)(key & ((1ull << v.pack_g_shift) - 1ull));
      } else {
        g = point_list[t.r0 + base + lane];
        if (TIGHT) u = reinterpret_cast<const uint32_t*>(keys)[2 * (size_t)(t.r0 + base + lane)];
      }
      const float4* rp = reinterpret_cast<const float4*>(krows + g);
      A = rp[0];
      B = rp[1];
      Cc = rp[2];
      if (!TIGHT) {
        // index of this (tile, Gaussian) duplicate in duplicate order: row-major inside the Gaussian's tile rect
        int minx, miny, maxx, maxy;
        dgs_get_rect(A.x, A.y, __float_as_int(Cc.w), v.gx, v.gy, minx, miny, maxx, maxy);
        u = dup_off[(size_t)t.k * v.P + g] + (uint32_t)((t.ty - miny) * (maxx - minx) + (t.tx - minx));
      }
    }
    uint64_t m[4] = {0, 0, 0, 0};
    uint64_t pdm = 0;   // entries with a positive definite conic (same predicate as the forward): no `power <= 0` compare
    if (base < maxc) {
      const CullGauss cg = make_cull(A.z, A.w, B.x, B.y);
      pdm = __ballot(!cg.always);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float ex = A.x - (qx0 + (float)((q & 1) * 8)), ey = A.y - (qy0 + (float)((q >> 1) * 8));
        const bool hit = has && (base + lane < maxq[q]) && cull_hit(cg, ex - 7.0f, ex, ey - 7.0f, ey);
        m[q] = __ballot(hit);
      }
#if DGS_EXACT_POWER
      s_row[w][3 * lane] = make_float4(A.x, A.y, A.z, A.w);
      s_row[w][3 * lane + 1] = make_float4(B.x, B.y, B.z, B.w);
#else
      s_row[w][3 * lane] = make_float4(A.x, A.y, A.z * K_HALF_LOG2E, A.w * K_LOG2E);
      s_row[w][3 * lane + 1] = make_float4(B.x * K_HALF_LOG2E, B.y, B.z, B.w);
#endif
      // the duplicate's contribution-row slot rides in the row (after the colour's third channel, where the depth sits
      // when a depth gradient is asked for): the entry loop gets it with the row's own LDS read instead of a v_readlane
      s_row[w][3 * lane + 2] = HASDEPTH ? make_float4(Cc.x, Cc.y, __uint_as_float(u), 0.0f)
                                        : make_float4(Cc.x, __uint_as_float(u), 0.0f, 0.0f);
    }
    const float4 z4 = make_float4(0, 0, 0, 0);
    // totals go straight to the contribution rows (one 40-byte store per entry); an entry that no pixel of the tile
    // reaches any more is never walked: its row is zero-filled here
    if (has && (((m[0] | m[1] | m[2] | m[3]) >> lane) & 1ull) == 0ull) {
      float4* dst = reinterpret_cast<float4*>(contrib + (size_t)u * DGS_CONTRIB_F);
      dst[0] = z4;
      dst[1] = z4;
      dst[2] = z4;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // (two copies of the batch loop, with and without the `power <= 0` compare: see the forward)
    const uint64_t mu_all = m[0] | m[1] | m[2] | m[3];
    auto run_batch = [&](auto checkc) {
    constexpr bool CHECK = decltype(checkc)::value;
    uint64_t mu = mu_all;
    while (mu) {
      const int j = 63 - __builtin_clzll(mu);  // back to front
      mu &= ~(1ull << j);
      const float4* rowj = &s_row[w][3 * j];
      const float4 a = rowj[0];
      const float4 b = rowj[1];
      float2 c;
      uint32_t uj;
      if (HASDEPTH) {
        const float4 c4 = rowj[2];
        c = make_float2(c4.x, c4.y);
        uj = __float_as_uint(c4.z);
      } else {
        const float2 c2 = *reinterpret_cast<const float2*>(rowj + 2);
        c = make_float2(c2.x, 0.0f);
        uj = __float_as_uint(c2.y);
      }
      const uint32_t pos = base + (uint32_t)j;  // 0-based position in the tile list
      // raw per-lane sums over this lane's pixels; with w = (opacity*G) * dL_dalpha every geometric gradient of
      // backward.cu:620-637 is a per-Gaussian linear map of {sum w, sum w*dx, sum w*dy, sum w*dx*dx, sum w*dx*dy,
      // sum w*dy*dy}; that map (conic / opacity / 0.5*W factors) is applied once per (subframe, Gaussian) in
      // geometry_bwd.hip instead of once per pixel here.
      // (starting the accumulators from three 16-byte LDS reads of zeros instead of v_mov measured slower: the compiler
      // already starts them with the first executed pass when quadrant 0 is hit)
      float S_w = 0, S_wx = 0, S_wy = 0, S_xx = 0, S_xy = 0, S_yy = 0;
      v2f sA = {0.0f, 0.0f}, sB = {0.0f, 0.0f};  // (dL_dr, dL_dg), (dL_db, dL_ddepth)
      const v2f colA = {b.z, b.w}, colB = {c.x, c.y};
      // one (Gaussian, quadrant) pass
      auto pass = [&](auto qc) __attribute__((always_inline)) {
          constexpr int q = decltype(qc)::value;
          const float dx = a.x - ((q & 1) ? pxf1 : pxf0);
          const float dy = a.y - ((q >> 1) ? pyf1 : pyf0);
          // backward.cu:566-637, branch-free per lane.  A pair that the reference skips gets alpha = 0: then
          // T/(1-alpha) = T, every gradient term is an exact 0 and the colour-behind recurrence below is the identity,
          // so the results are bit-identical to skipping.  The recurrence is applied eagerly (right after the pair is
          // used) instead of one pair late with a remembered last_alpha / last_color as in the reference: same
          // operations in the same order, 5 fewer live registers per pixel.
#if DGS_EXACT_POWER
          const float power = dgs_power_ref(a.z, a.w, b.x, dx, dy);
          const float au_any = b.y * __builtin_amdgcn_exp2f(power * 1.4426950408889634f);
          const bool ok = (pos < last[q]) && (power <= 0.0f) && (au_any >= 1.0f / 255.0f);
#else
          const float power = dgs_power2(a.z, a.w, b.x, dx, dy);  // log2(e) * the reference's `power`
          const float au_any = b.y * __builtin_amdgcn_exp2f(power);
          const bool ok = (pos < last[q]) && (!CHECK || power <= 0.0f) && (au_any >= 1.0f / 255.0f);
#endif
          const float au = ok ? au_any : 0.0f;  // opacity * G: the unclamped alpha the backward differentiates
          const float alpha = dgs_min_raw(0.99f, au);
#if DGS_EXACT_POWER
          T[q] = T[q] / (1.0f - alpha);
#elif DGS_DIV_REFINE
          // T / (1 - alpha) by the hardware reciprocal plus one residual correction: q = T r, q' = q + (T - d q) r.  The plain
          // product T * v_rcp_f32(d) carries up to ~1.5 ulp per pair and the recurrence compounds it over the hundreds of
          // pairs behind a pixel (dL_dconic sat at 1.03e-4 of its column scale at the metric size with the exact exempt set);
          // the corrected quotient is the division's to within an ulp for two more FMAs per pass.
          const float one_m = 1.0f - alpha;
          const float inv1ma = __builtin_amdgcn_rcpf(one_m);
          const float q0 = T[q] * inv1ma;
          T[q] = fmaf(fmaf(-one_m, q0, T[q]), inv1ma, q0);
#else
          const float inv1ma = __builtin_amdgcn_rcpf(1.0f - alpha);
          T[q] = T[q] * inv1ma;
#endif
          const float dchannel_dcolor = alpha * T[q];
          // dL_dalpha = sum_ch (c[ch] - accum_rec[ch]) * dL_dpixel[ch] (backward.cu:590-600) only ever uses the
          // colour behind the pair through its dot product with dL_dpixel, and that dot product obeys the same
          // recurrence as the colour itself (it is linear): keep the one scalar per pixel instead of four channels
          float cg = fmaf(colB.x, gB[q].x, fmaf(colA.y, gA[q].y, colA.x * gA[q].x));
          if (HASDEPTH) cg = fmaf(colB.y, gB[q].y, cg);
          const float behind = cg - accg[q];
          // acc <- alpha c + (1 - alpha) acc in its lerp form: reuses cg - acc (the identity for alpha = 0)
          accg[q] = fmaf(alpha, behind, accg[q]);
          const float wgt = au * (behind * T[q]);  // au == 0 for a skipped pair
          const float wx = wgt * dx, wy = wgt * dy;
          // (scalar FMAs: a packed v_pk_fma_f32 was measured slower than the two scalar ones it replaces)
          sA.x = fmaf(gA[q].x, dchannel_dcolor, sA.x);
          sA.y = fmaf(gA[q].y, dchannel_dcolor, sA.y);
          sB.x = fmaf(gB[q].x, dchannel_dcolor, sB.x);
          if (HASDEPTH) sB.y = fmaf(gB[q].y, dchannel_dcolor, sB.y);
          S_w += wgt;
          S_wx += wx;
          S_wy += wy;
          S_xx = fmaf(wx, dx, S_xx);
          S_xy = fmaf(wx, dy, S_xy);
          S_yy = fmaf(wy, dy, S_yy);
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      using I3 = std::integral_constant<int, 3>;
      // (every way of letting the first executed pass START the sums instead of zeroing nine accumulators -- 15 straight-line
      // hit-mask variants, only the four one-quadrant ones, or a compare / branch chain on the first hit quadrant -- made the
      // kernel 5-17 % SLOWER although it removes instructions: more copies of the pass body in the loop.  variants/NOTES.md)
      if ((m[0] >> j) & 1ull) pass(I0{});   // wave-uniform
      if ((m[1] >> j) & 1ull) pass(I1{});
      if ((m[2] >> j) & 1ull) pass(I2{});
      if ((m[3] >> j) & 1ull) pass(I3{});
      {
        // 10 wave sums through LDS instead of the VALU: ten conflict-free 4-byte stores per lane, then 40 lanes read 16
        // values each (four 16-byte loads) and add them in a fixed tree (deterministic); the cross-lane instructions of
        // round 2's VALU reduce-scatter (8 v_permlane*_swap, 7 DPP adds: quarter-rate classes) become 15 plain adds and
        // 2 DPP adds.  The kernel then sits between two limits (round 4, PMC): the VALU issues for ~78 % of the cycles
        // (instruction counts x per-class issue cost, profiles/valu_peak_r04.json) and the CU's LDS is 56 % busy by
        // SQ_LDS_IDX_ACTIVE -- 70-80 % when the stores are priced at what their address + data transfer takes.
        // Hence ds_write_addtid_b32 for the stores: LDS address = M0 + offset + 4 * lane, no address VGPR to ship, 2 cycles
        // instead of 4 (MI355X_MICROARCH.md, LDS table): 5.35 -> 5.12 ms per launch on the same box.  The s_nop: an SALU
        // write of M0 needs one wait state before an add-TID LDS instruction reads it -- the compiler inserts that for its
        // own code, not inside asm; without it the stores of some waves went to a stale M0 (tools/addtid_probe.hip).
        // A wave's LDS operations execute in order, so the loads below see these stores without a wait.
        if (HASDEPTH)
          asm volatile(
              "s_mov_b32 m0, %10\n\ts_nop 0\n\t"
              "ds_write_addtid_b32 %0 offset:0\n\tds_write_addtid_b32 %1 offset:272\n\t"
              "ds_write_addtid_b32 %2 offset:544\n\tds_write_addtid_b32 %3 offset:816\n\t"
              "ds_write_addtid_b32 %4 offset:1088\n\tds_write_addtid_b32 %5 offset:1360\n\t"
              "ds_write_addtid_b32 %6 offset:1632\n\tds_write_addtid_b32 %7 offset:1904\n\t"
              "ds_write_addtid_b32 %8 offset:2176\n\tds_write_addtid_b32 %9 offset:2448"
              :: "v"(S_wx), "v"(S_wy), "v"(S_xx), "v"(S_xy), "v"(S_yy), "v"(S_w), "v"(sA.x), "v"(sA.y), "v"(sB.x),
                 "v"(sB.y), "s"(part_base) : "m0", "memory");
        else
          asm volatile(
              "s_mov_b32 m0, %9\n\ts_nop 0\n\t"
              "ds_write_addtid_b32 %0 offset:0\n\tds_write_addtid_b32 %1 offset:272\n\t"
              "ds_write_addtid_b32 %2 offset:544\n\tds_write_addtid_b32 %3 offset:816\n\t"
              "ds_write_addtid_b32 %4 offset:1088\n\tds_write_addtid_b32 %5 offset:1360\n\t"
              "ds_write_addtid_b32 %6 offset:1632\n\tds_write_addtid_b32 %7 offset:1904\n\t"
              "ds_write_addtid_b32 %8 offset:2176"
              :: "v"(S_wx), "v"(S_wy), "v"(S_xx), "v"(S_xy), "v"(S_yy), "v"(S_w), "v"(sA.x), "v"(sA.y), "v"(sB.x),
                 "s"(part_base) : "m0", "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float tot = 0.0f;
        if (rlane) {
          const float4* pr = reinterpret_cast<const float4*>(&s_part[w][rrow][16 * rq]);
          const float4 x0 = pr[0], x1 = pr[1], x2 = pr[2], x3 = pr[3];
          // (a packed tree -- the 16-byte loads deliver aligned register pairs, 8 v_pk_add_f32 / v_add instead of 15
          // v_add, no moves -- measured slower on the same box: 5.47-5.52 against 5.41 ms)
          tot = (((x0.x + x0.y) + (x0.z + x0.w)) + ((x1.x + x1.y) + (x1.z + x1.w))) +
                (((x2.x + x2.y) + (x2.z + x2.w)) + ((x3.x + x3.y) + (x3.z + x3.w)));
        }
        tot = dgs_quad_sum(tot);
        asm volatile("" : "+v"(tot));
        {
          // (the row address is one quarter-rate v_mad_u64_u32 per entry; forming it on the scalar unit -- readlane, two
          // multiplies, add with carry, global_store with a scalar base -- measured the same: 5.41 / 5.45 ms.)  Without a
          // depth gradient the lanes of column 9 read nothing and store the 0 they started from; columns 10 and 11 of a
          // row are never read as values.
          if (rrow < 10 && rq == 0) contrib_lane[(size_t)uj * DGS_CONTRIB_F] = tot;
        }
        __builtin_amdgcn_wave_barrier();   // the next entry's stores follow this entry's loads (LDS runs a wave's ops in order)
      }
    }
    };
    if ((~pdm & mu_all) == 0ull)
      run_batch(std::false_type{});
    else
      run_batch(std::true_type{});
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
#if DGS_TIMELINE
  if (lane == 0 && g_timeline != nullptr) {
    const size_t gw = (size_t)t.k * v.T + (size_t)t.ty * v.gx + t.tx;
    g_timeline[3 * gw] = t_begin;
    g_timeline[3 * gw + 1] = wall_clock64();
    // the wave's slot (block, wave), the XCD that really ran it (XCC_ID, hardware register 20, bits [3:0]) and the one its
    // block index implies
    g_timeline[3 * gw + 2] = ((unsigned long long)(blockIdx.x * CW + w) << 16) |
                             ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 8) |
                             (unsigned long long)(blockIdx.x & 7u);
  }
#endif
}

}  // namespace

#if DGS_TIMELINE
extern "C" int dgs_debug_set_timeline(void* buffer) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(buffer);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &p, sizeof(p)) == hipSuccess ? 0 : -3;
}
#endif

static uint32_t per_xcd_blocks(const DgsView& v) {
  const uint64_t KT = (uint64_t)v.K * v.T;
  const uint64_t nblk = (KT + CW - 1) / CW;
  return (uint32_t)((nblk + 7) / 8);
}

hipError_t dgs_launch_composite_fwd(const DgsView& v, const DgsCarve& c, const float* bg, float* out_color,
                                    float* out_depth, hipStream_t s, uint32_t* checksum) {
  const uint32_t per = per_xcd_blocks(v);
  if (per == 0) return hipSuccess;
#define DGS_CFWD(WD, KP)                                                                                                  \
  hipLaunchKernelGGL((composite_fwd_kernel<WD, KP>), dim3(per * 8), dim3(64 * CW), 0, s, v, per, c.ranges, c.point_list, \
                     c.keys_sorted, c.rows, bg, c.final_T, c.n_contrib, out_color, out_depth)
  const bool keep = c.final_T != nullptr && c.n_contrib != nullptr;
  if (checksum != nullptr) {   // the parity tests' variant: depth and state always on
    if (out_depth == nullptr || !keep) return hipErrorInvalidValue;
    hipLaunchKernelGGL((composite_fwd_kernel<true, true, true>), dim3(per * 8), dim3(64 * CW), 0, s, v, per, c.ranges,
                       c.point_list, c.keys_sorted, c.rows, bg, c.final_T, c.n_contrib, out_color, out_depth, checksum);
    return hipGetLastError();
  }
  if (out_depth != nullptr) {
    if (keep) DGS_CFWD(true, true); else DGS_CFWD(true, false);
  } else {
    if (keep) DGS_CFWD(false, true); else DGS_CFWD(false, false);
  }
#undef DGS_CFWD
  return hipGetLastError();
}

hipError_t dgs_launch_composite_bwd(const DgsView& v_all, const DgsCarve& c, const float* bg, const float* dL_dpix,
                                    const float* dL_ddepth, float* contrib, hipStream_t s, int k0, int k1) {
  // subframes [k0, k1) of the view (k1 < 0: all of them): the kernel sees a view of k1 - k0 subframes whose per-subframe
  // arrays start at subframe k0; list positions (ranges, keys, duplicate offsets, contribution rows) are absolute
  if (k1 < 0) { k0 = 0; k1 = v_all.K; }
  DgsView v = v_all;
  v.K = k1 - k0;
  const size_t N = (size_t)v.W * v.H;
  const uint32_t per = per_xcd_blocks(v);
  if (per == 0) return hipSuccess;
  const uint2* ranges = c.ranges + (size_t)k0 * v.T;
  const DgsRow* rows = c.rows + (size_t)k0 * v.P;
  const float* final_T = c.final_T + (size_t)k0 * N;
  const uint32_t* n_contrib = c.n_contrib + (size_t)k0 * N;
  dL_dpix += (size_t)k0 * 3 * N;
  if (dL_ddepth != nullptr) dL_ddepth += (size_t)k0 * N;
#define DGS_CBWD(TI, HD)                                                                                            \
  hipLaunchKernelGGL((composite_bwd_kernel<TI, HD>), dim3(per * 8), dim3(64 * CW), 0, s, v, per, ranges, c.point_list, \
                     c.keys_sorted, rows, bg, final_T, n_contrib, dL_dpix, dL_ddepth, c.point_offsets, contrib)
  if (v.tile_cull) {
    if (dL_ddepth != nullptr) DGS_CBWD(true, true); else DGS_CBWD(true, false);
  } else {
    if (dL_ddepth != nullptr) DGS_CBWD(false, true); else DGS_CBWD(false, false);
  }
#undef DGS_CBWD
  return hipGetLastError();
}
