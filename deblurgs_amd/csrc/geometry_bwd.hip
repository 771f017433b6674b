// geometry_bwd.hip -- K-fused backward of the per-Gaussian stages
// (replaces computeCov2DCUDA, backward.cu:145-295, and preprocessCUDA<3> backward, backward.cu:367-460, with
// computeColorFromSH backward :20-140 and computeCov3D backward :299-362).
//
// One thread per Gaussian walks the K subframes:
//   * reads that (subframe, Gaussian)'s total of the contribution rows written by composite_bwd (summed in
//     duplicate order by contrib_reduce_kernel; replaces the reference's 10 float atomics per (pixel, Gaussian):
//     deterministic);
//   * conic -> cov2D -> cov3D / mean gradients, projection and depth terms, SH gradients;
//   * accumulates dL/d{mean3D, SH, opacity, cov3D} over the K subframes IN REGISTERS and writes them once
//     (the reference allocates, zero-fills and re-accumulates 220 B/Gaussian per subframe,
//     rasterize_points.cu:162-174).  cov3D -> scale/rotation is linear in dL_dcov3D, so it runs once on the
//     K-summed dL_dcov3D instead of K times.
//   * dL_dviewmatrix / dL_dprojmatrix (reference: 28 same-address float atomics per visible Gaussian,
//     backward.cu:279-293,432-457) are wave-reduced with DPP, combined per block in LDS in wave order and
//     written as per-block partials that a second kernel sums in block order: deterministic as well.
#include "dgs_common.h"

// (the 21 per-subframe pose terms of a wave are summed through LDS; with DPP butterflies, round 2, the launch took 0.69
// instead of 0.61 ms: variants/NOTES.md)

namespace {

__device__ const float SH_C0 = 0.28209479177387814f;
__device__ const float SH_C1 = 0.4886025119029199f;
__device__ const float SH_C2[] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                  -1.0925484305920792f, 0.5462742152960396f};
__device__ const float SH_C3[] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                  0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                  -0.5900435899266435f};

constexpr int GB_THREADS = 256;
constexpr int NMAT = 24;  // 12 view entries + 8 proj entries + 1 shared last-column value (+3 pad)

struct M3 {  // column-major like glm::mat3
  float m[3][3];
};
__device__ __forceinline__ M3 mul(const M3& A, const M3& B) {
  M3 R;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) R.m[c][r] = A.m[0][r] * B.m[c][0] + A.m[1][r] * B.m[c][1] + A.m[2][r] * B.m[c][2];
  return R;
}
__device__ __forceinline__ M3 tr(const M3& A) {
  M3 R;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) R.m[c][r] = A.m[r][c];
  return R;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Stage 1: the contribution rows of every visible (subframe, Gaussian) pair, summed in duplicate order (deterministic),
// written as ONE total per pair (12 floats in a 64-byte slot: whole aligned lines) at the pair's NATURAL index k * P + g.  Pairs are walked in the order the
// duplicates were laid out in ((k, depth, index): consecutive quads read consecutive row segments).
// The geometry kernel then reads its totals with coalesced, independent loads -- no duplicate offset to chase, and the
// emit pass of the forward no longer stores one.  A visible pair whose every tile was culled gets zeros.
// (Round 3 tried a streaming variant -- a wave owns 64 consecutive pairs and pulls their contiguous row span through LDS
// with fully coalesced 1 KB loads -- and measured the same time, 0.82 vs 0.80 ms: the kernel is bound by the 11.8 M
// scattered 48-byte WRITES of the totals to their natural index, not by how the rows are read.)
__global__ void __launch_bounds__(256)
contrib_reduce_kernel(uint64_t n, const uint32_t* __restrict__ status, const uint32_t* __restrict__ order,
                      const uint32_t* __restrict__ tt_visible, const uint32_t* __restrict__ tiles,
                      const uint32_t* __restrict__ offsets, const float* __restrict__ contrib,
                      float* __restrict__ sums) {
  if (status[5] != 0u) return;  // capacity mode, truncated lists: the offsets point past the rows that were written
  // four lanes per (subframe, Gaussian): lane part p in {0,1,2} owns the p-th float4 of every row of the segment
  // (part 3 idles), so a quad reads each 48-byte row with one contiguous access and no cross-lane sum is needed
  // (an XCD-contiguous block map -- every XCD's L2 seeing about two subframes' slots -- measured slower: 709 vs 671 us)
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t j = t >> 2;
  const uint32_t part = (uint32_t)t & 3u;
  if (j >= n || (part == 3u && DGS_SUMS_F < 16)) return;
  // the four index words are requested together (one round trip instead of a chain of three) ...
  const uint32_t vis = tt_visible[j], nt = tiles[j], off = offsets[j], dst = order[j];
  if (vis == 0u) return;  // invisible pair: the geometry kernel never reads its slot
  float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (nt > 0 && part < 3u) {
    const float4* cp = reinterpret_cast<const float4*>(contrib + (size_t)off * DGS_CONTRIB_F) + part;
    // ... and so are the first four rows of the segment (a pair has ~3 duplicates on average), then eight at a time:
    // the loop is never a chain of dependent HBM round trips.  Rows are added strictly in duplicate order, one after the
    // other: the duplicates tile culling removes would have contributed exact zeros, and x + 0 = x keeps the totals --
    // and every gradient -- bit-identical between tile_cull = 0 and 1 (a pairwise or grouped order does not: tried).
    float4 q0 = cp[0], q1 = a, q2 = a, q3 = a;
    if (nt > 1) q1 = cp[3];
    if (nt > 2) q2 = cp[6];
    if (nt > 3) q3 = cp[9];
    a = q0;
    if (nt > 1) { a.x += q1.x; a.y += q1.y; a.z += q1.z; a.w += q1.w; }
    if (nt > 2) { a.x += q2.x; a.y += q2.y; a.z += q2.z; a.w += q2.w; }
    if (nt > 3) { a.x += q3.x; a.y += q3.y; a.z += q3.z; a.w += q3.w; }
    uint32_t r = 4;
    for (; r + 8 <= nt; r += 8) {
      float4 q[8];
#pragma unroll
      for (int i = 0; i < 8; i++) q[i] = cp[3 * (r + i)];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        a.x += q[i].x;
        a.y += q[i].y;
        a.z += q[i].z;
        a.w += q[i].w;
      }
    }
    for (; r < nt; r++) {
      const float4 q = cp[3 * r];
      a.x += q.x;
      a.y += q.y;
      a.z += q.z;
      a.w += q.w;
    }
  }
  reinterpret_cast<float4*>(sums + (size_t)dst * DGS_SUMS_F)[part] = a;
}

// computeColorFromSH backward (backward.cu:20-140) of subframe k: adds to a_sh and to (dmean_x, dmean_y, dmean_z).  PS = pointer
// to the pair's three pre-activation colour values.  A macro, expanded in the single-loop kernels exactly where the block
// always stood (their code is unchanged) and in the second loop of the SPLIT kernel.
#define DGS_SH_BACKWARD(PS)  \
  {  \
        const float* cam = campos + 3 * k;  \
        const float dox = mx - cam[0], doy = my - cam[1], doz = mz - cam[2];  \
        const float len = sqrtf(dox * dox + doy * doy + doz * doz);  \
        const float x = dox / len, y = doy / len, z = doz / len;  \
        float dRGB[3];  \
_Pragma("unroll")  \
        for (int ch = 0; ch < 3; ch++) {  \
          const float psv = (PS)[ch];  \
          float f = psv;  \
          if (v.use_sigmoid) {  \
            const float sg = sigmoidf_(psv);  \
            f = sg * (1.0f - sg);  \
          }  \
          dRGB[ch] = dcol[ch] * f;  \
        }  \
        float ddir[3] = {0, 0, 0};  /* dL/ddir = sum_ch dRGBd{x,y,z}[ch] * dRGB[ch] */  \
        float cf[MAXC];  /* dRGB/dsh_j (same for the three channels) */  \
        cf[0] = SH_C0;  \
        if (MAXC > 1 && ncoef > 1) {  \
          cf[1] = -SH_C1 * y;  \
          cf[2] = SH_C1 * z;  \
          cf[3] = -SH_C1 * x;  \
_Pragma("unroll")  \
          for (int ch = 0; ch < 3; ch++) {  \
            ddir[0] += (-SH_C1 * sh[9 + ch]) * dRGB[ch];  \
            ddir[1] += (-SH_C1 * sh[3 + ch]) * dRGB[ch];  \
            ddir[2] += (SH_C1 * sh[6 + ch]) * dRGB[ch];  \
          }  \
        }  \
        if (MAXC > 4 && ncoef > 4) {  \
          const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;  \
          cf[4] = SH_C2[0] * xy;  \
          cf[5] = SH_C2[1] * yz;  \
          cf[6] = SH_C2[2] * (2.f * zz - xx - yy);  \
          cf[7] = SH_C2[3] * xz;  \
          cf[8] = SH_C2[4] * (xx - yy);  \
_Pragma("unroll")  \
          for (int ch = 0; ch < 3; ch++) {  \
            const float s4 = sh[12 + ch], s5 = sh[15 + ch], s6 = sh[18 + ch], s7 = sh[21 + ch], s8 = sh[24 + ch];  \
            ddir[0] += (SH_C2[0] * y * s4 + SH_C2[2] * 2.f * -x * s6 + SH_C2[3] * z * s7 + SH_C2[4] * 2.f * x * s8) *  \
                       dRGB[ch];  \
            ddir[1] += (SH_C2[0] * x * s4 + SH_C2[1] * z * s5 + SH_C2[2] * 2.f * -y * s6 + SH_C2[4] * 2.f * -y * s8) *  \
                       dRGB[ch];  \
            ddir[2] += (SH_C2[1] * y * s5 + SH_C2[2] * 2.f * 2.f * z * s6 + SH_C2[3] * x * s7) * dRGB[ch];  \
          }  \
          if (MAXC > 9 && ncoef > 9) {  \
            cf[9] = SH_C3[0] * y * (3.f * xx - yy);  \
            cf[10] = SH_C3[1] * xy * z;  \
            cf[11] = SH_C3[2] * y * (4.f * zz - xx - yy);  \
            cf[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);  \
            cf[13] = SH_C3[4] * x * (4.f * zz - xx - yy);  \
            cf[14] = SH_C3[5] * z * (xx - yy);  \
            cf[15] = SH_C3[6] * x * (xx - 3.f * yy);  \
_Pragma("unroll")  \
            for (int ch = 0; ch < 3; ch++) {  \
              const float s9 = sh[27 + ch], s10 = sh[30 + ch], s11 = sh[33 + ch], s12 = sh[36 + ch],  \
                          s13 = sh[39 + ch], s14 = sh[42 + ch], s15 = sh[45 + ch];  \
              ddir[0] += (SH_C3[0] * s9 * 3.f * 2.f * xy + SH_C3[1] * s10 * yz + SH_C3[2] * s11 * -2.f * xy +  \
                          SH_C3[3] * s12 * -3.f * 2.f * xz + SH_C3[4] * s13 * (-3.f * xx + 4.f * zz - yy) +  \
                          SH_C3[5] * s14 * 2.f * xz + SH_C3[6] * s15 * 3.f * (xx - yy)) * dRGB[ch];  \
              ddir[1] += (SH_C3[0] * s9 * 3.f * (xx - yy) + SH_C3[1] * s10 * xz +  \
                          SH_C3[2] * s11 * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * s12 * -3.f * 2.f * yz +  \
                          SH_C3[4] * s13 * -2.f * xy + SH_C3[5] * s14 * -2.f * yz +  \
                          SH_C3[6] * s15 * -3.f * 2.f * xy) * dRGB[ch];  \
              ddir[2] += (SH_C3[1] * s10 * xy + SH_C3[2] * s11 * 4.f * 2.f * yz +  \
                          SH_C3[3] * s12 * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * s13 * 4.f * 2.f * xz +  \
                          SH_C3[5] * s14 * (xx - yy)) * dRGB[ch];  \
            }  \
          }  \
        }  \
_Pragma("unroll")  \
        for (int j = 0; j < MAXC; j++)  \
          if (j < ncoef) {  \
            a_sh[3 * j + 0] += cf[j] * dRGB[0];  \
            a_sh[3 * j + 1] += cf[j] * dRGB[1];  \
            a_sh[3 * j + 2] += cf[j] * dRGB[2];  \
          }  \
  /* dnormvdv (auxiliary.h:107-117) */  \
        const float sum2 = dox * dox + doy * doy + doz * doz;  \
        const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);  \
        dmean_x += ((+sum2 - dox * dox) * ddir[0] - doy * dox * ddir[1] - doz * dox * ddir[2]) * invsum32;  \
        dmean_y += (-dox * doy * ddir[0] + (sum2 - doy * doy) * ddir[1] - doz * doy * ddir[2]) * invsum32;  \
        dmean_z += (-dox * doz * ddir[0] - doy * doz * ddir[1] + (sum2 - doz * doz) * ddir[2]) * invsum32;  \
  }

// waves per SIMD the register allocation must leave room for: the degree-3 instantiation came out at 255 VGPRs + 3 AGPRs,
// i.e. ONE wave per SIMD, without a bound (two waves = 256 registers in all); the others are left alone
template <int MAXC>  // MAXC = SH coefficients held in registers: 1, 4, 9 or 16
__global__ void __launch_bounds__(GB_THREADS) __attribute__((amdgpu_waves_per_eu(MAXC > 9 ? 2 : 1)))
geometry_bwd_kernel(DgsView v, const float* __restrict__ means3D, const float* __restrict__ shs,
                    const float* __restrict__ shs_rest, const float* __restrict__ opacities_raw,
                    const float* __restrict__ scales, const float* __restrict__ rotations,
                    const float* __restrict__ cov3D_precomp, const float* __restrict__ viewm,
                    const float* __restrict__ projm, const float* __restrict__ campos,
                    const DgsRow* __restrict__ rows,
                    const float* __restrict__ cov3Ds,
                    const float* __restrict__ pre_sigmoid, const uint32_t* __restrict__ tiles_touched,
                    const float* __restrict__ contrib, const uint32_t* __restrict__ status, float hinge_scale,
                    float* __restrict__ dL_dmeans3D,
                    float* __restrict__ dL_dmeans2D, float* __restrict__ dL_dsh, float* __restrict__ dL_dsh_rest,
                    float* __restrict__ dL_dcolors,
                    float* __restrict__ dL_dopacity, float* __restrict__ dL_dscales, float* __restrict__ dL_drots,
                    float* __restrict__ dL_dcov3D_out, double* __restrict__ partials, int g_begin, int g_end,
                    const int32_t* __restrict__ radii, float* __restrict__ st_max_radii, float* __restrict__ st_accum,
                    float* __restrict__ st_denom, float st_inc) {
  extern __shared__ __attribute__((aligned(16))) float s_part[];  // [waves][K][NMAT]
  __shared__ __attribute__((aligned(16))) float s_m[GB_THREADS / 64][21][68];  // one subframe's 21 pose terms of every lane
  if (status[5] != 0u) return;  // capacity mode, truncated lists (see contrib_reduce_kernel); the caller discards the step
  // Gaussians [g_begin, g_end) (g_begin a multiple of the block size): a caller may run the per-Gaussian half in index
  // chunks so that the all-reduce of one chunk's gradients overlaps the next chunk's kernel (dgs_backward_geometry)
  const int idx = g_begin + blockIdx.x * GB_THREADS + threadIdx.x;
  const bool valid = idx < g_end;
  const int gi = valid ? idx : 0;
  const int lane = dgs_lane(), w = threadIdx.x >> 6;
  const float mx = means3D[3 * gi], my = means3D[3 * gi + 1], mz = means3D[3 * gi + 2];
  const float* c3p = (cov3D_precomp != nullptr ? cov3D_precomp : cov3Ds) + 6 * (size_t)gi;
  float c3[6];
#pragma unroll
  for (int i = 0; i < 6; i++) c3[i] = c3p[i];
  const float h_x = v.focal_x, h_y = v.focal_y;
  const int ncoef = (v.D + 1) * (v.D + 1);

  float a_mean[3] = {0, 0, 0};
  float a_cov[6] = {0, 0, 0, 0, 0, 0};
  float a_col[3] = {0, 0, 0};
  float a_op = 0;
  // SH coefficients are read once per Gaussian and stay in registers across the K subframes (only the first
  // (D+1)^2 are touched, MAXC >= (D+1)^2 by construction of the launch).
  // SPLIT (MAXC = 16, SH degree 3): 48 coefficients + 48 gradient sums live across the subframe loop next to the
  // covariance / projection chain need the whole 256-VGPR budget and more (round 5: 256 VGPRs + AGPR copies, ONE wave per
  // SIMD, 1.24 ms against 0.63 ms for MAXC = 9).  The SH part then runs as a second loop over the subframes after the first
  // one has retired its temporaries: it re-reads the pair's colour-gradient totals (two 16-byte loads) and the activation
  // mask, and adds its direction term to dL/dmean afterwards.  MAXC <= 9 keeps the single loop (and its exact sums).
  constexpr bool SPLIT = MAXC > 9;
  float a_sh[MAXC * 3];
  float sh[MAXC * 3];
#define DGS_LOAD_SH()                                                                                               \
  {                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < MAXC * 3; i++) a_sh[i] = 0.0f;                                           \
    if (shs != nullptr) {                                                                                          \
      if (shs_rest == nullptr) {                                                                                   \
        const float* shp = shs + (size_t)gi * v.M * 3;                                                             \
        _Pragma("unroll") for (int i = 0; i < MAXC * 3; i++) sh[i] = (i < ncoef * 3) ? shp[i] : 0.0f;              \
      } else {                                                                                                     \
        const float* dcp = shs + (size_t)gi * 3;                                                                   \
        const float* rsp = shs_rest + (size_t)gi * (v.M - 1) * 3;                                                  \
        _Pragma("unroll") for (int i = 0; i < MAXC * 3; i++)                                                       \
            sh[i] = (i < ncoef * 3) ? ((i < 3) ? dcp[i] : rsp[i - 3]) : 0.0f;                                      \
      }                                                                                                            \
    }                                                                                                              \
  }
  if (!SPLIT) DGS_LOAD_SH()

  // Software pipeline over the subframes (this kernel runs at two waves per SIMD): the loads of one (subframe, Gaussian)
  // -- tiles_touched, the geometry row, the 48-byte total at the natural index -- and this kernel runs at two
  // waves per SIMD, so they are issued ahead: row of k+2 and contribution row of k+1 are in flight while k computes.
  struct RowPf {
    uint32_t nt;
    int32_t rad;    // fused densification statistics only
    uint32_t cmask; // relu activation: clamp-mask bits of the pair's colour (DgsRow::dup_offset)
    float4 ga, gb;  // x, y, cx, cy | cz, op, r, g
  };
  // densification statistics (train.py:188-193, scene/gaussian_model.py:456-458; DgsBackwardIO.stats_*): the thread has
  // every subframe's screen gradient of its Gaussian in hand, in subframe order -- the separate statistics launch re-read
  // the [K,P,3] gradient this kernel had just written (56 us and 0.4 GB per metric step)
  const bool stats = st_max_radii != nullptr;
  float st_mr = 0.0f, st_ac = 0.0f, st_dn = 0.0f;
  if (stats && valid) {
    st_mr = st_max_radii[gi];
    st_ac = st_accum[gi];
    st_dn = st_denom[gi];
  }
  auto load_row = [&](int k) {
    RowPf r;
    const size_t o = (size_t)k * v.P + gi;
    const float4* rowp = reinterpret_cast<const float4*>(rows + o);
    r.nt = valid ? tiles_touched[o] : 0u;
    r.rad = (stats && valid) ? radii[o] : 0;
    r.ga = rowp[0];   // unconditional (no dependent hop); rows of invisible pairs are never used
    r.gb = rowp[1];
    r.cmask = (shs != nullptr && !v.use_sigmoid) ? rows[o].dup_offset : 0u;
    return r;
  };
  struct SumPf {
    float4 r0, r1, r2;
  };
  auto load_sums = [&](int k) {
    // the (subframe, Gaussian) total contrib_reduce_kernel left at the pair's natural index (zeros when tile culling
    // left the pair no tile); loaded unconditionally -- the slot of an invisible pair is never used
    SumPf c;
    const float4* cp = reinterpret_cast<const float4*>(contrib + ((size_t)k * v.P + gi) * DGS_SUMS_F);
    c.r0 = cp[0];
    c.r1 = cp[1];
    c.r2 = cp[2];
    return c;
  };
  RowPf row1 = load_row(0);
  RowPf row2 = load_row(v.K > 1 ? 1 : 0);
  SumPf sum1 = load_sums(0);

  for (int k = 0; k < v.K; k++) {
    const float* V = viewm + 16 * k;
    const float* F = projm + 16 * k;
    const size_t o = (size_t)k * v.P + gi;
    float mat[NMAT];
#pragma unroll
    for (int i = 0; i < NMAT; i++) mat[i] = 0.0f;
    float g2x = 0.0f, g2y = 0.0f;
    const RowPf cur = row1;
    const SumPf cs = sum1;
    row1 = row2;
    if (k + 2 < v.K) row2 = load_row(k + 2);
    // (the slot of an invisible pair -- 22 % of them at the metric size, 64 bytes each -- is not fetched: row1 is subframe
    // k + 1's row by now, loaded two iterations ago)
    if (k + 1 < v.K && row1.nt > 0) sum1 = load_sums(k + 1);   // (two subframes ahead, 249 VGPRs: same time -- the kernel waits on its own
                                                // dependent arithmetic at two waves per SIMD, not on these loads; without
                                                // any SH state, 149 VGPRs and three waves, it still takes 0.50 of 0.68 ms)
    const uint32_t ntiles = cur.nt;
    if (ntiles > 0) {
      const float4 ga = cur.ga, gb = cur.gb;
      const float4 r0 = cs.r0, r1 = cs.r1, r2 = cs.r2;
      const float s[10] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y};
      // raw sums -> the reference's per-Gaussian sinks (backward.cu:620-637; see composite.hip):
      //   dL_dmean2D = -(0.5 W, 0.5 H) * (cx*Swx + cy*Swy, cz*Swy + cy*Swx),  dL_dconic = -0.5 * (Sxx, Sxy, Syy),
      //   dL_dopacity = Sw / opacity
      g2x = -(0.5f * (float)v.W) * (ga.z * s[0] + ga.w * s[1]);
      g2y = -(0.5f * (float)v.H) * (gb.x * s[1] + ga.w * s[0]);
      const float dcon_x = -0.5f * s[2], dcon_y = -0.5f * s[3], dcon_w = -0.5f * s[4];
      a_op += (gb.y > 0.0f) ? s[5] / gb.y : 0.0f;
      const float dcol[3] = {s[6], s[7], s[8]};
      const float ddepth = s[9];

      // ---- computeCov2DCUDA (backward.cu:145-295)
      float tx = V[0] * mx + V[4] * my + V[8] * mz + V[12];
      float ty = V[1] * mx + V[5] * my + V[9] * mz + V[13];
      const float tz_ = V[2] * mx + V[6] * my + V[10] * mz + V[14];
      const float limx = 1.3f * v.tanfovx, limy = 1.3f * v.tanfovy;
      const float txtz = tx / tz_, tytz = ty / tz_;
      tx = fminf(limx, fmaxf(-limx, txtz)) * tz_;
      ty = fminf(limy, fmaxf(-limy, tytz)) * tz_;
      const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.0f : 1.0f;
      const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.0f : 1.0f;
      M3 J = {{{h_x / tz_, 0.0f, -(h_x * tx) / (tz_ * tz_)}, {0.0f, h_y / tz_, -(h_y * ty) / (tz_ * tz_)}, {0, 0, 0}}};
      M3 Wm = {{{V[0], V[4], V[8]}, {V[1], V[5], V[9]}, {V[2], V[6], V[10]}}};
      M3 Vrk = {{{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}}};
      M3 T = mul(Wm, J);
      M3 cov2D = mul(mul(tr(T), tr(Vrk)), T);
      const float a = cov2D.m[0][0] + 0.3f;
      const float b = cov2D.m[0][1];
      const float c = cov2D.m[1][1] + 0.3f;
      const float denom = a * c - b * b;
      float dL_da = 0, dL_db = 0, dL_dc = 0;
      const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
      if (denom2inv != 0) {
        dL_da = denom2inv * (-c * c * dcon_x + 2 * b * c * dcon_y + (denom - a * c) * dcon_w);
        dL_dc = denom2inv * (-a * a * dcon_w + 2 * a * b * dcon_y + (denom - a * c) * dcon_x);
        dL_db = denom2inv * 2 * (b * c * dcon_x - (denom + 2 * b * b) * dcon_y + a * b * dcon_w);
        a_cov[0] += (T.m[0][0] * T.m[0][0] * dL_da + T.m[0][0] * T.m[1][0] * dL_db + T.m[1][0] * T.m[1][0] * dL_dc);
        a_cov[3] += (T.m[0][1] * T.m[0][1] * dL_da + T.m[0][1] * T.m[1][1] * dL_db + T.m[1][1] * T.m[1][1] * dL_dc);
        a_cov[5] += (T.m[0][2] * T.m[0][2] * dL_da + T.m[0][2] * T.m[1][2] * dL_db + T.m[1][2] * T.m[1][2] * dL_dc);
        a_cov[1] += 2 * T.m[0][0] * T.m[0][1] * dL_da + (T.m[0][0] * T.m[1][1] + T.m[0][1] * T.m[1][0]) * dL_db +
                    2 * T.m[1][0] * T.m[1][1] * dL_dc;
        a_cov[2] += 2 * T.m[0][0] * T.m[0][2] * dL_da + (T.m[0][0] * T.m[1][2] + T.m[0][2] * T.m[1][0]) * dL_db +
                    2 * T.m[1][0] * T.m[1][2] * dL_dc;
        a_cov[4] += 2 * T.m[0][2] * T.m[0][1] * dL_da + (T.m[0][1] * T.m[1][2] + T.m[0][2] * T.m[1][1]) * dL_db +
                    2 * T.m[1][1] * T.m[1][2] * dL_dc;
      }
      const float dL_dT00 = 2 * (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_da +
                            (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_db;
      const float dL_dT01 = 2 * (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_da +
                            (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_db;
      const float dL_dT02 = 2 * (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_da +
                            (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_db;
      const float dL_dT10 = 2 * (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_dc +
                            (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_db;
      const float dL_dT11 = 2 * (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_dc +
                            (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_db;
      const float dL_dT12 = 2 * (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_dc +
                            (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_db;
      const float dL_dJ00 = Wm.m[0][0] * dL_dT00 + Wm.m[0][1] * dL_dT01 + Wm.m[0][2] * dL_dT02;
      const float dL_dJ02 = Wm.m[2][0] * dL_dT00 + Wm.m[2][1] * dL_dT01 + Wm.m[2][2] * dL_dT02;
      const float dL_dJ11 = Wm.m[1][0] * dL_dT10 + Wm.m[1][1] * dL_dT11 + Wm.m[1][2] * dL_dT12;
      const float dL_dJ12 = Wm.m[2][0] * dL_dT10 + Wm.m[2][1] * dL_dT11 + Wm.m[2][2] * dL_dT12;
      const float tz = 1.f / tz_;
      const float tz2 = tz * tz;
      const float tz3 = tz2 * tz;
      const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
      const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
      const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * tx) * tz3 * dL_dJ02 +
                           (2 * h_y * ty) * tz3 * dL_dJ12;
      // transformVec4x3Transpose (auxiliary.h:90-98)
      float dmean_x = V[0] * dL_dtx + V[1] * dL_dty + V[2] * dL_dtz;
      float dmean_y = V[4] * dL_dtx + V[5] * dL_dty + V[6] * dL_dtz;
      float dmean_z = V[8] * dL_dtx + V[9] * dL_dty + V[10] * dL_dtz;
      // view-matrix gradient through t = view * mean only (backward.cu:277-294) ...
      mat[0] = dL_dtx * mx;  mat[1] = dL_dty * mx;  mat[2] = dL_dtz * mx;
      mat[3] = dL_dtx * my;  mat[4] = dL_dty * my;  mat[5] = dL_dtz * my;
      mat[6] = dL_dtx * mz;  mat[7] = dL_dty * mz;  mat[8] = dL_dtz * mz;
      mat[9] = dL_dtx;       mat[10] = dL_dty;      mat[11] = dL_dtz;
      // ... and through depth (backward.cu:454-457): view[2], [6], [10], [14]
      mat[2] += ddepth * mx;
      mat[5] += ddepth * my;
      mat[8] += ddepth * mz;
      mat[11] += ddepth;

      // ---- preprocessCUDA backward (backward.cu:367-460)
      const float mhx = F[0] * mx + F[4] * my + F[8] * mz + F[12];
      const float mhy = F[1] * mx + F[5] * my + F[9] * mz + F[13];
      const float mhw = F[3] * mx + F[7] * my + F[11] * mz + F[15];
      const float m_w = 1.0f / (mhw + 0.0000001f);
      const float mul1 = mhx * m_w * m_w;
      const float mul2 = mhy * m_w * m_w;
      dmean_x += (F[0] * m_w - F[3] * mul1) * g2x + (F[1] * m_w - F[3] * mul2) * g2y + ddepth * V[2];
      dmean_y += (F[4] * m_w - F[7] * mul1) * g2x + (F[5] * m_w - F[7] * mul2) * g2y + ddepth * V[6];
      dmean_z += (F[8] * m_w - F[11] * mul1) * g2x + (F[9] * m_w - F[11] * mul2) * g2y + ddepth * V[10];

      if (!SPLIT && shs != nullptr) {  // computeColorFromSH backward (backward.cu:20-140)
        // the activation's derivative input: the pre-activation values (sigmoid) or the clamp mask (relu; from the row)
        float ps3[3];
        if (v.use_sigmoid) {
          ps3[0] = pre_sigmoid[3 * o]; ps3[1] = pre_sigmoid[3 * o + 1]; ps3[2] = pre_sigmoid[3 * o + 2];
        } else {
          ps3[0] = (cur.cmask & 1u) ? 1.0f : 0.0f; ps3[1] = (cur.cmask & 2u) ? 1.0f : 0.0f; ps3[2] = (cur.cmask & 4u) ? 1.0f : 0.0f;
        }
        DGS_SH_BACKWARD(ps3)
      }
      a_col[0] += dcol[0];
      a_col[1] += dcol[1];
      a_col[2] += dcol[2];
      a_mean[0] += dmean_x;
      a_mean[1] += dmean_y;
      a_mean[2] += dmean_z;

      // projection-matrix gradient exactly as the reference computes it (backward.cu:423-450): double
      // arithmetic rounded to float per contribution; entries 3, 7, 11, 15 all receive -0.5*lastcol.
      const float lastcol = (mhx * v.W * g2x + mhy * v.H * g2y) * m_w * m_w;
      mat[12] = (float)(0.5 * g2x * mx * v.W * m_w);
      mat[13] = (float)(0.5 * g2y * mx * v.H * m_w);
      mat[14] = (float)(0.5 * g2x * my * v.W * m_w);
      mat[15] = (float)(0.5 * g2y * my * v.H * m_w);
      mat[16] = (float)(0.5 * g2x * mz * v.W * m_w);
      mat[17] = (float)(0.5 * g2y * mz * v.H * m_w);
      mat[18] = (float)(0.5 * g2x * v.W * m_w);
      mat[19] = (float)(0.5 * g2y * v.H * m_w);
      mat[20] = (float)(-0.5 * lastcol);
    }
    if (valid && dL_dmeans2D != nullptr) {
      float* d2 = dL_dmeans2D + 3 * o;
      d2[0] = g2x;
      d2[1] = g2y;
      d2[2] = 0.0f;
    }
    if (stats && cur.rad > 0) {   // dgs_densify_stats' update of this subframe, same operations in the same order
      st_mr = fmaxf(st_mr, (float)cur.rad);
      st_ac += sqrtf(g2x * g2x + g2y * g2y);
      st_dn += st_inc;
    }
    // ---- per-subframe pose gradients: wave sum (skipped when no lane of the wave is visible in k)
    // through LDS, like the compositing backward's per-entry reduction: 21 conflict-free 4-byte stores per lane, then
    // lane (row, quarter) adds 16 values of one sum and a quad sum finishes -- 34 VALU adds instead of 21 six-step DPP
    // butterflies (126 DPP adds, the most expensive plain class)
    {
      float* sp = s_part + ((size_t)w * v.K + k) * NMAT;
      if (__ballot(ntiles > 0) != 0ull) {
        float* pw = &s_m[w][0][lane];
#pragma unroll
        for (int i = 0; i < 21; i++) pw[i * 68] = mat[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int rrow = lane >> 2, rq = lane & 3;
#pragma unroll
        for (int half = 0; half < 2; half++) {
          const int row = rrow + 16 * half;
          float tot = 0.0f;
          if (row < 21) {
            const float4* pr = reinterpret_cast<const float4*>(&s_m[w][row][16 * rq]);
            const float4 x0 = pr[0], x1 = pr[1], x2 = pr[2], x3 = pr[3];
            tot = (((x0.x + x0.y) + (x0.z + x0.w)) + ((x1.x + x1.y) + (x1.z + x1.w))) +
                  (((x2.x + x2.y) + (x2.z + x2.w)) + ((x3.x + x3.y) + (x3.z + x3.w)));
          }
          tot = dgs_quad_sum(tot);
          if (row < 21 && rq == 0) sp[row] = tot;
        }
        __builtin_amdgcn_wave_barrier();
      } else if (lane < 21) {
        sp[lane] = 0.0f;
      }
    }
  }

  if (SPLIT && shs != nullptr) {
    // second loop over the subframes: the SH part alone (see SPLIT above).  The loads of subframe k + 1 are issued
    // before subframe k computes.
    DGS_LOAD_SH()
    struct ShPf {
      uint32_t nt;
      float4 r1, r2;
      float ps[3];
    };
    auto load_pf = [&](int k) {
      ShPf f;
      const size_t o = (size_t)k * v.P + gi;
      f.nt = valid ? tiles_touched[o] : 0u;
      const float4* cp = reinterpret_cast<const float4*>(contrib + o * DGS_SUMS_F);
      f.r1 = cp[1];
      f.r2 = cp[2];
      if (v.use_sigmoid) {
        f.ps[0] = pre_sigmoid[3 * o];
        f.ps[1] = pre_sigmoid[3 * o + 1];
        f.ps[2] = pre_sigmoid[3 * o + 2];
      } else {   // relu: the clamp mask rides in the row's spare word
        const uint32_t cm = rows[o].dup_offset;
        f.ps[0] = (cm & 1u) ? 1.0f : 0.0f;
        f.ps[1] = (cm & 2u) ? 1.0f : 0.0f;
        f.ps[2] = (cm & 4u) ? 1.0f : 0.0f;
      }
      return f;
    };
#ifndef DGS_SH_PREFETCH
#define DGS_SH_PREFETCH 0
#endif
    ShPf nxt = load_pf(0);
    for (int k = 0; k < v.K; k++) {
#if DGS_SH_PREFETCH
      const ShPf cur = nxt;
      if (k + 1 < v.K) nxt = load_pf(k + 1);
#else
      const ShPf cur = load_pf(k);
#endif
      if (cur.nt > 0) {
        const float dcol[3] = {cur.r1.z, cur.r1.w, cur.r2.x};
        float dmean_x = 0.0f, dmean_y = 0.0f, dmean_z = 0.0f;
        DGS_SH_BACKWARD(cur.ps)
        a_mean[0] += dmean_x;
        a_mean[1] += dmean_y;
        a_mean[2] += dmean_z;
      }
    }
  }

  if (valid && stats) {
    st_max_radii[gi] = st_mr;
    st_accum[gi] = st_ac;
    st_denom[gi] = st_dn;
  }
  if (valid) {
    dL_dmeans3D[3 * idx + 0] = a_mean[0];
    dL_dmeans3D[3 * idx + 1] = a_mean[1];
    dL_dmeans3D[3 * idx + 2] = a_mean[2];
    // raw parameters: torch.clamp passes the gradient where 0 <= x <= 1
    float g_op = (v.raw_params && !(opacities_raw[idx] >= 0.0f && opacities_raw[idx] <= 1.0f)) ? 0.0f : a_op;
    if (v.raw_params && hinge_scale != 0.0f) {
      // the opacity hinge of the training loss (utils/loss_utils.py:96-104: mean(x^2 [x <= 0] + (x - 1)^2 [x >= 1]),
      // train.py:156-163) differentiated here: hinge_scale = lambda_hinge * upstream / numel
      const float x = opacities_raw[idx];
      g_op += hinge_scale * ((x <= 0.0f) ? 2.0f * x : ((x >= 1.0f) ? 2.0f * (x - 1.0f) : 0.0f));
    }
    dL_dopacity[idx] = g_op;
    dL_dcolors[3 * idx + 0] = a_col[0];
    dL_dcolors[3 * idx + 1] = a_col[1];
    dL_dcolors[3 * idx + 2] = a_col[2];
#pragma unroll
    for (int i = 0; i < 6; i++) dL_dcov3D_out[6 * (size_t)idx + i] = a_cov[i];
    if (dL_dsh != nullptr && dL_dsh_rest != nullptr) {  // raw parameters: dc and rest gradients in two tensors
      float* ddc = dL_dsh + (size_t)idx * 3;
      float* drs = dL_dsh_rest + (size_t)idx * (v.M - 1) * 3;
      ddc[0] = a_sh[0];
      ddc[1] = a_sh[1];
      ddc[2] = a_sh[2];
#pragma unroll
      for (int jj = 1; jj < MAXC; jj++)
        if (jj < v.M) {
          drs[3 * (jj - 1)] = a_sh[3 * jj];
          drs[3 * (jj - 1) + 1] = a_sh[3 * jj + 1];
          drs[3 * (jj - 1) + 2] = a_sh[3 * jj + 2];
        }
      for (int j = MAXC; j < v.M; j++) {
        drs[3 * (j - 1)] = 0.0f;
        drs[3 * (j - 1) + 1] = 0.0f;
        drs[3 * (j - 1) + 2] = 0.0f;
      }
    } else if (dL_dsh != nullptr) {
      float* dsh = dL_dsh + (size_t)idx * v.M * 3;
#pragma unroll
      for (int jj = 0; jj < MAXC; jj++)
        if (jj < v.M) {
          dsh[3 * jj] = a_sh[3 * jj];
          dsh[3 * jj + 1] = a_sh[3 * jj + 1];
          dsh[3 * jj + 2] = a_sh[3 * jj + 2];
        }
      for (int j = MAXC; j < v.M; j++) {  // coefficients above the active degree get zero gradient
        dsh[3 * j] = 0.0f;
        dsh[3 * j + 1] = 0.0f;
        dsh[3 * j + 2] = 0.0f;
      }
    }
    if (scales != nullptr) {  // computeCov3D backward (backward.cu:299-362) on the K-summed dL_dcov3D
      float r = rotations[4 * idx], x = rotations[4 * idx + 1], y = rotations[4 * idx + 2], z = rotations[4 * idx + 3];
      float s0 = scales[3 * idx], s1 = scales[3 * idx + 1], s2 = scales[3 * idx + 2];
      float qd = 1.0f, e0 = 1.0f, e1 = 1.0f, e2 = 1.0f;
      if (v.raw_params) {
        if (v.iso_scale) s1 = s2 = s0;
        e0 = expf(s0); e1 = expf(s1); e2 = expf(s2);   // d(exp(x) + lb)/dx
        s0 = dgs_act_scale(s0, v.scale_lb);
        s1 = dgs_act_scale(s1, v.scale_lb);
        s2 = dgs_act_scale(s2, v.scale_lb);
        qd = dgs_quat_norm(r, x, y, z);
        r = r / qd; x = x / qd; y = y / qd; z = z / qd;
      }
      M3 R = {{{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
               {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
               {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}}};
      const float sx = v.scale_modifier * s0, sy = v.scale_modifier * s1, sz = v.scale_modifier * s2;
      M3 S = {{{sx, 0, 0}, {0, sy, 0}, {0, 0, sz}}};
      M3 Mm = mul(S, R);
      M3 dSig = {{{a_cov[0], 0.5f * a_cov[1], 0.5f * a_cov[2]},
                  {0.5f * a_cov[1], a_cov[3], 0.5f * a_cov[4]},
                  {0.5f * a_cov[2], 0.5f * a_cov[4], a_cov[5]}}};
      M3 M2;
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
        for (int rr = 0; rr < 3; rr++) M2.m[c][rr] = Mm.m[c][rr] * 2.0f;
      M3 dL_dM = mul(M2, dSig);
      M3 Rt = tr(R);
      M3 dMt = tr(dL_dM);
      float ds0 = (Rt.m[0][0] * dMt.m[0][0] + Rt.m[0][1] * dMt.m[0][1] + Rt.m[0][2] * dMt.m[0][2]) * e0;
      float ds1 = (Rt.m[1][0] * dMt.m[1][0] + Rt.m[1][1] * dMt.m[1][1] + Rt.m[1][2] * dMt.m[1][2]) * e1;
      float ds2 = (Rt.m[2][0] * dMt.m[2][0] + Rt.m[2][1] * dMt.m[2][1] + Rt.m[2][2] * dMt.m[2][2]) * e2;
      if (v.iso_scale) {  // expand(-1, 3) backward: the three columns' gradients land on column 0
        ds0 = (ds0 + ds1) + ds2;
        ds1 = 0.0f;
        ds2 = 0.0f;
      }
      dL_dscales[3 * idx + 0] = ds0;
      dL_dscales[3 * idx + 1] = ds1;
      dL_dscales[3 * idx + 2] = ds2;
#pragma unroll
      for (int rr = 0; rr < 3; rr++) {
        dMt.m[0][rr] *= sx;
        dMt.m[1][rr] *= sy;
        dMt.m[2][rr] *= sz;
      }
      float4 dq;
      dq.x = 2 * z * (dMt.m[0][1] - dMt.m[1][0]) + 2 * y * (dMt.m[2][0] - dMt.m[0][2]) +
             2 * x * (dMt.m[1][2] - dMt.m[2][1]);
      dq.y = 2 * y * (dMt.m[1][0] + dMt.m[0][1]) + 2 * z * (dMt.m[2][0] + dMt.m[0][2]) +
             2 * r * (dMt.m[1][2] - dMt.m[2][1]) - 4 * x * (dMt.m[2][2] + dMt.m[1][1]);
      dq.z = 2 * x * (dMt.m[1][0] + dMt.m[0][1]) + 2 * r * (dMt.m[2][0] - dMt.m[0][2]) +
             2 * z * (dMt.m[1][2] + dMt.m[2][1]) - 4 * y * (dMt.m[2][2] + dMt.m[0][0]);
      dq.w = 2 * r * (dMt.m[0][1] - dMt.m[1][0]) + 2 * x * (dMt.m[2][0] + dMt.m[0][2]) +
             2 * y * (dMt.m[1][2] + dMt.m[2][1]) - 4 * z * (dMt.m[1][1] + dMt.m[0][0]);
      if (v.raw_params) {
        // x / max(|x|, eps) backward: (g - n (n . g)) / |x|  (n = normalised quaternion; constant denominator if
        // the clamp is active)
        const float4 raw = reinterpret_cast<const float4*>(rotations)[idx];
        const bool clamped = sqrtf(raw.x * raw.x + raw.y * raw.y + raw.z * raw.z + raw.w * raw.w) < 1e-12f;
        const float dot = clamped ? 0.0f : (r * dq.x + x * dq.y + y * dq.z + z * dq.w);
        dq.x = (dq.x - r * dot) / qd;
        dq.y = (dq.y - x * dot) / qd;
        dq.z = (dq.z - y * dot) / qd;
        dq.w = (dq.w - z * dot) / qd;
      }
      reinterpret_cast<float4*>(dL_drots)[idx] = dq;
    }
  }

  // ---- combine the 4 waves in wave order and publish this block's partial pose gradients.  Beyond the wave sums
  // everything is added in double: dL_dview is a sum of ~10^6 signed terms that cancel to a small total, and fp32
  // partial sums of growing magnitude would put an error of several 1e-4 of the result on it (the reference's float
  // atomics do; the parity tests compare with a double-accumulating oracle)
  __syncthreads();
  const int total = v.K * NMAT;
  for (int i = threadIdx.x; i < total; i += GB_THREADS) {
    double acc = 0.0;
#pragma unroll
    for (int ww = 0; ww < GB_THREADS / 64; ww++) acc += (double)s_part[(size_t)ww * total + i];
    partials[((size_t)(g_begin / GB_THREADS) + blockIdx.x) * total + i] = acc;
  }
}

// Sums the per-block partials and scatters them into the two [K,4,4] outputs.  Deterministic: thread t adds
// blocks t, t+256, ... in order for all 21 values at once (one 84-byte row per block), then a fixed-shape tree
// over the 256 threads combines them.
__global__ void __launch_bounds__(256)
pose_grad_reduce_kernel(int K, int nblocks, const double* __restrict__ partials, float* __restrict__ dL_dview,
                        float* __restrict__ dL_dproj) {
  __shared__ double red[256][21 + 1];
  const int k = blockIdx.x;
  const size_t stride = (size_t)K * NMAT;
  double acc[21];
#pragma unroll
  for (int i = 0; i < 21; i++) acc[i] = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) {
    const double* p = partials + (size_t)b * stride + (size_t)k * NMAT;
#pragma unroll
    for (int i = 0; i < 21; i++) acc[i] += p[i];
  }
#pragma unroll
  for (int i = 0; i < 21; i++) red[threadIdx.x][i] = acc[i];
  __syncthreads();
  for (int h = 128; h >= 1; h >>= 1) {
    if ((int)threadIdx.x < h) {
#pragma unroll
      for (int i = 0; i < 21; i++) red[threadIdx.x][i] += red[threadIdx.x + h][i];
    }
    __syncthreads();
  }
  if (threadIdx.x < 32) {
    const int i = threadIdx.x;  // output slot: 0..15 view entry i, 16..31 proj entry i-16
    int src = -1;
    if (i < 16) {
      const int r = i >> 2, c = i & 3;
      if (c < 3) src = r * 3 + c;  // view[4r+c] <- mat[3r+c]
    } else {
      const int e = i - 16, r = e >> 2, c = e & 3;
      if (c < 2) src = 12 + r * 2 + c;  // proj[4r+c], c in {0,1}
      if (c == 3) src = 20;             // proj[3], [7], [11], [15]
    }
    const float v = src >= 0 ? (float)red[0][src] : 0.0f;
    if (i < 16)
      dL_dview[16 * k + i] = v;
    else
      dL_dproj[16 * k + (i - 16)] = v;
  }
}

}  // namespace

int dgs_geometry_bwd_blocks(int P) { return (P + GB_THREADS - 1) / GB_THREADS; }

// phases: 1 = per-pair totals of the contribution rows, 2 = the per-Gaussian kernel for Gaussians [g_begin, g_end),
// 4 = the final sum of the pose-gradient partials of ALL blocks (after every chunk has run)
hipError_t dgs_launch_geometry_bwd(const DgsProblem& p, const DgsView& v, const DgsCarve& c, const DgsBackwardIO& io,
                                   const float* contrib, float* sums, double* partials, hipStream_t s, int phases,
                                   int g_begin, int g_end, int k0, int k1) {
  const int all_blocks = dgs_geometry_bwd_blocks(v.P);
  const size_t lds = (size_t)(GB_THREADS / 64) * v.K * NMAT * sizeof(float);
  const int ncoef = (p.shs != nullptr) ? (v.D + 1) * (v.D + 1) : 1;
  const uint64_t kp = (uint64_t)v.K * v.P;
  // walk the (k, Gaussian) pairs in (k, depth, index) order: their row segments are then consecutive in memory
  // (that is the order the duplicates were laid out in)
  // (the depth order, tt_sorted / offs_sorted exist whenever a duplicate or -- with tile culling -- a visible pair does;
  // otherwise no pair is visible and the geometry kernel reads no total)
  // (phase 1 for subframes [k0, k1) only, k1 < 0 = all: the pairs of a subframe are one segment of the depth order, their
  // totals go to absolute natural indices)
  if ((phases & 1) && (io.num_rendered > 0 || v.tile_cull)) {
    if (k1 < 0) { k0 = 0; k1 = v.K; }
    const uint64_t j0 = (uint64_t)k0 * v.P, nj = (uint64_t)(k1 - k0) * v.P;
    if (nj > 0)
      hipLaunchKernelGGL(contrib_reduce_kernel, dim3((uint32_t)((4 * nj + 255) / 256)), dim3(256), 0, s, nj, c.num_rendered,
                         c.gsort_vals + j0, c.tt_sorted + j0, (v.tile_cull ? c.tt_tight : c.tt_sorted) + j0,
                         (v.tile_cull ? c.offs_tight : c.offs_sorted) + j0, contrib, sums);
  }
  const int blocks = (g_end - g_begin + GB_THREADS - 1) / GB_THREADS;
#define DGS_GB_LAUNCH(MAXC)                                                                                          \
  hipLaunchKernelGGL(geometry_bwd_kernel<MAXC>, dim3(blocks), dim3(GB_THREADS), lds, s, v, p.means3D, p.shs,         \
                     p.shs_rest, p.opacities,                                                                        \
                     p.scales, p.rotations, p.cov3D_precomp, p.viewmatrix, p.projmatrix, p.campos, c.rows,       \
                     c.cov3D,  \
                     c.pre_sigmoid, c.tiles_touched, sums, c.num_rendered, io.opacity_hinge_scale, io.dL_dmeans3D,  \
                     io.dL_dmeans2D, io.dL_dsh,                                                                       \
                     io.dL_dsh_rest,                                                                                 \
                     io.dL_dcolors, io.dL_dopacity, io.dL_dscales, io.dL_drotations, io.dL_dcov3D, partials, g_begin, \
                     g_end, io.radii, io.stats_max_radii2D, io.stats_grad_accum, io.stats_denom,                      \
                     (float)(1.0 / (double)(io.stats_K_total > 0 ? io.stats_K_total : v.K)))
  if ((phases & 2) && blocks > 0) {
    if (ncoef <= 1)
      DGS_GB_LAUNCH(1);
    else if (ncoef <= 4)
      DGS_GB_LAUNCH(4);
    else if (ncoef <= 9)
      DGS_GB_LAUNCH(9);
    else
      DGS_GB_LAUNCH(16);
  }
#undef DGS_GB_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (phases & 4)
    hipLaunchKernelGGL(pose_grad_reduce_kernel, dim3(v.K), dim3(256), 0, s, v.K, all_blocks, partials, io.dL_dviewmatrix,
                       io.dL_dprojmatrix);
  return hipGetLastError();
}
