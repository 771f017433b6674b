// preprocess.hip -- K-fused forward preprocess (replaces FORWARD::preprocess, forward.cu:166-268, and
// checkFrustum, rasterizer_impl.cu:54-66).
//
// One thread per Gaussian reads xyz / scale / rotation / opacity / SH ONCE and emits the K per-subframe
// geometry rows (the reference re-reads all of it K times, once per render() call).  cov3D is
// view-independent and computed once.
//
// BUILD WITH -ffp-contract=off: radii, tile rectangles, tiles_touched and depth bits must be bit-identical to
// the CPU oracle, so every expression below keeps the operation order of the reference source (GLM
// column-major mat3 products, left-to-right sums) and relies on IEEE fp32 add/mul/div/sqrt.
#include "dgs_common.h"

namespace {

__device__ const float SH_C0 = 0.28209479177387814f;
__device__ const float SH_C1 = 0.4886025119029199f;
__device__ const float SH_C2[] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                  -1.0925484305920792f, 0.5462742152960396f};
__device__ const float SH_C3[] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                  0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                  -0.5900435899266435f};

struct M3 {  // column-major like glm::mat3: m[col][row]
  float m[3][3];
};
__device__ __forceinline__ M3 mul(const M3& A, const M3& B) {  // glm operator*(mat3, mat3)
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

__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int DEG>  // active SH degree: the (DEG+1)^2 coefficients are read ONCE per Gaussian and kept in registers
__global__ void __launch_bounds__(256)
preprocess_fwd_kernel(DgsView v, const float* __restrict__ means3D, const float* __restrict__ scales,
                      const float* __restrict__ rotations, const float* __restrict__ opacities,
                      const float* __restrict__ shs, const float* __restrict__ shs_rest,
                      const float* __restrict__ cov3D_precomp,
                      const float* __restrict__ colors_precomp, const float* __restrict__ viewm,
                      const float* __restrict__ projm, const float* __restrict__ campos, DgsRow* __restrict__ rows,
                      float* __restrict__ cov3Ds, float* __restrict__ pre_sigmoid,
                      uint32_t* __restrict__ tiles_touched, int32_t* __restrict__ radii,
                      uint32_t* __restrict__ dkeys) {
  // Row and colour-mask stores are transposed through LDS per wave: a lane's 48-byte row becomes three
  // wave-wide 1 KB stores (64 consecutive float4) instead of three 16-byte stores at a 48-byte lane stride.
  __shared__ float4 s_row[4][3 * 64];
  __shared__ float s_pre[4][3 * 64];
  const int lane = dgs_lane(), wv_ = threadIdx.x >> 6;
  const int idx_raw = blockIdx.x * 256 + threadIdx.x;
  const bool live = idx_raw < v.P;
  const int idx = live ? idx_raw : v.P - 1;   // out-of-range lanes shadow the last Gaussian and store nothing
  const int wave_first = blockIdx.x * 256 + wv_ * 64;          // first Gaussian of this wave
  const int wave_count = min(64, v.P - wave_first);           // > 0 for every launched wave that has live lanes
  const float px = means3D[3 * idx], py = means3D[3 * idx + 1], pz = means3D[3 * idx + 2];
  const float opacity = v.raw_params ? dgs_act_opacity(opacities[idx]) : opacities[idx];

  float c3[6];
  if (cov3D_precomp != nullptr) {
#pragma unroll
    for (int i = 0; i < 6; i++) c3[i] = cov3D_precomp[6 * (size_t)idx + i];
  } else {  // computeCov3D, forward.cu:129-163 (quaternion as given)
    const float mod = v.scale_modifier;
    M3 S = {{{1.0f, 0.f, 0.f}, {0.f, 1.0f, 0.f}, {0.f, 0.f, 1.0f}}};
    float s0 = scales[3 * idx], s1 = scales[3 * idx + 1], s2 = scales[3 * idx + 2];
    float r = rotations[4 * idx], x = rotations[4 * idx + 1], y = rotations[4 * idx + 2], z = rotations[4 * idx + 3];
    if (v.raw_params) {
      if (v.iso_scale) s1 = s2 = s0;   // get_scaling of an isotropic cloud (scene/gaussian_model.py:115-118)
      s0 = dgs_act_scale(s0, v.scale_lb);
      s1 = dgs_act_scale(s1, v.scale_lb);
      s2 = dgs_act_scale(s2, v.scale_lb);
      const float d = dgs_quat_norm(r, x, y, z);
      r = r / d; x = x / d; y = y / d; z = z / d;
    }
    S.m[0][0] = mod * s0;
    S.m[1][1] = mod * s1;
    S.m[2][2] = mod * s2;
    M3 R = {{{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
             {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
             {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}}};
    M3 Mm = mul(S, R);
    M3 Sigma = mul(tr(Mm), Mm);
    c3[0] = Sigma.m[0][0];
    c3[1] = Sigma.m[0][1];
    c3[2] = Sigma.m[0][2];
    c3[3] = Sigma.m[1][1];
    c3[4] = Sigma.m[1][2];
    c3[5] = Sigma.m[2][2];
    if (live && cov3Ds != nullptr) {   // (NULL: an inference call keeps nothing for a backward)
#pragma unroll
      for (int i = 0; i < 6; i++) cov3Ds[6 * (size_t)idx + i] = c3[i];
    }
  }
  if (wave_count <= 0) return;  // wave-uniform
  constexpr int NC = (DEG + 1) * (DEG + 1);
  float sh[NC * 3];
  if (colors_precomp == nullptr) {
    if (shs_rest == nullptr) {
      const float* shp = shs + (size_t)idx * v.M * 3;
#pragma unroll
      for (int i = 0; i < NC * 3; i++) sh[i] = shp[i];
    } else {  // raw parameters: dc and rest live in two tensors (GaussianModel._features_dc / _features_rest)
      const float* dcp = shs + (size_t)idx * 3;
      const float* rsp = shs_rest + (size_t)idx * (v.M - 1) * 3;
#pragma unroll
      for (int i = 0; i < NC * 3; i++) sh[i] = (i < 3) ? dcp[i] : rsp[i - 3];
    }
  }

  for (int k = 0; k < v.K; k++) {
    const float* V = viewm + 16 * k;
    const float* F = projm + 16 * k;
    const size_t o = (size_t)k * v.P + idx;
    int out_radius = 0;
    uint32_t out_tiles = 0;
    float pre_out[3] = {0.0f, 0.0f, 0.0f};
    DgsRow row;
    row.x = row.y = row.cx = row.cy = row.cz = row.op = row.r = row.g = row.b = row.depth = 0.0f;
    row.dup_offset = 0;
    row.radius = 0;
    // in_frustum (auxiliary.h:144-169): near-plane cull only
    const float vz = V[2] * px + V[6] * py + V[10] * pz + V[14];
    if (vz > 0.2f) {
      const float hx = F[0] * px + F[4] * py + F[8] * pz + F[12];
      const float hy = F[1] * px + F[5] * py + F[9] * pz + F[13];
      const float hw = F[3] * px + F[7] * py + F[11] * pz + F[15];
      const float p_w = 1.0f / (hw + 0.0000001f);
      const float projx = hx * p_w, projy = hy * p_w;
      // computeCov2D, forward.cu:85-124
      float tx = V[0] * px + V[4] * py + V[8] * pz + V[12];
      float ty = V[1] * px + V[5] * py + V[9] * pz + V[13];
      const float tz = vz;
      const float limx = 1.3f * v.tanfovx;
      const float limy = 1.3f * v.tanfovy;
      const float txtz = tx / tz;
      const float tytz = ty / tz;
      tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
      ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
      M3 J = {{{v.focal_x / tz, 0.0f, -(v.focal_x * tx) / (tz * tz)},
               {0.0f, v.focal_y / tz, -(v.focal_y * ty) / (tz * tz)},
               {0.f, 0.f, 0.f}}};
      M3 Wm = {{{V[0], V[4], V[8]}, {V[1], V[5], V[9]}, {V[2], V[6], V[10]}}};
      M3 T = mul(Wm, J);
      M3 Vrk = {{{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}}};
      M3 cov = mul(mul(tr(T), tr(Vrk)), T);
      const float ca = cov.m[0][0] + 0.3f;
      const float cb = cov.m[0][1];
      const float cc = cov.m[1][1] + 0.3f;
      const float det = (ca * cc - cb * cb);
      if (det != 0.0f) {
        const float det_inv = 1.f / det;
        const float mid = 0.5f * (ca + cc);
        const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        const float pixx = ndc2pix(projx, v.W), pixy = ndc2pix(projy, v.H);
        int minx, miny, maxx, maxy;
        dgs_get_rect(pixx, pixy, (int)my_radius, v.gx, v.gy, minx, miny, maxx, maxy);
        const uint32_t area = (uint32_t)(maxx - minx) * (uint32_t)(maxy - miny);
        if (area != 0) {
          float cr, cg, cbl;
          if (colors_precomp != nullptr) {
            cr = colors_precomp[3 * (size_t)idx];
            cg = colors_precomp[3 * (size_t)idx + 1];
            cbl = colors_precomp[3 * (size_t)idx + 2];
          } else {  // computeColorFromSH, forward.cu:20-82
            const float* cam = campos + 3 * k;
            float dx = px - cam[0], dy = py - cam[1], dz = pz - cam[2];
            const float tx2 = dx * dx, ty2 = dy * dy, tz2 = dz * dz;
            const float len = sqrtf(tx2 + ty2 + tz2);
            dx = dx / len;
            dy = dy / len;
            dz = dz / len;
            float res[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
              float r_ = SH_C0 * sh[ch];
              if (DEG > 0) {
                const float x = dx, y = dy, z = dz;
                r_ = r_ - SH_C1 * y * sh[3 + ch] + SH_C1 * z * sh[6 + ch] - SH_C1 * x * sh[9 + ch];
                if (DEG > 1) {
                  const float xx = x * x, yy = y * y, zz = z * z;
                  const float xy = x * y, yz = y * z, xz = x * z;
                  r_ = r_ + SH_C2[0] * xy * sh[12 + ch] + SH_C2[1] * yz * sh[15 + ch] +
                       SH_C2[2] * (2.0f * zz - xx - yy) * sh[18 + ch] + SH_C2[3] * xz * sh[21 + ch] +
                       SH_C2[4] * (xx - yy) * sh[24 + ch];
                  if (DEG > 2) {
                    r_ = r_ + SH_C3[0] * y * (3.0f * xx - yy) * sh[27 + ch] + SH_C3[1] * xy * z * sh[30 + ch] +
                         SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[33 + ch] +
                         SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + ch] +
                         SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[39 + ch] + SH_C3[5] * z * (xx - yy) * sh[42 + ch] +
                         SH_C3[6] * x * (xx - 3.0f * yy) * sh[45 + ch];
                  }
                }
              }
              float pre;
              if (v.use_sigmoid) {
                pre = r_;
                r_ = sigmoidf_(r_);
              } else {
                r_ += 0.5f;
                pre = (r_ >= 0.0f) ? 1.0f : 0.0f;
                r_ = fmaxf(r_, 0.0f);
              }
              pre_out[ch] = pre;
              res[ch] = r_;
            }
            cr = res[0];
            cg = res[1];
            cbl = res[2];
          }
          row.x = pixx;
          row.y = pixy;
          row.cx = cc * det_inv;
          row.cy = -cb * det_inv;
          row.cz = ca * det_inv;
          row.op = opacity;
          row.r = cr;
          row.g = cg;
          row.b = cbl;
          row.depth = vz;
          // relu colour activation: the three clamp-mask bits the backward needs ride in the row's spare word (round 6:
          // the [K,P,3] float mask array -- 180 MB written here and read back by geometry_bwd at the metric size -- is then
          // neither written nor read); the sigmoid activation keeps its pre-activation values in pre_sigmoid
          row.dup_offset = v.use_sigmoid ? 0u
                                         : ((pre_out[0] != 0.0f ? 1u : 0u) | (pre_out[1] != 0.0f ? 2u : 0u) |
                                            (pre_out[2] != 0.0f ? 4u : 0u));
          row.radius = (int)my_radius;
          out_radius = (int)my_radius;
          out_tiles = area;
        }
      }
    } else if (v.prefiltered) {
      // auxiliary.h:161-165: the reference traps when a "prefiltered" point is culled
      __builtin_trap();
    }
    if (live) {
      radii[o] = out_radius;
      tiles_touched[o] = out_tiles;
      // sort key of the (k, Gaussian) pair for the depth-ordered duplication (binning.hip): invisible pairs sort
      // to the end of their subframe and emit nothing
      dkeys[o] = out_tiles ? __float_as_uint(row.depth) - DGS_DEPTH_KEY_BASE : 0xFFFFFFFFu;
    }
    // wave-transposed stores of the 64 rows / colour masks of this (wave, k); invisible pairs store zeros
    {
      const float4* src = reinterpret_cast<const float4*>(&row);
      s_row[wv_][3 * lane + 0] = src[0];
      s_row[wv_][3 * lane + 1] = src[1];
      s_row[wv_][3 * lane + 2] = src[2];
      s_pre[wv_][3 * lane + 0] = pre_out[0];
      s_pre[wv_][3 * lane + 1] = pre_out[1];
      s_pre[wv_][3 * lane + 2] = pre_out[2];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const size_t wbase = (size_t)k * v.P + wave_first;
      float4* dst = reinterpret_cast<float4*>(rows + wbase);
      float* dpre = pre_sigmoid + 3 * wbase;
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const int e = i * 64 + lane;
        if (e < 3 * wave_count) {
          dst[e] = s_row[wv_][e];
          if (pre_sigmoid != nullptr && v.use_sigmoid) dpre[e] = s_pre[wv_][e];
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

__global__ void mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ V,
                                    uint8_t* __restrict__ present) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  const float px = means3D[3 * idx], py = means3D[3 * idx + 1], pz = means3D[3 * idx + 2];
  present[idx] = (V[2] * px + V[6] * py + V[10] * pz + V[14]) > 0.2f;
}

}  // namespace

hipError_t dgs_launch_preprocess(const DgsProblem& p, const DgsView& v, const DgsCarve& c, int32_t* radii,
                                 hipStream_t s) {
  const int blocks = (v.P + 255) / 256;
#define DGS_PRE(DEG)                                                                                                  \
  hipLaunchKernelGGL(preprocess_fwd_kernel<DEG>, dim3(blocks), dim3(256), 0, s, v, p.means3D, p.scales, p.rotations, \
                     p.opacities, p.shs, p.shs_rest, p.cov3D_precomp, p.colors_precomp, p.viewmatrix, p.projmatrix, p.campos,     \
                     c.rows, c.cov3D, c.pre_sigmoid, c.tiles_touched, radii, c.gsort_keys)
  const int deg = (p.colors_precomp != nullptr) ? 0 : v.D;
  if (deg <= 0)
    DGS_PRE(0);
  else if (deg == 1)
    DGS_PRE(1);
  else if (deg == 2)
    DGS_PRE(2);
  else
    DGS_PRE(3);
#undef DGS_PRE
  return hipGetLastError();
}

// The activated values exactly as preprocess_fwd_kernel forms them in raw-parameter mode (same device functions, same
// translation unit and compile flags, hence the same bits): the device-side counterpart of the reference's getters
// get_scaling / get_rotation / get_opacity (scene/gaussian_model.py:114-137).
__global__ void __launch_bounds__(256)
cloud_activations_kernel(int P, const float* __restrict__ scales, const float* __restrict__ rotations,
                         const float* __restrict__ opacities, float scale_lb, float* __restrict__ out_scales,
                         float* __restrict__ out_rotations, float* __restrict__ out_opacities) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  if (out_scales != nullptr) {
#pragma unroll
    for (int i = 0; i < 3; i++) out_scales[3 * idx + i] = dgs_act_scale(scales[3 * idx + i], scale_lb);
  }
  if (out_rotations != nullptr) {
    const float r = rotations[4 * idx], x = rotations[4 * idx + 1], y = rotations[4 * idx + 2], z = rotations[4 * idx + 3];
    const float d = dgs_quat_norm(r, x, y, z);
    out_rotations[4 * idx] = r / d;
    out_rotations[4 * idx + 1] = x / d;
    out_rotations[4 * idx + 2] = y / d;
    out_rotations[4 * idx + 3] = z / d;
  }
  if (out_opacities != nullptr) out_opacities[idx] = dgs_act_opacity(opacities[idx]);
}

hipError_t dgs_launch_cloud_activations(int P, const float* scales, const float* rotations, const float* opacities,
                                        float scale_lb, float* out_scales, float* out_rotations, float* out_opacities,
                                        hipStream_t s) {
  hipLaunchKernelGGL(cloud_activations_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, scales, rotations, opacities,
                     scale_lb, out_scales, out_rotations, out_opacities);
  return hipGetLastError();
}

hipError_t dgs_launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s) {
  hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, view, present);
  return hipGetLastError();
}
