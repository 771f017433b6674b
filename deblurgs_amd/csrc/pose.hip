// pose.hip -- the pose path of the blur-integration loop as ONE kernel per direction:
//   Bezier(control points, nu) -> se(3) -> se3_exp_map -> world_view, full_proj, camera_center for all K
//   subframes, and its backward from dL/d{world_view, full_proj} to the control points and nu; curve_type
//   "quarternion_cartesian" (scene/motion.py:191-194,242-246) takes the other branch: a 4-component quaternion curve,
//   normalised, to a rotation matrix (x, y, z, w convention of roma.unitquat_to_rotmat) + a cartesian translation curve.
// Replaces ~250 micro-kernels per training step in the reference (scene/bezier.py:54-83,
// utils/pytorch3d_functions.py:218-247,373-457,546-573, scene/motion.py:248-294, scene/cameras.py:63-74).
//
// Numerics follow the reference: the Bernstein powers are float32 (torch pow on float32 nu), everything after
// the float64 binomial table is float64, the world_view matrix is rounded to float32 before full_proj = view @ P.
// The backward differentiates the exponential map with forward-mode dual numbers (6 directions), so there is
// no hand-derived Jacobian to get wrong; eps-clamping of the squared angle has zero gradient like torch.clamp.
#include "dgs_common.h"

namespace {

constexpr int MAX_ORDER = 32;

constexpr int NCOMP = 7;  // curve outputs per subframe: 6 se(3) coordinates, or 3 translation + 4 quaternion components
// Forward-mode dual numbers with ONE direction: the backward runs one thread per (subframe, curve output), each carrying
// the value and its derivative along that output -- the directions are independent, so seven threads do a quarter of the
// dependent double-precision work each that a single thread carrying all seven did (48 -> 16 us at 9 subframes).
constexpr int ND = 1;
struct D6 {  // value + derivative w.r.t. this thread's curve output (trans xyz, rot xyz [, w])
  double v;
  double d[ND];
};
__device__ __forceinline__ D6 cst(double x) {
  D6 r;
  r.v = x;
#pragma unroll
  for (int i = 0; i < ND; i++) r.d[i] = 0.0;
  return r;
}
__device__ __forceinline__ D6 var(double x, bool mine) {   // mine: this is the output the thread differentiates along
  D6 r = cst(x);
  r.d[0] = mine ? 1.0 : 0.0;
  return r;
}
__device__ __forceinline__ D6 operator+(const D6& a, const D6& b) {
  D6 r;
  r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < ND; i++) r.d[i] = a.d[i] + b.d[i];
  return r;
}
__device__ __forceinline__ D6 operator-(const D6& a, const D6& b) {
  D6 r;
  r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < ND; i++) r.d[i] = a.d[i] - b.d[i];
  return r;
}
__device__ __forceinline__ D6 operator-(const D6& a) {
  D6 r;
  r.v = -a.v;
#pragma unroll
  for (int i = 0; i < ND; i++) r.d[i] = -a.d[i];
  return r;
}
__device__ __forceinline__ D6 operator*(const D6& a, const D6& b) {
  D6 r;
  r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < ND; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
__device__ __forceinline__ D6 operator/(const D6& a, const D6& b) {
  D6 r;
  const double inv = 1.0 / b.v;
  r.v = a.v * inv;
#pragma unroll
  for (int i = 0; i < ND; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
  return r;
}
__device__ __forceinline__ D6 dsqrt(const D6& a) {
  D6 r;
  r.v = sqrt(a.v);
  const double k = 0.5 / r.v;
#pragma unroll
  for (int i = 0; i < ND; i++) r.d[i] = a.d[i] * k;
  return r;
}
__device__ __forceinline__ D6 dsin(const D6& a) {
  D6 r;
  r.v = sin(a.v);
  const double c = cos(a.v);
#pragma unroll
  for (int i = 0; i < ND; i++) r.d[i] = a.d[i] * c;
  return r;
}
__device__ __forceinline__ D6 dcos(const D6& a) {
  D6 r;
  r.v = cos(a.v);
  const double s = -sin(a.v);
#pragma unroll
  for (int i = 0; i < ND; i++) r.d[i] = a.d[i] * s;
  return r;
}
__device__ __forceinline__ D6 dclamp_min(const D6& a, double lo) { return a.v < lo ? cst(lo) : a; }

// scene/bezier.py:54-64: coeff_c = binom(C,c) * t^(C-c) * (1-t)^c with float32 powers; also d coeff / dt
__device__ __forceinline__ void bernstein(int C, float t, double* coeff, double* dcoeff) {
  double binom = 1.0;
  const float u = 1.0f - t;
  // integer powers for the derivative by repeated multiplication (a double pow() per term cost 2 (C + 1) calls of a few
  // hundred dependent instructions each on the one active wave: 54 us at C = 9)
  double tp[MAX_ORDER + 1], up[MAX_ORDER + 1];
  if (dcoeff != nullptr) {
    tp[0] = 1.0;
    up[0] = 1.0;
    for (int i = 1; i <= C; i++) {
      tp[i] = tp[i - 1] * (double)t;
      up[i] = up[i - 1] * (double)u;
    }
  }
  for (int c = 0; c <= C; c++) {
    if (c > 0) binom = binom * (double)(C - c + 1) / (double)c;
    const float pa = powf(t, (float)(C - c));
    const float pb = powf(u, (float)c);
    coeff[c] = (double)(pa * pb) * binom;
    if (dcoeff != nullptr) {
      const double da = (C - c) > 0 ? (double)(C - c) * tp[C - c - 1] : 0.0;
      const double db = c > 0 ? -(double)c * up[c - 1] : 0.0;
      dcoeff[c] = binom * (da * (double)pb + (double)pa * db);
    }
  }
}

// utils/pytorch3d_functions.py:218-247,373-457,546-573 on dual numbers: R (row-major 3x3) and T = V @ trans
__device__ void se3_exp(const D6 se3[6], D6 R[9], D6 T[3]) {
  const double eps = 1e-4;
  const D6 &wx = se3[3], &wy = se3[4], &wz = se3[5];
  const D6 nrm = wx * wx + wy * wy + wz * wz;
  const D6 ang = dsqrt(dclamp_min(nrm, eps));
  const D6 inv = cst(1.0) / ang;
  const D6 sn = dsin(ang), cs = dcos(ang);
  const D6 fac1 = inv * sn;
  const D6 fac2 = inv * inv * (cst(1.0) - cs);
  const D6 z = cst(0.0);
  // hat(w) (pytorch3d_functions.py:337-372) and its square
  const D6 S[9] = {z, -wz, wy, wz, z, -wx, -wy, wx, z};
  D6 S2[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) S2[3 * r + c] = S[3 * r] * S[c] + S[3 * r + 1] * S[3 + c] + S[3 * r + 2] * S[6 + c];
  const D6 va = (cst(1.0) - cs) / (ang * ang);
  const D6 vb = (ang - sn) / (ang * ang * ang);
  D6 V[9];
  for (int i = 0; i < 9; i++) {
    const D6 eye = cst((i % 4 == 0) ? 1.0 : 0.0);
    R[i] = fac1 * S[i] + fac2 * S2[i] + eye;
    V[i] = eye + S[i] * va + S2[i] * vb;
  }
  for (int r = 0; r < 3; r++) T[r] = V[3 * r] * se3[0] + V[3 * r + 1] * se3[1] + V[3 * r + 2] * se3[2];
}

// MODE 0: forward outputs, one thread per subframe.  MODE 1: dL_dse3[k][6] and dL_dnu[k], eight threads per subframe
// (thread (k, dir) owns curve output dir; dir >= NP idles).
// roma.unitquat_to_rotmat on dual numbers, after q / |q| (scene/motion.py:243-245); R row-major 3x3
__device__ void quat_rot(const D6 q[4], D6 R[9]) {
  const D6 n = dsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const D6 x = q[0] / n, y = q[1] / n, z = q[2] / n, w = q[3] / n;
  const D6 x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
  const D6 xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
  const D6 two = cst(2.0);
  R[0] = x2 - y2 - z2 + w2;  R[1] = two * (xy - zw);     R[2] = two * (xz + yw);
  R[3] = two * (xy + zw);    R[4] = y2 - x2 - z2 + w2;   R[5] = two * (yz - xw);
  R[6] = two * (xz - yw);    R[7] = two * (yz + xw);     R[8] = z2 - x2 - y2 + w2;
}

// QUAT = 0: se(3) curves (rotation control points [C+1,3]); 1: quaternion + cartesian curves ([C+1,4])
template <int MODE, int QUAT>
__global__ void pose_kernel(int C, int K, const float* __restrict__ ctrl_trans, const float* __restrict__ ctrl_rot,
                            const float* __restrict__ nu, const float* __restrict__ proj, float* __restrict__ view,
                            float* __restrict__ full, float* __restrict__ campos, const float* __restrict__ dL_dview,
                            const float* __restrict__ dL_dfull, double* __restrict__ dL_dse3,
                            float* __restrict__ dL_dnu, double* __restrict__ coeff_out) {
  constexpr int NR = QUAT ? 4 : 3, NP = 3 + NR;   // rotation components, curve outputs per subframe
  __shared__ double s_gn[8][8];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = (MODE == 1) ? (t >> 3) : t;
  const int dir = (MODE == 1) ? (t & 7) : 0;
  const bool live = k < K && dir < NP;
  if (MODE == 0 && !live) return;
  const int kk = live ? k : 0;
  double coeff[MAX_ORDER + 1], dcoeff[MAX_ORDER + 1];
  bernstein(C, nu[kk], coeff, MODE == 1 ? dcoeff : nullptr);
  double s[NCOMP] = {0, 0, 0, 0, 0, 0, 0}, ds[NCOMP] = {0, 0, 0, 0, 0, 0, 0};
  for (int c = 0; c <= C; c++) {
    for (int d = 0; d < 3; d++) {
      s[d] += coeff[c] * (double)ctrl_trans[3 * c + d];
      if (MODE == 1) ds[d] += dcoeff[c] * (double)ctrl_trans[3 * c + d];
    }
    for (int d = 0; d < NR; d++) {
      s[3 + d] += coeff[c] * (double)ctrl_rot[NR * c + d];
      if (MODE == 1) ds[3 + d] += dcoeff[c] * (double)ctrl_rot[NR * c + d];
    }
  }
  D6 se3[NCOMP];
  for (int i = 0; i < NCOMP; i++) se3[i] = var(s[i], i == dir);
  D6 R[9], T[3];
  if (QUAT) {
    quat_rot(se3 + 3, R);
    for (int d = 0; d < 3; d++) T[d] = se3[d];
  } else {
    se3_exp(se3, R, T);
  }
  // scene/motion.py:277-279 (c2w rotation = R, translation = T in the row-vector convention):
  //   world_view[:3,:3] = R, world_view[3,:3] = -T @ R, rounded to float32
  if (MODE == 0) {
    float wv[16];
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) wv[4 * r + c] = (float)R[3 * r + c].v;
      wv[4 * r + 3] = 0.0f;
    }
    for (int c = 0; c < 3; c++) wv[12 + c] = (float)(-(T[0].v * R[c].v + T[1].v * R[3 + c].v + T[2].v * R[6 + c].v));
    wv[15] = 1.0f;
    for (int i = 0; i < 16; i++) view[16 * k + i] = wv[i];
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) {
        float acc = 0.0f;
        for (int j = 0; j < 4; j++) acc += wv[4 * r + j] * proj[4 * j + c];
        full[16 * k + 4 * r + c] = acc;
      }
    // camera_center = inverse(world_view)[3,:3] (scene/cameras.py:73-74) == T for a rigid transform
    for (int d = 0; d < 3; d++) campos[3 * k + d] = (float)T[d].v;
    if (coeff_out != nullptr)
      for (int c = 0; c <= C; c++) coeff_out[(size_t)k * (MAX_ORDER + 1) + c] = coeff[c];
  } else {
    // G = dL/dworld_view = dL_dview + dL_dfull @ P^T
    double G[16];
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) {
        double acc = (double)dL_dview[16 * kk + 4 * r + c];
        for (int j = 0; j < 4; j++) acc += (double)dL_dfull[16 * kk + 4 * r + j] * (double)proj[4 * c + j];
        G[4 * r + c] = acc;
      }
    double g = 0.0;   // dL / d(curve output dir of subframe k)
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) {
        const double gR = G[4 * r + c] - G[12 + c] * T[r].v;  // through world_view[r][c] and world_view[3][c]
        const double gT = -G[12 + c] * R[3 * r + c].v;        // dL/dT[r] contribution
        g += gR * R[3 * r + c].d[0] + gT * T[r].d[0];
      }
    if (live) dL_dse3[NP * k + dir] = g;
    s_gn[threadIdx.x >> 3][dir] = live ? g * ds[dir < NCOMP ? dir : 0] : 0.0;
    __syncthreads();
    if (live && dir == 0) {
      double gn = 0.0;
      for (int i = 0; i < NP; i++) gn += s_gn[threadIdx.x >> 3][i];   // in output order
      dL_dnu[k] = (float)gn;
      for (int c = 0; c <= C; c++) coeff_out[(size_t)k * (MAX_ORDER + 1) + c] = coeff[c];
    }
  }
}

// dL/dctrl[c][d] = sum_k coeff[k][c] * dL_dse3[k][d], summed in k order (deterministic)
__global__ void pose_ctrl_grad_kernel(int C, int K, int NP, const double* __restrict__ coeff,
                                      const double* __restrict__ dL_dse3, float* __restrict__ dL_dctrl_trans,
                                      float* __restrict__ dL_dctrl_rot) {
  const int i = threadIdx.x;  // (c, d) pair, d in 0..NP-1 (NP = 6 curve outputs, 7 with a quaternion curve)
  if (i >= (C + 1) * NP) return;
  const int c = i / NP, d = i % NP;
  double acc = 0.0;
  for (int k = 0; k < K; k++) acc += coeff[(size_t)k * (MAX_ORDER + 1) + c] * dL_dse3[NP * k + d];
  if (d < 3)
    dL_dctrl_trans[3 * c + d] = (float)acc;
  else
    dL_dctrl_rot[(NP - 3) * c + d - 3] = (float)acc;
}

// Subframe times of one view from its alignment parameters (scene/motion.py:209-219):
//   nu = sort(clamp(cat(0, sigmoid(raw) [+ u / f - 1 / (2 f)], 1), 0, 1))
// raw is the view's row of CameraMotionModule._nu (f - 2 values), u optional uniform samples (curve_random_sample).
// One block; thread i owns candidate i (0 and f-1 are the fixed end points), its rank is the number of candidates that
// sort before it (value, then index: a stable ascending sort), nu[rank] = value, src[rank] = i.
__global__ void alignment_fwd_kernel(int f, int n_sub, const float* __restrict__ raw, const float* __restrict__ u,
                                     float* __restrict__ nu, int32_t* __restrict__ src) {
  __shared__ float vals[DGS_MAX_K];
  const int i = threadIdx.x;
  if (i < f) {
    float v = 0.0f;
    if (i == f - 1 && f > 1) v = 1.0f;
    if (i > 0 && i < f - 1) {
      v = 1.0f / (1.0f + expf(-raw[i - 1]));
      if (u != nullptr) v = v + u[i - 1] / (float)n_sub - (1.0f / (float)(2 * n_sub));
    }
    vals[i] = fminf(1.0f, fmaxf(0.0f, v));
  }
  __syncthreads();
  if (i < f) {
    const float v = vals[i];
    int rank = 0;
    for (int j = 0; j < f; j++) rank += (vals[j] < v || (vals[j] == v && j < i)) ? 1 : 0;
    nu[rank] = v;
    src[rank] = i;
  }
}

// dL/draw[i-1] = dL/dnu[rank(i)] * [0 <= pre-clamp value <= 1] * sigmoid'(raw[i-1])   (torch: sort -> clamp -> cat ->
// sigmoid backward); the end points have no parameter
__global__ void alignment_bwd_kernel(int f, int n_sub, const float* __restrict__ raw, const float* __restrict__ u,
                                     const int32_t* __restrict__ src, const float* __restrict__ dL_dnu,
                                     float* __restrict__ dL_draw) {
  const int r = threadIdx.x;
  if (r >= f) return;
  const int i = src[r];
  if (i <= 0 || i >= f - 1) return;
  const float sg = 1.0f / (1.0f + expf(-raw[i - 1]));
  float v = sg;
  if (u != nullptr) v = v + u[i - 1] / (float)n_sub - (1.0f / (float)(2 * n_sub));
  const float pass = (v >= 0.0f && v <= 1.0f) ? 1.0f : 0.0f;
  dL_draw[i - 1] = dL_dnu[r] * pass * (sg * (1.0f - sg));
}

}  // namespace

extern "C" {

int dgs_alignment_forward(const float* raw, const float* uniform, int32_t f, int32_t n_subframes, float* nu,
                          int32_t* src, dgs_stream_t stream) {
  if (f < 1 || f > DGS_MAX_K || n_subframes < 1 || (f > 2 && raw == nullptr) || nu == nullptr || src == nullptr)
    return DGS_E_ARG;
  hipLaunchKernelGGL(alignment_fwd_kernel, dim3(1), dim3(DGS_MAX_K), 0, reinterpret_cast<hipStream_t>(stream), f,
                     n_subframes, raw, uniform, nu, src);
  return hipGetLastError() == hipSuccess ? DGS_OK : DGS_E_HIP;
}

int dgs_alignment_backward(const float* raw, const float* uniform, int32_t f, int32_t n_subframes, const int32_t* src,
                           const float* dL_dnu, float* dL_draw, dgs_stream_t stream) {
  if (f < 1 || f > DGS_MAX_K || n_subframes < 1 || src == nullptr || dL_dnu == nullptr || (f > 2 && (raw == nullptr || dL_draw == nullptr)))
    return DGS_E_ARG;
  if (f <= 2) return DGS_OK;
  hipLaunchKernelGGL(alignment_bwd_kernel, dim3(1), dim3(DGS_MAX_K), 0, reinterpret_cast<hipStream_t>(stream), f,
                     n_subframes, raw, uniform, src, dL_dnu, dL_draw);
  return hipGetLastError() == hipSuccess ? DGS_OK : DGS_E_HIP;
}

size_t dgs_pose_scratch_bytes(int32_t K) { return (size_t)K * (NCOMP + MAX_ORDER + 1) * sizeof(double) + 256; }

int dgs_pose_forward(const float* ctrl_trans, const float* ctrl_rot, const float* nu, const float* proj, int32_t C,
                     int32_t K, int32_t quaternion, float* view, float* full, float* campos, dgs_stream_t stream) {
  if (C < 0 || C > MAX_ORDER || K < 1 || ctrl_trans == nullptr || ctrl_rot == nullptr || nu == nullptr ||
      proj == nullptr || view == nullptr || full == nullptr || campos == nullptr)
    return DGS_E_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (quaternion)
    hipLaunchKernelGGL((pose_kernel<0, 1>), dim3((K + 63) / 64), dim3(64), 0, s, C, K, ctrl_trans, ctrl_rot, nu, proj,
                       view, full, campos, nullptr, nullptr, nullptr, nullptr, nullptr);
  else
    hipLaunchKernelGGL((pose_kernel<0, 0>), dim3((K + 63) / 64), dim3(64), 0, s, C, K, ctrl_trans, ctrl_rot, nu, proj,
                       view, full, campos, nullptr, nullptr, nullptr, nullptr, nullptr);
  return hipGetLastError() == hipSuccess ? DGS_OK : DGS_E_HIP;
}

int dgs_pose_backward(const float* ctrl_trans, const float* ctrl_rot, const float* nu, const float* proj, int32_t C,
                      int32_t K, int32_t quaternion, const float* dL_dview, const float* dL_dfull, void* scratch,
                      float* dL_dctrl_trans, float* dL_dctrl_rot, float* dL_dnu, dgs_stream_t stream) {
  if (C < 0 || C > MAX_ORDER || K < 1 || (C + 1) * NCOMP > 256 || ctrl_trans == nullptr || ctrl_rot == nullptr ||
      nu == nullptr || proj == nullptr || dL_dview == nullptr || dL_dfull == nullptr || scratch == nullptr ||
      dL_dctrl_trans == nullptr || dL_dctrl_rot == nullptr || dL_dnu == nullptr)
    return DGS_E_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  double* dse3 = reinterpret_cast<double*>(scratch);
  double* coeff = dse3 + (size_t)NCOMP * K;
  if (quaternion)
    hipLaunchKernelGGL((pose_kernel<1, 1>), dim3((K + 7) / 8), dim3(64), 0, s, C, K, ctrl_trans, ctrl_rot, nu, proj,
                       nullptr, nullptr, nullptr, dL_dview, dL_dfull, dse3, dL_dnu, coeff);
  else
    hipLaunchKernelGGL((pose_kernel<1, 0>), dim3((K + 7) / 8), dim3(64), 0, s, C, K, ctrl_trans, ctrl_rot, nu, proj,
                       nullptr, nullptr, nullptr, dL_dview, dL_dfull, dse3, dL_dnu, coeff);
  hipLaunchKernelGGL(pose_ctrl_grad_kernel, dim3(1), dim3(256), 0, s, C, K, quaternion ? 7 : 6, coeff, dse3,
                     dL_dctrl_trans, dL_dctrl_rot);
  return hipGetLastError() == hipSuccess ? DGS_OK : DGS_E_HIP;
}

}  // extern "C"
