// dgs_oracle.cpp -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; never shipped, never on the product path).
//
// Single-thread C++17 restatement of the reference rasteriser
//   /root/reference/submodules/diff-gaussian-rasterization/cuda_rasterizer/{forward.cu,backward.cu,
//   rasterizer_impl.cu,auxiliary.h}
// written from the reference's *behaviour*, one subframe per call, emitting every intermediate the
// reference keeps in its Geometry/Binning/Image state so that the HIP kernels can be compared stage by
// stage.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
//
// PARITY STATUS: the reference holds NO tests, golden vectors or fixtures for this path and its CUDA
// sources cannot be built here (no nvcc; building them would need stand-ins for the CUDA runtime, CUB
// and cooperative_groups headers).  The L0 kernel semantics restated here are therefore "parity
// unpinned" against a reference binary; what IS pinned: (1) the importable pure-torch reference
// utilities (SH evaluation, scale/rotation -> covariance, projection matrices, se3_exp_map, losses) via
// tests/golden/*.npz, and (2) every analytic gradient via float64 autograd of oracle/torch_naive.py.
//
// Floating-point contract (shared with the HIP preprocess kernel so that integer outputs -- radii, tile
// rectangles, tiles_touched, sort keys -- are bit-exact): IEEE fp32, no FMA contraction (build with
// -ffp-contract=off), correctly rounded division and sqrt, the operation order of the reference source
// (GLM column-major mat3 products: third_party/glm/glm/detail/type_mat3x3.inl:486-518), ndc2Pix and
// the dL_dproj terms in double.  float->int conversions saturate like the GPU's v_cvt_i32_f32.
// Built twice from this one file (oracle/Makefile): libdgs_oracle.so (single thread, the parity checker) and
// libdgs_oracle_omp.so (-fopenmp: tile/Gaussian loops spread over the host cores; used by bench.py's cpu_baseline leg
// and by the BASELINE-size parity tests, where the single-thread build would take minutes).  The OpenMP build is
// DETERMINISTIC and independent of the thread count: every place where the reference (and the single-thread build)
// accumulates with float atomicAdds -- whose order the reference itself leaves undefined, backward.cu:599-637,
// :277-294, :434-459 -- the OpenMP build accumulates in DOUBLE: per (tile, Gaussian) duplicate in pixel order, then per
// Gaussian in duplicate order; pose sums per fixed 4096-Gaussian chunk, then over chunks in order.  Each individual
// contribution is the same fp32 value in both builds; only the summation differs (the OpenMP build returns the
// correctly rounded centre of the reference's order-dependent fp32 results).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#include <memory>
#include <parallel/algorithm>
#endif

namespace {

constexpr int BLOCK_X = 16;  // config.h:16-17
constexpr int BLOCK_Y = 16;
constexpr int BLOCK_SIZE = BLOCK_X * BLOCK_Y;

// auxiliary.h:22-39
const float SH_C0 = 0.28209479177387814f;
const float SH_C1 = 0.4886025119029199f;
const float SH_C2[] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                       -1.0925484305920792f, 0.5462742152960396f};
const float SH_C3[] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                       -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

struct vec3 {
  float x, y, z;
};
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(vec3 a, vec3 b) {  // glm compute_dot<vec3>: tmp = a*b; tmp.x + tmp.y + tmp.z
  float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
  return tx + ty + tz;
}
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }

// GLM-style column-major 3x3: m[col][row]; the 9-scalar constructor fills column by column.
struct mat3 {
  float m[3][3];
  mat3() { std::memset(m, 0, sizeof(m)); }
  mat3(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2) {
    m[0][0] = a0; m[0][1] = a1; m[0][2] = a2;
    m[1][0] = b0; m[1][1] = b1; m[1][2] = b2;
    m[2][0] = c0; m[2][1] = c1; m[2][2] = c2;
  }
  float* operator[](int c) { return m[c]; }
  const float* operator[](int c) const { return m[c]; }
};
// type_mat3x3.inl:486-518
inline mat3 operator*(const mat3& A, const mat3& B) {
  mat3 R;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) R[c][r] = A[0][r] * B[c][0] + A[1][r] * B[c][1] + A[2][r] * B[c][2];
  return R;
}
inline mat3 operator*(float s, const mat3& A) {
  mat3 R;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) R[c][r] = A[c][r] * s;
  return R;
}
inline mat3 transpose(const mat3& A) {
  mat3 R;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) R[c][r] = A[r][c];
  return R;
}

// GPU-style saturating float -> int32 (CUDA cvt.rzi.s32.f32 and AMD v_cvt_i32_f32 both saturate, NaN -> 0).
inline int f2i(float f) {
  if (f != f) return 0;
  if (f >= 2147483648.0f) return 2147483647;
  if (f <= -2147483648.0f) return (-2147483647 - 1);
  return (int)f;
}

// auxiliary.h:41-44 -- evaluated in double, returned as float.
inline float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// auxiliary.h:46-56
inline void getRect(float px, float py, int max_radius, uint32_t& minx, uint32_t& miny, uint32_t& maxx,
                    uint32_t& maxy, int gx, int gy) {
  minx = (uint32_t)std::min(gx, std::max(0, f2i((px - max_radius) / BLOCK_X)));
  miny = (uint32_t)std::min(gy, std::max(0, f2i((py - max_radius) / BLOCK_Y)));
  maxx = (uint32_t)std::min(gx, std::max(0, f2i((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
  maxy = (uint32_t)std::min(gy, std::max(0, f2i((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

// auxiliary.h:58-77
inline vec3 transformPoint4x3(vec3 p, const float* M) {
  return {M[0] * p.x + M[4] * p.y + M[8] * p.z + M[12], M[1] * p.x + M[5] * p.y + M[9] * p.z + M[13],
          M[2] * p.x + M[6] * p.y + M[10] * p.z + M[14]};
}
struct vec4 {
  float x, y, z, w;
};
inline vec4 transformPoint4x4(vec3 p, const float* M) {
  return {M[0] * p.x + M[4] * p.y + M[8] * p.z + M[12], M[1] * p.x + M[5] * p.y + M[9] * p.z + M[13],
          M[2] * p.x + M[6] * p.y + M[10] * p.z + M[14], M[3] * p.x + M[7] * p.y + M[11] * p.z + M[15]};
}
// auxiliary.h:90-98
inline vec3 transformVec4x3Transpose(vec3 p, const float* M) {
  return {M[0] * p.x + M[1] * p.y + M[2] * p.z, M[4] * p.x + M[5] * p.y + M[6] * p.z,
          M[8] * p.x + M[9] * p.y + M[10] * p.z};
}
// auxiliary.h:107-117
inline vec3 dnormvdv(vec3 v, vec3 dv) {
  float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
  float invsum32 = 1.0f / std::sqrt(sum2 * sum2 * sum2);
  vec3 r;
  r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
  r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
  r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
  return r;
}
inline float sigmoidf(float x) { return 1.0f / (1.0f + std::exp(-x)); }            // auxiliary.h:134-137
inline float sigmoid_derivative(float x) { return sigmoidf(x) * (1.0f - sigmoidf(x)); }  // :139-142

// rasterizer_impl.cu:35-50
uint32_t getHigherMsb(uint32_t n) {
  uint32_t msb = sizeof(n) * 4;
  uint32_t step = msb;
  while (step > 1) {
    step /= 2;
    if (n >> msb)
      msb += step;
    else
      msb -= step;
  }
  if (n >> msb) msb++;
  return msb;
}

// forward.cu:129-163 (quaternion used as given; scale_modifier applied)
void computeCov3D(vec3 scale, float mod, const float* rot, float* cov3D) {
  mat3 S(1.0f, 0, 0, 0, 1.0f, 0, 0, 0, 1.0f);
  S[0][0] = mod * scale.x;
  S[1][1] = mod * scale.y;
  S[2][2] = mod * scale.z;
  float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
  mat3 R(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
         2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
         2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
  mat3 M = S * R;
  mat3 Sigma = transpose(M) * M;
  cov3D[0] = Sigma[0][0];
  cov3D[1] = Sigma[0][1];
  cov3D[2] = Sigma[0][2];
  cov3D[3] = Sigma[1][1];
  cov3D[4] = Sigma[1][2];
  cov3D[5] = Sigma[2][2];
}

// forward.cu:85-124
void computeCov2D(vec3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy, const float* cov3D,
                  const float* V, float& ca, float& cb, float& cc) {
  vec3 t = transformPoint4x3(mean, V);
  const float limx = 1.3f * tan_fovx;
  const float limy = 1.3f * tan_fovy;
  const float txtz = t.x / t.z;
  const float tytz = t.y / t.z;
  t.x = std::min(limx, std::max(-limx, txtz)) * t.z;
  t.y = std::min(limy, std::max(-limy, tytz)) * t.z;
  mat3 J(focal_x / t.z, 0.0f, -(focal_x * t.x) / (t.z * t.z), 0.0f, focal_y / t.z, -(focal_y * t.y) / (t.z * t.z),
         0, 0, 0);
  mat3 W(V[0], V[4], V[8], V[1], V[5], V[9], V[2], V[6], V[10]);
  mat3 T = W * J;
  mat3 Vrk(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
  mat3 cov = transpose(T) * transpose(Vrk) * T;
  cov[0][0] += 0.3f;
  cov[1][1] += 0.3f;
  ca = cov[0][0];
  cb = cov[0][1];
  cc = cov[1][1];
}

// forward.cu:20-82
vec3 computeColorFromSH(int idx, int deg, int max_coeffs, const float* means, vec3 campos, const float* shs,
                        float* pre_sigmoid, bool use_sigmoid) {
  vec3 pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
  vec3 dir = pos - campos;
  dir = dir / length(dir);
  const vec3* sh = reinterpret_cast<const vec3*>(shs) + (size_t)idx * max_coeffs;
  vec3 result = SH_C0 * sh[0];
  if (deg > 0) {
    float x = dir.x, y = dir.y, z = dir.z;
    result = result - SH_C1 * y * sh[1] + SH_C1 * z * sh[2] - SH_C1 * x * sh[3];
    if (deg > 1) {
      float xx = x * x, yy = y * y, zz = z * z;
      float xy = x * y, yz = y * z, xz = x * z;
      result = result + SH_C2[0] * xy * sh[4] + SH_C2[1] * yz * sh[5] + SH_C2[2] * (2.0f * zz - xx - yy) * sh[6] +
               SH_C2[3] * xz * sh[7] + SH_C2[4] * (xx - yy) * sh[8];
      if (deg > 2) {
        result = result + SH_C3[0] * y * (3.0f * xx - yy) * sh[9] + SH_C3[1] * xy * z * sh[10] +
                 SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[11] +
                 SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12] +
                 SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[13] + SH_C3[5] * z * (xx - yy) * sh[14] +
                 SH_C3[6] * x * (xx - 3.0f * yy) * sh[15];
      }
    }
  }
  if (use_sigmoid) {
    pre_sigmoid[3 * idx + 0] = result.x;
    pre_sigmoid[3 * idx + 1] = result.y;
    pre_sigmoid[3 * idx + 2] = result.z;
    result.x = sigmoidf(result.x);
    result.y = sigmoidf(result.y);
    result.z = sigmoidf(result.z);
  } else {
    result = result + vec3{0.5f, 0.5f, 0.5f};
    pre_sigmoid[3 * idx + 0] = result.x >= 0.0f;
    pre_sigmoid[3 * idx + 1] = result.y >= 0.0f;
    pre_sigmoid[3 * idx + 2] = result.z >= 0.0f;
    result = {std::max(result.x, 0.0f), std::max(result.y, 0.0f), std::max(result.z, 0.0f)};
  }
  return result;
}

}  // namespace

// OpenMP build only: 1 = round every accumulation of the compositing backward AND of the pose sums to fp32 (same
// deterministic order as the double accumulation).  The difference between the two modes is the fp32 rounding noise of THAT summation structure
// (per duplicate, then per Gaussian over duplicates -- the structure the HIP path uses too); the parity tests use it as
// the noise floor next to the 1e-4 bar.
bool g_accum_f32 = false;

extern "C" {

void dgs_oracle_set_accum_f32(int on) { g_accum_f32 = on != 0; }

int dgs_oracle_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
// OpenMP threads of the parallel regions the CALLING host thread opens from now on (the ICV is per thread: several host
// threads may each run one subframe on a share of the cores).  Round 6: on the 256-thread host of the GPU box one
// subframe's backward takes 1.85 s with all 256 threads and 0.84 s with 32 (profiles/oracle_threads_r06.txt).
int dgs_oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}

uint32_t dgs_oracle_higher_msb(uint32_t n) { return getHigherMsb(n); }

// FORWARD::preprocess + InclusiveSum (forward.cu:166-268, rasterizer_impl.cu:253-287).
// All outputs are caller-allocated and must be zero-initialised.  Returns num_rendered.
// Null-able inputs: shs / colors_precomp (exactly one), scales+rotations / cov3D_precomp (exactly one).
int dgs_oracle_preprocess(int P, int D, int M, int W, int H, const float* means3D, const float* shs,
                          const float* colors_precomp, const float* opacities, const float* scales,
                          float scale_modifier, const float* rotations, const float* cov3D_precomp,
                          const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                          float tan_fovy, int use_sigmoid,
                          /* geometry state */ int* radii, float* depths, float* pre_sigmoid, float* means2D,
                          float* cov3Ds, float* conic_opacity, float* rgb, uint32_t* tiles_touched,
                          uint32_t* point_offsets) {
  const float focal_y = H / (2.0f * tan_fovy);  // rasterizer_impl.cu:227-228
  const float focal_x = W / (2.0f * tan_fovx);
  const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
  const vec3 cam = {campos[0], campos[1], campos[2]};
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < P; idx++) {
    radii[idx] = 0;
    tiles_touched[idx] = 0;
    vec3 p_orig = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    // in_frustum (auxiliary.h:144-169): near cull only
    vec3 p_view = transformPoint4x3(p_orig, viewmatrix);
    if (p_view.z <= 0.2f) continue;
    vec4 p_hom = transformPoint4x4(p_orig, projmatrix);
    float p_w = 1.0f / (p_hom.w + 0.0000001f);
    vec3 p_proj = {p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w};
    const float* cov3D;
    if (cov3D_precomp != nullptr) {
      cov3D = cov3D_precomp + (size_t)idx * 6;
    } else {
      computeCov3D({scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]}, scale_modifier, rotations + 4 * idx,
                   cov3Ds + (size_t)idx * 6);
      cov3D = cov3Ds + (size_t)idx * 6;
    }
    float ca, cb, cc;
    computeCov2D(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, ca, cb, cc);
    float det = (ca * cc - cb * cb);
    if (det == 0.0f) continue;
    float det_inv = 1.f / det;
    float conic_x = cc * det_inv, conic_y = -cb * det_inv, conic_z = ca * det_inv;
    float mid = 0.5f * (ca + cc);
    float lambda1 = mid + std::sqrt(std::max(0.1f, mid * mid - det));
    float lambda2 = mid - std::sqrt(std::max(0.1f, mid * mid - det));
    float my_radius = std::ceil(3.f * std::sqrt(std::max(lambda1, lambda2)));
    float pix_x = ndc2Pix(p_proj.x, W), pix_y = ndc2Pix(p_proj.y, H);
    uint32_t minx, miny, maxx, maxy;
    getRect(pix_x, pix_y, f2i(my_radius), minx, miny, maxx, maxy, gx, gy);
    if ((maxx - minx) * (maxy - miny) == 0) continue;
    if (colors_precomp == nullptr) {
      vec3 c = computeColorFromSH(idx, D, M, means3D, cam, shs, pre_sigmoid, use_sigmoid != 0);
      rgb[3 * idx + 0] = c.x;
      rgb[3 * idx + 1] = c.y;
      rgb[3 * idx + 2] = c.z;
    }
    depths[idx] = p_view.z;
    radii[idx] = f2i(my_radius);
    means2D[2 * idx] = pix_x;
    means2D[2 * idx + 1] = pix_y;
    conic_opacity[4 * idx + 0] = conic_x;
    conic_opacity[4 * idx + 1] = conic_y;
    conic_opacity[4 * idx + 2] = conic_z;
    conic_opacity[4 * idx + 3] = opacities[idx];
    tiles_touched[idx] = (maxy - miny) * (maxx - minx);
  }
  uint32_t run = 0;  // cub::DeviceScan::InclusiveSum (rasterizer_impl.cu:283)
  for (int i = 0; i < P; i++) {
    run += tiles_touched[i];
    point_offsets[i] = run;
  }
  return P > 0 ? (int)point_offsets[P - 1] : 0;
}

// duplicateWithKeys + stable radix sort on bits [0, 32+bit) + identifyTileRanges
// (rasterizer_impl.cu:70-138, 295-324).  ranges is [T][2], zero-initialised by the caller (memset :316).
void dgs_oracle_bin(int P, int W, int H, int R, const int* radii, const float* means2D, const float* depths,
                    const uint32_t* point_offsets, uint64_t* keys_unsorted, uint32_t* vals_unsorted,
                    uint64_t* keys_sorted, uint32_t* point_list, uint32_t* ranges) {
  const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < P; idx++) {
    if (radii[idx] > 0) {
      uint32_t off = (idx == 0) ? 0 : point_offsets[idx - 1];
      uint32_t minx, miny, maxx, maxy;
      getRect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], minx, miny, maxx, maxy, gx, gy);
      uint32_t dbits;
      std::memcpy(&dbits, &depths[idx], 4);
      for (uint32_t y = miny; y < maxy; y++)
        for (uint32_t x = minx; x < maxx; x++) {
          uint64_t key = y * (uint32_t)gx + x;
          key <<= 32;
          key |= dbits;
          keys_unsorted[off] = key;
          vals_unsorted[off] = (uint32_t)idx;
          off++;
        }
    }
  }
  const int bit = (int)getHigherMsb((uint32_t)(gx * gy));
  const uint64_t mask = (32 + bit >= 64) ? ~0ull : ((1ull << (32 + bit)) - 1);
  std::vector<uint32_t> order(R);
  std::iota(order.begin(), order.end(), 0u);
  auto less = [&](uint32_t a, uint32_t b) { return (keys_unsorted[a] & mask) < (keys_unsorted[b] & mask); };
#ifdef _OPENMP
  __gnu_parallel::stable_sort(order.begin(), order.end(), less);
#else
  std::stable_sort(order.begin(), order.end(), less);
#endif
#pragma omp parallel for schedule(static)
  for (int i = 0; i < R; i++) {
    keys_sorted[i] = keys_unsorted[order[i]];
    point_list[i] = vals_unsorted[order[i]];
  }
  for (int idx = 0; idx < R; idx++) {
    uint32_t currtile = (uint32_t)(keys_sorted[idx] >> 32);
    if (idx == 0)
      ranges[2 * currtile] = 0;
    else {
      uint32_t prevtile = (uint32_t)(keys_sorted[idx - 1] >> 32);
      if (currtile != prevtile) {
        ranges[2 * prevtile + 1] = idx;
        ranges[2 * currtile] = idx;
      }
    }
    if (idx == R - 1) ranges[2 * currtile + 1] = R;
  }
}

// FORWARD::render (forward.cu:273-392): per pixel front-to-back compositing of colour and depth.
void dgs_oracle_render(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* means2D,
                       const float* features, const float* depths, const float* conic_opacity, const float* bg,
                       float z_far, float* final_T, uint32_t* n_contrib, float* out_color, float* out_depth,
                       uint32_t* contrib_checksum) {
  // contrib_checksum (optional, test infrastructure for test infrastructure): per pixel, the wrap-around sum of
  // pos * 2654435761 over the list positions pos (1-based, inside the tile's list) of the pairs that CONTRIBUTE to the pixel --
  // those that pass the three tests below.  Two traversals of the same list took the same per-pair decisions at a pixel iff
  // their checksums agree (up to a 2^-32 collision): the parity tests compare it with the HIP forward's
  // (DgsForwardOut.debug_contrib_checksum) to find the pixels where an exp() ulp flipped a threshold, instead of masking
  // every pixel that sits within a margin of one.
  const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
  for (int ty = 0; ty < gy; ty++)
    for (int tx = 0; tx < gx; tx++) {
      const uint32_t r0 = ranges[2 * (ty * gx + tx)], r1 = ranges[2 * (ty * gx + tx) + 1];
      for (int ly = 0; ly < BLOCK_Y; ly++)
        for (int lx = 0; lx < BLOCK_X; lx++) {
          const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
          if (!(px < W && py < H)) continue;
          const uint32_t pix_id = (uint32_t)W * py + px;
          const float pfx = (float)px, pfy = (float)py;
          float T = 1.0f;
          uint32_t chk = 0;
          uint32_t contributor = 0, last_contributor = 0;
          float C[3] = {0, 0, 0};
          float Dacc = 0.0f;
          for (uint32_t s = r0; s < r1; s++) {
            contributor++;
            const uint32_t g = point_list[s];
            const float dx = means2D[2 * g] - pfx, dy = means2D[2 * g + 1] - pfy;
            const float* co = conic_opacity + 4 * (size_t)g;
            const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
            if (power > 0.0f) continue;
            const float alpha = std::min(0.99f, co[3] * std::exp(power));
            if (alpha < 1.0f / 255.0f) continue;
            const float test_T = T * (1 - alpha);
            if (test_T < 0.0001f) break;  // done = true
            for (int ch = 0; ch < 3; ch++) C[ch] += features[3 * (size_t)g + ch] * alpha * T;
            Dacc += depths[g] * alpha * T;
            T = test_T;
            last_contributor = contributor;
            chk += contributor * 2654435761u;
          }
          final_T[pix_id] = T;
          n_contrib[pix_id] = last_contributor;
          if (contrib_checksum != nullptr) contrib_checksum[pix_id] = chk;
          for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pix_id] = C[ch] + T * bg[ch];
          out_depth[pix_id] = Dacc + T * z_far;
        }
    }
}

// BACKWARD::render (backward.cu:463-640): back-to-front; the reference's atomicAdds become plain += in
// tile-major / pixel-major / back-to-front order.  dL_dmean2D is [P,3], dL_dconic [P,4] (x,y,-,w).
void dgs_oracle_render_backward(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* bg,
                                const float* means2D, const float* conic_opacity, const float* colors,
                                const float* depths, const float* final_Ts, const uint32_t* n_contrib,
                                const float* dL_dpixels, const float* dL_dpixeldepths, float z_far,
                                float* dL_dmean2D, float* dL_dconic2D, float* dL_dopacity, float* dL_dcolors,
                                float* dL_ddepths) {
  const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
  const float ddelx_dx = 0.5 * W;  // backward.cu:535-536
  const float ddely_dy = 0.5 * H;
#ifdef _OPENMP
  // one row of double accumulators per (tile, Gaussian) duplicate = per position of the sorted list
  uint32_t Rtot = 0;
  for (int t = 0; t < gx * gy; t++) Rtot = std::max(Rtot, ranges[2 * t + 1]);
  // (allocated without a serial zero-fill: 320 MB per metric subframe; the pages are first touched -- and zeroed -- in
  // parallel.  Round 6: the serial fill and the serial per-Gaussian sum below were most of what was left of a call once
  // the tile loop ran on a few dozen threads)
  std::unique_ptr<double[]> dacc_mem(new double[(size_t)Rtot * 10]);
  double* const dacc = dacc_mem.get();
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)Rtot * 10; i++) dacc[i] = 0.0;
  const bool f32 = g_accum_f32;  // emulate fp32 accumulation in the same order (noise-floor estimate)
#define DGS_ACC(slot, dst, val)                                   \
  do {                                                            \
    double& a__ = dacc[(size_t)s * 10 + (slot)];                  \
    a__ += (double)(val);                                         \
    if (f32) a__ = (double)(float)a__;                            \
  } while (0)
#else
#define DGS_ACC(slot, dst, val) dst += (val)
#endif
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
  for (int ty = 0; ty < gy; ty++)
    for (int tx = 0; tx < gx; tx++) {
      const uint32_t r0 = ranges[2 * (ty * gx + tx)], r1 = ranges[2 * (ty * gx + tx) + 1];
      for (int ly = 0; ly < BLOCK_Y; ly++)
        for (int lx = 0; lx < BLOCK_X; lx++) {
          const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
          if (!(px < W && py < H)) continue;
          const uint32_t pix_id = (uint32_t)W * py + px;
          const float pfx = (float)px, pfy = (float)py;
          const float T_final = final_Ts[pix_id];
          float T = T_final;
          uint32_t contributor = r1 - r0;
          const uint32_t last_contributor = n_contrib[pix_id];
          float accum_rec[3] = {0, 0, 0};
          float dL_dpixel[3];
          for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpixels[(size_t)i * H * W + pix_id];
          const float dL_dpixeldepth = dL_dpixeldepths[pix_id];
          float accum_depth_rec = 0;
          float last_alpha = 0;
          float last_color[3] = {0, 0, 0};
          float last_depth = 0;
          for (uint32_t s = r1; s-- > r0;) {
            contributor--;
            if (contributor >= last_contributor) continue;
            const uint32_t g = point_list[s];
            const float dx = means2D[2 * g] - pfx, dy = means2D[2 * g + 1] - pfy;
            const float* co = conic_opacity + 4 * (size_t)g;
            const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
            if (power > 0.0f) continue;
            const float G = std::exp(power);
            const float alpha = std::min(0.99f, co[3] * G);
            if (alpha < 1.0f / 255.0f) continue;
            T = T / (1.f - alpha);
            const float dchannel_dcolor = alpha * T;
            float dL_dalpha = 0.0f;
            for (int ch = 0; ch < 3; ch++) {
              const float c = colors[3 * (size_t)g + ch];
              accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
              last_color[ch] = c;
              const float dL_dchannel = dL_dpixel[ch];
              dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
              DGS_ACC(ch, dL_dcolors[3 * (size_t)g + ch], dchannel_dcolor * dL_dchannel);
            }
            const float c_d = depths[g];
            accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
            last_depth = c_d;
            dL_dalpha += (c_d - accum_depth_rec) * dL_dpixeldepth;
            DGS_ACC(3, dL_ddepths[g], dchannel_dcolor * dL_dpixeldepth);
            dL_dalpha *= T;
            last_alpha = alpha;
            float bg_dot_dpixel = 0;
            for (int i = 0; i < 3; i++) bg_dot_dpixel += bg[i] * dL_dpixel[i];
            bg_dot_dpixel += z_far * dL_dpixeldepth;
            dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
            const float dL_dG = co[3] * dL_dalpha;
            const float gdx = G * dx;
            const float gdy = G * dy;
            const float dG_ddelx = -gdx * co[0] - gdy * co[1];
            const float dG_ddely = -gdy * co[2] - gdx * co[1];
            DGS_ACC(4, dL_dmean2D[3 * (size_t)g + 0], dL_dG * dG_ddelx * ddelx_dx);
            DGS_ACC(5, dL_dmean2D[3 * (size_t)g + 1], dL_dG * dG_ddely * ddely_dy);
            DGS_ACC(6, dL_dconic2D[4 * (size_t)g + 0], -0.5f * gdx * dx * dL_dG);
            DGS_ACC(7, dL_dconic2D[4 * (size_t)g + 1], -0.5f * gdx * dy * dL_dG);
            DGS_ACC(8, dL_dconic2D[4 * (size_t)g + 3], -0.5f * gdy * dy * dL_dG);
            DGS_ACC(9, dL_dopacity[g], G * dL_dalpha);
          }
        }
    }
#undef DGS_ACC
#ifdef _OPENMP
  {
    // per Gaussian: sum its duplicates' rows in duplicate (= tile) order, in double, round once.  P is not an argument
    // of this entry point; the sums are formed in a map keyed by the point list instead.
    uint32_t maxg = 0;
    for (uint32_t s = 0; s < Rtot; s++) maxg = std::max(maxg, point_list[s]);
    std::vector<double> gsum(((size_t)maxg + 1) * 10, 0.0);
    std::vector<uint8_t> seen((size_t)maxg + 1, 0);
    // every thread owns a contiguous range of Gaussians and walks the whole list for them: each Gaussian's rows are still
    // added in list order by one thread (same sums as the serial loop, bit for bit, for any thread count)
#pragma omp parallel
    {
      const int nt = omp_get_num_threads(), me = omp_get_thread_num();
      const uint64_t span = (uint64_t)maxg + 1;
      const uint32_t g0 = (uint32_t)(span * me / nt), g1 = (uint32_t)(span * (me + 1) / nt);
      for (uint32_t s = 0; s < Rtot; s++) {
        const uint32_t gi = point_list[s];
        if (gi < g0 || gi >= g1) continue;
        const size_t g = gi;
        seen[g] = 1;
        for (int i = 0; i < 10; i++) {
          gsum[g * 10 + i] += dacc[(size_t)s * 10 + i];
          if (f32) gsum[g * 10 + i] = (double)(float)gsum[g * 10 + i];
        }
      }
    }
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g <= (int64_t)maxg; g++) {
      if (!seen[g]) continue;
      const double* a = &gsum[(size_t)g * 10];
      for (int ch = 0; ch < 3; ch++) dL_dcolors[3 * g + ch] = (float)a[ch];
      dL_ddepths[g] = (float)a[3];
      dL_dmean2D[3 * g + 0] = (float)a[4];
      dL_dmean2D[3 * g + 1] = (float)a[5];
      dL_dconic2D[4 * g + 0] = (float)a[6];
      dL_dconic2D[4 * g + 1] = (float)a[7];
      dL_dconic2D[4 * g + 3] = (float)a[8];
      dL_dopacity[g] = (float)a[9];
    }
  }
#endif
}

// BACKWARD::preprocess = computeCov2DCUDA (backward.cu:145-295) then preprocessCUDA (backward.cu:367-460)
// with computeColorFromSH bwd (:20-140) and computeCov3D bwd (:299-362).  cov3Ds is cov3D_precomp when that
// was given, else the forward's geometry-state cov3D (rasterizer_impl.cu:437).  Grad outputs zero-initialised.
void dgs_oracle_preprocess_backward(int P, int D, int M, int W, int H, const float* means3D, const int* radii,
                                    const float* shs, const float* pre_sigmoid, const float* scales,
                                    const float* rotations, float scale_modifier, const float* cov3Ds,
                                    const float* viewmatrix, const float* proj, const float* campos_,
                                    float tan_fovx, float tan_fovy, int use_sigmoid, const float* dL_dmean2D,
                                    const float* dL_dconics, float* dL_dmeans, float* dL_dcolor,
                                    const float* dL_ddepth, float* dL_dcov, float* dL_dsh, float* dL_dscale,
                                    float* dL_drot, float* dL_dview_matrix, float* dL_dproj) {
  const float h_y = H / (2.0f * tan_fovy);
  const float h_x = W / (2.0f * tan_fovx);
  const vec3 campos = {campos_[0], campos_[1], campos_[2]};
  // Pose sums: the single-thread build adds every contribution to the fp32 result in index order (the reference uses
  // float atomicAdds in an undefined order); the OpenMP build adds them in double per fixed chunk of POSE_CHUNK
  // Gaussians and then over the chunks in order (deterministic, thread-count independent, rounded once at the end).
  constexpr int POSE_CHUNK = 4096;
  const int nchunks = (P + POSE_CHUNK - 1) / POSE_CHUNK;
#ifdef _OPENMP
  std::vector<double> part_view((size_t)nchunks * 16 * 2, 0.0), part_proj((size_t)nchunks * 16, 0.0);
#endif
  // ---- computeCov2DCUDA
#pragma omp parallel for schedule(dynamic, 1)
  for (int chunk = 0; chunk < nchunks; chunk++) {
#ifdef _OPENMP
  double* dL_dview_matrix_acc = &part_view[(size_t)chunk * 16];
#else
  float* dL_dview_matrix_acc = dL_dview_matrix;
#endif
  for (int idx = chunk * POSE_CHUNK; idx < std::min(P, (chunk + 1) * POSE_CHUNK); idx++) {
    if (!(radii[idx] > 0)) continue;
    const float* cov3D = cov3Ds + 6 * (size_t)idx;
    vec3 mean = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    vec3 dL_dconic = {dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3]};
    vec3 t = transformPoint4x3(mean, viewmatrix);
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    const float txtz = t.x / t.z;
    const float tytz = t.y / t.z;
    t.x = std::min(limx, std::max(-limx, txtz)) * t.z;
    t.y = std::min(limy, std::max(-limy, tytz)) * t.z;
    const float x_grad_mul = txtz < -limx || txtz > limx ? 0 : 1;
    const float y_grad_mul = tytz < -limy || tytz > limy ? 0 : 1;
    mat3 J(h_x / t.z, 0.0f, -(h_x * t.x) / (t.z * t.z), 0.0f, h_y / t.z, -(h_y * t.y) / (t.z * t.z), 0, 0, 0);
    const float* V = viewmatrix;
    mat3 Wm(V[0], V[4], V[8], V[1], V[5], V[9], V[2], V[6], V[10]);
    mat3 Vrk(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    mat3 T = Wm * J;
    mat3 cov2D = transpose(T) * transpose(Vrk) * T;
    float a = cov2D[0][0] += 0.3f;
    float b = cov2D[0][1];
    float c = cov2D[1][1] += 0.3f;
    float denom = a * c - b * b;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float* dc = dL_dcov + 6 * (size_t)idx;
    if (denom2inv != 0) {
      dL_da = denom2inv * (-c * c * dL_dconic.x + 2 * b * c * dL_dconic.y + (denom - a * c) * dL_dconic.z);
      dL_dc = denom2inv * (-a * a * dL_dconic.z + 2 * a * b * dL_dconic.y + (denom - a * c) * dL_dconic.x);
      dL_db = denom2inv * 2 * (b * c * dL_dconic.x - (denom + 2 * b * b) * dL_dconic.y + a * b * dL_dconic.z);
      dc[0] = (T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc);
      dc[3] = (T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc);
      dc[5] = (T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc);
      dc[1] = 2 * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db +
              2 * T[1][0] * T[1][1] * dL_dc;
      dc[2] = 2 * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db +
              2 * T[1][0] * T[1][2] * dL_dc;
      dc[4] = 2 * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db +
              2 * T[1][1] * T[1][2] * dL_dc;
    } else {
      for (int i = 0; i < 6; i++) dc[i] = 0;
    }
    float dL_dT00 = 2 * (T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2]) * dL_da +
                    (T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2]) * dL_db;
    float dL_dT01 = 2 * (T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2]) * dL_da +
                    (T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2]) * dL_db;
    float dL_dT02 = 2 * (T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2]) * dL_da +
                    (T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2]) * dL_db;
    float dL_dT10 = 2 * (T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2]) * dL_dc +
                    (T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2]) * dL_db;
    float dL_dT11 = 2 * (T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2]) * dL_dc +
                    (T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2]) * dL_db;
    float dL_dT12 = 2 * (T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2]) * dL_dc +
                    (T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2]) * dL_db;
    float dL_dJ00 = Wm[0][0] * dL_dT00 + Wm[0][1] * dL_dT01 + Wm[0][2] * dL_dT02;
    float dL_dJ02 = Wm[2][0] * dL_dT00 + Wm[2][1] * dL_dT01 + Wm[2][2] * dL_dT02;
    float dL_dJ11 = Wm[1][0] * dL_dT10 + Wm[1][1] * dL_dT11 + Wm[1][2] * dL_dT12;
    float dL_dJ12 = Wm[2][0] * dL_dT10 + Wm[2][1] * dL_dT11 + Wm[2][2] * dL_dT12;
    float tz = 1.f / t.z;
    float tz2 = tz * tz;
    float tz3 = tz2 * tz;
    float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 +
                   (2 * h_y * t.y) * tz3 * dL_dJ12;
    vec3 dL_dmean = transformVec4x3Transpose({dL_dtx, dL_dty, dL_dtz}, viewmatrix);
    dL_dmeans[3 * idx + 0] = dL_dmean.x;
    dL_dmeans[3 * idx + 1] = dL_dmean.y;
    dL_dmeans[3 * idx + 2] = dL_dmean.z;
    // backward.cu:277-294 -- view-matrix gradient through t = view * mean only
    dL_dview_matrix_acc[0] += dL_dtx * mean.x;
    dL_dview_matrix_acc[1] += dL_dty * mean.x;
    dL_dview_matrix_acc[2] += dL_dtz * mean.x;
    dL_dview_matrix_acc[4] += dL_dtx * mean.y;
    dL_dview_matrix_acc[5] += dL_dty * mean.y;
    dL_dview_matrix_acc[6] += dL_dtz * mean.y;
    dL_dview_matrix_acc[8] += dL_dtx * mean.z;
    dL_dview_matrix_acc[9] += dL_dty * mean.z;
    dL_dview_matrix_acc[10] += dL_dtz * mean.z;
    dL_dview_matrix_acc[12] += dL_dtx;
    dL_dview_matrix_acc[13] += dL_dty;
    dL_dview_matrix_acc[14] += dL_dtz;
#ifdef _OPENMP
    if (g_accum_f32)
      for (int i = 0; i < 16; i++) dL_dview_matrix_acc[i] = (double)(float)dL_dview_matrix_acc[i];
#endif
  }
  }
  // ---- preprocessCUDA (bwd)
#pragma omp parallel for schedule(dynamic, 1)
  for (int chunk = 0; chunk < nchunks; chunk++) {
#ifdef _OPENMP
  double* dL_dview_matrix_acc = &part_view[(size_t)(nchunks + chunk) * 16];
  double* dL_dproj_acc = &part_proj[(size_t)chunk * 16];
#else
  float* dL_dview_matrix_acc = dL_dview_matrix;
  float* dL_dproj_acc = dL_dproj;
#endif
  for (int idx = chunk * POSE_CHUNK; idx < std::min(P, (chunk + 1) * POSE_CHUNK); idx++) {
    if (!(radii[idx] > 0)) continue;
    vec3 m = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    vec4 m_hom = transformPoint4x4(m, proj);
    float m_w = 1.0f / (m_hom.w + 0.0000001f);
    const float g2x = dL_dmean2D[3 * idx + 0], g2y = dL_dmean2D[3 * idx + 1];
    float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
    float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
    vec3 dL_dmean;
    dL_dmean.x = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y +
                 dL_ddepth[idx] * viewmatrix[2];
    dL_dmean.y = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y +
                 dL_ddepth[idx] * viewmatrix[6];
    dL_dmean.z = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y +
                 dL_ddepth[idx] * viewmatrix[10];
    dL_dmeans[3 * idx + 0] += dL_dmean.x;
    dL_dmeans[3 * idx + 1] += dL_dmean.y;
    dL_dmeans[3 * idx + 2] += dL_dmean.z;

    if (shs) {  // computeColorFromSH bwd, backward.cu:20-140
      vec3 dir_orig = m - campos;
      vec3 dir = dir_orig / length(dir_orig);
      const vec3* sh = reinterpret_cast<const vec3*>(shs) + (size_t)idx * M;
      vec3 dL_dRGB = {dL_dcolor[3 * idx], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2]};
      dL_dRGB.x *= use_sigmoid ? sigmoid_derivative(pre_sigmoid[3 * idx + 0]) : pre_sigmoid[3 * idx + 0];
      dL_dRGB.y *= use_sigmoid ? sigmoid_derivative(pre_sigmoid[3 * idx + 1]) : pre_sigmoid[3 * idx + 1];
      dL_dRGB.z *= use_sigmoid ? sigmoid_derivative(pre_sigmoid[3 * idx + 2]) : pre_sigmoid[3 * idx + 2];
      vec3 dRGBdx = {0, 0, 0}, dRGBdy = {0, 0, 0}, dRGBdz = {0, 0, 0};
      float x = dir.x, y = dir.y, z = dir.z;
      vec3* dsh = reinterpret_cast<vec3*>(dL_dsh) + (size_t)idx * M;
      float dRGBdsh0 = SH_C0;
      dsh[0] = dRGBdsh0 * dL_dRGB;
      if (D > 0) {
        float dRGBdsh1 = -SH_C1 * y;
        float dRGBdsh2 = SH_C1 * z;
        float dRGBdsh3 = -SH_C1 * x;
        dsh[1] = dRGBdsh1 * dL_dRGB;
        dsh[2] = dRGBdsh2 * dL_dRGB;
        dsh[3] = dRGBdsh3 * dL_dRGB;
        dRGBdx = -SH_C1 * sh[3];
        dRGBdy = -SH_C1 * sh[1];
        dRGBdz = SH_C1 * sh[2];
        if (D > 1) {
          float xx = x * x, yy = y * y, zz = z * z;
          float xy = x * y, yz = y * z, xz = x * z;
          float dRGBdsh4 = SH_C2[0] * xy;
          float dRGBdsh5 = SH_C2[1] * yz;
          float dRGBdsh6 = SH_C2[2] * (2.f * zz - xx - yy);
          float dRGBdsh7 = SH_C2[3] * xz;
          float dRGBdsh8 = SH_C2[4] * (xx - yy);
          dsh[4] = dRGBdsh4 * dL_dRGB;
          dsh[5] = dRGBdsh5 * dL_dRGB;
          dsh[6] = dRGBdsh6 * dL_dRGB;
          dsh[7] = dRGBdsh7 * dL_dRGB;
          dsh[8] = dRGBdsh8 * dL_dRGB;
          dRGBdx = dRGBdx + (SH_C2[0] * y * sh[4] + SH_C2[2] * 2.f * -x * sh[6] + SH_C2[3] * z * sh[7] +
                             SH_C2[4] * 2.f * x * sh[8]);
          dRGBdy = dRGBdy + (SH_C2[0] * x * sh[4] + SH_C2[1] * z * sh[5] + SH_C2[2] * 2.f * -y * sh[6] +
                             SH_C2[4] * 2.f * -y * sh[8]);
          dRGBdz = dRGBdz + (SH_C2[1] * y * sh[5] + SH_C2[2] * 2.f * 2.f * z * sh[6] + SH_C2[3] * x * sh[7]);
          if (D > 2) {
            float dRGBdsh9 = SH_C3[0] * y * (3.f * xx - yy);
            float dRGBdsh10 = SH_C3[1] * xy * z;
            float dRGBdsh11 = SH_C3[2] * y * (4.f * zz - xx - yy);
            float dRGBdsh12 = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
            float dRGBdsh13 = SH_C3[4] * x * (4.f * zz - xx - yy);
            float dRGBdsh14 = SH_C3[5] * z * (xx - yy);
            float dRGBdsh15 = SH_C3[6] * x * (xx - 3.f * yy);
            dsh[9] = dRGBdsh9 * dL_dRGB;
            dsh[10] = dRGBdsh10 * dL_dRGB;
            dsh[11] = dRGBdsh11 * dL_dRGB;
            dsh[12] = dRGBdsh12 * dL_dRGB;
            dsh[13] = dRGBdsh13 * dL_dRGB;
            dsh[14] = dRGBdsh14 * dL_dRGB;
            dsh[15] = dRGBdsh15 * dL_dRGB;
            dRGBdx = dRGBdx + (SH_C3[0] * sh[9] * 3.f * 2.f * xy + SH_C3[1] * sh[10] * yz +
                               SH_C3[2] * sh[11] * -2.f * xy + SH_C3[3] * sh[12] * -3.f * 2.f * xz +
                               SH_C3[4] * sh[13] * (-3.f * xx + 4.f * zz - yy) + SH_C3[5] * sh[14] * 2.f * xz +
                               SH_C3[6] * sh[15] * 3.f * (xx - yy));
            dRGBdy = dRGBdy + (SH_C3[0] * sh[9] * 3.f * (xx - yy) + SH_C3[1] * sh[10] * xz +
                               SH_C3[2] * sh[11] * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * sh[12] * -3.f * 2.f * yz +
                               SH_C3[4] * sh[13] * -2.f * xy + SH_C3[5] * sh[14] * -2.f * yz +
                               SH_C3[6] * sh[15] * -3.f * 2.f * xy);
            dRGBdz = dRGBdz + (SH_C3[1] * sh[10] * xy + SH_C3[2] * sh[11] * 4.f * 2.f * yz +
                               SH_C3[3] * sh[12] * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * sh[13] * 4.f * 2.f * xz +
                               SH_C3[5] * sh[14] * (xx - yy));
          }
        }
      }
      vec3 dL_ddir = {dot(dRGBdx, dL_dRGB), dot(dRGBdy, dL_dRGB), dot(dRGBdz, dL_dRGB)};
      vec3 dm = dnormvdv(dir_orig, dL_ddir);
      dL_dmeans[3 * idx + 0] += dm.x;
      dL_dmeans[3 * idx + 1] += dm.y;
      dL_dmeans[3 * idx + 2] += dm.z;
    }

    if (scales) {  // computeCov3D bwd, backward.cu:299-362
      const float* rot = rotations + 4 * idx;
      float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
      mat3 R(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y), 2.f * (x * y + r * z),
             1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x), 2.f * (x * z - r * y), 2.f * (y * z + r * x),
             1.f - 2.f * (x * x + y * y));
      mat3 S(1.0f, 0, 0, 0, 1.0f, 0, 0, 0, 1.0f);
      vec3 s = scale_modifier * vec3{scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]};
      S[0][0] = s.x;
      S[1][1] = s.y;
      S[2][2] = s.z;
      mat3 Mm = S * R;
      const float* d3 = dL_dcov + 6 * (size_t)idx;
      mat3 dL_dSigma(d3[0], 0.5f * d3[1], 0.5f * d3[2], 0.5f * d3[1], d3[3], 0.5f * d3[4], 0.5f * d3[2],
                     0.5f * d3[4], d3[5]);
      mat3 dL_dM = 2.0f * Mm * dL_dSigma;
      mat3 Rt = transpose(R);
      mat3 dL_dMt = transpose(dL_dM);
      auto col = [](const mat3& A, int c) { return vec3{A[c][0], A[c][1], A[c][2]}; };
      dL_dscale[3 * idx + 0] = dot(col(Rt, 0), col(dL_dMt, 0));
      dL_dscale[3 * idx + 1] = dot(col(Rt, 1), col(dL_dMt, 1));
      dL_dscale[3 * idx + 2] = dot(col(Rt, 2), col(dL_dMt, 2));
      for (int rr = 0; rr < 3; rr++) {
        dL_dMt[0][rr] *= s.x;
        dL_dMt[1][rr] *= s.y;
        dL_dMt[2][rr] *= s.z;
      }
      float qx = 2 * z * (dL_dMt[0][1] - dL_dMt[1][0]) + 2 * y * (dL_dMt[2][0] - dL_dMt[0][2]) +
                 2 * x * (dL_dMt[1][2] - dL_dMt[2][1]);
      float qy = 2 * y * (dL_dMt[1][0] + dL_dMt[0][1]) + 2 * z * (dL_dMt[2][0] + dL_dMt[0][2]) +
                 2 * r * (dL_dMt[1][2] - dL_dMt[2][1]) - 4 * x * (dL_dMt[2][2] + dL_dMt[1][1]);
      float qz = 2 * x * (dL_dMt[1][0] + dL_dMt[0][1]) + 2 * r * (dL_dMt[2][0] - dL_dMt[0][2]) +
                 2 * z * (dL_dMt[1][2] + dL_dMt[2][1]) - 4 * y * (dL_dMt[2][2] + dL_dMt[0][0]);
      float qw = 2 * r * (dL_dMt[0][1] - dL_dMt[1][0]) + 2 * x * (dL_dMt[2][0] + dL_dMt[0][2]) +
                 2 * y * (dL_dMt[1][2] + dL_dMt[2][1]) - 4 * z * (dL_dMt[1][1] + dL_dMt[0][0]);
      dL_drot[4 * idx + 0] = qx;
      dL_drot[4 * idx + 1] = qy;
      dL_drot[4 * idx + 2] = qz;
      dL_drot[4 * idx + 3] = qw;
    }

    // backward.cu:423-457 -- the reference's (non-analytic) projection-matrix gradient, double arithmetic
    // rounded to float per contribution, plus the depth term of the view-matrix gradient.
    const float lastcol_element = (m_hom.x * W * g2x + m_hom.y * H * g2y) * m_w * m_w;
    dL_dproj_acc[0] += (float)(0.5 * g2x * m.x * W * m_w);
    dL_dproj_acc[1] += (float)(0.5 * g2y * m.x * H * m_w);
    dL_dproj_acc[3] += (float)(-0.5 * lastcol_element);
    dL_dproj_acc[4] += (float)(0.5 * g2x * m.y * W * m_w);
    dL_dproj_acc[5] += (float)(0.5 * g2y * m.y * H * m_w);
    dL_dproj_acc[7] += (float)(-0.5 * lastcol_element);
    dL_dproj_acc[8] += (float)(0.5 * g2x * m.z * W * m_w);
    dL_dproj_acc[9] += (float)(0.5 * g2y * m.z * H * m_w);
    dL_dproj_acc[11] += (float)(-0.5 * lastcol_element);
    dL_dproj_acc[12] += (float)(0.5 * g2x * W * m_w);
    dL_dproj_acc[13] += (float)(0.5 * g2y * H * m_w);
    dL_dproj_acc[15] += (float)(-0.5 * lastcol_element);
    dL_dview_matrix_acc[2] += dL_ddepth[idx] * m.x;
    dL_dview_matrix_acc[6] += dL_ddepth[idx] * m.y;
    dL_dview_matrix_acc[10] += dL_ddepth[idx] * m.z;
    dL_dview_matrix_acc[14] += dL_ddepth[idx];
#ifdef _OPENMP
    if (g_accum_f32)
      for (int i = 0; i < 16; i++) {
        dL_dview_matrix_acc[i] = (double)(float)dL_dview_matrix_acc[i];
        dL_dproj_acc[i] = (double)(float)dL_dproj_acc[i];
      }
#endif
  }
  }
#ifdef _OPENMP
  for (int i = 0; i < 16; i++) {
    double v = 0.0, pj = 0.0;
    for (int c = 0; c < 2 * nchunks; c++) {
      v += part_view[(size_t)c * 16 + i];
      if (g_accum_f32) v = (double)(float)v;
    }
    for (int c = 0; c < nchunks; c++) {
      pj += part_proj[(size_t)c * 16 + i];
      if (g_accum_f32) pj = (double)(float)pj;
    }
    dL_dview_matrix[i] += (float)v;
    dL_dproj[i] += (float)pj;
  }
#endif
}

// Pixels whose ORACLE traversal has a (pixel, Gaussian) pair within a small margin of one of the reference's three
// thresholds (power > 0, alpha < 1/255, T (1 - alpha) < 1e-4: forward.cu:356-368).  exp() differs by an ulp or two
// between glibc, CUDA libdevice and the gfx950 v_exp_f32, so such a pair may legitimately fall on either side; the
// parity tests hold every other pixel to the 1e-4 bar and count these.  Same rule as tests/helpers.unstable_pixels
// (numpy, small scenes); this one is usable at BASELINE sizes.  flag is [H*W], zero-initialised by the caller.
void dgs_oracle_unstable(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* means2D,
                         const float* conic_opacity, float alpha_tol, float power_tol, float T_tol, uint8_t* flag) {
  const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
  for (int ty = 0; ty < gy; ty++)
    for (int tx = 0; tx < gx; tx++) {
      const uint32_t r0 = ranges[2 * (ty * gx + tx)], r1 = ranges[2 * (ty * gx + tx) + 1];
      for (int ly = 0; ly < BLOCK_Y; ly++)
        for (int lx = 0; lx < BLOCK_X; lx++) {
          const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
          if (!(px < W && py < H)) continue;
          const float pfx = (float)px, pfy = (float)py;
          float T = 1.0f;
          bool near = false;
          float T_relerr = 0.0f;  // accumulated relative uncertainty of T
          for (uint32_t s = r0; s < r1 && !near; s++) {
            const uint32_t g = point_list[s];
            const float dx = means2D[2 * g] - pfx, dy = means2D[2 * g + 1] - pfy;
            const float* co = conic_opacity + 4 * (size_t)g;
            const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
            const float alpha = std::min(0.99f, co[3] * std::exp(power));
            // fp32 evaluation uncertainty of `power`: a few ulps of its largest term (for an elongated splat far from its
            // centre the three terms are ~1e3 and cancel to ~-5, so two correct fp32 evaluation orders differ by ~1e-4);
            // it carries over to alpha = opacity * exp(power) as a relative error
            const float perr = 5e-7f * (0.5f * std::fabs(co[0] * dx * dx) + 0.5f * std::fabs(co[2] * dy * dy) +
                                        std::fabs(co[1] * dx * dy));
            if (std::fabs(alpha - 1.0f / 255.0f) < alpha_tol + alpha * perr || std::fabs(power) < power_tol + perr)
              near = true;
            if (power > 0.0f || alpha < 1.0f / 255.0f) continue;
            const float test_T = T * (1 - alpha);
            if (alpha < 0.99f) T_relerr += alpha * perr / (1 - alpha);
            if (std::fabs(test_T - 0.0001f) < T_tol + test_T * (T_relerr + 1e-6f)) near = true;
            if (test_T < 0.0001f) break;
            T = test_T;
          }
          if (near) flag[(size_t)W * py + px] = 1;
        }
    }
}

// checkFrustum / markVisible (rasterizer_impl.cu:54-66,141-153)
void dgs_oracle_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present) {
  for (int idx = 0; idx < P; idx++) {
    vec3 p = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    present[idx] = transformPoint4x3(p, viewmatrix).z > 0.2f;
  }
}

}  // extern "C"
