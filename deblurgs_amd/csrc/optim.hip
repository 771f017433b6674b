// optim.hip -- the per-Gaussian work that follows the rasteriser backward in a training iteration:
//   * multi-tensor Adam: every parameter group of the cloud in ONE launch (reference: torch.optim.Adam(lr=0, eps=1e-15)
//     built in scene/gaussian_model.py:170-195 and stepped in train.py:203-208, i.e. 6 tensors x ~10 elementwise
//     launches); HBM-bound, 28 bytes per parameter float (read p, g, m, v; write p, m, v);
//   * densify_and_prune (scene/gaussian_model.py:389-448): clone / split / prune decisions, the destination offsets of
//     every survivor and the compaction of the six parameter tensors plus both Adam moments in three small launches
//     and three scans, instead of ~100 boolean-mask / cat / index launches with optimiser-state surgery in Python.
// Build with -ffp-contract=off: each statement rounds like the torch elementwise op it restates.
#include "dgs_common.h"

namespace {

// ------------------------------------------------------------------------------------------------ Adam
struct AdamArgs {
  float* param[DGS_ADAM_MAX_GROUPS];
  const float* grad[DGS_ADAM_MAX_GROUPS];
  float* m[DGS_ADAM_MAX_GROUPS];
  float* v[DGS_ADAM_MAX_GROUPS];
  uint64_t numel[DGS_ADAM_MAX_GROUPS];
  uint32_t block_end[DGS_ADAM_MAX_GROUPS];  // inclusive prefix of blocks per group
  float neg_step_size[DGS_ADAM_MAX_GROUPS];  // -(lr / (1 - beta1^step))
  float bc2_sqrt[DGS_ADAM_MAX_GROUPS];       // sqrt(1 - beta2^step)
  uint8_t vec4[DGS_ADAM_MAX_GROUPS];         // all four pointers 16-byte aligned: float4 accesses
  uint8_t slot[DGS_ADAM_MAX_GROUPS];         // index of the group in the caller's array (dev_scalars is laid out by it)
  int n;
  float beta2, w1, w2, eps, clip;  // w1 = 1 - beta1, w2 = 1 - beta2
  const uint32_t* skip;            // optional device word: non-zero = leave everything untouched
  const float* dev_scalars;        // optional [2 n]: (neg_step_size, bc2_sqrt) per group read from device memory
};

constexpr int ADAM_THREADS = 256;
constexpr int ADAM_PER_BLOCK = ADAM_THREADS * 4;

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a, float neg_step_size,
                                         float bc2_sqrt) {
  if (a.clip > 0.0f) g = fminf(a.clip, fmaxf(-a.clip, g));   // clip_grad_value_ (train.py:204-205)
  m = m + a.w1 * (g - m);                                     // exp_avg.lerp_(grad, 1 - beta1)
  v = v * a.beta2;                                            // exp_avg_sq.mul_(beta2)
  v = v + (a.w2 * g) * g;                                     //   .addcmul_(grad, grad, value = 1 - beta2)
  const float denom = sqrtf(v) / bc2_sqrt + a.eps;            // (exp_avg_sq.sqrt() / sqrt(bc2)).add_(eps)
  p = p + neg_step_size * (m / denom);                        // param.addcdiv_(exp_avg, denom, value = -step_size)
}

__global__ void __launch_bounds__(ADAM_THREADS) adam_kernel(AdamArgs a) {
  if (a.skip != nullptr && a.skip[0] != 0u) return;   // the gradients come from a truncated (overflowed) forward
  int gi = 0;
#pragma unroll
  for (int i = 0; i < DGS_ADAM_MAX_GROUPS - 1; i++)
    if (i < a.n - 1 && blockIdx.x >= a.block_end[i]) gi = i + 1;
  const uint32_t first = gi == 0 ? 0u : a.block_end[gi - 1];
  const uint64_t base = (uint64_t)(blockIdx.x - first) * ADAM_PER_BLOCK + (uint64_t)threadIdx.x * 4;
  const uint64_t n = a.numel[gi];
  if (base >= n) return;
  // graph replay: this step's bias-corrected step size and sqrt(1 - beta2^t) come from device memory (the host wrote
  // them, computed exactly as below in dgs_adam_step, before launching the graph)
  const float nss = a.dev_scalars != nullptr ? a.dev_scalars[2 * a.slot[gi]] : a.neg_step_size[gi];
  const float bcs = a.dev_scalars != nullptr ? a.dev_scalars[2 * a.slot[gi] + 1] : a.bc2_sqrt[gi];
  float* p = a.param[gi] + base;
  const float* g = a.grad[gi] + base;
  float* m = a.m[gi] + base;
  float* v = a.v[gi] + base;
  if (base + 4 <= n && a.vec4[gi]) {
    float4 P4 = *reinterpret_cast<float4*>(p);
    const float4 G4 = *reinterpret_cast<const float4*>(g);
    float4 M4 = *reinterpret_cast<float4*>(m);
    float4 V4 = *reinterpret_cast<float4*>(v);
    adam_one(P4.x, G4.x, M4.x, V4.x, a, nss, bcs);
    adam_one(P4.y, G4.y, M4.y, V4.y, a, nss, bcs);
    adam_one(P4.z, G4.z, M4.z, V4.z, a, nss, bcs);
    adam_one(P4.w, G4.w, M4.w, V4.w, a, nss, bcs);
    *reinterpret_cast<float4*>(p) = P4;
    *reinterpret_cast<float4*>(m) = M4;
    *reinterpret_cast<float4*>(v) = V4;
  } else {
    for (int i = 0; i < 4 && base + i < n; i++) {
      float pp = p[i], mm = m[i], vv = v[i];
      adam_one(pp, g[i], mm, vv, a, nss, bcs);
      p[i] = pp;
      m[i] = mm;
      v[i] = vv;
    }
  }
}

// --------------------------------------------------------------------------------------- densification
// flags / offsets are [4][P]: 0 keep (original survives), 1 clone survives, 2 split survives (each of its two
// children), 3 split-selected regardless of opacity (indexes the caller's normal samples the way the reference
// draws them, one row per selected Gaussian and copy, before its opacity prune)
__global__ void __launch_bounds__(256)
densify_flags_kernel(int P, const float* __restrict__ accum, const float* __restrict__ denom,
                     const float* __restrict__ scaling, const float* __restrict__ opacity, float grad_thr,
                     float size_thr, float min_opacity, float scale_lb, int iso, uint32_t* __restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  float g = accum[i] / denom[i];            // gaussian_model.py:437-438
  if (g != g) g = 0.0f;
  const float s0 = expf(scaling[3 * i]) + scale_lb, s1 = iso ? s0 : expf(scaling[3 * i + 1]) + scale_lb,
              s2 = iso ? s0 : expf(scaling[3 * i + 2]) + scale_lb;
  const float smax = fmaxf(s0, fmaxf(s1, s2));
  const bool clone_sel = (fabsf(g) >= grad_thr) && (smax <= size_thr);   // :421-424
  const bool split_sel = (g >= grad_thr) && (smax > size_thr);           // :394-399
  const float op = fminf(1.0f, fmaxf(0.0f, opacity[i]));
  const bool alive = !(op < min_opacity);                                // :442-444
  flags[i] = (!split_sel && alive) ? 1u : 0u;
  flags[(size_t)P + i] = (clone_sel && alive) ? 1u : 0u;
  flags[2 * (size_t)P + i] = (split_sel && alive) ? 1u : 0u;
  flags[3 * (size_t)P + i] = split_sel ? 1u : 0u;
}

struct CloudPtrs {
  float* t[6];       // xyz, f_dc, f_rest, opacity, scaling, rotation
  float* m[6];
  float* v[6];
};

__device__ __forceinline__ void field_of(int e, int n_rest, int& f, int& off, int& len) {
  // element e of the concatenated row [xyz 3 | f_dc 3 | f_rest n_rest | opacity 1 | scaling 3 | rotation 4]
  if (e < 3) { f = 0; off = e; len = 3; return; }
  e -= 3;
  if (e < 3) { f = 1; off = e; len = 3; return; }
  e -= 3;
  if (e < n_rest) { f = 2; off = e; len = n_rest; return; }
  e -= n_rest;
  if (e < 1) { f = 3; off = 0; len = 1; return; }
  e -= 1;
  if (e < 3) { f = 4; off = e; len = 3; return; }
  e -= 3;
  f = 5; off = e; len = 4;
}

// One thread per (source Gaussian, row element): copies the element (and its two Adam moments) to the original's
// new slot, to the clone's slot and -- transformed for xyz / scaling -- to the two children's slots.
// New order = the reference's: [surviving originals | clones | children copy 0 | children copy 1].
__global__ void __launch_bounds__(256)
densify_apply_kernel(int P, int n_rest, uint32_t n_keep, uint32_t n_clone, uint32_t n_split, uint32_t m_all,
                     const uint32_t* __restrict__ flags, const uint32_t* __restrict__ offs, CloudPtrs src,
                     CloudPtrs dst, const float* __restrict__ noise, float scale_lb, int iso) {
  const int E = 14 + n_rest;
  const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (tid >= (uint64_t)P * E) return;
  const int i = (int)(tid / E), e = (int)(tid - (uint64_t)i * E);
  int f, off, len;
  field_of(e, n_rest, f, off, len);
  const size_t so = (size_t)i * len + off;
  const float val = src.t[f][so];
  if (flags[i]) {
    const size_t d = (size_t)offs[i] * len + off;
    dst.t[f][d] = val;
    dst.m[f][d] = src.m[f] ? src.m[f][so] : 0.0f;
    dst.v[f][d] = src.v[f] ? src.v[f][so] : 0.0f;
  }
  if (flags[(size_t)P + i]) {   // clone: same parameters, zero moments (cat_tensors_to_optimizer, :366-387)
    const size_t d = ((size_t)n_keep + offs[(size_t)P + i]) * len + off;
    dst.t[f][d] = val;
    dst.m[f][d] = 0.0f;
    dst.v[f][d] = 0.0f;
  }
  if (flags[2 * (size_t)P + i]) {
    const uint32_t r = offs[2 * (size_t)P + i], ra = offs[3 * (size_t)P + i];
#pragma unroll
    for (int c = 0; c < 2; c++) {
      float out = val;
      if (f == 0 || f == 4) {
        const float* sc = src.t[4] + 3 * (size_t)i;
        const float s0 = expf(sc[0]) + scale_lb, s1 = iso ? s0 : expf(sc[1]) + scale_lb,
                    s2 = iso ? s0 : expf(sc[2]) + scale_lb;   // get_scaling: column 0 for all three when isotropic
        if (f == 4) {
          // scaling_inverse_activation(get_scaling / (0.8 N)), N = 2 (:405; LowerBoundLog clamps at eps = 0.001)
          const float s = (off == 0 ? s0 : off == 1 ? s1 : s2) / 1.6f;
          out = logf(fmaxf(s - scale_lb, 0.001f));
        } else {
          const float* z = noise + 3 * ((size_t)c * m_all + ra);
          const float x0 = s0 * z[0], x1 = s1 * z[1], x2 = s2 * z[2];    // torch.normal(0, stds)
          const float* q = src.t[5] + 4 * (size_t)i;                     // build_rotation normalises (general_utils:117)
          const float nrm = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
          const float r_ = q[0] / nrm, x = q[1] / nrm, y = q[2] / nrm, zq = q[3] / nrm;
          float R0, R1, R2;
          if (off == 0) { R0 = 1.0f - 2.0f * (y * y + zq * zq); R1 = 2.0f * (x * y - r_ * zq); R2 = 2.0f * (x * zq + r_ * y); }
          else if (off == 1) { R0 = 2.0f * (x * y + r_ * zq); R1 = 1.0f - 2.0f * (x * x + zq * zq); R2 = 2.0f * (y * zq - r_ * x); }
          else { R0 = 2.0f * (x * zq - r_ * y); R1 = 2.0f * (y * zq + r_ * x); R2 = 1.0f - 2.0f * (x * x + y * y); }
          out = (R0 * x0 + R1 * x1 + R2 * x2) + val;                      // bmm(rots, samples) + xyz (:403)
        }
      }
      const size_t d = ((size_t)n_keep + n_clone + (size_t)c * n_split + r) * len + off;
      dst.t[f][d] = out;
      dst.m[f][d] = 0.0f;
      dst.v[f][d] = 0.0f;
    }
  }
}

// ------------------------------------------------------------------------- rank-ordered sum of received shards
// The second half of a direct reduce-scatter (deblurgs_amd/sharding.py, p2p_allreduce_): this rank holds its own shard
// and one received copy per peer; the reduced shard is their sum IN RANK ORDER -- ((s_0 + s_1) + s_2) + ... -- so that it
// is a fixed function of the inputs, then divided by `divisor` for a mean.  One pass, float4, instead of W - 1 torch
// adds + a clone + a copy (7 launches over 19 MB shards at 8 ranks).
constexpr int RSUM_MAX_W = 64;
__global__ void __launch_bounds__(256)
rank_ordered_sum_kernel(const float* __restrict__ recv, uint64_t stride, float* __restrict__ own, uint64_t n, int W,
                        int rank, float divisor) {
  const uint64_t i4 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  const bool vec = (i4 + 4 <= n) && ((stride & 3) == 0) &&
                   (((reinterpret_cast<uintptr_t>(recv) | reinterpret_cast<uintptr_t>(own)) & 15) == 0);
  if (vec) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < W; s++) {
      const float4 v = (s == rank) ? *reinterpret_cast<const float4*>(own + i4)
                                   : *reinterpret_cast<const float4*>(recv + (uint64_t)s * stride + i4);
      if (s == 0) acc = v;
      else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    }
    if (divisor != 1.0f) { acc.x /= divisor; acc.y /= divisor; acc.z /= divisor; acc.w /= divisor; }
    *reinterpret_cast<float4*>(own + i4) = acc;
  } else {
    for (uint64_t i = i4; i < n && i < i4 + 4; i++) {
      float acc = 0.f;
      for (int s = 0; s < W; s++) {
        const float v = (s == rank) ? own[i] : recv[(uint64_t)s * stride + i];
        acc = (s == 0) ? v : acc + v;
      }
      own[i] = (divisor != 1.0f) ? acc / divisor : acc;
    }
  }
}

}  // namespace

extern int dgs_fail_arg(const char* msg);
extern int dgs_fail_hip(hipError_t e, const char* where);

extern "C" {

static void adam_scalars_of(const DgsAdamGroup& g, double beta1, double beta2, float* neg_step_size, float* bc2_sqrt) {
  // torch/optim/adam.py (_single_tensor_adam): python-float bias corrections, cast to fp32 at the tensor op
  const double bc1 = 1.0 - pow(beta1, (double)g.step);
  const double bc2 = 1.0 - pow(beta2, (double)g.step);
  *neg_step_size = (float)(-(g.lr / bc1));
  *bc2_sqrt = (float)sqrt(bc2);
}

static int adam_step_impl(const DgsAdamGroup* groups, int32_t n_groups, double beta1, double beta2, double eps,
                          double clip_value, const uint32_t* skip_flag, const float* dev_scalars, dgs_stream_t stream) {
  if (n_groups < 0 || n_groups > DGS_ADAM_MAX_GROUPS || (n_groups > 0 && groups == nullptr))
    return dgs_fail_arg("adam_step: 0..DGS_ADAM_MAX_GROUPS groups");
  AdamArgs a;
  a.n = 0;
  uint64_t blocks = 0;
  for (int i = 0; i < n_groups; i++) {
    const DgsAdamGroup& g = groups[i];
    if (g.grad == nullptr || g.numel == 0) continue;   // torch skips parameters without a gradient
    if (g.param == nullptr || g.exp_avg == nullptr || g.exp_avg_sq == nullptr || g.step < 1)
      return dgs_fail_arg("adam_step: null state pointer or step < 1");
    if (((uintptr_t)g.param | (uintptr_t)g.grad | (uintptr_t)g.exp_avg | (uintptr_t)g.exp_avg_sq) & 3)
      return dgs_fail_arg("adam_step: tensors must be 4-byte aligned");
    const int j = a.n++;
    // views into packed buffers may start at any float: those groups take the scalar path
    a.vec4[j] = (((uintptr_t)g.param | (uintptr_t)g.grad | (uintptr_t)g.exp_avg | (uintptr_t)g.exp_avg_sq) & 15) == 0;
    a.param[j] = g.param; a.grad[j] = g.grad; a.m[j] = g.exp_avg; a.v[j] = g.exp_avg_sq;
    a.numel[j] = g.numel;
    blocks += (g.numel + ADAM_PER_BLOCK - 1) / ADAM_PER_BLOCK;
    if (blocks >= (1ull << 31)) return dgs_fail_arg("adam_step: too many elements for one launch");
    a.block_end[j] = (uint32_t)blocks;
    a.slot[j] = (uint8_t)i;
    adam_scalars_of(g, beta1, beta2, &a.neg_step_size[j], &a.bc2_sqrt[j]);
  }
  if (a.n == 0) return DGS_OK;
  for (int j = a.n; j < DGS_ADAM_MAX_GROUPS; j++) {
    a.param[j] = nullptr; a.grad[j] = nullptr; a.m[j] = nullptr; a.v[j] = nullptr;
    a.numel[j] = 0; a.block_end[j] = (uint32_t)blocks; a.neg_step_size[j] = 0.0f; a.bc2_sqrt[j] = 1.0f;
    a.vec4[j] = 0;
    a.slot[j] = 0;
  }
  a.dev_scalars = dev_scalars;
  a.beta2 = (float)beta2;
  a.w1 = (float)(1.0 - beta1);
  a.w2 = (float)(1.0 - beta2);
  a.eps = (float)eps;
  a.clip = (float)clip_value;
  a.skip = skip_flag;
  hipLaunchKernelGGL(adam_kernel, dim3((uint32_t)blocks), dim3(ADAM_THREADS), 0, reinterpret_cast<hipStream_t>(stream), a);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? DGS_OK : dgs_fail_hip(e, "adam_step");
}

int dgs_adam_step(const DgsAdamGroup* groups, int32_t n_groups, double beta1, double beta2, double eps,
                  double clip_value, const uint32_t* skip_flag, dgs_stream_t stream) {
  return adam_step_impl(groups, n_groups, beta1, beta2, eps, clip_value, skip_flag, nullptr, stream);
}
int dgs_adam_step_dev(const DgsAdamGroup* groups, int32_t n_groups, double beta1, double beta2, double eps,
                      double clip_value, const uint32_t* skip_flag, const float* dev_scalars, dgs_stream_t stream) {
  if (dev_scalars == nullptr) return dgs_fail_arg("adam_step_dev: dev_scalars is null");
  return adam_step_impl(groups, n_groups, beta1, beta2, eps, clip_value, skip_flag, dev_scalars, stream);
}
int dgs_adam_scalars(const DgsAdamGroup* groups, int32_t n_groups, double beta1, double beta2, float* out) {
  if (n_groups < 0 || n_groups > DGS_ADAM_MAX_GROUPS || (n_groups > 0 && (groups == nullptr || out == nullptr)))
    return dgs_fail_arg("adam_scalars: bad argument");
  for (int i = 0; i < n_groups; i++) {
    if (groups[i].step < 1) return dgs_fail_arg("adam_scalars: step < 1");
    adam_scalars_of(groups[i], beta1, beta2, &out[2 * i], &out[2 * i + 1]);
  }
  return DGS_OK;
}

int dgs_rank_ordered_sum(const float* recv, uint64_t stride, float* own, uint64_t n, int32_t world, int32_t rank,
                         float divisor, dgs_stream_t stream) {
  if (world < 1 || world > RSUM_MAX_W || rank < 0 || rank >= world || (n > 0 && own == nullptr) ||
      (n > 0 && world > 1 && recv == nullptr) || stride < n || !(divisor > 0.0f))
    return dgs_fail_arg("rank_ordered_sum: bad argument");
  if (n == 0) return DGS_OK;
  const uint64_t blocks = (n + 1023) / 1024;
  if (blocks >= (1ull << 31)) return dgs_fail_arg("rank_ordered_sum: n too large");
  hipLaunchKernelGGL(rank_ordered_sum_kernel, dim3((uint32_t)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     recv, stride, own, n, (int)world, (int)rank, divisor);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DGS_OK : dgs_fail_hip(e, "rank_ordered_sum");
}

size_t dgs_densify_tmp_bytes(int32_t P) { return dgs_scan_tmp_words((uint64_t)(P < 0 ? 0 : P)) * 4 + 256; }

int dgs_densify_plan(int32_t P, const float* xyz_gradient_accum, const float* denom, const float* scaling,
                     const float* opacity, float grad_threshold, float size_threshold, float min_opacity,
                     float scale_lb, int32_t isotropic, uint32_t* flags, uint32_t* offsets, uint32_t* counts_dev,
                     uint32_t* counts_host, void* tmp, dgs_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (P < 0 || counts_host == nullptr) return dgs_fail_arg("densify_plan: bad argument");
  if (P == 0) {
    counts_host[0] = counts_host[1] = counts_host[2] = counts_host[3] = 0;
    return DGS_OK;
  }
  if (xyz_gradient_accum == nullptr || denom == nullptr || scaling == nullptr || opacity == nullptr ||
      flags == nullptr || offsets == nullptr || counts_dev == nullptr || tmp == nullptr)
    return dgs_fail_arg("densify_plan: null pointer");
  hipLaunchKernelGGL(densify_flags_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, xyz_gradient_accum, denom, scaling,
                     opacity, grad_threshold, size_threshold, min_opacity, scale_lb, isotropic != 0 ? 1 : 0, flags);
  hipError_t e = hipGetLastError();
  for (int a = 0; a < 4 && e == hipSuccess; a++) {
    // counts_dev[2a], [2a+1] = the 64-bit total of flag array a
    e = dgs_launch_scan(flags + (size_t)a * P, offsets + (size_t)a * P, (uint64_t)P, reinterpret_cast<uint32_t*>(tmp),
                        counts_dev + 2 * a, s);
    if (e == hipSuccess)
      e = hipMemcpyAsync(counts_host + a, counts_dev + 2 * a, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
  }
  return e == hipSuccess ? DGS_OK : dgs_fail_hip(e, "densify_plan");
}

int dgs_densify_apply(int32_t P, int32_t n_rest, const uint32_t* counts, const uint32_t* flags, const uint32_t* offsets,
                      const DgsCloudArrays* src, const DgsCloudArrays* dst, const float* noise, float scale_lb,
                      int32_t isotropic, dgs_stream_t stream) {
  if (P < 0 || n_rest < 0 || counts == nullptr || src == nullptr || dst == nullptr)
    return dgs_fail_arg("densify_apply: bad argument");
  if (P == 0) return DGS_OK;
  if (flags == nullptr || offsets == nullptr) return dgs_fail_arg("densify_apply: null plan");
  if (counts[2] > 0 && noise == nullptr) return dgs_fail_arg("densify_apply: split needs normal samples");
  CloudPtrs sp, dp;
  for (int f = 0; f < 6; f++) {
    if (f == 2 && n_rest == 0) {
      sp.t[f] = sp.m[f] = sp.v[f] = dp.t[f] = dp.m[f] = dp.v[f] = nullptr;
      continue;
    }
    sp.t[f] = src->param[f]; sp.m[f] = src->exp_avg[f]; sp.v[f] = src->exp_avg_sq[f];
    dp.t[f] = dst->param[f]; dp.m[f] = dst->exp_avg[f]; dp.v[f] = dst->exp_avg_sq[f];
    if (sp.t[f] == nullptr) return dgs_fail_arg("densify_apply: null source tensor");
    if ((counts[0] + counts[1] + counts[2]) > 0 && (dp.t[f] == nullptr || dp.m[f] == nullptr || dp.v[f] == nullptr))
      return dgs_fail_arg("densify_apply: null destination tensor");
  }
  const uint64_t threads = (uint64_t)P * (14 + n_rest);
  hipLaunchKernelGGL(densify_apply_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), P, n_rest, counts[0], counts[1], counts[2], counts[3], flags,
                     offsets, sp, dp, noise, scale_lb, isotropic != 0 ? 1 : 0);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? DGS_OK : dgs_fail_hip(e, "densify_apply");
}

}  // extern "C"
