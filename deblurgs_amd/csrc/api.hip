// api.hip -- the extern "C" surface of libdgs_hip.so (include/dgs_hip.h): argument checks, blob carving,
// stage sequencing (replaces Rasterizer::forward / ::backward, rasterizer_impl.cu:198-463, and the torch glue
// in rasterize_points.cu:35-239), stage timing and the fused loss-gradient image kernel.
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <new>

#include "dgs_common.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, const char* a = "") {
  snprintf(g_err, sizeof(g_err), fmt, a);
  return code;
}
int fail_hip(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
  return DGS_E_HIP;
}

constexpr size_t ALIGN = 256;
size_t up(size_t x) { return (x + ALIGN - 1) / ALIGN * ALIGN; }

int sort_bits_for(int W, int H, int K) {
  const uint32_t T = (uint32_t)((W + DGS_TILE - 1) / DGS_TILE) * (uint32_t)((H + DGS_TILE - 1) / DGS_TILE);
  return 32 + (int)dgs_higher_msb(T * (uint32_t)K);  // rasterizer_impl.cu:306, with K*T tiles
}

// forward_only: the image blob holds the tile ranges alone (final_T / n_contrib are not stored: offsets 0, size 0)
void make_layout(int P, int W, int H, int K, uint64_t R, bool wide_records, DgsLayout* L, bool forward_only = false) {
  const size_t KP = (size_t)K * (size_t)P;
  const size_t N = (size_t)W * (size_t)H;
  const size_t T = (size_t)((W + DGS_TILE - 1) / DGS_TILE) * (size_t)((H + DGS_TILE - 1) / DGS_TILE);
  size_t o = 0;
  L->geom_rows = o;      o += up(KP * sizeof(DgsRow));
  L->cov3D = o;          o += up((size_t)P * 6 * 4);
  L->pre_sigmoid = o;    o += up(KP * 3 * 4);
  L->tiles_touched = o;  o += up(KP * 4);
  L->point_offsets = o;  o += up(KP * 4);
  L->scan_tmp = o;       o += up(dgs_scan_tmp_words(KP) * 4);
  L->num_rendered = o;   o += up(32);
  L->gsort_keys = o;     o += up(KP * 4);
  L->gsort_keys_alt = o; o += up(KP * 4);
  L->gsort_vals = o;     o += up(KP * 4);
  L->gsort_vals_alt = o; o += up(KP * 4);
  L->tt_sorted = o;      o += up(KP * 4);
  L->offs_sorted = o;    o += up(KP * 4);
  L->tt_tight = o;       o += up(KP * 4);
  L->offs_tight = o;     o += up(KP * 4);
  L->gsort_tmp = o;      o += up(dgs_depth_sort_tmp_words(K, (uint32_t)P) * 4);
  L->cull_rec = o;       o += up(KP * 16);
  L->cull_cnt = o;       o += up(KP * 4);
  L->geom_total = o;
  o = 0;
  L->final_T = o;        o += forward_only ? 0 : up((size_t)K * N * 4);
  L->n_contrib = o;      o += forward_only ? 0 : up((size_t)K * N * 4);
  L->ranges = o;         o += up((size_t)K * T * 8);
  L->image_total = o;
  o = 0;
  L->keys_sorted = o;    o += up(R * 8);
  L->point_list = o;     o += up(R * 4);
  L->keys_unsorted = o;  o += up(R * 8);
  L->vals_unsorted = o;  o += up(R * 4);
  L->sort_tmp = o;       o += up(dgs_sort_tmp_words(R) * 4);
  L->binning_total = o;
  L->sort_bits = sort_bits_for(W, H, K);
  L->sort_passes = dgs_sort_num_passes(32, L->sort_bits);
  // compact keys (tile_cull only): tile | Gaussian | emission index in one 64-bit word when the three fit
  const bool compact = !wide_records;   // DgsProblem.wide_records: keep key + value arrays
  const int u_bits = (int)dgs_higher_msb64(R), g_bits = (int)dgs_higher_msb((uint32_t)(P > 0 ? P : 1));
  L->pack_g_shift = 0;
  L->pack_tile_shift = 0;
  if (compact && R > 0 && u_bits + g_bits + (L->sort_bits - 32) <= 64) {
    L->pack_g_shift = u_bits;
    L->pack_tile_shift = u_bits + g_bits;
  }
}

// need_opacity = false: the backward of activated-value problems (raw_params = 0) -- the reference's backward entry point
// is not handed the opacities either (rasterize_points.cu:125-152: it reads them from the geometry state)
int check_problem(const DgsProblem* p, bool need_opacity = true) {
  if (p == nullptr) return fail(DGS_E_ARG, "null DgsProblem");
  if (p->P < 0 || p->W <= 0 || p->H <= 0) return fail(DGS_E_ARG, "bad P/W/H");
  // tile coordinates are packed into 12 bits each by the duplication kernels (and tile counts stay below 2^24)
  if (p->W > 4095 * DGS_TILE || p->H > 4095 * DGS_TILE) return fail(DGS_E_ARG, "W / H above 65520 pixels");
  if (p->K < 1 || p->K > DGS_MAX_K) return fail(DGS_E_ARG, "K must be in [1, DGS_MAX_K]");
  if (p->D < 0 || p->D > 3) return fail(DGS_E_ARG, "SH degree must be 0..3");
  if (p->forward_only != 0 && p->forward_only != 1) return fail(DGS_E_ARG, "forward_only must be 0 or 1");
  if (p->P == 0) return DGS_OK;
  if (p->means3D == nullptr || (need_opacity && p->opacities == nullptr))
    return fail(DGS_E_ARG, "means3D / opacities are null");
  if ((p->shs == nullptr) == (p->colors_precomp == nullptr))
    return fail(DGS_E_ARG, "Please provide excatly one of either SHs or precomputed colors!");
  const bool sr = (p->scales != nullptr) && (p->rotations != nullptr);
  if (((p->scales == nullptr || p->rotations == nullptr) && p->cov3D_precomp == nullptr) ||
      ((p->scales != nullptr || p->rotations != nullptr) && p->cov3D_precomp != nullptr))
    return fail(DGS_E_ARG, "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
  (void)sr;
  if (p->shs != nullptr && p->M < (p->D + 1) * (p->D + 1)) return fail(DGS_E_ARG, "M < (D+1)^2");
  if (p->raw_params != 0 && p->raw_params != 1 && p->raw_params != 3)
    return fail(DGS_E_ARG, "raw_params must be 0, 1 or 3 (1 | isotropic scale)");
  if (p->raw_params) {
    if (p->shs == nullptr || p->scales == nullptr || p->rotations == nullptr)
      return fail(DGS_E_ARG, "raw_params needs shs (dc), scales and rotations");
    if (p->M > 1 && p->shs_rest == nullptr) return fail(DGS_E_ARG, "raw_params with M > 1 needs shs_rest");
  } else if (p->shs_rest != nullptr) {
    return fail(DGS_E_ARG, "shs_rest is only meaningful with raw_params");
  }
  if (p->viewmatrix == nullptr || p->projmatrix == nullptr || p->campos == nullptr || p->bg == nullptr)
    return fail(DGS_E_ARG, "viewmatrix / projmatrix / campos / bg are null");
  if ((uint64_t)p->K * (uint64_t)p->P >= (1ull << 32)) return fail(DGS_E_ARG, "K*P must be < 2^32");
  return DGS_OK;
}

DgsView make_view(const DgsProblem* p) {
  DgsView v;
  v.P = p->P; v.D = p->D; v.M = p->M; v.W = p->W; v.H = p->H; v.K = p->K;
  v.gx = (p->W + DGS_TILE - 1) / DGS_TILE;
  v.gy = (p->H + DGS_TILE - 1) / DGS_TILE;
  v.T = v.gx * v.gy;
  v.tanfovx = p->tanfovx; v.tanfovy = p->tanfovy;
  v.focal_y = p->H / (2.0f * p->tanfovy);  // rasterizer_impl.cu:227-228
  v.focal_x = p->W / (2.0f * p->tanfovx);
  v.scale_modifier = p->scale_modifier;
  v.z_far = p->z_far;
  v.use_sigmoid = p->use_sigmoid; v.prefiltered = p->prefiltered;
  v.tile_cull = p->tile_cull != 0;
  v.raw_params = (p->raw_params & 1) != 0;
  v.iso_scale = (p->raw_params & 2) != 0;
  v.pack_g_shift = 0;      // set from the layout by the callers that know R (tile_cull only)
  v.pack_tile_shift = 0;
  v.scale_lb = p->scale_lb;
  return v;
}

void carve(const DgsProblem* p, const DgsLayout& L, DgsCarve* c) {
  char* g = reinterpret_cast<char*>(p->geom_state);
  char* im = reinterpret_cast<char*>(p->image_state);
  char* b = reinterpret_cast<char*>(p->binning_state);
  c->rows = reinterpret_cast<DgsRow*>(g + L.geom_rows);
  c->cov3D = p->forward_only ? nullptr : reinterpret_cast<float*>(g + L.cov3D);   // (not stored by an inference call)
  c->pre_sigmoid = p->forward_only ? nullptr : reinterpret_cast<float*>(g + L.pre_sigmoid);
  c->tiles_touched = reinterpret_cast<uint32_t*>(g + L.tiles_touched);
  c->point_offsets = reinterpret_cast<uint32_t*>(g + L.point_offsets);
  c->scan_tmp = reinterpret_cast<uint32_t*>(g + L.scan_tmp);
  c->num_rendered = reinterpret_cast<uint32_t*>(g + L.num_rendered);
  c->gsort_keys = reinterpret_cast<uint32_t*>(g + L.gsort_keys);
  c->gsort_keys_alt = reinterpret_cast<uint32_t*>(g + L.gsort_keys_alt);
  c->gsort_vals = reinterpret_cast<uint32_t*>(g + L.gsort_vals);
  c->gsort_vals_alt = reinterpret_cast<uint32_t*>(g + L.gsort_vals_alt);
  c->tt_sorted = reinterpret_cast<uint32_t*>(g + L.tt_sorted);
  c->offs_sorted = reinterpret_cast<uint32_t*>(g + L.offs_sorted);
  c->tt_tight = reinterpret_cast<uint32_t*>(g + L.tt_tight);
  c->offs_tight = reinterpret_cast<uint32_t*>(g + L.offs_tight);
  c->gsort_tmp = reinterpret_cast<uint32_t*>(g + L.gsort_tmp);
  c->cull_rec = reinterpret_cast<uint4*>(g + L.cull_rec);
  c->cull_cnt = reinterpret_cast<uint32_t*>(g + L.cull_cnt);
  c->final_T = p->forward_only ? nullptr : reinterpret_cast<float*>(im + L.final_T);
  c->n_contrib = p->forward_only ? nullptr : reinterpret_cast<uint32_t*>(im + L.n_contrib);
  c->ranges = reinterpret_cast<uint2*>(im + L.ranges);
  c->keys_sorted = b ? reinterpret_cast<uint64_t*>(b + L.keys_sorted) : nullptr;
  c->point_list = b ? reinterpret_cast<uint32_t*>(b + L.point_list) : nullptr;
  c->keys_unsorted = b ? reinterpret_cast<uint64_t*>(b + L.keys_unsorted) : nullptr;
  c->vals_unsorted = b ? reinterpret_cast<uint32_t*>(b + L.vals_unsorted) : nullptr;
  c->sort_tmp = b ? reinterpret_cast<uint32_t*>(b + L.sort_tmp) : nullptr;
}

// ------------------------------------------------------------------------------------------------ context
// Everything that outlives a call lives here, owned by the caller (include/dgs_hip.h, ABI 14): the side stream + events of
// the backward in parts (created on first use, on the device that is current then), its policy, the stage timers.
constexpr int PROF_MAX = 16384;
struct Prof {
  bool on = false;
  int n = 0;
  hipEvent_t* beg = nullptr;   // PROF_MAX each, allocated when the timers are first switched on
  hipEvent_t* end = nullptr;
  int* stage = nullptr;
  int created = 0;
};
}  // namespace

struct DgsContext {
  std::mutex mu;               // one enqueue sequence at a time per context
  DgsContextOptions opt;
  bool side_ready = false;
  int side_device = -1;
  hipStream_t s2 = nullptr;
  hipEvent_t done[DGS_MAX_BWD_PARTS] = {}, join = nullptr;
  Prof prof;
};

namespace {

int prof_begin(DgsContext* ctx, int stage, hipStream_t s) {
  if (ctx == nullptr || !ctx->prof.on) return -1;
  Prof& pr = ctx->prof;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (pr.n >= PROF_MAX) return -1;
  const int i = pr.n++;
  if (i >= pr.created) {
    hipEventCreate(&pr.beg[i]);
    hipEventCreate(&pr.end[i]);
    pr.created = i + 1;
  }
  pr.stage[i] = stage;
  hipEventRecord(pr.beg[i], s);
  return i;
}
void prof_end(DgsContext* ctx, int i, hipStream_t s) {
  if (i >= 0) hipEventRecord(ctx->prof.end[i], s);
}

struct StageTimer {
  DgsContext* ctx;
  int id;
  hipStream_t s;
  StageTimer(DgsContext* c, int stage, hipStream_t st) : ctx(c), id(prof_begin(c, stage, st)), s(st) {}
  ~StageTimer() { prof_end(ctx, id, s); }
};

#define DGS_STAGE(stage_id, where, expr)                                   \
  do {                                                                      \
    hipError_t e__;                                                         \
    {                                                                       \
      StageTimer tm__(p->context, stage_id, s);                                         \
      e__ = (expr);                                                         \
    }                                                                       \
    if (e__ != hipSuccess) return fail_hip(e__, where);                     \
    if (p->debug) {                                                         \
      e__ = hipStreamSynchronize(s);                                        \
      if (e__ != hipSuccess) return fail_hip(e__, where " (debug sync)");   \
    }                                                                       \
  } while (0)

// --------------------------------------------------------------------------- fused loss-gradient image (f1)
// train.py:143-165 with utils/loss_utils.py:17-18 (l1_loss) and :80-93 (batchwise_smoothness_loss):
//   blur = mean_k sub_k;  L = mean|blur - gt| + lambda_t * mean|sub_{k+1} - sub_k|
//   dL/dsub_k = sign(blur-gt)/(E*K) + lambda_t * [sign(sub_k - sub_{k-1}) - sign(sub_{k+1} - sub_k)] / (E*(K-1))
// with E = C*H*W.  One thread per (channel, pixel) element; replaces ~20 elementwise launches and ~1.5 GB of
// traffic between the fused forward and backward.
__device__ __forceinline__ float sgn(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

// Deterministic totals: a block's two sums (fixed-order fp32 trees of non-negative terms) are added as 2^-24 fixed-point
// integers -- integer addition is associative, so the result does not depend on the order in which the blocks arrive,
// unlike float atomics -- and the last block to arrive converts them.  losses is an 8-word work area: [0] L1,
// [1] smoothness, [2..3] / [4..5] the two 64-bit accumulators, [6] arrival counter, [7] sticky "not representable" flag
// (all zeroed by the launcher).  The fixed point covers totals below 2^40 ~ 1.1e12 at a resolution far below the fp32
// rounding of the block sums; a block sum that is NaN / Inf (a diverged render, a NaN in the ground truth) or a total that
// leaves the range sets the flag and both losses come back NaN, as float arithmetic would have reported them.
__device__ __forceinline__ void loss_totals_publish(float a, float c, float* __restrict__ losses, size_t E, int K) {
  constexpr double FX = 16777216.0;   // 2^24
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(losses + 2);
  unsigned int* flag = reinterpret_cast<unsigned int*>(losses + 7);
  const bool fin = __builtin_isfinite(a) && __builtin_isfinite(c) && a < 5.0e11f && c < 5.0e11f;
  if (!fin) {
    atomicOr(flag, 1u);
  } else {
    const unsigned long long ia = (unsigned long long)__double2ll_rn((double)a * FX);
    const unsigned long long ic = (unsigned long long)__double2ll_rn((double)c * FX);
    const unsigned long long oa = atomicAdd(&acc[0], ia), oc = atomicAdd(&acc[1], ic);
    if (oa + ia < oa || oc + ic < oc) atomicOr(flag, 1u);   // the 64-bit accumulator wrapped
  }
  __threadfence();
  const unsigned int ticket = atomicAdd(reinterpret_cast<unsigned int*>(losses + 6), 1u);
  if (ticket == gridDim.x - 1) {
    __threadfence();
    const unsigned long long t0 = atomicAdd(&acc[0], 0ull), t1 = atomicAdd(&acc[1], 0ull);
    const bool bad = atomicOr(flag, 0u) != 0u;
    const float nanv = __int_as_float(0x7fc00000);
    losses[0] = bad ? nanv : (float)(((double)t0 / FX) / (double)E);
    losses[1] = bad ? nanv : ((K > 1) ? (float)(((double)t1 / FX) / ((double)E * (double)(K - 1))) : 0.0f);
  }
}

// MODE 0: blur + loss values (forward).  MODE 1: dL/dsubframes, multiplied by the upstream scalar *scale read
// from device memory (backward; no host sync, no extra elementwise pass over [K,3,H,W]).  MODE 2: both at once.
template <int MODE, int V>  // V = elements per thread (4 -> 16-byte loads/stores when E % 4 == 0, else 1)
__global__ void __launch_bounds__(256)
blur_loss_kernel(const float* __restrict__ sub, const float* __restrict__ gt, int K, size_t E, float lambda_t,
                 const float* __restrict__ lambda_dev, const float* __restrict__ scale, float* __restrict__ blur,
                 float* __restrict__ dsub, float* __restrict__ losses) {
  if (lambda_dev != nullptr) lambda_t = lambda_dev[0];   // graph replay: the scheduled weight lives in device memory
  __shared__ float red[2][4];
  typedef float vec __attribute__((ext_vector_type(V)));
  float l1 = 0.0f, sm = 0.0f;
  // grid-stride loop: the launcher caps the grid (the loss totals cost three same-address atomics per block)
  for (size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * V; e < E; e += (size_t)gridDim.x * 256 * V) {
    auto ld = [](const float* p) { return *reinterpret_cast<const vec*>(p); };
    if (MODE == 0) {
      // forward: ONE pass over the K subframes (sum for the blur and the adjacent differences together), unrolled so
      // that several 16-byte loads per lane are in flight
      vec prev = ld(sub + e);
      vec acc = prev;
#pragma unroll 4
      for (int k = 1; k < K; k++) {
        const vec nxt = ld(sub + (size_t)k * E + e);
        const vec dd = nxt - prev;
        acc += nxt;
#pragma unroll
        for (int i = 0; i < V; i++) sm += fabsf(dd[i]);
        prev = nxt;
      }
      const vec b = acc / (float)K;
      *reinterpret_cast<vec*>(blur + e) = b;
      const vec d = b - ld(gt + e);
#pragma unroll
      for (int i = 0; i < V; i++) l1 += fabsf(d[i]);
    } else {
      vec b;
      if (MODE == 1 && blur != nullptr) {
        b = ld(blur + e);   // backward: the forward's blur is handed back in, no second summation
      } else {
        vec acc = ld(sub + e);
#pragma unroll 4
        for (int k = 1; k < K; k++) acc += ld(sub + (size_t)k * E + e);
        b = acc / (float)K;
        if (MODE == 2) *reinterpret_cast<vec*>(blur + e) = b;
      }
      const vec d = b - ld(gt + e);
      const float up = (scale != nullptr) ? scale[0] : 1.0f;
      const float c_l1 = up / ((float)E * (float)K);
      const float ws = (K > 1) ? up * lambda_t / ((float)E * (float)(K - 1)) : 0.0f;
      vec g_l1;
#pragma unroll
      for (int i = 0; i < V; i++) {
        l1 += fabsf(d[i]);
        g_l1[i] = c_l1 * sgn(d[i]);
      }
      vec prev = ld(sub + e);
      vec s_prev = (vec)(0.0f);  // sign(x_k - x_{k-1})
#pragma unroll 4
      for (int k = 0; k < K; k++) {
        vec s_next = (vec)(0.0f);
        vec nxt = prev;
        if (k + 1 < K) {
          nxt = ld(sub + (size_t)(k + 1) * E + e);
          const vec dd = nxt - prev;
#pragma unroll
          for (int i = 0; i < V; i++) {
            sm += fabsf(dd[i]);
            s_next[i] = sgn(dd[i]);
          }
        }
        *reinterpret_cast<vec*>(dsub + (size_t)k * E + e) = g_l1 + ws * (s_prev - s_next);
        s_prev = s_next;
        prev = nxt;
      }
    }
  }
  if (MODE == 1) return;
  l1 = dgs_wave_sum63(l1);
  sm = dgs_wave_sum63(sm);
  const int lane = dgs_lane(), w = threadIdx.x >> 6;
  if (lane == 63) {
    red[0][w] = l1;
    red[1][w] = sm;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const float c = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    loss_totals_publish(a, c, losses, E, K);
  }
}

// Forward + backward at once (MODE 2's results, bit for bit) for K <= KMAX with the K subframe values of an element
// held in registers: every subframe value is read ONCE (MODE 2 streams the subframes twice: 1.1 GB instead of 0.77 GB
// at the metric configuration).
template <int KMAX, int V>
__global__ void __launch_bounds__(256)
blur_loss_all_kernel(const float* __restrict__ sub, const float* __restrict__ gt, int K, size_t E, float lambda_t,
                     const float* __restrict__ lambda_dev, const float* __restrict__ scale, float* __restrict__ blur,
                     float* __restrict__ dsub, float* __restrict__ losses) {
  if (lambda_dev != nullptr) lambda_t = lambda_dev[0];
  __shared__ float red[2][4];
  typedef float vec __attribute__((ext_vector_type(V)));
  float l1 = 0.0f, sm = 0.0f;
  for (size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * V; e < E; e += (size_t)gridDim.x * 256 * V) {
    vec x[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++)
      if (k < K) x[k] = *reinterpret_cast<const vec*>(sub + (size_t)k * E + e);
    vec acc = x[0];
#pragma unroll
    for (int k = 1; k < KMAX; k++)
      if (k < K) acc += x[k];
    const vec b = acc / (float)K;
    *reinterpret_cast<vec*>(blur + e) = b;
    const vec d = b - *reinterpret_cast<const vec*>(gt + e);
    const float up = (scale != nullptr) ? scale[0] : 1.0f;
    const float c_l1 = up / ((float)E * (float)K);
    const float ws = (K > 1) ? up * lambda_t / ((float)E * (float)(K - 1)) : 0.0f;
    vec g_l1;
#pragma unroll
    for (int i = 0; i < V; i++) {
      l1 += fabsf(d[i]);
      g_l1[i] = c_l1 * sgn(d[i]);
    }
    vec s_prev = (vec)(0.0f);
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
      if (k < K) {
        vec s_next = (vec)(0.0f);
        if (k + 1 < K) {
          const vec dd = x[k + 1 < KMAX ? k + 1 : k] - x[k];
#pragma unroll
          for (int i = 0; i < V; i++) {
            sm += fabsf(dd[i]);
            s_next[i] = sgn(dd[i]);
          }
        }
        *reinterpret_cast<vec*>(dsub + (size_t)k * E + e) = g_l1 + ws * (s_prev - s_next);
        s_prev = s_next;
      }
    }
  }
  l1 = dgs_wave_sum63(l1);
  sm = dgs_wave_sum63(sm);
  const int lane = dgs_lane(), w = threadIdx.x >> 6;
  if (lane == 63) {
    red[0][w] = l1;
    red[1][w] = sm;
  }
  __syncthreads();
  if (threadIdx.x == 0) {   // deterministic fixed-point totals
    const float a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const float c = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    loss_totals_publish(a, c, losses, E, K);
  }
}

// The loss block of ONE RANK of a subframe-sharded view (deblurgs_amd/sharding.py; new work: the reference is
// single-GPU).  The rank holds Kl consecutive subframes of the view's K; `blur` is the view's blur image (mean over all K
// subframes: the ranks' partial sums were all-reduced), prev / next the neighbouring ranks' boundary subframes (NULL at
// the ends of the view).  Same arithmetic as blur_loss_all_kernel for the subframes held here:
//   dL/dsub_k = sign(blur - gt) / (E K) + lambda_t [sign(sub_k - sub_{k-1}) - sign(sub_{k+1} - sub_k)] / (E (K - 1)),
// losses[0] = mean |blur - gt| (the same on every rank), losses[1] = this rank's share of the smoothness value: the
// differences whose LEFT frame it holds, / (E (K - 1)) -- the caller sums the shares over the ranks.
template <int KMAX, int V>
__global__ void __launch_bounds__(256)
blur_loss_slice_kernel(const float* __restrict__ sub, const float* __restrict__ prev, const float* __restrict__ next,
                       const float* __restrict__ blur, const float* __restrict__ gt, int Kl, int K, size_t E,
                       float lambda_t, float* __restrict__ dsub, float* __restrict__ losses) {
  __shared__ float red[2][4];
  typedef float vec __attribute__((ext_vector_type(V)));
  float l1 = 0.0f, sm = 0.0f;
  for (size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * V; e < E; e += (size_t)gridDim.x * 256 * V) {
    vec x[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++)
      if (k < Kl) x[k] = *reinterpret_cast<const vec*>(sub + (size_t)k * E + e);
    const vec d = *reinterpret_cast<const vec*>(blur + e) - *reinterpret_cast<const vec*>(gt + e);
    const float c_l1 = 1.0f / ((float)E * (float)K);
    const float ws = (K > 1) ? lambda_t / ((float)E * (float)(K - 1)) : 0.0f;
    vec g_l1;
#pragma unroll
    for (int i = 0; i < V; i++) {
      l1 += fabsf(d[i]);
      g_l1[i] = c_l1 * sgn(d[i]);
    }
    vec s_prev = (vec)(0.0f);
    if (prev != nullptr) {   // the difference across the lower rank boundary belongs to the rank below: sign only
      const vec dd = x[0] - *reinterpret_cast<const vec*>(prev + e);
#pragma unroll
      for (int i = 0; i < V; i++) s_prev[i] = sgn(dd[i]);
    }
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
      if (k < Kl) {
        vec s_next = (vec)(0.0f);
        const bool inner = k + 1 < Kl;
        if (inner || next != nullptr) {
          const vec nx = inner ? x[k + 1 < KMAX ? k + 1 : k] : *reinterpret_cast<const vec*>(next + e);
          const vec dd = nx - x[k];
#pragma unroll
          for (int i = 0; i < V; i++) {
            sm += fabsf(dd[i]);
            s_next[i] = sgn(dd[i]);
          }
        }
        *reinterpret_cast<vec*>(dsub + (size_t)k * E + e) = g_l1 + ws * (s_prev - s_next);
        s_prev = s_next;
      }
    }
  }
  l1 = dgs_wave_sum63(l1);
  sm = dgs_wave_sum63(sm);
  const int lane = dgs_lane(), w = threadIdx.x >> 6;
  if (lane == 63) {
    red[0][w] = l1;
    red[1][w] = sm;
  }
  __syncthreads();
  if (threadIdx.x == 0) {   // deterministic fixed-point totals
    const float a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const float c = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    loss_totals_publish(a, c, losses, E, K);
  }
}

// train.py:188-193 + scene/gaussian_model.py:456-458 for the K subframes of one step, in subframe order
__global__ void __launch_bounds__(256)
densify_stats_kernel(const float* __restrict__ vgrad, const int32_t* __restrict__ radii, int K, int K_total, int P,
                     float* __restrict__ max_radii2D, float* __restrict__ accum, float* __restrict__ denom,
                     const uint32_t* __restrict__ skip) {
  if (skip != nullptr && skip[0] != 0u) return;
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= P) return;
  float mr = max_radii2D[g], ac = accum[g], dn = denom[g];
  const float inc = (float)(1.0 / (double)K_total);
  for (int k = 0; k < K; k++) {
    const size_t o = (size_t)k * P + g;
    const int r = radii[o];
    if (r > 0) {
      mr = fmaxf(mr, (float)r);
      const float gx = vgrad[3 * o], gy = vgrad[3 * o + 1];
      ac += sqrtf(gx * gx + gy * gy);
      dn += inc;
    }
  }
  max_radii2D[g] = mr;
  accum[g] = ac;
  denom[g] = dn;
}

}  // namespace

hipError_t dgs_launch_blur_loss(const float* sub, const float* gt, int K, int C, int HW, float lambda_t,
                                const float* lambda_dev, const float* scale, float* blur, float* dsub, float* losses,
                                hipStream_t s) {
  const size_t E = (size_t)C * HW;
  if (losses != nullptr) {
    hipError_t e = dgs_launch_clear_words(reinterpret_cast<uint32_t*>(losses), 8, s);   // results + accumulators + arrival counter
    if (e != hipSuccess) return e;
  }
  const bool v4 = (E % 4 == 0) && ((reinterpret_cast<uintptr_t>(sub) | reinterpret_cast<uintptr_t>(gt) |
                                    reinterpret_cast<uintptr_t>(blur) | reinterpret_cast<uintptr_t>(dsub)) % 16 == 0);
  const size_t per = v4 ? 4 : 1;
  // at most DGS_LOSS_BLOCKS blocks (2 per CU; every thread keeps K 16-byte loads in flight): enough to saturate HBM, few
  // enough that the per-block same-address atomics of the loss totals (~58 ns per block: 0.35 ms with 6075 blocks at
  // 1080p, still 0.11 ms of the 800x800 case's kernel with 1875) disappear.  Measured 2048 / 1024 / 512 blocks:
  // 1.204 / 1.157 / 1.144 ms per cfg2 step, 11.99 / 12.01 / 11.91 ms at the metric config.  The totals depend on the
  // grid, which is a function of E only -- still bitwise reproducible
  const size_t want = (E / per + 255) / 256;
#ifndef DGS_LOSS_BLOCKS
#define DGS_LOSS_BLOCKS 512
#endif
  const dim3 grid((uint32_t)(want < DGS_LOSS_BLOCKS ? (want == 0 ? 1 : want) : DGS_LOSS_BLOCKS));
#define DGS_BL(MODE)                                                                                              \
  do {                                                                                                            \
    if (v4)                                                                                                       \
      hipLaunchKernelGGL((blur_loss_kernel<MODE, 4>), grid, dim3(256), 0, s, sub, gt, K, E, lambda_t, lambda_dev, scale, blur, \
                         dsub, losses);                                                                           \
    else                                                                                                          \
      hipLaunchKernelGGL((blur_loss_kernel<MODE, 1>), grid, dim3(256), 0, s, sub, gt, K, E, lambda_t, lambda_dev, scale, blur, \
                         dsub, losses);                                                                           \
  } while (0)
  if (dsub == nullptr)
    DGS_BL(0);
  else if (losses == nullptr)
    DGS_BL(1);
  else if (v4 && K <= 16)
    hipLaunchKernelGGL((blur_loss_all_kernel<16, 4>), grid, dim3(256), 0, s, sub, gt, K, E, lambda_t, lambda_dev, scale, blur, dsub,
                       losses);
  else if (v4 && K <= 32)
    hipLaunchKernelGGL((blur_loss_all_kernel<32, 2>), grid, dim3(256), 0, s, sub, gt, K, E, lambda_t, lambda_dev, scale, blur, dsub,
                       losses);
  else
    DGS_BL(2);
#undef DGS_BL
  return hipGetLastError();
}

// error reporting for the other translation units of the library (optim.hip)
int dgs_fail_arg(const char* msg) { return fail(DGS_E_ARG, "%s", msg); }
int dgs_fail_hip(hipError_t e, const char* where) { return fail_hip(e, where); }

extern "C" {

int dgs_abi_version(void) { return DGS_ABI_VERSION; }
const char* dgs_last_error(void) { return g_err; }

size_t dgs_geom_state_bytes(int32_t P, int32_t K) {
  DgsLayout L;
  make_layout(P, 16, 16, K, 0, false, &L);
  return L.geom_total;
}
size_t dgs_image_state_bytes(int32_t W, int32_t H, int32_t K) {
  DgsLayout L;
  make_layout(0, W, H, K, 0, false, &L);
  return L.image_total;
}
size_t dgs_image_state_bytes_forward_only(int32_t W, int32_t H, int32_t K) {
  DgsLayout L;
  make_layout(0, W, H, K, 0, false, &L, true);
  return L.image_total;
}
size_t dgs_binning_state_bytes(uint64_t R, int32_t W, int32_t H, int32_t K) {
  DgsLayout L;
  make_layout(0, W, H, K, R, false, &L);
  return L.binning_total;
}
size_t dgs_backward_scratch_bytes(uint64_t R, int32_t P, int32_t K) {
  // contribution rows [R] + their per-pair totals by natural index [K*P] (48 bytes each) + pose-gradient partials (f64)
  return up((size_t)R * DGS_CONTRIB_F * 4) + up((size_t)K * (size_t)P * DGS_SUMS_F * 4) +
         up((size_t)dgs_geometry_bwd_blocks(P) * (size_t)K * 24 * 8) + ALIGN;
}
int dgs_layout(int32_t P, int32_t W, int32_t H, int32_t K, uint64_t R, int32_t wide_records, DgsLayout* out) {
  if (out == nullptr) return fail(DGS_E_ARG, "null DgsLayout");
  make_layout(P, W, H, K, R, wide_records != 0, out);
  return DGS_OK;
}
int dgs_backward_scratch_layout(uint64_t R, int32_t P, int32_t K, size_t* sums_offset, size_t* partials_offset) {
  const size_t so = up((size_t)R * DGS_CONTRIB_F * 4);
  if (sums_offset) *sums_offset = so;
  if (partials_offset) *partials_offset = so + up((size_t)K * (size_t)P * DGS_SUMS_F * 4);
  return DGS_OK;
}

// order the (k, Gaussian) pairs by (k, depth bits, index): segmented stable sort of the depth keys preprocess wrote;
// the result (flat indices) lands in c.gsort_vals
// (tile_cull: its last pass also writes the visibility flags in that order)
static hipError_t launch_depth_order(const DgsProblem* p, const DgsCarve& c, hipStream_t s, bool carry_counts = false) {
  const bool cull = p->tile_cull != 0;
  // (status word [6]: "a visible depth key needs more than 27 bits", zeroed with the other status words before preprocess)
  return dgs_launch_depth_sort(c.gsort_keys, c.gsort_keys_alt, c.gsort_vals, c.gsort_vals_alt, p->K, (uint32_t)p->P,
                               c.gsort_tmp, cull ? c.tt_sorted : nullptr, c.num_rendered + 6, s,
                               carry_counts ? c.cull_cnt : nullptr, carry_counts ? c.tt_tight : nullptr,
                               /* drop_invisible = */ cull);
}

// tile_cull: per-slot test in natural order (independent of the depth order), then records and counts into depth order and
// the scan of the counts (total -> status words [2], [3])
static hipError_t launch_tile_cull(const DgsView& v, const DgsCarve& c, hipStream_t s) {
  hipError_t e = dgs_launch_cull_count(v, c, s);
  if (e != hipSuccess) return e;
  return dgs_launch_cull_offsets(v, c, c.num_rendered + 2, s);
}
// the surviving-tile counts ride through the depth sort in the top byte of its 32-bit values when the flat (subframe,
// Gaussian) index fits 24 bits: COUNT first (natural order), then the depth order, then only the scan
static bool counts_ride_with_the_sort(const DgsProblem* p) { return (uint64_t)p->K * (uint64_t)p->P <= (1ull << 24); }

// copy_count = false: the caller (the capacity-mode forward) publishes the count words itself, from its finalize kernel
static int forward_geometry_impl(const DgsProblem* p, const DgsForwardOut* out, dgs_stream_t stream, bool copy_count) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rc = check_problem(p);
  if (rc != DGS_OK) return rc;
  if (out == nullptr || out->num_rendered_host == nullptr)
    return fail(DGS_E_ARG, "DgsForwardOut: num_rendered_host is null");
  if (p->P == 0) {  // rasterize_points.cu:85 -- nothing to launch
    out->num_rendered_host[0] = 0;
    out->num_rendered_host[1] = 0;
    return DGS_OK;
  }
  if (out->radii == nullptr) return fail(DGS_E_ARG, "DgsForwardOut: radii is null");
  DgsLayout L;
  make_layout(p->P, p->W, p->H, p->K, 0, p->wide_records != 0, &L, p->forward_only != 0);
  if (p->geom_state == nullptr || p->geom_bytes < L.geom_total) return fail(DGS_E_CAPACITY, "geom_state too small");
  DgsCarve c;
  carve(p, L, &c);
  const DgsView v = make_view(p);
  hipError_t e = dgs_launch_clear_words(c.num_rendered, 8, s);   // (a kernel, not a memset node: see finalize_count_kernel)
  if (e != hipSuccess) return fail_hip(e, "clear status");
  DGS_STAGE(DGS_STAGE_PREPROCESS, "preprocess", dgs_launch_preprocess(*p, v, c, out->radii, s));
  if (!v.tile_cull) {
    DGS_STAGE(DGS_STAGE_SCAN, "scan",
              dgs_launch_scan(c.tiles_touched, c.point_offsets, (uint64_t)p->K * p->P, c.scan_tmp, c.num_rendered, s));
    // two words: R and the high half of the 64-bit total (non-zero = the u32 duplicate offsets overflowed)
    if (copy_count)
      e = hipMemcpyAsync(out->num_rendered_host, c.num_rendered, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
  } else {
    // tile_cull: R is the number of surviving duplicates: per-slot test in natural order, depth ordering (which also
    // lays the per-pair records and counts out in its order), scan of the counts.  Status words [0], [1] stay 0.
    if (counts_ride_with_the_sort(p)) {
      DGS_STAGE(DGS_STAGE_TILE_CULL, "tile cull", dgs_launch_cull_count(v, c, s));
      DGS_STAGE(DGS_STAGE_DEPTH_ORDER, "depth order", launch_depth_order(p, c, s, true));
      DGS_STAGE(DGS_STAGE_SCAN, "scan", dgs_launch_cull_offsets(v, c, c.num_rendered + 2, s, true));
    } else {
      DGS_STAGE(DGS_STAGE_DEPTH_ORDER, "depth order", launch_depth_order(p, c, s));
      DGS_STAGE(DGS_STAGE_TILE_CULL, "tile cull", launch_tile_cull(v, c, s));
    }
    // two words: R and the high half of the 64-bit total (non-zero = the u32 duplicate offsets overflowed)
    if (copy_count)
      e = hipMemcpyAsync(out->num_rendered_host, c.num_rendered + 2, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
  }
  if (e != hipSuccess) return fail_hip(e, "copy num_rendered");
  return DGS_OK;
}

int dgs_forward_geometry(const DgsProblem* p, const DgsForwardOut* out, dgs_stream_t stream) {
  return forward_geometry_impl(p, out, stream, true);
}

// R = the exact duplicate count (two-phase forward), or the capacity of the duplicate arrays when `speculative`
// (then the kernels read the count from c.num_rendered[4], set by finalize_count)
static int forward_render_impl(const DgsProblem* p, const DgsForwardOut* out, uint32_t R, bool speculative,
                               hipStream_t s, int phases = 3) {   // 1 = duplicate lists + sort + ranges, 2 = compositing
  int rc = check_problem(p);
  if (rc != DGS_OK) return rc;
  if (out == nullptr || out->out_color == nullptr) return fail(DGS_E_ARG, "DgsForwardOut: out_color is null");
  if (out->debug_contrib_checksum != nullptr && (p->tile_cull || p->forward_only || out->out_depth == nullptr))
    return fail(DGS_E_ARG, "debug_contrib_checksum needs tile_cull = 0 (positions in the reference's lists), out_depth and a "
                           "problem that keeps its state");
  const size_t N = (size_t)p->W * p->H;
  if (p->P == 0) {  // the reference returns zero-filled images when P == 0 (rasterize_points.cu:70-71,85)
    hipError_t e = hipMemsetAsync(out->out_color, 0, (size_t)p->K * 3 * N * 4, s);
    if (e == hipSuccess && out->out_depth != nullptr) e = hipMemsetAsync(out->out_depth, 0, (size_t)p->K * N * 4, s);
    return e == hipSuccess ? DGS_OK : fail_hip(e, "memset outputs");
  }
  DgsLayout L;
  make_layout(p->P, p->W, p->H, p->K, R, p->wide_records != 0, &L, p->forward_only != 0);
  if (p->geom_state == nullptr || p->geom_bytes < L.geom_total) return fail(DGS_E_CAPACITY, "geom_state too small");
  if (p->image_state == nullptr || p->image_bytes < L.image_total) return fail(DGS_E_CAPACITY, "image_state too small");
  if (R > 0 && (p->binning_state == nullptr || p->binning_bytes < L.binning_total))
    return fail(DGS_E_CAPACITY, "binning_state too small");
  DgsCarve c;
  carve(p, L, &c);
  DgsView v = make_view(p);
  if (v.tile_cull) {
    v.pack_g_shift = L.pack_g_shift;
    v.pack_tile_shift = L.pack_tile_shift;
  }
  const int key_lo = v.pack_tile_shift > 0 ? v.pack_tile_shift : 32;   // first tile bit of the key
  const uint32_t* n_dev = speculative ? c.num_rendered + 4 : nullptr;
  if ((phases & 1) && (R > 0 || v.tile_cull)) {  // tile_cull with R == 0 still marks the visible pairs as "no surviving tile"
    // choose the sort's input pair so that the result always lands in keys_sorted / point_list
    DgsCarve cd = c;
    const bool even = (L.sort_passes % 2) == 0;
    if (even) {
      cd.keys_unsorted = c.keys_sorted;
      cd.vals_unsorted = c.point_list;
    }
    if (!v.tile_cull) {
      // (1) order the (k, Gaussian) pairs by (k, depth bits, index)
      DGS_STAGE(DGS_STAGE_DEPTH_ORDER, "depth order", launch_depth_order(p, c, s));
      const uint32_t* order = c.gsort_vals;
      // (2) duplicate in that order, (3) stable sort on the tile bits only
      DGS_STAGE(DGS_STAGE_DUPLICATE, "duplicateWithKeys",
                dgs_launch_duplicate_sorted(v, cd, order, c.tt_sorted, c.offs_sorted, c.scan_tmp, R, s));
    } else {
      // the ordering and the surviving-tile offsets were produced by dgs_forward_geometry
      const uint32_t* order = c.gsort_vals;
      DGS_STAGE(DGS_STAGE_DUPLICATE, "duplicateWithKeys", dgs_launch_duplicate_tight(v, cd, order, R, s));
    }
    int in_alt = 0;
    uint64_t* kalt = even ? c.keys_unsorted : c.keys_sorted;
    uint32_t* valt = even ? c.vals_unsorted : c.point_list;
    const bool packed = v.pack_tile_shift > 0;
    DGS_STAGE(DGS_STAGE_SORT, "radix sort",
              dgs_launch_sort(cd.keys_unsorted, packed ? nullptr : cd.vals_unsorted, kalt, packed ? nullptr : valt, R,
                              key_lo, key_lo + (L.sort_bits - 32), c.sort_tmp, &in_alt, s, n_dev));
  }
  if (phases & 1) DGS_STAGE(DGS_STAGE_RANGES, "identifyTileRanges", dgs_launch_ranges(v, c, R, s, n_dev, key_lo));
  if (phases & 2)
    DGS_STAGE(DGS_STAGE_COMPOSITE_FWD, "composite forward",
              dgs_launch_composite_fwd(v, c, p->bg, out->out_color, out->out_depth, s,   // (c.final_T == NULL: inference)
                                       out->debug_contrib_checksum));
  return DGS_OK;
}

int dgs_forward_render(const DgsProblem* p, const DgsForwardOut* out, uint32_t R, dgs_stream_t stream) {
  return forward_render_impl(p, out, R, false, reinterpret_cast<hipStream_t>(stream));
}

static int forward_capacity_impl(const DgsProblem* p, const DgsForwardOut* out, uint32_t capacity, dgs_stream_t stream,
                                 int phases);
int dgs_forward(const DgsProblem* p, const DgsForwardOut* out, uint32_t capacity, dgs_stream_t stream) {
  return forward_capacity_impl(p, out, capacity, stream, 3);
}
int dgs_forward_lists(const DgsProblem* p, const DgsForwardOut* out, uint32_t capacity, dgs_stream_t stream) {
  return forward_capacity_impl(p, out, capacity, stream, 1);
}
int dgs_forward_composite(const DgsProblem* p, const DgsForwardOut* out, uint32_t capacity, dgs_stream_t stream) {
  return forward_render_impl(p, out, capacity, true, reinterpret_cast<hipStream_t>(stream), 2);
}
static int forward_capacity_impl(const DgsProblem* p, const DgsForwardOut* out, uint32_t capacity, dgs_stream_t stream,
                                 int phases) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // The count words reach the host from the finalize kernel itself when num_rendered_host is device-accessible pinned
  // memory (hipHostMalloc / torch pin_memory: the usual case) -- no copy node in the launch chain; otherwise by copies.
  uint32_t* host_dev = nullptr;
  {   // arguments first: an invalid call is refused before anything touches the HIP runtime
    const int rc0 = check_problem(p);
    if (rc0 != DGS_OK) return rc0;
    if (out == nullptr || out->num_rendered_host == nullptr)
      return fail(DGS_E_ARG, "DgsForwardOut: num_rendered_host is null");
    if (p->P > 0) {
      if (out->radii == nullptr) return fail(DGS_E_ARG, "DgsForwardOut: radii is null");
      DgsLayout L0;
      make_layout(p->P, p->W, p->H, p->K, capacity, p->wide_records != 0, &L0, p->forward_only != 0);
      if (p->geom_state == nullptr || p->geom_bytes < L0.geom_total) return fail(DGS_E_CAPACITY, "geom_state too small");
      if ((phases & 2) && (p->image_state == nullptr || p->image_bytes < L0.image_total))
        return fail(DGS_E_CAPACITY, "image_state too small");
      if (capacity > 0 && (p->binning_state == nullptr || p->binning_bytes < L0.binning_total))
        return fail(DGS_E_CAPACITY, "binning_state too small");
    }
  }
  if (p->P > 0) {
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, out->num_rendered_host, 0) == hipSuccess && dp != nullptr)
      host_dev = reinterpret_cast<uint32_t*>(dp);
    else
      (void)hipGetLastError();   // not mapped: the copies below
  }
  int rc = forward_geometry_impl(p, out, stream, host_dev == nullptr);
  if (rc != DGS_OK) return rc;
  if (p->P > 0) {
    DgsLayout L;
    make_layout(p->P, p->W, p->H, p->K, 0, p->wide_records != 0, &L, p->forward_only != 0);
    DgsCarve c;
    carve(p, L, &c);
    hipError_t e = dgs_launch_finalize_count(c, p->tile_cull != 0, capacity, out->drop_counter, out->status_dev, host_dev,
                                             out->status_host_indirect, s);
    if (host_dev == nullptr) {
      if (e == hipSuccess && out->drop_counter != nullptr)   // [4] = overflowed forwards so far (caller's running counter)
        e = hipMemcpyAsync(out->num_rendered_host + 4, out->drop_counter, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
      if (e == hipSuccess)   // [2] = overflow flag, [3] = the count the lists were built with (0 on overflow)
        e = hipMemcpyAsync(out->num_rendered_host + 2, c.num_rendered + 5, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
      if (e == hipSuccess)
        e = hipMemcpyAsync(out->num_rendered_host + 3, c.num_rendered + 4, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    }
    if (e != hipSuccess) return fail_hip(e, "finalize count");
  } else {
    out->num_rendered_host[2] = 0;
    out->num_rendered_host[3] = 0;
  }
  return forward_render_impl(p, out, capacity, true, s, phases);
}

// ---- the compositing backward in parts, each part's row totals next to the next part's compositing
// composite_bwd is bound by VALU issue and the LDS, contrib_reduce by scattered HBM writes: side by side they share the
// device well.  The subframes are cut into (up to) three parts; their compositing kernels run back to back on the
// caller's stream, and the row totals of every part but the last run on a side stream owned by the library, each as soon as
// its part's compositing is done; the caller's stream waits for the side stream before the last part's totals, so all of
// the call's work is ordered before whatever the caller enqueues next.  The kernels and every sum are the ones of the single
// launch: bit-identical results (tools/grad_hash.py).  Only for an eagerly enqueued call (backward_impl: never inside a stream
// capture); measured at the metric configuration: the eager step 10.4-10.6 ms against 10.7-10.9 replayed or eager with one
// launch.
// Policy and streams come from the caller's DgsContext (no context: always one launch).
constexpr int BWD_MAX_PARTS = DGS_MAX_BWD_PARTS;
constexpr uint64_t BWD_OVERLAP_MIN_PAIRS = 4000000;   // below this the extra launches and events cost more than they hide

// the context's side stream and events, created on first use on the device that is current (mutex held by the caller)
static hipError_t side_stream(DgsContext* ctx) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (ctx->side_ready) return dev == ctx->side_device ? hipSuccess : hipErrorInvalidDevice;   // one context, one device
  e = hipStreamCreateWithFlags(&ctx->s2, hipStreamNonBlocking);
  for (int i = 0; i < BWD_MAX_PARTS && e == hipSuccess; i++) e = hipEventCreateWithFlags(&ctx->done[i], hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->join, hipEventDisableTiming);
  if (e != hipSuccess) {   // give back what was created: the next large backward tries again from nothing
    for (int i = 0; i < BWD_MAX_PARTS; i++) {
      if (ctx->done[i] != nullptr) (void)hipEventDestroy(ctx->done[i]);
      ctx->done[i] = nullptr;
    }
    if (ctx->join != nullptr) (void)hipEventDestroy(ctx->join);
    if (ctx->s2 != nullptr) (void)hipStreamDestroy(ctx->s2);
    ctx->join = nullptr;
    ctx->s2 = nullptr;
    return e;
  }
  ctx->side_device = dev;
  ctx->side_ready = true;
  return hipSuccess;
}

// cut[0..n]: part i = subframes [cut[i], cut[i+1]).  Default for K >= 6: a first part of two thirds of the subframes -- its
// totals, which take about twice as long beside compositing as alone, then have the compositing of all the others to run
// beside --, a second one of most of the rest and a last one of one subframe (its totals are the only ones nothing runs next
// to): K = 15 -> 10, 4, 1; K = 31 -> 21, 9, 1.  (Measured for the eagerly enqueued step, the only place the parts run, on three
// boxes: 10,4,1 10.40-10.44 against 7,6,2 10.47-10.55 ms; 10,4,1 10.86, 12,2,1 10.80-10.82, 12,3 10.84-10.90, 7,6,2 10.91-10.95,
// 13,2 10.94, 14,1 11.03; on the third 12,2,1 and 7,6,2 were equal within its noise, and K = 31 lost 0.5 % with 26,4,1 against
// 14,13,4 -- profiles/r05_ab_logs.txt, calls 31-34)
static int bwd_parts(const DgsContextOptions& opt, int K, int* cut, bool force) {
  int n = 0;
  cut[0] = 0;
  if (opt.bwd_n_parts > 0) {
    for (int i = 0; i < opt.bwd_n_parts && i < BWD_MAX_PARTS - 1 && n < BWD_MAX_PARTS - 1; i++) {
      const int len = opt.bwd_parts[i];
      if (len <= 0 || cut[n] + len >= K) break;
      cut[n + 1] = cut[n] + len;
      n++;
    }
  } else if (K >= 6) {
    cut[1] = (2 * K + 1) / 3;
    cut[2] = K - 1;
    n = 2;
  } else if (force && K >= 2) {
    cut[1] = (K + 1) / 2;
    n = 1;
  }
  cut[++n] = K;
  return n;
}

static hipError_t backward_composite_overlapped(const DgsProblem* p, const DgsBackwardIO* io, const DgsView& v, const DgsCarve& c,
                                                float* contrib, float* sums, double* partials, hipStream_t s, const int* cut,
                                                int n) {
  DgsContext* ctx = p->context;
  hipStream_t s2 = ctx->s2;
  hipError_t e = hipSuccess;
  bool forked = false;
  for (int i = 0; i < n && e == hipSuccess; i++) {
    e = dgs_launch_composite_bwd(v, c, p->bg, io->dL_dout_color, io->dL_dout_depth, contrib, s, cut[i], cut[i + 1]);
    if (i + 1 < n) {
      if (e == hipSuccess) e = hipEventRecord(ctx->done[i], s);
      if (e == hipSuccess) e = hipStreamWaitEvent(s2, ctx->done[i], 0);
      if (e == hipSuccess) {
        forked = true;
        e = dgs_launch_geometry_bwd(*p, v, c, *io, contrib, sums, partials, s2, 1, 0, 0, cut[i], cut[i + 1]);
      }
    }
  }
  // the join is enqueued also when something failed half-way: whatever already sits on the side stream is ordered before
  // the caller's next work on `s`
  if (forked) {
    hipError_t j = hipEventRecord(ctx->join, s2);
    if (j == hipSuccess) j = hipStreamWaitEvent(s, ctx->join, 0);
    if (j != hipSuccess && e == hipSuccess) e = j;
    if (j != hipSuccess) (void)hipStreamSynchronize(s2);
  }
  if (e == hipSuccess) e = dgs_launch_geometry_bwd(*p, v, c, *io, contrib, sums, partials, s, 1, 0, 0, cut[n - 1], cut[n]);
  return e;
}

// which != 0: 1 = compositing backward + per-pair totals, 2 = per-Gaussian kernel for [g_begin, g_end), 4 = pose sums
static int backward_impl(const DgsProblem* p, const DgsBackwardIO* io, int which, int32_t g_begin, int32_t g_end,
                         hipStream_t s) {
  int rc = check_problem(p, p != nullptr && p->raw_params != 0);
  if (rc != DGS_OK) return rc;
  if (p->forward_only) return fail(DGS_E_ARG, "backward: the problem is forward_only (nothing was kept for a backward)");
  if (io == nullptr) return fail(DGS_E_ARG, "null DgsBackwardIO");
  if (io->dL_dviewmatrix == nullptr || io->dL_dprojmatrix == nullptr)
    return fail(DGS_E_ARG, "dL_dviewmatrix / dL_dprojmatrix are null");
  if (p->P == 0) {
    if (!(which & 4)) return DGS_OK;
    hipError_t e = hipMemsetAsync(io->dL_dviewmatrix, 0, (size_t)p->K * 64, s);
    if (e == hipSuccess) e = hipMemsetAsync(io->dL_dprojmatrix, 0, (size_t)p->K * 64, s);
    return e == hipSuccess ? DGS_OK : fail_hip(e, "memset grads");
  }
  if (io->dL_dout_color == nullptr || io->radii == nullptr || io->dL_dmeans3D == nullptr ||
      (io->dL_dmeans2D == nullptr && io->stats_max_radii2D == nullptr) || io->dL_dcolors == nullptr ||
      io->dL_dopacity == nullptr ||
      io->dL_dcov3D == nullptr)
    return fail(DGS_E_ARG, "DgsBackwardIO: a required pointer is null");
  if (p->shs != nullptr && io->dL_dsh == nullptr) return fail(DGS_E_ARG, "dL_dsh is null");
  if (p->raw_params && p->M > 1 && io->dL_dsh_rest == nullptr) return fail(DGS_E_ARG, "dL_dsh_rest is null");
  if (p->scales != nullptr && (io->dL_dscales == nullptr || io->dL_drotations == nullptr))
    return fail(DGS_E_ARG, "dL_dscales / dL_drotations are null");
  if (g_begin < 0 || g_end > p->P || g_begin > g_end || (g_begin % 256) != 0)
    return fail(DGS_E_ARG, "backward: bad Gaussian range (g_begin must be a multiple of 256)");
  const uint64_t R = io->num_rendered;
  DgsLayout L;
  make_layout(p->P, p->W, p->H, p->K, R, p->wide_records != 0, &L);
  if (p->geom_state == nullptr || p->geom_bytes < L.geom_total) return fail(DGS_E_CAPACITY, "geom_state too small");
  if (p->image_state == nullptr || p->image_bytes < L.image_total) return fail(DGS_E_CAPACITY, "image_state too small");
  if (R > 0 && (p->binning_state == nullptr || p->binning_bytes < L.binning_total))
    return fail(DGS_E_CAPACITY, "binning_state too small");
  if (io->scratch == nullptr || io->scratch_bytes < dgs_backward_scratch_bytes(R, p->P, p->K))
    return fail(DGS_E_CAPACITY, "backward scratch too small");
  DgsCarve c;
  carve(p, L, &c);
  DgsView v = make_view(p);
  if (v.tile_cull) {
    v.pack_g_shift = L.pack_g_shift;
    v.pack_tile_shift = L.pack_tile_shift;
  }
  float* contrib = reinterpret_cast<float*>(io->scratch);
  float* sums = reinterpret_cast<float*>(reinterpret_cast<char*>(io->scratch) + up((size_t)R * DGS_CONTRIB_F * 4));
  double* partials = reinterpret_cast<double*>(reinterpret_cast<char*>(sums) +
                                               up((size_t)p->K * (size_t)p->P * DGS_SUMS_F * 4));
  // (with the stage timers on -- dgs_profile_begin -- the two kernels run one after the other: their event pairs then time
  // what they say; so does debug mode, which synchronises after every stage)
  // ... and never inside a stream capture (unless DGS_BWD_OVERLAP=3 asks for it): on this runtime (ROCm 7.2) an executable
  // graph with a kernel on a forked branch does not give back all device memory when it is destroyed -- 4 MB per
  // capture / instantiate / destroy cycle in plain HIP (tools/graph_fork_leak.hip), and with torch's graph pools the
  // step's released buffers stay in use as well: ~0.9 GB per re-capture at 1.2 M Gaussians (tools/soak.py)
  int cut[BWD_MAX_PARTS + 1], parts = 1;
  DgsContext* ctx = p->context;
  const int omode = ctx != nullptr ? ctx->opt.bwd_overlap : 0;
  std::unique_lock<std::mutex> side_lock;
  if ((which & 1) && v.tile_cull && !p->debug && omode != 0 && !ctx->prof.on && (R >= BWD_OVERLAP_MIN_PAIRS || omode >= 2)) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (omode != 3 && hipStreamIsCapturing(s, &cs) != hipSuccess) cs = hipStreamCaptureStatusActive;   // (unknown: stay serial)
    if (cs == hipStreamCaptureStatusNone) parts = bwd_parts(ctx->opt, p->K, cut, omode >= 2);
    if (parts > 1) {
      side_lock = std::unique_lock<std::mutex>(ctx->mu);
      if (side_stream(ctx) != hipSuccess) {   // no side stream to be had: the single launch
        (void)hipGetLastError();
        side_lock.unlock();
        parts = 1;
      }
    }
  }
  if (parts > 1) {
    hipError_t e = backward_composite_overlapped(p, io, v, c, contrib, sums, partials, s, cut, parts);
    side_lock.unlock();
    if (e != hipSuccess) return fail_hip(e, "composite backward (parts)");
  } else if (which & 1) {
    DGS_STAGE(DGS_STAGE_COMPOSITE_BWD, "composite backward",
              dgs_launch_composite_bwd(v, c, p->bg, io->dL_dout_color, io->dL_dout_depth, contrib, s));
    DGS_STAGE(DGS_STAGE_CONTRIB_REDUCE, "contribution-row totals",
              dgs_launch_geometry_bwd(*p, v, c, *io, contrib, sums, partials, s, 1, 0, 0));
  }
  if (which & 6)
    DGS_STAGE(DGS_STAGE_GEOMETRY_BWD, "geometry backward",
              dgs_launch_geometry_bwd(*p, v, c, *io, contrib, sums, partials, s, which & 6, g_begin, g_end));
  return DGS_OK;
}

int dgs_backward(const DgsProblem* p, const DgsBackwardIO* io, dgs_stream_t stream) {
  return backward_impl(p, io, 7, 0, p != nullptr ? p->P : 0, reinterpret_cast<hipStream_t>(stream));
}
int dgs_backward_composite(const DgsProblem* p, const DgsBackwardIO* io, dgs_stream_t stream) {
  return backward_impl(p, io, 1, 0, 0, reinterpret_cast<hipStream_t>(stream));
}
int dgs_backward_geometry(const DgsProblem* p, const DgsBackwardIO* io, int32_t g_begin, int32_t g_end,
                          dgs_stream_t stream) {
  return backward_impl(p, io, 2, g_begin, g_end, reinterpret_cast<hipStream_t>(stream));
}
int32_t dgs_backward_parts(const DgsContext* ctx, int32_t K, uint64_t num_rendered, int32_t tile_cull) {
  const int omode = ctx != nullptr ? ctx->opt.bwd_overlap : 0;
  if (K < 1 || !tile_cull || omode == 0 || (num_rendered < BWD_OVERLAP_MIN_PAIRS && omode < 2)) return 1;
  int cut[BWD_MAX_PARTS + 1];
  return bwd_parts(ctx->opt, K, cut, omode >= 2);
}
int dgs_backward_pose(const DgsProblem* p, const DgsBackwardIO* io, dgs_stream_t stream) {
  return backward_impl(p, io, 4, 0, 0, reinterpret_cast<hipStream_t>(stream));
}

int dgs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, dgs_stream_t stream) {
  (void)projmatrix;  // in_frustum only uses the view depth (auxiliary.h:159)
  if (P < 0 || (P > 0 && (means3D == nullptr || viewmatrix == nullptr || present == nullptr)))
    return fail(DGS_E_ARG, "mark_visible: null pointer");
  if (P == 0) return DGS_OK;
  hipError_t e = dgs_launch_mark_visible(P, means3D, viewmatrix, present, reinterpret_cast<hipStream_t>(stream));
  return e == hipSuccess ? DGS_OK : fail_hip(e, "mark_visible");
}

int dgs_cloud_activations(int32_t P, const float* scaling, const float* rotation, const float* opacity, float scale_lb,
                          float* out_scaling, float* out_rotation, float* out_opacity, dgs_stream_t stream) {
  if (P < 0 || (out_scaling != nullptr && scaling == nullptr) || (out_rotation != nullptr && rotation == nullptr) ||
      (out_opacity != nullptr && opacity == nullptr))
    return fail(DGS_E_ARG, "cloud_activations: an output is requested without its input");
  if (P == 0) return DGS_OK;
  hipError_t e = dgs_launch_cloud_activations(P, scaling, rotation, opacity, scale_lb, out_scaling, out_rotation,
                                              out_opacity, reinterpret_cast<hipStream_t>(stream));
  return e == hipSuccess ? DGS_OK : fail_hip(e, "cloud_activations");
}

size_t dgs_scan_tmp_bytes(uint64_t n) { return dgs_scan_tmp_words(n) * 4; }
int dgs_exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint64_t n, void* tmp, uint32_t* total_out,
                           dgs_stream_t stream) {
  hipError_t e = dgs_launch_scan(in, out, n, reinterpret_cast<uint32_t*>(tmp), total_out,
                                 reinterpret_cast<hipStream_t>(stream));
  return e == hipSuccess ? DGS_OK : fail_hip(e, "scan");
}
size_t dgs_sort_tmp_bytes(uint64_t n) { return dgs_sort_tmp_words(n) * 4; }
int dgs_sort_pairs(uint64_t* keys, uint32_t* vals, uint64_t* keys_alt, uint32_t* vals_alt, uint64_t n,
                   int32_t begin_bit, int32_t end_bit, void* tmp, int32_t* result_in_alt, dgs_stream_t stream) {
  if (n >= (1ull << 32)) return fail(DGS_E_ARG, "sort: n must be < 2^32");
  if (begin_bit < 0 || end_bit > 64 || end_bit <= begin_bit) return fail(DGS_E_ARG, "sort: bad bit range");
  int alt = 0;
  hipError_t e = dgs_launch_sort(keys, vals, keys_alt, vals_alt, n, begin_bit, end_bit, reinterpret_cast<uint32_t*>(tmp), &alt,
                                 reinterpret_cast<hipStream_t>(stream));
  if (result_in_alt) *result_in_alt = alt;
  return e == hipSuccess ? DGS_OK : fail_hip(e, "sort");
}

size_t dgs_depth_order_tmp_bytes(int32_t K, int32_t P) {
  return (K > 0 && P > 0) ? dgs_depth_sort_tmp_words(K, (uint32_t)P) * 4 + 256 : 256;
}
int dgs_depth_order(uint32_t* keys, uint32_t* keys_alt, uint32_t* order, uint32_t* order_alt, int32_t K, int32_t P,
                    void* tmp, uint32_t* visible, dgs_stream_t stream) {
  if (K < 0 || P < 0) return fail(DGS_E_ARG, "depth_order: bad size");
  if (K == 0 || P == 0) return DGS_OK;
  if (keys == nullptr || keys_alt == nullptr || order == nullptr || order_alt == nullptr || tmp == nullptr)
    return fail(DGS_E_ARG, "depth_order: null buffer");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // the last 256 bytes of tmp hold the "a key needs the fourth pass" word
  uint32_t* words = reinterpret_cast<uint32_t*>(tmp);
  uint32_t* flag = words + dgs_depth_sort_tmp_words(K, (uint32_t)P);
  hipError_t e = hipMemsetAsync(flag, 0, 4, s);
  if (e == hipSuccess) e = dgs_launch_depth_sort(keys, keys_alt, order, order_alt, K, (uint32_t)P, words, visible, flag, s);
  return e == hipSuccess ? DGS_OK : fail_hip(e, "depth_order");
}

static int blur_loss_impl(const float* subframes, const float* gt, int32_t K, int32_t C, int32_t HW, float lambda_t,
                          const float* lambda_dev, const float* upstream, float* blur, float* dL_dsubframes, float* losses,
                          dgs_stream_t stream) {
  const bool fwd = (blur != nullptr && losses != nullptr);
  // losses given: blur is an output (forward, or forward + backward when dL_dsubframes is given too);
  // losses NULL: backward only, blur (optional) is the forward's blur handed back in
  if (subframes == nullptr || gt == nullptr || K < 1 || C < 1 || HW < 1 || (!fwd && dL_dsubframes == nullptr) ||
      (losses != nullptr && blur == nullptr))
    return fail(DGS_E_ARG, "blur_loss_grad: bad argument");
  hipError_t e = dgs_launch_blur_loss(subframes, gt, K, C, HW, lambda_t, lambda_dev, upstream, blur, dL_dsubframes,
                                      losses, reinterpret_cast<hipStream_t>(stream));
  return e == hipSuccess ? DGS_OK : fail_hip(e, "blur_loss_grad");
}
int dgs_blur_loss_grad(const float* subframes, const float* gt, int32_t K, int32_t C, int32_t HW, float lambda_t,
                       const float* upstream, float* blur, float* dL_dsubframes, float* losses, dgs_stream_t stream) {
  return blur_loss_impl(subframes, gt, K, C, HW, lambda_t, nullptr, upstream, blur, dL_dsubframes, losses, stream);
}
int dgs_blur_loss_grad_dev(const float* subframes, const float* gt, int32_t K, int32_t C, int32_t HW,
                           const float* lambda_t_dev, const float* upstream, float* blur, float* dL_dsubframes,
                           float* losses, dgs_stream_t stream) {
  if (lambda_t_dev == nullptr) return fail(DGS_E_ARG, "blur_loss_grad_dev: lambda_t_dev is null");
  return blur_loss_impl(subframes, gt, K, C, HW, 0.0f, lambda_t_dev, upstream, blur, dL_dsubframes, losses, stream);
}

int dgs_blur_loss_slice_grad(const float* subframes, const float* prev_last, const float* next_first, const float* blur,
                             const float* gt, int32_t K_local, int32_t K_total, int32_t C, int32_t HW, float lambda_t,
                             float* dL_dsubframes, float* losses, dgs_stream_t stream) {
  if (subframes == nullptr || blur == nullptr || gt == nullptr || dL_dsubframes == nullptr || losses == nullptr ||
      K_local < 1 || K_total < K_local || K_local > 32 || C < 1 || HW < 1)
    return fail(DGS_E_ARG, "blur_loss_slice_grad: bad argument (1 <= K_local <= 32, K_local <= K_total)");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t E = (size_t)C * HW;
  hipError_t e = dgs_launch_clear_words(reinterpret_cast<uint32_t*>(losses), 8, s);
  if (e != hipSuccess) return fail_hip(e, "blur_loss_slice_grad");
  const uintptr_t align = reinterpret_cast<uintptr_t>(subframes) | reinterpret_cast<uintptr_t>(blur) |
                          reinterpret_cast<uintptr_t>(gt) | reinterpret_cast<uintptr_t>(dL_dsubframes) |
                          reinterpret_cast<uintptr_t>(prev_last) | reinterpret_cast<uintptr_t>(next_first);
  const bool v4 = (E % 4 == 0) && (align % 16 == 0);
  const size_t per = v4 ? 4 : 1;
  const size_t want = (E / per + 255) / 256;
  const dim3 grid((uint32_t)(want < 512 ? (want == 0 ? 1 : want) : 512));   // (as dgs_launch_blur_loss: few blocks, few atomics)
#define DGS_BLS(KMAX, V)                                                                                              \
  hipLaunchKernelGGL((blur_loss_slice_kernel<KMAX, V>), grid, dim3(256), 0, s, subframes, prev_last, next_first, blur, gt, \
                     K_local, K_total, E, lambda_t, dL_dsubframes, losses)
  if (v4) {
    if (K_local <= 4) DGS_BLS(4, 4); else if (K_local <= 16) DGS_BLS(16, 4); else DGS_BLS(32, 2);
  } else {
    if (K_local <= 4) DGS_BLS(4, 1); else if (K_local <= 16) DGS_BLS(16, 1); else DGS_BLS(32, 1);
  }
#undef DGS_BLS
  e = hipGetLastError();
  return e == hipSuccess ? DGS_OK : fail_hip(e, "blur_loss_slice_grad");
}

// device-side address of a pointer that is either device memory or mapped pinned host memory; nullptr otherwise (pageable)
static void* device_view(const void* ptr) {
  void* dp = nullptr;
  if (hipHostGetDevicePointer(&dp, const_cast<void*>(ptr), 0) == hipSuccess && dp != nullptr) return dp;
  (void)hipGetLastError();
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, ptr) == hipSuccess && at.type == hipMemoryTypeDevice) return const_cast<void*>(ptr);
  (void)hipGetLastError();
  return nullptr;
}

int dgs_copy_words(void* dst, const void* src, int32_t n_words, dgs_stream_t stream) {
  if (n_words < 0 || n_words > 4096 || (n_words > 0 && (dst == nullptr || src == nullptr)))
    return fail(DGS_E_ARG, "copy_words: bad argument (at most 4096 words)");
  if (n_words == 0) return DGS_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  void* d = device_view(dst);
  void* sp = device_view(src);
  hipError_t e;
  if (d != nullptr && sp != nullptr)   // both reachable from a kernel (system-scope stores + fence: dst may be pinned host)
    e = dgs_launch_copy_words(reinterpret_cast<uint32_t*>(d), reinterpret_cast<const uint32_t*>(sp), n_words, s);
  else                                 // a pageable end: the runtime's copy
    e = hipMemcpyAsync(dst, src, (size_t)n_words * 4, hipMemcpyDefault, s);
  return e == hipSuccess ? DGS_OK : fail_hip(e, "copy_words");
}

int dgs_densify_stats(const float* viewspace_grad, const int32_t* radii, int32_t K, int32_t K_total, int32_t P,
                      float* max_radii2D, float* xyz_gradient_accum, float* denom, const uint32_t* skip_flag,
                      dgs_stream_t stream) {
  if (K_total <= 0) K_total = K;
  if (K < 1 || K_total < K || P < 0 || (P > 0 && (viewspace_grad == nullptr || radii == nullptr || max_radii2D == nullptr ||
                                   xyz_gradient_accum == nullptr || denom == nullptr)))
    return fail(DGS_E_ARG, "densify_stats: bad argument");
  if (P == 0) return DGS_OK;
  hipLaunchKernelGGL(densify_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     viewspace_grad, radii, K, K_total, P, max_radii2D, xyz_gradient_accum, denom, skip_flag);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DGS_OK : fail_hip(e, "densify_stats");
}

int dgs_context_create(const DgsContextOptions* options, DgsContext** out) {
  if (out == nullptr) return fail(DGS_E_ARG, "context_create: null output");
  *out = nullptr;
  DgsContextOptions o;
  memset(&o, 0, sizeof(o));
  o.bwd_overlap = 1;
  if (options != nullptr) o = *options;
  if (o.bwd_overlap < 0 || o.bwd_overlap > 3 || o.bwd_n_parts < 0 || o.bwd_n_parts > DGS_MAX_BWD_PARTS - 1)
    return fail(DGS_E_ARG, "context_create: bwd_overlap must be 0..3, bwd_n_parts 0..DGS_MAX_BWD_PARTS-1");
  for (int i = 0; i < o.bwd_n_parts; i++)
    if (o.bwd_parts[i] < 1) return fail(DGS_E_ARG, "context_create: a part must hold at least one subframe");
  DgsContext* c = new (std::nothrow) DgsContext();
  if (c == nullptr) return fail(DGS_E_ARG, "context_create: out of host memory");
  c->opt = o;
  *out = c;
  return DGS_OK;
}
int dgs_context_destroy(DgsContext* ctx) {
  if (ctx == nullptr) return DGS_OK;
  hipError_t e = hipSuccess;
  if (ctx->side_ready) {
    e = hipStreamSynchronize(ctx->s2);
    for (int i = 0; i < BWD_MAX_PARTS; i++) (void)hipEventDestroy(ctx->done[i]);
    (void)hipEventDestroy(ctx->join);
    (void)hipStreamDestroy(ctx->s2);
  }
  for (int i = 0; i < ctx->prof.created; i++) {
    (void)hipEventDestroy(ctx->prof.beg[i]);
    (void)hipEventDestroy(ctx->prof.end[i]);
  }
  delete[] ctx->prof.beg;
  delete[] ctx->prof.end;
  delete[] ctx->prof.stage;
  delete ctx;
  return e == hipSuccess ? DGS_OK : fail_hip(e, "context_destroy");
}

int dgs_profile_enable(DgsContext* ctx, int32_t on) {
  if (ctx == nullptr) return fail(DGS_E_ARG, "profile_enable: null context");
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (on) {   // (each array on its own: a call that ran out of host memory half-way is completed by the next one)
    if (ctx->prof.beg == nullptr) ctx->prof.beg = new (std::nothrow) hipEvent_t[PROF_MAX];
    if (ctx->prof.end == nullptr) ctx->prof.end = new (std::nothrow) hipEvent_t[PROF_MAX];
    if (ctx->prof.stage == nullptr) ctx->prof.stage = new (std::nothrow) int[PROF_MAX];
    if (ctx->prof.beg == nullptr || ctx->prof.end == nullptr || ctx->prof.stage == nullptr)
      return fail(DGS_E_ARG, "profile_enable: out of host memory");
  }
  ctx->prof.on = on != 0;
  return DGS_OK;
}
int dgs_profile_reset(DgsContext* ctx) {
  if (ctx == nullptr) return fail(DGS_E_ARG, "profile_reset: null context");
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->prof.n = 0;
  return DGS_OK;
}
int dgs_profile_read(DgsContext* ctx, float* ms, int32_t* calls, int32_t n) {
  if (ctx == nullptr) return fail(DGS_E_ARG, "profile_read: null context");
  std::lock_guard<std::mutex> lk(ctx->mu);
  Prof& pr = ctx->prof;
  for (int i = 0; i < n; i++) {
    if (ms) ms[i] = 0.0f;
    if (calls) calls[i] = 0;
  }
  for (int i = 0; i < pr.n; i++) {
    hipError_t e = hipEventSynchronize(pr.end[i]);
    if (e != hipSuccess) return fail_hip(e, "profile_read");
    float t = 0.0f;
    e = hipEventElapsedTime(&t, pr.beg[i], pr.end[i]);
    if (e != hipSuccess) return fail_hip(e, "profile_read");
    const int st = pr.stage[i];
    if (st >= 0 && st < n) {
      if (ms) ms[st] += t;
      if (calls) calls[st] += 1;
    }
  }
  return DGS_OK;
}

#ifndef DGS_BUILD_ID
#define DGS_BUILD_ID "unstamped"
#endif
const char* dgs_build_id(void) { return DGS_BUILD_ID; }

}  // extern "C"
