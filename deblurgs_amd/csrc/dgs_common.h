// dgs_common.h -- shared declarations of the gfx950 kernels behind include/dgs_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dgs_hip.h"

#define DGS_TILE 16                 // reference BLOCK_X/BLOCK_Y (config.h:16-17): tile ids must match
// Depth-sort key of a visible (subframe, Gaussian) pair = bits(view depth) - DGS_DEPTH_KEY_BASE.  in_frustum keeps
// depth > 0.2f only (auxiliary.h:159), positive floats order like their bit patterns, so the difference is positive,
// order-preserving and below 2^27 for every depth under 13107: the depth order is then three 9-bit radix passes
// (binning.hip; a key that needs more bits sets a flag that switches a fourth pass on).  Invisible pairs: 0xFFFFFFFF.
#define DGS_DEPTH_KEY_BASE 0x3E4CCCCDu   // bits(0.2f)
#define DGS_ROW_F 12                // floats per geometry row (48 B)
#define DGS_CONTRIB_F 12            // floats per backward contribution row (48 B): 10 used
#define DGS_SUMS_F 16               // float stride of the per-(subframe, Gaussian) totals of those rows: 64-byte slots, so the
                                    // scattered writes of contrib_reduce_kernel are whole aligned lines (12-float slots: 771 us, 16: 663 us)
#define DGS_WAVE 64

// Geometry row: everything the per-tile compositing needs about one (subframe, Gaussian), gathered by one
// lane with three 16-byte loads.  Replaces the reference's separate means2D / conic_opacity / rgb / depths
// arrays (rasterizer_impl.h:31-45) whose per-pair global reads (forward.cu:371-373) become LDS reads.
struct __attribute__((aligned(16))) DgsRow {
  float x, y;            // pixel-space mean (ndc2Pix)
  float cx, cy, cz, op;  // conic (a, b, c of the inverse 2-D covariance) and opacity
  float r, g, b, depth;  // activated colour, view-space depth
  uint32_t dup_offset;   // relu colour activation: bits 0..2 = the clamp mask of r, g, b (1 = the channel passed the clamp:
                         // what the SH backward multiplies by, backward.cu:35-47); 0 with the sigmoid activation
  int32_t radius;        // ceil(3 sigma_max) in pixels
};
static_assert(sizeof(DgsRow) == 4 * DGS_ROW_F, "row must be 48 bytes");

struct DgsCarve {  // resolved device pointers of the three blobs
  DgsRow* rows;
  float* cov3D;
  float* pre_sigmoid;
  uint32_t* tiles_touched;
  uint32_t* point_offsets;
  uint32_t* scan_tmp;
  uint32_t* num_rendered;
  uint32_t* gsort_keys;      // depth bits per (k, Gaussian) (0xFFFFFFFF = invisible), and the sort's ping-pong buffer
  uint32_t* gsort_keys_alt;
  uint32_t* gsort_vals;      // flat (k, Gaussian) indices in (k, depth, index) order (the depth sort's result)
  uint32_t* gsort_vals_alt;
  uint32_t* tt_sorted;       // tiles_touched in that order, and its exclusive scan
  uint32_t* offs_sorted;
  uint32_t* tt_tight;        // tile_cull: per (k, Gaussian) count of tiles that can reach alpha >= 1/255, same order,
  uint32_t* offs_tight;      //            and its exclusive scan (= duplicate / contribution-row offsets)
  uint32_t* gsort_tmp;
  uint4* cull_rec;           // tile_cull: per (k, Gaussian) rectangle, surviving-tile count, hit bits of the first 64 slots
  uint32_t* cull_cnt;        //            the counts alone (both by natural index)
  float* final_T;
  uint32_t* n_contrib;
  uint2* ranges;
  uint64_t* keys_sorted;
  uint32_t* point_list;
  uint64_t* keys_unsorted;
  uint32_t* vals_unsorted;
  uint32_t* sort_tmp;
};

static inline uint32_t dgs_higher_msb64(uint64_t n) {   // bits needed for values 0 .. n
  uint32_t b = 0;
  while (b < 64 && (n >> b) != 0) b++;
  return b;
}
// rasterizer_impl.cu:35-50
static inline uint32_t dgs_higher_msb(uint32_t n) {
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

// auxiliary.h:46-56.  v_cvt_i32_f32 saturates, like the reference's cvt.rzi.
__device__ __forceinline__ void dgs_get_rect(float px, float py, int max_radius, int gx, int gy, int& minx,
                                             int& miny, int& maxx, int& maxy) {
  minx = min(gx, max(0, (int)((px - max_radius) / DGS_TILE)));
  miny = min(gy, max(0, (int)((py - max_radius) / DGS_TILE)));
  maxx = min(gx, max(0, (int)((px + max_radius + DGS_TILE - 1) / DGS_TILE)));
  maxy = min(gy, max(0, (int)((py + max_radius + DGS_TILE - 1) / DGS_TILE)));
}

// ---- wave64 helpers -------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dgs_dpp(float v) {
  // lanes disabled by ROW_MASK (or reading out of range) see 0.0f
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}

// Sum over the 64 lanes of a wave; the total is valid in lane 63 (DPP butterflies inside each row of 16,
// then row_bcast15 / row_bcast31 across rows).  6 VALU ops, no LDS.
__device__ __forceinline__ float dgs_wave_sum63(float v) {
  v += dgs_dpp<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dgs_dpp<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dgs_dpp<0x141, 0xf>(v);  // row_half_mirror
  v += dgs_dpp<0x140, 0xf>(v);  // row_mirror
  v += dgs_dpp<0x142, 0xa>(v);  // row_bcast15 -> rows 1,3
  v += dgs_dpp<0x143, 0xc>(v);  // row_bcast31 -> rows 2,3
  return v;
}

__device__ __forceinline__ int dgs_lane() { return (int)(threadIdx.x & 63); }

// every lane of a quad <- the sum over the quad
__device__ __forceinline__ float dgs_quad_sum(float v) {
  v += dgs_dpp<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dgs_dpp<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  return v;
}
// every lane of a 16-lane row <- the sum over that row
__device__ __forceinline__ float dgs_row_sum(float v) {
  v += dgs_dpp<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dgs_dpp<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dgs_dpp<0x141, 0xf>(v);  // row_half_mirror
  v += dgs_dpp<0x140, 0xf>(v);  // row_mirror
  return v;
}

// ---- host-side launch entry points (one per .hip file) -----------------------------------------------------
struct DgsView {  // per-launch scalars shared by the kernels
  int P, D, M, W, H, K;
  int gx, gy, T;  // tile grid and tiles per subframe
  float tanfovx, tanfovy, focal_x, focal_y, scale_modifier, z_far;
  int use_sigmoid, prefiltered;
  int tile_cull;
  int raw_params;   // kernels apply the cloud's activations (DgsProblem.raw_params bit 0)
  int iso_scale;    // raw_params bit 1: one shared scale per Gaussian = column 0 of `scales` (use_isotrophic)
  // tile_cull with compact keys (DgsLayout.pack_*): key = tile << pack_tile_shift | Gaussian << pack_g_shift | emission
  // index, no value array; pack_tile_shift == 0: key = tile << 32 | emission index, Gaussian in point_list
  int pack_g_shift, pack_tile_shift;
  float scale_lb;
};

// ---- the reference's parameter activations (scene/gaussian_activation.py:29-52, torch.nn.functional.normalize) for
// DgsProblem.raw_params = 1; forward and backward call the same functions so that both see identical values
__device__ __forceinline__ float dgs_act_opacity(float x) { return fminf(1.0f, fmaxf(0.0f, x)); }
__device__ __forceinline__ float dgs_act_scale(float x, float lb) { return expf(x) + lb; }
__device__ __forceinline__ float dgs_quat_norm(float r, float x, float y, float z) {
  return fmaxf(sqrtf(r * r + x * x + y * y + z * z), 1e-12f);
}

// ---- exact "can this Gaussian reach alpha >= 1/255 anywhere in this pixel box" test -----------------------------
// alpha = op * exp(-q/2), q(d) = a dx^2 + 2 b dx dy + c dy^2 with d = mean - pixel (forward.cu:346-358), so a pair
// contributes only where q <= 2 ln(255 op).  The minimum of the convex q over an axis-aligned box of pixel centres
// is 0 if the mean is inside, else it lies on an edge where it has a closed form.  Used per 8x8 quadrant by the
// compositing kernels and per 16x16 tile by the tile_cull duplicate emission; always conservative (slack >> the
// fp32 error of `power`), so a culled pair is one the reference would have skipped at every pixel of the box.
struct DgsCull {
  float a, b, c, inv_a, inv_c, r2;
  bool always;  // degenerate conic: never cull
  bool never;   // opacity too small to ever reach 1/255
};
__device__ __forceinline__ DgsCull dgs_make_cull(float cx, float cy, float cz, float op) {
  DgsCull g;
  g.a = cx;
  g.b = cy;
  g.c = cz;
  g.r2 = 2.0f * __logf(255.0f * op) + 0.02f;
  const float det = cx * cz - cy * cy;
  g.always = !(det > 0.0f && cx > 0.0f && cz > 0.0f);  // also catches NaN
  g.never = (g.r2 < 0.0f);
  g.inv_a = 1.0f / cx;
  g.inv_c = 1.0f / cz;
  return g;
}
__device__ __forceinline__ float dgs_clampf(float v, float lo, float hi) { return fminf(hi, fmaxf(lo, v)); }
// min over dy in [lo, hi] of q(e, dy)
__device__ __forceinline__ float dgs_edge_min_x(const DgsCull& g, float e, float lo, float hi) {
  const float t = dgs_clampf(-g.b * e * g.inv_c, lo, hi);
  return g.a * e * e + (2.0f * g.b * e + g.c * t) * t;
}
// min over dx in [lo, hi] of q(dx, e)
__device__ __forceinline__ float dgs_edge_min_y(const DgsCull& g, float e, float lo, float hi) {
  const float t = dgs_clampf(-g.b * e * g.inv_a, lo, hi);
  return g.c * e * e + (2.0f * g.b * e + g.a * t) * t;
}
__device__ __forceinline__ bool dgs_cull_hit(const DgsCull& g, float dx_lo, float dx_hi, float dy_lo, float dy_hi) {
  if (g.always) return true;
  if (g.never) return false;
  // q is convex with its minimum (0) at the origin: over a box that does not contain the origin, the minimum lies on a
  // face that is VISIBLE from the origin (from the minimiser, q decreases along the segment towards the origin, so that
  // segment leaves the box through the face the minimiser sits on).  At most one x-face and one y-face are visible:
  // two clamped 1-D minimisations instead of four.
  const bool x_out = (dx_lo > 0.0f) || (dx_hi < 0.0f), y_out = (dy_lo > 0.0f) || (dy_hi < 0.0f);
  const float ex = (dx_lo > 0.0f) ? dx_lo : dx_hi, ey = (dy_lo > 0.0f) ? dy_lo : dy_hi;
  const float qx = dgs_edge_min_x(g, ex, dy_lo, dy_hi), qy = dgs_edge_min_y(g, ey, dx_lo, dx_hi);
  const float big = __int_as_float(0x7f800000);
  float qm = fminf(x_out ? qx : big, y_out ? qy : big);
  qm = (x_out || y_out) ? qm : 0.0f;     // origin inside the box
  return !(qm * 0.9999f > g.r2);  // NaN -> keep
}

hipError_t dgs_launch_preprocess(const DgsProblem& p, const DgsView& v, const DgsCarve& c, int32_t* radii,
                                 hipStream_t s);
hipError_t dgs_launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s);
hipError_t dgs_launch_cloud_activations(int P, const float* scales, const float* rotations, const float* opacities,
                                        float scale_lb, float* out_scales, float* out_rotations, float* out_opacities,
                                        hipStream_t s);
hipError_t dgs_launch_finalize_count(const DgsCarve& c, int cull, uint32_t cap, uint32_t* drops, uint32_t* status,
                                     uint32_t* host_words, const uint64_t* host_indirect, hipStream_t s);
hipError_t dgs_launch_clear_words(uint32_t* p, int n, hipStream_t s);
hipError_t dgs_launch_copy_words(uint32_t* dst, const uint32_t* src, int n, hipStream_t s);
hipError_t dgs_launch_ranges(const DgsView& v, const DgsCarve& c, uint32_t R, hipStream_t s,
                             const uint32_t* n_dev = nullptr, int tile_shift = 32);
hipError_t dgs_launch_scan(const uint32_t* in, uint32_t* out, uint64_t n, uint32_t* tmp, uint32_t* total,
                           hipStream_t s);
hipError_t dgs_launch_sort(uint64_t* keys, uint32_t* vals, uint64_t* keys_alt, uint32_t* vals_alt, uint64_t n,
                           int begin_bit, int end_bit, uint32_t* tmp, int* result_in_alt, hipStream_t s,
                           const uint32_t* n_dev = nullptr);
hipError_t dgs_launch_duplicate_sorted(const DgsView& v, const DgsCarve& c, const uint32_t* order, uint32_t* tt_sorted,
                                       uint32_t* offs_sorted, uint32_t* scan_tmp, uint32_t cap, hipStream_t s);
hipError_t dgs_launch_cull_count(const DgsView& v, const DgsCarve& c, hipStream_t s);
hipError_t dgs_launch_cull_offsets(const DgsView& v, const DgsCarve& c, uint32_t* total_tight, hipStream_t s,
                                   bool counts_in_order = false);
hipError_t dgs_launch_duplicate_tight(const DgsView& v, const DgsCarve& c, const uint32_t* order, uint32_t cap,
                                      hipStream_t s);
hipError_t dgs_launch_composite_fwd(const DgsView& v, const DgsCarve& c, const float* bg, float* out_color,
                                    float* out_depth, hipStream_t s, uint32_t* checksum = nullptr);
hipError_t dgs_launch_composite_bwd(const DgsView& v, const DgsCarve& c, const float* bg, const float* dL_dpix,
                                    const float* dL_ddepth, float* contrib, hipStream_t s, int k0 = 0, int k1 = -1);
hipError_t dgs_launch_geometry_bwd(const DgsProblem& p, const DgsView& v, const DgsCarve& c, const DgsBackwardIO& io,
                                   const float* contrib, float* sums, double* partials, hipStream_t s, int phases,
                                   int g_begin, int g_end, int k0 = 0, int k1 = -1);
hipError_t dgs_launch_blur_loss(const float* sub, const float* gt, int K, int C, int HW, float lambda_t,
                                const float* lambda_dev, const float* scale, float* blur, float* dsub, float* losses,
                                hipStream_t s);

hipError_t dgs_launch_depth_sort(uint32_t* keys, uint32_t* keys_alt, uint32_t* order, uint32_t* order_alt, int K,
                                 uint32_t P, uint32_t* tmp, uint32_t* vis_dst, uint32_t* wide_flag, hipStream_t s,
                                 const uint32_t* cnt_src = nullptr, uint32_t* cnt_dst = nullptr,
                                 bool drop_invisible = false);
size_t dgs_depth_sort_tmp_words(int K, uint32_t P);
size_t dgs_scan_tmp_words(uint64_t n);
size_t dgs_sort_tmp_words(uint64_t n);
int dgs_sort_num_passes(int begin_bit, int end_bit);
int dgs_geometry_bwd_blocks(int P);
