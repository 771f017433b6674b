/* dgs_hip.h -- C ABI of libdgs_hip.so: the MI355X (gfx950) differentiable Gaussian-splat rasteriser for
 * DeblurGS's blur-integration loop.
 *
 * This is the drop-in boundary for the reference's pybind module `diff_gaussian_rasterization._C`
 * (/root/reference/submodules/diff-gaussian-rasterization/ext.cpp:15-19,
 *  rasterize_points.h:18-73).  Differences, all deliberate:
 *   - plain C: raw device pointers, sizes and a HIP stream handle; no torch types.  The caller owns every
 *     byte (inputs, outputs, gradients, the three state blobs and the backward scratch); the library never
 *     allocates device memory.  Blob sizes come from the dgs_*_bytes() queries, and the carving of a blob
 *     into sub-arrays is a pure function of (P, W, H, K, R, wide_records, forward_only), replayed identically by forward and backward
 *     (the reference does the same with GeometryState/ImageState/BinningState::fromChunk,
 *     rasterizer_impl.cu:155-194,389-391).
 *   - the duplicates are generated in (k, depth, index) order (a 15 M-pair sort of the Gaussians) so that the
 *     R-pair sort only has to order the bits(K*T) tile bits; sorted keys / point list are bit-identical to the
 *     reference's single (tile | depth) sort.
 *   - K >= 1 subframes per call ("K-fused"): one launch chain rasterises all K poses of a blurry view
 *     (scene/motion.py:141-143 calls render() K times).  K = 1 is exactly `_C.rasterize_gaussians`.
 *   - the forward is split in two calls around the one host read of num_rendered
 *     (rasterizer_impl.cu:286-287): dgs_forward_geometry -> caller sizes the binning blob ->
 *     dgs_forward_render.
 * Every entry point returns 0 on success or a negative DGS_E_* code; dgs_last_error() gives the text.
 * All kernels are enqueued on the given stream; nothing here synchronises unless `debug` is set.
 *   - NO PROCESS STATE (ABI 14): the library reads no environment variable and keeps no global but the thread-local
 *     text of dgs_last_error().  What outlives a call -- the side stream and events the backward of a large view forks
 *     part of its work onto, the policy for that fork, the stage timers -- lives in a DgsContext the CALLER creates,
 *     hands over in DgsProblem.context and destroys; with context = NULL every call is one launch chain on the given
 *     stream and nothing else.  (The reference keeps its state in caller tensors only, rasterize_points.cu:27-33.)
 */
#ifndef DGS_HIP_H_INCLUDED
#define DGS_HIP_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGS_ABI_VERSION 14
#define DGS_MAX_K 128 /* subframes per fused call */

#define DGS_OK 0
#define DGS_E_ARG (-1)      /* bad shape / null pointer / exclusive-argument violation */
#define DGS_E_CAPACITY (-2) /* a caller-provided blob is too small */
#define DGS_E_HIP (-3)      /* a HIP runtime error (text in dgs_last_error) */

typedef void* dgs_stream_t; /* hipStream_t */

/* ---- caller-owned context (ABI 14) ---------------------------------------------------------------------------
 * Everything the library keeps between calls.  dgs_context_create, on the CURRENT device: one non-blocking side stream,
 * DGS_MAX_BWD_PARTS + 1 events (no device memory), the options below, and the stage timers of dgs_profile_*.  A context
 * serialises the calls that use it (a mutex around each enqueue sequence); use one per host thread that enqueues
 * concurrently.  It must outlive every call it was handed to and be destroyed on a quiet device. */
#define DGS_MAX_BWD_PARTS 8
typedef struct DgsContextOptions {
  int32_t bwd_overlap; /* the compositing backward in parts, each part's row totals on the side stream next to the next
                        * part's compositing (dgs_backward): 0 = never; 1 = for large views (tile_cull, K >= 6, >= 4 M
                        * duplicates), never inside a stream capture (the default); 2 = for any view with K >= 2 (tests);
                        * 3 = as 2, also inside a stream capture (measurements only: a forked executable graph does not
                        * return its memory on ROCm 7.2, tools/graph_fork_leak.hip) */
  int32_t bwd_n_parts; /* 0 = the library's cut (K = 15: 10, 4, 1); n >= 1: bwd_parts[0..n-1] subframes per part, what is
                        * left of K after them is the last part (entries that do not fit K are dropped) */
  int32_t bwd_parts[DGS_MAX_BWD_PARTS - 1];
} DgsContextOptions;
typedef struct DgsContext DgsContext; /* opaque */
/* options = NULL: {1, 0, {}}. */
int dgs_context_create(const DgsContextOptions* options, DgsContext** out);
int dgs_context_destroy(DgsContext* ctx); /* NULL is a no-op */

/* Arguments shared by forward and backward: replaces the 22 / 25 positional arguments of
 * RasterizeGaussiansCUDA / RasterizeGaussiansBackwardCUDA (rasterize_points.cu:35-59,125-152). */
typedef struct DgsProblem {
  int32_t P;       /* Gaussians */
  int32_t D;       /* active SH degree 0..3 */
  int32_t M;       /* SH coefficients per Gaussian, (max_sh_degree+1)^2; 0 when colors_precomp is used */
  int32_t W, H;    /* image size */
  int32_t K;       /* subframes (poses) in this call, 1..DGS_MAX_K */
  float tanfovx, tanfovy;
  float scale_modifier;
  float z_near, z_far; /* z_near is plumbed but unused, as in the reference */
  int32_t use_sigmoid; /* colour activation: 0 = relu(x+0.5), 1 = sigmoid (forward.cu:63-80) */
  int32_t prefiltered;
  int32_t debug;       /* sync + check after every stage (auxiliary.h:179-186) */
  int32_t tile_cull;   /* 0: one duplicate per tile of the 3-sigma rectangle, the reference's exact lists
                        *    (rasterizer_impl.cu:76-108); 1: drop the duplicates whose Gaussian cannot reach
                        *    alpha >= 1/255 at any pixel of the tile (pairs the reference skips at every pixel,
                        *    forward.cu:356-358), so R is smaller and the outputs are unchanged.  Must be the same
                        *    in the forward and backward calls of one problem. */
  int32_t raw_params;  /* 0: opacities / scales / rotations / shs are activated values, as the reference's render() passes
                        *    them (gaussian_renderer/__init__.py:60-77).  1: they are the cloud's raw parameters and the
                        *    kernels apply the reference's activations themselves (scene/gaussian_model.py:36-50,
                        *    scene/gaussian_activation.py): opacity clamp(x, 0, 1), scale exp(x) + scale_lb, rotation
                        *    x / max(|x|, 1e-12), SH = [shs (dc, [P,1,3]) | shs_rest ([P,M-1,3])]; the backward then
                        *    returns gradients with respect to the raw parameters.  3 (= 1 | 2): the same with ONE shared
                        *    scale per Gaussian, column 0 of `scales` used for all three axes (use_isotrophic,
                        *    scene/gaussian_model.py:115-118); dL_dscales then carries the summed gradient in column 0
                        *    and zeros in columns 1, 2. */
  float scale_lb;
  int32_t wide_records; /* tile_cull only.  0 (default): a duplicate is ONE 64-bit word, tile | Gaussian | emission index,
                         *    whenever the three fit (DgsLayout.pack_*), and the value arrays stay unused; 1: always keep
                         *    the key (tile << 32 | emission index) and the Gaussian index in separate arrays.  Part of
                         *    the blob carving: must be the same in the forward and backward calls of one problem. */
  int32_t forward_only; /* 1: an inference call -- the reference's test.py:117 / render_spiral.py:29 call render() under
                         *    no_grad --: nothing is kept for a backward.  The compositing does not store final_T /
                         *    n_contrib, preprocess does not store cov3D / the colour-activation mask, image_state needs
                         *    only dgs_image_state_bytes_forward_only(W,H,K) bytes (the tile ranges; a full-size blob is
                         *    accepted too), and dgs_backward* on such a problem returns DGS_E_ARG.  Outputs are
                         *    bit-identical to forward_only = 0. */
  /* inputs, device pointers, fp32 contiguous */
  const float* means3D;        /* [P,3] */
  const float* shs;            /* [P,M,3] or NULL */
  const float* shs_rest;       /* raw_params = 1 only: [P,M-1,3] (then shs is [P,1,3]); else NULL */
  const float* colors_precomp; /* [P,3]   or NULL   (exactly one of shs / colors_precomp) */
  const float* opacities;      /* [P]; may be NULL in the backward calls of a raw_params = 0 problem (the reference's backward
                                * is not handed them either, rasterize_points.cu:125-152) */
  const float* scales;         /* [P,3] or NULL */
  const float* rotations;      /* [P,4] or NULL */
  const float* cov3D_precomp;  /* [P,6] or NULL   (exactly one of scales+rotations / cov3D_precomp) */
  const float* viewmatrix;     /* [K,4,4] row-vector convention, flat index 4r+c (auxiliary.h:58-66) */
  const float* projmatrix;     /* [K,4,4] full projection = view @ proj */
  const float* campos;         /* [K,3] */
  const float* bg;             /* [3] */
  /* state blobs kept alive by the caller between forward and backward */
  void* geom_state;    size_t geom_bytes;    /* >= dgs_geom_state_bytes(P,K) */
  void* image_state;   size_t image_bytes;   /* >= dgs_image_state_bytes(W,H,K) */
  void* binning_state; size_t binning_bytes; /* >= dgs_binning_state_bytes(R,W,H,K); unused by dgs_forward_geometry */
  DgsContext* context; /* caller-owned (dgs_context_create) or NULL: no side stream, no stage timers -- every call is one
                        * launch chain on `stream`.  Not part of the blob carving; may differ between forward and backward. */
} DgsProblem;

typedef struct DgsForwardOut {
  float* out_color;   /* [K,3,H,W] */
  float* out_depth;   /* [K,1,H,W], or NULL: the caller does not consume the depth image (the default training loss
                       * never reads it) -- the depth channel then drops out of the compositing */
  int32_t* radii;     /* [K,P] */
  uint32_t* num_rendered_host; /* pinned host uint32[2] (dgs_forward: [5]): dgs_forward_geometry enqueues an async copy
                                * of {R, overflow}; overflow != 0 means the duplicate count exceeded 32 bits */
  uint32_t* drop_counter;      /* dgs_forward only, optional device word owned by the caller: incremented by every
                                * forward whose capacity overflowed and copied to num_rendered_host[4].  A caller that
                                * replays a captured graph reads its running value instead of one flag per replay. */
  uint32_t* status_dev;        /* dgs_forward only, optional DEVICE uint32[4] owned by the caller and outliving the state
                                * blobs: receives the same four words as num_rendered_host[0..3] ({count, its high word,
                                * overflow flag, count the lists were built with}).  Every replay of a captured graph copies
                                * into the SAME pinned num_rendered_host block; a caller that runs several replays ahead
                                * enqueues its own copy of status_dev into a per-step pinned slot right behind each replay
                                * and so reads the words of exactly that replay. */
  const uint64_t* status_host_indirect; /* dgs_forward only, optional DEVICE uint64: the address of a device-accessible
                                * PINNED HOST uint32[5] that this forward's finalize kernel writes the five words of
                                * num_rendered_host into, read from device memory when the kernel runs.  A caller that
                                * replays a captured graph stores a different block's address there before every replay
                                * (through its own device-side scalar block): every replay then reports into a pinned
                                * slot of its own without any copy behind it. */
  uint32_t* debug_contrib_checksum; /* optional [K, H*W], tile_cull = 0 only: per pixel, the wrap-around sum of
                                * pos * 2654435761 over the 1-based positions pos (inside the tile's list) of the pairs that
                                * contribute to the pixel.  What the parity tests compare with the CPU oracle's to find the
                                * pixels where the two traversals took a different per-pair decision (an exp() ulp at one of
                                * the three thresholds of forward.cu:354-368); NULL (the product): the compositing kernel
                                * without that sum. */
} DgsForwardOut;

typedef struct DgsBackwardIO {
  uint32_t num_rendered;      /* R returned by the forward */
  const int32_t* radii;       /* [K,P] from the forward */
  const float* dL_dout_color; /* [K,3,H,W] */
  const float* dL_dout_depth; /* [K,1,H,W] or NULL (= zeros) */
  void* scratch; size_t scratch_bytes; /* >= dgs_backward_scratch_bytes(R,P,K) */
  /* gradients, fully overwritten (no zero-fill needed).  Per-Gaussian grads are summed over the K
   * subframes; the screen-space and pose grads stay per subframe because densification and the trajectory
   * consume them per subframe (train.py:188-193, scene/motion.py:248-294). */
  float* dL_dmeans3D;  /* [P,3] */
  float* dL_dmeans2D;  /* [K,P,3] NDC-scaled screen gradient, .z = 0 (backward.cu:628-629); NULL allowed with stats_* */
  float* dL_dsh;       /* [P,M,3]  (NULL when colors_precomp was used); [P,1,3] with raw_params */
  float* dL_dsh_rest;  /* raw_params = 1 only: [P,M-1,3] */
  float* dL_dcolors;   /* [P,3]    (written always; the grad of colors_precomp when that was used) */
  float* dL_dopacity;  /* [P] */
  float* dL_dscales;   /* [P,3]  (NULL when cov3D_precomp was used) */
  float* dL_drotations;/* [P,4]  (NULL when cov3D_precomp was used) */
  float* dL_dcov3D;    /* [P,6] */
  float* dL_dviewmatrix; /* [K,4,4] */
  float* dL_dprojmatrix; /* [K,4,4] */
  /* raw_params = 1 only: adds opacity_hinge_scale * d/dx hinge_l2 terms to dL_dopacity, i.e. the gradient of the
   * training loss's lambda_hinge * hinge_l2(_opacity) (utils/loss_utils.py:96-104, train.py:156-163) with
   * opacity_hinge_scale = lambda_hinge / P; 0 = off.  Saves the caller a dozen elementwise launches per step. */
  float opacity_hinge_scale;
  /* Optional fused densification statistics (train.py:188-193, scene/gaussian_model.py:456-458): when
   * stats_max_radii2D is non-NULL the per-Gaussian kernel updates the three [P] accumulators in place exactly as
   * dgs_densify_stats(dL_dmeans2D, radii, K, stats_K_total, ...) would right after this backward (same operations,
   * subframe order; nothing is touched when the forward's overflow flag is set), and dL_dmeans2D may then be NULL:
   * the [K,P,3] screen gradient is neither written nor re-read.  stats_K_total: len(render_pkgs) of the whole view
   * (0 = K). */
  float* stats_max_radii2D;
  float* stats_grad_accum;
  float* stats_denom;
  int32_t stats_K_total;
} DgsBackwardIO;

/* Byte offsets of the sub-arrays inside the three blobs (for debuggers and the parity tests).
 * Element types: rows f32[12] = {x, y, conic.x, conic.y, conic.z, opacity, r, g, b, depth, u32 (unused),
 * i32 radius}; keys u64 = ((k*T + tile) << 32) | depth_bits; ranges uint2 per (k, tile). */
typedef struct DgsLayout {
  /* geometry blob */
  size_t geom_rows;      /* f32 [K,P,12] */
  size_t cov3D;          /* f32 [P,6] */
  size_t pre_sigmoid;    /* f32 [K,P,3]  pre-activation colour (sigmoid) or 0/1 clamp mask (relu) */
  size_t tiles_touched;  /* u32 [K*P] */
  size_t point_offsets;  /* u32 [K*P] tile_cull = 0 only, by natural (k, Gaussian) index: index of the pair's first
                          * duplicate = first contribution row of the backward (duplicates are laid out in (k, depth,
                          * index) order; undefined for invisible pairs).  With tile culling the backward uses
                          * offs_tight (same order as tt_tight) and the emission index in the low key word. */
  size_t scan_tmp;       /* u32 scan block sums */
  size_t num_rendered;   /* u32 [8] status words: [0],[1] rectangle total lo/hi (tile_cull = 0), [2],[3] surviving total lo/hi
                          * (tile_cull = 1), [4] the count the lists were built with, [5] overflow flag (capacity mode) */
  size_t gsort_keys;     /* u32 [K,P] bits(depth) - bits(0.2f) (0xFFFFFFFF = invisible): keys of the segmented depth sort */
  size_t gsort_keys_alt; /* u32 [K,P] its ping-pong buffer */
  size_t gsort_vals;     /* u32 [K*P] flat (k, Gaussian) indices in (k, depth, index) order (the sort's result).  tile_cull = 1:
                          * the invisible pairs are dropped by the first sort pass -- every subframe's segment holds its
                          * visible pairs first (tt_sorted = 1) and UNDEFINED indices behind them (tt_sorted = 0) */
  size_t gsort_vals_alt; /* u32 [K*P] */
  size_t tt_sorted;      /* u32 [K*P] tiles_touched in (k, depth, index) order (tile_cull: 1 / 0 = visible / not) */
  size_t offs_sorted;    /* u32 [K*P] its exclusive prefix sum (tile_cull = 0 only) */
  size_t tt_tight;       /* u32 [K*P] tile_cull: surviving tiles per (k, Gaussian), same order */
  size_t offs_tight;     /* u32 [K*P] its exclusive prefix sum (its total is R under tile_cull) */
  size_t gsort_tmp;      /* u32 radix tables of the Gaussian sort */
  size_t cull_rec;       /* u32 [K*P,4] tile_cull: per (k, Gaussian), natural order: rectangle (minx | miny << 12 | (width-1) << 24, or
                          * bit 31 = more than 64 tiles), surviving tiles, hit bits of the rectangle's first 64 tiles (u64) */
  size_t cull_cnt;       /* u32 [K*P] tile_cull: the surviving-tile counts alone, natural order */
  size_t geom_total;
  /* image blob */
  size_t final_T;        /* f32 [K,H*W] */
  size_t n_contrib;      /* u32 [K,H*W] */
  size_t ranges;         /* u32 [K*T,2] */
  size_t image_total;
  /* binning blob */
  size_t keys_sorted;    /* u64 [R] */
  size_t point_list;     /* u32 [R] sorted Gaussian ids */
  size_t keys_unsorted;  /* u64 [R] (also the sort's ping-pong buffer) */
  size_t vals_unsorted;  /* u32 [R] */
  size_t sort_tmp;       /* u32 radix histogram table */
  size_t binning_total;
  int32_t sort_bits;     /* 32 + bits(K*T): width of the reference-compatible key */
  int32_t sort_passes;   /* digit passes of the duplicate sort (over the tile bits only) */
  /* tile_cull = 1 with wide_records = 0 only, when bits(K*T) + bits(P) + bits(R) <= 64 (else both are 0 and keys /
   * point_list are as above):
   * the key is the whole record, (tile << pack_tile_shift) | (Gaussian << pack_g_shift) | emission index, point_list /
   * vals_unsorted stay unused and the sort moves 8 instead of 12 bytes per duplicate and pass. */
  int32_t pack_g_shift;
  int32_t pack_tile_shift;
} DgsLayout;

int dgs_abi_version(void);
const char* dgs_last_error(void);
/* SHA-256 (64 hex digits) over the sources this binary was built from -- every .hip and .h file of deblurgs_amd/csrc, include/dgs_hip.h --
 * and the compiler flag table of deblurgs_amd/build.py, computed by the build and compiled in.  A loader that has the
 * sources at hand (deblurgs_amd/_lib.py) recomputes it and refuses a stale binary; bench.py prints it. */
const char* dgs_build_id(void);

size_t dgs_geom_state_bytes(int32_t P, int32_t K);
size_t dgs_image_state_bytes(int32_t W, int32_t H, int32_t K);
size_t dgs_image_state_bytes_forward_only(int32_t W, int32_t H, int32_t K); /* DgsProblem.forward_only = 1: tile ranges only */
size_t dgs_binning_state_bytes(uint64_t R, int32_t W, int32_t H, int32_t K);
size_t dgs_backward_scratch_bytes(uint64_t R, int32_t P, int32_t K);
/* wide_records: DgsProblem.wide_records of the problem the layout is for (it only decides pack_g_shift / pack_tile_shift;
 * the offsets and totals do not depend on it). */
int dgs_layout(int32_t P, int32_t W, int32_t H, int32_t K, uint64_t R, int32_t wide_records, DgsLayout* out);
/* Byte offsets inside DgsBackwardIO.scratch (for debuggers and the parity tests): contribution rows f32 [R,12] at 0
 * ([S_wx, S_wy, S_xx, S_xy, S_yy, S_w, dL_dr, dL_dg, dL_db, dL_ddepth, -, -] per duplicate, in emission order), their
 * per-(subframe, Gaussian) totals f32 [K*P,16] at *sums_offset (the same 12 columns in 64-byte slots, the last 4 floats
 * unused; natural index k*P + g; defined for visible pairs only; the
 * reference's per-Gaussian sinks follow from them as dL_dconic = -0.5 (S_xx, S_xy, S_yy), dL_dopacity = S_w / opacity,
 * backward.cu:620-637), the per-block pose-gradient partials at *partials_offset. */
int dgs_backward_scratch_layout(uint64_t R, int32_t P, int32_t K, size_t* sums_offset, size_t* partials_offset);

/* Replaces Rasterizer::forward up to the host read of num_rendered (rasterizer_impl.cu:198-287):
 * preprocess for all K subframes + prefix sum, then an async copy of R to out->num_rendered_host. */
int dgs_forward_geometry(const DgsProblem* p, const DgsForwardOut* out, dgs_stream_t stream);
/* Replaces the rest of Rasterizer::forward (rasterizer_impl.cu:289-345): duplicateWithKeys, the stable
 * radix sort, identifyTileRanges and the per-tile alpha compositing, for all K subframes. */
int dgs_forward_render(const DgsProblem* p, const DgsForwardOut* out, uint32_t num_rendered, dgs_stream_t stream);
/* Single-call forward for callers that size the duplicate arrays AHEAD of the count (from the previous step's count,
 * say): dgs_forward_geometry + dgs_forward_render without the host read in between.  binning_state must hold
 * dgs_binning_state_bytes(capacity, ...).  num_rendered_host (pinned, >= 4 words) receives asynchronously
 * [0] the true duplicate count, [1] its u32-overflow word, [2] overflow flag (count > capacity or [1] != 0: the lists
 * were NOT built, the outputs are meaningless and dgs_backward on this state returns garbage-but-in-bounds results;
 * re-run with a larger capacity), [3] the count the lists were built with, [4] the running drop_counter (if given) --
 * written by the forward's finalize kernel itself when the block is device-accessible pinned memory (no copy in the
 * launch chain), by asynchronous copies otherwise.
 * The same flag is the device word at
 * geom_state + DgsLayout.num_rendered + 20 bytes, which dgs_adam_step / dgs_densify_stats accept as `skip_flag` so that
 * a whole training step can be enqueued without any host synchronisation and still never apply a truncated gradient.
 * dgs_backward takes num_rendered = capacity for such a state. */
int dgs_forward(const DgsProblem* p, const DgsForwardOut* out, uint32_t capacity, dgs_stream_t stream);
/* dgs_forward in two parts, for callers that pipeline subframe groups over streams: dgs_forward_lists (preprocess, depth
 * order, tile culling, duplication, sort, tile ranges: the HBM-bound half) and dgs_forward_composite (the per-tile
 * compositing of the lists the first call built: the VALU-bound half).  Same arguments as dgs_forward. */
int dgs_forward_lists(const DgsProblem* p, const DgsForwardOut* out, uint32_t capacity, dgs_stream_t stream);
int dgs_forward_composite(const DgsProblem* p, const DgsForwardOut* out, uint32_t capacity, dgs_stream_t stream);
/* Replaces Rasterizer::backward (rasterizer_impl.cu:350-463).
 * With a context (DgsProblem.context) whose bwd_overlap allows it, an EAGERLY enqueued call for a large view (tile_cull,
 * K >= 6, >= 4 M duplicates) runs the compositing backward in up to three parts of the subframes and the per-pair totals
 * of every part but the last on the context's side stream, forked from and joined back into `stream` inside the call
 * (events; joined also when an enqueue fails half-way): ordering on `stream` is unchanged, results are bit-identical.
 * Not while `stream` is being captured (unless bwd_overlap = 3), not while the context's stage timers are on, not in
 * debug mode, never without a context. */
int dgs_backward(const DgsProblem* p, const DgsBackwardIO* io, dgs_stream_t stream);
/* dgs_backward in three parts, for callers that overlap the gradient all-reduce of a sharded run with the backward's
 * tail: dgs_backward_composite (compositing backward + per-(subframe, Gaussian) totals), then dgs_backward_geometry for
 * consecutive Gaussian-index chunks [g_begin, g_end) (g_begin a multiple of 256; each call writes exactly those rows of
 * the per-Gaussian outputs, so chunk i can be all-reduced on another stream while chunk i + 1 runs), then
 * dgs_backward_pose (dL_dviewmatrix / dL_dprojmatrix from the partial sums all chunks left).  Bit-identical to
 * dgs_backward for any chunking. */
int dgs_backward_composite(const DgsProblem* p, const DgsBackwardIO* io, dgs_stream_t stream);
int dgs_backward_geometry(const DgsProblem* p, const DgsBackwardIO* io, int32_t g_begin, int32_t g_end,
                          dgs_stream_t stream);
int dgs_backward_pose(const DgsProblem* p, const DgsBackwardIO* io, dgs_stream_t stream);
/* How many parts an eagerly enqueued dgs_backward / dgs_backward_composite of such a view runs its compositing in with
 * this context (1 = one launch: no context, small view, K < 6, no tile culling, or bwd_overlap = 0).  For callers that
 * choose between replaying a captured step (always one launch) and enqueueing it eagerly: deblurgs_amd/fused_step.py does. */
int32_t dgs_backward_parts(const DgsContext* ctx, int32_t K, uint64_t num_rendered, int32_t tile_cull);
/* Replaces Rasterizer::markVisible (rasterizer_impl.cu:141-153); present is bool[P] as bytes. */
int dgs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, dgs_stream_t stream);

/* Stand-alone building blocks, exported so the parity tests can pin them against numpy
 * (cub::DeviceScan::InclusiveSum / cub::DeviceRadixSort::SortPairs call sites,
 * rasterizer_impl.cu:166,188-191,283,309-314). */
size_t dgs_scan_tmp_bytes(uint64_t n);
/* total_out (optional) receives uint32[2] = {sum mod 2^32, high 32 bits of the 64-bit sum} */
int dgs_exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint64_t n, void* tmp, uint32_t* total_out,
                           dgs_stream_t stream);
size_t dgs_sort_tmp_bytes(uint64_t n);
/* Stable LSD radix sort of (u64 key, u32 value) pairs on key bits [begin_bit, end_bit).  Both buffer pairs are
 * clobbered; *result_in_alt tells which pair holds the result (0: keys/vals, 1: keys_alt/vals_alt). */
int dgs_sort_pairs(uint64_t* keys, uint32_t* vals, uint64_t* keys_alt, uint32_t* vals_alt, uint64_t n,
                   int32_t begin_bit, int32_t end_bit, void* tmp, int32_t* result_in_alt, dgs_stream_t stream);

/* The depth order of the fused rasteriser on its own (the stage between preprocess and duplicateWithKeys; the reference
 * sorts depth inside its one 64-bit cub sort, rasterizer_impl.cu:306-314): K independent stable sorts of P u32 keys,
 * key = bits(view depth) - bits(0.2f) for a visible (subframe, Gaussian) pair, 0xFFFFFFFF for an invisible one.
 * order[K*P] receives the flat indices k * P + g in (k, key, g) order; visible (optional, [K*P]) 1 / 0 in that order.
 * keys, keys_alt, order_alt are clobbered; tmp: dgs_depth_order_tmp_bytes(K, P) bytes.  Three 9-bit passes when every
 * visible key is below 2^27 - 1 (depth < 13107), a fourth one otherwise -- decided on the device. */
size_t dgs_depth_order_tmp_bytes(int32_t K, int32_t P);
int dgs_depth_order(uint32_t* keys, uint32_t* keys_alt, uint32_t* order, uint32_t* order_alt, int32_t K, int32_t P,
                    void* tmp, uint32_t* visible, dgs_stream_t stream);

/* Fused loss-gradient image (train.py:143-165, utils/loss_utils.py:17-18,80-93): from the K rendered subframes
 * and the target, produces blur = mean_k, the L1 and temporal-smoothness loss values and/or dL/dsubframes in one pass.
 * `losses` is an 8-float (32-byte) work area whose first two words receive {l1, smooth}; the totals are formed with
 * integer (2^-24 fixed-point) atomics and are therefore bitwise reproducible; a NaN / Inf among the inputs, or a total
 * beyond the fixed-point range (2^40), makes both values NaN.  Forward call: blur + losses non-null, dL_dsubframes NULL.  Backward call: losses
 * NULL, dL_dsubframes non-null, `upstream` = device pointer to the scalar dL/d(l1 + lambda_t*smooth) (NULL = 1); blur
 * is then an optional INPUT (the forward call's blur, which saves re-summing the K subframes; NULL = recompute).
 * All three outputs non-null computes everything at once. */
int dgs_blur_loss_grad(const float* subframes, const float* gt, int32_t K, int32_t C, int32_t HW, float lambda_t,
                       const float* upstream, float* blur, float* dL_dsubframes, float* losses, dgs_stream_t stream);

/* The same with lambda_t read from device memory when the kernel runs: the scheduled weight of a captured (hipGraph)
 * training step changes between replays without re-capturing. */
int dgs_blur_loss_grad_dev(const float* subframes, const float* gt, int32_t K, int32_t C, int32_t HW,
                           const float* lambda_t_dev, const float* upstream, float* blur, float* dL_dsubframes,
                           float* losses, dgs_stream_t stream);

/* Multi-GPU runs, "subframes" sharding (new work, the reference is single-GPU): the loss block of ONE RANK, which
 * holds K_local consecutive subframes [K_local,C,HW] of a view's K_total.  blur [C,HW] = the view's blur image (mean over
 * all K_total subframes, after the ranks' partial sums have been all-reduced), prev_last / next_first [C,HW] = the boundary
 * subframes of the neighbouring ranks (NULL at the ends of the view).  Writes dL/dsubframes of the local subframes (the
 * formula of dgs_blur_loss_grad with K = K_total) and, in the 8-float work area `losses`: [0] mean |blur - gt| (the same
 * on every rank), [1] this rank's share of the temporal-smoothness value (the differences whose LEFT frame it holds,
 * over E (K_total - 1)): the caller sums the shares over the ranks.  Same deterministic fixed-point totals.  K_local <= 32. */
int dgs_blur_loss_slice_grad(const float* subframes, const float* prev_last, const float* next_first, const float* blur,
                             const float* gt, int32_t K_local, int32_t K_total, int32_t C, int32_t HW, float lambda_t,
                             float* dL_dsubframes, float* losses, dgs_stream_t stream);

/* A few words (n_words <= 4096) copied BY A KERNEL on `stream`: dst and src may each be device memory or pinned host
 * memory (hipHostMalloc / torch pin_memory / hipHostRegister: resolved through hipHostGetDevicePointer; the kernel stores
 * with system scope and fences, so a pinned destination is complete when the stream reaches the caller's next event).  What a captured training step's per-step scalars travel
 * with: an asynchronous host-to-device copy in front of every graph launch costs a hand-over between the copy engine and
 * the compute queue (tens of microseconds); a kernel in front of the graph costs one launch.  Pageable host memory falls
 * back to hipMemcpyAsync. */
int dgs_copy_words(void* dst_dev, const void* src, int32_t n_words, dgs_stream_t stream);

/* Densification-statistic consumers of the rasteriser's per-subframe outputs (train.py:188-193 with
 * scene/gaussian_model.py:456-458), for all K subframes in one pass and in subframe order:
 *   visible = radii[k] > 0;  max_radii2D = max(max_radii2D, radii[k]);
 *   xyz_gradient_accum += || viewspace_grad[k][:, :2] ||;  denom += 1/K_total.
 * viewspace_grad is the [K,P,3] gradient of the means2D carrier, radii is [K,P]; the three accumulators are [P].
 * K_total = len(render_pkgs) of the whole view (train.py:192): K itself (pass 0) unless the view's subframes are
 * split over ranks and this call covers only a rank's share. */
int dgs_densify_stats(const float* viewspace_grad, const int32_t* radii, int32_t K, int32_t K_total, int32_t P,
                      float* max_radii2D, float* xyz_gradient_accum, float* denom, const uint32_t* skip_flag,
                      dgs_stream_t stream);

/* ---- optimiser step and densification of the Gaussian cloud ----------------------------------
 * Multi-tensor Adam: all parameter groups in one launch.  Replaces torch.optim.Adam(l, lr=0.0, eps=1e-15).step()
 * over the six per-Gaussian groups (scene/gaussian_model.py:170-195, train.py:203-208) with the same arithmetic
 * (torch/optim/adam.py, single-tensor path: lerp, mul+addcmul, sqrt/div/add, addcdiv; bias corrections in double).
 * `step` is the 1-based count of THIS update (state["step"] after its increment).  A group whose grad is NULL is
 * skipped, like a parameter whose .grad is None.  clip_value > 0 clamps the gradient first
 * (torch.nn.utils.clip_grad_value_, train.py:204-205); the gradient buffer itself is left untouched.
 * skip_flag (optional device word, see dgs_forward): non-zero = the launch leaves parameters and moments untouched. */
#define DGS_ADAM_MAX_GROUPS 16
typedef struct DgsAdamGroup {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  uint64_t numel;
  double lr;      /* python floats stay double up to the point where torch casts them (step_size, 1 - beta) */
  int32_t step;
} DgsAdamGroup;
int dgs_adam_step(const DgsAdamGroup* groups, int32_t n_groups, double beta1, double beta2, double eps,
                  double clip_value, const uint32_t* skip_flag, dgs_stream_t stream);

/* densify_and_prune (scene/gaussian_model.py:436-448 = densify_and_clone :419-434, densify_and_split :389-417,
 * prune_points :336-349 with the optimiser-state surgery of :315-334 / :359-387) as plan + apply.
 * Arrays of the cloud, in the order xyz[P,3], f_dc[P,3], f_rest[P,n_rest], opacity[P,1], scaling[P,3],
 * rotation[P,4] (raw parameters), with the two Adam moments of each (exp_avg / exp_avg_sq may be NULL in the
 * source = no optimiser state yet = zeros). */
typedef struct DgsCloudArrays {
  float* param[6];
  float* exp_avg[6];
  float* exp_avg_sq[6];
} DgsCloudArrays;
/* For captured (hipGraph) training steps: the per-group scalars that depend on the step count and the learning rate --
 * out[2 i] = -(lr_i / (1 - beta1^step_i)), out[2 i + 1] = sqrt(1 - beta2^step_i), exactly as dgs_adam_step forms them --
 * are computed on the host by dgs_adam_scalars, copied by the caller into device memory before every replay, and read
 * from there (dev_scalars, same group order) by the kernel that dgs_adam_step_dev enqueues; the lr / step fields of the
 * groups passed to dgs_adam_step_dev itself are not used. */
int dgs_adam_scalars(const DgsAdamGroup* groups, int32_t n_groups, double beta1, double beta2, float* out);
int dgs_adam_step_dev(const DgsAdamGroup* groups, int32_t n_groups, double beta1, double beta2, double eps,
                      double clip_value, const uint32_t* skip_flag, const float* dev_scalars, dgs_stream_t stream);

/* Multi-GPU runs (new work, the reference is single-GPU): the local half of a direct reduce-scatter over
 * point-to-point transfers.  `own` [n] is this rank's shard of the gradient bucket, `recv` [world, stride] holds the copy
 * received from every peer in row s = its rank (row `rank` is not read); own <- ((row_0 + row_1) + row_2) + ... with `own`
 * standing in for row `rank`, added IN RANK ORDER so every rank computes the same bits for a given set of inputs, then
 * divided by `divisor` (1 = sum, world = mean).  stride >= n, world <= 64. */
int dgs_rank_ordered_sum(const float* recv, uint64_t stride, float* own, uint64_t n, int32_t world, int32_t rank,
                         float divisor, dgs_stream_t stream);

size_t dgs_densify_tmp_bytes(int32_t P);
/* plan: per Gaussian, grads = xyz_gradient_accum / denom (NaN -> 0); clone if |grads| >= grad_threshold and
 * max(exp(scaling)+scale_lb) <= size_threshold (= percent_dense * extent); split if grads >= grad_threshold and
 * max(...) > size_threshold; every Gaussian whose clamp(opacity, 0, 1) < min_opacity is pruned together with its
 * clone / children (they inherit its opacity).  flags and offsets are u32 [4,P] (keep, clone, split, split-selected
 * before the opacity prune) and their exclusive scans; counts_host (pinned, u32[4]) receives the four totals by an
 * async copy: {n_keep, n_clone, n_split, m_all}.  The new cloud has n_keep + n_clone + 2 n_split Gaussians. */
int dgs_densify_plan(int32_t P, const float* xyz_gradient_accum, const float* denom, const float* scaling,
                     const float* opacity, float grad_threshold, float size_threshold, float min_opacity,
                     float scale_lb, int32_t isotropic, uint32_t* flags, uint32_t* offsets, uint32_t* counts_dev,
                     uint32_t* counts_host, void* tmp, dgs_stream_t stream);
/* apply: writes the new cloud in the reference's order [surviving originals | clones | children copy 0 | children
 * copy 1]; clones and children get zero Adam moments; a child is xyz + R(rotation) (std * z), scaling
 * log(max(std / 1.6 - scale_lb, 0.001)) with std = exp(scaling) + scale_lb, and z = noise[c * m_all + rank] the
 * caller's standard-normal samples [2 m_all, 3] (the reference draws torch.normal(0, stds) for every selected
 * Gaussian and copy before the opacity prune).  isotropic != 0 (both calls): std = exp(scaling[:, 0]) + scale_lb for all
 * three axes (use_isotrophic clouds): size test, child offsets and all three child scaling columns use it; survivors
 * and clones copy the raw rows unchanged.  counts = the host copy {n_keep, n_clone, n_split, m_all}. */
int dgs_densify_apply(int32_t P, int32_t n_rest, const uint32_t* counts, const uint32_t* flags, const uint32_t* offsets,
                      const DgsCloudArrays* src, const DgsCloudArrays* dst, const float* noise, float scale_lb,
                      int32_t isotropic, dgs_stream_t stream);

/* ---- initialisation: mean squared distance to the 3 nearest neighbours -------------------------
 * Replaces simple_knn._C.distCUDA2 (submodules/simple-knn/spatial.cu:15-26, simple_knn.cu:138-221), which
 * create_from_pcd uses for the initial scales (scene/gaussian_model.py:148-156): mean_dist2[i] = mean of the three
 * smallest |p_j - p_i|^2 over j != i (exact search; duplicates count with distance 0; fewer than three neighbours
 * leave FLT_MAX terms, i.e. +inf, like the reference).  points is [P,3] fp32, tmp >= dgs_knn_tmp_bytes(P). */
size_t dgs_knn_tmp_bytes(int32_t P);
int dgs_knn_mean_dist2(int32_t P, const float* points, float* mean_dist2, void* tmp, dgs_stream_t stream);

/* Pose path of the blur-integration loop on device: Bezier curves in se(3) evaluated at the K
 * subframe times nu, se3_exp_map, and the three camera tensors render() reads -- scene/bezier.py:54-83,
 * utils/pytorch3d_functions.py:373-457, scene/motion.py:248-294, scene/cameras.py:63-74 -- as one kernel, and its
 * backward from dL/d{world_view, full_proj} to the control points and nu.  ctrl_* are [C+1,3] (one curve),
 * proj is the transposed projection matrix [4,4] (row-vector convention), outputs are [K,4,4], [K,4,4], [K,3].
 * quaternion != 0: curve_type "quarternion_cartesian" (scene/motion.py:191-194,242-246): ctrl_rot is [C+1,4], a
 * quaternion curve (x, y, z, w) that is normalised and turned into the c2w rotation, ctrl_trans the camera origin. */
/* Subframe times of one view from its alignment parameters (scene/motion.py:209-219):
 * nu = sort(clamp(cat(0, sigmoid(raw) [+ uniform / n_subframes - 1 / (2 n_subframes)], 1), 0, 1)), f values from the
 * f - 2 raw ones (uniform: optional [f-2] samples of U(0,1), the reference's curve_random_sample); src[r] = index of
 * the candidate that landed at sorted position r (stable ascending sort).  Backward: dL/draw from dL/dnu (zero where
 * the clamp was active, end points carry no parameter).  f <= DGS_MAX_K. */
int dgs_alignment_forward(const float* raw, const float* uniform, int32_t f, int32_t n_subframes, float* nu,
                          int32_t* src, dgs_stream_t stream);
int dgs_alignment_backward(const float* raw, const float* uniform, int32_t f, int32_t n_subframes, const int32_t* src,
                           const float* dL_dnu, float* dL_draw, dgs_stream_t stream);
size_t dgs_pose_scratch_bytes(int32_t K);
int dgs_pose_forward(const float* ctrl_trans, const float* ctrl_rot, const float* nu, const float* proj, int32_t C,
                     int32_t K, int32_t quaternion, float* view, float* full, float* campos, dgs_stream_t stream);
int dgs_pose_backward(const float* ctrl_trans, const float* ctrl_rot, const float* nu, const float* proj, int32_t C,
                      int32_t K, int32_t quaternion, const float* dL_dview, const float* dL_dfull, void* scratch,
                      float* dL_dctrl_trans, float* dL_dctrl_rot, float* dL_dnu, dgs_stream_t stream);

/* The cloud's activations as the raw_params kernels evaluate them -- clamp(opacity, 0, 1), exp(scaling) + scale_lb,
 * rotation / max(|rotation|, 1e-12): the reference's get_opacity / get_scaling / get_rotation getters
 * (scene/gaussian_model.py:114-137, scene/gaussian_activation.py:29-52) on device, bit-identical to what
 * dgs_forward_geometry uses with DgsProblem.raw_params = 1.  Any output pointer may be NULL (with its input). */
int dgs_cloud_activations(int32_t P, const float* scaling, const float* rotation, const float* opacity, float scale_lb,
                          float* out_scaling, float* out_rotation, float* out_opacity, dgs_stream_t stream);

/* Stage timing with HIP events recorded on the caller's stream (bench.py's roofline leg); the timers belong to the context
 * the timed calls are handed (DgsProblem.context). */
#define DGS_STAGE_PREPROCESS 0
#define DGS_STAGE_SCAN 1
#define DGS_STAGE_DUPLICATE 2
#define DGS_STAGE_SORT 3
#define DGS_STAGE_RANGES 4
#define DGS_STAGE_COMPOSITE_FWD 5
#define DGS_STAGE_COMPOSITE_BWD 6
#define DGS_STAGE_GEOMETRY_BWD 7
#define DGS_STAGE_DEPTH_ORDER 8 /* sort of the (k, Gaussian) pairs by depth that precedes the duplication */
#define DGS_STAGE_TILE_CULL 9   /* tile_cull: per-slot ellipse test + count, and the scan of the counts */
#define DGS_STAGE_CONTRIB_REDUCE 10 /* backward: per-(subframe, Gaussian) totals of the contribution rows (the second
                                     * stage of the atomics-free reduction; no counterpart in the reference) */
#define DGS_STAGE_COUNT 11
int dgs_profile_enable(DgsContext* ctx, int32_t on);
int dgs_profile_reset(DgsContext* ctx);
/* Synchronises on the recorded events; ms[i] = summed duration of stage i, calls[i] = launches timed. */
int dgs_profile_read(DgsContext* ctx, float* ms, int32_t* calls, int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* DGS_HIP_H_INCLUDED */
