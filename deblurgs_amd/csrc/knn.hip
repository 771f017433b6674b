// knn.hip -- mean squared distance to the three nearest neighbours of every point: the
// initial Gaussian scales of create_from_pcd (scene/gaussian_model.py:148-156) come from simple-knn's distCUDA2
// (submodules/simple-knn/spatial.cu:15-26 -> simple_knn.cu:176-221).  The reference's search is exact (its box
// pruning only skips boxes that cannot hold a closer point, simple_knn.cu:138-174), so the result is the exact
// 3-NN mean regardless of the traversal; what is reproduced is that definition, the self-exclusion by index (so
// duplicate points count with distance 0) and the FLT_MAX fill when fewer than three neighbours exist.
//
// gfx950 shape: Morton-order the points with the library's own radix sort, gather them once into a contiguous
// float4 array, one AABB per 256 consecutive points.  A wave owns 64 consecutive sorted points; for every box it
// takes ONE wave-uniform decision (does any lane still need this box?) and, if so, stages the box's points in LDS
// with coalesced loads and lets every lane scan them with conflict-free broadcast reads -- the reference has each
// thread chase points[indices[i]] through global memory on its own (simple_knn.cu:165-170).
#include <float.h>

#include "dgs_common.h"

namespace {

constexpr int KNN_BOX = 256;

struct Box {
  float lo[3], hi[3];
};

__device__ __forceinline__ uint32_t spread10(uint32_t x) {  // simple_knn.cu:41-48
  x = (x | (x << 16)) & 0x030000FF;
  x = (x | (x << 8)) & 0x0300F00F;
  x = (x | (x << 4)) & 0x030C30C3;
  x = (x | (x << 2)) & 0x09249249;
  return x;
}

// bounds[0..2] = min(0, points), bounds[3..5] = max(0, points): the reference reduces with init {0,0,0}
// (simple_knn.cu:182-190), so the Morton grid always contains the origin
__global__ void __launch_bounds__(256)
bounds_kernel(int P, const float* __restrict__ pts, float* __restrict__ partial, int nblk, float* __restrict__ bounds,
              uint32_t* __restrict__ ticket) {
  __shared__ float s[6][256];
  float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += nblk * 256)
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const float v = pts[3 * (size_t)i + a];
      lo[a] = fminf(lo[a], v);
      hi[a] = fmaxf(hi[a], v);
    }
#pragma unroll
  for (int a = 0; a < 3; a++) {
    s[a][threadIdx.x] = lo[a];
    s[3 + a][threadIdx.x] = hi[a];
  }
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off)
#pragma unroll
      for (int a = 0; a < 3; a++) {
        s[a][threadIdx.x] = fminf(s[a][threadIdx.x], s[a][threadIdx.x + off]);
        s[3 + a][threadIdx.x] = fmaxf(s[3 + a][threadIdx.x], s[3 + a][threadIdx.x + off]);
      }
    __syncthreads();
  }
  __shared__ bool is_last;
  if (threadIdx.x == 0) {
    for (int a = 0; a < 6; a++) partial[6 * blockIdx.x + a] = s[a][0];
    __threadfence();
    is_last = (atomicAdd(ticket, 1u) == (uint32_t)nblk - 1);
  }
  __syncthreads();
  if (is_last && threadIdx.x < 6) {  // min / max are order-independent: deterministic
    __threadfence();
    const int a = threadIdx.x;
    float r = 0.0f;
    for (int b = 0; b < nblk; b++) {
      const float v = reinterpret_cast<volatile float*>(partial)[6 * b + a];
      r = a < 3 ? fminf(r, v) : fmaxf(r, v);
    }
    bounds[a] = r;
    if (a == 0) *ticket = 0;
  }
}

__global__ void __launch_bounds__(256)
morton_kernel(int P, const float* __restrict__ pts, const float* __restrict__ bounds, uint64_t* __restrict__ keys,
              uint32_t* __restrict__ vals) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  uint32_t code = 0;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float lo = bounds[a], hi = bounds[3 + a];
    const float t = ((pts[3 * (size_t)i + a] - lo) / (hi - lo)) * 1023.0f;  // simple_knn.cu:52-58
    code |= spread10((uint32_t)t) << a;                                      // NaN / inf (flat axis) -> 0 or sat
  }
  keys[i] = code;
  vals[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(256)
gather_points_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order,
                     float4* __restrict__ sorted) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const uint32_t j = order[i];
  sorted[i] = make_float4(pts[3 * (size_t)j], pts[3 * (size_t)j + 1], pts[3 * (size_t)j + 2], 0.0f);
}

__global__ void __launch_bounds__(KNN_BOX)
box_kernel(int P, const float4* __restrict__ sorted, Box* __restrict__ boxes) {
  __shared__ float s[6][KNN_BOX];
  const int i = blockIdx.x * KNN_BOX + threadIdx.x;
  const bool in = i < P;
  const float4 p = in ? sorted[i] : make_float4(0, 0, 0, 0);
  s[0][threadIdx.x] = in ? p.x : FLT_MAX;  s[3][threadIdx.x] = in ? p.x : -FLT_MAX;
  s[1][threadIdx.x] = in ? p.y : FLT_MAX;  s[4][threadIdx.x] = in ? p.y : -FLT_MAX;
  s[2][threadIdx.x] = in ? p.z : FLT_MAX;  s[5][threadIdx.x] = in ? p.z : -FLT_MAX;
  __syncthreads();
  for (int off = KNN_BOX / 2; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off)
#pragma unroll
      for (int a = 0; a < 3; a++) {
        s[a][threadIdx.x] = fminf(s[a][threadIdx.x], s[a][threadIdx.x + off]);
        s[3 + a][threadIdx.x] = fmaxf(s[3 + a][threadIdx.x], s[3 + a][threadIdx.x + off]);
      }
    __syncthreads();
  }
  if (threadIdx.x < 3) {
    boxes[blockIdx.x].lo[threadIdx.x] = s[threadIdx.x][0];
    boxes[blockIdx.x].hi[threadIdx.x] = s[3 + threadIdx.x][0];
  }
}

__device__ __forceinline__ void update3(float px, float py, float pz, const float4 q, float* best) {
  const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;   // simple_knn.cu:123-136
  float dist = dx * dx + dy * dy + dz * dz;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const float t = best[j];
    const bool sw = t > dist;
    best[j] = sw ? dist : t;
    dist = sw ? t : dist;
  }
}

__device__ __forceinline__ float box_dist(const Box& b, float px, float py, float pz) {  // simple_knn.cu:110-120
  float dx = 0, dy = 0, dz = 0;
  if (px < b.lo[0] || px > b.hi[0]) dx = fminf(fabsf(px - b.lo[0]), fabsf(px - b.hi[0]));
  if (py < b.lo[1] || py > b.hi[1]) dy = fminf(fabsf(py - b.lo[1]), fabsf(py - b.hi[1]));
  if (pz < b.lo[2] || pz > b.hi[2]) dz = fminf(fabsf(pz - b.lo[2]), fabsf(pz - b.hi[2]));
  return dx * dx + dy * dy + dz * dz;
}

__global__ void __launch_bounds__(256)
knn_kernel(int P, int nboxes, const float4* __restrict__ sorted, const uint32_t* __restrict__ order,
           const Box* __restrict__ boxes, float* __restrict__ out) {
  __shared__ float4 s_pts[4][KNN_BOX];
  const int lane = dgs_lane(), w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const bool in = idx < P;
  const float4 me = in ? sorted[idx] : make_float4(0, 0, 0, 0);
  float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  if (in)
    for (int i = max(0, idx - 3); i <= min(P - 1, idx + 3); i++)   // simple_knn.cu:147-152
      if (i != idx) update3(me.x, me.y, me.z, sorted[i], best);
  const float reject = best[2];
  best[0] = best[1] = best[2] = FLT_MAX;
  for (int b = 0; b < nboxes; b++) {
    const Box box = boxes[b];   // uniform address: scalar loads
    const float d = box_dist(box, me.x, me.y, me.z);
    const bool need = in && !(d > reject || d > best[2]);   // simple_knn.cu:161-163
    if (__ballot(need) == 0ull) continue;
    const int first = b * KNN_BOX, cnt = min(KNN_BOX, P - first);
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < cnt; j += 64) s_pts[w][j] = sorted[first + j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // every lane scans the staged box (a lane that did not need it cannot get a wrong answer from extra candidates)
    const int self = idx - first;
    for (int j = 0; j < cnt; j++) {
      const float4 q = s_pts[w][j];
      if (j != self) update3(me.x, me.y, me.z, q, best);
    }
  }
  if (in) out[order[idx]] = (best[0] + best[1] + best[2]) / 3.0f;
}

constexpr size_t A = 256;
size_t up(size_t x) { return (x + A - 1) / A * A; }
constexpr int BOUNDS_BLOCKS = 256;

struct KnnCarve {
  uint64_t *keys, *keys_alt;
  uint32_t *vals, *vals_alt, *sort_tmp, *ticket;
  float4* sorted;
  Box* boxes;
  float *partial, *bounds;
  size_t total;
};
KnnCarve carve_knn(int P, char* base) {
  KnnCarve c;
  size_t o = 0;
  const size_t n = (size_t)P;
  // (offsets are added as integers: the size query carves from a null base, and pointer arithmetic on null is undefined)
  auto take = [&](size_t bytes) { char* p = reinterpret_cast<char*>(reinterpret_cast<uintptr_t>(base) + o); o += up(bytes); return p; };
  c.keys = reinterpret_cast<uint64_t*>(take(n * 8));
  c.keys_alt = reinterpret_cast<uint64_t*>(take(n * 8));
  c.vals = reinterpret_cast<uint32_t*>(take(n * 4));
  c.vals_alt = reinterpret_cast<uint32_t*>(take(n * 4));
  c.sort_tmp = reinterpret_cast<uint32_t*>(take(dgs_sort_tmp_words(n) * 4));
  c.sorted = reinterpret_cast<float4*>(take(n * 16));
  c.boxes = reinterpret_cast<Box*>(take(((n + KNN_BOX - 1) / KNN_BOX) * sizeof(Box)));
  c.partial = reinterpret_cast<float*>(take(BOUNDS_BLOCKS * 6 * 4));
  c.bounds = reinterpret_cast<float*>(take(6 * 4));
  c.ticket = reinterpret_cast<uint32_t*>(take(4));
  c.total = o;
  return c;
}

}  // namespace

extern int dgs_fail_arg(const char* msg);
extern int dgs_fail_hip(hipError_t e, const char* where);

extern "C" {

size_t dgs_knn_tmp_bytes(int32_t P) { return carve_knn(P < 0 ? 0 : P, nullptr).total; }

int dgs_knn_mean_dist2(int32_t P, const float* points, float* mean_dist2, void* tmp, dgs_stream_t stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (P < 0 || (P > 0 && (points == nullptr || mean_dist2 == nullptr || tmp == nullptr)))
    return dgs_fail_arg("knn_mean_dist2: bad argument");
  if (P == 0) return DGS_OK;
  const KnnCarve c = carve_knn(P, reinterpret_cast<char*>(tmp));
  hipError_t e = hipMemsetAsync(c.ticket, 0, 4, s);
  if (e != hipSuccess) return dgs_fail_hip(e, "knn memset");
  const int nblk = (int)fmin((double)BOUNDS_BLOCKS, (double)((P + 255) / 256));
  hipLaunchKernelGGL(bounds_kernel, dim3(nblk), dim3(256), 0, s, P, points, c.partial, nblk, c.bounds, c.ticket);
  const dim3 grid((P + 255) / 256);
  hipLaunchKernelGGL(morton_kernel, grid, dim3(256), 0, s, P, points, c.bounds, c.keys, c.vals);
  int in_alt = 0;
  e = dgs_launch_sort(c.keys, c.vals, c.keys_alt, c.vals_alt, (uint64_t)P, 0, 30, c.sort_tmp, &in_alt, s);
  if (e != hipSuccess) return dgs_fail_hip(e, "knn sort");
  const uint32_t* order = in_alt ? c.vals_alt : c.vals;
  hipLaunchKernelGGL(gather_points_kernel, grid, dim3(256), 0, s, P, points, order, c.sorted);
  const int nboxes = (P + KNN_BOX - 1) / KNN_BOX;
  hipLaunchKernelGGL(box_kernel, dim3(nboxes), dim3(KNN_BOX), 0, s, P, c.sorted, c.boxes);
  hipLaunchKernelGGL(knn_kernel, grid, dim3(256), 0, s, P, nboxes, c.sorted, order, c.boxes, mean_dist2);
  e = hipGetLastError();
  return e == hipSuccess ? DGS_OK : dgs_fail_hip(e, "knn_mean_dist2");
}

}  // extern "C"
