// Does a captured graph that forks onto a second stream lose device memory on this runtime -- per capture / instantiate /
// destroy cycle, or per launch?  Plain HIP, no torch: capture [A on s] -> event -> [B on s2] -> event -> [C on s], instantiate,
// launch `reps` times, destroy; print the device memory in use as the cycles go by.
//   hipcc --offload-arch=gfx950 -O2 -o tools/graph_fork_leak tools/graph_fork_leak.hip
//   tools/graph_fork_leak <mode> [cycles] [launches per cycle] [MB touched per kernel]
// mode 0: no fork (B on s)   1: fork, side stream non-blocking   2: fork, side stream with default flags
//      3: fork and join, but no kernel on the side stream
//      5: like 1, and the two buffers the kernels touch are allocated before every capture and freed after the destroy
//      6: like 0 (no fork), buffers allocated and freed per cycle
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void touch(float* p, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] * 0.5f + v;
}

static double used_mb() {
  size_t fr = 0, tot = 0;
  CK(hipMemGetInfo(&fr, &tot));
  return (double)(tot - fr) / (1 << 20);
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 1;
  const int cycles = argc > 2 ? atoi(argv[2]) : 100;
  const int reps = argc > 3 ? atoi(argv[3]) : 3;
  const size_t n = (size_t)(argc > 4 ? atoi(argv[4]) : 64) << 18;   // floats
  hipStream_t s, s2;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, mode == 2 ? hipStreamDefault : hipStreamNonBlocking));
  hipEvent_t e1, e2;
  CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
  float *a, *b;
  CK(hipMalloc(&a, n * 4));
  CK(hipMalloc(&b, n * 4));
  CK(hipMemset(a, 0, n * 4));
  CK(hipMemset(b, 0, n * 4));
  CK(hipDeviceSynchronize());
  if (mode >= 5) {
    CK(hipFree(a));
    CK(hipFree(b));
  }
  printf("mode %d: start %.1f MB in use\n", mode, used_mb());
  for (int c = 1; c <= cycles; c++) {
    if (mode >= 5) {
      CK(hipMalloc(&a, n * 4));
      CK(hipMalloc(&b, n * 4));
    }
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, s, a, n, 1.0f);
    if (mode == 0 || mode == 6) {
      hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, s, b, n, 2.0f);
    } else {
      CK(hipEventRecord(e1, s));
      CK(hipStreamWaitEvent(s2, e1, 0));
      if (mode != 3) hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, s2, b, n, 2.0f);
      hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, s, a, n, 3.0f);
      CK(hipEventRecord(e2, s2));
      CK(hipStreamWaitEvent(s, e2, 0));
    }
    hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, s, a, n, 4.0f);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    if (mode >= 5) {
      CK(hipFree(a));
      CK(hipFree(b));
    }
    if (c % (cycles / 5 > 0 ? cycles / 5 : 1) == 0) printf("  cycle %4d: %.1f MB in use\n", c, used_mb());
  }
  if (mode >= 5) {
    CK(hipMalloc(&a, n * 4));
    CK(hipMalloc(&b, n * 4));
  }
  // launches only
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, s, a, n, 1.0f);
  if (mode != 0 && mode != 6) {
    CK(hipEventRecord(e1, s));
    CK(hipStreamWaitEvent(s2, e1, 0));
    if (mode != 3) hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, s2, b, n, 2.0f);
    hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, s, a, n, 3.0f);
    CK(hipEventRecord(e2, s2));
    CK(hipStreamWaitEvent(s, e2, 0));
  }
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  const double before = used_mb();
  for (int r = 0; r < 2000; r++) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  printf("  2000 launches of one executable graph: %.1f -> %.1f MB in use\n", before, used_mb());
  return 0;
}
