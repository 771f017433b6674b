// Micro-benchmark: achieved fp32 VALU rate (scalar v_fma_f32 vs packed v_pk_fma_f32) at a given occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int PK>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  float x[8]; v2f y[8];
  for (int i = 0; i < 8; i++) { x[i] = threadIdx.x * 0.001f + i; y[i] = (v2f){x[i], x[i] + 1}; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (PK) y[i] = y[i] * (v2f){a, a} + (v2f){b, b};
      else x[i] = fmaf(x[i], a, b);
    }
  }
  float s = 0; for (int i = 0; i < 8; i++) s += PK ? (y[i].x + y[i].y) : x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 4096 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pk = 0; pk < 2; pk++) for (int blocks_per_cu = 1; blocks_per_cu <= 8; blocks_per_cu *= 2) {
    int iters = 20000; int grid = 256 * blocks_per_cu;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (pk) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
      else hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double inst = (double)grid * 4 * iters * 8;  // wave-instructions
    double flops = inst * 64 * 2 * (pk ? 2 : 1);
    printf("pk=%d waves/SIMD=%d  %.3f ms  %.1f TFLOP/s  wave-instr/cycle/SIMD(@2.4GHz)=%.3f\n", pk, blocks_per_cu, ms, flops / ms / 1e9, inst / (ms * 1e-3 * 2.4e9 * 1024));
  }
  return 0;
}
