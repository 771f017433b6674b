// Micro-benchmark: what one SIMD of this chip sustains, in wave64 instructions per CORE CLOCK CYCLE, for each
// instruction class the compositing kernels are made of -- plain fp32 FMA, packed FMA, the two transcendentals, compare +
// select, DPP adds, v_permlane32_swap / v_permlane16_swap, v_mov, broadcast ds_read_b128 -- at 1..8 waves per SIMD.
// Cycles are MEASURED (s_memtime deltas around the instruction stream of every wave, the longest wave of the launch
// counts), not derived from a nominal clock: the issue-slot cost of a class is cycles / instructions, and the
// "issue-slot-weighted" VALU roofline of a kernel is  sum_class(count_class * cost_class) / (SIMDs * kernel cycles)
// (profiles/make_valu.py combines this table with the ISA census and the PMC instruction counts).
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/valu_rate tools/valu_rate.hip && tools/valu_rate > profiles/valu_classes_r03.txt
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

enum Cls { FMA = 0, PKFMA, EXP, RCP, CMPSEL, DPPADD, PERM32, PERM16, MOV, DSREAD, NCLS };
static const char* NAMES[NCLS] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_rcp_f32", "v_cmp+v_cndmask (2 instr)",
                                  "v_add_f32_dpp", "v_permlane32_swap", "v_permlane16_swap", "v_mov_b32",
                                  "ds_read_b128 (broadcast)"};
constexpr int UNROLL = 8;   // independent chains per loop iteration

template <int C>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int iters, float a, float b) {
  __shared__ float4 lds[64];
  if (threadIdx.x < 64) lds[threadIdx.x] = make_float4(a, b, a, b);
  __syncthreads();
  float x[UNROLL];
  float y[UNROLL];
  for (int i = 0; i < UNROLL; i++) {
    x[i] = threadIdx.x * 0.001f + i;
    y[i] = x[i] + 1.0f;
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < UNROLL; i++) {
      if (C == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
      if (C == PKFMA) {
        typedef float v2f __attribute__((ext_vector_type(2)));
        v2f p = {x[i], y[i]};
        const v2f aa = {a, a}, bb = {b, b};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(aa), "v"(bb));
        x[i] = p.x;
        y[i] = p.y;
      }
      if (C == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
      if (C == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
      if (C == CMPSEL)
        asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %2, %0, vcc" : "+v"(x[i]) : "v"(a), "v"(b) : "vcc");
      if (C == DPPADD)
        asm volatile("v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(y[i]));
      if (C == PERM32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[i]), "+v"(y[i]));
      if (C == PERM16) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x[i]), "+v"(y[i]));
      if (C == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(y[i]));
      if (C == DSREAD) {
        float4 q;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q) : "v"((unsigned)((i & 7) * 16)));
        x[i] += q.x;
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < UNROLL; i++) s += x[i] + y[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) atomicMax(cyc, t1 - t0);
}

template <int C>
void run(float* out, unsigned long long* cyc, int waves_per_simd, int n_cu) {
  const int iters = 4000, grid = n_cu * waves_per_simd;   // 256 threads = 4 waves = one per SIMD of a CU
  double best = 1e30;
  for (int rep = 0; rep < 3; rep++) {
    hipMemset(cyc, 0, 8);
    hipLaunchKernelGGL(k<C>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    if ((double)c < best) best = (double)c;
  }
  const double per_wave = (double)iters * UNROLL * (C == CMPSEL ? 2 : 1);
  // one SIMD executed waves_per_simd waves' streams within `best` cycles (all blocks resident at once: grid = CUs x
  // waves per SIMD, 4 waves per block)
  printf("%-28s waves/SIMD %d  cycles/instr %.2f  instr/cycle/SIMD %.3f\n", NAMES[C], waves_per_simd,
         best / (per_wave * waves_per_simd), per_wave * waves_per_simd / best);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount;
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, (size_t)256 * n_cu * 8 * 4);
  hipMalloc(&cyc, 8);
  printf("# %s, %d CUs; cycles = s_memtime deltas (core clock as the shader sees it), longest wave of the launch\n",
         prop.gcnArchName, n_cu);
  for (int w = 1; w <= 8; w *= 2) {
    run<FMA>(out, cyc, w, n_cu);
    run<PKFMA>(out, cyc, w, n_cu);
    run<EXP>(out, cyc, w, n_cu);
    run<RCP>(out, cyc, w, n_cu);
    run<CMPSEL>(out, cyc, w, n_cu);
    run<DPPADD>(out, cyc, w, n_cu);
    run<PERM32>(out, cyc, w, n_cu);
    run<PERM16>(out, cyc, w, n_cu);
    run<MOV>(out, cyc, w, n_cu);
    run<DSREAD>(out, cyc, w, n_cu);
  }
  return 0;
}
