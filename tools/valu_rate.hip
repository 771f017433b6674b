// Micro-benchmark: what one SIMD of this chip sustains, in wave64 instructions per CORE CLOCK CYCLE, for each
// instruction class the compositing kernels are made of, at W = 1..8 waves per SIMD.
//
// Round 4: round 3 took the LONGEST wave of a launch whose blocks were not guaranteed to spread
// evenly over the CUs, and got "costs" the real kernels then beat (busy fractions above 1).  Now
//   * occupancy is FORCED: a block is 4 waves (one per SIMD) and allocates 160 KB / W of LDS, so exactly W blocks fit a
//     CU; the grid is CUs x W x ROUNDS blocks, so every CU stays full for several rounds whatever the dispatch order;
//   * the rate comes from the SUM of all waves' s_memtime deltas: W waves share a SIMD for their whole life, so
//     instr / cycle / SIMD = W x (instructions per wave) / (mean wave cycles);
//   * every class is its own kernel symbol (valu_rate_k<CLASS>), so the same binary under
//         rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES -- tools/valu_rate 8
//     gives the counter-side figure  SQ_INSTS_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)  per class
//     (profiles/make_valu_peak.py puts the two side by side).
//   * CAVEAT found with that cross-check (round 4): s_memtime does not tick at the core clock on this chip under load --
//     the tool's own "instr/cycle/SIMD" come out ~2.3x above the counter figures (1.03 for v_mul_f32 where the counters say
//     0.44 and the SIMD-32 peak is 0.5).  The ABSOLUTE rates are the counters' (profiles/valu_peak_r04.json); this program's
//     printed figures are good for comparing classes and occupancies with each other only.
//   * operand forms are separated: v_fma_f32 with three distinct VGPR sources, v_fmac_f32 (two sources + the
//     accumulator), v_mul_f32 / v_add_f32 (two sources), because the register-file read ports, not the ALU, set the rate.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/valu_rate tools/valu_rate.hip && tools/valu_rate > profiles/valu_classes_r04.txt
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

enum Cls { FMA3 = 0, FMAC, MUL2, ADD2, PKFMA, EXP, RCP, CMPSEL, DPPADD, MOV, READLANE, MADU64, NCLS };
static const char* NAMES[NCLS] = {"v_fma_f32 (3 VGPR sources)", "v_fmac_f32 (2 sources + acc)", "v_mul_f32 (2 sources)",
                                  "v_add_f32 (2 sources)", "v_pk_fma_f32", "v_exp_f32", "v_rcp_f32",
                                  "v_cmp+v_cndmask (2 instr)", "v_add_f32_dpp", "v_mov_b32", "v_readlane_b32",
                                  "v_mad_u64_u32"};
constexpr int UNROLL = 8;   // independent chains per loop iteration
constexpr int ITERS = 4000;

template <int C>
__global__ void __launch_bounds__(256) valu_rate_k(float* out, unsigned long long* acc, float a, float b) {
  extern __shared__ float dyn[];   // occupancy limiter only
  float x[UNROLL], y[UNROLL];
  for (int i = 0; i < UNROLL; i++) {
    x[i] = threadIdx.x * 0.001f + i;
    y[i] = x[i] + 1.0f;
  }
  float c0 = a * 1.5f, c1 = b * 0.25f;
  asm volatile("" : "+v"(c0), "+v"(c1));
  unsigned long long sink = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < UNROLL; i++) {
      if (C == FMA3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c0), "v"(c1));
      if (C == FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(c0), "v"(c1));
      if (C == MUL2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c0));
      if (C == ADD2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c1));
      if (C == PKFMA) {
        typedef float v2f __attribute__((ext_vector_type(2)));
        v2f p = {x[i], y[i]};
        const v2f aa = {c0, c0}, bb = {c1, c1};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(aa), "v"(bb));
        x[i] = p.x;
        y[i] = p.y;
      }
      if (C == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
      if (C == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
      if (C == CMPSEL)
        asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %2, %0, vcc" : "+v"(x[i]) : "v"(c0), "v"(c1) : "vcc");
      if (C == DPPADD)
        asm volatile("v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(y[i]));
      if (C == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(y[i]));
      if (C == READLANE) {
        int s;
        asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(x[i]));
        sink += (unsigned)s;
      }
      if (C == MADU64) {
        unsigned long long r;
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "=v"(r) : "v"(__float_as_uint(x[i])), "v"(48u) : "vcc");
        sink += r;
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = (float)sink;
  for (int i = 0; i < UNROLL; i++) s += x[i] + y[i];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s + dyn[threadIdx.x & 7] * 0.0f;
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&acc[0], t1 - t0);
    atomicAdd(&acc[1], 1ull);
  }
}

template <int C>
void run(float* out, unsigned long long* acc, int W, int n_cu) {
  const int rounds = 4, grid = n_cu * W * rounds;
  // exactly W blocks per CU: 160 KB of LDS per CU, a little slack for the allocation granule
  const size_t lds = (size_t)(160 * 1024) / W - (W > 1 ? 1024 : 2048);
  hipFuncSetAttribute(reinterpret_cast<const void*>(valu_rate_k<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  double best = 1e30;
  for (int rep = 0; rep < 3; rep++) {
    hipMemset(acc, 0, 16);
    hipLaunchKernelGGL(valu_rate_k<C>, dim3(grid), dim3(256), lds, 0, out, acc, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    unsigned long long h[2] = {0, 0};
    hipMemcpy(h, acc, 16, hipMemcpyDeviceToHost);
    const double mean = (double)h[0] / (double)(h[1] ? h[1] : 1);
    if (mean < best) best = mean;
  }
  const double per_wave = (double)ITERS * UNROLL * (C == CMPSEL ? 2 : 1);
  printf("%-30s waves/SIMD %d  cycles/instr %.2f  instr/cycle/SIMD %.3f  (kernel valu_rate_k<%d>)\n", NAMES[C], W,
         best / (per_wave * W), per_wave * W / best, C);
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount;
  float* out;
  unsigned long long* acc;
  hipMalloc(&out, (size_t)256 * n_cu * 8 * 4 * 4);
  hipMalloc(&acc, 16);
  const int only = argc > 1 ? atoi(argv[1]) : 0;
  printf("# %s, %d CUs; cycles = s_memtime deltas, MEAN over all waves of a launch that keeps every CU at exactly W blocks "
         "(4 waves each) for 4 rounds\n", prop.gcnArchName, n_cu);
  for (int W = 1; W <= 8; W *= 2) {
    if (only && W != only) continue;
    run<FMA3>(out, acc, W, n_cu);
    run<FMAC>(out, acc, W, n_cu);
    run<MUL2>(out, acc, W, n_cu);
    run<ADD2>(out, acc, W, n_cu);
    run<PKFMA>(out, acc, W, n_cu);
    run<EXP>(out, acc, W, n_cu);
    run<RCP>(out, acc, W, n_cu);
    run<CMPSEL>(out, acc, W, n_cu);
    run<DPPADD>(out, acc, W, n_cu);
    run<MOV>(out, acc, W, n_cu);
    run<READLANE>(out, acc, W, n_cu);
    run<MADU64>(out, acc, W, n_cu);
  }
  return 0;
}
