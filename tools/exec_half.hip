// Micro-benchmark: does gfx950 issue a wave64 VALU / LDS instruction in ONE pass of its SIMD-32
// when one 32-lane half of EXEC is zero?
//
// The compositing kernels run at ~39 % lane efficiency: a (Gaussian, quadrant) pass is issued for all 64 lanes of the
// quadrant (8 x 8 pixels, lanes 0-31 = rows 0-3, lanes 32-63 = rows 4-7) whenever ANY pixel of the quadrant can be reached.
// If the hardware skipped an all-zero half, culling at 8 x 4 half-quadrant granularity and setting EXEC per pass would
// halve the issue cost of the passes that reach one half only.  This program answers the "if".
//
// One kernel, the EXEC mask is a kernel argument (an SGPR pair moved into EXEC around a loop written entirely in
// assembly: the compiler never sees a divergent branch).  Every class runs with the same grid under each mask and is timed
// with HIP events (s_memtime does not tick at the core clock under load on this chip, tools/valu_rate.hip); the figure of
// merit is time(mask) / time(all 64 lanes).  0.5 for "low 32" / "high 32" = the empty half is skipped; 1.0 = it is not.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/exec_half tools/exec_half.hip && tools/exec_half > profiles/exec_half_r06.txt
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

enum Cls { FMA = 0, MUL, EXP, PKFMA, CNDMASK, DSREAD, DSWRITE, DSREAD128, CND_SGPR, CND_REFRESH, CND_INDEP, CND_CONSTSRC, CND_E64VCC, CND_1IN8, CND_SALU1IN8, FMA_7IN8, NCLS };
static const char* NAMES[NCLS] = {"v_fma_f32", "v_mul_f32", "v_exp_f32", "v_pk_fma_f32", "v_cndmask_b32",
                                  "ds_read_b32", "ds_write_b32", "ds_read_b128", "v_cndmask_e64 sgpr", "v_cmp+8 cndmask",
                                  "v_cndmask indep dst", "v_cndmask x,0,c", "v_cndmask_e64 vcc", "1 cndmask(vcc)+7 fma",
                                  "s_or vcc+cnd+7 fma", "7 fma (8 counted)"};
constexpr int ITERS = 2000;   // loop trips; 32 instructions of the class per trip


template <int C>
__global__ void __launch_bounds__(256) exec_half_k(float* out, unsigned long long mask, float a, float b) {
  __shared__ float lds[256 * 4 + 64];
  float x0 = threadIdx.x * 0.001f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  float c0 = a * 1.5f, c1 = b * 0.25f;
  typedef float v2f __attribute__((ext_vector_type(2)));
  typedef float v4f __attribute__((ext_vector_type(4)));
  lds[threadIdx.x] = x0;
  lds[threadIdx.x + 256] = x1;
  lds[threadIdx.x + 512] = x2;
  lds[threadIdx.x + 768] = x3;
  __syncthreads();
  const unsigned addr = (threadIdx.x & 63) * 4u;           // conflict-free 4-byte lanes
  const unsigned addr16 = (threadIdx.x & 63) * 16u;        // 16 bytes per lane
#define LOOP_HEAD                      \
  "s_mov_b64 s[20:21], exec\n\t"       \
  "s_mov_b64 exec, %[m]\n\t"           \
  "s_movk_i32 s22, %[n]\n\t"           \
  "1:\n\t"
#define LOOP_TAIL                      \
  "s_sub_u32 s22, s22, 1\n\t"          \
  "s_cmp_lg_u32 s22, 0\n\t"            \
  "s_cbranch_scc1 1b\n\t"              \
  "s_waitcnt lgkmcnt(0)\n\t"           \
  "s_mov_b64 exec, s[20:21]\n\t"
#define VREGS "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
#define CLOB "s20", "s21", "s22", "scc", "vcc", "memory"
#define FMA8                                                                                                     \
  "v_fma_f32 %0, %0, %[c0], %[c1]\n\t v_fma_f32 %1, %1, %[c0], %[c1]\n\t v_fma_f32 %2, %2, %[c0], %[c1]\n\t"       \
  "v_fma_f32 %3, %3, %[c0], %[c1]\n\t v_fma_f32 %4, %4, %[c0], %[c1]\n\t v_fma_f32 %5, %5, %[c0], %[c1]\n\t"       \
  "v_fma_f32 %6, %6, %[c0], %[c1]\n\t v_fma_f32 %7, %7, %[c0], %[c1]\n\t"
  if (C == FMA)
    asm volatile(LOOP_HEAD FMA8 FMA8 FMA8 FMA8 LOOP_TAIL
                 : VREGS
                 : [m] "s"(mask), [n] "n"(ITERS), [c0] "v"(c0), [c1] "v"(c1)
                 : CLOB);
#define ONE8(op)                                                                                                   \
  op " %0, %0, %[c0]\n\t" op " %1, %1, %[c0]\n\t" op " %2, %2, %[c0]\n\t" op " %3, %3, %[c0]\n\t" op " %4, %4, %[c0]\n\t" \
     op " %5, %5, %[c0]\n\t" op " %6, %6, %[c0]\n\t" op " %7, %7, %[c0]\n\t"
  if (C == MUL)
    asm volatile(LOOP_HEAD ONE8("v_mul_f32") ONE8("v_mul_f32") ONE8("v_mul_f32") ONE8("v_mul_f32") LOOP_TAIL
                 : VREGS
                 : [m] "s"(mask), [n] "n"(ITERS), [c0] "v"(c0)
                 : CLOB);
#define UN8(op) \
  op " %0, %0\n\t" op " %1, %1\n\t" op " %2, %2\n\t" op " %3, %3\n\t" op " %4, %4\n\t" op " %5, %5\n\t" op " %6, %6\n\t" op " %7, %7\n\t"
  if (C == EXP)
    asm volatile(LOOP_HEAD UN8("v_exp_f32") UN8("v_exp_f32") UN8("v_exp_f32") UN8("v_exp_f32") LOOP_TAIL
                 : VREGS
                 : [m] "s"(mask), [n] "n"(ITERS)
                 : CLOB);
  if (C == PKFMA) {
    v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
    const v2f aa = {c0, c0}, bb = {c1, c1};
#define PK8                                                                                                      \
  "v_pk_fma_f32 %0, %0, %[a], %[b]\n\t v_pk_fma_f32 %1, %1, %[a], %[b]\n\t v_pk_fma_f32 %2, %2, %[a], %[b]\n\t"  \
  "v_pk_fma_f32 %3, %3, %[a], %[b]\n\t v_pk_fma_f32 %4, %4, %[a], %[b]\n\t v_pk_fma_f32 %5, %5, %[a], %[b]\n\t"  \
  "v_pk_fma_f32 %6, %6, %[a], %[b]\n\t v_pk_fma_f32 %7, %7, %[a], %[b]\n\t"
    asm volatile(LOOP_HEAD PK8 PK8 PK8 PK8 LOOP_TAIL
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                 : [m] "s"(mask), [n] "n"(ITERS), [a] "v"(aa), [b] "v"(bb)
                 : CLOB);
    x0 = p0.x + p0.y + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.x;
  }
  if (C == CNDMASK) {
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x0), "v"(c0) : "vcc");
    asm volatile(LOOP_HEAD ONE8("v_cndmask_b32") ONE8("v_cndmask_b32") ONE8("v_cndmask_b32") ONE8("v_cndmask_b32") LOOP_TAIL
                 : VREGS
                 : [m] "s"(mask), [n] "n"(ITERS), [c0] "v"(c0)
                 : CLOB);
  }
#define RD8(op)                                                                                                   \
  op " %0, %[ad]\n\t" op " %1, %[ad] offset:256\n\t" op " %2, %[ad] offset:512\n\t" op " %3, %[ad] offset:768\n\t" \
     op " %4, %[ad] offset:1024\n\t" op " %5, %[ad] offset:1280\n\t" op " %6, %[ad] offset:1536\n\t" op " %7, %[ad] offset:1792\n\t" \
     "s_waitcnt lgkmcnt(4)\n\t"
  if (C == DSREAD)
    asm volatile(LOOP_HEAD RD8("ds_read_b32") RD8("ds_read_b32") RD8("ds_read_b32") RD8("ds_read_b32") LOOP_TAIL
                 : VREGS
                 : [m] "s"(mask), [n] "n"(ITERS), [ad] "v"(addr)
                 : CLOB);
#define WR8(op)                                                                                                   \
  op " %[ad], %0\n\t" op " %[ad], %1 offset:256\n\t" op " %[ad], %2 offset:512\n\t" op " %[ad], %3 offset:768\n\t" \
     op " %[ad], %4 offset:1024\n\t" op " %[ad], %5 offset:1280\n\t" op " %[ad], %6 offset:1536\n\t" op " %[ad], %7 offset:1792\n\t" \
     "s_waitcnt lgkmcnt(4)\n\t"
  if (C == DSWRITE)
    asm volatile(LOOP_HEAD WR8("ds_write_b32") WR8("ds_write_b32") WR8("ds_write_b32") WR8("ds_write_b32") LOOP_TAIL
                 : VREGS
                 : [m] "s"(mask), [n] "n"(ITERS), [ad] "v"(addr)
                 : CLOB);
  if (C == DSREAD128) {
    v4f q0 = {x0, x1, x2, x3}, q1 = q0, q2 = q0, q3 = q0;
#define RQ4 \
  "ds_read_b128 %0, %[ad]\n\t ds_read_b128 %1, %[ad] offset:1024\n\t ds_read_b128 %2, %[ad] offset:2048\n\t ds_read_b128 %3, %[ad] offset:3072\n\t s_waitcnt lgkmcnt(2)\n\t"
    asm volatile(LOOP_HEAD RQ4 RQ4 RQ4 RQ4 RQ4 RQ4 RQ4 RQ4 LOOP_TAIL
                 : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
                 : [m] "s"(mask), [n] "n"(ITERS), [ad] "v"(addr16)
                 : CLOB);
    x0 = q0.x + q1.y + q2.z + q3.w;
  }
  // v_cndmask variants (round 6: the VOP2 / VCC chain above came out ~9x slower than v_fma_f32 -- which operand form is slow?)
#define CS8 \
  "v_cndmask_b32_e64 %0, %0, %[c0], s[24:25]\n\t v_cndmask_b32_e64 %1, %1, %[c0], s[24:25]\n\t v_cndmask_b32_e64 %2, %2, %[c0], s[24:25]\n\t" \
  "v_cndmask_b32_e64 %3, %3, %[c0], s[24:25]\n\t v_cndmask_b32_e64 %4, %4, %[c0], s[24:25]\n\t v_cndmask_b32_e64 %5, %5, %[c0], s[24:25]\n\t" \
  "v_cndmask_b32_e64 %6, %6, %[c0], s[24:25]\n\t v_cndmask_b32_e64 %7, %7, %[c0], s[24:25]\n\t"
  if (C == CND_SGPR) {
    asm volatile("v_cmp_lt_f32_e64 s[24:25], %0, %1" : : "v"(x0), "v"(c0) : "s24", "s25");
    asm volatile(LOOP_HEAD CS8 CS8 CS8 CS8 LOOP_TAIL
                 : VREGS
                 : [m] "s"(mask), [n] "n"(ITERS), [c0] "v"(c0)
                 : CLOB, "s24", "s25");
  }
#define CR8 "v_cmp_lt_f32 vcc, %0, %[c0]\n\t" ONE8("v_cndmask_b32")
  if (C == CND_REFRESH)   // 9 instructions per group: the rate printed counts 8
    asm volatile(LOOP_HEAD CR8 CR8 CR8 CR8 LOOP_TAIL
                 : VREGS
                 : [m] "s"(mask), [n] "n"(ITERS), [c0] "v"(c0)
                 : CLOB);
#define CI8 \
  "v_cndmask_b32 %0, %[c1], %[c0]\n\t v_cndmask_b32 %1, %[c1], %[c0]\n\t v_cndmask_b32 %2, %[c1], %[c0]\n\t v_cndmask_b32 %3, %[c1], %[c0]\n\t" \
  "v_cndmask_b32 %4, %[c1], %[c0]\n\t v_cndmask_b32 %5, %[c1], %[c0]\n\t v_cndmask_b32 %6, %[c1], %[c0]\n\t v_cndmask_b32 %7, %[c1], %[c0]\n\t"
  if (C == CND_INDEP) {
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x0), "v"(c0) : "vcc");
    asm volatile(LOOP_HEAD CI8 CI8 CI8 CI8 LOOP_TAIL
                 : VREGS
                 : [m] "s"(mask), [n] "n"(ITERS), [c0] "v"(c0), [c1] "v"(c1)
                 : CLOB);
  }
#define CZ8 \
  "v_cndmask_b32 %0, 0, %0\n\t v_cndmask_b32 %1, 0, %1\n\t v_cndmask_b32 %2, 0, %2\n\t v_cndmask_b32 %3, 0, %3\n\t" \
  "v_cndmask_b32 %4, 0, %4\n\t v_cndmask_b32 %5, 0, %5\n\t v_cndmask_b32 %6, 0, %6\n\t v_cndmask_b32 %7, 0, %7\n\t"
  if (C == CND_CONSTSRC) {
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x0), "v"(c0) : "vcc");
    asm volatile(LOOP_HEAD CZ8 CZ8 CZ8 CZ8 LOOP_TAIL
                 : VREGS
                 : [m] "s"(mask), [n] "n"(ITERS)
                 : CLOB);
  }
#define CE8 \
  "v_cndmask_b32_e64 %0, %0, %[c0], vcc\n\t v_cndmask_b32_e64 %1, %1, %[c0], vcc\n\t v_cndmask_b32_e64 %2, %2, %[c0], vcc\n\t" \
  "v_cndmask_b32_e64 %3, %3, %[c0], vcc\n\t v_cndmask_b32_e64 %4, %4, %[c0], vcc\n\t v_cndmask_b32_e64 %5, %5, %[c0], vcc\n\t" \
  "v_cndmask_b32_e64 %6, %6, %[c0], vcc\n\t v_cndmask_b32_e64 %7, %7, %[c0], vcc\n\t"
  if (C == CND_E64VCC) {
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x0), "v"(c0) : "vcc");
    asm volatile(LOOP_HEAD CE8 CE8 CE8 CE8 LOOP_TAIL
                 : VREGS
                 : [m] "s"(mask), [n] "n"(ITERS), [c0] "v"(c0)
                 : CLOB);
  }
  // what the compositing pass bodies look like: one select among plain arithmetic
#define F7                                                                                                      \
  "v_fma_f32 %1, %1, %[c0], %[c1]\n\t v_fma_f32 %2, %2, %[c0], %[c1]\n\t v_fma_f32 %3, %3, %[c0], %[c1]\n\t"      \
  "v_fma_f32 %4, %4, %[c0], %[c1]\n\t v_fma_f32 %5, %5, %[c0], %[c1]\n\t v_fma_f32 %6, %6, %[c0], %[c1]\n\t"      \
  "v_fma_f32 %7, %7, %[c0], %[c1]\n\t"
#define M1 "v_cndmask_b32 %0, %0, %[c0]\n\t" F7
  if (C == CND_1IN8) {
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x0), "v"(c0) : "vcc");
    asm volatile(LOOP_HEAD M1 M1 M1 M1 LOOP_TAIL
                 : VREGS
                 : [m] "s"(mask), [n] "n"(ITERS), [c0] "v"(c0), [c1] "v"(c1)
                 : CLOB);
  }
#define M2 "s_or_b64 vcc, s[24:25], s[26:27]\n\t v_cndmask_b32 %0, %0, %[c0]\n\t" F7
  if (C == CND_SALU1IN8) {
    asm volatile("v_cmp_lt_f32_e64 s[24:25], %0, %1\n\t v_cmp_gt_f32_e64 s[26:27], %0, %1" : : "v"(x0), "v"(c0) : "s24", "s25", "s26", "s27");
    asm volatile(LOOP_HEAD M2 M2 M2 M2 LOOP_TAIL
                 : VREGS
                 : [m] "s"(mask), [n] "n"(ITERS), [c0] "v"(c0), [c1] "v"(c1)
                 : CLOB, "s24", "s25", "s26", "s27");
  }
  if (C == FMA_7IN8)
    asm volatile(LOOP_HEAD F7 F7 F7 F7 LOOP_TAIL
                 : VREGS
                 : [m] "s"(mask), [n] "n"(ITERS), [c0] "v"(c0), [c1] "v"(c1)
                 : CLOB);
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + lds[threadIdx.x];
}

struct Mask {
  const char* name;
  unsigned long long m;
};
static const Mask MASKS[] = {
    {"all 64 lanes", 0xFFFFFFFFFFFFFFFFull},  {"low 32 (rows 0-3)", 0x00000000FFFFFFFFull},
    {"high 32 (rows 4-7)", 0xFFFFFFFF00000000ull}, {"low 16", 0x000000000000FFFFull},
    {"lanes 16-31", 0x00000000FFFF0000ull},    {"16 low + 16 high", 0x0000FFFF0000FFFFull},
    {"even lanes", 0x5555555555555555ull},     {"lane 0 only", 0x1ull},
    {"lane 0 + lane 32", 0x0000000100000001ull}};

template <int C>
void run(float* out, int n_cu) {
  const int grid = n_cu * 8 * 4;   // 8 blocks of 4 waves per CU = 8 waves per SIMD, four rounds
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  double full = 0;
  for (const Mask& mk : MASKS) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(exec_half_k<C>, dim3(grid), dim3(256), 0, 0, out, mk.m, 1.0001f, 0.5f);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    if (mk.m == ~0ull) full = best;
    const int per_trip = (C == DSREAD128) ? 32 : 32;
    const double instr = (double)grid * 4 * ITERS * per_trip;
    printf("%-14s EXEC = %-20s %8.3f ms   %6.2f G wave-instr/s   time / full = %.3f\n", NAMES[C], mk.name, best,
           instr / best * 1e-6, best / full);
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount;
  float* out;
  hipMalloc(&out, (size_t)256 * n_cu * 32 * 4);
  printf("# %s, %d CUs, clock %d kHz; 8 waves per SIMD; each line: best of 3 timed launches (HIP events)\n", prop.gcnArchName,
         n_cu, prop.clockRate);
  run<FMA>(out, n_cu);
  run<MUL>(out, n_cu);
  run<EXP>(out, n_cu);
  run<PKFMA>(out, n_cu);
  run<CNDMASK>(out, n_cu);
  run<DSREAD>(out, n_cu);
  run<DSWRITE>(out, n_cu);
  run<DSREAD128>(out, n_cu);
  run<CND_SGPR>(out, n_cu);
  run<CND_REFRESH>(out, n_cu);
  run<CND_INDEP>(out, n_cu);
  run<CND_CONSTSRC>(out, n_cu);
  run<CND_E64VCC>(out, n_cu);
  run<CND_1IN8>(out, n_cu);
  run<CND_SALU1IN8>(out, n_cu);
  run<FMA_7IN8>(out, n_cu);
  hipFree(out);
  return 0;
}
