// Probe: where does ds_write_addtid_b32 put its data?  (address = M0[15:0] + offset + 4 * TID -- TID within the wave or
// within the workgroup?  ordering against a following ds_read of the same wave?)
//   hipcc --offload-arch=gfx950 -O2 -o tools/addtid_probe tools/addtid_probe.hip && tools/addtid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) probe(unsigned* out, unsigned* readback, int wait) {
  __shared__ unsigned lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned base = (unsigned)(size_t)(&lds[0]) + 2048u * (unsigned)w;   // wave w: bytes [2048 w, 2048 w + 256)
  const unsigned val = 0x1000u * (unsigned)w + (threadIdx.x & 63);
  unsigned rb;
  if (wait)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:16\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b32 %0, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=v"(rb) : "v"(val), "s"(base), "v"(base + 16u + 4u * (threadIdx.x & 63)) : "m0", "memory");
  else
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:16\n\tds_read_b32 %0, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=v"(rb) : "v"(val), "s"(base), "v"(base + 16u + 4u * (threadIdx.x & 63)) : "m0", "memory");
  readback[threadIdx.x] = rb;
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 256) out[i] = lds[i];
}

int main() {
  unsigned *out, *rb;
  hipMalloc(&out, 2048 * 4);
  hipMalloc(&rb, 256 * 4);
  for (int wait = 0; wait < 2; wait++) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, out, rb, wait);
    hipDeviceSynchronize();
    std::vector<unsigned> h(2048), r(256);
    hipMemcpy(h.data(), out, 2048 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), rb, 256 * 4, hipMemcpyDeviceToHost);
    printf("wait=%d\n", wait);
    for (int w = 0; w < 4; w++) {
      int first = -1, last = -1, n = 0;
      for (int i = 0; i < 2048; i++)
        if ((h[i] >> 12) == (unsigned)w && h[i] != 0xdeadbeefu && (h[i] & 0xfff) < 64) { if (first < 0) first = i; last = i; n++; }
      printf("  wave %d: %d words landed at dword [%d, %d] (expected [%d, %d]); lane0 value at dword %d; read-back lane 5 = 0x%x (expect 0x%x)\n",
             w, n, first, last, 512 * w + 4, 512 * w + 4 + 63, first, r[64 * w + 5], 0x1000 * w + 5);
    }
  }
  return 0;
}
