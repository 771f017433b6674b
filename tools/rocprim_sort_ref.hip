// Reference point only (NOT part of the product): how fast does AMD's own rocPRIM radix sort run the same
// (u64 key, u32 value) x 63M x 49-bit problem on this GPU?  hipcc --offload-arch=gfx950 -O3 -o /tmp/rps tools/rocprim_sort_ref.hip
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char** argv) {
  size_t n = argc > 1 ? atoll(argv[1]) : 63000000;
  int bits = argc > 2 ? atoi(argv[2]) : 49;
  std::vector<uint64_t> hk(n);
  uint64_t x = 88172645463325252ull;
  for (size_t i = 0; i < n; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; hk[i] = x & ((1ull << bits) - 1); }
  uint64_t *k0, *k1; uint32_t *v0, *v1;
  hipMalloc(&k0, n * 8); hipMalloc(&k1, n * 8); hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
  hipMemcpy(k0, hk.data(), n * 8, hipMemcpyHostToDevice);
  hipMemset(v0, 0, n * 4);
  size_t tmp_bytes = 0; void* tmp = nullptr;
  rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, bits);
  hipMalloc(&tmp, tmp_bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int it = 0; it < 6; it++) {
    hipMemcpy(k0, hk.data(), n * 8, hipMemcpyHostToDevice);
    hipEventRecord(e0);
    rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, bits);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it > 0 && ms < best) best = ms;
  }
  printf("rocprim radix_sort_pairs n=%zu bits=%d tmp=%.1f MB best=%.3f ms\n", n, bits, tmp_bytes / 1e6, best);
  return 0;
}
