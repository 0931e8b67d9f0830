// Measurement tool: where does the dispatcher put the workgroups of a small grid?  Every workgroup records (XCC id, HW_ID) and spins
// for `ticks` of the 100 MHz clock so that all of them are resident together.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/census tools/micro/census.hip && /tmp/census 576 256 0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void census_kernel(unsigned* out, long long ticks) {
  extern __shared__ char pad[];
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x + 0] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 576, threads = argc > 2 ? atoi(argv[2]) : 256, lds_kb = argc > 3 ? atoi(argv[3]) : 0;
  unsigned* d; hipMalloc(&d, wgs * 8);
  std::vector<unsigned> h(2 * wgs);
  hipFuncSetAttribute((const void*)census_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(census_kernel, dim3(wgs), dim3(threads), (size_t)lds_kb * 1024, 0, d, 3000LL);   // 30 us
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), d, wgs * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, int> per;
  for (int i = 0; i < wgs; ++i) per[((h[2 * i] & 0xF) << 16) | (h[2 * i + 1] & 0xFF00)]++;     // xcc | se / sh / cu fields of HW_ID
  std::map<int, int> hist;
  for (auto& kv : per) hist[kv.second]++;
  printf("%d workgroups of %d threads, %d KB dynamic LDS: %zu distinct CUs;", wgs, threads, lds_kb, per.size());
  for (auto& kv : hist) printf("  %d CUs hold %d", kv.second, kv.first);
  printf("\n");
  return 0;
}
