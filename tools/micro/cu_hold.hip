// Measurement tool (never part of libvtx.so): workgroups that HOLD compute units the way an RCCL collective's channels do while it is
// in flight.  One 256-thread workgroup per held CU (a wave on every SIMD: the 512-thread, 256-register GEMM workgroups cannot share
// that CU), launched on a side stream; they sleep-poll a device flag and leave when it is set or after `max_seconds`.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/micro/libcuhold.so tools/micro/cu_hold.hip      (tools/cu_contention.py does this)
#include <hip/hip_runtime.h>

__global__ void __launch_bounds__(256) cu_hold_kernel(const int* stop, long long max_ticks, unsigned* census) {
  if (threadIdx.x == 0) {
    census[2 * blockIdx.x + 0] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    census[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID (cu / sh / se fields)
  }
  const long long t0 = wall_clock64();                                          // 100 MHz
  while (wall_clock64() - t0 < max_ticks) {
    if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
    __builtin_amdgcn_s_sleep(127);
  }
}

extern "C" int cu_hold_launch(int n_wg, double max_seconds, const int* stop, unsigned* census, void* stream) {
  if (n_wg <= 0) return 0;
  hipLaunchKernelGGL(cu_hold_kernel, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, stop, (long long)(max_seconds * 1e8), census);
  return (int)hipGetLastError();
}
