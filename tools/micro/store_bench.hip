// Per-CU global-store throughput vs access shape (how many distinct rows one wave-instruction touches).
// hipcc --offload-arch=gfx950 -O3 -o store_bench store_bench.hip && ./store_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

// Each workgroup (512 threads) writes a [256 rows x 512 B] tile `reps` times (different tiles), row stride ld bytes.
// SEG = contiguous bytes one wave-instruction writes per row (128, 256, 512 -> 8, 4, 2 rows per instruction).
// Measured on MI355X: 80 GB/s per CU with 8 workgroups, 70 GB/s with 64, 25.7 GB/s (6.6 TB/s total) with 256,
// independent of SEG -- the per-CU store rate collapses only when every CU bursts at once.
template <int SEG>
__global__ __launch_bounds__(512) void store_kernel(char* out, long ld, int reps, int tiles_n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPR = SEG / 16;            // lanes per row
  constexpr int RPI = 64 / LPR;            // rows per instruction
  uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
  for (int r = 0; r < reps; ++r) {
    const int t = blockIdx.x * reps + r;
    const int tm = t / tiles_n, tn = t % tiles_n;
    char* base = out + (long)tm * 256 * ld + (long)tn * 512;
    // the tile is 256 rows x 512 B = 128 instructions of 1 KB; wave w issues 16 of them
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int idx = wave * 16 + i;                 // 0..127
      // instruction idx covers rows [ (idx / (512/SEG)) * RPI, +RPI ) and column segment (idx % (512/SEG)) * SEG
      const int cseg = idx % (512 / SEG), rblk = idx / (512 / SEG);
      const int row = rblk * RPI + lane / LPR;
      *reinterpret_cast<uint4*>(base + (long)row * ld + cseg * SEG + (lane % LPR) * 16) = v;
    }
  }
}

int main() {
  const long ld = 6144;                    // bytes per row (N = 3072 bf16)
  const int tiles_n = 12;
  const long bytes = 400L << 20;
  char* out;
  hipMalloc(&out, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {8, 64, 256}) {
    const int reps = 8;
    for (int seg : {128, 256, 512}) {
      float best = 1e9;
      for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0);
        if (seg == 128) hipLaunchKernelGGL(store_kernel<128>, dim3(grid), dim3(512), 0, 0, out, ld, reps, tiles_n);
        if (seg == 256) hipLaunchKernelGGL(store_kernel<256>, dim3(grid), dim3(512), 0, 0, out, ld, reps, tiles_n);
        if (seg == 512) hipLaunchKernelGGL(store_kernel<512>, dim3(grid), dim3(512), 0, 0, out, ld, reps, tiles_n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double per_tile_us = best * 1e3 / reps;
      printf("grid=%3d seg=%4d B: %.1f us per 128-KB tile per CU, %.1f GB/s per CU, %.2f TB/s total\n", grid, seg, per_tile_us,
             131072.0 / per_tile_us * 1e-3, 131072.0 * grid / per_tile_us * 1e-6);
    }
  }
  return 0;
}
