// Where do once-read weights come from fastest?  Streams one layer's MLP weights (91.5 MB) with 16-byte loads, plain or non-temporal,
// (a) cold: six buffers in rotation (549 MB, beyond the 256 MiB Infinity Cache), (b) hot: the same buffer again and again (resident in the
// Infinity Cache when the load policy allocates there).  Decides whether prefetching the next kernel's weights into the cache can pay.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mall_probe scripts/mall_probe.hip && /tmp/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int DEPTH, int NT>
__global__ void k_stream(const u32x4* __restrict__ src, size_t per_wave_vec, unsigned* sink) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const u32x4* p = src + (size_t)wave * per_wave_vec + lane;
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = 0; i < per_wave_vec; i += 64 * DEPTH) {
    u32x4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) v[d] = NT ? __builtin_nontemporal_load(p + i + 64 * d) : p[i + 64 * d];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) acc ^= v[d];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}
int main() {
  const size_t bytes = 91521024 / 4096 * 4096;
  const int NBUF = 6;
  std::vector<void*> bufs(NBUF);
  for (auto& b : bufs) { hipMalloc(&b, bytes + (1 << 20)); hipMemset(b, 1, bytes); }
  unsigned* sink; hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Cfg { int grid, threads; } cfgs[] = {{224, 1024}, {256, 1024}, {256, 512}, {512, 256}};
  for (auto c : cfgs) {
    const size_t waves = (size_t)c.grid * c.threads / 64;
    const size_t per_wave_vec = bytes / 16 / waves / (64 * 8) * (64 * 8);
    for (int nt = 0; nt < 2; nt++)
      for (int hot = 0; hot < 2; hot++) {
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
          hipEventRecord(e0);
          for (int i = 0; i < 12; i++) {
            const u32x4* s = (const u32x4*)bufs[hot ? 0 : i % NBUF];
            if (nt) hipLaunchKernelGGL((k_stream<8, 1>), dim3(c.grid), dim3(c.threads), 0, 0, s, per_wave_vec, sink);
            else hipLaunchKernelGGL((k_stream<8, 0>), dim3(c.grid), dim3(c.threads), 0, 0, s, per_wave_vec, sink);
          }
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          best = ms < best ? ms : best;
        }
        const double us = best * 1e3 / 12, gb = (double)per_wave_vec * 16 * waves / 1e9;
        printf("grid %4d x %4d thr, 8 KiB in flight per wave, %s loads, %s: %6.2f us per launch, %6.0f GB/s\n", c.grid, c.threads, nt ? "nt   " : "plain", hot ? "hot (same 91.5 MB)" : "cold (549 MB ring) ",
               us, gb / (us * 1e-6));
      }
  }
  return 0;
}
