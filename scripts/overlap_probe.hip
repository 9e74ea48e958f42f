// overlap_probe.hip -- can two consecutive decode kernels overlap on MI355X without a kernel boundary between them?
//   Stand-alone experiment, not part of the library.
//
// Two kernels ping-pong like qkv -> attention+o_proj -> qkv ...: kernel k streams its own weights (independent of k-1), then needs the
// 64-bit accumulators kernel k-1 produced with device-scope atomics, then adds into its own accumulators.
//   serial   : one stream, plain launches (what the library does today)
//   overlap  : kernels alternate between two streams (a kernel is ordered behind k-2 by its stream, so at most two are resident); kernel k
//              issues its weight loads, then spins on kernel k-1's completion counter, then reads the accumulators with agent-scope loads.
// Variants of the completion signal: FENCE = __threadfence() before the counter increment, or only s_waitcnt vmcnt(0).
// Every spin is bounded; a timeout sets an error flag instead of hanging.
//
// build: hipcc --offload-arch=gfx950 -O3 -o overlap_probe overlap_probe.hip ; run: ./overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned u4_t __attribute__((ext_vector_type(4)));

struct Args {
  const uint4* w; int w_per_wave;         // this kernel's weights: w_per_wave uint4 per lane per wave
  unsigned long long* acc_out; int n_out; // accumulators this kernel adds into
  const unsigned long long* acc_in; int n_in;  // accumulators of the previous kernel
  unsigned* done_prev; unsigned target_prev;   // completion counter of the previous kernel and the value that means "complete"
  unsigned* done_self;                         // this kernel's counter (one increment per block)
  unsigned* err; unsigned* gerr; unsigned long long* sink;
  int wait_mode;                               // 0: no wait (stream order guarantees it), 1: spin on done_prev
  int fence;                                   // 1: __threadfence() before signalling, 0: vmcnt(0) only
  unsigned long long expect_in;                // every acc_in entry must equal this
};

template <int NW>   // uint4 per lane held in flight
__global__ __launch_bounds__(512) void k_stage(Args a) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // (1) weights: independent of the previous kernel, all in flight now
  u4_t q[NW];
  const uint4* wp = a.w + ((size_t)(blockIdx.x * 8 + wave) * NW) * 64 + lane;
#pragma unroll
  for (int i = 0; i < NW; i++) q[i] = __builtin_nontemporal_load((const u4_t*)(wp + i * 64));
  // (2) wait for the previous kernel
  __shared__ unsigned ok;
  if (a.wait_mode) {
    if (tid == 0) {
      unsigned spins = 0, v = 0;
      const bool dead = __hip_atomic_load(a.gerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;   // an earlier timeout: stop waiting everywhere
      if (!dead) do {
        v = __hip_atomic_load(a.done_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= a.target_prev) break;
        __builtin_amdgcn_s_sleep(8);
      } while (++spins < 200000u);
      ok = v >= a.target_prev;
      if (!ok) { atomicAdd(a.err, 1u); atomicAdd(a.gerr, 1u); }
    }
    __syncthreads();
  }
  // (3) the previous kernel's accumulators (agent-scope loads: they were produced by device-scope atomics of another kernel that may
  //     still have been running when this one started)
  unsigned long long bad = 0, s = 0;
  for (int i = tid; i < a.n_in; i += 512) {
    const unsigned long long v = __hip_atomic_load(a.acc_in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bad += v != a.expect_in;
    s += v;
  }
  if (bad) atomicAdd(a.err + 1, 1u);
  // (4) "compute" and accumulate: one atomic per (wave, 64 columns)
  unsigned x = (unsigned)s;
#pragma unroll
  for (int i = 0; i < NW; i++) x += q[i].x ^ q[i].y ^ q[i].z ^ q[i].w;
  if (x == 0x12345678u) a.sink[0] = x;   // keeps the loads alive
  const int n = ((blockIdx.x * 8 + wave) * 64 + lane) % a.n_out;
  atomicAdd(a.acc_out + n, 1ull);
  // (5) completion signal
  if (a.fence) __threadfence(); else __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) expcnt(0) lgkmcnt(0)
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(a.done_self, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main() {
  CK(hipSetDevice(0));
  const int GA = 192, GB = 256, NWV = 8;                      // blocks; uint4 per lane per wave (8 KiB per wave)
  const size_t wa = (size_t)GA * 8 * NWV * 64, wb = (size_t)GB * 8 * NWV * 64;   // uint4 counts: 12.6 MB and 16.8 MB
  const int NA = 6144, NB = 4096;
  uint4 *WA, *WB; unsigned long long *accA, *accB, *sink; unsigned *cnt, *err;
  CK(hipMalloc(&WA, wa * 16)); CK(hipMalloc(&WB, wb * 16));
  CK(hipMemset(WA, 1, wa * 16)); CK(hipMemset(WB, 2, wb * 16));
  CK(hipMalloc(&accA, NA * 8)); CK(hipMalloc(&accB, NB * 8)); CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&cnt, 64)); CK(hipMalloc(&err, 64));
  hipStream_t sx, sy; CK(hipStreamCreateWithFlags(&sx, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sy, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int ITERS = 200;
  // per (wave) one atomic on 64 columns: A touches GA*8*64 = 98304 slots over NA=6144 -> each column gets 16 per launch; B: 131072 / 4096 = 32
  const unsigned long long perA = (unsigned long long)GA * 8 * 64 / NA, perB = (unsigned long long)GB * 8 * 64 / NB;
  for (int mode = 0; mode < 4; mode++) {       // 0 serial, 1 overlap + fence, 2 overlap, vmcnt only, 3 overlap in a graph (vmcnt only)
    CK(hipMemset(accA, 0, NA * 8)); CK(hipMemset(accB, 0, NB * 8)); CK(hipMemset(cnt, 0, 64)); CK(hipMemset(err, 0, 64));
    CK(hipDeviceSynchronize());
    const bool overlap = mode != 0, graph = mode == 3;
    const int fence = mode == 1;
    auto launch = [&](int it, hipStream_t s_a, hipStream_t s_b) {
      Args a{}; a.w = WA; a.w_per_wave = NWV; a.acc_out = accA; a.n_out = NA; a.acc_in = accB; a.n_in = NB; a.done_prev = cnt + 1; a.done_self = cnt;
      a.target_prev = (unsigned)GB * it; a.err = err; a.gerr = err + 8; a.sink = sink; a.wait_mode = overlap; a.fence = fence; a.expect_in = perB * it;
      hipLaunchKernelGGL(k_stage<NWV>, dim3(GA), dim3(512), 0, s_a, a);
      Args b{}; b.w = WB; b.w_per_wave = NWV; b.acc_out = accB; b.n_out = NB; b.acc_in = accA; b.n_in = NA; b.done_prev = cnt; b.done_self = cnt + 1;
      b.target_prev = (unsigned)GA * (it + 1); b.err = err + 2; b.gerr = err + 8; b.sink = sink; b.wait_mode = overlap; b.fence = fence; b.expect_in = perA * (it + 1);
      hipLaunchKernelGGL(k_stage<NWV>, dim3(GB), dim3(512), 0, s_b, b);
    };
    float ms = 0.f;
    if (!graph) {
      hipStream_t s_a = sx, s_b = overlap ? sy : sx;
      CK(hipEventRecord(e0, sx));
      if (overlap) { CK(hipStreamWaitEvent(sy, e0, 0)); }
      for (int it = 0; it < ITERS; it++) launch(it, s_a, s_b);
      if (overlap) { hipEvent_t ej; CK(hipEventCreate(&ej)); CK(hipEventRecord(ej, sy)); CK(hipStreamWaitEvent(sx, ej, 0)); }
      CK(hipEventRecord(e1, sx));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    } else {
      // capture ITERS pairs on two forked streams into one graph, replay it once (counters / expectations are baked in as absolute values)
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(sx, hipStreamCaptureModeThreadLocal));
      hipEvent_t ef, ej; CK(hipEventCreate(&ef)); CK(hipEventCreate(&ej));
      CK(hipEventRecord(ef, sx)); CK(hipStreamWaitEvent(sy, ef, 0));
      for (int it = 0; it < ITERS; it++) launch(it, sx, sy);
      CK(hipEventRecord(ej, sy)); CK(hipStreamWaitEvent(sx, ej, 0));
      CK(hipStreamEndCapture(sx, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipEventRecord(e0, sx));
      CK(hipGraphLaunch(ge, sx));
      CK(hipEventRecord(e1, sx));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipDeviceSynchronize());
    unsigned herr[4]; CK(hipMemcpy(herr, err, 16, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> ha(NA), hb(NB);
    CK(hipMemcpy(ha.data(), accA, NA * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), accB, NB * 8, hipMemcpyDeviceToHost));
    int wrong = 0;
    for (auto v : ha) wrong += v != perA * ITERS;
    for (auto v : hb) wrong += v != perB * ITERS;
    const char* names[4] = {"serial (one stream)", "overlap, __threadfence signal", "overlap, vmcnt(0) signal", "overlap in one hipGraph, vmcnt(0) signal"};
    printf("%-42s %7.2f us per pair   timeouts A/B %u/%u  stale reads A/B %u/%u  final sums wrong %d\n", names[mode], ms * 1000.f / ITERS, herr[0], herr[2],
           herr[1], herr[3], wrong);
    fflush(stdout);
  }
  return 0;
}
