// Prefill GEMM probe (tuning aid): Y[S][N] = X[S][K] . W[N][K]^T, bf16 operands, f32 accumulate, at the shapes the prefill paths launch.
// Holds the LDS-DMA (global_load_lds) form of the 128 x 128 x 64 tile kernel so that pipeline depth, K split and tile order can be timed and checked
// against a plain reference kernel outside the library.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/gemm_probe.bin scripts/gemm_probe.hip && scripts/gemm_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define GLDS16(gsrc, ldst) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc), (__attribute__((address_space(3))) void*)(ldst), 16, 0, 0)

// workgroup tile 128 x 128 x 64, 2 x 2 waves of 64 x 64; NB LDS buffers of 32 KB (A 16 KB | B 16 KB), rows of 128 B un-padded with the 16-byte pieces
// XOR-swizzled by (row & 7) -- on the SOURCE address (LDS-DMA writes lane-linear) and on the fragment read.  Prefetch distance NB - 1 tiles.
template <int NB>
__global__ __launch_bounds__(256) void k_g3(const unsigned short* __restrict__ X, const unsigned short* __restrict__ W, int S, int N, int K, float* __restrict__ Y,
                                            float* __restrict__ part, int KS, int mtiles, int ntiles) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int mt = jj % mtiles, rest = jj / mtiles, ks = rest % KS, nt = (rest / KS) * 8 + xcd;
  if (nt >= ntiles) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int m0 = mt * 128, n0 = nt * 128;
  const int nk = K >> 6, k_beg = (int)((long long)ks * nk / KS), k_end = (int)((long long)(ks + 1) * nk / KS), nsteps = k_end - k_beg;
  // staging: wave w fills row groups 4 w .. 4 w + 3 (8 rows x 128 B = one 1 KB LDS-DMA each) of A and of B
  const int lrow = lane >> 3, piece = (lane & 7) ^ lrow;
  unsigned xo[4], wo[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = (wave * 4 + i) * 8 + lrow;
    xo[i] = (unsigned)min(m0 + row, S - 1) * (unsigned)K + 8u * piece;
    wo[i] = (unsigned)min(n0 + row, N - 1) * (unsigned)K + 8u * piece;
  }
  auto issue = [&](int kt, int buf) {
    unsigned char* base = lds + buf * 32768 + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; i++) GLDS16(X + xo[i] + (size_t)kt * 64, base + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; i++) GLDS16(W + wo[i] + (size_t)kt * 64, base + 16384 + i * 1024);
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[t][j][i] = 0.f;
  if (nsteps > 0) {
#pragma unroll
    for (int p = 0; p < NB - 1; p++) issue(min(k_beg + p, k_end - 1), p);
    const int aoff = (wm * 64 + r) * 128, boff = 16384 + (wn * 64 + r) * 128, sw = r & 7;
    for (int it = 0; it < nsteps; it++) {
      // this thread's pieces of tile `it` have landed (NB - 2 later tiles may still be in flight); after the barrier everyone's have, and everyone
      // is done reading the buffer the next issue overwrites
      if constexpr (NB == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if constexpr (NB == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      issue(min(k_beg + it + NB - 1, k_end - 1), (it + NB - 1) % NB);          // clamped: redundant reloads at the tail, never a branch around a load
      const unsigned char* tb = lds + (it % NB) * 32768;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const int po = ((4 * h + s) ^ sw) * 16;
        bf16x8 af[2], bf[2];
#pragma unroll
        for (int t = 0; t < 2; t++) af[t] = *(const bf16x8*)(tb + aoff + t * 32 * 128 + po);
#pragma unroll
        for (int j = 0; j < 2; j++) bf[j] = *(const bf16x8*)(tb + boff + j * 32 * 128 + po);
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
          for (int j = 0; j < 2; j++) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t], bf[j], acc[t][j], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the tail's redundant LDS-DMAs land before the LDS allocation is released
  }
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int n = n0 + wn * 64 + 32 * j + r;
    if (n < N) {
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const int m = m0 + wm * 64 + 32 * t + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (m < S) {
            if (part) part[((size_t)ks * S + m) * N + n] = acc[t][j][i];
            else Y[(size_t)m * N + n] = acc[t][j][i];
          }
        }
    }
  }
}

__global__ void k_reduce(const float* part, int KS, size_t SN, float* Y) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < SN; i += (size_t)gridDim.x * 256) {
    float v = 0.f;
    for (int z = 0; z < KS; z++) v += part[(size_t)z * SN + i];
    Y[i] = v;
  }
}
__global__ void k_ref(const unsigned short* X, const unsigned short* W, int S, int N, int K, float* Y) {
  const int n = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
  if (n >= N) return;
  float a = 0.f;
  for (int k = 0; k < K; k++) a += __uint_as_float((unsigned)X[(size_t)m * K + k] << 16) * __uint_as_float((unsigned)W[(size_t)n * K + k] << 16);
  Y[(size_t)m * N + n] = a;
}

static unsigned short bf(float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }

template <int NB>
static void launch(const unsigned short* X, const unsigned short* W, int S, int N, int K, float* Y, float* part, int KS) {
  const int mtiles = (S + 127) / 128, ntiles = (N + 127) / 128;
  const unsigned grid = 8u * ((ntiles + 7) / 8) * mtiles * KS;
  static bool done = false;
  if (!done) { hipFuncSetAttribute((const void*)k_g3<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, NB * 32768); done = true; }
  hipLaunchKernelGGL(k_g3<NB>, dim3(grid), dim3(256), NB * 32768, 0, X, W, S, N, K, Y, KS > 1 ? part : nullptr, KS, mtiles, ntiles);
  if (KS > 1) hipLaunchKernelGGL(k_reduce, dim3(2048), dim3(256), 0, 0, part, KS, (size_t)S * N, Y);
}

int main() {
  struct Shape { int S, N, K; } shapes[] = {{512, 10576, 2560}, {512, 2560, 5120}, {512, 16384, 2048}, {512, 2048, 8192}, {2048, 6144, 4096}, {2048, 4096, 4096}, {64, 10576, 2560}, {200, 3072, 2048}};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto sh : shapes) {
    const int S = sh.S, N = sh.N, K = sh.K;
    std::vector<unsigned short> hx((size_t)S * K), hw((size_t)N * K);
    srand(1);
    for (auto& v : hx) v = bf((rand() % 2001 - 1000) / 1000.0f);
    for (auto& v : hw) v = bf((rand() % 2001 - 1000) / 1000.0f);
    unsigned short *X, *W; float *Y, *R, *P;
    hipMalloc(&X, hx.size() * 2); hipMalloc(&W, hw.size() * 2); hipMalloc(&Y, (size_t)S * N * 4); hipMalloc(&R, (size_t)S * N * 4); hipMalloc(&P, (size_t)16 * S * N * 4);
    hipMemcpy(X, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_ref, dim3((N + 255) / 256, S), dim3(256), 0, 0, X, W, S, N, K, R);
    std::vector<float> hr((size_t)S * N), hy((size_t)S * N);
    hipMemcpy(hr.data(), R, hr.size() * 4, hipMemcpyDeviceToHost);
    for (int nb : {2, 3, 4})
      for (int KS : {1, 2, 3, 4, 8}) {
        if (KS > K / 64 / 2) continue;
        hipMemset(Y, 0xff, (size_t)S * N * 4);
        auto go = [&]() { if (nb == 2) launch<2>(X, W, S, N, K, Y, P, KS); else if (nb == 3) launch<3>(X, W, S, N, K, Y, P, KS); else launch<4>(X, W, S, N, K, Y, P, KS); };
        go();
        hipMemcpy(hy.data(), Y, hy.size() * 4, hipMemcpyDeviceToHost);
        double maxd = 0, maxr = 0;
        for (size_t i = 0; i < hy.size(); i++) { maxd = fmax(maxd, fabs((double)hy[i] - hr[i])); maxr = fmax(maxr, fabs((double)hr[i])); }
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
          hipEventRecord(e0);
          for (int i = 0; i < 10; i++) go();
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1); best = fminf(best, ms);
        }
        const double us = best * 1e3 / 10;
        printf("S %4d N %5d K %4d  NB %d KS %d: %7.1f us  %6.1f TFLOP/s   max|d| %.3g (max|ref| %.3g)%s\n", S, N, K, nb, KS, us, 2.0 * S * N * K / us * 1e-6, maxd, maxr,
               maxd > 1e-3 * maxr ? "  MISMATCH" : "");
        fflush(stdout);
      }
    hipFree(X); hipFree(W); hipFree(Y); hipFree(R); hipFree(P);
  }
  return 0;
}
