// micro-benchmark: issue rate of v_dot4c_i32_i8 / v_dot8_i32_i4 vs v_fma_f32 vs v_and on gfx950 (wave64), 1..4 waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int OP>
__global__ void k(int* out, int iters, int a0, int b0) {
  int a = a0 + threadIdx.x, b = b0;
  int acc[8] = {0, 1, 2, 3, 4, 5, 6, 7};
  float f[8] = {0, 1, 2, 3, 4, 5, 6, 7};
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (OP == 0) acc[j] = __builtin_amdgcn_sdot4(a, b + j, acc[j], false);
      else if (OP == 1) f[j] = __builtin_fmaf(f[j], 1.0001f, 0.5f);
      else if (OP == 2) acc[j] = (acc[j] & a) + b;
      else if (OP == 3) acc[j] = __builtin_amdgcn_sdot8(a, b + j, acc[j], false);
      else if (OP == 4) { typedef _Float16 h2 __attribute__((ext_vector_type(2))); h2 x = __builtin_bit_cast(h2, a), y = __builtin_bit_cast(h2, b + j); f[j] = __builtin_amdgcn_fdot2(x, y, f[j], false); }
      else if (OP == 5) acc[j] = __builtin_amdgcn_perm(acc[j], a, b + j);
      else if (OP == 6) acc[j] = __builtin_amdgcn_udot8(a, b + j, acc[j], false);
      else { typedef _Float16 h2 __attribute__((ext_vector_type(2))); h2 x = __builtin_bit_cast(h2, acc[j]), y = __builtin_bit_cast(h2, a); x = x * y + y; acc[j] = __builtin_bit_cast(int, x); }
    }
  }
  int s = 0;
  for (int j = 0; j < 8; j++) s += acc[j] + (int)f[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  int* d; hipMalloc(&d, 1 << 24);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  const char* names[8] = {"dot4_i8", "fma_f32", "and+add", "dot8_i4", "dot2_f16", "perm", "udot8_u4", "pk_fma_f16"};
  for (int wps = 1; wps <= 4; wps *= 2) {
    for (int op = 0; op < 8; op++) {
      dim3 grid(256), block(256 * wps);   // 256 CUs, wps waves per SIMD
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (op == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, d, iters, 3, 5);
        else if (op == 1) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, d, iters, 3, 5);
        else if (op == 2) hipLaunchKernelGGL(k<2>, grid, block, 0, 0, d, iters, 3, 5);
        else if (op == 3) hipLaunchKernelGGL(k<3>, grid, block, 0, 0, d, iters, 3, 5);
        else if (op == 4) hipLaunchKernelGGL(k<4>, grid, block, 0, 0, d, iters, 3, 5);
        else if (op == 5) hipLaunchKernelGGL(k<5>, grid, block, 0, 0, d, iters, 3, 5);
        else if (op == 6) hipLaunchKernelGGL(k<6>, grid, block, 0, 0, d, iters, 3, 5);
        else hipLaunchKernelGGL(k<7>, grid, block, 0, 0, d, iters, 3, 5);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double instr_per_wave = (double)iters * 8 * (op == 2 ? 2 : 1);
      double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * wps);
      printf("waves/SIMD=%d op=%s: %.3f ms -> %.2f ns per wave-instr per SIMD (%.1f cycles @2.4GHz)\n", wps, names[op], ms,
             ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4);
    }
  }
  return 0;
}
