// probe: operand lane map of v_mfma_i32_32x32x32_i8 on gfx950 (exact integer data, asymmetric operands), as the guide asks before relying on it.
// hypothesis: lane l (r = l & 31, h = l >> 5) supplies A[row r][k = 16 h + j] and B[k = 16 h + j][col r] in byte j = 0..15 of its 4-VGPR fragment;
// C/D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 h.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void k(const signed char* A, const signed char* B, int* C) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  v4i a, b;
  signed char ab[16], bb[16];
  for (int j = 0; j < 16; j++) { ab[j] = A[r * 32 + 16 * h + j]; bb[j] = B[(16 * h + j) * 32 + r]; }
  __builtin_memcpy(&a, ab, 16); __builtin_memcpy(&b, bb, 16);
  v16i c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
  for (int i = 0; i < 16; i++) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];
}
int main() {
  signed char hA[1024], hB[1024]; int hC[1024], ref[1024];
  srand(7);
  for (int i = 0; i < 1024; i++) { hA[i] = (signed char)(rand() % 255 - 127); hB[i] = (signed char)(rand() % 255 - 127); }
  for (int m = 0; m < 32; m++) for (int n = 0; n < 32; n++) { int s = 0; for (int kk = 0; kk < 32; kk++) s += (int)hA[m * 32 + kk] * (int)hB[kk * 32 + n]; ref[m * 32 + n] = s; }
  signed char *dA, *dB; int* dC;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096);
  hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 1024; i++) bad += hC[i] != ref[i];
  printf("v_mfma_i32_32x32x32_i8 operand-map hypothesis: %d of 1024 outputs differ from the integer reference\n", bad);
  return bad != 0;
}
