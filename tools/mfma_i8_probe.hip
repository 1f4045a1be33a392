// Probe: operand / result layout of v_mfma_i32_16x16x64_i8 on gfx950 (used by the digit-sliced scan).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void probe(const int8_t* A, const int8_t* B, int* D) {
  const int l = threadIdx.x;
  v4i a, b, c = {0, 0, 0, 0};
  int8_t ta[16], tb[16];
  for (int t = 0; t < 16; ++t) {
    const int k = (l >> 4) * 16 + t;
    ta[t] = A[(l & 15) * 64 + k];   // A[m][k]
    tb[t] = B[k * 16 + (l & 15)];   // B[k][n]
  }
  __builtin_memcpy(&a, ta, 16);
  __builtin_memcpy(&b, tb, 16);
  c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[((l >> 4) * 4 + i) * 16 + (l & 15)] = c[i];   // D[m][n]
}

int main() {
  int8_t hA[16 * 64], hB[64 * 16];
  srand(7);
  for (auto& v : hA) v = (int8_t)(rand() % 256 - 128);
  for (auto& v : hB) v = (int8_t)(rand() % 256 - 128);
  int8_t *dA, *dB; int* dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dD);
  int hD[256];
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      int ref = 0;
      for (int k = 0; k < 64; ++k) ref += (int)hA[m * 64 + k] * (int)hB[k * 16 + n];
      if (ref != hD[m * 16 + n]) ++bad;
    }
  printf("mfma_i32_16x16x64_i8 layout probe: %d mismatches of 256\n", bad);
  return bad != 0;
}
