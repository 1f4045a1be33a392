// stream_bench.hip -- what HBM read bandwidth do the scan kernel's access patterns allow?
//  contig : every workgroup streams one contiguous slice (copy-style ceiling)
//  pattern: the scan kernel's pattern (4 rows x 1 KiB per wave per column, 64 KiB column stride), loads only
//  pattern+sv: same plus the selector re-reads from L2
// hipcc --offload-arch=gfx950 -O3 stream_bench.hip -o stream_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) read_contig(const u64x2* __restrict__ p, size_t n_vec, uint64_t* out, int unroll) {
  size_t per_block = (n_vec + gridDim.x - 1) / gridDim.x;
  size_t b0 = (size_t)blockIdx.x * per_block, b1 = b0 + per_block < n_vec ? b0 + per_block : n_vec;
  uint64_t acc = 0;
  for (size_t i = b0 + threadIdx.x; i < b1; i += 256 * 4) {
    u64x2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { size_t j = i + (size_t)u * 256; v[u] = j < b1 ? __builtin_nontemporal_load(p + j) : u64x2{0, 0}; }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc ^= v[u].x + v[u].y;
  }
  if (acc == 0x1234567) out[0] = acc;
}

template <int ROWS, bool SV, int DEPTH>
__global__ void __launch_bounds__(256) read_pattern(const uint64_t* __restrict__ db, const uint64_t* __restrict__ sv,
                                                    uint32_t rows, uint32_t cols, uint32_t kN, uint64_t* out) {
  const uint32_t c0 = (blockIdx.x * 256 + threadIdx.x) * 2;
  const uint32_t row0 = blockIdx.y * ROWS;
  const uint64_t* rp[ROWS];
  for (int r = 0; r < ROWS; ++r) { uint32_t row = row0 + r < rows ? row0 + r : rows - 1; rp[r] = db + (size_t)row * cols * kN + c0; }
  uint64_t acc = 0;
  for (uint32_t col = 0; col < cols; col += DEPTH) {
    u64x2 v[DEPTH][ROWS + 2];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      uint32_t cc = col + d < cols ? col + d : cols - 1;
#pragma unroll
      for (int r = 0; r < ROWS; ++r) v[d][r] = __builtin_nontemporal_load((const u64x2*)(rp[r] + (size_t)cc * kN));
      if (SV) { v[d][ROWS] = *(const u64x2*)(sv + (size_t)cc * 2 * kN + c0); v[d][ROWS + 1] = *(const u64x2*)(sv + (size_t)cc * 2 * kN + kN + c0); }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int r = 0; r < ROWS + (SV ? 2 : 0); ++r) acc ^= v[d][r].x + v[d][r].y;
  }
  if (acc == 0x1234567) out[0] = acc;
}

template <typename F> float time_it(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  float best = 1e30f;
  for (int i = 0; i < 5; ++i) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  return best;
}

int main() {
  const uint32_t rows = 162, cols = 162, kN = 8192;
  const size_t words = (size_t)rows * cols * kN;  // 1.72 GB
  uint64_t *db, *sv, *out;
  CHECK(hipMalloc(&db, words * 8)); CHECK(hipMalloc(&sv, (size_t)cols * 2 * kN * 8)); CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(db, 1, words * 8)); CHECK(hipMemset(sv, 2, (size_t)cols * 2 * kN * 8));
  const double gb = words * 8 / 1e9;
  for (int blocks : {256, 512, 768, 1024, 2048, 4096}) {
    float ms = time_it([&] { hipLaunchKernelGGL(read_contig, dim3(blocks), dim3(256), 0, 0, (const u64x2*)db, words / 2, out, 4); });
    printf("contig   blocks=%5d            %.3f ms  %.0f GB/s\n", blocks, ms, gb / ms * 1e3);
  }
  dim3 g4(kN / 2 / 256, (rows + 3) / 4), g2(kN / 2 / 256, (rows + 1) / 2), g8(kN / 2 / 256, (rows + 7) / 8);
#define RUN(R, SVF, D, G) { float ms = time_it([&] { hipLaunchKernelGGL((read_pattern<R, SVF, D>), G, dim3(256), 0, 0, db, sv, rows, cols, kN, out); }); \
    printf("pattern  rows=%d sv=%d depth=%d     %.3f ms  %.0f GB/s\n", R, (int)SVF, D, ms, gb / ms * 1e3); }
  RUN(4, false, 1, g4) RUN(4, false, 2, g4) RUN(4, false, 4, g4) RUN(4, true, 1, g4) RUN(4, true, 2, g4) RUN(4, true, 4, g4)
  RUN(2, false, 4, g2) RUN(2, true, 4, g2) RUN(8, false, 2, g8) RUN(8, true, 2, g8)
  return 0;
}
