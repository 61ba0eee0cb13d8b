// How fast can ONE CU store, and how fast can all of them together?  Decides whether a GEMM epilogue that writes a
// 256 x 256 bf16 tile (128 KiB per workgroup, 16-byte stores in 128-byte row segments) in ~8-10 k cycles is bound by the
// CU's own store path or by the chip-wide HBM write rate.
//   store_bench.bin                (sweeps the number of active workgroups, one per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) int i32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// each workgroup writes `tiles` tiles of 256 rows x 512 bytes (= 128 KiB) with the epilogue's pattern: a wave stores
// 8 rows x 128 B per instruction, 16 instructions per wave per tile; row stride `ld` bytes (a [M, 2304] bf16 matrix: 4608)
__global__ __launch_bounds__(512) void store_kernel(char* out, int tiles, size_t ld, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  i32x4 v = {lane, wave, 3, 4};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < tiles; ++t) {
    char* tile = out + ((size_t)blockIdx.x * tiles + t) * 256 * ld;
    char* base = tile + (size_t)(wm * 128 + (lane >> 3)) * ld + wn * 128 + (lane & 7) * 16;
#pragma unroll
    for (int it = 0; it < 16; ++it) *(i32x4*)(base + (size_t)(it * 8) * ld) = v;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const int tiles = 32; const size_t ld = 4608;
  char* out; unsigned long long* cyc;
  const size_t bytes = (size_t)256 * tiles * 256 * ld;
  CK(hipMalloc(&out, bytes)); CK(hipMalloc(&cyc, 256 * 8));
  CK(hipMemset(out, 0, bytes));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int G : {1, 2, 8, 32, 64, 128, 256}) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(a)); hipLaunchKernelGGL(store_kernel, dim3(G), dim3(512), 0, 0, out, tiles, ld, cyc); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      std::vector<unsigned long long> h(G); CK(hipMemcpy(h.data(), cyc, G * 8, hipMemcpyDeviceToHost));
      std::sort(h.begin(), h.end());
      const double per_wg = (double)tiles * 256 * 512;
      if (rep == 2) printf("%3d workgroups: %.1f us, %.2f TB/s in all; per CU: median %.0f cycles per 128-KiB tile = %.1f B/clk (slowest %.1f B/clk)\n", G, ms * 1e3,
                           G * per_wg / (ms * 1e-3) / 1e12, (double)h[G / 2] / tiles, per_wg / (double)h[G / 2], per_wg / (double)h[G - 1]);
    }
  }
  return 0;
}
