// What HW_REG_XCC_ID reports per workgroup of a 1-D grid (the fused LayerNorm of gemm256.hip.h keys its work queues on it):
//   xcc_probe.bin [blocks, default 1024]   ->  histogram of ids, and how often id == blockIdx % 8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
__global__ void probe(unsigned int* ids) {
  unsigned int r;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(r));
  if (threadIdx.x == 0) ids[blockIdx.x] = r;
}
int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 1024;
  unsigned int* d; CK(hipMalloc(&d, nb * 4));
  for (int threads : {64, 512}) {
    hipLaunchKernelGGL(probe, dim3(nb), dim3(threads), 0, 0, d);
    CK(hipDeviceSynchronize());
    std::vector<unsigned int> h(nb);
    CK(hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost));
    int hist[16] = {}, rr = 0; unsigned hi = 0;
    for (int i = 0; i < nb; ++i) { hist[h[i] & 15]++; rr += (int)(h[i] & 15) == i % 8; hi |= h[i] >> 4; }
    printf("%d blocks x %d threads: ids", nb, threads); for (int i = 0; i < 16; ++i) if (hist[i]) printf(" %d:%d", i, hist[i]);
    printf("   id == blockIdx %% 8 for %d of %d   (bits above 3:0 seen: 0x%x; first 16:", rr, nb, hi);
    for (int i = 0; i < 16 && i < nb; ++i) printf(" %u", h[i] & 15); printf(")\n");
  }
  return 0;
}
