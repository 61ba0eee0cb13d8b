// Standalone micro-benchmark of the d_h = 64 attention kernels on random bf16 qkv (HIP events, median of rounds):
//   attn_bench.bin [n_images] [heads] [tokens]       (tokens 197 -> 7 query tiles)
// Runs attn_fwd_kernel<64,7,false> (one item per workgroup) and attn64_persist_kernel<7> (persistent producer /
// consumer), bit-compares their outputs, and — built with -DATTN_STAMPS — prints the s_memtime phase stamps of
// workgroup 0 of the persistent kernel (per item: barrier wait, QK^T + softmax, P V, store; producer: DMA issue / flight).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../attn.hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }

// clock calibration: a chain of dependent v_add_f32 (4 issue cycles each on a 64-wide wave), s_memtime around it
__global__ void calib_kernel(float* out, unsigned long long* ticks, int iters) {
  float v = threadIdx.x;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 100; ++j) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v));
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
  if (v == 12345.f) out[0] = v;
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "calib")) {
    float* o; unsigned long long* t; CK(hipMalloc(&o, 64)); CK(hipMalloc(&t, 64));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int grid : {1, 256, 2048}) {
      const int iters = 20000;
      hipLaunchKernelGGL(calib_kernel, dim3(grid), dim3(64), 0, 0, o, t, iters);
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(calib_kernel, dim3(grid), dim3(64), 0, 0, o, t, iters);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      unsigned long long h; CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost));
      printf("calib grid %d: %.3f ticks per dependent v_add_f32, %.3f ns per v_add (wall) -> s_memtime runs at %.3f GHz\n", grid, (double)h / (iters * 100.0),
             ms * 1e6 / (iters * 100.0), (double)h / (ms * 1e6));
    }
    return 0;
  }
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int nCU = pr.multiProcessorCount;
  const int n = argc > 1 ? atoi(argv[1]) : 512, heads = argc > 2 ? atoi(argv[2]) : 12, tokens = argc > 3 ? atoi(argv[3]) : 197;
  if (tokens <= 192 || tokens > 224) { printf("this tool instantiates the 7-tile kernels: 193..224 tokens\n"); return 1; }
  const int dim = heads * 64, ld = 3 * dim;
  const size_t rows = (size_t)n * tokens;
  std::vector<uint16_t> h(rows * ld);
  srand(3);
  for (auto& v : h) v = f2bf((rand() / (float)RAND_MAX) * 4.f - 2.f);
  bf16 *qkv, *o1, *o2;
  CK(hipMalloc(&qkv, h.size() * 2)); CK(hipMalloc(&o1, rows * dim * 2)); CK(hipMalloc(&o2, rows * dim * 2));
  CK(hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(o1, 0, rows * dim * 2)); CK(hipMemset(o2, 0, rows * dim * 2));
  const RowMap rm{tokens, 0, 0};
  const float scale = 0.125f;
  constexpr int NT = 7;
  constexpr int smem1 = NT * 32 * 256 + 4 * 4096, smem2 = 2 * 2 * NT * 32 * 128 + 7 * 4096;
  CK(hipFuncSetAttribute((const void*)attn_fwd_kernel<64, NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem1));
  CK(hipFuncSetAttribute((const void*)attn64_persist_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem2));
  const int items = heads * n;
#ifdef ATTN_STAMPS
  unsigned long long* stamps; CK(hipMalloc(&stamps, 8 * 256 * 8)); CK(hipMemset(stamps, 0, 8 * 256 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(attn_stamp_ptr), &stamps, sizeof(stamps)));
#endif
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto run = [&](int which) {
    std::vector<float> ms;
    for (int r = 0; r < 9; ++r) {
      CK(hipEventRecord(a));
      if (which == 0)
        hipLaunchKernelGGL((attn_fwd_kernel<64, NT, false>), dim3(heads, n), dim3(256), smem1, 0, qkv, ld, qkv, (size_t)tokens * ld, ld, o1,
                           (size_t)tokens * dim, dim, tokens, dim, scale, rm);
      else
        hipLaunchKernelGGL((attn64_persist_kernel<NT>), dim3(std::min(items, nCU)), dim3(512), smem2, 0, qkv, ld, o2, dim, tokens, dim, heads, items, scale, rm);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float m; CK(hipEventElapsedTime(&m, a, b)); ms.push_back(m);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    const double bytes = (double)rows * (ld + dim) * 2;
    printf("%s: median %.1f us, %.2f TB/s, %.2f us per item per CU\n", which ? "persistent  " : "one-item    ", ms[4] * 1e3, bytes / (ms[4] * 1e-3) / 1e12,
           ms[4] * 1e3 / ((double)items / nCU));
  };
  run(0); run(1);
  std::vector<uint16_t> r1(rows * dim), r2(rows * dim);
  CK(hipMemcpy(r1.data(), o1, r1.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), o2, r2.size() * 2, hipMemcpyDeviceToHost));
  size_t bad = 0; for (size_t i = 0; i < r1.size(); ++i) bad += r1[i] != r2[i];
  printf("bit-compare: %zu of %zu elements differ\n", bad, r1.size());
#ifdef ATTN_STAMPS
  std::vector<unsigned long long> st(8 * 256); CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
  const int nit = std::min(64, (items + nCU - 1) / nCU);
  const unsigned long long t0 = st[7 * 256 + 0];
  printf("s_memtime ticks (shader clock), workgroup 0, relative to the producer's first 'landed'\n");
  printf("item | producer: landed  B-passed  next-issued | wave0: arrive  B-passed  softmax  PV-done | wave4: arrive  B-passed  softmax  PV-done | wave3: arrive softmax PV-done\n");
  for (int i = 0; i < nit; ++i) {
    auto T = [&](int w, int s) { return (long long)(st[w * 256 + i * 4 + s] - t0); };
    printf("%4d | %7lld %7lld %7lld | %7lld %7lld %7lld %7lld | %7lld %7lld %7lld %7lld | %7lld %7lld %7lld\n", i, T(7, 0), T(7, 1), T(7, 2), T(0, 0), T(0, 1), T(0, 2), T(0, 3),
           T(4, 0), T(4, 1), T(4, 2), T(4, 3), T(3, 0), T(3, 2), T(3, 3));
  }
#endif
  return 0;
}
