// How fast can ONE CU take operand tiles in?  The persistent 256x256 GEMM needs (256 + 256) rows x 128 B = 64 KiB per
// K-tile per CU against 2048 matrix-pipe cycles (32 B/clk at the MFMA peak).  This micro-benchmark issues exactly that
// access pattern (8 waves, one workgroup per CU, 1-KiB pieces of 8 rows x 128 B, source-swizzled) with NO compute:
//   mode 0  global_load_lds_dwordx4 (LDS-DMA), counted vmcnt, 3-deep
//   mode 1  global_load_dwordx4 -> VGPR, consumed by an empty asm (no LDS write)
//   mode 2  global_load_dwordx4 -> VGPR -> ds_write_b128
// src 0: every CU re-reads the same two 256-row panels (L2 hits); src 1: the A rows stream from a big matrix (HBM /
// Infinity Cache) while B stays a resident panel — the GEMM's real mix.
//   ingest_bench.bin [iters_kt]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../common.hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE, int NWV = 8>
__global__ __launch_bounds__(NWV * 64) void ingest_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W, int lda, int nkt, int kper,
                                                     long a_rows_per_wg, unsigned long long* cyc, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int P = 32 / NWV;                      // 1-KiB pieces per operand per wave per K-tile
  const bf16* a_src[P];
  const bf16* w_src[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int row = (wave + NWV * i) * 8 + (lane >> 3);
    const int c_src = (lane & 7) ^ ((row >> 1) & 7);
    a_src[i] = A + (size_t)row * lda + c_src * 8;
    w_src[i] = W + (size_t)row * lda + c_src * 8;
  }
  unsigned long long t0 = 0, t1 = 0;
  if (tid == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  f32x4 accv = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < nkt; ++it) {
    const int kt = it % kper;                       // walk K like the GEMM does, then start over (next tile)
    const int slot = it % 3;
    // src 1: like the GEMM's tile walk, 9 neighbouring workgroups of an XCD share an A panel, a new panel per tile
    const size_t a_tile = a_rows_per_wg ? (size_t)(((long)(xcd_remap(blockIdx.x, gridDim.x) / 9 + (long)(it / kper) * 29) * 256) % a_rows_per_wg) * lda : 0;
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < P; ++i) glds16(w_src[i] + kt * 64, smem + 98304 + (it & 1) * 32768 + (wave + NWV * i) * 1024);
#pragma unroll
      for (int i = 0; i < P; ++i) glds16(a_src[i] + a_tile + kt * 64, smem + slot * 32768 + (wave + NWV * i) * 1024);
      if (P == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (P == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else if (NWV == 8) {
      f32x4 v[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = *(const f32x4*)(w_src[i] + kt * 64);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[4 + i] = *(const f32x4*)(a_src[i] + a_tile + kt * 64);
      if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(v[i]));
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)(smem + 98304 + (it & 1) * 32768 + (wave + 8 * i) * 1024 + lane * 16) = v[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)(smem + slot * 32768 + (wave + 8 * i) * 1024 + lane * 16) = v[4 + i];
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    cyc[blockIdx.x] = t1 - t0;
  }
  if (MODE != 1 && sink && tid == 12345678) sink[0] = *(float*)(smem + lane * 4) + accv.x;
}

int main(int argc, char** argv) {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int nCU = pr.multiProcessorCount;
  const int nkt = argc > 1 ? atoi(argv[1]) : 1200;
  const int K = 768, kper = K / 64;
  const long big_rows = 63040;
  bf16 *A, *W; unsigned long long* cyc; float* sink;
  CK(hipMalloc(&A, (size_t)(big_rows + 512) * K * 2)); CK(hipMalloc(&W, (size_t)512 * K * 2));
  CK(hipMemset(A, 0x11, (size_t)(big_rows + 512) * K * 2)); CK(hipMemset(W, 0x22, (size_t)512 * K * 2));
  CK(hipMalloc(&cyc, nCU * 8)); CK(hipMalloc(&sink, 64));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto run = [&](int mode, int src) {
    const long rows_per_wg = src ? big_rows - 256 : 0;            // src 1: panels walk over the whole matrix
    const void* fn = mode == 0 ? (const void*)ingest_kernel<0> : mode == 1 ? (const void*)ingest_kernel<1> : (const void*)ingest_kernel<2>;
    CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    CK(hipFuncSetAttribute((const void*)ingest_kernel<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    CK(hipFuncSetAttribute((const void*)ingest_kernel<0, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    std::vector<float> ms;
    for (int r = 0; r < 7; ++r) {
      CK(hipEventRecord(a));
      if (mode == 0) hipLaunchKernelGGL(ingest_kernel<0>, dim3(nCU), dim3(512), 163840, 0, A, W, K, nkt, kper, rows_per_wg, cyc, sink);
      if (mode == 3) hipLaunchKernelGGL((ingest_kernel<0, 4>), dim3(nCU), dim3(256), 163840, 0, A, W, K, nkt, kper, rows_per_wg, cyc, sink);
      if (mode == 4) hipLaunchKernelGGL((ingest_kernel<0, 16>), dim3(nCU), dim3(1024), 163840, 0, A, W, K, nkt, kper, rows_per_wg, cyc, sink);
      if (mode == 1) hipLaunchKernelGGL(ingest_kernel<1>, dim3(nCU), dim3(512), 163840, 0, A, W, K, nkt, kper, rows_per_wg, cyc, sink);
      if (mode == 2) hipLaunchKernelGGL(ingest_kernel<2>, dim3(nCU), dim3(512), 163840, 0, A, W, K, nkt, kper, rows_per_wg, cyc, sink);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float m; CK(hipEventElapsedTime(&m, a, b)); ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    std::vector<unsigned long long> h(nCU); CK(hipMemcpy(h.data(), cyc, nCU * 8, hipMemcpyDeviceToHost));
    double c = 0; for (auto v : h) c += (double)v; c /= nCU;
    const double bytes = (double)nkt * 65536.0;
    printf("mode %d (%s) src %d (%s): median %.1f us, %.0f cycles per CU -> %.1f B/clk/CU, %.2f TB/s chip-wide, %.0f cycles per 64-KiB K-tile\n", mode,
           mode == 0 ? "LDS-DMA, 8 waves" : mode == 1 ? "VGPR load only" : mode == 2 ? "VGPR load + ds_write" : mode == 3 ? "LDS-DMA, 4 waves" : "LDS-DMA, 16 waves", src, src ? "A streams" : "L2-resident", ms[3] * 1e3, c,
           bytes / c, bytes * nCU / (ms[3] * 1e-3) / 1e12, c / nkt);
  };
  for (int src = 0; src < 2; ++src)
    for (int mode : {0, 3, 4, 1, 2}) run(mode, src);
  return 0;
}
