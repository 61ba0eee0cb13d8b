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

// ------------------------------------------------------------------------------------------------------------------
// EXPERIMENT (not launched by the engine) - d_h = 64, full attention, TWO-PASS tile: one workgroup of NT waves per (image, head), one 32-query tile per wave.
// The one-pass kernels keep a tile's whole score row in registers (NT x 16), which caps the CU at two waves per SIMD
// and leaves each wave's QK^T -> softmax -> P V dependency chain exposed.  Here pass 1 computes the scores of a key
// tile only to fold them into the running row maximum and drops them; pass 2 recomputes the key tile (the same four
// MFMAs, so the same bits), exponentiates, adds to the row sum in the same (key tile, register) order, and feeds P V
// at once.  28 more MFMAs per tile buy a register budget of <= 128: two 7-wave workgroups = 14 waves per CU.
// LDS per workgroup: K rows 0..8*ceil(tokens/8)-1, V rows 0..16*ceil(tokens/16)-1 (pad keys are masked by assignment,
// a wholly padded half key tile is skipped: its P is exactly 0), NT x 4 KiB of output staging: 80,896 B for 197 tokens,
// two workgroups per CU.  Outputs are bit-identical to attn_fwd_kernel<64, NT, false>.
template <int NT>
__global__ __launch_bounds__(NT * 64, 4) void attn64_two_pass_kernel(const bf16* __restrict__ qkv, int ld, bf16* __restrict__ out, int ldo,
                                                                    int tokens, int dim, float scale, RowMap rm, int k_bytes, int v_bytes,
                                                                    int head_stride = 64, int out_img_rows = -1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = 4, DT = 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int head = blockIdx.x, img = blockIdx.y;
  const size_t img_row = (size_t)row_of(rm, img);
  // head_stride = 64: the engine's [row][q|k|v][head][64] layout; the layout experiment of tools/attn_bench passes a
  // head-major image ([image][q|k|v][head][token][64]: ld = 64, dim = heads * tokens * 64, head_stride = tokens * 64)
  const bf16* base = qkv + img_row * ld + (size_t)head * head_stride;
  char* Ks = smem;
  char* Vs = smem + k_bytes;
  char* ost = smem + k_bytes + v_bytes + wave * 4096;

  // Q fragments of this wave's tile, then the K / V pieces (1 KiB = 8 rows each), round-robin over the waves
  bf16x8 qf[KS];
  {
    const int q = wave * 32 + l31;
    const int qc = q < tokens ? q : tokens - 1;
    const bf16* qp = base + (size_t)qc * ld + 8 * lh;
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = *(const bf16x8*)(qp + 16 * s);
  }
  const int c = lane & 7;
  for (int piece = wave; piece * 1024 < k_bytes; piece += NT) {
    const int row = piece * 8 + (lane >> 3);
    const int rc = row < tokens ? row : tokens - 1;
    glds16(base + (size_t)rc * ld + dim + ((c ^ ((row >> 1) & 7)) << 3), Ks + piece * 1024);
  }
  for (int piece = wave; piece * 1024 < v_bytes; piece += NT) {
    const int row = piece * 8 + (lane >> 3);
    const int rc = row < tokens ? row : tokens - 1;
    glds16(base + (size_t)rc * ld + 2 * dim + ((c ^ (((row >> 1) & 1) << 2)) << 3), Vs + piece * 1024);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g = (lane >> 4) & 1;
  const int v_lane_off = (4 * lh + tr_q) * 128 + (16 * tr_g + 4 * tr_p) * 2;
  const int v_sw = (tr_q >> 1) & 1;
  const int k_sw = (l31 >> 1) & 7;

  auto scores = [&](int kt, f32x16& sa) {
#pragma unroll
    for (int i = 0; i < 16; ++i) sa[i] = 0.f;
    const char* kp = Ks + (kt * 32 + l31) * 128;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bf16x8 kf = *(const bf16x8*)(kp + (((2 * s + lh) ^ k_sw) << 4));
      sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sa, 0, 0, 0);
    }
    if (kt == NT - 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
        if (key >= tokens) sa[i] = -INFINITY;
      }
    }
  };

  // ---- pass 1: row maximum
  float mx = -INFINITY;
#pragma unroll 1
  for (int kt = 0; kt < NT; ++kt) {      // rolled: one key tile's registers live at a time (the budget is 128)
    f32x16 sa;
    scores(kt, sa);
#pragma unroll
    for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sa[i]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  const float c2 = scale * 1.44269504088896340736f;
  const float mc = -mx * c2;

  // ---- pass 2: P = exp2(s * c2 + mc), row sum, O^T += V^T P^T
  // (the Q fragments go through an empty asm: otherwise the compiler sees that pass 2 repeats pass 1's MFMAs, keeps all
  //  NT x 16 scores alive instead and spills them)
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    f32x4 t = __builtin_bit_cast(f32x4, qf[s]);
    asm volatile("" : "+v"(t));
    qf[s] = __builtin_bit_cast(bf16x8, t);
  }
  float sum = 0.f;
  f32x16 oacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[dt][i] = 0.f;
#pragma unroll 1
  for (int kt = 0; kt < NT; ++kt) {
    f32x16 sa;
    scores(kt, sa);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float e = __builtin_amdgcn_exp2f(fmaf(sa[i], c2, mc));
      sa[i] = e;
      sum += e;
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      if (kt == NT - 1 && kt * 32 + 16 * s2 >= tokens) continue;      // wholly padded half tile: P == 0 exactly
      bf16x8 pf;
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = (bf16)sa[8 * s2 + j];
      const char* vp = Vs + (kt * 32 + 16 * s2) * 128 + v_lane_off;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (__attribute__((address_space(3))) bf16x4*)(vp + (dt ^ v_sw) * 64));
        const bf16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (__attribute__((address_space(3))) bf16x4*)(vp + (dt ^ v_sw) * 64 + 8 * 128));
        bf16x8 vf;
#pragma unroll
        for (int j = 0; j < 4; ++j) { vf[j] = v1[j]; vf[4 + j] = v2[j]; }
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[dt], 0, 0, 0);
      }
    }
  }
  sum += __shfl_xor(sum, 32);
  const float inv = 1.0f / sum;

  // ---- store through the wave-private LDS tile: whole 128-byte row segments
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      bf16x4 o4;
#pragma unroll
      for (int j = 0; j < 4; ++j) o4[j] = (bf16)(oacc[dt][4 * g4 + j] * inv);
      *(bf16x4*)(ost + l31 * 128 + (((dt * 4 + g4) ^ (l31 & 7)) << 4) + lh * 8) = o4;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  bf16x8 ov[4];
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const int r = r4 * 8 + (lane >> 3);
    ov[r4] = *(const bf16x8*)(ost + r * 128 + (((lane & 7) ^ (r & 7)) << 4));
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const int qq = wave * 32 + r4 * 8 + (lane >> 3);
    // (out_img_rows >= 0: the layout experiment reads a differently laid out q/k/v image; the output keeps its rows)
    const size_t orow = out_img_rows >= 0 ? (size_t)img * out_img_rows : img_row;
    if (qq < tokens) *(bf16x8*)(out + (orow + qq) * ldo + head * 64 + (lane & 7) * 8) = ov[r4];
  }
}


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
  {   // two-pass kernel: NT waves per item, K rows to a multiple of 8, V rows to a multiple of 16
    const int k_bytes = (tokens + 7) / 8 * 1024, v_bytes = (tokens + 15) / 16 * 2048;
    const int smem3 = k_bytes + v_bytes + NT * 4096;
    CK(hipFuncSetAttribute((const void*)attn64_two_pass_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem3));
    bf16* o3; CK(hipMalloc(&o3, rows * dim * 2)); CK(hipMemset(o3, 0, rows * dim * 2));
    std::vector<float> ms;
    for (int r = 0; r < 9; ++r) {
      CK(hipEventRecord(a));
      hipLaunchKernelGGL((attn64_two_pass_kernel<NT>), dim3(heads, n), dim3(NT * 64), smem3, 0, qkv, ld, o3, dim, tokens, dim, scale, rm, k_bytes, v_bytes);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float m; CK(hipEventElapsedTime(&m, a, b)); ms.push_back(m);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    const double bytes = (double)rows * (ld + dim) * 2;
    printf("two-pass    : median %.1f us, %.2f TB/s, %.2f us per item per CU (LDS %d B per workgroup)\n", ms[4] * 1e3, bytes / (ms[4] * 1e-3) / 1e12,
           ms[4] * 1e3 / ((double)items / nCU), smem3);
    {   // layout experiment: the same kernel reading a head-major q/k/v image (contiguous 25-KiB blocks per item)
      std::vector<float> ms2;
      const RowMap rmh{3 * heads * tokens, 0, 0};          // image stride = 3 * heads * tokens rows of 64
      for (int r = 0; r < 9; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((attn64_two_pass_kernel<NT>), dim3(heads, n), dim3(NT * 64), smem3, 0, qkv, 64, o3, dim, tokens, heads * tokens * 64, scale, rmh, k_bytes, v_bytes,
                           tokens * 64, tokens);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float m; CK(hipEventElapsedTime(&m, a, b)); ms2.push_back(m);
      }
      CK(hipGetLastError());
      std::sort(ms2.begin(), ms2.end());
      printf("two-pass, head-major q/k/v (timing only): median %.1f us, %.2f TB/s\n", ms2[4] * 1e3, bytes / (ms2[4] * 1e-3) / 1e12);
      hipLaunchKernelGGL((attn64_two_pass_kernel<NT>), dim3(heads, n), dim3(NT * 64), smem3, 0, qkv, ld, o3, dim, tokens, dim, scale, rm, k_bytes, v_bytes, 64);
      CK(hipDeviceSynchronize());
    }
    std::vector<uint16_t> r1(rows * dim), r3(rows * dim);
    CK(hipMemcpy(r1.data(), o1, r1.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(r3.data(), o3, r3.size() * 2, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < r1.size(); ++i) bad += r1[i] != r3[i];
    printf("two-pass bit-compare vs one-item: %zu of %zu elements differ\n", bad, r1.size());
  }
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
