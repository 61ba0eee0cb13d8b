// Standalone micro-benchmark of gemm_bf16_kernel<EPI> on random bf16 data (HIP events, median of rounds).
//   gemm_bench.bin M N K epi [iters] [tokens]       epi: 0 bf16, 1 resid, 2 fc1(+score), 3 patch, 4 f32 (128x128 kernel);
//   10/11/12 bf16/resid/fc1 on the persistent 256x256 kernel (13/14: + pre- / post-GELU stage-1 score),
//   15 resid + fused LayerNorm of the finished row panels (N = 768 / 1024 / 1280; x vs the 128x128 kernel, h vs layernorm_bf16_kernel),
//   40/41/42 the same on the four-wave kernel (gemm256w4.hip.h), 30..33 e4m3 operands (gemm256 F8)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include "../gemm256.hip.h"
#include "../misc.hip.h"
#include "gemm256w4.hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
// OCP e4m3fn: decode exactly, encode by nearest (ties to even mantissa) over the 256 codes, saturating at +-448
static float e4m3_to_f(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -f : f;
}
static uint8_t f_to_e4m3(float f) {
  static float tab[127]; static bool init = false;
  if (!init) { for (int i = 0; i < 127; ++i) tab[i] = e4m3_to_f((uint8_t)i); init = true; }      // 0 .. 0x7e = 448 (0x7f = NaN)
  const float a = fabsf(f) > 448.f ? 448.f : fabsf(f);
  int lo = 0, hi = 126;
  while (hi - lo > 1) { const int mid = (lo + hi) / 2; if (tab[mid] <= a) lo = mid; else hi = mid; }
  int best = lo;
  if (hi != lo) { const float dl = a - tab[lo], dh = tab[hi] - a; best = dh < dl ? hi : (dl < dh ? lo : ((lo & 1) ? hi : lo)); }
  return (uint8_t)(best | (f < 0 ? 0x80 : 0));
}

template <int EPI, int SCORE = 0> static void launch(const GemmArgs& g, hipStream_t s) {
  static bool done = false;
  if (!done) { CK(hipFuncSetAttribute((const void*)gemm_bf16_kernel<EPI, SCORE>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES)); done = true; }
  hipLaunchKernelGGL((gemm_bf16_kernel<EPI, SCORE>), dim3(g.tiles_m * g.tiles_n), dim3(256), GEMM_LDS_BYTES, s, g);
}

static int nCU = 256;
template <int EPI, int SCORE = 0> static void launch256(GemmArgs g, hipStream_t s) {
  static bool done = false;
  if (!done) { CK(hipFuncSetAttribute((const void*)gemm256_bf16_kernel<EPI, SCORE>, hipFuncAttributeMaxDynamicSharedMemorySize, G256::LDS)); done = true; }
  g.tiles_m = (g.M + 255) / 256; g.tiles_n = (g.N + 255) / 256;
  hipLaunchKernelGGL((gemm256_bf16_kernel<EPI, SCORE>), dim3(std::min(g.tiles_m * g.tiles_n, nCU)), dim3(512), G256::LDS, s, g);
}
template <int LNV> static void launch256ln(GemmArgs g, hipStream_t s) {   // residual + LayerNorm of the finished panels by their last-arriving workgroup
  static bool done = false;
  static unsigned int* sync = nullptr; static int set = 0;
  g.tiles_m = (g.M + 255) / 256; g.tiles_n = (g.N + 255) / 256;
  if (!done) {
    CK(hipFuncSetAttribute((const void*)gemm256_bf16_kernel<EPI_RESID, LNV>, hipFuncAttributeMaxDynamicSharedMemorySize, G256::LDS));
    CK(hipMalloc(&sync, (16 + g.tiles_m + 1) * 4)); CK(hipMemset(sync, 0, (16 + g.tiles_m + 1) * 4));
    done = true;
  }
  g.ln_sync = sync; g.ln_set = set; set ^= 1;
  hipLaunchKernelGGL((gemm256_bf16_kernel<EPI_RESID, LNV>), dim3(std::min(g.tiles_m * g.tiles_n, nCU)), dim3(512), G256::LDS, s, g);
}
template <int EPI> static void launch256w4(GemmArgs g, hipStream_t s) {   // four-wave kernel (one wave per SIMD, 128 x 128 per wave)
  static bool done = false;
  if (!done) { CK(hipFuncSetAttribute((const void*)gemm256w4_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, G256W4::LDS)); done = true; }
  g.tiles_m = (g.M + 255) / 256; g.tiles_n = (g.N + 255) / 256;
  hipLaunchKernelGGL((gemm256w4_kernel<EPI>), dim3(std::min(g.tiles_m * g.tiles_n, nCU)), dim3(256), G256W4::LDS, s, g);
}
template <int EPI, int SCORE = 0> static void launch256f8(GemmArgs g, hipStream_t s) {   // e4m3 operands (gemm256 with F8 = true)
  static bool done = false;
  if (!done) { CK(hipFuncSetAttribute((const void*)gemm256_bf16_kernel<EPI, SCORE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G256::LDS)); done = true; }
  g.tiles_m = (g.M + 255) / 256; g.tiles_n = (g.N + 255) / 256;
  hipLaunchKernelGGL((gemm256_bf16_kernel<EPI, SCORE, true>), dim3(std::min(g.tiles_m * g.tiles_n, nCU)), dim3(512), G256::LDS, s, g);
}
// epi 30 bf16-out (QKV), 31 resid (fc2), 32 fc1 (+GELU, e4m3 out), 33 fc1 + pre-GELU score (e4m3 out + slab): e4m3 operands.
// Operands are quantised on the host (A: direct cast, W: per-row scale amax/448), the reference is the fp32 sum of the
// exact products of the DEQUANTISED values, so the check isolates the kernel (layout, scale, epilogue) from quantisation.
static int run_fp8(int M, int N, int K, int epi, int iters, int tokens) {
  const int Npad = (N + 255) / 256 * 256, K8 = (K + 127) / 128 * 128;
  std::vector<uint8_t> hA((size_t)M * K8, 0), hW((size_t)Npad * K8, 0);
  std::vector<float> fA((size_t)M * K8, 0.f), fW((size_t)Npad * K8, 0.f), hs(Npad, 1.f), hb(Npad, 0.f);
  srand(2);
  for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) { const uint8_t q = f_to_e4m3(((rand() / (float)RAND_MAX) * 2.f - 1.f) * 3.f); hA[(size_t)m * K8 + k] = q; fA[(size_t)m * K8 + k] = e4m3_to_f(q); }
  for (int n = 0; n < N; ++n) {
    std::vector<float> row(K); float amax = 0.f;
    for (int k = 0; k < K; ++k) { row[k] = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.05f * (1 + n % 7); amax = std::max(amax, fabsf(row[k])); }
    const float sc = amax / 448.f; hs[n] = sc;
    for (int k = 0; k < K; ++k) { const uint8_t q = f_to_e4m3(row[k] / sc); hW[(size_t)n * K8 + k] = q; fW[(size_t)n * K8 + k] = e4m3_to_f(q); }
    hb[n] = bf2f(f2bf(0.01f * (n % 13 - 6)));
  }
  uint8_t *A, *W, *out8; bf16* out; float *bias, *ws, *x, *slab;
  CK(hipMalloc(&A, hA.size())); CK(hipMalloc(&W, hW.size())); CK(hipMalloc(&out, (size_t)M * Npad * 2)); CK(hipMalloc(&out8, (size_t)M * Npad));
  CK(hipMalloc(&bias, Npad * 4)); CK(hipMalloc(&ws, Npad * 4)); CK(hipMalloc(&x, (size_t)(M + 4) * Npad * 4)); CK(hipMalloc(&slab, (size_t)((M + 127) / 128) * 2 * Npad * 4));
  CK(hipMemcpy(A, hA.data(), hA.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW.data(), hW.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, hb.data(), Npad * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ws, hs.data(), Npad * 4, hipMemcpyHostToDevice));
  CK(hipMemset(x, 0, (size_t)(M + 4) * Npad * 4)); CK(hipMemset(out, 0, (size_t)M * Npad * 2)); CK(hipMemset(out8, 0, (size_t)M * Npad)); CK(hipMemset(slab, 0, (size_t)((M + 127) / 128) * 2 * Npad * 4));
  GemmArgs g{};
  g.A = (const bf16*)A; g.lda = K8; g.W = (const bf16*)W; g.ldw = K8; g.bias = bias; g.wscale = ws; g.M = M; g.N = N; g.K = K8;
  g.out = epi >= 32 ? (bf16*)out8 : out; g.ldo = Npad; g.x = x; g.ldx = Npad; g.tokens = tokens; g.slab = slab; g.slab_ld = Npad;
  g.score_site = epi == 33 ? 1 : 0;
  hipStream_t s; CK(hipStreamCreate(&s));
  auto run = [&]() { switch (epi) { case 30: launch256f8<EPI_BF16>(g, s); break; case 31: launch256f8<EPI_RESID>(g, s); break;
                                    case 32: launch256f8<EPI_FC1, 0>(g, s); break; case 33: launch256f8<EPI_FC1, 1>(g, s); break; } };
  run(); CK(hipStreamSynchronize(s));
  // reference on a sample of rows (all columns): full check is M*N*K host flops
  std::vector<uint16_t> ho((size_t)M * Npad); std::vector<uint8_t> ho8((size_t)M * Npad); std::vector<float> hx((size_t)(M + 4) * Npad);
  CK(hipMemcpy(ho.data(), out, ho.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(ho8.data(), out8, ho8.size(), hipMemcpyDeviceToHost));
  CK(hipMemcpy(hx.data(), x, hx.size() * 4, hipMemcpyDeviceToHost));
  size_t bad = 0, checked = 0; double maxrel = 0;
  const int step = std::max(1, M / 97);
  for (int m = 0; m < M; m += (m < 300 || m > M - 300) ? 1 : step)
    for (int n = 0; n < N; ++n) {
      double acc = 0, mag = 0;
      for (int k = 0; k < K; ++k) { const double t = (double)fA[(size_t)m * K8 + k] * fW[(size_t)n * K8 + k]; acc += t; mag += fabs(t); }
      const float pre = bf2f(f2bf((float)(acc * hs[n]) + hb[n]));
      const float sumtol = (float)(4e-6 * mag * hs[n]);          // accumulation error where the terms cancel (the scaled MFMA aligns the products of a 64-deep step before adding: ~2e-6 of sum|terms| measured)
      ++checked;
      if (epi == 30) { const float got = bf2f(ho[(size_t)m * Npad + n]); const float d = fabsf(got - pre); const float tol = fabsf(pre) * 0.0079f + sumtol + 1e-6f; if (d > tol) { if (bad < 8) printf("  out[%d][%d] = %g, ref %g\n", m, n, got, pre); ++bad; } maxrel = std::max(maxrel, (double)d / (fabsf(pre) + 1e-3)); }
      else if (epi == 31) { const float got = hx[(size_t)m * Npad + n]; const float d = fabsf(got - pre); const float tol = fabsf(pre) * 0.0079f + sumtol + 1e-6f; if (d > tol) { if (bad < 8) printf("  x[%d][%d] = %g, ref %g\n", m, n, got, pre); ++bad; } maxrel = std::max(maxrel, (double)d / (fabsf(pre) + 1e-3)); }
      else { const float ge = bf2f(f2bf(0.5f * pre * (1.f + erff(pre * 0.70710678f)))); const float got = e4m3_to_f(ho8[(size_t)m * Npad + n]);
             const float d = fabsf(got - ge); const float tol = fabsf(ge) * 0.13f + 0.004f; if (d > tol) { if (bad < 8) printf("  act8[%d][%d] = %g, ref gelu %g (pre %g)\n", m, n, got, ge, pre); ++bad; } }
    }
  if (epi == 31) { for (size_t i = (size_t)M * Npad; i < hx.size(); ++i) if (hx[i] != 0.f) { ++bad; } }
  printf("fp8 epi %d  M=%d N=%d K=%d: %zu of %zu checked elements outside tolerance%s  (max rel err %.3g)\n", epi, M, N, K, bad, checked, bad ? "  <-- FAIL" : " (ok)", maxrel);
  for (int i = 0; i < 3; ++i) run();
  CK(hipStreamSynchronize(s));
  std::vector<float> ms(iters);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < iters; ++i) { CK(hipEventRecord(a, s)); run(); CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms[i], a, b)); }
  std::sort(ms.begin(), ms.end());
  const double fl = 2.0 * M * (double)N * K;
  printf("M=%d N=%d K=%d epi=%d (fp8)  median %.1f us  min %.1f us  -> %.0f TFLOP/s (median) %.0f (min)\n", M, N, K, epi, ms[iters / 2] * 1e3, ms[0] * 1e3,
         fl / (ms[iters / 2] * 1e-3) / 1e12, fl / (ms[0] * 1e-3) / 1e12);
  return bad ? 2 : 0;
}

int main(int argc, char** argv) {
  { hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); nCU = pr.multiProcessorCount; }
  if (argc > 4 && atoi(argv[4]) >= 30 && atoi(argv[4]) <= 33)
    return run_fp8(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), argc > 5 ? atoi(argv[5]) : 20, argc > 6 ? atoi(argv[6]) : 197);
  int M = argc > 1 ? atoi(argv[1]) : 12608, N = argc > 2 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
  int epi = argc > 4 ? atoi(argv[4]) : 2, iters = argc > 5 ? atoi(argv[5]) : 20, tokens = argc > 6 ? atoi(argv[6]) : 197;
  int Npad = (N + 255) / 256 * 256;
  std::vector<uint16_t> hA((size_t)M * K), hW((size_t)Npad * K);
  srand(1);
  for (auto& v : hA) v = f2bf((rand() / (float)RAND_MAX) * 2.f - 1.f);
  for (auto& v : hW) v = f2bf(((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.05f);
  bf16 *A, *W, *out; float *bias, *x, *slab, *pos;
  CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&W, hW.size() * 2)); CK(hipMalloc(&out, (size_t)M * Npad * 2));
  CK(hipMalloc(&bias, Npad * 4)); CK(hipMalloc(&x, (size_t)(M + M / 196 + 2) * Npad * 4)); CK(hipMalloc(&slab, (size_t)((M + 127) / 128) * 2 * Npad * 4));
  CK(hipMalloc(&pos, (size_t)200 * Npad * 4));
  CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(bias, 0, Npad * 4)); CK(hipMemset(x, 0, (size_t)(M + M / 196 + 2) * Npad * 4)); CK(hipMemset(pos, 0, (size_t)200 * Npad * 4));
  GemmArgs g{};
  g.A = A; g.lda = K; g.W = W; g.ldw = K; g.bias = bias; g.M = M; g.N = N; g.K = K;
  g.tiles_m = (M + 127) / 128; g.tiles_n = Npad / 128; g.out = out; g.ldo = Npad; g.x = x; g.ldx = Npad;
  g.score_site = (epi == 2 || epi == 13) ? 1 : ((epi == 5 || epi == 14) ? 2 : 0); g.tokens = tokens; g.slab = slab; g.slab_ld = Npad; g.pos = pos; g.patches = 196; g.group_m = argc > 7 ? atoi(argv[7]) : 0;
#ifdef GEMM_STAMPS
  unsigned long long* stamps; CK(hipMalloc(&stamps, (size_t)g.tiles_m * g.tiles_n * 64 * 8)); CK(hipMemset(stamps, 0, (size_t)g.tiles_m * g.tiles_n * 64 * 8));
  g.stamps = stamps;
#endif
  bf16 *h = nullptr, *h_ref = nullptr; float *lng = nullptr, *lnb = nullptr;
  if (epi == 15) {
    if (N % 256 || N / 256 < 3 || N / 256 > 5) { printf("epi 15 needs N = 768, 1024 or 1280\n"); return 1; }
    std::vector<float> hg(N), hb(N);
    for (int i = 0; i < N; ++i) { hg[i] = 0.5f + (rand() / (float)RAND_MAX); hb[i] = (rand() / (float)RAND_MAX) - 0.5f; }
    CK(hipMalloc(&h, (size_t)(M + 2) * N * 2)); CK(hipMalloc(&h_ref, (size_t)(M + 2) * N * 2)); CK(hipMalloc(&lng, N * 4)); CK(hipMalloc(&lnb, N * 4));
    CK(hipMemcpy(lng, hg.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(lnb, hb.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(h, 0, (size_t)(M + 2) * N * 2)); CK(hipMemset(h_ref, 0, (size_t)(M + 2) * N * 2));
    g.ln_g = lng; g.ln_b = lnb; g.ln_out = h; g.ln_ld = N; g.ln_eps = 1e-6f;
  }
  hipStream_t s; CK(hipStreamCreate(&s));
  auto run = [&]() {
    switch (epi) { case 15: if (N == 768) launch256ln<3>(g, s); else if (N == 1024) launch256ln<4>(g, s); else launch256ln<5>(g, s); break;
                   case 0: launch<EPI_BF16>(g, s); break; case 1: launch<EPI_RESID>(g, s); break; case 2: launch<EPI_FC1, 1>(g, s); break; case 5: launch<EPI_FC1, 2>(g, s); break; case 6: launch<EPI_FC1, 0>(g, s); break;
                   case 3: launch<EPI_PATCH>(g, s); break; case 4: launch<EPI_F32>(g, s); break;
                   case 10: launch256<EPI_BF16>(g, s); break; case 11: launch256<EPI_RESID>(g, s); break;
                   case 12: launch256<EPI_FC1>(g, s); break; case 13: launch256<EPI_FC1, 1>(g, s); break; case 14: launch256<EPI_FC1, 2>(g, s); break;
                   case 40: launch256w4<EPI_BF16>(g, s); break; case 41: launch256w4<EPI_RESID>(g, s); break; case 42: launch256w4<EPI_FC1>(g, s); break;
                   default: break; }
  };
  if (epi >= 10) {   // verify the large-tile kernel against the (oracle-validated) 128x128 kernel, bit for bit
    const bool resid = (epi % 10) == 1 || epi == 15, fc1 = (epi % 10) == 2, sc1 = epi == 13, sc2 = epi == 14;
    const size_t se = (size_t)((M + 127) / 128) * 2 * Npad;
    std::vector<float> slab_ref(se);
    size_t xe = (size_t)(M + M / 196 + 2) * Npad, oe = (size_t)M * Npad;
    CK(hipMemset(x, 0, xe * 4)); CK(hipMemset(out, 0, oe * 2));
    GemmArgs r = g; r.tiles_m = (M + 127) / 128; r.tiles_n = Npad / 128;
    if (sc1 || sc2) CK(hipMemset(slab, 0, se * 4));
    if (resid) launch<EPI_RESID>(r, s); else if (fc1) launch<EPI_FC1, 0>(r, s); else if (sc1) launch<EPI_FC1, 1>(r, s); else if (sc2) launch<EPI_FC1, 2>(r, s); else launch<EPI_BF16>(r, s);
    CK(hipStreamSynchronize(s));
    if (sc1 || sc2) { CK(hipMemcpy(slab_ref.data(), slab, se * 4, hipMemcpyDeviceToHost)); CK(hipMemset(slab, 0, se * 4)); }
    std::vector<float> xr(resid ? xe : 0); std::vector<uint16_t> orf(resid ? 0 : oe);
    if (resid) CK(hipMemcpy(xr.data(), x, xe * 4, hipMemcpyDeviceToHost)); else CK(hipMemcpy(orf.data(), out, oe * 2, hipMemcpyDeviceToHost));
    CK(hipMemset(x, 0, xe * 4)); CK(hipMemset(out, 0, oe * 2));
    run(); CK(hipStreamSynchronize(s));
    size_t bad = 0;
    if (resid) { std::vector<float> xn(xe); CK(hipMemcpy(xn.data(), x, xe * 4, hipMemcpyDeviceToHost));
                 for (size_t i = 0; i < (size_t)M * Npad; ++i) { size_t rr = i / Npad, cc = i % Npad; if ((int)cc < N && memcmp(&xn[rr * Npad + cc], &xr[rr * Npad + cc], 4)) { if (bad < 16) printf("  x[%zu][%zu] = %.9g vs ref %.9g\n", rr, cc, xn[rr * Npad + cc], xr[rr * Npad + cc]); ++bad; } } }
    else { std::vector<uint16_t> on(oe); CK(hipMemcpy(on.data(), out, oe * 2, hipMemcpyDeviceToHost));
           for (size_t i = 0; i < oe; ++i) { if ((int)(i % Npad) < N && on[i] != orf[i]) ++bad; } }
    if (sc1 || sc2) {
      std::vector<float> sn(se); CK(hipMemcpy(sn.data(), slab, se * 4, hipMemcpyDeviceToHost));
      size_t sbad = 0; double ssum = 0;
      for (size_t i = 0; i < se; ++i) { if ((int)(i % Npad) < N) { if (memcmp(&sn[i], &slab_ref[i], 4)) { if (sbad < 12) printf("  slab[tile %zu seg %zu col %zu] = %.9g vs ref %.9g\n", i / Npad / 2, (i / Npad) % 2, i % Npad, sn[i], slab_ref[i]); ++sbad; } ssum += sn[i]; } }
      printf("score slab vs 128x128 kernel: %zu mismatching of %zu (sum %.6g)%s\n", sbad, se, ssum, sbad ? "  <-- FAIL" : " (bit-identical)");
      bad += sbad;
    }
    if (resid) {   // rows past M must stay untouched (x is allocated with a few spare rows, zeroed above)
      std::vector<float> xt(xe); CK(hipMemcpy(xt.data(), x, xe * 4, hipMemcpyDeviceToHost));
      size_t spill = 0; for (size_t i = (size_t)M * Npad; i < xe; ++i) if (xt[i] != 0.f) ++spill;
      if (spill) { printf("  %zu elements written past row M  <-- FAIL\n", spill); bad += spill; }
    }
    if (resid && bad) {
      std::vector<float> xn(xe); CK(hipMemcpy(xn.data(), x, xe * 4, hipMemcpyDeviceToHost));
      std::vector<long> hc(64, 0), hr(128, 0), ht(16, 0);
      for (size_t i = 0; i < (size_t)M * Npad; ++i) { size_t rr = i / Npad, cc = i % Npad; if ((int)cc < N && memcmp(&xn[i], &xr[i], 4)) { ++hc[cc % 64]; ++hr[rr % 128]; ++ht[(rr / 256) % 16]; } }
      printf("  mismatches by column %% 64:"); for (int i = 0; i < 64; ++i) if (hc[i]) printf(" %d:%ld", i, hc[i]); printf("\n");
      printf("  mismatches by row %% 128:"); for (int i = 0; i < 128; ++i) if (hr[i]) printf(" %d:%ld", i, hr[i]); printf("\n");
      printf("  mismatches by (row / 256) %% 16:"); for (int i = 0; i < 16; ++i) printf(" %ld", ht[i]); printf("\n");
    }
    printf("verify vs 128x128 kernel: %zu mismatching elements%s\n", bad, bad ? "  <-- FAIL" : " (bit-identical)");
    if (epi == 15) {   // h of the fused phase vs the standalone LayerNorm kernel on the x the GEMM left behind
      const int V = N / 256;
      dim3 grid((M + 3) / 4), blk(256);
      if (V == 3) hipLaunchKernelGGL(layernorm_bf16_kernel<3>, grid, blk, 0, s, x, (size_t)Npad, lng, lnb, h_ref, N, M, N, 1e-6f, RowMap{0, 0, 0}, (uint8_t*)nullptr);
      else if (V == 4) hipLaunchKernelGGL(layernorm_bf16_kernel<4>, grid, blk, 0, s, x, (size_t)Npad, lng, lnb, h_ref, N, M, N, 1e-6f, RowMap{0, 0, 0}, (uint8_t*)nullptr);
      else hipLaunchKernelGGL(layernorm_bf16_kernel<5>, grid, blk, 0, s, x, (size_t)Npad, lng, lnb, h_ref, N, M, N, 1e-6f, RowMap{0, 0, 0}, (uint8_t*)nullptr);
      CK(hipStreamSynchronize(s));
      std::vector<uint16_t> a((size_t)(M + 2) * N), b((size_t)(M + 2) * N);
      CK(hipMemcpy(a.data(), h, a.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), h_ref, b.size() * 2, hipMemcpyDeviceToHost));
      size_t hb = 0, nz = 0; for (size_t i = 0; i < a.size(); ++i) { if (a[i] != b[i]) { if (hb < 8) printf("  h[%zu][%zu] = %04x vs %04x\n", i / N, i % N, a[i], b[i]); ++hb; } nz += a[i] != 0; }
      printf("fused LayerNorm vs layernorm_bf16_kernel: %zu mismatching of %zu (%zu non-zero)%s\n", hb, a.size(), nz, (hb || !nz) ? "  <-- FAIL" : " (bit-identical, rows past M untouched)");
      bad += hb + (nz ? 0 : 1);
      {   // ... and both against a host LayerNorm in double on a sample of rows (one bf16 rounding + fp32 arithmetic)
        std::vector<float> hx((size_t)M * Npad), hg(N), hbt(N);
        CK(hipMemcpy(hx.data(), x, hx.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hg.data(), lng, N * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hbt.data(), lnb, N * 4, hipMemcpyDeviceToHost));
        size_t off = 0, chk = 0;
        for (int m = 0; m < M; m += (m < 600 || m > M - 600) ? 1 : 211) {
          double mu = 0, var = 0;
          for (int c = 0; c < N; ++c) mu += hx[(size_t)m * Npad + c];
          mu /= N;
          for (int c = 0; c < N; ++c) { const double d = hx[(size_t)m * Npad + c] - mu; var += d * d; }
          const double rstd = 1.0 / sqrt(var / N + 1e-6);
          for (int c = 0; c < N; ++c) {
            const double ref = (hx[(size_t)m * Npad + c] - mu) * rstd * hg[c] + hbt[c];
            const double got = bf2f(a[(size_t)m * N + c]);
            ++chk;
            if (fabs(got - ref) > fabs(ref) * 0.0040 + 2e-5) { if (off < 8) printf("  h[%d][%d] = %g, host %g\n", m, c, got, ref); ++off; }
          }
        }
        printf("fused LayerNorm vs host double: %zu of %zu sampled elements outside one bf16 rounding%s\n", off, chk, off ? "  <-- FAIL" : " (ok)");
        bad += off;
        if (hb) {   // diagnosis: which rows differ, and is a row explained by ONE stale piece of x (x was zero before this launch)?
          int shown = 0; size_t bad_rows = 0; int last_row = -1;
          for (int m = 0; m < M; ++m) {
            bool differs = false;
            for (int c = 0; c < N && !differs; ++c) differs = a[(size_t)m * N + c] != b[(size_t)m * N + c];
            if (!differs) continue;
            ++bad_rows; last_row = m;
            if (shown >= 6) continue;
            ++shown;
            auto score = [&](int z0, int z1) {   // elements of the device row within one bf16 step of a host LayerNorm with columns [z0, z1) zeroed
              double mu = 0, var = 0;
              for (int c = 0; c < N; ++c) mu += (c >= z0 && c < z1) ? 0.0 : hx[(size_t)m * Npad + c];
              mu /= N;
              for (int c = 0; c < N; ++c) { const double d = ((c >= z0 && c < z1) ? 0.0 : hx[(size_t)m * Npad + c]) - mu; var += d * d; }
              const double rstd = 1.0 / sqrt(var / N + 1e-6);
              int ok = 0;
              for (int c = 0; c < N; ++c) {
                const double ref = (((c >= z0 && c < z1) ? 0.0 : hx[(size_t)m * Npad + c]) - mu) * rstd * hg[c] + hbt[c];
                ok += fabs(bf2f(a[(size_t)m * N + c]) - ref) <= fabs(ref) * 0.0040 + 2e-5;
              }
              return ok;
            };
            int best = score(0, 0), bz0 = 0, bz1 = 0;
            for (int w : {4, 32, 64, 128, 256})
              for (int z = 0; z + w <= N; z += w) { const int sc = score(z, z + w); if (sc > best) { best = sc; bz0 = z; bz1 = z + w; } }
            printf("  row %d (panel %d, row %d of it; wave %d, turn %d): as stored %d of %d elements fit the final x; best single stale piece: columns [%d, %d) -> %d fit\n",
                   m, m / 256, m % 256, (m % 256) % 8, (m % 256) / 8, score(0, 0), N, bz0, bz1, best);
          }
          printf("  %zu rows differ (last: %d)\n", bad_rows, last_row);
        }
      }
      {   // what the fused phase replaces: the standalone kernel over the same rows
        const int V2 = N / 256; dim3 grid2((M + 3) / 4), blk2(256);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float best = 1e9f;
        for (int it = 0; it < 7; ++it) {
          CK(hipEventRecord(e0, s));
          if (V2 == 3) hipLaunchKernelGGL(layernorm_bf16_kernel<3>, grid2, blk2, 0, s, x, (size_t)Npad, lng, lnb, h_ref, N, M, N, 1e-6f, RowMap{0, 0, 0}, (uint8_t*)nullptr);
          else if (V2 == 4) hipLaunchKernelGGL(layernorm_bf16_kernel<4>, grid2, blk2, 0, s, x, (size_t)Npad, lng, lnb, h_ref, N, M, N, 1e-6f, RowMap{0, 0, 0}, (uint8_t*)nullptr);
          else hipLaunchKernelGGL(layernorm_bf16_kernel<5>, grid2, blk2, 0, s, x, (size_t)Npad, lng, lnb, h_ref, N, M, N, 1e-6f, RowMap{0, 0, 0}, (uint8_t*)nullptr);
          CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); best = std::min(best, t);
        }
        printf("standalone layernorm_bf16_kernel over the same %d rows: %.1f us\n", M, best * 1e3);
      }
      CK(hipMemset(h, 0, a.size() * 2));
    }
    CK(hipMemset(x, 0, xe * 4));
  }
  if (const char* st = getenv("GEMM_STRESS")) {   // race screen: many launches, every result compared with the first
    const int reps = atoi(st);
    const bool resid = (epi % 10) == 1;
    size_t xe = (size_t)(M + M / 196 + 2) * Npad, oe = (size_t)M * Npad, se = (size_t)((M + 127) / 128) * 2 * Npad;
    std::vector<float> x0(resid ? xe : 0), xi(resid ? xe : 0), s0(se), si(se); std::vector<uint16_t> o0(resid ? 0 : oe), oi(resid ? 0 : oe);
    int bad_runs = 0;
    for (int r = 0; r <= reps; ++r) {
      CK(hipMemset(x, 0, xe * 4)); CK(hipMemset(out, 0, oe * 2)); CK(hipMemset(slab, 0, se * 4));
      run(); CK(hipStreamSynchronize(s));
      if (resid) CK(hipMemcpy((r ? xi : x0).data(), x, xe * 4, hipMemcpyDeviceToHost)); else CK(hipMemcpy((r ? oi : o0).data(), out, oe * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy((r ? si : s0).data(), slab, se * 4, hipMemcpyDeviceToHost));
      if (r && ((resid ? memcmp(xi.data(), x0.data(), xe * 4) : memcmp(oi.data(), o0.data(), oe * 2)) || memcmp(si.data(), s0.data(), se * 4))) ++bad_runs;
    }
    printf("stress: %d of %d launches differ from the first%s\n", bad_runs, reps, bad_runs ? "  <-- FAIL" : " (deterministic)");
    CK(hipMemset(x, 0, xe * 4));
  }
  for (int i = 0; i < 3; ++i) run();
  CK(hipStreamSynchronize(s));
  std::vector<float> ms(iters);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < iters; ++i) { CK(hipEventRecord(a, s)); run(); CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms[i], a, b)); }
  std::sort(ms.begin(), ms.end());
  double fl = 2.0 * M * (double)N * K;
  printf("M=%d N=%d K=%d epi=%d blocks=%d  median %.1f us  min %.1f us  -> %.0f TFLOP/s (median) %.0f (min)\n", M, N, K, epi,
         g.tiles_m * g.tiles_n, ms[iters / 2] * 1e3, ms[0] * 1e3, fl / (ms[iters / 2] * 1e-3) / 1e12, fl / (ms[0] * 1e-3) / 1e12);
  if (const char* sus = getenv("GEMM_SUSTAIN")) {
    // SUSTAINED rate: launches back to back with no host synchronisation in between (the loop above waits for every launch, so
    // the card idles between two of them and the power controller sees bursts; the step's launches run back to back).  Chunks of
    // 50 launches between events, all recorded first and read after one final synchronisation.
    const int n = atoi(sus), per = 50, chunks = (n + per - 1) / per;
    std::vector<hipEvent_t> ev(chunks + 1);
    for (auto& e : ev) CK(hipEventCreate(&e));
    CK(hipEventRecord(ev[0], s));
    for (int c = 0; c < chunks; ++c) { for (int i = 0; i < per; ++i) run(); CK(hipEventRecord(ev[c + 1], s)); }
    CK(hipStreamSynchronize(s));
    std::vector<float> cm(chunks);
    for (int c = 0; c < chunks; ++c) { CK(hipEventElapsedTime(&cm[c], ev[c], ev[c + 1])); cm[c] /= per; }
    float tot = 0; for (float v : cm) tot += v; tot /= chunks;
    printf("  sustained (%d launches back to back): %.1f us per launch -> %.0f TFLOP/s   [first 50: %.1f us, last 50: %.1f us]\n", chunks * per,
           tot * 1e3, fl / (tot * 1e-3) / 1e12, cm[0] * 1e3, cm[chunks - 1] * 1e3);
  }
#ifdef GEMM_STAMPS
  {
    int nb = g.tiles_m * g.tiles_n, nk = K / 64;
    std::vector<unsigned long long> h((size_t)nb * 64);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    double mfma = 0, wait = 0, bar = 0, tot = 0; long cnt = 0;
    unsigned long long tmin = ~0ULL, tmax = 0;
    for (int b = 0; b < nb; ++b) {
      const unsigned long long* t = &h[(size_t)b * 64];
      tmin = std::min(tmin, t[0]); tmax = std::max(tmax, t[60]);
      unsigned long long prev = t[0];
      for (int k = 0; k < nk && 3 + 3 * k < 60; ++k) {
        mfma += (double)(t[1 + 3 * k] - prev); wait += (double)(t[2 + 3 * k] - t[1 + 3 * k]); bar += (double)(t[3 + 3 * k] - t[2 + 3 * k]);
        prev = t[3 + 3 * k]; ++cnt;
      }
      tot += (double)(t[60] - t[0]);
    }
    printf("stamps (cycles, wave 0 of every block): per-iteration issue+mfma %.0f  vmcnt-wait %.0f  barrier %.0f ; main loop per block %.0f ; kernel span %.0f\n",
           mfma / cnt, wait / cnt, bar / cnt, tot / nb, (double)(tmax - tmin));
    if (epi >= 40 && epi < 50) {   // four-wave kernel: third tile of every workgroup (wave 0)
      int G = std::min(((M + 255) / 256) * ((N + 255) / 256), nCU);
      double pro = 0, ml = 0, ep = 0, w_l = 0, w_v = 0, w_b = 0, per = 0; long n = 0, nw = 0, np = 0;
      for (int b = 0; b < G; ++b) {
        const unsigned long long* t = &h[(size_t)b * 64];
        if (!t[0] || !t[59] || !t[58]) continue;
        pro += (double)(t[1] - t[0]); ml += (double)(t[58] - t[1]); ep += (double)(t[59] - t[58]); ++n;
        for (int k = 0; k < nk - 1 && k < 14; ++k) {
          w_l += (double)(t[3 + 4 * k] - t[2 + 4 * k]); w_v += (double)(t[4 + 4 * k] - t[3 + 4 * k]); w_b += (double)(t[5 + 4 * k] - t[4 + 4 * k]); ++nw;
          if (k > 0) { per += (double)(t[2 + 4 * k] - t[2 + 4 * (k - 1)]); ++np; }
        }
      }
      if (n) printf("w4 third tile (wave 0, avg of %ld workgroups, cycles): prologue (issue + wait + barrier) %.0f | main loop %.0f (K-tile period %.0f; at the barrier: lgkmcnt wait %.0f, vmcnt wait %.0f, barrier %.0f) | epilogue %.0f\n",
                    n, pro / n, ml / n, np ? per / np : 0, w_l / nw, w_v / nw, w_b / nw, ep / n);
    } else
    if (epi >= 10) {   // persistent kernel: per-workgroup totals
      int G = std::min(((M + 255) / 256) * ((N + 255) / 256), nCU);
      double cyc = 0, rt = 0, tl = 0;
      for (int b = 0; b < G; ++b) { const unsigned long long* t = &h[(size_t)b * 64]; cyc += (double)(t[59] - t[0]); rt += (double)(t[62] - t[61]); tl += (double)t[63]; }
      printf("persistent: %d workgroups, avg tiles/wg %.2f, avg lifetime %.0f cycles = %.1f us (s_memrealtime, 100 MHz) -> clock %.2f GHz ; cycles per tile %.0f\n",
             G, tl / G, cyc / G, rt / G / 100.0, (cyc / G) / (rt / G / 100.0) / 1e3, cyc / tl);
      if (epi < 20) {  // third tile of every workgroup that ran >= 3: tile-start wait, main loop (compute / wait / barrier per K-tile), prefetch issue, epilogue
        double w0 = 0, ml = 0, pf = 0, ep = 0, d[4] = {0, 0, 0, 0}, per = 0; long n = 0, nkt = 0;
        for (int b = 0; b < G; ++b) {
          const unsigned long long* t = &h[(size_t)b * 64];
          if (t[63] < 3 || !t[40] || !t[41]) continue;
          w0 += (double)(t[41] - t[42]); ml += (double)(t[60] - t[41]); pf += (double)(t[43] - t[60]); ep += (double)(t[40] - t[43]); ++n;
          for (int k = 1; k < nk && k < 6; ++k) {
            for (int j = 0; j < 4; ++j) d[j] += (double)(t[2 + 6 * k + j] - t[1 + 6 * k + j]);
            per += (double)(t[1 + 6 * k] - t[1 + 6 * (k - 1)]); ++nkt;
          }
        }
        { double e[6] = {0,0,0,0,0,0}; long ne = 0;
          for (int b = 0; b < G; ++b) { const unsigned long long* t = &h[(size_t)b * 64]; if (t[63] < 3 || !t[44] || !t[48]) continue;
            e[0] += (double)(t[44] - t[43]); e[1] += (double)(t[45] - t[44]); e[2] += (double)(t[46] - t[45]); e[3] += (double)(t[47] - t[46]); e[4] += (double)(t[48] - t[47]); e[5] += (double)(t[40] - t[48]); ++ne; }
          if (ne) printf("epilogue of the third tile (wave 0; bf16 / gelu paths): bias shuffles %.0f | skew barrier %.0f | pass 0: convert + LDS writes %.0f, LDS read-back %.0f, 8 stores %.0f | pass 1 %.0f\n",
                         e[0] / ne, e[1] / ne, e[2] / ne, e[3] / ne, e[4] / ne, e[5] / ne); }
        if (n) printf("third tile (wave 0, avg of %ld workgroups, cycles): start-wait+barrier %.0f | main loop %.0f (K-tile period %.0f; first unit: 12 ds_read issue %.0f, lgkmcnt wait %.0f, barrier %.0f, 16 MFMA + 4 DMA issue %.0f) | next-tile set-up + DMA issue %.0f | epilogue %.0f\n",
                      n, w0 / n, ml / n, per / nkt, d[0] / nkt, d[1] / nkt, d[2] / nkt, d[3] / nkt, pf / n, ep / n);
      }
    }
    // distribution of block start times (first 16 and a few later)
    for (int b : {0, 1, 255, 256, 511, 512, 513, 1000, 2000}) if (b < nb) printf("  block %d start %+lld loop %lld\n", b, (long long)(h[(size_t)b * 64] - tmin), (long long)(h[(size_t)b * 64 + 60] - h[(size_t)b * 64]));
  }
#endif
  return 0;
}
