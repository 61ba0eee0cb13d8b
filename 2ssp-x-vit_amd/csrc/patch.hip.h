// Patch embedding (conv k = s = p as a GEMM) with the patch tiles staged through LDS straight from the fp32 pixels:
//
//   x[row(img) + 1 + q, :] = float(bf16(sum_k bf16(pixel[img, q, k]) * W[:, k] + bias)) + pos[1 + q, :]
//
// Round 1-2 went pixels -> im2col_patch_kernel -> bf16 matrix in HBM (154 MB per 512 images, written and read back) ->
// gemm_bf16_kernel<EPI_PATCH>.  Here the A operand never exists in HBM: a workgroup gathers its 128 patches x 64 k-elements
// of a K-tile from the NCHW image with 16-byte loads (k = c*p*p + ky*p + kx: for p a multiple of 8 a 16-byte LDS chunk is 8
// consecutive pixels of one image row), rounds them to bf16 in registers (RNE, what autocast does to the conv input) and
// writes the tile into LDS in exactly the swizzled image the LDS-DMA path of gemm.hip.h produces, so the fragment reads, the
// MFMA order and the epilogue are those of gemm_bf16_kernel<EPI_PATCH>: the results are the same bits (tests flip
// SSP2_OPT_PATCH_LDS).  LDS-DMA cannot do this (it cannot convert), hence the VGPR round trip for A; the weight panel still
// arrives by LDS-DMA.  Ring: A 2 deep (registers hold tile kt+1 while tile kt is multiplied), B 2 deep.
#pragma once
#include "gemm.hip.h"

struct PatchArgs {
  const float* px;            // [n, 3, img, img] fp32 NCHW
  const bf16* W; int ldw;     // [dim_pad, Kpad] bf16, k = c*p*p + ky*p + kx, zero beyond K
  const float* bias;          // [dim_pad]
  const float* pos;           // [tokens, dim] fp32
  float* x; int ldx;          // residual stream, row layout rm
  int n, img, p, side, K, Kpad, N;
  int tiles_n;
  RowMap rm;                  // tokens = patches + 1
};

#define PATCH_LDS_BYTES (4 * GEMM_STAGE_BYTES)     // A0 A1 B0 B1; the fp32 epilogue staging (4 x 16 KiB) reuses all of it

// (waves per SIMD pinned to 2 = two 64-KiB workgroups per CU: left to its own occupancy target hipcc keeps the kernel at 152 VGPRs
// and puts the three pixel-tile register sets into scratch memory)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void patch_embed_kernel(const PatchArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const smA = smem;
  char* const smB = smem + 2 * GEMM_STAGE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int patches = g.side * g.side, M = g.n * patches;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = lid / g.tiles_n, tn = lid - tm * g.tiles_n;
  const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;

  // ---- A staging: 1024 sixteen-byte chunks per K-tile, 4 per thread: chunk (row r = j*32 + tid/8, c = tid & 7)
  const int cc = tid & 7;
  const float* row_px[4];      // pixel (0, 0) of the patch, channel 0
  int lds_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = j * 32 + (tid >> 3);
    int gm = m0 + r; gm = gm < M ? gm : M - 1;                 // rows past M are never stored
    const int im = gm / patches, q = gm - im * patches;
    const int py = q / g.side, pxx = q - py * g.side;
    row_px[j] = g.px + ((size_t)im * 3 * g.img + (size_t)py * g.p) * g.img + (size_t)pxx * g.p;
    lds_off[j] = r * 128 + ((cc ^ ((r >> 1) & 7)) << 4);
  }
  const int pp2 = g.p * g.p;
  const bool vec = (g.p & 7) == 0;                              // 8 consecutive k = 8 consecutive pixels of one row
  // The pixel loads of K-tiles kt + 1, kt + 2 and kt + 3 are in flight while tile kt is multiplied: three register sets in a
  // ring (tile t lives in set t % 3).  One set (the first version) left every iteration waiting ~2 k cycles for loads issued
  // 16 MFMAs earlier — slower than the im2col image it replaces (profiles/r03_h_step_ab.txt).
  float av0[4][8], av1[4][8], av2[4][8];
  auto load_a = [&](int kt, float (&av)[4][8]) __attribute__((always_inline)) {
    const int k0 = kt * GEMM_BK + cc * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // EVERY chunk issues its loads (a k past K re-reads k = 0 and is zeroed afterwards): the number of vector-memory
      // operations per K-tile is then the same for every thread — 8 (vec) or 32 — which the counted wait of the loop relies on
      if (vec) {
        const int kk = k0 < g.K ? k0 : 0;
        const int c = kk / pp2, rem = kk - c * pp2;
        const int ky = rem / g.p, kx = rem - ky * g.p;
        const float* s = row_px[j] + ((size_t)c * g.img + ky) * g.img + kx;
        const f32x4 a = *(const f32x4*)s, b = *(const f32x4*)(s + 4);
        const float z = k0 < g.K ? 1.f : 0.f;
        av[j][0] = a[0] * z; av[j][1] = a[1] * z; av[j][2] = a[2] * z; av[j][3] = a[3] * z;
        av[j][4] = b[0] * z; av[j][5] = b[1] * z; av[j][6] = b[2] * z; av[j][7] = b[3] * z;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = k0 + e, kk = k < g.K ? k : 0;
          const int c = kk / pp2, rem = kk - c * pp2;
          const int ky = rem / g.p, kx = rem - ky * g.p;
          const float f = row_px[j][((size_t)c * g.img + ky) * g.img + kx];
          av[j][e] = k < g.K ? f : 0.f;
        }
      }
    }
  };
  auto store_a = [&](int slot, const float (&av)[4][8]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16)av[j][e];
      *(bf16x8*)(smA + slot * GEMM_STAGE_BYTES + lds_off[j]) = v;
    }
  };
  // ---- B staging by LDS-DMA, as gemm_bf16_kernel
  const bf16* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave + 4 * i) * 8 + (lane >> 3);
    const int c_src = (lane & 7) ^ ((row >> 1) & 7);
    w_src[i] = g.W + (size_t)(n0 + row) * g.ldw + c_src * 8;
  }
  auto stage_b = [&](int slot, int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(w_src[i] + kt * GEMM_BK, smB + slot * GEMM_STAGE_BYTES + (wave + 4 * i) * 1024);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  int a_off[2], b_off[2], a_swz[2], b_swz[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = wr * 64 + i * 32 + l31, rb = wc * 64 + i * 32 + l31;
    a_off[i] = ra * 128; a_swz[i] = (ra >> 1) & 7;
    b_off[i] = rb * 128; b_swz[i] = (rb >> 1) & 7;
  }
  const float bias_pre[2] = {g.bias[n0 + wc * 64 + l31], g.bias[n0 + wc * 64 + 32 + l31]};

  const int nk = g.Kpad / GEMM_BK;
  stage_b(0, 0);
  load_a(0, av0);
  if (nk > 1) load_a(1, av1);
  if (nk > 2) load_a(2, av2);
  store_a(0, av0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();                                               // (the DMA of B(0) is waited for at the end of iteration 0's prefetch below)
  // one K-tile: SET (compile time) = kt % 3 holds tile kt's registers (already in LDS), (SET + 1) % 3 tile kt + 1
  auto ktile = [&](int kt, auto set_c) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value, NEXT = (SET + 1) % 3;
    const int s = kt & 1;
    if (kt == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // B(0) (and the first three pixel tiles) — once
    if (kt == 0) __syncthreads();
    if (kt + 1 < nk) stage_b(s ^ 1, kt + 1);                           // slot s^1 was last read in iteration kt-1 (barrier below)
    if (kt + 3 < nk) {                                                 // tile kt's registers are free: it is in LDS
      if constexpr (SET == 0) load_a(kt + 3, av0); else if constexpr (SET == 1) load_a(kt + 3, av1); else load_a(kt + 3, av2);
    }
    const char* As = smA + s * GEMM_STAGE_BYTES;
    const char* Bs = smB + s * GEMM_STAGE_BYTES;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int chunk = 2 * st + lh;
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = *(const bf16x8*)(As + a_off[i] + ((chunk ^ a_swz[i]) << 4));
        fb[i] = *(const bf16x8*)(Bs + b_off[i] + ((chunk ^ b_swz[i]) << 4));
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    if (kt + 1 < nk) {                                                 // hipcc waits for exactly these registers' loads (the younger ones stay in flight)
      if constexpr (NEXT == 0) store_a(s ^ 1, av0); else if constexpr (NEXT == 1) store_a(s ^ 1, av1); else store_a(s ^ 1, av2);
    }
    // B(kt + 1) must have landed before the next iteration reads it; it is OLDER than the pixel loads of tile kt + 3 (8 per thread)
    if (kt + 3 >= nk) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if (vec)     asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else              asm volatile("s_waitcnt vmcnt(32) lgkmcnt(0)" ::: "memory");
    __syncthreads();
  };
  for (int kt = 0; kt < nk; kt += 3) {
    ktile(kt, std::integral_constant<int, 0>{});
    if (kt + 1 < nk) ktile(kt + 1, std::integral_constant<int, 1>{});
    if (kt + 2 < nk) ktile(kt + 2, std::integral_constant<int, 2>{});
  }

  // ---- epilogue = gemm_bf16_kernel<EPI_PATCH>: fp32 staging [64][64] per wave, then x = bf16(acc + bias) + pos
  const int row0 = m0 + wr * 64, col0 = n0 + wc * 64;
  char* stg = smem + wave * 16384;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const float bias = bias_pre[b];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rw = a * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
        *(float*)(stg + rw * 256 + (b * 32 + l31) * 4) = bf16_round(acc[a][b][i] + bias);
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (col0 >= g.N) return;
  const int c = (lane & 15) * 4;
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int r = it * 4 + (lane >> 4);
    const int m = row0 + r;
    if (m < M) {
      const int im = m / patches, q = m - im * patches;
      const f32x4 v = *(const f32x4*)(stg + r * 256 + c * 4);
      const f32x4 ps = *(const f32x4*)(g.pos + (size_t)(1 + q) * g.ldx + col0 + c);
      *(f32x4*)(g.x + (size_t)(row_of(g.rm, im) + 1 + q) * g.ldx + col0 + c) = ps + v;
    }
  }
}
