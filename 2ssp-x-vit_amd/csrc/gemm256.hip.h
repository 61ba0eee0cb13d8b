// Large-tile variant of the bf16 MFMA GEMM for the big-M projections (QKV, attention out-proj, fc2):
// 256 x BN x 64 tiles, 512 threads = 8 waves, ONE workgroup per CU.
//
// Why a second kernel: in the 128x128 kernel every v_mfma_f32_32x32x16 (32 cycles of matrix pipe) is accompanied by
// 0.5 LDS-DMA issues (~60 cycles each) and 1.0 ds_read_b128 (~23 cycles each) in the wave's own instruction stream,
// so two waves per SIMD cannot keep the matrix pipe busy (measured with s_memtime stamps: 1665 cycles per K-tile for
// 512 cycles of MFMA per wave).  A 256 x 256 tile with 128 x 64 per wave quarters the DMA issues per MFMA (0.125) and
// cuts the fragment reads to 0.75; 256 x 128 (64 x 64 per wave) keeps more tiles in flight for N = 768 and affords a
// three-deep ring.
//
//   BN = 256: waves 2(M) x 4(N), wave tile 128 x 64, LDS ring 2 x 64 KiB  (prefetch distance 1 tile = 32 MFMAs/wave)
//   BN = 128: waves 4(M) x 2(N), wave tile  64 x 64, LDS ring 3 x 48 KiB  (prefetch distance 2 tiles, counted vmcnt)
//
// Staging, swizzle, fragment layout and the LDS-staged vector epilogues are those of gemm.hip.h.
#pragma once
#include "gemm.hip.h"

template <int BN> struct G256 {
  static constexpr int BM = 256;
  static constexpr int WM = BN == 256 ? 2 : 4;          // waves along M
  static constexpr int WN = BN == 256 ? 4 : 2;          // waves along N
  static constexpr int TM = BM / WM / 32;               // 32x32 MFMA tiles per wave along M (4 or 2)
  static constexpr int TN = 2;
  static constexpr int S = BN == 256 ? 2 : 3;           // ring depth
  static constexpr int A_BYTES = BM * 128;              // 256 rows x 64 bf16
  static constexpr int B_BYTES = BN * 128;
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int LDS = S * STAGE;                 // 131072 / 147456
  static constexpr int GA = BM / 8 / 8;                 // LDS-DMA pieces per wave per stage (A): 4
  static constexpr int GB = BN / 8 / 8;                 // (B): 4 / 2
  static constexpr int G = GA + GB;
};

template <int EPI, int BN>
__global__ __launch_bounds__(512, 2) void gemm256_bf16_kernel(const GemmArgs g) {
  using C = G256<BN>;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WN, wn = wave - wm * C::WN;
  const int l31 = lane & 31, lh = lane >> 5;

  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = lid / g.tiles_n, tn = lid - tm * g.tiles_n;   // N fastest: neighbours share the A panel
  const int m0 = tm * C::BM, n0 = tn * BN;

  const bf16* a_src[C::GA];
  const bf16* w_src[C::GB];
#pragma unroll
  for (int i = 0; i < C::GA; ++i) {
    const int row = (wave + 8 * i) * 8 + (lane >> 3);
    const int c_src = (lane & 7) ^ ((row >> 1) & 7);
    int gr = m0 + row; gr = gr < g.M ? gr : g.M - 1;
    a_src[i] = g.A + (size_t)gr * g.lda + c_src * 8;
  }
#pragma unroll
  for (int i = 0; i < C::GB; ++i) {
    const int row = (wave + 8 * i) * 8 + (lane >> 3);
    const int c_src = (lane & 7) ^ ((row >> 1) & 7);
    w_src[i] = g.W + (size_t)(n0 + row) * g.ldw + c_src * 8;
  }
  auto stage = [&](int slot, int kt) {
    char* base = smem + slot * C::STAGE;
#pragma unroll
    for (int i = 0; i < C::GB; ++i) glds16(w_src[i] + kt * GEMM_BK, base + C::A_BYTES + (wave + 8 * i) * 1024);
#pragma unroll
    for (int i = 0; i < C::GA; ++i) glds16(a_src[i] + kt * GEMM_BK, base + (wave + 8 * i) * 1024);
  };

  f32x16 acc[C::TM][C::TN];
#pragma unroll
  for (int a = 0; a < C::TM; ++a)
#pragma unroll
    for (int b = 0; b < C::TN; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  int a_off[C::TM], a_swz[C::TM], b_off[C::TN], b_swz[C::TN];
#pragma unroll
  for (int i = 0; i < C::TM; ++i) { const int r = wm * (C::TM * 32) + i * 32 + l31; a_off[i] = r * 128; a_swz[i] = (r >> 1) & 7; }
#pragma unroll
  for (int i = 0; i < C::TN; ++i) { const int r = wn * 64 + i * 32 + l31; b_off[i] = C::A_BYTES + r * 128; b_swz[i] = (r >> 1) & 7; }

  const float bias_pre[2] = {g.bias[n0 + wn * 64 + l31], g.bias[n0 + wn * 64 + 32 + l31]};

  const int nk = g.K / GEMM_BK;
  // prologue: S-1 stages in flight, wait for the first
  stage(0, 0);
  if (C::S == 3) {
    if (nk > 1) { stage(1, 1); asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(C::G) : "memory"); }
    else        { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); }
  } else {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const int pre = kt + C::S - 1;
    if (pre < nk) { int ps = slot + C::S - 1; if (ps >= C::S) ps -= C::S; stage(ps, pre); }
    const char* St = smem + slot * C::STAGE;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int chunk = 2 * s + lh;
      bf16x8 fa[C::TM], fb[C::TN];
#pragma unroll
      for (int i = 0; i < C::TN; ++i) fb[i] = *(const bf16x8*)(St + b_off[i] + ((chunk ^ b_swz[i]) << 4));
#pragma unroll
      for (int i = 0; i < C::TM; ++i) fa[i] = *(const bf16x8*)(St + a_off[i] + ((chunk ^ a_swz[i]) << 4));
#pragma unroll
      for (int a = 0; a < C::TM; ++a)
#pragma unroll
        for (int b = 0; b < C::TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    // next tile must have landed; with S = 3 the stage issued in THIS iteration may stay in flight
    if (C::S == 3 && kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(C::G) : "memory");
    else                          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    slot = slot + 1 == C::S ? 0 : slot + 1;
  }

  // ---------------------------------------------------------------- epilogue (wave-private LDS staging)
  const int row0 = m0 + wm * (C::TM * 32), col0 = n0 + wn * 64;
  const bool wave_cols_ok = col0 < g.N;
  if (EPI == EPI_BF16) {
    char* stg = smem + wave * (C::TM * 32 * 128);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float bias = bias_pre[b];
      char* const stc = stg + (b * 32 + l31) * 2;
#pragma unroll
      for (int a = 0; a < C::TM; ++a)
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const int rw = a * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
          const bf16x2 ov = __builtin_bit_cast(bf16x2, pack_bf16x2(acc[a][b][i] + bias, acc[a][b][i + 1] + bias));
          *(bf16*)(stc + rw * 128) = ov[0];
          *(bf16*)(stc + (rw + 1) * 128) = ov[1];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (wave_cols_ok) {
#pragma unroll
      for (int it = 0; it < C::TM * 4; ++it) {
        const int r = it * 8 + (lane >> 3), c = (lane & 7) * 8;
        const bf16x8 v = *(const bf16x8*)(stg + r * 128 + c * 2);
        if (row0 + r < g.M) *(bf16x8*)(g.out + (size_t)(row0 + r) * g.ldo + col0 + c) = v;
      }
    }
  } else {   // EPI_RESID: per 32-row sub-tile, fp32 staging [32][64] then float4 read-modify-write of x
    char* stg = smem + wave * 8192;
    const int c = (lane & 15) * 4;
#pragma unroll
    for (int a = 0; a < C::TM; ++a) {
      f32x4 xin[8];
      float* dst[8];
      if (wave_cols_ok) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int m = row0 + a * 32 + it * 4 + (lane >> 4);
          const int mc = m < g.M ? m : g.M - 1;
          dst[it] = g.x + (size_t)mc * g.ldx + col0 + c;
          xin[it] = *(const f32x4*)dst[it];
        }
      }
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int rw = (i & 3) + 8 * (i >> 2) + 4 * lh;
          *(float*)(stg + rw * 256 + (b * 32 + l31) * 4) = bf16_round(acc[a][b][i] + bias_pre[b]);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (wave_cols_ok) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int r = it * 4 + (lane >> 4);
          const f32x4 v = *(const f32x4*)(stg + r * 256 + c * 4);
          if (row0 + a * 32 + r < g.M) *(f32x4*)dst[it] = xin[it] + v;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next sub-tile overwrites the staging
    }
  }
}
