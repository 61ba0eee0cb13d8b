// ROUND-1 BASELINE of the large-tile kernel (two-slot ring, full drains), kept only for A/B timing in gemm_bench.
// Large-tile variant of the bf16 MFMA GEMM for the big-M projections (QKV, attention out-proj, fc2):
// 256 x 256 x 64 tiles, 8 or 16 waves, ONE persistent workgroup per CU.
//
// Why a second kernel: in the 128x128 kernel every v_mfma_f32_32x32x16 (32 cycles of matrix pipe) is accompanied by
// 0.5 LDS-DMA issues (~60 cycles each) and 1.0 ds_read_b128 (~23 cycles each) in the wave's own instruction stream,
// so two waves per SIMD cannot keep the matrix pipe busy (measured with s_memtime stamps: 1665 cycles per K-tile for
// 512 cycles of MFMA per wave).  A 256 x 256 tile with 128 x 64 per wave quarters the DMA issues per MFMA (0.125) and
// cuts the fragment reads to 0.75; 256 x 128 (64 x 64 per wave) keeps more tiles in flight for N = 768 and affords a
// three-deep ring.
//
//   NW =  8: waves 2(M) x 4(N), wave tile 128 x 64 (0.75 fragment reads per MFMA), two waves per SIMD
//   NW = 16: waves 4(M) x 4(N), wave tile  64 x 64 (1.0 reads per MFMA), FOUR waves per SIMD (<= 128 VGPRs) — more
//            waves to cover each other's DMA-issue / LDS-read stalls at the price of LDS bandwidth
//   LDS ring 2 x 64 KiB (prefetch distance 1 tile).
//
// Staging, swizzle, fragment layout and the LDS-staged vector epilogues are those of gemm.hip.h.
#pragma once
#include "../gemm.hip.h"
#ifndef STAMP
#define STAMP(slot) do {} while (0)
#endif
#ifdef GEMM_STAMPS
#define RSTAMP(slot) do { if (tid == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g.stamps[(size_t)blockIdx.x * 64 + (slot)] = t_; } } while (0)
#else
#define RSTAMP(slot) do {} while (0)
#endif

template <int NW> struct G256v1 {
  static constexpr int BM = 256, BN = 256;
  static constexpr int WM = NW == 8 ? 2 : 4;            // waves along M
  static constexpr int WN = 4;                          // waves along N
  static constexpr int TM = BM / WM / 32;               // 32x32 MFMA tiles per wave along M (4 or 2)
  static constexpr int TN = 2;
  static constexpr int A_BYTES = BM * 128;              // 256 rows x 64 bf16
  static constexpr int B_BYTES = BN * 128;
  static constexpr int STAGE = A_BYTES + B_BYTES;       // 65536
  static constexpr int LDS = 2 * STAGE;                 // 131072
  static constexpr int GA = BM / 8 / NW;                // LDS-DMA pieces per wave per stage (A): 4 / 2
  static constexpr int GB = BN / 8 / NW;
  static constexpr int STG = STAGE / NW;                // wave-private epilogue staging bytes inside slot 1
  static constexpr int ROWS16 = STG / 128;              // bf16 staging rows per pass (64 / 32)
  static constexpr int ROWS32 = STG / 256;              // fp32 staging rows per pass (32 / 16)
};

// PERSISTENT: the grid is one workgroup per CU (or fewer tiles); each workgroup walks tiles wg, wg+G, wg+2G, ...
// A 1-workgroup-per-CU kernel exposes everything between two main loops (epilogue, workgroup launch, the first
// tile's HBM latency) — measured 46 % of the tile time for K = 768.  Here the NEXT tile's first K-stage is issued by
// LDS-DMA into ring slot 0 right after the main loop, and the epilogue stages through slot 1, so the DMA flight,
// the address set-up and the bias load overlap the epilogue's VALU / store work and no launch sits in between.
template <int EPI, int NW>
__global__ __launch_bounds__(NW * 64) void gemm256v1_bf16_kernel(const GemmArgs g) {
  using C = G256v1<NW>;
  constexpr int BN = C::BN;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WN, wn = wave - wm * C::WN;
  const int l31 = lane & 31, lh = lane >> 5;
  const int ntiles = g.tiles_m * g.tiles_n;
  const int G = gridDim.x;
  // workgroups b, b+8, .. share an XCD: give each XCD a contiguous run of tile ids inside every round of G tiles
  const int wg = xcd_remap(blockIdx.x, G);

  int a_off[C::TM], a_swz[C::TM], b_off[C::TN], b_swz[C::TN];
#pragma unroll
  for (int i = 0; i < C::TM; ++i) { const int r = wm * (C::TM * 32) + i * 32 + l31; a_off[i] = r * 128; a_swz[i] = (r >> 1) & 7; }
#pragma unroll
  for (int i = 0; i < C::TN; ++i) { const int r = wn * 64 + i * 32 + l31; b_off[i] = C::A_BYTES + r * 128; b_swz[i] = (r >> 1) & 7; }

  const bf16* a_src[C::GA];
  const bf16* w_src[C::GB];
  int m0 = 0, n0 = 0;
  float bias_pre[2];
  auto set_tile = [&](int tile) {
    const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;   // N fastest: neighbours share the A panel
    m0 = tm * C::BM; n0 = tn * BN;
#pragma unroll
    for (int i = 0; i < C::GA; ++i) {
      const int row = (wave + NW * i) * 8 + (lane >> 3);
      const int c_src = (lane & 7) ^ ((row >> 1) & 7);
      int gr = m0 + row; gr = gr < g.M ? gr : g.M - 1;
      a_src[i] = g.A + (size_t)gr * g.lda + c_src * 8;
    }
#pragma unroll
    for (int i = 0; i < C::GB; ++i) {
      const int row = (wave + NW * i) * 8 + (lane >> 3);
      const int c_src = (lane & 7) ^ ((row >> 1) & 7);
      w_src[i] = g.W + (size_t)(n0 + row) * g.ldw + c_src * 8;
    }
  };
  auto stage = [&](int slot, int kt) {
    char* base = smem + slot * C::STAGE;
#pragma unroll
    for (int i = 0; i < C::GB; ++i) glds16(w_src[i] + kt * GEMM_BK, base + C::A_BYTES + (wave + NW * i) * 1024);
#pragma unroll
    for (int i = 0; i < C::GA; ++i) glds16(a_src[i] + kt * GEMM_BK, base + (wave + NW * i) * 1024);
  };

  const int nk = g.K / GEMM_BK;
  int tile = wg;
  if (tile < ntiles) {
    set_tile(tile);
    bias_pre[0] = g.bias[n0 + wn * 64 + l31]; bias_pre[1] = g.bias[n0 + wn * 64 + 32 + l31];
    stage(0, 0);
  }
  STAMP(0);
  RSTAMP(61);
  int tiles_done = 0;
  for (; tile < ntiles; tile += G) {
    ++tiles_done;
    f32x16 acc[C::TM][C::TN];
#pragma unroll
    for (int a = 0; a < C::TM; ++a)
#pragma unroll
      for (int b = 0; b < C::TN; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    // stage 0 of this tile was issued before the previous epilogue (or above): drain it (and the epilogue's
    // stores), then everybody may read slot 0 and DMA may overwrite slot 1 (all staging reads are done).
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) stage(slot ^ 1, kt + 1);
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
      STAMP(1 + 3 * kt);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      STAMP(2 + 3 * kt);
      asm volatile("s_barrier" ::: "memory");
      STAMP(3 + 3 * kt);
      slot ^= 1;
    }
    STAMP(60);

    // ---------------------------------------------------------------- epilogue
    const int row0 = m0 + wm * (C::TM * 32), col0 = n0 + wn * 64;
    const bool wave_cols_ok = col0 < g.N;
    const float bias0 = bias_pre[0], bias1 = bias_pre[1];
    // next tile: addresses, bias and the first K-stage go out now, into slot 0 (free since the last barrier)
    const int next = tile + G;
    if (next < ntiles) {
      set_tile(next);
      bias_pre[0] = g.bias[n0 + wn * 64 + l31]; bias_pre[1] = g.bias[n0 + wn * 64 + 32 + l31];
      stage(0, 0);
    }
    char* stg = smem + C::STAGE + wave * C::STG;     // wave-private staging inside slot 1
    if (EPI == EPI_BF16 || EPI == EPI_FC1) {   // EPI_FC1 here = bias + erf-GELU, no scoring (evaluation passes)
      constexpr int PASSES = C::TM * 32 / C::ROWS16, APP = C::ROWS16 / 32;   // 32-row sub-tiles per pass
#pragma unroll
      for (int h = 0; h < PASSES; ++h) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const float bias = b ? bias1 : bias0;
          char* const stc = stg + (b * 32 + l31) * 2;
#pragma unroll
          for (int a2 = 0; a2 < APP; ++a2)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
              const int a = h * APP + a2;
              const int rw = a2 * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
              uint32_t pk = pack_bf16x2(acc[a][b][i] + bias, acc[a][b][i + 1] + bias);
              if (EPI == EPI_FC1) {
                f32x2 pre; pre.x = bf16lo_f32(pk); pre.y = bf16hi_f32(pk);
                const f32x2 gl = gelu_erf_fast2(pre);
                pk = pack_bf16x2(gl.x, gl.y);
              }
              const bf16x2 ov = __builtin_bit_cast(bf16x2, pk);
              *(bf16*)(stc + rw * 128) = ov[0];
              *(bf16*)(stc + (rw + 1) * 128) = ov[1];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (wave_cols_ok) {
#pragma unroll
          for (int it = 0; it < C::ROWS16 / 8; ++it) {
            const int r = it * 8 + (lane >> 3), c = (lane & 7) * 8;
            const bf16x8 v = *(const bf16x8*)(stg + r * 128 + c * 2);
            const int m = row0 + h * C::ROWS16 + r;
            if (m < g.M) *(bf16x8*)(g.out + (size_t)m * g.ldo + col0 + c) = v;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    } else {   // EPI_RESID: fp32 staging of ROWS32 rows x 64 cols per pass, then float4 read-modify-write of x
      const int c = (lane & 15) * 4;
      constexpr int PPA = 32 / C::ROWS32;            // passes per 32-row sub-tile (1 or 2)
      constexpr int IT = C::ROWS32 / 4;              // float4 rows handled per lane per pass (8 or 4)
#pragma unroll
      for (int a = 0; a < C::TM; ++a)
#pragma unroll
        for (int hp = 0; hp < PPA; ++hp) {
          const int rbase = row0 + a * 32 + hp * C::ROWS32;
          f32x4 xin[IT];
          float* dst[IT];
          if (wave_cols_ok) {
#pragma unroll
            for (int it = 0; it < IT; ++it) {
              const int m = rbase + it * 4 + (lane >> 4);
              const int mc = m < g.M ? m : g.M - 1;
              dst[it] = g.x + (size_t)mc * g.ldx + col0 + c;
              xin[it] = *(const f32x4*)dst[it];
            }
          }
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = hp * (16 / PPA); i < (hp + 1) * (16 / PPA); ++i) {
              const int rw = (i & 3) + 8 * (i >> 2) + 4 * lh - hp * C::ROWS32;     // row inside this pass
              *(float*)(stg + rw * 256 + (b * 32 + l31) * 4) = bf16_round(acc[a][b][i] + (b ? bias1 : bias0));
            }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (wave_cols_ok) {
#pragma unroll
            for (int it = 0; it < IT; ++it) {
              const int r = it * 4 + (lane >> 4);
              const f32x4 v = *(const f32x4*)(stg + r * 256 + c * 4);
              if (rbase + r < g.M) *(f32x4*)dst[it] = xin[it] + v;
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next pass overwrites the staging
        }
    }
  }
  STAMP(59);
  RSTAMP(62);
#ifdef GEMM_STAMPS
  if (tid == 0) g.stamps[(size_t)blockIdx.x * 64 + 63] = tiles_done;
#endif
}
