// bf16 x bf16 -> fp32 MFMA GEMM for gfx950 with the fused epilogues of the ViT forward.
//
//   C[m, n] = sum_k A[m, k] * W[n, k]      A [M, K] row-major (activations, K-contiguous)
//                                          W [Npad, K] row-major (nn.Linear [out, in] as stored)
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x16_bf16 tiles.
// Both operands are staged global -> LDS with 16-byte global_load_lds (LDS-DMA, no VGPR round trip) into
// a double buffer; the LDS image is lane-linear, so the bank swizzle chunk ^= (row>>1)&7 is applied to the
// per-lane SOURCE address and again on the ds_read_b128 fragment reads (an involution; conflict-free for
// the 16-lane groups ds_read_b128 is served in, 128-B rows).
//
// Epilogues (selected at compile time):
//   EPI_BF16   out = bf16(acc + bias)                                   QKV projection
//   EPI_RESID  x  += float(bf16(acc + bias))                            attention out-proj, fc2 (+residual)
//   EPI_FC1    pre = bf16(acc + bias); out = bf16(gelu(pre));           fc1 + GELU + fused stage-1 score:
//              per-(sample, neuron) partial sum of squares over this tile's tokens -> slab (2 segments/tile)
//   EPI_PATCH  x[img*N + 1 + p] = float(bf16(acc + bias)) + pos[1 + p]  patch-embed conv as GEMM
//   EPI_F32    out_f32 = float(bf16(acc + bias))                        classifier head
#pragma once
#include "common.hip.h"

enum { EPI_BF16 = 0, EPI_RESID = 1, EPI_FC1 = 2, EPI_PATCH = 3, EPI_F32 = 4 };

#define GEMM_BM 128
#define GEMM_BN 128
#define GEMM_BK 64

struct GemmArgs {
  const bf16* A; int lda;
  const bf16* W; int ldw;
  const float* bias;          // [Npad], values already bf16-rounded
  int M, N, K;                // N = number of columns to store (<= Npad); K multiple of 64
  int tiles_m, tiles_n;
  bf16* out; int ldo;         // EPI_BF16 / EPI_FC1
  bf16* out2;                 // EPI_FC1: optional pre-GELU copy (same ld)
  float* x; int ldx;          // EPI_RESID / EPI_PATCH / EPI_F32
  // EPI_FC1
  int score_site;             // 0 none, 1 pre-GELU, 2 post-GELU
  int tokens;                 // tokens per sample (>= GEMM_BM when score_site != 0)
  float* slab; int slab_ld;   // [tiles_m][2][slab_ld]
  // EPI_PATCH
  const float* pos; int patches;
};

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) char smem[65536];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = lid / g.tiles_n, tn = lid - tm * g.tiles_n;   // N fastest: neighbours share the A panel
  const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;

  // ---- staging: 16 LDS-DMA pieces of 1 KiB (8 rows x 128 B) per operand per K-tile, 4+4 per wave
  const int srow = lane >> 3;                 // row inside a piece
  const bf16* a_src[4];
  const bf16* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave + 4 * i) * 8 + srow;
    const int c_src = (lane & 7) ^ ((row >> 1) & 7);
    int gr = m0 + row; gr = gr < g.M ? gr : g.M - 1;          // clamp: rows past M are never stored
    a_src[i] = g.A + (size_t)gr * g.lda + c_src * 8;
    w_src[i] = g.W + (size_t)(n0 + row) * g.ldw + c_src * 8;
  }
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * 32768;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(a_src[i] + kt * GEMM_BK, base + (wave + 4 * i) * 1024);
      glds16(w_src[i] + kt * GEMM_BK, base + 16384 + (wave + 4 * i) * 1024);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  // fragment read offsets (bytes) inside one operand image: row*128 + ((chunk ^ swz) * 16)
  int a_off[2], b_off[2], a_swz[2], b_swz[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = wr * 64 + i * 32 + l31, rb = wc * 64 + i * 32 + l31;
    a_off[i] = ra * 128; a_swz[i] = (ra >> 1) & 7;
    b_off[i] = rb * 128; b_swz[i] = (rb >> 1) & 7;
  }

  const int nk = g.K / GEMM_BK;
  stage(0, 0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* As = smem + cur * 32768;
    const char* Bs = As + 16384;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int chunk = 2 * s + lh;
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
    __syncthreads();   // drains the LDS-DMA of tile kt+1 (vmcnt(0)) and fences the reads of tile kt
    cur ^= 1;
  }

  // ---------------------------------------------------------------- epilogue
  // acc[a][b][i]: row = m0 + wr*64 + a*32 + (i&3) + 8*(i>>2) + 4*lh ; col = n0 + wc*64 + b*32 + l31
  float ssq[2][2];             // EPI_FC1: [segment][b]
  int bnd = 1 << 30;           // first row (inside the tile) that belongs to the NEXT sample
  if (EPI == EPI_FC1) {
    ssq[0][0] = ssq[0][1] = ssq[1][0] = ssq[1][1] = 0.f;
    if (g.score_site) bnd = (m0 / g.tokens + 1) * g.tokens - m0;
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int n = n0 + wc * 64 + b * 32 + l31;
    const float bias = g.bias[n];
    const bool n_ok = n < g.N;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rt = wr * 64 + a * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;   // row inside the tile
        const int m = m0 + rt;
        const float v = acc[a][b][i] + bias;
        if (EPI == EPI_BF16) {
          if (m < g.M && n_ok) g.out[(size_t)m * g.ldo + n] = (bf16)v;
        } else if (EPI == EPI_RESID) {
          if (m < g.M && n_ok) { float* p = g.x + (size_t)m * g.ldx + n; *p = *p + bf16_round(v); }
        } else if (EPI == EPI_F32) {
          if (m < g.M && n_ok) g.x[(size_t)m * g.ldx + n] = bf16_round(v);
        } else if (EPI == EPI_PATCH) {
          if (m < g.M && n_ok) {
            const int img = m / g.patches, p = m - img * g.patches;
            g.x[((size_t)img * (g.patches + 1) + 1 + p) * g.ldx + n] =
                bf16_round(v) + g.pos[(size_t)(1 + p) * g.ldx + n];
          }
        } else {  // EPI_FC1
          const float pre = bf16_round(v);
          const bf16 gel = (bf16)gelu_erf(pre);
          if (m < g.M && n_ok) {
            g.out[(size_t)m * g.ldo + n] = gel;
            if (g.out2) g.out2[(size_t)m * g.ldo + n] = (bf16)pre;
          }
          if (g.score_site) {
            const float sv = (g.score_site == 1) ? pre : (float)gel;
            const float sq = (m < g.M) ? sv * sv : 0.f;
            if (rt < bnd) ssq[0][b] += sq; else ssq[1][b] += sq;
          }
        }
      }
    }
  }
  if (EPI == EPI_FC1) {
    if (g.score_site) {
      // lane l31 of both halves hold the same column: fold halves, then the two row-waves through LDS
      float* red = (float*)smem;   // [wr][seg][128]  (main-loop LDS is dead after the last barrier)
#pragma unroll
      for (int sgm = 0; sgm < 2; ++sgm)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          float t = ssq[sgm][b];
          t += __shfl_xor(t, 32);
          if (lh == 0) red[(wr * 2 + sgm) * 128 + wc * 64 + b * 32 + l31] = t;
        }
      __syncthreads();
      const int sgm = tid >> 7, col = tid & 127;
      const float tot = red[(0 * 2 + sgm) * 128 + col] + red[(1 * 2 + sgm) * 128 + col];
      if (n0 + col < g.slab_ld) g.slab[((size_t)tm * 2 + sgm) * g.slab_ld + n0 + col] = tot;
    }
  }
}
