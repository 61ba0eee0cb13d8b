// bf16 x bf16 -> fp32 MFMA GEMM for gfx950 with the fused epilogues of the ViT forward.
//
//   C[m, n] = sum_k A[m, k] * W[n, k]      A [M, K] row-major (activations, K-contiguous)
//                                          W [Npad, K] row-major (nn.Linear [out, in] as stored)
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x16_bf16 tiles,
// two workgroups per CU (80 KiB LDS each).
//
// Staging: both operands go global -> LDS by 16-byte LDS-DMA (global_load_lds, no VGPR round trip).  The LDS
// image is lane-linear, so the bank swizzle chunk ^= (row>>1)&7 sits on the per-lane SOURCE address and again
// on the ds_read_b128 fragment reads (an involution; conflict-free for ds_read_b128's 16-lane groups).
// The ring is asymmetric: the A panel streams from HBM / Infinity Cache (long latency) and is kept THREE tiles
// deep (prefetch distance 2), the weight panel is re-read by every row tile, is L2-resident, and is kept TWO deep.
// Per iteration a wave issues [B(t+1) x4, A(t+2) x4] and the tile boundary waits with a COUNTED
// s_waitcnt vmcnt(4) (never 0 in the loop) + raw s_barrier, so A(t+2) stays in flight across the barrier.
//
// Epilogues (compile time).  Results leave through a wave-private LDS staging tile so that every global access
// is a full 16-byte-per-lane, row-contiguous vector access (1 KiB per wave instruction):
//   EPI_BF16   out = bf16(acc + bias)                                   QKV projection
//   EPI_RESID  x  += float(bf16(acc + bias))                            attention out-proj, fc2 (+residual)
//   EPI_FC1    pre = bf16(acc + bias); out = bf16(gelu(pre));           fc1 + GELU + fused stage-1 score:
//              per-(sample, neuron) partial sum of squares over this tile's tokens -> slab (2 segments/tile)
//   EPI_PATCH  x[img*N + 1 + p] = float(bf16(acc + bias)) + pos[1 + p]  patch-embed conv as GEMM
//   EPI_F32    out_f32 = float(bf16(acc + bias))                        classifier head (tiny, scalar stores)
#pragma once
#include "common.hip.h"

enum { EPI_BF16 = 0, EPI_RESID = 1, EPI_FC1 = 2, EPI_PATCH = 3, EPI_F32 = 4 };

#define GEMM_BM 128
#define GEMM_BN 128
#define GEMM_BK 64
#define GEMM_A_STAGES 3
#define GEMM_B_STAGES 2
#define GEMM_STAGE_BYTES 16384
#define GEMM_LDS_BYTES ((GEMM_A_STAGES + GEMM_B_STAGES) * GEMM_STAGE_BYTES)   // 81920

struct GemmArgs {
  const bf16* A; int lda;
  const bf16* W; int ldw;
  const float* bias;          // [Npad], values already bf16-rounded
  int M, N, K;                // N = number of columns to store (<= Npad, multiple of 64 unless EPI_F32)
  int tiles_m, tiles_n;
  bf16* out; int ldo;         // EPI_BF16 / EPI_FC1
  bf16* out2;                 // EPI_FC1: optional pre-GELU copy (same ld)
  float* x; int ldx;          // EPI_RESID / EPI_PATCH / EPI_F32
  const float* xin;           // EPI_RESID: where the residual is READ (same shape / ld as x); nullptr = x itself (in place)
  // EPI_FC1
  int score_site;             // 0 none, 1 pre-GELU, 2 post-GELU
  int tokens;                 // tokens per sample (>= GEMM_BM when score_site != 0)
  int group, mpad, n_img;     // slab layout of the rows (RowMap); group == 0: contiguous
  float* slab; int slab_ld;   // [tiles_m][2][slab_ld]
  // EPI_PATCH
  const float* pos; int patches;
  int group_m;                // row tiles per L2 super-tile (0 = plain N-fastest order)
  int nt_out;                 // gemm256 kernel: bf16 / e4m3 activation outputs leave as NON-TEMPORAL stores (SSP2_OPT_NT_STORES)
  int reverse;                // gemm256: walk the tiles from the last to the first (see engine.hip, zigzag)
  // fp8 (e4m3) operands, gemm256 kernel with F8 = true only: A and W point at BYTES (lda / ldw / K count fp8 elements,
  // K a multiple of 128), acc is multiplied by wscale[n] (the weight row's dequantisation scale) before the bias;
  // EPI_FC1 then writes e4m3 bytes to `out` (ldo in bytes) for the fp8 fc2 that follows
  const float* wscale;
  const float* ascale;        // fp8: per-ROW dequantisation scale of the A operand ([M], written by the LayerNorm that produced it) or nullptr (= 1)
  float ascale_const;         // fp8, ascale == nullptr: one dequantisation scale for the whole A operand (0 = 1; the e4m3 attention output: 1 / 16)
  // LayerNorm fused behind the residual epilogue (gemm256 kernel, EPI_RESID with SCORE = N / 256 only): the workgroup that
  // writes the LAST of the N / 256 column tiles of a 256-row panel of x normalises those rows (gamma / beta fp32 [N], eps) into
  // ln_out (bf16, ld ln_ld elements) or, fp8 mode, ln_out8 (e4m3 bytes, ld ln_ld bytes) — the operand of the next projection
  const float* ln_g; const float* ln_b; bf16* ln_out; uint8_t* ln_out8; int ln_ld; float ln_eps;
  float* ln_ascale;           // with ln_out8: the rows' activation scales ([M]) the fused phase writes beside the e4m3 bytes
  // ... scheduling state of that form (see gemm256.hip.h, LNV): [0..15] two sets of 8 per-XCD tile-queue counters (a launch
  // uses set ln_set and zeroes the other one for the next fused launch), [16 + p] arrival counter of row panel p (left at 0)
  unsigned int* ln_sync; int ln_set;
  // Deferred residual (gemm256 kernel, EPI_RESID with SCORE = 1): per workgroup 256 x 256 bf16 (128 KiB) where a tile's
  // bf16(acc + bias) is parked until the NEXT tile's main loop adds it to x (see gemm256.hip.h, DG); [gridDim.x][8 waves][16 KiB]
  unsigned int* dg;
#ifdef GEMM_STAMPS
  unsigned long long* stamps; // diagnostic build only: [blocks][64] s_memtime values of wave 0
#endif
};

// erf-GELU with erfc(|z|) from Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the bf16 output
// resolution of 2^-9 relative): 2 transcendentals + ~10 VALU ops instead of the ~40-op libm erff.
// The negative branch uses erfc directly (x * 0.5 * erfc(|z|)), so there is no 1 - (1 - q) cancellation.
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  p *= t;
  const float q = p * __builtin_amdgcn_exp2f(z * z * -1.44269504088896340736f);   // erfc(z)
  const float h = 0.5f * q;
  return x * (x >= 0.f ? 1.0f - h : h);
}

// Two elements at a time: the polynomial runs on v_pk_*_f32 (2 fp32 lanes per instruction), only the two
// transcendentals per element stay scalar.  Instruction diet (19 VALU ops per pair instead of 24):
//   * u = |x| * sqrt(log2(e)/2), so that exp(-z^2) = exp2(-(u*u)) (|.| and the negation are operand modifiers)
//     and 1 + p*z = 1 + (p / sqrt(log2 e)) * u
//   * the polynomial's coefficients carry the factor 1/2 of Phi = erfc/2 (exact: a power of two)
//   * gelu(x) = max(x, 0) - |x| * h  with  h = erfc(|z|)/2: one fma instead of compare + select + multiply, and one
//     rounding less than x * (1 - h)
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
// four fp32 -> four e4m3 bytes (OCP fp8, RNE), saturating at +-448 (e4m3fn has no infinity: an overflow would become NaN)
__device__ __forceinline__ uint32_t pack_e4m3x4(float a, float b, float c, float d) {
  a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
  c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f); d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (uint32_t)w;
}
// max(x, 0) as ONE instruction (fmaxf costs two: hipcc canonicalises the operand with v_max_f32 x, x first)
__device__ __forceinline__ float relu_f32(float x) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}
// x and ax = |x| are passed separately: on the fc1 paths both come straight out of the packed bf16 pair
__device__ __forceinline__ f32x2 gelu_erf_core(f32x2 x, f32x2 ax) {
  const f32x2 u = ax * 0.84932180028801904272f;
  f32x2 d = u * 0.27273748087922250f + 1.0f, t;               // 0.3275911 / sqrt(log2 e)
  t.x = __builtin_amdgcn_rcpf(d.x); t.y = __builtin_amdgcn_rcpf(d.y);
  f32x2 p = t * 0.5307027145f + -0.7265760135f;               // A&S 7.1.26 coefficients, halved
  p = t * p + 0.7107068705f;
  p = t * p + -0.142248368f;
  p = t * p + 0.127414796f;
  p *= t;
  const f32x2 a = u * u;
  f32x2 e;
  e.x = __builtin_amdgcn_exp2f(-a.x); e.y = __builtin_amdgcn_exp2f(-a.y);
  const f32x2 h = p * e;                                      // erfc(|z|) / 2
  f32x2 rl;
  rl.x = relu_f32(x.x); rl.y = relu_f32(x.y);
  return -ax * h + rl;                                        // one v_pk_fma_f32 (neg modifier)
}
__device__ __forceinline__ f32x2 gelu_erf_fast2(f32x2 x) {
  f32x2 ax; ax.x = fabsf(x.x); ax.y = fabsf(x.y);
  return gelu_erf_core(x, ax);
}
// GELU of a packed bf16 pair (lo | hi << 16): four bit operations unpack the two values AND their magnitudes
__device__ __forceinline__ f32x2 gelu_erf_pk(uint32_t pk, f32x2& pre) {
  const uint32_t lo = pk << 16;
  pre.x = __builtin_bit_cast(float, lo); pre.y = __builtin_bit_cast(float, pk & 0xffff0000u);
  f32x2 ax;
  ax.x = __builtin_bit_cast(float, lo & 0x7fffffffu); ax.y = __builtin_bit_cast(float, pk & 0x7fff0000u);
  return gelu_erf_core(pre, ax);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  f32x2 f; f.x = lo; f.y = hi;
  const bf16x2 v = __builtin_convertvector(f, bf16x2);     // one v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16lo_f32(uint32_t pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float bf16hi_f32(uint32_t pk) { return __builtin_bit_cast(float, pk & 0xffff0000u); }

// Store a wave's 64x64 bf16 tile from its private LDS staging area: 8 x (ds_read_b128 + 16-B global store).
__device__ __forceinline__ void wave_store_bf16_tile(const char* stg, bf16* out, int ldo, int row0, int col0, int M, int lane) {
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int r = it * 8 + (lane >> 3), c = (lane & 7) * 8;
    const bf16x8 v = *(const bf16x8*)(stg + r * 128 + c * 2);
    if (row0 + r < M) *(bf16x8*)(out + (size_t)(row0 + r) * ldo + col0 + c) = v;
  }
}

#ifdef GEMM_STAMPS
#define STAMP(slot) do { if (tid == 0 && (slot) < 64) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g.stamps[(size_t)blockIdx.x * 64 + (slot)] = t_; } } while (0)
#else
#define STAMP(slot) do {} while (0)
#endif

template <int EPI, int SCORE = 0>   // SCORE (EPI_FC1 only): 0 none, 1 pre-GELU, 2 post-GELU
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const smA = smem;
  char* const smB = smem + GEMM_A_STAGES * GEMM_STAGE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  // Tile order.  An XCD walks a contiguous range of logical ids (xcd_remap) and runs 64 of them at once
  // (32 CUs x 2 workgroups).  Ids are laid out in groups of `group_m` row tiles x all column tiles, column-major
  // inside the group, so the 64 concurrent tiles form a group_m x (64/group_m) patch: each A panel and each weight
  // panel in flight is shared by 8 tiles through the XCD's 4 MiB L2 instead of being re-fetched from MALL/HBM.
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  int tm, tn;
  if (g.group_m > 0) {
    const int per_group = g.group_m * g.tiles_n;
    const int grp = lid / per_group, r = lid - grp * per_group;
    const int rows = min(g.group_m, g.tiles_m - grp * g.group_m);
    tn = r / rows;
    tm = grp * g.group_m + (r - tn * rows);
  } else {
    tm = lid / g.tiles_n; tn = lid - tm * g.tiles_n;
  }
  const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;

  // ---- staging: 16 LDS-DMA pieces of 1 KiB (8 rows x 128 B) per operand per K-tile, 4+4 per wave
  const bf16* a_src[4];
  const bf16* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave + 4 * i) * 8 + (lane >> 3);
    const int c_src = (lane & 7) ^ ((row >> 1) & 7);
    int gr = m0 + row; gr = gr < g.M ? gr : g.M - 1;          // clamp: rows past M are never stored
    a_src[i] = g.A + (size_t)gr * g.lda + c_src * 8;
    w_src[i] = g.W + (size_t)(n0 + row) * g.ldw + c_src * 8;
  }
  auto stage_a = [&](int slot, int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(a_src[i] + kt * GEMM_BK, smA + slot * GEMM_STAGE_BYTES + (wave + 4 * i) * 1024);
  };
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

  // fragment read offsets (bytes) inside one operand image: row*128 + ((chunk ^ swz) * 16)
  int a_off[2], b_off[2], a_swz[2], b_swz[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = wr * 64 + i * 32 + l31, rb = wc * 64 + i * 32 + l31;
    a_off[i] = ra * 128; a_swz[i] = (ra >> 1) & 7;
    b_off[i] = rb * 128; b_swz[i] = (rb >> 1) & 7;
  }

  // bias for this lane's two output columns: issued first (oldest VMEM op), consumed only in the epilogue
  const float bias_pre[2] = {g.bias[n0 + wc * 64 + l31], g.bias[n0 + wc * 64 + 32 + l31]};

  const int nk = g.K / GEMM_BK;
  // prologue: [B0, A0, A1] ; tile 0 needs B0, A0  -> all but the youngest 4 (A1)
  stage_b(0, 0);
  stage_a(0, 0);
  if (nk > 1) {
    stage_a(1, 1);
    asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }
  int sa = 0, sb = 0;          // ring slots of tile kt
  STAMP(0);
  for (int kt = 0; kt < nk; ++kt) {
    const int sa2 = sa >= 1 ? sa - 1 : 2;          // (kt+2) % 3
    if (kt + 1 < nk) stage_b(sb ^ 1, kt + 1);
    if (kt + 2 < nk) stage_a(sa2, kt + 2);
    const char* As = smA + sa * GEMM_STAGE_BYTES;
    const char* Bs = smB + sb * GEMM_STAGE_BYTES;
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
    STAMP(1 + 3 * kt);
    // tile kt+1 needs A(kt+1) [issued one iteration ago] and B(kt+1): everything but the youngest 4 (A(kt+2)).
    // lgkmcnt(0): this tile's fragment reads have landed before any wave may overwrite the slot by DMA.
#ifdef GEMM_STAMPS
    if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else             asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    STAMP(2 + 3 * kt);
    asm volatile("s_barrier" ::: "memory");
    STAMP(3 + 3 * kt);
#else
    if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else             asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    sa = sa == 2 ? 0 : sa + 1;
    sb ^= 1;
  }

  // ---------------------------------------------------------------- epilogue
  STAMP(60);
  // acc[a][b][i]: row = m0 + wr*64 + a*32 + (i&3) + 8*(i>>2) + 4*lh ; col = n0 + wc*64 + b*32 + l31
  const int row0 = m0 + wr * 64, col0 = n0 + wc * 64;
  const bool wave_cols_ok = col0 < g.N;      // N is a multiple of 64 for every vector epilogue: all-or-nothing

  if (EPI == EPI_F32) {                      // classifier head: M = images, N = classes (any value): scalar path
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int n = col0 + b * 32 + l31;
      const float bias = bias_pre[b];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int m = row0 + a * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
          if (m < g.M && n < g.N) g.x[(size_t)m * g.ldx + n] = bf16_round(acc[a][b][i] + bias);
        }
    }
    return;
  }

  if (EPI == EPI_BF16 || EPI == EPI_FC1) {
    char* stg = smem + wave * 8192;          // wave-private [64][64] bf16
    float ssq[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    // Row coordinates inside this tile's slab (contiguous layout = one slab): ml0 = first row of the tile,
    // rows_s = valid rows of the slab, bnd = first row (inside the tile) that belongs to the NEXT sample.
    int bnd = 1 << 30, row_lim = g.M - m0;       // rows of the tile at or past row_lim never count
    if (EPI == EPI_FC1 && SCORE) {
      int ml0 = m0;
      if (g.group > 0) {
        const int sb = m0 / g.mpad;
        ml0 = m0 - sb * g.mpad;
        const int imgs = min(g.group, g.n_img - sb * g.group);
        row_lim = imgs * g.tokens - ml0;
      }
      bnd = (ml0 / g.tokens + 1) * g.tokens - ml0;
    }
    const bool keep_pre = (EPI == EPI_FC1) && g.out2 != nullptr;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float bias = bias_pre[b];
      char* const stc = stg + (b * 32 + l31) * 2;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        // does this 32-row sub-tile lie wholly inside one sample and inside M?  (wave-uniform)
        const int rb = wr * 64 + a * 32;
        const bool plain = (rb + 32 <= bnd || rb >= bnd) && (rb + 32 <= row_lim);
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const int rw = a * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;   // row inside the wave tile (i even: rw, rw+1)
          uint32_t o;
          if (EPI == EPI_BF16) {
            o = pack_bf16x2(acc[a][b][i] + bias, acc[a][b][i + 1] + bias);
          } else {
            const uint32_t pk = pack_bf16x2(acc[a][b][i] + bias, acc[a][b][i + 1] + bias);
            f32x2 pre;
            const f32x2 gl = gelu_erf_pk(pk, pre);
            o = pack_bf16x2(gl.x, gl.y);
            if (keep_pre) { acc[a][b][i] = pre.x; acc[a][b][i + 1] = pre.y; }   // second staging pass below
            if (SCORE) {
              f32x2 sv = pre;
              if (SCORE == 2) { sv.x = bf16lo_f32(o); sv.y = bf16hi_f32(o); }
              if (plain) {
                s0 = fmaf(sv.x, sv.x, s0); s0 = fmaf(sv.y, sv.y, s0);
              } else {                                   // sub-tile straddles two samples or the end of M
                const int r0 = wr * 64 + rw;
                const float q0 = (r0 < row_lim) ? sv.x * sv.x : 0.f, q1 = (r0 + 1 < row_lim) ? sv.y * sv.y : 0.f;
                if (r0 < bnd) s0 += q0; else s1 += q0;
                if (r0 + 1 < bnd) s0 += q1; else s1 += q1;
              }
            }
          }
          const bf16x2 ov = __builtin_bit_cast(bf16x2, o);
          *(bf16*)(stc + rw * 128) = ov[0];
          *(bf16*)(stc + (rw + 1) * 128) = ov[1];
        }
        if (EPI == EPI_FC1 && SCORE) {
          if (plain && rb >= bnd) { ssq[1][b] += s0; } else { ssq[0][b] += s0; ssq[1][b] += s1; }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // staging writes before the vector reads (compiler fence too)
    if (wave_cols_ok) wave_store_bf16_tile(stg, g.out, g.ldo, row0, col0, g.M, lane);
    if (keep_pre) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            *(bf16*)(stg + (a * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh) * 128 + (b * 32 + l31) * 2) = (bf16)acc[a][b][i];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (wave_cols_ok) wave_store_bf16_tile(stg, g.out2, g.ldo, row0, col0, g.M, lane);
    }
    if (EPI == EPI_FC1 && SCORE) {
      // lane l31 of both halves hold the same column: fold halves, then the two row-waves through LDS
      float* red = (float*)(smem + 65536);   // [wr][seg][128], outside every staging area
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
    return;
  }

  // EPI_RESID / EPI_PATCH: fp32 staging [64][64] per wave, then float4 read-modify-write of x
  {
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
    if (!wave_cols_ok) return;
    const int c = (lane & 15) * 4;
    f32x4 xin[16];
    float* dst[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int m = row0 + it * 4 + (lane >> 4);
      const int mc = m < g.M ? m : g.M - 1;
      if (EPI == EPI_RESID) {
        dst[it] = g.x + (size_t)mc * g.ldx + col0 + c;
        xin[it] = *(const f32x4*)((g.xin ? g.xin : g.x) + (size_t)mc * g.ldx + col0 + c);
      } else {
        const int img = mc / g.patches, p = mc - img * g.patches;
        const RowMap rm{g.patches + 1, g.group, g.mpad};
        dst[it] = g.x + (size_t)(row_of(rm, img) + 1 + p) * g.ldx + col0 + c;
        xin[it] = *(const f32x4*)(g.pos + (size_t)(1 + p) * g.ldx + col0 + c);
      }
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int r = it * 4 + (lane >> 4);
      const f32x4 v = *(const f32x4*)(stg + r * 256 + c * 4);
      if (row0 + r < g.M) *(f32x4*)dst[it] = xin[it] + v;
    }
  }
}
