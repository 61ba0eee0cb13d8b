// Multi-head self-attention for ViT sequence lengths (197 / 257 tokens): one workgroup per (image, head),
// the whole K and V of that head live in LDS, the whole score row of a query lives in registers, so the
// softmax is exact single-pass (no online rescaling) and nothing N x N ever touches HBM.
//
// MFMA orientation (v_mfma_f32_32x32x16_bf16, 64-wide waves):
//   S^T tile [32 keys x 32 queries] = K_tile (A operand, ds_read_b128 from LDS)  x  Q^T (B operand, registers)
//     -> accumulator: lane <-> query, registers <-> keys: row max / row sum are in-lane reductions plus one
//        exchange between the two 32-lane halves.
//   O^T tile [32 d_h x 32 queries] = V^T (A operand) x P^T (B operand).  P^T is the S^T accumulator itself,
//     converted to bf16 in registers (an accumulator tile is a valid B operand when the next product sums over
//     its ROW index); V^T fragments come straight from the row-major V image with ds_read_b64_tr_b16
//     (hardware transpose read), so V is staged exactly like K — no transposed LDS writes.
//
// qkv layout: [n*tokens, 3*dim] bf16, columns [q | k | v][head][d_h] (timm's fused qkv == concat of HF q,k,v).
#pragma once
#include <type_traits>
#include "common.hip.h"

#define ATTN_OUT8_SCALE 16.0f        // fp8 mode, default: attention outputs are written as e4m3(o * 16) and the out-projection multiplies by 1 / 16;
                                     // a calibrated engine passes its own power of two per layer (kernel argument out8_scale, ssp2_fp8_calibrate_*)

// ------------------------------------------------------------------------------------------------------------------
// Softmax numerators of one 32-query tile, in place, shared by every attention kernel of this file (so they stay bit-identical
// to one another): sacc[kt][i] := exp2((s - rowmax) * scale * log2 e), returns the row sum (lane <-> query; registers and the two
// lane halves <-> keys: key = 32 kt + (i & 3) + 8 (i >> 2) + 4 lh).  scale > 0, so the row max is taken on the raw scores and the
// scale folds into the exponent.  Round 4 (the softmax is the kernels' bound: ~4 issue slots per v_exp_f32 — a quarter-rate
// transcendental — and, before, one v_fma + one v_add + half a v_max3 per score):
//   * exponent arguments and the row sum on PACKED instructions (v_pk_fma_f32 / v_pk_add_f32: two scores per issue slot); the sum
//     is therefore formed as two interleaved partial sums — a different association than rounds 1-3 (every kernel changed with it);
//   * (ATTN_SKIP_DEAD, built, measured, OFF) the LAST key tile is mostly padding (197 tokens: 5 valid keys of 32; 257 tokens: 1 of
//     32): register groups of four whose keys are past `tokens` on every lane need no exponential (they are 0, exactly what
//     exp2(-inf) gives) — a wave-uniform branch per group of the last tile, and `live_groups` (1..4) lets the P V loop skip the dead
//     half.  Same bits, and 3-5 % SLOWER on the 197-token kernel (tools/attn_bench, 3520 items: 1043-1060 us against 1000-1022 with
//     the padding computed; profiles/r04_g_attn_ab.txt): the branches cut the one basic block in which hipcc interleaves the
//     exponentials with the MFMAs of the next phase.  Packing alone: 1000 us against 1008-1022 (+1-2 %).
// Cache policy A/B switches: K / V are read exactly once per launch (ATTN_NT_KV: their LDS-DMA marked non-temporal), the output
// is read next by another kernel (ATTN_NT_OUT: non-temporal stores)
#ifndef ATTN_NT_KV
#define ATTN_NT_KV 0
#endif
#ifndef ATTN_NT_OUT
#define ATTN_NT_OUT 0
#endif
#if ATTN_NT_KV
#define GLDS_KV glds16_nt
#else
#define GLDS_KV glds16
#endif
// A/B switches (timing builds of tools/attn_bench only; the product builds with both on)
#ifndef ATTN_SOFTMAX_PACKED
#define ATTN_SOFTMAX_PACKED 1
#endif
#ifndef ATTN_SKIP_DEAD
#define ATTN_SKIP_DEAD 0
#endif
typedef __attribute__((ext_vector_type(2))) float attn_f32x2;
// LIVE (round 5): the number of live register groups of the last key tile as a COMPILE-TIME constant (1..3; 0 = not known: all four
// groups computed, or ATTN_SKIP_DEAD's run-time count in the tools).  The engine knows `tokens` when it launches, so the persistent
// kernels of the two hot geometries (197 tokens: 5 valid keys in the last tile; 257: 1) are instantiated with LIVE = 1: the dead
// groups' maxima, exponentials and sums and the dead half's P V step are not emitted at all — no branch, one basic block as before.
template <int NT, int LIVE = 0>
__device__ __forceinline__ float softmax_tile(f32x16 (&sacc)[NT], const int tokens, const int lh, const float scale, int& live_groups) {
  const int rem = tokens - 32 * (NT - 1);                  // valid keys of the last tile, 1..32 (wave-uniform)
  live_groups = LIVE > 0 ? LIVE : (ATTN_SKIP_DEAD ? __builtin_amdgcn_readfirstlane((rem + 7) >> 3) : 4);          // group g = registers 4g..4g+3 = keys 8g + 4 lh + 0..3 of the tile
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NT - 1; ++kt)
#pragma unroll
    for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[kt][i]);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (g < live_groups) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = (NT - 1) * 32 + r + 8 * g + 4 * lh;
        if (key >= tokens) sacc[NT - 1][4 * g + r] = -INFINITY;
        mx = fmaxf(mx, sacc[NT - 1][4 * g + r]);
      }
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  const float c2 = scale * 1.44269504088896340736f;
  const float mc = -mx * c2;
  attn_f32x2 c2v, mcv, sum2;
  c2v.x = c2v.y = c2; mcv.x = mcv.y = mc; sum2.x = sum2.y = 0.f;
  auto pair = [&](f32x16& t, int i) __attribute__((always_inline)) {
#if ATTN_SOFTMAX_PACKED
    attn_f32x2 v; v.x = t[i]; v.y = t[i + 1];
    v = v * c2v + mcv;                                     // one v_pk_fma_f32
    attn_f32x2 e; e.x = __builtin_amdgcn_exp2f(v.x); e.y = __builtin_amdgcn_exp2f(v.y);
    t[i] = e.x; t[i + 1] = e.y;
    sum2 += e;                                             // one v_pk_add_f32
#else
    const float e0 = __builtin_amdgcn_exp2f(fmaf(t[i], c2, mc)), e1 = __builtin_amdgcn_exp2f(fmaf(t[i + 1], c2, mc));
    t[i] = e0; t[i + 1] = e1;
    sum2.x += e0; sum2.x += e1;
#endif
  };
#pragma unroll
  for (int kt = 0; kt < NT - 1; ++kt)
#pragma unroll
    for (int i = 0; i < 16; i += 2) pair(sacc[kt], i);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (g < live_groups) { pair(sacc[NT - 1], 4 * g); pair(sacc[NT - 1], 4 * g + 2); }
    else {
#pragma unroll
      for (int r = 0; r < 4; ++r) sacc[NT - 1][4 * g + r] = 0.f;
    }
  }
  float sum = sum2.x + sum2.y;
  sum += __shfl_xor(sum, 32);
  return sum;
}

// CLS_ONLY (evaluation tail): only query 0 of every image is needed.  q then comes from a compact [n, dim] buffer
// (q_img_stride = dim, q_ld = 0: the whole 32-query tile reads the CLS row) and only row 0 is stored, into a compact
// [n, dim] output.  The arithmetic for query 0 is the same instruction sequence as in the full kernel, so the tail
// is bit-identical to running the whole block.
template <int DH, int NT, bool CLS_ONLY = false>
__global__ __launch_bounds__(256, (DH == 64 ? 2 : 1)) void attn_fwd_kernel(const bf16* __restrict__ qkv, int ld, const bf16* __restrict__ qsrc,
                                                      size_t q_img_stride, int q_ld, bf16* __restrict__ out,
                                                      size_t o_img_stride, int ldo, int tokens, int dim, float scale, RowMap rm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int DT = (DH + 31) / 32;     // 32-row tiles of the O^T output (d_h padded up)
  constexpr int KS = DH / 16;            // k-steps of the QK^T product
  // d_h = 64 (Ti/S/B/L): rows are exactly 128 B, so K and V are staged by LDS-DMA (lane-linear image, 8 rows per
  // 1-KiB piece, every piece of the head in flight at once) and bank conflicts are removed by XOR swizzles applied on
  // the DMA's source address and again on the reads:  K chunk ^= (row >> 1) & 7  (ds_read_b128, lane <-> row),
  // V 64-byte half ^= (row >> 1) & 1  (ds_read_b64_tr_b16: 4 rows x 64 B per 32-lane group).
  // d_h = 80 (H/14): the same lane-linear DMA image with 160-byte rows (10 chunks per row, a piece covers 6.4 rows),
  // no swizzle: the K reads of rows r and r + 8 and a quarter of the transposed V reads are 2-way bank conflicts, a
  // small price next to seven serialised global-load round trips of the register-staged fill.  The third 32-column
  // V^T tile reads 32 bytes past its row (columns 80..95): whatever is there only reaches output rows >= 80, which
  // are never stored.  Other head widths keep padded rows filled through registers.
  constexpr bool DMA64 = (DH == 64), DMA80 = (DH == 80 && (NT * 32 * (DH / 8)) % 64 == 0);
  constexpr bool DMA = DMA64 || DMA80;
  constexpr int KSB = DMA ? DH * 2 : DH * 2 + 16;    // K row stride (bytes); padded: conflict-free ds_read_b128
  constexpr int VSB = DMA ? DH * 2 : 192;            // V row stride (bytes); padded: 4 rows x 16 dwords tile 64 banks
  constexpr int NKEY = NT * 32;
  constexpr int CH = DH / 8;
  static_assert(DMA80 || DT * 32 * 2 <= VSB, "V row does not fit its LDS stride");

  char* Ks = smem;
  char* Vs = smem + NKEY * KSB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int head = blockIdx.x, img = blockIdx.y;
  const size_t img_row = (size_t)row_of(rm, img);
  const bf16* base = qkv + img_row * ld + head * DH;

  // Q fragments of the wave's first query tile go out first; the next tile's are fetched behind the current tile's
  // QK^T (software pipeline: no query load is ever waited for at the top of a tile)
  const bf16* qbase = (CLS_ONLY ? qsrc + (size_t)img * q_img_stride : qsrc + img_row * ld) + head * DH;
  auto load_q = [&](int qt, bf16x8 (&dst)[KS]) {
    const int q = qt * 32 + l31;
    const int qc = q < tokens ? q : tokens - 1;
    const bf16* qp = qbase + (size_t)qc * q_ld + 8 * lh;
#pragma unroll
    for (int s = 0; s < KS; ++s) dst[s] = *(const bf16x8*)(qp + 16 * s);
  };
  // ---- stage K, V.  SPLIT (d_h = 64, full kernel): every wave issues Q, then its NT pieces of K, then its NT pieces
  // of V, and waits with a COUNTED vmcnt(NT) — QK^T and the softmax run while V is still in flight; V is waited for
  // (and the second barrier paid) in front of the first P V product.  The Q loads of this flow are inline asm with
  // explicit waits tied to the registers: a compiler-tracked load would get a conservative vmcnt(0) in front of its
  // first use, which would also wait for V.
  constexpr bool SPLIT = DMA64 && !CLS_ONLY && NT >= 4;
  auto load_q_asm = [&](int qt, bf16x8 (&dst)[KS]) {
    const int q = qt * 32 + l31;
    const int qc = q < tokens ? q : tokens - 1;
    const bf16* qp = qbase + (size_t)qc * q_ld + 8 * lh;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      f32x4 t;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(t) : "v"(qp + 16 * s) : "memory");
      dst[s] = __builtin_bit_cast(bf16x8, t);
    }
  };
  auto wait_q = [&](bf16x8 (&dst)[KS], auto cnt) {    // s_waitcnt vmcnt(cnt) that the uses of dst[] cannot move above
    static_assert(KS == 4 || !SPLIT, "wait_q ties four fragments");
    f32x4 t0 = __builtin_bit_cast(f32x4, dst[0]), t1 = __builtin_bit_cast(f32x4, dst[1]);
    f32x4 t2 = __builtin_bit_cast(f32x4, dst[KS > 2 ? 2 : 0]), t3 = __builtin_bit_cast(f32x4, dst[KS > 3 ? 3 : 0]);
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "n"(decltype(cnt)::value) : "memory");
    dst[0] = __builtin_bit_cast(bf16x8, t0); dst[1] = __builtin_bit_cast(bf16x8, t1);
    if (KS > 2) dst[2] = __builtin_bit_cast(bf16x8, t2);
    if (KS > 3) dst[3] = __builtin_bit_cast(bf16x8, t3);
  };
  bf16x8 qnext[KS];
  if constexpr (SPLIT) load_q_asm(wave, qnext);
  else if (wave < (CLS_ONLY ? 1 : NT)) load_q(wave, qnext);
  if constexpr (DMA80) {
    for (int piece = wave; piece < NKEY * CH / 64; piece += 4) {
      const int gch = piece * 64 + lane, row = gch / CH, c = gch - row * CH;
      const int rc = row < tokens ? row : tokens - 1;
      const bf16* rowp = base + (size_t)rc * ld + c * 8;
      glds16(rowp + dim, Ks + piece * 1024);
      glds16(rowp + 2 * dim, Vs + piece * 1024);
    }
  } else if constexpr (DMA64) {
    // rows past `tokens` copy the last valid row: finite values, masked out of the softmax (p = 0) below
    if constexpr (SPLIT) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int piece = wave + 4 * i, row = piece * 8 + (lane >> 3), c = lane & 7;
        const int rc = row < tokens ? row : tokens - 1;
        glds16(base + (size_t)rc * ld + dim + ((c ^ ((row >> 1) & 7)) << 3), Ks + piece * 1024);
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int piece = wave + 4 * i, row = piece * 8 + (lane >> 3), c = lane & 7;
        const int rc = row < tokens ? row : tokens - 1;
        glds16(base + (size_t)rc * ld + 2 * dim + ((c ^ (((row >> 1) & 1) << 2)) << 3), Vs + piece * 1024);
      }
    } else
    for (int piece = wave; piece < NKEY / 8; piece += 4) {
      const int row = piece * 8 + (lane >> 3), c = lane & 7;
      const int rc = row < tokens ? row : tokens - 1;
      const bf16* rowp = base + (size_t)rc * ld;
      glds16(rowp + dim + ((c ^ ((row >> 1) & 7)) << 3), Ks + piece * 1024);
      glds16(rowp + 2 * dim + ((c ^ (((row >> 1) & 1) << 2)) << 3), Vs + piece * 1024);
    }
  } else
  // (zero rows past `tokens`, zero V columns past d_h: 0 * garbage would be NaN-unsafe)
  for (int idx = tid; idx < NKEY * CH; idx += 256) {
    const int key = idx / CH, c = idx - key * CH;
    bf16x8 kv, vv;
#pragma unroll
    for (int j = 0; j < 8; ++j) { kv[j] = (bf16)0.f; vv[j] = (bf16)0.f; }
    if (key < tokens) {
      const bf16* rowp = base + (size_t)key * ld + c * 8;
      kv = *(const bf16x8*)(rowp + dim);
      vv = *(const bf16x8*)(rowp + 2 * dim);
    }
    *(bf16x8*)(Ks + key * KSB + c * 16) = kv;
    *(bf16x8*)(Vs + key * VSB + c * 16) = vv;
  }
  if (!DMA && DT * 32 > DH) {
    constexpr int PC = (DT * 32 - DH) / 8;
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
    for (int idx = tid; idx < NKEY * PC; idx += 256) {
      const int key = idx / PC, c = idx - key * PC;
      *(bf16x8*)(Vs + key * VSB + (CH + c) * 16) = z;
    }
  }
  if constexpr (SPLIT) {
    wait_q(qnext, std::integral_constant<int, NT>{});               // Q and K have landed, V may still be in flight
    __builtin_amdgcn_s_waitcnt(0x0F70 | NT);                        // the same wait, visible to the compiler's bookkeeping
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  } else {
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // per-lane constants of the transposed V read: 16-lane group -> 16 d_h columns, lane 4q+p -> row q, cols 4p..
  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g = (lane >> 4) & 1;
  const int v_lane_off = (4 * lh + tr_q) * VSB + (16 * tr_g + 4 * tr_p) * 2;
  const int v_sw = DMA64 ? ((tr_q >> 1) & 1) : 0;        // swizzle bit of this lane's V rows (row bit 1 == tr_q bit 1)
  const int k_sw = DMA64 ? ((l31 >> 1) & 7) : 0;

  // One 32-query tile.  FIRST (SPLIT flow only): the wave's first tile, peeled out of the loop below so that the wait
  // for V and its barrier are straight-line code — the compiler's own wait bookkeeping then knows the LDS-DMA has
  // drained and adds nothing in later tiles (a wait it places there also waits for the previous tile's output stores).
  auto tile = [&](const int qt, auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    const int q = qt * 32 + l31;
    bf16x8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = qnext[s];
    if constexpr (SPLIT) { if (qt + 4 < NT) load_q_asm(qt + 4, qnext); }
    else if (!CLS_ONLY && qt + 4 < NT) load_q(qt + 4, qnext);

    f32x16 sacc[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) sacc[kt][i] = 0.f;
      const char* kp = Ks + (kt * 32 + l31) * KSB + (DMA64 ? 0 : 16 * lh);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bf16x8 kf = *(const bf16x8*)(kp + (DMA64 ? (((2 * s + lh) ^ k_sw) << 4) : 32 * s));
        sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc[kt], 0, 0, 0);
      }
    }
    // ---- softmax over keys (registers + the other lane half): softmax_tile above
    int live_groups;
    float sum = softmax_tile<NT>(sacc, tokens, lh, scale, live_groups);
    const float inv = 1.0f / sum;

    // ---- O^T = V^T P^T
    if constexpr (FIRST)                 // NT >= 4: all four waves get here exactly once.  Tied to the row sum: the
      // wait must not be scheduled above the softmax (a memory clobber alone orders only the LDS reads)
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" : "+v"(sum) :: "memory");
    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) oacc[dt][i] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        if (kt == NT - 1 && s2 == 1 && live_groups <= 2) continue;       // keys 16..31 of the last tile are all padding: P = 0 (wave-uniform)
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (bf16)sacc[kt][8 * s2 + j];
        const char* vp = Vs + (kt * 32 + 16 * s2) * VSB + v_lane_off;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (__attribute__((address_space(3))) bf16x4*)(vp + (dt ^ v_sw) * 64));
          const bf16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (__attribute__((address_space(3))) bf16x4*)(vp + (dt ^ v_sw) * 64 + 8 * VSB));
          bf16x8 vf;
#pragma unroll
          for (int j = 0; j < 4; ++j) { vf[j] = v1[j]; vf[4 + j] = v2[j]; }
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[dt], 0, 0, 0);
        }
      }
    }
    // ---- store
    if constexpr (SPLIT) {               // next tile's Q (issued a whole tile ago); never a wait for the previous stores
      if (qt + 4 < NT) wait_q(qnext, std::integral_constant<int, 0>{});
    }
    if constexpr (DMA64 && !CLS_ONLY) {
      // through a wave-private [32 queries][128 B] LDS tile (chunk ^= row & 7), so that every global store is a whole
      // 128-byte row segment of 8 lanes x 16 B instead of 8 bytes per lane at a row stride
      char* ost = smem + 2 * NKEY * 128 + wave * 4096;
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
      for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + (lane >> 3);
        ov[it] = *(const bf16x8*)(ost + r * 128 + (((lane & 7) ^ (r & 7)) << 4));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int qq = qt * 32 + it * 8 + (lane >> 3);
        if (qq < tokens) *(bf16x8*)(out + img_row * ldo + (size_t)qq * ldo + head * DH + (lane & 7) * 8) = ov[it];
      }
    } else
    // lane <-> query row, 4 consecutive d_h per 8-byte store
    if (CLS_ONLY ? (q == 0) : (q < tokens)) {
      bf16* op = (CLS_ONLY ? out + (size_t)img * o_img_stride : out + img_row * ldo) + (size_t)q * ldo + head * DH;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int d0 = dt * 32 + 8 * g4 + 4 * lh;
          if (d0 < DH) {
            bf16x4 o4;
#pragma unroll
            for (int j = 0; j < 4; ++j) o4[j] = (bf16)(oacc[dt][4 * g4 + j] * inv);
            *(bf16x4*)(op + d0) = o4;
          }
        }
    }
  };
  if constexpr (SPLIT) {
    tile(wave, std::true_type{});
    for (int qt = wave + 4; qt < NT; qt += 4) tile(qt, std::false_type{});
  } else {
    for (int qt = wave; qt < (CLS_ONLY ? 1 : NT); qt += 4) tile(qt, std::false_type{});   // wave-uniform trip count: EXEC stays full
  }
}

// ------------------------------------------------------------------------------------------------------------------
// d_h = 64, full attention, persistent form: ONE 8-wave workgroup per CU walks the (image, head) items
// `blockIdx.x, blockIdx.x + gridDim.x, ...`.  K and V are double-buffered in LDS (2 x 2 x NT*32 rows x 128 B) and
// wave 7 is the producer: right after the barrier that starts item i it issues the whole LDS-DMA of item i + 1 into
// the other buffer, so the loads of the next item are in flight during ALL of this item's arithmetic (the one-item-per-
// workgroup kernel above can only overlap loads and arithmetic across the two workgroups of a CU).  Waves 0..NT-1 are
// the consumers, one 32-query tile each per item — the 2,2,2,1 tile split of the four-wave kernel is gone — and their
// next Q fragments are fetched a whole item ahead.  One s_barrier per item:
//     producer:  [DMA(i) landed: vmcnt(0)]  B(i)  issue DMA(i+1) -> buf[(i+1)&1]
//     consumer:  [item i-1 finished]        B(i)  tile(i) from buf[i&1]
// buf[(i+1)&1] held item i-1, which every consumer has finished before it arrives at B(i).
// Measured with tools/attn_bench.hip (-DATTN_STAMPS): the item time of a SIMD's two consumer waves is the SUM of their
// MFMA and VALU phases (11 k ticks per item = 2 x 5.2 k), DMA(i+1) has landed ~4 k ticks before the consumers finish
// item i: not HBM-bound.  Three different organisations (this one, the one-item kernel, and a two-pass tile with 14
// waves per CU kept in tools/attn_bench.hip) and a head-major q/k/v layout all land on 6.1-6.9 us per item per CU: the
// common term is the softmax's VALU issue (per 32-query tile ~700 VALU + 112 v_exp, 1.75 tiles per SIMD per item).
// Replacing the barrier with LDS flags (landed / done counters, waves free to drift apart) was tried: the older wave
// of each SIMD then runs a whole item ahead, waits for data gated by the younger one, and the item time is unchanged
// (155.8 vs 158.2 us on 512 images) — not kept.
// The per-tile arithmetic is the instruction-for-instruction order of attn_fwd_kernel<64, NT, false> (same MFMA
// accumulation order, same softmax sum order): the outputs are bit-identical.
#ifdef ATTN_STAMPS
__device__ unsigned long long* attn_stamp_ptr;    // [8 waves][256 slots] of workgroup 0 (tools/attn_bench.hip)
#define ASTAMP(slot) do { if (blockIdx.x == 0 && lane == 0 && (slot) < 256) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); attn_stamp_ptr[wave * 256 + (slot)] = t_; } } while (0)
#else
#define ASTAMP(slot) do {} while (0)
#endif

template <int NT, int LIVE = 0>
__global__ __launch_bounds__(512, 2) void attn64_persist_kernel(const bf16* __restrict__ qkv, int ld, bf16* __restrict__ out, int ldo,
                                                               int tokens, int dim, int heads, int n_items, float scale, RowMap rm,
                                                               int reverse = 0, int stagger = 0, uint8_t* __restrict__ out8 = nullptr, int ldo8 = 0,
                                                               unsigned int* __restrict__ sat_counter = nullptr, float out8_scale = ATTN_OUT8_SCALE) {
  // out8 (fp8 mode, round 3): the output is ALSO the A operand of an e4m3 out-projection — written as e4m3(o * ATTN_OUT8_SCALE)
  // bytes [rows, ldo8] instead of bf16 (a fixed power-of-two scale: an attention output is a convex combination of V rows, O(0.1-1);
  // x 16 puts it into the upper half of the e4m3 range, the projection's epilogue divides it out exactly)
  static_assert(NT >= 1 && NT <= 7, "one consumer wave per query tile, wave 7 produces");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = 4, DT = 2, NKEY = NT * 32, KV = NKEY * 128, BUF = 2 * KV;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int G = gridDim.x;

  // reverse: the items are walked from the last image to the first (engine.hip, zigzag launch order)
  auto row0_of = [&](int it, int& head) -> size_t { if (reverse) it = n_items - 1 - it; const int img = it / heads; head = it - img * heads; return (size_t)row_of(rm, img); };

  // ---- producer side
  auto issue = [&](int it, int b) {
    int head; const size_t r0 = row0_of(it, head);
    const bf16* base = qkv + r0 * ld + head * 64;
    char* Kd = smem + b * BUF;
    char* Vd = Kd + KV;
    const int c = lane & 7;
#pragma unroll 4
    for (int piece = 0; piece < NT * 4; ++piece) {
      const int row = piece * 8 + (lane >> 3);
      const int rc = row < tokens ? row : tokens - 1;
      GLDS_KV(base + (size_t)rc * ld + dim + ((c ^ ((row >> 1) & 7)) << 3), Kd + piece * 1024);
    }
#pragma unroll 4
    for (int piece = 0; piece < NT * 4; ++piece) {
      const int row = piece * 8 + (lane >> 3);
      const int rc = row < tokens ? row : tokens - 1;
      GLDS_KV(base + (size_t)rc * ld + 2 * dim + ((c ^ (((row >> 1) & 1) << 2)) << 3), Vd + piece * 1024);
    }
  };

  // ---- consumer side: Q fragments by inline asm (explicit waits tied to the registers, see wait_q)
  auto load_q_asm = [&](int it, bf16x8 (&dst)[KS]) {
    int head; const size_t r0 = row0_of(it, head);
    const int q = wave * 32 + l31;
    const int qc = q < tokens ? q : tokens - 1;
    const bf16* qp = qkv + (r0 + qc) * ld + head * 64 + 8 * lh;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      f32x4 t;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(t) : "v"(qp + 16 * s) : "memory");
      dst[s] = __builtin_bit_cast(bf16x8, t);
    }
  };
  auto wait_q0 = [&](bf16x8 (&dst)[KS]) {          // s_waitcnt vmcnt(0) that the uses of dst[] cannot move above
    f32x4 t0 = __builtin_bit_cast(f32x4, dst[0]), t1 = __builtin_bit_cast(f32x4, dst[1]);
    f32x4 t2 = __builtin_bit_cast(f32x4, dst[2]), t3 = __builtin_bit_cast(f32x4, dst[3]);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) :: "memory");
    dst[0] = __builtin_bit_cast(bf16x8, t0); dst[1] = __builtin_bit_cast(bf16x8, t1);
    dst[2] = __builtin_bit_cast(bf16x8, t2); dst[3] = __builtin_bit_cast(bf16x8, t3);
  };

  const bool consumer = (wave < NT);
  int it = blockIdx.x;
  // The producer's loop is separate code and returns: no LDS-DMA instruction lies on a consumer's control-flow path, so
  // the compiler's wait bookkeeping never places a vmcnt wait (which would also wait for the consumer's own next-Q
  // loads and previous stores) in front of the consumers' LDS reads.
  if (wave == 7) {
    if (it < n_items) issue(it, 0);
    for (int b = 0, n = 0; it < n_items; it += G, b ^= 1, ++n) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ASTAMP(n * 4 + 0);                                     // DMA(i) landed
      asm volatile("s_barrier" ::: "memory");
      ASTAMP(n * 4 + 1);                                     // B(i) passed
      if (it + G < n_items) issue(it + G, b ^ 1);
      ASTAMP(n * 4 + 2);                                     // DMA(i+1) issued
    }
    return;
  }
  bf16x8 qnext[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) qnext[s][j] = (bf16)0.f;
  if (consumer && it < n_items) { load_q_asm(it, qnext); wait_q0(qnext); }

  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g = (lane >> 4) & 1;
  const int v_lane_off = (4 * lh + tr_q) * 128 + (16 * tr_g + 4 * tr_p) * 2;
  const int v_sw = (tr_q >> 1) & 1;
  const int k_sw = (l31 >> 1) & 7;
  char* const ost = smem + 2 * BUF + (wave < 7 ? wave : 0) * 4096;

  for (int b = 0, n = 0; it < n_items; it += G, b ^= 1, ++n) {     // workgroup-uniform trip count: every wave meets every barrier
    ASTAMP(n * 4 + 0);                                       // arrived at B(i)
    asm volatile("s_barrier" ::: "memory");
    ASTAMP(n * 4 + 1);                                       // B(i) passed
    if (!consumer) continue;
    // STAGGER: waves 4..6 — the SIMD partners of waves 0..2 — start their tile `stagger` x 64 cycles late, so that their
    // Q K^T MFMAs run beside the partner's softmax (VALU) instead of beside the partner's own Q K^T, and their softmax beside the
    // partner's P V: two waves that run the same program from the same barrier otherwise meet on the same pipe in every phase
    // (MI355X_MICROARCH.md "two waves per SIMD", item 9).  The delayed wave's idle time is its partner's uncontended time.
    if (wave >= 4) for (int c = 0; c < stagger; ++c) __builtin_amdgcn_s_sleep(1);

    const char* Ks = smem + b * BUF;
    const char* Vs = Ks + KV;
    bf16x8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = qnext[s];
    const bool more = it + G < n_items;
    if (more) load_q_asm(it + G, qnext);

    f32x16 sacc[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) sacc[kt][i] = 0.f;
      const char* kp = Ks + (kt * 32 + l31) * 128;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bf16x8 kf = *(const bf16x8*)(kp + (((2 * s + lh) ^ k_sw) << 4));
        sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc[kt], 0, 0, 0);
      }
    }
    int live_groups;
    float sum = softmax_tile<NT, LIVE>(sacc, tokens, lh, scale, live_groups);
#ifdef ATTN_STAMPS
    asm volatile("" : "+v"(sum));
    ASTAMP(n * 4 + 2);                                       // softmax sum known
#endif
    const float inv = 1.0f / sum;

    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) oacc[dt][i] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        if (kt == NT - 1 && s2 == 1 && (LIVE > 0 ? LIVE : live_groups) <= 2) continue;       // keys 16..31 of the last tile are all padding: P = 0 (wave-uniform)
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (bf16)sacc[kt][8 * s2 + j];
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
    // next item's Q (issued a whole tile ago) and, older still, the previous item's stores
    if (more) wait_q0(qnext);
#ifdef ATTN_STAMPS
    asm volatile("" : "+v"(oacc[0]), "+v"(oacc[1]));
    ASTAMP(n * 4 + 3);                                       // P V done, Q of the next item here
#endif
    if (out8) {
      // e4m3 bytes: the wave's [32 queries][64 B] tile through its staging area (a lane's four consecutive d_h are one dword), then
      // 128 sixteen-byte row chunks, two per lane
      const float sc8 = inv * out8_scale;
      bool sat8 = false;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          *(uint32_t*)(ost + l31 * 64 + dt * 32 + 8 * g4 + 4 * lh) = pack_e4m3x4_sat(oacc[dt][4 * g4] * sc8, oacc[dt][4 * g4 + 1] * sc8,
                                                                                   oacc[dt][4 * g4 + 2] * sc8, oacc[dt][4 * g4 + 3] * sc8, sat8);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      f32x4 o8[2];
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2) o8[r2] = *(const f32x4*)(ost + (r2 * 64 + lane) * 16);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      int head8; const size_t r08 = row0_of(it, head8);
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2) {
        const int c = r2 * 64 + lane, qq = wave * 32 + (c >> 2);
        if (qq < tokens) *(f32x4*)(out8 + (r08 + qq) * (size_t)ldo8 + head8 * 64 + (c & 3) * 16) = o8[r2];
      }
      report_sat(sat8, sat_counter);          // (padding queries repeat the last token's row: they cannot saturate alone)
      continue;
    }
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
    int head; const size_t r0 = row0_of(it, head);
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int qq = wave * 32 + r4 * 8 + (lane >> 3);
      if (qq < tokens) { if (ATTN_NT_OUT) __builtin_nontemporal_store(ov[r4], (bf16x8*)(out + (r0 + qq) * ldo + head * 64 + (lane & 7) * 8)); else *(bf16x8*)(out + (r0 + qq) * ldo + head * 64 + (lane & 7) * 8) = ov[r4]; }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// d_h = 80 (ViT-H/14: 257 tokens, 9 query tiles), full attention, persistent form.  K + V of one head are 92 KB, so the two
// whole-item buffers of attn64_persist_kernel do not fit the 160 KiB; and nine query tiles do not leave a wave over for a
// producer.  Instead: ONE 8-wave workgroup per CU walks the (image, head) items with a ring of TWO K buffers and ONE V buffer
// (3 x 45 KiB + slack) and every wave is a consumer that also issues its share of the LDS-DMA (6 one-KiB pieces per operand;
// the 45 pieces of an operand are dealt round-robin, three waves re-send a piece to keep the count uniform).  Two barriers per item:
//     B1(i)  everybody has finished P V of item i-1 and its K(i) pieces have landed (vmcnt(0))
//            -> issue [Q fragments of item i+1 | V(i) -> V | K(i+1) -> K[(i+1) & 1]]; S = K Q^T and the softmax of item i run on K[i & 1]
//     B2(i)  behind vmcnt(6): V(i) has landed, K(i+1) may still be in flight -> O = P V from V, output stores
// K[(i+1) & 1] was last read by item i-1 (finished before B2(i-1)), V by P V of item i-1 (finished before B1(i)).  So the next
// item's K and this item's V arrive during arithmetic; the one-item kernel (one 4-wave workgroup per CU at d_h = 80) waits for
// every byte of K and V before its first MFMA.  Waves 0..7 take query tiles 0..7, wave 0 also the ninth (one valid query).
// The per-tile arithmetic is the instruction order of attn_fwd_kernel<80, NT, false> (DMA80 layout: 160-byte rows, unswizzled):
// bit-identical outputs (tests flip SSP2_OPT_ATTN_PERSIST).
template <int NT, int LIVE = 0>
__global__ __launch_bounds__(512) void attn80_persist_kernel(const bf16* __restrict__ qkv, int ld, bf16* __restrict__ out, int ldo,
                                                            int tokens, int dim, int heads, int n_items, float scale, RowMap rm,
                                                            int reverse = 0, int stagger = 0, uint8_t* __restrict__ out8 = nullptr, int ldo8 = 0,
                                                            unsigned int* __restrict__ sat_counter = nullptr, float out8_scale = ATTN_OUT8_SCALE) {
  static_assert(NT == 9, "waves 0..7 take tiles 0..7; the ninth tile (one query) is split over the waves by key tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int DH = 80, KS = 5, DT = 3, CH = 10, NKEY = NT * 32, RSB = 160, KV = NKEY * RSB;     // 46080 B per operand
  constexpr int NP = NKEY * CH / 64;                                                              // 45 one-KiB pieces per operand
  static_assert((NKEY * CH) % 64 == 0, "whole DMA pieces");
  constexpr int PPW = (NP + 7) / 8;                                                               // 6 pieces per wave and operand
  char* const K0 = smem;
  char* const Vb = smem + 2 * KV;                                                                  // + 64 bytes of slack behind V (read past the last row's 160 B)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int G = gridDim.x;
  auto row0_of = [&](int it, int& head) -> size_t { if (reverse) it = n_items - 1 - it; const int img = it / heads; head = it - img * heads; return (size_t)row_of(rm, img); };

  auto issue = [&](int it, int which /* 1 = K, 2 = V */, char* dst) {
    int head; const size_t r0 = row0_of(it, head);
    const bf16* base = qkv + r0 * ld + head * DH + which * dim;
    int ls = lane;
    asm volatile("" : "+v"(ls));                               // opaque: the per-lane source offsets are re-formed per call, not carried across the item loop in registers
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      int piece = wave + 8 * j;
      piece = piece < NP ? piece : piece - 8;                  // the last round re-sends a piece (same bytes to the same place)
      const int gch = piece * 64 + ls, row = gch / CH, c = gch - row * CH;
      const int rc = row < tokens ? row : tokens - 1;
      GLDS_KV(base + (size_t)rc * ld + c * 8, dst + piece * 1024);
    }
  };
  auto load_q_asm = [&](int it, int qt, bf16x8 (&dst)[KS]) {
    int head; const size_t r0 = row0_of(it, head);
    int ls = lane;
    asm volatile("" : "+v"(ls));
    const int q = qt * 32 + (ls & 31);
    const int qc = q < tokens ? q : tokens - 1;
    const bf16* qp = qkv + (r0 + qc) * ld + head * DH + 8 * (ls >> 5);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      f32x4 t;
      asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(t) : "v"(qp), "n"(32 * s) : "memory");
      dst[s] = __builtin_bit_cast(bf16x8, t);
    }
  };

  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g = (lane >> 4) & 1;
  const int v_lane_off = (4 * lh + tr_q) * RSB + (16 * tr_g + 4 * tr_p) * 2;
  // The ninth query tile.  ViT-H/14 has 257 tokens: ONE valid query in it.  Rounds 1-3 gave the whole tile to wave 0 (a second full
  // tile — 45 + 54 MFMAs, 144 exponentials — while seven waves idled at the next barrier: 200 of 1335 ms of the fp8 step).  Round 4
  // (`split9`, taken when the tile holds exactly one query): the tile is split by KEY tiles over all eight waves, flash-attention
  // style — wave w takes the query against key tile w (wave 0 also key tile 8, one valid key) right behind its own tile's stores, while
  // K(i) and V(i) are still in LDS: 5 (10) Q K^T MFMAs, a LOCAL softmax (own maximum m_w, own sum l_w), 6 (9) P V MFMAs; the
  // partials (m_w, l_w, O_w[80]) go to a 3-KiB LDS scratch, and behind the NEXT item's first barrier one wave combines them:
  //   M = max m_w,  a_w = exp2((m_w - M) c2),  o = sum a_w O_w / sum a_w l_w
  // — no extra barrier, no extra register set (the query's fragments come out of LDS: wave 0 sends the row as ONE more DMA piece,
  // oldest of its item, so the counted wait in front of B2 covers it).  All 32 query columns of those MFMAs hold the same query
  // (rows past `tokens` are clamped to the last one), so every lane of a half carries the whole result: no extraction step.
  // Rows 0..255 keep the instruction order of attn_fwd_kernel<80, 9>; row 256 is now summed in another order (local maxima, eight
  // partial sums): equal within the kernel's tolerance, not bit for bit.
  // The engine routes a geometry here only when the ninth tile holds exactly one query (257 tokens); anything else runs attn_fwd_kernel.
  constexpr bool split9 = (NT == 9);
  char* const xq = smem + 3 * KV + 64;                           // 1 KiB: the extra query's row (160 B used), one DMA piece
  float* const xpart = (float*)(xq + 1024);                      // [8 waves][96]: O_w[0..79], m_w at 80, l_w at 81
  // Q fragments: qa = the wave's tile of the CURRENT item, na = of the next item (in flight during this one).
  // 512 threads = 256 registers per lane: 144 score + 48 output accumulators leave room for the two fragment sets.
  bf16x8 qa[KS], na[KS];
  const float c2 = scale * 1.44269504088896340736f;

  auto issue_xq = [&](int it) {                                  // wave 0: the extra query's 160-byte row -> xq (lanes >= 10 re-send chunk 9)
    int head; const size_t r0 = row0_of(it, head);
    int ls = lane;
    asm volatile("" : "+v"(ls));
    const int c = ls < CH ? ls : CH - 1;
    glds16(qkv + (r0 + (size_t)(tokens - 1)) * ld + head * DH + c * 8, xq);
  };
  // the partials of item `it` -> its output row (one wave; lanes 0..39 own two adjacent d each)
  auto combine_x = [&](int it) {
    int head; const size_t r0 = row0_of(it, head);
    int ls = lane;
    asm volatile("" : "+v"(ls));
    float m[8], M = -INFINITY;
#pragma unroll
    for (int w = 0; w < 8; ++w) { m[w] = xpart[w * 96 + 80]; M = fmaxf(M, m[w]); }
    float L = 0.f, o0 = 0.f, o1 = 0.f;
    const int d = 2 * (ls < 40 ? ls : 39);
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const float a = __builtin_amdgcn_exp2f((m[w] - M) * c2);
      L = fmaf(a, xpart[w * 96 + 81], L);
      o0 = fmaf(a, xpart[w * 96 + d], o0);
      o1 = fmaf(a, xpart[w * 96 + d + 1], o1);
    }
    const float inv = 1.0f / L;
    if (ls < 40) {
      const size_t row = r0 + (size_t)(tokens - 1);
      if (out8) {
        bool sat8 = false;
        const uint32_t pk = pack_e4m3x4_sat(o0 * inv * out8_scale, o1 * inv * out8_scale, 0.f, 0.f, sat8);
        if (sat8 && sat_counter) atomicAdd(sat_counter, 1u);
        *(uint16_t*)(out8 + row * (size_t)ldo8 + head * DH + d) = (uint16_t)pk;
      } else {
        bf16x2 o2; o2[0] = (bf16)(o0 * inv); o2[1] = (bf16)(o1 * inv);
        *(bf16x2*)(out + row * ldo + head * DH + d) = o2;
      }
    }
  };

  int it = blockIdx.x;
  if (it < n_items) {
    if (split9 && wave == 0) issue_xq(it);
    issue(it, 1, K0);
    load_q_asm(it, wave, na);
  }
  int prev_it = -1;
  for (int b = 0; it < n_items; it += G, b ^= 1) {            // workgroup-uniform trip count
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                   // B1(i)  (lgkmcnt: the extra query's partials are LDS stores)
    if (split9 && prev_it >= 0) {
      if (wave == 7) combine_x(prev_it);                       // the previous item's extra query (its partials predate B1)
      if (wave == 0) issue_xq(it);                             // this item's: every wave has read the previous row (before B1); OLDEST DMA of the item
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) qa[s] = na[s];
    const bool more = it + G < n_items;
    if (more) load_q_asm(it + G, wave, na);
    issue(it, 2, Vb);
    if (more) issue(it + G, 1, K0 + (b ^ 1) * KV);
    const char* Ks = K0 + b * KV;
    int head; const size_t r0 = row0_of(it, head);
    if (wave >= 4) for (int c = 0; c < stagger; ++c) __builtin_amdgcn_s_sleep(1);      // see attn64_persist_kernel (STAGGER)

    // S^T = K Q^T and the softmax for the wave's tile(s); the score accumulators stay in registers across B2
    f32x16 sacc[NT];
    float inv;
    int live_groups = 4;
    auto scores = [&](const bf16x8 (&qf)[KS]) {
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[kt][i] = 0.f;
        const char* kp = Ks + (kt * 32 + l31) * RSB + 16 * lh;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const bf16x8 kf = *(const bf16x8*)(kp + 32 * s);
          sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc[kt], 0, 0, 0);
        }
      }
      const float sum = softmax_tile<NT, LIVE>(sacc, tokens, lh, scale, live_groups);
      inv = 1.0f / sum;
    };
    auto pv_store = [&](int qt) {
      f32x16 oacc[DT];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[dt][i] = 0.f;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          if (kt == NT - 1 && s2 == 1 && (LIVE > 0 ? LIVE : live_groups) <= 2) continue;     // keys 16..31 of the last tile are all padding: P = 0 (wave-uniform)
          bf16x8 pf;
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[j] = (bf16)sacc[kt][8 * s2 + j];
          const char* vp = Vb + (kt * 32 + 16 * s2) * RSB + v_lane_off;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vp + dt * 64));
            const bf16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vp + dt * 64 + 8 * RSB));
            bf16x8 vf;
#pragma unroll
            for (int j = 0; j < 4; ++j) { vf[j] = v1[j]; vf[4 + j] = v2[j]; }
            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[dt], 0, 0, 0);
          }
        }
      }
      const int q = qt * 32 + l31;
      if (q < tokens) {
        bf16* op = out + (r0 + q) * ldo + head * DH;
        uint8_t* op8 = out8 ? out8 + (r0 + q) * (size_t)ldo8 + head * DH : nullptr;       // fp8 mode: e4m3(o * 16) bytes (see attn64_persist_kernel)
        const float sc8 = inv * out8_scale;
        bool sat8 = false;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int d0 = dt * 32 + 8 * g4 + 4 * lh;
            if (d0 < DH) {
              if (op8) {
                *(uint32_t*)(op8 + d0) = pack_e4m3x4_sat(oacc[dt][4 * g4] * sc8, oacc[dt][4 * g4 + 1] * sc8, oacc[dt][4 * g4 + 2] * sc8, oacc[dt][4 * g4 + 3] * sc8, sat8);
              } else {
                bf16x4 o4;
#pragma unroll
                for (int j = 0; j < 4; ++j) o4[j] = (bf16)(oacc[dt][4 * g4 + j] * inv);
                *(bf16x4*)(op + d0) = o4;
              }
            }
          }
        if (op8) report_sat(sat8, sat_counter);
      }
    };
    // The extra query against this wave's key tile(s): local softmax, partial O -> xpart (see `split9` above)
    auto extra_partial = [&]() {
      bf16x8 qx[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) qx[s] = *(const bf16x8*)(xq + 32 * s + 16 * lh);       // every lane: the same query (broadcast reads)
      f32x16 sx[2];
      const bool second = (wave == 0);                           // wave 0: key tiles 0 and 8
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sx[j][i] = 0.f;
        if (j == 1 && !second) continue;
        const int kt = j ? NT - 1 : wave;
        const char* kp = Ks + (kt * 32 + l31) * RSB + 16 * lh;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const bf16x8 kf = *(const bf16x8*)(kp + 32 * s);
          sx[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qx[s], sx[j], 0, 0, 0);
        }
      }
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sx[0][i]);
      // key tile 8: only key 256 = register 0 of the lower lane half is a token (tokens = 257)
      if (second && lh == 0) mx = fmaxf(mx, sx[1][0]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float mc = -mx * c2;
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float e = __builtin_amdgcn_exp2f(fmaf(sx[0][i], c2, mc)); sx[0][i] = e; sum += e; }
      {
        const float e = (second && lh == 0) ? __builtin_amdgcn_exp2f(fmaf(sx[1][0], c2, mc)) : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) sx[1][i] = 0.f;
        sx[1][0] = e; sum += e;
      }
      sum += __shfl_xor(sum, 32);
      f32x16 ox[DT];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) ox[dt][i] = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (j == 1 && !second) continue;
        const int kt = j ? NT - 1 : wave;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          if (j == 1 && s2 == 1) continue;                       // keys 272.. : padding
          bf16x8 pf;
#pragma unroll
          for (int k = 0; k < 8; ++k) pf[k] = (bf16)sx[j][8 * s2 + k];
          const char* vp = Vb + (kt * 32 + 16 * s2) * RSB + v_lane_off;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vp + dt * 64));
            const bf16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vp + dt * 64 + 8 * RSB));
            bf16x8 vf;
#pragma unroll
            for (int k = 0; k < 4; ++k) { vf[k] = v1[k]; vf[4 + k] = v2[k]; }
            ox[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, ox[dt], 0, 0, 0);
          }
        }
      }
      // query column 0 (lanes 0 and 32) writes: d = dt * 32 + 8 g4 + 4 lh + 0..3
      if (l31 == 0) {
        float* const dst = xpart + wave * 96;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int d0 = dt * 32 + 8 * g4 + 4 * lh;
            if (d0 < DH) {
              f32x4 v; v.x = ox[dt][4 * g4]; v.y = ox[dt][4 * g4 + 1]; v.z = ox[dt][4 * g4 + 2]; v.w = ox[dt][4 * g4 + 3];
              *(f32x4*)(dst + d0) = v;
            }
          }
        if (lh == 0) { dst[80] = mx; dst[81] = sum; }
      }
    };

    scores(qa);
    // B2(i): V(i) has landed on every wave (its 6 pieces are older than the 6 of K(i+1), when there is a next item; wave 0's
    // extra-query piece is older still)
    if (more) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
    else      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    pv_store(wave);
    if (split9) extra_partial();
    prev_it = it;
  }
  if (split9 && prev_it >= 0) {                                // the last item's extra query
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (wave == 7) combine_x(prev_it);
  }
}

