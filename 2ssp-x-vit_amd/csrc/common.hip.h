// Shared device helpers for the gfx950 (CDNA4) kernels of libssp2vit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define SSP2_WAVE 64

// fp32 -> bf16 -> fp32 (round-to-nearest-even; lowers to v_cvt_pk_bf16_f32 on gfx950, NaN stays NaN)
__device__ __forceinline__ float bf16_round(float v) { return (float)(bf16)v; }

// async global -> LDS copy of 16 B per lane: LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// the same with the NT (non-temporal) cache policy: a streamed operand takes the L2's evict-first position, so that it
// does not push the panels that every tile re-reads (weights) out of the 4 MiB L2
__device__ __forceinline__ void glds16_nt(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2);
}

// LDS stores that hipcc does not see as LDS stores.  While LDS-DMA operations are in flight the compiler puts an
// `s_waitcnt vmcnt(0)` in front of the next ordinary store to LDS (it cannot tell that the two do not alias): in the GEMM
// epilogues that drained the next tile's first K-tiles before the first staging write of EVERY wave.  The epilogues' staging
// areas are never a DMA target while they are in use, so their stores go out as inline asm; the caller orders them against
// the reads that follow with its own `s_waitcnt lgkmcnt(0)` (asm volatile with a memory clobber).
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ void lds_st_b8(void* p, uint32_t v) { asm volatile("ds_write_b8 %0, %1" :: "v"(lds_addr(p)), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_st_b16(void* p, uint32_t v) { asm volatile("ds_write_b16 %0, %1" :: "v"(lds_addr(p)), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_st_b32(void* p, uint32_t v) { asm volatile("ds_write_b32 %0, %1" :: "v"(lds_addr(p)), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_st_b64(void* p, uint2 v) {
  const unsigned long long d = ((unsigned long long)v.y << 32) | v.x;
  asm volatile("ds_write_b64 %0, %1" :: "v"(lds_addr(p)), "v"(d) : "memory");
}
__device__ __forceinline__ void lds_st_b128(void* p, f32x4 v) { asm volatile("ds_write_b128 %0, %1" :: "v"(lds_addr(p)), "v"(v) : "memory"); }

// exact (erf) GELU in fp32, the nn.GELU() default the reference models use
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// Sum over the 64 lanes, every lane gets the total, on the VALU only: four DPP adds (xor 1, xor 2, the other quad pair of
// the 8-group, the other 8-group of the row) and the two gfx950 row / half swaps — ~50 cycles of dependent latency where
// the ds_bpermute butterfly of wave_sum takes ~600.  Association (fixed): ((pairs) quads) 8-groups, rows 0+1 | 2+3,
// halves.  (The permlane swaps are inline asm: with both operands equal, hipcc 7.2 folds the builtin's two results
// into one and emits v_add v, v, v.)
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));   // row_mirror
  float a = v, b = v;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));      // a: rows 0 0 2 2, b: rows 1 1 3 3
  v = a + b; a = v; b = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));      // a: lower half twice, b: upper half twice
  return a + b;
}

// four fp32 -> four e4m3 bytes (OCP fp8), saturating at +-448
__device__ __forceinline__ uint32_t pack_e4m3x4_from(float a, float b, float c, float d) {
  a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
  c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f); d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (uint32_t)w;
}

// The same, noting in `sat` whether any of the four values lay outside the e4m3 range (|v| > 448: the cast clips it).  The fp8 path's
// unscaled hand-offs (GELU output -> fc2 operand, attention output x 16 -> out-projection operand) count such events per engine
// (ssp2_query SSP2_Q_FP8_SATURATED): a checkpoint whose activations leave the range is REPORTED instead of silently clipped.
__device__ __forceinline__ uint32_t pack_e4m3x4_sat(float a, float b, float c, float d, bool& sat) {
  const float m = fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(d)));
  sat = sat || (m > 448.f);
  return pack_e4m3x4_from(a, b, c, d);
}
// one wave: if any lane saw a clipped value, lane 0 of the wave adds 1 to *counter (a no-op branch otherwise; counter may be null)
__device__ __forceinline__ void report_sat(bool sat, unsigned int* counter) {
  if (counter && __builtin_amdgcn_ballot_w64(sat) != 0 && (threadIdx.x & 63) == 0) atomicAdd(counter, 1u);
}

// One LayerNorm row, held by ONE wave: lane owns the float4 chunks lane, lane + 64, ... of the row (nv = D / 4 chunks,
// chunks past nv are skipped).  Two-pass mean / variance in fp32, every multiply-add spelled out, so that the standalone
// kernel (misc.hip.h) and the LayerNorm phase of the residual GEMM (gemm256.hip.h) produce the same bits from the
// same row.  g4 / b4: the lane's chunks of gamma / beta; inv_d = 1 / D.  Output: bf16 rows (OUT8 = false, y = bf16*) or,
// fp8 mode, the e4m3 bytes of the bf16-rounded values (OUT8 = true, y = uint8_t*).  FULL: D = 256 * MAXV exactly, no
// chunk is predicated (straight-line code for the GEMM's phase).
// The row statistics are wave-uniform values computed on the VALU of every lane, so their cost is per row, not per
// element: sum * (1 / D) instead of a division, v_rsq_f32 + one Newton step instead of 1 / sqrt (both within 1 ulp of the
// correctly rounded forms: 12 + 27 instructions of ~130 per row saved, which matters where the VALU is the bound — the GEMM's phase).
template <int MAXV, bool FULL>
__device__ __forceinline__ void ln_row_stats(const f32x4 (&v)[MAXV], int lane, int nv, float inv_d, float eps, float& mean, float& rstd) {
  // no contraction beyond the fmas written out below: left to hipcc, ONE of the four instances (the GEMM phase's e4m3 form)
  // folded mean = sum * inv_d into the subtractions, v - sum * inv_d as one fma, and a few e4m3 bytes per launch differed
#pragma clang fp contract(off)
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (FULL || i * 64 + lane < nv) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  mean = wave_sum_dpp(s) * inv_d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (FULL || i * 64 + lane < nv) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float d = v[i][k] - mean; q = __builtin_fmaf(d, d, q); }
    }
  const float var = __builtin_fmaf(wave_sum_dpp(q), inv_d, eps);
  rstd = __builtin_amdgcn_rsqf(var);
  rstd = rstd * __builtin_fmaf(-0.5f * var * rstd, rstd, 1.5f);           // one Newton step on v_rsq_f32's 1-ulp estimate
}
// chunk i of the row: ((v - mean) * rstd) * gamma + beta, bf16-rounded, stored as bf16 or as the e4m3 bytes of the bf16 values
template <bool OUT8>
__device__ __forceinline__ void ln_chunk_write(const f32x4& v, float mean, float rstd, const f32x4& g4, const f32x4& b4, void* y, int c) {
#pragma clang fp contract(off)
  bf16x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = (bf16)__builtin_fmaf((v[k] - mean) * rstd, g4[k], b4[k]);
#ifndef LN_NT_STORES
#define LN_NT_STORES 0
#endif
  if constexpr (OUT8) {
    const uint32_t pk = pack_e4m3x4_from((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
    if (LN_NT_STORES) __builtin_nontemporal_store(pk, (uint32_t*)((uint8_t*)y + c * 4)); else *(uint32_t*)((uint8_t*)y + c * 4) = pk;
  } else {
    if (LN_NT_STORES) __builtin_nontemporal_store(o, (bf16x4*)((bf16*)y + c * 4)); else *(bf16x4*)((bf16*)y + c * 4) = o;
  }
}
// Maximum over the 64 lanes on the VALU (the pattern of wave_sum_dpp).
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true)));
  float a = v, b = v;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  v = fmaxf(a, b); a = v; b = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}
// fp8 mode with PER-ROW activation scales (round 3): the row's bf16-rounded LayerNorm values o are written as
// e4m3(o * 448 / amax(|o|)) and *ascale_row = amax / 448 — the projection that reads the row multiplies its accumulators by it
// (gemm256.hip.h, GemmArgs.ascale).  The bare saturating cast of round 2 clipped at +-448 and spent the e4m3 subnormals on every
// value below 2^-6; with the row's own scale the largest element sits at the top of the range whatever gamma / beta a model has.
template <int MAXV, bool FULL>
__device__ __forceinline__ void ln_row_write8_scaled(const f32x4 (&v)[MAXV], int lane, int nv, float mean, float rstd, const f32x4 (&g4)[MAXV],
                                                     const f32x4 (&b4)[MAXV], uint8_t* y8, float* ascale_row) {
#pragma clang fp contract(off)
  f32x4 o[MAXV];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (FULL || i * 64 + lane < nv) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        o[i][k] = (float)(bf16)__builtin_fmaf((v[i][k] - mean) * rstd, g4[i][k], b4[i][k]);
        amax = fmaxf(amax, fabsf(o[i][k]));
      }
    }
  amax = wave_max_dpp(amax);
  const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
  const float inv = amax > 0.f ? 448.0f / amax : 1.0f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (FULL || i * 64 + lane < nv)
      *(uint32_t*)(y8 + (i * 64 + lane) * 4) = pack_e4m3x4_from(o[i][0] * inv, o[i][1] * inv, o[i][2] * inv, o[i][3] * inv);
  if (lane == 0) *ascale_row = sc;
}
template <int MAXV, bool FULL, bool OUT8>
__device__ __forceinline__ void ln_row_finish(const f32x4 (&v)[MAXV], int lane, int nv, float inv_d, float eps, const f32x4 (&g4)[MAXV],
                                              const f32x4 (&b4)[MAXV], void* y, float* ascale_row = nullptr) {
  float mean, rstd;
  ln_row_stats<MAXV, FULL>(v, lane, nv, inv_d, eps, mean, rstd);
  if constexpr (OUT8) {
    if (ascale_row) { ln_row_write8_scaled<MAXV, FULL>(v, lane, nv, mean, rstd, g4, b4, (uint8_t*)y, ascale_row); return; }
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (FULL || i * 64 + lane < nv) ln_chunk_write<OUT8>(v[i], mean, rstd, g4[i], b4[i], y, i * 64 + lane);
}

// XCD-aware bijective remap of a 1-D grid (blocks b and b+8 share an XCD): each XCD walks a contiguous
// range of logical tile ids, so neighbours that share an operand panel hit the same 4 MiB L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// Row layout of the token matrix.  group == 0: images are contiguous (row of image i = i * tokens).
// group  > 0: the images are laid out in SLABS of `group` images (= one dataloader batch each), every slab padded to
// `mpad` rows, a multiple of kSlabAlign = 128: the stage-1 hook folds a sample's sum of squares per 128-row tile (a tile of the 128 x 128
// kernel = a wave's tile of the 256 x 256 kernel), so no such tile straddles two slabs and a sample sits at the same offset inside its tile
// grid whichever launch it is part of — which makes the per-tile partial sums, and so the scores, independent of how many batches share a
// launch.  (Rounds 1-5a aligned slabs to 256 rows; only the 128-row scoring tiles need the alignment — a 256-row workgroup tile may span two
// slabs — and with the search's streams in this layout since round 5 the pad rows are work: 64 instead of 192 per 64 images of ViT-B/16.)
static constexpr int kSlabAlign = 128;
struct RowMap { int tokens, group, mpad; };
__host__ __device__ __forceinline__ long row_of(const RowMap r, int img) {
  return r.group > 0 ? (long)(img / r.group) * r.mpad + (long)(img % r.group) * r.tokens : (long)img * r.tokens;
}
