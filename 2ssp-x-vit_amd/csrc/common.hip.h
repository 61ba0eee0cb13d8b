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

// XCD-aware bijective remap of a 1-D grid (blocks b and b+8 share an XCD): each XCD walks a contiguous
// range of logical tile ids, so neighbours that share an operand panel hit the same 4 MiB L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// Row layout of the token matrix.  group == 0: images are contiguous (row of image i = i * tokens).
// group  > 0: the images are laid out in SLABS of `group` images (= one dataloader batch each), every slab padded to
// `mpad` rows (a multiple of 256, so no GEMM tile straddles two slabs).  A sample then sits at the same offset inside
// its slab whichever launch it is part of, which makes the per-tile partial sums of the fused stage-1 score — and so
// the scores — independent of how many batches share a launch.
struct RowMap { int tokens, group, mpad; };
__host__ __device__ __forceinline__ long row_of(const RowMap r, int img) {
  return r.group > 0 ? (long)(img / r.group) * r.mpad + (long)(img % r.group) * r.tokens : (long)img * r.tokens;
}
