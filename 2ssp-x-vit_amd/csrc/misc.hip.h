// Bandwidth-bound helper kernels: im2col for the patch-embed GEMM, LayerNorm, stage-1 score finishing,
// the standalone activation-L2 kernel, argmax/top-1.
#pragma once
#include "common.hip.h"

// ------------------------------------------------------------------------------------------------
// im2col for conv k=s=p (a pure gather): pixels f32 NCHW [n,3,img,img] -> A bf16 [n*P, Kpad],
// column = c*p*p + ky*p + kx (the flattened conv-weight order), zero-filled up to Kpad.
// One thread per 8 output columns (16-B store); reads are contiguous runs of p floats.
__global__ void im2col_patch_kernel(const float* __restrict__ px, bf16* __restrict__ out, int n, int img, int p,
                                    int side, int K, int Kpad) {
  const int chunks = Kpad / 8;
  const long total = (long)n * side * side * chunks;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(idx % chunks);
    const long row = idx / chunks;
    const int pp = (int)(row % (side * side));
    const int im = (int)(row / (side * side));
    const int py = pp / side, pxx = pp - py * side;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = ch * 8 + j;
      float f = 0.f;
      if (col < K) {
        const int c = col / (p * p), rem = col - c * p * p;
        const int ky = rem / p, kx = rem - ky * p;
        f = px[(((size_t)im * 3 + c) * img + (py * p + ky)) * img + (pxx * p + kx)];
      }
      v[j] = (bf16)f;
    }
    *(bf16x8*)(out + row * Kpad + ch * 8) = v;
  }
}

// x[img*N + 0] = cls + pos[0]   (fp32; the concat promotes to fp32 under autocast)
// What the hardware calls the XCD a workgroup runs on (HW_REG_XCC_ID, bits 3:0) — the fused LayerNorm of gemm256.hip.h keys its
// work queues on it; the engine looks at one grid's worth of answers at create time.
__global__ void xcc_probe_kernel(unsigned int* __restrict__ ids) {
  unsigned int r;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(r));
  if (threadIdx.x == 0) ids[blockIdx.x] = r & 15u;
}

__global__ void cls_row_kernel(float* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos,
                               int n, RowMap rm, int dim) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n * dim) {
    const int im = i / dim, d = i - im * dim;
    x[(size_t)row_of(rm, im) * dim + d] = cls[d] + pos[d];
  }
}

// CLS rows of a residual stream in the slab layout -> compact [n, dim] fp32 (the tail of the search; the contiguous layout uses a 2-D copy)
__global__ void gather_cls_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int n, RowMap rm, int dim) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = dim / 4;
  if (i < (long)n * q) {
    const int im = (int)(i / q), c = (int)(i - (long)im * q);
    ((f32x4*)out)[i] = ((const f32x4*)(x + (size_t)row_of(rm, im) * dim))[c];
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over the last dim: fp32 rows -> bf16 rows.  One wave per row, the row lives in registers
// (two-pass mean / variance in fp32), 16-byte loads and 8-byte stores (lane owns float4 chunks lane, lane+64, ..).
// Row r is read at x + r*in_stride (in_stride = tokens*dim picks the CLS rows for the classifier head).
// D multiple of 4, D <= 256*MAXV.
// FULL: D = 256 * MAXV exactly (ViT-B 768, ViT-L 1024, ViT-H 1280): no chunk is predicated, the row's loads, statistics and
// stores are straight-line code.  Round 2 predicated every chunk (`if (chunk < nv)`); for MAXV = 5 the merges of the
// conditionally loaded row cost 270 register moves and 138 VGPRs — 3 waves per SIMD, 3.1 TB/s on ViT-H/14 against 5.7 for the
// narrower rows (profiles/r03_d_other_models.jsonl).  The results do not depend on FULL (same ln_row_stats / ln_chunk_write).
// Cache policy (A/B switches; see DESIGN.md section 6 for what was measured): the fp32 row is read exactly once here
#ifndef LN_NT_LOADS
#define LN_NT_LOADS 1
#endif
template <int MAXV, bool FULL = false>
__global__ __launch_bounds__(256) void layernorm_bf16_kernel(const float* __restrict__ x, size_t in_stride,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16* __restrict__ y,
                                                            int out_ld, int rows, int D, float eps, RowMap gather,
                                                            uint8_t* __restrict__ y8 = nullptr, int reverse = 0,
                                                            float* __restrict__ ascale = nullptr) {
  const int lane = threadIdx.x & 63;
  const int row = (reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  // gather.tokens > 0: output row r is the CLS row of image r (slab layout aware); else input row r*in_stride
  const f32x4* xr = (const f32x4*)(gather.tokens > 0 ? x + (size_t)row_of(gather, row) * D : x + (size_t)row * in_stride);
  const int nv = D >> 2;                      // float4 chunks in the row
  f32x4 v[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (FULL || i * 64 + lane < nv) v[i] = LN_NT_LOADS ? __builtin_nontemporal_load(xr + i * 64 + lane) : xr[i * 64 + lane];
  float mean, rstd;
  ln_row_stats<MAXV, FULL>(v, lane, nv, 1.0f / (float)D, eps, mean, rstd);
  // gamma / beta only now (L2 hits): held across the statistics they cost 8 registers per chunk and most of the eight waves per
  // SIMD — the kernel lives on memory-level parallelism (84 VGPRs: 60 us for 63040 rows, 32: 50 us).  The lane index their
  // addresses are formed from is made to DEPEND on the statistics: for MAXV = 5 and 8 (ViT-H/14, row widths up to 2048) hipcc
  // otherwise hoists all 2 x MAXV loads above them — 138 / 188 VGPRs, 3 / 2 waves per SIMD, 3.1 TB/s where the narrower
  // instantiations (38-45 VGPRs) run at 5.7 (round 3: profiles/r03_d_other_models.jsonl).
  int lane_gb = lane;
  asm volatile("" : "+v"(lane_gb) : "v"(rstd));
  if (y8 && ascale) {                                 // fp8 mode, per-row activation scale (common.hip.h ln_row_write8_scaled)
    f32x4 g4[MAXV], b4[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (FULL || i * 64 + lane_gb < nv) { g4[i] = ((const f32x4*)gamma)[i * 64 + lane_gb]; b4[i] = ((const f32x4*)beta)[i * 64 + lane_gb]; }
    ln_row_write8_scaled<MAXV, FULL>(v, lane_gb, nv, mean, rstd, g4, b4, y8 + (size_t)row * out_ld, ascale + row);
    return;
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = i * 64 + lane_gb;
    if (FULL || c < nv) {
      const f32x4 g4 = ((const f32x4*)gamma)[c], b4 = ((const f32x4*)beta)[c];
      // fp8 mode: e4m3 bytes (of the bf16-rounded value), out_ld in bytes
      if (y8) ln_chunk_write<true>(v[i], mean, rstd, g4, b4, y8 + (size_t)row * out_ld, c);
      else    ln_chunk_write<false>(v[i], mean, rstd, g4, b4, y + (size_t)row * out_ld, c);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Stage-1 score, fused path, step 2: per-(sample, neuron) norm from the fc1 epilogue's slab.
//   slab [tiles_m][2][ld]: tile t covers rows [128t, 128t+128); segment 0 = rows of sample floor(128t/N),
//   segment 1 = rows of the following sample.  norms[s][j] = sqrt(sum of the segments that belong to s).
__global__ void score_norms_from_slab_kernel(const float* __restrict__ slab, float* __restrict__ norms, int n,
                                             RowMap rm, int ld, int chain) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int s = blockIdx.y;
  if (j >= ld) return;
  const int tokens = rm.tokens;
  const long r0 = row_of(rm, s);                                  // first row of the sample
  const long slab0 = rm.group > 0 ? (long)(s / rm.group) * rm.mpad : 0;   // first row of its slab (multiple of 128)
  const int sl = rm.group > 0 ? s % rm.group : s;                 // index of the sample inside its slab
  const int t0 = (int)(r0 / 128), t1 = (int)((r0 + tokens - 1) / 128);
  float acc = 0.f;
  for (int t = t0; t <= t1; ++t) {
    const int first = (int)(((long)t * 128 - slab0) / tokens);    // slab-local sample that owns the tile's first row
    const int seg = (first == sl) ? 0 : 1;
    acc += slab[((size_t)t * 2 + seg) * ld + j];
  }
  float nrm = sqrtf(acc);
  if (chain) nrm = bf16_round(nrm);
  norms[(size_t)s * ld + j] = nrm;
}

// step 3 (both paths): out[g][j] = sum over the samples s of group g (group*g <= s < group*(g+1), s < n) of
// norms[s][j], samples in index order (fixed association).  grid (ld/256, groups).
__global__ void score_colsum_kernel(const float* __restrict__ norms, float* __restrict__ out, size_t out_stride, int n,
                                    int group, int ld, int chain) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ld) return;
  const int s0 = blockIdx.y * group;
  const int s1 = min(n, s0 + group);
  float acc = 0.f;
  int s = s0;
  for (; s + 16 <= s1; s += 16) {         // 16 loads in flight, then the adds in sample order (same association as a
    float v[16];                          // one-by-one loop; that loop waited for every load: 16 us for 64 samples)
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = norms[(size_t)(s + k) * ld + j];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += v[k];
  }
  for (; s < s1; ++s) acc += norms[(size_t)s * ld + j];
  out[(size_t)blockIdx.y * out_stride + j] = chain ? bf16_round(acc) : acc;
}

// ------------------------------------------------------------------------------------------------
// Standalone activation-L2 kernel (the hook body on an activation tensor resident in HBM) — HBM-bound.
//   act [n, tokens, ld] (bf16 or f32) ; norms[s][j] = sqrt(sum_t act[s,t,j]^2)
// grid (ld/512, n); 256 threads: wave w streams tokens w, w+4, ... ; lane owns 8 consecutive neurons
// (one 16-B load per token row for bf16), 16 token rows in flight per wave; cross-wave fold through LDS.
template <typename T, int NT = 0>
__device__ __forceinline__ void load8(const T* p, float (&f)[8]);
template <>
__device__ __forceinline__ void load8<bf16, 0>(const bf16* p, float (&f)[8]) {
  const bf16x8 v = *(const bf16x8*)p;
#pragma unroll
  for (int k = 0; k < 8; ++k) f[k] = (float)v[k];
}
template <>
__device__ __forceinline__ void load8<bf16, 1>(const bf16* p, float (&f)[8]) {
  const bf16x8 v = __builtin_nontemporal_load((const bf16x8*)p);
#pragma unroll
  for (int k = 0; k < 8; ++k) f[k] = (float)v[k];
}
template <>
__device__ __forceinline__ void load8<float, 0>(const float* p, float (&f)[8]) {
  const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
  for (int k = 0; k < 4; ++k) { f[k] = a[k]; f[4 + k] = b[k]; }
}
template <>
__device__ __forceinline__ void load8<float, 1>(const float* p, float (&f)[8]) {
  const f32x4 a = __builtin_nontemporal_load((const f32x4*)p), b = __builtin_nontemporal_load((const f32x4*)(p + 4));
#pragma unroll
  for (int k = 0; k < 4; ++k) { f[k] = a[k]; f[4 + k] = b[k]; }
}

template <typename T, int NT = 0>
__global__ __launch_bounds__(256) void act_l2_norms_kernel(const T* __restrict__ act, float* __restrict__ ssq_ws,
                                                          RowMap rm, int ld, int n) {
  // grid (ld/512, n, 2): blockIdx.z = token half.  One (sample, 512-column chunk) per block gave 6 x 64 = 384 blocks for
  // a B/16 calibration batch — 1.5 per CU, so half the chip idled through the second round; with the tokens cut in two
  // it is 768 = 3 per CU.  The halves' sums of squares go to ssq_ws[half][s][j]; score_colsum_kernel adds them
  // (first half + second half: a fixed association), takes the root and sums the samples in order.
  const int tokens = rm.tokens;
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = blockIdx.y;
  const int col = blockIdx.x * 512 + lane * 8;
  const bool ok = col < ld;
  const int mid = (tokens + 1) >> 1;
  const int t_begin = blockIdx.z ? mid : 0, t_end = blockIdx.z ? tokens : mid;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  if (ok) {
    const T* base = act + (size_t)row_of(rm, s) * ld + col;
    int t = t_begin + wave;
    // 16 token rows in flight per wave (16 KiB; 12 waves per CU => ~190 KiB per CU outstanding): with 4 rows the loop
    // was latency-bound (13 round trips of ~1.2 us per wave = the whole 16 us of the launch)
    for (; t + 60 < t_end; t += 64) {
      float f[16][8];
#pragma unroll
      for (int r = 0; r < 16; ++r) load8<T, NT>(base + (size_t)(t + 4 * r) * ld, f[r]);
#pragma unroll
      for (int r = 0; r < 16; r += 4)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += f[r][k] * f[r][k] + f[r + 1][k] * f[r + 1][k] + f[r + 2][k] * f[r + 2][k] + f[r + 3][k] * f[r + 3][k];
    }
    for (; t + 28 < t_end; t += 32) {
      float f[8][8];
#pragma unroll
      for (int r = 0; r < 8; ++r) load8<T, NT>(base + (size_t)(t + 4 * r) * ld, f[r]);
#pragma unroll
      for (int r = 0; r < 8; r += 4)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += f[r][k] * f[r][k] + f[r + 1][k] * f[r + 1][k] + f[r + 2][k] * f[r + 2][k] + f[r + 3][k] * f[r + 3][k];
    }
    for (; t + 12 < t_end; t += 16) {
      float f0[8], f1[8], f2[8], f3[8];
      load8<T, NT>(base + (size_t)t * ld, f0);
      load8<T, NT>(base + (size_t)(t + 4) * ld, f1);
      load8<T, NT>(base + (size_t)(t + 8) * ld, f2);
      load8<T, NT>(base + (size_t)(t + 12) * ld, f3);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += f0[k] * f0[k] + f1[k] * f1[k] + f2[k] * f2[k] + f3[k] * f3[k];
    }
    for (; t < t_end; t += 4) {
      float f0[8];
      load8<T, NT>(base + (size_t)t * ld, f0);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += f0[k] * f0[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[wave][lane * 8 + k] = acc[k];
  __syncthreads();
  for (int c = threadIdx.x; c < 512; c += 256) {
    if (blockIdx.x * 512 + c < ld)
      ssq_ws[((size_t)blockIdx.z * n + s) * ld + blockIdx.x * 512 + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
  }
}

// step 2 of the standalone path: out[g][j] = sum over the samples s of group g, in index order, of
// sqrt(ssq[0][s][j] + ssq[1][s][j]) (bf16-rounded per sample and per group sum on the reference chain).
__global__ __launch_bounds__(256) void score_colsum_halves_kernel(const float* __restrict__ ssq_ws, float* __restrict__ out, size_t out_stride, int n,
                                                                   int group, int ld, int chain) {
  // grid (ld/64, groups); 256 threads = 64 columns x 4 sample lanes: lane q adds samples s0+q, s0+q+4, ... of the group in
  // that order, the four partial sums are folded as (p0 + p1) + (p2 + p3) — a fixed association that depends only on the
  // group's own samples, so the result does not depend on what else shares the launch.  (One thread per column walked
  // the 64 samples alone: 7.5 us on 12 blocks for a B/16 batch.)
  __shared__ float part[4][64];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  const int s0 = blockIdx.y * group;
  const int s1 = min(n, s0 + group);
  const float* h0 = ssq_ws;
  const float* h1 = ssq_ws + (size_t)n * ld;
  float acc = 0.f;
  if (j < ld) {
    int s = s0 + q;
    for (; s + 28 < s1; s += 32) {          // 8 samples (16 loads) in flight
      float a[8], b[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { a[k] = h0[(size_t)(s + 4 * k) * ld + j]; b[k] = h1[(size_t)(s + 4 * k) * ld + j]; }
#pragma unroll
      for (int k = 0; k < 8; ++k) { float nrm = sqrtf(a[k] + b[k]); if (chain) nrm = bf16_round(nrm); acc += nrm; }
    }
    for (; s < s1; s += 4) { float nrm = sqrtf(h0[(size_t)s * ld + j] + h1[(size_t)s * ld + j]); if (chain) nrm = bf16_round(nrm); acc += nrm; }
  }
  part[q][c] = acc;
  __syncthreads();
  if (q == 0 && j < ld) {
    const float t = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
    out[(size_t)blockIdx.y * out_stride + j] = chain ? bf16_round(t) : t;
  }
}

// ------------------------------------------------------------------------------------------------
// top-1: one wave per image; first-max-index rule (torch.argmax: lowest index among equal maxima; NaN
// counts as maximal, as torch does).  correct += (pred == label) with an integer atomic (exact, order-free).
__global__ __launch_bounds__(256) void argmax_top1_kernel(const float* __restrict__ logits, int n, int classes,
                                                         int32_t* __restrict__ pred, const int64_t* __restrict__ labels,
                                                         unsigned long long* __restrict__ correct, int period = 0) {
  // period > 0 (ssp2_tail_slots): the n rows are `n / period` SLOTS of `period` images that share one label vector; row r is
  // compared with labels[r % period] and counted in correct[r / period]
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float* lr = logits + (size_t)row * classes;
  float best = -INFINITY; int bi = 0x7fffffff; bool bnan = false;
  for (int c = lane; c < classes; c += 64) {
    const float v = lr[c];
    const bool vnan = v != v;
    if (vnan) { bnan = true; bi = c; break; }            // first NaN wins for this lane (lowest index first)
    if (v > best || bi == 0x7fffffff) { best = v; bi = c; }   // strict >: lowest index among equals
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o); const int on = __shfl_xor((int)bnan, o);
    bool take;
    if (bnan != (bool)on) take = (bool)on;               // NaN beats non-NaN
    else if (bnan) take = oi < bi;                       // both NaN: lower index
    else take = (ob > best) || (ob == best && oi < bi);
    if (take) { best = ob; bi = oi; bnan = (bool)on; }
  }
  if (lane == 0) {
    if (pred) pred[row] = bi;
    if (labels && correct) {
      const int li = period > 0 ? row % period : row, ci = period > 0 ? row / period : 0;
      if (labels[li] == (int64_t)bi) atomicAdd(correct + ci, 1ULL);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Weight compaction for the width prune (a8): dst[r][c] = src[map_r(r)][map_c(c)] over bf16 matrices, zero padding.
//   rows_keep != nullptr : row r of dst comes from row rows_keep[r] of src (fc1: neurons are rows)
//   cols_keep != nullptr : col c of dst comes from col cols_keep[c] of src (fc2: neurons are columns)
__global__ void gather_matrix_kernel(const bf16* __restrict__ src, int src_ld, bf16* __restrict__ dst, int dst_ld, int dst_rows_pad,
                                     int n_rows, int n_cols, const int* __restrict__ rows_keep, const int* __restrict__ cols_keep) {
  const long total = (long)dst_rows_pad * dst_ld;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / dst_ld), c = (int)(i - (long)r * dst_ld);
    bf16 v = (bf16)0.f;
    if (r < n_rows && c < n_cols) v = src[(size_t)(rows_keep ? rows_keep[r] : r) * src_ld + (cols_keep ? cols_keep[c] : c)];
    dst[i] = v;
  }
}
__global__ void gather_vector_kernel(const float* __restrict__ src, float* __restrict__ dst, int n_pad, int n, const int* __restrict__ keep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pad) dst[i] = i < n ? src[keep[i]] : 0.f;
}


// ------------------------------------------------------------------------------------------------
// Weight ingest: fp32 [rows, cols] (nn.Linear [out, in]) -> bf16 [rows_pad, ld], zero padded, RNE (NaN stays NaN) —
// the cast torch.autocast applies to the weight.  One thread per 8 output columns (16-byte store).
__global__ void convert_pad_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int rows, int cols, int rows_pad, int ld) {
  const int chunks = ld / 8;
  const long total = (long)rows_pad * chunks;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / chunks), c0 = (int)(i - (long)r * chunks) * 8;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16)((r < rows && c0 + j < cols) ? src[(size_t)r * cols + c0 + j] : 0.f);
    *(bf16x8*)(dst + (size_t)r * ld + c0) = v;
  }
}
// bias: rounded to bf16, kept as fp32 (the epilogues add it in fp32 registers), zero padded
__global__ void round_bias_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int n_pad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pad) dst[i] = i < n ? bf16_round(src[i]) : 0.f;
}

// Weight ingest of MANY device-resident tensors in one launch (ssp2_load_tensors_dev).  A live module hands over ~150 tensors (ViT-B/16);
// one convert / round / copy launch each was 1.5 ms of launches for 0.1 ms of memory traffic inside the caller's prune bracket.  The
// descriptors travel as the kernel argument (no table in HBM, nothing to keep alive on the host); a work unit is one 8-element chunk of
// a destination — the very chunk of convert_pad_bf16_kernel / 8 elements of round_bias_kernel / 8 floats of a plain copy —, units are
// numbered through all tensors of the batch (work_end = running total) and a thread finds its tensor by bisection in LDS.
struct IngestDesc {
  const float* src; void* dst;
  int rows, cols, rows_pad, ld;       // matrix: source [rows, cols], image [rows_pad, ld]; bias: rows = n, rows_pad = n_pad; copy: rows = n
  int type;                           // 0 matrix -> bf16 image, 1 bias -> bf16-rounded fp32, 2 fp32 copy
  unsigned work_end;                  // units of tensors 0 .. this one
};
constexpr int kIngestBatch = 64;      // 64 x 40 B + 8 = 2568 B of kernel argument (limit 4 KiB)
struct IngestBatch { IngestDesc d[kIngestBatch]; int count; };
__global__ __launch_bounds__(256) void ingest_batch_kernel(const IngestBatch b) {
  __shared__ IngestDesc ds[kIngestBatch];
  for (int i = threadIdx.x; i < b.count; i += blockDim.x) ds[i] = b.d[i];
  __syncthreads();
  const unsigned total = ds[b.count - 1].work_end;
  for (unsigned u = blockIdx.x * blockDim.x + threadIdx.x; u < total; u += gridDim.x * blockDim.x) {
    int lo = 0, hi = b.count - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (ds[mid].work_end > u) hi = mid; else lo = mid + 1; }
    const IngestDesc d = ds[lo];
    const unsigned i = u - (lo ? ds[lo - 1].work_end : 0u);
    if (d.type == 0) {
      const int chunks = d.ld / 8;
      const int r = (int)(i / chunks), c0 = (int)(i - (unsigned)r * chunks) * 8;
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (bf16)((r < d.rows && c0 + j < d.cols) ? d.src[(size_t)r * d.cols + c0 + j] : 0.f);
      *(bf16x8*)((bf16*)d.dst + (size_t)r * d.ld + c0) = v;
    } else if (d.type == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int k = (int)i * 8 + j; if (k < d.rows_pad) ((float*)d.dst)[k] = k < d.rows ? bf16_round(d.src[k]) : 0.f; }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int k = (int)i * 8 + j; if (k < d.rows) ((float*)d.dst)[k] = d.src[k]; }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// fp8 weight image: bf16 [rows_pad, ld] -> e4m3 bytes [rows_pad, ld8] with one dequantisation scale per row
// (scale = amax / 448, the e4m3 maximum; 1 for an all-zero row), zero padded.  One wave per row.
__global__ __launch_bounds__(256) void quant_rows_e4m3_kernel(const bf16* __restrict__ w, int ld, uint8_t* __restrict__ w8, int ld8,
                                                             float* __restrict__ scale, int rows, int cols, int rows_pad) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows_pad) return;
  float amax = 0.f;
  if (row < rows)
    for (int c = lane; c < cols; c += 64) amax = fmaxf(amax, fabsf((float)w[(size_t)row * ld + c]));
  amax = wave_max(amax);
  const float sc = amax > 0.f ? amax / 448.f : 1.f;
  const float inv = 1.f / sc;
  for (int c4 = lane; c4 < ld8 / 4; c4 += 64) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int c = c4 * 4 + k; v[k] = (row < rows && c < cols) ? (float)w[(size_t)row * ld + c] * inv : 0.f; }
    *(uint32_t*)(w8 + (size_t)row * ld8 + c4 * 4) = pack_e4m3x4_from(v[0], v[1], v[2], v[3]);
  }
  if (lane == 0) scale[row] = sc;
}

// fp8 calibration (ssp2_fp8_calibrate_*): how many e4m3 bytes of the fc1 -> fc2 hand-off [rows, cols] (leading dimension ld bytes) sit ON the
// top code (+-448, 0x7e / 0xfe) — what the saturating cast of the GELU output writes for every value at or beyond the e4m3 range.  The fc1
// epilogue cannot count them itself (no register left: 253 VGPRs); a calibration pass can afford one more read of the activation.
__global__ void e4m3_top_code_count_kernel(const uint8_t* __restrict__ a, long rows, int cols, int ld, unsigned long long* __restrict__ count) {
  unsigned int c = 0;
  const int c16 = cols / 16;
  const long n16 = rows * c16;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
    const long r = i / c16; const int k = (int)(i - r * c16);
    const i32x4 v = *(const i32x4*)(a + r * ld + k * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned int w = (unsigned int)v[j] & 0x7f7f7f7fu;
#pragma unroll
      for (int b = 0; b < 4; ++b) c += ((w >> (8 * b)) & 0xffu) == 0x7eu;
    }
  }
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, (unsigned long long)c);
}

// ------------------------------------------------------------------------------------------------
// fp8 calibration (ssp2_fp8_calibrate_*): largest |value| of a bf16 matrix [rows, cols] (leading dimension ld) -> atomicMax on the bits of
// a non-negative float (*amax_bits; non-negative floats order as unsigned integers).
__global__ void absmax_bf16_kernel(const bf16* __restrict__ a, long rows, int cols, int ld, unsigned int* __restrict__ amax_bits) {
  float m = 0.f;
  const long n8 = rows * (cols / 8);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const long r = i / (cols / 8); const int c = (int)(i - r * (cols / 8));
    const bf16x8 v = *(const bf16x8*)(a + r * ld + c * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) m = fmaxf(m, fabsf((float)v[k]));
  }
  m = wave_max_dpp(m);
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(amax_bits, __builtin_bit_cast(unsigned int, m));
}
