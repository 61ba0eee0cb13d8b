// EXPERIMENT (round 2, not part of libssp2vit; built only into tools/gemm_bench, epi 40/41/42) — measured result at the end.
// 256 x 256 x 64 bf16 MFMA GEMM, FOUR waves per workgroup — one wave per SIMD, 128 x 128 outputs per wave, the 256
// accumulator registers in the AGPR half of the unified file (512 registers per lane at one wave per SIMD).  Persistent,
// one workgroup per CU; same LDS image, ring (A0 A1 B0 B1 A2), DMA pieces and swizzle as gemm256_bf16_kernel; results are
// bit-identical to it and to gemm_bf16_kernel (same K order, same rounding points; tools/gemm_bench checks).
//
// Why a second large-tile kernel.  The 8-wave ping-pong loop of gemm256.hip.h spends 3.1 k cycles per K-tile against
// 2.05 k cycles of matrix work: two waves share a SIMD, each phase ends in a workgroup barrier (4 per K-tile), and a
// wave that loads cannot issue MFMAs.  Here a wave never waits for a partner:
//   * fragments are DOUBLE-BUFFERED in registers (2 x 8 x 4 VGPRs): the eight ds_read_b128 of k-step u+1 and the four
//     LDS-DMA pieces of the period are issued in the gaps between the 16 MFMAs of k-step u (a gap hides ~5 single-issue
//     instructions beside the matrix pipe, MI355X_MICROARCH.md);
//   * ONE barrier per K-tile, at the top of its last k-step: by then every wave holds that step's fragments in registers,
//     so all LDS reads of the K-tile are over and its two ring slots are re-filled at once (B(kt+2), A(kt+3));
//   * the counted wait at that barrier is vmcnt(8): only the A(kt+2) pieces issued in the previous period may still fly.
// LDS fragment traffic is 64 B/clk/CU instead of 96 (a 128 x 128 wave tile reads 8 fragments per 16 MFMAs, a 128 x 64
// one 6 per 8).
//
// RESULT (MI355X, scripts/gemm_w4.sh, gpurun_out/r02_gemm_w4_a.txt): bit-identical to the other two kernels on every edge
// shape at the first run, and 5-15 % SLOWER than the 8-wave ping-pong kernel (QKV 63040x2304x768: 772 vs 852 TF; fc2
// K = 3072: 731 vs 908; H/14 QKV K = 1280: 934 vs 1027).  s_memtime stamps (r02_w4_stamps_a.txt): K-tile period 3.3 k
// cycles against 2.56 k for the 8-wave loop under the same stamps: the 16 LDS-DMA pieces a wave issues per K-tile cost
// ~50 cycles EACH of its own instruction stream (the vector-memory path is shared by the four SIMDs and busy > 55 % of
// the time with the tile's 64 KiB per K-tile), and with one wave per SIMD nothing else feeds the matrix pipe meanwhile;
// plus 470 cycles at the K-tile barrier (lgkmcnt 110, vmcnt 258, barrier 101).  The serial epilogue of one wave per SIMD
// is slower too (GELU 22.8 k cycles, VALU issue at 4 cycles per instruction for a lone wave).  Kept as a record.
#pragma once
#include "../gemm.hip.h"
#include <type_traits>

#ifdef GEMM_STAMPS
#define W4STAMP(slot) do { if (tiles_done == 2 && tid == 0 && (slot) < 60) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g.stamps[(size_t)blockIdx.x * 64 + (slot)] = t_; } } while (0)
#else
#define W4STAMP(slot) do {} while (0)
#endif

struct G256W4 {
  static constexpr int BM = 256, BN = 256, NW = 4;
  static constexpr int TM = 4, TN = 4;                  // 32x32 MFMA tiles per wave along M / N
  static constexpr int SLOT = 32768;
  static constexpr int A0 = 0, A1 = SLOT, B0 = 2 * SLOT, B1 = 3 * SLOT, A2 = 4 * SLOT;
  static constexpr int LDS = 5 * SLOT;
};

template <int EPI>
__global__ __launch_bounds__(256) void gemm256w4_kernel(const GemmArgs g) {
  using C = G256W4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int ntiles = g.tiles_m * g.tiles_n;
  const int G = gridDim.x;
  const int wg = xcd_remap(blockIdx.x, G);
  const int nk = g.K / GEMM_BK;

  // fragment reads: row (w?*128 + i*32 + l31) of the slot, 16-byte chunk (2s + lh) ^ ((row >> 1) & 7)
  const int lane_a = (wm * 128 + l31) * 128, lane_b = (wn * 128 + l31) * 128;
  int t16[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) t16[k] = ((2 * k + lh) ^ ((l31 >> 1) & 7)) << 4;

  // LDS-DMA pieces: wave w carries pieces w, w + 4, ..., w + 28 of each operand's 32 (8 rows x 128 B each)
  uint32_t a_src[8], w_src[8];
  const char* a_tile = nullptr;
  const char* w_tile = nullptr;
  int m0 = 0, n0 = 0;
  auto set_tile = [&](int tile) {
    const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;       // N fastest: neighbours share the A panel
    m0 = tm * C::BM; n0 = tn * C::BN;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = (wave + 4 * i) * 8 + (lane >> 3);
      const int c_src = (lane & 7) ^ ((row >> 1) & 7);
      int gr = m0 + row; gr = gr < g.M ? gr : g.M - 1;
      a_src[i] = (uint32_t)(gr - m0) * (uint32_t)g.lda * 2 + c_src * 16;
      w_src[i] = (uint32_t)row * (uint32_t)g.ldw * 2 + c_src * 16;
    }
    a_tile = (const char*)(g.A + (size_t)m0 * g.lda);
    w_tile = (const char*)(g.W + (size_t)n0 * g.ldw);
  };
  auto piece_a = [&](int slot, int kt, int i) { glds16(a_tile + (size_t)kt * (GEMM_BK * 2) + a_src[i], smem + slot + (wave + 4 * i) * 1024); };
  auto piece_b = [&](int slot, int kt, int i) { glds16(w_tile + (size_t)kt * (GEMM_BK * 2) + w_src[i], smem + slot + (wave + 4 * i) * 1024); };
  auto a_slot = [](int kt) { const int r = kt % 3; return r == 0 ? C::A0 : (r == 1 ? C::A1 : C::A2); };
  auto b_slot = [](int kt) { return (kt & 1) ? C::B1 : C::B0; };

  int tiles_done = -1;
  for (int tile = wg; tile < ntiles; tile += G) {
    ++tiles_done;
    set_tile(tile);
    W4STAMP(0);
    const float bias_lane[2] = {g.bias[n0 + wn * 128 + lane], g.bias[n0 + wn * 128 + 64 + lane]};
    // ---- prologue: [B(0) A(0)] [B(1) A(1)] [A(2)]; K-tile 0 needs the first 16 pieces
    __builtin_amdgcn_s_barrier();                          // the previous tile's epilogue staging is done with the slots
#pragma unroll
    for (int i = 0; i < 8; ++i) piece_b(C::B0, 0, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) piece_a(C::A0, 0, i);
    if (nk > 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) piece_b(C::B1, 1, i);
#pragma unroll
      for (int i = 0; i < 8; ++i) piece_a(C::A1, 1, i);
    }
    if (nk > 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) piece_a(C::A2, 2, i);
    }
    f32x16 acc[C::TM][C::TN];
    if (nk > 2) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    bf16x8 fa[2][C::TM], fb[2][C::TN];
    auto read_frags = [&](int set, int kt, int s) {
      const char* As = smem + a_slot(kt) + lane_a + t16[s];
      const char* Bs = smem + b_slot(kt) + lane_b + t16[s];
#pragma unroll
      for (int i = 0; i < C::TN; ++i) fb[set][i] = *(const bf16x8*)(Bs + i * 4096);
#pragma unroll
      for (int i = 0; i < C::TM; ++i) fa[set][i] = *(const bf16x8*)(As + i * 4096);
    };
    W4STAMP(1);
    read_frags(0, 0, 0);

    // One k-step: 16 MFMAs on fragment set `set`, with the reads of the next step and 4 DMA pieces in their gaps.
    //   pb / pa : K-tiles whose B / A pieces this period issues (-1: none); piece indices i0 .. i0+3 of the period's 16
    auto kstep = [&](auto zero_c, int set, int nkt, int ns, bool do_read, int pb, int pa, int i0) {
      constexpr bool ZERO = decltype(zero_c)::value;            // first k-step of a tile: C = 0 (no accumulator init pass)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this step's fragments (issued one step ago) have landed
      __builtin_amdgcn_sched_barrier(0);
      const f32x16 zero = {};
#pragma unroll
      for (int a = 0; a < C::TM; ++a) {
#pragma unroll
        for (int b = 0; b < C::TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[set][b], fa[set][a], ZERO ? zero : acc[a][b], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // behind the first two groups of four MFMAs: the eight fragment reads of the next step (they then have the other
        // two groups' 256 cycles to land before the lgkmcnt wait above); behind every group: one DMA piece of the period
        if (do_read && a < 2) {
          const char* As = smem + a_slot(nkt) + lane_a + t16[ns];
          const char* Bs = smem + b_slot(nkt) + lane_b + t16[ns];
          fb[set ^ 1][2 * a] = *(const bf16x8*)(Bs + (2 * a) * 4096);
          fb[set ^ 1][2 * a + 1] = *(const bf16x8*)(Bs + (2 * a + 1) * 4096);
          fa[set ^ 1][2 * a] = *(const bf16x8*)(As + (2 * a) * 4096);
          fa[set ^ 1][2 * a + 1] = *(const bf16x8*)(As + (2 * a + 1) * 4096);
        }
        const int p = i0 + a;                                   // 0..15: 0..7 = B pieces, 8..15 = A pieces
        if (p < 8) { if (pb >= 0) piece_b(b_slot(pb), pb, p); }
        else { if (pa >= 0) piece_a(a_slot(pa), pa, p - 8); }
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    // k-step (0, 0) is peeled: it starts the accumulation from C = 0 (no 256-register zeroing pass per tile)
    kstep(std::true_type{}, 0, 0, 1, true, -1, -1, 4);
    for (int kt = 0; kt < nk; ++kt) {
      // the period that started at the barrier of K-tile kt-1 issues B(kt+1), A(kt+2): pieces 0..3 in that K-tile's last
      // k-step, 4..7 in k-step 0 of kt (below, at the end of the previous iteration), 8..15 in k-steps 1 and 2
      const int pb_prev = (kt >= 1 && kt + 1 < nk) ? kt + 1 : -1, pa_prev = (kt >= 1 && kt + 2 < nk) ? kt + 2 : -1;
      kstep(std::false_type{}, 1, kt, 2, true, pb_prev, pa_prev, 8);
      kstep(std::false_type{}, 0, kt, 3, true, pb_prev, pa_prev, 12);
      // ---- the barrier of K-tile kt: every wave holds the step-3 fragments after its lgkmcnt wait, all reads of kt are over
      const bool more = kt + 1 < nk;
      const int pb = (kt + 2 < nk) ? kt + 2 : -1, pa = (kt + 3 < nk) ? kt + 3 : -1;
      if (more) {
        if (kt < 14) W4STAMP(2 + 4 * kt);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (kt < 14) W4STAMP(3 + 4 * kt);
        if (kt + 3 < nk && kt >= 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // steady state: only A(kt+2) may still fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (kt < 14) W4STAMP(4 + 4 * kt);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt < 14) W4STAMP(5 + 4 * kt);
      }
      kstep(std::false_type{}, 1, kt + 1, 0, more, pb, pa, 0);
      if (more) kstep(std::false_type{}, 0, kt + 1, 1, true, pb, pa, 4);
    }
    W4STAMP(58);
    // pin the accumulators in the AGPR half (and above the epilogue)
#pragma unroll
    for (int a = 0; a < C::TM; ++a)
#pragma unroll
      for (int b = 0; b < C::TN; ++b) asm volatile("" : "+a"(acc[a][b]));

    // ---------------------------------------------------------------- epilogue (serial, the ring is idle)
    // acc[a][b][4q + r]: row = row0 + a*32 + l31, column = col0 + b*32 + 8q + 4*lh + r (operands swapped)
    __builtin_amdgcn_s_barrier();                        // every wave is done reading the last K-tile: slots may be reused as staging
    const int row0 = m0 + wm * 128, col0 = n0 + wn * 128;
    char* const stg = smem + wave * 40960;               // 40 KiB per wave: [0, 512) bias, [1024, 1024 + 32 KiB) tile staging
    if (col0 < g.N) {
      ((float*)stg)[lane] = bias_lane[0];
      ((float*)stg)[64 + lane] = bias_lane[1];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      char* const tile_stg = stg + 1024;
#pragma unroll
      for (int a = 0; a < C::TM; ++a) {
        // 32 rows x 128 columns
        if (EPI == EPI_RESID) {
          // fp32 staging: 32 rows x 512 B; chunk (16 B = 4 columns) index cq = b*8 + 2q + lh, swizzled by row & 31
#pragma unroll
          for (int b = 0; b < C::TN; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 bq = *(const f32x4*)(stg + (b * 32 + 8 * q + 4 * lh) * 4);
              const uint32_t p0 = pack_bf16x2(acc[a][b][4 * q] + bq[0], acc[a][b][4 * q + 1] + bq[1]);
              const uint32_t p1 = pack_bf16x2(acc[a][b][4 * q + 2] + bq[2], acc[a][b][4 * q + 3] + bq[3]);
              f32x4 v; v[0] = bf16lo_f32(p0); v[1] = bf16hi_f32(p0); v[2] = bf16lo_f32(p1); v[3] = bf16hi_f32(p1);
              const int cq = b * 8 + 2 * q + lh;
              *(f32x4*)(tile_stg + l31 * 512 + ((cq ^ l31) & 31) * 16) = v;
            }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          // 32 rows x 32 chunks = 1024 chunks, 16 per lane: lane -> chunk (lane & 31), rows it*2 + (lane >> 5)
#pragma unroll
          for (int it = 0; it < 16; ++it) {
            const int r = it * 2 + (lane >> 5), c = lane & 31;
            const f32x4 v = *(const f32x4*)(tile_stg + r * 512 + ((c ^ r) & 31) * 16);
            const int m = row0 + a * 32 + r;
            if (m < g.M) {
              float* px = g.x + (size_t)m * g.ldx + col0 + c * 4;
              f32x4 xv = *(const f32x4*)px;
              xv += v;
              *(f32x4*)px = xv;
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
          // bf16 staging: 32 rows x 256 B; 8-byte piece (4 columns) index cq = b*8 + 2q + lh; 16-byte chunk = cq >> 1, swizzled
#pragma unroll
          for (int b = 0; b < C::TN; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 bq = *(const f32x4*)(stg + (b * 32 + 8 * q + 4 * lh) * 4);
              uint32_t pk[2];
#pragma unroll
              for (int p = 0; p < 2; ++p) {
                pk[p] = pack_bf16x2(acc[a][b][4 * q + 2 * p] + bq[2 * p], acc[a][b][4 * q + 2 * p + 1] + bq[2 * p + 1]);
                if (EPI == EPI_FC1) {
                  f32x2 pre;
                  const f32x2 gl = gelu_erf_pk(pk[p], pre);
                  pk[p] = pack_bf16x2(gl.x, gl.y);
                }
              }
              uint2 v; v.x = pk[0]; v.y = pk[1];
              const int chunk = b * 4 + q;                     // 16 chunks of 16 B per row; lh picks the half
              *(uint2*)(tile_stg + l31 * 256 + ((chunk ^ (l31 & 15)) << 4) + lh * 8) = v;
            }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          // 32 rows x 16 chunks = 512 chunks, 8 per lane: lane -> chunk (lane & 15), rows it*4 + (lane >> 4)
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int r = it * 4 + (lane >> 4), c = lane & 15;
            const bf16x8 v = *(const bf16x8*)(tile_stg + r * 256 + ((c ^ (r & 15)) << 4));
            const int m = row0 + a * 32 + r;
            if (m < g.M) *(bf16x8*)(g.out + (size_t)m * g.ldo + col0 + c * 8) = v;
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    W4STAMP(59);
  }
}
