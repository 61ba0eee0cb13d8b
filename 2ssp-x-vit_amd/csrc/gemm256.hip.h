// Large-tile bf16 MFMA GEMM for the big-M projections (QKV, attention out-proj, fc1 of evaluation passes, fc2):
// 256 x 256 x 64 tiles, 8 waves (2 along M x 4 along N, 128 x 64 per wave), ONE persistent workgroup per CU that
// walks tiles wg, wg+G, wg+2G, ...  Results are bit-identical to gemm_bf16_kernel (same K order, same rounding
// points); csrc/tools/gemm_bench.hip checks that on every run.
//
// LDS (all 160 KiB, one array):   A0 | A1 | B0 | B1 | A2     32 KiB each
//   main loop   asymmetric ring as in gemm.hip.h: the A panel (streams from HBM / Infinity Cache) is three K-tiles
//               deep, the weight panel (L2-resident) two, filled by 16-byte LDS-DMA (1-KiB pieces, 8 per wave per
//               K-tile).  The two wave groups (waves 0-3 / 4-7 = the two waves of every SIMD) run in PING-PONG,
//               half a K-tile per step: one group reads its 12 fragments from LDS while the other issues 16 MFMAs
//               with its 4 DMA pieces between them (see the phase table at the loop); 4 raw s_barriers and ONE
//               counted `s_waitcnt vmcnt(4)` per K-tile, so A(kt+2) stays in flight across the K-tile boundary.
//               The steady-state K-tile body is a separate compile-time instance without any conditional.
//   tile change right after the last K-tile the NEXT tile's B(0), A(0), A(1) go out into B0, A0, A1; the epilogue
//               then runs out of B1 | A2 (64 KiB = 8 KiB per wave, wave-private, no barrier inside).  vmcnt counts
//               loads, stores and LDS-DMA together in issue order, so the next main loop starts behind a counted
//               wait that leaves only the epilogue's LAST stores in flight — nothing is ever drained to zero between
//               two main loops (full tiles; an edge tile takes the conservative vmcnt(0) path).
//
// Epilogues: the MFMA operands are swapped (acc = W_frag x A_frag), so a lane owns ONE output row and FOUR
// consecutive columns per register quad.  Against the row-of-4-rows layout of the plain order this makes the bias
// add and the bf16 conversion packed and replaces 16 two-byte LDS stores per 32x32 block by 4 eight-byte ones;
// outputs leave through a wave-private LDS tile as whole 128-byte row segments.  The residual epilogue keeps 16
// coalesced loads of the fp32 x tile in flight per wave (raw-buffer addressing: no 64-bit address registers) and
// adds the accumulators to them in that row-contiguous layout.  The stage-1 scoring variant keeps the plain operand
// order (see the template comment).
#pragma once
#include "gemm.hip.h"
#include <type_traits>
#ifndef STAMP
#define STAMP(slot) do {} while (0)
#endif
#ifdef GEMM_STAMPS
#define TSTAMP(slot) do { if (tiles_done == 3) STAMP(slot); } while (0)
#define RSTAMP(slot) do { if (tid == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g.stamps[(size_t)blockIdx.x * 64 + (slot)] = t_; } } while (0)
#else
#define TSTAMP(slot) do {} while (0)
#define RSTAMP(slot) do {} while (0)
#endif

// Cache policy of the streamed traffic (activation panel, x tile, outputs).  GEMM_NT = 1 marks all of it non-temporal
// so that it cannot evict the weight panels from the XCD's 4 MiB L2; 2 = the LDS-DMA loads only, 3 = the stores only.
// Measured (gemm_bench + rocprofv3 PMC, 63040-row shapes, interleaved runs): with NT loads QKV's HBM-side fetch halves
// (182 -> 99 MB per launch, L2 hit rate 72 -> 79 %) and yet every shape runs 5-9 % SLOWER; NT stores alone are within
// noise of the default.  The default policy stays; the switch is kept for A/B runs only.
// LDS-DMA pieces (of 4 per unit) issued in the LOAD phase instead of between the MFMAs.  Measured on the 63040-row
// shapes (interleaved runs): 1 -> 1-2 % slower, 2 -> 3 % slower, 4 (a burst, ~140 cycles per piece) -> 25 % slower than
// 0, although the loading wave then idles ~400 cycles per phase at the barrier: a piece issued beside the partner's
// MFMAs slows those down by more than it saves.  Kept as a switch for A/B runs.
#ifndef PP_NL
#define PP_NL 0
#endif
#ifndef GEMM_NT
#define GEMM_NT 0
#endif
#ifndef GEMM_MFMA16
#define GEMM_MFMA16 0
#endif
#if GEMM_NT == 2
#define GLDS_A glds16_nt
#define ST_OUT(p, v) (*(p) = (v))
#elif GEMM_NT == 3
#define GLDS_A glds16
#define ST_OUT(p, v) __builtin_nontemporal_store((v), (p))
#elif GEMM_NT
#define GLDS_A glds16_nt
#define ST_OUT(p, v) __builtin_nontemporal_store((v), (p))
#else
#define GLDS_A glds16
#ifdef GEMM_ABL_NOSTORE   // ablation (timing builds of tools/gemm_bench only): the bf16 / GELU epilogue keeps its arithmetic and
#define ST_OUT(p, v) asm volatile("" :: "v"(v))   // its LDS round trip but sends nothing to memory
#else
#define ST_OUT(p, v) (*(p) = (v))
#endif
#endif

struct G256 {
  static constexpr int BM = 256, BN = 256, NW = 8;
  static constexpr int TM = 4, TN = 2;                  // 32x32 MFMA tiles per wave along M / N
  static constexpr int SLOT = 32768;                    // 256 rows x 64 bf16
  static constexpr int A0 = 0, A1 = SLOT, B0 = 2 * SLOT, B1 = 3 * SLOT, A2 = 4 * SLOT;
  static constexpr int STG = B1;                        // epilogue staging: B1 | A2, 8 KiB per wave
  static constexpr int LDS = 5 * SLOT;                  // 163840
};

#define WAITV(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define WAITL0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// SCORE (EPI_FC1 only): 0 none (evaluation passes), 1 pre-GELU, 2 post-GELU stage-1 score.  The scoring variant
// keeps the PLAIN operand order (lane <-> column, registers <-> rows): the per-(sample, neuron) sums of squares are then
// in-lane accumulations, and every partial sum is formed in exactly the order of gemm_bf16_kernel<EPI_FC1, SCORE>
// (a wave's 128 rows = that kernel's 128-row tile, passes h = 0, 1 = its row-waves), so the slab — and the stage-1
// scores — are bit-identical whichever kernel a launch is routed to.
// F8: both operands are e4m3 bytes (OCP fp8), 128 per 128-byte LDS row, so a K-tile carries K = 128 through the very same
// DMA pieces, ring slots and swizzle; the matrix instruction is v_mfma_scale_f32_32x32x64_f8f6f4 with unit block
// scales (twice the cycles of the bf16 32x32x16 form at four times the K: 2x the rate), 8 per unit instead of 16, fed by
// the same 12 ds_read_b128; the weight row's dequantisation scale is applied in the epilogue (acc * wscale[n] + bias).
// Which 32 of the 64 k-bytes a lane half supplies is free as long as both operands use one rule (lane half h takes
// bytes 32h .. 32h+31 of the 64-byte step, tools/fp8_probe.hip checks the pairing on the hardware).
template <int EPI, int SCORE = 0, bool F8 = false>
__global__ __launch_bounds__(512) void gemm256_bf16_kernel(const GemmArgs g) {
  using C = G256;
  constexpr bool SWAP = !(EPI == EPI_FC1 && SCORE != 0);
  // PRODUCT vs LAB builds.  libssp2vit.so instantiates only what the forward launches: {EPI_BF16, EPI_FC1 x SCORE 0 / 1 / 2, EPI_RESID}
  // x {bf16, e4m3}.  The forms that were built, measured and found slower — the LayerNorm behind the residual epilogue (LNV), the
  // deferred residual (DG), the column-group tile orders (GROUP256) — stay in this file behind SSP2_LAB (lib/libssp2vit_lab.so,
  // tools/gemm_bench: scripts/build_variant.py, _lib.build_library(variant="lab")), with their bit-identity tests run against that
  // build: the product kernel does not carry their template instantiations, branches or registers.
#ifdef SSP2_LAB
  constexpr bool LAB = true;
#else
  constexpr bool LAB = false;
#endif
  constexpr int LNV = (LAB && EPI == EPI_RESID && SCORE >= 3) ? SCORE : 0;     // > 0: LayerNorm of the finished row panels, N = LNV * 256 (see below)
  // DG (EPI_RESID, SCORE == 1, bf16 operands): the DEFERRED residual.  The epilogue only PARKS the tile — bf16(acc + bias), 128 KiB per
  // workgroup in g.dg, written row-contiguous through the usual LDS transposition — and the read-add-write of the fp32 x tile
  // happens during the NEXT tile's main loop, one sixteenth of a wave's 128 x 64 window (16 rows x 32 columns: one 16-byte load of parked
  // values, two 16-byte loads of x, two stores) per K-tile, in the LOAD phase of the wave — while its SIMD partner feeds the matrix pipe.
  // x_new = x + float(bf16(acc + bias)) exactly as in the direct form, and every sum over K keeps its order: results are bit-identical.
  constexpr bool DG = LAB && EPI == EPI_RESID && SCORE == 1 && !F8;
  constexpr int ESZ = F8 ? 1 : 2;                       // bytes per operand element
  constexpr int KT = F8 ? 128 : GEMM_BK;                // K elements per K-tile (128 bytes per LDS row either way)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, lh = lane >> 5;
  const int ntiles = g.tiles_m * g.tiles_n;
  const int G = gridDim.x;
  // workgroups b, b+8, .. share an XCD: give each XCD a contiguous run of tile ids inside every round of G tiles
  const int wg = xcd_remap(blockIdx.x, G);

  // Fragment reads.  A lane reads row (wm*128 + i*32 + l31) of the A slot and row (wn*64 + i*32 + l31) of the B slot;
  // the swizzle (row >> 1) & 7 only sees l31, so one set of four chunk offsets serves every fragment of both
  // operands and the i*32 rows are immediate offsets of the ds_read.
  const int lane_a = (wm * 128 + l31) * 128, lane_b = (wn * 64 + l31) * 128;
  int t16[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)      // bf16: k-step k takes chunk 2k + lh; fp8: step s = k >> 1 takes chunks 4s + 2lh + (k & 1)
    t16[k] = ((F8 ? 4 * (k >> 1) + 2 * lh + (k & 1) : 2 * k + lh) ^ ((l31 >> 1) & 7)) << 4;

  // per-lane source of the wave's 4 LDS-DMA pieces per operand, as 32-bit BYTE offsets from g.A / g.W: the pieces are
  // issued as global_load_lds with an SGPR base (operand + K offset, scalar arithmetic) and this VGPR offset, so a
  // piece between two MFMAs costs no vector instruction besides itself
  uint32_t a_src[4], w_src[4];
  // what a wave carries in the first / second half of a K-tile (see the main loop): g0 B then A, g1 A then B
  // All per-lane offsets are relative to the TILE's operand panels (64-bit wave-uniform bases, set per tile): a launch
  // may span more than 4 GiB of activations (the layer-major search runs up to depth x 320 images per launch).
  const char* a_tile = (const char*)g.A;                // g.A + m0 * lda
  const char* w_tile = (const char*)g.W;                // g.W + n0 * ldw
  const char* ride_base0 = nullptr;                     // half 0: g0 carries the weight panel, g1 the activation panel
  const char* ride_base1 = nullptr;
  int m0 = 0, n0 = 0;
  float bias_next = 0.f;                                // bias[n0 + wn*64 + lane] of the tile being prefetched
  float wsc_next = 1.f;                                 // F8: wscale[...] of the same column
  float asc_next[4] = {1.f, 1.f, 1.f, 1.f};             // F8: ascale[row] of the lane's four rows (row0 + a*32 + l31) of that tile
  auto load_asc = [&](int m0_) {                        // (rows past M: the last valid row's — never stored)
    if constexpr (F8) {
      if (g.ascale) {
#pragma unroll
        for (int a = 0; a < 4; ++a) { int r = m0_ + wm * 128 + a * 32 + l31; r = r < g.M ? r : g.M - 1; asc_next[a] = g.ascale[r]; }
      } else if (g.ascale_const != 0.f) {
#pragma unroll
        for (int a = 0; a < 4; ++a) asc_next[a] = g.ascale_const;
      }
    }
  };
  auto set_tile = [&](int tile) {
    int tm, tn;
    if (g.reverse) tile = ntiles - 1 - tile;
    const int gn = LAB ? g.group_m % 100 : 0, gm = LAB ? g.group_m / 100 : 0;          // group_m = 100 * GM + GN (0: plain order; lab builds only)
    if (LAB && gn > 0 && gn < g.tiles_n) {
      // Super-tiles for the 4 MiB L2 of an XCD: blocks of GM row panels (all of them if GM = 0); inside a block column
      // groups of GN tiles, inside a group N fastest.  An XCD's 32 concurrent tiles then touch GN weight panels instead
      // of all of them, and the next column group of the block re-reads the block's A panels from L2 / Infinity Cache.
      const int bm = gm > 0 ? gm : g.tiles_m;
      const int per_block = bm * g.tiles_n;
      const int mb = tile / per_block, r = tile - mb * per_block;
      const int left_m = g.tiles_m - mb * bm, rows = left_m < bm ? left_m : bm;
      const int per_group = rows * gn;
      const int cg = r / per_group, r2 = r - cg * per_group;
      const int left_n = g.tiles_n - cg * gn, gw = left_n < gn ? left_n : gn;
      const int tr = r2 / gw;
      tm = mb * bm + tr; tn = cg * gn + (r2 - tr * gw);
    } else {
      tm = tile / g.tiles_n; tn = tile - tm * g.tiles_n;            // N fastest: neighbours share the A panel
    }
    m0 = tm * C::BM; n0 = tn * C::BN;
    // the lane-derived parts are recomputed per tile from an opaque lane id (a handful of VALU operations): hoisted out of
    // the tile loop they are 12 more VGPRs across a main loop that lives at the 256-register limit
    int ls = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(ls));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wave + 8 * i) * 8 + (ls >> 3);
      const int c_src = (ls & 7) ^ ((row >> 1) & 7);
      int gr = m0 + row; gr = gr < g.M ? gr : g.M - 1;
      a_src[i] = (uint32_t)(gr - m0) * (uint32_t)g.lda * ESZ + c_src * 16;
      w_src[i] = (uint32_t)row * (uint32_t)g.ldw * ESZ + c_src * 16;
    }
    a_tile = (const char*)g.A + (size_t)m0 * g.lda * ESZ;
    w_tile = (const char*)g.W + (size_t)n0 * g.ldw * ESZ;
  };
  auto piece_a = [&](int off, int kt, int i) { GLDS_A(a_tile + (size_t)kt * (GEMM_BK * 2) + a_src[i], smem + off + (wave + 8 * i) * 1024); };
  auto piece_b = [&](int off, int kt, int i) { glds16(w_tile + (size_t)kt * (GEMM_BK * 2) + w_src[i], smem + off + (wave + 8 * i) * 1024); };
  auto stage_a = [&](int off, int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) piece_a(off, kt, i);
  };
  auto stage_b = [&](int off, int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) piece_b(off, kt, i);
  };

  const int nk = g.K / KT;
  // STREAM (nk >= 2): the K-tile stream does not stop at a tile boundary — the last two K-tiles of a tile carry the NEXT
  // tile's A(0), B(0), A(1) in the very piece slots that would hold K-tiles nk, nk+1, and the ring (sa, sb) keeps turning
  // across tiles.  Round 1 issued those 12 pieces per wave in one burst after the last K-tile (~3 k cycles per tile with
  // nothing on the matrix pipe); riding behind MFMAs they cost ~50 cycles each of a COMPUTE phase.
  const bool stream = nk >= 2;
  int sa = C::A0, sb = C::B0;            // ring slots of the K-tile about to be read (persist across tiles)
  // Tile walk.  Plain: tiles wg, wg + G, ... of the N-fastest numbering.
  // LNV (LayerNorm behind the residual epilogue): the panel's LAST-ARRIVING workgroup normalises it, so (1) the tiles_n column
  // tiles of a panel run at the same time on different CUs and share the A panel through the L2, as in the plain order (round 2's
  // form made ONE workgroup own the whole panel and read its A panel tiles_n times), and (2) the rows it reads were written by
  // other CUs — they are coherent in the L2 of ONE XCD only.  Hence a work queue per XCD, keyed on the hardware's XCC_ID
  // (not on the blockIdx -> XCD convention xcd_remap assumes for speed): XCD x owns panels x, x + 8, ...; queue position p is
  // column tile p % tiles_n of its (p / tiles_n)-th panel, claimed with one atomic add per tile.  Claims run two tiles ahead (the
  // next tile's first K-tiles ride on this tile's last two, so its id must be known a whole tile early); every claim and every
  // arrival is ONE returning atomic of wave 4 issued at the start of an epilogue and read at its end — no spin, no extra
  // barrier anywhere: a workgroup learns at the top of tile t + 2 whether its tile t completed a panel.
  int tile = wg, lnq_next = ntiles, lnq_slot = 0, lnq_prev_panel = -1, lnq_cnt = 0, lnq_claimed = 0;
  uint32_t lnq_res = 0;                // wave 4: lane 0 the claimed queue position, lane 1 the panel's arrivals before ours
  int xcc = 0;
  unsigned int* const lnq_queue = LNV ? g.ln_sync + g.ln_set * 8 : nullptr;
  auto pos_tile = [&](int p) { const int pl = p / g.tiles_n, panel = xcc + 8 * pl; return panel < g.tiles_m ? panel * g.tiles_n + (p - pl * g.tiles_n) : ntiles; };
  if constexpr (LNV > 0) {
    uint32_t r;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(r));
    xcc = r & 7;
    if (blockIdx.x == 0 && tid < 8) g.ln_sync[(g.ln_set ^ 1) * 8 + tid] = 0;      // the NEXT fused launch's queue heads
    if (tid == 256) *(volatile uint32_t*)(smem + C::A2) = atomicAdd(lnq_queue + xcc, 2u);    // A2: the first DMAs go to B0, A0, A1
    __syncthreads();
    const int p0 = __builtin_amdgcn_readfirstlane(*(volatile uint32_t*)(smem + C::A2));
    __syncthreads();
    tile = pos_tile(p0); lnq_next = pos_tile(p0 + 1);
  }
  auto next_of = [&](int t) { return LNV ? lnq_next : t + G; };
  // LayerNorm of a finished 256-row panel (every wave of the workgroup calls it at the same point; no barrier, no LDS inside).
  // All tiles_n column tiles of the panel have arrived — their stores were acknowledged by this XCD's L2 before the arrivals
  // were counted — so the rows are read with device-scope loads (sc1: past the CU's vector L1, which may hold this
  // workgroup's own, older, view of its column range).  Wave w normalises rows w, w + 8, ... exactly as
  // layernorm_bf16_kernel does (one wave per row, ln_row_finish).  The standalone kernel moves 4.5 KB per row through HBM;
  // here the reads are L2 hits and only the bf16 / e4m3 rows leave.
  auto ln_phase = [&](int panel) __attribute__((always_inline)) {
    if constexpr (LNV > 0) {
#ifdef GEMM_ABL_NOLN   // ablation (timing builds of tools/gemm_bench only): queues and arrivals as usual, the rows are not normalised
      if (tid == 0) g.ln_sync[16 + panel] = 0;
      return;
#endif
      const int pm0 = panel * C::BM;
      int ll = lane;
      asm volatile("" : "+v"(ll));                 // opaque: nothing below may be hoisted over a main loop (registers)
      if (tid == 0) g.ln_sync[16 + panel] = 0;      // the arrival counter is left at zero for the next launch
      const int nv = g.N >> 2;
      f32x4 g4[LNV > 0 ? LNV : 1], b4[LNV > 0 ? LNV : 1];
#pragma unroll
      for (int i = 0; i < LNV; ++i) {
        g4[i] = ((const f32x4*)g.ln_g)[i * 64 + ll]; b4[i] = ((const f32x4*)g.ln_b)[i * 64 + ll];     // N = 256 * LNV: every chunk exists
      }
      const int rows_here = g.M - pm0 < C::BM ? g.M - pm0 : C::BM;
      const float inv_d = 1.0f / (float)g.N;
      // wave w: rows w, w + 8, ... (32 of them), two register sets in ping-pong: the loads of row i + 1 are in flight
      // while row i is normalised.  The loads are inline asm with a counted wait of our own (as for the LDS-DMA): left to
      // hipcc, every row waited with vmcnt(0) — for its own loads AND the previous row's stores, 1.9 us per row.
      const uint32_t ll16 = (uint32_t)ll * 16;
      auto load_row = [&](f32x4 (&dst)[LNV > 0 ? LNV : 1], int i) __attribute__((always_inline)) {
        int r = wave + 8 * i;
        r = r < rows_here ? r : rows_here - 1;
        const char* xr = (const char*)(g.x + (size_t)(pm0 + r) * g.ldx);
#pragma unroll
        for (int c = 0; c < LNV; ++c)
          // s_nop 4: the SGPR base may come straight out of a v_readlane_b32 (hipcc reloads spilled scalars that way), and a
          // vector-memory instruction that reads an SGPR a VALU instruction wrote needs five wait states in between — which hipcc
          // pads for instructions it knows, not for inline asm.  (Found the hard way: without it ~30 of 82 240 rows of a dim-1280
          // launch were normalised from another row's chunk, always a wave's SECOND row: the one whose address was a spill reload.)
          asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 sc1" : "=&v"(dst[c]) : "v"(ll16), "s"(xr + c * 1024) : "memory");
      };
      // a counted wait that the row registers pass through (no use moves above it): all but the N youngest vector-memory
      // operations have landed
      // a counted wait that TWO rows' registers pass through (no use moves above it): all but the N youngest vector-memory
      // operations have landed
      auto landed2 = [&](f32x4 (&d)[LNV > 0 ? LNV : 1], f32x4 (&e)[LNV > 0 ? LNV : 1], auto n_c) __attribute__((always_inline)) {
        constexpr int N = decltype(n_c)::value;
        if constexpr (LNV == 3) asm volatile("s_waitcnt vmcnt(%6)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(e[0]), "+v"(e[1]), "+v"(e[2]) : "n"(N) : "memory");
        if constexpr (LNV == 4) asm volatile("s_waitcnt vmcnt(%8)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]) : "n"(N) : "memory");
        if constexpr (LNV == 5) asm volatile("s_waitcnt vmcnt(%10)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]) : "n"(N) : "memory");
      };
      auto rows = [&](auto out8_c) __attribute__((always_inline)) {
        constexpr bool OUT8 = decltype(out8_c)::value;
        // rows past the panel's end (last panel only) are clamped to its last row for the load AND the store: the same
        // bits land on the same address again, and every row costs at least LNV loads + LNV stores — the counts below
        auto finish = [&](const f32x4 (&src)[LNV > 0 ? LNV : 1], int i) __attribute__((always_inline)) {
          int r = wave + 8 * i;
          r = r < rows_here ? r : rows_here - 1;
          const size_t ro = (size_t)(pm0 + r) * g.ln_ld;
          ln_row_finish<LNV, true, OUT8>(src, ll, nv, inv_d, g.ln_eps, g4, b4, OUT8 ? (void*)(g.ln_out8 + ro) : (void*)(g.ln_out + ro),
                                         (OUT8 && g.ln_ascale) ? g.ln_ascale + pm0 + r : nullptr);
        };
        // Rows go through in PAIRS out of R = EIGHT register sets: while a pair is normalised, the loads of the six other rows are in
        // flight.  Why eight: the panel was finished a tile ago and its XCD has written 8 MB since, so the rows come from the
        // Infinity Cache or HBM, ~3 us a load on a chip busy with GEMM traffic; with four sets (one pair in flight, 48 KB per
        // CU) the phase ran at 15 GB/s per CU = 52 us per panel — exactly the standalone kernel's HBM-bound rate, i.e. no gain.
        // Why pairs: a row is one long dependent chain (two wave reductions, a sequential sum of squares); two independent
        // rows in one basic block let hipcc interleave the chains at two waves per SIMD.
        // Issue order: L0 .. L7 | wait(L0, L1) S0 S1 L8 L9 | wait(L2, L3) S2 S3 L10 L11 | ...   Younger than a pair at its wait
        // in the steady state: six load sets and six store sets (12 * LNV <= 60 operations: vmcnt counts to 63); an e4m3 row's
        // scale is one store more — the wait then covers an operation more than it needs to.
        // (dim 1280: six sets — eight of 20 registers each do not fit beside gamma and beta; the 32 rows then take six rounds of
        // six, the last four turns repeat row 31: same bits to the same addresses.)
        constexpr int R = LNV <= 4 ? 8 : 6;
        constexpr int TURNS = (32 + R - 1) / R * R;
        auto turn = [](int i) { return i < 32 ? i : 31; };
        f32x4 rs[R][LNV > 0 ? LNV : 1];
#pragma unroll
        for (int k = 0; k < R; ++k) load_row(rs[k], k);
        // first group: fewer stores are under way (R - 2, R, R + 2, ... younger sets)
#pragma unroll
        for (int q = 0; q < R / 2; ++q) {
          if (q == 0) landed2(rs[0], rs[1], std::integral_constant<int, (R - 2) * LNV>{});
          else if (q == 1) landed2(rs[2], rs[3], std::integral_constant<int, R * LNV>{});
          else if (q == 2) landed2(rs[4], rs[5], std::integral_constant<int, (R + 2) * LNV>{});
          else landed2(rs[R - 2], rs[R - 1], std::integral_constant<int, (R + 4) * LNV>{});      // R = 8 only
          finish(rs[2 * q], 2 * q); finish(rs[2 * q + 1], 2 * q + 1);
          load_row(rs[2 * q], turn(R + 2 * q)); load_row(rs[2 * q + 1], turn(R + 2 * q + 1));
        }
        using NS = std::integral_constant<int, 2 * (R - 2) * LNV>;
#pragma unroll 1
        for (int i = R; i < TURNS; i += R) {
#pragma unroll
          for (int q = 0; q < R / 2; ++q) {
            landed2(rs[2 * q], rs[2 * q + 1], NS{}); finish(rs[2 * q], turn(i + 2 * q)); finish(rs[2 * q + 1], turn(i + 2 * q + 1));
            load_row(rs[2 * q], turn(i + R + 2 * q)); load_row(rs[2 * q + 1], turn(i + R + 2 * q + 1));     // (past the end: redundant re-loads, the counts stay uniform)
          }
        }
#pragma unroll
        for (int q = 0; q < R / 2; ++q) landed2(rs[2 * q], rs[2 * q + 1], std::integral_constant<int, 0>{});   // nothing may still be landing in registers the compiler re-uses
      };
      if (g.ln_out8) rows(std::true_type{}); else rows(std::false_type{});
    }
  };
  // ------------------------------------------------------------------------------------------------ DG: the deferred residual
  // A STEP s (0..15) of a parked tile is half a block p = s >> 1 = (a, b) of the direct epilogue below: rows a*32 + 8j + dr (j = 2(s & 1),
  // 2(s & 1) + 1; lane -> dr = lane >> 3), columns b*32 + 4 (lane & 7) .. + 3 of the wave's 128 x 64 window — the parked values as ONE
  // 16-byte load (lane-linear: the lane that stored them), x as two 16-byte raw-buffer loads (rows past M: out-of-range offsets, loads
  // return 0 and stores are dropped, as in the direct form).  The three loads of the step that a main-loop slot will finish NEXT are inline
  // asm with a counted wait of our own (as for the LDS-DMA pieces; hipcc would drain the piece stream in front of a loop-carried load):
  //   slot of K-tile kt = 1 .. 16 (g0: LOAD(kt,1), g1: LOAD(kt,0)):  wait | 8 adds | 2 stores | 3 loads of the next step
  //   younger than a slot's loads at the next slot's wait: the 8 pieces of one K-tile -> vmcnt(8) (step 0 goes out at the tile top and is
  //   retired by K-tile 0's closing wait: the first slot's vmcnt(8) is then already met);
  //   younger than the pieces a K-tile's closing wait retires: that slot's 5 operations + 4 pieces -> vmcnt(9) instead of vmcnt(4).
  typedef __attribute__((ext_vector_type(4))) unsigned int dg_u32x4;
  dg_u32x4 dgd = {0u, 0u, 0u, 0u};
  f32x4 dgx0 = {0.f, 0.f, 0.f, 0.f}, dgx1 = {0.f, 0.f, 0.f, 0.f};
  int dg_pm0 = 0, dg_pn0 = 0, dg_prows = 0;      // the parked tile: origin, valid rows of this wave's window (0: nothing parked)
  int dg_s = 0;                                   // the step whose loads are in flight / in the registers
  constexpr int DG_STEPS = 16;
  auto dg_words = [&](const void* ptr) __attribute__((always_inline)) -> i32x4 {      // raw-buffer descriptor: base, stride 0, range 2 GiB
    const unsigned long long a = (unsigned long long)ptr;
    i32x4 d;
    d.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a); d.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    d.z = 0x7ffffff0; d.w = 0x00020000;
    return d;
  };
  struct DgLane { int lane_off, rows_left, l16; };
  auto dg_lane = [&]() __attribute__((always_inline)) -> DgLane {
    int le_ = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(le_));                                      // opaque: recomputed per slot, never parked in registers across the main loop
    const int dr_ = le_ >> 3, cc_ = le_ & 7;
    return DgLane{(int)(__umul24((unsigned)dr_, (unsigned)g.ldx) + cc_ * 4) * 4, dg_prows - dr_, le_ * 16};     // ldx < 2^24: one full-rate v_mad_u32_u24
  };
  auto dg_row = [&](int s_, int jj) { return (s_ >> 2) * 32 + 8 * (2 * (s_ & 1) + jj); };
  auto dg_soff = [&](int s_, int jj) { return (dg_row(s_, jj) * g.ldx + ((s_ >> 1) & 1) * 32) * 4; };
  auto dg_xw = [&](const float* base) { return const_cast<float*>(base) + (size_t)(dg_pm0 + wm * 128) * g.ldx + dg_pn0 + wn * 64; };
  auto dg_scratch = [&]() { return g.dg + ((size_t)blockIdx.x * 8 + wave) * 4096; };          // this wave's 16 KiB
  // the three loads of step s_ (steps past the last: every x offset out of range — the counts stay uniform)
  auto dg_issue = [&](int s_, const DgLane& L) __attribute__((always_inline)) {
    if constexpr (DG) {
      const bool live = s_ < DG_STEPS;
      const int sc = live ? s_ : DG_STEPS - 1;
      const i32x4 rin = dg_words(dg_xw(g.xin ? g.xin : g.x));
      const i32x4 rsc = dg_words(dg_scratch());
      const int v0 = (live && dg_row(sc, 0) < L.rows_left) ? L.lane_off : 0x7fffffff;
      const int v1 = (live && dg_row(sc, 1) < L.rows_left) ? L.lane_off : 0x7fffffff;
      const int so0 = __builtin_amdgcn_readfirstlane(dg_soff(sc, 0)), so1 = __builtin_amdgcn_readfirstlane(dg_soff(sc, 1));
      const int sos = __builtin_amdgcn_readfirstlane(sc * 1024);
#if defined(DG_ABL) && (DG_ABL & 1)     // ablation (timing only): the slot's arithmetic without its memory operations — three pseudo-loads keep the vmcnt counts
      asm volatile("s_nop 4\n\tbuffer_load_dword %0, %3, %6, %8 offen\n\tbuffer_load_dword %1, %3, %6, %8 offen\n\tbuffer_load_dword %2, %3, %6, %8 offen"
                   : "=&v"(dgd.x), "=&v"(dgx0.x), "=&v"(dgx1.x)
                   : "v"(L.l16), "v"(v0), "v"(v1), "s"(rsc), "s"(rin), "s"(sos), "s"(so0), "s"(so1) : "memory");
#else
      asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %3, %6, %8 offen\n\tbuffer_load_dwordx4 %1, %4, %7, %9 offen nt\n\tbuffer_load_dwordx4 %2, %5, %7, %10 offen nt"
                   : "=&v"(dgd), "=&v"(dgx0), "=&v"(dgx1)
                   : "v"(L.l16), "v"(v0), "v"(v1), "s"(rsc), "s"(rin), "s"(sos), "s"(so0), "s"(so1) : "memory");
#endif
    }
  };
  // x += parked for the step in the registers (they have LANDED: the caller's wait), and its two stores
  auto dg_finish = [&](int s_, dg_u32x4 d_, f32x4 x0_, f32x4 x1_, const DgLane& L) __attribute__((always_inline)) {
    if constexpr (DG) {
      const bool live = s_ < DG_STEPS;
      const int sc = live ? s_ : DG_STEPS - 1;
      const __amdgpu_buffer_rsrc_t xr_ = __builtin_amdgcn_make_buffer_rsrc(dg_xw(g.x), 0, 0x7ffffff0, 0x00020000);
      const int v0 = (live && dg_row(sc, 0) < L.rows_left) ? L.lane_off : 0x7fffffff;
      const int v1 = (live && dg_row(sc, 1) < L.rows_left) ? L.lane_off : 0x7fffffff;
      {   // eight fp32 adds as four v_pk_add_f32 (IEEE adds, lane by lane: the same sums)
        f32x2 p0, p1, p2, p3, q0, q1, q2, q3;
        p0.x = bf16lo_f32(d_.x); p0.y = bf16hi_f32(d_.x); p1.x = bf16lo_f32(d_.y); p1.y = bf16hi_f32(d_.y);
        p2.x = bf16lo_f32(d_.z); p2.y = bf16hi_f32(d_.z); p3.x = bf16lo_f32(d_.w); p3.y = bf16hi_f32(d_.w);
        q0.x = x0_.x; q0.y = x0_.y; q1.x = x0_.z; q1.y = x0_.w; q2.x = x1_.x; q2.y = x1_.y; q3.x = x1_.z; q3.y = x1_.w;
        q0 += p0; q1 += p1; q2 += p2; q3 += p3;
        x0_.x = q0.x; x0_.y = q0.y; x0_.z = q1.x; x0_.w = q1.y; x1_.x = q2.x; x1_.y = q2.y; x1_.z = q3.x; x1_.w = q3.y;
      }
      __builtin_amdgcn_sched_barrier(0);
#if defined(DG_ABL) && (DG_ABL & 2)     // ablation (timing only): two one-dword stores into the parking area instead of the x stores
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(dg_u32x4, x0_).x, __builtin_amdgcn_make_buffer_rsrc(dg_scratch(), 0, 0x7ffffff0, 0x00020000), L.l16, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(dg_u32x4, x1_).x, __builtin_amdgcn_make_buffer_rsrc(dg_scratch(), 0, 0x7ffffff0, 0x00020000), L.l16, 0, 0);
#else
#ifndef DG_ST_AUX
#define DG_ST_AUX 0
#endif
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dg_u32x4, x0_), xr_, v0, dg_soff(sc, 0), DG_ST_AUX);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dg_u32x4, x1_), xr_, v1, dg_soff(sc, 1), DG_ST_AUX);
#endif
      asm volatile("s_nop 1" ::: "memory");                            // store-data hazard guard, see the direct epilogue
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // a main-loop slot: the step in flight lands, is finished, and the next one goes out
  auto dg_slot = [&]() __attribute__((always_inline)) {
    if constexpr (DG) {
      asm volatile("s_waitcnt vmcnt(8)" : "+v"(dgd), "+v"(dgx0), "+v"(dgx1) :: "memory");
      const DgLane L = dg_lane();
      dg_finish(dg_s, dgd, dgx0, dgx1, L);
      ++dg_s;
      dg_issue(dg_s, L);
    }
  };
  // whatever is left of the parked tile, outside a main loop (a tile's epilogue: steps the main loop had no K-tiles for; after the last
  // tile: all of it): compiler-visible loads, FOUR steps in flight.  `in_regs`: step dg_s sits in dgd / dgx0 / dgx1 and has landed.
  auto dg_drain = [&](bool in_regs) __attribute__((always_inline)) {
    if constexpr (DG) {
      if (in_regs && dg_s < DG_STEPS) {
        asm volatile("" : "+v"(dgd), "+v"(dgx0), "+v"(dgx1));
        dg_finish(dg_s, dgd, dgx0, dgx1, dg_lane());
        ++dg_s;
      }
      const __amdgpu_buffer_rsrc_t rin_ = __builtin_amdgcn_make_buffer_rsrc(dg_xw(g.xin ? g.xin : g.x), 0, 0x7ffffff0, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsc_ = __builtin_amdgcn_make_buffer_rsrc(dg_scratch(), 0, 0x7ffffff0, 0x00020000);
#pragma unroll 1
      for (; dg_s < DG_STEPS; dg_s += 4) {
        const DgLane L = dg_lane();
        dg_u32x4 d_[4]; f32x4 a_[4], b_[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int s_ = dg_s + i;
          const bool live = s_ < DG_STEPS;
          const int sc = live ? s_ : DG_STEPS - 1;
          const int v0 = (live && dg_row(sc, 0) < L.rows_left) ? L.lane_off : 0x7fffffff;
          const int v1 = (live && dg_row(sc, 1) < L.rows_left) ? L.lane_off : 0x7fffffff;
          d_[i] = __builtin_bit_cast(dg_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsc_, L.l16, sc * 1024, 0));
          a_[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin_, v0, dg_soff(sc, 0), 2));
          b_[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin_, v1, dg_soff(sc, 1), 2));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) dg_finish(dg_s + i, d_[i], a_[i], b_[i], L);
      }
    }
  };
#ifdef GEMM_STAGGER
  // experiment (timing builds of tools/gemm_bench only): the workgroups of XCD group x = blockIdx.x & 7 start
  // x * (group_m / 100) * 256 cycles late, so that the eight XCDs' epilogue bursts do not meet in HBM (workgroups of one XCD
  // stay in step and keep sharing their operand panels through its L2)
  for (int c = 0, n = (g.group_m / 100) * (blockIdx.x & 7); c < n; ++c) __builtin_amdgcn_s_sleep(4);
#endif
  if (tile < ntiles) {
    set_tile(tile);
    bias_next = g.bias[n0 + wn * 64 + lane];
    if (F8) wsc_next = g.wscale[n0 + wn * 64 + lane];
    load_asc(m0);
    stage_b(C::B0, 0);
    stage_a(C::A0, 0);
    if (nk > 1) stage_a(C::A1, 1);
  }
  STAMP(0);
  RSTAMP(61);
  int tiles_done = 0;
  bool counted = false;                // previous epilogue took the full-tile path: its last stores may stay in flight
  int tile_adv = ntiles;
  for (; tile < ntiles; tile = tile_adv) {
    ++tiles_done;
    f32x16 acc[C::TM][C::TN];
#pragma unroll
    for (int a = 0; a < C::TM; ++a)
#pragma unroll
      for (int b = 0; b < C::TN; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    TSTAMP(42);
    // K-tile 0 needs B(0), A(0).  Younger than those: A(1) (4 DMAs) on the first tile; on later tiles the whole
    // epilogue — whose last 4 VMEM operations are stores when it ran the full-tile path.
    // (Round 2 experiment, null result: leaving the epilogue's 16 tail stores in flight here — vmcnt(20) instead of vmcnt(4) —
    // changes nothing within the +-3 % run-to-run spread on any shape: wave 0's 5 k cycles at this barrier are the other
    // waves' epilogues, which run at the HBM write rate when all 256 CUs store together, not its own store drain.)
    // (DG: the epilogue always ends with the 16 park stores)
    if (DG || counted || (tiles_done == 1 && nk > 1)) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else                                              asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (LNV > 0) {
      // what wave 4 left in its staging area at the end of the previous tile: [panel to normalise or -1 | queue position claimed
      // for the tile after this one].  That area is the A slot of the previous tile's last K-tile: the ring re-fills it with this
      // tile's A(2), whose first piece goes out behind at least one more barrier.
      if (tiles_done > 1) {
        const uint32_t w0 = *(volatile uint32_t*)(smem + lnq_slot), w1 = *(volatile uint32_t*)(smem + lnq_slot + 4);
        const int ln_panel = __builtin_amdgcn_readfirstlane(w0);
        lnq_next = pos_tile(__builtin_amdgcn_readfirstlane(w1));
        if (ln_panel >= 0) ln_phase(ln_panel);
      }
    }
    // This tile's bias (one column per lane; fp8: and its dequantisation scale) is taken over HERE: hipcc waits with
    // vmcnt(0) for the (long finished) load in front of its first use, and this is the one point of the tile where nothing
    // worth keeping in flight is in flight (at most the 4 A(1) pieces and the previous epilogue's last 4 stores) — at the
    // start of the epilogue the same wait would drain the next tile's first K-tiles, which ride on the last two K-tiles.
    float bias_lane = bias_next;
    asm volatile("v_mov_b32 %0, %0" : "+v"(bias_lane));
    float wsc_lane = wsc_next;
    if (F8) asm volatile("v_mov_b32 %0, %0" : "+v"(wsc_lane));
    float asc[4] = {asc_next[0], asc_next[1], asc_next[2], asc_next[3]};     // taken over here for the same reason as the bias
    if (F8) asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2\n\tv_mov_b32 %3, %3" : "+v"(asc[0]), "+v"(asc[1]), "+v"(asc[2]), "+v"(asc[3]));
    // DG: the three loads of step 0 of the parked tile go out HERE — behind the copy above (hipcc's vmcnt(0) in front of it would wait for
    // them) and a whole K-tile ahead of their slot: K-tile 0 carries none, its closing wait (everything but its own 4 youngest pieces)
    // retires them.  On the first tile nothing is parked (dg_prows = 0: every x offset out of range, same counts).
    if constexpr (DG) dg_issue(0, dg_lane());
    TSTAMP(41);
    // ---- main loop: two wave groups in ping-pong.  A UNIT is half a K-tile: LOAD = 12 fragment reads
    // (ds_read_b128), COMPUTE = 16 MFMAs with 4 LDS-DMA pieces issued BETWEEN the MFMAs (behind pairs 0, 2, 4, 6; s_memtime stamps: a piece
    // costs ~140 cycles of issue in a burst of four, ~30 behind an MFMA), one s_barrier after each.  Waves 4-7 (the
    // SIMD partners of waves 0-3) run one barrier behind, so between any two barriers one wave of every SIMD feeds
    // the matrix pipe while its partner reads LDS:
    //   phase 4kt  : g0 LOAD(kt,0)                    g1 COMPUTE(kt-1,1) + its B(kt+1) pieces
    //   phase 4kt+1: g0 COMPUTE(kt,0) + its B(kt+1)   g1 LOAD(kt,0)
    //   phase 4kt+2: g0 LOAD(kt,1)                    g1 COMPUTE(kt,0) + its A(kt+2) pieces
    //   phase 4kt+3: g0 COMPUTE(kt,1) + its A(kt+2)   g1 LOAD(kt,1)           every wave: counted vmcnt(4)
    // K-tile kt is read in phases 4kt..4kt+3 only, so from phase 4kt on B(kt+1) / A(kt+2) may overwrite the slots of
    // B(kt-1) / A(kt-1).  At the end of phase 4kt+3 the 4 youngest DMAs of every wave are its A(kt+2) pieces: the
    // counted wait retires its B(kt+1) and A(kt+1) pieces one barrier before their first read (phase 4kt+4).
    bf16x8 fa[2][C::TM], fb[2][C::TN];
    i32x8 ga[C::TM], gb[C::TN];                                // F8: one 64-deep step per unit, 32 bytes per lane and fragment
    uint32_t ride0[4], ride1[4];                               // the pieces carried in half 0 / half 1 (selected once per tile)
#pragma unroll
    for (int i = 0; i < 4; ++i) { ride0[i] = wm ? a_src[i] : w_src[i]; ride1[i] = wm ? w_src[i] : a_src[i]; }
    ride_base0 = wm ? a_tile : w_tile;
    ride_base1 = wm ? w_tile : a_tile;
    if (wm) {                                                  // phase 0 of the tile: g1 has no unit to compute yet
      if (nk > 1) stage_b(sb ^ (C::B0 ^ C::B1), 1);            // its B(1) pieces (that slot is free: the tile-start barrier)
      asm volatile("s_barrier" ::: "memory");                  // the skew
    }
    // One K-tile.  B_ON / A_ON (is there a B(kt+1) / an A(kt+2) to fetch?) are compile-time: the steady state
    // (kt < nk - 2) carries no conditional around its DMA pieces, the last two K-tiles are separate instances.
    //   kidx0 / kidx1: K-tile index (inside the panel ride_base0 / ride_base1 points at) of the pieces carried in half 0 / 1
    //   STEADY (compile time): every unit carries its pieces, no conditional anywhere.  Otherwise `flags` decides at run
    //   time (the last two K-tiles of a tile): bit 0 = g0's half-0 unit carries pieces, bit 2 = g1's half-1 unit does,
    //   bit 1 = the other two units do.
    auto ktile = [&](int kt, int kidx0, int kidx1, auto steady_c, int flags, auto dg_c) {
      constexpr bool STEADY = decltype(steady_c)::value;
      constexpr int SLOT = decltype(dg_c)::value;       // 0: none; 1 (DG): this K-tile carries a step of the deferred residual (steady K-tiles only)
      constexpr bool DGA = SLOT == 1;
      constexpr int SLOT_OPS = SLOT == 1 ? 5 : 0;       // vector-memory operations of the slot: younger than the pieces the closing wait retires
      const bool b_on = STEADY || (flags & 1), a_on = STEADY || (flags & 2), g1h1_on = STEADY || (flags & 4);
      const int sa1 = sa == C::A0 ? C::A1 : (sa == C::A1 ? C::A2 : C::A0);
      const int sa2 = sa1 == C::A0 ? C::A1 : (sa1 == C::A1 ? C::A2 : C::A0);
      const int sb1 = sb ^ (C::B0 ^ C::B1);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        //   half 0: g0 carries B(kt+1) -> sb1, g1 carries A(kt+2) -> sa2
        //   half 1: g0 carries A(kt+2) -> sa2, g1 carries B(kt+2) -> sb (K-tile kt's slot: its reads end with this LOAD)
        const bool take_a = (half == 0) == (wm != 0);
        const bool dma_on = (half == 0 && !wm) ? b_on : ((half == 1 && wm) ? g1h1_on : a_on);
        const char* const dbase = (half == 0 ? ride_base0 : ride_base1)                      // wave-uniform: SGPRs
                                  + (size_t)((half == 0 ? kidx0 : kidx1) * (GEMM_BK * 2));
        const int dlds = take_a ? sa2 : (half == 0 ? sb1 : sb);
        const uint32_t (&dsrc)[4] = half == 0 ? ride0 : ride1;
        // PP_NL of the unit's 4 pieces go out in its LOAD phase (behind the fragment reads, in the time the wave would
        // otherwise wait at the barrier), the rest between the MFMAs — except for g1's half 1, whose target slot is
        // still being read during that LOAD
        const bool early_ok = !(half == 1 && wm);
        auto piece = [&](int i) {
          if (take_a) GLDS_A(dbase + dsrc[i], smem + dlds + (wave + 8 * i) * 1024);
          else        glds16(dbase + dsrc[i], smem + dlds + (wave + 8 * i) * 1024);
        };
        // ---------------- LOAD
        if (half == 0 && kt < 6) TSTAMP(1 + 6 * kt);
        {
          const char* As = smem + sa + lane_a;
          const char* Bs = smem + sb + lane_b;
          if constexpr (F8) {
            const int o0 = t16[2 * half], o1 = t16[2 * half + 1];
#pragma unroll
            for (int i = 0; i < C::TN; ++i) {
              const i32x4 lo = *(const i32x4*)(Bs + o0 + i * 4096), hi = *(const i32x4*)(Bs + o1 + i * 4096);
              gb[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int i = 0; i < C::TM; ++i) {
              const i32x4 lo = *(const i32x4*)(As + o0 + i * 4096), hi = *(const i32x4*)(As + o1 + i * 4096);
              ga[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
          } else {
#if GEMM_MFMA16 == 1
          // TIMING PROBE (tools/gemm_bench -DGEMM_MFMA16=1 only; results are NOT valid): the unit's 32 k-elements as ONE step of
          // v_mfma_f32_16x16x32_bf16 — lane -> row (lane & 15) of a 16-row fragment, 16-byte chunk 4 * half + (lane >> 4); the
          // swizzle (row >> 1) & 7 keeps every 16-lane group of the ds_read_b128 on 16 distinct 16-byte slots.  12 reads per unit, as before.
          {
            const int l15 = lane & 15, lq = lane >> 4;
            const int o = ((4 * half + lq) ^ ((l15 >> 1) & 7)) << 4;
            const char* Bs16 = smem + sb + (wn * 64 + l15) * 128 + o;
            const char* As16 = smem + sa + (wm * 128 + l15) * 128 + o;
#pragma unroll
            for (int i = 0; i < 4; ++i) fb[i >> 1][i & 1] = *(const bf16x8*)(Bs16 + i * 2048);
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[i >> 2][i & 3] = *(const bf16x8*)(As16 + i * 2048);
          }
#else
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const int o = t16[2 * half + s2];
#pragma unroll
            for (int i = 0; i < C::TN; ++i) fb[s2][i] = *(const bf16x8*)(Bs + o + i * 4096);
#pragma unroll
            for (int i = 0; i < C::TM; ++i) fa[s2][i] = *(const bf16x8*)(As + o + i * 4096);
          }
#endif
          }
        }
        if (PP_NL > 0 && dma_on && early_ok) {
#pragma unroll
          for (int i = 0; i < PP_NL; ++i) piece(i);
        }
        if constexpr (SLOT != 0) {
          // the slot: behind the fragment reads (their LDS latency runs meanwhile), g0 in LOAD(kt,1), g1 in LOAD(kt,0) — in both
          // groups that is the LOAD phase in FRONT of the unit that carries the A(kt+2) pieces
          if ((half == 1) != (wm != 0)) {
            __builtin_amdgcn_sched_barrier(0);
            dg_slot();
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (half == 0 && kt < 6) TSTAMP(2 + 6 * kt);
        if (half == 1 && wm) {      // g1: end of phase 4kt+3
          if constexpr (SLOT != 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(4 + SLOT_OPS) : "memory");
          else if (a_on) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
          else      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (half == 0 && kt < 6) TSTAMP(3 + 6 * kt);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (half == 0 && kt < 6) TSTAMP(4 + 6 * kt);
        // ---------------- COMPUTE (+ the remaining DMA pieces behind MFMA pairs 0, 2, 4, 6)
        if constexpr (F8) {
#pragma unroll
          for (int a = 0; a < C::TM; ++a) {       // 8 MFMAs of 64 cycles; a piece behind every second one (= the bf16 cadence)
#pragma unroll
            for (int b = 0; b < C::TN; ++b)
              acc[a][b] = SWAP ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(gb[b], ga[a], acc[a][b], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f)
                               : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ga[a], gb[b], acc[a][b], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            __builtin_amdgcn_sched_barrier(0);
            if (dma_on && !(PP_NL > 0 && early_ok && a < PP_NL)) piece(a);
            __builtin_amdgcn_sched_barrier(0);
          }
          // Pin the accumulators HERE: the scaled MFMA is a register-only instruction without side effects, and without an
          // ordered use hipcc's machine sinking moved this unit's eight MFMAs down behind the next barrier pair, next to
          // the following unit's (16 MFMAs in one phase, the partner idle, both units' fragments live: 68 spilled VGPRs
          // whose scratch reloads sat in front of the LDS-DMA issues with a vmcnt(0) each).  An empty volatile asm that
          // reads and writes each accumulator keeps the producers above it; volatile asms keep their order with the barriers.
#pragma unroll
          for (int a = 0; a < C::TM; ++a)
#pragma unroll
            for (int b = 0; b < C::TN; ++b) asm volatile("" : "+v"(acc[a][b]));
        } else
#if GEMM_MFMA16 == 1
        {
          // 32 MFMAs of 16 cycles: 16-row fragment tm of the activations x 16-row fragment tn of the weights; accumulator (tm, tn) lives in
          // quad (tm & 1) * 2 + (tn & 1) of acc[tm >> 1][tn >> 1] (probe: the epilogue's lane map does not apply).  A DMA piece behind
          // every eighth MFMA = the 128-cycle cadence of the 32x32x16 loop.
#pragma unroll
          for (int tm = 0; tm < 8; ++tm) {
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
              f32x16& A_ = acc[tm >> 1][tn >> 1];
              constexpr int dummy = 0; (void)dummy;
              const int j = (tm & 1) * 2 + (tn & 1);
              f32x4 c4; c4.x = A_[4 * j]; c4.y = A_[4 * j + 1]; c4.z = A_[4 * j + 2]; c4.w = A_[4 * j + 3];
              c4 = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[tn >> 1][tn & 1], fa[tm >> 2][tm & 3], c4, 0, 0, 0)
                        : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[tm >> 2][tm & 3], fb[tn >> 1][tn & 1], c4, 0, 0, 0);
              A_[4 * j] = c4.x; A_[4 * j + 1] = c4.y; A_[4 * j + 2] = c4.z; A_[4 * j + 3] = c4.w;
            }
            if (!(tm & 1)) {
              __builtin_amdgcn_sched_barrier(0);
              if (dma_on && !(PP_NL > 0 && early_ok && (tm >> 1) < PP_NL)) piece(tm >> 1);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
#else
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int a = 0; a < C::TM; ++a) {
#pragma unroll
            for (int b = 0; b < C::TN; ++b)     // operands swapped: D rows <-> n (weight rows), D columns <-> m
              acc[a][b] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[s2][b], fa[s2][a], acc[a][b], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s2][a], fb[s2][b], acc[a][b], 0, 0, 0);
            const int pair = s2 * 4 + a;        // 8 MFMA pairs per unit; a piece behind pairs 0, 2, 4, 6: the last one
            if (!(pair & 1)) {                  // still has MFMAs behind it
              __builtin_amdgcn_sched_barrier(0);
              if (dma_on && !(PP_NL > 0 && early_ok && (pair >> 1) < PP_NL)) piece(pair >> 1);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
#endif
        if (half == 0 && kt < 6) TSTAMP(5 + 6 * kt);
        if (half == 1 && !wm) {     // g0: end of phase 4kt+3
          if constexpr (SLOT != 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 + SLOT_OPS) : "memory");
          else if (a_on) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          else      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      sa = sa1; sb = sb1;
    };
    const int cur_m0 = m0, cur_n0 = n0;                        // this tile's origin (m0 / n0 move on to the next tile below)
    const int next = next_of(tile);
    tile_adv = next;
    const bool has_next = stream && next < ntiles;
    {
      using T_ = std::true_type; using F_ = std::false_type;
      int kt = 0;
      using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
      if constexpr (DG) {      // K-tile 0: no slot (step 0 is still on its way); K-tiles 1 .. 16: one step each
        if (kt + 2 < nk) { ktile(kt, wm ? kt + 2 : kt + 1, kt + 2, T_{}, 7, S0{}); ++kt; }
        for (; kt + 2 < nk && kt <= DG_STEPS; ++kt) ktile(kt, wm ? kt + 2 : kt + 1, kt + 2, T_{}, 7, S1{});
      }
      for (; kt + 2 < nk; ++kt) ktile(kt, wm ? kt + 2 : kt + 1, kt + 2, T_{}, 7, S0{});
      if (kt + 1 < nk) {
        // K-tile nk-2: g0's half 0 still carries this tile's B(nk-1); everything "two ahead" is the next tile's K-tile 0
        int k0 = wm ? kt + 2 : kt + 1, k1 = kt + 2, fl = 1;
        if (has_next) {
          const char* const cur_w = w_tile;
          set_tile(next);                                        // a_src, w_src, a_tile, w_tile, m0, n0 := the next tile's
#pragma unroll
          for (int i = 0; i < 4; ++i) { ride0[i] = wm ? a_src[i] : w_src[i]; ride1[i] = wm ? w_src[i] : a_src[i]; }
          ride_base0 = wm ? a_tile : cur_w;
          ride_base1 = wm ? w_tile : a_tile;
          k0 = wm ? 0 : kt + 1; k1 = 0; fl = 7;
        }
        ktile(kt, k0, k1, F_{}, fl, S0{});
        ++kt;
      }
      if (kt < nk) {
        // K-tile nk-1: g0 carries B'(0) and A'(1), g1 A'(1); g1's half-1 unit (it would be B'(1), into THIS K-tile's B slot)
        // stays empty — that slot and this K-tile's A slot are the epilogue's staging area; g1 fetches B'(1) at the tile start
        if (has_next) ride_base0 = wm ? a_tile : w_tile;
        ktile(kt, wm ? 1 : 0, 1, F_{}, has_next ? 3 : 0, S0{});
      }
    }
    TSTAMP(60);
    if constexpr (LNV > 0) {
      // The main loop is over: every wave of the workgroup passed this loop's counted waits long ago, and the first of them retired
      // the PREVIOUS tile's last stores (they are older than every piece of this loop) — that tile's rows are in the L2.  Wave 4
      // sends ONE returning atomic: lane 0 claims a queue position (the tile after the next one), lane 1 counts the previous
      // tile's arrival at its panel.  A compiler-visible atomic: hipcc waits for the result where it is first used — at the
      // end of this epilogue, microseconds later — and every vector-memory operation younger than it is the epilogue's own, so
      // the wait it computes is exact and drains nothing of the next tile's stream (those pieces are older).
      if (wave == 4) {
        int ls = lane;
        asm volatile("" : "+v"(ls));
        if (ls < 2) {
          const bool arrive = ls == 1 && lnq_prev_panel >= 0;
          unsigned int* const ap = arrive ? g.ln_sync + 16 + lnq_prev_panel : lnq_queue + xcc;
          lnq_res = __hip_atomic_fetch_add(ap, (ls == 0 || arrive) ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }

    // ---------------------------------------------------------------- epilogue
    // acc[a][b][4q + r]: row m = row0 + a*32 + l31 ; column n = col0 + b*32 + 8q + 4*lh + r
    const int row0 = cur_m0 + wm * 128, col0 = cur_n0 + wn * 64;
    const bool full = (cur_m0 + C::BM <= g.M) && (cur_n0 + C::BN <= g.N);    // workgroup-uniform
    const bool wave_cols_ok = col0 < g.N;
    // staging area of the epilogue = the LAST K-tile's two slots (its reads are over; the next tile's first K-tiles go
    // to the other three): waves 0-3 in its B slot, waves 4-7 in its A slot.  With nk a multiple of 6 that is B1 | A2.
    const int stg_b = sb ^ (C::B0 ^ C::B1);
    const int stg_a = sa == C::A0 ? C::A2 : (sa == C::A1 ? C::A0 : C::A1);
    if (next < ntiles) {
      if (!has_next) {            // nk == 1: no K-tile to ride on — the next tile's first panels go out here, ring-relative
        set_tile(next);
        stage_b(sb, 0);
        stage_a(sa, 0);
      }
      bias_next = g.bias[n0 + wn * 64 + lane];
      if (F8) wsc_next = g.wscale[n0 + wn * 64 + lane];
      load_asc(m0);
    }
    TSTAMP(43);
    // Lane-derived epilogue constants are recomputed per tile behind an opaque copy of the lane id: hoisted out of the
    // tile loop they would sit in VGPRs across the main loop, which lives at the 256-register limit (a spill reload
    // is a VMEM operation and would break the counted vmcnt waits below).
    int le = lane;
    asm volatile("" : "+v"(le));
    const int l31e = le & 31, lhe = le >> 5;
    // The 64 biases (and fp8 scales) of this wave's columns, four consecutive ones per (b, q): through the wave's own
    // staging area — one ds_write_b32 per lane, eight ds_read_b128 — instead of 32 ds_bpermute shuffles (1.06 k cycles of
    // every wave's epilogue under the stamps, a fifth of the bf16 epilogue).
    char* const stg = smem + (wave < 4 ? stg_b : stg_a) + (wave & 3) * 8192;     // wave-private
    float bb[2][4][4];
    float sc[2][4][4];                                 // F8: the columns' dequantisation scales, same lane -> column map
    if constexpr (SWAP) {
      lds_st_b32(stg + le * 4, __builtin_bit_cast(uint32_t, bias_lane));
      if constexpr (F8) lds_st_b32(stg + 256 + le * 4, __builtin_bit_cast(uint32_t, wsc_lane));
      WAITL0();
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = *(const f32x4*)(stg + (b * 32 + 8 * q + 4 * lhe) * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) bb[b][q][r] = v[r];
          if constexpr (F8) {
            const f32x4 w = *(const f32x4*)(stg + 256 + (b * 32 + 8 * q + 4 * lhe) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[b][q][r] = w[r];
          }
        }
      WAITL0();                                        // the values are in registers before the tile data overwrites the area
    }
    // pre-activation of element (a, b, 4q + r) before the bf16 rounding: acc + bias, or acc * wscale + bias on fp8 operands
    auto pre_f32 = [&](int a, int b, int q, int r) -> float {
      if constexpr (F8) return fmaf(acc[a][b][4 * q + r] * asc[a], sc[b][q][r], bb[b][q][r]);     // asc = 1 without activation scales: exact
      else return acc[a][b][4 * q + r] + bb[b][q][r];
    };
    counted = full && SWAP;
    // g0 is one barrier short of g1 (the skew): it pays it here, a few hundred cycles into its epilogue, while g1
    // issues its last 16 MFMAs.
    TSTAMP(44);
    if (!wm) asm volatile("s_barrier" ::: "memory");
    TSTAMP(45);

    if constexpr (!SWAP) {
      // ---- fc1 + GELU + stage-1 score, plain layout: acc[a][b][i]: row = row0 + a*32 + (i&3) + 8*(i>>2) + 4*lh,
      // column = col0 + b*32 + l31.  Same arithmetic, in the same order, as gemm_bf16_kernel<EPI_FC1, SCORE>.
      const float bias2[2] = {__shfl(bias_lane, l31e), __shfl(bias_lane, 32 + l31e)};
      const float scale2[2] = {F8 ? __shfl(wsc_lane, l31e) : 1.f, F8 ? __shfl(wsc_lane, 32 + l31e) : 1.f};
      const int m128 = row0;                           // this wave's 128-row tile (= one tile of the 128x128 kernel); m0 already names the NEXT tile
      int bnd = 1 << 30, row_lim = g.M - m128;         // rows of the tile at or past row_lim never count
      {
        int ml0 = m128;
        if (g.group > 0) {
          const int sbk = m128 / g.mpad;
          ml0 = m128 - sbk * g.mpad;
          const int imgs = min(g.group, g.n_img - sbk * g.group);
          row_lim = imgs * g.tokens - ml0;
        }
        bnd = (ml0 / g.tokens + 1) * g.tokens - ml0;   // first row (inside the tile) that belongs to the NEXT sample
      }
      float ssq[2][2][2] = {{{0.f, 0.f}, {0.f, 0.f}}, {{0.f, 0.f}, {0.f, 0.f}}};     // [pass h][segment][b]
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // fp8 activation scales of the pass's rows (plain order: register i of sub-tile a2 is row a2*32 + (i&3) + 8*(i>>2) + 4*lh)
        float arow[2][16];
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            arow[a2][i] = 1.f;
            if constexpr (F8) { if (g.ascale) arow[a2][i] = g.ascale[min(row0 + h * 64 + a2 * 32 + (i & 3) + 8 * (i >> 2) + 4 * lhe, g.M - 1)]; }
          }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const float bias = bias2[b];
          const float scl = scale2[b];
          char* const stc = stg + (b * 32 + l31e) * (F8 ? 1 : 2);
#pragma unroll
          for (int a2 = 0; a2 < 2; ++a2) {
            const int a = 2 * h + a2;
            const int rb = h * 64 + a2 * 32;           // does this 32-row sub-tile lie wholly inside one sample and inside M?
            const bool plain = (rb + 32 <= bnd || rb >= bnd) && (rb + 32 <= row_lim);
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
              const int rw = a2 * 32 + (i & 3) + 8 * (i >> 2) + 4 * lhe;     // row inside the 64-row pass (i even: rw, rw+1)
              const uint32_t pk = F8 ? pack_bf16x2(fmaf(acc[a][b][i] * arow[a2][i], scl, bias), fmaf(acc[a][b][i + 1] * arow[a2][i + 1], scl, bias))
                                     : pack_bf16x2(acc[a][b][i] + bias, acc[a][b][i + 1] + bias);
              f32x2 pre;
              const f32x2 gl = gelu_erf_pk(pk, pre);
              const uint32_t o = pack_bf16x2(gl.x, gl.y);
              f32x2 sv = pre;
              if (SCORE == 2) { sv.x = bf16lo_f32(o); sv.y = bf16hi_f32(o); }
              if (plain) {
                s0 = fmaf(sv.x, sv.x, s0); s0 = fmaf(sv.y, sv.y, s0);
              } else {                                   // sub-tile straddles two samples or the end of the slab / of M
                const int r0 = h * 64 + rw;
                const float q0 = (r0 < row_lim) ? sv.x * sv.x : 0.f, q1 = (r0 + 1 < row_lim) ? sv.y * sv.y : 0.f;
                if (r0 < bnd) s0 += q0; else s1 += q0;
                if (r0 + 1 < bnd) s0 += q1; else s1 += q1;
              }
              if constexpr (F8) {                         // e4m3 bytes of the bf16 activation, 64-byte staging rows
                const uint32_t e = pack_e4m3x4(bf16lo_f32(o), bf16hi_f32(o), 0.f, 0.f);
                lds_st_b8(stc + rw * 64, e);
                lds_st_b8(stc + (rw + 1) * 64, e >> 8);
              } else {
                const bf16x2 ov = __builtin_bit_cast(bf16x2, o);
                (void)ov;
                lds_st_b16(stc + rw * 128, o);
                lds_st_b16(stc + (rw + 1) * 128, o >> 16);
              }
            }
            if (plain && rb >= bnd) { ssq[h][1][b] += s0; } else { ssq[h][0][b] += s0; ssq[h][1][b] += s1; }
          }
        }
        WAITL0();
        if (wave_cols_ok) {
          if constexpr (F8) {                             // 64 rows x 64 bytes: 4 sixteen-byte chunks per row
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              const int r = it * 16 + (le >> 2), c = (le & 3) * 16;
              const i32x4 v = *(const i32x4*)(stg + r * 64 + c);
              const int m = row0 + h * 64 + r;
              if (m < g.M) { if (g.nt_out) __builtin_nontemporal_store(v, (i32x4*)((char*)g.out + (size_t)m * g.ldo + col0 + c)); else *(i32x4*)((char*)g.out + (size_t)m * g.ldo + col0 + c) = v; }
            }
          } else {
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int r = it * 8 + (le >> 3), c = (le & 7) * 8;
            const bf16x8 v = *(const bf16x8*)(stg + r * 128 + c * 2);
            const int m = row0 + h * 64 + r;
            if (m < g.M) { if (g.nt_out) __builtin_nontemporal_store(v, (bf16x8*)(g.out + (size_t)m * g.ldo + col0 + c)); else *(bf16x8*)(g.out + (size_t)m * g.ldo + col0 + c) = v; }
          }
          }
        }
        WAITL0();
      }
      // lane l31 of both halves holds the same column: fold the halves of each pass, then the two passes
#pragma unroll
      for (int sgm = 0; sgm < 2; ++sgm)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          float t0 = ssq[0][sgm][b], t1 = ssq[1][sgm][b];
          t0 += __shfl_xor(t0, 32);
          t1 += __shfl_xor(t1, 32);
          const int col = col0 + b * 32 + l31e;
          if (lhe == 0 && col < g.slab_ld) g.slab[((size_t)(m128 >> 7) * 2 + sgm) * g.slab_ld + col] = t0 + t1;
        }
    } else
    if (EPI == EPI_FC1 && F8) {   // fc1 on fp8 operands: bias + erf-GELU on the bf16 pre-activation, e4m3 bytes out (for the fp8 fc2)
      // two passes of 64 rows x 64 bytes (4 KiB): a lane's 4 consecutive columns are one dword; 16-byte chunk c of row r sits
      // at chunk c ^ ((r >> 1) & 3)
      if (wave_cols_ok) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int a = 2 * h + a2;
                float gv[4];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                  const uint32_t pk = pack_bf16x2(pre_f32(a, b, q, 2 * p), pre_f32(a, b, q, 2 * p + 1));
                  f32x2 pre;
                  const f32x2 gl = gelu_erf_pk(pk, pre);
                  const uint32_t o = pack_bf16x2(gl.x, gl.y);
                  gv[2 * p] = bf16lo_f32(o); gv[2 * p + 1] = bf16hi_f32(o);
                }
                const int r = a2 * 32 + l31e, chunk = 2 * b + (q >> 1);
                lds_st_b32(stg + r * 64 + ((chunk ^ ((r >> 1) & 3)) << 4) + 8 * (q & 1) + 4 * lhe, pack_e4m3x4(gv[0], gv[1], gv[2], gv[3]));
              }
          WAITL0();
          i32x4 v[4];
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int r = it * 16 + (le >> 2), c = le & 3;
            v[it] = *(const i32x4*)(stg + r * 64 + ((c ^ ((r >> 1) & 3)) << 4));
          }
          WAITL0();
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int r = it * 16 + (le >> 2), c = le & 3;
            const int m = row0 + h * 64 + r;
            if (m < g.M) { if (g.nt_out) __builtin_nontemporal_store(v[it], (i32x4*)((char*)g.out + (size_t)m * g.ldo + col0 + c * 16)); else *(i32x4*)((char*)g.out + (size_t)m * g.ldo + col0 + c * 16) = v[it]; }
          }
        }
      }
    } else
    if (EPI == EPI_BF16 || EPI == EPI_FC1) {   // EPI_FC1 here = bias + erf-GELU, no scoring (evaluation passes)
      // two passes of 64 rows x 64 columns bf16 (8 KiB): 16-byte chunk c of row r sits at chunk c ^ (r & 7)
      if (wave_cols_ok) {
        char* const wr_lane = stg + l31e * 128 + lhe * 8;                       // + a2*4096 + swizzled chunk
        const char* const rd_lane = stg + (le >> 3) * 128 + (((le & 7) ^ ((le >> 3) & 7)) << 4);   // + it*1024 (8 rows: same r & 7)
        bf16* const out_lane = g.out + (size_t)(row0 + (le >> 3)) * g.ldo + col0 + (le & 7) * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int a = 2 * h + a2;
                uint32_t pk[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                  pk[p] = pack_bf16x2(pre_f32(a, b, q, 2 * p), pre_f32(a, b, q, 2 * p + 1));
                  if (EPI == EPI_FC1) {
                    f32x2 pre;
                    const f32x2 gl = gelu_erf_pk(pk[p], pre);
                    pk[p] = pack_bf16x2(gl.x, gl.y);
                  }
                }
                uint2 v; v.x = pk[0]; v.y = pk[1];
                lds_st_b64(wr_lane + a2 * 4096 + (((b * 4 + q) ^ (l31e & 7)) << 4), v);
              }
          if (h == 0) TSTAMP(46);
          WAITL0();
          bf16x8 v[8];
#pragma unroll
          for (int it = 0; it < 8; ++it) v[it] = *(const bf16x8*)(rd_lane + it * 1024);
          WAITL0();
          if (h == 0) TSTAMP(47);
          if (g.nt_out) {           // (wave-uniform) non-temporal: the tile is read next by ANOTHER kernel, it need not stay in this XCD's L2
#pragma unroll
            for (int it = 0; it < 8; ++it) {
              bf16* dst = out_lane + (size_t)(h * 64 + it * 8) * g.ldo;
              if (full || row0 + h * 64 + it * 8 + (le >> 3) < g.M) __builtin_nontemporal_store(v[it], (bf16x8*)dst);
            }
          } else {
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            bf16* dst = out_lane + (size_t)(h * 64 + it * 8) * g.ldo;
            if (full) ST_OUT((bf16x8*)dst, v[it]);
            else if (row0 + h * 64 + it * 8 + (le >> 3) < g.M) ST_OUT((bf16x8*)dst, v[it]);
          }
          }
          if (h == 0) TSTAMP(48);
        }
      }
      // full path: the last 8 operations of this wave are the second pass's stores; the next main loop starts behind
      // vmcnt(4), i.e. it waits for everything but the youngest four of them
    } else {   // EPI_RESID: x += float(bf16(acc + bias))
      // The fp32 x tile makes this epilogue a memory-latency problem: 32 KiB per wave must come in and go out.  All the
      // loads that fit in registers go out AT ONCE, coalesced (a block p = (a, b) of 32 rows x 32 fp32 columns is 4
      // loads of 8 rows x 128 B: lane -> row 8j + dr, 16-byte chunk le & 7), NXB = 4 blocks = 64 VGPRs (the fragment
      // and bias registers are dead by then), block p + 4 as soon as block p has been stored.
      // The accumulators meet that layout through a wave-private LDS buffer: written fragment-shaped (lane -> row
      // l31, chunk 2q + lh, swizzled by (row >> 1) & 7: conflict-free), read back row-contiguous.
      // (Round-1 form: two x blocks in flight per wave through LDS-DMA; every pass then waited ~2.3 k cycles for
      // its block, 20.7 k cycles per tile.)
      constexpr int NXB = 4;
      if constexpr (DG) {
        // (1) what the main loop left of the PREVIOUSLY parked tile (K = 768: 10 of its 16 steps ride on K-tiles; K >= 1152: all of them).
        //     The step in the registers has landed: the closing wait of K-tile nk - 2 retired everything older than its own pieces.
        dg_drain(true);
        // (2) park this tile: bf16(acc + bias), row-contiguous through the wave's LDS cells exactly as the direct form reads them,
        //     16 x 1 KiB lane-linear stores (the lane that stores a 16-byte piece is the lane that will load it)
        {
          const int dr = le >> 3, cc = le & 7;
          char* const cell_lane = stg + l31e * 128;
          const int csw = (l31e >> 1) & 7;
          const char* const rd_lane = stg + dr * 128;
          const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(dg_scratch(), 0, 0x7ffffff0, 0x00020000);
          const int l16 = le * 16;
#pragma unroll
          for (int p = 0; p < 8; ++p) {
            const int a = p >> 1, b = p & 1;
            const int buf = (p & 1) * 4096;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              f32x4 v; v.x = pre_f32(a, b, q, 0); v.y = pre_f32(a, b, q, 1); v.z = pre_f32(a, b, q, 2); v.w = pre_f32(a, b, q, 3);
              lds_st_b128(cell_lane + buf + (((2 * q + lhe) ^ csw) << 4), v);
            }
            WAITL0();
            f32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *(const f32x4*)(rd_lane + buf + j * 1024 + ((cc ^ (((8 * j + dr) >> 1) & 7)) << 4));
            WAITL0();
            dg_u32x4 w0, w1;
            w0.x = pack_bf16x2(v[0].x, v[0].y); w0.y = pack_bf16x2(v[0].z, v[0].w); w0.z = pack_bf16x2(v[1].x, v[1].y); w0.w = pack_bf16x2(v[1].z, v[1].w);
            w1.x = pack_bf16x2(v[2].x, v[2].y); w1.y = pack_bf16x2(v[2].z, v[2].w); w1.z = pack_bf16x2(v[3].x, v[3].y); w1.w = pack_bf16x2(v[3].z, v[3].w);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_raw_buffer_store_b128(w0, rsc, l16, (2 * p) * 1024, 0);
            __builtin_amdgcn_raw_buffer_store_b128(w1, rsc, l16, (2 * p + 1) * 1024, 0);
            asm volatile("s_nop 1" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        // (3) the parked tile is now this one (its step 0 goes out at the top of the next tile, or in the drain behind the last one)
        dg_pm0 = cur_m0; dg_pn0 = cur_n0; dg_prows = g.M - row0; dg_s = 0;
      } else
      if (wave_cols_ok) {
        const int dr = le >> 3, cc = le & 7;
        float* const xw = g.x + (size_t)row0 * g.ldx + col0;         // wave's 128 x 64 window of x
        // bias + bf16 rounding in place first (frees the 32 bias registers before the x registers are taken)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const uint32_t p0 = pack_bf16x2(pre_f32(a, b, q, 0), pre_f32(a, b, q, 1));
              const uint32_t p1 = pack_bf16x2(pre_f32(a, b, q, 2), pre_f32(a, b, q, 3));
              acc[a][b][4 * q] = bf16lo_f32(p0); acc[a][b][4 * q + 1] = bf16hi_f32(p0);
              acc[a][b][4 * q + 2] = bf16lo_f32(p1); acc[a][b][4 * q + 3] = bf16hi_f32(p1);
            }
        char* const cell_lane = stg + l31e * 128;
        const int csw = (l31e >> 1) & 7;
        const char* const rd_lane = stg + dr * 128;                  // + j*1024 + swizzled chunk (rows 8j + dr)
        // x is addressed as a raw buffer: SGPR descriptor + SGPR block offset + ONE per-lane VGPR offset, so the 16
        // loads in flight and their 16 stores need no 64-bit address registers (32 VGPRs the allocator does not
        // have here).  Rows past M get a per-lane offset beyond the buffer's range (the range check sees the VGPR
        // offset): their loads return 0 and their stores are dropped by the hardware — edge tiles take the same
        // straight-line code, no branches, no exec masks.
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(xw, 0, 0x7ffffff0, 0x00020000);   // base = the wave's window
        // the residual may be READ from another stream of the same shape (g.xin: the search's candidate l takes block l's input
        // from the baseline's slot and writes its own, csrc/engine.hip ssp2_layers_from): a second descriptor, 4 SGPRs
        float* const xw_in = const_cast<float*>(g.xin ? g.xin : g.x) + (size_t)row0 * g.ldx + col0;
        const __amdgpu_buffer_rsrc_t xr_in = __builtin_amdgcn_make_buffer_rsrc(xw_in, 0, 0x7ffffff0, 0x00020000);
        const int lane_off = (dr * g.ldx + cc * 4) * 4;              // bytes: row dr of a piece, 16-byte chunk cc
        auto xoff = [&](int a, int b, int j) -> int {                // bytes, wave-uniform
          return ((a * 32 + 8 * j) * g.ldx + b * 32) * 4;
        };
        const int rows_left = g.M - row0 - dr;                       // row a*32 + 8j + dr is valid iff a*32 + 8j < rows_left
        auto voff = [&](int a, int j) -> int { return (a * 32 + 8 * j < rows_left) ? lane_off : 0x7fffffff; };
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
        // Cache policy of the x tile (aux operand of the raw-buffer instructions: bit 1 = nt).  GEMM_XNT: bit 0 = loads non-temporal,
        // bit 1 = stores.  Default 1: the tile is READ non-temporally — it is read exactly once here and rewritten.  On launches whose x
        // exceeds the 256 MiB Infinity Cache (the layer-major search: 630 400 rows = 1.9 GB) that is + 8.8 % on the out-projection
        // (1131 -> 1040 us sustained), nothing either way at 63 040 rows; non-temporal STORES of x do nothing, both together lose
        // (profiles/r04_h_resid_policy.jsonl).  Compile time only: a run-time choice between the two load forms spilled 379 registers.
#ifndef GEMM_XNT
#define GEMM_XNT 1
#endif
        constexpr int XLD_AUX = (GEMM_XNT & 1) ? 2 : 0, XST_AUX = (GEMM_XNT & 2) ? 2 : 0;
        auto xload = [&](int vo, int so) __attribute__((always_inline)) -> f32x4 {
          return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr_in, vo, so, XLD_AUX));
        };
        f32x4 xin[NXB][4];
#pragma unroll
        for (int p = 0; p < NXB; ++p)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            xin[p][j] = xload(voff(p >> 1, j), xoff(p >> 1, p & 1, j));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int a = p >> 1, b = p & 1;
          const int buf = (p & 1) * 4096;                            // two buffers: block p+1 is written while p's reads land
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v; v.x = acc[a][b][4 * q]; v.y = acc[a][b][4 * q + 1]; v.z = acc[a][b][4 * q + 2]; v.w = acc[a][b][4 * q + 3];
            lds_st_b128(cell_lane + buf + (((2 * q + lhe) ^ csw) << 4), v);
          }
          WAITL0();
          f32x4 v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = *(const f32x4*)(rd_lane + buf + j * 1024 + ((cc ^ (((8 * j + dr) >> 1) & 7)) << 4));
          WAITL0();
          // all four sums first, into four different register quads, THEN the four stores: when hipcc re-used one quad
          // (add, store, add, store, ...) the last dword of the 16-byte store data came out overwritten by the next
          // add on lanes 12-15 / 28-31 / 44-47 / 60-63 — a store-data hazard it does not pad on gfx950
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += xin[p % NXB][j];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[j]), xr, voff(a, j), xoff(a, b, j), XST_AUX);
          // ... and two wait states before anything may rewrite the stores' data registers (16-byte store data with an
          // SGPR offset: hipcc / ROCm 7.2 does not pad this case on gfx950, see above) — an instruction-order-independent
          // guard, so that a scheduling change in a later compiler cannot bring the corruption back
          asm volatile("s_nop 1" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          if (p + NXB < 8) {                   // block p + NXB takes the registers of block p (pinned: hipcc would
            __builtin_amdgcn_sched_barrier(0); // hoist these loads to the top and spill their destinations)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              xin[p % NXB][j] = xload(voff((p + NXB) >> 1, j), xoff((p + NXB) >> 1, (p + NXB) & 1, j));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    if constexpr (LNV > 0) {
      // wave 4, done with its own staging area, leaves the two words the next tile's top reads (see there)
      if (wave == 4) {
        lnq_claimed = __builtin_amdgcn_readlane(lnq_res, 0);
        lnq_cnt = __builtin_amdgcn_readlane(lnq_res, 1);
        const int flagged = (lnq_prev_panel >= 0 && lnq_cnt == g.tiles_n - 1) ? lnq_prev_panel : -1;
        if (lane == 0) { lds_st_b32(stg, (uint32_t)flagged); lds_st_b32(stg + 4, (uint32_t)lnq_claimed); }
      }
      lnq_slot = stg_a;
      lnq_prev_panel = cur_m0 >> 8;
    }
    TSTAMP(40);
  }
  if constexpr (DG) {
    if (tiles_done > 0) dg_drain(false);           // the last tile's residual: nothing rides on anything any more
  }
  if constexpr (LNV > 0) {
    // The queue is empty.  Two panels may still wait for this workgroup: the one its second-to-last tile belongs to (arrival
    // counted during the last main loop, result in wave 4's words) and the one of its last tile, whose arrival is sent here,
    // behind a full drain of every wave's stores.
    if (tiles_done > 0) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const int first = __builtin_amdgcn_readfirstlane(*(volatile uint32_t*)(smem + lnq_slot));
      if (tid == 256) {
        const uint32_t before = atomicAdd(g.ln_sync + 16 + lnq_prev_panel, 1u);
        *(volatile uint32_t*)(smem + lnq_slot + 8) = before == (uint32_t)(g.tiles_n - 1) ? (uint32_t)lnq_prev_panel : 0xffffffffu;
      }
      __syncthreads();
      const int second = __builtin_amdgcn_readfirstlane(*(volatile uint32_t*)(smem + lnq_slot + 8));
      if (first >= 0) ln_phase(first);
      if (second >= 0) ln_phase(second);
    }
  }
  STAMP(59);
  RSTAMP(62);
#ifdef GEMM_STAMPS
  if (tid == 0) g.stamps[(size_t)blockIdx.x * 64 + 63] = tiles_done;
#endif
}
