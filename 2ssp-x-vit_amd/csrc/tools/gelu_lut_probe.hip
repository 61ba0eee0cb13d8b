// Probe (not part of libssp2vit): what would a table-driven erf-GELU cost in the fc1 epilogue?  The pre-activation is a bf16 value, so
// GELU is a function of 16 bits; outside ~3 k inputs it is x, x/2 or -0, and a 6 KiB LDS table over the rest is exact by construction.
// The open question is LDS bank conflicts of 64 data-dependent 2-byte reads per wave-instruction against ~100 VALU cycles per pair.
//   gelu_lut_probe.bin [iters]  ->  cycles per wave-tile (64 pairs per lane = one 128 x 64 wave tile) for the VALU form and the table form,
//   eight waves per CU on all CUs, inputs ~ N(0, sigma) for a few sigma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include <cstring>
static uint32_t bf_host(float f) { uint32_t u; std::memcpy(&u, &f, 4); return (u + 0x7fffu + ((u >> 16) & 1)) >> 16; }
#include "../gemm.hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int LO_EXP = 127 - 9, HI_EXP = 127 + 3;           // table rows: exponents 2^-9 .. 2^2 (|x| in [2^-9, 8)), both signs
constexpr int ROWS = HI_EXP - LO_EXP;                       // 12
constexpr int TABLE = ROWS * 128 * 2;                       // entries (uint16)

__device__ __forceinline__ uint32_t gelu_bits_valu(uint32_t pk) {
  f32x2 pre;
  const f32x2 g = gelu_erf_pk(pk, pre);
  return pack_bf16x2(g.x, g.y);
}

__global__ void fill_table(uint16_t* t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= TABLE) return;
  const int sign = i / (ROWS * 128), r = i % (ROWS * 128);
  const uint32_t bits = (uint32_t)(sign << 15) | (uint32_t)((LO_EXP << 7) + r);
  t[i] = (uint16_t)(gelu_bits_valu(bits) & 0xffffu);
}

// one element: table where |x| in [2^-9, 8), x itself above, x/2 below (exact: one exponent step), -0 * ... for large negative handled by "above": gelu(-8..) = -0
__device__ __forceinline__ uint32_t gelu_lut_elem(uint32_t b16, const uint16_t* lds_t) {
  const uint32_t mag = b16 & 0x7fffu, sign = b16 >> 15;
  const int rel = (int)mag - (LO_EXP << 7);
  const bool in = rel >= 0 && rel < ROWS * 128;
  const int idx = in ? rel + (int)sign * (ROWS * 128) : 0;
  const uint32_t tv = lds_t[idx];
  const uint32_t big = sign ? 0x8000u : b16;                       // x >= 8: x; x <= -8: -0
  const uint32_t small = mag >= 0x0100u ? b16 - 0x0080u : 0u;      // |x| < 2^-9: x / 2 (exponent - 1); subnormal inputs: 0 (probe only)
  return in ? tv : (rel < 0 ? small : big);
}

template <int MODE>
__global__ __launch_bounds__(512) void probe(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, const uint16_t* __restrict__ table, int iters,
                                             unsigned long long* cyc) {
  __shared__ uint16_t t[TABLE];
  for (int i = threadIdx.x; i < TABLE; i += 512) t[i] = table[i];
  __syncthreads();
  uint32_t v[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) v[i] = in[(size_t)(blockIdx.x * 512 + threadIdx.x) * 64 + i];
  uint32_t acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      uint32_t r;
      if (MODE == 0) r = gelu_bits_valu(v[i]);
      else r = gelu_lut_elem(v[i] & 0xffffu, t) | (gelu_lut_elem(v[i] >> 16, t) << 16);
      acc ^= r; v[i] ^= 0x00010001u;                // keep the work inside the loop (the low mantissa bits flip every iteration) without chaining one result into the next input
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 512 + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 50, nb = 256;
  uint16_t* table; CK(hipMalloc(&table, TABLE * 2));
  hipLaunchKernelGGL(fill_table, dim3((TABLE + 255) / 256), dim3(256), 0, 0, table);
  uint32_t *in, *out; unsigned long long* cyc;
  CK(hipMalloc(&in, (size_t)nb * 512 * 64 * 4)); CK(hipMalloc(&out, (size_t)nb * 512 * 4)); CK(hipMalloc(&cyc, nb * 8));
  for (float sigma : {0.25f, 1.0f, 3.0f}) {
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, sigma);
    std::vector<uint32_t> h((size_t)nb * 512 * 64);

    for (auto& w : h) w = bf_host(nd(rng)) | (bf_host(nd(rng)) << 16);
    CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    // exactness of the table form on these inputs
    std::vector<uint32_t> o0((size_t)nb * 512), o1((size_t)nb * 512);
    for (int mode = 0; mode < 2; ++mode) {
      std::vector<unsigned long long> hc(nb);
      for (int rep = 0; rep < 3; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(nb), dim3(512), 0, 0, in, out, table, iters, cyc);
        else hipLaunchKernelGGL(probe<1>, dim3(nb), dim3(512), 0, 0, in, out, table, iters, cyc);
        CK(hipDeviceSynchronize());
      }
      CK(hipMemcpy(hc.data(), cyc, nb * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy((mode ? o1 : o0).data(), out, (size_t)nb * 512 * 4, hipMemcpyDeviceToHost));
      double s = 0; for (auto c : hc) s += (double)c; s /= nb;
      printf("sigma %.2f  %s: %.0f ticks per wave-tile of 64 pairs (wave 0 of each CU, 8 waves per CU, %d iterations)\n", sigma, mode ? "table" : "VALU ", s / iters, iters);
    }
    size_t diff = 0; for (size_t i = 0; i < o0.size(); ++i) diff += o0[i] != o1[i];
    printf("sigma %.2f  xor-accumulators differing between the two forms: %zu of %zu threads (0 = the table reproduced the VALU form on every input drawn)\n", sigma, diff, o0.size());
  }
  return 0;
}
