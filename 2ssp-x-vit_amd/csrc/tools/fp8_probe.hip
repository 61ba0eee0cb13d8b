// Layout probe for v_mfma_scale_f32_32x32x64_f8f6f4 on gfx950 (no ISA document is at hand): with exact small values it
// determines (1) which output row / column a lane's A / B bytes belong to and that the C/D map is the 32x32 one of the
// bf16 instruction, (3) the format code of cbsz /
// blgp = 0 (e4m3 vs e5m2), (4) that an E8M0 scale byte of 127 is 1.0 and 128 doubles, and which lanes' scale bytes count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// a, b: [64 lanes][32 bytes]; sa, sb: [64 lanes] scale dwords; d: [64][16]
__global__ void one(const unsigned char* a, const unsigned char* b, const int* sa, const int* sb, float* d) {
  i32x8 A, B;
  const int l = threadIdx.x;
  for (int i = 0; i < 8; ++i) { A[i] = ((const int*)a)[l * 8 + i]; B[i] = ((const int*)b)[l * 8 + i]; }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 0, 0, 0, sa[l], 0, sb[l]);
  for (int i = 0; i < 16; ++i) d[l * 16 + i] = c[i];
}
int main() {
  unsigned char *a, *b; int *sa, *sb; float* d;
  CK(hipMalloc(&a, 2048)); CK(hipMalloc(&b, 2048)); CK(hipMalloc(&sa, 256)); CK(hipMalloc(&sb, 256)); CK(hipMalloc(&d, 4096));
  std::vector<unsigned char> ha(2048), hb(2048); std::vector<int> hsa(64, 0x7f7f7f7f), hsb(64, 0x7f7f7f7f); std::vector<float> hd(1024);
  auto run = [&]() -> int {
    CK(hipMemcpy(a, ha.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), 2048, hipMemcpyHostToDevice));
    CK(hipMemcpy(sa, hsa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(sb, hsb.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, a, b, sa, sb, d);
    CK(hipMemcpy(hd.data(), d, 4096, hipMemcpyDeviceToHost));
    return 0;
  };
  // D(row, col) under the bf16 32x32 map: col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)
  auto D = [&](int row, int col) { for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) if ((l & 31) == col && (i & 3) + 8 * (i >> 2) + 4 * (l >> 5) == row) return hd[l * 16 + i]; return -1.f; };
  // (3) format: all bytes 0x3C: e4m3 1.5 * 1.5 * 64 = 144 ; e5m2 1.0 * 1.0 * 64 = 64
  memset(ha.data(), 0x3C, 2048); memset(hb.data(), 0x3C, 2048);
  if (run()) return 1;
  printf("format: all bytes 0x3C -> D[0][0] = %g  (144 = e4m3 x e4m3, 64 = e5m2, 96 = mixed)\n", hd[0]);
  // (1) rows: A lane l all 1.0, B all 1.0 -> which rows are 32 (a lane's 32 k values)
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    memset(ha.data(), 0, 2048); memset(ha.data() + l * 32, 0x38, 32); memset(hb.data(), 0x38, 2048);
    if (run()) return 1;
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) { const float want = (r == (l & 31)) ? 32.f : 0.f; if (D(r, c) != want) { if (bad < 5) printf("  A lane %d: D[%d][%d] = %g, expected %g\n", l, r, c, D(r, c), want); ++bad; } }
  }
  printf("A operand: lane l holds row l & 31 (32 k values), C/D = the bf16 32x32 map: %s\n", bad ? "NO" : "yes");
  bad = 0;
  for (int l = 0; l < 64; ++l) {
    memset(hb.data(), 0, 2048); memset(hb.data() + l * 32, 0x38, 32); memset(ha.data(), 0x38, 2048);
    if (run()) return 1;
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) { const float want = (c == (l & 31)) ? 32.f : 0.f; if (D(r, c) != want) { if (bad < 5) printf("  B lane %d: D[%d][%d] = %g, expected %g\n", l, r, c, D(r, c), want); ++bad; } }
  }
  printf("B operand: lane l holds column l & 31: %s\n", bad ? "NO" : "yes");
  // (2) which A byte pairs with which B byte (same k) is checked end to end by gemm_bench epi 30..33 against a host reference
  // (4) scales: exact values, A = B = all ones
  memset(ha.data(), 0x38, 2048); memset(hb.data(), 0x38, 2048);
  for (int which = 0; which < 2; ++which) {
    std::fill(hsa.begin(), hsa.end(), 0x7f7f7f7f); std::fill(hsb.begin(), hsb.end(), 0x7f7f7f7f);
    if (run()) return 1;
    const float base = hd[0];
    (which ? hsb : hsa)[5] = 0x7f7f7f80;                 // lane 5 (row / col 5, k half 0): byte 0 = 128 -> x2 for its 32 k values
    if (run()) return 1;
    printf("scale_%c: unit scales D = %g; lane 5 byte0 = 128 -> D[5][0] = %g D[0][5] = %g D[0][0] = %g (48 = 32 + 2 x ... expected 96 = 32 + 2*32 on its row/col)\n",
           which ? 'b' : 'a', base, D(5, 0), D(0, 5), D(0, 0));
    (which ? hsb : hsa)[5] = 0x7f7f807f;                 // byte 1 instead: must NOT count with opsel 0
    if (run()) return 1;
    printf("         lane 5 byte1 = 128 (opsel 0) -> D[5][0] = %g D[0][5] = %g\n", D(5, 0), D(0, 5));
    (which ? hsb : hsa)[5] = 0x7f7f7f7f; (which ? hsb : hsa)[37] = 0x7f7f7f80;   // lane 37 = row 5, k half 1
    if (run()) return 1;
    printf("         lane 37 byte0 = 128 -> D[5][0] = %g D[0][5] = %g\n", D(5, 0), D(0, 5));
  }
  return 0;
}
