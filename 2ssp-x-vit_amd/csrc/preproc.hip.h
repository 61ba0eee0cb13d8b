// GPU input pipeline: uint8 HWC images -> resize (bicubic) -> [optional horizontal flip] ->
// ToTensor (/255) -> Normalize((x - mean) / std) -> fp32 NCHW, i.e. the reference's torchvision chain
// (adaptation-for-Pures-framework/auto_2ssp.py:290-301: Resize(BICUBIC) on a PIL image, RandomHorizontalFlip, ToTensor,
// Normalize).  torchvision's Resize on PIL images is `Image.resize`, whose 8-bit resampler is integer arithmetic:
// two separable passes (horizontal, then vertical), per-output-pixel windows [xmin, xmin+count), coefficients
// normalised in double and rounded to 22-bit fixed point, an 8-bit rounded/clipped intermediate image.  The same
// integers are computed here, so the uint8 result is bit-identical to Pillow's (checked against Pillow 12 goldens)
// and the fp32 result bit-identical to ToTensor+Normalize.
#pragma once
#include "common.hip.h"

#define PREPROC_PRECISION_BITS 22   // Pillow: 32 - 8 - 2

__device__ __forceinline__ uint8_t clip8_fixed(int v) {
  v >>= PREPROC_PRECISION_BITS;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: in [n, h, w, 3] u8 -> tmp [n, h, W, 3] u8
__global__ void resize_h_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ tmp, int n, int h, int w, int W,
                                   const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
  const long total = (long)n * h * W * 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % 3);
    const int xx = (int)((i / 3) % W);
    const long row = i / (3L * W);                       // n*h + y
    const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
    const int* k = kk + xx * ksize;
    const uint8_t* src = in + (row * w + xmin) * 3 + c;
    int ss = 1 << (PREPROC_PRECISION_BITS - 1);
    for (int x = 0; x < cnt; ++x) ss += (int)src[3 * x] * k[x];
    tmp[i] = clip8_fixed(ss);
  }
}

// vertical pass + flip + ToTensor + Normalize: tmp [n, h, W, 3] u8 -> out [n, 3, H, W] f32 (and optional u8 copy)
__global__ void resize_v_norm_kernel(const uint8_t* __restrict__ tmp, float* __restrict__ out, uint8_t* __restrict__ out_u8,
                                     int n, int h, int W, int H, const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                     const uint8_t* __restrict__ hflip, float m0, float m1, float m2, float s0, float s1, float s2) {
  const long total = (long)n * 3 * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xo = (int)(i % W);
    const int yy = (int)((i / W) % H);
    const int c = (int)((i / ((long)W * H)) % 3);
    const int im = (int)(i / (3L * W * H));
    const int xs = (hflip && hflip[im]) ? W - 1 - xo : xo;          // flip of the RESIZED image
    const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
    const int* k = kk + yy * ksize;
    const uint8_t* src = tmp + (((long)im * h + ymin) * W + xs) * 3 + c;
    int ss = 1 << (PREPROC_PRECISION_BITS - 1);
    for (int y = 0; y < cnt; ++y) ss += (int)src[(long)y * W * 3] * k[y];
    const uint8_t v = clip8_fixed(ss);
    if (out_u8) out_u8[(((long)im * H + yy) * W + xo) * 3 + c] = v;
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    out[i] = ((float)v / 255.0f - mean) / sd;                       // ToTensor: .div(255); Normalize: sub, div
  }
}
