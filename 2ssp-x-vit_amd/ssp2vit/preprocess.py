"""GPU input pipeline: the reference's torchvision chain
`Resize((S,S), BICUBIC) -> [RandomHorizontalFlip] -> ToTensor -> Normalize(mean, std)`
(adaptation-for-Pures-framework/auto_2ssp.py:290-301) on uint8 images that already sit in HBM.  CIFAR bytes are
3 KB per image; the fp32 224x224 tensor they become is 602 KB — resizing on the device keeps PCIe out of the way."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import Ssp2Error, check


class GpuPreprocessor:
    def __init__(self, in_hw: Sequence[int], out_size: int = 224, mean: Sequence[float] = (0.5, 0.5, 0.5),
                 std: Sequence[float] = (0.5, 0.5, 0.5), device: str | torch.device = "cuda:0"):
        if not torch.cuda.is_available():
            raise Ssp2Error("ssp2vit needs an MI355X (HIP device): there is no CPU fallback in the product path")
        self.device = torch.device(device)
        self.in_h, self.in_w, self.out = int(in_hw[0]), int(in_hw[1]), int(out_size)
        self.lib = _lib.load()
        m = (C.c_float * 3)(*[float(v) for v in mean])
        s = (C.c_float * 3)(*[float(v) for v in std])
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.ssp2_preproc_create(self.in_h, self.in_w, self.out, m, s, C.byref(h)))
        self.h = h

    def __call__(self, images_u8: torch.Tensor, hflip: Optional[torch.Tensor] = None, return_u8: bool = False):
        """images_u8: uint8 [n, H, W, 3] (host or device); hflip: bool/uint8 [n] or None -> f32 [n, 3, S, S] on device."""
        if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or tuple(images_u8.shape[1:]) != (self.in_h, self.in_w, 3):
            raise ValueError(f"images must be uint8 [n,{self.in_h},{self.in_w},3], got {images_u8.dtype} {tuple(images_u8.shape)}")
        img = images_u8.to(self.device, non_blocking=True).contiguous()
        n = img.shape[0]
        flip = None if hflip is None else hflip.to(self.device, torch.uint8).contiguous()
        tmp = torch.empty(n, self.in_h, self.out, 3, dtype=torch.uint8, device=self.device)
        out = torch.empty(n, 3, self.out, self.out, dtype=torch.float32, device=self.device)
        u8 = torch.empty(n, self.out, self.out, 3, dtype=torch.uint8, device=self.device) if return_u8 else None
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
        check(self.lib.ssp2_preproc_run(self.h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream), p(img), n, p(flip),
                                        p(tmp), p(out), p(u8)))
        return (out, u8) if return_u8 else out

    def close(self):
        if getattr(self, "h", None):
            self.lib.ssp2_preproc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
