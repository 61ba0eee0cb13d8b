"""VitEngine: thin object wrapper over the C ABI.  PyTorch is used here only as plumbing — device buffers,
the current HIP stream, and (elsewhere) torch.distributed; every arithmetic step of the hot path runs in
libssp2vit's HIP kernels.  There is no CPU path: constructing an engine without a GPU raises."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib
from ._lib import K_CLASS, OPTIONS, SCORE_CHAIN, SCORE_SITE, T_KINDS, Ssp2Error, VitDesc, check


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class VitEngine:
    def _check(self, rc: int) -> None:
        check(rc, self.lib)

    def __init__(self, weights: Optional[Dict], device: str | torch.device = "cuda:0", max_images: int = 64, *,
                 precision: str = "bf16", lib_variant: Optional[str] = None, _twin_of: Optional["VitEngine"] = None,
                 _d_int: Optional[Sequence[int]] = None):
        """`lib_variant`: a named build of the library instead of the product one — "lab" carries the opt-in kernel forms the product
        does not instantiate (options ln_fusion / defer_resid / group256; ssp2vit/_lib.py VARIANT_FLAGS)."""
        if precision not in ("bf16", "fp8"):
            raise ValueError(f"precision must be 'bf16' or 'fp8', got {precision!r}")
        self.precision = precision
        if _twin_of is not None:          # same architecture, other FFN widths, weights cloned on the device
            src = _twin_of
            weights = dict(depth=src.depth, dim=src.dim, classes=src.classes, img=src.img, patch=src.patch, heads=src.heads,
                           eps=src.eps, **{f"fc1_w.{i}": torch.empty(int(d), 0) for i, d in enumerate(_d_int)})
            device = src.device
        if not torch.cuda.is_available():
            raise Ssp2Error("ssp2vit needs an MI355X (HIP device): there is no CPU fallback in the product path")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise Ssp2Error(f"ssp2vit runs on HIP devices only, got device={device!r}")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.lib_variant = lib_variant if _twin_of is None else _twin_of.lib_variant
        self.lib = _lib.load(variant=self.lib_variant)
        self.depth = int(weights["depth"])
        self.dim = int(weights["dim"])
        self.classes = int(weights["classes"])
        self.img = int(weights["img"])
        self.patch, self.heads, self.eps = int(weights["patch"]), int(weights["heads"]), float(weights.get("eps", 1e-6))
        self.d_int = [int(weights[f"fc1_w.{i}"].shape[0]) for i in range(self.depth)]
        self.score_ld = max((d + 63) // 64 * 64 for d in self.d_int)
        self.max_images = int(max_images)
        self.absent = [bool(weights.get(f"attn_absent.{i}", False)) for i in range(self.depth)]
        arr = (C.c_int32 * self.depth)(*self.d_int)
        desc = VitDesc(self.img, int(weights["patch"]), self.dim, int(weights["heads"]), self.depth, self.classes,
                       float(weights.get("eps", 1e-6)), self.max_images, arr)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            self._check(self.lib.ssp2_create(C.byref(desc), C.byref(h)))
            self.h = h
            self.tokens = self.lib.ssp2_tokens(self.h)
            if _twin_of is None:
                self._load(weights)
                if precision == "fp8":
                    self._bind_stream()
                    self._check(self.lib.ssp2_set_precision(self.h, 1))
            else:
                self._bind_stream()
                self._check(self.lib.ssp2_clone_weights(self.h, _twin_of.h))
                self.absent = list(_twin_of.absent)

    # ------------------------------------------------------------------ weights
    def _load(self, w: Dict) -> None:
        """fp32 tensors in nn.Linear layout -> the engine's bf16 / padded weight image.  The rounding and the padding run
        in a kernel; a tensor that already lives on this GPU (a module moved to the device, as the reference keeps it)
        is read in place, a host tensor costs one copy of its fp32 bytes."""
        self._bind_stream()
        keep, batch = [], []                                     # device tensors: alive until the stream has consumed them; (kind, layer, tensor)

        def put(kind: str, layer: int, t: torch.Tensor):
            t = t.detach()
            if t.device == self.device:
                t = t.to(torch.float32).contiguous()
                keep.append(t)
                batch.append((T_KINDS.index(kind), layer, t))
            else:
                t = t.to("cpu", torch.float32).contiguous()
                self._check(self.lib.ssp2_load_tensor(self.h, T_KINDS.index(kind), layer,
                                                C.cast(t.data_ptr(), C.POINTER(C.c_float)), t.numel()))
        for k in ("patch_w", "patch_b", "cls", "pos", "lnf_g", "lnf_b", "head_w", "head_b"):
            put(k, 0, w[k])
        for i in range(self.depth):
            for k in ("ln1_g", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ln2_g", "ln2_b",
                      "fc1_w", "fc1_b", "fc2_w", "fc2_b"):
                put(k, i, w[f"{k}.{i}"])
        if batch:
            # a live module's ~150 tensors in ceil(count / 64) launches (ssp2_load_tensors_dev, ABI 5) instead of one launch each: 1.5 ms of
            # launches for 0.1 ms of traffic inside the caller's prune bracket (scripts/api_profile.py)
            m = len(batch)
            kinds = (C.c_int * m)(*[b[0] for b in batch])
            layers = (C.c_int * m)(*[b[1] for b in batch])
            ptrs = (C.c_void_p * m)(*[b[2].data_ptr() for b in batch])
            numels = (C.c_size_t * m)(*[b[2].numel() for b in batch])
            self._check(self.lib.ssp2_load_tensors_dev(self.h, m, kinds, layers, ptrs, numels))
            torch.cuda.current_stream(self.device).synchronize()

    def geometry_key(self):
        """What two engines must share for one to take over the other's handle (workspace, weight storage): see `reload`."""
        return (str(self.device), self.precision, self.lib_variant, self.img, self.patch, self.dim, self.heads, self.depth, self.classes,
                float(self.eps), tuple(int(d) for d in self.d_int))

    def reload(self, weights: Dict) -> "VitEngine":
        """Another model of the SAME geometry into this engine: every tensor is ingested again (device-to-device rounding kernels), the
        attention flags are reset to the new model's — handle, weight storage and the (already touched) activation workspace stay.
        What vit_pruning's engine pool does when a sweep builds a fresh module per target: a fresh engine costs its build (2.6 ms for
        the 13 GB layer-major workspace of ViT-B/16) plus ~3 ms of first use of fresh memory, per prune (profiles/r05_m_api_profile.txt)."""
        if self.precision != "bf16":
            raise Ssp2Error("reload: bf16 engines only (an fp8 engine carries quantised weight images and calibrated scales)")
        if [int(weights[f"fc1_w.{i}"].shape[0]) for i in range(self.depth)] != [int(d) for d in self.d_int]:
            raise Ssp2Error("reload: FFN widths differ")
        with torch.cuda.device(self.device):
            self._load(weights)
            for l in range(self.depth):
                gone = bool(weights.get(f"attn_absent.{l}", False))
                self._check((self.lib.ssp2_drop_attention if gone else self.lib.ssp2_restore_attention)(self.h, l))
                self.absent[l] = gone
        return self

    def close(self) -> None:
        if getattr(self, "h", None):
            if getattr(self, "precision", "bf16") == "fp8":
                try:                                              # an fp8 engine that clipped attention outputs says so when it goes away
                    n = self.fp8_saturation()
                    if n > 0:
                        import warnings
                        warnings.warn(f"ssp2vit fp8 engine: {n} wave(s) clipped attention outputs at the e4m3 range of their hand-off scale "
                                      "(|o| > 28 uncalibrated); use calibrate_fp8(...), set_option('fp8_proj', 0) or precision='bf16' for this "
                                      "checkpoint", RuntimeWarning, stacklevel=2)
                except Exception:
                    pass
            self.lib.ssp2_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ applying a prune on the device (f2)
    def prune_ffn(self, layer: int, keep: Sequence[int]) -> None:
        """Keep only the FFN neurons `keep` (ascending) of block `layer`: fc1 rows / bias and fc2 columns are gathered
        in HBM (reference src/vit_pruning.py:297-311 on the module).  No re-upload of weights."""
        arr = (C.c_int32 * len(keep))(*[int(k) for k in keep])
        with torch.cuda.device(self.device):
            self._check(self.lib.ssp2_prune_ffn(self.h, int(layer), arr, len(keep)))
        self.d_int[layer] = len(keep)

    def apply_ffn_masks(self, masks: Sequence[Sequence[int]]) -> None:
        """masks[l][j] == 1 -> prune neuron j of block l (the `ffn_prune_masks` of prune_vit_mlp_width)."""
        for l, m in enumerate(masks):
            keep = [j for j, bit in enumerate(m) if not bit]
            if len(keep) != self.d_int[l]:
                self.prune_ffn(l, keep)

    def drop_attention(self, layers: Sequence[int]) -> None:
        for l in layers:
            self._check(self.lib.ssp2_drop_attention(self.h, int(l)))
            self.absent[int(l)] = True

    def set_cu_limit(self, n_cu: int) -> None:
        """Cap the grids of this engine's persistent kernels at n_cu workgroups (0 = all CUs)."""
        self._check(self.lib.ssp2_set_cu_limit(self.h, int(n_cu)))

    def set_option(self, name: str, value: int) -> None:
        """Run-time switch of this engine (include/ssp2vit.h SSP2_OPT_*: "zigzag", "attn_persist", "ln_fusion", "big_tiles",
        "fc1_big_tiles", "group256").  None of them changes a result bit; they exist for the tests and A/B scripts that prove it."""
        self._check(self.lib.ssp2_set_option(self.h, OPTIONS[name], int(value)))

    def get_option(self, name: str) -> int:
        v = self.lib.ssp2_get_option(self.h, OPTIONS[name])
        if v < 0:
            self._check(v)
        return int(v)

    def calibrate_fp8(self, pixels: torch.Tensor, headroom: float = 4.0) -> List[float]:
        """fp8 mode: measure the attention outputs of `pixels` (one or more chunks of <= max_images images, each launch >= 4096 token
        rows) with the out-projection on bf16 and set every block's hand-off scale to the largest power of two that keeps
        max|o| x headroom inside the e4m3 range (ssp2_fp8_calibrate_*).  Explicit and deterministic: only these images count.
        Returns the scales in force afterwards, one per block (16.0 = the uncalibrated default)."""
        if self.precision != "fp8":
            raise Ssp2Error("calibrate_fp8 needs precision='fp8'")
        self._bind_stream()
        self._check(self.lib.ssp2_fp8_calibrate_begin(self.h))
        try:
            for s0 in range(0, pixels.shape[0], self.max_images):      # EVERY block through the full path (forward_logits ends in the CLS-only tail)
                chunk = pixels[s0:s0 + self.max_images]
                x = self.embed(chunk)
                self.layers(x, chunk.shape[0], 0, self.depth)
        finally:
            self._check(self.lib.ssp2_fp8_calibrate_end(self.h, float(headroom)))
        top = self.fp8_fc2_top_codes()
        if top > 0:
            import warnings
            warnings.warn(f"ssp2vit fp8 calibration: {top} FFN activation value(s) of the calibration images sit on the e4m3 top code (|GELU output| >= 448): "
                          "the fc1 -> fc2 hand-off of this checkpoint clips in fp8 mode; use precision='bf16'", RuntimeWarning, stacklevel=2)
        return [float(self.lib.ssp2_fp8_attn_scale(self.h, l)) for l in range(self.depth)]

    def fp8_fc2_top_codes(self) -> int:
        """fp8 mode: how many e4m3 bytes of the fc1 -> fc2 hand-off the LAST calibration pass (calibrate_fp8) found on the top code +-448 — GELU
        outputs at or beyond the e4m3 range, which the saturating cast clips (SSP2_Q_FP8_FC2_TOP_CODES).  0: that hand-off is safe on those images."""
        v = self.lib.ssp2_query(self.h, 9)
        if v < 0:
            self._check(v)
        return int(v)

    def fp8_saturation(self, reset: bool = False) -> int:
        """fp8 mode: how many waves have CLIPPED a value when the attention output was handed to the out-projection as e4m3(o x 16)
        (|o| > 28) since the engine was built / the counter was reset.  0 for a model whose attention outputs stay in range; > 0
        says this checkpoint wants `set_option("fp8_proj", 0)` or bf16.  Waits for the stream."""
        v = self.lib.ssp2_query(self.h, 8 if reset else 7)            # SSP2_Q_FP8_SATURATED_RESET / SSP2_Q_FP8_SATURATED
        if v < 0:
            self._check(v)
        return int(v)

    def pruned_twin(self, d_int: Sequence[int], max_images: int = 64) -> "VitEngine":
        """A second engine of the same architecture with FFN widths `d_int`, every other weight cloned device to
        device: the container `apply_into` fills.  Built once (outside a timed region); a prune then costs gathers only."""
        return VitEngine(None, max_images=max_images, _twin_of=self, _d_int=list(d_int))

    def apply_into(self, twin: "VitEngine", masks: Sequence[Sequence[int]], drop_blocks: Sequence[int]) -> "VitEngine":
        """Stage-1 + stage-2 APPLY (reference src/vit_pruning.py:297-311 and :499-504) into `twin`, leaving this dense
        engine untouched: kept FFN neurons of every block are gathered in HBM, the chosen blocks lose their attention.
        Asynchronous on the current stream.  masks[l][j] == 1 -> neuron j of block l is pruned."""
        self.apply_ffn_into(twin, masks)
        return self.apply_attention_into(twin, drop_blocks)

    def apply_ffn_into(self, twin: "VitEngine", masks: Sequence[Sequence[int]]) -> "VitEngine":
        """The width half of `apply_into` (a8): the gathers are ENQUEUED on the current stream and do not depend on the depth search, so
        a caller that has the masks before the search has finished can queue them behind it and spare the GPU the host's latency
        at the end of the prune (bench.py does)."""
        twin._bind_stream()
        for l, m in enumerate(masks):
            keep = m if isinstance(m, torch.Tensor) else torch.as_tensor(m)
            keep = torch.nonzero(keep == 0).view(-1).to(torch.int32).contiguous()
            self._check(self.lib.ssp2_prune_ffn_into(twin.h, self.h, l, C.cast(keep.data_ptr(), C.POINTER(C.c_int32)), keep.numel()))
        return twin

    def apply_attention_into(self, twin: "VitEngine", drop_blocks: Sequence[int]) -> "VitEngine":
        """The depth half (a9): host-side flags only, no device work."""
        drop = set(int(b) for b in drop_blocks)
        for l in range(self.depth):
            if l in drop or self.absent[l]:
                self._check(self.lib.ssp2_drop_attention(twin.h, l)); twin.absent[l] = True
            else:
                self._check(self.lib.ssp2_restore_attention(twin.h, l)); twin.absent[l] = False
        return twin

    # ------------------------------------------------------------------ plumbing
    def _bind_stream(self) -> None:
        self._check(self.lib.ssp2_set_stream(self.h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def _skip_array(self, attn_skip: Optional[Sequence[int]]):
        flags = [1 if a else 0 for a in self.absent]
        if attn_skip is not None:
            for i in attn_skip:
                flags[int(i)] = 1
        if not any(flags):
            return None
        return (C.c_uint8 * self.depth)(*flags)

    def rows(self, n: int, group: int = 0) -> int:
        return int(self.lib.ssp2_rows(self.h, n, int(group)))

    def new_x(self, n: int, group: int = 0) -> torch.Tensor:
        return torch.empty(self.rows(n, group), self.dim, dtype=torch.float32, device=self.device)

    def new_scores(self, groups: int = 1) -> torch.Tensor:
        return torch.zeros(groups, self.depth, self.score_ld, dtype=torch.float32, device=self.device)

    # ------------------------------------------------------------------ the four device entry points
    batch_lists = True          # embed / forward_scores take a LIST of batches: each lands in its rows of x, nothing is concatenated
    prefix_scoring = True       # layers(score_images=...) / tail(group=...): the search's baseline can carry the stage-1 hook (core.prune_pass)

    def embed(self, pixels, x: Optional[torch.Tensor] = None, group: int = 0) -> torch.Tensor:
        if isinstance(pixels, (list, tuple)):
            # several dataloader batches -> one token matrix, each batch embedded straight into its rows (round 2 built one
            # pixel tensor with torch.cat first: 308 MB copied per 512-image stage-1 launch).  Slab layout (group > 0): batch
            # b is slab b and must hold `group` images, except the last; contiguous layout: batches back to back.
            n = sum(int(p.shape[0]) for p in pixels)
            x = self.new_x(n, group) if x is None else x
            if group > 0 and group < n:
                mpad = -(-group * self.tokens // _lib.SLAB_ALIGN) * _lib.SLAB_ALIGN
                pad = mpad - group * self.tokens
                if pad:                                    # pad rows between slabs: finite values (they flow through LN / GEMMs)
                    x[: (len(pixels) - 1) * mpad].view(len(pixels) - 1, mpad, self.dim)[:, group * self.tokens:, :].zero_()
                for b, p in enumerate(pixels):
                    if int(p.shape[0]) != group and b != len(pixels) - 1:
                        raise ValueError("slab layout: every batch but the last must hold `group` images")
                    self.embed(p, x=x[b * mpad:], group=0)
            else:
                row = 0
                for p in pixels:
                    self.embed(p, x=x[row:], group=0)
                    row += int(p.shape[0]) * self.tokens
            return x
        if pixels.dim() != 4 or pixels.shape[1] != 3 or pixels.shape[2] != self.img or pixels.shape[3] != self.img:
            raise ValueError(f"pixel_values must be [n,3,{self.img},{self.img}], got {tuple(pixels.shape)}")
        px = pixels.to(self.device, torch.float32, non_blocking=True).contiguous()
        n = px.shape[0]
        x = self.new_x(n, group) if x is None else x
        self._bind_stream()
        self._check(self.lib.ssp2_embed(self.h, _ptr(px), n, _ptr(x), int(group)))
        return x

    def layers(self, x: torch.Tensor, n: int, l_begin: int = 0, l_end: Optional[int] = None,
               attn_skip: Optional[Sequence[int]] = None, score_site: str = "none", score_chain: str = "fp32",
               batch_scores: Optional[torch.Tensor] = None, score_group: int = 0, scores_only: bool = False,
               x_in: Optional[torch.Tensor] = None, score_images: Optional[int] = None) -> Optional[torch.Tensor]:
        """Returns f32 [groups, depth, score_ld] when scoring (groups = ceil(n / score_group), 1 if score_group=0).
        `scores_only`: x is scratch afterwards — the last block stops behind its hooked activation (SSP2_SCORE_ONLY).
        `x_in`: the stream entering block l_begin is read from there and left untouched (ssp2_layers_from); x receives the result.
        `score_images`: only the leading images of the launch are hooked (whole slabs; ssp2_layers_prefix) — the search's baseline
        doubling as the stage-1 pass; groups = ceil(score_images / score_group)."""
        l_end = self.depth if l_end is None else l_end
        site = SCORE_SITE[score_site] | (0x10 if (scores_only and SCORE_SITE[score_site]) else 0)
        ns = n if score_images is None else int(score_images)
        grp = ns if (score_group <= 0 or score_group > ns) else score_group
        if site and batch_scores is None:
            batch_scores = self.new_scores((ns + grp - 1) // grp)
        self._bind_stream()
        if x_in is not None and (x_in.shape[0] < x.shape[0] or x_in.shape[1:] != x.shape[1:] or x_in.dtype != x.dtype or not x_in.is_contiguous()):
            raise ValueError("x_in must be a contiguous stream of the shape of x")
        self._check(self.lib.ssp2_layers_prefix(self.h, _ptr(x_in), _ptr(x), n, l_begin, l_end, self._skip_array(attn_skip), site,
                                          SCORE_CHAIN[score_chain], int(score_group), ns, _ptr(batch_scores if site else None), self.score_ld))
        return batch_scores if site else None

    def head(self, x: torch.Tensor, n: int, labels: Optional[torch.Tensor] = None,
             correct: Optional[torch.Tensor] = None, want_logits: bool = False, want_pred: bool = False, group: int = 0):
        logits = torch.empty(n, self.classes, dtype=torch.float32, device=self.device) if want_logits else None
        pred = torch.empty(n, dtype=torch.int32, device=self.device) if want_pred else None
        if labels is not None:
            labels = labels.to(self.device, torch.int64, non_blocking=True).contiguous()
            if correct is None:
                correct = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._bind_stream()
        self._check(self.lib.ssp2_head(self.h, _ptr(x), n, int(group), _ptr(logits), _ptr(pred), _ptr(labels),
                                 _ptr(correct if labels is not None else None)))
        return logits, pred, correct

    def tail(self, x: torch.Tensor, n: int, attn_skip: Optional[Sequence[int]] = None,
             labels: Optional[torch.Tensor] = None, correct: Optional[torch.Tensor] = None, want_logits: bool = False,
             want_pred: bool = False, slots: int = 1, group: int = 0):
        """Last encoder block + head on the CLS rows only (x must hold the residual stream ENTERING the last block;
        it is not modified).  Same results, bit for bit, as layers(x, depth-1, depth) + head(x).
        `slots` > 1: x holds that many streams of n images side by side (ssp2_tail_slots); labels [n] are shared, `correct`
        must have `slots` entries (slot s is counted in correct[s]), logits / pred cover slots * n images.
        `group` > 0: the streams are in the slab layout of `rows(slots * n, group)` (ssp2_tail_group)."""
        tot = n * int(slots)
        logits = torch.empty(tot, self.classes, dtype=torch.float32, device=self.device) if want_logits else None
        pred = torch.empty(tot, dtype=torch.int32, device=self.device) if want_pred else None
        if labels is not None:
            labels = labels.to(self.device, torch.int64, non_blocking=True).contiguous()
            if correct is None:
                correct = torch.zeros(int(slots), dtype=torch.int64, device=self.device)
            if correct.numel() < int(slots) or not correct.is_contiguous():
                raise ValueError("`correct` needs one contiguous int64 entry per slot")
        skip_last = bool(self.absent[self.depth - 1]) or (attn_skip is not None and (self.depth - 1) in [int(i) for i in attn_skip])
        self._bind_stream()
        self._check(self.lib.ssp2_tail_group(self.h, _ptr(x), n, int(slots), int(group), int(skip_last), _ptr(logits), _ptr(pred), _ptr(labels),
                                       _ptr(correct if labels is not None else None)))
        return logits, pred, correct

    def act_l2_accum(self, act: torch.Tensor, score_chain: str = "fp32") -> torch.Tensor:
        """Standalone hook-body kernel: act [n, tokens, d] (bf16 or f32, contiguous, d % 8 == 0) -> f32 [d]."""
        if act.dim() != 3 or not act.is_contiguous() or act.device.type != "cuda":
            raise ValueError("act must be a contiguous [n, tokens, d] device tensor")
        n, t, d = act.shape
        dtype = {torch.bfloat16: 0, torch.float32: 1}[act.dtype]
        ws = torch.empty(2, n, d, dtype=torch.float32, device=act.device)
        out = torch.empty(d, dtype=torch.float32, device=act.device)
        self._check(self.lib.ssp2_act_l2_accum(C.c_void_p(torch.cuda.current_stream(act.device).cuda_stream), _ptr(act), dtype,
                                         n, t, d, d, SCORE_CHAIN[score_chain], 0, _ptr(ws), _ptr(out), d))
        return out

    def linear(self, a: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], epilogue: str = "bf16",
               x: Optional[torch.Tensor] = None, kernel: str = "auto") -> torch.Tensor:
        """One projection of the forward on its own (ssp2_linear_bf16): a bf16 [M, K], weight [N, K] (any float dtype,
        rounded to bf16 like autocast does), bias [N] or None.  epilogue "bf16" / "gelu" -> bf16 [M, N];
        "resid" -> x (f32 [M, N]) updated in place and returned."""
        epi = {"bf16": 0, "resid": 1, "gelu": 2}[epilogue]
        M, K = a.shape
        N = weight.shape[0]
        npad = (N + 255) // 256 * 256
        wp = torch.zeros(npad, K, dtype=torch.bfloat16, device=self.device)
        wp[:N].copy_(weight.to(self.device, torch.bfloat16))
        bp = torch.zeros(npad, dtype=torch.float32, device=self.device)
        if bias is not None:
            bp[:N].copy_(bias.to(self.device, torch.bfloat16).float())
        a = a.to(self.device, torch.bfloat16).contiguous()
        out = None if epi == 1 else torch.empty(M, N, dtype=torch.bfloat16, device=self.device)
        if epi == 1 and (x is None or x.dtype != torch.float32 or not x.is_contiguous() or x.shape[1] != N or x.shape[0] < M):
            raise ValueError("resid epilogue needs a contiguous f32 x [>= M, N]")
        self._check(self.lib.ssp2_linear_bf16(C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream), epi, _ptr(a), K,
                                        _ptr(wp), K, _ptr(bp), M, N, K, _ptr(out), N, _ptr(x if epi == 1 else None), N,
                                        {"auto": 0, "small": 1, "big": 2}[kernel]))
        return x if epi == 1 else out

    # ------------------------------------------------------------------ compositions used by the host API
    def forward_scores(self, pixels: torch.Tensor, score_site: str, score_chain: str = "fp32",
                       group: int = 0) -> torch.Tensor:
        """One forward over `pixels` (one or several dataloader batches of `group` images each, concatenated).
        Returns f32 [n_groups, depth, score_ld]; entry [g, l] = sum over group g's samples of the per-sample
        token-L2 of block l's FFN activation (reference hook body, src/vit_pruning.py:151-152)."""
        n = sum(int(p.shape[0]) for p in pixels) if isinstance(pixels, (list, tuple)) else pixels.shape[0]
        if n > self.max_images:
            raise Ssp2Error(f"chunk of {n} images exceeds engine capacity {self.max_images}")
        x = self.embed(pixels, group=group)             # slab layout: one 128-row-aligned slab per batch
        return self.layers(x, n, 0, self.depth, None, score_site, score_chain, None, group, scores_only=True)

    def forward_logits(self, pixels: torch.Tensor, attn_skip: Optional[Sequence[int]] = None) -> torch.Tensor:
        outs = []
        for s in range(0, pixels.shape[0], self.max_images):
            chunk = pixels[s:s + self.max_images]
            x = self.embed(chunk)
            self.layers(x, chunk.shape[0], 0, self.depth - 1, attn_skip)
            outs.append(self.tail(x, chunk.shape[0], attn_skip, want_logits=True)[0])
        return torch.cat(outs, 0)

    def profile(self, klass: str):
        """HIP events around every launch of one kernel class ("gemm_fc1", ...) or of every class ("all"); after the block:
        .total_ms / .launches / .flops, and with "all" also .by_class[name] = dict(ms, launches, flops, bytes)."""
        eng = self

        class _Ctx:
            def __enter__(self_inner):
                eng._bind_stream()
                eng._check(eng.lib.ssp2_profile_begin(eng.h, len(K_CLASS) if klass == "all" else K_CLASS[klass]))
                return self_inner

            def __exit__(self_inner, *exc):
                ms, cnt, fl = C.c_double(), C.c_int64(), C.c_double()
                if klass == "all":
                    self_inner.by_class = {}
                    for name, k in K_CLASS.items():
                        by = C.c_double()
                        eng._check(eng.lib.ssp2_profile_query(eng.h, k, C.byref(ms), C.byref(cnt), C.byref(fl), C.byref(by)))
                        if cnt.value:
                            self_inner.by_class[name] = {"ms": ms.value, "launches": cnt.value, "flops": fl.value, "bytes": by.value}
                eng._check(eng.lib.ssp2_profile_end(eng.h, C.byref(ms), C.byref(cnt), C.byref(fl)))
                self_inner.total_ms, self_inner.launches, self_inner.flops = ms.value, cnt.value, fl.value
                return False
        return _Ctx()
