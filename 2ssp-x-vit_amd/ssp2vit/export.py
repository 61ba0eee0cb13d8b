"""Pruned-model export (reference adaptation-for-Pures-framework/auto_2ssp.py:415-424 and :878-901).

Two on-disk forms, as in the reference CLI:
  * HF directory — `model.save_pretrained(dir)` (+ `processor.save_pretrained(dir)`): `config.json` + `model.safetensors`.
    A module that has `save_pretrained` (a transformers model) is saved by that very call, exactly like the reference; any
    other HF-/timm-layout nn.Module (the build-owned containers, a bare state dict) gets the same two files written here
    with `safetensors`, plus `pruning_meta.json` (per-block FFN width and the bypassed attention blocks — the reference's
    config.json keeps the ORIGINAL intermediate_size after a width prune, so a loader needs this to rebuild the shapes).
  * timm state dict — `torch.save(model.state_dict(), dir/timm_model.pth)` + `srp_meta.json` (:881-894).
"""
from __future__ import annotations

import json
import os
from pathlib import Path
from typing import Any, Dict, Optional

import torch

from . import vit_pruning as _vp


def pruning_meta(model) -> Dict[str, Any]:
    blocks, kind = _vp._blocks(model)
    widths = [int(fc1.out_features) for fc1, _ in _vp._gather_mlp_pairs(model)]
    absent = [i for i, b in enumerate(blocks)
              if sum(p.numel() for p in (_vp._attn_module(b, kind).parameters() if _vp._attn_module(b, kind) is not None else [])) == 0]
    from . import weights as _w
    site = getattr(model, "ssp2_score_site", None) or _w.score_site_for("timm" if kind == "timm" else "hf")
    cfg = getattr(model, "config", None)
    eps = getattr(cfg, "layer_norm_eps", None)
    if eps is None:
        eps = next((float(m.eps) for m in model.modules() if isinstance(m, torch.nn.LayerNorm)), None)
    heads = getattr(cfg, "num_attention_heads", None)
    # `layout` = the key layout of the saved state dict; `origin_layout` / `score_site` / `layer_norm_eps` = model facts a key layout
    # does not carry (an HF-origin model held in the timm-layout container still hooks post-GELU and normalises with eps 1e-12)
    return {"layout": kind, "origin_layout": getattr(model, "ssp2_origin_layout", kind), "score_site": site,
            "layer_norm_eps": None if eps is None else float(eps), "num_attention_heads": None if heads is None else int(heads),
            "num_blocks": len(blocks), "ffn_width_per_block": widths, "attention_removed_blocks": absent,
            "total_params": _vp.count_total_params(model)}


def _config_dict(model) -> Dict[str, Any]:
    cfg = getattr(model, "config", None)
    if cfg is not None and hasattr(cfg, "to_dict"):
        return cfg.to_dict()
    hidden, inters = _vp._get_hidden_and_inter_sizes(model)
    out = {"model_type": "vit", "hidden_size": int(hidden), "num_hidden_layers": len(inters),
           "intermediate_size": int(max(inters)) if inters else None}
    for k in ("num_attention_heads", "num_labels", "image_size", "patch_size", "layer_norm_eps"):
        v = getattr(cfg, k, None) if cfg is not None else None
        if v is not None:
            out[k] = v
    return out


def save_pretrained_dir(model, out_dir: str) -> str:
    """HF-format directory for `model` (see the module docstring)."""
    os.makedirs(out_dir, exist_ok=True)
    if hasattr(model, "save_pretrained"):
        model.save_pretrained(out_dir)                                    # the reference's own call (:419)
    else:
        from safetensors.torch import save_file
        sd = {k: v.detach().to("cpu").contiguous() for k, v in model.state_dict().items()}
        save_file(sd, os.path.join(out_dir, "model.safetensors"), metadata={"format": "pt"})
        with open(os.path.join(out_dir, "config.json"), "w", encoding="utf-8") as f:
            json.dump(_config_dict(model), f, indent=2, default=str)
    with open(os.path.join(out_dir, "pruning_meta.json"), "w", encoding="utf-8") as f:
        json.dump(pruning_meta(model), f, indent=2)
    return out_dir


def save_pruned_model_and_processor(model, processor, out_root, run_id: str) -> str:
    """Reference :415-424, same name and arguments."""
    out_dir = Path(out_root) / run_id
    save_pretrained_dir(model, out_dir.as_posix())
    try:
        processor.save_pretrained(out_dir.as_posix())
    except Exception:
        pass
    return out_dir.as_posix()


def save_timm_state_dict(model, out_root, run_id: str, srp_meta: Optional[Dict] = None) -> str:
    """Reference :881-894 (the SRP / timm branch)."""
    pdir = Path(out_root) / run_id
    pdir.mkdir(parents=True, exist_ok=True)
    torch.save(model.state_dict(), (pdir / "timm_model.pth").as_posix())
    with open(pdir / "srp_meta.json", "w", encoding="utf-8") as f:
        json.dump(srp_meta or {}, f, indent=2)
    with open(pdir / "pruning_meta.json", "w", encoding="utf-8") as f:        # (this build's addition: see pruning_meta)
        json.dump(pruning_meta(model), f, indent=2)
    return pdir.as_posix()
