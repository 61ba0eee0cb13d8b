"""Parameter containers in the two anatomies the reference duck-types over, whose `forward` runs on the MI355X
engine (there is no PyTorch-eager forward here and no CPU path).  They exist so that code written against a
timm / HF ViT module — `model(px)`, `model(pixel_values=px).logits`, in-place weight slicing, attention bypass,
`count_total_params` — also works on synthetic or converted weights without timm/transformers installed."""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict

import torch
import torch.nn as nn

from . import vit_pruning as _vp


class _EngineAttn(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)

    def forward(self, *a, **k):
        raise RuntimeError("sub-module forward is not available: call the whole model (it runs on the HIP engine)")


class _EngineMlp(nn.Module):
    def __init__(self, dim, inter):
        super().__init__()
        self.fc1 = nn.Linear(dim, inter)
        self.fc2 = nn.Linear(inter, dim)


class _EngineBlock(nn.Module):
    def __init__(self, dim, heads, inter, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = _EngineAttn(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = _EngineMlp(dim, inter)


class _PatchEmbed(nn.Module):
    def __init__(self, dim, patch):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=patch, stride=patch)


class EngineViT(nn.Module):
    """timm-layout container (`.blocks[i].{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}`, `.patch_embed.proj`,
    `.cls_token`, `.pos_embed`, `.norm`, `.head`); `model(px)` returns logits computed by libssp2vit."""

    def __init__(self, w: Dict):
        super().__init__()
        dim, heads, depth = int(w["dim"]), int(w["heads"]), int(w["depth"])
        eps = float(w.get("eps", 1e-6))
        self.patch_embed = _PatchEmbed(dim, int(w["patch"]))
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, w["pos"].shape[1], dim))
        self.blocks = nn.ModuleList([_EngineBlock(dim, heads, int(w[f"fc1_w.{i}"].shape[0]), eps) for i in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=eps)
        self.head = nn.Linear(dim, int(w["classes"]))
        self.config = SimpleNamespace(hidden_size=dim, num_attention_heads=heads, num_labels=int(w["classes"]),
                                      image_size=int(w["img"]), patch_size=int(w["patch"]), layer_norm_eps=eps)
        self.ssp2_origin_layout = str(w.get("origin_layout") or w.get("layout") or "timm")     # where the weights came from (export writes it)
        with torch.no_grad():
            put = lambda p, t: p.copy_(t.reshape(p.shape))
            put(self.patch_embed.proj.weight, w["patch_w"]); put(self.patch_embed.proj.bias, w["patch_b"])
            put(self.cls_token, w["cls"]); put(self.pos_embed, w["pos"])
            for i, b in enumerate(self.blocks):
                put(b.norm1.weight, w[f"ln1_g.{i}"]); put(b.norm1.bias, w[f"ln1_b.{i}"])
                put(b.attn.qkv.weight, w[f"qkv_w.{i}"]); put(b.attn.qkv.bias, w[f"qkv_b.{i}"])
                put(b.attn.proj.weight, w[f"proj_w.{i}"]); put(b.attn.proj.bias, w[f"proj_b.{i}"])
                put(b.norm2.weight, w[f"ln2_g.{i}"]); put(b.norm2.bias, w[f"ln2_b.{i}"])
                put(b.mlp.fc1.weight, w[f"fc1_w.{i}"]); put(b.mlp.fc1.bias, w[f"fc1_b.{i}"])
                put(b.mlp.fc2.weight, w[f"fc2_w.{i}"]); put(b.mlp.fc2.bias, w[f"fc2_b.{i}"])
            put(self.norm.weight, w["lnf_g"]); put(self.norm.bias, w["lnf_b"])
            put(self.head.weight, w["head_w"]); put(self.head.bias, w["head_b"])
        for i, b in enumerate(self.blocks):              # a checkpoint saved after stage 2: those blocks have no attention
            if w.get(f"attn_absent.{i}"):
                b.attn = _vp.TimmAttentionBypass()
        if w.get("score_site") in ("pre_gelu", "post_gelu"):     # a checkpoint this build exported: the hook site travels with it
            self.ssp2_score_site = w["score_site"]
        elif w.get("layout") in ("hf", "hf5"):         # weights converted from an HF checkpoint: the reference's hook site there
            self.ssp2_score_site = "post_gelu"
        self.eval()

    @torch.no_grad()
    def forward(self, x=None, pixel_values=None):
        px = x if x is not None else pixel_values
        eng = _vp.engine_for(self, "cuda", max_images=max(64, int(px.shape[0])))
        return eng.forward_logits(px)
