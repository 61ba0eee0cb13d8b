"""TEST INFRASTRUCTURE — CPU oracle model definitions (never imported by the product path).

Two plain-PyTorch ViT modules, one per anatomy the reference duck-types over
(/root/reference/src/vit_pruning.py:27-75):

* ``TimmLayoutViT``  — ``.blocks[i].{norm1,attn.{qkv,proj},norm2,mlp.{fc1,fc2}}``; called as
  ``model(px)``; the reference hooks ``mlp.fc1``  => PRE-GELU scores   (vit_pruning.py:135).
* ``HFLayoutViT``    — ``.vit.encoder.layer[i].{layernorm_before,attention,layernorm_after,
  intermediate.dense,output.dense}``; called as ``model(pixel_values=px)`` and returns an object with
  ``.logits``; attention returns a tuple; the reference hooks ``layer.intermediate`` => POST-GELU
  scores (vit_pruning.py:130).

timm / the HF version the reference was written against are not installed in this image
(SURVEY.md §8c), so these are build-owned modules with the same arithmetic: pre-norm blocks,
fused scaled-dot-product attention, erf-GELU, CLS-token pooling.  Both are constructed from the same
flat ``ViTWeights`` dictionary the product engine consumes (``ssp2vit.weights``), so a parity test feeds
identical numbers to both sides.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- timm layout
class _TimmAttention(nn.Module):
    def __init__(self, dim: int, heads: int):
        super().__init__()
        self.num_heads = heads
        self.head_dim = dim // heads
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        b, n, c = x.shape
        qkv = self.qkv(x).reshape(b, n, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        o = F.scaled_dot_product_attention(q, k, v)
        return self.proj(o.transpose(1, 2).reshape(b, n, c))


class _TimmMlp(nn.Module):
    def __init__(self, dim: int, inter: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, inter)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(inter, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _TimmBlock(nn.Module):
    def __init__(self, dim, heads, inter, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = _TimmAttention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = _TimmMlp(dim, inter)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class _PatchEmbed(nn.Module):
    def __init__(self, dim, patch, chans=3):
        super().__init__()
        self.proj = nn.Conv2d(chans, dim, kernel_size=patch, stride=patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class TimmLayoutViT(nn.Module):
    def __init__(self, *, img, patch, dim, heads, inter, depth, classes, eps=1e-6):
        super().__init__()
        n_tok = (img // patch) ** 2 + 1
        self.patch_embed = _PatchEmbed(dim, patch)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n_tok, dim))
        inters = inter if isinstance(inter, (list, tuple)) else [inter] * depth
        self.blocks = nn.ModuleList([_TimmBlock(dim, heads, inters[i], eps) for i in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=eps)
        self.head = nn.Linear(dim, classes)

    def forward(self, x):
        x = self.patch_embed(x)
        x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1), x), dim=1)
        x = x + self.pos_embed
        for blk in self.blocks:
            x = blk(x)
        x = self.norm(x)
        return self.head(x[:, 0])


# ----------------------------------------------------------------------------- old-HF layout
class _HFSelfAttention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_attention_heads = heads
        self.attention_head_size = dim // heads
        self.query = nn.Linear(dim, dim)
        self.key = nn.Linear(dim, dim)
        self.value = nn.Linear(dim, dim)

    def _split(self, t):
        b, n, _ = t.shape
        return t.view(b, n, self.num_attention_heads, self.attention_head_size).transpose(1, 2)

    def forward(self, hidden_states):
        q = self._split(self.query(hidden_states))
        k = self._split(self.key(hidden_states))
        v = self._split(self.value(hidden_states))
        o = F.scaled_dot_product_attention(q, k, v)
        b, h, n, d = o.shape
        return o.transpose(1, 2).reshape(b, n, h * d)


class _HFSelfOutput(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dense = nn.Linear(dim, dim)

    def forward(self, hidden_states):
        return self.dense(hidden_states)


class _HFAttention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.attention = _HFSelfAttention(dim, heads)
        self.output = _HFSelfOutput(dim)

    def forward(self, hidden_states, head_mask=None, output_attentions=False):
        return (self.output(self.attention(hidden_states)),)


class _HFIntermediate(nn.Module):
    def __init__(self, dim, inter):
        super().__init__()
        self.dense = nn.Linear(dim, inter)
        self.intermediate_act_fn = nn.GELU()

    def forward(self, hidden_states):
        return self.intermediate_act_fn(self.dense(hidden_states))


class _HFOutput(nn.Module):
    def __init__(self, dim, inter):
        super().__init__()
        self.dense = nn.Linear(inter, dim)

    def forward(self, hidden_states, input_tensor):
        return self.dense(hidden_states) + input_tensor


class _HFLayer(nn.Module):
    def __init__(self, dim, heads, inter, eps):
        super().__init__()
        self.layernorm_before = nn.LayerNorm(dim, eps=eps)
        self.attention = _HFAttention(dim, heads)
        self.layernorm_after = nn.LayerNorm(dim, eps=eps)
        self.intermediate = _HFIntermediate(dim, inter)
        self.output = _HFOutput(dim, inter)

    def forward(self, hidden_states):
        attn = self.attention(self.layernorm_before(hidden_states))[0]
        hidden_states = attn + hidden_states
        y = self.intermediate(self.layernorm_after(hidden_states))
        return self.output(y, hidden_states)


class _HFEncoder(nn.Module):
    def __init__(self, dim, heads, inters, eps):
        super().__init__()
        self.layer = nn.ModuleList([_HFLayer(dim, heads, i, eps) for i in inters])

    def forward(self, x):
        for lyr in self.layer:
            x = lyr(x)
        return x


class _HFPatchEmbeddings(nn.Module):
    def __init__(self, dim, patch):
        super().__init__()
        self.projection = nn.Conv2d(3, dim, kernel_size=patch, stride=patch)

    def forward(self, px):
        return self.projection(px).flatten(2).transpose(1, 2)


class _HFEmbeddings(nn.Module):
    def __init__(self, dim, patch, n_tok):
        super().__init__()
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.position_embeddings = nn.Parameter(torch.zeros(1, n_tok, dim))
        self.patch_embeddings = _HFPatchEmbeddings(dim, patch)

    def forward(self, px):
        e = self.patch_embeddings(px)
        e = torch.cat((self.cls_token.expand(e.shape[0], -1, -1), e), dim=1)
        return e + self.position_embeddings


class _HFViTModel(nn.Module):
    def __init__(self, dim, heads, inters, patch, n_tok, eps):
        super().__init__()
        self.embeddings = _HFEmbeddings(dim, patch, n_tok)
        self.encoder = _HFEncoder(dim, heads, inters, eps)
        self.layernorm = nn.LayerNorm(dim, eps=eps)

    def forward(self, px):
        return self.layernorm(self.encoder(self.embeddings(px)))


class HFLayoutViT(nn.Module):
    def __init__(self, *, img, patch, dim, heads, inter, depth, classes, eps=1e-12):
        super().__init__()
        n_tok = (img // patch) ** 2 + 1
        inters = inter if isinstance(inter, (list, tuple)) else [inter] * depth
        self.config = SimpleNamespace(hidden_size=dim, num_attention_heads=heads, num_labels=classes,
                                      intermediate_size=inters[0], num_hidden_layers=depth,
                                      image_size=img, patch_size=patch, layer_norm_eps=eps)
        self.vit = _HFViTModel(dim, heads, inters, patch, n_tok, eps)
        self.classifier = nn.Linear(dim, classes)

    def forward(self, pixel_values=None):
        seq = self.vit(pixel_values)
        return SimpleNamespace(logits=self.classifier(seq[:, 0, :]))


# ----------------------------------------------------------------------------- weight plumbing
@torch.no_grad()
def load_flat_weights(model: nn.Module, w: Dict[str, torch.Tensor]) -> nn.Module:
    """Copy a flat ``ssp2vit.weights`` dictionary (see that module for the key names) into either module."""
    L = int(w["depth"])

    def put(param, t):
        param.copy_(t.reshape(param.shape))

    if isinstance(model, TimmLayoutViT):
        put(model.patch_embed.proj.weight, w["patch_w"]); put(model.patch_embed.proj.bias, w["patch_b"])
        put(model.cls_token, w["cls"]); put(model.pos_embed, w["pos"])
        for i, blk in enumerate(model.blocks):
            put(blk.norm1.weight, w[f"ln1_g.{i}"]); put(blk.norm1.bias, w[f"ln1_b.{i}"])
            put(blk.attn.qkv.weight, w[f"qkv_w.{i}"]); put(blk.attn.qkv.bias, w[f"qkv_b.{i}"])
            put(blk.attn.proj.weight, w[f"proj_w.{i}"]); put(blk.attn.proj.bias, w[f"proj_b.{i}"])
            put(blk.norm2.weight, w[f"ln2_g.{i}"]); put(blk.norm2.bias, w[f"ln2_b.{i}"])
            put(blk.mlp.fc1.weight, w[f"fc1_w.{i}"]); put(blk.mlp.fc1.bias, w[f"fc1_b.{i}"])
            put(blk.mlp.fc2.weight, w[f"fc2_w.{i}"]); put(blk.mlp.fc2.bias, w[f"fc2_b.{i}"])
        put(model.norm.weight, w["lnf_g"]); put(model.norm.bias, w["lnf_b"])
        put(model.head.weight, w["head_w"]); put(model.head.bias, w["head_b"])
    elif isinstance(model, HFLayoutViT):
        emb = model.vit.embeddings
        put(emb.patch_embeddings.projection.weight, w["patch_w"])
        put(emb.patch_embeddings.projection.bias, w["patch_b"])
        put(emb.cls_token, w["cls"]); put(emb.position_embeddings, w["pos"])
        for i, lyr in enumerate(model.vit.encoder.layer):
            d = lyr.layernorm_before.weight.numel()
            put(lyr.layernorm_before.weight, w[f"ln1_g.{i}"]); put(lyr.layernorm_before.bias, w[f"ln1_b.{i}"])
            sa = lyr.attention.attention
            qw, kw, vw = w[f"qkv_w.{i}"].reshape(3, d, d)
            qb, kb, vb = w[f"qkv_b.{i}"].reshape(3, d)
            put(sa.query.weight, qw); put(sa.key.weight, kw); put(sa.value.weight, vw)
            put(sa.query.bias, qb); put(sa.key.bias, kb); put(sa.value.bias, vb)
            put(lyr.attention.output.dense.weight, w[f"proj_w.{i}"])
            put(lyr.attention.output.dense.bias, w[f"proj_b.{i}"])
            put(lyr.layernorm_after.weight, w[f"ln2_g.{i}"]); put(lyr.layernorm_after.bias, w[f"ln2_b.{i}"])
            put(lyr.intermediate.dense.weight, w[f"fc1_w.{i}"]); put(lyr.intermediate.dense.bias, w[f"fc1_b.{i}"])
            put(lyr.output.dense.weight, w[f"fc2_w.{i}"]); put(lyr.output.dense.bias, w[f"fc2_b.{i}"])
        put(model.vit.layernorm.weight, w["lnf_g"]); put(model.vit.layernorm.bias, w["lnf_b"])
        put(model.classifier.weight, w["head_w"]); put(model.classifier.bias, w["head_b"])
    else:
        raise TypeError(type(model))
    assert L == (len(model.blocks) if isinstance(model, TimmLayoutViT) else len(model.vit.encoder.layer))
    return model


def build_from_flat(w: Dict[str, torch.Tensor], layout: str) -> nn.Module:
    """``layout`` in {"timm", "hf"}; returns an eval-mode fp32 CPU module holding exactly ``w``."""
    cfg = dict(img=int(w["img"]), patch=int(w["patch"]), dim=int(w["dim"]), heads=int(w["heads"]),
               inter=[int(w[f"fc1_w.{i}"].shape[0]) for i in range(int(w["depth"]))],
               depth=int(w["depth"]), classes=int(w["classes"]))
    if layout == "timm":
        m = TimmLayoutViT(eps=float(w.get("eps", 1e-6)), **cfg)
    elif layout == "hf":
        m = HFLayoutViT(eps=float(w.get("eps", 1e-12)), **cfg)
    else:
        raise ValueError(layout)
    return load_flat_weights(m, w).eval()
