"""Flat ViT weight dictionary: the one format the HIP engine is loaded from.

Keys (all tensors fp32, CPU, contiguous; layer index ``i`` in ``0..depth-1``)::

    img patch dim heads depth classes eps            python scalars
    patch_w [dim,3,p,p]  patch_b [dim]               conv k=s=p   (a3 in SURVEY.md §8a)
    cls [1,1,dim]  pos [1,N,dim]
    ln1_g.i ln1_b.i [dim]
    qkv_w.i [3*dim,dim]  qkv_b.i [3*dim]             rows ordered [q|k|v][head][d_h]
    proj_w.i [dim,dim]   proj_b.i [dim]
    ln2_g.i ln2_b.i [dim]
    fc1_w.i [d_int_i,dim] fc1_b.i [d_int_i]
    fc2_w.i [dim,d_int_i] fc2_b.i [dim]
    lnf_g lnf_b [dim]    head_w [classes,dim]  head_b [classes]

``nn.Linear`` row-major ``[out,in]`` is kept as is: it is already the K-contiguous "B^T" layout
the MFMA GEMM wants.  Anatomy adapters (``from_module``) accept the same three layouts the
reference duck-types over (/root/reference/src/vit_pruning.py:27-67) plus the renamed layout of
transformers>=5.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

# name -> (img, patch, dim, heads, d_int, depth)
VIT_CONFIGS = {
    "vit_tiny_patch16_224": (224, 16, 192, 3, 768, 12),
    "vit_small_patch16_224": (224, 16, 384, 6, 1536, 12),
    "vit_base_patch16_224": (224, 16, 768, 12, 3072, 12),
    "vit_large_patch16_224": (224, 16, 1024, 16, 4096, 24),
    "vit_huge_patch14_224": (224, 14, 1280, 16, 5120, 32),
    # the reference's own smoke-test config (experiments/vit_pruning/test_stage2_attention_only.py:44-53)
    "vit_test_patch16_32": (32, 16, 64, 4, 128, 4),
    # two-block cuts of the large geometries (same kernels / tile shapes as the full models, test-sized)
    "vit_base_patch16_224_d3": (224, 16, 768, 12, 3072, 3),
    "vit_large_patch16_224_d2": (224, 16, 1024, 16, 4096, 2),
    "vit_huge_patch14_224_d2": (224, 14, 1280, 16, 5120, 2),
    "vit_small_patch16_224_d2": (224, 16, 384, 6, 1536, 2),
    # geometries beside BASELINE's: ViT-L/14 (257 tokens at d_h = 64), ViT-B/32 (50 tokens, 3072-wide patches), B/16 at 160 / 208 pixels
    "vit_large_patch14_224": (224, 14, 1024, 16, 4096, 24),
    "vit_base_patch32_224": (224, 32, 768, 12, 3072, 12),
    "vit_large_patch14_224_d2": (224, 14, 1024, 16, 4096, 2),
    "vit_base_patch32_224_d2": (224, 32, 768, 12, 3072, 2),
    "vit_base_patch16_160_d2": (160, 16, 768, 12, 3072, 2),
    "vit_base_patch16_208_d2": (208, 16, 768, 12, 3072, 2),
    "vit_base_patch16_240_d2": (240, 16, 768, 12, 3072, 2),
}


def n_tokens(w: Dict) -> int:
    return (int(w["img"]) // int(w["patch"])) ** 2 + 1


def synthetic_weights(name: str = "vit_base_patch16_224", classes: int = 1000, seed: int = 0,
                      std: float = 0.02, eps: float = 1e-6, spread: Optional[float] = None,
                      bias_std: float = 0.0) -> Dict:
    """Seeded random-init weights of a named architecture (no checkpoints exist offline).

    trunc-normal(std) matrices, LN gamma=1 beta=0, zero (or N(0,bias_std)) biases.  ``spread`` scales
    fc1 rows by a log-uniform factor in [1/spread, spread] so stage-1 scores are well separated
    (SURVEY.md §8d).
    """
    img, patch, dim, heads, inter, depth = VIT_CONFIGS[name]
    g = torch.Generator().manual_seed(seed)

    def tn(*shape, s=std):
        t = torch.empty(*shape)
        torch.nn.init.trunc_normal_(t, std=s, a=-2 * s, b=2 * s, generator=g)
        return t

    def bias(n):
        return torch.randn(n, generator=g) * bias_std if bias_std > 0 else torch.zeros(n)

    n_tok = (img // patch) ** 2 + 1
    w: Dict = dict(img=img, patch=patch, dim=dim, heads=heads, depth=depth, classes=classes, eps=eps)
    w["patch_w"] = tn(dim, 3, patch, patch)
    w["patch_b"] = bias(dim)
    w["cls"] = tn(1, 1, dim)
    w["pos"] = tn(1, n_tok, dim)
    for i in range(depth):
        w[f"ln1_g.{i}"] = torch.ones(dim); w[f"ln1_b.{i}"] = torch.zeros(dim)
        w[f"qkv_w.{i}"] = tn(3 * dim, dim); w[f"qkv_b.{i}"] = bias(3 * dim)
        w[f"proj_w.{i}"] = tn(dim, dim); w[f"proj_b.{i}"] = bias(dim)
        w[f"ln2_g.{i}"] = torch.ones(dim); w[f"ln2_b.{i}"] = torch.zeros(dim)
        fc1 = tn(inter, dim)
        if spread is not None and spread > 1.0:
            u = torch.rand(inter, generator=g) * 2 - 1
            fc1 = fc1 * torch.exp(u * math.log(spread)).unsqueeze(1)
        w[f"fc1_w.{i}"] = fc1; w[f"fc1_b.{i}"] = bias(inter)
        w[f"fc2_w.{i}"] = tn(dim, inter); w[f"fc2_b.{i}"] = bias(dim)
    w["lnf_g"] = torch.ones(dim); w["lnf_b"] = torch.zeros(dim)
    w["head_w"] = tn(classes, dim); w["head_b"] = bias(classes)
    return w


def count_params(w: Dict) -> int:
    return sum(int(v.numel()) for v in w.values() if isinstance(v, torch.Tensor))


# ----------------------------------------------------------------------------- anatomy adapters
def _f32(t) -> torch.Tensor:
    """fp32 view / copy ON THE TENSOR'S OWN DEVICE: the engine reads device-resident weights in place
    (ssp2_load_tensor_dev), so a module that lives on the GPU never round-trips through the host."""
    return t.detach().to(torch.float32).contiguous()


def _bias_or_zeros(lin) -> torch.Tensor:
    """nn.Linear bias, or zeros when the layer was built with bias=False (timm's qkv_bias=False)."""
    return _f32(lin.bias) if getattr(lin, "bias", None) is not None else torch.zeros(lin.weight.shape[0], device=lin.weight.device)


def detect_layout(model) -> str:
    """'timm' | 'hf' (transformers<5: vit.encoder.layer) | 'hf5' (transformers>=5: vit.layers)."""
    if hasattr(model, "blocks") and hasattr(model, "patch_embed"):
        return "timm"
    base = getattr(model, "vit", None) or getattr(model, "base_model", None) or model
    enc = getattr(base, "encoder", base)
    if hasattr(enc, "layer"):
        return "hf"
    if hasattr(base, "layers") or hasattr(enc, "layers"):
        return "hf5"
    raise AttributeError("Unsupported ViT model structure: expected encoder.layer or blocks")


def score_site_for(layout: str) -> str:
    """Where the reference's stage-1 hook lands (vit_pruning.py:130 vs :135)."""
    return "pre_gelu" if layout == "timm" else "post_gelu"


def from_module(model) -> Dict:
    """Extract the flat dictionary from a live module (weights are copied, the module is untouched)."""
    layout = detect_layout(model)
    w: Dict = {}
    if layout == "timm":
        pe = model.patch_embed.proj
        blocks = list(model.blocks)
        w["patch_w"], w["patch_b"] = _f32(pe.weight), _f32(pe.bias)
        w["cls"], w["pos"] = _f32(model.cls_token), _f32(model.pos_embed)
        for i, b in enumerate(blocks):
            w[f"ln1_g.{i}"], w[f"ln1_b.{i}"] = _f32(b.norm1.weight), _f32(b.norm1.bias)
            attn = getattr(b, "attn", None)
            if attn is not None and hasattr(attn, "qkv"):
                w[f"qkv_w.{i}"], w[f"qkv_b.{i}"] = _f32(attn.qkv.weight), _bias_or_zeros(attn.qkv)
                w[f"proj_w.{i}"], w[f"proj_b.{i}"] = _f32(attn.proj.weight), _bias_or_zeros(attn.proj)
            else:  # attention already replaced by a bypass (a6): zero weights + skip flag
                w[f"attn_absent.{i}"] = True
            w[f"ln2_g.{i}"], w[f"ln2_b.{i}"] = _f32(b.norm2.weight), _f32(b.norm2.bias)
            w[f"fc1_w.{i}"], w[f"fc1_b.{i}"] = _f32(b.mlp.fc1.weight), _f32(b.mlp.fc1.bias)
            w[f"fc2_w.{i}"], w[f"fc2_b.{i}"] = _f32(b.mlp.fc2.weight), _f32(b.mlp.fc2.bias)
        w["lnf_g"], w["lnf_b"] = _f32(model.norm.weight), _f32(model.norm.bias)
        w["head_w"], w["head_b"] = _f32(model.head.weight), _f32(model.head.bias)
        eps = float(model.norm.eps)
        # the head count lives on the attention modules; block 0 may already be a bypass (the heuristic prunes it
        # first), so take it from the first block that still has one, then from a config object
        heads = next((int(b.attn.num_heads) for b in blocks if hasattr(getattr(b, "attn", None), "num_heads")), None)
        if heads is None:
            heads = getattr(getattr(model, "config", None), "num_attention_heads", None)
            heads = None if heads is None else int(heads)
    elif layout == "hf":
        vit = model.vit if hasattr(model, "vit") else model
        emb = vit.embeddings
        pe = emb.patch_embeddings.projection
        layers = list(vit.encoder.layer)
        w["patch_w"], w["patch_b"] = _f32(pe.weight), _f32(pe.bias)
        w["cls"], w["pos"] = _f32(emb.cls_token), _f32(emb.position_embeddings)
        for i, l in enumerate(layers):
            w[f"ln1_g.{i}"], w[f"ln1_b.{i}"] = _f32(l.layernorm_before.weight), _f32(l.layernorm_before.bias)
            att = getattr(l, "attention", None)
            sa = getattr(att, "attention", None)
            if sa is not None and hasattr(sa, "query"):
                w[f"qkv_w.{i}"] = torch.cat([_f32(sa.query.weight), _f32(sa.key.weight), _f32(sa.value.weight)], 0)
                w[f"qkv_b.{i}"] = torch.cat([_bias_or_zeros(sa.query), _bias_or_zeros(sa.key), _bias_or_zeros(sa.value)], 0)
                w[f"proj_w.{i}"], w[f"proj_b.{i}"] = _f32(att.output.dense.weight), _f32(att.output.dense.bias)
            else:
                w[f"attn_absent.{i}"] = True
            w[f"ln2_g.{i}"], w[f"ln2_b.{i}"] = _f32(l.layernorm_after.weight), _f32(l.layernorm_after.bias)
            w[f"fc1_w.{i}"], w[f"fc1_b.{i}"] = _f32(l.intermediate.dense.weight), _f32(l.intermediate.dense.bias)
            w[f"fc2_w.{i}"], w[f"fc2_b.{i}"] = _f32(l.output.dense.weight), _f32(l.output.dense.bias)
        w["lnf_g"], w["lnf_b"] = _f32(vit.layernorm.weight), _f32(vit.layernorm.bias)
        w["head_w"], w["head_b"] = _f32(model.classifier.weight), _f32(model.classifier.bias)
        eps = float(vit.layernorm.eps)
        heads = int(getattr(model.config, "num_attention_heads"))
    else:  # hf5: vit.layers[i].{layernorm_before, attention.{q,k,v,o}_proj, layernorm_after, mlp.fc1/fc2}
        vit = model.vit if hasattr(model, "vit") else model
        emb = vit.embeddings
        pe = emb.patch_embeddings.projection
        layers = list(getattr(vit, "layers", None) or vit.encoder.layers)
        w["patch_w"], w["patch_b"] = _f32(pe.weight), _f32(pe.bias)
        w["cls"], w["pos"] = _f32(emb.cls_token), _f32(emb.position_embeddings)
        for i, l in enumerate(layers):
            w[f"ln1_g.{i}"], w[f"ln1_b.{i}"] = _f32(l.layernorm_before.weight), _f32(l.layernorm_before.bias)
            a = l.attention
            if hasattr(a, "q_proj"):
                w[f"qkv_w.{i}"] = torch.cat([_f32(a.q_proj.weight), _f32(a.k_proj.weight), _f32(a.v_proj.weight)], 0)
                w[f"qkv_b.{i}"] = torch.cat([_bias_or_zeros(a.q_proj), _bias_or_zeros(a.k_proj), _bias_or_zeros(a.v_proj)], 0)
                w[f"proj_w.{i}"], w[f"proj_b.{i}"] = _f32(a.o_proj.weight), _bias_or_zeros(a.o_proj)
            else:
                w[f"attn_absent.{i}"] = True
            w[f"ln2_g.{i}"], w[f"ln2_b.{i}"] = _f32(l.layernorm_after.weight), _f32(l.layernorm_after.bias)
            w[f"fc1_w.{i}"], w[f"fc1_b.{i}"] = _f32(l.mlp.fc1.weight), _f32(l.mlp.fc1.bias)
            w[f"fc2_w.{i}"], w[f"fc2_b.{i}"] = _f32(l.mlp.fc2.weight), _f32(l.mlp.fc2.bias)
        w["lnf_g"], w["lnf_b"] = _f32(vit.layernorm.weight), _f32(vit.layernorm.bias)
        w["head_w"], w["head_b"] = _f32(model.classifier.weight), _f32(model.classifier.bias)
        eps = float(vit.layernorm.eps)
        heads = int(getattr(model.config, "num_attention_heads"))

    dim = int(w["patch_w"].shape[0])
    patch = int(w["patch_w"].shape[-1])
    n_tok = int(w["pos"].shape[1])
    side = int(round(math.sqrt(n_tok - 1)))
    depth = sum(1 for k in w if k.startswith("ln1_g."))
    if heads is None:
        if depth and all(w.get(f"attn_absent.{i}") for i in range(depth)):
            heads = max(1, dim // 64)       # no attention left anywhere: the value is never used by a kernel
        else:
            raise AttributeError("cannot determine the number of attention heads: no block exposes attn.num_heads and "
                                 "the model has no config.num_attention_heads")
    for i in range(depth):  # bypassed attention: supply zero weights, the engine skips the block anyway
        if w.get(f"attn_absent.{i}"):
            dv = w["patch_w"].device
            w[f"qkv_w.{i}"] = torch.zeros(3 * dim, dim, device=dv); w[f"qkv_b.{i}"] = torch.zeros(3 * dim, device=dv)
            w[f"proj_w.{i}"] = torch.zeros(dim, dim, device=dv); w[f"proj_b.{i}"] = torch.zeros(dim, device=dv)
    w.update(img=side * patch, patch=patch, dim=dim, heads=heads, depth=depth,
             classes=int(w["head_w"].shape[0]), eps=eps, layout=layout)
    return w


# ----------------------------------------------------------------------------- local checkpoints (CLI --weights)
_KNOWN_HEADS = {192: 3, 384: 6, 768: 12, 1024: 16, 1280: 16}       # ViT-Ti / S / B / L / H: the published geometries


def from_state_dict(sd: Dict[str, torch.Tensor], config: Optional[Dict] = None, meta: Optional[Dict] = None) -> Dict:
    """Flat dictionary from a checkpoint's state dict in any of the three key layouts `from_module` reads from live
    modules: timm (`blocks.N.attn.qkv.weight`, ...), transformers < 5 (`vit.encoder.layer.N.attention.attention.query...`)
    and transformers >= 5 (`vit.layers.N.attention.q_proj...`; also under `vit.encoder.layers`).  `config` = the HF
    config.json (heads, layer_norm_eps); `meta` = this build's pruning_meta.json (bypassed attention blocks).  FFN widths
    come from the tensor shapes, so a width-pruned checkpoint loads as it is.  The reference loads its models with
    `from_pretrained` / `timm.create_model` + `load_state_dict` (adaptation-for-Pures-framework/auto_2ssp.py:636-667);
    neither library's model code is needed here."""
    config, meta = dict(config or {}), dict(meta or {})
    f = lambda k: sd[k].detach().to(torch.float32).contiguous()
    has = lambda k: k in sd
    opt = lambda k, n: f(k) if has(k) else torch.zeros(n)
    w: Dict = {}
    absent = set(int(i) for i in meta.get("attention_removed_blocks", []))
    if has("patch_embed.proj.weight"):                                   # ---- timm
        layout = "timm"
        w["patch_w"], w["patch_b"] = f("patch_embed.proj.weight"), f("patch_embed.proj.bias")
        w["cls"], w["pos"] = f("cls_token"), f("pos_embed")
        dim = int(w["patch_w"].shape[0])
        i = 0
        while has(f"blocks.{i}.norm1.weight"):
            p = f"blocks.{i}."
            w[f"ln1_g.{i}"], w[f"ln1_b.{i}"] = f(p + "norm1.weight"), f(p + "norm1.bias")
            if has(p + "attn.qkv.weight"):
                w[f"qkv_w.{i}"], w[f"qkv_b.{i}"] = f(p + "attn.qkv.weight"), opt(p + "attn.qkv.bias", 3 * dim)
                w[f"proj_w.{i}"], w[f"proj_b.{i}"] = f(p + "attn.proj.weight"), opt(p + "attn.proj.bias", dim)
            else:
                absent.add(i)
            w[f"ln2_g.{i}"], w[f"ln2_b.{i}"] = f(p + "norm2.weight"), f(p + "norm2.bias")
            w[f"fc1_w.{i}"], w[f"fc1_b.{i}"] = f(p + "mlp.fc1.weight"), f(p + "mlp.fc1.bias")
            w[f"fc2_w.{i}"], w[f"fc2_b.{i}"] = f(p + "mlp.fc2.weight"), f(p + "mlp.fc2.bias")
            i += 1
        w["lnf_g"], w["lnf_b"] = f("norm.weight"), f("norm.bias")
        if not has("head.weight"):
            raise AttributeError("Unsupported ViT checkpoint: no classification head (head.weight) — the search evaluates top-1")
        w["head_w"], w["head_b"] = f("head.weight"), opt("head.bias", int(sd["head.weight"].shape[0]))
        eps = float(config.get("layer_norm_eps", meta.get("layer_norm_eps", 1e-6)))
    else:
        pre = "vit." if any(k.startswith("vit.") for k in sd) else ""
        emb = pre + "embeddings."
        if not has(emb + "patch_embeddings.projection.weight"):
            raise AttributeError("Unsupported ViT checkpoint: neither timm (patch_embed.proj) nor HF (embeddings.patch_embeddings) keys")
        w["patch_w"], w["patch_b"] = f(emb + "patch_embeddings.projection.weight"), f(emb + "patch_embeddings.projection.bias")
        w["cls"], w["pos"] = f(emb + "cls_token"), f(emb + "position_embeddings")
        dim = int(w["patch_w"].shape[0])
        if has(pre + "encoder.layer.0.layernorm_before.weight"):        # ---- transformers < 5
            layout, base = "hf", pre + "encoder.layer."
        else:                                                           # ---- transformers >= 5
            layout = "hf5"
            base = pre + ("layers." if has(pre + "layers.0.layernorm_before.weight") else "encoder.layers.")
        i = 0
        while has(f"{base}{i}.layernorm_before.weight"):
            p = f"{base}{i}."
            w[f"ln1_g.{i}"], w[f"ln1_b.{i}"] = f(p + "layernorm_before.weight"), f(p + "layernorm_before.bias")
            if layout == "hf":
                q, k_, v, o = (p + "attention.attention.query", p + "attention.attention.key", p + "attention.attention.value",
                               p + "attention.output.dense")
                fc1, fc2 = p + "intermediate.dense", p + "output.dense"
            else:
                q, k_, v, o = p + "attention.q_proj", p + "attention.k_proj", p + "attention.v_proj", p + "attention.o_proj"
                fc1, fc2 = p + "mlp.fc1", p + "mlp.fc2"
            if has(q + ".weight"):
                w[f"qkv_w.{i}"] = torch.cat([f(q + ".weight"), f(k_ + ".weight"), f(v + ".weight")], 0)
                w[f"qkv_b.{i}"] = torch.cat([opt(q + ".bias", dim), opt(k_ + ".bias", dim), opt(v + ".bias", dim)], 0)
                w[f"proj_w.{i}"], w[f"proj_b.{i}"] = f(o + ".weight"), opt(o + ".bias", dim)
            else:
                absent.add(i)
            w[f"ln2_g.{i}"], w[f"ln2_b.{i}"] = f(p + "layernorm_after.weight"), f(p + "layernorm_after.bias")
            w[f"fc1_w.{i}"], w[f"fc1_b.{i}"] = f(fc1 + ".weight"), f(fc1 + ".bias")
            w[f"fc2_w.{i}"], w[f"fc2_b.{i}"] = f(fc2 + ".weight"), f(fc2 + ".bias")
            i += 1
        w["lnf_g"], w["lnf_b"] = f(pre + "layernorm.weight"), f(pre + "layernorm.bias")
        if not has("classifier.weight"):
            raise AttributeError("Unsupported ViT checkpoint: no classification head (classifier.weight; a bare ViTModel?) — the search evaluates top-1")
        w["head_w"], w["head_b"] = f("classifier.weight"), opt("classifier.bias", int(sd["classifier.weight"].shape[0]))
        eps = float(config.get("layer_norm_eps", meta.get("layer_norm_eps", 1e-12)))
    depth = i
    if depth == 0:
        raise AttributeError("Unsupported ViT checkpoint: no encoder blocks found")
    heads = config.get("num_attention_heads", meta.get("num_attention_heads"))
    if heads is None:
        heads = _KNOWN_HEADS.get(dim)
    if heads is None:
        raise AttributeError(f"cannot determine the number of attention heads for hidden size {dim}: give config.json "
                             "(num_attention_heads) or --heads")
    for b in range(depth):
        if b in absent:
            w[f"attn_absent.{b}"] = True
            w[f"qkv_w.{b}"] = torch.zeros(3 * dim, dim); w[f"qkv_b.{b}"] = torch.zeros(3 * dim)
            w[f"proj_w.{b}"] = torch.zeros(dim, dim); w[f"proj_b.{b}"] = torch.zeros(dim)
    patch = int(w["patch_w"].shape[-1])
    side = int(round(math.sqrt(int(w["pos"].shape[1]) - 1)))
    w.update(img=side * patch, patch=patch, dim=dim, heads=int(heads), depth=depth, classes=int(w["head_w"].shape[0]),
             eps=eps, layout=layout)
    # A checkpoint this build exported says where it came from (pruning_meta.json): an HF-origin model re-saved in the timm key
    # layout keeps its hook site (post-GELU, reference src/vit_pruning.py:130 vs :135) and its LayerNorm eps across the round trip
    if meta.get("score_site") in ("pre_gelu", "post_gelu"):
        w["score_site"] = meta["score_site"]
    if meta.get("origin_layout"):
        w["origin_layout"] = meta["origin_layout"]
    return w


def load_checkpoint(path: str, heads: Optional[int] = None) -> Dict:
    """CLI `--weights`: a LOCAL checkpoint -> the flat dictionary.  Accepted: a `.safetensors` file, a `.pth` / `.pt` /
    `.bin` state dict (loaded with `weights_only=True`: nothing in the file is executed), or a directory holding
    `model.safetensors` / `pytorch_model.bin` / `timm_model.pth` with an optional `config.json` and this build's
    `pruning_meta.json` — i.e. an HF `save_pretrained` directory, or what `ssp2vit.export` writes.  No network access."""
    import json
    import os
    config, meta = {}, {}
    file = path
    if os.path.isdir(path):
        for name in ("model.safetensors", "pytorch_model.bin", "timm_model.pth"):
            if os.path.exists(os.path.join(path, name)):
                file = os.path.join(path, name)
                break
        else:
            raise FileNotFoundError(f"{path}: no model.safetensors / pytorch_model.bin / timm_model.pth inside")
        for name, dst in (("config.json", config), ("pruning_meta.json", meta)):
            p = os.path.join(path, name)
            if os.path.exists(p):
                with open(p, encoding="utf-8") as fh:
                    dst.update(json.load(fh))
    if file.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(file)
    else:
        sd = torch.load(file, map_location="cpu", weights_only=True)
        if isinstance(sd, dict) and "state_dict" in sd and not any(torch.is_tensor(v) for v in sd.values()):
            sd = sd["state_dict"]
    if heads is not None:
        config["num_attention_heads"] = int(heads)
    return from_state_dict(sd, config, meta)
