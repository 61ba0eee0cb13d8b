"""ctypes binding of libssp2vit.so (include/ssp2vit.h).  No compute happens here and there is no fallback:
if the shared library (or a GPU) is missing the product path raises."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)                       # 2ssp-x-vit_amd/
CSRC = os.path.join(PKG_ROOT, "csrc")
LIB_PATH = os.path.join(PKG_ROOT, "lib", "libssp2vit.so")
INCLUDE = os.path.join(os.path.dirname(PKG_ROOT), "include")

SLAB_ALIGN = 128                                        # csrc/common.hip.h kSlabAlign: slabs of the slab layout are padded to a multiple of this many rows
ABI_VERSION = 5                                         # SSP2_ABI_VERSION of include/ssp2vit.h

# every symbol include/ssp2vit.h declares
SYMBOLS = [
    "ssp2_abi_version", "ssp2_last_error", "ssp2_create", "ssp2_destroy", "ssp2_set_stream", "ssp2_load_tensor",
    "ssp2_load_tensor_dev", "ssp2_load_tensors_dev", "ssp2_linear_bf16", "ssp2_query", "ssp2_restore_attention", "ssp2_clone_weights",
    "ssp2_prune_ffn_into", "ssp2_set_precision", "ssp2_set_cu_limit", "ssp2_set_option", "ssp2_get_option", "ssp2_profile_query",
    "ssp2_embed", "ssp2_layers", "ssp2_layers_from", "ssp2_layers_prefix", "ssp2_head", "ssp2_tail", "ssp2_tail_slots", "ssp2_tail_group", "ssp2_prune_ffn", "ssp2_drop_attention", "ssp2_d_int", "ssp2_act_l2_accum", "ssp2_profile_begin", "ssp2_profile_end",
    "ssp2_tokens", "ssp2_rows", "ssp2_workspace_bytes", "ssp2_preproc_create", "ssp2_preproc_run", "ssp2_preproc_destroy",
    "ssp2_fp8_calibrate_begin", "ssp2_fp8_calibrate_end", "ssp2_fp8_attn_scale", "ssp2_fp8_set_attn_scale",
]

T_KINDS = ["patch_w", "patch_b", "cls", "pos", "ln1_g", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b",
           "ln2_g", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "lnf_g", "lnf_b", "head_w", "head_b"]
SCORE_SITE = {"none": 0, "pre_gelu": 1, "post_gelu": 2}
SCORE_CHAIN = {"fp32": 0, "bf16_ref": 1}
OPTIONS = {"zigzag": 0, "attn_persist": 1, "ln_fusion": 2, "big_tiles": 3, "fc1_big_tiles": 4, "group256": 5, "patch_lds": 6, "attn_stagger": 7, "fp8_proj": 8, "big_tile_min_rows": 9, "nt_stores": 10, "defer_resid": 11, "attn_live": 12}   # SSP2_OPT_*
K_CLASS = {"gemm_fc1": 0, "gemm_fc2": 1, "gemm_qkv": 2, "gemm_proj": 3, "gemm_patch": 4, "gemm_head": 5,
           "attn": 6, "ln": 7, "score_finish": 8, "act_l2": 9, "other": 10}


class VitDesc(C.Structure):
    _fields_ = [("img", C.c_int32), ("patch", C.c_int32), ("dim", C.c_int32), ("heads", C.c_int32),
                ("depth", C.c_int32), ("classes", C.c_int32), ("ln_eps", C.c_float), ("max_images", C.c_int32),
                ("d_int", C.POINTER(C.c_int32))]


class Ssp2Error(RuntimeError):
    pass


# Named builds beside the product library: lib/libssp2vit_<name>.so = the same sources with extra -D flags.  "lab" carries the opt-in
# kernel forms the product does not instantiate (LayerNorm behind the residual epilogue, deferred residual, column-group tile orders:
# csrc/gemm256.hip.h SSP2_LAB) — their bit-identity tests run against it (VitEngine(lib_variant="lab")).
VARIANT_FLAGS = {"lab": ("-DSSP2_LAB=1",)}


def variant_path(variant: Optional[str]) -> str:
    return LIB_PATH if not variant else os.path.join(os.path.dirname(LIB_PATH), f"libssp2vit_{variant}.so")


def build_library(verbose: bool = False, only_if_stale: bool = False, variant: Optional[str] = None, flags=()) -> str:
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU).  Output stays in-tree: 2ssp-x-vit_amd/lib/."""
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = variant_path(variant)
    flags = tuple(flags) or VARIANT_FLAGS.get(variant or "", ())
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", *flags,
           os.path.join(CSRC, "engine.hip"), "-o", out]
    if verbose:
        print(" ".join(cmd))
    import fcntl
    with open(out + ".lock", "w") as lock:          # one builder at a time (several ranks may start together)
        fcntl.flock(lock, fcntl.LOCK_EX)
        if only_if_stale and not _needs_rebuild(variant):
            return out
        tmp = out + f".tmp{os.getpid()}"
        cmd[-1] = tmp
        subprocess.run(cmd, check=True, cwd=CSRC)
        os.replace(tmp, out)
        with open(out + ".srchash", "w") as f:
            f.write(_source_hash() + " " + " ".join(flags))
    return out


HASH_PATH = LIB_PATH + ".srchash"


def _source_hash() -> str:
    """Content hash of every source the library is built from (file mtimes do not survive the copy to a GPU box)."""
    import hashlib
    h = hashlib.sha256()
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if os.path.isfile(os.path.join(CSRC, f)))
    for s in srcs + [os.path.join(INCLUDE, "ssp2vit.h")]:
        h.update(os.path.basename(s).encode())
        with open(s, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _needs_rebuild(variant: Optional[str] = None) -> bool:
    path = variant_path(variant)
    if not os.path.exists(path) or not os.path.exists(path + ".srchash"):
        return True
    with open(path + ".srchash") as f:
        return f.read().split()[:1] != [_source_hash()]


TORCH_OPS_PATH = os.path.join(PKG_ROOT, "lib", "libssp2vit_torch.so")


def build_torch_ops(verbose: bool = False) -> str:
    """g++ build of csrc/torch_ops.cpp (TORCH_LIBRARY shim, host code only) -> lib/libssp2vit_torch.so, linked against
    libssp2vit.so (rpath $ORIGIN) and the installed torch.  Rebuilt when its source, the C header or torch changes."""
    import hashlib
    import torch
    from torch.utils import cpp_extension as ce
    src = os.path.join(CSRC, "torch_ops.cpp")
    h = hashlib.sha256()
    for f in (src, os.path.join(INCLUDE, "ssp2vit.h")):
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(torch.__version__.encode())
    want = h.hexdigest()
    stamp = TORCH_OPS_PATH + ".srchash"
    if os.path.exists(TORCH_OPS_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return TORCH_OPS_PATH
    if _needs_rebuild():
        build_library(only_if_stale=True)
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           *[f"-I{p}" for p in ce.include_paths()], "-I/opt/rocm/include", src, "-o", TORCH_OPS_PATH + f".tmp{os.getpid()}",
           f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", f"-L{os.path.dirname(LIB_PATH)}", "-lssp2vit",
           "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(cmd[cmd.index("-o") + 1], TORCH_OPS_PATH)
    with open(stamp, "w") as f:
        f.write(want)
    return TORCH_OPS_PATH


_ops_loaded = False


def load_torch_ops():
    """Registers torch.ops.ssp2vit.{forward, act_l2_accum, top1_count} (builds the shim if it is missing or stale)."""
    global _ops_loaded
    import torch
    if not _ops_loaded:
        load()                                           # libssp2vit.so first: the shim links against it
        torch.ops.load_library(build_torch_ops())
        _ops_loaded = True
    return torch.ops.ssp2vit


TOOLS = os.path.join(CSRC, "tools")


def build_tool(name: str, defines=(), out: Optional[str] = None) -> str:
    """Build (or reuse) a micro-benchmark of csrc/tools: <name>.hip -> <out or name>.bin.  Staleness is decided by a
    content hash of the tool's source, EVERY kernel header of csrc/ (the kernels under test live there) and the
    defines — never by file times, which do not survive the copy to a GPU box."""
    import hashlib
    if not any(d.startswith(("-DSSP2_LAB", "-USSP2_LAB")) for d in defines):     # ("-USSP2_LAB": a tool binary with the PRODUCT's kernel forms)
        defines = ("-DSSP2_LAB=1", *defines)             # the micro-benchmarks exercise the lab forms too (gemm_bench epi 15, GEMM_SUSTAIN of the deferred residual)
    src = os.path.join(TOOLS, name + ".hip")
    exe = os.path.join(TOOLS, (out or name) + ".bin")
    h = hashlib.sha256(_source_hash().encode())
    for f in sorted(os.listdir(TOOLS)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            with open(os.path.join(TOOLS, f), "rb") as fh:
                h.update(fh.read())
    h.update(" ".join(defines).encode())
    want = h.hexdigest()
    stamp = exe + ".srchash"
    if os.path.exists(exe) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return exe
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", *defines, src, "-o", exe],
                   check=True, cwd=TOOLS, timeout=900)
    with open(stamp, "w") as f:
        f.write(want)
    return exe


_libs: dict = {}


def load(build_if_missing: bool = True, variant: Optional[str] = None) -> C.CDLL:
    """The product library, or a named build of the same sources (`variant`; default: SSP2_LIB_VARIANT from the environment, for A/B
    runs).  A known variant ("lab") is built like the product when missing or stale; any other must exist (scripts/build_variant.py).
    Every library is checked against this binding's ABI version before a symbol is touched."""
    variant = variant or os.environ.get("SSP2_LIB_VARIANT") or None
    if variant in _libs:
        return _libs[variant]
    path = variant_path(variant)
    if variant and variant not in VARIANT_FLAGS:
        if not os.path.exists(path):
            raise Ssp2Error(f"library variant {variant!r}: {path} does not exist (scripts/build_variant.py builds it)")
    elif _needs_rebuild(variant):
        if not build_if_missing:
            raise Ssp2Error(f"{path} missing or stale; run `python -c 'import __graft_entry__ as g; g.build()'`")
        build_library(only_if_stale=True, variant=variant)
    lib = C.CDLL(path)
    lib.ssp2_abi_version.restype = C.c_int
    if lib.ssp2_abi_version() != ABI_VERSION:
        raise Ssp2Error(f"{path}: ABI {lib.ssp2_abi_version()} but this binding speaks ABI {ABI_VERSION} — a stale variant build; rebuild it")
    missing = [n for n in SYMBOLS if not hasattr(lib, n)]
    if missing:
        raise Ssp2Error(f"{path} lacks {missing}: a stale variant build; rebuild it")
    vp, i32, i64p = C.c_void_p, C.c_int, C.POINTER(C.c_int64)
    lib.ssp2_abi_version.restype = i32
    lib.ssp2_last_error.restype = C.c_char_p
    lib.ssp2_create.argtypes = [C.POINTER(VitDesc), C.POINTER(vp)]
    lib.ssp2_destroy.argtypes = [vp]
    lib.ssp2_set_stream.argtypes = [vp, vp]
    lib.ssp2_load_tensor.argtypes = [vp, i32, i32, C.POINTER(C.c_float), C.c_size_t]
    lib.ssp2_load_tensor_dev.argtypes = [vp, i32, i32, vp, C.c_size_t]
    lib.ssp2_load_tensors_dev.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.ssp2_linear_bf16.argtypes = [vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, vp, i32, vp, i32, i32]
    lib.ssp2_embed.argtypes = [vp, vp, i32, vp, i32]
    lib.ssp2_rows.argtypes = [vp, i32, i32]
    lib.ssp2_rows.restype = C.c_long
    lib.ssp2_layers.argtypes = [vp, vp, i32, i32, i32, C.POINTER(C.c_uint8), i32, i32, i32, vp, i32]
    lib.ssp2_layers_from.argtypes = [vp, vp, vp, i32, i32, i32, C.POINTER(C.c_uint8), i32, i32, i32, vp, i32]
    lib.ssp2_layers_prefix.argtypes = [vp, vp, vp, i32, i32, i32, C.POINTER(C.c_uint8), i32, i32, i32, i32, vp, i32]
    lib.ssp2_head.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp]
    lib.ssp2_tail.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp]
    lib.ssp2_tail_slots.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp]
    lib.ssp2_tail_group.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]
    lib.ssp2_prune_ffn.argtypes = [vp, i32, C.POINTER(C.c_int32), i32]
    lib.ssp2_drop_attention.argtypes = [vp, i32]
    lib.ssp2_restore_attention.argtypes = [vp, i32]
    lib.ssp2_clone_weights.argtypes = [vp, vp]
    lib.ssp2_prune_ffn_into.argtypes = [vp, vp, i32, C.POINTER(C.c_int32), i32]
    lib.ssp2_d_int.argtypes = [vp, i32]
    lib.ssp2_act_l2_accum.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, C.c_size_t]
    lib.ssp2_profile_begin.argtypes = [vp, i32]
    lib.ssp2_profile_end.argtypes = [vp, C.POINTER(C.c_double), i64p, C.POINTER(C.c_double)]
    lib.ssp2_profile_query.argtypes = [vp, i32, C.POINTER(C.c_double), i64p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.ssp2_set_option.argtypes = [vp, i32, i32]
    lib.ssp2_get_option.argtypes = [vp, i32]
    lib.ssp2_tokens.argtypes = [vp]
    lib.ssp2_query.argtypes = [vp, i32]
    lib.ssp2_set_precision.argtypes = [vp, i32]
    lib.ssp2_set_cu_limit.argtypes = [vp, i32]
    lib.ssp2_fp8_calibrate_begin.argtypes = [vp]
    lib.ssp2_fp8_calibrate_end.argtypes = [vp, C.c_float]
    lib.ssp2_fp8_attn_scale.argtypes = [vp, i32]
    lib.ssp2_fp8_attn_scale.restype = C.c_float
    lib.ssp2_fp8_set_attn_scale.argtypes = [vp, i32, C.c_float]
    lib.ssp2_workspace_bytes.argtypes = [vp]
    lib.ssp2_workspace_bytes.restype = C.c_size_t
    lib.ssp2_preproc_create.argtypes = [i32, i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(vp)]
    lib.ssp2_preproc_run.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp]
    lib.ssp2_preproc_destroy.argtypes = [vp]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("ssp2_last_error", "ssp2_workspace_bytes", "ssp2_rows", "ssp2_fp8_attn_scale"):
            fn.restype = i32
    _libs[variant] = lib
    return lib


def check(rc: int, lib: Optional[C.CDLL] = None) -> None:
    """`lib`: the library the failing call went to (every build keeps its own last-error text); default: the product library."""
    if rc != 0:
        msg = (lib or load()).ssp2_last_error()
        raise Ssp2Error(f"libssp2vit error {rc}: {msg.decode() if msg else '?'}")
