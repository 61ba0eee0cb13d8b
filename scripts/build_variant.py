#!/usr/bin/env python3
"""Builds lib/libssp2vit_<name>.so = the product library with extra -D flags (compile-time A/B switches of the kernels), for same-box
step A/Bs:  SSP2_LIB_VARIANT=<name> python bench.py ...   (ssp2vit/_lib.py).  Cross-compiles here, travels to the GPU box.
    python scripts/build_variant.py ln_nt -DLN_NT_LOADS=1
The named variant "lab" (-DSSP2_LAB=1: LayerNorm behind the residual epilogue, deferred residual, column-group tile orders) is built by
__graft_entry__.build() itself (ssp2vit/_lib.py VARIANT_FLAGS); add -DSSP2_LAB=1 here when an A/B needs those forms in another variant."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "2ssp-x-vit_amd", "lib", f"libssp2vit_{name}.so")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", *flags,
       os.path.join(ROOT, "2ssp-x-vit_amd", "csrc", "engine.hip"), "-o", out]
subprocess.run(cmd, check=True, cwd=os.path.join(ROOT, "2ssp-x-vit_amd", "csrc"))
print(out)
