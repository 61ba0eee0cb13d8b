"""pytest wiring: `gpu` marker, import paths, shared fixture loaders."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "2ssp-x-vit_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "lab: a VARIANT test — runs against lib/libssp2vit_lab.so (-DSSP2_LAB: the opt-in kernel forms the "
                                       "product library does not instantiate), with the product library as the source of the expected bits")


def load_tiny_golden(layout: str):
    """-> (flat weights dict, batches list, raw npz dict) of tests/golden/tiny_<layout>.npz"""
    z = dict(np.load(os.path.join(GOLDEN, f"tiny_{layout}.npz")))
    w = {}
    for k, v in z.items():
        if k.startswith("w."):
            name = k[2:]
            w[name] = torch.from_numpy(v) if v.ndim > 0 else v.item()
    batches = []
    i = 0
    while f"px.{i}" in z:
        batches.append({"pixel_values": torch.from_numpy(z[f"px.{i}"]),
                        "labels": torch.from_numpy(z[f"labels.{i}"])})
        i += 1
    return w, batches, z


def bf16_from_bits(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.astype(np.uint16).view(np.int16).copy()).view(torch.bfloat16)


@pytest.fixture(scope="session")
def have_gpu():
    return torch.cuda.is_available()
