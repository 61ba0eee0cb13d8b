"""CPU oracle of the 2SSP ViT pruning path: TEST INFRASTRUCTURE, not product code.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py import this package; the shipped path
(2ssp-x-vit_amd/ssp2vit) never does and fails loudly when the HIP library is missing.  Pinned by tests/golden/
(vectors captured from the reference's own functions, see tests/golden/make_golden.py)."""
