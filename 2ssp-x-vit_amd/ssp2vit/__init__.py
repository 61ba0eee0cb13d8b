"""ssp2vit — MI355X-native hot path of 2SSP pruning for Vision Transformers.

Python host layer over libssp2vit.so (C ABI in include/ssp2vit.h).  Mirrors the reference's function API
(`ssp2vit.vit_pruning`) and plug-in class (`ssp2vit.mask_conjunction.Auto2SSPInterface`).
Importing the package does not need a GPU; running any scoring / evaluation entry point does.
"""
from .weights import VIT_CONFIGS, synthetic_weights, from_module  # noqa: F401
from .planner import TwoSSPPlan, plan_from_stats, stats_from_shapes  # noqa: F401

__version__ = "0.1.0"
