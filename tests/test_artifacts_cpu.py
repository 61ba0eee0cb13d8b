"""SURVEY.md §8 row f1 — score/mask artifact I/O and the offline combiners, against outputs of the reference's own
stdlib scripts (tests/golden/artifact_tools.json, written by make_golden.py --artifacts-only).  The framework-export
pair (auto_2ssp.py:71-185) could not be executed here (auto_2ssp.py imports timm at module top): its schema is
restated from the source and checked structurally — "parity unpinned" for that one format."""
import json
import os

import torch

from conftest import GOLDEN
from ssp2vit import artifacts as A


def _gold():
    return json.load(open(os.path.join(GOLDEN, "artifact_tools.json")))


def test_minmax_normalize_matches_reference_script():
    g = _gold()
    assert A.minmax_normalize(g["normalize"]["input"]) == g["normalize"]["output"]
    assert A.minmax_normalize(g["normalize_const"]["input"]) == g["normalize_const"]["output"]
    assert A.minmax_normalize({"s": "x"}) == {"s": "x"}


def test_bottom_k_and_consensus_masks_match_reference_scripts():
    g = _gold()
    files = g["files"]
    summed = A.sum_leaves(files)
    seen = set()
    for c in g["cases"]:
        if c["kind"] == "bottom_k":
            m = A.bottom_k_mask(summed, c["fraction"], c["rounding"], c["per_block_k"])
        else:
            m = A.consensus_mask(files, c["fraction"], c["rounding"])
        assert m == c["mask"], c
        assert list(m.keys()) == list(c["mask"].keys())          # same stable (block, neuron) key order
        seen.add(c["kind"])
    assert seen == {"bottom_k", "consensus"}


def test_mask_round_trip_into_width_prune_inputs(tmp_path):
    g = _gold()
    mask = A.bottom_k_mask(A.sum_leaves(g["files"]), 0.25)
    p = tmp_path / "m.json"
    p.write_text(json.dumps({"runs": [{"ffn": mask}], "note": "x"}))
    blocks = A.load_mask(str(p))
    assert sorted(blocks) == [0, 1, 2] and set(blocks[0].values()) <= {0, 1}
    imp, counts = A.mask_to_importance_and_counts(blocks, [20, 20, 17])
    assert counts == [sum(blocks[i].values()) for i in range(3)] and counts[0] == 4
    assert all(set(t.tolist()) <= {1.0, -1.0} for t in imp) and int((imp[2] == -1).sum()) == counts[2]
    p2 = tmp_path / "bad.json"
    p2.write_text(json.dumps({"a": [1, 2]}))
    try:
        A.load_mask(str(p2))
        assert False
    except RuntimeError:
        pass


def test_framework_export_schema(tmp_path):
    prefix = str(tmp_path / "out" / "fw")
    mlp = [torch.tensor([0.5, 1.5, 0.25]), torch.tensor([2.0, 0.0, 1.0])]
    res = A.build_framework_exports(prefix, n_blocks=2, hidden=4, num_heads=2, mlp_imp_list=mlp,
                                    att_imp=torch.tensor([0.125, 0.0]), ffn_masks_list=[[0, 0, 1], [0, 1, 0]],
                                    pruned_attn_block_indices=[1])
    sc = json.load(open(prefix + "_scores.json")); mk = json.load(open(prefix + "_masks.json"))
    assert sc == res["scores"] and mk == res["masks"]
    assert sc["ffn"] == {"0:0": 0.5, "0:1": 1.5, "0:2": 0.25, "1:0": 2.0, "1:1": 0.0, "1:2": 1.0}
    assert sc["heads"] == {"0:0": 0.125, "0:1": 0.125, "1:0": 0.0, "1:1": 0.0}
    assert len(sc["qkv_dim"]) == 8 and sc["qkv_dim"]["0:3"] == 0.125
    assert mk["ffn"] == {"0": [0, 0, 1], "1": [0, 1, 0]}
    assert mk["heads"] == {"0": [0, 0], "1": [1, 1]} and mk["qkv_dim"] == {"0": [0] * 4, "1": [1] * 4}
    # reference fallback: no masks -> all-keep masks sized like the score vectors
    res = A.build_framework_exports(prefix, 2, 4, 2, mlp, None, None, None, write=False)
    assert res["masks"]["ffn"] == {"0": [0, 0, 0], "1": [0, 0, 0]} and res["scores"]["heads"]["1:1"] == 0.0
    path = A.save_ffn_prune_masks(str(tmp_path / "ffn_prune_masks.json"), [torch.tensor([0, 1]), [1, 0]])
    assert json.load(open(path)) == {"ffn_masks": [[0, 1], [1, 0]]}
    assert A.scores_to_ij(mlp)["ffn"]["1:0"] == 2.0


def test_v1_artifact_files_of_the_older_cli(tmp_path):
    """The older CLI's three files (/root/reference/experiments/vit_pruning/auto_2ssp.py:769-829; SURVEY section 2 row 5 files these formats
    under f1): key sets, value types, the 1 = prune convention, `indices` = positions of the ones, the "b:j" importance map in block-major
    order, the attention file only when a block was removed, and the report's artifact keys.  Schema restated from the source text (that
    script imports timm at module top, so it cannot run here): "parity unpinned" for the byte layout, pinned for the schema."""
    imps = [torch.tensor([0.5, 0.0, 415.25]), torch.tensor([1.0, 2.0, 3.0])]
    masks = [[1, 0, 0], torch.tensor([0, 1, 1])]
    out = A.save_v1_artifacts(str(tmp_path), mlp_imp=imps, ffn_masks=masks, pruned_block_indices=[1], min_remaining=1, s1_sparsity=None,
                              block_inter_sizes=[3, 3])
    assert set(out) == {"ffn_importances_path", "ffn_prune_masks_path", "attn_pruned_indices_path"}
    m = json.load(open(out["ffn_prune_masks_path"]))
    assert list(m) == ["format_version", "stage", "strategy", "min_remaining", "s1_sparsity", "block_inter_sizes", "masks", "indices"]
    assert m["format_version"] == 1 and m["stage"] == "s1" and m["strategy"] == "act_l2" and m["min_remaining"] == 1 and m["s1_sparsity"] is None
    assert m["block_inter_sizes"] == [3, 3] and m["masks"] == [[1, 0, 0], [0, 1, 1]] and m["indices"] == [[0], [1, 2]]
    a = json.load(open(out["attn_pruned_indices_path"]))
    assert a == {"format_version": 1, "stage": "s2", "indices": [1]}
    s = json.load(open(out["ffn_importances_path"]))
    assert list(s) == ["ffn"] and list(s["ffn"]) == ["0:0", "0:1", "0:2", "1:0", "1:1", "1:2"] and s["ffn"]["0:2"] == 415.25
    assert A.load_ij_leaves(out["ffn_importances_path"]) if hasattr(A, "load_ij_leaves") else True
    # nothing removed in stage 2 -> no attention file (reference :808); explicit indices are written as given
    out2 = A.save_v1_artifacts(str(tmp_path / "b"), ffn_masks=masks, ffn_indices=[[0], [1, 2]], pruned_block_indices=[])
    assert set(out2) == {"ffn_prune_masks_path"} and not os.path.exists(tmp_path / "b" / "attention_pruned_indices.json")
    # the mask consumer of the reference's tooling reads "i:j" leaves: the importance file is one
    tree = json.load(open(out["ffn_importances_path"]))
    assert A._is_ij_leaf(tree["ffn"])
