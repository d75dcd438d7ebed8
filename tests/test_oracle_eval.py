"""Pins oracle/eval_oracle.py (evaluate_proposal, extract_traces_and_update_boxes, forward) to golden
vectors produced by the reference's own code (tests/golden/make_golden_forward.py). CPU."""
import os

import numpy as np

G = os.path.join(os.path.dirname(__file__), "golden")
KEYS = ["component_rad1x25", "component_rad0x75", "component_rad0x25"]


def load():
    return np.load(os.path.join(G, "forward_small.npz"))


def boxes_of(g):
    return dict(attr=g["in_gt_box_attr"], frame=g["in_gt_box_frame"], trace_id=g["in_gt_box_track_label"],
                cls_label=g["in_gt_box_cls_label"], moving=g["in_moving"])


def all_points_of(g):
    hm = g["in_full_point_height"] > 0.0
    return dict(fxyz=g["in_full_point_fxyz"][hm], frame=g["in_full_point_sweep"].reshape(-1)[hm],
                full_segmentation_label=g["in_full_segmentation_label"][hm],
                full_instance_label=g["in_full_instance_label"][hm])


def test_evaluate_proposal_vs_reference():
    from oracle import eval_oracle as E
    g = load()
    out = E.evaluate_proposal(g["in_point_fxyz"], [g[f"in_point_{k}"] for k in KEYS], boxes_of(g))
    for k in ("point_gt_box_id", "point_gt_trace_id", "point_pred_trace_id", "point_pred_box_id"):
        np.testing.assert_array_equal(out[k], g[f"eval_{k}"], err_msg=k)
    assert (out["point_gt_box_id"] >= 0).sum() > 1000 and (out["point_pred_box_id"] >= 0).sum() > 1000
    np.testing.assert_allclose(out["gt_box_best_iou"], g["eval_gt_box_best_iou"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(out["gt_trace_best_iou"], g["eval_gt_trace_best_iou"], rtol=0, atol=1e-6)
    assert (out["gt_box_best_iou"] > 0.5).sum() > 20


def test_evaluate_proposal_without_boxes():
    from oracle import eval_oracle as E
    g = load()
    empty = dict(attr=np.zeros((0, 7), np.float32), frame=np.zeros(0, np.int64), trace_id=np.zeros(0, np.int64))
    out = E.evaluate_proposal(g["in_point_fxyz"][:100], [g["in_point_component_rad1x25"][:100]], empty)
    assert all((out[f"point_{k}"] == -1).all() for k in ("gt_box_id", "gt_trace_id", "pred_trace_id", "pred_box_id"))


def extracted_of(g):
    return {k: g[f"ext_in_{k}"].copy() for k in ("fxyz", "component", "moving", "transforms", "original_indices",
                                                 "frame_indices")}


def test_extract_traces_vs_reference():
    """The stage is deterministic given its input (the reference's own track_frame extraction, stored in
    the golden file): ids bit-exact, IoUs to 1e-6."""
    from oracle import eval_oracle as E
    g = load()
    boxes = boxes_of(g)
    boxes["best_iou"] = np.zeros(boxes["attr"].shape[0], np.float32)
    out = E.extract_traces_and_update_boxes(all_points_of(g), extracted_of(g), boxes, 0.5)
    for k in ("component", "segmentation_label", "instance_label", "original_indices", "frame_indices", "moving",
              "component_hit", "component_size", "fxyz"):
        np.testing.assert_array_equal(out[k], g[f"ext_out_{k}"], err_msg=k)
    assert out["fxyz"].shape[0] > g["ext_in_fxyz"].shape[0] and out["component_hit"].sum() > 10
    np.testing.assert_allclose(boxes["best_iou"], g["ext_out_best_iou"], rtol=0, atol=1e-6)


def seq_of(g):
    seq = {k[3:]: g[k] for k in g.files if k.startswith("in_")}
    seq["boxes"] = boxes_of(g)
    seq["component_keys"] = KEYS
    return seq


def test_forward_vs_reference():
    """The oracle-side forward against what the reference's forward saved. The tracking chain inside is
    chaotic (test_oracle_tracking.py), so units are compared by overlap and the box IoUs by their bulk."""
    from oracle import eval_oracle as E
    g = load()
    saved, boxes = E.forward(seq_of(g))
    assert sorted(saved) == g["fwd_units"].tolist()
    for u in saved:
        want = set(g[f"fwd_{u}_original_indices"].reshape(-1).tolist())
        got = set(saved[u]["original_indices"].reshape(-1).tolist())
        assert len(want & got) / len(want | got) > 0.85, (u, len(want), len(got))
        # (component_hit / component_size have max(tracked component) + 1 rows: chaotic like the ids)
        assert abs(int((saved[u]["component_hit"] > 0).sum()) - int((g[f"fwd_{u}_component_hit"] > 0).sum())) <= 3
    d = np.abs(boxes["best_iou"] - g["fwd_all_best_iou"])
    assert (d < 1e-6).mean() > 0.8 and abs(boxes["best_iou"].mean() - g["fwd_all_best_iou"].mean()) < 0.02, \
        ((d < 1e-6).mean(), boxes["best_iou"].mean(), g["fwd_all_best_iou"].mean())


def test_format_boxes_velocity():
    """format_boxes: a trace moving 1 m per frame has velocity 1 everywhere (the first box takes the
    second's), a parked one 0; empty boxes are dropped; track labels follow the sorted id strings."""
    from oracle import eval_oracle as E
    F = 4
    attr = np.zeros((F, 3, 7), np.float32)
    attr[:, 0] = [0, 0, 0, 4, 2, 1.5, 0.3]
    attr[:, 0, 0] = np.arange(F)
    attr[:, 1] = [10, 5, 0, 1, 1, 2, 0]
    out = E.format_boxes(attr, np.ones((F, 3), np.int64), np.array([["b", "a", "c"]] * F), F)
    assert out["attr"].shape[0] == 2 * F
    np.testing.assert_array_equal(out["frame"], np.repeat(np.arange(F), 2))
    np.testing.assert_array_equal(out["trace_id"], np.tile([1, 0], F))
    np.testing.assert_allclose(out["velo"], np.tile([1.0, 0.0], F), atol=1e-6)
    np.testing.assert_array_equal(out["moving"], np.tile([True, False], F))
