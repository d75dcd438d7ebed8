"""Host side of the sequence loader (no GPU): sample index, sweep order, pose products, per-box
arithmetic, prepare_data and collate_batch of the product against the reference's own outputs
(tests/golden/loader_*.npz)."""
import os

import numpy as np
import pytest

from tests import loader_fixture as LF

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
OBJECT_KEYS = ("gt_box_cls_label", "gt_box_attr", "augmented", "num_points_in_gt", "gt_box_corners_3d", "gt_boxes")


@pytest.mark.parametrize("name", ["loader_fwd", "loader_bwd"])
def test_loader_host_logic_vs_reference(name, tmp_path):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    ds = LF.build_dataset(LF.write_sequence(tmp_path, g), g)
    assert len(ds) == 1
    sweeps = ds.sweep_infos(0)
    want_frames = g["out_frame_id"][0]
    assert [s["frame_id"] for s in sweeps] == list(want_frames)
    poses, Ts = ds.sweep_transforms(sweeps)
    np.testing.assert_array_equal(np.stack(poses), g["out_pose"][0])
    np.testing.assert_array_equal(Ts[-1], np.linalg.inv(poses[-1]) @ poses[-1])
    for s in sweeps:  # only lidar 0 is kept: one leading run
        sel = ds.kept_rows(s, 10 ** 6)
        assert sel == slice(0, s["num_points_of_each_lidar"][0])
    import copy
    object_wise, box_list = ds.assemble_objects([copy.deepcopy(s) for s in sweeps], Ts)
    sample = ds.prepare_data(dict(point_wise={}, object_wise=object_wise,
                                  scene_wise=dict(frame_id=np.stack([np.asarray(s["frame_id"]) for s in sweeps]),
                                                  pose=np.stack(poses))))
    sample["scene_wise"]["num_sweeps"] = len(sweeps)
    batch = ds.collate_batch([sample])
    for k in OBJECT_KEYS + ("pose", "num_sweeps", "batch_size"):
        want = g["out_" + k]
        got = np.asarray(batch[k])
        assert got.dtype == want.dtype, (k, got.dtype, want.dtype)
        np.testing.assert_array_equal(got, want, err_msg=k)
    np.testing.assert_array_equal(np.asarray(batch["obj_ids"][0]).astype(str), g["out_obj_ids"])
    np.testing.assert_array_equal(batch["frame_id"].astype(str), g["out_frame_id"])
    # every sweep's boxes in its own frame (is_foreground) are the filtered inputs
    for s, b in zip(sweeps, box_list):
        f = s["point_cloud"]["sample_idx"]
        keep = ~np.isin(g[f"names_{f}"], ["unknown", "Sign"])
        np.testing.assert_array_equal(b, g[f"boxes_{f}"][keep])


def test_loader_refuses_what_it_does_not_implement(tmp_path):
    g = np.load(os.path.join(GOLDEN, "loader_fwd.npz"))
    root = LF.write_sequence(tmp_path, g)
    from pcseqlearning_amd.pcdet.datasets import WaymoDataset
    cfg = LF.dataset_cfg(root, 5, 1, [0, 1])
    cfg["SPHERICAL_RESAMPLING"] = True  # (SYNC_MOVING_POINTS, WITH_TIME_FEAT, REMOVE_GROUND, FILTER_FOREGROUND are in)
    with pytest.raises(NotImplementedError):
        WaymoDataset(cfg, training=False, root_path=root)
    with pytest.raises(NotImplementedError):
        WaymoDataset(LF.dataset_cfg(root, 5, 1, [0, 1]), training=True, root_path=root)
    ds = WaymoDataset(LF.dataset_cfg(root, 5, 1, [0, 1]), training=False, root_path=root)
    with pytest.raises(RuntimeError):  # the point path needs the HIP library: no CPU fallback
        ds.load_sample_to_gpu(0, "cpu")


def test_collate_batch_keeps_tensors_where_they_are():
    """The device path hands collate_batch torch tensors (here on the CPU): concatenation and the
    `point_bxyz` pass-through must give what the numpy path gives."""
    import torch
    from pcseqlearning_amd.pcdet.datasets import DatasetTemplate
    rng = np.random.default_rng(0)

    def sample(n, b, as_tensor):
        bxyz = np.concatenate([np.full((n, 1), b, np.float32), rng.normal(size=(n, 3)).astype(np.float32)], 1)
        pw = dict(point_bxyz=bxyz, point_feat=rng.normal(size=(n, 3)).astype(np.float32),
                  point_sweep=rng.integers(0, 5, (n, 1)).astype(np.int32))
        if as_tensor:
            pw = {k: torch.from_numpy(v) for k, v in pw.items()}
        ow = dict(gt_box_attr=rng.normal(size=(b + 2, 7)).astype(np.float32), obj_ids=np.array(["a"] * (b + 2)))
        return dict(point_wise=pw, object_wise=ow, scene_wise=dict(frame_id=np.array(["f0", "f1"])))

    rng = np.random.default_rng(0)
    a = DatasetTemplate.collate_batch([sample(5, 0, False), sample(7, 1, False)])
    rng = np.random.default_rng(0)
    b = DatasetTemplate.collate_batch([sample(5, 0, True), sample(7, 1, True)])
    assert a["batch_size"] == b["batch_size"] == 2 and a["gt_box_attr"].shape == (2, 3, 7)
    for k in ("point_bxyz", "point_feat", "point_sweep"):
        assert torch.is_tensor(b[k]) and isinstance(a[k], np.ndarray)
        np.testing.assert_array_equal(a[k], b[k].numpy())
    np.testing.assert_array_equal(a["gt_box_attr"], b["gt_box_attr"])
    assert a["gt_box_attr"][0, 2].sum() == 0  # padded row of the sample with fewer boxes
