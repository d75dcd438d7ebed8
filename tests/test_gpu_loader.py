"""Sequence loader on the device: `WaymoDataset.load_batch_to_gpu` /
`__getitem__` + `collate_batch` against the reference's own outputs (tests/golden/loader_*.npz),
`pcs_assemble_sweeps` against the loader oracle on a larger cloud, and a pose round trip at the
full sequence size (20 sweeps x 160 k points)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from pcseqlearning_amd import ops  # noqa: E402
from tests import loader_fixture as LF  # noqa: E402

DEV = "cuda:0"
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def ulp_diff_f32(a, b):
    """Distance in representable fp32 values."""
    ia = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    ib = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, np.int64(-2 ** 31) - ia, ia)
    ib = np.where(ib < 0, np.int64(-2 ** 31) - ib, ib)
    return np.abs(ia - ib)


def check_batch(batch, g):
    host = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in batch.items()}
    want_keys = {k[4:] for k in g.files if k.startswith("out_")}
    assert set(host.keys()) == want_keys
    # integer / boolean / string arrays and everything per box: bit-exact
    for k in ("segmentation_label", "instance_label", "point_sweep", "is_foreground", "gt_box_cls_label",
              "gt_box_attr", "augmented", "num_points_in_gt", "gt_box_corners_3d", "gt_boxes", "pose",
              "num_sweeps", "batch_size"):
        want = g["out_" + k]
        got = np.asarray(host[k])
        assert got.dtype == want.dtype and got.shape == want.shape, (k, got.dtype, got.shape, want.dtype, want.shape)
        np.testing.assert_array_equal(got, want, err_msg=k)
    np.testing.assert_array_equal(np.asarray(host["obj_ids"][0]).astype(str), g["out_obj_ids"])
    np.testing.assert_array_equal(np.asarray(host["frame_id"]).astype(str), g["out_frame_id"])
    # aligned coordinates: the f64 products rounded to fp32 -- the reference's dgemm may fuse
    # differently, which can move a result by one fp32 value at most (none observed)
    assert host["point_bxyz"].dtype == np.float32 and host["point_bxyz"].shape == g["out_point_bxyz"].shape
    d = ulp_diff_f32(host["point_bxyz"], g["out_point_bxyz"])
    assert d.max() <= 1 and (d > 0).mean() < 1e-4, (d.max(), (d > 0).mean())
    # features: elongation and range / 75 exact, tanh(intensity) within one fp32 value of np.tanh
    np.testing.assert_array_equal(host["point_feat"][:, 1:], g["out_point_feat"][:, 1:])
    assert ulp_diff_f32(host["point_feat"][:, 0], g["out_point_feat"][:, 0]).max() <= 1


@pytest.mark.parametrize("name", ["loader_fwd", "loader_bwd"])
def test_loader_vs_reference(name, tmp_path):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    ds = LF.build_dataset(LF.write_sequence(tmp_path, g), g)
    batch = ds.load_batch_to_gpu([0], DEV)
    assert batch["point_bxyz"].is_cuda and batch["point_feat"].is_cuda  # never left the device
    check_batch(batch, g)
    # the reference's host contract: __getitem__ + collate_batch
    sample = ds[0]
    assert isinstance(sample["point_wise"]["point_xyz"], np.ndarray) and sample["point_wise"]["point_xyz"].shape[1] == 3
    check_batch(ds.collate_batch([sample]), g)
    # two samples in one batch: batch index column and box padding per sample
    both = ds.load_batch_to_gpu([0, 0], DEV)
    n = g["out_point_bxyz"].shape[0]
    assert both["batch_size"] == 2 and both["point_bxyz"].shape[0] == 2 * n
    assert torch.equal(both["point_bxyz"][:n, 1:], both["point_bxyz"][n:, 1:])
    assert float(both["point_bxyz"][:n, 0].max()) == 0.0 and float(both["point_bxyz"][n:, 0].min()) == 1.0
    assert both["gt_boxes"].shape[0] == 2


def test_loader_with_time_feat(tmp_path):
    """WITH_TIME_FEAT (waymo_dataset.py:625-630): the sweep's frame index / (sweeps - 1) in front of the features,
    against a numpy restatement of those lines applied to the loader's own output without the option."""
    g = np.load(os.path.join(GOLDEN, "loader_fwd.npz"))
    root = LF.write_sequence(tmp_path, g)
    plain = LF.build_dataset(root, g).load_batch_to_gpu([0], DEV)
    timed = LF.build_dataset(root, g, WITH_TIME_FEAT=True).load_batch_to_gpu([0], DEV)
    F = int(g["num_sweeps"])
    assert F > 1
    sweep = plain["point_sweep"].cpu().numpy()
    # (the feature encoder runs AFTER the concatenation and picks columns by their index in src_feature_list,
    #  point_feature_encoder.py:41-55: with the time column in front, the fixture's three used features are
    #  columns 0..2 of [time, intensity, elongation, range, ...] -- the reference's behaviour, reproduced)
    ncol = plain["point_feat"].shape[1]
    want = np.concatenate([sweep.reshape(-1, 1) / (F - 1), plain["point_feat"].cpu().numpy()],
                          axis=-1).astype(np.float32)[:, :ncol]
    got = timed["point_feat"].cpu().numpy()
    assert got.dtype == np.float32 and got.shape == want.shape
    np.testing.assert_array_equal(got, want)
    for k in ("point_bxyz", "point_sweep", "is_foreground"):
        assert torch.equal(timed[k], plain[k])


def test_loader_filter_foreground_and_remove_ground(tmp_path):
    """FILTER_FOREGROUND + REMOVE_GROUND (waymo_dataset.py:479-492): per-frame row filters read from files, against
    the same masks applied to the unfiltered loader output (every point-wise array is row-filtered alike)."""
    g = np.load(os.path.join(GOLDEN, "loader_fwd.npz"))
    root = LF.write_sequence(tmp_path, g)
    plain_ds = LF.build_dataset(root, g)
    plain = plain_ds.load_batch_to_gpu([0], DEV)
    rng = np.random.default_rng(9)
    pred_dir, height_dir = tmp_path / "pred", tmp_path / "height"
    keep_all = []
    for info in plain_ds.sweep_infos(0):  # the sample's sweeps in output order
        seq, idx = info["point_cloud"]["lidar_sequence"], info["point_cloud"]["sample_idx"]
        n_raw = plain_ds.read_lidar_file(seq, idx).shape[0]
        sel = plain_ds.kept_rows(info, n_raw)
        n = len(np.arange(n_raw)[sel])
        pred = rng.integers(0, 6, n).astype(np.int32)
        fg_keep = ~np.isin(pred, [1, 4])
        height = rng.uniform(-0.5, 2.0, int(fg_keep.sum())).astype(np.float32)
        (pred_dir / seq).mkdir(parents=True, exist_ok=True)
        (height_dir / seq).mkdir(parents=True, exist_ok=True)
        np.save(pred_dir / seq / f"{idx:03d}_pred.npy", pred)
        np.save(height_dir / seq / ("%04d_point_height.npy" % idx), height)
        keep = fg_keep.copy()
        keep[np.nonzero(fg_keep)[0][height < 0.3]] = False
        keep_all.append(keep)
    keep_all = np.concatenate(keep_all)
    assert 0 < keep_all.sum() < keep_all.shape[0] == plain["point_bxyz"].shape[0]
    ds = LF.build_dataset(root, g, FILTER_FOREGROUND={"PRED_LABEL_PATH": str(pred_dir), "FILTER_CLASS": [1, 4]},
                          REMOVE_GROUND=True, POINT_HEIGHT_PATH=height_dir, TRUNCATE_GROUND_HEIGHT=0.3)
    got = ds.load_batch_to_gpu([0], DEV)
    m = torch.from_numpy(keep_all).to(DEV)
    for k in ("point_bxyz", "point_feat", "point_sweep", "segmentation_label", "instance_label", "is_foreground"):
        assert torch.equal(got[k], plain[k][m]), k
    # only one of the two
    ds = LF.build_dataset(root, g, FILTER_FOREGROUND={"PRED_LABEL_PATH": str(pred_dir), "FILTER_CLASS": [1, 4]})
    assert ds.load_batch_to_gpu([0], DEV)["point_bxyz"].shape[0] > int(keep_all.sum())


def random_sweeps(rng, F, n_per):
    from tests.golden.make_golden_loader import pose_of
    raws, poses = [], []
    for f in range(F):
        n = n_per + int(rng.integers(0, 100))
        pts = np.zeros((n, 8), dtype=np.float32)
        pts[:, :3] = rng.uniform(-75, 75, (n, 3)) * [1, 1, 0.05]
        pts[:, 3] = rng.gamma(1.0, 0.8, n)
        pts[:, 4] = rng.uniform(0, 1.5, n)
        pts[:, 5] = np.linalg.norm(pts[:, :3], axis=1)
        pts[:, 6:] = rng.uniform(0, 1, (n, 2))
        raws.append(pts)
        poses.append(pose_of(rng, f))
    return raws, poses


def device_inputs(raws, Ts, ids):
    counts = np.array([r.shape[0] for r in raws])
    off = np.zeros(len(raws) + 1, dtype=np.int64)
    np.cumsum(counts, out=off[1:])
    d_raw = torch.from_numpy(np.concatenate(raws, axis=0)).to(DEV)
    d_T = torch.from_numpy(np.ascontiguousarray(np.stack([T[:3, :4] for T in Ts]))).to(DEV)
    return d_raw, torch.from_numpy(off).to(DEV), d_T, torch.from_numpy(np.asarray(ids, dtype=np.int32)).to(DEV), off


def test_assemble_sweeps_vs_oracle():
    from oracle import loader_oracle as LO
    rng = np.random.default_rng(5)
    raws, poses = random_sweeps(rng, 7, 30000)
    T0_inv = np.linalg.inv(poses[-1])
    Ts = [T0_inv @ p for p in poses]
    seg = rng.integers(0, 1000, (sum(r.shape[0] for r in raws), 2)).astype(np.int32)
    ids = [3, 4, 5, 6, 7, 8, 9]
    d_raw, d_off, d_T, d_ids, off = device_inputs(raws, Ts, ids)
    bxyz, feat, sweep, seg_l, inst_l = ops.assemble_sweeps(d_raw, torch.from_numpy(seg).to(DEV), d_off, d_T, d_ids, 2)
    want_xyz, want_feat, want_sweep = [], [], []
    for r, T, i in zip(raws, Ts, ids):
        p = LO.get_lidar(r).astype(np.float32)
        xyz = p[:, :3].copy()
        xyz[:, :3] = xyz[:, :3] @ T[:3, :3].T + T[:3, 3]
        want_xyz.append(xyz)
        want_feat.append(p[:, 3:6])
        want_sweep.append(np.full((r.shape[0], 1), i, dtype=np.int32))
    want_xyz, want_feat = np.concatenate(want_xyz), np.concatenate(want_feat)
    got = bxyz.cpu().numpy()
    assert (got[:, 0] == 2.0).all()
    d = ulp_diff_f32(got[:, 1:], want_xyz)
    assert d.max() <= 1 and (d > 0).mean() < 1e-4, (d.max(), (d > 0).mean())
    f = feat.cpu().numpy()
    np.testing.assert_array_equal(f[:, 1:], want_feat[:, 1:])
    assert ulp_diff_f32(f[:, 0], want_feat[:, 0]).max() <= 1
    np.testing.assert_array_equal(sweep.cpu().numpy(), np.concatenate(want_sweep))
    np.testing.assert_array_equal(seg_l.cpu().numpy(), seg[:, 1])
    np.testing.assert_array_equal(inst_l.cpu().numpy(), seg[:, 0])
    # no labels, empty input
    out = ops.assemble_sweeps(d_raw, None, d_off, d_T, d_ids, 0)
    assert out[3] is None and out[4] is None and torch.equal(out[0][:, 1:], bxyz[:, 1:])
    e = ops.assemble_sweeps(torch.empty(0, 8, device=DEV), None, torch.zeros(2, dtype=torch.int64, device=DEV),
                            d_T[:1].contiguous(), d_ids[:1].contiguous(), 0)
    assert e[0].shape == (0, 4) and e[1].shape == (0, 3)


def test_assemble_sweeps_full_size_round_trip():
    """configs[2] size (20 sweeps x 160 k points): moving every sweep into the last sweep's frame and
    back with the inverse products returns the input to fp32 rounding of ~1e3-metre coordinates; the
    sweep tags follow the offsets; identity poses change nothing."""
    rng = np.random.default_rng(9)
    raws, poses = random_sweeps(rng, 20, 160000)
    T0_inv = np.linalg.inv(poses[-1])
    Ts = [T0_inv @ p for p in poses]
    ids = list(range(20))
    d_raw, d_off, d_T, d_ids, off = device_inputs(raws, Ts, ids)
    bxyz, feat, sweep, _, _ = ops.assemble_sweeps(d_raw, None, d_off, d_T, d_ids, 0)
    N = d_raw.shape[0]
    assert bxyz.shape == (N, 4) and int(sweep.min()) == 0 and int(sweep.max()) == 19
    counts = torch.bincount(sweep[:, 0].long(), minlength=20).cpu().numpy()
    np.testing.assert_array_equal(counts, np.diff(off))
    assert bool((sweep[1:, 0] >= sweep[:-1, 0]).all())
    # back with the inverse transforms
    back_raw = d_raw.clone()
    back_raw[:, :3] = bxyz[:, 1:]
    inv = torch.from_numpy(np.ascontiguousarray(np.stack([np.linalg.inv(T)[:3, :4] for T in Ts]))).to(DEV)
    back = ops.assemble_sweeps(back_raw, None, d_off, inv, d_ids, 0)[0]
    err = (back[:, 1:] - d_raw[:, :3]).abs().max().item()
    assert err < 2e-4, err  # two roundings to fp32 of coordinates up to ~1.1e2 m, rotated lever arm
    # the last sweep is the reference frame: inv(P) @ P is the identity to ~1e-13, so its points move
    # by one fp32 value at most
    lo = int(off[-2])
    assert ulp_diff_f32(bxyz[lo:, 1:].cpu().numpy(), d_raw[lo:, :3].cpu().numpy()).max() <= 1
    eye = torch.eye(4, dtype=torch.float64, device=DEV)[:3].expand(20, 3, 4).contiguous()
    same = ops.assemble_sweeps(d_raw, None, d_off, eye, d_ids, 0)[0]
    assert torch.equal(same[:, 1:], d_raw[:, :3])
