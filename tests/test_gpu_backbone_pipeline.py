"""VoxelBackBone8x on the HIP sparse conv vs a layer-by-layer oracle chain; ClusterProposal and
the SimpleReg subsample vs the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from pcseqlearning_amd import ops, synthetic  # noqa: E402
from pcseqlearning_amd.pcdet.models.backbones_3d import VoxelBackBone8x  # noqa: E402
from pcseqlearning_amd.pcdet.models.backbones_3d.vfe import DynamicMeanVFE  # noqa: E402
from pcseqlearning_amd.pcdet.utils.common_utils import EasyDict  # noqa: E402

DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _bn_relu(x, bn):
    w, b = bn.weight.detach().cpu().numpy(), bn.bias.detach().cpu().numpy()
    m, v = bn.running_mean.cpu().numpy(), bn.running_var.cpu().numpy()
    return np.maximum((x - m) / np.sqrt(v + bn.eps) * w + b, 0).astype(np.float32)


@pytest.mark.parametrize("n_points", [30000, 160000])
def test_voxel_backbone_8x_vs_oracle(oracle, n_points):
    """BASELINE configs[4] (a reduced frame and the full 160k-pt frame): dynamic VFE -> 12 sparse convs /
    8 rulebooks, eval BN, against the oracle chain: indices bit-exact at every scale, features 1e-4."""
    torch.manual_seed(0)
    pts = synthetic.gen_voxel_frame(2, n_points)
    vfe = DynamicMeanVFE({}, 5, synthetic.VOXEL_SIZE, synthetic.GRID_SIZE, synthetic.PC_RANGE)
    net = VoxelBackBone8x({}, 5, synthetic.GRID_SIZE.tolist()).to(DEV).eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
    batch = vfe({"points": dev(pts), "batch_size": 1})
    with torch.no_grad():
        out = net(batch)
    # oracle chain
    feat, coords, _ = oracle.dynamic_voxelize(pts, synthetic.PC_RANGE, synthetic.VOXEL_SIZE, synthetic.GRID_SIZE)
    np.testing.assert_array_equal(batch["voxel_coords"].cpu().numpy(), coords)
    x, idx, shape = feat, coords, [41, 1504, 1504]
    cache = {}

    def conv(seq, x, idx, shape, ks, st, pd, subm, key):
        convm, bn = seq[0], seq[1]
        if key in cache:
            out_idx, pairs, num, oshape = cache[key]
        else:
            oshape = shape if subm else oracle.conv_out_shape(shape, ks, st, pd, [1, 1, 1])
            out_idx, pairs, num = oracle.get_indice_pairs(idx, oshape, ks, st, pd, [1, 1, 1], subm)
            cache[key] = (out_idx, pairs, num, oshape)
        y = oracle.indice_conv(x, convm.weight.detach().cpu().numpy(), pairs, num, out_idx.shape[0], False, subm)
        return _bn_relu(y, bn), out_idx, oshape

    k3, s1, s2, p1 = [3, 3, 3], [1, 1, 1], [2, 2, 2], [1, 1, 1]
    x, idx, shape = conv(net.conv_input, x, idx, shape, k3, s1, p1, True, "subm1")
    x, idx, shape = conv(net.conv1[0], x, idx, shape, k3, s1, p1, True, "subm1")
    stages = {}
    stages["x_conv1"] = (x, idx)
    for name, stage, pad in (("x_conv2", net.conv2, p1), ("x_conv3", net.conv3, p1), ("x_conv4", net.conv4, [0, 1, 1])):
        x, idx, shape = conv(stage[0], x, idx, shape, k3, s2, pad, False, name + "_down")
        x, idx, shape = conv(stage[1], x, idx, shape, k3, s1, p1, True, name + "_subm")
        x, idx, shape = conv(stage[2], x, idx, shape, k3, s1, p1, True, name + "_subm")
        stages[name] = (x, idx)
    x, idx, shape = conv(net.conv_out, x, idx, shape, [3, 1, 1], [2, 1, 1], [0, 0, 0], False, "down2")
    for name, (ox, oidx) in stages.items():
        t = out["multi_scale_3d_features"][name]
        np.testing.assert_array_equal(t.indices.cpu().numpy(), oidx)
        np.testing.assert_allclose(t.features.cpu().numpy(), ox, rtol=1e-4, atol=1e-4)
    enc = out["encoded_spconv_tensor"]
    assert enc.spatial_shape == [2, 188, 188] and shape == [2, 188, 188]
    np.testing.assert_array_equal(enc.indices.cpu().numpy(), idx)
    np.testing.assert_allclose(enc.features.cpu().numpy(), x, rtol=1e-4, atol=1e-4)


def test_cluster_proposal_and_subsample_vs_oracle(oracle):
    from oracle import track_oracle as T
    from pcseqlearning_amd.pcdet.models.registration.preprocessors.cluster_proposal import ClusterProposal
    from pcseqlearning_amd.pcdet.models.model_utils.grid_sampling import GridSampling3D
    batch = synthetic.gen_sequence(4, 12, 8000, n_static=14, n_vehicle=8, n_ped=4)
    fxyz = np.concatenate([batch["point_sweep"].astype(np.float32), batch["point_bxyz"][:, 1:]], 1)
    # SimpleReg subsample: max original index per occupied 8 cm cell, in cell-rank order
    _, inv, order, seg = oracle.grid_sample(fxyz, [0.08, 0.08, 0.08])
    want_pick = order[seg[1:] - 1]
    sampler = GridSampling3D([0.08, 0.08, 0.08])
    pick = ops.segment_pick(sampler.segments(dev(fxyz)), last=True).cpu().numpy()
    np.testing.assert_array_equal(pick, want_pick)
    keep = batch["segmentation_label"][pick] != 17
    pts = fxyz[pick][keep]
    cfg = EasyDict(GRAPH=dict(TYPE="RadiusGraph", RADIUS=[1.25, 0.75, 0.25], MAX_NUM_NEIGHBORS=32,
                              SORT_BY_DIST=True, RELATIVE_KEY="fxyz"),
                   COMPONENT_KEYS=["component_rad1x25", "component_rad0x75", "component_rad0x25"], DIR="/tmp/x")
    prop = ClusterProposal(cfg, {})
    seq = {"point_fxyz": dev(pts), "point_sweep": dev(pts[:, :1].astype(np.int64)), "frame_id": ["seq_000"]}
    seq = prop.propose_cluster(seq)
    for key, r in zip(cfg.COMPONENT_KEYS, cfg.GRAPH["RADIUS"]):
        want = T.propose_clusters(pts, pts[:, 0].astype(np.int64), r)
        np.testing.assert_array_equal(seq[f"point_{key}"].cpu().numpy(), want)


def test_points_in_boxes_vs_oracle(oracle):
    rng = np.random.default_rng(1)
    pts = rng.uniform(-10, 10, (5000, 3)).astype(np.float32)
    boxes = np.concatenate([rng.uniform(-8, 8, (40, 3)), rng.uniform(0.5, 5, (40, 3)), rng.uniform(-4, 4, (40, 1))], 1).astype(np.float32)
    got = ops.points_in_boxes(dev(pts), dev(boxes)).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.points_in_boxes(pts, boxes))


def test_full_pipeline_lockstep_vs_unit_by_unit():
    """SimpleReg end to end on a small sequence: all tracking units in lock-step in one native call,
    the same twice (run to run), one unit per native call, and the op-by-op host loop must all give
    bit-identical results."""
    from pcseqlearning_amd.pcdet import models
    from tools.run_sequence import build_model
    batch = synthetic.gen_sequence(5, 10, 12000, n_static=14, n_vehicle=8, n_ped=4)
    batch = models.load_data_to_gpu(batch, DEV)
    outs = []
    for per_call, native in ((0, True), (0, True), (1, True), (4, True), (0, False)):
        model = build_model(save=False, device=DEV, units_per_call=per_call)
        model.preprocessors[2].native = native
        with torch.no_grad():
            model(batch)
        outs.append(model.last_outputs[0])
    ref = outs[0]["cluster_tracking_results"]
    assert len(ref) == 6 and any(v["fxyz"].shape[0] > 0 for v in ref.values())
    for other in outs[1:]:
        res = other["cluster_tracking_results"]
        assert res.keys() == ref.keys()
        for key in ref:
            for field in ("fxyz", "component", "original_indices", "frame_indices", "moving", "component_hit"):
                if field in ref[key]:
                    assert torch.equal(ref[key][field], res[key][field]), (key, field)
        assert torch.equal(outs[0]["gt_box_best_iou_tracking"], other["gt_box_best_iou_tracking"])
        for k in ("point_component_rad1x25", "point_component_rad0x25", "point_gt_box_id"):
            assert torch.equal(outs[0][k], other[k])


def test_extraction_of_all_units_together_vs_unit_by_unit():
    """ClusterTracking's tail on the native tracker's outputs: the final validity filter and the extraction of traces
    for all units of a sequence at once (EXTRACT_UNITS_TOGETHER, the default: gathers and per-unit statistics on the
    concatenated rows) against the unit-by-unit form (`_extracted_from_native` per unit, then
    `extract_traces_of_units`): every field of every unit and the boxes' best IoUs identical."""
    from pcseqlearning_amd.pcdet import models
    from tools.run_sequence import build_model
    batch = models.load_data_to_gpu(synthetic.gen_sequence(9, 10, 12000, n_static=14, n_vehicle=8, n_ped=4), DEV)
    outs = []
    for together in (True, False):
        model = build_model(save=False, device=DEV)
        model.preprocessors[2].model_cfg["EXTRACT_UNITS_TOGETHER"] = together
        with torch.no_grad():
            model(batch)
        outs.append(model.last_outputs[0])
    a, b = outs[0]["cluster_tracking_results"], outs[1]["cluster_tracking_results"]
    assert a.keys() == b.keys() and any(v["fxyz"].shape[0] > 0 for v in a.values())
    for key in a:
        assert set(a[key].keys()) == set(b[key].keys()), key
        for field, t in a[key].items():
            if torch.is_tensor(t):
                assert t.dtype == b[key][field].dtype and torch.equal(t, b[key][field]), (key, field)
    assert torch.equal(outs[0]["gt_box_best_iou_tracking"], outs[1]["gt_box_best_iou_tracking"])
    # no component survives the final filter (MIN_MOVE_FRAME beyond the sequence): every unit comes back with its own
    # empty dictionary in both forms
    empty = []
    for together in (True, False):
        model = build_model(save=False, device=DEV)
        model.preprocessors[2].model_cfg["EXTRACT_UNITS_TOGETHER"] = together
        model.preprocessors[2].min_move_frame = 100
        with torch.no_grad():
            model(batch)
        empty.append(model.last_outputs[0]["cluster_tracking_results"])
    assert empty[0].keys() == empty[1].keys()
    for key in empty[0]:
        assert empty[0][key]["fxyz"].shape[0] == 0 and set(empty[0][key].keys()) == set(empty[1][key].keys())


def test_batch_of_sequences_concurrently_vs_one_by_one():
    """SimpleReg.forward on a batch of three sequences: processed at once (SEQUENCES_IN_FLIGHT 3: one host
    thread, stream and preprocessor copy per sequence, one lock-step group each) and one after the other
    (SEQUENCES_IN_FLIGHT 1, four groups) -- every sequence's outputs bit-identical."""
    from pcseqlearning_amd.pcdet import models
    from tools.run_sequence import build_model
    seqs = [synthetic.gen_sequence(s, 10, 9000, n_static=12, n_vehicle=6, n_ped=3) for s in (5, 6, 7)]
    batch = models.load_data_to_gpu(synthetic.collate_sequences(seqs), DEV)
    outs = []
    for inflight in (3, 1):
        model = build_model(save=False, device=DEV)
        model.model_cfg["SEQUENCES_IN_FLIGHT"] = inflight
        with torch.no_grad():
            model(batch)
        outs.append(model.last_outputs)
    assert len(outs[0]) == len(outs[1]) == 3
    for a, b in zip(*outs):
        ra, rb = a["cluster_tracking_results"], b["cluster_tracking_results"]
        assert ra.keys() == rb.keys() and len(ra) == 6
        assert any(v["fxyz"].shape[0] > 0 for v in ra.values())
        for key in ra:
            for field in ("fxyz", "component", "original_indices", "frame_indices", "moving", "component_hit"):
                if field in ra[key]:
                    assert torch.equal(ra[key][field], rb[key][field]), (key, field)
        for k in ("gt_box_best_iou_tracking", "point_component_rad1x25", "point_component_rad0x25", "point_gt_box_id",
                  "point_height"):
            assert torch.equal(a[k], b[k]), k
    # the three sequences differ from each other (the batch index really selects them)
    assert outs[0][0]["point_fxyz"].shape != outs[0][1]["point_fxyz"].shape
