"""Tracking-stage parity against golden vectors produced by the reference's own track_frame /
smooth_velo (tests/golden/make_golden_track.py): extracted point and component ids bit-exact,
per-component transforms within 1e-5."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from pcseqlearning_amd import ops  # noqa: E402
from pcseqlearning_amd.pcdet.models.registration.preprocessors import cluster_tracking as CT  # noqa: E402
from pcseqlearning_amd.pcdet.utils.common_utils import EasyDict  # noqa: E402
from tests.golden.make_golden_track import MODEL_CFG  # noqa: E402

DEV = "cuda:0"
G = os.path.join(os.path.dirname(__file__), "golden")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_smooth_velo_vs_reference():
    g = np.load(os.path.join(G, "smooth_velo.npz"))
    for name in ("fwd", "bwd", "one", "same"):
        a, b = g[f"range_{name}"]
        v = dev(g["velos"].copy())
        out = ops.smooth_velo(v, dev(g["diffs"]), int(a), int(b))
        assert out.data_ptr() == v.data_ptr()  # in place, like the reference's aliasing Parameter
        np.testing.assert_allclose(out.cpu().numpy(), g[f"out_{name}"], rtol=0, atol=2e-5)


def test_smooth_velo_many_components_vs_oracle():
    """The multi-workgroup form (thousands of components: one grid barrier per AdamW step) and the
    single-workgroup forms against the CPU oracle."""
    from oracle import track_oracle as T
    rng = np.random.default_rng(3)
    for C, F, a, b in ((3000, 12, 3, 11), (700, 9, 8, 2), (40000, 6, 1, 4)):
        diffs = (rng.standard_normal((C, F, 3)) * 0.3).astype(np.float32)
        velos = (diffs + rng.standard_normal((C, F, 3)) * 0.05).astype(np.float32)
        want = T.smooth_velo(velos.copy(), diffs, a, b)
        got = ops.smooth_velo(dev(velos.copy()), dev(diffs), a, b)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=2e-5)


def test_smooth_velo_batch_many_multi_workgroup_problems():
    """Fifty-two problems in ONE call -- more than one job list holds (48), so the call spans two sets of
    launches; seven of them run on ~400 workgroups each (2800 workgroups of 256 threads: far more than the
    chip keeps resident, which is why the multi-workgroup form has no barrier: window pass, last-arrival
    decision, replay), some stop in the first window, some need several, some run to num_itr; small
    problems ride along in the one-workgroup form. Results = separate calls, bit for bit."""
    rng = np.random.default_rng(11)
    shapes = [(20000, 20, 4, 14)] * 7 + [(300, 20, 9, 2), (5, 20, 0, 19)] + [(1500 + 37 * i, 17, 8, 8 + 1 + i % 8)
                                                                               for i in range(43)]
    probs, want, iters = [], [], []
    for q, (C, F, a, b) in enumerate(shapes):
        diffs = (rng.standard_normal((C, F, 3)) * (0.3 if q % 3 else 0.01)).astype(np.float32)
        velos = (diffs + rng.standard_normal((C, F, 3)) * (0.05 if q % 3 else 0.002)).astype(np.float32)
        want.append(ops.smooth_velo(dev(velos.copy()), dev(diffs), a, b).cpu().numpy())
        probs.append((dev(velos.copy()), dev(diffs), a, b))
    got = ops.smooth_velo_batch(probs)
    for g_, w_ in zip(got, want):
        np.testing.assert_array_equal(g_.cpu().numpy(), w_)


def test_smooth_velo_multi_workgroup_vs_oracle_step_counts():
    """The multi-workgroup form against the oracle where the stop rule fires in the first window, in a
    later window, and never (num_itr reached): same iterate to 2e-5 means the same number of steps."""
    from oracle import track_oracle as T
    rng = np.random.default_rng(23)
    for scale, noise in ((0.004, 0.001), (0.05, 0.02), (0.3, 0.05)):
        C, F, a, b = 2500, 17, 8, 15
        diffs = (rng.standard_normal((C, F, 3)) * scale).astype(np.float32)
        velos = (diffs + rng.standard_normal((C, F, 3)) * noise).astype(np.float32)
        want = T.smooth_velo(velos.copy(), diffs, a, b)
        got = ops.smooth_velo(dev(velos.copy()), dev(diffs), a, b)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=2e-5)


def _seq_points(g):
    sp = EasyDict(fxyz=dev(g["fxyz"].copy()), frame=dev(g["fxyz"][:, :1].astype(np.int64)),
                  gt_box_id=dev(np.full(len(g["fxyz"]), -1, np.int64)),
                  segmentation_label=dev(g["seg_label"]), component=dev(g["component"]))
    return sp


def test_component_diameter_vs_reference():
    g = np.load(os.path.join(G, "track_small.npz"))
    sp = _seq_points(g)
    diam = CT.component_diameter(sp)
    np.testing.assert_allclose(diam.cpu().numpy(), g["comp_diameter"], rtol=1e-5, atol=1e-5)


def _tracker():
    cfg = EasyDict({k: (EasyDict(v) if isinstance(v, dict) else v) for k, v in MODEL_CFG.items()})
    return CT.ClusterTracking(cfg, {}).to(DEV)


def _prepare(g, native=False):
    sp = _seq_points(g)
    diam = CT.component_diameter(sp)
    sp.component_diameter = diam[sp.component]
    sp.stationary = sp.component_diameter > 12.5
    sp.extracted = torch.zeros_like(sp.fxyz[:, 0]).bool()
    if native:
        sp.frame_offsets, sp.frame_perm = CT.ClusterTracking.frame_offsets_of(sp.frame)
        assert sp.frame_offsets is not None
    return sp


def _track(tracker, sp, frame_id):
    from pcseqlearning_amd.pcdet.utils.common_utils import filter_dict
    mask = (sp.fxyz[:, 0] == frame_id).reshape(-1)
    fp = EasyDict(filter_dict(sp, mask, ignore_keys=("frame_offsets", "frame_perm")))
    fp.component = fp.component - fp.component.min()
    return tracker.track_frame(sp, fp, None)


@pytest.mark.parametrize("native", [False, True])
def test_track_frame_vs_oracle(native):
    """Cluster / track ids bit-exact against the CPU oracle, transforms within 1e-5, on the small
    sequence of the golden file (inputs only are taken from it); for the op-by-op host loop and
    for the native loop (pcs_track_frame)."""
    from oracle import track_oracle as T
    g = np.load(os.path.join(G, "track_small.npz"))
    comp = g["component"]
    C = int(comp.max()) + 1
    o_diam = T.component_diameter(g["fxyz"], comp, C)
    seq = dict(fxyz=g["fxyz"].copy(), frame=g["fxyz"][:, 0].astype(np.int64), component=comp,
               stationary=o_diam[comp] > 12.5, extracted=np.zeros(len(comp), bool))
    oracle_tracker = T.Tracker()
    tracker, sp = _tracker(), _prepare(g, native)
    np.testing.assert_array_equal(sp.stationary.cpu().numpy(), seq["stationary"])
    for frame_id in (0, 8):
        want = oracle_tracker.track_frame(seq, frame_id)
        ex = _track(tracker, sp, frame_id)
        np.testing.assert_array_equal(ex.original_indices.cpu().numpy(), want["original_indices"])
        np.testing.assert_array_equal(ex.frame_indices.cpu().numpy(), want["frame_indices"])
        np.testing.assert_array_equal(ex.component.cpu().numpy(), want["component"])
        np.testing.assert_array_equal(ex.moving.cpu().numpy(), want["moving"])
        np.testing.assert_array_equal(ex.fxyz.cpu().numpy(), want["fxyz"])
        np.testing.assert_allclose(ex.transforms.cpu().numpy(), want["transforms"], rtol=0, atol=1e-5)
    np.testing.assert_array_equal(sp.extracted.cpu().numpy(), seq["extracted"])


@pytest.mark.parametrize("schedule", [(1, True), (1, False)], ids=["resident_tiles", "resident_cache"])
def test_track_frame_vs_oracle_resident_schedules(schedule):
    """The native tracking loop with the ICP iterations on the unit-resident loops (one workgroup per
    registration; LDS-staged searches / hash-grid searches): the same ids as the oracle, bit for bit."""
    ops.icp_set_schedule(*schedule)
    try:
        test_track_frame_vs_oracle(True)
    finally:
        ops.icp_set_schedule(-1, True)


def test_track_frame_vs_oracle_grid_rebuilt_every_iteration():
    """The same with the moving points' grid rebuilt at every ICP iteration (ops.icp_set_grid_mode): the in-place
    form (default, the tests above) and this one both give the oracle's ids."""
    ops.icp_set_grid_mode(True)
    try:
        test_track_frame_vs_oracle(True)
    finally:
        ops.icp_set_grid_mode(False)


@pytest.mark.parametrize("chunk", [3, 8, 16])
def test_track_frame_vs_oracle_chunk_lengths(chunk):
    """pcs_icp_set_chunk_iterations is a scheduling parameter (default 10): the engines read their units' states every
    `chunk` iterations, the stop rule is decided on the device at every iteration -- the oracle's ids and transforms
    for chunks shorter and longer than the default, and for one that does not divide the iteration limit."""
    from pcseqlearning_amd import lib as L
    import ctypes
    L.check(L.load().pcs_icp_set_chunk_iterations(ctypes.c_int(chunk)), "pcs_icp_set_chunk_iterations")
    try:
        test_track_frame_vs_oracle(True)
    finally:
        L.check(L.load().pcs_icp_set_chunk_iterations(ctypes.c_int(10)), "pcs_icp_set_chunk_iterations")


def test_track_frame_vs_reference():
    """Against the reference's own track_frame output: the extraction overlaps > 90 % and the
    first tracked frames agree to fp32 noise (see tests/test_oracle_tracking.py for why the
    multi-frame chain cannot be compared more tightly with the reference's unordered fp32 sums)."""
    g = np.load(os.path.join(G, "track_small.npz"))
    tracker, sp = _tracker(), _prepare(g)
    for frame_id in (0, 8):
        ex = _track(tracker, sp, frame_id)
        want = set(g[f"a{frame_id}_original_indices"].tolist())
        got = set(ex.original_indices.cpu().numpy().tolist())
        assert len(want & got) / len(want | got) > 0.9
        both = sorted(set(g[f"a{frame_id}_component"].tolist()) & set(ex.component.cpu().numpy().tolist()))
        W, M = g[f"a{frame_id}_transforms"][both], ex.transforms.cpu().numpy()[both]
        anchor_col = frame_id - max(0, frame_id - 8)
        near = [c for c in (anchor_col - 1, anchor_col + 1) if 0 <= c < W.shape[1]]
        assert np.median(np.abs(W[:, near] - M[:, near]).max(axis=(1, 2, 3))) < 1e-4


def test_track_frame_native_vs_host():
    """pcs_track_frame (the whole loop in one native call) against the op-by-op host loop on the
    same kernels: identical extraction, masks, frame ranges and debug records; transforms equal to
    f64 round-off."""
    g = np.load(os.path.join(G, "track_small.npz"))
    for frame_id in (0, 8, 16):
        if not (g["fxyz"][:, 0] == frame_id).any():
            continue
        host, sp_h = _tracker(), _prepare(g)
        nat, sp_n = _tracker(), _prepare(g, native=True)
        a = _track(host, sp_h, frame_id)
        b = _track(nat, sp_n, frame_id)
        assert nat.num_registration_calls == host.num_registration_calls > 0
        for key in ("original_indices", "frame_indices", "component", "moving", "valid_comp_mask",
                    "segmentation_label", "fxyz"):
            np.testing.assert_array_equal(a[key].cpu().numpy(), b[key].cpu().numpy(), err_msg=key)
        np.testing.assert_allclose(a.transforms.cpu().numpy(), b.transforms.cpu().numpy(), rtol=0, atol=1e-9)
        np.testing.assert_array_equal(sp_h.extracted.cpu().numpy(), sp_n.extracted.cpu().numpy())
        for key in ("comp_min_frame_id", "comp_max_frame_id"):
            np.testing.assert_array_equal(host.last_track_debug[key].cpu().numpy(),
                                          nat.last_track_debug[key].cpu().numpy(), err_msg=key)
        for key in ("reg_errors", "comp_edge_ratios", "comp_velos"):
            np.testing.assert_allclose(host.last_track_debug[key].cpu().numpy(),
                                       nat.last_track_debug[key].cpu().numpy(), rtol=0, atol=1e-6,
                                       err_msg=key)


def _units_of(tracker, sp, anchors):
    offsets = sp.frame_offsets
    comp, stat = sp.component.long().contiguous(), sp.stationary.to(torch.uint8).contiguous()
    units = []
    for frame_id in anchors:
        lo, hi = offsets[frame_id], offsets[frame_id + 1]
        cmin, cmax = int(comp[lo:hi].min()), int(comp[lo:hi].max())
        w0, w1 = tracker._native_window(offsets, frame_id)
        units.append(dict(component=comp, stationary=stat, anchor_frame=frame_id, min_frame_id=w0,
                          max_frame_id=w1, comp_base=cmin, num_components=cmax - cmin + 1))
    return units


def test_track_batch_vs_track_sequence_vs_track_frame():
    """pcs_track_batch (several sequences handed to the device's lock-step engine by one call),
    pcs_track_sequence (one sequence per call; here also from three host threads at once, so that units
    of different calls share launches and join / leave at different ticks) and pcs_track_frame (one
    unit): every unit's outputs identical, bit for bit."""
    from concurrent.futures import ThreadPoolExecutor
    g = np.load(os.path.join(G, "track_small.npz"))
    tracker, sp = _tracker(), _prepare(g, native=True)
    if sp.frame_perm is not None:  # rows grouped by ascending frame, as the native entries take them
        for k in ("fxyz", "component", "stationary"):
            sp[k] = sp[k][sp.frame_perm].contiguous()
    prm = tracker._native_params()
    anchors = [f for f in (0, 8, 16) if (g["fxyz"][:, 0] == f).any()]
    fxyz = sp.fxyz.contiguous()
    seqs = [(fxyz, sp.frame_offsets, _units_of(tracker, sp, anchors)),
            (fxyz, sp.frame_offsets, _units_of(tracker, sp, anchors[:1])),
            (fxyz, sp.frame_offsets, _units_of(tracker, sp, anchors[::-1]))]
    batch = ops.track_batch(prm, seqs)
    one_by_one = [ops.track_sequence(prm, *q) for q in seqs]

    def worker(q):
        with torch.cuda.stream(torch.cuda.Stream(device=DEV)):
            return ops.track_sequence(prm, *q)

    torch.cuda.synchronize()
    with ThreadPoolExecutor(max_workers=3) as pool:
        threaded = list(pool.map(worker, seqs))
    keys = ("ex_index", "ex_local", "ex_component", "transforms", "reg_errors", "comp_edge_ratios", "comp_velos",
            "comp_min_frame_id", "comp_max_frame_id", "moving", "valid_comp")
    for q, (fx, offs, units) in enumerate(seqs):
        for i, u in enumerate(units):
            single, info = ops.track_frame(prm, fx, u["component"], u["stationary"], offs, u["anchor_frame"],
                                           u["min_frame_id"], u["max_frame_id"], u["comp_base"], u["num_components"])
            for other in (batch, one_by_one, threaded):
                out, info2 = other[q][i]
                assert info2 == info
                for k in keys:
                    assert torch.equal(out[k], single[k]), (q, i, k)


def test_track_frame_with_empty_frames_in_the_window():
    """Frames without any point inside the tracking window (3 and 10 here). The reference cannot
    process them (its per-frame grid sampling takes the min of an empty tensor and raises), and the
    op-by-op host loop refuses the same way; the native tracker treats them as frames in which nothing
    is seen: no harvest from them, every index valid, deterministic."""
    from pcseqlearning_amd.lib import PcsError
    g0 = np.load(os.path.join(G, "track_small.npz"))
    keep = ~np.isin(g0["fxyz"][:, 0].astype(np.int64), [3, 10])
    g = {k: (g0[k][keep] if g0[k].shape[:1] == keep.shape else g0[k]) for k in ("fxyz", "seg_label", "component")}
    for frame_id in (0, 8):
        if not (g["fxyz"][:, 0] == frame_id).any():
            continue
        host, sp_h = _tracker(), _prepare(g)
        with pytest.raises(PcsError):
            _track(host, sp_h, frame_id)
        outs = []
        for _ in range(2):
            nat, sp_n = _tracker(), _prepare(g, native=True)
            outs.append(_track(nat, sp_n, frame_id))
        b = outs[0]
        n = g["fxyz"].shape[0]
        idx = b["original_indices"].cpu().numpy()
        assert idx.size == 0 or (idx.min() >= 0 and idx.max() < n)
        assert not np.isin(b["fxyz"][:, 0].cpu().numpy().astype(np.int64), [3, 10]).any()
        for key in ("original_indices", "frame_indices", "component", "moving"):
            np.testing.assert_array_equal(b[key].cpu().numpy(), outs[1][key].cpu().numpy(), err_msg=key)


def test_track_frame_steps_teacher_forced():
    """Every frame step of the op-by-op loop (the kernels the native tracker runs, bit-identical to it:
    test_track_frame_native_vs_host) from the REFERENCE's own state before the step
    (tests/_track_steps.py, golden track_steps.npz): decisions exact, state to fp32 noise."""
    from pcseqlearning_amd.pcdet.utils.common_utils import filter_dict
    from tests import _track_steps as TS
    g, gs = TS.load()
    tracker, sp = _tracker(), _prepare(gs)
    anchor = int(g["anchor"])
    mask = (sp.fxyz[:, 0] == anchor).reshape(-1)
    fp = EasyDict(filter_dict(sp, mask, ignore_keys=("frame_offsets", "frame_perm")))
    fp.component = fp.component - fp.component.min()
    st = tracker.host_init(sp, fp)
    names = dict(stopped="stopped", moving="moving", transforms="transforms", comp_velos="comp_velos",
                 comp_centers="comp_centers", comp_center_diffs="comp_center_diffs",
                 comp_min_frame_id="comp_min_frame_id", comp_max_frame_id="comp_max_frame_id",
                 last_xyz="last_xyz")
    prev_dir = None
    for s in range(int(g["num_steps"])):
        d, nf = int(g[f"s{s}_track_dir"]), int(g[f"s{s}_next_frame_id"])
        if s > 0:  # the reference's state after its step s-1
            for ref_key, key in names.items():
                st[key] = dev(g[f"s{s - 1}_{ref_key}"].copy())
            st.frame.fxyz = dev(g[f"s{s - 1}_frame_fxyz"].copy())
            last = int(g[f"s{s - 1}_next_frame_id"])
            st.last_velo = st.comp_velos[:, last]
            st.track_dir = int(g[f"s{s - 1}_track_dir"])
            st.next_frame_id = last + st.track_dir
        if d != prev_dir:
            tracker.host_start_dir(st, sp, d)
        prev_dir = d
        assert tracker.host_can_step(st) and st.next_frame_id == nf
        live = (~st.stopped).cpu().numpy()
        st.smooth_iters = int(g[f"s{s}_smooth_iters"])
        f_this, f_next = tracker.host_step(st, sp)
        got = {k: st[k].cpu().numpy() for k in TS.DECISIONS + TS.CONTINUOUS + ("l1_reg_error", "comp_edge_ratio")}
        got["frame_fxyz"] = st.frame.fxyz.cpu().numpy()
        got["component"] = st.frame.component.cpu().numpy()
        TS.check_step(s, g, got, live, (f_this.cpu().numpy(), f_next.cpu().numpy()))
