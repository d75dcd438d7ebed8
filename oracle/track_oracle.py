"""numpy restatement of the tracking stage of the reference
(pcdet/models/registration/preprocessors/cluster_tracking.py): sample_frame (:39-51),
component_diameter / component_center / filter_components (:89-148), dist_compensate (:80-87),
smooth_velo (:162-199, torch.optim.AdamW + MultiStepLR restated), track_frame (:430-787) and the
cluster proposal loop of cluster_proposal.py:57-83.

TEST INFRASTRUCTURE ONLY (see oracle/oracle.py). Leaf ops come from the C++ oracle (radius graph,
grid sampling, connected components) and reg_oracle (register_to_next_frame). Pinned by
tests/golden/track_small.npz and smooth_velo.npz, produced by the reference's own code.
Serial numpy on one core: this is also what bench.py times as the CPU baseline.
"""
import numpy as np

from . import oracle as O
from . import reg_oracle as R

F32 = np.float32


# ---------------------------------------------------------------------------------------------
# per-component reductions. fp32 sums follow a fixed order: elements of a segment in ascending
# point index are dealt round-robin to 64 accumulators, each summed serially, then combined by a
# xor butterfly (32,16,...,1). That is the order of the device's wave-per-segment reduction; any
# fixed order is as valid a restatement of torch_scatter's unordered atomics.
def _wave_sum(vals):
    """vals [m, d] f32 -> [d] f32."""
    m, d = vals.shape
    lanes = np.zeros((64, d), F32)
    for start in range(0, m, 64):
        chunk = vals[start:start + 64]
        lanes[:chunk.shape[0]] = lanes[:chunk.shape[0]] + chunk
    idx = np.arange(64)
    for o in (32, 16, 8, 4, 2, 1):
        lanes = lanes + lanes[idx ^ o]
    return lanes[0]


class Segs:
    def __init__(self, index, C):
        index = np.asarray(index, np.int64)
        self.index = index
        self.C = C
        self.order = np.argsort(index, kind="stable")
        sidx = index[self.order]
        self.ids, self.start = np.unique(sidx, return_index=True)
        self.end = np.append(self.start[1:], len(sidx))


def seg_sum(data, sg, mean=False):
    data = np.asarray(data)
    flat = data.reshape(data.shape[0], -1).astype(F32)
    out = np.zeros((sg.C, flat.shape[1]), F32)
    for i, b, e in zip(sg.ids, sg.start, sg.end):
        s = _wave_sum(flat[sg.order[b:e]])
        out[i] = s / F32(e - b) if mean else s
    return out.reshape((sg.C,) + data.shape[1:]).astype(data.dtype if data.dtype.kind == "f" else F32)


def seg_max(data, sg):
    out = np.zeros(sg.C, F32)
    for i, b, e in zip(sg.ids, sg.start, sg.end):
        out[i] = data[sg.order[b:e]].max()
    return out


def norm3(x):
    x = x.astype(F32)
    return np.sqrt((x * x).sum(-1, dtype=F32)).astype(F32)


# ---------------------------------------------------------------------------------------------
def sample_frame(grid_size, frame):
    means, inv, order, seg = O.grid_sample(frame["fxyz"], grid_size)
    v = means.shape[0]
    stat = O.segment_reduce(frame["stationary"].astype(F32), order, seg, mean=True) > 0.5
    comp = O.robust_median(frame["component"], inv, v)
    frm = O.robust_median(frame["frame"].reshape(-1), inv, v)
    return dict(fxyz=means, stationary=stat, component=comp, frame=frm)


def component_stats(fxyz, comp, C, sg=None):
    sg = sg or Segs(comp, C)
    xyz = fxyz[:, 1:]
    deg = seg_sum(np.ones(len(comp), F32), sg)
    valid = deg > 0.5
    center = seg_sum(xyz, sg)
    center[valid] = center[valid] / deg[valid, None]
    return sg, deg, valid, center


def component_diameter(fxyz, comp, C, sg=None):
    sg, _, _, center = component_stats(fxyz, comp, C, sg)
    dist = norm3(fxyz[:, 1:] - center[comp])
    return seg_max(dist, sg) * F32(2)


def dist_compensate(deg):
    th = [0, 10, 40, 100, 200, 400, 10000000]
    val = [1, 0.5, 0.3, 0.2, 0.1, 0.0]
    out = np.zeros(len(deg), F32)
    for lo, hi, v in zip(th[:-1], th[1:], val):
        out[(deg >= lo) & (deg < hi)] = v
    return out


def smooth_velo(v, d, frame_id, next_frame_id, weight0=1.0, weight=10.0, num_itr=300, stopping=1e-3,
                force_iters=None, iters_out=None):
    """AdamW(lr 1e-2, betas (.9,.999), eps 1e-8, wd 1e-2) + MultiStepLR([100,200,300]) on the whole
    [C,F,3] tensor, in place; loss and gradients in closed form.
    force_iters (tests only): run exactly that many AdamW steps instead of the reference's stop rule
    (`last_loss - loss < 1e-3` on an fp32 mean: marginal by one step run to run); iters_out: 1-element
    list receiving the number of steps taken."""
    if frame_id == next_frame_id:
        return v
    a, b = (frame_id, next_frame_id) if frame_id < next_frame_id else (next_frame_id, frame_id)
    C = v.shape[0]
    nf = b - a + 1
    inv_fit = F32(1.0) / F32(C * nf * 2)
    inv_sm = F32(1.0) / F32(C * (nf - 1) * 2)
    w0, w1 = F32(weight0), F32(weight)
    m = np.zeros((C, nf, 2), F32)
    s = np.zeros((C, nf, 2), F32)
    last_loss, countdown = F32(1e10), 3
    if iters_out is not None:
        iters_out[0] = 0
    for it in range(num_itr if force_iters is None else int(force_iters)):
        lr = 1e-2 * (0.1 ** sum(it >= ms for ms in (100, 200, 300)))
        x = v[:, a:b + 1, :2]
        dd = d[:, a:b + 1, :2]
        fit = x - dd
        sm = x[:, :-1] - x[:, 1:]
        loss = F32((fit * fit).astype(np.float64).sum() * inv_fit * w0 +
                   np.abs(sm).astype(np.float64).sum() * inv_sm * w1)
        g = fit * F32(2) * (inv_fit * w0)
        sg = np.sign(sm).astype(F32) * (inv_sm * w1)
        g[:, :-1] += sg
        g[:, 1:] -= sg
        m = (m + F32(0.1) * (g - m)).astype(F32)
        s = (s * F32(0.999) + F32(0.001) * g * g).astype(F32)
        bc1 = 1.0 - 0.9 ** (it + 1)
        bc2 = 1.0 - 0.999 ** (it + 1)
        step, bc2s, decay = F32(lr / bc1), F32(np.sqrt(bc2)), F32(1.0 - lr * 1e-2)
        vin = (x * decay).astype(F32)
        denom = (np.sqrt(s) / bc2s + F32(1e-8)).astype(F32)
        new = (vin - step * (m / denom)).astype(F32)
        v *= decay                      # decoupled weight decay reaches every element
        v[:, a:b + 1, :2] = new
        if iters_out is not None:
            iters_out[0] = it + 1
        if force_iters is not None:
            continue
        if last_loss - loss < stopping:
            countdown -= 1
        else:
            countdown = 3
        if countdown <= 0:
            break
        last_loss = loss
    return v


# ---------------------------------------------------------------------------------------------
class Tracker:
    def __init__(self, radius_list=(2.5, 1.25, 1.0), voxel_sizes=((0.4, 0.4, 0.6), (0.2, 0.2, 0.3), (0.1, 0.1, 0.15)),
                 stopping_delta=(0.05, 0.05, 0.05), nn_radius=0.5, angle_regularizer=10.0,
                 reg_error_coeff=0.13, track_interval=8, angle_threshold=45, min_move_frame=6, icp_threads=None):
        # icp_threads: None = the numpy ICP (reg_oracle.register_to_next_frame); an int = the C++ form of
        # the same function (pcs_oracle_icp.cpp, bit-compatible; 0 = all cores) for full-size inputs
        self.icp_threads = icp_threads
        self.radius_list, self.voxel_sizes, self.stopping_delta = radius_list, voxel_sizes, stopping_delta
        self.nn_radius, self.angle_regularizer = nn_radius, angle_regularizer
        self.reg_error_coeff, self.track_interval = reg_error_coeff, track_interval
        self.angle_threshold, self.min_move_frame = angle_threshold, min_move_frame

    # The loop of cluster_tracking.py:430-787 in three pieces, so that tests can run it one frame step
    # at a time from a given state (teacher forcing with the reference's own states,
    # tests/golden/track_steps.npz): init_state -> for each direction: start_dir -> step ... -> finish.
    def init_state(self, seq, frame_id):
        """seq: dict(fxyz [N,4] f32, frame [N] i64, component [N] i64, stationary [N] bool,
        extracted [N] bool)."""
        S = {}
        fmask = seq["frame"] == frame_id
        S["fmask"], S["orig_idx"], S["frame_id"] = fmask, np.nonzero(fmask)[0], frame_id
        fr = dict(fxyz=seq["fxyz"][fmask].copy(), component=seq["component"][fmask].copy(),
                  stationary=seq["stationary"][fmask].copy(), frame=seq["frame"][fmask].copy())
        fr["component"] = fr["component"] - fr["component"].min()
        comp = fr["component"]
        C = int(comp.max()) + 1
        S["fr"], S["comp"], S["C"] = fr, comp, C
        S["min_f"] = max(int(seq["frame"].min()), frame_id - self.track_interval)
        S["max_f"] = min(int(seq["frame"].max()), frame_id + self.track_interval)
        sg = Segs(comp, C)
        S["sg"] = sg
        S["comp_deg"] = np.bincount(comp, minlength=C)
        S["comp_diameter"] = component_diameter(fr["fxyz"], comp, C, sg)
        Fn = S["max_f"] + 1
        S["transforms"] = np.tile(np.eye(4), (C, S["max_f"] - S["min_f"] + 1, 1, 1))
        S["comp_min"] = np.full(C, frame_id, np.int64)
        S["comp_max"] = np.full(C, frame_id, np.int64)
        S["comp_velos"] = np.zeros((C, Fn, 3), F32)
        S["comp_centers"] = np.zeros((C, Fn, 3), F32)
        S["comp_centers"][:, frame_id] = component_stats(fr["fxyz"], comp, C, sg)[3]
        S["comp_center_diffs"] = np.zeros((C, Fn, 3), F32)
        _, _, nonempty, _ = component_stats(fr["fxyz"], comp, C, sg)
        S["valid_comp"] = nonempty & (S["comp_diameter"] < 12.5)
        vpm = S["valid_comp"][comp]
        S["ex"] = dict(fxyz=[fr["fxyz"][vpm].copy()], component=[comp[vpm]], frame_indices=[np.nonzero(vpm)[0]],
                       original_indices=[S["orig_idx"][vpm]])
        S["last_velo"] = None  # (column index) -> live view semantics of the reference
        S["moving"] = None
        return S

    def start_dir(self, S, seq, track_dir):
        S["track_dir"] = track_dir
        S["nf"] = S["frame_id"] + track_dir
        S["fr"]["fxyz"] = seq["fxyz"][S["fmask"]].copy()
        S["stopped"] = ~S["valid_comp"]
        S["moving"] = S["valid_comp"].copy()
        S["last_xyz"] = S["fr"]["fxyz"][:, 1:].copy()
        if track_dir == 1 and S["frame_id"] > 0:
            S["last_velo"] = S["frame_id"]

    def can_step(self, S):
        return S["min_f"] <= S["nf"] <= S["max_f"] and bool((~S["stopped"]).any())

    def step(self, S, seq):
        """One frame step (the body of the reference's while loop). Returns the harvest's edges
        (f_this, f_next) before the stop filter."""
        fr, comp, C, sg = S["fr"], S["comp"], S["C"], S["sg"]
        nf, track_dir, frame_id, min_f = S["nf"], S["track_dir"], S["frame_id"], S["min_f"]
        transforms, comp_velos, comp_centers = S["transforms"], S["comp_velos"], S["comp_centers"]
        comp_center_diffs, comp_diameter, comp_deg = S["comp_center_diffs"], S["comp_diameter"], S["comp_deg"]
        stopped, moving, last_velo = S["stopped"], S["moving"], S["last_velo"]
        nmask = seq["frame"] == nf
        n_orig = np.nonzero(nmask)[0]
        nxt = dict(fxyz=seq["fxyz"][nmask], component=seq["component"][nmask],
                   stationary=seq["stationary"][nmask], frame=seq["frame"][nmask])
        col = nf - min_f
        transforms[:, col] = transforms[:, col - track_dir]
        if last_velo is not None:
            trans = comp_velos[:, last_velo].copy()
            trans[stopped] = 0
            fr["fxyz"][:, 1:] += trans[comp] * F32(track_dir)
            transforms[:, col, :3, 3] += trans.astype(np.float64) * track_dir
        for i, radius in enumerate(self.radius_list):
            sub = sample_frame(self.voxel_sizes[i], fr)
            sub_next = sample_frame(self.voxel_sizes[i], nxt)
            if self.icp_threads is None:
                graph = R.Graph(radius)
                _, T, l1, ratio, _ = R.register_to_next_frame(graph, sub, sub_next, C, self.angle_regularizer,
                                                              80, self.stopping_delta[i])
            else:
                _, T, l1, ratio, _ = R.register_to_next_frame_cpp(radius, sub, sub_next, C, self.angle_regularizer,
                                                                  80, self.stopping_delta[i], threads=self.icp_threads)
            if i == 0:
                comp_edge_ratio = ratio
            if i == len(self.radius_list) - 1:
                l1_reg_error = l1
            Tc = T[comp]
            rot = np.einsum("nab,nb->na", Tc[:, :3, :3], fr["fxyz"][:, 1:].astype(np.float64))
            fr["fxyz"][:, 1:] = rot.astype(F32) + Tc[:, :3, 3].astype(F32)
            transforms[:, col] = T @ transforms[:, col]
        comp_centers[:, nf] = component_stats(fr["fxyz"], comp, C, sg)[3]
        point_velo = (fr["fxyz"][:, 1:] - S["last_xyz"]) * F32(track_dir)
        comp_velo = seg_sum(point_velo, sg, mean=True)
        comp_velo[:, 2] = 0
        comp_velos[:, nf] = comp_velo
        comp_center_diffs[:, nf] = (comp_centers[:, nf] - comp_centers[:, nf - track_dir]) * F32(track_dir)
        iters = [0]
        smooth_velo(comp_velos, comp_center_diffs, frame_id + track_dir, nf,
                    force_iters=S.get("force_smooth_iters"), iters_out=iters)
        S["smooth_iters"] = iters[0]
        delta = comp_velos[:, nf] - comp_velo
        fr["fxyz"][:, 1:] += delta[comp] * F32(track_dir)
        transforms[:, col, :3, 3] += (delta * F32(track_dir)).astype(np.float64)
        S["last_xyz"] = fr["fxyz"][:, 1:].copy()
        cv = comp_velos[:, nf]
        stopped = stopped | (l1_reg_error > (F32(self.reg_error_coeff) * comp_diameter *
                                             (F32(1) + dist_compensate(comp_deg))).astype(F32))
        stopped = stopped | (comp_edge_ratio < 0.5)
        if (nf - frame_id) * track_dir == self.min_move_frame:
            moved = norm3(comp_centers[:, nf] - comp_centers[:, frame_id])
            moving = moving & (moved > F32(0.08) * comp_diameter)
        if last_velo is not None:
            lv = comp_velos[:, last_velo]
            dev = norm3(cv - lv)
            stopped = stopped | (dev > F32(0.24) * comp_diameter)
            pv = comp_velos[:, nf - track_dir]
            nrm = np.maximum(norm3(cv) * norm3(pv), F32(1e-6))
            cosang = np.clip((cv * pv).sum(-1, dtype=F32) / nrm, -1, 1)
            angle = np.arccos(cosang).astype(F32) / F32(np.pi) * F32(180.0)
            stopped = stopped | ((angle > self.angle_threshold) & (norm3(comp_velos[:, nf, :2]) > F32(0.01)))
        S["last_velo"] = nf
        if nf == frame_id - 1:
            comp_velos[:, frame_id] = cv
        if track_dir == -1:
            S["comp_min"][~stopped] = nf
        else:
            S["comp_max"][~stopped] = nf
        q = fr["fxyz"].copy()
        q[:, 0] = nf
        f_this, f_next = O.radius_graph(q, nxt["fxyz"], self.nn_radius, 1, True)
        ok = (~stopped)[comp[f_this]]
        ex = S["ex"]
        ex["fxyz"].append(nxt["fxyz"][f_next[ok]])
        ex["component"].append(comp[f_this[ok]])
        ex["frame_indices"].append(f_next[ok])
        ex["original_indices"].append(n_orig[f_next[ok]])
        S["stopped"], S["moving"] = stopped, moving
        S["l1_reg_error"], S["comp_edge_ratio"] = l1_reg_error, comp_edge_ratio
        S["nf"] = nf + track_dir
        return f_this, f_next

    def finish(self, S, seq):
        frame_id = S["frame_id"]
        out = {k: np.concatenate(v) for k, v in S["ex"].items()}
        out["moving"] = S["moving"][out["component"]]
        valid_comp = S["valid_comp"] & ((S["comp_max"] >= frame_id + self.min_move_frame) |
                                        (S["comp_min"] <= frame_id - self.min_move_frame))
        keep = valid_comp[out["component"]]
        out = {k: v[keep] for k, v in out.items()}
        out["transforms"] = S["transforms"]
        seq["extracted"][out["original_indices"]] = True
        return out

    def track_frame(self, seq, frame_id):
        """Returns the `extracted` dict of the reference."""
        S = self.init_state(seq, frame_id)
        for track_dir in (-1, 1):
            self.start_dir(S, seq, track_dir)
            while self.can_step(S):
                self.step(S, seq)
        S["fr"]["fxyz"] = seq["fxyz"][S["fmask"]].copy()
        return self.finish(S, seq)

    # ---- teacher forcing: overwrite the loop state with the reference's state after its step `s`
    STATE_MAP = dict(stopped="stopped", moving="moving", transforms="transforms", comp_velos="comp_velos",
                     comp_centers="comp_centers", comp_center_diffs="comp_center_diffs",
                     comp_min_frame_id="comp_min", comp_max_frame_id="comp_max", last_xyz="last_xyz")

    def load_reference_state(self, S, g, s):
        for ref_key, key in self.STATE_MAP.items():
            S[key] = g[f"s{s}_{ref_key}"].copy()
        S["fr"]["fxyz"] = g[f"s{s}_frame_fxyz"].copy()
        S["track_dir"] = int(g[f"s{s}_track_dir"])
        S["last_velo"] = int(g[f"s{s}_next_frame_id"])
        S["nf"] = S["last_velo"] + S["track_dir"]


def propose_clusters(fxyz, frame, radius, max_nbr=32, chunk=10):
    """cluster_proposal.py:57-83 for one radius key."""
    comp = np.zeros(len(fxyz), np.int64)
    total = 0
    for f0 in range(0, int(frame.max()) + 1, chunk):
        mask = (frame >= f0) & (frame < f0 + chunk)
        if not mask.any():
            continue
        pts = fxyz[mask]
        e0, e1 = O.radius_graph(pts, pts, radius, max_nbr, True)
        num, c = O.connected_components(e0, e1, len(pts))
        comp[mask] = c + total
        total += num
    return comp
