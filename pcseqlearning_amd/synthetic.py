"""Synthetic Waymo-shaped LiDAR frames and sequences.

A 64-beam spinning LiDAR (inclinations -17.6..+2.4 deg, 2650 azimuth steps = 169 600 rays) is
ray-cast against a gently sloped noisy ground plane, static boxes / walls / poles and moving
vehicle / pedestrian boxes with constant velocity and yaw rate. All frames of a sequence are
expressed in one common frame (the reference transforms every sweep into the last frame's pose,
waymo_dataset.py:575-598), the sensor itself moves with the ego vehicle. Pure numpy; used by the
tests, __graft_entry__.smoke() and bench.py. There is no dataset on the GPU box.
"""
import numpy as np

N_BEAMS = 64
N_AZIMUTH = 2650
SENSOR_HEIGHT = 2.1
MAX_RANGE = 75.0
PC_RANGE = np.array([-75.2, -75.2, -2.0, 75.2, 75.2, 4.0], np.float32)  # all_sequence.yaml:65
VOXEL_SIZE = np.array([0.1, 0.1, 0.15], np.float32)                       # all_sequence.yaml:66
GRID_SIZE = np.array([1504, 1504, 40], np.int64)


def _ray_box(o, d, box):
    """Nearest positive hit distance of rays (o [3], d [R,3]) with a yawed box
    (cx, cy, cz, dx, dy, dz, yaw); inf where missed."""
    c, s = np.cos(box[6]), np.sin(box[6])
    rel = o - box[:3]
    ol = np.array([c * rel[0] + s * rel[1], -s * rel[0] + c * rel[1], rel[2]])
    dl = np.stack([c * d[:, 0] + s * d[:, 1], -s * d[:, 0] + c * d[:, 1], d[:, 2]], axis=1)
    half = box[3:6] / 2
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / dl
        t1 = (-half - ol) * inv
        t2 = (half - ol) * inv
    tmin = np.nanmax(np.minimum(t1, t2), axis=1)
    tmax = np.nanmin(np.maximum(t1, t2), axis=1)
    hit = (tmax >= np.maximum(tmin, 0.0)) & (tmin > 0.0)
    return np.where(hit, tmin, np.inf)


class Scene:
    """Static world + dynamic objects of one sequence (seed -> numpy.random.default_rng(20221+s))."""

    def __init__(self, seq_seed=0, n_static=60, n_vehicle=25, n_ped=15, ego_speed=1.0):
        rng = np.random.default_rng(20221 + seq_seed)
        self.seq_seed = seq_seed
        self.ego_speed = ego_speed  # m per frame along +x
        self.slope = 0.01
        st = []
        for _ in range(n_static):
            kind = rng.integers(0, 4)
            a = rng.uniform(0, 2 * np.pi)
            if kind == 0:    # building-like box, far from the road
                r = rng.uniform(30, 70)
                dx, dy, dz = rng.uniform(4, 12), rng.uniform(4, 12), rng.uniform(3, 8)
            elif kind == 1:  # wall / fence
                r = rng.uniform(18, 70)
                dx, dy, dz = rng.uniform(6, 16), rng.uniform(0.3, 0.6), rng.uniform(1.2, 2.5)
            else:            # pole / trunk
                r = rng.uniform(6, 60)
                dx, dy, dz = rng.uniform(0.25, 0.5), rng.uniform(0.25, 0.5), rng.uniform(3, 6)
            x, y = r * np.cos(a), r * np.sin(a)
            st.append([x, y, self.slope * x + dz / 2, dx, dy, dz, rng.uniform(0, np.pi)])
        self.static = np.array(st, np.float64)
        dyn, vel, yawrate, cls = [], [], [], []
        for i in range(n_vehicle + n_ped):
            veh = i < n_vehicle
            r, a = rng.uniform(6, 55), rng.uniform(0, 2 * np.pi)
            x, y = r * np.cos(a), r * np.sin(a)
            yaw = rng.uniform(0, 2 * np.pi)
            size = (4.6, 1.9, 1.6) if veh else (0.8, 0.8, 1.7)
            parked = rng.uniform() < 0.3
            speed = 0.0 if parked else (rng.uniform(0.2, 1.5) if veh else rng.uniform(0.05, 0.2))
            dyn.append([x, y, self.slope * x + size[2] / 2 + 0.02, *size, yaw])
            vel.append(speed)
            yawrate.append(0.0 if parked else rng.uniform(-0.02, 0.02))
            cls.append(1 if veh else 2)
        self.dynamic0 = np.array(dyn, np.float64)
        self.speed = np.array(vel)
        self.yawrate = np.array(yawrate)
        self.cls = np.array(cls, np.int64)

    def dynamic_boxes(self, f):
        """7-DoF boxes of the dynamic objects at frame f (constant speed along heading)."""
        b = self.dynamic0.copy()
        yaw = b[:, 6] + self.yawrate * f
        # integrate the heading change in closed form (small yaw rates): straight-line approximation
        b[:, 0] += self.speed * f * np.cos(b[:, 6] + self.yawrate * f / 2)
        b[:, 1] += self.speed * f * np.sin(b[:, 6] + self.yawrate * f / 2)
        b[:, 2] = self.slope * b[:, 0] + b[:, 5] / 2 + 0.02
        b[:, 6] = yaw
        return b

    def sensor_origin(self, f):
        x = self.ego_speed * f
        return np.array([x, 0.0, SENSOR_HEIGHT + self.slope * x])


def gen_frame(scene, f, n_target=160_000, seed=None):
    """One frame: dict(points [n,3] f32, feat [n,3] f32 (intensity, elongation, range/75),
    seg [n] i64 (17 ground, 1 vehicle, 2 pedestrian, 10 static), inst [n] i64 (-1 or object id),
    boxes [B,7] f32, box_cls [B], obj_ids [B])."""
    rng = np.random.default_rng(1000 + f + 7919 * scene.seq_seed if seed is None else seed)
    o = scene.sensor_origin(f)
    dyn = scene.dynamic_boxes(f)
    out_xyz, out_seg, out_inst, out_rng = [], [], [], []
    have = 0
    for attempt in range(4):
        incl = np.deg2rad(np.linspace(-17.6, 2.4, N_BEAMS)) + rng.normal(0, 2e-4, N_BEAMS)
        az = (np.arange(N_AZIMUTH) + rng.uniform(0, 1)) * (2 * np.pi / N_AZIMUTH)
        az = az[None, :] + rng.normal(0, 1e-4, (N_BEAMS, N_AZIMUTH))
        ci = np.cos(incl)[:, None]
        d = np.stack([ci * np.cos(az), ci * np.sin(az), np.broadcast_to(np.sin(incl)[:, None], az.shape)],
                     axis=-1).reshape(-1, 3)
        # ground: o_z + t d_z = slope (o_x + t d_x)
        den = scene.slope * d[:, 0] - d[:, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            tg = (o[2] - scene.slope * o[0]) / den
        t = np.where((den > 1e-9) & (tg > 0), tg, np.inf)
        seg = np.full(d.shape[0], 17, np.int64)
        inst = np.full(d.shape[0], -1, np.int64)
        for b in scene.static:
            tb = _ray_box(o, d, b)
            m = tb < t
            t = np.where(m, tb, t)
            seg[m] = 10
        for j, b in enumerate(dyn):
            tb = _ray_box(o, d, b)
            m = tb < t
            t = np.where(m, tb, t)
            seg[m] = scene.cls[j]
            inst[m] = j
        t = t + rng.normal(0, 0.01, t.shape)
        ok = np.isfinite(t) & (t < MAX_RANGE) & (t > 0.5)
        xyz = o[None, :] + t[:, None] * d
        xyz[:, 2] += np.where(seg == 17, rng.uniform(-0.03, 0.03, t.shape), 0.0)
        ok &= (xyz[:, 0] > PC_RANGE[0]) & (xyz[:, 0] < PC_RANGE[3]) & (xyz[:, 1] > PC_RANGE[1]) & \
              (xyz[:, 1] < PC_RANGE[4]) & (xyz[:, 2] > PC_RANGE[2]) & (xyz[:, 2] < PC_RANGE[5])
        out_xyz.append(xyz[ok]); out_seg.append(seg[ok]); out_inst.append(inst[ok]); out_rng.append(t[ok])
        have += int(ok.sum())
        if have >= n_target:
            break
    xyz = np.concatenate(out_xyz); seg = np.concatenate(out_seg)
    inst = np.concatenate(out_inst); rr = np.concatenate(out_rng)
    if xyz.shape[0] > n_target:
        keep = np.sort(rng.choice(xyz.shape[0], n_target, replace=False))
        xyz, seg, inst, rr = xyz[keep], seg[keep], inst[keep], rr[keep]
    n = xyz.shape[0]
    feat = np.stack([np.tanh(rng.uniform(0, 2, n)), rng.uniform(0, 1, n), rr / MAX_RANGE], axis=1)
    return dict(points=xyz.astype(np.float32), feat=feat.astype(np.float32), seg=seg, inst=inst,
                boxes=dyn.astype(np.float32), box_cls=scene.cls.copy(),
                obj_ids=np.arange(dyn.shape[0]))


def gen_sequence(seq_seed=0, num_frames=20, n_target=160_000, **scene_kw):
    """A whole sequence in the reference's sample contract after collate:
    point_bxyz [N,4] (batch 0), point_sweep [N,1] int64 frame id, point_feat, segmentation_label,
    instance_label, per-frame GT boxes flattened frame-major, obj_ids, frame_id strings."""
    scene = Scene(seq_seed, **scene_kw)
    P, S, F, SEG, INST, B, BC, OID = [], [], [], [], [], [], [], []
    for f in range(num_frames):
        fr = gen_frame(scene, f, n_target)
        n = fr["points"].shape[0]
        P.append(fr["points"]); S.append(np.full((n, 1), f, np.int64)); F.append(fr["feat"])
        SEG.append(fr["seg"]); INST.append(fr["inst"])
        B.append(fr["boxes"]); BC.append(fr["box_cls"]); OID.append(fr["obj_ids"])
    pts = np.concatenate(P)
    batch = dict(
        batch_size=1,
        point_bxyz=np.concatenate([np.zeros((pts.shape[0], 1), np.float32), pts], axis=1),
        point_sweep=np.concatenate(S),
        point_feat=np.concatenate(F),
        segmentation_label=np.concatenate(SEG),
        instance_label=np.concatenate(INST),
        gt_box_attr=[np.concatenate(B)],            # [F*B, 7], frame-major
        gt_box_cls_label=[np.concatenate(BC)],
        obj_ids=[np.array([f"obj{seq_seed}_{i}" for i in np.concatenate(OID)])],
        frame_id=[[f"synthetic_seq{seq_seed:03d}_{f:03d}" for f in range(num_frames)]],
        num_frames=num_frames,
    )
    return batch


def collate_sequences(batches):
    """Several gen_sequence() results as ONE batch (collate_batch of the reference's dataset,
    dataset.py:252-298, for this sample contract): point arrays concatenated with the batch index in
    column 0 of point_bxyz, per-sequence entries as lists."""
    out = dict(batch_size=len(batches), num_frames=batches[0]["num_frames"])
    pts = []
    for b, s in enumerate(batches):
        p = s["point_bxyz"].copy()
        p[:, 0] = b
        pts.append(p)
    out["point_bxyz"] = np.concatenate(pts)
    for key in ("point_sweep", "point_feat", "segmentation_label", "instance_label"):
        out[key] = np.concatenate([s[key] for s in batches])
    for key in ("gt_box_attr", "gt_box_cls_label", "obj_ids", "frame_id"):
        out[key] = [s[key][0] for s in batches]
    return out


def gen_voxel_frame(seed=0, n_target=160_000):
    """Config 2 input: one 160k-pt frame as `points` [n, 1+3+2] (batch, x, y, z, intensity,
    elongation) -- 5 features after the mean VFE, as with Waymo in OpenPCDet."""
    scene = Scene(seed)
    fr = gen_frame(scene, 0, n_target)
    n = fr["points"].shape[0]
    pts = np.concatenate([np.zeros((n, 1), np.float32), fr["points"], fr["feat"][:, :2]], axis=1)
    return pts.astype(np.float32)
