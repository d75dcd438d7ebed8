"""Ground-plane estimation of pcdet/models/registration/preprocessors/preprocessor_utils.py.

Stages (same order and formulas as the reference, :21-30, :274-311, :84-272, :313-350, :352-419):
grid subsample -> 2 m pillars -> plane hypotheses per 8 m super-pillar by iteratively reweighted
least squares over 30 height ratios -> "truncated least squares" pruning of hypotheses by k-NN
curvature -> nearest-confidence propagation to every pillar -> L1 + second-difference smoothing of
the pillar heights. Grid sampling, the per-pillar reductions and the smoothing optimiser run in
the HIP kernels; the small dense algebra on <= a few hundred super-pillars (3x3 eigh, 8-NN among
hypotheses) is torch device code. Not yet fused: see DESIGN.md "what comes next".
"""
import numpy as np
import torch

from ..... import ops
from .....lib import PcsError
from ....utils.common_utils import EasyDict, filter_dict
from ....utils.timer import Timer
from ...model_utils.grid_sampling import GridSampling3D
from .registration_utils import IndexSegments, efficient_robust_sum, robust_mean, segment_max, segment_min


def grid_sample(point_bxyz, grid_size):
    """Voxel means with the frame column zeroed (all frames merged), and the point->voxel map."""
    sampler = GridSampling3D(grid_size)
    merged = point_bxyz.clone()
    merged[:, 0] = 0
    means, inv = sampler(merged, return_inverse=True)
    return EasyDict(bxyz=means), inv


def _seg_sum(data, sg, n):
    return efficient_robust_sum(data, sg, n, block=True)


def iterative_reweighted_ransac(points, pillars, w0, num_pillars, sigma2, stopping_delta=1e-2, sg=None,
                                origin=None, seg_start=None):
    """<= 50 IRLS plane fits per pillar: weighted centre, weighted covariance, smallest
    eigenvector as the normal, Cauchy weights on the point-plane error times a 0.5 m distance
    falloff; stops when no weight moves by more than stopping_delta (preprocessor_utils.py:32-80).
    Runs as ops.irls_plane_fit: two HIP kernels per round, convergence tested on the device."""
    xyz = points.bxyz[:, 1:4]
    if origin is None:
        if sg is None:
            sg = IndexSegments(points.pillar_idx, num_pillars)
        origin = robust_mean(xyz, sg, num_pillars)
    pillars.center, pillars.normal, points.plane_fitting_error = ops.irls_plane_fit(
        xyz, points.pillar_idx, w0, origin, num_pillars, sigma2, stopping_delta, 50, seg_start=seg_start)
    return points, pillars


def _knn_self(xyz, k):
    """torch_cluster.knn(x, x, k): for every point its k nearest (itself included), row-major
    (query, neighbour) pairs; brute force, ties by index (upstream order is unspecified)."""
    n = xyz.shape[0]
    k = min(k, n)
    d = torch.cdist(xyz.double(), xyz.double())
    nbr = d.topk(k, dim=1, largest=False).indices
    e0 = torch.arange(n, device=xyz.device)[:, None].expand(-1, k).reshape(-1)
    return e0, nbr.reshape(-1), k


@torch.no_grad()
def compute_min_height_from_ransac(pillar_dims, num_pillars, voxels, pillars, cfg, window_size=4):
    dev = voxels.bxyz.device
    ar = torch.arange(num_pillars, device=dev)
    # pillar_dims: host integers (format_pillars read them once). The pillars enumerate the whole X x Y grid, so the
    # super-pillar grid's size follows on the host (the reference takes coords.max() + 1: the same numbers)
    X, Y = int(pillar_dims[0]), int(pillar_dims[1])
    coords = torch.stack([ar // Y, ar % Y], dim=-1) // window_size
    new_dims = ((X - 1) // window_size + 1, (Y - 1) // window_size + 1)
    num_new = new_dims[0] * new_dims[1]
    pillars.pillar_coords = coords
    pillars.pillar_idx = coords[:, 0] * new_dims[1] + coords[:, 1]
    pillars.best_confidence = torch.zeros_like(pillars.density).reshape(-1)
    pillars.best_normal = pillars.min_z.new_zeros(num_pillars, 3)
    pillars.best_center = pillars.min_z.new_zeros(num_pillars, 3)

    # voxels regrouped into super-pillars, sorted by super-pillar like the reference
    nv = EasyDict(bxyz=voxels.bxyz)
    nv.pillar_coords = voxels.pillar_coords // window_size
    nv.pillar_idx = nv.pillar_coords[:, 0] * new_dims[1] + nv.pillar_coords[:, 1]
    order = torch.sort(nv.pillar_idx, stable=True).indices
    nv = EasyDict(filter_dict(nv, order))
    sg = IndexSegments(nv.pillar_idx, num_new)
    z = nv.bxyz[:, -1]
    npil = EasyDict()
    npil.min_z = segment_min(z, sg, num_new)
    npil.max_z = segment_max(z, sg, num_new)
    npil.best_confidence = torch.zeros_like(npil.min_z).reshape(-1)
    npil.best_normal = npil.min_z.new_zeros(num_new, 3)
    npil.best_normal[:, -1] = 1.0
    npil.best_center = npil.min_z.new_zeros(num_new, 3)

    t_fit = Timer("ground: plane hypotheses (IRLS x 30 ratios)")
    t_fit.__enter__()
    sigma = cfg["SIGMA2"] ** 0.5
    origin = robust_mean(nv.bxyz[:, 1:4], sg, num_new)
    # rows are sorted by super-pillar: dense row ranges for the one-launch-per-round plane fit
    seg_start = torch.searchsorted(nv.pillar_idx.contiguous(),
                                   torch.arange(num_new + 1, device=dev)).int()
    ratios = torch.linspace(0.3, 1, 30).tolist()
    if cfg.get("BATCHED_HYPOTHESES", True):
        # the 30 fits are independent: side by side in one call (pcs_ransac_plane_hypotheses)
        npil.best_center, npil.best_normal, npil.best_confidence = ops.ransac_plane_hypotheses(
            nv.bxyz[:, 1:4], seg_start, npil.min_z, npil.max_z, ratios, origin, cfg["SIGMA2"],
            hit_below=sigma)
    else:
        for ratio in ratios:
            cur_z = npil.min_z * ratio + npil.max_z * (1 - ratio)
            z_diff = cur_z[nv.pillar_idx] - z
            w0 = (cfg["SIGMA2"] / (z_diff.square() + cfg["SIGMA2"])).reshape(-1, 1)
            nv, npil = iterative_reweighted_ransac(nv, npil, w0, num_new, sigma2=cfg["SIGMA2"], sg=sg,
                                                   origin=origin, seg_start=seg_start)
            hit = (nv.plane_fitting_error < sigma).float()
            num_hit = _seg_sum(hit, sg, num_new)
            better = npil.best_confidence < num_hit
            if bool(better.any()):
                npil.best_normal[better] = npil.normal[better]
                npil.best_center[better] = npil.center[better]
                npil.best_confidence[better] = num_hit[better]
    t_fit.__exit__(None, None, None)
    t_tls = Timer("ground: hypothesis pruning (100 thresholds)")
    t_tls.__enter__()
    # "truncated least squares": drop hypotheses whose 8-NN mean curvature exceeds a shrinking bound
    xyz, normal, conf = npil.best_center, npil.best_normal, npil.best_confidence

    def mean_curvature(xyz, normal):
        if xyz.shape[0] == 0:
            return xyz.new_zeros(0), float("-inf")
        e0, e1, k = _knn_self(xyz, cfg["K"])
        diff = xyz[e1] - xyz[e0]
        p2p = (diff * normal[e0]).sum(dim=-1).abs()
        curvature = p2p / (diff.norm(p=2, dim=-1) + 1e-4)
        mean_curv = curvature.reshape(-1, k).mean(-1)
        return mean_curv, float(mean_curv.max().item())

    thresholds = np.logspace(np.log(5) / np.log(10), np.log(0.01) / np.log(10), 100)
    pruned = False
    if cfg.get("PRUNE_KERNEL", True) and xyz.is_cuda and 0 < xyz.shape[0] and cfg["K"] <= 8:
        # the whole loop in one launch (pcs_prune_plane_hypotheses); more than 2048 hypotheses fall through
        try:
            keep = ops.prune_plane_hypotheses(xyz, normal, thresholds, cfg["K"]).nonzero(as_tuple=True)[0]
            xyz, normal, conf = xyz[keep], normal[keep], conf[keep]  # (one compaction of the mask, not three)
            pruned = True
        except PcsError as err:
            if "PCS_ERR_RANGE" not in str(err):
                raise
    if not pruned:
        # the k-NN curvature only changes when a hypothesis is dropped: evaluated once per pruning
        # event, not once per threshold (the reference recomputes it for each of the 100 thresholds)
        mean_curv, max_curv = mean_curvature(xyz, normal)
        for threshold in thresholds:
            if np.float32(threshold) > np.float32(max_curv):  # fp32, like the tensor comparison it replaces
                continue
            keep = mean_curv < threshold
            xyz, normal, conf = xyz[keep], normal[keep], conf[keep]
            mean_curv, max_curv = mean_curvature(xyz, normal)
    conf[:] = 1
    t_tls.__exit__(None, None, None)
    t_prop = Timer("ground: propagation to pillars")
    t_prop.__enter__()

    # every pillar takes the hypothesis with the best confidence / (distance + 1); the reference
    # scans the hypotheses in list order with a strict '>' update, i.e. the FIRST maximum wins:
    # one [pillars, hypotheses] table and a first-occurrence max
    if conf.shape[0] > 0:
        # (elementwise products and sums, not library norm / sum reductions: their internal order and
        # contraction are not ours to fix, and oracle/pcs_oracle_ground.cpp restates this arithmetic)
        dx = pillars.xyz[:, None, 0] - xyz[None, :, 0]
        dy = pillars.xyz[:, None, 1] - xyz[None, :, 1]
        dist = (dx * dx + dy * dy).sqrt()
        ind = conf[None, :] / (dist + 1)
        best_val, best_idx = ind.max(dim=1)  # (documented: the first maximal value's index)
        mask = best_val > pillars.best_confidence
        # (selects instead of masked assignments: the same values where the mask holds, no host synchronisation)
        pillars.best_center = torch.where(mask[:, None], xyz[best_idx], pillars.best_center)
        pillars.best_normal = torch.where(mask[:, None], normal[best_idx], pillars.best_normal)
        pillars.best_confidence = torch.where(mask, best_val, pillars.best_confidence)
    t_prop.__exit__(None, None, None)
    vn = pillars.best_normal[voxels.pillar_idx]
    vc = pillars.best_center[voxels.pillar_idx]
    vd = voxels.bxyz[:, 1:] - vc
    nz = vn[:, -1].abs().clamp(min=0.01) * ((vn[:, -1] >= 0).float() + 1) / 2
    vh = ((vd[:, 0] * vn[:, 0] + vd[:, 1] * vn[:, 1]) + vd[:, 2] * vn[:, 2]) / nz
    pillars.min_z = robust_mean(voxels.bxyz[:, -1] - vh, voxels.pillar_idx, num_pillars).reshape(X, Y)
    pillars.height = pillars.min_z.clone()
    return voxels, pillars


def format_pillars(points, pillar_size, pc_range_min):
    pillars = EasyDict()
    points["pillar_coords"] = torch.div(points.bxyz[:, 1:3] - pc_range_min, pillar_size,
                                        rounding_mode="floor").round().long()
    X, Y = (points.pillar_coords.max(0)[0].long() + 1).tolist()  # (one host read; host integers from here on)
    dims = (X, Y)
    num_pillars = X * Y
    points["pillar_idx"] = points.pillar_coords[:, 0] * Y + points.pillar_coords[:, 1]
    sg = IndexSegments(points.pillar_idx, num_pillars)
    pillars["density"] = _seg_sum(torch.ones_like(points.bxyz[:, 0]), sg, num_pillars).reshape(X, Y)
    pillars["min_z"] = segment_min(points.bxyz[:, -1], sg, num_pillars).reshape(X, Y)
    pillars["xyz"] = robust_mean(points.bxyz[:, 1:], sg, num_pillars).reshape(-1, 3)
    pillars["weight"] = (pillars.density > 0.5).float().reshape(-1)
    return dims, num_pillars, points, pillars


def l1_minimization(points, pillars, pillar_dims, cfg, max_countdown=3):
    X, Y = int(pillar_dims[0]), int(pillar_dims[1])
    pillars["height"] = ops.l1_minimization(pillars.min_z.reshape(X, Y), pillars.weight.reshape(X, Y),
                                            cfg["LR"], cfg["DECAY_STEPS"], cfg["RIGID_WEIGHT"],
                                            cfg["MAX_NUM_ITERS"])
    return pillars


def ground_plane_removal(point_bxyz, cfg, warmup=None):
    """Returns (height above ground, above-horizon flag, fitting error) per point and the pillar
    height / min_z grids (preprocessor_utils.py:352-419)."""
    points = EasyDict(bxyz=point_bxyz)
    pillar_size = torch.tensor(cfg["PILLAR_SIZE"]).to(points.bxyz)
    pc_range_min = points.bxyz[:, 1:3].min(0)[0] - 0.05
    with Timer("ground: grid sample + pillars"):
        voxels, point_voxel_index = grid_sample(points.bxyz, [0.10, 0.10, 0.03])
        pillar_dims, num_pillars, voxels, pillars = format_pillars(voxels, pillar_size, pc_range_min)
    if warmup is not None:
        pillars.height = warmup["pillar_height"]
        pillars.min_z = warmup["pillar_min_z"]
    else:
        if cfg.get("RANSAC", False):
            voxels, pillars = compute_min_height_from_ransac(pillar_dims, num_pillars, voxels, pillars, cfg)
        if cfg.get("JointOpt", False):
            with Timer("ground: l1 minimization"):
                pillars = l1_minimization(voxels, pillars, pillar_dims, cfg)
    ij = (voxels.pillar_coords[:, 0], voxels.pillar_coords[:, 1])
    voxel_height = pillars.height[ij]
    voxel_min_z = pillars.min_z[ij]
    voxel_horizon = voxels.bxyz[:, -1] > voxel_min_z
    voxel_height = voxels.bxyz[:, -1] - voxel_height
    fitting_error = voxel_height - voxel_min_z
    return voxel_height[point_voxel_index], voxel_horizon[point_voxel_index], \
        fitting_error[point_voxel_index], pillars.height, pillars.min_z
