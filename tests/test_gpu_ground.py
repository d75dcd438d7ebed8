"""Ground-plane estimation pieces on the HIP path vs plain torch restatements of the reference's
formulas (preprocessor_utils.py:32-80 IRLS, :313-350 l1_minimization). Float parity: tolerances
stated per check (these stages have no integer outputs; their third-party leaves are unpinned)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from pcseqlearning_amd import ops  # noqa: E402

DEV = "cuda:0"


def _irls_torch(xyz, pidx, w0, P, sigma2, delta=1e-2):
    """The reference loop, literally, with torch ops (fp32) and normals flipped to n_z >= 0."""
    w = w0.reshape(-1, 1)
    cnt = torch.zeros(P, device=xyz.device).index_add_(0, pidx, torch.ones_like(xyz[:, 0])).clamp(min=1)
    for _ in range(50):
        center = torch.zeros(P, 3, device=xyz.device).index_add_(0, pidx, xyz * w)
        wsum = torch.zeros(P, 1, device=xyz.device).index_add_(0, pidx, w) + 1e-6
        center = center / wsum
        d = xyz - center[pidx]
        ddT = (w[:, :, None] * d[:, :, None]) * d[:, None, :]
        cov = torch.zeros(P, 3, 3, device=xyz.device).index_add_(0, pidx, ddT) / cnt[:, None, None]
        _, Q = torch.linalg.eigh(cov.double())
        normal = Q[:, :, 0].float()
        normal = normal * torch.where(normal[:, 2:3] < 0, -1.0, 1.0)
        err = (d * normal[pidx]).sum(-1).abs()
        new_w = (sigma2 / (err.square() + sigma2)) * (0.25 / (d.square().sum(-1) + 0.25))
        new_w = new_w.reshape(-1, 1)
        if (new_w - w).abs().max() < delta:
            break
        w = new_w
    return center, normal, err


def test_irls_plane_fit_vs_torch():
    g = torch.Generator(device="cpu").manual_seed(0)
    P, per = 40, 600
    pts, idx = [], []
    for p in range(P):
        cx, cy = torch.rand(2, generator=g) * 100 - 50
        n = torch.tensor([0.05 * (p % 5 - 2), 0.03 * (p % 3 - 1), 1.0]); n = n / n.norm()
        xy = torch.rand(per, 2, generator=g) * 8 - 4
        z = -(n[0] * xy[:, 0] + n[1] * xy[:, 1]) / n[2] + torch.randn(per, generator=g) * 0.02
        z[: per // 5] += torch.rand(per // 5, generator=g) * 2.0  # objects above the ground
        pts.append(torch.stack([xy[:, 0] + cx, xy[:, 1] + cy, z], 1)); idx.append(torch.full((per,), p))
    xyz = torch.cat(pts).float().to(DEV)
    pidx = torch.cat(idx).long().to(DEV)
    w0 = (0.0025 / ((xyz[:, 2] - 0.0).square() + 0.0025)).reshape(-1, 1)
    origin = torch.zeros(P, 3, device=DEV).index_add_(0, pidx, xyz) / per
    rc, rn, re = _irls_torch(xyz, pidx, w0.reshape(-1), P, 0.0025)
    seg_start = torch.searchsorted(pidx, torch.arange(P + 1, device=DEV)).int()
    for seg in (None, seg_start):  # atomics path, one-launch-per-round path on grouped rows
        c, n, e = ops.irls_plane_fit(xyz, pidx, w0, origin, P, 0.0025, seg_start=seg)
        assert torch.allclose(c, rc, atol=2e-4), (c - rc).abs().max()
        assert torch.allclose(n, rn, atol=2e-4), (n - rn).abs().max()
        assert torch.allclose(e, re, atol=3e-4), (e - re).abs().max()
        assert (n[:, 2] > 0.9).all()
    # fixed-order sums: the grouped path is bit-reproducible
    c2, n2, e2 = ops.irls_plane_fit(xyz, pidx, w0, origin, P, 0.0025, seg_start=seg_start)
    assert torch.equal(c, c2) and torch.equal(n, n2) and torch.equal(e, e2)


def test_l1_minimization_vs_torch_adamw():
    """The persistent kernel against torch.optim.AdamW on the reference's loss, same schedule."""
    torch.manual_seed(0)
    X, Y = 24, 31
    min_z = (torch.randn(X, Y) * 0.3).to(DEV)
    weight = (torch.rand(X, Y) > 0.3).float().to(DEV)
    got = ops.l1_minimization(min_z, weight, 0.01, [1600], 0.5, 3000)
    h = torch.nn.Parameter(torch.zeros(X, Y, device=DEV))
    opt = torch.optim.AdamW([h], lr=0.01)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, [1600])
    last, countdown = 1e10, 3
    w = weight
    for _ in range(3000):
        opt.zero_grad()
        l1 = ((h - min_z) * w).abs().mean()
        left = ((h[:-2] - 2 * h[1:-1] + h[2:]) * (w[1:-1] + 1e-2)).abs().mean()
        up = ((h[:, :-2] - 2 * h[:, 1:-1] + h[:, 2:]) * (w[:, 1:-1] + 1e-2)).abs().mean()
        t1 = ((h[:-2, :-2] - 2 * h[1:-1, 1:-1] + h[2:, 2:]) * (w[1:-1, 1:-1] + 1e-2)).abs().mean()
        t2 = ((h[2:, :-2] - 2 * h[1:-1, 1:-1] + h[:-2, 2:]) * (w[1:-1, 1:-1] + 1e-2)).abs().mean()
        loss = l1 + (left + up + t1 + t2) * 0.5
        loss.backward()
        opt.step(); sch.step()
        if last - loss.item() < 1e-4:
            countdown -= 1
        else:
            countdown = 3
        if countdown == 0:
            break
        last = loss.item()
    # sign() kinks make individual trajectories diverge slightly; the optimum they reach agrees
    assert (got - h.data).abs().mean() < 5e-3, (got - h.data).abs().mean()
    assert (got - h.data).abs().max() < 5e-2, (got - h.data).abs().max()
