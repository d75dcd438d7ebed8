"""component_box_iou (vectorised GT bookkeeping) against the reference's per-component loop
(cluster_proposal.py:238-255 / cluster_tracking.py:388-411 restated literally). CPU torch."""
import torch

from pcseqlearning_amd.pcdet.models.registration.preprocessors.eval_utils import component_box_iou, scatter_max_


def test_component_box_iou_matches_loop():
    g = torch.Generator().manual_seed(0)
    n, B = 3000, 12
    comp = torch.randint(0, 60, (n,), generator=g) * 3  # non-contiguous ids
    bp = (torch.rand(B, n, generator=g) < 0.05).long()
    in_box = (bp == 1).any(0)
    gt = torch.full((n,), -1, dtype=torch.long)
    gt[in_box] = bp[:, in_box].argmax(0)
    gt_cnt = torch.bincount(gt[in_box], minlength=B)
    uniq, inv, has_box, box, iou = component_box_iou(comp, bp, gt, gt_cnt)
    best = torch.zeros(B)
    want_best = torch.zeros(B)
    for k, c in enumerate(uniq.tolist()):
        cm = comp == c
        bi = bp[:, cm]
        assert bool(bi.any()) == bool(has_box[k])
        if not bi.any():
            continue
        b = int(bi.sum(-1).argmax())
        assert b == int(box[k])
        m1, m2 = gt == b, cm
        w = (m1 & m2).float().sum().item() / ((m1 | m2).float().sum().item() + 1e-6)
        assert abs(w - float(iou[k])) < 1e-12
        if w > want_best[b]:
            want_best[b] = w
    scatter_max_(best, box[has_box], iou[has_box].float())
    assert torch.allclose(best, want_best)


def test_component_box_iou_frames_matches_per_frame():
    """The all-frames form (groups = (frame, component), boxes indexed within their frame) against the
    per-frame component_box_iou it replaces in the evaluation / extraction loops."""
    from pcseqlearning_amd.pcdet.models.registration.preprocessors.eval_utils import component_box_iou_frames
    g = torch.Generator().manual_seed(1)
    F, n = 6, 4000
    nbox = torch.tensor([5, 0, 9, 3, 0, 7])            # frames 1 and 4 have no boxes
    box_off = torch.zeros(F + 1, dtype=torch.long)
    box_off[1:] = nbox.cumsum(0)
    bmax = int(nbox.max())
    frame = torch.randint(0, F, (n,), generator=g)
    frame = frame[(nbox[frame] > 0)]                    # the callers drop frames without boxes
    n = frame.numel()
    comp = torch.randint(0, 40, (n,), generator=g) * 2
    member = torch.zeros(n, bmax, dtype=torch.int32)
    hit = torch.rand(n, bmax, generator=g) < 0.08
    member[hit & (torch.arange(bmax)[None, :] < nbox[frame][:, None])] = 1
    in_box = member.any(1)
    gt_local = torch.where(in_box, member.argmax(1), torch.full((n,), -1))
    gt_cnt_all = torch.bincount((box_off[frame] + gt_local)[in_box], minlength=int(box_off[-1]))
    gframe, gcomp, inv, has_box, box, iou = component_box_iou_frames(comp, frame, member, gt_local, box_off, gt_cnt_all)
    pos = 0
    for f in range(F):
        sel = (frame == f).nonzero(as_tuple=True)[0]
        if sel.numel() == 0:
            continue
        B = int(nbox[f])
        bp = member[sel, :B].t().long()
        uniq, inv_f, hb, bx, io = component_box_iou(comp[sel], bp, gt_local[sel],
                                                  gt_cnt_all[box_off[f]:box_off[f + 1]])
        k = uniq.numel()
        assert torch.equal(gframe[pos:pos + k], torch.full((k,), f)) and torch.equal(gcomp[pos:pos + k], uniq)
        assert torch.equal(has_box[pos:pos + k], hb)
        assert torch.equal(box[pos:pos + k][hb], bx[hb])
        assert torch.equal(iou[pos:pos + k][hb], io[hb])
        assert torch.equal(inv[sel] - pos, inv_f)
        pos += k
    assert pos == gframe.numel()


def test_no_boxes_is_not_an_error():
    """A sequence without GT boxes: counting into zero bins and a running maximum over zero boxes
    are no-ops, as bincount(minlength=0) / an empty best_iou are in the reference."""
    from pcseqlearning_amd.pcdet.models.registration.preprocessors.eval_utils import count_into, scatter_max_
    idx = torch.zeros(5, dtype=torch.int64)
    assert count_into(idx, 0).numel() == 0
    assert count_into(idx, 0, torch.ones(5, dtype=torch.bool)).numel() == 0
    best = torch.zeros(0, dtype=torch.float64)
    assert scatter_max_(best, idx, torch.ones(5, dtype=torch.float64)).numel() == 0
    assert count_into(torch.tensor([0, 2, 2]), 3).tolist() == [1, 0, 2]
