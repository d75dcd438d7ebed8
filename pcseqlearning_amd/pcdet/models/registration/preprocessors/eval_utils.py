"""GT-box bookkeeping shared by ClusterProposal.evaluate_proposal (cluster_proposal.py:238-255) and
ClusterTracking.extract_traces_and_update_boxes (cluster_tracking.py:388-411): the reference loops
over components in Python (a mask, two reductions and two `.item()` syncs per component); the same
quantities are histograms, computed here in a handful of device ops."""
import torch


def count_into(index, size, weight=None):
    """torch.bincount(index[weight], minlength=size) for indices known to be < size, without bincount's and the mask's
    host synchronisations (integer adds: the same counts)."""
    out = torch.zeros(size, dtype=torch.int64, device=index.device)
    if index.numel() and size:  # (size 0 -- a sequence without GT boxes: bincount(minlength=0) of nothing is empty)
        out.index_add_(0, index, torch.ones_like(index) if weight is None else weight.to(torch.int64))
    return out


def component_box_iou(component, bp_mask, point_gt_box_id, gt_count_per_box):
    """For every distinct component of `component` [n]:
      box  = argmax_b #points of the component inside box b   (bp_mask [B,n], 0/1)
      iou  = |comp & gt(box)| / (|comp| + |gt(box)| - |comp & gt(box)| + 1e-6)
    with gt(box) = points whose (single) gt box id is `box`; |gt(box)| given by gt_count_per_box [B].
    Returns (uniq [U], inv [n], has_box [U] bool, box [U] long, iou [U] float64)."""
    uniq, inv = component.unique(return_inverse=True)
    U, B = uniq.numel(), bp_mask.shape[0]
    counts = torch.zeros(U, B, dtype=torch.int64, device=component.device)
    counts.index_add_(0, inv, bp_mask.t().long())
    has_box = counts.sum(1) > 0
    box = counts.argmax(1)
    size = count_into(inv, U)
    in_gt = point_gt_box_id >= 0
    hist = count_into(inv * B + point_gt_box_id.long().clamp(min=0), U * B, in_gt).reshape(U, B)
    inter = hist.gather(1, box[:, None])[:, 0]
    union = size + gt_count_per_box.long()[box] - inter
    iou = inter.double() / (union.double() + 1e-6)
    return uniq, inv, has_box, box, iou


def scatter_max_(dst, index, values):
    """dst[index] = max(dst[index], values) (values already in dst's dtype)."""
    if index.numel() and dst.numel():  # (no boxes: nothing to update)
        dst.scatter_reduce_(0, index, values, reduce="amax", include_self=True)
    return dst


def component_box_iou_frames(component, frame, member, gt_local, box_offsets, gt_count_all, span=None):
    """component_box_iou for many frames at once: a group is a (frame, component) pair, boxes are
    indexed within their frame (`member` [n,Bmax] from ops.points_in_frame_boxes, `gt_local` [n] the
    point's gt box within its frame or -1, `box_offsets` [F+1] long, `gt_count_all` [B] per global box).
    Every point's frame must have boxes. Returns (group_frame [G], group_comp [G], inv [n],
    has_box [G], box_local [G], iou [G] float64), groups ordered by (frame, component) as the
    per-frame loop visits them."""
    # `member`: the [n, bmax] matrix, or a pair (bmax, counts_of) with counts_of(inv, G) -> [G, bmax] hits per group
    # (ops.frame_box_counts: the matrix is then never formed)
    if isinstance(member, tuple):
        bmax, counts_of = member
        n = component.shape[0]
    else:
        (n, bmax), counts_of = member.shape, None
    comp = component.long()
    if span is None:  # (any bound above the largest component id gives the same groups in the same order; a caller
        span = int(comp.max().item()) + 1 if n else 1  # that knows one saves the host read)
    key = frame.long() * span + comp
    uniq, inv, size = key.unique(return_inverse=True, return_counts=True)  # (the groups' sizes come with the sort)
    G = uniq.numel()
    gframe, gcomp = uniq // span, uniq % span
    # (in the membership's own int32: a group holds far fewer than 2^31 points, and the int64 copy of [n, bmax] was
    #  the largest array of the whole extraction)
    if counts_of is not None:
        counts = counts_of(inv, G)
    else:
        counts = torch.zeros(G, bmax, dtype=member.dtype, device=comp.device)
        counts.index_add_(0, inv, member)
    has_box = counts.sum(1) > 0
    box = counts.argmax(1)
    # |component & gt(box)|: the group's points whose own gt box is the group's box (one count per group, not the
    # [G, bmax] histogram of which only column `box` was read)
    inter = count_into(inv, G, (gt_local >= 0) & (gt_local.long() == box[inv]))
    gbox = (box_offsets[gframe] + box).clamp(max=max(gt_count_all.numel() - 1, 0))
    union = size + gt_count_all.long()[gbox] - inter
    iou = inter.double() / (union.double() + 1e-6)
    return gframe, gcomp, inv, has_box, box, iou
