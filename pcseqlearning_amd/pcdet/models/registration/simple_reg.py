"""SimpleReg (pcdet/models/registration/simple_reg.py:13-157): per batch item build the
`point_fxyz` sequence, subsample it on an 8 cm 4-D grid (keeping the largest original index of
every occupied cell), format the GT boxes and run the preprocessors in order."""
import os

import numpy as np
import torch
import torch.nn as nn

from .... import ops
from ...utils import timer
from ...utils.common_utils import EasyDict, filter_dict, run_deferred, stream_slot, thread_stream
from ...utils.timer import Timer
from ..model_utils.grid_sampling import GridSampling3D
from .registration_module_template import RegistrationTemplate


def boxes_to_corners_3d(boxes):
    """[N,7] (x,y,z,dx,dy,dz,heading) -> [N,8,3] (pcdet/utils/box_utils.py, same corner order)."""
    template = boxes.new_tensor([[1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1],
                                 [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1]]) / 2
    corners = boxes[:, None, 3:6] * template[None]
    c, s = torch.cos(boxes[:, 6]), torch.sin(boxes[:, 6])
    rot = torch.stack([torch.stack([c, s, torch.zeros_like(c)], -1),
                       torch.stack([-s, c, torch.zeros_like(c)], -1),
                       torch.stack([torch.zeros_like(c), torch.zeros_like(c), torch.ones_like(c)], -1)], 1)
    return corners @ rot + boxes[:, None, 0:3]


def _record_stream(obj, stream):
    """record_stream on every device tensor of a (nested) result dictionary"""
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream)


_SEQUENCE_POOLS = {}
_SEQUENCE_POOLS_LOCK = __import__("threading").Lock()


def _sequence_pool(workers):
    """Process-wide pool of `workers` host threads for the sequences of a batch (kept outside the module:
    a model stays picklable / deep-copyable and its workers survive from call to call)."""
    from concurrent.futures import ThreadPoolExecutor
    with _SEQUENCE_POOLS_LOCK:
        pool = _SEQUENCE_POOLS.get(workers)
        if pool is None:
            pool = _SEQUENCE_POOLS[workers] = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="pcs-seq")
        return pool


class SimpleReg(RegistrationTemplate):
    def __init__(self, model_cfg, runtime_cfg, dataset):
        super().__init__(model_cfg, runtime_cfg, dataset)
        self.module_list = self.build_networks()
        self.fake_param = nn.Parameter(torch.zeros(1), requires_grad=True)
        self.forward_dict = EasyDict()
        self.subsample = model_cfg.get("SUBSAMPLE", False)
        if self.subsample:
            self.grid_sampler = GridSampling3D(grid_size=[0.08, 0.08, 0.08])

    def process_sequence(self, seq_dict, preprocessors=None, gate=None):
        """`gate` (a semaphore, _process_batch_concurrently): at most that many sequences of a batch are in the stages
        BEFORE tracking at once. Twelve sequences that all start with ground removal and the proposal graphs share
        the chip twelve ways and reach the tracker's engines together and late; a few at a time reach them one after
        the other, early, and the engines -- whose ticks are what a batch takes -- start while the rest queue."""
        mods = list((self.preprocessors or []) if preprocessors is None else preprocessors)
        n_pre = len(mods) - 1 if gate is not None else 0  # (the last module is the tracker)
        if n_pre > 0:
            with gate:
                for module in mods[:n_pre]:
                    seq_dict = module(seq_dict)
        for module in mods[n_pre:]:
            seq_dict = module(seq_dict)
        return run_deferred(seq_dict)

    def _process_batch_concurrently(self, batch_dict, todo, inflight):
        """The sequences of a batch are independent (the reference loops over them, simple_reg.py:105-147):
        `inflight` of them are processed at once, each on its own host thread, HIP stream and private copy
        of the preprocessor modules (made per call: they hold no parameters, only small configuration
        objects; nothing of a worker outlives the call). A sequence's tracking stage is a chain of small,
        latency-bound launches whose duration hardly depends on their width: the native tracker merges the
        units of ALL sequences in flight into the launches of the device's lock-step engines (pcs_track_sequence),
        while the ground / proposal / extraction stages of the other sequences run on their own streams.
        Results do not depend on what runs beside a sequence (tests/test_gpu_backbone_pipeline.py)."""
        import copy
        import threading
        device = batch_dict["point_bxyz"].device
        main = torch.cuda.current_stream(device)
        grad = torch.is_grad_enabled()
        results, errors = {}, []
        lock = threading.Lock()
        # worker i takes sequences i, i + inflight, ... of the batch: the same worker (= the same streams and the same
        # pools of the caching allocator) sees the same sequence sizes at every step; handing the sequences out first
        # come first served let every pool grow to the largest request of any sequence (reserved memory 88 -> 106 GiB
        # over 100 batches)
        todo = list(todo)

        # workers start 5 ms apart (SEQUENCE_START_STAGGER_MS): sequences that enter the ground / proposal
        # stages at the same instant compete for the same resources at the same moments (measured with 12
        # in flight: 0 / 5 / 25 ms -> 1.78 / 1.75 / 1.83 s per batch; 25 ms was round 2's value for 4 in
        # flight and leaves the twelfth worker idle for 275 ms)
        stagger = float(self.model_cfg.get("SEQUENCE_START_STAGGER_MS", 5)) * 1e-3
        shared = str(self.model_cfg.get("WORKER_STREAMS", "own")) == "caller"
        # PRE_TRACKING_CONCURRENCY: sequences allowed in ground removal / proposal at once (0: no limit)
        slots = int(self.model_cfg.get("PRE_TRACKING_CONCURRENCY", 0))
        gate = threading.BoundedSemaphore(slots) if 0 < slots < inflight else None

        def worker(i=0):
            if stagger > 0 and i > 0:
                import time
                time.sleep(stagger * i)
            mods = copy.deepcopy(self.preprocessors)
            slot = stream_slot(i)
            slot.__enter__()
            try:
                # this worker's own stream, the same at every call -- or (WORKER_STREAMS: "caller") the caller's
                # stream for every worker: the runtime deals streams to its four hardware queues, three of which
                # carry the tracker's engines, and a worker's 8 ms search kernel on an engine's queue holds up that
                # engine's chain
                stream = main if shared else thread_stream(device)
                stream.wait_stream(main)
                with torch.cuda.device(device), torch.cuda.stream(stream), torch.set_grad_enabled(grad):
                    for b in todo[i::inflight]:
                        with lock:
                            if errors:
                                break
                        out = self.process_sequence(self.build_sequence(batch_dict, b), mods, gate)
                        _record_stream(out, main)  # the caller's stream consumes what this stream allocated
                        results[b] = out
                    stream.synchronize()
            except BaseException as e:  # re-raised on the calling thread
                with lock:
                    errors.append(e)
            finally:
                slot.__exit__()

        # persistent workers (module-level, not module state): the native tracker keeps a stream, an event
        # and a pinned status page per host thread, so short-lived threads would leave hundreds of them
        # behind (measured: 6.3 -> 2.2 sequences/s over 40 batches with a fresh thread per sequence slot)
        futures = [_sequence_pool(inflight).submit(worker, i) for i in range(inflight)]
        for f in futures:
            f.result()
        if errors:
            raise errors[0]
        return results

    def format_boxes(self, seq_dict):
        """Frame ids of the (frame-major, equally many per frame) GT boxes, track labels from the
        object ids, per-trace velocity from consecutive corner displacement, `moving` flag."""
        sweeps = seq_dict["point_sweep"]
        lo, hi = torch.stack([sweeps.min(), sweeps.max()]).long().tolist()  # (one host read)
        num_frames = hi - lo + 1
        attr = seq_dict["gt_box_attr"].reshape(-1, 7)
        cls = seq_dict["gt_box_cls_label"].reshape(-1)
        assert attr.shape[0] % num_frames == 0, "expects the same number of boxes in every frame"
        frame = torch.repeat_interleave(torch.arange(0, num_frames), cls.numel() // num_frames).to(cls)
        keep = (attr[:, 3:6].norm(p=2, dim=-1) > 1e-5).nonzero(as_tuple=True)[0]  # (one compaction, one host copy)
        attr, cls, frame = attr[keep], cls[keep], frame[keep]
        obj_ids = np.asarray(seq_dict["obj_ids"]).astype(str).reshape(-1)[keep.cpu().numpy()]
        track = torch.from_numpy(np.unique(obj_ids, return_inverse=True)[1]).to(cls).long()
        # per trace, boxes in frame order: velocity = mean corner displacement to the previous box of
        # the trace, the first box takes the second's (the reference loops over the traces; here all
        # traces at once on the (trace, frame)-sorted list)
        velo = torch.zeros_like(attr[:, 0])
        if attr.shape[0] > 1:
            # (any multiplier above the largest frame id gives the same (trace, frame) order: no host read for it)
            order = torch.argsort(track * num_frames + frame.long(), stable=True)
            tr_s = track[order]
            corners = boxes_to_corners_3d(attr[order])
            same = tr_s[1:] == tr_s[:-1]
            step = (corners[1:] - corners[:-1]).norm(p=2, dim=-1).mean(dim=-1)
            tv = torch.zeros_like(velo)
            tv[1:] = torch.where(same, step, torch.zeros_like(step))
            first = torch.ones_like(tr_s, dtype=torch.bool)
            first[1:] = ~same
            nxt = torch.zeros_like(tv)
            nxt[:-1] = tv[1:]                       # value of the following row ...
            has_next = torch.zeros_like(first)
            has_next[:-1] = same                    # ... if it belongs to the same trace
            tv = torch.where(first, torch.where(has_next, nxt, torch.zeros_like(tv)), tv)
            velo[order] = tv
        seq_dict.update(gt_box_attr=attr, gt_box_cls_label=cls, gt_box_frame=frame,
                        gt_box_track_label=track, gt_box_velo=velo, moving=velo > 5e-2,
                        obj_ids=obj_ids)
        return seq_dict

    def build_sequence(self, batch_dict, b):
        seq_dict = EasyDict()
        mask = (batch_dict["point_bxyz"][:, 0] == b).reshape(-1).nonzero(as_tuple=True)[0]  # (one compaction for all keys)
        for key in ("point_bxyz", "point_feat", "segmentation_label", "instance_label",
                    "is_foreground", "point_sweep"):
            if key in batch_dict:
                seq_dict[key] = batch_dict[key][mask]
        seq_dict["point_fxyz"] = torch.cat([seq_dict["point_sweep"].reshape(-1, 1).to(seq_dict["point_bxyz"]),
                                            seq_dict["point_bxyz"][:, 1:]], dim=-1)
        seq_dict.pop("point_bxyz")
        if self.subsample:
            with Timer("Subsample (8 cm grid pick)"):
                n_in = seq_dict["point_fxyz"].shape[0]
                seg = self.grid_sampler.segments(seq_dict["point_fxyz"].contiguous())
                pick = ops.segment_pick(seg, last=True)  # scatter(arange, inv, 'max')
                for key in ("point_fxyz", "point_feat", "segmentation_label", "instance_label",
                            "is_foreground", "point_sweep"):
                    if key in seq_dict:
                        seq_dict[key] = seq_dict[key][pick]
                timer.count("subsample points in", n_in)
                timer.count("subsample cells out", pick.shape[0])
        for key in ("gt_box_cls_label", "gt_box_attr", "obj_ids", "frame_id"):
            if key in batch_dict:
                seq_dict[key] = batch_dict[key][b]
        return self.format_boxes(seq_dict)

    def forward(self, batch_dict):
        batch_size = batch_dict["batch_size"]
        if batch_dict["point_bxyz"].is_cuda:
            # the tracker's engines before any worker creates a stream of its own (see pcs_track_engines_init)
            ops.track_engines_init(batch_dict["point_bxyz"].device)
        save_dir = self.model_cfg.get("SAVE_DIR", None)

        def done_before(b):
            sequence_id = batch_dict["frame_id"][b][0][:-4]
            return save_dir is not None and os.path.exists(f"{save_dir}/{sequence_id}/all.pth")

        # SEQUENCES_IN_FLIGHT (default 12): how many sequences of a batch are processed at once (their
        # tracking units share the launches of the device's lock-step engine)
        inflight = min(int(self.model_cfg.get("SEQUENCES_IN_FLIGHT", 12)), batch_size)
        todo = [b for b in range(batch_size) if not done_before(b)]
        processed = {}
        if inflight > 1 and len(todo) > 1 and batch_dict["point_bxyz"].is_cuda:
            processed = self._process_batch_concurrently(batch_dict, todo, min(inflight, len(todo)))
        outputs = []
        for b in range(batch_size):
            if b in processed:
                outputs.append(processed[b])
                continue
            seq_dict = self.build_sequence(batch_dict, b)
            if b in todo:
                seq_dict = self.process_sequence(seq_dict)
            outputs.append(seq_dict)
        self.last_outputs = outputs
        if self.training:
            device = batch_dict["point_bxyz"].device
            return dict(loss=torch.zeros(1, device=device, requires_grad=True)), {}, {}
        return {}, None
