#!/usr/bin/env python3
"""Experiment (invalid as a result, valid as a clock): bench.py with the proposal's evaluation and / or the extraction
of traces SKIPPED -- an upper bound of what taking their ~900 tensor-library launches per sequence off the hardware
queues could buy.   python tools/exp_skip_stages.py {none|eval|extract|eval_extract} [bench.py arguments]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode = sys.argv[1]
    sys.argv = [sys.argv[0]] + sys.argv[2:]
    import bench
    from pcseqlearning_amd.pcdet.models.registration.preprocessors import cluster_proposal as cp, cluster_tracking as ct
    if "eval" in mode:
        orig_fwd = cp.ClusterProposal.forward

        def fwd(self, seq_dict):
            out = orig_fwd(self, seq_dict)
            out["_deferred"] = []
            return out
        cp.ClusterProposal.forward = fwd
    if "extract" in mode:
        from pcseqlearning_amd.pcdet.utils.common_utils import EasyDict
        ct.ClusterTracking.extract_traces_of_units = (
            lambda self, all_points, extracted_list, seq_boxes, views=None: list(extracted_list))
        ct.ClusterTracking._extract_units_native = (
            lambda self, seq_fxyz, units, tracked, perm, views, seq_boxes, windows:
            [EasyDict(fxyz=seq_fxyz[:0]) for _ in units])
    bench.main()


if __name__ == "__main__":
    main()
