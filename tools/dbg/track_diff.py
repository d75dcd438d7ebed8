import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_gpu_tracking as TT
from oracle import track_oracle as T
g = np.load(os.path.join(TT.G, "track_small.npz"))
comp = g["component"]; C = int(comp.max()) + 1
o_diam = T.component_diameter(g["fxyz"], comp, C)
for native in (False, True):
    seq = dict(fxyz=g["fxyz"].copy(), frame=g["fxyz"][:, 0].astype(np.int64), component=comp,
               stationary=o_diam[comp] > 12.5, extracted=np.zeros(len(comp), bool))
    ot = T.Tracker()
    tracker, sp = TT._tracker(), TT._prepare(g, native)
    for frame_id in (0, 8):
        want = ot.track_frame(seq, frame_id)
        ex = TT._track(tracker, sp, frame_id)
        d = np.abs(ex.transforms.cpu().numpy() - want["transforms"])
        idx = np.argwhere(d > 1e-5)
        print("native", native, "frame", frame_id, "max", d.max(), "viol", idx.tolist()[:10])
        for k in ("original_indices", "component"):
            print("  ", k, np.array_equal(ex[k].cpu().numpy(), want[k]))
        if len(idx):
            c = idx[0][0]
            print("  comp", c, "npts anchor", int((ex.component.cpu().numpy() == c).sum()))
            print(ex.transforms.cpu().numpy()[c, idx[0][1]]); print(want["transforms"][c, idx[0][1]])
