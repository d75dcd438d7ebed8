"""Per-workgroup timestamps of the three ICP kernels (a build with -DPCS_WG_TRACE: tools/wg_trace.sh):
where inside a launch the time goes -- start spread of the workgroups, mean and longest workgroup,
phases. One sequence, single-group schedule (every launch serves all 9 units)."""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PCS_TRACK_GROUPS", "1")
import numpy as np
import torch
import bench
from pcseqlearning_amd import lib as L
from pcseqlearning_amd.pcdet import models
from tools.run_sequence import build_model

batch = models.load_data_to_gpu(bench.load_sequence(0, 20, 160000))
model = build_model(save=False, device="cuda:0")
with torch.no_grad():
    model(batch)
torch.cuda.synchronize()
lib = L.load()
CAP = 6 << 20
buf = torch.zeros(CAP * 5, dtype=torch.int64, device="cuda:0")
lib.pcs_wg_trace_set.restype = ctypes.c_int
lib.pcs_wg_trace_count.restype = ctypes.c_uint
assert lib.pcs_wg_trace_set(ctypes.c_void_p(buf.data_ptr())) == 0
with torch.no_grad():
    model(batch)
torch.cuda.synchronize()
n = min(int(lib.pcs_wg_trace_count()), CAP)
print("records", n)
a = buf[: n * 5].cpu().numpy().astype(np.uint64).reshape(n, 5)
tag, t0, t1, m1, m2 = a[:, 0], a[:, 1].astype(np.int64), a[:, 2].astype(np.int64), a[:, 3].astype(np.int64), a[:, 4].astype(np.int64)
kid = (tag >> np.uint64(56)).astype(np.int64)
kk = ((tag >> np.uint64(40)) & np.uint64(0xFFFF)).astype(np.int64)
yy = ((tag >> np.uint64(32)) & np.uint64(0xFF)).astype(np.int64)
TICK = 0.01  # us per s_memrealtime tick (100 MHz)
names = {1: "icp_nn (budget test)", 2: "search, cooperative", 3: "search, seeded", 4: "apply"}
for K in (1, 2, 3, 4):
    sel = np.nonzero(kid == K)[0]
    if len(sel) == 0:
        continue
    # launches: sort by start, split where the start jumps by more than 8 us
    o = sel[np.argsort(t0[sel], kind="stable")]
    cuts = np.nonzero(np.diff(t0[o]) > 800)[0] + 1
    groups = np.split(o, cuts)
    span, spread, mean_d, max_d, nwg, ph1, ph2, ph3, worst_y = [], [], [], [], [], [], [], [], []
    for g in groups:
        if len(g) < 4:
            continue
        s0 = t0[g].min()
        span.append((t1[g].max() - s0) * TICK)
        spread.append((t0[g].max() - s0) * TICK)
        d = (t1[g] - t0[g]) * TICK
        mean_d.append(d.mean())
        max_d.append(d.max())
        nwg.append(len(g))
        w = g[np.argmax(d)]
        worst_y.append(int(yy[w]))
        if K in (1, 4):
            ph1.append(((m1[g] - t0[g]) * TICK).mean())
            ph2.append(((m2[g] - m1[g]) * TICK).mean())
            ph3.append(((t1[g] - m2[g]) * TICK).mean())
        else:
            ph2.append(((m2[g] - t0[g]) * TICK).mean())
            ph3.append(((t1[g] - m2[g]) * TICK).mean())
    f = lambda x: f"{np.mean(x):7.2f}"
    print(f"{names[K]:24s} launches {len(span):5d}  workgroups/launch {np.mean(nwg):7.1f}  span {f(span)} us  "
          f"start spread {f(spread)}  mean wg {f(mean_d)}  longest wg {f(max_d)}")
    if ph1:
        print(f"{'':24s} phases (mean over wgs): head {f(ph1)}  body {f(ph2)}  tail {f(ph3)}")
    else:
        print(f"{'':24s} phases (mean over wgs): body {f(ph2)}  flush {f(ph3)}")
    span = np.array(span); max_d = np.array(max_d); spread = np.array(spread)
    print(f"{'':24s} span percentiles 50/90/99: {np.percentile(span, 50):.1f} {np.percentile(span, 90):.1f} {np.percentile(span, 99):.1f};"
          f" longest-wg 50/90/99: {np.percentile(max_d, 50):.1f} {np.percentile(max_d, 90):.1f} {np.percentile(max_d, 99):.1f};"
          f" spread 50/90/99: {np.percentile(spread, 50):.1f} {np.percentile(spread, 90):.1f} {np.percentile(spread, 99):.1f}")
    print(f"{'':24s} unit slot of the longest workgroup: {np.bincount(np.array(worst_y), minlength=9).tolist()}")

# apply: what distinguishes the longest workgroup of a launch (rounds it ran, component owners among its threads)
sel = np.nonzero(kid == 4)[0]
if len(sel):
    o = sel[np.argsort(t0[sel], kind="stable")]
    cuts = np.nonzero(np.diff(t0[o]) > 800)[0] + 1
    rl, ol, rm, om = [], [], [], []
    for g in np.split(o, cuts):
        if len(g) < 4:
            continue
        d = t1[g] - t0[g]
        w = g[np.argmax(d)]
        rounds = (a[g, 3] >> np.uint64(32)).astype(np.int64)
        owners = (a[g, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
        rl.append(int(a[w, 3] >> np.uint64(32))); ol.append(int(a[w, 3] & np.uint64(0xFFFFFFFF)))
        rm.append(rounds.mean()); om.append(owners.mean())
    print(f"apply: longest workgroup ran {np.mean(rl):.2f} rounds with {np.mean(ol):.1f} owners; the mean workgroup {np.mean(rm):.2f} rounds, {np.mean(om):.1f} owners")
cnt = (ctypes.c_ulonglong * 16)()
if lib.pcs_g2_counters(cnt) == 0:
    c = list(cnt)
    print(f"cooperative searches {c[0]}: far walk in {c[1]} ({100.0 * c[1] / max(c[0], 1):.1f} %), far rounds per walking query "
          f"{c[2] / max(c[1], 1):.2f}, far cells per walking query {c[3] / max(c[1], 1):.1f}; rounds histogram 1/2/3-4/5-8/9+: {c[4:9]}")
    print(f"apply: waves {c[11]}, with at least one lane on the Jacobi SVD path {c[10]} ({100.0 * c[10] / max(c[11], 1):.1f} %), such lanes {c[9]}")
