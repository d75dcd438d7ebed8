import sys, os, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from pcseqlearning_amd import ops, lib as L
from pcseqlearning_amd.pcdet import models
from tools.run_sequence import build_model
def main():
    batch = models.load_data_to_gpu(bench.load_sequence(0, 20, 160000))
    model = build_model(save=False)
    lib = L.load()
    orig = ops.radius_graph
    def rg(ref, query, radius, max_nbr, *a, **k):
        out = orig(ref, query, radius, max_nbr, *a, **k)
        if max_nbr == 1 and query.shape[0] > 1000000:
            N, M = ref.shape[0], query.shape[0]
            ws = L.workspace(lib.pcs_radius_graph_workspace_bytes(L.i64(N), L.i64(M), L.i32(1)), ref.device)
            # the search's own workspace is gone; run it again into ws to read the counter
            d = torch.empty(1, dtype=torch.int64, device=ref.device)
            L.check(lib.pcs_radius_graph_search(L.ptr(ref.float().contiguous()), L.i64(N), L.ptr(query.float().contiguous()), L.i64(M),
                    L.f32(radius), L.f32(k.get("cell_size", 0.0)), L.i32(a[1] if len(a) > 1 else k.get("dframe_min", 0)),
                    L.i32(a[2] if len(a) > 2 else k.get("dframe_max", 0)), L.i32(1), L.i32(1), L.ptr(d), L.ptr(ws), L.i64(ws.numel()), L.stream()), "s")
            cnt = (ctypes.c_int32 * 8)()
            L.check(lib.pcs_radius_graph_counters(L.i64(N), L.i64(M), L.i32(1), L.ptr(ws), L.i64(ws.numel()), cnt, L.stream()), "c")
            print("nn1: refs", N, "queries", M, "passed filter", cnt[0], "left to the cooperative search", cnt[1], "spilled points", cnt[2], "edges", int(d.item()), "radius", radius, flush=True)
        return out
    ops.radius_graph = rg
    with torch.no_grad():
        model(batch)
        torch.cuda.synchronize()
if __name__ == "__main__":
    main()
