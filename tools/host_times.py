"""Where the HOST time of one sequence goes: per stage wall span and CPU time of the interpreter thread
(the part that holds the GIL and therefore serialises when several sequences are in flight), plus the
cProfile top list by own time.  python tools/host_times.py [top]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pcseqlearning_amd.pcdet import models  # noqa: E402
from pcseqlearning_amd.pcdet.utils import timer  # noqa: E402
from tools.run_sequence import build_model  # noqa: E402


def main():
    top = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    batch = models.load_data_to_gpu(bench.load_sequence(0, 20, 160000))
    model = build_model(save=False)
    cpu = {}
    orig_enter, orig_exit = timer.Timer.__enter__, timer.Timer.__exit__

    def enter(self):
        self._c0 = time.thread_time()
        return orig_enter(self)

    def exit_(self, *exc):
        cpu[self.name] = cpu.get(self.name, 0.0) + time.thread_time() - self._c0
        return orig_exit(self, *exc)

    timer.Timer.__enter__, timer.Timer.__exit__ = enter, exit_
    with torch.no_grad():
        for _ in range(2):
            model(batch)
        torch.cuda.synchronize()
        timer.RECORDS.clear()
        cpu.clear()
        reps = 5
        w0, c0 = time.perf_counter(), time.thread_time()
        for _ in range(reps):
            model(batch)
        torch.cuda.synchronize()
        wall, cput = (time.perf_counter() - w0) / reps, (time.thread_time() - c0) / reps
    print(f"one sequence alone: wall {wall * 1e3:.1f} ms, CPU time of the calling thread {cput * 1e3:.1f} ms")
    for k, v in sorted(timer.RECORDS.items(), key=lambda kv: -kv[1]):
        if k.startswith("Tracking Frame"):
            continue
        print(f"  {k:55s} wall {v / reps * 1e3:8.2f} ms   cpu {cpu.get(k, 0.0) / reps * 1e3:8.2f} ms")
    pr = cProfile.Profile()
    with torch.no_grad():
        pr.enable()
        model(batch)
        torch.cuda.synchronize()
        pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(top)


if __name__ == "__main__":
    main()
