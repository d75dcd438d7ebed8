"""The tracker's schedule must be robust, not lucky: the engines' chains land on hardware queues of their own
whatever streams the embedding application created first (round 3 found 1506 against 1750 ms per batch
depending on the creation order; pcs_track_engines_init now creates the engines' streams back to back, so any
three consecutive streams map to three different queues of the runtime's four)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "4", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-stages"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_throughput_does_not_depend_on_streams_created_before():
    base = _bench([])
    assert base["host_fallbacks"] == 0
    for n in (1, 4):
        other = _bench(["--pre-streams", str(n)])
        assert other["config"]["pre_created_streams"] == n
        # same work, same schedule: the median step within 10 % (the verdict's bench criterion is 3 % on the
        # full batch; a batch of four on a shared test box gets more slack)
        a, b = base["step_ms"]["median"], other["step_ms"]["median"]
        assert abs(a - b) <= 0.10 * min(a, b), (n, a, b)
