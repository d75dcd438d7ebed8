"""`ClusterTracking.plan_units` (host scheduling of the (component key, anchor) tracking units onto
the worker streams): every unit exactly once, never worse than greedy LPT, optimal on small cases."""
import itertools

import numpy as np

from pcseqlearning_amd.pcdet.models.registration.preprocessors.cluster_tracking import ClusterTracking


def makespan(plan, cost):
    return max((sum(cost[u] for u in p) for p in plan), default=0.0)


def lpt(units, cost, n):
    load = [0.0] * n
    for u in sorted(units, key=lambda u: -cost[u]):
        load[load.index(min(load))] += cost[u]
    return max(load)


def brute(units, cost, n):
    best = float("inf")
    for assign in itertools.product(range(n), repeat=len(units)):
        load = [0.0] * n
        for u, w in zip(units, assign):
            load[w] += cost[u]
        best = min(best, max(load))
    return best


def test_plan_units_properties():
    rng = np.random.default_rng(0)
    for trial in range(40):
        n_units, n_workers = int(rng.integers(1, 10)), int(rng.integers(1, 5))
        units = [(f"key{i % 3}", 8 * (i // 3)) for i in range(n_units)]
        cost = {u: float(rng.uniform(0.02, 0.12)) for u in units}
        plan = ClusterTracking.plan_units(units, cost, n_workers)
        assert len(plan) == n_workers
        flat = [u for p in plan for u in p]
        assert sorted(flat) == sorted(units)                       # every unit exactly once
        for p in plan:                                             # dearest first on every worker
            assert [cost[u] for u in p] == sorted((cost[u] for u in p), reverse=True)
        ms = makespan(plan, cost)
        assert ms <= lpt(units, cost, n_workers) + 1e-12           # never worse than greedy LPT
        if n_units <= 7:
            assert ms <= brute(units, cost, n_workers) * 1.12 + 1e-12  # near-optimal on small cases


def test_plan_units_the_shipped_case():
    """9 units of the every-8 config on 4 workers with the measured shape of their costs (the anchors
    in the middle of the sequence track both ways and cost about twice as much)."""
    units = [(k, a) for k in ("component_rad1x25", "component_rad0x75", "component_rad0x25") for a in (0, 8, 16)]
    cost = {u: {0: 0.05, 8: 0.09, 16: 0.05}[u[1]] * (1.3 if u[0].endswith("0x25") else 1.0) for u in units}
    plan = ClusterTracking.plan_units(units, cost, 4)
    total = sum(cost.values())
    assert makespan(plan, cost) <= 1.1 * max(total / 4, max(cost.values())) + 0.02
