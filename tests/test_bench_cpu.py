"""CPU: the host-side pieces of bench.py that need no GPU -- the metric string is BASELINE.json's, the strong-scaling partition
covers the named ensemble exactly once, and the LAPACK-backed Pade expm of the `cpu_baseline` leg (what Julia's exp(::Matrix)
does per prop_step! [UPSTREAM-RECALLED]) agrees with the oracle's and SciPy's."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_metric_string_is_baseline_json():
    import bench
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        assert bench.METRIC_FMT % (64, 1000, 256) == json.load(f)["metric"]


def test_strong_scaling_partition_covers_the_ensemble_once():
    from krotov_jl_amd.distributed import shard_range
    for K in (256, 1024, 64, 37):
        for world in (1, 2, 4, 8):
            parts = [shard_range(K, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == K
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == K


def test_expm_lapack_matches_oracle_and_scipy(oracle_c):
    import bench
    from scipy.linalg import expm
    rng = np.random.default_rng(5)
    for n, scale in ((2, 0.01), (5, 0.2), (16, 0.9), (16, 2.0), (64, 0.07), (33, 40.0)):
        A = (rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))) * scale / np.sqrt(n)
        E = bench.expm_lapack(A)
        ref = expm(A)
        assert np.max(np.abs(E - ref)) <= 1e-13 * max(1.0, np.max(np.abs(ref)))
        assert np.max(np.abs(E - oracle_c.expm(A))) <= 1e-13 * max(1.0, np.max(np.abs(ref)))


def test_cpu_reference_order_variant_runs_and_updates_the_pulse():
    import bench
    from krotov_jl_amd import workloads as wl
    p = wl.dense_problem(6, 12, 3, 2, 77, 0.1, 1.0)
    out = bench.cpu_reference_order_lapack(p, 3, 12, lambda m: None)
    assert out["value"] > 0 and out["cores"] == 1 and out["K"] == 3 and out["NT"] == 12
