"""GPU: bench.py prints ONE JSON line that carries the driver's contract keys plus `roofline` and `cpu_baseline`,
and __graft_entry__.smoke() passes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_smoke_entry():
    import __graft_entry__ as ge
    ge.smoke()


def test_bench_json_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--config", "C2"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "f64"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    # "mfma" = fp64 matrix cores, "valu_fp64" = fp64 vector unit (same 78.6 TFLOP/s peak): the bound names the unit the
    # dominant kernel really computes on (round-1 review: a VALU kernel must not be labelled "mfma")
    assert r["bound"] in ("hbm", "mfma", "valu_fp64") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["bound"] == ("mfma" if d["config"]["engine"].get("matrix_core_sweep") else "valu_fp64")
    assert "N=16 nt=500 K=32" in d["metric"]          # the metric string follows --config
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1
    assert d["value"] > 0 and d["J_T"]["after"] < d["J_T"]["before"]


def _bench(args, ranks=1, env_extra=None):
    env = dict(os.environ, **(env_extra or {}))
    if ranks > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
               "--master-port", str(29600 + os.getpid() % 300), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks)] + args
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_two_ranks_rehearsal_strong_scaling_default():
    """bench.py's N > 1 path (torch.distributed launch, IPC mailbox wiring, max-over-ranks timing) with two ranks sharing the one
    GPU (gloo instead of RCCL; each persistent kernel limited to 100 workgroups so that both are resident).  DEFAULT = strong
    scaling: the config's ensemble is block-partitioned, the metric string does not change with N."""
    d = _bench(["--steps", "2", "--warmup", "1", "--config", "C2", "--no-cpu-baseline"], ranks=2,
               env_extra={"KROTOV_BENCH_BACKEND": "gloo", "KROTOV_HIP_MAX_WG": "100"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert "N=16 nt=500 K=32" in d["metric"]
    assert d["config"]["K_global"] == 32 and d["config"]["K_per_gpu"] == 16
    assert "over 2 GPU" in d["config"]["parallelism"] and "mailboxes" in d["config"]["exchange"]
    assert d["J_T"]["after"] < d["J_T"]["before"]


def test_bench_headline_metric_is_baselines_at_every_n():
    """BASELINE.json's metric names a FIXED ensemble (N=64, nt=1000, K=256): `--gpus 2` must shard those 256 trajectories (128
    per rank), print the identical metric string, and -- same problem, same arithmetic up to the order of the cross-rank sum --
    reach the same J_T as the one-GPU run (/root/reference/src/optimize.jl:182,303,321,339-349,360)."""
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        metric = json.load(f)["metric"]
    one = _bench(["--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    two = _bench(["--steps", "2", "--warmup", "1", "--no-cpu-baseline"], ranks=2,
                 env_extra={"KROTOV_BENCH_BACKEND": "gloo", "KROTOV_HIP_MAX_WG": "100"})
    assert one["metric"] == metric and two["metric"] == metric
    assert one["scaling"] == "strong" and two["scaling"] == "strong"
    assert two["config"]["K_global"] == 256 and two["config"]["K_per_gpu"] == 128 and one["config"]["K_per_gpu"] == 256
    assert abs(one["J_T"]["before"] - two["J_T"]["before"]) < 1e-12
    assert abs(one["J_T"]["after"] - two["J_T"]["after"]) < 1e-10
    assert two["J_T"]["after"] < two["J_T"]["before"]


def test_bench_weak_scaling_is_opt_in():
    d = _bench(["--steps", "2", "--warmup", "1", "--traj", "16", "--config", "C2", "--no-cpu-baseline", "--scaling", "weak"], ranks=2,
               env_extra={"KROTOV_BENCH_BACKEND": "gloo", "KROTOV_HIP_MAX_WG": "100"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["K_global"] == 32 and d["config"]["K_per_gpu"] == 16
    assert "K=32" in d["metric"]
