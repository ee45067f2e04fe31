"""bench.py's one-line JSON contract (the driver parses it), checked on the GPU box: single process, and the N = 2 control
flow with bench.py starting its own ranks (both ranks on cuda:0, exchange over gloo: rehearsal only)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "strong"}


def _run(*args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900,
                       env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_single_gpu():
    d = _run("--steps", "10", "--warmup", "3", "--cpu-sample", "65536")
    assert REQUIRED <= set(d) and "cpu_baseline" in d
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["warmup"] == 3 and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    roof = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(roof) and roof["bound"] == "hbm"
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12 and roof["kernel_ms"] <= d["ms_per_step"]
    assert abs(d["value"] - d["config"]["configs_per_gpu"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    cpu = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cpu) and cpu["kind"] in ("port", "reference")
    assert 0.55 < d["config"]["valid_fraction"] < 0.70
    assert d["value"] > 1e8  # the north-star floor, by a wide margin
    two = d["two_streams"]  # informational leg: the same steps on two alternating streams, never `value`
    assert two["same_words_as_the_timed_run"] is True and two["value"] > 0 and two["unit"] == d["unit"]
    probe = d["shard_probe"]["shards"]  # the N = 2, 4, 8 shards of the 1M job timed on this GPU
    assert set(probe) == {"2", "4", "8"} and probe["8"]["configs"] == d["config"]["configs_per_gpu"] // 8
    assert all(p["ms_per_step"] > 0 and 0.5 < p["ceiling"] <= int(k) * 1.5 for k, p in probe.items())
    short = _run("--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-two-streams", "--no-shard-probe")
    assert "two_streams" not in short and "shard_probe" not in short


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_and_reports_weak_and_strong():
    d = _run("--gpus", "2", "--rehearse-on-one-gpu", "--steps", "6", "--warmup", "2")
    assert REQUIRED <= set(d) and d["n_gpus"] == 2 and "cpu_baseline" not in d and "two_streams" not in d
    assert d["strong"]["configs_per_job"] == d["config"]["configs_per_gpu"] and d["strong"]["value"] > 0
    assert abs(d["value"] - 2 * d["config"]["configs_per_gpu"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
