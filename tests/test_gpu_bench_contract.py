"""bench.py's output contract on a small instance of the default workload: exactly one JSON line on stdout with the fields the driver
reads, a roofline object measured inside the run and a CPU baseline timed beside it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def test_bench_prints_one_json_line_with_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--bs", "2", "--size", "128",
                        "--cpu-steps", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                   ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[k], typ), (k, d[k])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 2 * 1000.0 / d["ms_per_step"]) <= 1e-6 * d["value"]            # images/s = bs / step time
    assert d["launch_mode"] in ("eager", "replay", "hipgraph")
    roof = d["roofline"]
    assert roof["bound"] in ("mfma", "hbm") and roof["unit"] in ("TFLOP/s", "GB/s") and roof["peak"] > 0
    assert roof["achieved"] > 0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert roof["launches"] > 0 and roof["avg_launch_ms"] > 0 and sum(roof["by_kernel_ms_per_step"].values()) > 0
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["unit"] == d["unit"] and cpu["sample"]
    # the headline family is the one with the largest time of this run, whichever it is
    assert roof["by_kernel_ms_per_step"][roof["kernel"]] >= max(roof["by_kernel_ms_per_step"].values()) - 1e-3      # (rounded figures)
    # the bounded f32 (parity-mode) leg of the same workload rides in the same line
    par = d["parity_mode"]
    assert par["dtype"] == "f32" and par["value"] > 0 and par["ms_per_step"] > 0 and "error" not in par
    assert abs(par["value"] - 2 * 1000.0 / par["ms_per_step"]) <= 1e-6 * par["value"]


def test_plain_launch_with_gpus_2_starts_two_ranks():
    """`python bench.py --gpus 2` without a launcher around it must start its own two ranks (here both on cuda:0 over gloo — the
    one-GPU rehearsal; on a multi-GPU node the same path runs RCCL) and print ONE line with n_gpus 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-gpu", "--backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--bs", "2", "--size", "128", "--no-roofline", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2" and d["value"] > 0
