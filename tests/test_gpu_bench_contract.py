"""bench.py prints ONE JSON line with the contract's fields (task statement section 4): run here on a small shape so that the
whole thing, CPU baseline included, takes seconds."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--alg", "projals"], ["--alg", "multdiv"]])
def test_bench_json_contract(built, extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--p", "2048", "--n", "2048", "--k", "128", "--steps", "8", "--warmup", "2",
           "--cpu-sample-cols", "256"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                     ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict),
                     ("cpu_baseline", dict)):
        assert key in d and isinstance(d[key], typ), key
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32" and "workload" in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) <= 1e-3 * d["value"]                  # iterations per second of the whole job
    r = d["roofline"]
    assert r["bound"] in ("mfma", "hbm") and r["unit"] in ("TFLOP/s", "GB/s") and 0 < r["achieved"] <= r["peak"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str) and c["unit"] == d["unit"]
    assert d["ms_per_step_no_events"] > 0 and isinstance(d["event_brackets"], str)        # the same steps without hipEvent brackets
    if not extra:   # multmse: the CPU baseline states where its time goes (every mul!, every loop) and what its GEMM phases reach alone
        assert len(c["phase_seconds"]) == 12 and c["gemm_phases_gflops"] > 0 and c["gemm_seconds"] + c["non_gemm_seconds"] > 0


def test_bench_help_runs_without_a_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout
