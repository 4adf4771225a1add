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


@pytest.mark.gpu
def test_bench_gpus_2_launches_its_own_ranks(built):
    """`python bench.py --gpus 2` typed as it stands (no torch.distributed.run in front, WORLD_SIZE unset) becomes two ranks by itself;
    on this 1-GPU box both sit on device 0 and exchange through the peer windows (NMFX_BENCH_BACKEND=gloo-p2p).  One JSON line, from
    rank 0, with n_gpus = 2 and the ranks holding the same W bits."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "NMFX_BENCH_DEVICE")}
    env["NMFX_BENCH_BACKEND"] = "gloo-p2p"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--p", "2048", "--n", "2048", "--k", "128",
           "--prewarm-ms", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0
    assert d["multi_gpu_consistency"] == {"W_identical_on_all_ranks": True, "objective_identical_on_all_ranks": True, "finite": True}
    assert "cpu_baseline" not in d      # rank 0 at N = 1 only
    # the transport is chosen BY THE CLOCK (round 6): every exchange candidate that passed its verification was also timed before the
    # timed region, the times are in the line, and the region ran on the fastest one
    ev = d["exchange_verification"]
    cands = ev["candidates"]
    assert len(cands) >= 2 and ev["selected_by"].startswith("fastest")
    timed = [c for c in cands if c["ok"]]
    assert timed and all(isinstance(c["ms_per_step"], float) and c["ms_per_step"] > 0 for c in timed)
    assert ev["ran"] == min(timed, key=lambda c: c["ms_per_step"])["candidate"]
    assert all(c["ms_per_step"] is None for c in cands if not c["ok"])


def test_bench_gpus_n_without_a_launcher_spawns_torch_distributed_run(monkeypatch):
    """CPU: the self-launch builds a torch.distributed.run command for N ranks on 127.0.0.1 with the caller's arguments."""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    monkeypatch.setenv("NMFX_BENCH_BACKEND", "gloo-p2p")
    monkeypatch.delenv("NMFX_BENCH_DEVICE", raising=False)
    assert bench.self_launch(4) == 0
    c = seen["cmd"]
    assert c[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in c and "127.0.0.1" in c
    assert c[-1].endswith("bench.py") and json.loads(seen["env"]["NMFX_BENCH_ARGV"]) == ["--gpus", "4", "--steps", "7"]
    assert seen["env"]["NMFX_BENCH_DEVICE"] == "0"       # fewer than 4 devices here: every rank on device 0
