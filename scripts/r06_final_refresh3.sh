#!/bin/bash
export NMFX_DEV=1
# Round 6, last refresh after the big products of MultUpdate-MSE / CoordinateDescent went unsplit at the headline shape: the full GPU suite,
# smoke, the rocprofv3 view of the default command, the driver-style lines, the CoordinateDescent line.
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06z"; mkdir -p "$O"; cd "$R"
export TMPDIR=/tmp
python -m pytest tests -q -m gpu > "$O/pytest_full.log" 2>&1; grep -E "passed|failed" "$O/pytest_full.log" | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
export GPU_MAX_HW_QUEUES=24
bash scripts/profile_bench.sh r06z/prof_multmse > "$O/prof_multmse.log" 2>&1
python bench.py > "$O/bench_default.json" 2>/dev/null
python bench.py --steps 20 --warmup 5 > "$O/driver_20_steps.json" 2>/dev/null
python bench.py --steps 20 --warmup 5 --prewarm-ms 0 --no-cpu-baseline > "$O/driver_20_steps_without_prewarm.json" 2>/dev/null
B="python bench.py --no-cpu-baseline"
$B --all-events > "$O/multmse_all_events.json" 2>/dev/null
$B --alg cd > "$O/cd_default.json" 2>/dev/null
$B --alg cd --all-events > "$O/cd_all_events.json" 2>/dev/null
python - <<'PY'
import json
for f in ("bench_default", "driver_20_steps", "driver_20_steps_without_prewarm", "multmse_all_events", "cd_default", "cd_all_events"):
    d = json.load(open("gpurun_out/r06z/%s.json" % f)); r = d.get("roofline") or {}
    print(f, d["ms_per_step"], d.get("ms_per_step_no_events"), d.get("frac_of_mfma_peak"), r.get("kernel"), r.get("frac"), r.get("achieved"), r.get("traffic"))
d = json.load(open("gpurun_out/r06z/multmse_all_events.json"))
for v in d["kernels"]: print("  ", v["name"], v["avg_us"])
PY
head -12 "$O/prof_multmse/summary.md"
