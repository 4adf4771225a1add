#!/bin/bash
export NMFX_DEV=1
# Round 6, end of the round: the evidence that changed after scripts/r06_profiles.sh ran (ProjectedALS without the memset / with the pack
# on the main stream / with the faster trtri): its rocprofv3 view, its lines, one bench line per configuration again, the driver-style line.
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06f"; mkdir -p "$O"; cd "$R"
export GPU_MAX_HW_QUEUES=24
python bench.py --steps 20 --warmup 5 > "$O/driver_20_steps.json" 2>/dev/null
BENCH_ARGS="--alg projals --steps 30 --warmup 10" bash scripts/profile_bench.sh r06f/prof_projals > "$O/prof_projals.log" 2>&1
bash scripts/bench_configs.sh > "$O/bench_configs.jsonl" 2> "$O/bench_configs.err"
B="python bench.py --no-cpu-baseline"
: > "$O/projals_lines.jsonl"
$B --alg projals --steps 30 --warmup 10 --no-events >> "$O/projals_lines.jsonl" 2>/dev/null
$B --alg projals --p 4096 --n 4096 --steps 50 --warmup 10 --no-events >> "$O/projals_lines.jsonl" 2>/dev/null
NMFX_POTRF_REG=0 NMFX_CHOL_UNDER_US=0 $B --alg projals --p 4096 --n 4096 --steps 50 --warmup 10 --no-events >> "$O/projals_lines.jsonl" 2>/dev/null
$B --alg projals --p 8192 --n 8192 --steps 50 --warmup 10 --no-events >> "$O/projals_lines.jsonl" 2>/dev/null
$B --sim-ranks 8 --alg projals --p 16384 --n 131072 --k 256 --steps 10 --warmup 3 --no-events --transport rccl >> "$O/projals_lines.jsonl" 2>/dev/null
python - <<'PY'
import json
for f in ("bench_configs.jsonl", "projals_lines.jsonl"):
    print(f)
    for l in open("gpurun_out/r06f/" + f):
        try: d = json.loads(l)
        except Exception: continue
        print("  ", d["ms_per_step"], d.get("ms_per_step_no_events"), d.get("frac_of_mfma_peak"), d.get("sim_ranks"), d["config"].get("workload", "")[:60])
d = json.load(open("gpurun_out/r06f/driver_20_steps.json")); print("driver", d["ms_per_step"], d["roofline"]["frac"])
PY
