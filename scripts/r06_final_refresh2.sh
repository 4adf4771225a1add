#!/bin/bash
export NMFX_DEV=1
# Round 6, last refresh: ProjectedALS with XH' on the transposed images -- its rocprofv3 view, its bench lines, C4 at full size.
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06g"; mkdir -p "$O"; cd "$R"
export GPU_MAX_HW_QUEUES=24
BENCH_ARGS="--alg projals --steps 30 --warmup 10" bash scripts/profile_bench.sh r06g/prof_projals > "$O/prof_projals.log" 2>&1
B="python bench.py --no-cpu-baseline"
: > "$O/projals_lines.jsonl"
$B --alg projals --steps 30 --warmup 10 --no-events >> "$O/projals_lines.jsonl" 2>/dev/null
$B --alg projals --p 4096 --n 4096 --steps 50 --warmup 10 --no-events >> "$O/projals_lines.jsonl" 2>/dev/null
$B --alg projals --p 8192 --n 8192 --steps 50 --warmup 10 --no-events >> "$O/projals_lines.jsonl" 2>/dev/null
$B --sim-ranks 8 --alg projals --p 16384 --n 131072 --k 256 --steps 10 --warmup 3 --no-events --transport rccl >> "$O/projals_lines.jsonl" 2>/dev/null
$B --alg projals --p 16384 --n 131072 --k 256 --steps 10 --warmup 3 >> "$O/projals_lines.jsonl" 2>/dev/null
python - <<'PY'
import json
for l in open("gpurun_out/r06g/projals_lines.jsonl"):
    d = json.loads(l); print(d["ms_per_step"], d.get("frac_of_mfma_peak"), d.get("sim_ranks"), d["config"]["workload"][:50])
d = json.load(open("gpurun_out/r06g/prof_projals/bench_default.json")); print("default line", d["ms_per_step"], d["ms_per_step_no_events"], d["frac_of_mfma_peak"], d["roofline"]["kernel"], d["roofline"]["frac"])
PY
