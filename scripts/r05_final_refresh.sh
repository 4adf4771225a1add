#!/bin/bash
# Round 5, second half: the evidence that changed with the strip-kernel H solve of ProjectedALS and the backwards-walking split-K combine --
# one bench line per configuration (each with its cpu_baseline), the driver-style headline line, the rocprofv3 view of the ProjectedALS line.
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r05f"; mkdir -p "$O"; cd "$R"
export GPU_MAX_HW_QUEUES=24
python bench.py --steps 20 --warmup 5 > "$O/driver_20_steps.json" 2>/dev/null
bash scripts/bench_configs.sh > "$O/bench_configs.jsonl" 2> "$O/bench_configs.err"
python bench.py --no-cpu-baseline --alg projals --steps 30 --warmup 10 --all-events --traffic none > "$O/projals_all_events.json" 2>/dev/null
python bench.py --no-cpu-baseline --sim-ranks 8 --alg projals --p 16384 --n 131072 --k 256 --steps 10 --warmup 3 --no-events --transport rccl > "$O/simranks8_projals_c4.json" 2>/dev/null
BENCH_ARGS="--alg projals --steps 30 --warmup 10" bash scripts/profile_bench.sh r05f/prof_projals > "$O/prof_projals.log" 2>&1
ls -la "$O"
python - <<'PY'
import json
for l in open("gpurun_out/r05f/bench_configs.jsonl"):
    try: d=json.loads(l)
    except Exception: continue
    print(d["ms_per_step"], d.get("frac_of_mfma_peak"), d["config"].get("workload","")[:60])
PY
