#!/bin/bash
export NMFX_DEV=1   # the NMFX_* development switches below are honoured only with it (csrc/comm.hpp)
# Round-6 evidence run (GPU box): rocprofv3 summaries (kernel stats + FETCH / WRITE / SQ counter passes) of the headline command and of
# the multdiv, GreedyCD and ProjectedALS lines, the driver-style line, one bench line per BASELINE config (each with its cpu_baseline),
# the ProjectedALS lines the round-5 verdict asked for (4096^2 with the factorisation on the critical path, C4's 8-rank shard), the
# simulated-rank timings of the headline step on both launch sequences.  Everything lands under gpurun_out/r06p/.
R="$(cd "$(dirname "$0")/.." && pwd)"
O="$R/gpurun_out/r06p"; mkdir -p "$O"
cd "$R"
export GPU_MAX_HW_QUEUES=24
bash scripts/profile_bench.sh r06p/prof_multmse > "$O/prof_multmse.log" 2>&1
BENCH_ARGS="--alg multdiv --steps 30 --warmup 10" bash scripts/profile_bench.sh r06p/prof_multdiv > "$O/prof_multdiv.log" 2>&1
BENCH_ARGS="--alg greedycd --steps 20 --warmup 10" bash scripts/profile_bench.sh r06p/prof_greedycd > "$O/prof_greedycd.log" 2>&1
BENCH_ARGS="--alg projals --steps 30 --warmup 10" bash scripts/profile_bench.sh r06p/prof_projals > "$O/prof_projals.log" 2>&1
python bench.py --steps 20 --warmup 5 > "$O/driver_20_steps.json" 2>/dev/null
python bench.py --steps 20 --warmup 5 --prewarm-ms 0 --no-cpu-baseline > "$O/driver_20_steps_without_prewarm.json" 2>/dev/null
bash scripts/bench_configs.sh > "$O/bench_configs.jsonl" 2> "$O/bench_configs.err"
B="python bench.py --no-cpu-baseline"
$B --p 4096 --n 4096 --k 64 --steps 500 --warmup 50 --no-events > "$O/c2_no_events.json" 2>/dev/null
: > "$O/projals_lines.jsonl"
$B --alg projals --steps 30 --warmup 10 --no-events >> "$O/projals_lines.jsonl" 2>/dev/null
$B --alg projals --p 4096 --n 4096 --steps 50 --warmup 10 --no-events >> "$O/projals_lines.jsonl" 2>/dev/null
NMFX_POTRF_REG=0 NMFX_CHOL_UNDER_US=0 $B --alg projals --p 4096 --n 4096 --steps 50 --warmup 10 --no-events >> "$O/projals_lines.jsonl" 2>/dev/null
$B --alg projals --p 8192 --n 8192 --steps 50 --warmup 10 --no-events >> "$O/projals_lines.jsonl" 2>/dev/null
$B --sim-ranks 8 --alg projals --p 16384 --n 131072 --k 256 --steps 10 --warmup 3 --no-events --transport rccl >> "$O/projals_lines.jsonl" 2>/dev/null
: > "$O/simranks.jsonl"
for g in 2 4 8; do for tr in rccl p2p; do
  $B --sim-ranks $g --steps 50 --no-events --transport $tr >> "$O/simranks.jsonl" 2>/dev/null
done; done
$B --sim-ranks 8 --alg alspgrad --dtype f64 --p 32768 --n 32768 --k 512 --steps 2 --warmup 1 --transport rccl > "$O/simranks8_alspgrad_c5.json" 2>/dev/null
NMFX_BENCH_BACKEND=gloo-p2p python bench.py --gpus 4 --steps 20 --warmup 5 --p 8192 --n 8192 --no-cpu-baseline > "$O/bench_gpus4_self_launched_one_gpu.json" 2> "$O/bench_gpus4.err"
ls -la "$O"
python - <<'PY'
import json
for f in ("bench_configs.jsonl", "projals_lines.jsonl", "simranks.jsonl"):
    print(f)
    for l in open("gpurun_out/r06p/" + f):
        try: d = json.loads(l)
        except Exception: continue
        print("  ", d["ms_per_step"], d.get("frac_of_mfma_peak"), d.get("sim_ranks"), d["config"].get("workload", "")[:70], d["config"].get("parallelism"))
PY
