#!/bin/bash
export NMFX_DEV=1   # the NMFX_* development switches below are honoured only with it (csrc/comm.hpp)
# Round-3 evidence run (GPU box): rocprofv3 summaries of the headline and the multdiv command, one bench line per config,
# the per-launch event tables and the simulated-rank timings.  Everything lands under gpurun_out/r03p/.
R="$(cd "$(dirname "$0")/.." && pwd)"
O="$R/gpurun_out/r03p"; mkdir -p "$O"
cd "$R"
bash scripts/profile_bench.sh r03p/prof_multmse > "$O/prof_multmse.log" 2>&1
BENCH_ARGS="--alg multdiv --steps 30 --warmup 10" bash scripts/profile_bench.sh r03p/prof_multdiv > "$O/prof_multdiv.log" 2>&1
bash scripts/bench_configs.sh > "$O/bench_configs.jsonl" 2> "$O/bench_configs.err"
python bench.py --no-cpu-baseline --p 4096 --n 4096 --k 64 --steps 200 --warmup 50 --check-every 4 > "$O/c2_check_every_4.json" 2>/dev/null
python bench.py --no-cpu-baseline --p 4096 --n 4096 --k 64 --steps 200 --warmup 50 --all-events > "$O/c2_all_events.json" 2>/dev/null
python bench.py --no-cpu-baseline --alg multdiv --steps 30 --warmup 10 --all-events > "$O/multdiv_all_events.json" 2>/dev/null
python bench.py --no-cpu-baseline --alg projals --steps 30 --warmup 10 --all-events > "$O/projals_all_events.json" 2>/dev/null
python bench.py --no-cpu-baseline --alg projals --steps 30 --warmup 10 --no-events > "$O/projals_no_events.json" 2>/dev/null
python bench.py --no-cpu-baseline --alg greedycd --steps 20 --warmup 10 --all-events > "$O/greedycd_all_events.json" 2>/dev/null
: > "$O/simranks.jsonl"
for g in 2 4 8; do for m in row_sharded pipelined; do
  python bench.py --sim-ranks $g --steps 50 --no-cpu-baseline --no-events --comm-mode $m >> "$O/simranks.jsonl" 2>/dev/null
done; done
python bench.py --sim-ranks 8 --steps 50 --no-cpu-baseline --all-events > "$O/simranks8_all_events.json" 2>/dev/null
NMFX_RS_FUSED=0 python bench.py --sim-ranks 8 --steps 50 --no-cpu-baseline --no-events > "$O/simranks8_unfused.json" 2>/dev/null
python bench.py --sim-ranks 8 --alg projals --p 16384 --n 131072 --k 256 --steps 10 --warmup 3 --no-cpu-baseline --no-events > "$O/simranks8_projals_c4.json" 2>/dev/null
python bench.py --sim-ranks 8 --alg alspgrad --dtype f64 --p 32768 --n 32768 --k 512 --steps 2 --warmup 1 --no-cpu-baseline > "$O/simranks8_alspgrad_c5.json" 2>/dev/null
ls -la "$O"
