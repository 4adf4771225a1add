#!/bin/bash
# Round 6: one-launch stop statistics + stop rule of the one-GPU MultUpdate-MSE step; ratio pass with the reciprocal + one-residual division.
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06j"; mkdir -p "$O"; cd "$R"
timeout 1200 python -m pytest tests/test_gpu_multupd.py tests/test_golden.py tests/test_gpu_track_stop.py tests/test_gpu_k_granularity.py -x -q -m gpu > "$O/pytest_a.log" 2>&1
tail -5 "$O/pytest_a.log"
B="python bench.py --no-cpu-baseline --no-events"
: > "$O/lines.jsonl"
for rep in 1 2 3; do
  $B --steps 50 --warmup 10 >> "$O/lines.jsonl" 2>> "$O/err.log"
  NMFX_STATS_FUSED=0 $B --steps 50 --warmup 10 >> "$O/lines.jsonl" 2>> "$O/err.log"
done
for rep in 1 2; do
  $B --alg multdiv --steps 30 --warmup 10 >> "$O/lines.jsonl" 2>> "$O/err.log"
  NMFX_DIV_IEEE=1 $B --alg multdiv --steps 30 --warmup 10 >> "$O/lines.jsonl" 2>> "$O/err.log"
done
python bench.py --no-cpu-baseline --alg multdiv --steps 30 --warmup 10 --all-events > "$O/multdiv_all_events.json" 2>> "$O/err.log"
NMFX_DIV_IEEE=1 python bench.py --no-cpu-baseline --alg multdiv --steps 30 --warmup 10 --all-events > "$O/multdiv_ieee_all_events.json" 2>> "$O/err.log"
python bench.py --no-cpu-baseline --steps 50 --warmup 10 --all-events > "$O/multmse_all_events.json" 2>> "$O/err.log"
python - <<'PY'
import json
for l in open('gpurun_out/r06j/lines.jsonl'):
    d=json.loads(l); print(d['config'].get('workload')[:60], d['ms_per_step'])
for f in ('multdiv_all_events','multdiv_ieee_all_events','multmse_all_events'):
    d=json.load(open('gpurun_out/r06j/%s.json'%f))
    print(f, d['ms_per_step'], [(k['name'],round(k['avg_us'],1)) for k in d['kernels']])
PY
tail -3 "$O/err.log" | grep -v amdgpu.ids
