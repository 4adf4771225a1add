#!/bin/bash
# Round 6: two-launch stop statistics + stop rule of the one-GPU MultUpdate-MSE step, A/B against the four launches.
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06j"; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_multupd.py tests/test_gpu_track_stop.py -x -q -m gpu > "$O/pytest_b.log" 2>&1
tail -3 "$O/pytest_b.log"
B="python bench.py --no-cpu-baseline --no-events"
: > "$O/lines_b.jsonl"
for rep in 1 2 3; do
  $B --steps 50 --warmup 10 >> "$O/lines_b.jsonl" 2>> "$O/err.log"
  NMFX_STATS_FUSED=0 $B --steps 50 --warmup 10 >> "$O/lines_b.jsonl" 2>> "$O/err.log"
done
python bench.py --no-cpu-baseline --steps 50 --warmup 10 --all-events > "$O/multmse_all_events.json" 2>> "$O/err.log"
python - <<'PY'
import json
for l in open('gpurun_out/r06j/lines_b.jsonl'):
    d=json.loads(l); print(d['config'].get('workload')[:60], d['ms_per_step'])
d=json.load(open('gpurun_out/r06j/multmse_all_events.json'))
print(d['ms_per_step'], [(k['name'],round(k['avg_us'],1)) for k in d['kernels']])
PY
