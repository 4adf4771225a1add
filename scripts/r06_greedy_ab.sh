#!/bin/bash
# Round 6: GreedyCD sweep with 6 resident workgroups per CU (default) against 8 and 5, on the bench's standard run.
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06x"; mkdir -p "$O"; cd "$R"
timeout 600 python -m pytest tests/test_gpu_cd.py -x -q -m gpu -k greedy > "$O/pytest_a.log" 2>&1; tail -2 "$O/pytest_a.log"
B="python bench.py --no-cpu-baseline --alg greedycd --steps 20 --warmup 10 --no-events"
: > "$O/lines.jsonl"
for rep in 1 2; do
  $B >> "$O/lines.jsonl" 2>> "$O/err.log"
  NMFX_GREEDY_WGS_PER_CU=8 $B >> "$O/lines.jsonl" 2>> "$O/err.log"
  NMFX_GREEDY_WGS_PER_CU=5 $B >> "$O/lines.jsonl" 2>> "$O/err.log"
done
python - <<'PY'
import json
for i,l in enumerate(open('gpurun_out/r06x/lines.jsonl')):
    d=json.loads(l); print(('default 6','8','5')[i%3], d['ms_per_step'])
PY
