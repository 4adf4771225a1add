#!/bin/bash
# Round 6: GreedyCD's step division from Float32 operations (greedy_div_fma): parity tests, lines, per-launch view.
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06v"; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_cd.py tests/test_golden.py tests/test_frontend.py tests/test_gpu_fullsize.py -x -q -m gpu -k "greedy or golden or nnmf or frontend" > "$O/pytest_a.log" 2>&1
tail -3 "$O/pytest_a.log"
B="python bench.py --no-cpu-baseline --alg greedycd --steps 20 --warmup 10"
: > "$O/lines.jsonl"
for rep in 1 2 3; do $B --no-events >> "$O/lines.jsonl" 2>> "$O/err.log"; done
$B --all-events > "$O/greedycd_all_events.json" 2>> "$O/err.log"
python - <<'PY'
import json
for l in open('gpurun_out/r06v/lines.jsonl'):
    d=json.loads(l); print(d['ms_per_step'], d.get('inner_iters_per_step'))
d=json.load(open('gpurun_out/r06v/greedycd_all_events.json'))
print(d['ms_per_step'], [(k['name'],round(k['avg_us'],1)) for k in d['kernels']])
PY
