#!/bin/bash
# Round 6: CoordinateDescent / GreedyCD with X*H' on the transposed images (A/B against NMFX_XT=0), the new division test.
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06k"; mkdir -p "$O"; cd "$R"
timeout 1200 python -m pytest tests/test_gpu_cd.py tests/test_gpu_multupd.py tests/test_golden.py tests/test_frontend.py -x -q -m gpu > "$O/pytest_a.log" 2>&1
tail -4 "$O/pytest_a.log"
B="python bench.py --no-cpu-baseline --no-events"
: > "$O/lines.jsonl"
for rep in 1 2; do
  for alg in cd greedycd; do
    $B --alg $alg --steps 20 --warmup 10 >> "$O/lines.jsonl" 2>> "$O/err.log"
    NMFX_XT=0 $B --alg $alg --steps 20 --warmup 10 >> "$O/lines.jsonl" 2>> "$O/err.log"
  done
done
python bench.py --no-cpu-baseline --alg greedycd --steps 20 --warmup 10 --all-events > "$O/greedycd_all_events.json" 2>> "$O/err.log"
python bench.py --no-cpu-baseline --alg cd --steps 20 --warmup 10 --all-events > "$O/cd_all_events.json" 2>> "$O/err.log"
python - <<'PY'
import json
for l in open('gpurun_out/r06k/lines.jsonl'):
    d=json.loads(l); print(d['config'].get('workload')[:60], d['ms_per_step'])
for f in ('greedycd_all_events','cd_all_events'):
    d=json.load(open('gpurun_out/r06k/%s.json'%f))
    print(f, d['ms_per_step'], [(k['name'],round(k['avg_us'],1)) for k in d['kernels']])
PY
tail -3 "$O/err.log" | grep -v amdgpu.ids
