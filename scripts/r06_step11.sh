#!/bin/bash
# Round 6: trtri_offdiag with its operand loads hoisted (88 -> 34 us at k = 256): ProjectedALS lines, the crossover sweep again.
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06u"; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_utils.py tests/test_gpu_projals_alspgrad.py tests/test_golden.py tests/test_frontend.py -x -q -m gpu -k "projals or pdsolve or pdrsolve or adddiag or posdef or rsvd or nndsvd" > "$O/pytest_a.log" 2>&1
tail -3 "$O/pytest_a.log"
B="python bench.py --no-cpu-baseline --alg projals --steps 40 --warmup 10 --no-events"
: > "$O/lines.jsonl"
for rep in 1 2; do $B >> "$O/lines.jsonl" 2>> "$O/err.log"; done
for shape in "4096 4096" "8192 4096" "8192 8192" "12288 8192"; do
  set -- $shape
  NMFX_CHOL_UNDER_US=1000000000 $B --p $1 --n $2 >> "$O/lines.jsonl" 2>> "$O/err.log"
  NMFX_CHOL_UNDER_US=0 $B --p $1 --n $2 >> "$O/lines.jsonl" 2>> "$O/err.log"
done
python bench.py --no-cpu-baseline --alg projals --steps 30 --warmup 10 --all-events > "$O/projals_all_events.json" 2>> "$O/err.log"
python bench.py --no-cpu-baseline --alg projals --p 4096 --n 4096 --steps 50 --warmup 10 --all-events > "$O/projals_4096_all_events.json" 2>> "$O/err.log"
python - <<'PY'
import json
for l in open('gpurun_out/r06u/lines.jsonl'):
    d=json.loads(l); print(d['config']['p'], d['config']['n'], d['ms_per_step'])
for f in ('projals_all_events','projals_4096_all_events'):
    d=json.load(open('gpurun_out/r06u/%s.json'%f))
    print(f, d['ms_per_step'], [(k['name'],round(k['avg_us'],1)) for k in d['kernels']])
PY
