#!/bin/bash
# Round 6: ProjectedALS without the per-iteration memset in front of the factorisation, the pack as 16 blocks under the product, XH' on
# the transposed images (A/B against NMFX_PROJALS_XT=0).
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06s"; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_utils.py tests/test_gpu_projals_alspgrad.py tests/test_golden.py tests/test_gpu_c4_c5.py tests/test_frontend.py -x -q -m gpu -k "projals or pdsolve or pdrsolve or adddiag or posdef or c4 or rsvd or nndsvd" > "$O/pytest_a.log" 2>&1
tail -4 "$O/pytest_a.log"
B="python bench.py --no-cpu-baseline --alg projals --steps 30 --warmup 10"
: > "$O/projals.jsonl"
for rep in 1 2 3; do
  $B --no-events >> "$O/projals.jsonl" 2>> "$O/err.log"
  NMFX_PROJALS_XT=0 $B --no-events >> "$O/projals.jsonl" 2>> "$O/err.log"
done
NMFX_CHOL_UNROLLED=0 $B --no-events >> "$O/projals.jsonl" 2>> "$O/err.log"
$B --p 8192 --n 16384 --no-events >> "$O/projals.jsonl" 2>> "$O/err.log"
NMFX_CHOL_UNDER_US=0 $B --p 8192 --n 16384 --no-events >> "$O/projals.jsonl" 2>> "$O/err.log"
$B --all-events > "$O/projals_all_events.json" 2>> "$O/err.log"
NMFX_PROJALS_XT=0 $B --all-events > "$O/projals_all_events_row_contiguous_xht.json" 2>> "$O/err.log"
python - <<'PY'
import json
for l in open('gpurun_out/r06s/projals.jsonl'):
    d=json.loads(l); print(d['config'].get('workload')[:40], d['ms_per_step'])
for f in ('projals_all_events','projals_all_events_row_contiguous_xht'):
    d=json.load(open('gpurun_out/r06s/%s.json'%f))
    print(f, d['ms_per_step'], [(k['name'],round(k['avg_us'],1)) for k in d['kernels']])
PY
