#!/bin/bash
# Round 6: the big products unsplit (one block per CU) where the output is one tile per CU: MultUpdate-MSE, CoordinateDescent, GreedyCD,
# default against NMFX_UNSPLIT=0, then the tests that run those algorithms at that shape.
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06a"; mkdir -p "$O"; cd "$R"
: > "$O/lines.jsonl"
for alg in multmse cd greedycd; do
  B="python bench.py --no-cpu-baseline --alg $alg --steps 20 --warmup 5 --no-events"
  for rep in 1 2 3; do
    NMFX_UNSPLIT=0 $B >> "$O/lines.jsonl" 2>> "$O/err.log"
    $B >> "$O/lines.jsonl" 2>> "$O/err.log"
  done
done
python - <<'PY'
import json
for i,l in enumerate(open('gpurun_out/r06a/lines.jsonl')):
    d=json.loads(l); print(('2-way split','unsplit')[i%2], d['metric'], d['ms_per_step'], d['objvalue'])
PY
python bench.py > "$O/bench_default.json" 2>> "$O/err.log"; cut -c1-300 "$O/bench_default.json"
timeout 1500 python -m pytest tests/test_gpu_multupd.py tests/test_gpu_cd.py tests/test_gpu_fullsize.py tests/test_gpu_bench_contract.py tests/test_golden.py -x -q -m gpu > "$O/pytest.log" 2>&1
tail -2 "$O/pytest.log"; tail -3 "$O/err.log"
