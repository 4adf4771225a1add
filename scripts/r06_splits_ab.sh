#!/bin/bash
# Round 6: the two big products UNSPLIT (one 128 x 128 block per CU) against the 2-way split (two per CU) on the headline and the other algorithms.
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06s"; mkdir -p "$O"; cd "$R"
: > "$O/lines.jsonl"
for alg in multmse multdiv cd; do
  B="python bench.py --no-cpu-baseline --alg $alg --steps 20 --warmup 5 --no-events"
  for rep in 1 2; do
    $B >> "$O/lines.jsonl" 2>> "$O/err.log"
    NMFX_BIG_SPLITS=1 $B >> "$O/lines.jsonl" 2>> "$O/err.log"
  done
done
NMFX_BIG_SPLITS=1 python bench.py --no-cpu-baseline --alg multmse --steps 20 --warmup 5 --all-events > "$O/multmse_unsplit_all_events.json" 2>> "$O/err.log"
python - <<'PY'
import json
for i,l in enumerate(open('gpurun_out/r06s/lines.jsonl')):
    d=json.loads(l); print(('split','unsplit')[i%2], d['metric'], d['ms_per_step'])
d=json.load(open('gpurun_out/r06s/multmse_unsplit_all_events.json'))
print(d['ms_per_step'], d.get('ms_per_step_no_events'))
for v in d['kernels']: print('  ', v['name'], v['avg_us'])
PY
tail -5 "$O/err.log"
