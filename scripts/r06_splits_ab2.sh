#!/bin/bash
# Round 6: MultUpdate-MSE with the big products unsplit AND the Grams by their own launches (instead of as tail pieces of the products).
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06s2"; mkdir -p "$O"; cd "$R"
: > "$O/lines.jsonl"
B="python bench.py --no-cpu-baseline --alg multmse --steps 20 --warmup 5"
for rep in 1 2 3; do
  $B --no-events >> "$O/lines.jsonl" 2>> "$O/err.log"
  NMFX_BIG_SPLITS=1 $B --no-events >> "$O/lines.jsonl" 2>> "$O/err.log"
  NMFX_BIG_SPLITS=1 NMFX_FUSE_GRAM=0 $B --no-events >> "$O/lines.jsonl" 2>> "$O/err.log"
  NMFX_FUSE_GRAM=0 $B --no-events >> "$O/lines.jsonl" 2>> "$O/err.log"
done
NMFX_BIG_SPLITS=1 NMFX_FUSE_GRAM=0 $B --all-events > "$O/multmse_unsplit_sep_all_events.json" 2>> "$O/err.log"
python - <<'PY'
import json
for i,l in enumerate(open('gpurun_out/r06s2/lines.jsonl')):
    d=json.loads(l); print(('split+fused','unsplit+fused','unsplit+separate','split+separate')[i%4], d['ms_per_step'])
d=json.load(open('gpurun_out/r06s2/multmse_unsplit_sep_all_events.json'))
print(d['ms_per_step'], d.get('ms_per_step_no_events'))
for v in d['kernels']: print('  ', v['name'], v['avg_us'])
PY
tail -5 "$O/err.log"
