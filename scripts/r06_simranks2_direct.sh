#!/bin/bash
# Round 6: the row-sharded fused step at 2 (and 4) simulated ranks with X_g H_g' unsplit into the blocked send buffer also at the long local
# contraction of the 2-rank shard (256 k-tiles; until now only up to 128) against the 2-way split + combine.
export NMFX_DEV=1 TMPDIR=/tmp
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06d"; mkdir -p "$O"; cd "$R"
: > "$O/lines.jsonl"
for g in 2 4; do for tr in rccl p2p; do for rep in 1 2; do
  python bench.py --no-cpu-baseline --sim-ranks $g --steps 50 --no-events --transport $tr >> "$O/lines.jsonl" 2>/dev/null
  NMFX_DIRECT_MAX_KTILES=100000 python bench.py --no-cpu-baseline --sim-ranks $g --steps 50 --no-events --transport $tr >> "$O/lines.jsonl" 2>/dev/null
done; done; done
python - <<'PY'
import json
for i,l in enumerate(open('gpurun_out/r06d/lines.jsonl')):
    d=json.loads(l); print(d.get('sim_ranks'), d['config']['parallelism'], ('split','direct')[i%2], d['ms_per_step'], d.get('objvalue'))
PY
