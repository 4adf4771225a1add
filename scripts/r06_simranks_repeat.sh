#!/bin/bash
# Round 6: run-to-run spread of the simulated 8-rank step on both launch sequences (the peer sequence's stand-in stores are uncached).
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06q"; mkdir -p "$O"; cd "$R"
B="python bench.py --no-cpu-baseline --sim-ranks 8 --steps 50 --warmup 10 --no-events"
: > "$O/simranks8_repeat.jsonl"
for i in 1 2 3 4; do
  $B --transport p2p >> "$O/simranks8_repeat.jsonl" 2>> "$O/err.log"
  $B --transport rccl >> "$O/simranks8_repeat.jsonl" 2>> "$O/err.log"
done
$B --transport p2p --all-events > "$O/simranks8_p2p_all_events.json" 2>> "$O/err.log"
python bench.py --no-cpu-baseline --alg projals --p 4096 --n 4096 --steps 50 --warmup 10 --all-events > "$O/projals_4096_all_events.json" 2>> "$O/err.log"
python - <<'PY'
import json
for l in open('gpurun_out/r06q/simranks8_repeat.jsonl'):
    d=json.loads(l); print(d['config']['parallelism'], d['ms_per_step'])
for f in ('simranks8_p2p_all_events','projals_4096_all_events'):
    d=json.load(open('gpurun_out/r06q/%s.json'%f))
    print(f, d['ms_per_step'], [(k['name'],round(k['avg_us'],1)) for k in d['kernels']])
PY
