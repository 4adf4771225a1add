#!/bin/bash
# Round 6: the row-sharded ProjectedALS step with XH' on the transposed images too: sharded tests, C4's 8-rank shard line A/B.
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06h2"; mkdir -p "$O"; cd "$R"
timeout 1500 python -m pytest tests/test_gpu_localcomm.py tests/test_gpu_peer.py tests/test_gpu_comm.py tests/test_gpu_c4_c5.py -x -q -m gpu -k "projals or c4" > "$O/pytest_a.log" 2>&1
tail -3 "$O/pytest_a.log"
B="python bench.py --no-cpu-baseline --sim-ranks 8 --alg projals --p 16384 --n 131072 --k 256 --steps 10 --warmup 3 --no-events --transport rccl"
: > "$O/lines.jsonl"
for rep in 1 2; do
  $B >> "$O/lines.jsonl" 2>> "$O/err.log"
  NMFX_PROJALS_XT=0 $B >> "$O/lines.jsonl" 2>> "$O/err.log"
done
python - <<'PY'
import json
for i,l in enumerate(open('gpurun_out/r06h2/lines.jsonl')):
    d=json.loads(l); print(('images','row-contiguous')[i%2], d['ms_per_step'])
PY
