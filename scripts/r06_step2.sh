#!/bin/bash
# Round 6: the 8-rank launch sequences after the descriptor fix (pull and push forms of the peer exchange), then the exchange tests.
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06c"; mkdir -p "$O"; cd "$R"
B="python bench.py --no-cpu-baseline --sim-ranks 8 --steps 50 --warmup 10"
: > "$O/simranks.jsonl"
$B --transport p2p --no-events >> "$O/simranks.jsonl" 2>> "$O/err.log"
NMFX_P2P_PULL=0 $B --transport p2p --no-events >> "$O/simranks.jsonl" 2>> "$O/err.log"
$B --transport rccl --no-events >> "$O/simranks.jsonl" 2>> "$O/err.log"
$B --transport p2p --all-events > "$O/simranks8_p2p_all_events.json" 2>> "$O/err.log"
NMFX_P2P_PULL=0 $B --transport p2p --all-events > "$O/simranks8_p2p_push_all_events.json" 2>> "$O/err.log"
for g in 2 4; do for tr in p2p rccl; do python bench.py --no-cpu-baseline --sim-ranks $g --steps 50 --warmup 10 --transport $tr --no-events >> "$O/simranks.jsonl" 2>> "$O/err.log"; done; done
python - <<'PY'
import json
for l in open('gpurun_out/r06c/simranks.jsonl'):
    d=json.loads(l); print(d.get('sim_ranks'), d['config']['parallelism'], d['ms_per_step'])
for f in ('p2p','p2p_push'):
    d=json.load(open(f'gpurun_out/r06c/simranks8_{f}_all_events.json'))
    print(f, d['ms_per_step'], [(k['name'],k['avg_us']) for k in d['kernels']])
PY
python -m pytest tests/test_gpu_peer.py tests/test_gpu_localcomm.py tests/test_gpu_comm.py tests/test_gpu_projals_alspgrad.py -x -q -m gpu > "$O/pytest_exchange.log" 2>&1
tail -3 "$O/pytest_exchange.log"
