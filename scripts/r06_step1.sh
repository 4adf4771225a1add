#!/bin/bash
# Round 6, first GPU call: the peer transport's second exchange as a pull (tests + the 8-rank launch sequences), the barrier probe.
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06a"; mkdir -p "$O"; cd "$R"
python -m pytest tests/test_gpu_peer.py tests/test_gpu_localcomm.py tests/test_gpu_comm.py tests/test_gpu_bench_contract.py -x -q -m gpu > "$O/pytest_exchange.log" 2>&1
tail -5 "$O/pytest_exchange.log"
B="python bench.py --no-cpu-baseline --sim-ranks 8 --steps 50 --warmup 10"
: > "$O/simranks8.jsonl"
$B --transport p2p --no-events >> "$O/simranks8.jsonl" 2>> "$O/err.log"
NMFX_P2P_PULL=0 $B --transport p2p --no-events >> "$O/simranks8.jsonl" 2>> "$O/err.log"
$B --transport rccl --no-events >> "$O/simranks8.jsonl" 2>> "$O/err.log"
$B --transport p2p --all-events > "$O/simranks8_p2p_all_events.json" 2>> "$O/err.log"
$B --transport rccl --all-events > "$O/simranks8_rccl_all_events.json" 2>> "$O/err.log"
for g in 2 4; do python bench.py --no-cpu-baseline --sim-ranks $g --steps 50 --warmup 10 --transport p2p --no-events >> "$O/simranks8.jsonl" 2>> "$O/err.log"; done
python - <<'PY'
import json
for l in open('gpurun_out/r06a/simranks8.jsonl'):
    d=json.loads(l); print(d.get('sim_ranks'), d['config']['parallelism'], d['ms_per_step'])
for f in ('p2p','rccl'):
    d=json.load(open(f'gpurun_out/r06a/simranks8_{f}_all_events.json'))
    print(f, d['ms_per_step'], [(k['name'],k['avg_us']) for k in d['kernels']])
PY
hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/kbench/xcd_barrier_probe.hip -o /tmp/xcd_barrier_probe 2>/dev/null && timeout 120 /tmp/xcd_barrier_probe 500 > "$O/xcd_barrier_probe.log" 2>&1
cat "$O/xcd_barrier_probe.log"
