#!/bin/bash
# Round 6: ALSPGrad with rotating gradient sets + light apply pass; deferred stop rule of the row-sharded step.
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06f"; mkdir -p "$O"; cd "$R"
python -m pytest tests/test_gpu_projals_alspgrad.py tests/test_golden.py tests/test_gpu_track_stop.py -x -q -m gpu > "$O/pytest_a.log" 2>&1
tail -3 "$O/pytest_a.log"
python bench.py --no-cpu-baseline --alg alspgrad --dtype f64 --p 32768 --n 4096 --k 512 --steps 3 --warmup 1 --no-events > "$O/bench_alspgrad_c5_shard.json" 2>> "$O/err.log"
python bench.py --no-cpu-baseline --alg alspgrad --dtype f64 --p 32768 --n 4096 --k 512 --steps 3 --warmup 1 --sim-ranks 8 --no-events > "$O/bench_alspgrad_c5_simranks8.json" 2>> "$O/err.log"
B="python bench.py --no-cpu-baseline --sim-ranks 8 --steps 50 --warmup 10"
: > "$O/simranks.jsonl"
$B --transport p2p --no-events >> "$O/simranks.jsonl" 2>> "$O/err.log"
NMFX_DEFER_CHECK=0 $B --transport p2p --no-events >> "$O/simranks.jsonl" 2>> "$O/err.log"
$B --transport rccl --no-events >> "$O/simranks.jsonl" 2>> "$O/err.log"
NMFX_DEFER_CHECK=0 $B --transport rccl --no-events >> "$O/simranks.jsonl" 2>> "$O/err.log"
python - <<'PY'
import json
for f in ('bench_alspgrad_c5_shard','bench_alspgrad_c5_simranks8'):
    try:
        d=json.load(open(f'gpurun_out/r06f/{f}.json')); print(f, d['ms_per_step'], d.get('inner_iters_per_step'), d.get('backtracks_per_step'), d['objvalue'])
    except Exception as e: print(f, 'ERR', e)
for l in open('gpurun_out/r06f/simranks.jsonl'):
    d=json.loads(l); print(d.get('sim_ranks'), d['config']['parallelism'], d['ms_per_step'])
PY
python -m pytest tests/test_gpu_c4_c5.py tests/test_gpu_localcomm.py tests/test_gpu_peer.py tests/test_gpu_comm.py -x -q -m gpu > "$O/pytest_b.log" 2>&1
tail -3 "$O/pytest_b.log"
