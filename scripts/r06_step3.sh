#!/bin/bash
# Round 6: ALSPGrad on rotating (Z, G) sets -- the tests that pin its counters and bits, then the C5 shard line.
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06d"; mkdir -p "$O"; cd "$R"
python -m pytest tests/test_gpu_projals_alspgrad.py tests/test_golden.py tests/test_gpu_track_stop.py -x -q -m gpu -k "alspgrad or pg or golden" > "$O/pytest_alspgrad.log" 2>&1
tail -5 "$O/pytest_alspgrad.log"
python bench.py --no-cpu-baseline --alg alspgrad --dtype f64 --p 32768 --n 4096 --k 512 --steps 3 --warmup 1 --no-events > "$O/bench_alspgrad_c5_shard.json" 2>> "$O/err.log"
python bench.py --no-cpu-baseline --alg alspgrad --dtype f64 --p 32768 --n 4096 --k 512 --steps 3 --warmup 1 --sim-ranks 8 --no-events > "$O/bench_alspgrad_c5_simranks8.json" 2>> "$O/err.log"
python - <<'PY'
import json
for f in ('bench_alspgrad_c5_shard','bench_alspgrad_c5_simranks8'):
    try:
        d=json.load(open(f'gpurun_out/r06d/{f}.json')); print(f, d['ms_per_step'], d.get('inner_iters_per_step'), d.get('backtracks_per_step'), d['objvalue'])
    except Exception as e: print(f, 'ERR', e)
PY
python -m pytest tests/test_gpu_c4_c5.py tests/test_gpu_localcomm.py tests/test_gpu_peer.py -x -q -m gpu -k "alspgrad or c5" > "$O/pytest_alspgrad2.log" 2>&1
tail -5 "$O/pytest_alspgrad2.log"
