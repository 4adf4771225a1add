#!/bin/bash
# Round 6: 16-byte staged peer stores of the X_g H_g' epilogue; ALSPGrad rotating sets restored.
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06h"; mkdir -p "$O"; cd "$R"
python -m pytest tests/test_gpu_peer.py tests/test_gpu_localcomm.py tests/test_gpu_comm.py -x -q -m gpu > "$O/pytest_a.log" 2>&1
tail -3 "$O/pytest_a.log"
B="python bench.py --no-cpu-baseline --sim-ranks 8 --steps 50 --warmup 10"
: > "$O/simranks.jsonl"
for i in 1 2; do $B --transport p2p --no-events >> "$O/simranks.jsonl" 2>> "$O/err.log"; done
$B --transport rccl --no-events >> "$O/simranks.jsonl" 2>> "$O/err.log"
$B --transport p2p --all-events > "$O/simranks8_p2p_all_events.json" 2>> "$O/err.log"
for g in 2 4; do for tr in p2p rccl; do python bench.py --no-cpu-baseline --sim-ranks $g --steps 50 --warmup 10 --transport $tr --no-events >> "$O/simranks.jsonl" 2>> "$O/err.log"; done; done
python - <<'PY'
import json
for l in open('gpurun_out/r06h/simranks.jsonl'):
    d=json.loads(l); print(d.get('sim_ranks'), d['config']['parallelism'], d['ms_per_step'])
d=json.load(open('gpurun_out/r06h/simranks8_p2p_all_events.json'))
print(d['ms_per_step'], [(k['name'],k['avg_us']) for k in d['kernels']])
PY
python -m pytest tests/test_gpu_projals_alspgrad.py tests/test_golden.py tests/test_gpu_c4_c5.py -x -q -m gpu -k "alspgrad or c5 or golden" > "$O/pytest_b.log" 2>&1
tail -3 "$O/pytest_b.log"
python bench.py --no-cpu-baseline --alg alspgrad --dtype f64 --p 32768 --n 4096 --k 512 --steps 3 --warmup 1 --no-events > "$O/bench_alspgrad_c5_shard.json" 2>> "$O/err.log"
python -c "
import json; d=json.load(open('gpurun_out/r06h/bench_alspgrad_c5_shard.json')); print('c5 shard', d['ms_per_step'])"
