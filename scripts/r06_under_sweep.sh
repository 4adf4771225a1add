#!/bin/bash
# Round 6: where does hiding ProjectedALS's factorisation chain under the products start to pay?  (chain in stream order vs under the products)
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06t"; mkdir -p "$O"; cd "$R"
B="python bench.py --no-cpu-baseline --alg projals --steps 40 --warmup 10 --no-events"
: > "$O/sweep.jsonl"
for shape in "4096 4096" "8192 4096" "8192 8192" "12288 8192" "8192 16384" "16384 12288"; do
  set -- $shape
  NMFX_CHOL_UNDER_US=1000000000 $B --p $1 --n $2 >> "$O/sweep.jsonl" 2>> "$O/err.log"
  NMFX_CHOL_UNDER_US=0 $B --p $1 --n $2 >> "$O/sweep.jsonl" 2>> "$O/err.log"
done
python - <<'PY'
import json
ls=[json.loads(l) for l in open('gpurun_out/r06t/sweep.jsonl')]
for a,b in zip(ls[0::2], ls[1::2]):
    p,n,k=a['config']['p'],a['config']['n'],a['config']['k']
    print(p,n,'product est us %.0f'%(2.0*p*n*k/150e6),'stream order',a['ms_per_step'],'under',b['ms_per_step'])
PY
