export NMFX_DEV=1 TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06m
B="python bench.py --no-cpu-baseline --alg multdiv --steps 20 --warmup 5 --all-events"
$B > gpurun_out/r06m/split.json 2>/dev/null
NMFX_BIG_SPLITS=1 $B > gpurun_out/r06m/unsplit.json 2>/dev/null
python - <<'PY'
import json
for f in ("split","unsplit"):
    d=json.load(open("gpurun_out/r06m/%s.json"%f)); print(f, d["ms_per_step"], d["ms_per_step_no_events"])
    for v in d["kernels"]: print("   ", v["name"], v["avg_us"])
PY
