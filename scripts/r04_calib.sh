#!/bin/bash
# (GPU box) 1. FETCH_SIZE / WRITE_SIZE per byte for the access widths used here; 2. the big products of greedycd vs multmse: busy cycles vs wall time
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r04c"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -std=c++17 "$R/scripts/kbench/fetch_calib.hip" -o /tmp/fetch_calib 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/calib_$c -- /tmp/fetch_calib > /dev/null 2>&1
  python "$R/scripts/pmc_summary.py" /tmp/calib_$c "" > "$O/calib_$c.txt" 2>&1
done
cat "$O"/calib_*.txt
for alg in multmse greedycd; do
  rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/clk_$alg -- python "$R/bench.py" --alg $alg --steps 4 --warmup 2 --no-cpu-baseline --no-events --traffic none > /dev/null 2>&1
  python - "$alg" <<'PY' > "$O/clock_$alg.txt" 2>&1
import csv, glob, sys, collections
alg = sys.argv[1]
dur = collections.defaultdict(list); cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"/tmp/clk_{alg}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_mfma_kernel<float, 1, 1, 128, 128, 2, 2, nmfx::EpiStore<float>" in r["Kernel_Name"] or "gemm_mfma_kernel<float, 0, 0, 128, 128, 2, 2, nmfx::EpiStore<float>" in r["Kernel_Name"]:
            dur[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for f in glob.glob(f"/tmp/clk_{alg}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "128, 128, 2, 2, nmfx::EpiStore<float>" in r["Kernel_Name"]:
            cnt[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in dur:
    d = sum(dur[k]) / len(dur[k])
    print(alg, k, "launches", len(dur[k]), "avg us %.1f" % d, {c: round(sum(v) / len(v)) for c, v in cnt[k].items()},
          {c + "_per_us": round(sum(v) / len(v) / d, 1) for c, v in cnt[k].items()})
PY
  cat "$O/clock_$alg.txt" 2>/dev/null
done
