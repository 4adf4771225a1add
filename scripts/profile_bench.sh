#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box): kernel-trace stats of the default command, then separate PMC passes
# (FETCH_SIZE, WRITE_SIZE, SQ counters -- never combined with sys/runtime traces), summarised into profiles/-style markdown.
# usage: [BENCH_ARGS="--precision bf16x3"] bash scripts/profile_bench.sh <outdir under gpurun_out>
R="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$R/gpurun_out/${1:-prof_final}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python "$R/bench.py" $BENCH_ARGS > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o r03 -- python "$R/bench.py" $BENCH_ARGS --no-cpu-baseline --traffic static > "$OUT/bench_stats.log" 2>&1
SHORT="$BENCH_ARGS --steps 4 --warmup 2 --no-cpu-baseline --no-events --traffic none"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o r03 -- python "$R/bench.py" $SHORT > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o r03 -- python "$R/bench.py" $SHORT > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT \
    --kernel-trace -d "$OUT/pmc_sq" -o r03 -- python "$R/bench.py" $SHORT > /dev/null 2>&1
find "$OUT" -name "*.db" | head
python "$R/scripts/rocpd_summary.py" "$(find "$OUT/stats" -name '*.db' | head -1)" \
    FETCH_SIZE="$(find "$OUT/pmc_fetch" -name '*.db' | head -1)" WRITE_SIZE="$(find "$OUT/pmc_write" -name '*.db' | head -1)" \
    SQ="$(find "$OUT/pmc_sq" -name '*.db' | head -1)" > "$OUT/summary.md" 2> "$OUT/summary.err"
tail -3 "$OUT/summary.err"; head -8 "$OUT/summary.md"; cat "$OUT/bench_default.json" | cut -c1-400
# the raw rocpd databases are ~40 MB per pass: keep only the summaries (gpurun copies back at most 64 MiB)
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq"
