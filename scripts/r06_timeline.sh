#!/bin/bash
# Round 6: when does the potrf on the factorisation stream actually start and end?  rocprofv3 kernel trace of a short ProjectedALS run,
# dispatch by dispatch (scripts/rocpd_timeline.py).  (profiles/r06_projals_chain_timeline_and_potrf_stamps.txt also holds the run with
# the products unsplit, an experimental switch that was removed again: DESIGN.md section 3.2, item (6).)
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06t"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$O/base" -o t -- python "$R/bench.py" --no-cpu-baseline --alg projals --p 16384 --n 16384 --k 256 --steps 4 --warmup 2 --no-events --traffic none > "$O/base.log" 2>&1
python "$R/scripts/rocpd_timeline.py" "$(find "$O/base" -name '*.db' | head -1)" 44 > "$O/timeline_base.txt" 2>&1
rm -rf "$O/base"
cat "$O/timeline_base.txt"
