#!/bin/bash
# Round 6: where does potrf_reg_kernel spend its time when it runs "under" a product?  Needs an INSTRUMENTED library (not the shipped one):
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DNMFX_POTRF_TIMING -DNMFX_POTRF_TIMING_LIB -c nmf.jl_amd/csrc/solver_f32.hip -o /tmp/solver_f32.o
#   hipcc --offload-arch=gfx950 -shared -fPIC nmf.jl_amd/lib/obj/nmfx_api.o /tmp/solver_f32.o nmf.jl_amd/lib/obj/solver_f64.o -o nmf.jl_amd/lib/libnmfx.so -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
# (then rebuild the shipped one: python -c "import __graft_entry__ as g; g.build()").  NMFX_POTRF_DUMP=1 prints the cycle stamps of the
# last launch of a solve: entry -> first block step, then phases A / B / C of every block step for waves 0 and 7.
# Result (profiles/r06_projals_chain_timeline_and_potrf_stamps.txt): ~160 k cycles from the first to the last instruction in stream order
# and under the short grid alike -- the 460 us between its dispatch and its end under the product are spent before the first instruction.
export NMFX_DEV=1 NMFX_POTRF_DUMP=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06p"; mkdir -p "$O"; cd "$R"
B="python bench.py --no-cpu-baseline --alg projals --p 16384 --n 16384 --k 256 --steps 4 --warmup 2 --no-events --traffic none"
$B > "$O/split.json" 2> "$O/split.err"
NMFX_CHOL_UNDER_US=1e9 $B > "$O/stream_order.json" 2> "$O/stream_order.err"
for v in split stream_order; do echo "== $v"; grep -E "step|entry" "$O/$v.err" | tail -19; done
