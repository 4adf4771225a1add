#!/bin/bash
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/s3"; mkdir -p "$O"; cd "$R"
export GPU_MAX_HW_QUEUES=24
python scripts/projals_f32_error.py > "$O/projals_f32_error.log" 2>&1
timeout 1500 python -m pytest tests/test_gpu_projals_alspgrad.py tests/test_gpu_utils.py tests/test_gpu_c4_c5.py tests/test_golden.py -q -m gpu -k "projals or pdsolve or c4 or golden" 2>&1 | tail -25 > "$O/tests_potrs2.log"
timeout 900 python -m pytest tests/test_gpu_localcomm.py tests/test_gpu_comm.py tests/test_gpu_peer.py -q -m gpu -k "projals" 2>&1 | tail -8 > "$O/tests_potrs3.log"
cat "$O/projals_f32_error.log"; tail -25 "$O/tests_potrs2.log"; tail -5 "$O/tests_potrs3.log"
