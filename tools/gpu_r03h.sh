#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_general_b.py tests/test_gpu_parity.py -x -q -m gpu -k "general or block or six or nest or odd" > gpurun_out/t_gb.log 2>&1; rc=$?
tail -5 gpurun_out/t_gb.log
[ $rc -ne 0 ] && exit $rc
bash tools/gpu_r03g.sh
