#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "config5" > gpurun_out/t_cfg5.log 2>&1; rc=$?
tail -15 gpurun_out/t_cfg5.log
exit $rc
