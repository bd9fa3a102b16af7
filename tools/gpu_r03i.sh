#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_resident.py -x -q -m gpu > gpurun_out/t_res.log 2>&1; rc=$?
tail -25 gpurun_out/t_res.log
exit $rc
