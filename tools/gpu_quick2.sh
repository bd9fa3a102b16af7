#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/t_res.log 2>&1; rc=$?
tail -4 gpurun_out/t_res.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python bench.py --steps 300 --warmup 30 --grid 256 --pc jacobi --no-cpu-baseline > gpurun_out/bench_256j.json 2>&1; echo "256j rc $?"
timeout -k 10 120 python bench.py --steps 300 --warmup 30 --grid 256 --no-cpu-baseline > gpurun_out/bench_256s.json 2>&1; echo "256s rc $?"
timeout -k 10 120 python bench.py --steps 300 --warmup 30 --grid 1024 --grid-y 128 --no-cpu-baseline > gpurun_out/bench_slab8.json 2>&1; echo "slab8 rc $?"
python tools/bench_summary.py
