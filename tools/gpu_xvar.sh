#!/bin/bash
set -o pipefail
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1; rc=$?
tail -4 gpurun_out/t_all.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 60 --warmup 10 --dim 3 --grid 96 --no-cpu-baseline > gpurun_out/bench_3d96.json 2>/dev/null
timeout -k 10 300 python bench.py --steps 60 --warmup 10 --dim 3 --grid 96 --pc jacobi --inner-sweeps 3 --no-cpu-baseline > gpurun_out/bench_3d96_inner.json 2>/dev/null
SPK_DICT3_PIPELINE=1 timeout -k 10 300 python bench.py --steps 60 --warmup 10 --dim 3 --grid 96 --pc jacobi --inner-sweeps 3 --no-cpu-baseline > gpurun_out/bench_3d96_inner_pipe.json 2>/dev/null
python tools/bench_summary.py
