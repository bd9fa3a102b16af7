#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_dict.py tests/test_gpu_resident.py -x -q -m gpu > gpurun_out/t_dict.log 2>&1; rc=$?
tail -3 gpurun_out/t_dict.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/bench_1024.json 2> gpurun_out/bench_1024.err; echo "bench rc $?"
timeout -k 10 120 python bench.py --steps 300 --warmup 30 --grid 512 --no-cpu-baseline > gpurun_out/bench_512.json 2>&1; echo "512 rc $?"
timeout -k 10 120 python bench.py --steps 300 --warmup 30 --grid 256 --pc jacobi --no-cpu-baseline > gpurun_out/bench_256j.json 2>&1; echo "256j rc $?"
timeout -k 10 120 python bench.py --steps 300 --warmup 30 --grid 1024 --grid-y 128 --no-cpu-baseline > gpurun_out/bench_slab8.json 2>&1; echo "slab8 rc $?"
timeout -k 10 120 python bench.py --steps 300 --warmup 30 --grid 1024 --grid-y 256 --no-cpu-baseline > gpurun_out/bench_slab4.json 2>&1; echo "slab4 rc $?"
timeout -k 10 120 python bench.py --steps 300 --warmup 30 --grid 1024 --grid-y 512 --no-cpu-baseline > gpurun_out/bench_slab2.json 2>&1; echo "slab2 rc $?"
python tools/bench_summary.py
