#!/bin/bash
# round-3 first GPU pass: dictionary SpMV tests, the whole GPU suite, bench lines
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout -k 10 300 python -m pytest tests/test_gpu_dict.py -x -q -m gpu > gpurun_out/t_dict.log 2>&1; rc=$?
tail -5 gpurun_out/t_dict.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py --steps 300 --warmup 30 > gpurun_out/bench_1024.json 2> gpurun_out/bench_1024.err; echo "bench rc $?"
tail -c 1500 gpurun_out/bench_1024.json
timeout -k 10 120 python bench.py --steps 300 --warmup 30 --grid 1024 --grid-y 128 --no-cpu-baseline > gpurun_out/bench_slab8.json 2>&1; echo "slab rc $?"
timeout -k 10 120 python bench.py --steps 300 --warmup 30 --grid 256 --pc jacobi --no-cpu-baseline > gpurun_out/bench_256j.json 2>&1; echo "256j rc $?"
timeout -k 10 120 python bench.py --steps 300 --warmup 30 --grid 512 --no-cpu-baseline > gpurun_out/bench_512.json 2>&1; echo "512 rc $?"
timeout -k 10 300 python bench.py --steps 60 --warmup 10 --dim 3 --grid 256 --grid-y 256 --grid-z 32 --pc jacobi --inner-sweeps 3 --no-cpu-baseline > gpurun_out/bench_3dslab.json 2>&1; echo "3d rc $?"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1; rc=$?
tail -5 gpurun_out/t_all.log
exit $rc
