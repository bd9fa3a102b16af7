#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
SPK_DICT_VERBOSE=1 timeout -k 10 600 python -m pytest tests/test_gpu_dict.py tests/test_gpu_resident.py tests/test_gpu_parity.py -x -q -m gpu -k "dict or resident or spmv or fp32 or 3d or dictionary" > gpurun_out/t_dict.log 2>&1; rc=$?
tail -6 gpurun_out/t_dict.log; grep "row types" gpurun_out/t_dict.log | sort | uniq -c | sort -rn | head -5
[ $rc -ne 0 ] && exit $rc
SPK_DICT_VERBOSE=1 timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/bench_1024.json 2> gpurun_out/bench_1024.err; echo "bench rc $?"; grep "row types" gpurun_out/bench_1024.err
timeout -k 10 120 python bench.py --steps 300 --warmup 30 --grid 1024 --grid-y 128 --no-cpu-baseline > gpurun_out/bench_slab8.json 2>&1; echo "slab rc $?"
timeout -k 10 120 python bench.py --steps 300 --warmup 30 --grid 256 --pc jacobi --no-cpu-baseline > gpurun_out/bench_256j.json 2>&1; echo "256j rc $?"
timeout -k 10 120 python bench.py --steps 300 --warmup 30 --grid 512 --no-cpu-baseline > gpurun_out/bench_512.json 2>&1; echo "512 rc $?"
SPK_DICT_VERBOSE=1 timeout -k 10 300 python bench.py --steps 60 --warmup 10 --dim 3 --grid 256 --grid-y 256 --grid-z 32 --pc jacobi --inner-sweeps 3 --no-cpu-baseline > gpurun_out/bench_3dslab.json 2> gpurun_out/bench_3d.err; echo "3d rc $?"; grep "row types" gpurun_out/bench_3d.err
python tools/bench_summary.py
