#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_dict.py tests/test_gpu_configs.py -x -q -m gpu -k "3d or config5 or dictionary" 2>&1 | tail -4
timeout -k 10 300 python bench.py --steps 60 --warmup 10 --dim 3 --grid 256 --grid-y 256 --grid-z 32 --pc jacobi --inner-sweeps 3 --no-cpu-baseline > gpurun_out/bench_3dslab_inner.json 2>/dev/null
timeout -k 10 300 python bench.py --steps 60 --warmup 10 --dim 3 --grid 256 --grid-y 256 --grid-z 32 --no-cpu-baseline > gpurun_out/bench_3dslab_schur.json 2>/dev/null
timeout -k 10 300 python bench.py --steps 60 --warmup 10 --dim 3 --grid 96 --no-cpu-baseline > gpurun_out/bench_3d96.json 2>/dev/null
python tools/bench_summary.py
