#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_dict.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 300 python bench.py --steps 100 --warmup 5 --grid 2048 --no-cpu-baseline > gpurun_out/bench_2048.json 2>/dev/null
timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/bench_1024.json 2>/dev/null
python tools/bench_summary.py
