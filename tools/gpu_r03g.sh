#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 60 --warmup 10 --dim 3 --grid 96 --pc schur-full --no-cpu-baseline > gpurun_out/bench_3d96_moments.json 2>&1; echo "rc $?"
timeout -k 10 300 python bench.py --steps 60 --warmup 10 --dim 3 --grid 96 --pc schur-full --constraints div3d --no-cpu-baseline > gpurun_out/bench_3d96_div3d.json 2>&1; echo "rc $?"
tail -c 600 gpurun_out/bench_3d96_div3d.json
python tools/bench_summary.py
