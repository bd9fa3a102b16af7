#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for w in 256 384 512; do
  SPK_DICT2_WGS=$w timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/bench_1024_w$w.json 2> gpurun_out/bench_1024.err; echo "bench $w rc $?"
  SPK_DICT2_WGS=$w timeout -k 10 120 python bench.py --steps 300 --warmup 30 --grid 512 --no-cpu-baseline > gpurun_out/bench_512_w$w.json 2>&1; echo "512 rc $?"
done
python tools/bench_summary.py gpurun_out/bench_1024_w*.json gpurun_out/bench_512_w*.json
