#!/bin/bash
# developer: plane skew and workgroup count of the pipelined product
set -o pipefail
for sk in 1280 2304 4352 4608 12544 20736 1048832; do
  for w in 512 768; do
    echo -n "skew $sk wgs $w  "; SPK_DICT_SKEW=$sk SPK_DICT2_WGS=$w timeout -k 10 200 python tools/kbench.py --grid 1024 --kernels spmv_dict --reps 200 2>&1 | grep -E "^spmv"
  done
done
for sk in 4352 0; do for w in 256 512; do
SPK_DICT_SKEW=$sk SPK_DICT2_WGS=$w timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/bench_1024_s${sk}_w$w.json 2> gpurun_out/bench_1024.err
done; done
python tools/bench_summary.py
