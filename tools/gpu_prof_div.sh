#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_div -o run -- python3 $R/bench.py --steps 60 --warmup 10 --dim 3 --grid 96 --pc schur-full --constraints div3d --no-cpu-baseline > $R/gpurun_out/prof_div.json 2> $R/gpurun_out/prof_div.err
python3 $R/tools/rocpd_stats.py /tmp/prof_div/run_results.db $R/gpurun_out/kernel_stats_div3d_96.csv | head -40
