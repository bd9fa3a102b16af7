#!/bin/bash
# kernel timeline around a restart boundary: bash tools/gpu_timeline.sh [--grid-y 256] (default: the 1/8 slab)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
GY=${1:-128}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof_t -o run -- python3 $R/bench.py --grid 1024 --grid-y $GY --no-cpu-baseline --steps 120 --warmup 30 > /dev/null 2> $R/gpurun_out/timeline.err
python3 $R/tools/rocpd_timeline.py /tmp/prof_t/run_results.db 14 10 3
