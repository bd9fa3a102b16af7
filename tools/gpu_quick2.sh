#!/bin/bash
SPK_BENCH_COMM=gloo timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --grid 192 --steps 45 --warmup 5 --no-cpu-baseline --spmv-reps 5 > gpurun_out/two.out 2> gpurun_out/two.err
echo rc $?
grep -v "^W\|^I\|amdgpu.ids" gpurun_out/two.err | head -40
