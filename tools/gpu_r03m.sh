#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_general_b.py -x -q -m gpu > gpurun_out/t_gb.log 2>&1; rc=$?
tail -4 gpurun_out/t_gb.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 300 --warmup 30 --dim 3 --grid 96 --constraints div3d --no-cpu-baseline > gpurun_out/bench_3d96_div3d.json 2> gpurun_out/bench_3d.err; echo "div3d rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_3d96_div3d.json'))
print('div3d', d['value'], 'it/s', d['ms_per_step']*1e3, 'us/it', 'rep', d.get('value_representative'))
PY
