#!/bin/bash
# developer: long restarts
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "restart or refine or history or form" > gpurun_out/t_lr.log 2>&1; rc=$?
tail -5 gpurun_out/t_lr.log
[ $rc -ne 0 ] && exit $rc
for r in 62 64 100; do
timeout -k 10 300 python bench.py --steps 300 --warmup 30 --restart $r --no-cpu-baseline > gpurun_out/bench_1024_restart$r.json 2> gpurun_out/bench_1024.err
python - <<PY
import json
d=json.load(open('gpurun_out/bench_1024_restart$r.json'))
m=d['iteration_model']
print($r, 'it/s', d['value'], 'us/it', round(d['ms_per_step']*1e3,1), 'form', d['config']['iteration_form_run'], {k:(round(v['frac_of_peak'],3), round(v['bytes_per_iteration_layout']/1e6)) for k,v in m.items()}, d.get('error'))
PY
done
