#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_dict.py -x -q -m gpu > gpurun_out/t_dict.log 2>&1; rc=$?
tail -4 gpurun_out/t_dict.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_1024.json 2> gpurun_out/bench_1024.err; echo "1024 rc $?"
timeout -k 10 120 python bench.py --steps 300 --warmup 30 --grid 256 --pc jacobi --no-cpu-baseline > gpurun_out/bench_256j.json 2>&1; echo "256j rc $?"
timeout -k 10 120 python bench.py --steps 300 --warmup 30 --grid 512 --no-cpu-baseline > gpurun_out/bench_512.json 2>&1; echo "512 rc $?"
timeout -k 10 300 python bench.py --steps 60 --warmup 10 --dim 3 --grid 256 --grid-y 256 --grid-z 32 --no-cpu-baseline > gpurun_out/bench_3dslab_schur.json 2>/dev/null; echo "3d rc $?"
timeout -k 10 300 python bench.py --steps 60 --warmup 10 --dim 3 --grid 96 --constraints div3d --no-cpu-baseline > gpurun_out/bench_3d96div.json 2>/dev/null; echo "div rc $?"
python tools/bench_summary.py
python - <<'PY'
import json
for f in ("bench_1024","bench_256j","bench_512","bench_3dslab_schur","bench_3d96div"):
    d=json.load(open(f"gpurun_out/{f}.json")); r=d["roofline"]
    print(f, round(r["ms"]*1e3,2), round(r["ms_back_to_back"]*1e3,2), round(r["frac"],3), r["ms_source"][:60], "...", r["ms_source"][-70:])
PY
