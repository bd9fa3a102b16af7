#!/bin/bash
# developer: workgroup count of the pipelined 2x2 product against its time INSIDE a solve
for w in 256 384 512 768 1024; do
  SPK_DICT2_WGS=$w timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_1024_w$w.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_1024_w$w.json")); r=d["roofline"]
print("wgs $w: in-solve", round(r["ms"]*1e3,2), "batch", round(r["ms_back_to_back"]*1e3,2), "it/s full cycles", round(d["value_full_cycles"]))
PY
done
