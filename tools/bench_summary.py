"""prints one line per gpurun_out/bench_*.json (the JSON line bench.py printed)"""
import glob, json
for f in sorted(glob.glob("gpurun_out/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
        continue
    r = d["roofline"]
    print(f"{f}: {d['value']:.0f} it/s {d['ms_per_step'] * 1e3:.1f} us/it | whole cycles {d['value_full_cycles']:.0f} | loop spmv {r['ms'] * 1e3:.1f} us "
          f"{r['format']} frac {r['frac']:.3f} | plain {d['spmv_ms'] * 1e3:.1f} us | setup {d['setup_breakdown']['set_operators_upload']:.3f} s")
