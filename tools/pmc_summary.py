#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as
MI355X_MICROARCH.md prescribes: FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2)
into profiles/<tag>_pmc_summary.json and profiles/spmv_traffic.json.

gfx950 correction (same guide, section HBM): FETCH_SIZE counts 64 B per 128-B
request of a wide (16 B/lane) coalesced read, i.e. exactly half the bytes ->
doubled here; WRITE_SIZE is exact.  The correction is calibrated in the same
pass on scale_dev_kernel, a pure 16 B/lane stream of known size.

usage: tools/pmc_summary.py gpurun_out/pmc_r01_FETCH_SIZE gpurun_out/pmc_r01_WRITE_SIZE r01 [notraffic]
(notraffic: a pass on another workload than the 1024 x 1024 grid -- profiles/spmv_traffic.json is left alone)
"""
import collections, csv, glob, json, os, statistics, sys

fetch_dir, write_dir, tag = sys.argv[1:4]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def med(d, counter):
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: statistics.median(v) for k, v in agg.items()}


fe, wr = med(fetch_dir, "FETCH_SIZE"), med(write_dir, "WRITE_SIZE")
out = {"_method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two passes), median over dispatches, "
                  "counter unit KiB; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 wide-read correction)",
       "kernels": {}}
for k in sorted(set(fe) | set(wr)):
    if "spk::" not in k:
        continue
    f, w = fe.get(k, 0.0), wr.get(k, 0.0)
    out["kernels"][k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_corrected": (2 * f + w) * 1024}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1)
# the variant bench.py's roofline names first: y += A x with the Givens rider (<NT, ACC=true, RIDE=true>), then any other;
# a format this pass did not run keeps the figure (and source) already in the file
if len(sys.argv) > 4 and sys.argv[4] == "notraffic":
    print(json.dumps(out, indent=1))
    sys.exit(0)
tpath = os.path.join(ROOT, "profiles", "spmv_traffic.json")
try:
    tr = json.load(open(tpath))
except Exception:  # noqa: BLE001
    tr = {}
tr["workload"] = "1024x1024 grid A-block SpMV"
for key, pat in (("csr", "spmv_stream"), ("bcsr2x2", "spmv_bcsr_kernel"), ("dict2x2", "spmv_dict2_kernel")):
    # (first the variant the iteration launches: ACC = true, RIDE = true)
    sp = sorted((k for k in out["kernels"] if pat in k),
                key=lambda k: (0 if ("true, true, true>" in k or "true, true, true, false>" in k or "<2, true, true, false>" in k or
                                      "<true, true, false, 9" in k) else 1, k))
    if sp:
        tr["hbm_bytes_per_launch_" + key] = out["kernels"][sp[0]]["hbm_bytes_corrected"]
        tr["source_" + key] = f"profiles/{tag}_pmc_summary.json ({sp[0]})"
json.dump(tr, open(tpath, "w"), indent=1)
print(json.dumps(out, indent=1))
