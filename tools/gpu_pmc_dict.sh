#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc_x
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_x --output-format csv -- python3 $R/tools/kbench.py --grid 1024 --kernels spmv_dict --reps 20 > /dev/null 2> /tmp/pmc_x.err
  f=$(ls /tmp/pmc_x/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -z "$f" ] && { echo "no csv for $set"; tail -3 /tmp/pmc_x.err; continue; }
  python3 - "$f" <<'PY'
import csv, sys, collections, statistics
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "spmv_dict2_kernel" in k:
        agg[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(statistics.median(v)) for c, v in d.items()})
PY
done
