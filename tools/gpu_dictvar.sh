#!/bin/bash
# experiment: load-group size / occupancy of spmv_dict_kernel<2> (rebuilds spk_k_dict.o per variant on the box)
cd saddle_point_petsc_amd/csrc
for cfg in "9 1" "5 1" "5 6" "3 1" "3 7"; do
  set -- $cfg
  touch spk_k_dict.hip
  make -s XDEFS="-DSPK_DICT_G=$1 -DSPK_DICT_MINW=$2" > /dev/null 2>&1
  echo "== G $1 minw $2"
  (cd ../.. && timeout -k 10 200 python tools/kbench.py --grid 1024 --kernels spmv_dict --reps 200 2>&1 | grep spmv_dict | head -1)
done
