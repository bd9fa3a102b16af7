#!/bin/bash
# timing experiments on spmv_dict_kernel<2> (rebuilds the kernels per variant on the box; some variants compute wrong results)
cd saddle_point_petsc_amd/csrc
for cfg in "-DSPK_DICT_X_BASE" "-DSPK_DICT_X_NOFIELD" "-DSPK_DICT_X_ONESHIFT"; do
  touch spk_k_dict.hip spk_k_resident.hip
  make -s XDEFS="$cfg" > /dev/null 2>&1
  echo "== $cfg"
  (cd ../.. && timeout -k 10 200 python tools/kbench.py --grid 1024 --kernels spmv_dict,scale --reps 200 2>&1 | grep -E "^spmv|^scale" )
done
