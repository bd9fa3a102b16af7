#!/bin/bash
# developer: timing variants of the pipelined product
set -o pipefail
cd saddle_point_petsc_amd/csrc
for v in "-DSPK_X_NOGATHER -DSPK_X_NOCODES" "-DSPK_X_NOGATHER -DSPK_X_NOCODES -DSPK_X_NOLDS"; do
  touch spk_k_dict.hip
  make -j16 XDEFS="$v" > /dev/null 2>&1 || { echo "build failed $v"; exit 1; }
  echo "variant [$v]"
  for w in 256 512 768; do
  (cd ../.. && echo "wgs $w" && SPK_DICT2_WGS=$w timeout -k 10 200 python tools/kbench.py --grid 1024 --kernels spmv_dict --reps 200 2>&1 | grep -E "^spmv")
  done
done
