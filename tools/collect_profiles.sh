#!/bin/bash
# Collects the per-round evidence under gpurun_out/<tag>/ on the GPU box (copied to profiles/ afterwards):
#   bench lines (plain and under rocprofv3), kernel statistics, PMC traffic (separate --pmc passes)
# usage (on the box, from the repo root): bash tools/collect_profiles.sh r02
tag=${1:-r02}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $out/${tag}_bench_default.json 2> $out/bench_default.err
python3 $R/bench.py --grid 1024 --grid-y 128 --no-cpu-baseline > $out/${tag}_bench_slab_1of8.json 2> $out/bench_slab.err
python3 $R/bench.py --grid 256 --pc jacobi --no-cpu-baseline > $out/${tag}_bench_256_jacobi.json 2> $out/bench_256.err
python3 $R/bench.py --grid 512 --no-cpu-baseline > $out/${tag}_bench_512.json 2> $out/bench_512.err
rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o run -- python3 $R/bench.py --no-cpu-baseline > $out/${tag}_bench_default_under_rocprof.json 2> $out/rocprof_default.err
python3 $R/tools/rocpd_stats.py /tmp/prof_d/run_results.db $out/${tag}_kernel_stats_bench_default.csv > $out/stats_default.txt
rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o run -- python3 $R/bench.py --grid 1024 --grid-y 128 --no-cpu-baseline > $out/${tag}_bench_slab_1of8_under_rocprof.json 2> $out/rocprof_slab.err
python3 $R/tools/rocpd_stats.py /tmp/prof_s/run_results.db $out/${tag}_kernel_stats_slab_1of8.csv > $out/stats_slab.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $out/pmc_full_$c --output-format csv -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --spmv-reps 20 > /dev/null 2> $out/pmc_full_$c.err
  rocprofv3 --kernel-trace --pmc $c -d $out/pmc_slab_$c --output-format csv -- python3 $R/bench.py --grid 1024 --grid-y 128 --steps 60 --warmup 10 --no-cpu-baseline --spmv-reps 20 > /dev/null 2> $out/pmc_slab_$c.err
done
python3 $R/tools/kbench.py --grid 1024 > $out/${tag}_kbench_1024.txt 2>&1
python3 $R/tools/kbench.py --grid 1024 --grid-y 128 --kernels spmv_bcsr,mdot,maxpy --nvs 1,4,8,12,16,20,24,30 > $out/${tag}_kbench_slab_1of8.txt 2>&1
ls $out | head -50
