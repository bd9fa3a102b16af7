#!/bin/bash
# Collects the per-round evidence under gpurun_out/<tag>/ on the GPU box (copied to profiles/ afterwards):
#   bench lines (plain and under rocprofv3), kernel statistics, PMC traffic (separate --pmc passes)
# usage (on the box, from the repo root): bash tools/collect_profiles.sh r03
tag=${1:-r03}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $out/${tag}_bench_default.json 2> $out/bench_default.err
python3 $R/bench.py --grid 1024 --grid-y 128 --no-cpu-baseline > $out/${tag}_bench_slab_1of8.json 2> $out/bench_slab.err
python3 $R/bench.py --grid 1024 --grid-y 128 --iter-form 5 --no-cpu-baseline > $out/${tag}_bench_slab_1of8_form5.json 2> $out/bench_slab5.err
python3 $R/bench.py --grid 256 --pc jacobi --no-cpu-baseline > $out/${tag}_bench_256_jacobi.json 2> $out/bench_256.err
python3 $R/bench.py --grid 256 --no-cpu-baseline > $out/${tag}_bench_256_schur.json 2> $out/bench_256s.err
python3 $R/bench.py --grid 512 --no-cpu-baseline > $out/${tag}_bench_512.json 2> $out/bench_512.err
python3 $R/bench.py --grid 1024 --grid-y 256 --no-cpu-baseline > $out/${tag}_bench_slab_1of4.json 2> $out/bench_slab4.err
python3 $R/bench.py --grid 1024 --grid-y 512 --no-cpu-baseline > $out/${tag}_bench_slab_1of2.json 2> $out/bench_slab2.err
python3 $R/bench.py --restart 100 --no-cpu-baseline > $out/${tag}_bench_restart100.json 2> $out/bench_r100.err
python3 $R/bench.py --steps 60 --warmup 10 --dim 3 --grid 256 --grid-y 256 --grid-z 32 --pc jacobi --inner-sweeps 3 --no-cpu-baseline > $out/${tag}_bench_3d_slab_256x256x32_fp32_inner.json 2> $out/bench_3d.err
python3 $R/bench.py --steps 60 --warmup 10 --dim 3 --grid 256 --grid-y 256 --grid-z 32 --no-cpu-baseline > $out/${tag}_bench_3d_slab_256x256x32_schur.json 2> $out/bench_3ds.err
python3 $R/bench.py --steps 60 --warmup 10 --dim 3 --grid 96 --no-cpu-baseline > $out/${tag}_bench_3d_96_moments.json 2> $out/bench_3d96.err
python3 $R/bench.py --steps 60 --warmup 10 --dim 3 --grid 96 --constraints div3d --no-cpu-baseline > $out/${tag}_bench_3d_96_div3d.json 2> $out/bench_3d96d.err
rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o run -- python3 $R/bench.py --no-cpu-baseline > $out/${tag}_bench_default_under_rocprof.json 2> $out/rocprof_default.err
python3 $R/tools/rocpd_stats.py /tmp/prof_d/run_results.db $out/${tag}_kernel_stats_bench_default.csv > $out/stats_default.txt
python3 $R/tools/rocpd_product.py /tmp/prof_d/run_results.db > $out/${tag}_product_in_solve.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o run -- python3 $R/bench.py --grid 1024 --grid-y 128 --no-cpu-baseline > $out/${tag}_bench_slab_1of8_under_rocprof.json 2> $out/rocprof_slab.err
python3 $R/tools/rocpd_stats.py /tmp/prof_s/run_results.db $out/${tag}_kernel_stats_slab_1of8.csv > $out/stats_slab.txt
rocprofv3 --kernel-trace --stats -d /tmp/prof_j -o run -- python3 $R/bench.py --grid 256 --pc jacobi --no-cpu-baseline > $out/${tag}_bench_256_jacobi_under_rocprof.json 2> $out/rocprof_256.err
python3 $R/tools/rocpd_stats.py /tmp/prof_j/run_results.db $out/${tag}_kernel_stats_256_jacobi.csv > $out/stats_256.txt
rocprofv3 --kernel-trace --stats -d /tmp/prof_3 -o run -- python3 $R/bench.py --steps 60 --warmup 10 --dim 3 --grid 256 --grid-y 256 --grid-z 32 --pc jacobi --inner-sweeps 3 --no-cpu-baseline > /dev/null 2> $out/rocprof_3d.err
python3 $R/tools/rocpd_stats.py /tmp/prof_3/run_results.db $out/${tag}_kernel_stats_3d_slab_fp32_inner.csv > $out/stats_3d.txt
rocprofv3 --kernel-trace --stats -d /tmp/prof_v -o run -- python3 $R/bench.py --steps 60 --warmup 10 --dim 3 --grid 96 --constraints div3d --no-cpu-baseline > /dev/null 2> $out/rocprof_div.err
python3 $R/tools/rocpd_stats.py /tmp/prof_v/run_results.db $out/${tag}_kernel_stats_3d_96_div3d.csv > $out/stats_div.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $out/pmc_full_$c --output-format csv -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --spmv-reps 20 > /dev/null 2> $out/pmc_full_$c.err
  rocprofv3 --kernel-trace --pmc $c -d $out/pmc_slab_$c --output-format csv -- python3 $R/bench.py --grid 1024 --grid-y 128 --iter-form 5 --steps 60 --warmup 10 --no-cpu-baseline --spmv-reps 20 > /dev/null 2> $out/pmc_slab_$c.err
done
python3 $R/tools/kbench.py --grid 1024 > $out/${tag}_kbench_1024.txt 2>&1
ls $out | head -60
