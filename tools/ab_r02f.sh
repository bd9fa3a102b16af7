mkdir -p gpurun_out/r02f
for f in 1 2 3; do
  timeout -k 10 100 python bench.py --grid 1024 --grid-y 128 --no-cpu-baseline --iter-form $f > gpurun_out/r02f/slab_form$f.json 2> gpurun_out/r02f/slab_form$f.err || echo "slab form $f failed"
done
SPK_ITERA_OCC=3 timeout -k 10 100 python bench.py --grid 1024 --grid-y 128 --no-cpu-baseline --iter-form 2 > gpurun_out/r02f/slab_form2_occ3.json 2> gpurun_out/r02f/slab_form2_occ3.err || echo "occ3 failed"
for f in 1 2 3; do
  timeout -k 10 100 python bench.py --grid 256 --no-cpu-baseline --iter-form $f > gpurun_out/r02f/g256_form$f.json 2> gpurun_out/r02f/g256_form$f.err || echo "256 form $f failed"
  timeout -k 10 100 python bench.py --grid 512 --no-cpu-baseline --iter-form $f > gpurun_out/r02f/g512_form$f.json 2> gpurun_out/r02f/g512_form$f.err || echo "512 form $f failed"
  timeout -k 10 100 python bench.py --no-cpu-baseline --iter-form $f > gpurun_out/r02f/full_form$f.json 2> gpurun_out/r02f/full_form$f.err || echo "full form $f failed"
done
cd /tmp && export TMPDIR=/tmp
for f in 2 3; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$f -o slab -- python3 $GRAFT_REPO_ROOT/bench.py --grid 1024 --grid-y 128 --no-cpu-baseline --iter-form $f --steps 120 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/prof_$f/slab_results.db $GRAFT_REPO_ROOT/gpurun_out/r02f/slab_form${f}_kernel_stats.csv
done
cd $GRAFT_REPO_ROOT
(SPK_ITER_FORM=2 timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r02f/pytest_form2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02f/pytest_form2.log); tail -3 gpurun_out/r02f/pytest_form2.log
