#!/bin/bash
# developer tool: the GPU-side half of the profiles/ refresh (run through gpurun); tools/make_profile_summary.py is the other half
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err
tail -c 600 gpurun_out/bench_r01.json; echo
rm -rf gpurun_out/prof_final gpurun_out/pmc6_fetch gpurun_out/pmc6_write
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_final -o r01 --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_final.log 2>&1
rm -f gpurun_out/prof_final/r01_kernel_trace.csv
echo "kernel stats done"
SDT_GRAPH=0 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc6_fetch -o f --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc6a.log 2>&1
echo "fetch pass done"
SDT_GRAPH=0 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc6_write -o w --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc6b.log 2>&1
echo "write pass done"
ls -la gpurun_out/prof_final gpurun_out/pmc6_fetch gpurun_out/pmc6_write | head -30
