#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c6
timeout -k 10 600 python -m pytest tests/test_gpu_dp.py tests/test_gpu_kernels.py -q -m gpu -x -p no:cacheprovider -k "two_ranks_one_gpu or lion" > gpurun_out/r4c6/t.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r4c6/t.log
tools/ab_bench.sh "SDT_GRAD_BF16=0" "SDT_GRAD_BF16=1"
