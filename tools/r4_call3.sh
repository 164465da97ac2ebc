#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c3
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -p no:cacheprovider > gpurun_out/r4c3/kernels.log 2>&1; echo "kernels rc=$?"; tail -4 gpurun_out/r4c3/kernels.log
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -q -m gpu -x -p no:cacheprovider -k "graphed_step or checkpoint_save or bitwise or regression" > gpurun_out/r4c3/model.log 2>&1; echo "model rc=$?"; tail -15 gpurun_out/r4c3/model.log
tools/ab_bench.sh "SDT_NT_DEEP_RING=0" "SDT_NT_DEEP_RING=1"
