#!/bin/bash
# developer tool: the driver's GPU test command (without -x, to see every failure), then optional extra steps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4full
timeout -k 10 ${FULL_TIMEOUT:-1000} python -m pytest tests/ -q -m gpu -p no:cacheprovider --timeout=300 > gpurun_out/r4full/gpu_tests.log 2>&1
echo "pytest rc=$?"
tail -${TAIL:-40} gpurun_out/r4full/gpu_tests.log
