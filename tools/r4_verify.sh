#!/bin/bash
# developer tool (round 4): smoke(), the two-rank gloo rehearsal of `bench.py --gpus 2` on one GPU (both exchange modes + digests), the
# HIP regression fixture of the current build, then the driver's GPU test command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4verify
mkdir -p $O
python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
SDT_BENCH_BACKEND=gloo SDT_BENCH_ONE_DEVICE=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29733 \
  bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo bench rc=$?"
tail -c 2500 $O/bench_gloo2.json; echo; tail -3 $O/bench_gloo2.err
python tests/golden/make_hip_regression.py $O/tiny_step_hip.npz > $O/fixture.log 2>&1; echo "fixture rc=$?"; tail -4 $O/fixture.log
if [ "$1" = "full" ]; then
  timeout -k 10 900 python -m pytest tests/ -q -m gpu -p no:cacheprovider --timeout=300 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -5 $O/gpu_tests.log
fi
