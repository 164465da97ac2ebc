#!/bin/bash
# developer tool: same-box A/B of the default build against csrc/libsdtrain_hip_alt.so (tools/build_variant.sh): conv / GEMM
# micro shapes, then the whole step (each twice, interleaved)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ALT=$GRAFT_REPO_ROOT/stable_diffusion_training_amd/csrc/libsdtrain_hip_alt.so
for sh in ${SHAPES:-convvae128 convvae256 conv512 conv320 conv1280}; do
  echo "== default"; python tools/gemm_micro.py $sh 20 2>/dev/null
  echo "== alt";     SDT_LIB=$ALT python tools/gemm_micro.py $sh 20 2>/dev/null
done
for round in 1 2; do
  for lib in "" "$ALT"; do
    r=$(SDT_LIB=$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "lib=${lib:-default} round $round: $r ms/step"
  done
done
