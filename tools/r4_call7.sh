#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/stable_diffusion_training_amd/csrc
mkdir -p gpurun_out/r4c7
for t in gn0 default; do
  echo "== $t"
  if [ $t = default ]; then unset SDT_LIB; else export SDT_LIB=$C/libsdtrain_hip_$t.so; fi
  python tools/gn_micro.py 2>/dev/null | awk '{print $1, $2, $3, $4, $5, "apply", $7, $8, $9, "fwd", $12, $13}' | tail -12
done
unset SDT_LIB
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -p no:cacheprovider -k "groupnorm or norms_are or epilogue_groupnorm" > gpurun_out/r4c7/t.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4c7/t.log
FAM="lion sqnorm gemm_tn conv_wgrad gn_apply" LIBS="default" tools/r4_ab.sh kstats
tools/ab_libs.sh
