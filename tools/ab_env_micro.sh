#!/bin/bash
# developer tool: conv / GEMM micro shapes under environment switches on one box; usage: SHAPES="..." tools/ab_env_micro.sh "VAR=a" "VAR=b"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sh in ${SHAPES:-convvae128 convvae256 conv512 conv320}; do
  for setting in "$@"; do
    echo "== $setting"; env $setting python tools/gemm_micro.py $sh 20 2>/dev/null
  done
done
