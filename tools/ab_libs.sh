#!/bin/bash
# developer tool: same-box A/B of several in-tree builds (tools/build_variant.sh with OUT=libsdtrain_hip_<tag>.so):
#   LIBS="default n3 n2" SHAPES="lin640 clip" tools/ab_libs.sh     micro shapes, then the whole step twice per build, interleaved
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/stable_diffusion_training_amd/csrc
lib() { [ "$1" = default ] && echo "" || echo "$C/libsdtrain_hip_$1.so"; }
for sh in ${SHAPES:-}; do
  for t in $LIBS; do echo "== $t"; SDT_LIB=$(lib $t) python tools/gemm_micro.py $sh 50 2>/dev/null; done
done
for round in 1 2; do
  for t in $LIBS; do
    r=$(SDT_LIB=$(lib $t) python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "lib=$t round $round: $r ms/step"
  done
done
