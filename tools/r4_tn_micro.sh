#!/bin/bash
# developer tool (round 4): dump the weight-gradient groups of one real step, then replay them per build (LIBS) with tools/tn_group_micro.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/stable_diffusion_training_amd/csrc
O=gpurun_out/r4tn
mkdir -p $O
G=${GROUPS_FILE:-tools/data/sd15_b4_wgrad_groups.txt}
mkdir -p $O
lib() { [ "$1" = default ] && echo "" || echo "$C/libsdtrain_hip_$1.so"; }
if [ -n "$REDUMP" ]; then G=$O/groups.txt
  SDT_WGRAD_DUMP=1 SDT_GRAPH=0 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline ${BENCH_ARGS:-} 2>/dev/null | grep WGRAD_GROUP | sort | uniq -c | awk '{c=$1; $1=""; for(i=0;i<c/3;i++) print substr($0,2)}' > $G
  wc -l $G
fi
for t in $LIBS; do
  echo "== $t"
  SDT_LIB=$(lib $t) python tools/tn_group_micro.py $G ${REPS:-20} 2>&1 | tee $O/micro_$t.txt | ${FILTER:-cat}
done
