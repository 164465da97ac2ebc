#!/bin/bash
# developer tool (round 4): where the waves of the grouped weight-gradient kernels spend their cycles (tools/pmc_waits.py) on the
# replayed groups of tools/r4_tn_micro.sh; usage: LIBS="default" tools/r4_tn_pmc.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/stable_diffusion_training_amd/csrc
O=gpurun_out/r4tn
mkdir -p $O
G=${GROUPS_FILE:-tools/data/sd15_b4_wgrad_groups.txt}
lib() { [ "$1" = default ] && echo "" || echo "$C/libsdtrain_hip_$1.so"; }
for t in $LIBS; do
  rm -rf $O/pmc_$t
  export SDT_LIB=$(lib $t)
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmc_$t -o c --output-format csv -- python3 tools/tn_group_micro.py $G 2 > $O/pmc_$t.log 2>&1
  unset SDT_LIB
  f=$(find $O/pmc_$t -name "*counter_collection.csv" | head -1)
  echo "== $t"
  python tools/pmc_waits.py $f 6
  rm -rf $O/pmc_$t
done
