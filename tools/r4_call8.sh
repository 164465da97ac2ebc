#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 1024 256; do echo "== SDT_WGRAD3_MIN_M=$v"; SDT_WGRAD3_MIN_M=$v python tools/tn_group_micro.py tools/data/sd15_b4_wgrad_groups.txt 20 2>&1 | grep "^conv\|^sum" | cut -c1-60; done
tools/ab_bench.sh "SDT_WGRAD3_MIN_M=1024" "SDT_WGRAD3_MIN_M=256"
