#!/bin/bash
# developer tool: same-box A/B of environment switches; usage: tools/ab_bench.sh "VAR=a" "VAR=b" ... (each run twice, interleaved)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for setting in "$@"; do
    r=$(env $setting python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$setting round $round: $r ms/step"
  done
done
