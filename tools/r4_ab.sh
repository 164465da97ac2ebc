#!/bin/bash
# developer tool (round 4): same-box comparison of in-tree builds - kernel tests, whole-step A/B, per-kernel device time and
# fabric reads of the weight-gradient family.  usage: LIBS="r3 default" tools/r4_ab.sh [tests] [bench] [kstats] [pmc]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/stable_diffusion_training_amd/csrc
O=gpurun_out/r4ab
mkdir -p $O
lib() { [ "$1" = default ] && echo "" || echo "$C/libsdtrain_hip_$1.so"; }
want() { case " $STEPS " in *" $1 "*) return 0;; esac; return 1; }
STEPS="${@:-tests bench kstats pmc}"
FAM="${FAM:-gemm_tn conv_wgrad}"
if want tests; then
  timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "${TESTS_K:-wgrad or weight_gradient or split_reduction or linear_fwd_bwd or conv2d}" -p no:cacheprovider > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
  tail -3 $O/tests.log
fi
if want bench; then
  for round in 1 2; do
    for t in $LIBS; do
      r=$(SDT_LIB=$(lib $t) python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline ${BENCH_ARGS:-} 2>$O/bench_$t.err | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
      echo "lib=$t round $round: $r ms/step"
    done
  done
fi
summ() {  # $1 = csv, $2 = label
python - "$1" "$2" $FAM <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
fam = sys.argv[3:]
tot = 0.0
for r in rows:
    if any(k in r['Name'] for k in fam):
        tot += float(r['TotalDurationNs'])
        print(f"{sys.argv[2]:8s} {r['Name'][5:70]:66s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:8.1f} us total={float(r['TotalDurationNs'])/1e6:8.2f} ms", flush=True)
print(f"{sys.argv[2]:8s} family total {tot/1e6:.2f} ms")
PY
}
if want kstats; then
  for t in $LIBS; do
    rm -rf $O/ks_$t
    export SDT_LIB=$(lib $t)
    rocprofv3 --kernel-trace --stats -d $O/ks_$t -o k --output-format csv -- python3 bench.py --steps 20 --warmup 0 --no-cpu-baseline --no-roofline ${BENCH_ARGS:-} > $O/ks_$t.log 2>&1
    unset SDT_LIB
    find $O/ks_$t -name "*kernel_trace.csv" -delete
    f=$(find $O/ks_$t -name "*kernel_stats.csv" | head -1)
    summ $f $t
  done
fi
if want pmc; then
  export SDT_GRAPH=0
  for t in $LIBS; do
    rm -rf $O/pmc_$t
    export SDT_LIB=$(lib $t)
    rocprofv3 --pmc FETCH_SIZE -d $O/pmc_$t -o f --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline ${BENCH_ARGS:-} > $O/pmc_$t.log 2>&1
    unset SDT_LIB
    f=$(find $O/pmc_$t -name "*counter_collection.csv" | head -1)
    python - $f $t $FAM <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].replace('void ', '').split('(')[0]
    if any(k in n for k in sys.argv[3:]) and r['Counter_Name'] == 'FETCH_SIZE':
        agg[n][0] += float(r['Counter_Value']); agg[n][1] += 1
for n, (v, c) in sorted(agg.items()):
    # FETCH_SIZE is in KB and counts 64 B per 128-B request on gfx950: read bytes = 2 x FETCH_SIZE KB
    print(f"{sys.argv[2]:8s} {n[:60]:60s} launches={c:4d} read/launch={2*v*1024/c/1e6:9.1f} MB total={2*v*1024/1e9:7.2f} GB")
PY
    find $O/pmc_$t -name "*.csv" -size +20M -delete
  done
fi
