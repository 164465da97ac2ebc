#!/bin/bash
# developer tool: hardware counters of the halo convolution on one micro shape (separate passes per counter group)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
sh=${1:-conv512}
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_SALU"; do
  rm -rf gpurun_out/hpmc
  rocprofv3 --pmc $grp -d gpurun_out/hpmc -o c --output-format csv -- python3 tools/gemm_micro.py $sh 5 > /dev/null 2>&1
  python - "$grp" <<'PY'
import csv, sys, glob, collections
f = glob.glob('gpurun_out/hpmc/**/*counter_collection.csv', recursive=True)
if not f:
    print(sys.argv[1], "-> no output (counter unavailable?)"); sys.exit(0)
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f[0])):
    if 'conv3x3_halo_kernel<false, true>' in r['Kernel_Name']:
        a = agg[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in agg.items():
    print(f"{k:28s} per launch {v / max(n, 1):16.0f}  ({n} launches)", flush=True)
PY
done
