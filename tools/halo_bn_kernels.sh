#!/bin/bash
# developer tool: per-kernel device time of the halo convolution (forward <.., true, BN>, input gradient <.., false, BN>) for both tile widths
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for bn in ${BNS:-128 64}; do export SDT_HALO_MFMA=${MFMA:-32};
  for sh in ${SHAPES:-conv320 conv512 conv1280 conv1280s}; do
    rm -rf gpurun_out/hbn
    SDT_HALO_BN=$bn rocprofv3 --kernel-trace --stats -d gpurun_out/hbn -o s --output-format csv -- python3 tools/gemm_micro.py $sh 20 > /dev/null 2>&1
    python - "$bn" "$sh" <<'PY'
import csv, sys
for r in csv.DictReader(open('gpurun_out/hbn/s_kernel_stats.csv')):
    if 'halo' in r['Name'] or 'wgrad3' in r['Name']:
        print(f"bn={sys.argv[1]:>3s} {sys.argv[2]:10s} {r['Name'][5:48]:44s} calls={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:8.1f} us", flush=True)
PY
  done
done
