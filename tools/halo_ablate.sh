#!/bin/bash
# developer tool: halo-convolution device time under the SDT_NT_DBG ablations (16: no weight DMA in the loop, 32: no halo prefetch)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for dbg in ${DBGS:-0 16 32 48}; do
  for sh in ${SHAPES:-convvae128 conv512 conv320}; do
    rm -rf gpurun_out/hps
    SDT_NT_DBG=$dbg rocprofv3 --kernel-trace --stats -d gpurun_out/hps -o s --output-format csv -- python3 tools/gemm_micro.py $sh 20 > /dev/null 2>&1
    python - "$dbg" "$sh" <<'PY'
import csv, sys
for r in csv.DictReader(open('gpurun_out/hps/s_kernel_stats.csv')):
    if 'halo' in r['Name']:
        print(f"dbg={sys.argv[1]:>3s} {sys.argv[2]:11s} {r['Name'][5:40]:36s} calls={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:8.1f} us", flush=True)
PY
  done
done
