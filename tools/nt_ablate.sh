#!/bin/bash
# developer tool: plain NT GEMM device time under the SDT_NT_DBG ablations (4: no staging DMA in the loop, 2: no MFMAs, 1: no DMA
# waits; wrong results).  usage: tools/nt_ablate.sh "0 4 2" "lin320 lin1280"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for dbg in ${1:-0 4 2}; do
  for sh in ${2:-lin320 lin640 lin1280 ff320 qkv640}; do
    rm -rf gpurun_out/nta
    SDT_LIB=${SDT_LIB:-} SDT_NT_DBG=$dbg rocprofv3 --kernel-trace --stats -d gpurun_out/nta -o s --output-format csv -- python3 tools/gemm_micro.py $sh 30 > /dev/null 2>&1
    python - "$dbg" "$sh" <<'PY'
import csv, sys
for r in csv.DictReader(open('gpurun_out/nta/s_kernel_stats.csv')):
    if 'gemm_nt' in r['Name'] or 'empty' in r['Name'].lower():
        print(f"dbg={sys.argv[1]:>2s} {sys.argv[2]:9s} {r['Name'][5:40]:36s} calls={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:7.1f} us min={float(r['MinNs'])/1e3:7.1f}", flush=True)
PY
  done
done
