#!/bin/bash
# developer tool: per-kernel device time of the attention kernels (rocprofv3) for the default build and csrc/libsdtrain_hip_alt.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "" "$GRAFT_REPO_ROOT/stable_diffusion_training_amd/csrc/libsdtrain_hip_alt.so"; do
  [ -n "$lib" ] && [ ! -f "$lib" ] && continue
  export SDT_LIB=$lib
  rm -rf gpurun_out/hat
  rocprofv3 --kernel-trace --stats -d gpurun_out/hat -o s --output-format csv -- python3 tools/attn_micro.py ${1:-4096} ${2:-40} > /dev/null 2>&1
  python - "${lib:-default}" <<'PY'
import csv, sys, os
for r in csv.DictReader(open('gpurun_out/hat/s_kernel_stats.csv')):
    if 'attn_' in r['Name']:
        print(f"{os.path.basename(sys.argv[1]):24s} {r['Name'][5:52]:48s} calls={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:8.1f} us", flush=True)
PY
done
