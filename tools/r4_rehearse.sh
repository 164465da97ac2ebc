#!/bin/bash
# developer tool: rehearse `bench.py --gpus N` for N = 3 (cannot be sliced: all-reduce only) and N = 4 (both exchange modes) with every
# rank on cuda:0 and gloo collectives (RCCL refuses several ranks on one device)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4rehearse
mkdir -p $O
for n in ${NS:-3 4}; do
  SDT_BENCH_BACKEND=gloo SDT_BENCH_ONE_DEVICE=1 SDT_GRAPH=${GRAPH:-1} timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29740 + n)) \
    bench.py --gpus $n --steps 1 --warmup 0 --batch 1 > $O/bench_gloo$n.json 2> $O/bench_gloo$n.err; echo "N=$n rc=$?"
  python - $O/bench_gloo$n.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
m = d["exchange_modes"]
print({k: (round(v["ms_per_step"]), v["ranks_agree"], v["state_digest"][:10], v["payload_bytes"]) for k, v in m.items() if isinstance(v, dict)},
      {k: v for k, v in m.items() if not isinstance(v, dict)}, d["dist"])
PY
done
