#!/bin/bash
# developer tool: one-rank torchrun launches of bench.py with the RCCL exchange forced on, eager vs split step graphs, several bucket sizes
run() { # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench.py --gpus 1 --steps 8 --warmup 2 --no-roofline --no-cpu-baseline > gpurun_out/probe_$name.log 2>&1 || { echo "$name FAILED"; tail -5 gpurun_out/probe_$name.log; return 1; }
  echo "$name $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/probe_$name.log)"
}
for cfg in "$@"; do
  case $cfg in
    graph_nodp) run $cfg SDT_GRAPH=1 ;;
    eager_nodp) run $cfg SDT_GRAPH=0 ;;
    eager_dp) run $cfg SDT_GRAPH=0 SDT_DP_FORCE=2 ;;
    graph_dp_noev) run $cfg SDT_GRAPH=1 SDT_DP_FORCE=2 SDT_DP_EVENTS=0 ;;
    graph_dp_nocoll) run $cfg SDT_GRAPH=1 SDT_DP_FORCE=1 ;;
    graph_dp_nocoll_noev) run $cfg SDT_GRAPH=1 SDT_DP_FORCE=1 SDT_DP_EVENTS=0 ;;
    eager_shard) run $cfg SDT_GRAPH=0 SDT_DP_FORCE=2 SDT_DP_SHARD=1 ;;
    graph_shard) run $cfg SDT_GRAPH=1 SDT_DP_FORCE=2 SDT_DP_SHARD=1 ;;
    graph_dp*) run $cfg SDT_GRAPH=1 SDT_DP_FORCE=2 SDT_DP_BUCKET_MB=${cfg#graph_dp} ;;
  esac || exit 1
done
