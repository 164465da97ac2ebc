cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_nodp -- python3 bench.py --steps 6 --warmup 2 --no-roofline --no-cpu-baseline > gpurun_out/tr_nodp.log 2>&1
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29711 SDT_DP_FORCE=1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_dp -- python3 bench.py --steps 6 --warmup 2 --no-roofline --no-cpu-baseline > gpurun_out/tr_dp.log 2>&1
for d in tr_nodp tr_dp; do f=$(find gpurun_out/$d -name '*kernel_trace.csv' | head -1); echo $d $f; python tools/gap_analysis.py $f 3 > gpurun_out/$d.gaps.txt 2>&1; rm -rf gpurun_out/$d; done

