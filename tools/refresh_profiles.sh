#!/bin/bash
# developer tool: the GPU-side half of the profiles/ refresh (run through gpurun in two calls: `part1`, `part2`);
# tools/make_profile_summary.py is the other half.  rocprofv3 gets the program itself after `--` (no env / bash -c hop).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${SDT_ROUND:-r03}p
mkdir -p $O
if [ "$1" = "part1" ]; then
  python bench.py > $O/bench_sd15.json 2> $O/bench_sd15.err
  tail -c 400 $O/bench_sd15.json; echo
  rm -rf $O/kstats $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/markers
  rocprofv3 --kernel-trace --stats -d $O/kstats -o k --output-format csv -- python3 bench.py --steps 20 --warmup 0 --no-cpu-baseline --no-roofline > $O/kstats.log 2>&1
  find $O/kstats -name "*kernel_trace.csv" -delete
  echo "kernel stats done"
  export SDT_GRAPH=0
  rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_fetch.log 2>&1
  echo "fetch pass done"
  rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_write.log 2>&1
  echo "write pass done"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma -o m --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_mfma.log 2>&1
  echo "mfma pass done"
  export SDT_ROCTX_SYNC=1   # host ranges bracket the device work of each phase
  rocprofv3 --marker-trace --kernel-trace --stats -d $O/markers -o t --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/markers.log 2>&1
  python tools/phase_times.py $O/markers > $O/phase_times.md || true
  find $O/markers -name "*_trace.csv" -delete
  echo "marker pass done"
  find $O -name "*.csv" -size +20M -delete
  ls -la $O $O/kstats $O/pmc_mfma | head -40
elif [ "$1" = "markers" ]; then
  export SDT_GRAPH=0 SDT_ROCTX_SYNC=1
  rm -rf $O/markers
  rocprofv3 --marker-trace --kernel-trace --stats -d $O/markers -o t --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/markers.log 2>&1
  python tools/phase_times.py $O/markers > $O/phase_times.md || true
  find $O/markers -name "*_trace.csv" -delete
  cat $O/phase_times.md
elif [ "$1" = "cpu" ]; then
  python bench.py --steps 5 --warmup 2 --no-roofline --cpu-baseline-full > $O/bench_cpu_full.json 2> $O/bench_cpu_full.err
  tail -c 900 $O/bench_cpu_full.json
else
  for cfg in sd21_768 sdxl_1024; do
    python bench.py --config $cfg --steps 8 --warmup 2 > $O/bench_$cfg.json 2> $O/bench_$cfg.err
    tail -c 700 $O/bench_$cfg.json; echo
    rm -rf $O/kstats_$cfg
    rocprofv3 --kernel-trace --stats -d $O/kstats_$cfg -o k --output-format csv -- python3 bench.py --config $cfg --steps 8 --warmup 0 --no-roofline --no-cpu-baseline > $O/kstats_$cfg.log 2>&1
    find $O/kstats_$cfg -name "*kernel_trace.csv" -delete
    echo "$cfg done"
  done
fi
