#!/bin/bash
# Developer helper: run GPU test groups one process at a time; stop at the first crash/timeout (exit code other
# than 0/1) so that no further GPU work is started after a fault.  Logs go to gpurun_out/.
mkdir -p gpurun_out
rc_all=0
for grp in "$@"; do
  name=$(echo "$grp" | tr '/:[] ' '_____')
  echo "=== $grp"
  timeout -k 10 ${GROUP_TIMEOUT:-420} python -m pytest $grp -q -m gpu --timeout=${TEST_TIMEOUT:-150} -p no:cacheprovider > gpurun_out/$name.log 2>&1
  rc=$?
  tail -n 25 gpurun_out/$name.log
  echo "=== exit $rc"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "STOP: abnormal exit $rc"; exit $rc; fi
  if [ $rc -ne 0 ]; then rc_all=1; fi
done
exit $rc_all
