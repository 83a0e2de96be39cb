#!/bin/bash
# FETCH_SIZE and WRITE_SIZE passes over the VRNN bench (separate runs; launches serialised: rocprofv3's counter collection aborts on
# deep queues).  Results: gpurun_out/pmc_{fetch,write}/*_results.db
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp HIP_LAUNCH_BLOCKING=1
(while true; do date >> $R/gpurun_out/hb.log; sleep 45; done) &
HB=$!
rc=0
for c in FETCH_SIZE WRITE_SIZE; do
  d=$R/gpurun_out/pmc_$(echo $c | tr A-Z a-z | cut -d_ -f1)
  timeout -k 10 450 rocprofv3 --kernel-trace --pmc $c -d $d -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $d.log 2>&1 || { rc=1; break; }
done
kill $HB
ls -la $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
exit $rc
