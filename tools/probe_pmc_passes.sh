#!/bin/bash
# FETCH_SIZE and WRITE_SIZE passes over the VRNN bench (separate runs, counters only: no other trace domain).
# Round 1 needed HIP_LAUNCH_BLOCKING=1 here: with ~4 500 launches per step in flight rocprofv3's counter collection aborted with
# "AQL packet is malformed" at ~550 queued packets.  With the recurrent chain as two persistent launches per step (~100 launches
# per step in all) the UNBLOCKED passes complete: the abort was a queue-depth limit of the profiler's packet interception, not a
# fault of a product kernel (the chain's kernels were the same then; tools/launch_gaps.hip reproduces deep queues without them).
# Set PMC_BLOCKING=1 to serialise launches anyway.  Results: gpurun_out/pmc_{fetch,write}/*_results.db
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
[ "${PMC_BLOCKING:-0}" = "1" ] && export HIP_LAUNCH_BLOCKING=1
(while true; do date >> $R/gpurun_out/hb.log; sleep 45; done) &
HB=$!
rc=0
for c in FETCH_SIZE WRITE_SIZE; do
  d=$R/gpurun_out/pmc_$(echo $c | tr A-Z a-z | cut -d_ -f1)
  timeout -k 10 450 rocprofv3 --kernel-trace --pmc $c -d $d -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sweep > $d.log 2>&1 || { rc=1; break; }
done
kill $HB
ls -la $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
exit $rc
