#!/bin/bash
# Matrix-pipe / wave-state counters of the VRNN bench's kernels ("MFMA utilisation on the fused-gate GEMM from
# rocprof counters"): ONE --pmc pass (8 SQ slots + GRBM), counters only + the kernel trace, the program directly after `--`.
#   SQ_VALU_MFMA_BUSY_CYCLES  cycles a SIMD's matrix pipe is busy (summed over SIMDs)     SQ_BUSY_CYCLES  cycles the SQs are busy
#   SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY  quad-cycles of wave lifetime: parked / issue-stalled / issuing
#   SQ_WAVES  waves launched                                                 GRBM_GUI_ACTIVE  busy cycles summed over the 8 XCDs
# -> gpurun_out/pmc_mfma/p_results.db ; per-kernel sums: python tools/rocpd_pmc.py <db> <steps> ; summary: tools/pmc_mfma_summary.py
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
STEPS=${PMC_STEPS:-2}
(while true; do date >> $R/gpurun_out/hb.log; sleep 45; done) &
HB=$!
d=$R/gpurun_out/pmc_mfma
rm -rf $d
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE \
  -d $d -o p -- python3 $R/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-sweep "$@" > $d.log 2>&1
rc=$?
kill $HB
tail -3 $d.log
exit $rc
