#!/bin/bash
# End-of-round refresh (GPU box): default bench line, kernel stats of the same command, both operand modes of every model,
# the counter passes.  Outputs under gpurun_out/final_*; copy the summaries into profiles/.  ROUND=rNN names the files.
R=${GRAFT_REPO_ROOT:-/root/repo}
RN=${ROUND:-r03}
cd $R
timeout -k 10 500 python bench.py > gpurun_out/final_bench_$RN.json 2> gpurun_out/final_bench_$RN.log || { tail -5 gpurun_out/final_bench_$RN.log; exit 1; }
tail -1 gpurun_out/final_bench_$RN.json | cut -c1-400
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fs && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fs -o s -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sweep > /tmp/fs.log 2>&1; cp $(find /tmp/fs -name "*kernel_stats.csv" | head -1) $R/gpurun_out/final_kernel_stats.csv; grep "timed 5" /tmp/fs.log | cut -c1-120) || exit 1
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fc && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fc -o s -- python3 $R/bench.py --model cwvae --steps 3 --warmup 1 --no-cpu-baseline > /tmp/fc.log 2>&1; cp $(find /tmp/fc -name "*kernel_stats.csv" | head -1) $R/gpurun_out/final_cwvae_kernel_stats.csv; grep "timed 3" /tmp/fc.log | cut -c1-120) || exit 1
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fw && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fw -o s -- python3 $R/bench.py --model wavenet --batch 4 --steps 4 --warmup 2 --no-cpu-baseline --no-sweep > /tmp/fw.log 2>&1; cp $(find /tmp/fw -name "*kernel_stats.csv" | head -1) $R/gpurun_out/final_wavenet_b4_kernel_stats.csv; grep "timed 4" /tmp/fw.log | cut -c1-120) || exit 1
bash tools/probe_dtype_modes.sh > gpurun_out/final_dtype_modes.txt 2>&1; cat gpurun_out/final_dtype_modes.txt
python tools/dmol_bench.py > gpurun_out/final_dmol_bench.txt 2>&1; cat gpurun_out/final_dmol_bench.txt
bash tools/probe_pmc_passes.sh > gpurun_out/final_pmc.log 2>&1 || { tail -5 gpurun_out/final_pmc.log; exit 1; }
python tools/rocpd_pmc.py gpurun_out/pmc_fetch/p_results.db 3 > gpurun_out/final_pmc_fetch.csv
python tools/rocpd_pmc.py gpurun_out/pmc_write/p_results.db 3 > gpurun_out/final_pmc_write.csv
head -4 gpurun_out/final_pmc_fetch.csv gpurun_out/final_pmc_write.csv | cut -c1-200
bash tools/probe_pmc_mfma.sh > gpurun_out/final_pmc_mfma.log 2>&1 || { tail -5 gpurun_out/final_pmc_mfma.log; exit 1; }
python tools/pmc_mfma_summary.py gpurun_out/pmc_mfma/p_results.db 3 gpurun_out/final_pmc_mfma.json > gpurun_out/final_pmc_mfma.csv
cat gpurun_out/final_pmc_mfma.json
python tools/pmc_traffic_summary.py gpurun_out/final_pmc_fetch.csv gpurun_out/final_pmc_write.csv gpurun_out/final_pmc_traffic.json "final tree of round ${RN#r0}" > gpurun_out/final_pmc_traffic.csv
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma  # the result databases (tens of MB each) stay on the box: gpurun returns at most 64 MiB
