#!/bin/bash
# usage (on the GPU box): bash tools/probe_trace_all.sh "srnn:64 lstm:64 lstm:8:4000 stcn:64"  -> gpurun_out/trace_<model>_<batch>.log
# per-kernel totals and idle gaps of a few train steps of each bench model (rocprofv3 --kernel-trace + tools/trace_gaps.py)
cd /tmp && export TMPDIR=/tmp
for spec in $1; do
  IFS=: read -r m b t <<< "$spec"
  extra=""; [ -n "$t" ] && extra="--length $t"
  rm -rf /tmp/tr_$m
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$m -o tr -- python /root/repo/bench.py --model $m --batch $b $extra --steps 4 --warmup 2 --no-cpu-baseline --no-sweep > /tmp/tr_$m.log 2>&1
  f=$(find /tmp/tr_$m -name "*kernel_trace.csv" | head -1)
  python /root/repo/tools/trace_gaps.py $f tail > /root/repo/gpurun_out/trace_${m}_${b}.log 2>&1
  grep "step wall" /root/repo/gpurun_out/trace_${m}_${b}.log | sed "s/^/$spec: /"
done
