#!/bin/bash
# usage (GPU box): LIBS="name -" bash tools/probe_rt_deal.sh [batches...]  — the two deals of the VRNN row-group programs on ONE box:
# BLVM_PCHAIN_SHARED=0 (own range for the gentle link) | 1 (shared deal, vrnn.hip vrnn_shared_deal).
# LIB = scratch/variants/<name> ("-": the in-tree library).  -> gpurun_out/rt_deal.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; out=$R/gpurun_out/rt_deal.txt; : > $out
for b in ${@:-128 192 256}; do
  for lib in ${LIBS:--}; do
    for sh in 0 1; do
      e="BLVM_PCHAIN_SHARED=$sh"
      [ "$lib" != "-" ] && e="$e BLVM_HIP_LIB=$R/scratch/variants/$lib/libblvm_hip.so"
      o=$(env $e timeout -k 10 240 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-sweep 2>&1 | tail -1)
      echo "B=$b lib=$lib shared=$sh $(echo "$o" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.3f ms/step  fwd %.3f  bwd %.3f  bpd %.6f  aborts %s" % (d["ms_per_step"], d["roofline"]["fwd_ms"], d["roofline"]["bwd_ms"], d["bits_per_dim"], d.get("async_errors")))' 2>&1 | tail -1)" | tee -a $out
    done
  done
done
