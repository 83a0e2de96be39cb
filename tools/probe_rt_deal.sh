#!/bin/bash
# usage (GPU box): LIBS="fr12 fr16 -" CFGS="1,6 2,0 2,6" bash tools/probe_rt_deal.sh [batches...]  — row-group deal variants of the VRNN
# programs on ONE box: CFG = BLVM_PCHAIN_RT_COLS,BLVM_PCHAIN_RT_SHARED_TL; LIB = scratch/variants/<name> ("-": the in-tree library).
# -> gpurun_out/rt_deal.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; out=$R/gpurun_out/rt_deal.txt; : > $out
for b in ${@:-128 192 256}; do
  for lib in ${LIBS:--}; do
    for cfg in ${CFGS:-1,0 2,0 1,6 2,6}; do
      e="BLVM_PCHAIN_RT_COLS=${cfg%,*} BLVM_PCHAIN_RT_SHARED_TL=${cfg#*,}"
      [ "$lib" != "-" ] && e="$e BLVM_HIP_LIB=$R/scratch/variants/$lib/libblvm_hip.so"
      o=$(env $e timeout -k 10 240 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-sweep 2>&1 | tail -1)
      echo "B=$b lib=$lib cols,shared=$cfg $(echo "$o" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.3f ms/step  fwd %.3f  bwd %.3f  bpd %.6f  aborts %s" % (d["ms_per_step"], d["roofline"]["fwd_ms"], d["roofline"]["bwd_ms"], d["bits_per_dim"], d.get("async_errors")))' 2>&1 | tail -1)" | tee -a $out
    done
  done
done
