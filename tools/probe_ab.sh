#!/bin/bash
# usage: tools/probe_ab.sh "ENV_A" "ENV_B" [rounds] [bench args...]  — alternate two environments of `python bench.py` on ONE box
# (box-to-box differences are larger than most single changes): prints ms/step, forward and backward chain time of every run.
A="$1"; B="$2"; N="${3:-3}"; shift 3 || true
for i in $(seq 1 $N); do
  for tag in A B; do
    if [ $tag = A ]; then E="$A"; else E="$B"; fi
    out=$(env $E python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep "$@" 2>/dev/null)
    echo "$tag [$E] $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.3f ms/step  fwd %.3f  bwd %.3f  bpd %.6f" % (d["ms_per_step"], d["roofline"]["fwd_ms"], d["roofline"]["bwd_ms"], d["bits_per_dim"]))')"
  done
done
