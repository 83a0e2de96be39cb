#!/bin/bash
# usage (GPU box): bash tools/probe_pmc_kernel.sh OUTNAME "COUNTER ..." -- python3 script args   -> gpurun_out/OUTNAME.txt
# one rocprofv3 --pmc pass per counter group (counters only, program directly after --), per-kernel sums
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$1; shift; ctrs=$1; shift; shift
cd /tmp; export TMPDIR=/tmp
: > $R/gpurun_out/$out.txt
for c in $ctrs; do
  rm -rf /tmp/pk; timeout -k 10 300 rocprofv3 --kernel-trace --pmc $(echo $c | tr , ' ') --output-format csv -d /tmp/pk -o p -- "$@" > /tmp/pk.log 2>&1 || { tail -5 /tmp/pk.log; exit 1; }
  f=$(find /tmp/pk -name "*counter_collection.csv" | head -1)
  python3 - "$f" "${PMC_KERNELS:-gemm,wgrad}" >> $R/gpurun_out/$out.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60], r["Counter_Name"])
    acc[k] += float(r["Counter_Value"]); n[k] += 1
for (k, c), v in sorted(acc.items()):
    if any(t in k for t in (sys.argv[2] if len(sys.argv) > 2 else "gemm,wgrad").split(",")): print(f"{k:60s} {c:28s} per launch {v / n[(k, c)]:.4g}  (x{n[(k, c)]})")
PY
done
cat $R/gpurun_out/$out.txt
