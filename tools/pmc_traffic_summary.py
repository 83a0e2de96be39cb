"""Memory-side bytes per train step from the two counter passes of tools/probe_pmc_passes.sh:

    python tools/pmc_traffic_summary.py gpurun_out/final_pmc_fetch.csv gpurun_out/final_pmc_write.csv OUT.json "note" > OUT.csv

The csv inputs are tools/rocpd_pmc.py's per-kernel sums (KB per step in the last column).  FETCH_SIZE is doubled (gfx950 tallies
128-byte requests at 64 bytes, MI355X_MICROARCH.md "HBM / rocprofv3"); WRITE_SIZE is exact.  The json is what bench.py reads as
`roofline.traffic` (the persistent chain kernel's read + written bytes)."""
import csv
import json
import sys


def load(path):
    return [r for r in csv.DictReader(open(path))]


fetch, write = load(sys.argv[1]), load(sys.argv[2])
out = csv.writer(sys.stdout)
out.writerow(["kernel", "counter", "dispatches", "sum", "sum_per_step"])
for r in fetch + write:
    out.writerow([r["kernel"], r["counter"], r["dispatches"], r["sum"], r["sum_per_step"]])


def total(rows, pat=None):
    return sum(float(r["sum_per_step"]) for r in rows if pat is None or pat in r["kernel"]) * 1024.0


d = {
    "workload": "VRNN [64,1,16000] train step",
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes: tools/probe_pmc_passes.sh) -- python bench.py --steps 2 "
              "--warmup 1 --no-cpu-baseline --no-sweep; 3 steps averaged; FETCH_SIZE doubled per MI355X_MICROARCH.md; " + (sys.argv[4] if len(sys.argv) > 4 else ""),
    "cell_stage_kernels_read_bytes_per_step": 2 * total(fetch, "pchain_kernel"),
    "cell_stage_kernels_write_bytes_per_step": total(write, "pchain_kernel"),
    "wgrad_gemm_read_bytes_per_step": 2 * total(fetch, "gemm_group_kernel"),  # the grouped weight-gradient launches (chain + encoder / decoder MLPs)
    "all_kernels_read_bytes_per_step": 2 * total(fetch),
    "all_kernels_write_bytes_per_step": total(write),
}
json.dump(d, open(sys.argv[3], "w"), indent=1)
