"""Per-kernel sums of one rocprofv3 --pmc counter from its results database:
python tools/rocpd_pmc.py gpurun_out/pmc_fetch/p_results.db [n_steps]  ->  kernel, dispatches, counter sum (KB for FETCH_SIZE / WRITE_SIZE)."""
import sqlite3
import sys


def per_kernel(db):
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name "
                       "order by 4 desc").fetchall()
    return rows


if __name__ == "__main__":
    rows = per_kernel(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    import csv
    out = csv.writer(sys.stdout)  # kernel names hold commas (argument lists)
    out.writerow(["kernel", "counter", "dispatches", "sum", "sum_per_step"])
    for k, c, n, v in rows:
        out.writerow([k[:140], c, n, f"{v:.1f}", f"{v / steps:.1f}"])
