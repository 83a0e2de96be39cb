"""Per-kernel totals from a rocprofv3 results database (rocprofv3 --kernel-trace writes <name>_results.db):
python tools/rocpd_stats.py gpurun_out/prof/x_results.db [out.csv]  ->  name, calls, total ms, avg us, % of GPU time."""
import csv
import sqlite3
import sys


def main(db, out=None):
    con = sqlite3.connect(db)
    tables = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    disp = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
    rows = con.execute(f"select s.kernel_name, count(*), sum(d.end - d.start) from {disp} d join {sym} s on d.kernel_id = s.id "
                       "group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    table = [(n, c, t / 1e6, t / c / 1e3, 100.0 * t / total) for n, c, t in rows]
    w = csv.writer(open(out, "w", newline="") if out else sys.stdout)
    w.writerow(["kernel", "calls", "total_ms", "avg_us", "percent"])
    for n, c, ms, us, pct in table:
        w.writerow([n[:160], c, f"{ms:.3f}", f"{us:.2f}", f"{pct:.2f}"])


if __name__ == "__main__":
    main(*sys.argv[1:3])
