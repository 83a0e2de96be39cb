"""Diagnostic: the kernel sequence of the LAST train step in a rocprofv3 kernel trace (csv): start offset, duration, gap to the
previous kernel's end, grid, name.  usage: python tools/trace_step.py trace.csv [anchor-substring] (default anchor: pchain_kernel = the last
two chain launches to the end; an explicit anchor must occur once per step: one period between its last two occurrences)"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[2] if len(sys.argv) > 2 else "pchain_kernel"
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
# the last step starts at the second-to-last anchor launch (forward chain), give or take the kernels in front of it
start = idx[-2] if len(idx) >= 2 else 0
stop = len(rows)
if len(sys.argv) > 2 and len(idx) >= 2:  # an explicit anchor that occurs ONCE per step: exactly one period
    stop = idx[-1]
rows = rows[:stop]
prev_end = int(rows[start]["Start_Timestamp"])
t0 = prev_end
busy = 0
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("blvm::(anonymous namespace)::", "").replace("void ", "")[:70]
    grid = f'{int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]}'
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {grid:>14s}  {name}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms")
