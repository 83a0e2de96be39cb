"""Diagnostic: idle time between kernels in a rocprofv3 --kernel-trace csv (argument: the *_kernel_trace.csv).  Prints, for the
last bench step (between the last two optimizer launches), busy time, idle time and the largest gaps with their neighbours."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
# step boundary: the clip / Adam multi_tensor_apply burst; take the last two occurrences of the first kernel after a pchain pair
marks = [i for i, e in enumerate(ev) if "pchain_kernel" in e[2]]
if len(sys.argv) > 2 or len(marks) < 4:  # no persistent launches to cut steps at (or "tail" asked for): the last third of the trace
    a, b = len(ev) * 2 // 3, len(ev) - 1
else:
    fw = marks[0::2]  # forward launches are every other one (VRNN: one forward + one backward per step)
    a, b = fw[-2], fw[-1]
seg = ev[a:b]
busy = 0
cur_end = seg[0][0]
gaps = []
for s, e, n in seg:
    if s > cur_end:
        gaps.append((s - cur_end, n))
    busy += max(0, e - max(s, cur_end))
    cur_end = max(cur_end, e)
wall = seg[-1][1] - seg[0][0]
wall = ev[b][0] - seg[0][0]
print(f"step wall {wall / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(wall - busy) / 1e6:.3f} ms, kernels {len(seg)}")
gaps.sort(reverse=True)
for g, n in gaps[:15]:
    print(f"  gap {g / 1e3:8.1f} us before {n[:90]}")
small = [g for g, _ in gaps]
print(f"  gaps: n={len(small)}, sum {sum(small) / 1e6:.3f} ms, median {sorted(small)[len(small) // 2] / 1e3:.2f} us")
by = {}
for s, e, n in seg:
    k = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    by.setdefault(k, [0, 0])
    by[k][0] += 1
    by[k][1] += e - s
print("per-kernel totals of the step (durations overlap where launches do):")
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"  {t / 1e3:9.1f} us  x{c:3d}  {k}")
