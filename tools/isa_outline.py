"""Outline of a kernel's ISA: loads / waits / MFMAs / branches in order, runs of the same opcode collapsed.
usage: python tools/isa_outline.py file.s kernel_name_substring"""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith("E") or (l.startswith("_Z") and key in l and ":" in l))
out, prev, cnt = [], None, 0
for l in src[start + 1:]:
    l = l.strip()
    if l.startswith("s_endpgm"):
        break
    m = re.match(r"(global_load\w*|buffer_load\w*|s_load\w*|s_waitcnt|v_mfma\w*|s_cbranch\w*|s_branch|\.LBB\w*|s_barrier|ds_write\w*|ds_read\w*|global_store\w*|global_atomic\w*)", l)
    if not m:
        continue
    op = m.group(1)
    if op == "s_waitcnt":
        op = l.split(";")[0].strip()
    if op == prev:
        cnt += 1
    else:
        if prev:
            out.append(f"{prev}" + (f" x{cnt}" if cnt > 1 else ""))
        prev, cnt = op, 1
out.append(f"{prev} x{cnt}")
print("\n".join(out))
