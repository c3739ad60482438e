#!/usr/bin/env python3
"""Per-SDE-step kernel breakdown from a rocprofv3 --kernel-trace CSV of bench.py (graph replay region)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "em_update_kernel" in n]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
step = rows[idx[k] + 1: idx[k + 1] + 1]
wall = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
agg = collections.OrderedDict()
for r in step:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1
    a[1] += d
print(f"one SDE step: {wall:.1f} us wall, {len(step)} kernels, {sum(v[1] for v in agg.values()):.1f} us summed")
fam = collections.defaultdict(float)
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n[:72]:74s} x{c:<3d} {d:8.1f} us")
    fam["conv_igemm" if n.startswith("conv_igemm") or n.startswith("splitk") else n.split("<")[0]] += d
print("families:")
for n, d in sorted(fam.items(), key=lambda kv: -kv[1]):
    print(f"  {n:40s} {d:8.1f} us  {100 * d / wall:5.1f} %")
