#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV: per kernel name (last `tail` dispatches of each), mean counters."""
import collections, csv, re, sys
path = sys.argv[1]; tail = int(sys.argv[2]) if len(sys.argv) > 2 else 4; filt = sys.argv[3] if len(sys.argv) > 3 else ""
disp = collections.OrderedDict()
for r in csv.DictReader(open(path)):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
    if filt and filt not in n: continue
    d = disp.setdefault((n, int(r["Grid_Size"]), int(r["Dispatch_Id"])), {"dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
by = collections.defaultdict(list)
for (n, grid, did), d in disp.items(): by[(n, grid)].append(d)
rows = []
for (n, grid), ds in by.items():
    ds = ds[-tail:]
    m = {k: sum(d.get(k, 0) for d in ds) / len(ds) for k in ds[0]}
    rows.append((m["dur"], n, grid, m))
for dur, n, grid, m in sorted(rows, reverse=True)[:int(sys.argv[4]) if len(sys.argv) > 4 else 25]:
    extra = "  ".join(f"{k}={v:.4g}" for k, v in m.items() if k != "dur")
    print(f"{n[:60]:62s} grid={grid:<9d} {dur/1e3:8.1f} us  {extra}")
