#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel-trace CSV over a time window (last `frac` of the run): per-kernel totals and GPU busy %."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t_end = int(rows[-1]["End_Timestamp"]); t_beg = int(rows[0]["Start_Timestamp"])
cut = t_end - int((t_end - t_beg) * frac)
sel = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
wall = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
    a = agg[n]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
busy = sum(v[1] for v in agg.values())
print(f"window {wall/1e3:.2f} ms, {len(sel)} kernels, GPU busy {busy/1e3:.2f} ms ({100*busy/wall:.1f} %)")
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"  {n[:80]:82s} x{c:<5d} {d/1e3:8.3f} ms  avg {d/c:8.1f} us")
