"""Effective shader clock per kernel: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (MI355X_MICROARCH.md, DVFS give-back).
python3 tools/eff_clock.py <counter_collection.csv> <kernel_trace.csv>"""
import collections, csv, re, sys
clean = lambda n: re.sub(r"\(anonymous namespace\)::|void ", "", n).split("(")[0]
dur = {}
with open(sys.argv[2]) as f:
    for r in csv.DictReader(f):
        dur[r["Dispatch_Id"]] = (clean(r["Kernel_Name"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur:
            continue
        name, ns = dur[r["Dispatch_Id"]]
        a = agg[name]
        a[0] += float(r["Counter_Value"]); a[1] += ns; a[2] += 1
for name, (cyc, ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{name[:56]:56s} n={n:4d} avg {ns / n / 1e3:8.1f} us   effective clock {cyc / 8 / ns:5.2f} GHz")
