"""Per-kernel SQ stall split from one rocprofv3 --pmc pass (counter_collection.csv):
python3 tools/sq_stalls.py <counter_collection.csv>  ->  per kernel name: share of wave cycles parked (s_waitcnt / barrier), stalled at
issue, issuing; MFMA-busy cycles per SIMD-cycle.  Counters: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES (guide: MI355X_MICROARCH.md, rocprofv3 PMC slots)."""
import collections
import csv
import re
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(float))
count = collections.Counter()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
        rows[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            count[name] += 1
tot = sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))
print(f"{'kernel':58s} {'n':>4s} {'parked':>7s} {'istall':>7s} {'(lds)':>6s} {'issue':>6s} {'mfma/busy':>9s}")
for name, c in tot[:24]:
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{name[:58]:58s} {count[name]:4d} {c.get('SQ_WAIT_ANY', 0) / wc:7.2f} {c.get('SQ_WAIT_INST_ANY', 0) / wc:7.2f} "
          f"{c.get('SQ_WAIT_INST_LDS', 0) / wc:6.2f} {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:6.2f} "
          f"{c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(1.0, c.get('SQ_BUSY_CYCLES', 0)):9.3f}")
