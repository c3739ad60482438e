import collections, csv, re, sys
rows = collections.defaultdict(lambda: collections.defaultdict(float))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
        rows[name][r["Counter_Name"]] += float(r["Counter_Value"])
names = sorted(rows, key=lambda n: -rows[n].get("SQ_BUSY_CYCLES", 0))[:16]
ctrs = sorted({c for n in names for c in rows[n]})
print("kernel".ljust(50), " ".join(c.replace("SQ_", "")[:14].rjust(14) for c in ctrs))
for n in names:
    b = rows[n].get("SQ_BUSY_CYCLES", 1)
    print(n[:50].ljust(50), " ".join(f"{rows[n].get(c, 0) / b:14.3f}" for c in ctrs))
