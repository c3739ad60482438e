#!/usr/bin/env python3
"""HBM-side traffic per kernel instantiation from two rocprofv3 PMC passes of bench.py (FETCH_SIZE and WRITE_SIZE are
collected in separate passes: together they do not fit the TCC counter slots).

    python tools/pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv [out.json [workload]]

Only the dispatches of the last complete SDE step (between the last two em_update launches) are used, i.e. the tuned
steady-state kernels, not the autotuner's candidates.  gfx950 correction (MI355X_MICROARCH.md, HBM section):
FETCH_SIZE counts 128-byte read requests as 64 bytes for wide coalesced loads, so bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB.
"""
import collections
import csv
import json
import re
import sys


def last_step(path, counter):
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "v": 0.0})
        d["v"] += float(r["Counter_Value"])
    ids = sorted(disp)
    ends = [i for i in ids if "em_update_kernel" in disp[i]["name"]]
    lo, hi = ends[-2], ends[-1]
    return [(disp[i]["name"], disp[i]["v"]) for i in ids if lo < i <= hi]


def clean(n):
    return re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0].replace("void ", "")


def main():
    fetch = last_step(sys.argv[1], "FETCH_SIZE")
    write = last_step(sys.argv[2], "WRITE_SIZE")
    assert [n for n, _ in fetch] == [n for n, _ in write], "the two passes ran different kernel sequences"
    per = collections.OrderedDict()
    for (n, f), (_, w) in zip(fetch, write):
        k = per.setdefault(clean(n), {"launches": 0, "fetch_kib": 0.0, "write_kib": 0.0})
        k["launches"] += 1
        k["fetch_kib"] += f
        k["write_kib"] += w
    out = {"unit": "bytes per launch (average over the launches of one SDE step)", "correction": "2*FETCH_SIZE + WRITE_SIZE, KiB",
           "kernels": {}}
    tot = 0.0
    for n, k in sorted(per.items(), key=lambda kv: -(2 * kv[1]["fetch_kib"] + kv[1]["write_kib"])):
        b = (2 * k["fetch_kib"] + k["write_kib"]) * 1024
        tot += b
        out["kernels"][n] = {"launches": k["launches"], "bytes_per_launch": b / k["launches"],
                             "read_bytes_per_launch": 2 * k["fetch_kib"] * 1024 / k["launches"],
                             "write_bytes_per_launch": k["write_kib"] * 1024 / k["launches"]}
        print(f"{n[:64]:66s} x{k['launches']:<3d} read {2 * k['fetch_kib'] / 1024 / k['launches']:9.2f} MiB  write "
              f"{k['write_kib'] / 1024 / k['launches']:9.2f} MiB per launch")
    out["bytes_per_step"] = tot
    print(f"total per SDE step: {tot / 1e6:.1f} MB")
    if len(sys.argv) > 3:
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        out["source_hash"] = bench.kernel_source_hash()          # bench.py refuses this file once the kernel sources change
        if len(sys.argv) > 4:
            out["workload"] = sys.argv[4]
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
