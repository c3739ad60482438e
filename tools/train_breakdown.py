"""Steady-state kernel breakdown of `bench.py --mode train` from a rocprofv3 kernel trace (the last ~10 graph replays).
usage: python tools/train_breakdown.py <kernel_trace.csv> <ms_per_step>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
step_ns = float(sys.argv[2]) * 1e6
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
end = max(r["e"] for r in rows)
win = [r for r in rows if r["s"] > end - 10 * step_ns]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in win:
    k = r["Kernel_Name"][:110]
    agg[k][0] += 1
    agg[k][1] += (r["e"] - r["s"]) / 1e3
tot = sum(v[1] for v in agg.values())
n = max([v[0] for k, v in agg.items() if "pack_conv_weights_batched" in k] + [1])      # launched once per step
print(f"window of {n} steps: busy {tot / n:.0f} us/step, {sum(v[0] for v in agg.values()) / n:.0f} kernels/step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:50]:
    print(f"{v[1] / n:8.1f} us/step {v[0] / n:6.1f} calls {v[1] / v[0]:7.1f} avg  {k}")
