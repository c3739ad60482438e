"""Steady-state kernel breakdown of `bench.py --mode train` from a rocprofv3 kernel trace: the last 10 complete graph replays,
delimited by the once-per-step batched weight-pack launch.
usage: python tools/train_breakdown.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
marks = [r["s"] for r in rows if "pack_conv_weights_batched" in r["Kernel_Name"]]
n = min(10, len(marks) - 1)
t0, t1 = marks[-n - 1], marks[-1]
win = [r for r in rows if t0 <= r["s"] < t1]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in win:
    k = r["Kernel_Name"][:110]
    agg[k][0] += 1
    agg[k][1] += (r["e"] - r["s"]) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"{n} steps: {(t1 - t0) / n / 1e3:.0f} us wall per step, {tot / n:.0f} us summed kernel time, {sum(v[0] for v in agg.values()) / n:.0f} kernels per step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:50]:
    print(f"{v[1] / n:8.1f} us/step {v[0] / n:6.1f} calls {v[1] / v[0]:7.1f} avg  {k}")
