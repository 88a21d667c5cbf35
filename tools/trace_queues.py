#!/usr/bin/env python3
"""Per-queue picture of the LAST n_steps steps of a rocprofv3 kernel trace of any runner: busy time and gaps of every hardware queue, the
main queue's largest kernel classes, and which kernels of the other queues run while the main queue is idle.
    tools/trace_queues.py <kernel_trace.csv> <marker kernel substring> [n_steps=4]
The marker is a kernel that runs once per step (e.g. k_ema for the DiT runners); the window is [marker[-n-1], marker[-1]]."""
import re
import sys
import pandas as pd

df = pd.read_csv(sys.argv[1]).sort_values("Start_Timestamp").reset_index(drop=True)
marker, n = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 4


def short(s):
    s = re.sub(r"\(anonymous namespace\)::", "", s)
    s = re.sub(r"^void ", "", s)
    return s.split("(")[0][:60]


df["n"] = df.Kernel_Name.map(short)
df["dur"] = (df.End_Timestamp - df.Start_Timestamp) / 1e3
mk = df[df.n.str.contains(marker)]
t0, t1 = mk.iloc[-n - 1].End_Timestamp, mk.iloc[-1].End_Timestamp
w = df[(df.Start_Timestamp >= t0) & (df.End_Timestamp <= t1)]
print(f"window: {n} steps, {(t1 - t0) / 1e6 / n:.3f} ms per step, {len(w) / n:.0f} kernels per step")
mainq = w.groupby("Queue_Id").dur.sum().idxmax()
for q, g in w.groupby("Queue_Id"):
    g = g.sort_values("Start_Timestamp")
    gaps = (g.Start_Timestamp.values[1:] - g.End_Timestamp.values[:-1]) / 1e3
    print(f"queue {q}{'*' if q == mainq else ' '}: {len(g) / n:6.0f} kernels / step, busy {g.dur.sum() / 1e3 / n:7.3f} ms / step, "
          f"gaps > 3 us: {(gaps > 3).sum() / n:5.0f} / step totalling {gaps[gaps > 3].sum() / 1e3 / n:6.3f} ms / step")
m = w[w.Queue_Id == mainq]
print("\nmain queue, by kernel class (ms per step, launches per step, mean us):")
t = m.groupby("n").dur.agg(["sum", "count", "mean"]).sort_values("sum", ascending=False).head(22)
for k, r in t.iterrows():
    print(f"  {r['sum'] / 1e3 / n:7.3f}  {r['count'] / n:6.1f}  {r['mean']:7.1f}  {k}")
o = w[w.Queue_Id != mainq]
if len(o):
    print("\nother queues, by kernel class (ms per step, launches per step, mean us):")
    t = o.groupby("n").dur.agg(["sum", "count", "mean"]).sort_values("sum", ascending=False).head(10)
    for k, r in t.iterrows():
        print(f"  {r['sum'] / 1e3 / n:7.3f}  {r['count'] / n:6.1f}  {r['mean']:7.1f}  {k}")
