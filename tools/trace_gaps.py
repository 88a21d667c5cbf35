#!/usr/bin/env python3
"""Where a single-stream kernel chain spends its time: kernel durations against the GAPS between the end of one kernel and the start of the
next on the same queue, from a rocprofv3 kernel trace.  tools/trace_gaps.py <kernel_trace.csv> [skip_fraction=0.4]
Prints totals for the busiest queue over the last (1 - skip_fraction) of the trace, the gap histogram, and per kernel class: count, mean
duration, mean gap BEHIND it (the gap is charged to the kernel that just ended: its drain + write-back + the next dispatch)."""
import re
import sys

import pandas as pd

df = pd.read_csv(sys.argv[1]).sort_values("Start_Timestamp").reset_index(drop=True)
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:64]


q = df.Queue_Id.value_counts().idxmax()
d = df[df.Queue_Id == q].reset_index(drop=True)
d = d.iloc[int(len(d) * skip):].reset_index(drop=True)
d["n"] = d.Kernel_Name.map(short)
d["dur"] = (d.End_Timestamp - d.Start_Timestamp) / 1e3
d["gap"] = (d.Start_Timestamp.shift(-1) - d.End_Timestamp) / 1e3
d = d.iloc[:-1]
span = (d.End_Timestamp.iloc[-1] - d.Start_Timestamp.iloc[0]) / 1e3
g = d.gap.clip(lower=0)
print(f"queue {q}: {len(d)} kernels over {span / 1e3:.2f} ms; kernel time {d.dur.sum() / 1e3:.2f} ms ({100 * d.dur.sum() / span:.1f} %), gaps {g.sum() / 1e3:.2f} ms "
      f"({100 * g.sum() / span:.1f} %); mean kernel {d.dur.mean():.1f} us, mean gap {g.mean():.2f} us, median gap {g.median():.2f} us")
for lo, hi in ((0, 1), (1, 2), (2, 3), (3, 5), (5, 8), (8, 15), (15, 50), (50, 1e9)):
    m = (g >= lo) & (g < hi)
    print(f"   gaps in [{lo}, {hi if hi < 1e8 else 'inf'}) us: {int(m.sum()):6d}  total {g[m].sum() / 1e3:7.2f} ms")
t = d.groupby("n").agg(count=("dur", "size"), dur=("dur", "mean"), gap=("gap", lambda x: x.clip(lower=0).mean()), tot=("dur", "sum"),
                       gtot=("gap", lambda x: x.clip(lower=0).sum()))
t["both"] = t.tot + t.gtot
t = t.sort_values("both", ascending=False)
print(f"{'kernel':64s} {'count':>6s} {'mean us':>8s} {'gap us':>7s} {'kernel ms':>10s} {'gap ms':>8s}")
for n, r in t.head(40).iterrows():
    print(f"{n:64s} {int(r['count']):6d} {r.dur:8.1f} {r.gap:7.2f} {r.tot / 1e3:10.2f} {r.gtot / 1e3:8.2f}")
