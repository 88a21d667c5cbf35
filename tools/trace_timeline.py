#!/usr/bin/env python3
"""Two-stream timeline of the last SFR-on step in a rocprofv3 kernel trace: tools/trace_timeline.py <kernel_trace.csv> [n_print]"""
import sys, re
import pandas as pd, numpy as np
df = pd.read_csv(sys.argv[1]).sort_values('Start_Timestamp').reset_index(drop=True)
npr = int(sys.argv[2]) if len(sys.argv) > 2 else 40
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n); return n.split('(')[0][:44]
df['n'] = df.Kernel_Name.map(short)
df['dur'] = (df.End_Timestamp - df.Start_Timestamp) / 1e3
adam = df[df.n.str.contains('k_masked_clip_adam')]
t0 = adam.iloc[-3].End_Timestamp; t1 = adam.iloc[-1].End_Timestamp
print("step ms", (t1 - t0) / 1e6)
st = df[(df.Start_Timestamp >= t0) & (df.End_Timestamp <= t1)].copy()
st['s'] = (st.Start_Timestamp - t0) / 1e3; st['e'] = (st.End_Timestamp - t0) / 1e3
qs = sorted(st.Queue_Id.unique()); mainq = st.Queue_Id.value_counts().idxmax()
for q in qs:
    x = st[st.Queue_Id == q]
    print("queue", q, "kernels", len(x), "busy ms %.2f" % (x.dur.sum() / 1e3), "span %.2f..%.2f" % (x.s.min() / 1e3, x.e.max() / 1e3))
mid = adam.iloc[-2]
rem = st[st.Start_Timestamp > mid.End_Timestamp]
side = rem[rem.Queue_Id != mainq]; main = rem[rem.Queue_Id == mainq]
print("remain pass: fwd %.2f ms, bwd window %.2f ms (side busy %.2f, main busy %.2f)" % (
    (side.s.min() - main.s.min()) / 1e3, (side.e.max() - side.s.min()) / 1e3, side.dur.sum() / 1e3,
    main[(main.s >= side.s.min()) & (main.e <= side.e.max())].dur.sum() / 1e3))
mm = main[main.s >= side.s.min()].sort_values('s'); g = mm.s.values[1:] - mm.e.values[:-1]
print("main gaps in bwd: total %.2f ms, >20us: %d" % (g[g > 0].sum() / 1e3, (g > 20).sum()))
ss = side.sort_values('s'); g = ss.s.values[1:] - ss.e.values[:-1]
print("side gaps in bwd: total %.2f ms, >20us: %d" % (g[g > 0].sum() / 1e3, (g > 20).sum()))
print("join: side last end %.1f, main last end before %.1f" % (side.e.max(), main[main.e <= side.e.max()].e.max()))
print(rem.groupby('n').dur.agg(['mean', 'count', 'sum']).sort_values('sum', ascending=False).head(22))
c = (side.s.min() + side.e.max()) / 2
w = rem[(rem.s >= c)].head(npr)
for _, r in w.iterrows():
    print(f"q{r.Queue_Id} {r.s-c:8.1f} {r.e-c:8.1f} {r.dur:7.1f} {r.n}  grid {r.Grid_Size_X//r.Workgroup_Size_X}x{r.Grid_Size_Y}")
