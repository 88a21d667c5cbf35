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
# step boundary: the EMA of the frozen pos_embed (k_ema), the last launch of an iteration; stage boundary: the rank-(batch) AdamW of the
# adaLN matrix (k_adam_lowrank: one per stage) or, in traces of earlier rounds, the single k_masked_clip_adam launch per stage
ema = df[df.n.str.contains('k_ema')]
if len(ema) >= 2:
    t0, t1 = ema.iloc[-2].End_Timestamp, ema.iloc[-1].End_Timestamp
    stage = df[df.n.str.contains('k_adam_lowrank') & (df.Start_Timestamp >= t0) & (df.End_Timestamp <= t1)]
    adam = stage if len(stage) >= 2 else df[df.n.str.contains('k_masked_clip_adam') & (df.Start_Timestamp >= t0) & (df.End_Timestamp <= t1)]
else:
    adam = df[df.n.str.contains('k_masked_clip_adam')]
    t0, t1 = adam.iloc[-3].End_Timestamp, adam.iloc[-1].End_Timestamp
print("step ms", (t1 - t0) / 1e6)
st = df[(df.Start_Timestamp >= t0) & (df.End_Timestamp <= t1)].copy()
st['s'] = (st.Start_Timestamp - t0) / 1e3; st['e'] = (st.End_Timestamp - t0) / 1e3
qs = sorted(st.Queue_Id.unique()); mainq = st.Queue_Id.value_counts().idxmax()
for q in qs:
    x = st[st.Queue_Id == q]
    print("queue", q, "kernels", len(x), "busy ms %.2f" % (x.dur.sum() / 1e3), "span %.2f..%.2f" % (x.s.min() / 1e3, x.e.max() / 1e3))
if len(ema) >= 2:
    # end of the forget-stage sweep on the main stream: the last flat AdamW launch that follows the stage's rank-(batch) launch there
    after = st[(st.Queue_Id == mainq) & st.n.str.contains('k_masked_clip_adam') & (st.Start_Timestamp > adam.iloc[0].End_Timestamp) &
               (st.End_Timestamp < adam.iloc[-1].Start_Timestamp)]
    mid = after.iloc[0] if len(after) else adam.iloc[0]
else:
    mid = adam.iloc[-2]
rem = st[st.Start_Timestamp > mid.End_Timestamp]
others = rem[rem.Queue_Id != mainq]
sideq = others.Queue_Id.value_counts().idxmax()               # the weight-gradient stream (a third queue carries the block sweeps)
side = rem[rem.Queue_Id == sideq]; main = rem[rem.Queue_Id == mainq]
for q in sorted(others.Queue_Id.unique()):
    if q != sideq:
        x = rem[rem.Queue_Id == q]
        print("sweep stream q%d: %d kernels beside the forward pass, busy %.2f ms, span %.2f..%.2f ms after the stage boundary" % (
            q, len(x), x.dur.sum() / 1e3, (x.s.min() - main.s.min()) / 1e3, (x.e.max() - main.s.min()) / 1e3))
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
