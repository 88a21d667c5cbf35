#!/usr/bin/env python3
"""Every kernel around the two stage boundaries of the last SFR-on step in a rocprofv3 kernel trace (what sits between the last backward
kernel of a stage and the first forward GEMM of the next): tools/trace_boundary.py <kernel_trace.csv>"""
import sys, re
import pandas as pd
df = pd.read_csv(sys.argv[1]).sort_values('Start_Timestamp').reset_index(drop=True)
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n); return n.split('(')[0][:52]
df['n'] = df.Kernel_Name.map(short)
df['dur'] = (df.End_Timestamp - df.Start_Timestamp) / 1e3
ema = df[df.n.str.contains('k_ema')]
t0, t1 = ema.iloc[-2].End_Timestamp, ema.iloc[-1].End_Timestamp
st = df[(df.Start_Timestamp >= t0 - 3e6) & (df.End_Timestamp <= t1 + 1e6)].copy()
mainq = st.Queue_Id.value_counts().idxmax()
low = st[st.n.str.contains('k_adam_lowrank') & (st.Start_Timestamp >= t0 - 3e6)]
for k, (_, a) in enumerate(low.iterrows()):
    # window: from the last attention-backward kernel before this launch to the first attention-forward kernel after it
    before = st[st.n.str.contains('k_attn_bwd') & (st.End_Timestamp < a.Start_Timestamp)]
    after = st[st.n.str.contains('k_attn_fwd') & (st.Start_Timestamp > a.End_Timestamp)]
    if not len(before) or not len(after): continue
    w0, w1 = before.iloc[-1].Start_Timestamp, after.iloc[0].End_Timestamp
    w = st[(st.End_Timestamp >= w0) & (st.Start_Timestamp <= w1)]
    print(f"--- boundary {k}: window {(w1 - w0) / 1e3:.1f} us; main-stream busy {w[w.Queue_Id == mainq].dur.sum():.1f} us")
    for _, r in w.iterrows():
        print(f"q{r.Queue_Id}{'*' if r.Queue_Id == mainq else ' '} {(r.Start_Timestamp - w0) / 1e3:8.1f} {(r.End_Timestamp - w0) / 1e3:8.1f} {r.dur:7.1f} {r.n}  grid {r.Grid_Size_X // r.Workgroup_Size_X}x{r.Grid_Size_Y}")
