#!/usr/bin/env python3
"""Which launches of a step are NOT this library's kernels (torch glue, runtime copies), where they sit and what they cost:
tools/trace_glue.py <kernel_trace.csv>   -- last step of a bench.py trace (steps are delimited by k_guard_finite)."""
import re, sys
import pandas as pd
df = pd.read_csv(sys.argv[1]).sort_values("Start_Timestamp").reset_index(drop=True)
def short(x):
    x = re.sub(r"\(anonymous namespace\)::", "", x); x = re.sub(r"^void ", "", x); return x.split("(")[0][:70]
df["n"] = df.Kernel_Name.map(short)
ends = df.index[df.n.str.contains("k_guard_finite")].tolist()
a, b = (ends[-2] + 1, ends[-1] + 1) if len(ends) >= 2 else (0, len(df))
st = df.iloc[a:b]
ours = st.n.str.startswith("k_") | st.n.str.contains("k_gemm|k_attn|k_row|k_ln|k_adam|k_masked|k_sumsq|k_reduce|k_colsum|k_cast|k_cond|k_guard|k_patch|k_unpatch|k_dit_loss|k_q_sample|k_timestep|k_silu|k_gated|k_clip|k_ema|k_label")
g = st[~ours]
print(f"step: {len(st)} launches, {len(g)} not ours, {((g.End_Timestamp - g.Start_Timestamp).sum()) / 1e3:.1f} us of kernel time")
print(g.groupby("n").agg(count=("n", "size"), us=("End_Timestamp", lambda s: 0)).head(0))
agg = g.assign(us=(g.End_Timestamp - g.Start_Timestamp) / 1e3).groupby("n").us.agg(["count", "sum", "mean"]).sort_values("sum", ascending=False)
print(agg.to_string())
prev = None
print("\nin order (index in step, queue, name, us, previous library kernel):")
for i, (_, r) in enumerate(st.iterrows()):
    if r.n in g.n.values and not (r.n.startswith("k_")):
        print(f"{i:5d} q{r.Queue_Id} {r.n:70s} {(r.End_Timestamp - r.Start_Timestamp) / 1e3:6.1f}  after {prev}")
    else:
        prev = r.n
