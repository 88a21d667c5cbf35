#!/usr/bin/env python3
"""Kernel sequence of a rocprofv3 kernel trace: tools/trace_sequence.py <kernel_trace.csv> <first> <count>  (launch order, name, us, grid)"""
import re, sys
import pandas as pd
df = pd.read_csv(sys.argv[1]).sort_values("Start_Timestamp").reset_index(drop=True)
a, n = int(sys.argv[2]), int(sys.argv[3])
def short(x):
    x = re.sub(r"\(anonymous namespace\)::", "", x); x = re.sub(r"^void ", "", x); return x.split("(")[0][:60]
t0 = df.Start_Timestamp.iloc[a]
for i in range(a, min(a + n, len(df))):
    r = df.iloc[i]
    print(f"{i:6d} q{r.Queue_Id} +{(r.Start_Timestamp - t0) / 1e3:9.1f} {(r.End_Timestamp - r.Start_Timestamp) / 1e3:7.1f} us  {short(r.Kernel_Name):60s} grid {r.Grid_Size_X // r.Workgroup_Size_X}x{r.Grid_Size_Y}x{r.Grid_Size_Z}")
