#!/usr/bin/env python3
"""The launches of ONE solveMG step in stream order, from a rocprofv3 --kernel-trace CSV of scripts/solve_probe.py: kernel, grid, duration,
gap in front of it.  usage: step_sequence.py <kernel_trace.csv>   (prints the step between the 10th and 11th four-stage launch)"""
import sys
import pandas as pd
df = pd.read_csv(sys.argv[1]).sort_values("Start_Timestamp").reset_index(drop=True)
df["kernel"] = df.Kernel_Name.str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.replace("mgk::", "")
idx = [i for i, k in enumerate(df.kernel) if "march4" in k]
if len(idx) < 12:
    sys.exit("fewer than 12 four-stage launches in the trace")
a, b = idx[10], idx[11]
prev_end = df.End_Timestamp[a - 1]
tot = 0.0
for i in range(a, b):
    d = (df.End_Timestamp[i] - df.Start_Timestamp[i]) / 1e3
    g = (df.Start_Timestamp[i] - prev_end) / 1e3
    prev_end = df.End_Timestamp[i]
    tot += d + max(g, 0.0)
    gs = df.Grid_Size[i] if "Grid_Size" in df else (df.Grid_Size_X[i] if "Grid_Size_X" in df else 0)
    print(f"{d:8.2f} us  gap {g:6.2f}  grid {gs:>9}  {df.kernel[i][:90]}")
print(f"step: {b - a} launches, {tot:.1f} us")
