#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid size): count, avg/min/max duration (us).
The same kernel symbol runs on every level, so the grid size separates the fine level from the rest.
usage: summarize_trace.py <kernel_trace.csv> [out.md]"""
import sys
import pandas as pd

df = pd.read_csv(sys.argv[1])
df["dur_us"] = (df.End_Timestamp - df.Start_Timestamp) / 1e3
df["kernel"] = df.Kernel_Name.str.replace(r"\(.*", "", regex=True).str.replace("void ", "")
g = df.groupby(["kernel", "Grid_Size_X"]).dur_us.agg(["count", "mean", "min", "max", "sum"]).reset_index()
g = g.sort_values("sum", ascending=False)
tot = g["sum"].sum()
lines = ["| kernel | grid (threads) | launches | avg us | min us | max us | total ms | share |", "|---|---|---|---|---|---|---|---|"]
for _, r in g.iterrows():
    lines.append(f"| {r.kernel} | {int(r.Grid_Size_X)} | {int(r['count'])} | {r['mean']:.2f} | {r['min']:.2f} | {r['max']:.2f} | "
                 f"{r['sum'] / 1e3:.3f} | {r['sum'] / tot:.3f} |")
out = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
else:
    print(out)
