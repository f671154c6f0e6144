#!/usr/bin/env python3
"""Average a rocprofv3 --pmc counter per (kernel, grid).  usage: summarize_pmc.py <counter_collection.csv>"""
import sys
import pandas as pd
df = pd.read_csv(sys.argv[1])
df["kernel"] = df.Kernel_Name.str.replace(r"\(.*", "", regex=True).str.replace("void ", "")
g = df.groupby(["kernel", "Grid_Size", "Counter_Name"]).Counter_Value.agg(["count", "mean"]).reset_index()
print(g.to_string())
