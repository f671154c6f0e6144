#!/usr/bin/env python3
"""GPU idle time between the kernels of a rocprofv3 --kernel-trace CSV: per gap class (which kernel follows the gap).
usage: gap_analysis.py <kernel_trace.csv> [min_gap_us]"""
import sys
import pandas as pd
df = pd.read_csv(sys.argv[1]).sort_values("Start_Timestamp")
df["kernel"] = df.Kernel_Name.str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.replace("mgk::", "")
st, en, nm = df.Start_Timestamp.values, df.End_Timestamp.values, df.kernel.values
gaps = {}
busy = 0.0
for i in range(len(df)):
    busy += (en[i] - st[i]) / 1e3
    if i:
        g = (st[i] - en[i - 1]) / 1e3
        if g < 200.0:                       # (longer: between solves / setup)
            k = (nm[i - 1][:34], nm[i][:34])
            a = gaps.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += g
tot = sum(v[1] for v in gaps.values())
print(f"kernels {len(df)}, busy {busy / 1e3:.3f} ms, idle in gaps < 200 us: {tot / 1e3:.3f} ms")
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {v[1] / v[0]:7.2f} us avg x {v[0]:5d} = {v[1] / 1e3:7.3f} ms   after {k[0]:36s} before {k[1]}")
