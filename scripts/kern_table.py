"""Print the per-kernel table of a bench.py JSON line.  usage: python scripts/kern_table.py file.json [prefix]"""
import json, sys
d = json.load(open(sys.argv[1]))
pre = sys.argv[2] if len(sys.argv) > 2 else ""
print("ms_per_step", d["ms_per_step"], "regions", d.get("timed_regions_ms_per_step"))
for k, v in d["roofline"]["kernels"].items():
    if k.startswith(pre):
        print(f"  {k:28s} {v['avg_ms']*1e3:8.1f} us x {v['launches_per_step']:4.2f}  moved {v['moved_MB']:8.1f} MB  frac {v['frac']:.3f}")
