"""strong_ceiling_vs_n1 from the bench lines of scripts/round5_evidence.sh part 3 (one box, one job):
   T1(512^3 on one GPU) / T(one GPU's share of the 8-GPU run).  Two figures:
   * box_world1: the 257^3 box through the sharded path at a world of one (no ghost layers: what VERDICT r4 asked for);
   * dry_rank:   rank 7 of 8 alone on the GPU in the ghost-layer form - 269^3 extended box, exchanges pack / unpack without
                 travelling: the ceiling WITH the redundant ghost rows, communication free.
usage: python3 scripts/strong_ceiling.py <dir>"""
import json
import os
import sys


def line(path):
    try:
        return json.loads(open(path).read().strip().splitlines()[-1])
    except Exception:
        return None


d = sys.argv[1]
t1 = line(os.path.join(d, "c2_512_bench.json"))
w1 = line(os.path.join(d, "c2_sharded_w1_ghost.json"))
out = {"T1_512_ms_per_step": t1 and t1["ms_per_step"], "box_world1_ms_per_step": w1 and w1["ms_per_step"]}
if t1 and w1:
    out["strong_ceiling_vs_n1"] = round(t1["ms_per_step"] / w1["ms_per_step"], 3)
    out["sharded_over_single_same_job"] = w1.get("same_job_single_gpu", {}).get("sharded_over_single")
for tag, f in (("rank0_of_8", "c4_dry_rank0_of_8.json"), ("rank7_of_8", "c4_dry_rank7_of_8.json"), ("rank1_of_2", "c4_dry_rank1_of_2.json"),
               ("rank3_of_4", "c4_dry_rank3_of_4.json")):
    r = line(os.path.join(d, f))
    if r:
        out[f"dry_{tag}_ms_per_step"] = r["ms_per_step"]
        out[f"dry_{tag}_extended_over_owned"] = r["ghost"]["extended_over_owned_rows_fine"]
        out[f"dry_{tag}_exchanges_per_step"] = r["ghost"]["exchanges_per_step"]
r7 = line(os.path.join(d, "c4_dry_rank7_of_8.json"))
if t1 and r7:
    out["strong_ceiling_with_ghost_rows_free_communication"] = round(t1["ms_per_step"] / r7["ms_per_step"], 3)
print(json.dumps(out, indent=1))
