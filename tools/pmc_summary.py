#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter tables into profiles/rN/pmc_hbm_traffic.json:
per (kernel, counter): dispatches, mean and max of the per-dispatch value (KB for FETCH_SIZE / WRITE_SIZE)."""
import csv, glob, json, sys
from collections import defaultdict


def main(dirs, out):
    acc = defaultdict(list)
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            per_dispatch = defaultdict(float)
            for r in csv.DictReader(open(f)):
                name = r.get("Kernel_Name") or r.get("Kernel Name")
                cname = r.get("Counter_Name") or r.get("Counter Name")
                val = float(r.get("Counter_Value") or r.get("Counter Value") or 0)
                disp = r.get("Dispatch_Id") or r.get("Dispatch Id") or r.get("Correlation_Id")
                per_dispatch[(name, cname, disp)] += val      # one row per XCD / dimension instance: sum them
            for (name, cname, _), v in per_dispatch.items():
                acc[(name, cname)].append(v)
    rows = [{"kernel": k, "counter": c, "dispatches": len(v), "mean_KB": sum(v) / len(v), "max_KB": max(v)}
            for (k, c), v in acc.items()]
    rows.sort(key=lambda r: (r["counter"], r["kernel"]))
    json.dump(rows, open(out, "w"), indent=1)
    for r in rows:
        if "score_groups" in r["kernel"] or "refit_mask" in r["kernel"]:
            print("%-12s %-70s n=%3d mean %.1f KB" % (r["counter"], r["kernel"][:70], r["dispatches"], r["mean_KB"]))


if __name__ == "__main__":
    main(sys.argv[2:], sys.argv[1])
